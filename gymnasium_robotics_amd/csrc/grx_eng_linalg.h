// grx_eng_linalg.h -- dense symmetric factor / solve: in LDS, and in registers (lane i owns row i, v_readlane broadcasts, MuJoCo's elimination order).
// A FRAGMENT of csrc/grx_engine.h: textually included INSIDE `template <class S> struct GrxEngine { ... }` (every function here is a static member), in the order the engine
// header lists; not a standalone header.  The split is purely textual (round 5): the token stream of the translation units is unchanged.
// ------------------------------------------------------------------------------------------
// K4 dense symmetric solve in LDS:  A x = b, A overwritten.  Elimination runs from the last
// dof to the first (A = L' D L), i.e. leaves of the kinematic tree before the root -- the same
// order MuJoCo's sparse L'DL uses, which keeps the 1e11-damped base dofs out of the pivots of
// everything else.  x is returned in b.
// ------------------------------------------------------------------------------------------
GRX_MEM int grx_sym_factor(float* A, int n, int lane_) {
  int bad = 0;
  for (int k = n - 1; k >= 0; k--) {
    float d = A[k * n + k];
    if (!(d > 1e-30f)) { bad = 1; d = 1e-30f; }
    float rinv = 1.0f / d;
    FOR_LANES {
      int li = lane >> 3, lj = lane & 7;
      for (int i = li; i < k; i += 8) {
        float ti = A[k * n + i] * rinv;
        for (int j = lj; j < k; j += 8) A[i * n + j] -= ti * A[k * n + j];
      }
    }
    WAVE_SYNC();
    LANE0 { A[k * n + k] = rinv; }
  }
  WAVE_SYNC();
  return bad;
}
// A holds the factor from grx_sym_factor: row k = [t_k0 .. t_k,k-1, 1/d_k]
GRX_MEM void grx_sym_solve(const float* A, int n, float* x, int lane_) {
  // L' y = b  (y_i = b_i - sum_{k>i} (t_ki/d_k) y_k)
  for (int k = n - 1; k > 0; k--) {
    float yk = x[k] * A[k * n + k];
    FOR_LANES { for (int i = lane; i < k; i += 64) x[i] -= A[k * n + i] * yk; }
    WAVE_SYNC();
  }
  FOR_LANES { for (int i = lane; i < n; i += 64) x[i] *= A[i * n + i]; }
  WAVE_SYNC();
  // L x = z  (x_k = z_k - sum_{i<k} (t_ki/d_k) x_i)
  for (int i = 0; i < n - 1; i++) {
    float xi = x[i];
    FOR_LANES { for (int k = i + 1 + lane; k < n; k += 64) x[k] -= A[k * n + i] * A[k * n + k] * xi; }
    WAVE_SYNC();
  }
}

// dof_parentid of the Shadow hand's 24 dofs (see kGrxHandAnc below; checked by the host before a hand shape is selected)
#define GRX_HAND_DOF_PARENTS {-1, 0, 1, 2, 3, 4, 1, 6, 7, 8, 1, 10, 11, 12, 1, 14, 15, 16, 17, 1, 19, 20, 21, 22}
// A x = b in one call.  On the GPU, for the dof counts of the models in scope, the whole system is held in
// registers: lane j owns column j of A (lane nv owns b), the pivot column is broadcast with v_readlane and the
// elimination runs from the last dof to the first exactly like grx_sym_factor -- no LDS round trips, no barriers.
#if GRX_ON_DEVICE
// v_readlane_b32 moves raw bits: the builtin is typed (int,int), so floats go through a bit cast
static __device__ __forceinline__ float grx_readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// reciprocal of a strictly positive pivot: v_rcp_f32 (1 ulp) + one Newton step
static __device__ __forceinline__ float grx_rcp_refined(float d) { float r = __builtin_amdgcn_rcpf(d); return fmaf(fmaf(-d, r, 1.0f), r, r); }
// Lane i (< NS) owns ROW i of the symmetric matrix and b_i.  Step k of the elimination broadcasts row k with v_readlane
// (one readlane + one fma per remaining column) and every lane i < k subtracts its multiple of it; rows end up lower
// triangular, pivots final when they are used.  The forward substitution then needs one broadcast per unknown.
// Dof tree of the Shadow hand (shared_asset / robot.xml of the hand models): wrist 0-1, then five chains hanging off dof 1
// (FF 2-5, MF 6-9, RF 10-13, LF 14-18, TH 19-23).  kGrxHandAnc[k] = the ancestor dofs of dof k as a bit mask.  M and M + h B have
// exactly this pattern below the diagonal, and the last-to-first elimination creates no fill-in (the LTDL argument mj_factorM relies
// on), so the HAND variant of the register solve broadcasts only those columns: 83 instead of 276, decided at compile time, the
// skipped updates being exact zeros of the dense elimination.  grx_fill_model_scalars sets m->handtree only if dof_parentid matches.
static constexpr unsigned kGrxHandAnc[24] = {0x0, 0x1, 0x3, 0x7, 0xF, 0x1F, 0x3, 0x43, 0xC3, 0x1C3, 0x3, 0x403, 0xC03, 0x1C03,
                                             0x3, 0x4003, 0xC003, 0x1C003, 0x3C003, 0x3, 0x80003, 0x180003, 0x380003, 0x780003};
template <int NS, bool HAND = false>
  static __device__ __forceinline__ int grx_sym_solve_reg(const float* A, int ld, float* x, int lane_) {
  float a[NS];
  const int row = lane_ < NS ? lane_ : 0;
#pragma unroll
  for (int i = 0; i < NS; i++) a[i] = A[row * ld + i];     // ld: row stride (a diagonal block of a larger matrix can be solved in place)
  float b = x[row], rd = 0.0f;
#pragma unroll
  for (int k = NS - 1; k > 0; k--) {
    const float pinv = grx_rcp_refined(grx_readlane_f(a[k], k));
    rd = (lane_ == k) ? pinv : rd;
    const float mi = (lane_ < k) ? -a[k] * pinv : 0.0f;
#pragma unroll
    for (int j = 0; j < k; j++) { if (!HAND || ((kGrxHandAnc[k % 24] >> j) & 1u)) a[j] = fmaf(mi, grx_readlane_f(a[j], k), a[j]); }
    b = fmaf(mi, grx_readlane_f(b, k), b);
  }
  { const float pinv = grx_rcp_refined(grx_readlane_f(a[0], 0)); rd = (lane_ == 0) ? pinv : rd; }
  float xo = 0.0f;
#pragma unroll
  for (int k = 0; k < NS; k++) {
    const float t = b * rd;                 // lane k: b_k / L_kk = x_k (b_k is final once x_0 .. x_k-1 have been applied)
    const float xk = grx_readlane_f(t, k);
    xo = (lane_ == k) ? t : xo;
    b = fmaf(-a[k], xk, b);                 // lanes i > k: b_i -= L_ik x_k ; lanes <= k are finished, their b is dead
  }
  __syncthreads();
  if (lane_ < NS) x[lane_] = xo;
  __syncthreads();
  return __ballot((lane_ < NS) && !(rd > 0.0f)) != 0ull;   // a non-positive (or NaN) pivot: the matrix was not positive definite (GRX_ST_FACTOR)
}
// Newton Hessian of the hand models, H = M + J' D J (24 hand dofs [+ NOBJ = 6 dofs of a free object]).  M has the tree pattern; a limit / tendon row
// touches dofs of one chain; a contact between a finger and the object couples that finger's chain (and the wrist) with the object dofs -- so, unless
// two DIFFERENT fingers touch each other, H's hand block keeps the tree pattern and everything else sits in the object's rows and columns.
// Eliminating the hand dofs leaf-to-root FIRST and the object dofs LAST creates no fill outside that pattern (a pivot k couples anc(k) and the
// object among themselves: ancestors of one dof form a chain), so a pivot broadcasts |anc(k)| + NOBJ columns instead of all the remaining ones:
// 83 + 24 NOBJ + NOBJ (NOBJ - 1) / 2 = 242 against 435 for the 30 dofs of hand + object; the skipped updates are exact zeros.  LINKED = false:
// no active row links hand and object (the caller looked), the object columns are skipped as well (83 + 15).
// The pattern is CHECKED on the values (every lane scans the off-pattern part of its row: 24 compares); returns -1 without touching x when an
// off-pattern entry is non-zero (finger-finger contact): the caller falls back to the dense elimination.
static constexpr unsigned kGrxHandAncTable[32] = {0x0, 0x1, 0x3, 0x7, 0xF, 0x1F, 0x3, 0x43, 0xC3, 0x1C3, 0x3, 0x403, 0xC03, 0x1C03,
                                                  0x3, 0x4003, 0xC003, 0x1C003, 0x3C003, 0x3, 0x80003, 0x180003, 0x380003, 0x780003,
                                                  0xFFFFFF, 0xFFFFFF, 0xFFFFFF, 0xFFFFFF, 0xFFFFFF, 0xFFFFFF, 0xFFFFFF, 0xFFFFFF};
template <int NOBJ, bool LINKED>
  static __device__ __forceinline__ int grx_sym_solve_hand(const float* A, int ld, float* x, int lane_) {
  constexpr int NH = 24, NS = NH + NOBJ;
  float a[NS];
  const int row = lane_ < NS ? lane_ : 0;
#pragma unroll
  for (int i = 0; i < NS; i++) a[i] = A[row * ld + i];
  {
    const unsigned anc = kGrxHandAncTable[row & 31];
    int off = 0;
#pragma unroll
    for (int j = 0; j < NH - 1; j++) off |= (j < row) && !((anc >> j) & 1u) && (a[j] != 0.0f);
    if (__ballot(off && lane_ < NH) != 0ull) return -1;
  }
  float b = x[row], rd = 0.0f;
  const bool obj = lane_ >= NH;
#pragma unroll
  for (int k = NH - 1; k >= 0; k--) {          // hand pivots, leaf to root; remaining rows: hand dofs < k (only the ancestors hold a non-zero a[k]) and the object
    const float pinv = grx_rcp_refined(grx_readlane_f(a[k], k));
    rd = (lane_ == k) ? pinv : rd;
    const float mi = (lane_ < k || (LINKED && obj)) ? -a[k] * pinv : 0.0f;
#pragma unroll
    for (int j = 0; j < k; j++) { if ((kGrxHandAnc[k] >> j) & 1u) a[j] = fmaf(mi, grx_readlane_f(a[j], k), a[j]); }
    if (LINKED) {
#pragma unroll
      for (int j = NH; j < NS; j++) a[j] = fmaf(mi, grx_readlane_f(a[j], k), a[j]);
    }
    b = fmaf(mi, grx_readlane_f(b, k), b);
  }
#pragma unroll
  for (int k = NS - 1; k >= NH; k--) {         // object pivots: a dense NOBJ x NOBJ block
    const float pinv = grx_rcp_refined(grx_readlane_f(a[k], k));
    rd = (lane_ == k) ? pinv : rd;
    const float mi = (obj && lane_ < k) ? -a[k] * pinv : 0.0f;
#pragma unroll
    for (int j = NH; j < k; j++) a[j] = fmaf(mi, grx_readlane_f(a[j], k), a[j]);
    b = fmaf(mi, grx_readlane_f(b, k), b);
  }
  // substitution in the reverse order of the elimination: object dofs first, then the hand dofs root to leaf
  float xo = 0.0f;
#pragma unroll
  for (int kk = 0; kk < NS; kk++) {
    const int k = kk < NOBJ ? NH + kk : kk - NOBJ;
    const float t = b * rd;
    const float xk = grx_readlane_f(t, k);
    xo = (lane_ == k) ? t : xo;
    if (k >= NH && !LINKED) b = obj ? fmaf(-a[k], xk, b) : b;
    else b = fmaf(-a[k], xk, b);
  }
  __syncthreads();
  if (lane_ < NS) x[lane_] = xo;
  __syncthreads();
  return __ballot((lane_ < NS) && !(rd > 0.0f)) != 0ull;
}
// In-place Gauss-Jordan inverse of a symmetric positive definite matrix, same register layout (lane i = row i, NS registers): step k broadcasts
// row k with v_readlane, every other row subtracts its multiple of it, the pivot column becomes the k-th column of the inverse.  No LDS traffic,
// no barrier: ~2 NS^2 instructions against ~NS^3 / 8 dependent LDS round trips of the factor-and-substitute route (noslip needs all of M^-1).
template <int NS>
  static __device__ __forceinline__ void grx_sym_inverse_reg(const float* A, int ld, float* out, int lane_) {
  float a[NS];
  const int row = lane_ < NS ? lane_ : 0;
#pragma unroll
  for (int i = 0; i < NS; i++) a[i] = A[row * ld + i];
#pragma unroll
  for (int k = 0; k < NS; k++) {
    const float p = grx_rcp_refined(grx_readlane_f(a[k], k));
    const bool own = (lane_ == k);
    const float f = own ? 0.0f : a[k] * p;
#pragma unroll
    for (int j = 0; j < NS; j++) {
      if (j == k) continue;
      const float akj = grx_readlane_f(a[j], k);
      a[j] = own ? akj * p : fmaf(-f, akj, a[j]);
    }
    a[k] = own ? p : -f;
  }
  __syncthreads();
  if (lane_ < NS) {
#pragma unroll
    for (int i = 0; i < NS; i++) out[lane_ * ld + i] = a[i];
  }
  __syncthreads();
}
#endif

// nsplit: the last nsplit (= 6) dofs are a free object whose block of A is decoupled from the rest (all entries between the two
// blocks are exactly zero: always true for M + h B, true for the Hessian while no contact links object and robot).  The two diagonal
// blocks are then solved one after the other -- the same arithmetic as the full elimination, in which every multiplier between the
// blocks is an exact zero, at (nr^2 + 36) / n^2 of its broadcasts (15 + 6 instead of 21: 43 % fewer).
GRX_MEM int grx_sym_solve_full(float* A, int n, float* x, int lane_, int nsplit = 0, int tree = 0) {
#if GRX_ON_DEVICE
  if (S::kFixed && S::NF == 24 && tree) {      // hand shapes (the host matched m->handtree): M / M + h B solves
    int bad_ = grx_sym_solve_reg<24, true>(A, n, x, lane_);
    if (S::NV == 30) bad_ |= grx_sym_solve_reg<6>(A + 24 * n + 24, n, x + 24, lane_);
    return bad_;
  }
  if (S::kFixed && S::NF == 24 && !tree) {     // hand shapes, Newton Hessian: structured elimination while no two fingers touch each other
    int r_;
    if (S::NV == 30) r_ = nsplit == 6 ? grx_sym_solve_hand<6, false>(A, n, x, lane_) : grx_sym_solve_hand<6, true>(A, n, x, lane_);
    else r_ = grx_sym_solve_hand<0, false>(A, n, x, lane_);
    if (r_ >= 0) return r_;
  }
  if (nsplit == 6 && n == 21) { int bad_ = grx_sym_solve_reg<15>(A, n, x, lane_); return bad_ | grx_sym_solve_reg<6>(A + 15 * n + 15, n, x + 15, lane_); }
  if (nsplit == 6 && n == 30) { int bad_ = grx_sym_solve_reg<24>(A, n, x, lane_); return bad_ | grx_sym_solve_reg<6>(A + 24 * n + 24, n, x + 24, lane_); }
  if (n == 21) return grx_sym_solve_reg<21>(A, n, x, lane_);
  if (n == 14) return grx_sym_solve_reg<14>(A, n, x, lane_);
  if (n == 15) return grx_sym_solve_reg<15>(A, n, x, lane_);
  if (n == 24) return grx_sym_solve_reg<24>(A, n, x, lane_);
  if (n == 29) return grx_sym_solve_reg<29>(A, n, x, lane_);
  if (n == 30) return grx_sym_solve_reg<30>(A, n, x, lane_);
  if (n == 33) return grx_sym_solve_reg<33>(A, n, x, lane_);
  if (n == 36) return grx_sym_solve_reg<36>(A, n, x, lane_);
#endif
#if !GRX_ON_DEVICE
  if (n == 21 || n == 14 || n == 15 || n == 24 || n == 29 || n == 30 || n == 33 || n == 36) {   // mirror the device: these sizes are solved without touching A
    static float copy[36 * 36];
    for (int i = 0; i < n * n; i++) copy[i] = A[i];
    int bad_ = grx_sym_factor(copy, n, lane_);
    grx_sym_solve(copy, n, x, lane_);
    return bad_;
  }
#endif
  int bad = grx_sym_factor(A, n, lane_);
  grx_sym_solve(A, n, x, lane_);
  return bad;
}

