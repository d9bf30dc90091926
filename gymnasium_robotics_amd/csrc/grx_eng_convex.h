// grx_eng_convex.h -- K8 narrow phase, general convex pairs: support functions (primitives, hull scans with guesses and candidate lists), Minkowski Portal Refinement, root-body frames in fp64.
// A FRAGMENT of csrc/grx_engine.h: textually included INSIDE `template <class S> struct GrxEngine { ... }` (every function here is a static member), in the order the engine
// header lists; not a standalone header.  The split is purely textual (round 5): the token stream of the translation units is unchanged.
// ------------------------------------------------------------------------------------------
// General convex pairs (ellipsoid / cylinder against sphere, capsule, ellipsoid, cylinder, box): Minkowski Portal Refinement, one
// lane per pair, one contact per pair (what MuJoCo's libccd route produces; see the oracle's header comment on the algorithm and on
// what is not restated).  Everything is computed relative to the centre of geom 1, so the fp32 support points are O(geom size)
// instead of O(world coordinates); a Minkowski point is kept with its witness on geom 1 (the witness on geom 2 is w - v).
// ------------------------------------------------------------------------------------------
// The "is it zero / are they equal" thresholds of the portal routine are part of the ALGORITHM MuJoCo runs (libccd's CCD_EPS, built in double
// precision: 2.2e-16), not a statement about this build's arithmetic: several of the tests compare triple products of portal vertices (scale
// size^3 ~ 1e-5 for centimetre geoms) against it, and with the fp32 machine epsilon (1.2e-7: what rounds 1 to 1 in THIS arithmetic) the routine
// took other branches than the reference's in 20 % of the resting egg contacts -- all of the egg / puck / door discrepancy of round 2 was this
// one constant (the fp64 oracle compiled with 1.2e-7 reproduces the round-2 error table digit for digit; tools/emu_tolerances.py).  The rounding
// noise of fp32 in the same tests only moves decisions that are ties in exact arithmetic.
#ifndef GRX_MPR_EPS
#define GRX_MPR_EPS 2.220446e-16f
#endif
// Arithmetic type of the general convex routine (portal refinement + its support functions): GRX_MPR_REAL.  The routine's branch decisions compare
// triple products of nearly coplanar portal vertices and its final triangle is the size of a resting contact's depth, so fp32 rounding inside it moves the
// contact POINT of a line / face contact by centimetres (tools/emu_trace.py); it runs for a handful of pairs per substep, which is why it can afford fp64.
#ifndef GRX_MPR_REAL
#define GRX_MPR_REAL double
#endif
typedef GRX_MPR_REAL MF;
#ifndef GRX_HULL_REAL
#define GRX_HULL_REAL float
#endif
typedef GRX_HULL_REAL HF;   // arithmetic of the hull support scan (vertex tables are fp32)
#ifndef GRX_TIE_REAL
#define GRX_TIE_REAL double
#endif
typedef GRX_TIE_REAL TF;    // arithmetic that decides between hull vertices whose fp32 projections tie (grx_mesh_support)
GRX_MEM float grx_sqrt(float x) { return sqrtf(x); }
GRX_MEM float grx_fabs(float x) { return fabsf(x); }
GRX_MEM float grx_fmin(float a, float b) { return fminf(a, b); }
GRX_MEM float grx_fmax(float a, float b) { return fmaxf(a, b); }
#if !GRX_REAL_IS_DOUBLE
GRX_MEM double grx_sqrt(double x) { return sqrt(x); }
GRX_MEM double grx_fabs(double x) { return fabs(x); }
GRX_MEM double grx_fmin(double a, double b) { return fmin(a, b); }
GRX_MEM double grx_fmax(double a, double b) { return fmax(a, b); }
#endif
struct GrxMprPt { MF v[3], w[3]; };
GRX_MEM int grx_mpr_zero(MF x) { return grx_fabs(x) < GRX_MPR_EPS; }
GRX_MEM int grx_mpr_eq(MF a, MF b) {
  MF ab = grx_fabs(a - b);
  if (ab < GRX_MPR_EPS) return 1;
  a = grx_fabs(a); b = grx_fabs(b);
  return ab < GRX_MPR_EPS * (b > a ? b : a);
}
GRX_MEM MF grx_sgn1f(MF x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
GRX_MEM void grx_normalize3f(MF* v) { MF n2 = dot3f(v, v); if (n2 > 0.0f) { MF s = 1.0f / grx_sqrt(n2); v[0] *= s; v[1] *= s; v[2] *= s; } }
// farthest point of the geom along the world direction d, relative to the geom centre
template <typename RF>
GRX_MEM void grx_geom_support(const RF* R, const RF* szf, int type, const MF* d, MF* out) {
  MF dl[3], r[3] = {0.0f, 0.0f, 0.0f};
  const MF sz[3] = {szf[0], szf[1], szf[2]};
  mulMatTVec3f(dl, R, d);
  if (type == 2) { r[0] = dl[0] * sz[0]; r[1] = dl[1] * sz[0]; r[2] = dl[2] * sz[0]; }
  else if (type == 3) { r[0] = dl[0] * sz[0]; r[1] = dl[1] * sz[0]; r[2] = dl[2] * sz[0] + grx_sgn1f(dl[2]) * sz[1]; }
  else if (type == 4) {
    MF t[3] = {dl[0] * sz[0], dl[1] * sz[1], dl[2] * sz[2]};
    grx_normalize3f(t);
    r[0] = t[0] * sz[0]; r[1] = t[1] * sz[1]; r[2] = t[2] * sz[2];
  } else if (type == 5) {
    MF h = grx_sqrt(dl[0] * dl[0] + dl[1] * dl[1]);
    if (h > GRX_MINVAL) { MF ih = sz[0] / h; r[0] = dl[0] * ih; r[1] = dl[1] * ih; }
    r[2] = grx_sgn1f(dl[2]) * sz[1];
  } else if (type == 6) { r[0] = grx_sgn1f(dl[0]) * sz[0]; r[1] = grx_sgn1f(dl[1]) * sz[1]; r[2] = grx_sgn1f(dl[2]) * sz[2]; }
  mulMatVec3f(out, R, r);
}
template <typename RF>   // storage of the two frames and sizes: MF where the caller derived them in MF (grx_geom_frame_mf), float where they are the fp32 values of the kinematics stage (hull pairs: half the registers)
struct GrxMprPairT { RF R1[9], R2[9], s1[3], s2[3]; MF c21[3], hm; int t1, t2;   // the two frames are copied into registers: ~20 support evaluations each read them twice
                    const float *v1, *v2; int n1, n2, lane; const int *aadr1, *anum1, *aadr2, *anum2, *adj;   // hull adjacency (per hull vertex: first neighbour / count into adj)
                                        // hull vertices (geom frame) of mesh geoms: only read by the wave-cooperative variant
                    GrxMprPt* pts;
                    const float *nbr1, *nbr2;   // neighbour records of the two hulls (GrxModel::mesh_nbr + 64 * first hull vertex), or null
                    const int *cell1, *cell2; const float* cellrec;   // support-candidate lists of the two hulls (GrxModel::mesh_cellhdr + 2 * geom_cellbase, mesh_cellrec), or null
                    mutable int hint, hk;       // wave-cooperative variant: lane e holds the guessed support vertices of evaluation e ((v1 + 1) | (v2 + 1) << 16); evaluations so far
#if !GRX_ON_DEVICE
                    mutable int hints[16];      // the emulator's stand-in for `hint` (one register per lane on the device: lane e holds word e)
#endif
#if GRX_DEVICE_PROFILE
                    long long* prof;
#endif
                  };                                            // wave-cooperative variant: LDS storage of the five portal points (keeps them out of the VGPR budget)
typedef GrxMprPairT<MF> GrxMprPair;       // lane-per-pair convex routine (primitive pairs)
typedef GrxMprPairT<float> GrxMprPairW;   // wave-cooperative hull pairs
// (Round 4, measured and removed: the scan as a leaf function behind a real call or inline with 16-byte vertex records and 8 - 16 loads in flight per lane -- one memory
// round per hull instead of three -- is 12 % SLOWER on the Fetch launch, profiles/ab_r04_fetch_scan4.txt: the registers it needs are spilled by the substep loop.)
// Convex hull of a mesh: the hull vertex farthest along the (geom-frame) direction dl; the lowest vertex index wins ties, like the oracle's
// exhaustive scan.  Called from wave-uniform code: on the GPU the 64 lanes share the scan (lane l takes the vertices l, l + 64, ...; the
// loads are coalesced) and agree on the winner through two DPP reductions -- a hull of 500 vertices costs 8 loads per lane.
// fp64 support vertex from the fp32 scan's winner: the vertices whose projection is within fp32 rounding of the maximum form a connected cap of the convex
// hull (a face lying flat on a table: all of its vertices tie to ~1e-7), and the reference -- a double precision scan -- picks among them by the digits the
// fp32 products do not have; another pick moves the portal's first vertex and with it the contact normal by 0.1 rad (FetchHullContacts fixture, snapshot 93).
// Hill climbing over the hull's edge graph in MF arithmetic from the fp32 winner reaches the fp64 winner in one or two rounds of neighbour loads; the lowest
// index wins exact ties, like the reference's scan.  aadr / anum: per-vertex adjacency of THIS hull, adj: the model's neighbour table.
GRX_MEM int grx_mesh_support_refine(const float* verts, const int* aadr, const int* anum, const int* adj, const MF* dlm, int cur, int lane_) {
  if (sizeof(TF) == sizeof(HF) || aadr == nullptr) return cur;
  const TF dlt[3] = {(TF)dlm[0], (TF)dlm[1], (TF)dlm[2]};
  for (int guard = 0; guard < 64; guard++) {
    const TF tc = (TF)verts[3 * cur] * dlt[0] + (TF)verts[3 * cur + 1] * dlt[1] + (TF)verts[3 * cur + 2] * dlt[2];
    const int aa = aadr[cur], an = anum[cur];
    TF tb = tc; int nb = cur;
#if !GRX_ON_DEVICE
    (void)lane_;
    for (int k = 0; k < an; k++) {
      const int v = adj[aa + k];
      const TF t = (TF)verts[3 * v] * dlt[0] + (TF)verts[3 * v + 1] * dlt[1] + (TF)verts[3 * v + 2] * dlt[2];
      if (t > tb || (t == tb && v < nb)) { tb = t; nb = v; }
    }
#else
    for (int k = lane_; k < an; k += 64) {
      const int v = adj[aa + k];
      const TF t = (TF)verts[3 * v] * dlt[0] + (TF)verts[3 * v + 1] * dlt[1] + (TF)verts[3 * v + 2] * dlt[2];
      if (t > tb || (t == tb && v < nb)) { tb = t; nb = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {   // wave argmax in MF (rare path: a few times per portal search)
      const TF to = __shfl_xor(tb, o, 64); const int no = __shfl_xor(nb, o, 64);
      if (to > tb || (to == tb && no < nb)) { tb = to; nb = no; }
    }
#endif
    if (nb == cur) break;
    cur = nb;
  }
  return cur;
}
// hint / nbr: a GUESS of the support vertex (the one the same evaluation of the same pair's portal search found in the previous substep) and the hull's neighbour records.
// The guess is accepted only if its projection exceeds that of every hull neighbour by 1e-6 |d| (metres): a clear local maximum over the hull's edge graph is the unique
// global maximum (convexity), so the exhaustive scan below -- fp32 scan, fp64 decision among the near-ties -- returns the same vertex.  The margin is what makes this
// rigorous on REAL hull tables: qhull's triangulation of the float32-rounded vertices contains near-coplanar facets whose diagonals are "concave" at the 1e-9 m level, so
// a vertex can top all of its listed neighbours by up to 7e-9 m without being the maximum (tests/test_cpu_hull_hints.py measures this on every packaged hull: nothing
// above 1e-7 m over 10^5 face-normal, chord and random directions).  One coalesced fetch of 16 records instead of a scan of the whole hull; anything else (a near-tie, a
// vertex with more than 15 neighbours, a stale guess) falls through to the scan.
GRX_MEM int grx_mesh_support(const float* verts, int n, const MF* dlm, MF* r, int lane_, const int* aadr = nullptr, const int* anum = nullptr, const int* adj = nullptr, int hint = -1,
                             const float* nbr = nullptr, const int* cellhdr = nullptr, const float* cellrec = nullptr) {
  r[0] = r[1] = r[2] = 0.0f;
  if (n <= 0) return -1;
#if GRX_DEVICE_HULL_HINTS
  if (hint >= 0 && hint < n && nbr != nullptr) {
    const float4 p = ((const float4*)nbr)[GRX_NBR_RECS * hint + (lane_ & (GRX_NBR_RECS - 1))];
    const int deg = (int)grx_readlane_f(p.w, 0);
    if (deg >= 1) {
      const double t = (double)p.x * (double)dlm[0] + (double)p.y * (double)dlm[1] + (double)p.z * (double)dlm[2];
      const unsigned long long tb = (unsigned long long)__double_as_longlong(t);
      const double t0 = __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(tb >> 32), 0) << 32) | (unsigned)__builtin_amdgcn_readlane((int)tb, 0)));
      const double dn = sqrt((double)dlm[0] * (double)dlm[0] + (double)dlm[1] * (double)dlm[1] + (double)dlm[2] * (double)dlm[2]);
      const bool beaten = lane_ >= 1 && lane_ <= deg && !(t0 - t > 1.0e-6 * dn);
      if (__ballot(beaten) == 0ull) {
        r[0] = grx_readlane_f(p.x, 0); r[1] = grx_readlane_f(p.y, 0); r[2] = grx_readlane_f(p.z, 0);
        return hint;
      }
    }
  }
#elif GRX_TWIN_HULL_HINTS
  // emulator twin of the guess check above (same records, same margin, fp64 projections): the CPU suite then exercises "a stale / foreign / wrong guess never changes a result"
  // and "an accepted guess is the scan's vertex" on every hull fixture (tests/test_cpu_hull_hints.py)
  if (g_grx_emu_hints_on && hint >= 0 && hint < n && nbr != nullptr) {
    const float* p0 = nbr + 4 * GRX_NBR_RECS * (size_t)hint;
    const int deg = (int)p0[3];
    if (deg >= 1) {
      const double dn = sqrt((double)dlm[0] * (double)dlm[0] + (double)dlm[1] * (double)dlm[1] + (double)dlm[2] * (double)dlm[2]);
      const double t0 = (double)p0[0] * (double)dlm[0] + (double)p0[1] * (double)dlm[1] + (double)p0[2] * (double)dlm[2];
      int beaten = 0;
      for (int k = 1; k <= deg; k++) {
        const float* p = p0 + 4 * k;
        const double t = (double)p[0] * (double)dlm[0] + (double)p[1] * (double)dlm[1] + (double)p[2] * (double)dlm[2];
        if (!(t0 - t > 1.0e-6 * dn)) beaten = 1;
      }
      if (!beaten) { r[0] = p0[0]; r[1] = p0[1]; r[2] = p0[2]; g_grx_hint_stats[0]++; return hint; }
    }
    g_grx_hint_stats[1]++;
  }
#else
  (void)hint; (void)nbr;
#endif
  const HF dl[3] = {(HF)dlm[0], (HF)dlm[1], (HF)dlm[2]};   // the scan's own arithmetic type (GRX_HULL_REAL)
#if !GRX_ON_DEVICE
  HF best = -3.0e38f; int bi = 0;
  for (int v = 0; v < n; v++) { const HF t = verts[3 * v] * dl[0] + verts[3 * v + 1] * dl[1] + verts[3 * v + 2] * dl[2]; if (t > best) { best = t; bi = v; } }
  if (cellhdr) {   // the emulator scans the hull; it CHECKS that the device's candidate list of this direction's cell holds every vertex inside the tie band (what the device reads instead)
    const int cell = grx_hull_cell((float)dl[0], (float)dl[1], (float)dl[2]), off = cellhdr[2 * cell], cnt = cellhdr[2 * cell + 1];
    g_grx_cell_stats[0]++;
    if (cnt > 0) {
      g_grx_cell_stats[1]++; g_grx_cell_stats[2] += cnt;
      const HF near_ = best - 1.0e-6f * fmaxf(1.0f, fabsf(best));
      for (int v = 0; v < n; v++) {
        const HF t = verts[3 * v] * dl[0] + verts[3 * v + 1] * dl[1] + verts[3 * v + 2] * dl[2];
        if (t < near_) continue;
        int found = 0;
        for (int k = 0; k < cnt; k++) {
          const float* rec = cellrec + 4 * (size_t)(off + k); int id; memcpy(&id, rec + 3, 4);
          if (id == v) { found = (rec[0] == verts[3 * v] && rec[1] == verts[3 * v + 1] && rec[2] == verts[3 * v + 2]); break; }
        }
        if (!found) g_grx_cell_stats[3]++;   // a vertex the device would not have seen: must stay 0 (tests/test_cpu_hull_cells.py)
      }
    }
  }
#else
#ifndef GRX_HULL_INFLIGHT
#define GRX_HULL_INFLIGHT 4
#endif
  float best = -3.0e38f, second = -3.0e38f, bx = 0.0f, by = 0.0f, bz = 0.0f; int mine = 0;   // second: this lane's runner-up (is the winner unique beyond fp32 rounding?)
  int listed = 0;
  if (cellhdr) {   // the candidate list of the direction's cell: every vertex that can win or tie is in it (GrxModel::mesh_cellhdr), one record per lane
    const int cell = grx_hull_cell(dl[0], dl[1], dl[2]);
    const int off = __builtin_amdgcn_readfirstlane(cellhdr[2 * cell]), cnt = __builtin_amdgcn_readfirstlane(cellhdr[2 * cell + 1]);
    if (cnt > 0) {
      listed = 1;
      if (lane_ < cnt) {
        const float4 p = ((const float4*)cellrec)[off + lane_];
        best = p.x * dl[0] + p.y * dl[1] + p.z * dl[2]; mine = __float_as_int(p.w); bx = p.x; by = p.y; bz = p.z;
      }
    }
  }
  for (int v0 = lane_; !listed && v0 < n; v0 += 64 * GRX_HULL_INFLIGHT) {   // several independent vertex fetches in flight per lane: one memory latency per 64 * GRX_HULL_INFLIGHT vertices
    float x[GRX_HULL_INFLIGHT], y[GRX_HULL_INFLIGHT], z[GRX_HULL_INFLIGHT];
#pragma unroll
    for (int u = 0; u < GRX_HULL_INFLIGHT; u++) { const int v = v0 + 64 * u < n ? v0 + 64 * u : n - 1; x[u] = verts[3 * v]; y[u] = verts[3 * v + 1]; z[u] = verts[3 * v + 2]; }
#pragma unroll
    for (int u = 0; u < GRX_HULL_INFLIGHT; u++) {
      const float t = x[u] * dl[0] + y[u] * dl[1] + z[u] * dl[2];
#ifdef GRX_NO_SECOND   // (A/B only: misses two tied vertices of one lane)
      if (v0 + 64 * u < n && t > best) { best = t; mine = v0 + 64 * u; bx = x[u]; by = y[u]; bz = z[u]; }
#else
      // the lane's runner-up costs ONE instruction per vertex: the second largest of {best, second, t} is their median (best >= second).  (As a compare + two
      // selects it cost 9 % of the Fetch launch, profiles/ab_r04_fetch_tiebreak.txt: the scan loop is the hot spot of the worlds that end a launch.)
      const float tt = (v0 + 64 * u < n) ? t : -3.0e38f;
      second = __builtin_amdgcn_fmed3f(best, second, tt);
      if (tt > best) { best = tt; mine = v0 + 64 * u; bx = x[u]; by = y[u]; bz = z[u]; }
#endif
    }
  }
  const float mx = grx_reduce_max(best);
  int bi = (int)(-grx_reduce_max((best == mx) ? -(float)mine : -3.0e38f));   // vertex indices are far below 2^24: exact in fp32
  {
    // the winner is unique beyond the rounding of the fp32 products (|t| < 1 m: error < 3e-7) in all but face-on / edge-on directions: no refinement, and its
    // coordinates are in the registers of the lane that scanned it -- no second trip to memory
    const float near_ = mx - 1.0e-6f * fmaxf(1.0f, fabsf(mx));
#ifdef GRX_NO_HULL_REFINE   // (A/B: the fp32 winner as it is)
    const unsigned long long cand = 1ull, cand2 = 0ull;
#else
    const unsigned long long cand = __ballot(best >= near_), cand2 = __ballot(second >= near_);
#endif
    if (sizeof(TF) == sizeof(HF) || aadr == nullptr || (__builtin_popcountll(cand) <= 1 && cand2 == 0ull)) {
      const unsigned long long own = __ballot(best == mx && mine == bi);
      const int src = own ? __builtin_ctzll(own) : 0;
      r[0] = grx_readlane_f(bx, src); r[1] = grx_readlane_f(by, src); r[2] = grx_readlane_f(bz, src);
      return bi;
    }
    if (cand2 == 0ull) {
      // the tied vertices are the winners of different lanes (the common case: a face of a few vertices): their fp64 projections come from the coordinates the
      // lanes still hold -- no further memory traffic -- and one wave argmax picks the reference's vertex (lowest index on an exact tie)
      const bool c_ = best >= near_;
      const double tb = c_ ? (double)bx * (double)dlm[0] + (double)by * (double)dlm[1] + (double)bz * (double)dlm[2] : 0.0;
      // wave maximum of the fp64 projections through their order-preserving 64-bit integer images (DPP butterflies, no LDS round trips; a resting hull face ties
      // with ALL of its vertices -- tens of candidates in every support evaluation of exactly the worlds that end a Fetch launch -- so a scalar walk over the
      // candidate lanes cost 6 % of the launch); candidates get a key >= 1, everything else 0
      const unsigned long long bits = (unsigned long long)__double_as_longlong(tb);
      const unsigned long long key = c_ ? ((bits >> 63) ? ~bits : (bits | 0x8000000000000000ull)) : 0ull;
      const unsigned long long kmax = grx_reduce_max_u64(key);
      const unsigned long long top = __ballot(c_ && key == kmax);
      int src = __builtin_ctzll(top), nb = __builtin_amdgcn_readlane(mine, src);
      if (top & (top - 1ull)) {   // an exact fp64 tie: the lowest vertex index wins (the reference's scan keeps the first maximum)
        unsigned long long mk = top & (top - 1ull);
        while (mk) { const int l = __builtin_ctzll(mk); mk &= mk - 1ull; const int il = __builtin_amdgcn_readlane(mine, l); if (il < nb) { nb = il; src = l; } }
      }
      r[0] = grx_readlane_f(bx, src); r[1] = grx_readlane_f(by, src); r[2] = grx_readlane_f(bz, src);
      return nb;
    }
  }
#endif
  bi = grx_mesh_support_refine(verts, aadr, anum, adj, dlm, bi, lane_);
  r[0] = verts[3 * bi]; r[1] = verts[3 * bi + 1]; r[2] = verts[3 * bi + 2];
  return bi;
}
// The same with a guess: a hull vertex that is not lower than any of its hull neighbours along dl IS the support vertex (convexity), so a
// vertex remembered from the previous substep is verified with one round of neighbour loads instead of a scan of the whole hull.
// Returns the support vertex (hint, or the winner of the full scan).
GRX_MEM int grx_mesh_support_hint(const GrxModel* m, int adr, int n, const MF* dlm, int hint, MF* r, int lane_, const int* cellhdr = nullptr) {
  const float* verts = m->mesh_vert + 3 * adr;
  // The guess is verified in the scan's arithmetic (fp32): this routine only serves the re-check of a cached separating direction, whose test keeps 1e-6 of
  // slack -- ten times what a tie between fp32 projections can hide.  The portal search proper goes through grx_mesh_support (fp64 tie-break).
  const HF dl[3] = {(HF)dlm[0], (HF)dlm[1], (HF)dlm[2]};
  if (hint >= 0 && hint < n) {
    const int aa = m->mesh_adjadr[adr + hint], an = m->mesh_adjnum[adr + hint];
    const HF t0 = verts[3 * hint] * dl[0] + verts[3 * hint + 1] * dl[1] + verts[3 * hint + 2] * dl[2];
#if !GRX_ON_DEVICE
    int higher = 0;
    for (int k = 0; k < an; k++) { const int nb = m->mesh_adj[aa + k]; higher |= (verts[3 * nb] * dl[0] + verts[3 * nb + 1] * dl[1] + verts[3 * nb + 2] * dl[2] > t0); }
#else
    int hi_ = 0;
    for (int k = lane_; k < an; k += 64) { const int nb = m->mesh_adj[aa + k]; hi_ |= (verts[3 * nb] * dl[0] + verts[3 * nb + 1] * dl[1] + verts[3 * nb + 2] * dl[2] > t0); }
    const int higher = __ballot(hi_ != 0) != 0ull;
#endif
    if (!higher) { r[0] = verts[3 * hint]; r[1] = verts[3 * hint + 1]; r[2] = verts[3 * hint + 2]; return hint; }
  }
  return grx_mesh_support(verts, n, dlm, r, lane_, m->mesh_adjadr + adr, m->mesh_adjnum + adr, m->mesh_adj, -1, nullptr, cellhdr, m->mesh_cellrec);
}
// W: wave-cooperative variant (uniform control flow, every lane holds the same values; mesh geoms allowed)
template <bool W, typename Q>
GRX_MEM void grx_mpr_support(const Q* q, const MF* d, GrxMprPt* o) {
  MF nd[3] = {-d[0], -d[1], -d[2]}, b[3];
#if GRX_DEVICE_PROFILE
  if (W && q->lane == 0) { q->prof[16 + 26] += 1; q->prof[16 + 27] += (q->t1 == 7 ? q->n1 : 0) + (q->t2 == 7 ? q->n2 : 0); }
#endif
#if GRX_DEVICE_PROFILE
  const long long tp0_ = clock64();
#endif
  int h1 = -1, h2 = -1, f1 = -1, f2 = -1, ek = 0;
#if GRX_DEVICE_HULL_HINTS
  if (W) {
    ek = __builtin_amdgcn_readfirstlane(q->hk);
    if (ek < 16) { const int pk = __builtin_amdgcn_readlane(q->hint, ek); h1 = (pk & 0xFFFF) - 1; h2 = (int)((unsigned)pk >> 16) - 1; }
    q->hk = ek + 1;
  }
#elif GRX_TWIN_HULL_HINTS
  if (W) {
    ek = q->hk;
    if (ek < 16) { const int pk = q->hints[ek]; h1 = (pk & 0xFFFF) - 1; h2 = (int)((unsigned)pk >> 16) - 1; }
    q->hk = ek + 1;
  }
#endif
  bool got1 = false, got2 = false;
#if GRX_DEVICE_HULL_HINTS && !defined(GRX_NO_DUAL_HINT)
  // Hull against hull with a guess for BOTH (the arm resting on the head link: the worlds that end a Fetch launch): the two guesses are verified in ONE round -- lanes 0-15 fetch
  // the 16 neighbour records of hull 1's guess, lanes 16-31 those of hull 2's, each half projects on ITS direction -- instead of two dependent fetch / project / ballot rounds one
  // after the other.  Same records, same fp64 projections, same margin as the guess check of grx_mesh_support: an accepted guess is the vertex that routine returns; a
  // hull whose guess fails goes through it without a guess (which is what it does itself after a failed check).
  if (W && q->t1 == 7 && q->t2 == 7 && h1 >= 0 && h1 < q->n1 && h2 >= 0 && h2 < q->n2 && q->nbr1 != nullptr && q->nbr2 != nullptr) {
    MF dl1[3], dl2[3];
    mulMatTVec3f(dl1, q->R1, d); mulMatTVec3f(dl2, q->R2, nd);
    const int l_ = q->lane, k_ = l_ & (GRX_NBR_RECS - 1); const bool sec = (l_ & GRX_NBR_RECS) != 0;
    const float4 p = ((const float4*)(sec ? q->nbr2 : q->nbr1))[GRX_NBR_RECS * (sec ? h2 : h1) + k_];
    const int deg1 = (int)grx_readlane_f(p.w, 0), deg2 = (int)grx_readlane_f(p.w, GRX_NBR_RECS);
    const double dx = sec ? (double)dl2[0] : (double)dl1[0], dy = sec ? (double)dl2[1] : (double)dl1[1], dz = sec ? (double)dl2[2] : (double)dl1[2];
    const double t = (double)p.x * dx + (double)p.y * dy + (double)p.z * dz;
    const unsigned long long tb = (unsigned long long)__double_as_longlong(t);
    const double t01 = __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(tb >> 32), 0) << 32) | (unsigned)__builtin_amdgcn_readlane((int)tb, 0)));
    const double t02 = __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(tb >> 32), GRX_NBR_RECS) << 32) | (unsigned)__builtin_amdgcn_readlane((int)tb, GRX_NBR_RECS)));
    const double dn = sqrt(dx * dx + dy * dy + dz * dz), t0 = sec ? t02 : t01;
    const bool beaten = k_ >= 1 && k_ <= (sec ? deg2 : deg1) && !(t0 - t > 1.0e-6 * dn);
    const unsigned long long bal = __ballot(beaten);
    got1 = deg1 >= 1 && (bal & 0xFFFFull) == 0ull; got2 = deg2 >= 1 && ((bal >> GRX_NBR_RECS) & 0xFFFFull) == 0ull;
    if (got1) { MF r[3] = {grx_readlane_f(p.x, 0), grx_readlane_f(p.y, 0), grx_readlane_f(p.z, 0)}; f1 = h1; mulMatVec3f(o->w, q->R1, r); }
    if (got2) { MF r[3] = {grx_readlane_f(p.x, GRX_NBR_RECS), grx_readlane_f(p.y, GRX_NBR_RECS), grx_readlane_f(p.z, GRX_NBR_RECS)}; f2 = h2; mulMatVec3f(b, q->R2, r); }
    h1 = h2 = -1;   // checked: a failed guess is not checked again
  }
#endif
  if (W && q->t1 == 7) { if (!got1) { MF dl[3], r[3]; mulMatTVec3f(dl, q->R1, d); f1 = grx_mesh_support(q->v1, q->n1, dl, r, q->lane, q->aadr1, q->anum1, q->adj, h1, q->nbr1, q->cell1, q->cellrec); mulMatVec3f(o->w, q->R1, r); } }
  else grx_geom_support(q->R1, q->s1, q->t1, d, o->w);
  if (W && q->t2 == 7) { if (!got2) { MF dl[3], r[3]; mulMatTVec3f(dl, q->R2, nd); f2 = grx_mesh_support(q->v2, q->n2, dl, r, q->lane, q->aadr2, q->anum2, q->adj, h2, q->nbr2, q->cell2, q->cellrec); mulMatVec3f(b, q->R2, r); } }
  else grx_geom_support(q->R2, q->s2, q->t2, nd, b);
#if GRX_DEVICE_HULL_HINTS
  // the winners become the guesses of this evaluation in the next substep.  (The guess words live in the world's HBM row, written and read by the lanes of ONE wave without a
  // fence: a stale, torn or foreign word can never change a result, because a guess is only ever a CANDIDATE -- grx_mesh_support accepts it when it provably is the support
  // vertex (tops every hull neighbour by the margin) and scans otherwise; tests/test_gpu_fetch.py::test_hull_caches_do_not_change_the_rollout.)
  if (W && ek < 16 && q->lane == ek) q->hint = ((f1 + 1) & 0xFFFF) | ((f2 + 1) << 16);
#elif GRX_TWIN_HULL_HINTS
  if (W && ek < 16) q->hints[ek] = ((f1 + 1) & 0xFFFF) | ((f2 + 1) << 16);
#endif
#if GRX_DEVICE_PROFILE
  if (W && q->lane == 0) q->prof[16 + 28] += clock64() - tp0_;
#endif
  for (int k = 0; k < 3; k++) { o->w[k] += d[k] * q->hm; o->v[k] = o->w[k] - (b[k] + q->c21[k] - d[k] * q->hm); }
}
// the portal is kept as four separate points (not an array): every access is to a named variable, so the 30 floats stay in registers
GRX_MEM void grx_mpr_portal_dir(const GrxMprPt& P1, const GrxMprPt& P2, const GrxMprPt& P3, MF* dir) {
  MF a[3], b[3];
  for (int k = 0; k < 3; k++) { a[k] = P2.v[k] - P1.v[k]; b[k] = P3.v[k] - P1.v[k]; }
  cross3f(dir, a, b); grx_normalize3f(dir);
}
GRX_MEM int grx_mpr_reach_tolerance(const GrxMprPt& P1, const GrxMprPt& P2, const GrxMprPt& P3, const GrxMprPt& v4, const MF* dir, MF tol) {
  MF d4 = dot3f(v4.v, dir), mn = grx_fmin(d4 - dot3f(P1.v, dir), grx_fmin(d4 - dot3f(P2.v, dir), d4 - dot3f(P3.v, dir)));
  return grx_mpr_eq(mn, tol) || mn < tol;
}
GRX_MEM void grx_mpr_set(GrxMprPt& dst, const GrxMprPt& src, int take) {
  for (int k = 0; k < 3; k++) { dst.v[k] = take ? src.v[k] : dst.v[k]; dst.w[k] = take ? src.w[k] : dst.w[k]; }
}
GRX_MEM void grx_mpr_expand(const GrxMprPt& P0, GrxMprPt& P1, GrxMprPt& P2, GrxMprPt& P3, const GrxMprPt& v4) {
  MF cr[3];
  cross3f(cr, v4.v, P0.v);
  const int s1 = dot3f(P1.v, cr) > 0.0f, s2 = dot3f(P2.v, cr) > 0.0f, s3 = dot3f(P3.v, cr) > 0.0f;
  // s1: (s2 ? P1 : P3) <- v4;   !s1: (s3 ? P2 : P1) <- v4
  const int to1 = (s1 && s2) || (!s1 && !s3), to2 = !s1 && s3, to3 = s1 && !s2;
  grx_mpr_set(P1, v4, to1); grx_mpr_set(P2, v4, to2); grx_mpr_set(P3, v4, to3);
}
GRX_MEM MF grx_mpr_seg_dist2(const MF* a, const MF* b, MF* w) {
  MF d[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, t = -dot3f(a, d), dd = dot3f(d, d);
  t = dd > 0.0f ? grx_fmin((MF)1.0f, grx_fmax((MF)0.0f, t / dd)) : (MF)0.0f;
  for (int k = 0; k < 3; k++) w[k] = a[k] + t * d[k];
  return dot3f(w, w);
}
GRX_MEM MF grx_mpr_tri_dist2(const MF* x0, const MF* b, const MF* cc, MF* w) {
  MF d1[3], d2[3];
  for (int k = 0; k < 3; k++) { d1[k] = b[k] - x0[k]; d2[k] = cc[k] - x0[k]; }
  MF u = dot3f(x0, x0), v = dot3f(d1, d1), ww = dot3f(d2, d2), p = dot3f(x0, d1), q = dot3f(x0, d2), r = dot3f(d1, d2);
  MF den = ww * v - r * r, best;
  if (!grx_mpr_zero(den)) {
    MF sp = (q * r - ww * p) / den, tp = (-sp * r - q) / ww;
    if ((grx_mpr_zero(sp) || sp > 0.0f) && (grx_mpr_eq(sp, 1.0f) || sp < 1.0f) && (grx_mpr_zero(tp) || tp > 0.0f) && (grx_mpr_eq(tp, 1.0f) || tp < 1.0f) &&
        (grx_mpr_eq(tp + sp, 1.0f) || tp + sp < 1.0f)) {
      for (int k = 0; k < 3; k++) w[k] = x0[k] + sp * d1[k] + tp * d2[k];
      // |w|^2, not the expanded quadratic form sp^2 v + tp^2 ww + 2 sp tp r + 2 sp p + 2 tp q + u of the published routine: for a portal whose vertices are
      // decimetres from an origin 0.2 mm off its plane the form's terms are ~0.1 and cancel to 4e-8, which in fp32 is rounding noise -- 25 um of depth at 0.2 mm,
      // measured by tests/test_gpu_anchors.py (a mesh cube standing on a vertex, away from the slab's centre).  The components of w cancel too, but to 1e-4
      // relative.  Same value in exact arithmetic (and in the fp64 oracle).
      best = dot3f(w, w);
      return best;
    }
  }
  MF w2[3], dist;
  best = grx_mpr_seg_dist2(x0, b, w);
  dist = grx_mpr_seg_dist2(x0, cc, w2); if (dist < best) { best = dist; w[0] = w2[0]; w[1] = w2[1]; w[2] = w2[2]; }
  dist = grx_mpr_seg_dist2(b, cc, w2); if (dist < best) { best = dist; w[0] = w2[0]; w[1] = w2[1]; w[2] = w2[2]; }
  return best;
}
// 0 = penetration (depth, dir, pos, surface witnesses w1 on geom 1 / w2 on geom 2 -- all relative to the centre of geom 1), -1 = separated
// sep (may be null): on a -1 return caused by a support point on the far side of the origin (v . d <= 0), sep[0..2] <- that direction d
// and sep[3] <- 1: d separates the two (inflated) geoms, which any later call can re-check with ONE support evaluation (grx_mesh_pairs)
#ifdef GRX_MPR_CALL   // the portal search behind a real call: its (fp64) register appetite stays out of the substep loop's allocation
#define GRX_MPR_FN GRX_MEM_CALL
#else
#define GRX_MPR_FN GRX_MEM
#endif
template <bool W, typename Q>
GRX_MPR_FN int grx_mpr_penetration(const Q* q, MF tol, int maxit, MF* depth, MF* dir, MF* pos, MF* w1, MF* w2, MF* sep = nullptr) {
#define GRX_MPR_SEP(D) do { if (W && sep) { sep[0] = (D)[0]; sep[1] = (D)[1]; sep[2] = (D)[2]; sep[3] = 1.0f; } } while (0)
  // lane-per-pair variant: the portal lives in registers; wave-cooperative variant: in LDS (every lane writes the same values)
  GrxMprPt r0_, r1_, r2_, r3_, r4_;
#ifdef GRX_MPR_PORTAL_REGS
  constexpr bool kLds = false;
#else
  constexpr bool kLds = W;
#endif
  GrxMprPt& P0 = kLds ? q->pts[0] : r0_; GrxMprPt& P1 = kLds ? q->pts[1] : r1_; GrxMprPt& P2 = kLds ? q->pts[2] : r2_; GrxMprPt& P3 = kLds ? q->pts[3] : r3_;
  GrxMprPt& v4 = kLds ? q->pts[4] : r4_;
  MF d[3], a[3], b[3], dotv;
  for (int k = 0; k < 3; k++) { P0.w[k] = 0.0f; P0.v[k] = -q->c21[k]; }
  if (grx_mpr_eq(P0.v[0], 0.0f) && grx_mpr_eq(P0.v[1], 0.0f) && grx_mpr_eq(P0.v[2], 0.0f)) P0.v[0] += GRX_MPR_EPS * 10.0f;
  for (int k = 0; k < 3; k++) d[k] = -P0.v[k];
  grx_normalize3f(d);
  grx_mpr_support<W>(q, d, &P1);
  dotv = dot3f(P1.v, d);
  if (grx_mpr_zero(dotv) || dotv < 0.0f) { GRX_MPR_SEP(d); return -1; }
  cross3f(d, P0.v, P1.v);
  if (grx_mpr_zero(dot3f(d, d))) {
    for (int k = 0; k < 3; k++) { w1[k] = P1.w[k]; w2[k] = P1.w[k] - P1.v[k]; pos[k] = 0.5f * (w1[k] + w2[k]); }
    if (grx_mpr_eq(P1.v[0], 0.0f) && grx_mpr_eq(P1.v[1], 0.0f) && grx_mpr_eq(P1.v[2], 0.0f)) { *depth = 0.0f; dir[0] = dir[1] = dir[2] = 0.0f; return 0; }
    dir[0] = P1.v[0]; dir[1] = P1.v[1]; dir[2] = P1.v[2]; *depth = grx_sqrt(dot3f(dir, dir)); grx_normalize3f(dir);
    return 0;
  }
  grx_normalize3f(d);
  grx_mpr_support<W>(q, d, &P2);
  dotv = dot3f(P2.v, d);
  if (grx_mpr_zero(dotv) || dotv < 0.0f) { GRX_MPR_SEP(d); return -1; }
  for (int k = 0; k < 3; k++) { a[k] = P1.v[k] - P0.v[k]; b[k] = P2.v[k] - P0.v[k]; }
  cross3f(d, a, b); grx_normalize3f(d);
  if (dot3f(d, P0.v) > 0.0f) { GrxMprPt t = P1; P1 = P2; P2 = t; d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2]; }
  for (int guard = 0;; guard++) {
    if (guard > 200) return -1;
    grx_mpr_support<W>(q, d, &P3);
    dotv = dot3f(P3.v, d);
    if (grx_mpr_zero(dotv) || dotv < 0.0f) { GRX_MPR_SEP(d); return -1; }
    int cont = 0;
    cross3f(a, P1.v, P3.v); dotv = dot3f(a, P0.v);
    if (dotv < 0.0f && !grx_mpr_zero(dotv)) { P2 = P3; cont = 1; }
    if (!cont) {
      cross3f(a, P3.v, P2.v); dotv = dot3f(a, P0.v);
      if (dotv < 0.0f && !grx_mpr_zero(dotv)) { P1 = P3; cont = 1; }
    }
    if (!cont) break;
    for (int k = 0; k < 3; k++) { a[k] = P1.v[k] - P0.v[k]; b[k] = P2.v[k] - P0.v[k]; }
    cross3f(d, a, b); grx_normalize3f(d);
  }
  for (int guard = 0;; guard++) {
    if (guard > 200) return -1;
    grx_mpr_portal_dir(P1, P2, P3, d);
    dotv = dot3f(d, P1.v);
    if (grx_mpr_zero(dotv) || dotv > 0.0f) break;
    grx_mpr_support<W>(q, d, &v4);
    dotv = dot3f(v4.v, d);
    if (!(grx_mpr_zero(dotv) || dotv > 0.0f)) { GRX_MPR_SEP(d); return -1; }
    if (grx_mpr_reach_tolerance(P1, P2, P3, v4, d, tol)) return -1;
    grx_mpr_expand(P0, P1, P2, P3, v4);
  }
  for (int it = 0;; it++) {
    grx_mpr_portal_dir(P1, P2, P3, d);
    grx_mpr_support<W>(q, d, &v4);
#if GRX_TWIN_MPR_STATS
    if (W) { g_grx_mesh_stats[2]++; if (it > maxit) g_grx_mesh_stats[3]++; }
#endif
    if (grx_mpr_reach_tolerance(P1, P2, P3, v4, d, tol) || it > maxit) {
      MF w[3];
      *depth = grx_sqrt(grx_mpr_tri_dist2(P1.v, P2.v, P3.v, w));
      if (grx_mpr_zero(w[0]) && grx_mpr_zero(w[1]) && grx_mpr_zero(w[2])) { w[0] = d[0]; w[1] = d[1]; w[2] = d[2]; }
      grx_normalize3f(w); dir[0] = w[0]; dir[1] = w[1]; dir[2] = w[2];
      MF bc[4], cr[3], sum;
      cross3f(cr, P1.v, P2.v); bc[0] = dot3f(cr, P3.v);
      cross3f(cr, P3.v, P2.v); bc[1] = dot3f(cr, P0.v);
      cross3f(cr, P0.v, P1.v); bc[2] = dot3f(cr, P3.v);
      cross3f(cr, P2.v, P1.v); bc[3] = dot3f(cr, P0.v);
      sum = bc[0] + bc[1] + bc[2] + bc[3];
      if (grx_mpr_zero(sum) || sum < 0.0f) {
        bc[0] = 0.0f;
        cross3f(cr, P2.v, P3.v); bc[1] = dot3f(cr, d);
        cross3f(cr, P3.v, P1.v); bc[2] = dot3f(cr, d);
        cross3f(cr, P1.v, P2.v); bc[3] = dot3f(cr, d);
        sum = bc[1] + bc[2] + bc[3];
      }
      // witness on geom 2 = w - v (+ the centre offset, which cancels in the relative frame except for P0: its witnesses are the two centres)
      const MF is = 1.0f / sum;
      for (int k = 0; k < 3; k++) {
        MF p1 = 0.0f, p2 = bc[0] * q->c21[k];
        p1 += bc[1] * P1.w[k] + bc[2] * P2.w[k] + bc[3] * P3.w[k];
        p2 += bc[1] * (P1.w[k] - P1.v[k]) + bc[2] * (P2.w[k] - P2.v[k]) + bc[3] * (P3.w[k] - P3.v[k]);
        pos[k] = 0.5f * (p1 + p2) * is;
      }
      // surface witnesses: the foot of the origin on the portal plane in barycentric coordinates of the triangle alone
      cross3f(cr, P2.v, P3.v); bc[1] = dot3f(cr, d);
      cross3f(cr, P3.v, P1.v); bc[2] = dot3f(cr, d);
      cross3f(cr, P1.v, P2.v); bc[3] = dot3f(cr, d);
      const MF it3 = 1.0f / (bc[1] + bc[2] + bc[3]);
      for (int k = 0; k < 3; k++) {
        w1[k] = (bc[1] * P1.w[k] + bc[2] * P2.w[k] + bc[3] * P3.w[k]) * it3;
        w2[k] = (bc[1] * (P1.w[k] - P1.v[k]) + bc[2] * (P2.w[k] - P2.v[k]) + bc[3] * (P3.w[k] - P3.v[k])) * it3;
      }
      return 0;
    }
    grx_mpr_expand(P0, P1, P2, P3, v4);
  }
}
#undef GRX_MPR_SEP
// analytic outward normal of a smooth geom (sphere, capsule, ellipsoid) at the world point p (see the oracle: the portal direction of a
// shallow contact is ill-conditioned, MuJoCo replaces it for smooth geoms); returns 0 for the other types
template <typename RF>
GRX_MEM int grx_smooth_normal(const RF* R, const MF* ce, const RF* szf, int type, const MF* p, MF* n) {
  const MF sz[3] = {szf[0], szf[1], szf[2]};
  MF d[3] = {p[0] - ce[0], p[1] - ce[1], p[2] - ce[2]}, loc[3], nl[3];
  mulMatTVec3f(loc, R, d);
  if (type == 2) { nl[0] = loc[0]; nl[1] = loc[1]; nl[2] = loc[2]; }
  else if (type == 3) { nl[0] = loc[0]; nl[1] = loc[1]; nl[2] = loc[2] > sz[1] ? loc[2] - sz[1] : (loc[2] < -sz[1] ? loc[2] + sz[1] : 0.0f); }
  else if (type == 4) { nl[0] = loc[0] / (sz[0] * sz[0]); nl[1] = loc[1] / (sz[1] * sz[1]); nl[2] = loc[2] / (sz[2] * sz[2]); }
  else return 0;
  const MF l2 = dot3f(nl, nl);
  if (l2 < 1e-30f) return 0;
  const MF il = 1.0f / grx_sqrt(l2);
  nl[0] *= il; nl[1] *= il; nl[2] *= il;
  mulMatVec3f(n, R, nl);
  return 1;
}
// Frame of geom g for the convex routine, in MF.  A geom of a FREE ROOT body (a free joint directly under the world: the manipulated objects) gets its frame
// straight from the world's qpos in MF arithmetic -- normalised quaternion -> body frame -> geom frame, the oracle's operation order -- instead of the fp32 frames of
// the kinematics stage: an object lying flat on a table is a line / face contact whose single contact point is decided by a tilt of ~1e-6 rad, which the ~1e-7
// rounding of the fp32 frames moves by centimetres (tools/emu_mixed.py: the kinematics stage was the only fp32 stage the AdroitHammer fixtures noticed).
GRX_MEM void grx_quat2mat_mf(MF* X, const MF* q) {
  const MF w = q[0], x = q[1], y = q[2], z = q[3];
  X[0] = w * w + x * x - y * y - z * z; X[1] = 2 * (x * y - w * z); X[2] = 2 * (x * z + w * y);
  X[3] = 2 * (x * y + w * z); X[4] = w * w - x * x + y * y - z * z; X[5] = 2 * (y * z - w * x);
  X[6] = 2 * (x * z - w * y); X[7] = 2 * (y * z + w * x); X[8] = w * w - x * x - y * y + z * z;
}
GRX_MEM void grx_geom_frame_mf(const GrxModel* m, const GrxCtx* c, int g, MF* R, MF* pos) {
  const int b = m->geom_bodyid[g];
#ifndef GRX_NO_FREE_FRAMES
  // root bodies (children of the world that are not mocap bodies and not members of a shift group): the oracle's kinematics of ONE body, in MF
  // (joint types this routine restates: free 0, slide 2, hinge 3.  A BALL joint -- type 1 -- on a root body is not restated: such a body keeps the fp32 frame of the kinematics
  // stage, which is also what the Jacobians of its contacts are built from; no packaged model has one, compile_mjcf is a general compiler)
  int supported = b > 0 && m->body_parent[b] == 0 && m->body_mocapid[b] < 0 && !(S::kShift && m->nshift && (m->geom_shift[g] || m->body_shift[b]));
  if (supported) { const int jn0 = m->body_jntnum[b], ja0 = m->body_jntadr[b]; for (int kk = 0; kk < jn0; kk++) { const int ty = m->jnt_type[ja0 + kk]; if (ty != 0 && ty != 2 && ty != 3) supported = 0; } }
  if (supported) {
    const int jn = m->body_jntnum[b], ja = m->body_jntadr[b];
    MF p[3], q[4];
    if (jn == 1 && m->jnt_type[ja] == 0) {
      const int qa = m->jnt_qposadr[ja];
      for (int k = 0; k < 3; k++) p[k] = c->qpos[qa + k];
      for (int k = 0; k < 4; k++) q[k] = c->qpos[qa + 3 + k];
    } else {
      for (int k = 0; k < 3; k++) p[k] = m->body_pos[3 * b + k];
      for (int k = 0; k < 4; k++) q[k] = m->body_quat[4 * b + k];
      for (int kk = 0; kk < jn; kk++) {
        const int j = ja + kk;
        MF Rq[9]; grx_quat2mat_mf(Rq, q);
        const MF jp[3] = {m->jnt_pos[3 * j], m->jnt_pos[3 * j + 1], m->jnt_pos[3 * j + 2]}, jx[3] = {m->jnt_axis[3 * j], m->jnt_axis[3 * j + 1], m->jnt_axis[3 * j + 2]};
        MF anchor[3], axis[3];
        mulMatVec3f(anchor, Rq, jp); anchor[0] += p[0]; anchor[1] += p[1]; anchor[2] += p[2];
        mulMatVec3f(axis, Rq, jx);
        const MF dq = (MF)c->qpos[m->jnt_qposadr[j]] - (MF)m->qpos0[m->jnt_qposadr[j]];
        if (m->jnt_type[j] == 2) { p[0] += axis[0] * dq; p[1] += axis[1] * dq; p[2] += axis[2] * dq; }
        else if (m->jnt_type[j] == 3) {
          const MF sn = sin(0.5 * (double)dq), cs = cos(0.5 * (double)dq);
          const MF qr[4] = {cs, jx[0] * sn, jx[1] * sn, jx[2] * sn};
          const MF qn[4] = {q[0] * qr[0] - q[1] * qr[1] - q[2] * qr[2] - q[3] * qr[3], q[0] * qr[1] + q[1] * qr[0] + q[2] * qr[3] - q[3] * qr[2],
                            q[0] * qr[2] - q[1] * qr[3] + q[2] * qr[0] + q[3] * qr[1], q[0] * qr[3] + q[1] * qr[2] - q[2] * qr[1] + q[3] * qr[0]};
          for (int k = 0; k < 4; k++) q[k] = qn[k];
          MF Rn[9], off[3]; grx_quat2mat_mf(Rn, q); mulMatVec3f(off, Rn, jp);
          p[0] = anchor[0] - off[0]; p[1] = anchor[1] - off[1]; p[2] = anchor[2] - off[2];
        }
      }
    }
    const MF n = grx_sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n > 1e-12f) { const MF r = 1.0f / n; q[0] *= r; q[1] *= r; q[2] *= r; q[3] *= r; }
    MF X[9], L[9];
    grx_quat2mat_mf(X, q);
    const MF lq[4] = {m->geom_quat[4 * g], m->geom_quat[4 * g + 1], m->geom_quat[4 * g + 2], m->geom_quat[4 * g + 3]};
    grx_quat2mat_mf(L, lq);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) R[3 * i + j] = X[3 * i] * L[j] + X[3 * i + 1] * L[3 + j] + X[3 * i + 2] * L[6 + j];
    const MF lp[3] = {m->geom_pos[3 * g], m->geom_pos[3 * g + 1], m->geom_pos[3 * g + 2]};
    for (int i = 0; i < 3; i++) pos[i] = p[i] + (X[3 * i] * lp[0] + X[3 * i + 1] * lp[1] + X[3 * i + 2] * lp[2]);
    return;
  }
#endif
  for (int k = 0; k < 9; k++) R[k] = c->gxmat[9 * g + k];
  for (int k = 0; k < 3; k++) pos[k] = c->gxpos[3 * g + k];
}
GRX_MEM void grx_convex_pair(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, int t1, int t2, float margin) {
  GrxMprPair q;
  MF p1[3], p2[3];
  grx_geom_frame_mf(m, c, g1, q.R1, p1); grx_geom_frame_mf(m, c, g2, q.R2, p2);
  q.t1 = t1; q.t2 = t2; q.hm = 0.5f * margin;
  for (int k = 0; k < 3; k++) { q.s1[k] = m->geom_size[3 * g1 + k]; q.s2[k] = m->geom_size[3 * g2 + k]; q.c21[k] = p2[k] - p1[k]; }
  MF depth, dir[3], pos[3], w1[3], w2[3];
  q.v1 = q.v2 = nullptr; q.n1 = q.n2 = 0; q.lane = 0; q.pts = nullptr; q.aadr1 = q.anum1 = q.aadr2 = q.anum2 = q.adj = nullptr; q.nbr1 = q.nbr2 = nullptr; q.hint = q.hk = 0; q.cell1 = q.cell2 = nullptr; q.cellrec = nullptr;
  if (grx_mpr_penetration<false>(&q, m->mpr_tolerance, m->mpr_iterations, &depth, dir, pos, w1, w2) != 0) return;
#if GRX_TWIN_TRACE
  if (getenv("GRX_TRACE_MPR")) {
    fprintf(stderr, "MPR pair %d g %d %d t %d %d\n R1", pair, g1, g2, t1, t2);
    for (int k = 0; k < 9; k++) fprintf(stderr, " %.17g", (double)q.R1[k]);
    fprintf(stderr, "\n R2"); for (int k = 0; k < 9; k++) fprintf(stderr, " %.17g", (double)q.R2[k]);
    fprintf(stderr, "\n c21 %.17g %.17g %.17g s1 %.9g %.9g %.9g s2 %.9g %.9g %.9g hm %.9g\n depth %.17g dir %.17g %.17g %.17g pos %.17g %.17g %.17g\n", (double)q.c21[0], (double)q.c21[1], (double)q.c21[2],
            (double)q.s1[0], (double)q.s1[1], (double)q.s1[2], (double)q.s2[0], (double)q.s2[1], (double)q.s2[2], (double)q.hm, (double)depth, (double)dir[0], (double)dir[1], (double)dir[2], (double)pos[0], (double)pos[1], (double)pos[2]);
  }
#endif
  if (dir[0] == 0.0f && dir[1] == 0.0f && dir[2] == 0.0f) return;
  // still relative to the centre of geom 1: the smooth normals are taken in that frame as well (the world offset only enters the stored contact position)
  const MF ce1[3] = {0.0f, 0.0f, 0.0f};
  MF n1[3] = {0.0f, 0.0f, 0.0f}, n2[3] = {0.0f, 0.0f, 0.0f};
  const int h1 = grx_smooth_normal(q.R1, ce1, q.s1, t1, pos, n1), h2 = grx_smooth_normal(q.R2, q.c21, q.s2, t2, pos, n2);
  if (h1 || h2) {
    MF n[3] = {n1[0] - n2[0], n1[1] - n2[1], n1[2] - n2[2]};
    const MF l2 = dot3f(n, n);
    if (l2 > 1e-30f) {
      const MF il = 1.0f / grx_sqrt(l2); dir[0] = n[0] * il; dir[1] = n[1] * il; dir[2] = n[2] * il;
      // penetration along the corrected normal: extreme point of a smooth geom, portal witness of a box / cylinder (see the oracle)
      MF nd[3] = {-dir[0], -dir[1], -dir[2]};
      if (h1) { grx_geom_support(q.R1, q.s1, t1, dir, w1); for (int k = 0; k < 3; k++) w1[k] += dir[k] * q.hm; }
      if (h2) { grx_geom_support(q.R2, q.s2, t2, nd, w2); for (int k = 0; k < 3; k++) w2[k] += q.c21[k] - dir[k] * q.hm; }
      depth = (w1[0] - w2[0]) * dir[0] + (w1[1] - w2[1]) * dir[1] + (w1[2] - w2[2]) * dir[2];
    }
  }
  const float posw[3] = {(float)(pos[0] + p1[0]), (float)(pos[1] + p1[1]), (float)(pos[2] + p1[2])}, dirf[3] = {(float)dir[0], (float)dir[1], (float)dir[2]};
  grx_add_contact(c, pair, posw, dirf, (float)(margin - depth));
}
