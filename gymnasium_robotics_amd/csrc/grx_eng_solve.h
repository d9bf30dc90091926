// grx_eng_solve.h -- K10b noslip (dual Gauss-Seidel), object-block refinement, and the solve + integrate state machine of one substep.
// A FRAGMENT of csrc/grx_engine.h: textually included INSIDE `template <class S> struct GrxEngine { ... }` (every function here is a static member), in the order the engine
// header lists; not a standalone header.  The split is purely textual (round 5): the token stream of the translation units is unchanged.
// ------------------------------------------------------------------------------------------
// K10b noslip post-solver (MuJoCo option noslip_iterations; Adroit: assets/adroit_hand/adroit_assets.xml:3): projected Gauss-Seidel on the
// dual with the regulariser removed, over the friction-loss rows and the pairs of opposing pyramid edges of the frictional contacts (the
// oracle's solve_noslip restates the reference algorithm with an explicit A = J M^-1 J').  Here it is matrix-free, at wavefront level: the
// rows are visited one after the other, every dot product runs across the lanes (lane i = dof i):
//     t = M^-1 J_r'            (lane i: sum over the row's span of Minv[i][d] J_r[d])
//     A_rr = J_r . t ,  res_r = J_r . a - aref_r          (a = the acceleration implied by the current forces, kept in c->qacc)
//     f_r <- projected update ,  a += t * delta
// M^-1 is formed once per substep (LDL' of M, one right-hand side per lane).  Ends with M a in c->Ma (so that the caller's
// qfrc_constraint = M a - qfrc_smooth holds for the new forces).
// ------------------------------------------------------------------------------------------
GRX_MEM void grx_noslip(const GrxModel* m, GrxCtx* c, int nefc, int lane_) {
  const int nv = GRX_NVC, maxiter = m->noslip_iterations;
  const int ne = c->cnt[3], nf = c->cnt[4], ncon = c->cnt[0] < c->maxcon ? c->cnt[0] : c->maxcon;
  GRX_TICK(c, GRX_P_NEVAL);
  // ---- M^-1 into c->minv (= the Hessian's buffer: Newton is done with it).  Specialised shapes on the GPU: Gauss-Jordan in registers.
#if GRX_ON_DEVICE
  if (S::kFixed && S::NV > 0 && S::NV <= 40) grx_sym_inverse_reg<(S::NV > 0 && S::NV <= 40) ? S::NV : 1>(c->M, nv, c->minv, lane_);
  else
#endif
  {
    // in-place Gauss-Jordan in LDS (lane i = row i; the matrix is positive definite: no pivoting): step k eliminates column k from every other
    // row and turns it into the k-th column of the inverse, then row k is scaled -- the arithmetic of grx_sym_inverse_reg, through LDS
    FOR_LANES { for (int i = lane; i < nv * nv; i += 64) c->minv[i] = c->M[i]; }
    WAVE_SYNC();
    int bad = 0;
    for (int k = 0; k < nv; k++) {
      const float d = c->minv[k * nv + k];
      bad |= !(d > 0.0f);
      const float pinv = 1.0f / d;
      FOR_LANES {
        for (int i = lane; i < nv; i += 64) {
          if (i == k) continue;
          float* row = c->minv + i * nv; const float* piv = c->minv + k * nv;
          const float f = row[k] * pinv;
          for (int j = 0; j < nv; j++) if (j != k) row[j] = fmaf(-f, piv[j], row[j]);
          row[k] = -f;
        }
      }
      WAVE_SYNC();
      FOR_LANES { for (int j = lane; j < nv; j += 64) c->minv[k * nv + j] = (j == k) ? pinv : c->minv[k * nv + j] * pinv; }
      WAVE_SYNC();
    }
    if (bad) { LANE0 { c->cnt[2] |= GRX_ST_FACTOR; } }
  }
  GRX_SUBTICK(c, 17);
  const float scale = 1.0f / (m->meaninertia * (float)(nv > 1 ? nv : 1));
  float improvement0 = 0.0f;   // cost change of dropping the regulariser: 0.5 sum f^2 R (enters the first sweep's improvement)
  {
    GRX_LANEVAR(ip);
    FOR_LANES { float sacc = 0.0f; for (int r = lane; r < nefc; r += 64) { const float f = c->efc_force[r]; sacc += 0.5f * f * f / c->efc_D[r]; } LV(ip) = sacc; }
    improvement0 = grx_reduce_sum(ip);
  }
#if GRX_ON_DEVICE
  {
    // GPU: the sweep state lives in registers -- lane i holds a_i, lane r holds the r-th friction-loss row (dof, aref, bound, force, A_rr);
    // a row update is a handful of v_readlane broadcasts plus one LDS read of the M^-1 column, no barrier.  Same arithmetic, same order
    // as the plain version below (which the lane emulator runs).
    float a_l = lane_ < nv ? c->qacc[lane_] : 0.0f;
    int fr_d = 0; float fr_aref = 0.0f, fr_fl = 0.0f, fr_f = 0.0f, fr_arr = 1.0f, fr_rinv = 1.0f;
    if (lane_ < nf) {
      const int r = ne + lane_;
      fr_d = GRX_ROW_IDOF(c->efc_id[r]); fr_aref = c->efc_aref[r]; fr_fl = c->efc_floss[r]; fr_f = c->efc_force[r]; fr_arr = c->minv[fr_d * nv + fr_d];
      fr_rinv = 1.0f / fmaxf(GRX_MINVAL, fr_arr);
    }
    // Sweep-invariant part of a contact pair: t = M^-1 J' of its two rows (one word per lane each) and A00 / A01 / A11.  For the first KC pairs (sweep
    // order) they are formed once per substep and parked in the Newton scratch that is dead by now (grad, search, Mv, tmpv, efc_jar, efc_jv: contiguous);
    // a sweep then costs such a pair two LDS reads per lane and the two residual reductions instead of ~4 x len LDS reads and five reductions.
    float* const tc = c->grad;
    const int tstride = 2 * nv + 4;
    int KC = (int)(c->efc_force - c->grad) / tstride;
    if (KC > 24) KC = 24;
    {
      int pi = 0;
      for (int k = 0; k < ncon && pi < KC; k++) {
        const int r0 = c->con_efc[k], nr = c->con_nr[k];
        if (r0 < 0 || nr < 2) continue;
        for (int j = r0; j + 1 < r0 + nr && j + 1 < nefc && pi < KC; j += 2, pi++) {
          const int infoA = c->efc_row[j], idA = c->efc_id[j], infoB = c->efc_row[j + 1], idB = c->efc_id[j + 1];
          float ta = 0.0f, tb = 0.0f, ja = 0.0f, jb = 0.0f;
          if (lane_ < nv) {
            const float* mi = c->minv + lane_ * nv;
            const int offA = GRX_ROW_OFF(infoA), loA = GRX_ROW_LO(infoA), lenA = GRX_ROW_LEN(infoA), offB = GRX_ROW_OFF(infoB), loB = GRX_ROW_LO(infoB), lenB = GRX_ROW_LEN(infoB);
            for (int e = 0; e < lenA; e++) ta += c->Jp[offA + e] * mi[loA + e];
            for (int e = 0; e < lenB; e++) tb += c->Jp[offB + e] * mi[loB + e];
            if (S::kTwoSpan) {
              const int lo2A = GRX_ROWB_LO(idA), len2A = GRX_ROWB_LEN(idA), lo2B = GRX_ROWB_LO(idB), len2B = GRX_ROWB_LEN(idB);
              for (int e = 0; e < len2A; e++) ta += c->Jp[offA + lenA + e] * mi[lo2A + e];
              for (int e = 0; e < len2B; e++) tb += c->Jp[offB + lenB + e] * mi[lo2B + e];
            }
            const int pa = grx_row_pos(infoA, idA, lane_), pb = grx_row_pos(infoB, idB, lane_);
            ja = pa >= 0 ? c->Jp[offA + pa] : 0.0f; jb = pb >= 0 ? c->Jp[offB + pb] : 0.0f;
          }
          const float A00 = grx_reduce_sum(ja * ta), A01 = grx_reduce_sum(ja * tb), A11 = grx_reduce_sum(jb * tb);
          float* slot = tc + pi * tstride;
          if (lane_ < nv) { slot[lane_] = ta; slot[nv + lane_] = tb; }
          if (lane_ == 0) { slot[2 * nv] = A00; slot[2 * nv + 1] = A01; slot[2 * nv + 2] = A11; }
        }
      }
      __syncthreads();
    }
    for (int iter = 0; iter < maxiter; iter++) {
      float improvement = iter == 0 ? improvement0 : 0.0f;
      // one row after the other (Gauss-Seidel): the only LDS access of a row -- its column of M^-1 -- is fetched one row ahead, the division by A_rr
      // became a multiplication by the reciprocal formed with the row state (the plain version below divides: last-ulp difference)
      float col_next = (lane_ < nv && nf > 0) ? c->minv[lane_ * nv + __builtin_amdgcn_readlane(fr_d, 0)] : 0.0f;
      for (int r = 0; r < nf; r++) {
        const int d = __builtin_amdgcn_readlane(fr_d, r);
        const float col = col_next;
        const int dn = __builtin_amdgcn_readlane(fr_d, r + 1 < nf ? r + 1 : r);
        col_next = lane_ < nv ? c->minv[lane_ * nv + dn] : 0.0f;
        const float Arr = grx_readlane_f(fr_arr, r), rinv = grx_readlane_f(fr_rinv, r), res = grx_readlane_f(a_l, d) - grx_readlane_f(fr_aref, r), old = grx_readlane_f(fr_f, r),
                    fl = grx_readlane_f(fr_fl, r);
        float fn = old - res * rinv;
        fn = fn < -fl ? -fl : (fn > fl ? fl : fn);
        const float dl = fn - old;
        improvement -= 0.5f * dl * dl * Arr + dl * res;
        fr_f = (lane_ == r) ? fn : fr_f;
        a_l = fmaf(col, dl, a_l);
      }
      int pi = 0;
      for (int k = 0; k < ncon; k++) {
        const int r0 = c->con_efc[k], nr = c->con_nr[k];
        if (r0 < 0 || nr < 2) continue;
        for (int j = r0; j + 1 < r0 + nr && j + 1 < nefc; j += 2, pi++) {
          const int infoA = c->efc_row[j], idA = c->efc_id[j], infoB = c->efc_row[j + 1], idB = c->efc_id[j + 1];
          float ta = 0.0f, tb = 0.0f, ja = 0.0f, jb = 0.0f, A00, A01, A11;
          if (lane_ < nv) {
            const int offA = GRX_ROW_OFF(infoA), offB = GRX_ROW_OFF(infoB);
            const int pa = grx_row_pos(infoA, idA, lane_), pb = grx_row_pos(infoB, idB, lane_);
            ja = pa >= 0 ? c->Jp[offA + pa] : 0.0f; jb = pb >= 0 ? c->Jp[offB + pb] : 0.0f;
          }
          if (pi < KC) {   // parked above
            const float* slot = tc + pi * tstride;
            if (lane_ < nv) { ta = slot[lane_]; tb = slot[nv + lane_]; }
            A00 = slot[2 * nv]; A01 = slot[2 * nv + 1]; A11 = slot[2 * nv + 2];
          } else {
            if (lane_ < nv) {
              const float* mi = c->minv + lane_ * nv;
              const int offA = GRX_ROW_OFF(infoA), loA = GRX_ROW_LO(infoA), lenA = GRX_ROW_LEN(infoA), offB = GRX_ROW_OFF(infoB), loB = GRX_ROW_LO(infoB), lenB = GRX_ROW_LEN(infoB);
              for (int e = 0; e < lenA; e++) ta += c->Jp[offA + e] * mi[loA + e];
              for (int e = 0; e < lenB; e++) tb += c->Jp[offB + e] * mi[loB + e];
              if (S::kTwoSpan) {
                const int lo2A = GRX_ROWB_LO(idA), len2A = GRX_ROWB_LEN(idA), lo2B = GRX_ROWB_LO(idB), len2B = GRX_ROWB_LEN(idB);
                for (int e = 0; e < len2A; e++) ta += c->Jp[offA + lenA + e] * mi[lo2A + e];
                for (int e = 0; e < len2B; e++) tb += c->Jp[offB + lenB + e] * mi[lo2B + e];
              }
            }
            A00 = grx_reduce_sum(ja * ta); A01 = grx_reduce_sum(ja * tb); A11 = grx_reduce_sum(jb * tb);
          }
          const float res0 = grx_reduce_sum(ja * a_l) - c->efc_aref[j], res1 = grx_reduce_sum(jb * a_l) - c->efc_aref[j + 1];
          const float o0 = c->efc_force[j], o1 = c->efc_force[j + 1];
          const float bc0 = res0 - (A00 * o0 + A01 * o1), bc1 = res1 - (A01 * o0 + A11 * o1);
          const float mid = 0.5f * (o0 + o1), K1 = A00 + A11 - 2.0f * A01, K0 = mid * (A00 - A11) + bc0 - bc1;
          float f0, f1;
          if (K1 < GRX_MINVAL) { f0 = f1 = mid; }
          else {
            const float y = -K0 / K1;
            if (y < -mid) { f0 = 0.0f; f1 = 2.0f * mid; } else if (y > mid) { f0 = 2.0f * mid; f1 = 0.0f; } else { f0 = mid + y; f1 = mid - y; }
          }
          const float d0 = f0 - o0, d1 = f1 - o1;
          improvement -= 0.5f * (d0 * (A00 * d0 + A01 * d1) + d1 * (A01 * d0 + A11 * d1)) + d0 * res0 + d1 * res1;
          __syncthreads();
          if (lane_ == 0) { c->efc_force[j] = f0; c->efc_force[j + 1] = f1; }
          __syncthreads();
          a_l += ta * d0 + tb * d1;
        }
      }
      if (improvement * scale < m->noslip_tolerance) break;
    }
    __syncthreads();
    if (lane_ < nv) c->qacc[lane_] = a_l;
    if (lane_ < nf) c->efc_force[ne + lane_] = fr_f;
    __syncthreads();
  }
#else
  for (int iter = 0; iter < maxiter; iter++) {
    float improvement = iter == 0 ? improvement0 : 0.0f;
    // ---- dry friction: J = e_d, so t is a column of M^-1 and no reduction is needed
    for (int r = ne; r < ne + nf; r++) {
      const int d = GRX_ROW_IDOF(c->efc_id[r]);
      const float Arr = c->minv[d * nv + d], res = c->qacc[d] - c->efc_aref[r], old = c->efc_force[r], fl = c->efc_floss[r];
      float fn = old - res / fmaxf(GRX_MINVAL, Arr);
      fn = fn < -fl ? -fl : (fn > fl ? fl : fn);
      const float dl = fn - old;
      improvement -= 0.5f * dl * dl * Arr + dl * res;
      WAVE_SYNC();
      c->efc_force[r] = fn;
      FOR_LANES { if (lane < nv) c->qacc[lane] += c->minv[lane * nv + d] * dl; }
      WAVE_SYNC();
    }
    // ---- contact friction: pairs of opposing pyramid edges (their sum, the normal force, is kept)
    for (int k = 0; k < ncon; k++) {
      const int r0 = c->con_efc[k], nr = c->con_nr[k];
      if (r0 < 0 || nr < 2) continue;
      for (int j = r0; j + 1 < r0 + nr && j + 1 < nefc; j += 2) {
        GRX_LANEVAR(tA); GRX_LANEVAR(tB); GRX_LANEVAR(p00); GRX_LANEVAR(p01); GRX_LANEVAR(p11); GRX_LANEVAR(pr0); GRX_LANEVAR(pr1);
        const int infoA = c->efc_row[j], idA = c->efc_id[j], infoB = c->efc_row[j + 1], idB = c->efc_id[j + 1];
        FOR_LANES {
          float ta = 0.0f, tb = 0.0f, ja = 0.0f, jb = 0.0f, al = 0.0f;
          if (lane < nv) {
            const float* mi = c->minv + lane * nv;
            const int offA = GRX_ROW_OFF(infoA), loA = GRX_ROW_LO(infoA), lenA = GRX_ROW_LEN(infoA), offB = GRX_ROW_OFF(infoB), loB = GRX_ROW_LO(infoB), lenB = GRX_ROW_LEN(infoB);
            for (int e = 0; e < lenA; e++) ta += c->Jp[offA + e] * mi[loA + e];
            for (int e = 0; e < lenB; e++) tb += c->Jp[offB + e] * mi[loB + e];
            if (S::kTwoSpan) {
              const int lo2A = GRX_ROWB_LO(idA), len2A = GRX_ROWB_LEN(idA), lo2B = GRX_ROWB_LO(idB), len2B = GRX_ROWB_LEN(idB);
              for (int e = 0; e < len2A; e++) ta += c->Jp[offA + lenA + e] * mi[lo2A + e];
              for (int e = 0; e < len2B; e++) tb += c->Jp[offB + lenB + e] * mi[lo2B + e];
            }
            const int pa = grx_row_pos(infoA, idA, lane), pb = grx_row_pos(infoB, idB, lane);
            ja = pa >= 0 ? c->Jp[offA + pa] : 0.0f; jb = pb >= 0 ? c->Jp[offB + pb] : 0.0f;
            al = c->qacc[lane];
          }
          LV(tA) = ta; LV(tB) = tb; LV(p00) = ja * ta; LV(p01) = ja * tb; LV(p11) = jb * tb; LV(pr0) = ja * al; LV(pr1) = jb * al;
        }
        const float A00 = grx_reduce_sum(p00), A01 = grx_reduce_sum(p01), A11 = grx_reduce_sum(p11);
        const float res0 = grx_reduce_sum(pr0) - c->efc_aref[j], res1 = grx_reduce_sum(pr1) - c->efc_aref[j + 1];
        const float o0 = c->efc_force[j], o1 = c->efc_force[j + 1];
        const float bc0 = res0 - (A00 * o0 + A01 * o1), bc1 = res1 - (A01 * o0 + A11 * o1);
        const float mid = 0.5f * (o0 + o1), K1 = A00 + A11 - 2.0f * A01, K0 = mid * (A00 - A11) + bc0 - bc1;
        float f0, f1;
        if (K1 < GRX_MINVAL) { f0 = f1 = mid; }
        else {
          const float y = -K0 / K1;
          if (y < -mid) { f0 = 0.0f; f1 = 2.0f * mid; } else if (y > mid) { f0 = 2.0f * mid; f1 = 0.0f; } else { f0 = mid + y; f1 = mid - y; }
        }
        const float d0 = f0 - o0, d1 = f1 - o1;
        improvement -= 0.5f * (d0 * (A00 * d0 + A01 * d1) + d1 * (A01 * d0 + A11 * d1)) + d0 * res0 + d1 * res1;
        WAVE_SYNC();
        c->efc_force[j] = f0; c->efc_force[j + 1] = f1;
        FOR_LANES { if (lane < nv) c->qacc[lane] += LV(tA) * d0 + LV(tB) * d1; }
        WAVE_SYNC();
      }
    }
    if (improvement * scale < m->noslip_tolerance) break;
  }
#endif
  GRX_SUBTICK(c, 18);
  // M a for the caller (qfrc_constraint = M a - qfrc_smooth)
  FOR_LANES {
    for (int i = lane; i < nv; i += 64) {
      float sacc = 0.0f;
      for (int j = 0; j < nv; j++) sacc += c->M[i * nv + j] * c->qacc[j];
      c->Ma[i] = sacc;
    }
  }
  WAVE_SYNC();
}

// One more Newton step in the subspace of a DECOUPLED trailing free object (m->nfreeobj = 6, no active row links it to the robot), after Newton has converged by an
// exact full step.  H = M + J'DJ of a light body under a stiff contact carries the body's inertia at ~1e-4 of the contact's entries (the puck of FetchSlide:
// I = 5.8e-4 against D r^2 = 4.3), so the fp32 Hessian resolves the curvature of the body's weak mode -- rocking about the contact point -- to ~4e-4 and a full
// step of size 200 rad/s^2 leaves that mode 3e-2 rad/s^2 off the minimiser: the whole rotation-velocity discrepancy of the FetchSlide fixtures (tools/emu_mixed.py,
// tools/emu_trace.py).  The GRADIENT in that mode, taken from the rows (M a - qfrc_smooth - J'f with f from the carried J a - aref), has no such loss, so one more
// step with the same 6 x 6 Hessian block contracts the error by another 4e-4.  The step is only applied when it leaves every row of the object in its state (the
// block is then exact for the piece): cost one pass over the rows for 6 lanes and a 6 x 6 solve.
GRX_MEM void grx_refine_object_block(const GrxModel* m, GrxCtx* c, int nefc, int lane_) {
  GRX_FRESH_MODEL(m, c);
  const int nv = GRX_NVC, o0 = nv - 6;
  // J'f over the object's six dofs: one lane per row (the rows that touch the object are few), six wave sums
  GRX_LANEVAR(j0); GRX_LANEVAR(j1); GRX_LANEVAR(j2); GRX_LANEVAR(j3); GRX_LANEVAR(j4); GRX_LANEVAR(j5);
  FOR_LANES {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f, a4 = 0.0f, a5 = 0.0f;
    for (int r = lane; r < nefc; r += 64) {
      const int info = c->efc_row[r], idb = S::kTwoSpan ? c->efc_id[r] : 0;
      const int p0 = grx_row_pos(info, idb, o0), p5 = grx_row_pos(info, idb, o0 + 5);
      if (p0 < 0 && p5 < 0) continue;   // spans are contiguous dof ranges: a row that holds neither end of the object's six dofs holds none of them
      const float x = c->efc_jar[r], D = c->efc_D[r]; const int kind = c->efc_kind[r];
      float f;
      if (kind == GRX_ROW_EQ) f = -D * x;
      else if (kind == GRX_ROW_FRICTION) { const float fl = c->efc_floss[r], Rf = fl / D; f = (x <= -Rf) ? fl : ((x >= Rf) ? -fl : -D * x); }
      else f = (x < 0.0f) ? -D * x : 0.0f;
      const float* J = c->Jp + GRX_ROW_OFF(info);
      const int q1 = grx_row_pos(info, idb, o0 + 1), q2 = grx_row_pos(info, idb, o0 + 2), q3 = grx_row_pos(info, idb, o0 + 3), q4 = grx_row_pos(info, idb, o0 + 4);
      a0 += (p0 >= 0 ? J[p0] : 0.0f) * f; a1 += (q1 >= 0 ? J[q1] : 0.0f) * f; a2 += (q2 >= 0 ? J[q2] : 0.0f) * f;
      a3 += (q3 >= 0 ? J[q3] : 0.0f) * f; a4 += (q4 >= 0 ? J[q4] : 0.0f) * f; a5 += (p5 >= 0 ? J[p5] : 0.0f) * f;
    }
    LV(j0) = a0; LV(j1) = a1; LV(j2) = a2; LV(j3) = a3; LV(j4) = a4; LV(j5) = a5;
  }
  const float jf[6] = {grx_reduce_sum(j0), grx_reduce_sum(j1), grx_reduce_sum(j2), grx_reduce_sum(j3), grx_reduce_sum(j4), grx_reduce_sum(j5)};
  FOR_LANES { if (lane < 6) c->search[o0 + lane] = -(c->Ma[o0 + lane] - c->qfrc_smooth[o0 + lane] - GRX_SEL6(jf, lane)); }
  WAVE_SYNC();
#if !GRX_ON_DEVICE
  {
    static float blk[36];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) blk[6 * i + j] = c->A[(o0 + i) * nv + o0 + j];
    if (grx_sym_factor(blk, 6, lane_)) return;
    grx_sym_solve(blk, 6, c->search + o0, lane_);
  }
#else
  if (grx_sym_solve_reg<6>(c->A + o0 * nv + o0, nv, c->search + o0, lane_)) return;
#endif
  WAVE_SYNC();
  // the step must leave every row that touches the object in its state; jv of those rows
  const float d0 = c->search[o0], d1 = c->search[o0 + 1], d2 = c->search[o0 + 2], d3 = c->search[o0 + 3], d4 = c->search[o0 + 4], d5 = c->search[o0 + 5];
  GRX_LANEVAR_I(flipp);
  FOR_LANES {
    int flip = 0;
    for (int r = lane; r < nefc; r += 64) {
      const int info = c->efc_row[r], idb = S::kTwoSpan ? c->efc_id[r] : 0;
      const int p0 = grx_row_pos(info, idb, o0), p5 = grx_row_pos(info, idb, o0 + 5);
      float jv = 0.0f;
      if (p0 >= 0 || p5 >= 0) {
        const float* J = c->Jp + GRX_ROW_OFF(info);
        const int q1 = grx_row_pos(info, idb, o0 + 1), q2 = grx_row_pos(info, idb, o0 + 2), q3 = grx_row_pos(info, idb, o0 + 3), q4 = grx_row_pos(info, idb, o0 + 4);
        jv = (p0 >= 0 ? J[p0] : 0.0f) * d0 + (q1 >= 0 ? J[q1] : 0.0f) * d1 + (q2 >= 0 ? J[q2] : 0.0f) * d2 + (q3 >= 0 ? J[q3] : 0.0f) * d3 + (q4 >= 0 ? J[q4] : 0.0f) * d4 + (p5 >= 0 ? J[p5] : 0.0f) * d5;
        const float x0 = c->efc_jar[r], x1 = x0 + jv; const int kind = c->efc_kind[r];
        if (kind == GRX_ROW_FRICTION) { const float Rf = c->efc_floss[r] / c->efc_D[r]; flip |= ((x0 <= -Rf) != (x1 <= -Rf)) | ((x0 >= Rf) != (x1 >= Rf)); }
        else if (kind != GRX_ROW_EQ) flip |= ((x0 < 0.0f) != (x1 < 0.0f));
      }
      c->efc_jv[r] = jv;
    }
    LV(flipp) = flip;
  }
  if (GRX_BALLOT(flipp) != 0ull) return;
  WAVE_SYNC();
  FOR_LANES {
    if (lane < 6) {
      const int i = o0 + lane;
      const float* Mi = c->M + i * nv + o0;
      c->qacc[i] += c->search[i]; c->Ma[i] += Mi[0] * d0 + Mi[1] * d1 + Mi[2] * d2 + Mi[3] * d3 + Mi[4] * d4 + Mi[5] * d5;
    }
    for (int r = lane; r < nefc; r += 64) c->efc_jar[r] += c->efc_jv[r];
  }
  WAVE_SYNC();
}

// Constraint solve (Newton) + optional semi-implicit Euler step as ONE state machine, so that the three heavy
// primitives -- row evaluation, Hessian assembly and the register-resident linear solve -- each have a single call site
// in the kernel: the fused 20-substep loop has to stay inside the instruction cache.
//   phase 0: Newton iterations on the primal problem        (A = M + J' D J,      rhs = -gradient)
//   phase 2: no constraint rows at all                      (A = M,               rhs = qfrc_smooth)
//   phase 1: Euler velocity update with implicit damping    (A = M + h diag(B),   rhs = qfrc_smooth + qfrc_constraint)
GRX_MEM void grx_solve_integrate(const GrxModel* m, GrxCtx* c, int do_euler, int lane_) {
  GRX_OPAQUE_STAGE(lane_);
  GRX_FRESH_MODEL(m, c);
  const int nv = GRX_NVC; const float h = m->timestep;
  const int nefc = c->cnt[1];
  const float scale = 1.0f / (m->meaninertia * (float)(nv > 1 ? nv : 1));
  const int implicit_damp = (m->anydamp && m->eulerdamp);
  int phase = nefc ? 0 : 2, it = 0, done = 0, full_step = 0;
  float last_stepmax = 0.0f;   // largest component of the last accepted Newton step
  int exact_exit = 0, last_split = 0;   // converged by an exact full step (no row changed state) / the last linear solve ran on the decoupled robot | object blocks
  float alpha_prev = 0.0f;   // the step length accepted by the previous Newton iteration (gradient advance of the incremental path)
#if GRX_TWIN_STAGEHOOK
  if (g_grx_solve_mode == 2 && nefc) phase = 1;
#endif
  GRX_COUNT(c, 30, 1);     // profiling build: constrained solves (substeps) of the step
  if (nefc) GRX_TWIN_STAT(0);
  GRX_COUNT(c, 31, nefc);  // ... and their constraint rows
  // the linear solve leaves c->A intact where it runs from registers (grx_sym_solve_full): the Hessian can then be corrected in place
  const int keepA = S::kIncrHess && (nv == 21 || nv == 14 || nv == 15 || nv == 24 || nv == 29 || nv == 30 || nv == 33 || nv == 36);
  // Newton starts from the previous solution (qacc_warmstart).  MuJoCo starts from the cheaper of (warmstart,
  // M^-1 qfrc_smooth); the minimiser of the strictly convex problem does not depend on the start, and skipping the
  // comparison saves one factorisation of M per substep.
  FOR_LANES { for (int i = lane; i < nv; i += 64) c->qacc[i] = c->qacc_ws[i]; }
  WAVE_SYNC();
  GRX_TICK(c, GRX_P_MSOLVE);
  for (;;) {
    float* rhs;
    if (phase == 0) {
      // After a step, M a and J a - aref are current (carried) and convergence has already been decided: the only thing the evaluation would
      // still produce are the row forces, which nothing reads after the solve unless the model has touch sensors.
      const int noslip = S::kNoslip && m->noslip_iterations > 0;
      const int skip_eval = done && it > 0 && (S::kFixed ? S::NT : m->ntouch) == 0 && !noslip;
      const int changed = skip_eval ? 0 : grx_newton_eval(m, c, c->qacc, nefc, it > 0, lane_);
      GRX_TICK(c, GRX_P_NEVAL);
      // a full Newton step (alpha = 1 accepted) that did not change any row state landed on the exact minimiser of the
      // piecewise-quadratic cost: no further iteration can move it beyond rounding
      if (it > 0 && full_step && !changed) { done = 1; exact_exit = 1; }
      if (done || it >= m->iterations) {   // MuJoCo's option iterations (default 100; the hand models: 20)
#ifndef GRX_NO_OBJ_REFINE
        // only models whose free object can rest on ONE contact of the general convex routine (puck, egg, pen: a flat-on-flat or line contact stands on a single point, the
        // object block of the Hessian has a weak rocking mode and fp32 resolves the full step to ~4e-4 of its size there); a box object stands on its corner contacts and
        // the refinement changes nothing at the 1e-7 level (tools/emu_tolerances.py with -DGRX_NO_OBJ_REFINE: FetchPush / PickAndPlace identical), at 3 % of the step
        const int weak_object = (S::kFixed ? S::kConvex : (m->nconvex != 0));
#if !GRX_ON_DEVICE
        if (exact_exit && last_split == 6 && keepA && weak_object) { g_grx_newton_stats[4]++; if (last_stepmax > GRX_OBJ_REFINE_MINSTEP) g_grx_newton_stats[5]++; }
#endif
        if (exact_exit && last_split == 6 && keepA && weak_object && last_stepmax > GRX_OBJ_REFINE_MINSTEP) grx_refine_object_block(m, c, nefc, lane_);
#endif
        if (noslip) grx_noslip(m, c, nefc, lane_);   // re-solves the friction forces without regularisation: new qacc, new M a
        // converged: at the minimiser the gradient M a - qfrc_smooth - J'f vanishes, so the joint-space constraint force
        // J'f of the final evaluation is M a - qfrc_smooth (to the solver's residual) -- no further pass over the rows
        FOR_LANES { for (int i = lane; i < nv; i += 64) { c->qfrc_constraint[i] = c->Ma[i] - c->qfrc_smooth[i]; c->qacc_ws[i] = c->qacc[i]; } }
        WAVE_SYNC();
        GRX_TICK(c, GRX_P_NFINAL);
#if GRX_TWIN_STAGEHOOK
        if (g_grx_solve_mode == 1) break;
        if (do_euler) GRX_STAGE_HOOK(7);
#endif
        if (!do_euler) break;
        phase = 1;
        continue;
      }
      // Hessian of the current active set.  First iteration of a substep: assembled over all rows together with J'f of the current
      // row forces (one pass).  Later iterations: rank-1 corrections for the rows that flipped (grx_hessian_update), and the gradient is
      // ADVANCED along the accepted step instead of being re-formed: g_new = g_old + alpha H_old v is exact while no row changes state
      // (v = the step just taken, still in c->search; H_old = c->A, which the register solve leaves intact), the flipped rows add their
      // force change.  Every term is of the size of the gradient itself -- no cancellation of D |aref| |J|-sized numbers.
      int incremental = 0;
      if (it > 0 && keepA) {
        FOR_LANES {
          for (int i = lane; i < nv; i += 64) {
            float sacc = 0.0f;
#pragma unroll 8
            for (int j = 0; j < nv; j++) sacc += c->A[i * nv + j] * c->search[j];
            c->tmpv[i] = c->grad[i] + alpha_prev * sacc;
          }
        }
        WAVE_SYNC();
        incremental = grx_hessian_update(m, c, nefc, c->tmpv, lane_);
      }
      GRX_TWIN_STAT(incremental ? 3 : 2);
      if (!incremental) grx_hessian(m, c, nefc, lane_);
      GRX_TICK(c, GRX_P_NHESS);
      GRX_LANEVAR(gnp);
      if (incremental) {
        FOR_LANES {
          float part = 0;
          for (int i = lane; i < nv; i += 64) { const float sacc = c->tmpv[i]; c->grad[i] = sacc; c->search[i] = -sacc; part += sacc * sacc; }
          LV(gnp) = part;
        }
      } else {
        // gradient = M a - qfrc_smooth - J' f
        FOR_LANES {
          float part = 0;
          for (int i = lane; i < nv; i += 64) {
            const float sacc = c->Ma[i] - c->qfrc_smooth[i] - c->grad[i];
            c->grad[i] = sacc; c->search[i] = -sacc; part += sacc * sacc;
          }
          LV(gnp) = part;
        }
      }
      WAVE_SYNC();
      float gn = sqrtf(grx_reduce_sum(gnp));
      GRX_TICK(c, GRX_P_NGRAD);
#if GRX_TWIN_TRACE
      if (getenv("GRX_TRACE_NEWTON")) fprintf(stderr, "NEWTON it %d gn %.6e scale*gn %.3e incremental %d\n", it, (double)gn, (double)(scale * gn), incremental);
#endif
      if (scale * gn < 1e-8f) { done = 1; continue; }
      rhs = c->search;
    } else if (phase == 1) {
      if (!implicit_damp) {
        FOR_LANES { for (int i = lane; i < nv; i += 64) c->tmpv[i] = c->qacc[i]; }
        WAVE_SYNC();
      } else {
        FOR_LANES {
          for (int i = lane; i < nv * nv; i += 64) c->A[i] = c->M[i];
          for (int i = lane; i < nv; i += 64) c->tmpv[i] = c->qfrc_smooth[i] + c->qfrc_constraint[i];
        }
        WAVE_SYNC();
        FOR_LANES { for (int i = lane; i < nv; i += 64) c->A[i * nv + i] += h * m->dof_damping[i]; }
        WAVE_SYNC();
      }
      rhs = c->tmpv;
    } else {
      FOR_LANES { for (int i = lane; i < nv * nv; i += 64) c->A[i] = c->M[i]; }
      WAVE_SYNC();
      rhs = c->qacc_smooth;
    }
    // ---- the one linear solve
    if (!(phase == 1 && !implicit_damp)) {
      // a trailing free object (m->nfreeobj = 6): M + h B is always block diagonal; the Hessian is while no active row links object and robot
      int nsplit = 0;
      if (m->nfreeobj == 6 && (nv == 21 || nv == 30)) {
        if (phase == 0) {
          GRX_LANEVAR_I(nzp);
          FOR_LANES {
            int nz = 0;
            for (int e = lane; e < (nv - 6) * 6; e += 64) { const int i = e / 6, j = nv - 6 + (e - 6 * i); nz |= (c->A[i * nv + j] != 0.0f) | (c->A[j * nv + i] != 0.0f); }
            LV(nzp) = nz;
          }
          nsplit = (GRX_BALLOT(nzp) == 0ull) ? 6 : 0;
        } else nsplit = 6;
      }
      if (phase == 0) last_split = nsplit;
      if (grx_sym_solve_full(c->A, nv, rhs, lane_, nsplit, phase != 0)) { LANE0 { c->cnt[2] |= GRX_ST_FACTOR; } }
    }
    if (phase == 0) {
      GRX_TICK(c, GRX_P_NFACTOR);
      // Mv, Jv, quadratic coefficients of the Gauss term along the direction
      GRX_LANEVAR(q1p); GRX_LANEVAR(q2p); GRX_LANEVAR(g0p);
      FOR_LANES {
        float p1 = 0, p2 = 0, p0 = 0;
        for (int i = lane; i < nv; i += 64) {
          p0 += c->grad[i] * c->search[i];
          float sacc = 0;
#pragma unroll 8
          for (int j = 0; j < nv; j++) sacc += c->M[i * nv + j] * c->search[j];
          c->Mv[i] = sacc;
          p1 += c->search[i] * (c->Ma[i] - c->qfrc_smooth[i]); p2 += c->search[i] * sacc;
        }
        for (int r = lane; r < nefc; r += 64) {
          c->efc_jv[r] = grx_row_dot(c, r, c->search);
        }
        LV(q1p) = p1; LV(q2p) = p2; LV(g0p) = p0;
      }
      WAVE_SYNC();
      const float q1 = grx_reduce_sum(q1p), q2 = grx_reduce_sum(q2p), dphi0 = grx_reduce_sum(g0p);
      // exact line search: root of the monotone piecewise-linear derivative, starting from the Newton step
      // phi'(0) = gradient . search (exact, from the pass above); the first row pass is at the full Newton step
      float d1, d2, alpha = 1.0f, lo = 0.0f, hi = 0.0f, dlo = dphi0, dhi = 0.0f;
      const float gtol = 1e-6f * fabsf(dphi0);
      int have_hi = 0;
      const int stop = !(dphi0 < 0);
      full_step = 0;
      for (int k = 0; k < GRX_LS_MAXIT + 1 && !stop; k++) {
        int same = 0;
        grx_ls_eval(c, nefc, alpha, q1, q2, &d1, &d2, k == 0, &same, lane_);
        // The search direction is the exact Newton step of the current active set: when no row changes state on [0, 1] the cost is
        // quadratic there and alpha = 1 is its minimiser, whatever rounding left in d1 (a difference of two numbers of size |phi'(0)|).
        if (k == 0 && same) { full_step = 2; break; }
        if (fabsf(d1) <= gtol) { full_step = (k == 0); break; }
        if (d1 < 0) { lo = alpha; dlo = d1; } else { hi = alpha; dhi = d1; have_hi = 1; }
        float na = alpha - d1 / d2;
        if (have_hi) { if (!(na > lo && na < hi)) na = lo + (hi - lo) * (dlo / (dlo - dhi)); if (!(na > lo && na < hi)) na = 0.5f * (lo + hi); }
        else if (!(na > lo)) na = 2.0f * alpha;
        alpha = na;
      }
#if GRX_TWIN_TRACE
      if (getenv("GRX_TRACE_NEWTON")) fprintf(stderr, "   dphi0 %.6e stop %d alpha %.6f full_step %d\n", (double)dphi0, stop, (double)alpha, full_step);
#endif
      if (stop) { done = 1; continue; }  // not a descent direction any more: converged to rounding
      alpha_prev = alpha;
      GRX_LANEVAR(msp); GRX_LANEVAR(map_);
      FOR_LANES {
        float ms = 0, ma = 0;
        for (int i = lane; i < nv; i += 64) { float d = alpha * c->search[i]; float q = c->qacc[i] + d; c->qacc[i] = q; ms = fmaxf(ms, fabsf(d)); ma = fmaxf(ma, fabsf(q)); }
        {   // carry M a and J a - aref along the step: the next evaluation only re-derives row states and forces
          for (int i = lane; i < nv; i += 64) c->Ma[i] += alpha * c->Mv[i];
          for (int r = lane; r < nefc; r += 64) c->efc_jar[r] += alpha * c->efc_jv[r];
        }
        LV(msp) = ms; LV(map_) = ma;
      }
      WAVE_SYNC();
      const float stepmax = grx_reduce_max(msp), qmax = grx_reduce_max(map_);
      last_stepmax = stepmax;
#if GRX_TWIN_TRACE
      if (getenv("GRX_TRACE_NEWTON")) { const int d_ = atoi(getenv("GRX_TRACE_NEWTON")); fprintf(stderr, "   stepmax %.6e qmax %.4e search[d] %.6e qacc[d] %.9e grad[d] %.6e\n", (double)stepmax, (double)qmax, (double)c->search[d_], (double)c->qacc[d_], (double)c->grad[d_]); }
#endif
      LANE0 { c->cnt[6] += 1; }
      GRX_TWIN_STAT(1);
      GRX_COUNT(c, 29, 1);   // profiling build: Newton iterations of the step
#ifdef GRX_LS_STATS
      { extern int g_ls_iters, g_ls_full; g_ls_iters++; g_ls_full += full_step; }
#endif
      GRX_TICK(c, GRX_P_NLS);
      it++;
      // converged when the accepted step is below the resolution we can hold in fp32 (quadratic convergence: the
      // step just applied is ~ the error BEFORE it, the error after it is far smaller)
      if (stepmax <= GRX_NEWTON_RTOL * qmax + GRX_NEWTON_ATOL) done = 1;
      // an exact full step (no row changes state on [0,1], decided with the very arithmetic the carried evaluation would repeat) lands on
      // the minimiser of the current piece and leaves every row in its state: converged
      if (full_step == 2) { done = 1; exact_exit = 1; }
    } else if (phase == 2) {
      FOR_LANES { for (int i = lane; i < nv; i += 64) { float q = c->qacc_smooth[i]; c->qacc[i] = q; c->qacc_ws[i] = q; c->qfrc_constraint[i] = 0; } }
      WAVE_SYNC();
      if (!do_euler) break;
      phase = 1;
    } else {
      // ---- semi-implicit Euler: velocities, then positions with the new velocities
      FOR_LANES { for (int i = lane; i < nv; i += 64) c->qvel[i] += h * c->tmpv[i]; }
      WAVE_SYNC();
      FOR_LANES {
        for (int j = lane; j < GRX_NJC; j += 64) {
          int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
          if (m->jnt_type[j] == 0) {
            for (int k = 0; k < 3; k++) c->qpos[qa + k] += h * c->qvel[da + k];
            float w[3] = {c->qvel[da + 3], c->qvel[da + 4], c->qvel[da + 5]};
            float n = sqrtf(dot3f(w, w));
            if (n > 1e-12f) {
              float sn, cs; sincosf(0.5f * h * n, &sn, &cs);
              float ri = sn / n, qr[4] = {cs, w[0] * ri, w[1] * ri, w[2] * ri}, q[4] = {c->qpos[qa + 3], c->qpos[qa + 4], c->qpos[qa + 5], c->qpos[qa + 6]}, qn[4];
              mulQuatf(qn, q, qr); normalize4f(qn);
              for (int k = 0; k < 4; k++) c->qpos[qa + 3 + k] = qn[k];
            }
          } else c->qpos[qa] += h * c->qvel[da];
        }
      }
      WAVE_SYNC();
      GRX_TICK(c, GRX_P_EULER);
      break;
    }
  }
}

