// grx_eng_newton.h -- K10: Newton on the primal problem -- cost / gradient evaluation, exact line search, the Hessian on the matrix cores, incremental Hessian updates.
// A FRAGMENT of csrc/grx_engine.h: textually included INSIDE `template <class S> struct GrxEngine { ... }` (every function here is a static member), in the order the engine
// header lists; not a standalone header.  The split is purely textual (round 5): the token stream of the translation units is unchanged.
// ------------------------------------------------------------------------------------------
// K10 constraint solve: Newton on the primal problem (MuJoCo's default solver [3P]) with an
// exact line search; wave-parallel over dofs / rows / Hessian entries.
// ------------------------------------------------------------------------------------------
// Ma = M a ; jar = J a - aref ; force / active flags ; returns total cost if want_cost
// Row states: 0 = inactive (satisfied inequality), 1 = quadratic, 2 / 3 = friction-loss row saturated at -f / +f.
// Returns 1 if any row changed state with respect to the previous evaluation (bits 0-1 of efc_quad), else 0.
// carried != 0: Ma and jar were advanced along the accepted step (Ma += alpha Mv, jar += alpha Jv) by the caller, only the row states
// and forces are re-derived (no mat-vec, no row dot products).
GRX_MEM int grx_newton_eval(const GrxModel* m, GrxCtx* c, const float* a, int nefc, int carried, int lane_) {
  const int nv = GRX_NVC;
  GRX_LANEVAR(chgp);
  FOR_LANES {
    float chg = 0;
    if (!carried) {
      for (int i = lane; i < nv; i += 64) {
        float s = 0;
#pragma unroll 8
        for (int j = 0; j < nv; j++) s += c->M[i * nv + j] * a[j];
        c->Ma[i] = s;
      }
    }
    for (int r = lane; r < nefc; r += 64) {
      float x = carried ? c->efc_jar[r] : (float)(grx_row_dot(c, r, a) - c->efc_aref[r]), D = c->efc_D[r], f; int st;
      int kind = c->efc_kind[r];
      if (kind == GRX_ROW_EQ) { f = -D * x; st = 1; }
      else if (kind == GRX_ROW_FRICTION) {
        float fl = c->efc_floss[r], Rf = fl / D;
        if (x <= -Rf) { f = fl; st = 3; } else if (x >= Rf) { f = -fl; st = 2; } else { f = -D * x; st = 1; }
      } else {
        if (x < 0) { f = -D * x; st = 1; } else { f = 0; st = 0; }
      }
      const int old = c->efc_quad[r];
      if (st != (old & 3)) chg = 1.0f;
      c->efc_jar[r] = x; c->efc_force[r] = f; c->efc_quad[r] = (old & 0x30) | st;   // bits 4-5: the state this row has in the assembled Hessian
    }
    LV(chgp) = chg;
  }
  WAVE_SYNC();
  return grx_reduce_max(chgp) > 0.5f;
}

// derivative (d1) and curvature (d2) of the cost along the search direction at step alpha
// *same <- (want_same and) every row is, at step alpha, in the state it has at alpha = 0 (efc_quad): the cost is then exactly
// quadratic on [0, alpha]
GRX_MEM void grx_ls_eval(GrxCtx* c, int nefc, float alpha, float q1, float q2, float* d1, float* d2, int want_same, int* same, int lane_) {
#ifdef GRX_LS_STATS
  { extern int g_ls_calls; g_ls_calls++; }
#endif
  GRX_LANEVAR(gp); GRX_LANEVAR(hp); GRX_LANEVAR_I(difp);
  FOR_LANES {
    float g = 0, h = 0; int dif = 0;
    for (int r = lane; r < nefc; r += 64) {
      float jv = c->efc_jv[r], D = c->efc_D[r], x = c->efc_jar[r] + alpha * jv;
      int kind = c->efc_kind[r], st;
      if (kind == GRX_ROW_EQ) { g += D * x * jv; h += D * jv * jv; st = 1; }
      else if (kind == GRX_ROW_FRICTION) {
        float fl = c->efc_floss[r], Rf = fl / D;
        if (x <= -Rf) { g -= fl * jv; st = 3; } else if (x >= Rf) { g += fl * jv; st = 2; } else { g += D * x * jv; h += D * jv * jv; st = 1; }
      } else if (x < 0) { g += D * x * jv; h += D * jv * jv; st = 1; } else st = 0;
      if (want_same) dif |= (st != (c->efc_quad[r] & 3));
    }
    LV(gp) = g; LV(hp) = h; LV(difp) = dif;
  }
  float g = grx_reduce_sum(gp), h = grx_reduce_sum(hp);
  *d1 = q1 + alpha * q2 + g; *d2 = q2 + h;
  *same = want_same && (GRX_BALLOT(difp) == 0ull);
}

// H = M + J' diag(D_active) J  ->  c->A   (efc_jv is used as scratch for the masked D)
// Also returns J' f (the constraint force in joint space for the row forces of the last evaluation) in c->grad: on the
// matrix-core path it rides along as one extra output column of the same MFMA chain.
GRX_MEM void grx_hessian(const GrxModel* m, GrxCtx* c, int nefc, int lane_) {
  GRX_OPAQUE_STAGE(lane_);
  const int nv = GRX_NVC;
    // Hessian H = M + J' diag(D_active) J
  FOR_LANES {
    for (int r = lane; r < nefc; r += 64) {
      const int st = c->efc_quad[r] & 3;
      c->efc_jv[r] = (st == 1) ? c->efc_D[r] : 0.0f;   // efc_jv reused as scratch
      c->efc_quad[r] = st | (st << 4);                  // the Hessian now represents this row in state st (grx_hessian_update)
    }
  }
  WAVE_SYNC();
#if GRX_ON_DEVICE
  if (nv < 32 || (S::kFixed && S::NV <= 40)) {
    // matrix cores: [H | J'f] = J' [D J | f] as a chain of v_mfma_f32_32x32x2_f32 (exact f32, two constraint rows per instruction).
    // nv >= 32 (Adroit: 33): the chain forms the leading 32 x 32 block; the remaining rows / columns and J'f follow in a lane-per-dof pass.
    const int nvm = nv < 32 ? nv : 32;
    // operand maps: A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]; C: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5)
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.0f;
    const int idx = lane_ & 31, half = lane_ >> 5;
    const bool incol = idx < nvm;
    // Branch-free operand fetch (clamped addresses, selects instead of divergent paths), software-pipelined by hand over
    // four row pairs: 16 independent LDS reads (row info, second span, masked D, force), then 4 reads of the packed Jacobian, then 4 MFMAs.
    const bool isf = (idx == nv);   // the spare column carries J'f (only when nv < 32)
    for (int r0 = 0; r0 < nefc; r0 += 8) {
      int info[4], idb[4]; float dq[4], fr[4], v[4]; bool rowok[4], in[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int row = r0 + 2 * u + half;
        rowok[u] = row < nefc;
        const int rr = rowok[u] ? row : 0;
        info[u] = c->efc_row[rr]; idb[u] = S::kTwoSpan ? c->efc_id[rr] : 0; dq[u] = c->efc_jv[rr]; fr[u] = c->efc_force[rr];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int pos = grx_row_pos(info[u], idb[u], idx);
        in[u] = rowok[u] && incol && pos >= 0;
        v[u] = c->Jp[GRX_ROW_OFF(info[u]) + (in[u] ? pos : 0)];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const float a = in[u] ? v[u] : 0.0f;                                        // A[i = idx][k = row] = J[row][idx]
        const float b = in[u] ? v[u] * dq[u] : ((isf && rowok[u]) ? fr[u] : 0.0f);  // B[k = row][j = idx] = D J[row][idx]; column nv: f[row]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int i = (e & 3) + 8 * (e >> 2) + 4 * half;
      if (i < nvm) {
        if (incol) c->A[i * nv + idx] = c->M[i * nv + idx] + acc[e];
        else if (idx == nv) c->grad[i] = acc[e];
      }
    }
    if (nv >= 32) {   // rows / columns 32 .. nv-1 of H and the whole of J'f: lane j = dof j, one pass over the rows per extra dof
      if (lane_ < nv) {
        const int j = lane_;
        float gj = 0.0f;
        for (int r = 0; r < nefc; r++) {
          const int info = c->efc_row[r], idb = S::kTwoSpan ? c->efc_id[r] : 0, pj = grx_row_pos(info, idb, j);
          if (pj >= 0) gj += c->Jp[GRX_ROW_OFF(info) + pj] * c->efc_force[r];
        }
        c->grad[j] = gj;
        for (int i = 32; i < nv; i++) {
          float hij = 0.0f;
          for (int r = 0; r < nefc; r++) {
            const float dq = c->efc_jv[r];
            if (dq == 0.0f) continue;
            const int info = c->efc_row[r], idb = S::kTwoSpan ? c->efc_id[r] : 0, pi = grx_row_pos(info, idb, i), pj = grx_row_pos(info, idb, j);
            if (pi >= 0 && pj >= 0) hij += dq * c->Jp[GRX_ROW_OFF(info) + pi] * c->Jp[GRX_ROW_OFF(info) + pj];
          }
          const float v = c->M[i * nv + j] + hij;
          c->A[i * nv + j] = v; c->A[j * nv + i] = v;
        }
      }
    }
    __syncthreads();
  } else
#else
  // Emulator twin of the matrix-core path above: the SAME operand maps and the SAME result layout, fed through a scalar restatement of the instruction
  // (grx_emu_mfma_32x32x2, csrc/grx_engine.h), so that the CPU suite exercises the index arithmetic the device runs -- which lane supplies which J entry, which accumulator
  // register is which row of H, the spare column that carries J'f, the lane-per-dof pass for dofs 32 .. nv-1 -- instead of only the 3 x 3 register-tile path below.
  if (nv <= 40) {
    const int nvm = nv < 32 ? nv : 32;
    float acc[64][16];
    for (int l = 0; l < 64; l++) for (int e = 0; e < 16; e++) acc[l][e] = 0.0f;
    for (int r0 = 0; r0 < nefc; r0 += 8)
      for (int u = 0; u < 4; u++) {
        float a[64], b[64];
        for (int l = 0; l < 64; l++) {
          const int idx = l & 31, half = l >> 5;
          const bool incol = idx < nvm, isf = (idx == nv);
          const int row = r0 + 2 * u + half;
          const bool rowok = row < nefc;
          const int rr = rowok ? row : 0;
          const int info = c->efc_row[rr], idb = S::kTwoSpan ? c->efc_id[rr] : 0;
          const float dq = c->efc_jv[rr], fr = c->efc_force[rr];
          const int pos = grx_row_pos(info, idb, idx);
          const bool in = rowok && incol && pos >= 0;
          const float v = c->Jp[GRX_ROW_OFF(info) + (in ? pos : 0)];
          a[l] = in ? v : 0.0f;                                        // A[i = idx][k = row] = J[row][idx]
          b[l] = in ? v * dq : ((isf && rowok) ? fr : 0.0f);           // B[k = row][j = idx] = D J[row][idx]; column nv: f[row]
        }
        grx_emu_mfma_32x32x2(a, b, acc);
      }
    for (int l = 0; l < 64; l++) {
      const int idx = l & 31, half = l >> 5;
      const bool incol = idx < nvm;
      for (int e = 0; e < 16; e++) {
        const int i = (e & 3) + 8 * (e >> 2) + 4 * half;
        if (i < nvm) {
          if (incol) c->A[i * nv + idx] = c->M[i * nv + idx] + acc[l][e];
          else if (idx == nv) c->grad[i] = acc[l][e];
        }
      }
    }
    if (nv >= 32) {   // rows / columns 32 .. nv-1 of H and the whole of J'f: lane j = dof j, one pass over the rows per extra dof (the device's code, lane by lane)
      for (int j = 0; j < nv; j++) {
        float gj = 0.0f;
        for (int r = 0; r < nefc; r++) {
          const int info = c->efc_row[r], idb = S::kTwoSpan ? c->efc_id[r] : 0, pj = grx_row_pos(info, idb, j);
          if (pj >= 0) gj += c->Jp[GRX_ROW_OFF(info) + pj] * c->efc_force[r];
        }
        c->grad[j] = gj;
        for (int i = 32; i < nv; i++) {
          float hij = 0.0f;
          for (int r = 0; r < nefc; r++) {
            const float dq = c->efc_jv[r];
            if (dq == 0.0f) continue;
            const int info = c->efc_row[r], idb = S::kTwoSpan ? c->efc_id[r] : 0, pi = grx_row_pos(info, idb, i), pj = grx_row_pos(info, idb, j);
            if (pi >= 0 && pj >= 0) hij += dq * c->Jp[GRX_ROW_OFF(info) + pi] * c->Jp[GRX_ROW_OFF(info) + pj];
          }
          const float v = c->M[i * nv + j] + hij;
          c->A[i * nv + j] = v; c->A[j * nv + i] = v;
        }
      }
    }
  } else
#endif
  {
  FOR_LANES {
    const int li = lane >> 3, lj = lane & 7;
    for (int i0 = li; i0 < nv; i0 += 24)
      for (int j0 = lj; j0 < nv && j0 <= i0 + 16; j0 += 24) {
        // 3x3 register tile: rows i0, i0+8, i0+16 ; cols j0, j0+8, j0+16
        float acc[3][3];
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) acc[a][b] = 0.0f;
        const int i1 = i0 + 8, i2 = i0 + 16, j1 = j0 + 8, j2 = j0 + 16;
        const int vi1 = i1 < nv, vi2 = i2 < nv, vj1 = j1 < nv, vj2 = j2 < nv;
        for (int r = 0; r < nefc; r++) {
          const int info = c->efc_row[r], idb = c->efc_id[r], off = GRX_ROW_OFF(info);
#define GRX_JAT(dof) (grx_row_pos(info, idb, (dof)) >= 0 ? c->Jp[off + grx_row_pos(info, idb, (dof))] : 0.0f)
          float d = c->efc_jv[r];
          float a0 = GRX_JAT(i0) * d, a1 = vi1 ? GRX_JAT(i1) * d : 0.0f, a2 = vi2 ? GRX_JAT(i2) * d : 0.0f;
          float b0 = GRX_JAT(j0), b1 = vj1 ? GRX_JAT(j1) : 0.0f, b2 = vj2 ? GRX_JAT(j2) : 0.0f;
#undef GRX_JAT
          acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[0][2] += a0 * b2;
          acc[1][0] += a1 * b0; acc[1][1] += a1 * b1; acc[1][2] += a1 * b2;
          acc[2][0] += a2 * b0; acc[2][1] += a2 * b1; acc[2][2] += a2 * b2;
        }
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++) {
            int i = i0 + 8 * a, j = j0 + 8 * b;
            if (i < nv && j < nv && j <= i) { float v = c->M[i * nv + j] + acc[a][b]; c->A[i * nv + j] = v; c->A[j * nv + i] = v; }
          }
      }
    for (int i = lane; i < nv; i += 64) {
      float sacc = 0;
      for (int r = 0; r < nefc; r++) { const int info = c->efc_row[r], pos = grx_row_pos(info, c->efc_id[r], i); if (pos >= 0) sacc += c->Jp[GRX_ROW_OFF(info) + pos] * c->efc_force[r]; }
      c->grad[i] = sacc;
    }
  }
  WAVE_SYNC();
  }
}

// Incremental Hessian.  A row contributes to the problem through its state: force = -d (J a - aref) + kf, with d = D in the quadratic
// state, (d, kf) = (0, -+floss) for a saturated friction-loss row and (0, 0) when inactive; H = M + sum d J'J.  Between two Newton
// iterations of one substep only the rows whose state flipped change d: apply their rank-1 corrections to A instead of re-assembling H
// over all rows (one flip is the common case; the iterations after the first are what separates an expensive world from a cheap one).
// The GRADIENT is advanced the same way (the caller has put g_old + alpha H_old v into `gnew`: exact while no row changes state); a row
// that flipped adds J_r' (f_new - f_old-state(jar_new)) = J_r' (-(d_new - d_old) jar_new + (kf_new - kf_old)), all of it small near the
// solution.  (Round 3 formed the gradient as H a - qfrc_smooth - sum k J with k = D aref: terms of size D |aref| |J| ~ 400 cancelling to
// 1e-6 -- in fp32 a noise of 2e-5 on the puck's angular dof, whose Hessian entry is 6e-4: an acceleration error of 3e-2 rad/s^2 per
// substep, the whole FetchSlide rotation-velocity discrepancy; tools/emu_mixed.py, tools/emu_trace.py.)  Returns 0 when more than
// GRX_HUPD_MAX rows flipped (the caller re-assembles).  Uses c->ired (row list) and c->Mv (the row, expanded) as scratch.
#define GRX_HUPD_MAX 8
GRX_MEM int grx_hessian_update(const GrxModel* m, GrxCtx* c, int nefc, float* gnew, int lane_) {
  const int nv = GRX_NVC;
  int* list = c->ired;
  int nd = 0;
  for (int base = 0; base < nefc; base += 64) {
    GRX_LANEVAR_I(dirty);
    FOR_LANES { const int r = base + lane; const int q = r < nefc ? c->efc_quad[r] : 0; LV(dirty) = (r < nefc) && ((q & 3) != ((q >> 4) & 3)); }
    const unsigned long long bm = GRX_BALLOT(dirty);
    FOR_LANES { if (LV(dirty)) { const int k = nd + __builtin_popcountll(bm & ((1ull << lane) - 1ull)); if (k < GRX_HUPD_MAX) list[k] = base + lane; } }
    nd += __builtin_popcountll(bm);
  }
  WAVE_SYNC();
  if (nd > GRX_HUPD_MAX) return 0;
  for (int e = 0; e < nd; e++) {
    const int r = list[e];
    const int q = c->efc_quad[r], st = q & 3, hs = (q >> 4) & 3, info = c->efc_row[r], idb = S::kTwoSpan ? c->efc_id[r] : 0;
    const float D = c->efc_D[r];
    const float fl = (st >= 2 || hs >= 2) ? c->efc_floss[r] : 0.0f;
    const float dd = (st == 1 ? D : 0.0f) - (hs == 1 ? D : 0.0f);
    // change of the row force at the current point: -(d_new - d_old) jar + (kf_new - kf_old)
    const float df = -dd * c->efc_jar[r] + ((st == 2 ? -fl : (st == 3 ? fl : 0.0f)) - (hs == 2 ? -fl : (hs == 3 ? fl : 0.0f)));
    FOR_LANES { for (int i = lane; i < nv; i += 64) { const int pos = grx_row_pos(info, idb, i); c->Mv[i] = pos >= 0 ? c->Jp[GRX_ROW_OFF(info) + pos] : 0.0f; } }
    WAVE_SYNC();
    FOR_LANES {
      for (int i = lane; i < nv; i += 64) {
        const float vi = c->Mv[i];
        if (vi != 0.0f) {
          const float s_ = dd * vi;
          for (int j = 0; j < nv; j++) c->A[i * nv + j] += s_ * c->Mv[j];
          gnew[i] -= df * vi;   // gradient = M a - qfrc_smooth - J'f
        }
      }
    }
    WAVE_SYNC();
    LANE0 { c->efc_quad[r] = st | (st << 4); }
  }
  WAVE_SYNC();
  return 1;
}

