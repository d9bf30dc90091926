// grx_eng_velocity.h -- K4 - K7: body velocities, bias force (RNE), passive forces, actuation.
// A FRAGMENT of csrc/grx_engine.h: textually included INSIDE `template <class S> struct GrxEngine { ... }` (every function here is a static member), in the order the engine
// header lists; not a standalone header.  The split is purely textual (round 5): the token stream of the translation units is unchanged.
// ------------------------------------------------------------------------------------------
// velocity stage: cvel, cdof_dot, RNE bias (K2/K5), passive (K6), actuation (K7)
// ------------------------------------------------------------------------------------------
GRX_MEM void grx_velocity(const GrxModel* m, GrxCtx* c, int lane_) {
  GRX_FRESH_MODEL(m, c);
  const int nv = GRX_NVC;
  // body spatial velocities = sum over the dof chain (parallel, no tree walk)
  FOR_LANES {
    for (int it = lane; it < 6 * GRX_NBC; it += 64) {
      int b = it / 6, k = it - 6 * b;
      unsigned mlo = (unsigned)m->dof_chainmask[2 * b], mhi = (unsigned)m->dof_chainmask[2 * b + 1];
      float s = 0;
#pragma unroll 8
      for (int d = 0; d < nv; d++) { unsigned bit = d < 32 ? (mlo >> d) & 1u : (mhi >> (d - 32)) & 1u; s += bit ? c->cdof[6 * d + k] * c->qvel[d] : 0.0f; }
      c->cvel[it] = s;
    }
    // passive forces
    for (int d = lane; d < nv; d += 64) {
      float f = -m->dof_damping[d] * c->qvel[d];
      int j = m->dof_jntid[d];
      if (m->jnt_stiffness[j] != 0.0f && m->jnt_type[j] >= 2) f -= m->jnt_stiffness[j] * (c->qpos[m->jnt_qposadr[j]] - m->jnt_springref[j]);
      c->qfrc_passive[d] = f;
      c->qfrc_actuator[d] = 0;
    }
  }
  WAVE_SYNC();
  FOR_LANES {
    // cdof_dot = crossMotion(velocity just before this dof, cdof).  That velocity is the spatial velocity of the body
    // owning dof_cvelstart[d], minus the dofs of that body that come after it (only multi-dof joints have any).
    for (int d = lane; d < nv; d += 64) {
      float v[6] = {0, 0, 0, 0, 0, 0}, cd[6], r[6];
      const int e0 = m->dof_cvelstart[d];
      if (e0 >= 0) {
        const int bb = m->dof_bodyid[e0], last = m->body_dofadr[bb] + m->body_dofnum[bb] - 1;
        for (int k = 0; k < 6; k++) v[k] = c->cvel[6 * bb + k];
        for (int e = e0 + 1; e <= last; e++) { float qd = c->qvel[e]; for (int k = 0; k < 6; k++) v[k] -= c->cdof[6 * e + k] * qd; }
      }
      for (int k = 0; k < 6; k++) cd[k] = c->cdof[6 * d + k];
      int jt = m->jnt_type[m->dof_jntid[d]];
      if (jt == 0 && d - m->jnt_dofadr[m->dof_jntid[d]] < 3) { for (int k = 0; k < 6; k++) r[k] = 0; }
      else crossMotionf(r, v, cd);
      for (int k = 0; k < 6; k++) c->cdof_dot[6 * d + k] = r[k];
    }
  }
  WAVE_SYNC();
  GRX_SUBTICK(c, 6);
  FOR_LANES {
    // accelerations with qacc = 0 and per-body inertial forces
    for (int b = 1 + lane; b < GRX_NBC; b += 64) {
      float a[6] = {0, 0, 0, -m->gravity[0], -m->gravity[1], -m->gravity[2]}, v[6], Ia[6], Iv[6], t[6];
      unsigned mlo = (unsigned)m->dof_chainmask[2 * b], mhi = (unsigned)m->dof_chainmask[2 * b + 1];
#pragma unroll 4
      for (int d = 0; d < nv; d++) {
        unsigned bit = d < 32 ? (mlo >> d) & 1u : (mhi >> (d - 32)) & 1u;
        float qd = bit ? c->qvel[d] : 0.0f;
        for (int k = 0; k < 6; k++) a[k] += c->cdof_dot[6 * d + k] * qd;
      }
      for (int k = 0; k < 6; k++) v[k] = c->cvel[6 * b + k];
      inertMulf(Ia, c->cinert + 10 * b, a); inertMulf(Iv, c->cinert + 10 * b, v);
      crossForcef(t, v, Iv);
      for (int k = 0; k < 6; k++) c->cacc[6 * b + k] = Ia[k] + t[k];
    }
    // actuators (one lane each; joint transmission)
    for (int i = lane; i < GRX_NUC; i += 64) {
      int j = m->act_trnid[i]; float gear = m->act_gear[i];
      float len = gear * c->qpos[m->jnt_qposadr[j]], vel = gear * c->qvel[m->jnt_dofadr[j]];
      float u = c->ctrl[i];
      if (m->act_ctrllimited[i]) u = fminf(m->act_ctrlrange[2 * i + 1], fmaxf(m->act_ctrlrange[2 * i], u));
      float gain = m->act_gainprm[3 * i];
      if (m->act_gaintype[i] == 1) gain += m->act_gainprm[3 * i + 1] * len + m->act_gainprm[3 * i + 2] * vel;
      float bias = 0;
      if (m->act_biastype[i] == 1) bias = m->act_biasprm[3 * i] + m->act_biasprm[3 * i + 1] * len + m->act_biasprm[3 * i + 2] * vel;
      float f = gain * u + bias;
      if (m->act_forcelimited[i]) f = fminf(m->act_forcerange[2 * i + 1], fmaxf(m->act_forcerange[2 * i], f));
      c->qfrc_actuator[m->jnt_dofadr[j]] = gear * f;  // models in scope have at most one actuator per dof
    }
  }
  WAVE_SYNC();
  GRX_SUBTICK(c, 7);
  FOR_LANES {
    // subtree force sums
    for (int it = lane; it < 6 * GRX_NBC; it += 64) {
      int b = it / 6, k = it - 6 * b;
      unsigned mlo = (unsigned)m->body_submask[2 * b], mhi = (S::kFixed && S::NB <= 32) ? 0u : (unsigned)m->body_submask[2 * b + 1];
      float s = 0;
#pragma unroll 16
      for (int e = 1; e < GRX_NBC; e++) { unsigned bit = e < 32 ? (mlo >> e) & 1u : (mhi >> (e - 32)) & 1u; s += bit ? c->cacc[6 * e + k] : 0.0f; }
      c->cfrc[it] = s;
    }
  }
  WAVE_SYNC();
  GRX_SUBTICK(c, 8);
  FOR_LANES {
    for (int d = lane; d < nv; d += 64) {
      float s = 0; int b = m->dof_bodyid[d];
      for (int k = 0; k < 6; k++) s += c->cdof[6 * d + k] * c->cfrc[6 * b + k];
      c->qfrc_bias[d] = s;
      float f = c->qfrc_passive[d] - s + c->qfrc_actuator[d];
      c->qfrc_smooth[d] = f; c->qacc_smooth[d] = f;
    }
  }
  WAVE_SYNC();
}

