// grx_eng_velocity.h -- K4 - K7: body velocities, bias force (RNE), passive forces, actuation.
// A FRAGMENT of csrc/grx_engine.h: textually included INSIDE `template <class S> struct GrxEngine { ... }` (every function here is a static member), in the order the engine
// header lists; not a standalone header.  The split is purely textual (round 5): the token stream of the translation units is unchanged.
// ------------------------------------------------------------------------------------------
// velocity stage: cvel, cdof_dot, RNE bias (K2/K5), passive (K6), actuation (K7)
// ------------------------------------------------------------------------------------------
GRX_MEM void grx_velocity(const GrxModel* m, GrxCtx* c, int lane_) {
  GRX_OPAQUE_STAGE(lane_);   // record addresses and lane masks of this stage are recomputed here, not carried (spilled) across the substep loop
  GRX_FRESH_MODEL(m, c);
  const int nv = GRX_NVC;
  // One lane per dof (grx_model_create refuses more than 64): everything dof d's lane needs from the model in this stage comes from ONE record (GrxModel::reci_dof / recf_dof),
  // read in one volley -- the table walk it replaces (`j = dof_jntid[d]` ... `jnt_stiffness[j]` ... `jnt_qposadr[j]`, `dof_bodyid[dof_cvelstart[d]]` ... `body_dofadr[..]`) was
  // three dependent vector-L1 round trips (profiles/cpi_r06_fetch.txt).
  GRX_LANEVAR(damp); GRX_LANEVAR(stiff); GRX_LANEVAR(sref); GRX_LANEVAR_I(jqa); GRX_LANEVAR_I(jtyp); GRX_LANEVAR_I(jda); GRX_LANEVAR_I(ve0); GRX_LANEVAR_I(vbb); GRX_LANEVAR_I(vlast); GRX_LANEVAR_I(dbody);
  FOR_LANES {
    const int d = lane < nv ? lane : 0;
    const int* DI = m->reci_dof + GRX_RDI * d; const float* DF = m->recf_dof + GRX_RDF * d;
    LV(jqa) = DI[0]; LV(jtyp) = DI[1]; LV(jda) = DI[2]; LV(ve0) = DI[3]; LV(vbb) = DI[4]; LV(vlast) = DI[5]; LV(dbody) = DI[6];
    LV(damp) = DF[0]; LV(stiff) = DF[1]; LV(sref) = DF[2];
  }
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
    if (lane < nv) {
      const int d = lane;
      float f = -LV(damp) * c->qvel[d];
      if (LV(stiff) != 0.0f && LV(jtyp) >= 2) f -= LV(stiff) * (c->qpos[LV(jqa)] - LV(sref));
      c->qfrc_passive[d] = f;
      c->qfrc_actuator[d] = 0;
    }
  }
  WAVE_SYNC();
  FOR_LANES {
    // cdof_dot = crossMotion(velocity just before this dof, cdof).  That velocity is the spatial velocity of the body
    // owning dof_cvelstart[d], minus the dofs of that body that come after it (only multi-dof joints have any).
    if (lane < nv) {
      const int d = lane;
      float v[6] = {0, 0, 0, 0, 0, 0}, cd[6], r[6];
      const int e0 = LV(ve0);
      if (e0 >= 0) {
        const int bb = LV(vbb), last = LV(vlast);
        for (int k = 0; k < 6; k++) v[k] = c->cvel[6 * bb + k];
        for (int e = e0 + 1; e <= last; e++) { float qd = c->qvel[e]; for (int k = 0; k < 6; k++) v[k] -= c->cdof[6 * e + k] * qd; }
      }
      for (int k = 0; k < 6; k++) cd[k] = c->cdof[6 * d + k];
      if (LV(jtyp) == 0 && d - LV(jda) < 3) { for (int k = 0; k < 6; k++) r[k] = 0; }
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
      // one record per actuator (GrxModel::reci_act / recf_act), every field read up front: the flags used to guard the reads of the values they select, one round trip each
      const int* AI = m->reci_act + GRX_RAI * i; const float* AF = m->recf_act + GRX_RAF * i;
      const int qadr = AI[0], dadr = AI[1], ctrllimited = AI[2], gaintype = AI[3], biastype = AI[4], forcelimited = AI[5];
      const float gear = AF[0], cr0 = AF[1], cr1 = AF[2], g0 = AF[3], g1 = AF[4], g2 = AF[5], b0 = AF[6], b1 = AF[7], b2 = AF[8], fr0 = AF[9], fr1 = AF[10];
      float len = gear * c->qpos[qadr], vel = gear * c->qvel[dadr];
      float u = c->ctrl[i];
      if (ctrllimited) u = fminf(cr1, fmaxf(cr0, u));
      float gain = g0;
      if (gaintype == 1) gain += g1 * len + g2 * vel;
      float bias = 0;
      if (biastype == 1) bias = b0 + b1 * len + b2 * vel;
      float f = gain * u + bias;
      if (forcelimited) f = fminf(fr1, fmaxf(fr0, f));
      c->qfrc_actuator[dadr] = gear * f;  // models in scope have at most one actuator per dof
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
    if (lane < nv) {
      const int d = lane;
      float s = 0; int b = LV(dbody);
      for (int k = 0; k < 6; k++) s += c->cdof[6 * d + k] * c->cfrc[6 * b + k];
      c->qfrc_bias[d] = s;
      float f = c->qfrc_passive[d] - s + c->qfrc_actuator[d];
      c->qfrc_smooth[d] = f; c->qacc_smooth[d] = f;
    }
  }
  WAVE_SYNC();
}

