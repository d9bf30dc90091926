// grx_eng_integrate.h -- K11: forward pass of one substep, position integration, RK4 stages.
// A FRAGMENT of csrc/grx_engine.h: textually included INSIDE `template <class S> struct GrxEngine { ... }` (every function here is a static member), in the order the engine
// header lists; not a standalone header.  The split is purely textual (round 5): the token stream of the translation units is unchanged.
// ------------------------------------------------------------------------------------------
// mj_forward (do_euler = 0) / mj_step (do_euler = 1) for one world
// ------------------------------------------------------------------------------------------
GRX_MEM void grx_forward_euler(const GrxModel* m, GrxCtx* c, int do_euler, int lane_) {
  GRX_TICK(c, GRX_P_OTHER);
  GRX_RNDINJ(6, (grx_rnd(c->qpos, m->nq), grx_rnd(c->qvel, m->nv), grx_rnd(c->qacc_ws, m->nv)));
  GRX_STAGE_HOOK(-1);
  grx_kinematics(m, c, lane_);
  GRX_STAGE_HOOK(0);
  GRX_RNDINJ(0, (grx_rnd(c->xpos, 3 * m->nbody), grx_rnd(c->xquat, 4 * m->nbody), grx_rnd(c->xmat, 9 * m->nbody), grx_rnd(c->sxpos, 3 * m->nsite), grx_rnd(c->sxmat, 9 * m->nsite), grx_rnd(c->janchor, 3 * m->njnt), grx_rnd(c->jaxis, 3 * m->njnt)));
  GRX_TICK(c, GRX_P_KIN);
  grx_inertia_cdof(m, c, lane_);
  GRX_STAGE_HOOK(1);
  GRX_RNDINJ(1, (grx_rnd(c->cinert, 10 * m->nbody), grx_rnd(c->cdof, 6 * m->nv), grx_rnd(c->M, m->nv * m->nv)));
  GRX_TICK(c, GRX_P_INERTIA);
  grx_collision(m, c, lane_);
  if (S::kHandoff && (c->cnt[2] & GRX_ST_HULL)) return;   // a hull pair came near in a kernel without the hull routine: the state is untouched, the caller hands the world off at this substep
  GRX_STAGE_HOOK(2);
  GRX_RNDINJ(2, (grx_rnd(c->con_dist, c->maxcon), grx_rnd(c->con_pos, 3 * c->maxcon), grx_rnd(c->con_frame, 3 * c->maxcon)));
  GRX_TICK(c, GRX_P_COLLIDE);
  grx_make_constraint(m, c, lane_);
  if (grx_handoff_due(c)) return;   // (kernels with a hand-off row only) a table capacity is exceeded: stop BEFORE the solve touches the state, the world resumes at this substep on larger tables
  GRX_STAGE_HOOK(3);
  GRX_RNDINJ(3, (grx_rnd(c->Jp, c->jpool), grx_rnd(c->efc_D, c->maxefc), grx_rnd(c->efc_aref, c->maxefc)));
  GRX_TICK(c, GRX_P_CONSTR);
  grx_velocity(m, c, lane_);
  GRX_STAGE_HOOK(4);
  GRX_RNDINJ(4, (grx_rnd(c->qfrc_smooth, m->nv), grx_rnd(c->qacc_smooth, m->nv), grx_rnd(c->efc_aref, c->maxefc)));
#if GRX_TWIN_TRACE
  grx_emu_trace(m, c, 0);   // test infrastructure (tools/emu_trace.py): contact list / rows of this pass
#endif
  grx_solve_integrate(m, c, do_euler, lane_);
  GRX_STAGE_HOOK(do_euler ? 5 : 6);
  GRX_RNDINJ(5, (grx_rnd(c->qpos, m->nq), grx_rnd(c->qvel, m->nv), grx_rnd(c->qacc_ws, m->nv)));
#if GRX_TWIN_TRACE
  grx_emu_trace(m, c, 1);
#endif
}

// qpos <- q0 (+) hh * v  (mj_integratePos semantics: quaternion exponential for free joints), one lane per joint
GRX_MEM void grx_integrate_pos(const GrxModel* m, GrxCtx* c, const float* q0, const float* v, float hh, int lane_) {
  FOR_LANES {
    for (int j = lane; j < GRX_NJC; j += 64) {
      int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
      if (m->jnt_type[j] == 0) {
        for (int k = 0; k < 3; k++) c->qpos[qa + k] = q0[qa + k] + hh * v[da + k];
        float w[3] = {v[da + 3], v[da + 4], v[da + 5]};
        float n = sqrtf(dot3f(w, w));
        float q[4] = {q0[qa + 3], q0[qa + 4], q0[qa + 5], q0[qa + 6]};
        if (n > 1e-12f) {
          float sn, cs; sincosf(0.5f * hh * n, &sn, &cs);
          float ri = sn / n, qr[4] = {cs, w[0] * ri, w[1] * ri, w[2] * ri}, qn[4];
          mulQuatf(qn, q, qr); normalize4f(qn);
          for (int k = 0; k < 4; k++) q[k] = qn[k];
        }
        for (int k = 0; k < 4; k++) c->qpos[qa + 3 + k] = q[k];
      } else c->qpos[qa] = q0[qa] + hh * v[da];
    }
  }
  WAVE_SYNC();
}

// Runge-Kutta 4 (mj_RungeKutta [3P]).  Call after the forward pass of stage `stage` (0..3): records the
// stage derivative and moves the state to the next stage point (stages 0..2) or to the end of the step (stage 3).
GRX_MEM void grx_rk4_after_forward(const GrxModel* m, GrxCtx* c, int stage, int lane_) {
  const int nv = GRX_NVC; const float h = m->timestep;
  FOR_LANES {
    if (stage == 0) {
      for (int i = lane; i < GRX_NQC; i += 64) c->rk_q0[i] = c->qpos[i];
      for (int i = lane; i < nv; i += 64) c->rk_v0[i] = c->qvel[i];
    }
    for (int i = lane; i < nv; i += 64) { c->rk_Fv[stage * nv + i] = c->qvel[i]; c->rk_Fa[stage * nv + i] = c->qacc[i]; }
  }
  WAVE_SYNC();
  const float hh = (stage < 2) ? 0.5f * h : h;
  FOR_LANES {
    for (int i = lane; i < nv; i += 64) {
      float dv, da;
      if (stage < 3) { dv = c->rk_Fv[stage * nv + i]; da = c->rk_Fa[stage * nv + i]; }
      else {
        dv = (c->rk_Fv[i] + 2.0f * c->rk_Fv[nv + i] + 2.0f * c->rk_Fv[2 * nv + i] + c->rk_Fv[3 * nv + i]) * (1.0f / 6.0f);
        da = (c->rk_Fa[i] + 2.0f * c->rk_Fa[nv + i] + 2.0f * c->rk_Fa[2 * nv + i] + c->rk_Fa[3 * nv + i]) * (1.0f / 6.0f);
      }
      c->tmpv[i] = dv;
      c->qvel[i] = c->rk_v0[i] + hh * da;
    }
  }
  WAVE_SYNC();
  grx_integrate_pos(m, c, c->rk_q0, c->tmpv, hh, lane_);
}

