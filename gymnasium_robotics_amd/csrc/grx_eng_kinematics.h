// grx_eng_kinematics.h -- K1 - K3: kinematics (pointer jumping), spatial inertias, motion axes, composite inertias, the mass matrix.
// A FRAGMENT of csrc/grx_engine.h: textually included INSIDE `template <class S> struct GrxEngine { ... }` (every function here is a static member), in the order the engine
// header lists; not a standalone header.  The split is purely textual (round 5): the token stream of the translation units is unchanged.
// ------------------------------------------------------------------------------------------
// K1 forward kinematics
// ------------------------------------------------------------------------------------------
GRX_MEM void grx_kinematics(const GrxModel* m, GrxCtx* c, int lane_) {
  GRX_OPAQUE_STAGE(lane_);   // record addresses and lane masks of this stage are recomputed here, not carried (spilled) across the substep loop
  GRX_FRESH_MODEL(m, c);
  // One lane per body (grx_model_create refuses more than 64).  Everything the body's lane needs from the model in this stage -- its own pose, its first joint, the
  // pointer-jumping schedule of the rounds below, whether its quaternion is re-normalised at the end -- comes from ONE record (GrxModel::reci_body / recf_body), read in one
  // volley at the top: the table walk it replaces (body_mocapid -> body_jntadr / body_jntnum -> jnt_type / jnt_qposadr -> body_pos / body_quat -> jnt_pos / jnt_axis ->
  // qpos0[qposadr], then body_jump once per round) was a chain of ~10 dependent vector-L1 round trips per substep (profiles/cpi_r06_fetch.txt).
  GRX_LANEVAR_I(bj0); GRX_LANEVAR_I(bj1); GRX_LANEVAR_I(bj2); GRX_LANEVAR_I(bj3); GRX_LANEVAR_I(bj4); GRX_LANEVAR_I(bj5); GRX_LANEVAR_I(bj6); GRX_LANEVAR_I(bj7); GRX_LANEVAR_I(bfl);
  FOR_LANES {
    LV(bj0) = LV(bj1) = LV(bj2) = LV(bj3) = LV(bj4) = LV(bj5) = LV(bj6) = LV(bj7) = 0; LV(bfl) = 0;
    if (lane < GRX_NBC) {
      const int b = lane;
      const int* BI = m->reci_body + GRX_RBI * b; const float* BF = m->recf_body + GRX_RBF * b;
      const int mid = BI[0], jn = BI[2], jt0 = BI[3], qa0 = BI[4], bflags = BI[5], jtf = BI[6]; const unsigned ja = (unsigned)BI[1];
      LV(bj0) = BI[8]; LV(bj1) = BI[9]; LV(bj2) = BI[10]; LV(bj3) = BI[11]; LV(bj4) = BI[12]; LV(bj5) = BI[13]; LV(bj6) = BI[14]; LV(bj7) = BI[15]; LV(bfl) = bflags;
      float p[3] = {BF[0], BF[1], BF[2]}, q[4] = {BF[4], BF[5], BF[6], BF[7]};
      const float q00 = BF[3], jp0[3] = {BF[8], BF[9], BF[10]}, jx0[3] = {BF[12], BF[13], BF[14]};
      float* pl = c->ploc + 3 * b; float* ql = c->qloc + 4 * b;
      if (b == 0) {
        c->xpos[0] = c->xpos[1] = c->xpos[2] = 0; c->xquat[0] = 1; c->xquat[1] = c->xquat[2] = c->xquat[3] = 0;
        for (int k = 0; k < 9; k++) c->xmat[k] = (k % 4 == 0) ? 1.0f : 0.0f;
      } else if (mid >= 0) {
        float qm[4] = {c->mocap_quat[4 * mid], c->mocap_quat[4 * mid + 1], c->mocap_quat[4 * mid + 2], c->mocap_quat[4 * mid + 3]};
        normalize4f(qm);
        pl[0] = c->mocap_pos[3 * mid]; pl[1] = c->mocap_pos[3 * mid + 1]; pl[2] = c->mocap_pos[3 * mid + 2];
        ql[0] = qm[0]; ql[1] = qm[1]; ql[2] = qm[2]; ql[3] = qm[3];
      } else if (jt0 == 0) {  // free joint: qpos is the world pose
        const int qa = qa0;
        float qf[4] = {c->qpos[qa + 3], c->qpos[qa + 4], c->qpos[qa + 5], c->qpos[qa + 6]};
        if (!c->resume_first) normalize4f(qf);   // (a world resumed mid-step: this substep's normalisation was done, in place, by the kernel that handed it off; doing it twice is not bit-neutral)
        pl[0] = c->qpos[qa]; pl[1] = c->qpos[qa + 1]; pl[2] = c->qpos[qa + 2];
        for (int k = 0; k < 4; k++) { ql[k] = qf[k]; c->qpos[qa + 3 + k] = qf[k]; }
        for (int k = 0; k < 3; k++) { c->janchor[3 * ja + k] = pl[k]; c->jaxis[3 * ja + k] = (k == 2) ? 1.0f : 0.0f; }
      } else {
        if (S::kShift && m->nshift && (bflags & 1)) { p[0] += c->shift[0]; p[1] += c->shift[1]; p[2] += c->shift[2]; }   // child of the (world-fixed) shift group
        for (int k = 0; k < jn; k++) {
          const unsigned j = ja + (unsigned)k;
          int qa = qa0, jt = jtf; float jp[3] = {jp0[0], jp0[1], jp0[2]}, jx[3] = {jx0[0], jx0[1], jx0[2]}, q0 = q00;
          if (k > 0) {   // further joints of a multi-joint body (the Fetch base's three slides): one joint record each
            const int* JI = m->reci_jnt + GRX_RJI * j; const float* JF = m->recf_jnt + GRX_RJF * j;
            qa = JI[0]; jt = JI[1]; q0 = JF[3];
            jp[0] = JF[0]; jp[1] = JF[1]; jp[2] = JF[2]; jx[0] = JF[4]; jx[1] = JF[5]; jx[2] = JF[6];
          }
          float anchor[3], axis[3];
          rotVecQuatf(anchor, jp, q); anchor[0] += p[0]; anchor[1] += p[1]; anchor[2] += p[2];
          rotVecQuatf(axis, jx, q);
          for (int t = 0; t < 3; t++) { c->janchor[3 * j + t] = anchor[t]; c->jaxis[3 * j + t] = axis[t]; }
          float dq = c->qpos[qa] - q0;
          if (jt == 2) {
            p[0] += axis[0] * dq; p[1] += axis[1] * dq; p[2] += axis[2] * dq;
          } else if (jt == 3) {
            float sn, cs; sincosf(0.5f * dq, &sn, &cs);
            float qr[4] = {cs, jx[0] * sn, jx[1] * sn, jx[2] * sn}, qn[4], off[3];
            mulQuatf(qn, q, qr); normalize4f(qn);
            q[0] = qn[0]; q[1] = qn[1]; q[2] = qn[2]; q[3] = qn[3];
            rotVecQuatf(off, jp, q);
            p[0] = anchor[0] - off[0]; p[1] = anchor[1] - off[1]; p[2] = anchor[2] - off[2];
          }
        }
        pl[0] = p[0]; pl[1] = p[1]; pl[2] = p[2]; ql[0] = q[0]; ql[1] = q[1]; ql[2] = q[2]; ql[3] = q[3];
      }
    }
  }
  WAVE_SYNC();
  GRX_SUBTICK(c, 9);
  // world poses by pointer jumping: in round s every body composes its pose (relative to the ancestor 2^s levels up) with
  // that ancestor's pose (relative to ITS ancestor 2^s levels up): ceil(log2(depth)) rounds instead of one composition per
  // ancestor.  Rounds ping-pong between {ploc,qloc} and {xpos,xquat}; body_jump is the static schedule (in the body record: at most 8 rounds, 64 bodies deep).
  const int nbk = GRX_NBC, nj = m->njump;
  for (int s = 0; s < nj; s++) {
    const float* sp = (s & 1) ? c->xpos : c->ploc; const float* sq = (s & 1) ? c->xquat : c->qloc;
    float* dp = (s & 1) ? c->ploc : c->xpos; float* dq = (s & 1) ? c->qloc : c->xquat;
    FOR_LANES {
      if (lane >= 1 && lane < nbk) {
        const int b = lane;
        float p[3] = {sp[3 * b], sp[3 * b + 1], sp[3 * b + 2]}, q[4] = {sq[4 * b], sq[4 * b + 1], sq[4 * b + 2], sq[4 * b + 3]};
        const int anc = s == 0 ? LV(bj0) : (s == 1 ? LV(bj1) : (s == 2 ? LV(bj2) : (s == 3 ? LV(bj3) : (s == 4 ? LV(bj4) : (s == 5 ? LV(bj5) : (s == 6 ? LV(bj6) : LV(bj7)))))));
        if (anc > 0) {
          float qa[4] = {sq[4 * anc], sq[4 * anc + 1], sq[4 * anc + 2], sq[4 * anc + 3]}, v[3], qn[4];
          rotVecQuatf(v, p, qa);
          p[0] = sp[3 * anc] + v[0]; p[1] = sp[3 * anc + 1] + v[1]; p[2] = sp[3 * anc + 2] + v[2];
          mulQuatf(qn, qa, q);
          q[0] = qn[0]; q[1] = qn[1]; q[2] = qn[2]; q[3] = qn[3];
        }
        for (int e = 0; e < 3; e++) dp[3 * b + e] = p[e];
        for (int e = 0; e < 4; e++) dq[4 * b + e] = q[e];
      }
    }
    WAVE_SYNC();
  }
  {
    const float* sp = (nj & 1) ? c->xpos : c->ploc; const float* sq = (nj & 1) ? c->xquat : c->qloc;
    FOR_LANES {
      if (lane >= 1 && lane < nbk) {
        const int b = lane;
        float p[3] = {sp[3 * b], sp[3 * b + 1], sp[3 * b + 2]}, q[4] = {sq[4 * b], sq[4 * b + 1], sq[4 * b + 2], sq[4 * b + 3]};
        if (!(LV(bfl) & 2)) normalize4f(q);   // (mocap and free bodies carry their quaternion as given)
        float R[9]; quat2matf(R, q);
        for (int e = 0; e < 3; e++) c->xpos[3 * b + e] = p[e];
        for (int e = 0; e < 4; e++) c->xquat[4 * b + e] = q[e];
        for (int e = 0; e < 9; e++) c->xmat[9 * b + e] = R[e];
      }
    }
  }
  WAVE_SYNC();
  GRX_SUBTICK(c, 10);
  FOR_LANES {
    // joint anchors / axes to the world frame (they were expressed in the parent frame; free joints already are world)
    for (int j = lane; j < GRX_NJC; j += 64) {
      const int* JI = m->reci_jnt + GRX_RJI * j;   // joint record: type and the parent of the joint's body in one read
      const int jt = JI[1], par = JI[4];
      if (jt == 0) continue;
      float a_[3] = {c->janchor[3 * j], c->janchor[3 * j + 1], c->janchor[3 * j + 2]}, x_[3] = {c->jaxis[3 * j], c->jaxis[3 * j + 1], c->jaxis[3 * j + 2]}, ta[3], tx[3];
      mulMatVec3f(ta, c->xmat + 9 * par, a_); mulMatVec3f(tx, c->xmat + 9 * par, x_);
      for (int e = 0; e < 3; e++) { c->janchor[3 * j + e] = ta[e] + c->xpos[3 * par + e]; c->jaxis[3 * j + e] = tx[e]; }
    }
  }
  FOR_LANES {
    for (int i = lane; i < GRX_NSC; i += 64) {
      int b = m->site_bodyid[i];
      float lpv[3] = {m->site_pos[3 * i], m->site_pos[3 * i + 1], m->site_pos[3 * i + 2]}, lqv[4] = {m->site_quat[4 * i], m->site_quat[4 * i + 1], m->site_quat[4 * i + 2], m->site_quat[4 * i + 3]}, v[3], R[9], Rw[9];
      mulMatVec3f(v, c->xmat + 9 * b, lpv);
      const int sh = (S::kShift && m->nshift) ? m->site_shift[i] : 0;
      for (int e = 0; e < 3; e++) v[e] += c->xpos[3 * b + e];
      quat2matf(R, lqv); mulMat3f(Rw, c->xmat + 9 * b, R);
      if (S::kShiftRot && sh == 2) grx_apply_group_rotation(c->shift + 3, v, Rw);
      for (int e = 0; e < 3; e++) c->sxpos[3 * i + e] = v[e] + (sh ? c->shift[e] : 0.0f);
      for (int e = 0; e < 9; e++) c->sxmat[9 * i + e] = Rw[e];
    }
  }
  WAVE_SYNC();
}

// ------------------------------------------------------------------------------------------
// K2/K3 spatial inertias, motion axes, composite inertias, mass matrix
// reference point of each kinematic tree = xpos of its root body (any point is valid)
// ------------------------------------------------------------------------------------------
GRX_MEM void grx_inertia_cdof(const GrxModel* m, GrxCtx* c, int lane_) {
  GRX_OPAQUE_STAGE(lane_);   // record addresses and lane masks of this stage are recomputed here, not carried (spilled) across the substep loop
  GRX_FRESH_MODEL(m, c);
  FOR_LANES {
    for (int b = 1 + lane; b < GRX_NBC; b += 64) {
      const float* R = c->xmat + 9 * b; const float* in = m->body_inertia + 6 * b;
      const float* cref = c->xpos + 3 * m->body_rootid[b];
      float ip[3] = {m->body_ipos[3 * b], m->body_ipos[3 * b + 1], m->body_ipos[3 * b + 2]}, r[3];
      mulMatVec3f(r, R, ip);
      r[0] += c->xpos[3 * b] - cref[0]; r[1] += c->xpos[3 * b + 1] - cref[1]; r[2] += c->xpos[3 * b + 2] - cref[2];
      float Ib[9] = {in[0], in[3], in[4], in[3], in[1], in[5], in[4], in[5], in[2]}, t[9], Rt[9], Iw[9];
      for (int a = 0; a < 3; a++) for (int e = 0; e < 3; e++) Rt[3 * a + e] = R[3 * e + a];
      mulMat3f(t, R, Ib); mulMat3f(Iw, t, Rt);
      float mass = m->body_mass[b], rr = dot3f(r, r);
      float* I = c->cinert + 10 * b;
      I[0] = Iw[0] + mass * (rr - r[0] * r[0]); I[1] = Iw[4] + mass * (rr - r[1] * r[1]); I[2] = Iw[8] + mass * (rr - r[2] * r[2]);
      I[3] = Iw[1] - mass * r[0] * r[1]; I[4] = Iw[2] - mass * r[0] * r[2]; I[5] = Iw[5] - mass * r[1] * r[2];
      I[6] = mass * r[0]; I[7] = mass * r[1]; I[8] = mass * r[2]; I[9] = mass;
    }
    for (int j = lane; j < GRX_NJC; j += 64) {
      const int* JI = m->reci_jnt + GRX_RJI * j;   // joint record: body, dof address, type and the root of the body's tree in one read
      const int b = JI[2], da = JI[3], jt = JI[1];
      const float* cref = c->xpos + 3 * JI[5];
      float off[3] = {cref[0] - c->janchor[3 * j], cref[1] - c->janchor[3 * j + 1], cref[2] - c->janchor[3 * j + 2]};
      const float* ax = c->jaxis + 3 * j;
      if (jt == 2) {
        float* cd = c->cdof + 6 * da; cd[0] = cd[1] = cd[2] = 0; cd[3] = ax[0]; cd[4] = ax[1]; cd[5] = ax[2];
      } else if (jt == 3) {
        float* cd = c->cdof + 6 * da; float axv[3] = {ax[0], ax[1], ax[2]}, t[3];
        cross3f(t, axv, off);
        cd[0] = axv[0]; cd[1] = axv[1]; cd[2] = axv[2]; cd[3] = t[0]; cd[4] = t[1]; cd[5] = t[2];
      } else if (jt == 0) {
        for (int k = 0; k < 3; k++) { float* cd = c->cdof + 6 * (da + k); for (int e = 0; e < 6; e++) cd[e] = (e == 3 + k) ? 1.0f : 0.0f; }
        for (int k = 0; k < 3; k++) {
          float* cd = c->cdof + 6 * (da + 3 + k);
          float axv[3] = {c->xmat[9 * b + k], c->xmat[9 * b + 3 + k], c->xmat[9 * b + 6 + k]}, t[3];
          cross3f(t, axv, off);  // off = cref - xpos(body) = 0 for a root free body
          cd[0] = axv[0]; cd[1] = axv[1]; cd[2] = axv[2]; cd[3] = t[0]; cd[4] = t[1]; cd[5] = t[2];
        }
      }
    }
  }
  WAVE_SYNC();
  GRX_SUBTICK(c, 14);
  // composite inertia = sum over the subtree (no serial tree walk; membership from the static 64-bit subtree masks)
  FOR_LANES {
    for (int it = lane; it < 10 * GRX_NBC; it += 64) {
      int b = it / 10, k = it - 10 * b;
      unsigned mlo = (unsigned)m->body_submask[2 * b], mhi = (S::kFixed && S::NB <= 32) ? 0u : (unsigned)m->body_submask[2 * b + 1];
      float s = 0;
#pragma unroll 16
      for (int e = 1; e < GRX_NBC; e++) { unsigned bit = e < 32 ? (mlo >> e) & 1u : (mhi >> (e - 32)) & 1u; s += bit ? c->cinert[10 * e + k] : 0.0f; }
      c->crb[it] = s;
    }
  }
  WAVE_SYNC();
  GRX_SUBTICK(c, 15);
  FOR_LANES {
    for (int e = lane; e < m->nmpair; e += 64) {
      const int* MI = m->reci_mpair + GRX_RMI * e;   // entry record: i, j and the body of dof i in one read; the armature (diagonal entries) in the same volley
      const int i = MI[0], j = MI[1], bi = MI[2];
      const float arm = m->recf_mpair[e];
      float buf[6], cd[6];
      for (int t = 0; t < 6; t++) cd[t] = c->cdof[6 * i + t];
      inertMulf(buf, c->crb + 10 * bi, cd);
      float v = 0;
      for (int t = 0; t < 6; t++) v += c->cdof[6 * j + t] * buf[t];
      if (i == j) v += arm;
      c->M[i * GRX_NVC + j] = v; c->M[j * GRX_NVC + i] = v;
    }
  }
  WAVE_SYNC();
}

