// grx_eng_kinematics.h -- K1 - K3: kinematics (pointer jumping), spatial inertias, motion axes, composite inertias, the mass matrix.
// A FRAGMENT of csrc/grx_engine.h: textually included INSIDE `template <class S> struct GrxEngine { ... }` (every function here is a static member), in the order the engine
// header lists; not a standalone header.  The split is purely textual (round 5): the token stream of the translation units is unchanged.
// ------------------------------------------------------------------------------------------
// K1 forward kinematics
// ------------------------------------------------------------------------------------------
GRX_MEM void grx_kinematics(const GrxModel* m, GrxCtx* c, int lane_) {
  GRX_FRESH_MODEL(m, c);
  FOR_LANES {
    for (int b = lane; b < GRX_NBC; b += 64) {
      float* pl = c->ploc + 3 * b; float* ql = c->qloc + 4 * b;
      if (b == 0) {
        c->xpos[0] = c->xpos[1] = c->xpos[2] = 0; c->xquat[0] = 1; c->xquat[1] = c->xquat[2] = c->xquat[3] = 0;
        for (int k = 0; k < 9; k++) c->xmat[k] = (k % 4 == 0) ? 1.0f : 0.0f;
        continue;
      }
      int mid = m->body_mocapid[b];
      if (mid >= 0) {
        float q[4] = {c->mocap_quat[4 * mid], c->mocap_quat[4 * mid + 1], c->mocap_quat[4 * mid + 2], c->mocap_quat[4 * mid + 3]};
        normalize4f(q);
        pl[0] = c->mocap_pos[3 * mid]; pl[1] = c->mocap_pos[3 * mid + 1]; pl[2] = c->mocap_pos[3 * mid + 2];
        ql[0] = q[0]; ql[1] = q[1]; ql[2] = q[2]; ql[3] = q[3];
        continue;
      }
      const unsigned ja = (unsigned)m->body_jntadr[b]; const int jn = m->body_jntnum[b];   // unsigned: no sign-extended 64-bit index pair kept live
      int jt0 = -1, qa = 0;
      if (jn == 1) { jt0 = m->jnt_type[ja]; qa = m->jnt_qposadr[ja]; }   // both table reads before the divergent branches
#ifndef GRX_EMU
      asm volatile("" : "+v"(qa));   // keep the read here (sunk to its use, the 64-bit index pair is spilled to scratch across the branches)
#endif
      if (jt0 == 0) {  // free joint: qpos is the world pose
        float q[4] = {c->qpos[qa + 3], c->qpos[qa + 4], c->qpos[qa + 5], c->qpos[qa + 6]};
        if (!c->resume_first) normalize4f(q);   // (a world resumed mid-step: this substep's normalisation was done, in place, by the kernel that handed it off; doing it twice is not bit-neutral)
        pl[0] = c->qpos[qa]; pl[1] = c->qpos[qa + 1]; pl[2] = c->qpos[qa + 2];
        for (int k = 0; k < 4; k++) { ql[k] = q[k]; c->qpos[qa + 3 + k] = q[k]; }
        for (int k = 0; k < 3; k++) { c->janchor[3 * ja + k] = pl[k]; c->jaxis[3 * ja + k] = (k == 2) ? 1.0f : 0.0f; }
        continue;
      }
      float p[3] = {m->body_pos[3 * b], m->body_pos[3 * b + 1], m->body_pos[3 * b + 2]};
      if (S::kShift && m->nshift && m->body_shift[b]) { p[0] += c->shift[0]; p[1] += c->shift[1]; p[2] += c->shift[2]; }   // child of the (world-fixed) shift group
      float q[4] = {m->body_quat[4 * b], m->body_quat[4 * b + 1], m->body_quat[4 * b + 2], m->body_quat[4 * b + 3]};
      for (int k = 0; k < jn; k++) {
        const unsigned j = ja + (unsigned)k; const int qa = m->jnt_qposadr[j];
        float jp[3] = {m->jnt_pos[3 * j], m->jnt_pos[3 * j + 1], m->jnt_pos[3 * j + 2]};
        float jx[3] = {m->jnt_axis[3 * j], m->jnt_axis[3 * j + 1], m->jnt_axis[3 * j + 2]};
        float anchor[3], axis[3];
        rotVecQuatf(anchor, jp, q); anchor[0] += p[0]; anchor[1] += p[1]; anchor[2] += p[2];
        rotVecQuatf(axis, jx, q);
        for (int t = 0; t < 3; t++) { c->janchor[3 * j + t] = anchor[t]; c->jaxis[3 * j + t] = axis[t]; }
        float dq = c->qpos[qa] - m->qpos0[qa];
        if (m->jnt_type[j] == 2) {
          p[0] += axis[0] * dq; p[1] += axis[1] * dq; p[2] += axis[2] * dq;
        } else if (m->jnt_type[j] == 3) {
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
  WAVE_SYNC();
  GRX_SUBTICK(c, 9);
  // world poses by pointer jumping: in round s every body composes its pose (relative to the ancestor 2^s levels up) with
  // that ancestor's pose (relative to ITS ancestor 2^s levels up): ceil(log2(depth)) rounds instead of one composition per
  // ancestor.  Rounds ping-pong between {ploc,qloc} and {xpos,xquat}; body_jump is the static schedule.
  const int nbk = GRX_NBC, nj = m->njump;
  for (int s = 0; s < nj; s++) {
    const float* sp = (s & 1) ? c->xpos : c->ploc; const float* sq = (s & 1) ? c->xquat : c->qloc;
    float* dp = (s & 1) ? c->ploc : c->xpos; float* dq = (s & 1) ? c->qloc : c->xquat;
    FOR_LANES {
      for (int b = 1 + lane; b < nbk; b += 64) {
        float p[3] = {sp[3 * b], sp[3 * b + 1], sp[3 * b + 2]}, q[4] = {sq[4 * b], sq[4 * b + 1], sq[4 * b + 2], sq[4 * b + 3]};
        const int anc = m->body_jump[s * nbk + b];
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
      for (int b = 1 + lane; b < nbk; b += 64) {
        float p[3] = {sp[3 * b], sp[3 * b + 1], sp[3 * b + 2]}, q[4] = {sq[4 * b], sq[4 * b + 1], sq[4 * b + 2], sq[4 * b + 3]};
        int isfree = (m->body_jntnum[b] == 1 && m->jnt_type[m->body_jntadr[b]] == 0);
        if (!(m->body_mocapid[b] >= 0 || isfree)) normalize4f(q);
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
      if (m->jnt_type[j] == 0) continue;
      int par = m->body_parent[m->jnt_bodyid[j]];
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
      int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j], jt = m->jnt_type[j];
      const float* cref = c->xpos + 3 * m->body_rootid[b];
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
      int i = m->mpair_i[e], j = m->mpair_j[e];
      float buf[6], cd[6];
      for (int t = 0; t < 6; t++) cd[t] = c->cdof[6 * i + t];
      inertMulf(buf, c->crb + 10 * m->dof_bodyid[i], cd);
      float v = 0;
      for (int t = 0; t < 6; t++) v += c->cdof[6 * j + t] * buf[t];
      if (i == j) v += m->dof_armature[i];
      c->M[i * GRX_NVC + j] = v; c->M[j * GRX_NVC + i] = v;
    }
  }
  WAVE_SYNC();
}

