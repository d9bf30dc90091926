// grx_eng_sensors.h -- K12 - K14: touch sensors, bad-number check / state reset.
// A FRAGMENT of csrc/grx_engine.h: textually included INSIDE `template <class S> struct GrxEngine { ... }` (every function here is a static member), in the order the engine
// header lists; not a standalone header.  The split is purely textual (round 5): the token stream of the translation units is unchanged.
// ------------------------------------------------------------------------------------------
// K12 touch sensors (MuJoCo mjSENS_TOUCH): out[t] = sum of the normal forces of the active contacts that involve the zone's body
// and whose ray (from the contact point along the contact normal, flipped when the zone's body is the contact's second body)
// meets the zone (sphere or box site).  Runs after the constraint solve of the same forward pass (row forces in efc_force).
// mode 1: raw value, 2: value > 0, 3: log(value + 1)  (manipulate_touch_sensors.py:124-131), 4: clip(value, -1, 1) (adroit_hammer.py:344-346)
// ------------------------------------------------------------------------------------------
GRX_MEM float grx_ray_sphere(const float* p, const float* d, float r) {
  const float a = dot3f(d, d), b = dot3f(d, p), cc = dot3f(p, p) - r * r, det = b * b - a * cc;
  if (det < GRX_MINVAL || a < GRX_MINVAL) return -1.0f;
  const float sq = sqrtf(det), x0 = (-b - sq) / a, x1 = (-b + sq) / a;
  return x0 >= 0 ? x0 : (x1 >= 0 ? x1 : -1.0f);
}
GRX_MEM float grx_ray_box(const float* p, const float* d, const float* sz) {
  float best = -1.0f;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    if (fabsf(d[i]) < GRX_MINVAL) continue;
#pragma unroll
    for (int side = -1; side <= 1; side += 2) {
      const float t = ((float)side * sz[i] - p[i]) / d[i];
      if (t >= 0 && fabsf(p[j] + t * d[j]) <= sz[j] && fabsf(p[k] + t * d[k]) <= sz[k] && (best < 0 || t < best)) best = t;
    }
  }
  return best;
}
// cylinder zone (radius r, half height h along z): nearest non-negative hit of the side or a cap (the oracle's ray_cylinder)
GRX_MEM float grx_ray_cylinder(const float* p, const float* d, float r, float h) {
  float best = -1.0f;
  const float a = d[0] * d[0] + d[1] * d[1], b = d[0] * p[0] + d[1] * p[1], cc = p[0] * p[0] + p[1] * p[1] - r * r;
  if (a > GRX_MINVAL) {
    const float det = b * b - a * cc;
    if (det >= 0.0f) {
      const float sq = sqrtf(det), t0 = (-b - sq) / a, t1 = (-b + sq) / a;
      if (t0 >= 0.0f && fabsf(p[2] + t0 * d[2]) <= h) best = t0;
      if (t1 >= 0.0f && fabsf(p[2] + t1 * d[2]) <= h && (best < 0.0f || t1 < best)) best = t1;
    }
  }
  if (fabsf(d[2]) > GRX_MINVAL) {
#pragma unroll
    for (int side = -1; side <= 1; side += 2) {
      const float t = ((float)side * h - p[2]) / d[2], x = p[0] + t * d[0], y = p[1] + t * d[1];
      if (t >= 0.0f && x * x + y * y <= r * r && (best < 0.0f || t < best)) best = t;
    }
  }
  return best;
}
GRX_MEM void grx_touch_sensors(const GrxModel* m, const GrxCtx* c, float* out, int mode, int lane_) {
  GRX_FRESH_MODEL(m, c);
  const int ncon = c->cnt[0] < c->maxcon ? c->cnt[0] : c->maxcon, nefc = c->cnt[1];
  FOR_LANES {
    for (int t = lane; t < m->ntouch; t += 64) {
      const int b = m->touch_body[t], type = m->touch_type[t];
      const float lp[3] = {m->touch_pos[3 * t], m->touch_pos[3 * t + 1], m->touch_pos[3 * t + 2]};
      const float lq[4] = {m->touch_quat[4 * t], m->touch_quat[4 * t + 1], m->touch_quat[4 * t + 2], m->touch_quat[4 * t + 3]};
      const float sz[3] = {m->touch_size[3 * t], m->touch_size[3 * t + 1], m->touch_size[3 * t + 2]};
      float zp[3], zl[9], zR[9], v[3], val = 0.0f;
      mulMatVec3f(v, c->xmat + 9 * b, lp);
      for (int k = 0; k < 3; k++) zp[k] = c->xpos[3 * b + k] + v[k];
      quat2matf(zl, lq); mulMat3f(zR, c->xmat + 9 * b, zl);
      for (int k = 0; k < ncon; k++) {
        const int r0 = c->con_efc[k];
        if (r0 < 0) continue;
        const int pr = c->con_pair[k], b1 = m->geom_bodyid[m->pair_geom1[pr]], b2 = m->geom_bodyid[m->pair_geom2[pr]];
        if (b != b1 && b != b2) continue;
        float fn = 0.0f;
        for (int q = 0; q < c->con_nr[k] && r0 + q < nefc; q++) fn += c->efc_force[r0 + q];
        if (!(fn > 0.0f)) continue;
        const float sg = (b == b2) ? -1.0f : 1.0f;
        const float dw[3] = {sg * c->con_frame[3 * k], sg * c->con_frame[3 * k + 1], sg * c->con_frame[3 * k + 2]};
        const float pw[3] = {c->con_pos[3 * k] - zp[0], c->con_pos[3 * k + 1] - zp[1], c->con_pos[3 * k + 2] - zp[2]};
        float pl[3], dl[3];
        mulMatTVec3f(pl, zR, pw); mulMatTVec3f(dl, zR, dw);
        const float hit = (type == 2) ? grx_ray_sphere(pl, dl, sz[0]) : (type == 5 ? grx_ray_cylinder(pl, dl, sz[0], sz[1]) : grx_ray_box(pl, dl, sz));
        if (hit >= 0.0f) val += fn;
      }
      out[t] = (mode == 2) ? (val > 0.0f ? 1.0f : 0.0f) : (mode == 3 ? logf(val + 1.0f) : (mode == 4 ? fminf(1.0f, fmaxf(-1.0f, val)) : val));
    }
  }
  WAVE_SYNC();
}

GRX_MEM void grx_check_state(const GrxModel* m, GrxCtx* c, int lane_) {
  GRX_FRESH_MODEL(m, c);
  // mj_checkPos / mj_checkVel (engine_forward.c): a non-finite or huge coordinate resets the world to the model's
  // initial state (mj_resetData) and raises the warning; the status word plays the role of the warning counter.
  GRX_LANEVAR(badp);
  FOR_LANES {
    int bad = 0;
    for (int i = lane; i < GRX_NQC; i += 64) { float v = c->qpos[i]; if (!(v == v) || fabsf(v) > 1e10f) bad = 1; }
    for (int i = lane; i < GRX_NVC; i += 64) { float v = c->qvel[i]; if (!(v == v) || fabsf(v) > 1e10f) bad = 1; }
    LV(badp) = bad ? 1.0f : 0.0f;
  }
  WAVE_SYNC();
  if (grx_reduce_max(badp) > 0.5f) {
    FOR_LANES {
      for (int i = lane; i < GRX_NQC; i += 64) c->qpos[i] = m->qpos0[i];
      for (int i = lane; i < GRX_NVC; i += 64) { c->qvel[i] = 0.0f; c->qacc_ws[i] = 0.0f; }
      for (int i = lane; i < 3 * GRX_NMC; i += 64) c->mocap_pos[i] = m->mocap_pos0[i];
      for (int i = lane; i < 4 * GRX_NMC; i += 64) c->mocap_quat[i] = m->mocap_quat0[i];
    }
    LANE0 { c->cnt[2] |= GRX_ST_BADNUM; }
    WAVE_SYNC();
  }
}

