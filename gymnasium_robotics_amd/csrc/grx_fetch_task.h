// grx_fetch_task.h -- Fetch family task code fused around the physics substeps.
//
// Device restatement of the per-step Python the reference runs around mj_step:
//   _set_action ........ /root/reference/gymnasium_robotics/envs/fetch/fetch_env.py:85-105,305-310
//   ctrl_set_action .... /root/reference/gymnasium_robotics/utils/mujoco_utils.py:34-48
//   mocap_set_action ... mujoco_utils.py:51-71 (+ reset_mocap2body_xpos :83-107)
//   _step_callback ..... fetch_env.py:295-303
//   _get_obs ........... fetch_env.py:107-143,312-360 (mat2euler: utils/rotations.py:162-184)
//   compute_reward ..... fetch_env.py:74-80, goal_distance :16-18, _is_success :168-170
#pragma once
#include "grx_engine.h"

struct GrxFetchTask {
  int has_object, block_gripper, n_substeps, sparse_reward;
  int grip_body;                 // fused body that carries robot0:gripper_link
  float grip_relpos[3], grip_relquat[4];
  int site_grip, site_obj;       // site ids
  int jq_rf, jq_lf, jd_rf, jd_lf;  // qpos / dof addresses of the r / l finger joints
  int obs_dim, goal_dim;
  float dt;
  double distance_threshold;   // fp64: the success test / sparse reward compare the fp64 distance with the reference's fp64 threshold (fetch_env.py:74-80,168-170)
};

// per-world HBM buffers (world-major rows; one contiguous vector per world per field)
struct GrxFetchBuffers {
  float *qpos, *qvel, *qacc_ws, *mocap;  // [N,nq] [N,nv] [N,nv] [N,7*nmocap]
  float* aux;                            // [N,8]  pose (pos3, quat4) of gripper_link at the last forward pass
  const float* goal;                     // [N,3]
  const float* action;                   // [N,4]
  float *obs, *achieved;                 // [N,obs_dim] [N,3]
  float* reward;                         // [N]
  unsigned char* success;                // [N]
  int* status;                           // [N]
  const unsigned char* mask;             // [N] or null: worlds to process
  const int* order;                      // [grid] or null: world handled by workgroup b (dispatch order = cost order, see grx_fetch_step_kernel)
  int* cost;                             // [N] or null: out, cost estimate of this world (the next launch's ordering key)
  float* packed;                         // [N, obs_dim + 3 + 3 + 2] or null: out, the row [obs | achieved | desired | reward | success] (what the cross-rank gather ships)
  float* hullcache;                      // [N, GRX_HULLCACHE_WORDS] or null: in/out, GrxCtx::meshcache carried across launches + the support-vertex guesses (include/grx_capi.h)
  float* handoff;                        // [N, handoff_stride] or null: the worlds' mid-step hand-off rows (include/grx_capi.h; GrxCtx::handoff); word 0 of a row = substep + 1, 0 = none
  int handoff_stride, handoff_large;     // words per row (>= grx_handoff_words); handoff_large: entries claimed by THIS launch need the large tables (see grx_overflow_lane)
  int* split_state;                      // [N, 2] or null: split step (include/grx_capi.h): [2 w] = parts of world w done in this launch (< 0: aborted, re-run booked), [2 w + 1] = their measured duration
  int split_parts, split_pad_;           // >= 2: the launch has split_parts workgroups per world, each running its share of the substeps (the hand-off rows carry the state between them)
  GrxLane lane;                          // the overflow lane (include/grx_capi.h grx_overflow_lane): no dropped contacts
};

// Goal distance in fp64 (goal_distance, fetch_env.py:16-18: np.linalg.norm of the fp64 difference).  The flags and the sparse reward are then EXACTLY the
// reference's functions of the returned (fp32) achieved / desired goals -- d < 0.05 decided in the reference's own arithmetic, no rounding band around the
// threshold -- and the fused step kernel, the standalone reward kernel and the HER relabel share this one routine (reward == compute_reward(achieved, desired)
// bit for bit, core.py:59-62).  Three subtractions, three products (exact: 24-bit inputs) and one square root per world.
GRX_DEV double grx_goal_distance3(const float* a, const float* b) {
  const double dx = (double)a[0] - (double)b[0], dy = (double)a[1] - (double)b[1], dz = (double)a[2] - (double)b[2];
  return sqrt(dx * dx + dy * dy + dz * dz);
}
GRX_DEV float grx_fetch_reward(double d, double thresh, int sparse) { return sparse ? -((d > thresh) ? 1.0f : 0.0f) : (float)(-d); }

GRX_DEV void grx_mat2euler(const float* R, float* e) {
  const float eps4 = 4.0f * 1.1920929e-07f;  // the reference uses 4*eps of float64; only the gimbal branch differs
  float cy = sqrtf(R[8] * R[8] + R[5] * R[5]);
  if (cy > eps4) { e[0] = -atan2f(R[5], R[8]); e[1] = -atan2f(-R[2], cy); e[2] = -atan2f(R[1], R[0]); }
  else { e[0] = 0.0f; e[1] = -atan2f(-R[2], cy); e[2] = -atan2f(-R[3], R[4]); }
}

template <class S>
struct GrxFetch {
  typedef GrxEngine<S> E;
// linear / angular velocity of a world point fixed to body b: J(point) * qvel using the
// motion axes of the LAST forward pass and the CURRENT qvel (what mj_jacSite @ qvel gives)
GRX_MEM void grx_point_velocity(const GrxModel* m, const GrxCtx* c, int b, const float* point, float* vp, float* vr) {
  const float* cref = c->xpos + 3 * m->body_rootid[b];
  float off[3] = {point[0] - cref[0], point[1] - cref[1], point[2] - cref[2]};
  float w[3] = {0, 0, 0}, v[3] = {0, 0, 0};
  for (int d = m->body_lastdof[b]; d >= 0; d = m->dof_parentid[d]) {
    float qd = c->qvel[d]; const float* cd = c->cdof + 6 * d;
    w[0] += cd[0] * qd; w[1] += cd[1] * qd; w[2] += cd[2] * qd; v[0] += cd[3] * qd; v[1] += cd[4] * qd; v[2] += cd[5] * qd;
  }
  float t[3]; cross3f(t, w, off);
  vp[0] = v[0] + t[0]; vp[1] = v[1] + t[1]; vp[2] = v[2] + t[2]; vr[0] = w[0]; vr[1] = w[1]; vr[2] = w[2];
}

GRX_MEM void grx_fetch_set_action(const GrxModel* m, const GrxFetchTask* t, GrxCtx* c, const float* aux, const float* action, int lane_) {
  GRX_FRESH_MODEL(m, c);
  LANE0 {
    float a[4];
    for (int k = 0; k < 4; k++) a[k] = fminf(1.0f, fmaxf(-1.0f, action[k]));
    float g = t->block_gripper ? 0.0f : a[3];
    for (int i = 0; i < GRX_NUC; i++) {
      int j = m->act_trnid[i];
      c->ctrl[i] = (m->act_biastype[i] == 0) ? g : c->qpos[m->jnt_qposadr[j]] + g;
    }
    // mocap <- pose of the welded body at the last forward pass, then += deltas (quaternion ADDED, normalised in kinematics)
    for (int k = 0; k < 3; k++) c->mocap_pos[k] = aux[k] + 0.05f * a[k];
    c->mocap_quat[0] = aux[3] + 1.0f; c->mocap_quat[1] = aux[4] + 0.0f; c->mocap_quat[2] = aux[5] + 1.0f; c->mocap_quat[3] = aux[6] + 0.0f;
  }
  WAVE_SYNC();
}

// writes aux (gripper_link pose of the current kinematics), obs, achieved goal
GRX_MEM void grx_fetch_outputs(const GrxModel* m, const GrxFetchTask* t, const GrxCtx* c, float* aux, float* obs, float* achieved, int lane_) {
  GRX_FRESH_MODEL(m, c);
  LANE0 {
    int b = t->grip_body; float v[3], q[4];
    mulMatVec3f(v, c->xmat + 9 * b, t->grip_relpos);
    for (int k = 0; k < 3; k++) aux[k] = c->xpos[3 * b + k] + v[k];
    mulQuatf(q, c->xquat + 4 * b, t->grip_relquat);
    for (int k = 0; k < 4; k++) aux[3 + k] = q[k];
    aux[7] = 0;
    float dt = t->dt;
    const float* gp = c->sxpos + 3 * t->site_grip;
    float gvp[3], gvr[3];
    grx_point_velocity(m, c, m->site_bodyid[t->site_grip], gp, gvp, gvr);
    int o = 0;
    for (int k = 0; k < 3; k++) obs[o++] = gp[k];
    if (t->has_object) {
      const float* op = c->sxpos + 3 * t->site_obj;
      float ovp[3], ovr[3], e[3];
      grx_point_velocity(m, c, m->site_bodyid[t->site_obj], op, ovp, ovr);
      grx_mat2euler(c->sxmat + 9 * t->site_obj, e);
      for (int k = 0; k < 3; k++) obs[o++] = op[k];
      for (int k = 0; k < 3; k++) obs[o++] = op[k] - gp[k];
      obs[o++] = c->qpos[t->jq_rf]; obs[o++] = c->qpos[t->jq_lf];
      for (int k = 0; k < 3; k++) obs[o++] = e[k];
      for (int k = 0; k < 3; k++) obs[o++] = ovp[k] * dt - gvp[k] * dt;
      for (int k = 0; k < 3; k++) obs[o++] = ovr[k] * dt;
      for (int k = 0; k < 3; k++) achieved[k] = op[k];
    } else {
      obs[o++] = c->qpos[t->jq_rf]; obs[o++] = c->qpos[t->jq_lf];
      for (int k = 0; k < 3; k++) achieved[k] = gp[k];
    }
    for (int k = 0; k < 3; k++) obs[o++] = gvp[k] * dt;
    obs[o++] = c->qvel[t->jd_rf] * dt; obs[o++] = c->qvel[t->jd_lf] * dt;
  }
  WAVE_SYNC();
}

// whole env.step() for one world whose state is already in the LDS context
GRX_MEM void grx_fetch_step_world(const GrxModel* m, const GrxFetchTask* t, GrxCtx* c, const float* aux_in, const float* action,
                                  float* aux_out, float* obs, float* achieved, int lane_) {
  grx_fetch_sim_world(m, t, c, aux_in, action, lane_);
  grx_fetch_outputs(m, t, c, aux_out, obs, achieved, lane_);
}
// the simulation part alone: the step kernel derives the output pointers after it, so no global address stays live across the substeps.
// s0 > 0: the world RESUMES at substep s0 -- the caller has restored ctrl, mocap, qpos, qvel and the warm start from its hand-off row (GrxCtx::handoff) and set c->resume_first.
// s_end >= 0: stop BEFORE substep s_end (a part of a split step, include/grx_capi.h grx_fetch_buffers.split_parts: the caller writes the state to the world's row).
GRX_MEM void grx_fetch_sim_world(const GrxModel* m, const GrxFetchTask* t, GrxCtx* c, const float* aux_in, const float* action, int lane_, int s0 = 0, int s_end = -1) {
  if (s0 == 0) grx_fetch_set_action(m, t, c, aux_in, action, lane_);
  // n_substeps x mj_step, plus (block_gripper tasks) the _step_callback: zero the finger qpos and run one mj_forward.
  // One loop, one call site of the physics, so the loop body stays resident in the instruction cache.
  const int total = s_end >= 0 ? s_end : t->n_substeps + (t->block_gripper ? 1 : 0);
  for (int s = s0; s < total; s++) {
    const int callback = (s == t->n_substeps);
    if (callback) { LANE0 { c->qpos[t->jq_lf] = 0.0f; c->qpos[t->jq_rf] = 0.0f; } WAVE_SYNC(); }
    else E::grx_check_state(m, c, lane_);
    E::grx_forward_euler(m, c, !callback, lane_);
    c->resume_first = 0;
    if (grx_handoff_due(c)) {   // (kernels launched with a hand-off row) the substep stopped before it touched the state: hand the world off AT this substep
      if (grx_lane_handoff(c, s, GRX_NQC, GRX_NVC, GRX_NUC, GRX_NMC, lane_)) break;
      // the step's entry list is full (c->bail is 0 now): this kernel finishes the substep itself, dropping what it cannot hold / collide; the sticky status flag says so
      LANE0 { c->cnt[2] &= ~GRX_ST_HULL; }
      WAVE_SYNC();
      c->resume_first = 1; s--;
      continue;
    }
    if (c->bail && grx_lane_claim(c, lane_)) break;   // a capacity overflowed and the re-run on the large tables is booked: this run will be discarded
  }
}
};  // struct GrxFetch
