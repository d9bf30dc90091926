// grx_hand_task.h -- Shadow Dexterous Hand reach task fused around the physics substeps.
//
// Device restatement of
//   BaseRobotEnv.step ............ /root/reference/gymnasium_robotics/envs/robot_env.py:114-152
//   MujocoHandEnv._set_action .... envs/shadow_dexterous_hand/hand_env.py:36-58 (absolute control: ctrl = centre + action * half
//                                  range of the actuator's ctrlrange, clipped to it)
//   MujocoHandReachEnv._get_obs .. envs/shadow_dexterous_hand/reach.py:398-428 (robot qpos | qvel | five fingertip site positions;
//                                  read AFTER mj_step, i.e. site positions of the last substep's forward pass with the new qpos/qvel)
//   compute_reward / _is_success . reach.py:92-97, 128-130 (15-dim Euclidean distance, threshold 0.01)
#pragma once
#include "grx_engine.h"

#define GRX_HAND_NTIPS 5

// kind 0: HandReach (goal = 5 fingertip positions, 15 numbers).
// kind 1: HandManipulate* (goal = object pose, 7 numbers: manipulate.py:87-142, 298-316):
//   obs = robot qpos[nq_robot] | robot qvel[nq_robot] | object qvel[6] | object qpos[7];  achieved = object qpos[7]
struct GrxHandTask {
  int n_substeps, sparse_reward;
  int site[GRX_HAND_NTIPS];  // fingertip sites, reach.py:8-14 order
  int palm_body;             // body whose position _sample_goal offsets from (reach.py:413-416)
  double distance_threshold; // fp64, see grx_goal_distance3 (csrc/grx_fetch_task.h)
  int kind, nq_robot, obj_qadr, obj_dadr;
  int ignore_position, ignore_rotation;   // target_position == "ignore" / target_rotation == "ignore" (manipulate.py:92-97)
  float rotation_threshold;
  int ignore_z;     // ignore_z_target_rotation (pen variants, manipulate.py:100-108)
  int touch_mode;   // 0: no touch values in the observation; 1 sensordata, 2 boolean, 3 log(x+1) (manipulate_touch_sensors.py:124-131)
};
GRX_DEV int grx_hand_goal_dim(const GrxHandTask* t) { return t->kind ? 7 : 3 * GRX_HAND_NTIPS; }
GRX_DEV int grx_hand_obs_dim(const GrxHandTask* t, int nq, int nv, int ntouch) {
  return t->kind ? 2 * t->nq_robot + 6 + 7 + (t->touch_mode ? ntouch : 0) : nq + nv + 3 * GRX_HAND_NTIPS;
}

struct GrxHandBuffers {
  float *qpos, *qvel, *qacc_ws;  // [N,nq] [N,nv] [N,nv]
  const float* goal;             // [N,15]
  const float* action;           // [N,nu]   (null for the forward-only entry point)
  float *obs, *achieved;         // [N,nq+nv+15] [N,15]
  float* palm;                   // [N,3]  position of the palm body (env setup)
  float* reward;                 // [N]
  unsigned char* success;        // [N]
  int* status;                   // [N]
  const unsigned char* mask;     // [N] or null
  const int* order;              // [grid] or null: world handled by workgroup b (cost-ordered dispatch, see grx_fetch_buffers)
  int* cost;                     // [N] or null: out, cost estimate of this world
  float* packed;                 // [N, obs_dim + 2 goal_dim + 2] or null: out, the row [obs | achieved | desired | reward | success]
  GrxLane lane;                   // the overflow lane (include/grx_capi.h grx_overflow_lane): no dropped contacts
  float* split_rows;              // [N, split_stride] or null: carrier rows of the split step [qpos | qvel | warm start] (include/grx_capi.h)
  int* split_state;               // [N, 4] or null: [4 w] = parts of world w done in this launch (< 0: re-run booked), [4 w + 1] = their status flags, [4 w + 2] = their measured duration
  int split_stride, split_parts;  // words per carrier row (>= nq + 2 nv); >= 2: the step launch has split_parts workgroups per world
};

// Euclidean distance with a fixed accumulation order, shared by the step kernel and the recompute kernel so that
// reward == compute_reward(achieved, desired) bit for bit (core.py:59-62)
GRX_DEV double grx_goal_distance_n(const float* a, const float* b, int n) {
  double s = 0.0;
  for (int k = 0; k < n; k++) { const double d = (double)a[k] - (double)b[k]; s += d * d; }
  return sqrt(s);
}
GRX_DEV float grx_hand_reward(double d, double thr, int sparse) { return sparse ? ((d > thr) ? -1.0f : -0.0f) : (float)(-d); }

// manipulate.py:87-142.  The reference takes the angle as 2 acos(clip(w)) of quat_a * conj(quat_b), whose scalar part is the
// 4-vector dot product; 2 atan2(|vector part|, w) is the same angle for unit quaternions and keeps fp32 accuracy near 0.
// Euler angles of utils/rotations.py (quat2euler = mat2euler(quat2mat), rotations.py:162-184,227-272) and back (euler2quat,
// rotations.py:140-159 = qx(e0) qy(e1) qz(e2)); used only by the ignore-z special case
GRX_DEV void grx_quat2euler(const float* q, float* e) {
  const float w = q[0], x = q[1], y = q[2], z = q[3], n = w * w + x * x + y * y + z * z, s = n > 1.1920929e-07f * 4.0f ? 2.0f / n : 0.0f;
  const float m00 = 1.0f - s * (y * y + z * z), m01 = s * (x * y - w * z), m02 = s * (x * z + w * y);
  const float m10 = s * (x * y + w * z), m11 = 1.0f - s * (x * x + z * z), m12 = s * (y * z - w * x), m22 = 1.0f - s * (x * x + y * y);
  const float cy = sqrtf(m22 * m22 + m12 * m12);
  if (cy > 8.8817842e-16f) { e[2] = -atan2f(m01, m00); e[1] = -atan2f(-m02, cy); e[0] = -atan2f(m12, m22); }
  else { e[2] = -atan2f(-m10, m11); e[1] = -atan2f(-m02, cy); e[0] = 0.0f; }
}
GRX_DEV void grx_euler2quat(const float* e, float* q) {
  const float cx = cosf(0.5f * e[0]), sx = sinf(0.5f * e[0]), cy = cosf(0.5f * e[1]), sy = sinf(0.5f * e[1]), cz = cosf(0.5f * e[2]), sz = sinf(0.5f * e[2]);
  // qy qz
  const float aw = cy * cz, ax = sy * sz, ay = sy * cz, az = cy * sz;
  q[0] = cx * aw - sx * ax; q[1] = cx * ax + sx * aw; q[2] = cx * ay - sx * az; q[3] = cx * az + sx * ay;
}

GRX_DEV void grx_manip_distance(const float* a_in, const float* b, int ignore_pos, int ignore_rot, int ignore_z, float* d_pos, float* d_rot) {
  *d_pos = ignore_pos ? 0.0f : grx_goal_distance_n(a_in, b, 3);
  float dr = 0.0f;
  if (!ignore_rot) {
    float a[7] = {a_in[0], a_in[1], a_in[2], a_in[3], a_in[4], a_in[5], a_in[6]};
    if (ignore_z) {   // give quat_a the z Euler angle of quat_b, then compare (manipulate.py:100-108)
      float ea[3], eb[3];
      grx_quat2euler(a + 3, ea); grx_quat2euler(b + 3, eb);
      ea[2] = eb[2];
      grx_euler2quat(ea, a + 3);
    }
    const float w0 = a[3], x0 = a[4], y0 = a[5], z0 = a[6], w1 = b[3], x1 = -b[4], y1 = -b[5], z1 = -b[6];
    const float w = w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1;
    const float x = w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1, y = w0 * y1 + y0 * w1 + z0 * x1 - x0 * z1, z = w0 * z1 + z0 * w1 + x0 * y1 - y0 * x1;
    dr = 2.0f * atan2f(sqrtf(x * x + y * y + z * z), w);
  }
  *d_rot = dr;
}
GRX_DEV int grx_manip_success(float d_pos, float d_rot, float thr_pos, float thr_rot) { return (d_pos < thr_pos) && (d_rot < thr_rot); }
GRX_DEV float grx_manip_reward(float d_pos, float d_rot, float thr_pos, float thr_rot, int sparse) {
  return sparse ? (grx_manip_success(d_pos, d_rot, thr_pos, thr_rot) ? 0.0f : -1.0f) : -(10.0f * d_pos + d_rot);
}

template <class S>
struct GrxHand {
  typedef GrxEngine<S> E;
  GRX_MEM void grx_hand_outputs(const GrxModel* m, const GrxHandTask* t, const GrxCtx* c, float* obs, float* achieved, float* palm, int lane_) {
    GRX_FRESH_MODEL(m, c);
    const int nq = GRX_NQC, nv = GRX_NVC;
    if (t->kind) {
      const int nr = t->nq_robot;
      FOR_LANES {
        for (int i = lane; i < nr; i += 64) { obs[i] = c->qpos[i]; obs[nr + i] = c->qvel[i]; }
        for (int i = lane; i < 6; i += 64) obs[2 * nr + i] = c->qvel[t->obj_dadr + i];
        // the object's position leaves the engine in the MJCF's world frame (origin added in fp64, grx_engine.h grx_world_out); the quaternion as it is
        for (int i = lane; i < 7; i += 64) { const float q = c->qpos[t->obj_qadr + i], v = i < 3 ? grx_world_out(q, m->origin[i]) : q; obs[2 * nr + 6 + i] = v; achieved[i] = v; }
        for (int i = lane; i < 3; i += 64) palm[i] = grx_world_out(c->xpos[3 * t->palm_body + i], m->origin[i]);
      }
      WAVE_SYNC();
      if (t->touch_mode) E::grx_touch_sensors(m, c, obs + 2 * nr + 13, t->touch_mode, lane_);   // same forward pass as the contacts
      return;
    }
    FOR_LANES {
      for (int i = lane; i < nq; i += 64) obs[i] = c->qpos[i];
      for (int i = lane; i < nv; i += 64) obs[nq + i] = c->qvel[i];
      for (int i = lane; i < 3 * GRX_HAND_NTIPS; i += 64) {
        const int k = i / 3, e = i - 3 * k;
        const float v = grx_world_out(c->sxpos[3 * t->site[k] + e], m->origin[e]);
        obs[nq + nv + i] = v; achieved[i] = v;
      }
      for (int i = lane; i < 3; i += 64) palm[i] = grx_world_out(c->xpos[3 * t->palm_body + i], m->origin[i]);
    }
    WAVE_SYNC();
  }

  // s0 / s1: the substeps [s0, s1) of the step, outputs: write the observation rows behind them (parts of a split step, include/grx_capi.h grx_hand_buffers.split_parts; default: the whole step)
  GRX_MEM void grx_hand_step_world(const GrxModel* m, const GrxHandTask* t, GrxCtx* c, const float* action, float* obs, float* achieved, float* palm,
                                   int lane_, int s0 = 0, int s1 = -1, bool outputs = true) {
    GRX_FRESH_MODEL(m, c);
    FOR_LANES {
      for (int i = lane; i < GRX_NUC; i += 64) {
        const float lo = m->act_ctrlrange[2 * i], hi = m->act_ctrlrange[2 * i + 1];
        const float a = fminf(1.0f, fmaxf(-1.0f, action[i]));   // robot_env.py:132 clips to the action space
        c->ctrl[i] = fminf(hi, fmaxf(lo, 0.5f * (hi + lo) + a * (0.5f * (hi - lo))));
      }
    }
    WAVE_SYNC();
    if (s1 < 0) s1 = t->n_substeps;
    for (int s = s0; s < s1; s++) {
      E::grx_check_state(m, c, lane_);
      E::grx_forward_euler(m, c, 1, lane_);
      if (c->bail && grx_lane_claim(c, lane_)) break;   // a capacity overflowed and the re-run on the large tables is booked: this run will be discarded
    }
    if (!outputs) return;
    // A discarded run must not touch the output rows either: its re-run may already be under way in a polling workgroup of the standing lane launch on ANOTHER XCD,
    // and a stale observation written here would sit dirty in this XCD's L2 until the end of the launch and then overwrite the re-run's row (state right,
    // observation / achieved goal of the truncated run: found by tests/test_gpu_manipulate.py::test_overflow_lane_polling_equals_the_serialised_rerun).
    if (c->bail == 2) return;
    grx_hand_outputs(m, t, c, obs, achieved, palm, lane_);
  }
};
