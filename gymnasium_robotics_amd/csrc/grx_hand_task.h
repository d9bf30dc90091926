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

struct GrxHandTask {
  int n_substeps, sparse_reward;
  int site[GRX_HAND_NTIPS];  // fingertip sites, reach.py:8-14 order
  int palm_body;             // body whose position _sample_goal offsets from (reach.py:413-416)
  float distance_threshold;
};

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
};

// Euclidean distance with a fixed accumulation order, shared by the step kernel and the recompute kernel so that
// reward == compute_reward(achieved, desired) bit for bit (core.py:59-62)
GRX_DEV float grx_goal_distance_n(const float* a, const float* b, int n) {
  float s = 0.0f;
  for (int k = 0; k < n; k++) { float d = a[k] - b[k]; s = fmaf(d, d, s); }
  return sqrtf(s);
}
GRX_DEV float grx_hand_reward(float d, float thr, int sparse) { return sparse ? ((d > thr) ? -1.0f : -0.0f) : -d; }

template <class S>
struct GrxHand {
  typedef GrxEngine<S> E;
  GRX_MEM void grx_hand_outputs(const GrxModel* m, const GrxHandTask* t, const GrxCtx* c, float* obs, float* achieved, float* palm, int lane_) {
    GRX_FRESH_MODEL(m, c);
    const int nq = GRX_NQC, nv = GRX_NVC;
    FOR_LANES {
      for (int i = lane; i < nq; i += 64) obs[i] = c->qpos[i];
      for (int i = lane; i < nv; i += 64) obs[nq + i] = c->qvel[i];
      for (int i = lane; i < 3 * GRX_HAND_NTIPS; i += 64) {
        const int k = i / 3, e = i - 3 * k;
        const float v = c->sxpos[3 * t->site[k] + e];
        obs[nq + nv + i] = v; achieved[i] = v;
      }
      for (int i = lane; i < 3; i += 64) palm[i] = c->xpos[3 * t->palm_body + i];
    }
    WAVE_SYNC();
  }

  GRX_MEM void grx_hand_step_world(const GrxModel* m, const GrxHandTask* t, GrxCtx* c, const float* action, float* obs, float* achieved, float* palm,
                                   int lane_) {
    GRX_FRESH_MODEL(m, c);
    FOR_LANES {
      for (int i = lane; i < GRX_NUC; i += 64) {
        const float lo = m->act_ctrlrange[2 * i], hi = m->act_ctrlrange[2 * i + 1];
        const float a = fminf(1.0f, fmaxf(-1.0f, action[i]));   // robot_env.py:132 clips to the action space
        c->ctrl[i] = fminf(hi, fmaxf(lo, 0.5f * (hi + lo) + a * (0.5f * (hi - lo))));
      }
    }
    WAVE_SYNC();
    for (int s = 0; s < t->n_substeps; s++) {
      E::grx_check_state(m, c, lane_);
      E::grx_forward_euler(m, c, 1, lane_);
    }
    grx_hand_outputs(m, t, c, obs, achieved, palm, lane_);
  }
};
