// grx_adroit_task.h -- Adroit hand task code fused around the physics substeps.
//
// Device restatement of /root/reference/gymnasium_robotics/envs/adroit_hand/adroit_hammer.py
//   step ............ :291-329 (clip, a = act_mean + a * act_rng, do_simulation(a, frame_skip = 5) [MujocoEnv, 3P], reward, success)
//   _get_obs ........ :331-357 (qpos[:-6] | clip(qvel[-6:], +-1) | palm site | hammer body position | quat2euler(hammer quaternion) |
//                      nail-head site | clip(touch sensor on the nail head, +-1))
//   reset_model ..... :372-378 (the board height is per-world state here: GrxAdroitBuffers.shift; mj_forward + _get_obs = forward_only launch)
// As in the reference, the observation mixes the integrated qpos / qvel with the body / site poses and the sensor value of the LAST forward pass
// (mj_step does not recompute kinematics after integrating).
#pragma once
#include "grx_engine.h"
#include "grx_hand_task.h"   // grx_quat2euler

struct GrxAdroitTask {
  int n_substeps, sparse_reward;
  int site_grasp, site_target, site_goal, site_tool;   // S_grasp, S_target, nail_goal, tool
  int obj_body;                                         // the hammer ("Object")
  int nq_obs;                                           // leading qpos entries in the observation (nq - 6)
  int obs_dim;                                          // 46
};

struct GrxAdroitBuffers {
  float *qpos, *qvel, *qacc_ws;   // [N,nq] [N,nv] [N,nv]
  const float* shift;             // [N,3] offset of the board group (model.body_pos[nail_board] - its XML value)
  const float* action;            // [N,nu] (null for forward_only)
  const float *act_mean, *act_rng;  // [nu] action scaling (adroit_hammer.py:269-272)
  float* obs;                     // [N,obs_dim]
  float* reward;                  // [N]
  unsigned char* success;         // [N]
  int* status;                    // [N]
  const unsigned char* mask;      // [N] or null
};

template <class S>
struct GrxAdroit {
  typedef GrxEngine<S> E;
  // observation + reward + success from the context of the last forward pass; lane 0 writes the scalars
  GRX_MEM void grx_adroit_outputs(const GrxModel* m, const GrxAdroitTask* t, const GrxCtx* c, float* obs, float* reward, unsigned char* success, int lane_) {
    GRX_FRESH_MODEL(m, c);
    const int nq = GRX_NQC, nv = GRX_NVC, no = t->nq_obs;
    E::grx_touch_sensors(m, c, obs + no + 18, 4, lane_);   // clip(sensordata[S_nail], -1, 1): the model's only touch zone
    FOR_LANES {
      for (int i = lane; i < no; i += 64) obs[i] = c->qpos[i];
      for (int i = lane; i < 6; i += 64) obs[no + i] = fminf(1.0f, fmaxf(-1.0f, c->qvel[nv - 6 + i]));
      for (int i = lane; i < 3; i += 64) {
        obs[no + 6 + i] = c->sxpos[3 * t->site_grasp + i];
        obs[no + 9 + i] = c->xpos[3 * t->obj_body + i];
        obs[no + 15 + i] = c->sxpos[3 * t->site_target + i];
      }
    }
    LANE0 {
      float e[3];
      grx_quat2euler(c->xquat + 4 * t->obj_body, e);
      for (int k = 0; k < 3; k++) obs[no + 12 + k] = e[k];
      const float* palm = c->sxpos + 3 * t->site_grasp; const float* hamm = c->xpos + 3 * t->obj_body; const float* head = c->sxpos + 3 * t->site_tool;
      const float* nail = c->sxpos + 3 * t->site_target; const float* goal = c->sxpos + 3 * t->site_goal;
      float d_ph = 0, d_hn = 0, d_ng = 0, qv = 0;
      for (int k = 0; k < 3; k++) { d_ph += (palm[k] - hamm[k]) * (palm[k] - hamm[k]); d_hn += (head[k] - nail[k]) * (head[k] - nail[k]); d_ng += (nail[k] - goal[k]) * (nail[k] - goal[k]); }
      for (int i = 0; i < nv; i++) qv += c->qvel[i] * c->qvel[i];
      d_ph = sqrtf(d_ph); d_hn = sqrtf(d_hn); d_ng = sqrtf(d_ng); qv = sqrtf(qv);
      const int achieved = d_ng < 0.01f;
      float r = achieved ? 10.0f : -0.1f;
      if (!t->sparse_reward) {
        r = -0.1f * d_ph - d_hn - 10.0f * d_ng - 1e-2f * qv;
        if (hamm[2] > 0.04f && head[2] > 0.04f) r += 2.0f;
        if (d_ng < 0.020f) r += 25.0f;
        if (d_ng < 0.010f) r += 75.0f;
      }
      *reward = r; *success = achieved ? 1 : 0;
    }
    (void)nq;
    WAVE_SYNC();
  }

  GRX_MEM void grx_adroit_sim_world(const GrxModel* m, const GrxAdroitTask* t, GrxCtx* c, const float* action, const float* act_mean, const float* act_rng, int lane_) {
    FOR_LANES { for (int i = lane; i < GRX_NUC; i += 64) c->ctrl[i] = act_mean[i] + fminf(1.0f, fmaxf(-1.0f, action[i])) * act_rng[i]; }
    WAVE_SYNC();
    for (int s = 0; s < t->n_substeps; s++) {
      E::grx_check_state(m, c, lane_);
      E::grx_forward_euler(m, c, 1, lane_);
    }
  }
};
