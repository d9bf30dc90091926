// grx_kitchen_task.h -- FrankaKitchen-v1 task code fused around the physics substeps.
//
// Device restatement of /root/reference/gymnasium_robotics/envs/franka_kitchen/
//   franka_env.py   FrankaRobot.step :92-112 (clip, denormalise with act_mid 0 / act_rng 2, _ctrl_velocity_limits :136-156 -- the position target is built on
//                   the PREVIOUS, noisy joint reading --, _ctrl_position_limits :158-171, do_simulation(ctrl, frame_skip = 40) [MujocoEnv, 3P]) and
//                   _get_obs :114-131 (robot qpos / qvel + ratio * amplitude * uniform(-1, 1))
//   kitchen_env.py  _get_obs :356-384 (object qpos / qvel with their noise, achieved goals = TRUE qpos slices), compute_reward :340-354 (a task is complete
//                   when |qpos[idx] - goal| < BONUS_THRESH)
// The 59 uniform(-1, 1) draws of an observation come from the host (numpy's PCG64 streams advanced bit-exactly in C: grx_sample_uniform_rows) so that a
// seeded run reproduces the reference's noise; GrxKitchenBuffers.noise == null means both noise ratios are 0.
#pragma once
#include "grx_engine.h"

#define GRX_KITCHEN_NROBOT 9
#define GRX_KITCHEN_OBS 59
#define GRX_KITCHEN_NTASK 7

struct GrxKitchenTask {
  int n_substeps, obs_dim;
  float dt;                                     // timestep * frame_skip (MujocoEnv.dt)
  float vel_lo[9], vel_hi[9], pos_lo[9], pos_hi[9];   // franka_config.xml bounds of the nine robot joints
  float noise_scale[59];                        // ratio * amplitude per observation element (kitchen_spec.noise_scales)
  int task_adr[7], task_num[7];                 // qpos slice of every task (kitchen_env.py:19-27)
  float task_goal[17];                          // the goals, concatenated (:28-37)
  float bonus_thresh;                           // 0.3
};

struct GrxKitchenBuffers {
  float *qpos, *qvel, *qacc_ws;   // [N,30] [N,29] [N,29]
  float* last_qpos;               // [N,9] in/out: the robot joint reading of the previous observation (_last_robot_qpos)
  const float* action;            // [N,9] (null for forward_only)
  const float* noise;             // [N,59] uniform(-1, 1) draws of this observation, or null (noise ratios 0)
  float* obs;                     // [N,59]
  int* completed;                 // [N] bit k: task k's qpos slice is within bonus_thresh of its goal
  int* status;                    // [N]
  const unsigned char* mask;      // [N] or null
  int* skin;                      // [N, skin_stride] broad-phase skin lists (GrxEngine::grx_collision), zeroed by the host once; or null
  int skin_stride;
  float skin_radius;
  const int* order;               // [grid] or null: workgroup j steps world order[j] (cost-ordered dispatch, include/grx_capi.h)
  int* cost;                      // [N] or null: measured duration of each world's step (80 ns units), written by step launches
  GrxLane lane;                   // the overflow lane (include/grx_capi.h grx_overflow_lane): no dropped contacts
  float* split_rows;              // [N, split_stride] or null: carrier rows of the split step [qpos | qvel | warm start] (include/grx_capi.h)
  int* split_state;               // [N, 4] or null: [4 w] = parts of world w done in this launch (< 0: re-run booked), [4 w + 1] = their status flags, [4 w + 2] = their measured duration
  int split_stride, split_parts;  // words per carrier row (>= nq + 2 nv); >= 2: the step launch has split_parts workgroups per world
};

template <class S>
struct GrxKitchen {
  typedef GrxEngine<S> E;
  GRX_MEM void grx_kitchen_outputs(const GrxModel* m, const GrxKitchenTask* t, const GrxCtx* c, const float* noise, float* obs, float* last_qpos, int* completed,
                                   int lane_) {
    GRX_FRESH_MODEL(m, c);
    const int nq = GRX_NQC, nv = GRX_NVC, nr = GRX_KITCHEN_NROBOT;
    FOR_LANES {
      for (int i = lane; i < t->obs_dim; i += 64) {
        // observation order: robot qpos (9), robot qvel (9), object qpos (nq - 9), object qvel (nv - 9)
        float v = i < nr ? c->qpos[i] : (i < 2 * nr ? c->qvel[i - nr] : (i < nr + nq ? c->qpos[i - nr] : c->qvel[i - nq]));
        if (noise) v += t->noise_scale[i] * noise[i];
        obs[i] = v;
        if (i < nr) last_qpos[i] = v;
      }
    }
    LANE0 {
      int done = 0, k = 0;
      for (int j = 0; j < GRX_KITCHEN_NTASK; j++) {
        float d2 = 0;
        for (int e = 0; e < t->task_num[j]; e++, k++) { const float d = c->qpos[t->task_adr[j] + e] - t->task_goal[k]; d2 += d * d; }
        if (sqrtf(d2) < t->bonus_thresh) done |= 1 << j;
      }
      *completed = done;
    }
    WAVE_SYNC();
  }

  // s0 / s1: the substeps [s0, s1) of the step (a part of a split step, include/grx_capi.h grx_kitchen_buffers.split_parts); default: all of them
  GRX_MEM void grx_kitchen_sim_world(const GrxModel* m, const GrxKitchenTask* t, GrxCtx* c, const float* action, const float* last_qpos, int lane_, int s0 = 0, int s1 = -1) {
    FOR_LANES {
      for (int i = lane; i < GRX_KITCHEN_NROBOT; i += 64) {
        const float a = 2.0f * fminf(1.0f, fmaxf(-1.0f, action[i]));
        const float vel = fminf(t->vel_hi[i], fmaxf(t->vel_lo[i], a));
        c->ctrl[i] = fminf(t->pos_hi[i], fmaxf(t->pos_lo[i], last_qpos[i] + vel * t->dt));
      }
    }
    WAVE_SYNC();
    if (s1 < 0) s1 = t->n_substeps;
    for (int s = s0; s < s1; s++) {
      E::grx_check_state(m, c, lane_);
      E::grx_forward_euler(m, c, 1, lane_);
      if (c->bail && grx_lane_claim(c, lane_)) break;   // a capacity overflowed and the re-run on the large tables is booked: this run will be discarded
    }
  }
};
