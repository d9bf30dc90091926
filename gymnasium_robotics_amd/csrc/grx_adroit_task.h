// grx_adroit_task.h -- Adroit hand task code fused around the physics substeps.
//
// Device restatement of /root/reference/gymnasium_robotics/envs/adroit_hand/
//   adroit_hammer.py   step :291-329, _get_obs :331-357, reset_model :372-378 (board height = the world's shift)
//   adroit_door.py     step :281-322, _get_obs :324-347, reset_model :361-373 (door frame position = the world's shift)
//   adroit_pen.py      step :288-338, _get_obs :340-365, reset_model :379-397 (target orientation = the rotation of the world's shift group)
//   adroit_relocate.py step :290-326, _get_obs :328-338, reset_model :352-371 (ball body offset = the world's shift, target site = GrxAdroitBuffers.target)
// All four: clip, a = act_mean + a * act_rng, do_simulation(a, frame_skip = 5) [MujocoEnv, 3P], observation, reward, success.
// As in the reference, the observation mixes the integrated qpos / qvel with the body / site poses and the sensor value of the LAST forward pass
// (mj_step does not recompute kinematics after integrating).
#pragma once
#include "grx_engine.h"
#include "grx_hand_task.h"   // grx_quat2euler

#define GRX_ADROIT_HAMMER 0
#define GRX_ADROIT_DOOR 1
#define GRX_ADROIT_PEN 2
#define GRX_ADROIT_RELOCATE 3

struct GrxAdroitTask {
  int n_substeps, sparse_reward;
  int kind;          // GRX_ADROIT_*
  int site[5];       // hammer: S_grasp, S_target, nail_goal, tool | door: S_grasp, S_handle | pen: eps_ball, object_top, object_bottom, target_top,
                     // target_bottom | relocate: S_grasp
  int obj_body;      // "Object" (hammer / pen / ball); door: unused
  int nq_obs;        // leading qpos entries in the observation (hammer, pen, relocate: nq - 6; door: 27 = qpos[1:-2])
  int obs_dim;       // 46 / 39 / 45 / 39
  int qadr[2];       // door: qpos index read as the hinge angle (the reference indexes qpos with the hinge's DOF address, adroit_door.py:264-266,287) and the latch
  float len[2];      // pen: pen_length, tar_length (adroit_pen.py:385-392)
};

struct GrxAdroitBuffers {
  float *qpos, *qvel, *qacc_ws;   // [N,nq] [N,nv] [N,nv]
  const float* shift;             // [N,7] pose of the world's shift group: offset t[3], rotation q[4] (model.body_pos / body_quat edits of reset_model)
  const float* target;            // [N,3] relocate: model.site_pos[target] (adroit_relocate.py:364-372); null otherwise
  const float* action;            // [N,nu] (null for forward_only)
  const float *act_mean, *act_rng;  // [nu] action scaling (adroit_hammer.py:269-272)
  float* obs;                     // [N,obs_dim]
  float* reward;                  // [N]
  unsigned char* success;         // [N]
  int* status;                    // [N]
  const unsigned char* mask;      // [N] or null
  GrxLane lane;                   // the overflow lane (include/grx_capi.h grx_overflow_lane): no dropped contacts
  const long long* compact;       // [n_compact] world indices or null: workgroup j of a launch of n_compact workgroups handles world compact[j]
  int n_compact;
  const int* order;               // [grid] or null: workgroup j steps world order[j] (cost-ordered dispatch, include/grx_capi.h); ignored by compact launches
  int* cost;                      // [N] or null: measured duration of each world's step (80 ns units), written by step launches
  float* split_rows;              // [N, split_stride] or null: carrier rows of the split step [qpos | qvel | warm start] (include/grx_capi.h)
  int* split_state;               // [N, 4] or null: [4 w] = parts of world w done in this launch (< 0: re-run booked), [4 w + 1] = their status flags, [4 w + 2] = their measured duration
  int split_stride, split_parts;  // words per carrier row (>= nq + 2 nv); >= 2: the step launch has split_parts workgroups per world
};

template <class S>
struct GrxAdroit {
  typedef GrxEngine<S> E;
  GRX_MEM float dist3(const float* a, const float* b) {
    const float x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2];
    return sqrtf(x * x + y * y + z * z);
  }
  // observation + reward + success from the context of the last forward pass; lane 0 writes the scalars
  GRX_MEM void grx_adroit_outputs(const GrxModel* m, const GrxAdroitTask* t, const GrxCtx* c, const float* target, float* obs, float* reward, unsigned char* success,
                                  int lane_) {
    GRX_FRESH_MODEL(m, c);
    const int nq = GRX_NQC, nv = GRX_NVC, no = t->nq_obs;
    if (t->kind == GRX_ADROIT_HAMMER) {
      E::grx_touch_sensors(m, c, obs + no + 18, 4, lane_);   // clip(sensordata[S_nail], -1, 1): the model's only touch zone
      FOR_LANES {
        for (int i = lane; i < no; i += 64) obs[i] = c->qpos[i];
        for (int i = lane; i < 6; i += 64) obs[no + i] = fminf(1.0f, fmaxf(-1.0f, c->qvel[nv - 6 + i]));
        for (int i = lane; i < 3; i += 64) {
          obs[no + 6 + i] = c->sxpos[3 * t->site[0] + i];
          obs[no + 9 + i] = c->xpos[3 * t->obj_body + i];
          obs[no + 15 + i] = c->sxpos[3 * t->site[1] + i];
        }
      }
      LANE0 {
        float e[3];
        grx_quat2euler(c->xquat + 4 * t->obj_body, e);
        for (int k = 0; k < 3; k++) obs[no + 12 + k] = e[k];
        const float* palm = c->sxpos + 3 * t->site[0]; const float* hamm = c->xpos + 3 * t->obj_body; const float* head = c->sxpos + 3 * t->site[3];
        const float* nail = c->sxpos + 3 * t->site[1]; const float* goal = c->sxpos + 3 * t->site[2];
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
    } else if (t->kind == GRX_ADROIT_DOOR) {
      const float* palm = c->sxpos + 3 * t->site[0]; const float* handle = c->sxpos + 3 * t->site[1];
      const float door = c->qpos[t->qadr[0]];
      FOR_LANES {
        for (int i = lane; i < no; i += 64) obs[i] = c->qpos[1 + i];                 // qpos[1:-2]
        for (int i = lane; i < 3; i += 64) { obs[no + 2 + i] = palm[i]; obs[no + 5 + i] = handle[i]; obs[no + 8 + i] = palm[i] - handle[i]; }
      }
      LANE0 {
        obs[no] = c->qpos[t->qadr[1]]; obs[no + 1] = door; obs[no + 11] = door > 1.0f ? 1.0f : -1.0f;
        const int achieved = door >= 1.35f;
        float r = achieved ? 10.0f : -0.1f;
        if (!t->sparse_reward) {
          float qv = 0;
          for (int i = 0; i < nv; i++) qv += c->qvel[i] * c->qvel[i];
          r = -0.1f * dist3(palm, handle);
          r += -0.1f * (door - 1.57f) * (door - 1.57f);
          r += -1e-5f * qv;
          if (door > 0.2f) r += 2.0f;
          if (door > 1.0f) r += 8.0f;
          if (door > 1.35f) r += 10.0f;
        }
        *reward = r; *success = achieved ? 1 : 0;
      }
    } else if (t->kind == GRX_ADROIT_PEN) {
      const float* obj = c->xpos + 3 * t->obj_body; const float* want = c->sxpos + 3 * t->site[0];
      const float* ot = c->sxpos + 3 * t->site[1]; const float* ob = c->sxpos + 3 * t->site[2]; const float* tt = c->sxpos + 3 * t->site[3]; const float* tb = c->sxpos + 3 * t->site[4];
      FOR_LANES {
        for (int i = lane; i < no; i += 64) obs[i] = c->qpos[i];
        for (int i = lane; i < 6; i += 64) obs[no + 3 + i] = c->qvel[nv - 6 + i];
        for (int i = lane; i < 3; i += 64) {
          const float oo = (ot[i] - ob[i]) / t->len[0], dd = (tt[i] - tb[i]) / t->len[1];
          obs[no + i] = obj[i]; obs[no + 9 + i] = oo; obs[no + 12 + i] = dd; obs[no + 15 + i] = obj[i] - want[i]; obs[no + 18 + i] = oo - dd;
        }
      }
      LANE0 {
        float sim = 0;
        for (int k = 0; k < 3; k++) sim += ((ot[k] - ob[k]) / t->len[0]) * ((tt[k] - tb[k]) / t->len[1]);
        const float gd = dist3(obj, want);
        const int achieved = gd < 0.075f && sim > 0.95f;
        float r = achieved ? 10.0f : -0.1f;
        if (!t->sparse_reward) {
          r = -gd + sim;
          if (gd < 0.075f && sim > 0.9f) r += 10.0f;
          if (gd < 0.075f && sim > 0.95f) r += 50.0f;
          if (obj[2] < 0.075f) r -= 5.0f;
        }
        *reward = r; *success = achieved ? 1 : 0;
      }
    } else {   // GRX_ADROIT_RELOCATE
      const float* obj = c->xpos + 3 * t->obj_body; const float* palm = c->sxpos + 3 * t->site[0];
      FOR_LANES {
        for (int i = lane; i < no; i += 64) obs[i] = c->qpos[i];
        for (int i = lane; i < 3; i += 64) { obs[no + i] = palm[i] - obj[i]; obs[no + 3 + i] = palm[i] - target[i]; obs[no + 6 + i] = obj[i] - target[i]; }
      }
      LANE0 {
        const float tg[3] = {target[0], target[1], target[2]};
        const float gd = dist3(obj, tg);
        const int achieved = gd < 0.1f;
        float r = achieved ? 10.0f : -0.1f;
        if (!t->sparse_reward) {
          r = -0.1f * dist3(palm, obj);
          if (obj[2] > 0.04f) { r += 1.0f; r += -0.5f * dist3(palm, tg); r += -0.5f * dist3(obj, tg); }
          if (gd < 0.1f) r += 10.0f;
          if (gd < 0.05f) r += 20.0f;
        }
        *reward = r; *success = achieved ? 1 : 0;
      }
    }
    (void)nq;
    WAVE_SYNC();
  }

  // s0 / s1: the substeps [s0, s1) of the step (a part of a split step, include/grx_capi.h grx_adroit_buffers.split_parts); default: all of them
  GRX_MEM void grx_adroit_sim_world(const GrxModel* m, const GrxAdroitTask* t, GrxCtx* c, const float* action, const float* act_mean, const float* act_rng, int lane_, int s0 = 0, int s1 = -1) {
    FOR_LANES { for (int i = lane; i < GRX_NUC; i += 64) c->ctrl[i] = act_mean[i] + fminf(1.0f, fmaxf(-1.0f, action[i])) * act_rng[i]; }
    WAVE_SYNC();
    if (s1 < 0) s1 = t->n_substeps;
    for (int s = s0; s < s1; s++) {
      E::grx_check_state(m, c, lane_);
      E::grx_forward_euler(m, c, 1, lane_);
      if (c->bail && grx_lane_claim(c, lane_)) break;   // a capacity overflowed and the re-run on the large tables is booked: this run will be discarded
    }
  }
};
