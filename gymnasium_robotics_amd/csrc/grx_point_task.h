// grx_point_task.h -- PointMaze task code fused around the physics substep.
//
// Device restatement of
//   PointEnv.step ............ /root/reference/gymnasium_robotics/envs/maze/point.py:55-77  (clip action, clip qvel to +-5,
//                              do_simulation(a, frame_skip = 1), obs = qpos | qvel)
//   PointMazeEnv.step ........ envs/maze/point_maze.py:392-406 (achieved goal = xy, reward, terminated, success)
//   MazeEnv.compute_reward ... envs/maze/maze_v4.py:381-388   (dense exp(-d), sparse d <= 0.45)
//   MazeEnv.compute_terminated envs/maze/maze_v4.py:390-398
//   AntMazeEnv.step / _get_obs envs/maze/ant_maze_v5.py:295-320 (AntEnv.step [3P]: ctrl = action, do_simulation(a, 5) with the
//                              RK4 integrator of ant.xml; obs = qpos | qvel; achieved goal = xy, observation = the rest)
#pragma once
#include "grx_engine.h"

struct GrxPointTask {
  int n_substeps, sparse_reward, continuing_task;
  int agent;  // 0 = point mass (point.py), 1 = ant (gymnasium AntEnv-v5 [3P] wrapped by ant_maze_v5.py:295-320)
  double goal_radius;   // fp64, see grx_goal_distance3 (csrc/grx_fetch_task.h)
  float vel_clip;
};

struct GrxPointBuffers {
  float *qpos, *qvel, *qacc_ws;  // [N,nq] [N,nv] [N,nv]
  const float* goal;             // [N,2]
  const float* action;           // [N,nu]
  float *obs, *achieved;         // [N,nq+nv] [N,2]
  float* reward;                 // [N]
  unsigned char *success, *terminated;  // [N]
  int* status;                   // [N]
  const unsigned char* mask;     // [N] or null
  float* packed;                 // [N, obs_dim + 2 + 2 + 2] or null: out, the row [obs | achieved | desired | reward | success]
  int* split_state;              // [N, 2] or null: split step (include/grx_capi.h): [2 w] = parts of world w done in this launch, [2 w + 1] = their status flags
  int split_parts, split_pad_;   // >= 2: the launch has split_parts workgroups per world, each running its share of the substeps
};

GRX_DEV double grx_goal_distance2(const float* a, const float* b) {
  const double dx = (double)a[0] - (double)b[0], dy = (double)a[1] - (double)b[1];
  return sqrt(dx * dx + dy * dy);
}
GRX_DEV float grx_maze_reward(double d, double radius, int sparse) { return sparse ? ((d <= radius) ? 1.0f : 0.0f) : expf(-(float)d); }

template <class S>
struct GrxPoint {
  typedef GrxEngine<S> E;
  GRX_MEM void grx_point_step_world(const GrxModel* m, const GrxPointTask* t, GrxCtx* c, const float* action, float* obs, float* achieved,
                                    int lane_) {
    grx_point_sim_world(m, t, c, action, lane_);
    grx_point_outputs(m, t, c, obs, achieved, lane_);
  }
  // simulation and observation parts: the step kernel derives the output addresses between the two (nothing global live across the substeps)
  // s0 / s1: the substeps [s0, s1) of the step (a part of a split step, include/grx_capi.h grx_point_buffers.split_parts); default: all of them
  GRX_MEM void grx_point_sim_world(const GrxModel* m, const GrxPointTask* t, GrxCtx* c, const float* action, int lane_, int s0 = 0, int s1 = -1) {
    const int ant = t->agent;
    if (s1 < 0) s1 = t->n_substeps;
    FOR_LANES {
      // point: np.clip(action, -1, 1) and the velocity clip of point.py:57,73-77 (once per step: before its first substep); ant: ctrl = action (ctrlrange is applied by the actuator model)
      for (int i = lane; i < GRX_NUC; i += 64) c->ctrl[i] = ant ? action[i] : fminf(1.0f, fmaxf(-1.0f, action[i]));
      if (!ant && s0 == 0) for (int i = lane; i < GRX_NVC; i += 64) c->qvel[i] = fminf(t->vel_clip, fmaxf(-t->vel_clip, c->qvel[i]));
    }
    WAVE_SYNC();
    // Euler models: one forward+integrate per substep.  RK4 models: four forward passes per substep.  One loop, one call site.
    const int rk4 = (m->integrator == 1);
    const int total = rk4 ? 4 * s1 : s1;
    for (int it = rk4 ? 4 * s0 : s0; it < total; it++) {
      const int stage = it & 3;
      if (!rk4 || stage == 0) E::grx_check_state(m, c, lane_);
      E::grx_forward_euler(m, c, !rk4, lane_);
      if (rk4) E::grx_rk4_after_forward(m, c, stage, lane_);
    }
  }
  GRX_MEM void grx_point_outputs(const GrxModel* m, const GrxPointTask* t, GrxCtx* c, float* obs, float* achieved, int lane_) {
    const int skip = t->agent ? 2 : 0;  // AntMaze strips the xy position from the observation (it is the achieved goal)
    FOR_LANES {
      for (int i = lane; i < GRX_NQC; i += 64) { float q = c->qpos[i]; if (i >= skip) obs[i - skip] = q; if (i < 2) achieved[i] = q; }
      for (int i = lane; i < GRX_NVC; i += 64) obs[GRX_NQC - skip + i] = c->qvel[i];
    }
    WAVE_SYNC();
  }
};
