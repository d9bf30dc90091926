/* grx_capi.h -- C ABI of the MI355X batched-physics library (libgrx_hip.so).
 *
 * The reference has no native FFI: its hot path crosses Python -> C only at the
 * third-party MuJoCo bindings.  Each entry point below names the reference call
 * site(s) it replaces (paths relative to /root/reference/gymnasium_robotics):
 *
 *   grx_model_create ........ mujoco.MjModel.from_xml_path + MjData      envs/robot_env.py:293-294
 *                              (the MJCF itself is compiled on the host by
 *                              gymnasium_robotics_amd/mjcf/compiler.py; this uploads the tables)
 *   grx_model_set_table ..... in-place MjModel edits, e.g. eq_data      utils/mujoco_utils.py:74-80
 *   grx_fetch_step .......... BaseRobotEnv.step for N worlds            envs/robot_env.py:114-152
 *                              = _set_action (fetch/fetch_env.py:85-105,305-310)
 *                              + mujoco.mj_step(nstep=20)                (envs/robot_env.py:341)
 *                              + _step_callback (fetch/fetch_env.py:295-303)
 *                              + _get_obs / _is_success / compute_reward (fetch/fetch_env.py:74-80,107-170,312-360)
 *   grx_fetch_forward ....... mujoco.mj_forward after a reset + _get_obs fetch/fetch_env.py:401, envs/robot_env.py:183
 *                              (nstep > 0: the raw mj_step settle loop of _env_setup, fetch/fetch_env.py:419-420)
 *   grx_fetch_compute_reward  GoalEnv.compute_reward on a batch (HER)   fetch/fetch_env.py:74-80, core.py:45-67
 *   grx_maze_sample_resets_device ... np_random.integers / uniform draws of MazeEnv.reset       maze/maze_v4.py:299-358
 *   grx_adroit_sample_resets_device . np_random.uniform draws of the Adroit reset_model methods  adroit_hammer.py:374-376, adroit_door.py:362-370, adroit_relocate.py:353-372
 *   grx_fetch_sample_resets[_device] . np_random.uniform draws of _reset_sim/_sample_goal  fetch/fetch_env.py:153-166,388-391 (host / on the device)
 *   grx_fetch_commit_rows ........... commit of a Fetch reset that ran BESIDE the step kernel into staged rows      fetch/fetch_env.py:375-402, envs/robot_env.py:154-186
 *   grx_adroit_commit_rows .......... the same for the Adroit tasks (reset_model + mj_forward on staged rows)       adroit_hammer.py:372-378, adroit_door.py:359-371, adroit_pen.py:379-397, adroit_relocate.py:354-373
 *
 * All array arguments are plain device (HBM) pointers; rows are world-major.  `stream` is a
 * hipStream_t passed as void*.  Every function returns 0 on success, a negative value on error
 * (message via grx_last_error()).
 */
#ifndef GRX_CAPI_H
#define GRX_CAPI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct grx_model grx_model;

/* The OVERFLOW LANE (all families): the shape-specialised ("fast") step kernels have fixed table capacities (contacts / constraint rows / Jacobian-pool words per
 * world) sized so that 5-10 worlds share a CU.  A world that exceeds one in some substep would have to drop contacts, which the reference never does
 * (mujoco.mj_step, envs/robot_env.py:341).  Instead:
 *   fast kernel   (entry_count != NULL): leaves the worlds with skip[w] != 0 alone; a world that overflows writes NOTHING (state, outputs, status keep their
 *                 pre-step values), appends itself to entry_list (atomic counter entry_count) and stops simulating.  A world that does not overflow but comes
 *                 within the soft_* thresholds of a capacity (the caller passes a fraction of the tables) commits its step as usual and appends itself to
 *                 next_list: it moves to the lane BEFORE it can overflow, so that most worlds enter without a serialised re-run.
 *   large kernel  (list != NULL; the SAME model created with larger capacities, CompiledModel.with_capacity): one workgroup per entry of the COMPACTED list of
 *                 worlds list[0 .. min(*count, grid)) steps its world on the large tables, commits the result, checks in every substep whether the FAST
 *                 kernel's tables (soft_*) would have overflowed and keeps the world's ticket: ttl[w] = ttl_init after a step in which they would have, else
 *                 ttl[w] - 1; while ttl[w] > 0 the world is appended to next_list / next_flags: it stays in the lane.
 * One step: the caller zeroes the two counters and next_flags; launches the large kernel over the lane's current list on a second stream CONCURRENTLY with the fast
 * kernel (skip = the current flags); when both are done, the large kernel once more over entry_list (the worlds that entered the lane in this step: the only
 * serialised part, a launch of zero worlds otherwise); then next_* become the current list / flags.  Nothing crosses the host: which worlds are in the lane is
 * known to the kernels only.  All pointers are device pointers; an all-zero struct switches the mechanism off (capacity overflows then drop contacts and raise
 * GRX_STATUS_CON_OVERFLOW / GRX_STATUS_EFC_OVERFLOW, sticky in `status`).
 * LIMITS -- "no dropped contacts" is conditional: the large tables are finite too (the packaged environments create them with 64 contacts / 256 rows / 4 080 pool words,
 * the kitchen 400 / 8 160; a world's contact list is at most 64 long: one lane per contact, and row offsets are 14 bits).  A world that exceeds the LARGE tables, more
 * than 256 first-time entrants in one step, or a lane longer than the grid of its launch keep stepping with the excess contacts of that substep dropped and the sticky
 * status bit raised -- never silently: count `status` (bench.py: capacity_overflow_worlds, 0 in every measured workload). */
typedef struct grx_overflow_lane {
  const unsigned char* skip;            /* fast kernel, [N]: worlds in the lane this step */
  int* entry_count; int* entry_list;    /* fast kernel: out, the worlds that overflowed ([1], [N]); an entry = world id | (1 << 30 when it needs the LARGE tables: claimed by a lane launch that
                                         * already runs the middle ones, grx_fetch_buffers.handoff_large -- polling workgroups of such a launch give these back to the entry launch) */
  const int* list; const int* count;    /* large kernel: in, the worlds to step ([N], [1]) */
  unsigned char* next_flags; int* next_count; int* next_list;   /* large kernel: out, the lane of the next step ([N], [1], [N]) */
  signed char* ttl;                     /* large kernel: in/out [N] */
  int soft_maxefc, soft_jpool, soft_maxcon, ttl_init;           /* large kernel: the fast kernel's capacities; steps a world stays after its last soft overflow */
  int grid;                             /* large kernel: workgroups of the launch = the entries of `list` it covers (one each); 0 = 64 */
  int entry_cap, next_cap;              /* capacities of entry_list / next_list = the grids of the launches that will walk them.  A world that finds next_list full stays on
                                         * (returns to) the fast kernel; a world that overflows when entry_list is full goes on with the excess contacts dropped and the sticky
                                         * status flag says so (more than entry_cap worlds overflowing for the first time in ONE step) */
  /* Entrants without the serialised re-run (optional, all NULL / 0 = off).  fast kernel: ready[idx] <- 1 when entry idx of entry_list is published (idx < ready_cap), every
   * workgroup adds 1 to *progress when it ends.  large kernel, standing launch: the LAST poll_grid workgroups of its grid poll -- workgroup p sleeps until ready[p] == 1, claims
   * the entry (ready[p] <- 2) and steps world poll_list[p] (= this step's entry_list) on the large tables while the fast launch is still running; it gives up when
   * *progress == progress_total (the fast launch's workgroups) or after a bounded number of polls.  large kernel, entry launch (poll_grid == 0, ready != NULL): skips the entries
   * a polling workgroup has claimed. */
  int* ready; int* progress; const int* poll_list;
  int ready_cap, progress_total, poll_grid, pad_;
} grx_overflow_lane;

/* mirrors struct GrxFetchTask (csrc/grx_fetch_task.h) */
typedef struct grx_fetch_task {
  int has_object, block_gripper, n_substeps, sparse_reward;
  int grip_body;
  float grip_relpos[3], grip_relquat[4];
  int site_grip, site_obj;
  int jq_rf, jq_lf, jd_rf, jd_lf;
  int obs_dim, goal_dim;
  float dt;
  double distance_threshold; /* fp64: success / sparse reward are decided in the reference's arithmetic on the returned goals (fetch_env.py:74-80,168-170) */
} grx_fetch_task;

/* bits of the per-world `status` words the step kernels write (0 = healthy; csrc/grx_engine.h GRX_ST_*).
 * Bits 0-15 hold the flags of the LAST launch; bits 16-31 hold the same flags OR-accumulated over every launch since the caller last
 * zeroed the buffer (sticky: `status >> 16` answers "did this world ever drop a contact / hit a bad number since I last looked"). */
enum grx_status_bits {
  GRX_STATUS_BADNUM = 1,        /* a NaN / overflowing coordinate was found: the world was reset to qpos0 (MuJoCo's mj_checkPos / mj_checkVel behaviour) */
  GRX_STATUS_CON_OVERFLOW = 2,  /* more contacts than the model's contact capacity in some substep: the excess was dropped for that substep */
  GRX_STATUS_EFC_OVERFLOW = 4,  /* more constraint rows / Jacobian-pool words than the capacity: the excess contacts were dropped for that substep */
  GRX_STATUS_FACTOR = 8         /* a non-positive pivot in a Cholesky factorisation (fallback solver path) */
};

/* mirrors struct GrxFetchBuffers: device pointers, world-major rows */
typedef struct grx_fetch_buffers {
  float *qpos, *qvel, *qacc_ws, *mocap; /* [N,nq] [N,nv] [N,nv] [N,7*nmocap] */
  float* aux;                           /* [N,8]  */
  const float* goal;                    /* [N,3]  */
  const float* action;                  /* [N,4]  */
  float *obs, *achieved;                /* [N,obs_dim] [N,3] */
  float* reward;                        /* [N]    */
  unsigned char* success;               /* [N]    */
  int* status;                          /* [N]  GRX_ST_* bits, 0 = healthy */
  const unsigned char* mask;            /* [N] or NULL */
  /* Optional load balancing (both NULL = off).  A world costs between 0.9x and 2.2x the median (contacts, Newton iterations) and a
   * launch of 4096 worlds is only ~2 worlds per resident wave slot, so the launch ends when the unluckiest slot does.  `cost` receives
   * every world's cost estimate (12 x Newton iterations of the step + 24 x contacts of its last substep ~ microseconds above the base); `order[b]` names the world workgroup b handles, so the caller can start the expensive
   * worlds first (workgroups start in index order; b & 7 is the XCD: keep a world in the XCD slice that owns its neighbours). */
  const int* order;                     /* [8 * ceil(N / 8)] or NULL; entries >= N are idle workgroups */
  int* cost;                            /* [N] or NULL */
  float* packed;                        /* [N, obs_dim+3+3+2] or NULL: the row [obs | achieved | desired | reward | success], written by the step kernel itself so
                                         * that the cross-rank exchange (RCCL all-gather) ships one buffer and needs no pack kernels */
  float* hullcache;                     /* [N, 90] or NULL (zero-initialised; 21 words of separating directions + 69 words of support-vertex guesses of persistent hull contacts): the world's cache of separating directions of its hull-vs-convex pairs (engine: GrxCtx::meshcache), carried from one
                                         * env.step() to the next.  A remembered direction is re-verified before it is trusted (it proves "no contact", exactly what the portal
                                         * search would report), so the rows never change a result: they save the search every launch otherwise starts with. */
  /* MID-STEP HAND-OFF (round 6; NULL / 0 = off).  A launch that has an entry list to claim from (lane.entry_count) and these rows does not re-run a world it cannot finish from
   * the first substep: it stops at the START of the substep in question -- a table capacity is exceeded, or the kernel carries no hull routine (the fast FetchPickAndPlace
   * kernel: 168 VGPRs, a third wave per SIMD) and a hull pair has passed the bounding-box filter; nothing of that substep has touched the state -- writes the world's row
   * [substep + 1 | status flags | ctrl[nu] | mocap[7 nmocap] | qpos[nq] | qvel[nv] | qacc_warmstart[nv]] and publishes the entry (world id, bit 30 = handoff_large).  A lane
   * launch (lane.list != NULL) that finds word 0 of a world's row non-zero restores the row, clears the word and resumes AT that substep; results are bit-identical to a
   * step run by the lane's kernel alone (tests/test_gpu_fetch.py::test_handoff_is_the_full_kernels_rollout).  Rows must be zero before the first launch. */
  float* handoff;                       /* [N, handoff_stride] or NULL */
  int handoff_stride;                   /* words per row, >= 2 + nu + 7 nmocap + nq + 2 nv */
  int handoff_large;                    /* 1: entries claimed by THIS launch need the large tables (it runs the middle ones): polling workgroups give them back to the entry launch */
  /* SPLIT STEP (round 6; step launches only, not lane launches, not the hull-less fast kernel).  A launch of N worlds on S wave slots ends Sum(work) / S + the duration of the LAST
   * workgroup started -- a whole world-step (0.85 ms of FetchPickAndPlace's 2.9 ms launch at 4 096 worlds).  With split_parts = P >= 2 the launch has P workgroups per world,
   * workgroup p * grid + g running substeps [p T / P, (p + 1) T / P) of the world of slot g (same XCD for every part of a world): part p < P - 1 ends by writing the world's hand-off row
   * (the `handoff` rows above serve as the carrier: same layout, same bit-identical resume as the mid-step hand-off) and publishing p + 1 in split_state[2 w]; part p > 0 waits for
   * that word (its predecessor was dispatched before it: workgroups of one XCD start in index order), resumes from the row; only the last part writes state rows and outputs.  A part
   * that exceeds a table books the world's re-run as usual and marks the word negative (the later parts return).  split_state[2 w + 1] carries the measured duration of the
   * earlier parts (cost).  Results are bit-identical to the unsplit launch (tests/test_gpu_fetch.py::test_split_step_is_the_plain_step).  Zeroed before the first launch. */
  int* split_state;                     /* [N, 2] or NULL */
  int split_parts;                      /* 0 / 1: one workgroup per world */
  int split_pad_;
  grx_overflow_lane lane;               /* capacity overflows are re-run on larger tables instead of dropping contacts: see grx_overflow_lane above and its LIMITS */
} grx_fetch_buffers;

/* mirrors struct GrxPointTask / GrxPointBuffers (csrc/grx_point_task.h) */
typedef struct grx_point_task {
  int n_substeps, sparse_reward, continuing_task;
  int agent; /* 0 = PointMaze particle, 1 = AntMaze ant (RK4, obs without xy) */
  double goal_radius; /* fp64 (maze_v4.py:381-388) */
  float vel_clip;
} grx_point_task;
typedef struct grx_point_buffers {
  float *qpos, *qvel, *qacc_ws; /* [N,nq] [N,nv] [N,nv] */
  const float* goal;            /* [N,2] */
  const float* action;          /* [N,nu] */
  float *obs, *achieved;        /* [N,nq+nv] (ant: [N,nq+nv-2]) [N,2] */
  float* reward;                /* [N] */
  unsigned char *success, *terminated; /* [N] */
  int* status;                  /* [N] */
  const unsigned char* mask;    /* [N] or NULL */
  float* packed;                /* [N, obs_dim+2+2+2] or NULL: [obs | achieved | desired | reward | success] (see grx_fetch_buffers.packed) */
  /* SPLIT STEP (round 6; grx_fetch_buffers.split_parts explains the mechanism).  8 192 ant worlds on 3 072 wave slots are 2.67 rounds of workgroups, i.e. three: the last one runs
   * on a chip that is two thirds full.  With split_parts = P (2 <= P <= frame_skip) the launch has P workgroups per world, each running its share of the substeps and handing the
   * world on through its own state row (at a substep boundary qpos / qvel / warm start are the whole state; the RK4 stages of a substep stay in one part); only the last part writes
   * outputs.  Results are bit-identical to the one-workgroup launch (tests/test_gpu_maze.py::test_split_step_is_the_plain_step).  split_state: zeroed before the first launch. */
  int* split_state;             /* [N, 2] or NULL */
  int split_parts;              /* 0 / 1: one workgroup per world */
  int split_pad_;
} grx_point_buffers;

/* mirrors struct GrxHandTask / GrxHandBuffers (csrc/grx_hand_task.h): Shadow Dexterous Hand reach task */
typedef struct grx_hand_task {
  int n_substeps, sparse_reward;
  int site[5];   /* fingertip sites, envs/shadow_dexterous_hand/reach.py:8-14 order */
  int palm_body; /* body used by _sample_goal (reach.py:413-416) */
  double distance_threshold; /* fp64 (reach.py:92-100) */
  int kind;      /* 0 = HandReach (goal dim 15, obs nq+nv+15); 1 = HandManipulate* (goal dim 7, obs 2 nq_robot + 6 + 7) */
  int nq_robot, obj_qadr, obj_dadr;      /* kind 1: robot joints come first; qpos / dof address of object:joint */
  int ignore_position, ignore_rotation;  /* kind 1: target_position / target_rotation == "ignore" (manipulate.py:92-97) */
  float rotation_threshold;              /* kind 1: manipulate.py:33 */
  int ignore_z;   /* kind 1: ignore_z_target_rotation (pen variants, manipulate.py:100-108) */
  int touch_mode; /* kind 1: 0 none, 1 sensordata, 2 boolean, 3 log(x+1): touch values appended to the observation (manipulate_touch_sensors.py:113-137) */
} grx_hand_task;
typedef struct grx_hand_buffers {
  float *qpos, *qvel, *qacc_ws; /* [N,nq] [N,nv] [N,nv] */
  const float* goal;            /* [N,goal_dim] */
  const float* action;          /* [N,nu] (may be NULL when forward_only) */
  float *obs, *achieved;        /* [N,obs_dim] [N,goal_dim] */
  float* palm;                  /* [N,3] */
  float* reward;                /* [N] */
  unsigned char* success;       /* [N] */
  int* status;                  /* [N] */
  const unsigned char* mask;    /* [N] or NULL */
  const int* order;             /* [8 * ceil(N / 8)] or NULL: cost-ordered dispatch, as in grx_fetch_buffers */
  int* cost;                    /* [N] or NULL */
  float* packed;                /* [N, obs_dim+2*goal_dim+2] or NULL: [obs | achieved | desired | reward | success] */
  grx_overflow_lane lane;          /* capacity overflows re-run on larger tables: see grx_overflow_lane and its LIMITS */
  /* SPLIT STEP (round 6; as grx_adroit_buffers.split_parts): split_parts = P (2 <= P <= 8) workgroups per world, each running its share of the substeps and handing the world on through
   * its row of split_rows; plain step launches only (forward_only = 0, not grx_hand_step_repeat, not the lane's).  Bit-identical to split_parts = 0
   * (tests/test_gpu_hand.py::test_split_step_is_the_plain_step). */
  float* split_rows;               /* [N, split_stride >= nq + 2 nv] or NULL */
  int* split_state;                /* [N, 4] or NULL, zeroed before the first launch */
  int split_stride, split_parts;
} grx_hand_buffers;

/* mirrors struct GrxAdroitTask / GrxAdroitBuffers (csrc/grx_adroit_task.h): AdroitHandHammer / Door / Pen / Relocate */
#define GRX_ADROIT_HAMMER 0
#define GRX_ADROIT_DOOR 1
#define GRX_ADROIT_PEN 2
#define GRX_ADROIT_RELOCATE 3
typedef struct grx_adroit_task {
  int n_substeps, sparse_reward;
  int kind;     /* GRX_ADROIT_* */
  int site[5];  /* hammer: S_grasp, S_target, nail_goal, tool (adroit_hammer.py:264-268) | door: S_grasp, S_handle (adroit_door.py:267-268) |
                   pen: eps_ball, object_top, object_bottom, target_top, target_bottom (adroit_pen.py:259-263) | relocate: S_grasp (adroit_relocate.py:259) */
  int obj_body; /* "Object" (hammer / pen / ball); unused by the door */
  int nq_obs;   /* leading qpos entries in the observation: nq - 6 (door: nq - 3 = qpos[1:-2]) */
  int obs_dim;  /* 46 / 39 / 45 / 39 */
  int qadr[2];  /* door: the qpos index read as the hinge angle (the reference indexes qpos with jnt_dofadr, adroit_door.py:264-266) and the latch (nq - 1) */
  float len[2]; /* pen: pen_length, tar_length (adroit_pen.py:385-392) */
} grx_adroit_task;
typedef struct grx_adroit_buffers {
  float *qpos, *qvel, *qacc_ws;    /* [N,nq] [N,nv] [N,nv] */
  const float* shift;              /* [N,7] per-world pose of the model's shift group: offset t[3] (= model.body_pos[body] - XML value: adroit_hammer.py:374-376,
                                      adroit_door.py:362-370, adroit_relocate.py:358-363) and rotation q[4] (model.body_quat[target], adroit_pen.py:381; identity otherwise) */
  const float* target;             /* [N,3] relocate: model.site_pos[target] (adroit_relocate.py:364-372); NULL for the other tasks */
  const float* action;             /* [N,nu] (may be NULL when forward_only) */
  const float *act_mean, *act_rng; /* [nu] (adroit_hammer.py:269-272) */
  float* obs;                      /* [N,obs_dim] */
  float* reward;                   /* [N] */
  unsigned char* success;          /* [N] */
  int* status;                     /* [N] */
  const unsigned char* mask;       /* [N] or NULL */
  grx_overflow_lane lane;          /* capacity overflows re-run on larger tables: see grx_overflow_lane and its LIMITS */
  const int64_t* compact;          /* [n_compact] device world indices or NULL: a launch of n_compact workgroups, workgroup j handles world compact[j] (the reset-time forward pass of the ~0.5 %
                                      of the worlds an env.step() resets: a masked launch over all N worlds spends 0.5 ms dispatching workgroups that return at once) */
  int n_compact;
  const int* order;                /* [grid] or NULL: cost-ordered dispatch, see grx_kitchen_buffers.order; ignored by compact launches */
  int* cost;                       /* [N] or NULL: measured duration of each world's step (80 ns units), written by step launches */
  /* SPLIT STEP (round 6; grx_fetch_buffers.split_parts explains the mechanism): split_parts = P (2 <= P <= frame_skip) workgroups per world, each running its share of the substeps and
   * handing the world on through its row of split_rows ([qpos | qvel | warm start]; the state rows stay untouched until the last part, so a world that exceeds a table is re-run from
   * them as usual).  An Adroit world's cost changes from step to step (the hammer meets the nail for a few steps), so the cost order predicts it poorly: launches ended 15 - 40 % after
   * the mean of their wave slots (profiles/tail_probe_r06.txt).  Step launches only (not compact, not forward_only, not the lane's).  Bit-identical to split_parts = 0
   * (tests/test_gpu_adroit.py::test_split_step_is_the_plain_step).  With polling workgroups the standing lane launch's grx_overflow_lane.progress_total must count every part
   * (split_parts x the grid of a plain launch): every workgroup of the fast launch reports its end. */
  float* split_rows;               /* [N, split_stride] or NULL */
  int* split_state;                /* [N, 4] or NULL, zeroed before the first launch */
  int split_stride, split_parts;
} grx_adroit_buffers;

/* mirrors struct GrxKitchenTask / GrxKitchenBuffers (csrc/grx_kitchen_task.h): FrankaKitchen-v1 */
typedef struct grx_kitchen_task {
  int n_substeps, obs_dim;          /* 40 (franka_env.py:54), 59 (kitchen_env.py:109) */
  float dt;                         /* MujocoEnv.dt = timestep * frame_skip */
  float vel_lo[9], vel_hi[9], pos_lo[9], pos_hi[9];   /* franka_config.xml: vel_bound / pos_bound of the nine robot joints (franka_env.py:136-171) */
  float noise_scale[59];            /* noise ratio * amplitude per observation element (franka_env.py:118-127, kitchen_env.py:361-369) */
  int task_adr[7], task_num[7];     /* qpos slices OBS_ELEMENT_INDICES (kitchen_env.py:19-27) */
  float task_goal[17];              /* OBS_ELEMENT_GOALS, concatenated (:28-37) */
  float bonus_thresh;               /* BONUS_THRESH 0.3 (:38) */
} grx_kitchen_task;
typedef struct grx_kitchen_buffers {
  float *qpos, *qvel, *qacc_ws;     /* [N,30] [N,29] [N,29] */
  float* last_qpos;                 /* [N,9] in/out: FrankaRobot._last_robot_qpos, the (noisy) joint reading of the previous observation */
  const float* action;              /* [N,9] (may be NULL when forward_only) */
  const float* noise;               /* [N,59] uniform(-1, 1) draws of this observation (grx_sample_uniform_rows) or NULL */
  float* obs;                       /* [N,59] */
  int* completed;                   /* [N] bit k: |qpos[task k] - goal k| < bonus_thresh (compute_reward's per-task test, kitchen_env.py:346-351) */
  int* status;                      /* [N] */
  const unsigned char* mask;        /* [N] or NULL */
  int* skin;                        /* [N, skin_stride] zero-initialised scratch the library owns between calls (broad-phase skin list: 4 + 3 ngeom + ndevpair words
                                       per world), or NULL: every substep sweeps the full candidate list.  Results are identical either way. */
  int skin_stride;
  float skin_radius;                /* metres by which the broad-phase radius is inflated when a world's list is built (0.1) */
  /* COST-ORDERED DISPATCH (round 6; the Fetch family has had it since round 2: grx_fetch_buffers.order / .cost).  A launch of N worlds on S wave slots ends when the queue runs
   * dry PLUS the longest remaining world; a kitchen world takes 2.9 ms at the median and 11.5 ms at the worst (profiles/stragglers_r06_kitchen.txt), so a launch in index order
   * ends ~5 ms (14 %) after the slots' mean.  With `order` workgroup j steps world order[j]; every step launch writes each world's measured duration to cost[w]
   * and grx_order_by_cost turns that into the next launch's order (longest first, per XCD slice).  Results do not depend on the order. */
  const int* order;                 /* [grid] or NULL */
  int* cost;                        /* [N] or NULL: 80 ns units */
  grx_overflow_lane lane;          /* capacity overflows re-run on larger tables: see grx_overflow_lane and its LIMITS */
  /* SPLIT STEP (round 6; as grx_adroit_buffers.split_parts): split_parts = P (2 <= P <= 8) workgroups per world, each running its share of the 40 substeps and handing the world on
   * through its row of split_rows.  A part does not read the skin list another part (another CU) wrote: it rebuilds it in its first substep (one full candidate sweep per part).
   * Bit-identical to split_parts = 0 (tests/test_gpu_kitchen.py::test_split_step_is_the_plain_step).  Step launches only. */
  float* split_rows;               /* [N, split_stride >= nq + 2 nv] or NULL */
  int* split_state;                /* [N, 4] or NULL, zeroed before the first launch */
  int split_stride, split_parts;
} grx_kitchen_buffers;

/* At most 32 models (descriptor slots in constant memory) exist per process at a time; creation beyond that fails with an error, destroy frees the slot.  A model is immutable
 * once created apart from grx_model_set_table, so callers may share one handle between any number of world batches (the Python side does: _native.acquire_model). */
int grx_model_create(const int32_t* H, int nH, const int32_t* I, int nI, const double* F, int nF, int device, grx_model** out);
int grx_model_destroy(grx_model* m);
int grx_model_set_table(grx_model* m, const char* name, const double* data, int n);
int grx_model_lds_bytes(const grx_model* m);
int grx_model_dim(const grx_model* m, const char* name);   /* "nq" "nv" "nu" "nbody" "nmocap" "ndevpair"; "shape" = id of the shape-specialised kernel (0 generic); "handtree"; -1 unknown */

int grx_fetch_step(const grx_model* m, const grx_fetch_task* task, const grx_fetch_buffers* buf, int n_worlds, void* stream);
int grx_fetch_forward(const grx_model* m, const grx_fetch_task* task, const grx_fetch_buffers* buf, int n_worlds, int nstep, void* stream);
/* Episode reset of a COMPACTED list of worlds: _reset_sim (fetch/fetch_env.py:375-402: initial state, object xy) + _sample_goal (:153-166)
 * + mj_forward + _get_obs (envs/robot_env.py:154-186) for the n_reset worlds idx[0..n_reset): one workgroup per listed world, shape-specialised
 * like the step kernel.  The draws come from grx_fetch_sample_resets_device (a kernel on the same stream; or grx_fetch_sample_resets on the host, staged with one
 * asynchronous copy from pinned memory); all arrays here are DEVICE pointers: nothing on the host has to wait for the device. */
typedef struct grx_fetch_reset_args {
  const int* idx;        /* [n_reset] world indices */
  const float* samples;  /* [n_reset,5] object x, y, goal x, y, z */
  const float *init_qpos, *init_qvel, *init_mocap; /* [nq] [nv] [7*nmocap]: the state _env_setup left (fetch_env.py:404-428) */
  int obj_qadr;          /* qpos address of object0:joint, -1 for the tasks without object */
  int keep_outcome;      /* != 0 (same-step autoreset): reward[w] / success[w] and the last two words of the packed row keep the finished episode's values */
  float* final_packed;   /* [N, obs_dim + 8] or NULL: before world w's packed row is overwritten by the reset observation, the row the step kernel wrote (the TERMINAL
                          * observation / achieved goal / goal / reward / success of the finished episode) is parked in final_packed[w]: info["final_obs"] and the last
                          * transition of the episode for the HER buffer (grx_her_args.term_rows) read it from there */
} grx_fetch_reset_args;
int grx_fetch_reset(const grx_model* m, const grx_fetch_task* task, const grx_fetch_buffers* buf, const grx_fetch_reset_args* args, int n_reset, void* stream);
/* order <- the dispatch order for the next step launch from the per-world costs the last one wrote (grx_fetch_buffers.cost /
 * grx_hand_buffers.cost): per XCD slice of n_worlds / 8 contiguous worlds, decreasing cost.  With `ema` the key is the exponential moving
 * average ema <- (1 - alpha) ema + alpha cost, kept in the caller's buffer (measured best around alpha = 0.15).  n_worlds must be a multiple of 8. */
int grx_order_by_cost(const int* cost, float* ema /* [N] in/out or NULL */, float alpha, int n_worlds, int* order, void* stream);
/* The same for a launch of at most two worlds per wave slot (slots_per_xcd = worlds resident per XCD: worlds per CU x 32; 0 = the plain order above): the worlds
 * that will be a slot's THIRD world (as many as predicted stragglers hold a slot for the whole launch) are chosen and placed so that those slots run three cheap worlds
 * back to back instead of two median worlds and one more (csrc/grx_kernels.hip, grx_order_kernel). */
int grx_order_by_cost_slots(const int* cost, float* ema, float alpha, int n_worlds, int slots_per_xcd, int* order, void* stream);
int grx_fetch_compute_reward(const float* achieved, const float* desired, int64_t batch, double distance_threshold, int sparse,
                             float* reward_out, void* stream);
/* Maze family (PointMaze and AntMaze, selected by task->agent; AntMazeEnv.step: envs/maze/ant_maze_v5.py:295-310).
 * PointMaze: PointMazeEnv.step for N worlds = clip + velocity clip + mj_step(1) + obs/reward/terminated/success
 * (envs/maze/point.py:55-77, envs/maze/point_maze.py:392-406); batched MazeEnv.compute_reward (envs/maze/maze_v4.py:381-388). */
int grx_point_step(const grx_model* m, const grx_point_task* task, const grx_point_buffers* buf, int n_worlds, void* stream);
int grx_maze_compute_reward(const float* achieved, const float* desired, int64_t batch, double goal_radius, int sparse, float* reward_out,
                            void* stream);

/* Shadow hand reach: BaseRobotEnv.step with MujocoHandEnv._set_action (absolute position control) + mj_step(n_substeps) +
 * MujocoHandReachEnv._get_obs / compute_reward / _is_success for N worlds (envs/robot_env.py:114-152,
 * envs/shadow_dexterous_hand/hand_env.py:36-58, reach.py:92-134,398-428).  forward_only != 0: mj_forward + outputs (reset path,
 * robot_env.py:300-313, and _env_setup, reach.py:408-416).  forward_only takes 0 or 1 (values >= 2 are the repeat count of grx_hand_step_repeat below: call that instead).
 * grx_goal_compute_reward: batched compute_reward for dim-vector goals. */
int grx_hand_step(const grx_model* m, const grx_hand_task* task, const grx_hand_buffers* buf, int n_worlds, int forward_only, void* stream);
/* `repeat` consecutive env.step()s of the same action rows in ONE launch -- the loop of MujocoManipulateEnv._reset_sim, `for _ in range(10): self._set_action(np.zeros(20));
 * mj_step(nstep=n_substeps)` (envs/shadow_dexterous_hand/manipulate.py:205-224), for the worlds of `buf`.  Each repetition is the whole step of grx_hand_step, its state rows written
 * and read back as between two launches: results are bit-identical to `repeat` calls of grx_hand_step (tests/test_gpu_manipulate.py::test_repeat_launch_is_the_sequence_of_launches),
 * outputs are the last repetition's, status flags accumulate in the sticky half.  Why: beside a step kernel that fills the chip every launch of a reset's 164 worlds waits for wave
 * slots and ends with its slowest world; ten dependent launches took 22 - 34 ms, more than the two env.step()s the overlapped reset has (round 6).  Not for launches with an overflow lane. */
int grx_hand_step_repeat(const grx_model* m, const grx_hand_task* task, const grx_hand_buffers* buf, int n_worlds, int repeat, void* stream);
/* AdroitHand{Hammer,Door,Pen,Relocate}Env.step for N worlds: clip + a = act_mean + a * act_rng + do_simulation(a, 5) (= mj_step x 5 with the
 * noslip post-solver, adroit_assets.xml:3) + _get_obs + reward + success (envs/adroit_hand/adroit_hammer.py:291-357, adroit_door.py:281-347,
 * adroit_pen.py:288-365, adroit_relocate.py:290-338); forward_only != 0: the reset path (set_state -> mj_forward, _get_obs) */
int grx_adroit_step(const grx_model* m, const grx_adroit_task* task, const grx_adroit_buffers* buf, int n_worlds, int forward_only, void* stream);
/* KitchenEnv.step for N worlds: FrankaRobot.step (clip, velocity command -> position target on the previous noisy reading, bounds, do_simulation(ctrl, 40))
 * + FrankaRobot._get_obs + KitchenEnv._get_obs (59-vector with observation noise) + the per-task completion tests of compute_reward
 * (envs/franka_kitchen/franka_env.py:92-171, kitchen_env.py:340-384); forward_only != 0: the reset path (set_state -> mj_forward, _get_obs) */
int grx_kitchen_step(const grx_model* m, const grx_kitchen_task* task, const grx_kitchen_buffers* buf, int n_worlds, int forward_only, void* stream);
/* `count` consecutive np_random.uniform(-1, 1) draws per listed world, float32 rows (states / idx as in grx_fetch_sample_resets; idx NULL = worlds 0..n-1).  HOST pointers. */
int grx_sample_uniform_rows(uint64_t* states, const int64_t* idx, int n, int count, float* out);
/* The same draws ON THE DEVICE: states [N,4] uint64 and out [N,count] float32 are DEVICE pointers, mask [N] uint8 or NULL selects the worlds whose streams advance
 * (the others' rows and states are left alone).  Bit-equal to grx_sample_uniform_rows / numpy's Generator.uniform(-1, 1) (franka_env.py:118-127, kitchen_env.py:361-369):
 * the observation noise of 16 384 kitchen worlds no longer crosses PCIe (3.9 MB per step) and no host loop runs per step. */
int grx_uniform_rows_device(uint64_t* states, const unsigned char* mask, int n, int count, float* out, void* stream);
/* KitchenEnv.step's bookkeeping after the physics (kitchen_env.py:386-423) for N worlds on the device: step completions = completion bits & tasks_to_complete,
 * reward = their count, tasks_to_complete / episode_task_completions updates, terminated (every task of the episode completed), TimeLimit truncation, and the
 * autoreset decision: reset_now[w] = 1 for the worlds whose state rows were just rewound to init_qpos / zero velocity (mode 2 same_step: done in this step; mode 1
 * next_step: done in the previous one) -- the caller draws their observation noise (grx_uniform_rows_device with reset_now as mask) and runs the masked forward launch.
 * Nothing is read back: KitchenVecEnv.step(output="torch") has no host synchronisation. */
typedef struct grx_kitchen_book {
  const int* completed; const unsigned char* stepped;
  int *tasks_to_complete, *episode_completions, *elapsed, *step_completions;
  float* reward; unsigned char *terminated, *truncated, *needs_reset, *reset_now;
  float *qpos, *qvel, *qacc_ws; const float* init_qpos;
  int nq, nv, all_mask, max_steps, remove_when_completed, terminate_when_completed, mode;
  int* final_info;      /* [N,3] or NULL: (tasks_to_complete, step_task_completions, episode_task_completions) of the episode a same-step reset world has just finished */
} grx_kitchen_book;
int grx_kitchen_bookkeeping(const grx_kitchen_book* args, int n_worlds, void* stream);
int grx_goal_compute_reward(const float* achieved, const float* desired, int64_t batch, int dim, double distance_threshold, int sparse,
                            float* reward_out, void* stream);
/* batched MujocoManipulateEnv.compute_reward on 7-vector pose goals (shadow_dexterous_hand/manipulate.py:87-142) */
int grx_manip_compute_reward(const float* achieved, const float* desired, int64_t batch, int ignore_position, int ignore_rotation,
                             int ignore_z, float distance_threshold, float rotation_threshold, int sparse, float* reward_out, void* stream);

/* On-device HER relabel + replay write -- the caller of GoalEnv.compute_reward (/root/reference/README.md:72-76, gymnasium_robotics/core.py:45-67:
 * "substitute the goal, recompute the reward").  rows: [T+1, N, W] a ring of the packed output rows the step kernels write ([obs | achieved |
 * desired | reward | success]; a reset row or the row after a step), acts: [T+1, N, act_dim], acts[r] = the action that led to row r.  Row indices are
 * absolute step counts, used modulo T+1 (an episode buffer of T steps never wraps: row 0 = the reset, acts[0] unused).  For sample b the transition
 * from row t_idx[b] of world w_idx[b] is written to out[b] = [obs_t | achieved_t | goal | action_t | reward | obs_t+1 | achieved_t+1 | success] with
 * goal = the goal achieved at row t_goal[b] (t_goal[b] < 0: the episode's own goal) and reward / success recomputed from (achieved_t+1, goal) by the
 * device functions behind grx_*_compute_reward.  All pointers are device pointers. */
typedef struct grx_her_args {
  const float* rows; const float* acts;
  int T, N, W, obs_dim, goal_dim, act_dim;
  const int *t_idx, *w_idx, *t_goal;   /* [batch] */
  int kind;                            /* 0 Fetch (fetch_env.py:74-80), 1 HandReach (reach.py:92-100: -0.0 for success), 2 maze (maze_v4.py:381-388), 3 manipulate poses (manipulate.py:87-142) */
  double p0, p1;                       /* 0 / 1: distance_threshold; 2: goal radius 0.45; 3: distance_threshold, rotation_threshold */
  int sparse, ignore_pos, ignore_rot, ignore_z;
  float* out;                          /* [batch, 2 obs_dim + 3 goal_dim + act_dim + 2] */
  /* Same-step autoreset: the ring row of the step that resets world w already holds the FIRST row of w's new episode, the terminal row of the episode that just
   * ended is in term_rows[w] ([N, W], e.g. grx_fetch_reset_args.final_packed) and term_t[w] = the absolute row index it belongs to.  A sample whose next row
   * (t_idx + 1) or goal row (t_goal) equals term_t[w] reads the terminal row: the LAST transition of every episode is relabelled like any other.  Both NULL: off. */
  const float* term_rows; const int* term_t;
} grx_her_args;
int grx_her_relabel(const grx_her_args* args, int64_t batch, void* stream);
/* The index draws of HER's "future" strategy (Andrychowicz et al. 2017) for `batch` samples in one kernel: a uniform world among those whose current
 * episode has a transition in the ring (rejection sampling on episode_start), a uniform transition t of it, and with probability k / (k + 1) a uniformly
 * drawn LATER row of the same episode as the goal (t_goal), else -1.  t_now = absolute index of the newest row, T = ring length - 1.  Counter-based
 * generator: the same (seed, call) pair reproduces the same draws.  All pointers are device pointers. */
int grx_her_sample(const int* episode_start, int n_worlds, int t_now, int T, int k_future, uint64_t seed, uint64_t call, int64_t batch,
                   int* t_idx, int* w_idx, int* t_goal, void* stream);
/* The same with finished episodes: a world whose episode ended in THIS step (term_t[w] == t_now, see grx_her_args) is sampled from the episode that just ended
 * (rows prev_start[w] .. t_now, the last one being the terminal row) instead of being skipped; identical draws for every other world. */
int grx_her_sample_final(const int* episode_start, const int* prev_start, const int* term_t, int n_worlds, int t_now, int T, int k_future, uint64_t seed, uint64_t call,
                         int64_t batch, int* t_idx, int* w_idx, int* t_goal, void* stream);
/* Episode bookkeeping of the worlds reset in the step that produced row t (reset_mask[w] != 0): prev_start[w] <- episode_start[w], term_t[w] <- t,
 * episode_start[w] <- t.  One kernel.  prev_start / term_t may be NULL (then only episode_start is updated). */
int grx_her_mark_resets(const unsigned char* reset_mask, int n_worlds, int t, int* episode_start, int* prev_start, int* term_t, void* stream);

/* Episode reset of a COMPACTED list of maze worlds (maze/point_maze.py:377-390 / ant_maze_v5.py reset_model, maze_v4.py:299-358: qpos = init_qpos with
 * xy <- the drawn reset position, qvel = 0, new goal, observation of the reset state): one kernel writes state, goal, obs / achieved / success and the
 * packed row of the n_reset worlds idx[0..n_reset).  The draws (generate_reset_pos / generate_target_goal) stay on the host; stage = [n_reset, 4] rows
 * (start x, start y, goal x, goal y).  keep_outcome != 0 (same-step autoreset): the last two words of the packed row keep the finished episode's
 * reward / success; keep_outcome == 0 (reset(), next-step autoreset): reward[w] and the packed row's reward word are set to 0 (the reset step reports
 * reward 0).  All pointers are device pointers. */
typedef struct grx_maze_reset_args {
  const int* idx; const float* stage; const float* qpos0;
  int nq, nv, obs_dim, obs_skip;        /* obs_skip: leading qpos entries left out of the observation (2 for the ant, 0 for the point mass) */
  double goal_radius;
  int keep_outcome;
  float *qpos, *qvel, *qacc_ws, *goal, *obs, *achieved, *reward; unsigned char* success; float* packed;
} grx_maze_reset_args;
int grx_maze_reset_rows(const grx_maze_reset_args* args, int n_reset, void* stream);

/* Commit of an overlapped hand-manipulate reset (envs/hand.py: the settle steps of _reset_sim, manipulate.py:154-224, ran on compacted side rows while the
 * old episode was finishing): row j of the source block replaces world idx[j] of the destination buffers -- qpos, qvel, qacc_ws, obs, achieved, palm,
 * goal, the packed row except its last two words (reward / success stay those of the finished episode, same-step autoreset) with the new goal in its
 * desired-goal slot, and the sticky (upper 16) status bits are OR-ed in.  One launch instead of ~30 indexed copies.  All pointers are device pointers. */
typedef struct grx_hand_commit_args {
  const int64_t* idx; int k;
  int nq, nv, obs_dim, goal_dim;
  const float *s_qpos, *s_qvel, *s_qacc_ws, *s_obs, *s_achieved, *s_palm, *s_goal, *s_packed; const int* s_status;
  float *qpos, *qvel, *qacc_ws, *obs, *achieved, *palm, *goal, *packed; int* status;
} grx_hand_commit_args;
int grx_hand_commit_rows(const grx_hand_commit_args* args, void* stream);

/* Commit of an overlapped Fetch reset (envs/fetch.py).  Fetch episodes end by the time limit only (compute_terminated is constant False, fetch_env.py / robot_env.py:140-148),
 * so the worlds a step will reset are known BEFORE it is launched and their reset state depends on nothing the step computes: grx_fetch_reset runs for them on a side stream,
 * beside the step kernel, into a SECOND set of [N, ...] buffers (same world rows), and this call -- behind the step kernel -- parks the terminal packed row of world idx[j] in
 * final_packed, copies the staged qpos / qvel / qacc_ws / mocap / aux / goal / obs / achieved rows over the live ones, rewrites the packed row's [obs | achieved | desired] words
 * (reward / success stay those of the finished episode: same-step autoreset) and folds the reset launch's status flags into the world's status word exactly as
 * grx_fetch_reset would have.  Bit-identical to the in-line reset (tests/test_gpu_fetch.py).  All pointers are device pointers; idx as in grx_fetch_reset_args. */
typedef struct grx_fetch_commit_args {
  const int* idx; int k;
  int nq, nv, mocap_words, obs_dim;     /* mocap_words = 7 * nmocap */
  const float *s_qpos, *s_qvel, *s_qacc_ws, *s_mocap, *s_aux, *s_goal, *s_obs, *s_achieved; const int* s_status;
  float *qpos, *qvel, *qacc_ws, *mocap, *aux, *goal, *obs, *achieved, *packed, *final_packed; int* status;
} grx_fetch_commit_args;
int grx_fetch_commit_rows(const grx_fetch_commit_args* args, void* stream);

/* Commit of an overlapped Adroit reset (envs/adroit.py; hammer / door / relocate, whose reset_model draws are made on the device).  The Adroit tasks never terminate
 * (adroit_hammer.py:291-329 returns terminated = False), so the worlds a step truncates are known before it is launched: the draws (grx_adroit_sample_resets_device writing the
 * STAGED shift / target rows -- the running step still reads the finished episode's), init rows and the reset-time forward pass (grx_adroit_step, forward_only, `compact` list, a
 * second grx_adroit_buffers over staged [N, ...] arrays) run on a side stream beside the step kernel; this call, behind it, copies the staged qpos / qvel / qacc_ws / shift /
 * target / obs rows of world idx[j] over the live ones and ORs the forward pass's status flags into the sticky half of the world's status word.  reward / success / the step's
 * own status flags stay those of the finished episode (same-step autoreset).  Bit-identical -- status words included -- to the in-line reset AS AdroitVecEnv.step RUNS IT: there the
 * masked forward launch writes its own flags into the low half (grx_status_word) and the host then puts the step's low half back, keeping the accumulated sticky half
 * (envs/adroit.py: `status = (step's & 0xFFFF) | (now & ~0xFFFF)`); this kernel does both in one go.  (The Fetch commit differs on purpose: its in-line reset leaves the reset
 * launch's flags in the low half, and so does grx_fetch_commit_rows.)  tests/test_gpu_adroit.py::test_overlapped_reset_is_the_inline_reset compares `status` after every step.  Device pointers. */
typedef struct grx_adroit_commit_args {
  const int64_t* idx; int k;
  int nq, nv, obs_dim;
  const float *s_qpos, *s_qvel, *s_qacc_ws, *s_shift, *s_target, *s_obs; const int* s_status;    /* shift / target pairs may be NULL (no shift group / not relocate) */
  float *qpos, *qvel, *qacc_ws, *shift, *target, *obs; int* status;
} grx_adroit_commit_args;
int grx_adroit_commit_rows(const grx_adroit_commit_args* args, void* stream);

/* Host-side reset sampling: replaces the numpy PCG64 draws of _reset_sim / _sample_goal (fetch/fetch_env.py:153-166,388-391)
 * for the listed worlds, bit-exactly.  states: [n_total,4] uint64 = (state_hi, state_lo, inc_hi, inc_lo) of each world's
 * numpy PCG64 (created and seeded by numpy on the Python side), advanced in place.  All pointers are HOST pointers. */
int grx_fetch_sample_resets(uint64_t* states, const int64_t* idx, int n, int has_object, int target_in_the_air, double obj_range,
                            double target_range, const double* target_offset, const double* gripper_xpos, double height_offset,
                            double* out_oxy, double* out_goal);
/* The same draws ON THE DEVICE (fetch/fetch_env.py:153-166, 388-391): states [n_total,4] uint64 in HBM (advanced in place), idx [n] int32 device world indices,
 * samples [n,5] float32 out (object x, y, goal x, y, z) -- the row grx_fetch_reset reads.  Bit-equal to grx_fetch_sample_resets / numpy; the host draws nothing. */
int grx_fetch_sample_resets_device(uint64_t* states, const int* idx, int n, int has_object, int target_in_the_air, double obj_range, double target_range,
                                   const double* target_offset, const double* gripper_xpos, double height_offset, float* samples, void* stream);
/* reset_model's draws of AdroitHandHammer (kind 0: board z, adroit_hammer.py:374-376), Door (1: frame position, adroit_door.py:362-370) and Relocate (3: ball x / y + target site,
 * adroit_relocate.py:353-372) ON THE DEVICE: states [N,4] uint64 PCG64 streams in HBM (advanced in place), idx [n] int64 device world indices, shift_pos0 [3] HOST (the XML
 * body_pos), edit [N,3] float64 DEVICE in / out (model.body_pos of every world as get_env_state reports it; components the task does not redraw are kept), target64 [N,3] float64 and
 * target [N,3] float32 (relocate only, else NULL), shift [N,7] float32 (the pose the engine applies).  Bit-equal to np_random.uniform(low, high) in the reference's order. */
int grx_adroit_sample_resets_device(uint64_t* states, const int64_t* idx, int n, int kind, const double* shift_pos0, double* edit, double* target64, float* shift, float* target,
                                    void* stream);
/* MazeEnv.reset's draws ON THE DEVICE (maze/maze_v4.py:299-358: generate_target_goal, generate_reset_pos, add_xy_position_noise): states [N,5] uint64 in HBM = the world's numpy
 * PCG64 (state_hi, state_lo, inc_hi, inc_lo) and its buffered 32-bit half (has_uint32 << 32 | uinteger), advanced in place; idx [n] int32 DEVICE world indices; goal_xy [n_goal,2]
 * / reset_xy [n_reset,2] float64 DEVICE cell centres (maze.unique_goal_locations / unique_reset_locations); fixed_goal_xy / fixed_reset_xy: HOST [2] or NULL = options["goal_cell"]
 * / ["reset_cell"] of reset(); stage [n,4] float32 out (start x, y, goal x, y): the rows grx_maze_reset_rows reads.  Bit-equal to Generator.integers / uniform in the
 * reference's order. */
int grx_maze_sample_resets_device(uint64_t* states, const int* idx, int n, const double* goal_xy, int n_goal, const double* reset_xy, int n_reset, double noise_range, double scaling,
                                  const double* fixed_goal_xy, const double* fixed_reset_xy, float* stage, void* stream);
const char* grx_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
