// grx_engine.h -- wave-per-world batched rigid-body engine for gfx950 (CDNA4).
//
// One 64-lane wavefront owns ONE world.  All per-world working state lives in LDS
// for the whole fused env-step (n_substeps physics steps + task code), so HBM is
// touched once per env-step.  Lanes are mapped to bodies / dofs / constraint rows /
// matrix entries stage by stage; stages are separated by wave barriers.
//
// This is the replacement for the third-party call
//   mujoco.mj_step(model, data, nstep)   /root/reference/gymnasium_robotics/envs/robot_env.py:341
//   mujoco.mj_forward(model, data)       /root/reference/gymnasium_robotics/envs/fetch/fetch_env.py:303,401
// (stage inventory K1-K14 in SURVEY.md §8(a)); arithmetic is fp32.
//
// Programming model used below (so that tests can run the SAME source through a
// sequential lane emulator on a machine without a GPU, see tests/emu/):
//   * code outside FOR_LANES is wave-uniform (every lane computes the same thing),
//   * code inside FOR_LANES{...} is per lane; it may only read LDS data written
//     before the previous WAVE_SYNC(), and lane-private values never outlive the block,
//   * LANE0{...} marks single-writer uniform stores.
#pragma once
#include <math.h>
#include <stdint.h>

// ---- Build personality.  This header is the ONLY engine file that names GRX_EMU (the lane-emulator build of tests/emu: test infrastructure, never shipped): the stage fragments
// csrc/grx_eng_*.h ask for a FEATURE below.  GRX_ON_DEVICE: wave intrinsics exist (readlane / DPP / MFMA / ballot: the register-resident solvers, the matrix-core Hessian, the
// HBM-resident skin lists); the emulator personality runs the LDS twins of the same arithmetic.  The other switches are test / profiling instrumentation of one personality.
#if defined(GRX_EMU)
#define GRX_ON_DEVICE 0
#else
#define GRX_ON_DEVICE 1
#endif
#if GRX_ON_DEVICE && defined(GRX_PROFILE)
#define GRX_DEVICE_PROFILE 1      // per-stage cycle counters (tools/profile_stages.py)
#else
#define GRX_DEVICE_PROFILE 0
#endif
#if defined(GRX_HULL_HINTS)
#define GRX_DEVICE_HULL_HINTS GRX_ON_DEVICE        // guessed support vertices kept in HBM rows
#define GRX_TWIN_HULL_HINTS (!GRX_ON_DEVICE)       // ... and their emulator twin (a host array)
#else
#define GRX_DEVICE_HULL_HINTS 0
#define GRX_TWIN_HULL_HINTS 0
#endif
#if defined(GRX_EMU) && defined(GRX_EMU_TRACE)
#define GRX_TWIN_TRACE 1          // tools/emu_trace.py: contact lists / Newton iterations printed per pass
#else
#define GRX_TWIN_TRACE 0
#endif
#if defined(GRX_EMU) && defined(GRX_EMU_STAGEHOOK)
#define GRX_TWIN_STAGEHOOK 1      // tools/emu_mixed.py: per-stage precision experiments
#else
#define GRX_TWIN_STAGEHOOK 0
#endif
#if defined(GRX_EMU) && defined(GRX_MPR_STATS)
#define GRX_TWIN_MPR_STATS 1
#else
#define GRX_TWIN_MPR_STATS 0
#endif
#if defined(GRX_EMU) && defined(GRX_MESH_DEBUG)
#define GRX_TWIN_MESH_DEBUG 1
#else
#define GRX_TWIN_MESH_DEBUG 0
#endif
#if defined(GRX_EMU_FP64)
#define GRX_REAL_IS_DOUBLE 1      // the emulator's fp64 personality (float is #defined to double: tools/emu_fp64_check.py): the (double) overloads would be duplicates
#else
#define GRX_REAL_IS_DOUBLE 0
#endif
#if defined(GRX_EMU)
#define GRX_TWIN_STAT(k) (g_grx_newton_stats[k]++)      // solver statistics read by tools/emu_tolerances.py
#else
#define GRX_TWIN_STAT(k) ((void)0)
#endif

#if defined(GRX_EMU)
#define GRX_DEV static inline
#define GRX_MEM static inline
#define GRX_MEM_CALL static inline
#define GRX_HD static inline
#define FOR_LANES for (int lane = 0; lane < 64; ++lane)
#define LANE0 if (1)
#define WAVE_SYNC() ((void)0)
#define GRX_ATOMIC_ADD(p, v) grx_emu_fetch_add((p), (v))
static inline int grx_emu_fetch_add(int* p, int v) { int o = *p; *p = o + v; return o; }
#else
#define GRX_DEV __device__ __forceinline__
#define GRX_MEM static __device__ __forceinline__
#define GRX_MEM_CALL static __device__ __attribute__((noinline))   // a real call: the routine gets its own register budget (rare, register-hungry paths)
#define GRX_HD __host__ __device__ inline
#define FOR_LANES for (int lane = lane_, once_ = 1; once_; once_ = 0)
#define LANE0 if (lane_ == 0)
#define WAVE_SYNC() __syncthreads()
#define GRX_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#endif

// optional per-stage cycle accounting (tools/profile_stages.py builds a -DGRX_PROFILE variant; empty otherwise)
#if defined(GRX_PROFILE) && !defined(GRX_EMU)
#define GRX_NPROF 56
// sub-stage buckets 16.. (finer breakdown inside a stage; what remains goes to the stage's own bucket)
#define GRX_SUBTICK(c, k) GRX_TICK(c, 16 + (k))
#define GRX_TICK(c, id) do { if (lane_ == 0) { long long t_ = clock64(); (c)->prof[id] += t_ - (c)->prof_last[0]; (c)->prof_last[0] = t_; } } while (0)
#define GRX_COUNT(c, k, n) do { if (lane_ == 0) (c)->prof[16 + (k)] += (n); } while (0)
#define GRX_PMAX(c, k, n) do { if (lane_ == 0 && (c)->prof[16 + (k)] < (n)) (c)->prof[16 + (k)] = (n); } while (0)   // per-step maximum (table demand: tools/demand_probe.py)
#else
#define GRX_COUNT(c, k, n) ((void)0)
#define GRX_PMAX(c, k, n) ((void)0)
#define GRX_TICK(c, id) ((void)0)
#define GRX_SUBTICK(c, k) ((void)0)
#endif
enum { GRX_P_KIN = 0, GRX_P_INERTIA, GRX_P_COLLIDE, GRX_P_CONSTR, GRX_P_VEL, GRX_P_MSOLVE, GRX_P_NEVAL, GRX_P_NGRAD, GRX_P_NHESS, GRX_P_NFACTOR,
       GRX_P_NLS, GRX_P_NFINAL, GRX_P_EULER, GRX_P_OTHER };

#define GRX_MINVAL 1e-15f
#define GRX_MINIMP 0.0001f
#define GRX_MAXIMP 0.9999f
#define GRX_MAXCON 64    // largest contact-list capacity per world: one lane per contact in the per-contact passes (the large tables of the overflow lane)
#define GRX_MAXCON_DEFAULT 32   // capacity of a model that does not request one (the fast kernels: 16 - 32)
#define GRX_MAXEFC 144   // default constraint rows per world (models with wide contact rows get more: grx_pack_model)
#define GRX_JPOOL 2032    // default words of packed Jacobian storage per world (rows are stored over their dof span only); < 16384 (14-bit row offsets)
#ifndef GRX_NEWTON_RTOL
#define GRX_NEWTON_RTOL 1e-5f
#define GRX_NEWTON_ATOL 1e-5f
#endif
#define GRX_LS_MAXIT 12
#ifndef GRX_OBJ_REFINE_MINSTEP
#define GRX_OBJ_REFINE_MINSTEP 0.05f   // the object-block refinement (grx_refine_object_block) corrects ~4e-4 of the last full step: below this step size there is nothing to correct
#endif

// status bits reported per world
#define GRX_ST_BADNUM 1
#define GRX_ST_CON_OVERFLOW 2
#define GRX_ST_EFC_OVERFLOW 4
#define GRX_ST_FACTOR 8
#define GRX_ST_SOFT 16      // internal (never reported): the tables of the model's FAST kernel would have overflowed in some substep (GrxCtx::soft_*; the overflow lane, include/grx_capi.h)
#define GRX_ST_HULL 32      // internal (never reported): a hull-vs-convex pair passed the bounding-box filter in a kernel that carries no hull routine (GrxShape::kHandoff): the world is handed off mid-step

enum { GRX_ROW_EQ = 0, GRX_ROW_FRICTION = 1, GRX_ROW_LIMIT = 2, GRX_ROW_CONTACT = 3, GRX_ROW_TENDON = 4 };

// ------------------------------------------------------------------------------------------
// model: device-resident fp32 / int32 copies of the tables in include/grx_model_fields.def
// ------------------------------------------------------------------------------------------
struct GrxModel {
#define GRX_FI(name) const int* name;
#define GRX_FF(name) const float* name;
#include "../../include/grx_model_fields.def"
#undef GRX_FI
#undef GRX_FF
  int nq, nv, nu, nbody, njnt, ngeom, nsite, nmocap, neq, npair, ndevpair, nmpair, maxdepth, eulerdamp, anydamp, nfric, nweld, integrator, njump, wpool, ntendon, maxefc, jpool, ntouch, maxcon, twospan, nconvex, nfreeobj, ngridgeom, ngridwall, gridnx, gridny, handtree, nmeshpair, nshift, noslip_iterations, iterations, njeq, ngate;
  float timestep, gravity[3], meaninertia, impratio, mpr_tolerance, gridx0, gridy0, gridinv, noslip_tolerance;
  int mpr_iterations;
  // MJCF coordinates of this model's world origin (include/grx_model.h GRX_ORIGIN_*; zeros for a model compiled in the MJCF's own frame).  The engine never reads it: it works in
  // the model's frame.  The task code adds it back, in fp64, to every world position it writes out (grx_world_out below).
  double origin[3];
  // derived at model creation (grx_host_model.h), not part of the compiled blob: per hull vertex 16 records of 4 floats -- the vertex itself (x, y, z, degree; degree -1 when
  // it has more than 15 hull neighbours) and its hull neighbours (x, y, z, local id) -- so that a GUESSED support vertex is verified with ONE coalesced fetch (grx_mesh_support)
  const float* mesh_nbr;
  // derived at model creation too (grx_host_model.h, grx_build_hull_cells): the SUPPORT-CANDIDATE LISTS of every hull.  The unit sphere of directions (in the geom frame) is cut
  // into 6 x GRX_CELL_G x GRX_CELL_G cube-map cells; for every cell the list holds every hull vertex that can be the support vertex -- or tie with it inside the scan's 1e-6 m
  // band -- for ANY direction of the cell (a rigorous superset: v is listed unless (w - v) . dc > rho |w - v| + band, w = the support vertex of the cell's centre direction dc,
  // rho = the cell's chord radius).  A support evaluation whose guess (mesh_nbr) failed reads ONE cell header and ONE round of <= 64 coalesced 16-byte records (x, y, z, vertex
  // id) instead of scanning the hull (the Fetch head link: 1061 vertices, 17 rounds of loads): same winner, same tie-break, by construction and by test
  // (tests/test_cpu_hull_cells.py; the lane emulator checks every evaluation of every fixture against the list).
  const int* mesh_cellhdr;     // per (hull, cell): offset into mesh_cellrec (records), count (0: no list, scan the hull)
  const float* mesh_cellrec;
  const int* geom_cellbase;    // per geom: first header of its hull (in header pairs), -1 = none
  // derived at model creation too (grx_host_model.h, grx_build_records): per-stage RECORDS.  A stage's lane used to walk the model tables level by level -- `j = dof_jntid[d]`,
  // then `jnt_stiffness[j]`, then `qpos0[jnt_qposadr[j]]` -- one vector-L1 round trip per level and per divergent branch (profiles/cpi_r06_fetch.txt: ~60 dependent round trips per
  // substep, a third of the memory wait).  A record holds everything ONE entity's lane needs in ONE stage, gathered at model creation: one volley of loads per stage.  Integer
  // fields live in reci_* (int table), fp fields in recf_* (fp table), same index; the values are the tables' own fp32 values (sums that the stage used to form, like
  // margin - gap, are formed here in the same arithmetic), so results are bit-identical to the table walk.  Layouts: the GRX_R* enums below.
  const int *reci_body, *reci_jnt, *reci_pair, *reci_chain, *reci_weld, *reci_act, *reci_dof, *reci_mpair;
  const float *recf_body, *recf_jnt, *recf_pair, *recf_weld, *recf_eq, *recf_act, *recf_dof, *recf_ten, *recf_mpair;
};
// record strides (words) and field offsets
enum { GRX_RBI = 16, GRX_RBF = 16,    // body (kinematics): I mocapid, jntadr, jntnum, jt0 (type of the ONLY joint, else -1), qa0 (qpos address of the first joint), flags (1: shift child, 2: free or mocap body),
                                      //   type of the first joint, -, then body_jump[s] for s < 8;  F pos[3], qpos0[qa0], quat[4], first joint's pos[3], -, first joint's axis[3], -
       GRX_RJI = 8, GRX_RJF = 24,     // joint: I qposadr, type, bodyid, dofadr, parent of its body, root of its body, limited (jnt_limited && type >= 2), -;
                                      //   F pos[3], qpos0[qposadr], axis[3], -, range[2], -, -, then the row-parameter block (GRX_PRM_*) of its limit rows
       GRX_RPI = 8, GRX_RPF = 28,     // candidate pair: I condim, geom1, geom2, body1, body2, span, -, -;  F margin, gap, margin - gap, friction[5], row-parameter block [8, 18), -, -, size of geom1 [20, 23), -, size of geom2 [24, 27), -
       GRX_RCI = 4,                   // body (Jacobian columns): I dof_chainmask lo, hi, rootid, -
       GRX_RWI = 12, GRX_RWF = 28,    // active weld: I eq, body1, body2, weld_row, chainmask lo / hi / root of body1, of body2, -, -;  F eq_data[11], -, eq_relpose[14], -, -
       GRX_REF = 12,                  // equality (row parameters): F row-parameter block with invweight[0] at GRX_PRM_DA and invweight[1] at GRX_PRM_DA2, -
       GRX_RAI = 8, GRX_RAF = 12,     // actuator: I qposadr / dofadr of its joint, ctrllimited, gaintype, biastype, forcelimited, -, -;  F gear, ctrlrange[2], gainprm[3], biasprm[3], forcerange[2], -
       GRX_RDI = 8, GRX_RDF = 16,     // dof: I qposadr / type / dofadr of its joint, cvelstart, body of that dof, last dof of that body, bodyid, -;  F damping, stiffness, springref, armature, row-parameter block [4, 14) of its friction-loss row, -, -
       GRX_RMI = 4,                   // mass-matrix entry: I i, j, body of dof i, -;  F (one word) armature of dof i
       GRX_RTF = 12 };                // fixed tendon (row parameters): F row-parameter block, -, -
enum { GRX_PRM_SOLREF = 0, GRX_PRM_SOLIMP = 2, GRX_PRM_MARGIN = 7, GRX_PRM_DA = 8, GRX_PRM_AUX = 9, GRX_PRM_DA2 = 10 };   // AUX: friction[0] of a contact pair, frictionloss of a dof
#define GRX_NBR_RECS 16
#define GRX_CELL_G 16
#define GRX_CELL_MAX 64      // a cell whose list would be longer keeps none (count 0): the hull is scanned
// cube-map cell of a direction (any length > 0): the same arithmetic on the host (list construction, with the cells dilated by more than its rounding) and on the device
GRX_HD int grx_hull_cell(float x, float y, float z) {
  const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
  int face; float ma, u, v;
  if (ax >= ay && ax >= az) { face = x < 0.0f ? 1 : 0; ma = ax; u = y; v = z; }
  else if (ay >= az) { face = y < 0.0f ? 3 : 2; ma = ay; u = x; v = z; }
  else { face = z < 0.0f ? 5 : 4; ma = az; u = x; v = y; }
  const float s = (0.5f * GRX_CELL_G) / fmaxf(ma, 1e-30f);
  int iu = (int)floorf(u * s + 0.5f * GRX_CELL_G), iv = (int)floorf(v * s + 0.5f * GRX_CELL_G);
  iu = iu < 0 ? 0 : (iu > GRX_CELL_G - 1 ? GRX_CELL_G - 1 : iu); iv = iv < 0 ? 0 : (iv > GRX_CELL_G - 1 ? GRX_CELL_G - 1 : iv);
  return (face * GRX_CELL_G + iu) * GRX_CELL_G + iv;
}
#define GRX_HULLCACHE_WORDS 90   // per-world HBM row of the hull pairs: 21 words of cached separating directions (GrxCtx::meshcache) + 4 x (key + 16 guess words) + a round-robin counter

// mirrors grx_overflow_lane (include/grx_capi.h): where the worlds go that exceed a table capacity of the fast kernel
struct GrxLane { const unsigned char* skip; int* entry_count; int* entry_list; const int* list; const int* count; unsigned char* next_flags; int* next_count; int* next_list; signed char* ttl; int soft_maxefc, soft_jpool, soft_maxcon, ttl_init, grid, entry_cap, next_cap;
                 int* ready; int* progress; const int* poll_list; int ready_cap, progress_total, poll_grid, pad_; };   // the polling workgroups of the standing lane launch (include/grx_capi.h)

// per-world LDS working set; every pointer addresses LDS (or host memory in the emulator)
struct GrxCtx {
  // state
  float *qpos, *qvel, *qacc_ws, *mocap_pos, *mocap_quat, *ctrl;
  // position stage
  float *ploc, *qloc, *janchor, *jaxis;  // local poses (3,4 per body), joint anchor/axis (3,3 per joint) in parent frame -> world
  float *xpos, *xquat, *xmat, *cinert, *crb, *cvel, *cacc, *cfrc;
  float *gxpos, *gxmat, *sxpos, *sxmat;
  float *cdof, *cdof_dot;
  float *M, *A;
  float *qfrc_bias, *qfrc_passive, *qfrc_actuator, *qfrc_smooth, *qacc_smooth, *qfrc_constraint, *qacc, *Ma, *grad, *search,
      *Mv, *tmpv;
  // contacts
  float *con_dist, *con_pos, *con_frame;
  int *con_pair, *con_efc, *con_nr, *con_b1, *con_b2, *con_span, *con_ioff;  // con_span = loA | lenA << 8 | loB << 16 | lenB << 24 (pair_span); con_ioff = first item of the contact-Jacobian pass
  // constraint rows
  // Jacobian rows are stored packed: row r covers dofs [lo, lo+len) at Jp[off .. off+len); efc_row[r] = off | lo << 14 | len << 21 (GRX_ROW_PACK)
  float *Jp, *efc_pos, *efc_D, *efc_aref, *efc_jar, *efc_jv, *efc_force, *efc_floss;
  int *efc_kind, *efc_id, *efc_quad, *efc_row;  // efc_id packs (id << 4) | sub
  // scratch
  float *rk_q0, *rk_v0, *rk_Fv, *rk_Fa;  // RK4: state at the start of the step, per-stage velocities / accelerations
  float* red;  // 128 floats
  int* ired;   // 64 ints
  int* cnt;    // [0]=ncon [1]=nefc [2]=status [3]=ne [4]=nlimit ...
  float* shift;      // models with a shift group: the world's offset t[3] and rotation q[4] (state, loaded with qpos); flag 1 = x + t, flag 2 = R(q) x + t
  float* minv;       // models with the noslip post-solver: M^-1 (nv x nv), formed once per substep
  float* hullhint;   // HBM, or null: 4 x (pair + 1, 16 words of support-vertex guesses) + a round-robin counter, kept across substeps and steps (grx_mesh_pairs)
  float* meshcache;  // models with hull-vs-convex pairs: 4 x (pair + 1, separating direction, the two support vertices) + the slot to evict next, kept across the substeps of a step (grx_mesh_pairs)
  int mslot;   // slot of this world's model in g_grx_models (GPU build)
  int bail;    // != 0: a step kernel that hands capacity overflows to a re-run at a larger capacity stops simulating at the first overflowing substep (nothing of this run is kept)
  int soft_maxefc, soft_jpool, soft_maxcon;   // > 0 (the large-table kernel of the overflow lane): the capacities of the FAST kernel; exceeding one of them raises GRX_ST_SOFT
  int *lane_entry_count, *lane_entry_list; int lane_entry_cap, lane_world; int* lane_ready; int lane_ready_cap;   // fast kernel with an overflow lane: where a world that overflows claims its re-run (grx_lane_claim)
  // MID-STEP HAND-OFF (include/grx_capi.h, grx_fetch_buffers.handoff): a kernel that cannot go on with this world -- a table capacity is exceeded, or it carries no hull routine and a
  // hull pair came near -- stops at the START of the substep in question (nothing of it has touched the state yet), writes the world's row [substep + 1 | status | ctrl | mocap |
  // qpos | qvel | qacc_ws] and claims an entry; the kernel that takes the entry resumes AT that substep instead of re-running the step from its first one.
  float* handoff; int handoff_stride, handoff_large;   // handoff_large: claims from this launch need the LARGE tables (it already runs the middle ones): polling workgroups leave them to the entry launch
  int resume_first;   // 1 during the first substep of a resumed world: the free-joint quaternions in qpos were normalised by the kernel that handed it off (grx_kinematics normalises in place)
  int* skin;   // large scenes: this world's skin list in HBM (grx_collision), or null: [0] entries, [1] valid, [4, 4 + 3 ngeom) reference geom positions, then the list
  float skin_r;
  int maxefc, jpool, maxcon;  // capacities of the row tables / the packed Jacobian pool / the contact list of this model
#if defined(GRX_PROFILE) && !defined(GRX_EMU)
  long long* prof; long long* prof_last;
#endif
};

// The GPU build keeps the model descriptors (table pointers + scalars) in constant memory.  Every stage re-derives its
// model pointer from the slot index through an opaque scalar: the table pointers a stage needs are then fetched with
// s_load at the top of that stage and die at its end, instead of ~100 pointers staying live (and spilled) across the
// whole substep loop.
// GRX_OPAQUE_LANE(l): the lane index is made opaque at the top of a stage: its lane-derived values (record addresses lane * stride, `lane < n` masks) are recomputed there --
// two or three VALU instructions -- instead of being hoisted out of the 20-substep loop, kept live across it and spilled to scratch (a dword per lane each, written once by
// every wave and written back to HBM when the line leaves the L2: profiles/ab_r06_opaque_lane.txt).
// Which stages do it is a property of the kernel shape (GrxShape::kOpaque), measured per family (profiles/ab_r06_opaque_lane.txt): every stage for Fetch, the mazes, Adroit and the
// kitchen (kitchen scratch 496 -> 256 B per lane, +4 %; ant +4 %; Fetch HBM traffic 17.6 x -> 6.2 x algorithmic, +2 %), only the constraint stage for the hand models (-2 % otherwise).
#if defined(GRX_EMU)
#define GRX_OPAQUE_LANE(l) ((void)0)
#define GRX_OPAQUE_STAGE(l) ((void)0)
#define GRX_FRESH_MODEL(m, c) ((void)0)
#else
#define GRX_OPAQUE_LANE(l) asm volatile("" : "+v"(l))
#define GRX_OPAQUE_STAGE(l) do { if constexpr (S::kOpaque) GRX_OPAQUE_LANE(l); } while (0)
#define GRX_MAX_MODELS 32
static __constant__ GrxModel g_grx_models[GRX_MAX_MODELS];   // one copy per translation unit (see csrc/grx_kernels.hip on the build)
#define GRX_FRESH_MODEL(m, c) do { int s_ = __builtin_amdgcn_readfirstlane((c)->mslot); asm volatile("" : "+s"(s_)); (m) = g_grx_models + s_; } while (0)
#endif

// LDS layout.  Arrays that only live in the position/velocity stages of a substep (P1: local poses, spatial inertias,
// body velocities/forces, geom frames, contacts) and arrays that only live in the solve/integrate stage (P2: Hessian,
// Newton vectors, per-row solver scratch) share one overlay region; everything that must survive a whole substep (state,
// body frames, motion axes, M, J, row parameters) is persistent.
struct GrxDims { int nq, nv, nu, nbody, njnt, ngeom, nsite, nmocap, nfric, integrator, maxefc, jpool, ntouch, maxcon, nmesh, nshift, noslip; };
GRX_HD int grx_ctx_words(int nq, int nv, int nu, int nbody, int njnt, int ngeom, int nsite, int nmocap, int nfric, int integrator, int maxefc = GRX_MAXEFC,
                          int jpool = GRX_JPOOL, int ntouch = 0, int maxcon = GRX_MAXCON_DEFAULT, int nmesh = 0, int nshift = 0, int noslip = 0) {
  int pers = nq + nv + nv + 3 * nmocap + 4 * nmocap + nu;           // state
  pers += (3 + 4 + 9) * nbody + 12 * nsite + 6 * nv;                // xpos xquat xmat, sites, cdof
  pers += nv * nv + 4 * nv;                                          // M, qfrc_smooth qacc_smooth qfrc_constraint qacc
  pers += jpool + maxefc * (5 + (nfric ? 1 : 0));            // packed J, efc D aref kind id|sub row (+ floss)
  pers += (njnt > 32 ? 64 : 32) + 8 + (nmesh ? 21 : 0) + (nshift ? 8 : 0);   // noslip: M^-1 lives in the Hessian buffer (dead once Newton has finished)
  if (integrator == 1) pers += nq + nv + 8 * nv;                    // RK4 stage storage                                                   // ired, cnt
  int u1a = 7 * nbody + 6 * njnt, u1b = 18 * nbody;                  // {ploc qloc janchor jaxis} | {cvel cacc cfrc}
  int u2a = 10 * nbody, u2b = 12 * ngeom;                            // {crb} | {gxpos gxmat}
  const int ckeep = (ntouch || noslip) ? maxcon * (3 + 3 + 3) : 0;   // touch sensors and the noslip pass read the contacts after the solve: keep pos / normal / pair / rows out of the overlay
  pers += ckeep;
  int p1 = (u1a > u1b ? u1a : u1b) + (u2a > u2b ? u2a : u2b) + 10 * nbody + 6 * nv + 3 * nv + maxcon * (1 + 3 + 3 + 7) - ckeep;
  int p2 = nv * nv + 5 * nv + 4 * maxefc;
  return pers + (p1 > p2 ? p1 : p2) + 8;
}

#if defined(GRX_EMU) && defined(GRX_EMU_STAGEHOOK)
struct GrxEmuField { const char* name; void* ptr; int n, isint; };
static GrxEmuField g_grx_emu_fields[128]; static int g_grx_emu_nfields = 0;
static void grx_emu_carve_rec(const char* name, void* ptr, int n, int isint) { if (g_grx_emu_nfields < 128) { GrxEmuField f = {name, ptr, n, isint}; g_grx_emu_fields[g_grx_emu_nfields++] = f; } }
static void (*g_grx_stage_hook)(int stage) = nullptr;
static int g_grx_solve_mode = 0;   // 0 normal, 1 Newton only, 2 Euler stage only (tools/emu_mixed.py splits the solve stage)   // called by grx_forward_euler after each stage (-1: before the first)
#define GRX_STAGE_HOOK(k) do { if (g_grx_stage_hook) g_grx_stage_hook(k); } while (0)
#else
#define GRX_STAGE_HOOK(k) ((void)0)
#endif
// dims by value: when they are compile-time constants (a specialised kernel) every LDS address below folds to an
// immediate offset of the ds_read/ds_write instructions
GRX_DEV void grx_ctx_carve(GrxCtx* c, float* base, const GrxDims d) {
  const GrxDims* m = &d;
  float* p = base;
  c->hullhint = nullptr; c->skin = nullptr; c->skin_r = 0.0f; c->bail = 0; c->handoff = nullptr; c->handoff_stride = 0; c->handoff_large = 0; c->resume_first = 0; c->soft_maxefc = c->soft_jpool = c->soft_maxcon = 0; c->lane_entry_count = nullptr; c->lane_entry_list = nullptr; c->lane_entry_cap = 0; c->lane_world = 0; c->lane_ready = nullptr; c->lane_ready_cap = 0;
#if defined(GRX_EMU) && defined(GRX_EMU_STAGEHOOK)   // test infrastructure (tools/emu_mixed.py): the field map of the working set
#define GRX_CARVE_REC(field, n, isint) grx_emu_carve_rec(#field, (void*)p, (n), (isint));
#else
#define GRX_CARVE_REC(field, n, isint)
#endif
#define CARVE(field, n) c->field = p; GRX_CARVE_REC(field, n, 0) p += (n);
#define CARVEI(field, n) c->field = (int*)p; GRX_CARVE_REC(field, n, 1) p += (n);
  // ---- persistent
  CARVE(qpos, m->nq) CARVE(qvel, m->nv) CARVE(qacc_ws, m->nv) CARVE(mocap_pos, 3 * m->nmocap) CARVE(mocap_quat, 4 * m->nmocap)
  CARVE(ctrl, m->nu)
  CARVE(xpos, 3 * m->nbody) CARVE(xquat, 4 * m->nbody) CARVE(xmat, 9 * m->nbody) CARVE(sxpos, 3 * m->nsite) CARVE(sxmat, 9 * m->nsite)
  CARVE(cdof, 6 * m->nv) CARVE(M, m->nv * m->nv)
  CARVE(qfrc_smooth, m->nv) CARVE(qacc_smooth, m->nv) CARVE(qfrc_constraint, m->nv) CARVE(qacc, m->nv)
  c->red = p;  // 128-float scratch of the big-mesh collision path; the Jacobian pool is not written before the constraint stage
  c->maxefc = m->maxefc; c->jpool = m->jpool; c->maxcon = m->maxcon;
  CARVE(Jp, m->jpool) CARVE(efc_D, m->maxefc) CARVE(efc_aref, m->maxefc)
  c->efc_pos = c->efc_aref;  // residuals live in the aref slot until the per-row pass turns them into aref
  c->efc_floss = p; if (m->nfric) { GRX_CARVE_REC(efc_floss, m->maxefc, 0) p += m->maxefc; }
  CARVEI(efc_kind, m->maxefc) CARVEI(efc_id, m->maxefc) CARVEI(efc_row, m->maxefc)
  CARVEI(ired, m->njnt > 32 ? 64 : 32) CARVEI(cnt, 8)
  CARVE(shift, m->nshift ? 8 : 0)
  CARVE(meshcache, m->nmesh ? 21 : 0)
  if (m->integrator == 1) { CARVE(rk_q0, m->nq) CARVE(rk_v0, m->nv) CARVE(rk_Fv, 4 * m->nv) CARVE(rk_Fa, 4 * m->nv) }
  if (m->ntouch || m->noslip) {
    CARVE(con_pos, 3 * m->maxcon) CARVE(con_frame, 3 * m->maxcon)
    CARVEI(con_pair, m->maxcon) CARVEI(con_efc, m->maxcon) CARVEI(con_nr, m->maxcon)
  }
  float* overlay = p;
  // ---- P1 (kinematics .. velocity stage)
  {
    float* u = p;  // union 1: local poses + joint frames (kinematics, inertia stage) | body velocity/force vectors (velocity stage)
    CARVE(ploc, 3 * m->nbody) CARVE(qloc, 4 * m->nbody) CARVE(janchor, 3 * m->njnt) CARVE(jaxis, 3 * m->njnt)
    float* e1 = p; p = u;
    CARVE(cvel, 6 * m->nbody) CARVE(cacc, 6 * m->nbody) CARVE(cfrc, 6 * m->nbody)
    if (e1 > p) p = e1;
    u = p;         // union 2: composite inertias (inertia stage) | geom frames (collision stage)
    CARVE(crb, 10 * m->nbody)
    e1 = p; p = u;
    CARVE(gxpos, 3 * m->ngeom) CARVE(gxmat, 9 * m->ngeom)
    if (e1 > p) p = e1;
  }
  CARVE(cinert, 10 * m->nbody) CARVE(cdof_dot, 6 * m->nv)
  CARVE(qfrc_bias, m->nv) CARVE(qfrc_passive, m->nv) CARVE(qfrc_actuator, m->nv)
  CARVE(con_dist, m->maxcon) CARVEI(con_span, m->maxcon) CARVEI(con_ioff, m->maxcon) CARVEI(con_b1, m->maxcon) CARVEI(con_b2, m->maxcon)
  if (!(m->ntouch || m->noslip)) {
    CARVE(con_pos, 3 * m->maxcon) CARVE(con_frame, 3 * m->maxcon)  // con_frame: contact normal only
    CARVEI(con_pair, m->maxcon) CARVEI(con_efc, m->maxcon) CARVEI(con_nr, m->maxcon)
  }
  // ---- P2 (solve / integrate) on top of P1
  p = overlay;
  c->minv = p;   // noslip's M^-1 takes the Hessian's place: the pass runs after the last Newton iteration, the Euler stage rebuilds A afterwards
  CARVE(A, m->nv * m->nv) CARVE(Ma, m->nv) CARVE(grad, m->nv) CARVE(search, m->nv) CARVE(Mv, m->nv) CARVE(tmpv, m->nv)
  CARVE(efc_jar, m->maxefc) CARVE(efc_jv, m->maxefc) CARVE(efc_force, m->maxefc) CARVEI(efc_quad, m->maxefc)
#undef CARVE
#undef CARVEI
}

// Fast kernel with an overflow lane (include/grx_capi.h, grx_overflow_lane), called after every substep: once a table capacity has overflowed the world tries to CLAIM
// a slot of the step's entry list (the worlds that are re-run on the large tables after this launch).  1 = claimed: stop simulating, nothing of this run is kept
// (c->bail = 2).  0 = no overflow, or the list is full (more than entry_cap worlds overflowed in this very step): the world goes on with the excess contacts dropped,
// as without a lane, and the sticky status flag says so.
GRX_MEM int grx_lane_claim(GrxCtx* c, int lane_) {
#if defined(GRX_EMU)
  (void)c; (void)lane_;
  return 0;
#else
  // a contact-list overflow is only worth the re-run when the large tables hold more contacts than this kernel's: at the engine's limit (GRX_MAXCON = one lane per
  // contact) the re-run would drop the same contacts again -- after a serialised 5 - 9 ms for a hand jammed into the door (profiles/lane_probe_r03_door.txt)
  const int worth = (c->cnt[2] & GRX_ST_EFC_OVERFLOW) || ((c->cnt[2] & GRX_ST_CON_OVERFLOW) && c->maxcon < GRX_MAXCON);
  if (c->bail != 1 || !worth) return 0;
  int idx = 0;
  if (lane_ == 0) idx = atomicAdd(c->lane_entry_count, 1);
  idx = __builtin_amdgcn_readfirstlane(idx);
  if (idx >= c->lane_entry_cap) { c->bail = 0; return 0; }
  if (lane_ == 0) {
    c->lane_entry_list[idx] = c->lane_world;
    if (c->lane_ready && idx < c->lane_ready_cap) { __threadfence(); atomicExch(c->lane_ready + idx, 1); }   // published: a polling workgroup of the standing lane launch may take it now
  }
  c->bail = 2;
  return 1;
#endif
}

// words of a hand-off row (GrxCtx::handoff): [0] substep + 1 (0 = none) as int, [1] status flags as int, ctrl[nu], mocap pos / quat [7 nmocap], qpos, qvel, qacc_ws
GRX_HD int grx_handoff_words(int nq, int nv, int nu, int nmocap) { return 2 + nu + 7 * nmocap + nq + 2 * nv; }
// is a hand-off due?  (wave-uniform: the flags live in LDS, read after a WAVE_SYNC)
GRX_DEV int grx_handoff_due(const GrxCtx* c) {
  return c->handoff != nullptr && c->bail == 1 &&
         ((c->cnt[2] & (GRX_ST_HULL | GRX_ST_EFC_OVERFLOW)) || ((c->cnt[2] & GRX_ST_CON_OVERFLOW) && c->maxcon < GRX_MAXCON));
}
// The hand-off itself, called at the substep boundary: 1 = the world's row is written and its entry published: stop, nothing of this run is kept (c->bail = 2).
// 0 = the step's entry list is full: the world goes on in THIS kernel without a lane (c->bail = 0; the caller repeats the substep to its end -- the contacts that do not
// fit, or the hull pairs this kernel cannot collide, are dropped and the sticky status flag says so).
GRX_MEM int grx_lane_handoff(GrxCtx* c, int substep, int nq, int nv, int nu, int nmocap, int lane_) {
#if defined(GRX_EMU)
  (void)c; (void)substep; (void)nq; (void)nv; (void)nu; (void)nmocap; (void)lane_;
  return 0;
#else
  int idx = 0;
  if (lane_ == 0) idx = atomicAdd(c->lane_entry_count, 1);
  idx = __builtin_amdgcn_readfirstlane(idx);
  if (idx >= c->lane_entry_cap) { c->bail = 0; return 0; }
  float* row = c->handoff + (size_t)c->lane_world * c->handoff_stride;
  int o = 2;
  for (int i = lane_; i < nu; i += 64) row[o + i] = c->ctrl[i];
  o += nu;
  for (int i = lane_; i < 7 * nmocap; i += 64) { const int k = i / 7, e = i - 7 * k; row[o + i] = (e < 3) ? c->mocap_pos[3 * k + e] : c->mocap_quat[4 * k + e - 3]; }
  o += 7 * nmocap;
  for (int i = lane_; i < nq; i += 64) row[o + i] = c->qpos[i];
  o += nq;
  for (int i = lane_; i < nv; i += 64) { row[o + i] = c->qvel[i]; row[o + nv + i] = c->qacc_ws[i]; }
  if (lane_ == 0) { ((int*)row)[1] = c->cnt[2] & ~(GRX_ST_HULL | GRX_ST_EFC_OVERFLOW | GRX_ST_CON_OVERFLOW); ((int*)row)[0] = substep + 1; }
  __threadfence();      // every lane: its part of the row is written back before the entry can be seen (the taker may run on another XCD, behind another L2)
  __syncthreads();
  if (lane_ == 0) {
    c->lane_entry_list[idx] = c->lane_world | (c->handoff_large ? (1 << 30) : 0);
    __threadfence();
    if (c->lane_ready && idx < c->lane_ready_cap) atomicExch(c->lane_ready + idx, 1);   // published: a polling workgroup of the standing lane launch may take it now
  }
  c->bail = 2;
  return 1;
#endif
}

GRX_HD GrxDims grx_dims_of(const GrxModel* m) {
  GrxDims d = {m->nq, m->nv, m->nu, m->nbody, m->njnt, m->ngeom, m->nsite, m->nmocap, m->nfric, m->integrator, m->maxefc, m->jpool, m->ntouch, m->maxcon, m->nmeshpair > 0, m->nshift > 0, m->noslip_iterations > 0};
  return d;
}
GRX_HD int grx_ctx_words(const GrxDims d) { return grx_ctx_words(d.nq, d.nv, d.nu, d.nbody, d.njnt, d.ngeom, d.nsite, d.nmocap, d.nfric, d.integrator, d.maxefc, d.jpool, d.ntouch, d.maxcon, d.nmesh, d.nshift, d.noslip); }

// ------------------------------------------------------------------------------------------
// small math (all per-lane, registers)
// ------------------------------------------------------------------------------------------
// a world position of the model's (workspace-centred) frame -> the MJCF's world frame, rounded once to the output type
GRX_DEV float grx_world_out(float x, double origin) { return (float)((double)x + origin); }
GRX_DEV float dot3f(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
GRX_DEV void cross3f(float* r, const float* a, const float* b) {
  float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
GRX_DEV void mulQuatf(float* r, const float* a, const float* b) {
  float w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  float x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  float y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  float z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
GRX_DEV void normalize4f(float* q) {
  float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < 1e-12f) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  float r = 1.0f / n;
  q[0] *= r; q[1] *= r; q[2] *= r; q[3] *= r;
}
GRX_DEV void quat2matf(float* m, const float* q) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
GRX_DEV void rotVecQuatf(float* r, const float* v, const float* q) {
  float m[9]; quat2matf(m, q);
  float x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2], z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
GRX_DEV void mulMatVec3f(float* r, const float* m, const float* v) {
  float x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2], z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
GRX_DEV void mulMatTVec3f(float* r, const float* m, const float* v) {
  float x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2], y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2], z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
#ifndef GRX_EMU_FP64   // fp64 twins for the stages that run in double precision on purpose (GRX_MPR_REAL)
GRX_DEV double dot3f(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
GRX_DEV void cross3f(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
GRX_DEV void mulMatVec3f(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2], z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
GRX_DEV void mulMatTVec3f(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2], y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2], z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
// fp32 matrix (a frame as the kinematics stage left it in LDS), fp64 vectors: the entries are widened at use -- the same products as with a widened copy of the matrix,
// which would cost two registers per entry for the whole portal search (GrxMprPairT<float>)
GRX_DEV void mulMatVec3f(double* r, const float* m, const double* v) {
  double x = (double)m[0] * v[0] + (double)m[1] * v[1] + (double)m[2] * v[2], y = (double)m[3] * v[0] + (double)m[4] * v[1] + (double)m[5] * v[2], z = (double)m[6] * v[0] + (double)m[7] * v[1] + (double)m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
GRX_DEV void mulMatTVec3f(double* r, const float* m, const double* v) {
  double x = (double)m[0] * v[0] + (double)m[3] * v[1] + (double)m[6] * v[2], y = (double)m[1] * v[0] + (double)m[4] * v[1] + (double)m[7] * v[2], z = (double)m[2] * v[0] + (double)m[5] * v[1] + (double)m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
#endif
GRX_DEV void mulMat3f(float* r, const float* a, const float* b) {
  float t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  for (int i = 0; i < 9; i++) r[i] = t[i];
}
// pose (flag 2) members of the per-world shift group: x <- R(q) x, frame <- R(q) frame (the offset t is added by the caller)
GRX_DEV void grx_apply_group_rotation(const float* q, float* x, float* frame) {
  float Rg[9], t[9];
  quat2matf(Rg, q);
  mulMatVec3f(x, Rg, x);
  mulMat3f(t, Rg, frame);
  for (int e = 0; e < 9; e++) frame[e] = t[e];
}
// spatial inertia (10: Ixx Iyy Izz Ixy Ixz Iyz hx hy hz m, about the tree reference point) times motion vector [w; v]
GRX_DEV void inertMulf(float* f, const float* I, const float* v) {
  f[0] = I[0] * v[0] + I[3] * v[1] + I[4] * v[2] + (I[7] * v[5] - I[8] * v[4]);
  f[1] = I[3] * v[0] + I[1] * v[1] + I[5] * v[2] + (I[8] * v[3] - I[6] * v[5]);
  f[2] = I[4] * v[0] + I[5] * v[1] + I[2] * v[2] + (I[6] * v[4] - I[7] * v[3]);
  f[3] = I[9] * v[3] + (v[1] * I[8] - v[2] * I[7]);
  f[4] = I[9] * v[4] + (v[2] * I[6] - v[0] * I[8]);
  f[5] = I[9] * v[5] + (v[0] * I[7] - v[1] * I[6]);
}
GRX_DEV void crossMotionf(float* r, const float* v, const float* m) {
  float a[3], b[3], c[3];
  cross3f(a, v, m); cross3f(b, v, m + 3); cross3f(c, v + 3, m);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
GRX_DEV void crossForcef(float* r, const float* v, const float* f) {
  float a[3], b[3], c[3];
  cross3f(a, v, f); cross3f(b, v + 3, f + 3); cross3f(c, v, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}

// lane-private values that must survive until a wave-wide reduction: a register on the GPU, a 64-array in the emulator
#if defined(GRX_EMU)
#define GRX_LANEVAR(name) float name[64]
#define LV(name) name[lane]
static inline float grx_reduce_sum(const float* v) { float s = 0; for (int i = 0; i < 64; i++) s += v[i]; return s; }
static inline float grx_reduce_max(const float* v) { float s = v[0]; for (int i = 1; i < 64; i++) s = fmaxf(s, v[i]); return s; }
#else
#define GRX_LANEVAR(name) float name
#define LV(name) name
// cross-lane butterflies on the DPP path (no LDS): xor 1, xor 2 (quad_perm), row_half_mirror, row_mirror, then the four
// row totals are combined through v_readlane.  Must be called with all 64 lanes active.  Result is wave-uniform.
__device__ __forceinline__ float grx_dpp_f(float v, const int ctrl_unused) { return v; }
#define GRX_DPP_MOV(v, CTRL) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true))
__device__ __forceinline__ float grx_reduce_sum(float v) {
  v += GRX_DPP_MOV(v, 0xB1); v += GRX_DPP_MOV(v, 0x4E); v += GRX_DPP_MOV(v, 0x141); v += GRX_DPP_MOV(v, 0x140);
  return (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48)));
}
// the same butterflies on an unsigned 64-bit key (two 32-bit DPP moves per step; lanes outside a row read 0 = the smallest key): wave-uniform maximum
__device__ __forceinline__ unsigned long long grx_reduce_max_u64(unsigned long long k) {
#define GRX_DPP_U64_STEP(CTRL) { const unsigned lo2_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)k, CTRL, 0xF, 0xF, true), \
                                 hi2_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(k >> 32), CTRL, 0xF, 0xF, true); \
                                 const unsigned long long k2_ = ((unsigned long long)hi2_ << 32) | lo2_; k = k2_ > k ? k2_ : k; }
  GRX_DPP_U64_STEP(0xB1) GRX_DPP_U64_STEP(0x4E) GRX_DPP_U64_STEP(0x141) GRX_DPP_U64_STEP(0x140)
#undef GRX_DPP_U64_STEP
  unsigned long long m = 0ull;
#pragma unroll
  for (int l = 0; l < 64; l += 16) {
    const unsigned long long v = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k, l);
    m = v > m ? v : m;
  }
  return m;
}
__device__ __forceinline__ float grx_reduce_max(float v) {
  v = fmaxf(v, GRX_DPP_MOV(v, 0xB1)); v = fmaxf(v, GRX_DPP_MOV(v, 0x4E)); v = fmaxf(v, GRX_DPP_MOV(v, 0x141)); v = fmaxf(v, GRX_DPP_MOV(v, 0x140));
  return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16))),
               fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48))));
}
#endif

// integer lane variables, exclusive prefix sums over the 64 lanes and reads of one lane's value (uniform index)
#if defined(GRX_EMU)
#define GRX_LANEVAR_I(name) int name[64]
#define GRX_SCAN_EXCL(in, out, total) do { int s_ = 0; for (int i_ = 0; i_ < 64; i_++) { int t_ = (in)[i_]; (out)[i_] = s_; s_ += t_; } (total) = s_; } while (0)
#define GRX_LANE_READ_I(var, idx) ((var)[idx])
#else
#define GRX_LANEVAR_I(name) int name
// Hillis-Steele inside each row of 16 lanes (row_shr 1,2,4,8, zero fill), then row_bcast15 into rows 1,3 and row_bcast31 into rows 2,3
__device__ __forceinline__ int grx_scan_incl_i(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
  return v;
}
#define GRX_SCAN_EXCL(in, out, total) do { const int inc_ = grx_scan_incl_i(in); (out) = inc_ - (in); (total) = __builtin_amdgcn_readlane(inc_, 63); } while (0)
#define GRX_LANE_READ_I(var, idx) __builtin_amdgcn_readlane(var, idx)
#endif

// reductions inside groups of 8 consecutive lanes (all 8 lanes receive the result) and wave ballots
#if defined(GRX_EMU)
#define GRX_OCT_MAX(in, out) do { for (int g_ = 0; g_ < 8; g_++) { float m_ = (in)[8 * g_]; for (int i_ = 1; i_ < 8; i_++) m_ = fmaxf(m_, (in)[8 * g_ + i_]); \
    for (int i_ = 0; i_ < 8; i_++) (out)[8 * g_ + i_] = m_; } } while (0)
#define GRX_OCT_MIN_I(in, out) do { for (int g_ = 0; g_ < 8; g_++) { int m_ = (in)[8 * g_]; for (int i_ = 1; i_ < 8; i_++) m_ = (in)[8 * g_ + i_] < m_ ? (in)[8 * g_ + i_] : m_; \
    for (int i_ = 0; i_ < 8; i_++) (out)[8 * g_ + i_] = m_; } } while (0)
static inline unsigned long long grx_emu_ballot(const int* v) { unsigned long long b = 0; for (int i = 0; i < 64; i++) if (v[i]) b |= 1ull << i; return b; }
#define GRX_BALLOT(var) grx_emu_ballot(var)
#else
__device__ __forceinline__ float grx_oct_max_f(float v) {  // xor 1, xor 2 (quad_perm), row_half_mirror
  v = fmaxf(v, GRX_DPP_MOV(v, 0xB1)); v = fmaxf(v, GRX_DPP_MOV(v, 0x4E)); v = fmaxf(v, GRX_DPP_MOV(v, 0x141));
  return v;
}
__device__ __forceinline__ int grx_oct_min_i(int v) {
  int t;
  t = __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); v = t < v ? t : v;
  t = __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true); v = t < v ? t : v;
  t = __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true); v = t < v ? t : v;
  return v;
}
#define GRX_OCT_MAX(in, out) do { (out) = grx_oct_max_f(in); } while (0)
#define GRX_OCT_MIN_I(in, out) do { (out) = grx_oct_min_i(in); } while (0)
#define GRX_BALLOT(var) ((unsigned long long)__ballot((var) != 0))
#endif

// wave-wide sums of per-lane partials staged in red[0..63] (and red[64..127] for the second value).
// Uniform context.
#if defined(GRX_EMU)
GRX_DEV float grx_wave_sum(const float* red, int lane_) { (void)lane_; float s = 0; for (int i = 0; i < 64; i++) s += red[i]; return s; }
#else
GRX_DEV float grx_wave_sum(const float* red, int lane_) {
  float v = red[lane_];
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
#endif

#if defined(GRX_EMU)
GRX_DEV float grx_wave_max(const float* red, int lane_) { (void)lane_; float s = red[0]; for (int i = 1; i < 64; i++) s = fmaxf(s, red[i]); return s; }
#else
GRX_DEV float grx_wave_max(const float* red, int lane_) {
  float v = red[lane_];
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
#endif

// All stages live in a class template so that the dof count can be a compile-time constant (NV > 0: inner loops over
// dofs unroll and their LDS loads batch) or a runtime value (NV == 0: generic fallback, also used by the emulator).
// Model shape: the ten layout dims as compile-time constants (0 = read from the model at run time).
template <int NQ_, int NV_, int NU_, int NBODY_, int NJNT_, int NGEOM_, int NSITE_, int NMOCAP_, int NFRIC_ = 0, int INTEG_ = 0, int MAXEFC_ = GRX_MAXEFC, int JPOOL_ = GRX_JPOOL,
          int NTOUCH_ = 0, int MAXCON_ = GRX_MAXCON_DEFAULT, int TWOSPAN_ = 0, int CONVEX_ = 0>
struct GrxShape {
  static constexpr int NQ = NQ_, NV = NV_, NU = NU_, NB = NBODY_, NJ = NJNT_, NG = NGEOM_, NS = NSITE_, NM = NMOCAP_, NF = NFRIC_, INTEG = INTEG_, ME = MAXEFC_, JP = JPOOL_, NT = NTOUCH_, MC = MAXCON_;
  // rows may carry a second dof span (models compiled with split pair spans); fixed shapes without it skip that bookkeeping
  static constexpr bool kTwoSpan = (NV_ == 0) || (TWOSPAN_ != 0);
  static constexpr bool kFixed = NV_ > 0;   // nu / nmocap may legitimately be 0 in a fixed shape
  // incremental Hessian corrections between Newton iterations (grx_hessian_update): compiled into the kernels of models with a free
  // object in contact (several iterations per substep are common there); articulated-only models and the RK4 ant converge in one
  static constexpr bool kIncrHess = (NV_ == 0) || (NQ_ != NV_ && INTEG_ == 0) || NV_ > 33;   // ... and AdroitHandRelocate (nv 36: a full assembly = the 32 x 32 tile + four more rows / columns; A/B 17.4 -> 15.9 ms per step; door / pen / hammer measured slower or equal with it)
  static constexpr bool kConvex = (NV_ == 0) || ((CONVEX_ & 1) != 0);   // carries the general convex (MPR) narrow phase for primitive pairs: the generic kernels and the shapes of models that need it
  static constexpr bool kMesh = (NV_ == 0) || ((CONVEX_ & 2) != 0);     // carries the wave-cooperative hull-vs-convex routine (models with mesh-mesh / mesh-primitive pairs)
  // bit 5: the model HAS hull pairs but this kernel carries no routine for them (it fits 168 VGPRs = a third wave per SIMD): gates and the bounding-box filter run as usual,
  // a pair that passes them hands the world off mid-step to a kernel that has the routine (GRX_ST_HULL, grx_lane_handoff)
  static constexpr bool kHandoff = (NV_ != 0) && ((CONVEX_ & 32) != 0);
  static constexpr bool kHullFilter = kMesh || kHandoff;
  static constexpr int NMESH = (CONVEX_ & 2) ? 1 : 0;
  static constexpr int NSHIFT = (CONVEX_ & 4) ? 1 : 0;   // the model has a per-world shift group (Adroit's nail board)
  static constexpr int NOSLIP = (CONVEX_ & 8) ? 1 : 0;   // the model runs the noslip post-solver
  static constexpr bool kShift = (NV_ == 0) || NSHIFT, kNoslip = (NV_ == 0) || NOSLIP;
  static constexpr bool kShiftRot = (NV_ == 0) || ((CONVEX_ & 16) != 0);   // the shift group also rotates (flag 2: Adroit pen's target body, model.body_quat edits)
  static constexpr bool kOpaque = !(NV_ > 0 && NFRIC_ == 24);   // GRX_OPAQUE_STAGE: every family but the Shadow hand (24 friction-loss dofs), where recomputing the lane-derived values costs more than their spills
};
typedef GrxShape<0, 0, 0, 0, 0, 0, 0, 0> GrxShapeAny;
#define GRX_NVC (S::kFixed ? S::NV : m->nv)
#define GRX_NQC (S::kFixed ? S::NQ : m->nq)
#define GRX_NUC (S::kFixed ? S::NU : m->nu)
#define GRX_NBC (S::kFixed ? S::NB : m->nbody)
#define GRX_NJC (S::kFixed ? S::NJ : m->njnt)
#define GRX_NGC (S::kFixed ? S::NG : m->ngeom)
#define GRX_NSC (S::kFixed ? S::NS : m->nsite)
#define GRX_NMC (S::kFixed ? S::NM : m->nmocap)
#if defined(GRX_EMU)
// v_mfma_f32_32x32x2_f32 restated for the lane emulator: D (32 x 32) += A (32 x 2) B (2 x 32).  Lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; the element
// D[row][col] with col = l & 31 and row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5) lives in accumulator register `reg` of lane l (the layout grx_hessian's device path is written for).
static inline void grx_emu_mfma_32x32x2(const float* a, const float* b, float (*acc)[16]) {
  for (int l = 0; l < 64; l++)
    for (int reg = 0; reg < 16; reg++) {
      const int col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
      float s = acc[l][reg];
      for (int k = 0; k < 2; k++) s += a[row + 32 * k] * b[col + 32 * k];     // A[row][k] sits in lane row + 32 k, B[k][col] in lane col + 32 k
      acc[l][reg] = s;
    }
}
static long g_grx_mesh_stats[4];   // emulator diagnostics: hull pairs skipped by a cached separating direction / sent through the portal search
static int g_grx_emu_hints_on = 1; static long g_grx_hint_stats[2];   // emulator twin of the guessed support vertices (GRX_HULL_HINTS): switch (tests A/B it), guesses accepted / rejected
static long g_grx_cell_stats[4];   // emulator: hull support evaluations with a cell table / with a list in their cell / total list entries seen / near-tie vertices MISSING from a list (must be 0)
static long g_grx_newton_stats[6];   // emulator diagnostics: constrained solves, Newton iterations, full Hessian assemblies, incremental updates
#if defined(GRX_EMU_TRACE)
static void grx_emu_trace(const GrxModel* m, const GrxCtx* c, int phase);   // defined at the end of this file
#endif
#endif
#if defined(GRX_EMU) && defined(GRX_EMU_FP64) && defined(GRX_EMU_RNDINJ)
// test infrastructure (tools/emu_tolerances.py --inject): the fp64 build with fp32 ROUNDING injected at chosen stage boundaries -- which stage's fp32 storage costs the parity?
// GRX_RND_DITHER=seed (!= 0): the rounded value is additionally moved by up to one fp32 ulp at random -- the family of engines that hold this quantity in fp32
static void grx_rnd(float* p, int n) {
  static unsigned long long st_ = 0; static int dith_ = -1;
  if (dith_ < 0) { const char* e_ = getenv("GRX_RND_DITHER"); dith_ = e_ ? atoi(e_) : 0; st_ = 0x9E3779B97F4A7C15ull * (unsigned long long)(dith_ + 1); }
  for (int i = 0; i < n; i++) {
    double v = p[i];
    if (dith_) { st_ = st_ * 6364136223846793005ull + 1442695040888963407ull; const double u = (double)(st_ >> 11) * (1.0 / 9007199254740992.0); v *= 1.0 + (2.0 * u - 1.0) * 1.1920929e-07; }
    p[i] = (double)(grx_f32_t)v;
  }
}
#define GRX_RNDINJ(bit, body) do { static int mask_ = -1; if (mask_ < 0) { const char* e_ = getenv("GRX_RND_MASK"); mask_ = e_ ? atoi(e_) : 0; } if (mask_ & (1 << (bit))) { body; } } while (0)
#else
#define GRX_RNDINJ(bit, body) ((void)0)
#endif
template <class S>
struct GrxEngine {
// The stages live in fragments that are textually included HERE, inside the struct (every function is a static member of GrxEngine<S>); the order matters (a stage calls the ones above it).
#include "grx_eng_kinematics.h"
#include "grx_eng_linalg.h"
#include "grx_eng_velocity.h"
#include "grx_eng_narrowphase.h"
#include "grx_eng_convex.h"
#include "grx_eng_collision.h"
#include "grx_eng_constraint.h"
#include "grx_eng_newton.h"
#include "grx_eng_solve.h"
#include "grx_eng_integrate.h"
#include "grx_eng_sensors.h"
};  // struct GrxEngine

#if defined(GRX_EMU)
#if defined(GRX_EMU_TRACE)
// test infrastructure (tools/emu_trace.py): one record per forward pass -- contact list before the solve (phase 0), qacc / qvel after it (phase 1)
#include <stdio.h>
static void grx_emu_trace(const GrxModel* m, const GrxCtx* c, int phase) {
  static FILE* f = nullptr;
  if (!f) { const char* p = getenv("GRX_TRACE_FILE"); f = fopen(p ? p : "/tmp/grx_trace.txt", "w"); }
  if (phase == 0) {
    const int ncon = c->cnt[0] < c->maxcon ? c->cnt[0] : c->maxcon;
    fprintf(f, "PASS ncon %d nefc %d\n", ncon, c->cnt[1]);
    for (int k = 0; k < ncon; k++)
      fprintf(f, "CON pair %d g %d %d dist %.12g pos %.12g %.12g %.12g n %.12g %.12g %.12g efc %d\n", c->con_pair[k], m->pair_geom1[c->con_pair[k]], m->pair_geom2[c->con_pair[k]], (double)c->con_dist[k],
              (double)c->con_pos[3 * k], (double)c->con_pos[3 * k + 1], (double)c->con_pos[3 * k + 2], (double)c->con_frame[3 * k], (double)c->con_frame[3 * k + 1], (double)c->con_frame[3 * k + 2], c->con_efc[k]);
    fprintf(f, "SMOOTH"); for (int i = 0; i < m->nv; i++) fprintf(f, " %.12g", (double)c->qfrc_smooth[i]); fprintf(f, "\n");
    fprintf(f, "AREF"); for (int i = 0; i < c->cnt[1]; i++) fprintf(f, " %.12g", (double)c->efc_aref[i]); fprintf(f, "\n");
    fprintf(f, "EFCD"); for (int i = 0; i < c->cnt[1]; i++) fprintf(f, " %.12g", (double)c->efc_D[i]); fprintf(f, "\n");
    for (int r = 0; r < c->cnt[1]; r++) { fprintf(f, "JROW %d kind %d", r, c->efc_kind[r]); for (int i = 0; i < m->nv; i++) { const int info = c->efc_row[r], pos = GrxEngine<GrxShapeAny>::grx_row_pos(info, c->efc_id[r], i); fprintf(f, " %.12g", pos >= 0 ? (double)c->Jp[GRX_ROW_OFF(info) + pos] : 0.0); } fprintf(f, "\n"); }
    if (m->nfric) { fprintf(f, "FLOSS"); for (int i = 0; i < c->cnt[1]; i++) fprintf(f, " %.12g", (double)c->efc_floss[i]); fprintf(f, "\n"); }
    fprintf(f, "MDIAG"); for (int i = 0; i < m->nv; i++) fprintf(f, " %.12g", (double)c->M[i * m->nv + i]); fprintf(f, "\n");
    fprintf(f, "MFULL"); for (int i = 0; i < m->nv * m->nv; i++) fprintf(f, " %.12g", (double)c->M[i]); fprintf(f, "\n");
  } else {
    fprintf(f, "QVEL"); for (int i = 0; i < m->nv; i++) fprintf(f, " %.12g", (double)c->qvel[i]); fprintf(f, "\n");
    fprintf(f, "QPOS"); for (int i = 0; i < m->nq; i++) fprintf(f, " %.12g", (double)c->qpos[i]); fprintf(f, "\n");
    fprintf(f, "QACC"); for (int i = 0; i < m->nv; i++) fprintf(f, " %.12g", (double)c->qacc[i]); fprintf(f, "\n");
    fprintf(f, "EFCF"); for (int i = 0; i < c->cnt[1]; i++) fprintf(f, " %.12g", (double)c->efc_force[i]); fprintf(f, "\n");
    fprintf(f, "NEWT %d\n", c->cnt[6]);
    fflush(f);
  }
}
#endif
#endif
