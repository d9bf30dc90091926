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
};
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
#if defined(GRX_EMU)
#define GRX_FRESH_MODEL(m, c) ((void)0)
#else
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
  c->hullhint = nullptr; c->skin = nullptr; c->skin_r = 0.0f; c->bail = 0; c->soft_maxefc = c->soft_jpool = c->soft_maxcon = 0; c->lane_entry_count = nullptr; c->lane_entry_list = nullptr; c->lane_entry_cap = 0; c->lane_world = 0; c->lane_ready = nullptr; c->lane_ready_cap = 0;
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

GRX_HD GrxDims grx_dims_of(const GrxModel* m) {
  GrxDims d = {m->nq, m->nv, m->nu, m->nbody, m->njnt, m->ngeom, m->nsite, m->nmocap, m->nfric, m->integrator, m->maxefc, m->jpool, m->ntouch, m->maxcon, m->nmeshpair > 0, m->nshift > 0, m->noslip_iterations > 0};
  return d;
}
GRX_HD int grx_ctx_words(const GrxDims d) { return grx_ctx_words(d.nq, d.nv, d.nu, d.nbody, d.njnt, d.ngeom, d.nsite, d.nmocap, d.nfric, d.integrator, d.maxefc, d.jpool, d.ntouch, d.maxcon, d.nmesh, d.nshift, d.noslip); }

// ------------------------------------------------------------------------------------------
// small math (all per-lane, registers)
// ------------------------------------------------------------------------------------------
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
  static constexpr int NMESH = (CONVEX_ & 2) ? 1 : 0;
  static constexpr int NSHIFT = (CONVEX_ & 4) ? 1 : 0;   // the model has a per-world shift group (Adroit's nail board)
  static constexpr int NOSLIP = (CONVEX_ & 8) ? 1 : 0;   // the model runs the noslip post-solver
  static constexpr bool kShift = (NV_ == 0) || NSHIFT, kNoslip = (NV_ == 0) || NOSLIP;
  static constexpr bool kShiftRot = (NV_ == 0) || ((CONVEX_ & 16) != 0);   // the shift group also rotates (flag 2: Adroit pen's target body, model.body_quat edits)
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
static long g_grx_mesh_stats[4];   // emulator diagnostics: hull pairs skipped by a cached separating direction / sent through the portal search
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
// ------------------------------------------------------------------------------------------
// K1 forward kinematics
// ------------------------------------------------------------------------------------------
GRX_MEM void grx_kinematics(const GrxModel* m, GrxCtx* c, int lane_) {
  GRX_FRESH_MODEL(m, c);
  FOR_LANES {
    for (int b = lane; b < GRX_NBC; b += 64) {
      float* pl = c->ploc + 3 * b; float* ql = c->qloc + 4 * b;
      if (b == 0) {
        c->xpos[0] = c->xpos[1] = c->xpos[2] = 0; c->xquat[0] = 1; c->xquat[1] = c->xquat[2] = c->xquat[3] = 0;
        for (int k = 0; k < 9; k++) c->xmat[k] = (k % 4 == 0) ? 1.0f : 0.0f;
        continue;
      }
      int mid = m->body_mocapid[b];
      if (mid >= 0) {
        float q[4] = {c->mocap_quat[4 * mid], c->mocap_quat[4 * mid + 1], c->mocap_quat[4 * mid + 2], c->mocap_quat[4 * mid + 3]};
        normalize4f(q);
        pl[0] = c->mocap_pos[3 * mid]; pl[1] = c->mocap_pos[3 * mid + 1]; pl[2] = c->mocap_pos[3 * mid + 2];
        ql[0] = q[0]; ql[1] = q[1]; ql[2] = q[2]; ql[3] = q[3];
        continue;
      }
      const unsigned ja = (unsigned)m->body_jntadr[b]; const int jn = m->body_jntnum[b];   // unsigned: no sign-extended 64-bit index pair kept live
      int jt0 = -1, qa = 0;
      if (jn == 1) { jt0 = m->jnt_type[ja]; qa = m->jnt_qposadr[ja]; }   // both table reads before the divergent branches
#ifndef GRX_EMU
      asm volatile("" : "+v"(qa));   // keep the read here (sunk to its use, the 64-bit index pair is spilled to scratch across the branches)
#endif
      if (jt0 == 0) {  // free joint: qpos is the world pose
        float q[4] = {c->qpos[qa + 3], c->qpos[qa + 4], c->qpos[qa + 5], c->qpos[qa + 6]};
        normalize4f(q);
        pl[0] = c->qpos[qa]; pl[1] = c->qpos[qa + 1]; pl[2] = c->qpos[qa + 2];
        for (int k = 0; k < 4; k++) { ql[k] = q[k]; c->qpos[qa + 3 + k] = q[k]; }
        for (int k = 0; k < 3; k++) { c->janchor[3 * ja + k] = pl[k]; c->jaxis[3 * ja + k] = (k == 2) ? 1.0f : 0.0f; }
        continue;
      }
      float p[3] = {m->body_pos[3 * b], m->body_pos[3 * b + 1], m->body_pos[3 * b + 2]};
      if (S::kShift && m->nshift && m->body_shift[b]) { p[0] += c->shift[0]; p[1] += c->shift[1]; p[2] += c->shift[2]; }   // child of the (world-fixed) shift group
      float q[4] = {m->body_quat[4 * b], m->body_quat[4 * b + 1], m->body_quat[4 * b + 2], m->body_quat[4 * b + 3]};
      for (int k = 0; k < jn; k++) {
        const unsigned j = ja + (unsigned)k; const int qa = m->jnt_qposadr[j];
        float jp[3] = {m->jnt_pos[3 * j], m->jnt_pos[3 * j + 1], m->jnt_pos[3 * j + 2]};
        float jx[3] = {m->jnt_axis[3 * j], m->jnt_axis[3 * j + 1], m->jnt_axis[3 * j + 2]};
        float anchor[3], axis[3];
        rotVecQuatf(anchor, jp, q); anchor[0] += p[0]; anchor[1] += p[1]; anchor[2] += p[2];
        rotVecQuatf(axis, jx, q);
        for (int t = 0; t < 3; t++) { c->janchor[3 * j + t] = anchor[t]; c->jaxis[3 * j + t] = axis[t]; }
        float dq = c->qpos[qa] - m->qpos0[qa];
        if (m->jnt_type[j] == 2) {
          p[0] += axis[0] * dq; p[1] += axis[1] * dq; p[2] += axis[2] * dq;
        } else if (m->jnt_type[j] == 3) {
          float sn, cs; sincosf(0.5f * dq, &sn, &cs);
          float qr[4] = {cs, jx[0] * sn, jx[1] * sn, jx[2] * sn}, qn[4], off[3];
          mulQuatf(qn, q, qr); normalize4f(qn);
          q[0] = qn[0]; q[1] = qn[1]; q[2] = qn[2]; q[3] = qn[3];
          rotVecQuatf(off, jp, q);
          p[0] = anchor[0] - off[0]; p[1] = anchor[1] - off[1]; p[2] = anchor[2] - off[2];
        }
      }
      pl[0] = p[0]; pl[1] = p[1]; pl[2] = p[2]; ql[0] = q[0]; ql[1] = q[1]; ql[2] = q[2]; ql[3] = q[3];
    }
  }
  WAVE_SYNC();
  GRX_SUBTICK(c, 9);
  // world poses by pointer jumping: in round s every body composes its pose (relative to the ancestor 2^s levels up) with
  // that ancestor's pose (relative to ITS ancestor 2^s levels up): ceil(log2(depth)) rounds instead of one composition per
  // ancestor.  Rounds ping-pong between {ploc,qloc} and {xpos,xquat}; body_jump is the static schedule.
  const int nbk = GRX_NBC, nj = m->njump;
  for (int s = 0; s < nj; s++) {
    const float* sp = (s & 1) ? c->xpos : c->ploc; const float* sq = (s & 1) ? c->xquat : c->qloc;
    float* dp = (s & 1) ? c->ploc : c->xpos; float* dq = (s & 1) ? c->qloc : c->xquat;
    FOR_LANES {
      for (int b = 1 + lane; b < nbk; b += 64) {
        float p[3] = {sp[3 * b], sp[3 * b + 1], sp[3 * b + 2]}, q[4] = {sq[4 * b], sq[4 * b + 1], sq[4 * b + 2], sq[4 * b + 3]};
        const int anc = m->body_jump[s * nbk + b];
        if (anc > 0) {
          float qa[4] = {sq[4 * anc], sq[4 * anc + 1], sq[4 * anc + 2], sq[4 * anc + 3]}, v[3], qn[4];
          rotVecQuatf(v, p, qa);
          p[0] = sp[3 * anc] + v[0]; p[1] = sp[3 * anc + 1] + v[1]; p[2] = sp[3 * anc + 2] + v[2];
          mulQuatf(qn, qa, q);
          q[0] = qn[0]; q[1] = qn[1]; q[2] = qn[2]; q[3] = qn[3];
        }
        for (int e = 0; e < 3; e++) dp[3 * b + e] = p[e];
        for (int e = 0; e < 4; e++) dq[4 * b + e] = q[e];
      }
    }
    WAVE_SYNC();
  }
  {
    const float* sp = (nj & 1) ? c->xpos : c->ploc; const float* sq = (nj & 1) ? c->xquat : c->qloc;
    FOR_LANES {
      for (int b = 1 + lane; b < nbk; b += 64) {
        float p[3] = {sp[3 * b], sp[3 * b + 1], sp[3 * b + 2]}, q[4] = {sq[4 * b], sq[4 * b + 1], sq[4 * b + 2], sq[4 * b + 3]};
        int isfree = (m->body_jntnum[b] == 1 && m->jnt_type[m->body_jntadr[b]] == 0);
        if (!(m->body_mocapid[b] >= 0 || isfree)) normalize4f(q);
        float R[9]; quat2matf(R, q);
        for (int e = 0; e < 3; e++) c->xpos[3 * b + e] = p[e];
        for (int e = 0; e < 4; e++) c->xquat[4 * b + e] = q[e];
        for (int e = 0; e < 9; e++) c->xmat[9 * b + e] = R[e];
      }
    }
  }
  WAVE_SYNC();
  GRX_SUBTICK(c, 10);
  FOR_LANES {
    // joint anchors / axes to the world frame (they were expressed in the parent frame; free joints already are world)
    for (int j = lane; j < GRX_NJC; j += 64) {
      if (m->jnt_type[j] == 0) continue;
      int par = m->body_parent[m->jnt_bodyid[j]];
      float a_[3] = {c->janchor[3 * j], c->janchor[3 * j + 1], c->janchor[3 * j + 2]}, x_[3] = {c->jaxis[3 * j], c->jaxis[3 * j + 1], c->jaxis[3 * j + 2]}, ta[3], tx[3];
      mulMatVec3f(ta, c->xmat + 9 * par, a_); mulMatVec3f(tx, c->xmat + 9 * par, x_);
      for (int e = 0; e < 3; e++) { c->janchor[3 * j + e] = ta[e] + c->xpos[3 * par + e]; c->jaxis[3 * j + e] = tx[e]; }
    }
  }
  FOR_LANES {
    for (int i = lane; i < GRX_NSC; i += 64) {
      int b = m->site_bodyid[i];
      float lpv[3] = {m->site_pos[3 * i], m->site_pos[3 * i + 1], m->site_pos[3 * i + 2]}, lqv[4] = {m->site_quat[4 * i], m->site_quat[4 * i + 1], m->site_quat[4 * i + 2], m->site_quat[4 * i + 3]}, v[3], R[9], Rw[9];
      mulMatVec3f(v, c->xmat + 9 * b, lpv);
      const int sh = (S::kShift && m->nshift) ? m->site_shift[i] : 0;
      for (int e = 0; e < 3; e++) v[e] += c->xpos[3 * b + e];
      quat2matf(R, lqv); mulMat3f(Rw, c->xmat + 9 * b, R);
      if (S::kShiftRot && sh == 2) grx_apply_group_rotation(c->shift + 3, v, Rw);
      for (int e = 0; e < 3; e++) c->sxpos[3 * i + e] = v[e] + (sh ? c->shift[e] : 0.0f);
      for (int e = 0; e < 9; e++) c->sxmat[9 * i + e] = Rw[e];
    }
  }
  WAVE_SYNC();
}

// ------------------------------------------------------------------------------------------
// K2/K3 spatial inertias, motion axes, composite inertias, mass matrix
// reference point of each kinematic tree = xpos of its root body (any point is valid)
// ------------------------------------------------------------------------------------------
GRX_MEM void grx_inertia_cdof(const GrxModel* m, GrxCtx* c, int lane_) {
  GRX_FRESH_MODEL(m, c);
  FOR_LANES {
    for (int b = 1 + lane; b < GRX_NBC; b += 64) {
      const float* R = c->xmat + 9 * b; const float* in = m->body_inertia + 6 * b;
      const float* cref = c->xpos + 3 * m->body_rootid[b];
      float ip[3] = {m->body_ipos[3 * b], m->body_ipos[3 * b + 1], m->body_ipos[3 * b + 2]}, r[3];
      mulMatVec3f(r, R, ip);
      r[0] += c->xpos[3 * b] - cref[0]; r[1] += c->xpos[3 * b + 1] - cref[1]; r[2] += c->xpos[3 * b + 2] - cref[2];
      float Ib[9] = {in[0], in[3], in[4], in[3], in[1], in[5], in[4], in[5], in[2]}, t[9], Rt[9], Iw[9];
      for (int a = 0; a < 3; a++) for (int e = 0; e < 3; e++) Rt[3 * a + e] = R[3 * e + a];
      mulMat3f(t, R, Ib); mulMat3f(Iw, t, Rt);
      float mass = m->body_mass[b], rr = dot3f(r, r);
      float* I = c->cinert + 10 * b;
      I[0] = Iw[0] + mass * (rr - r[0] * r[0]); I[1] = Iw[4] + mass * (rr - r[1] * r[1]); I[2] = Iw[8] + mass * (rr - r[2] * r[2]);
      I[3] = Iw[1] - mass * r[0] * r[1]; I[4] = Iw[2] - mass * r[0] * r[2]; I[5] = Iw[5] - mass * r[1] * r[2];
      I[6] = mass * r[0]; I[7] = mass * r[1]; I[8] = mass * r[2]; I[9] = mass;
    }
    for (int j = lane; j < GRX_NJC; j += 64) {
      int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j], jt = m->jnt_type[j];
      const float* cref = c->xpos + 3 * m->body_rootid[b];
      float off[3] = {cref[0] - c->janchor[3 * j], cref[1] - c->janchor[3 * j + 1], cref[2] - c->janchor[3 * j + 2]};
      const float* ax = c->jaxis + 3 * j;
      if (jt == 2) {
        float* cd = c->cdof + 6 * da; cd[0] = cd[1] = cd[2] = 0; cd[3] = ax[0]; cd[4] = ax[1]; cd[5] = ax[2];
      } else if (jt == 3) {
        float* cd = c->cdof + 6 * da; float axv[3] = {ax[0], ax[1], ax[2]}, t[3];
        cross3f(t, axv, off);
        cd[0] = axv[0]; cd[1] = axv[1]; cd[2] = axv[2]; cd[3] = t[0]; cd[4] = t[1]; cd[5] = t[2];
      } else if (jt == 0) {
        for (int k = 0; k < 3; k++) { float* cd = c->cdof + 6 * (da + k); for (int e = 0; e < 6; e++) cd[e] = (e == 3 + k) ? 1.0f : 0.0f; }
        for (int k = 0; k < 3; k++) {
          float* cd = c->cdof + 6 * (da + 3 + k);
          float axv[3] = {c->xmat[9 * b + k], c->xmat[9 * b + 3 + k], c->xmat[9 * b + 6 + k]}, t[3];
          cross3f(t, axv, off);  // off = cref - xpos(body) = 0 for a root free body
          cd[0] = axv[0]; cd[1] = axv[1]; cd[2] = axv[2]; cd[3] = t[0]; cd[4] = t[1]; cd[5] = t[2];
        }
      }
    }
  }
  WAVE_SYNC();
  GRX_SUBTICK(c, 14);
  // composite inertia = sum over the subtree (no serial tree walk; membership from the static 64-bit subtree masks)
  FOR_LANES {
    for (int it = lane; it < 10 * GRX_NBC; it += 64) {
      int b = it / 10, k = it - 10 * b;
      unsigned mlo = (unsigned)m->body_submask[2 * b], mhi = (S::kFixed && S::NB <= 32) ? 0u : (unsigned)m->body_submask[2 * b + 1];
      float s = 0;
#pragma unroll 16
      for (int e = 1; e < GRX_NBC; e++) { unsigned bit = e < 32 ? (mlo >> e) & 1u : (mhi >> (e - 32)) & 1u; s += bit ? c->cinert[10 * e + k] : 0.0f; }
      c->crb[it] = s;
    }
  }
  WAVE_SYNC();
  GRX_SUBTICK(c, 15);
  FOR_LANES {
    for (int e = lane; e < m->nmpair; e += 64) {
      int i = m->mpair_i[e], j = m->mpair_j[e];
      float buf[6], cd[6];
      for (int t = 0; t < 6; t++) cd[t] = c->cdof[6 * i + t];
      inertMulf(buf, c->crb + 10 * m->dof_bodyid[i], cd);
      float v = 0;
      for (int t = 0; t < 6; t++) v += c->cdof[6 * j + t] * buf[t];
      if (i == j) v += m->dof_armature[i];
      c->M[i * GRX_NVC + j] = v; c->M[j * GRX_NVC + i] = v;
    }
  }
  WAVE_SYNC();
}

// ------------------------------------------------------------------------------------------
// K4 dense symmetric solve in LDS:  A x = b, A overwritten.  Elimination runs from the last
// dof to the first (A = L' D L), i.e. leaves of the kinematic tree before the root -- the same
// order MuJoCo's sparse L'DL uses, which keeps the 1e11-damped base dofs out of the pivots of
// everything else.  x is returned in b.
// ------------------------------------------------------------------------------------------
GRX_MEM int grx_sym_factor(float* A, int n, int lane_) {
  int bad = 0;
  for (int k = n - 1; k >= 0; k--) {
    float d = A[k * n + k];
    if (!(d > 1e-30f)) { bad = 1; d = 1e-30f; }
    float rinv = 1.0f / d;
    FOR_LANES {
      int li = lane >> 3, lj = lane & 7;
      for (int i = li; i < k; i += 8) {
        float ti = A[k * n + i] * rinv;
        for (int j = lj; j < k; j += 8) A[i * n + j] -= ti * A[k * n + j];
      }
    }
    WAVE_SYNC();
    LANE0 { A[k * n + k] = rinv; }
  }
  WAVE_SYNC();
  return bad;
}
// A holds the factor from grx_sym_factor: row k = [t_k0 .. t_k,k-1, 1/d_k]
GRX_MEM void grx_sym_solve(const float* A, int n, float* x, int lane_) {
  // L' y = b  (y_i = b_i - sum_{k>i} (t_ki/d_k) y_k)
  for (int k = n - 1; k > 0; k--) {
    float yk = x[k] * A[k * n + k];
    FOR_LANES { for (int i = lane; i < k; i += 64) x[i] -= A[k * n + i] * yk; }
    WAVE_SYNC();
  }
  FOR_LANES { for (int i = lane; i < n; i += 64) x[i] *= A[i * n + i]; }
  WAVE_SYNC();
  // L x = z  (x_k = z_k - sum_{i<k} (t_ki/d_k) x_i)
  for (int i = 0; i < n - 1; i++) {
    float xi = x[i];
    FOR_LANES { for (int k = i + 1 + lane; k < n; k += 64) x[k] -= A[k * n + i] * A[k * n + k] * xi; }
    WAVE_SYNC();
  }
}

// dof_parentid of the Shadow hand's 24 dofs (see kGrxHandAnc below; checked by the host before a hand shape is selected)
#define GRX_HAND_DOF_PARENTS {-1, 0, 1, 2, 3, 4, 1, 6, 7, 8, 1, 10, 11, 12, 1, 14, 15, 16, 17, 1, 19, 20, 21, 22}
// A x = b in one call.  On the GPU, for the dof counts of the models in scope, the whole system is held in
// registers: lane j owns column j of A (lane nv owns b), the pivot column is broadcast with v_readlane and the
// elimination runs from the last dof to the first exactly like grx_sym_factor -- no LDS round trips, no barriers.
#if !defined(GRX_EMU)
// v_readlane_b32 moves raw bits: the builtin is typed (int,int), so floats go through a bit cast
static __device__ __forceinline__ float grx_readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// reciprocal of a strictly positive pivot: v_rcp_f32 (1 ulp) + one Newton step
static __device__ __forceinline__ float grx_rcp_refined(float d) { float r = __builtin_amdgcn_rcpf(d); return fmaf(fmaf(-d, r, 1.0f), r, r); }
// Lane i (< NS) owns ROW i of the symmetric matrix and b_i.  Step k of the elimination broadcasts row k with v_readlane
// (one readlane + one fma per remaining column) and every lane i < k subtracts its multiple of it; rows end up lower
// triangular, pivots final when they are used.  The forward substitution then needs one broadcast per unknown.
// Dof tree of the Shadow hand (shared_asset / robot.xml of the hand models): wrist 0-1, then five chains hanging off dof 1
// (FF 2-5, MF 6-9, RF 10-13, LF 14-18, TH 19-23).  kGrxHandAnc[k] = the ancestor dofs of dof k as a bit mask.  M and M + h B have
// exactly this pattern below the diagonal, and the last-to-first elimination creates no fill-in (the LTDL argument mj_factorM relies
// on), so the HAND variant of the register solve broadcasts only those columns: 83 instead of 276, decided at compile time, the
// skipped updates being exact zeros of the dense elimination.  grx_fill_model_scalars sets m->handtree only if dof_parentid matches.
static constexpr unsigned kGrxHandAnc[24] = {0x0, 0x1, 0x3, 0x7, 0xF, 0x1F, 0x3, 0x43, 0xC3, 0x1C3, 0x3, 0x403, 0xC03, 0x1C03,
                                             0x3, 0x4003, 0xC003, 0x1C003, 0x3C003, 0x3, 0x80003, 0x180003, 0x380003, 0x780003};
template <int NS, bool HAND = false>
  static __device__ __forceinline__ int grx_sym_solve_reg(const float* A, int ld, float* x, int lane_) {
  float a[NS];
  const int row = lane_ < NS ? lane_ : 0;
#pragma unroll
  for (int i = 0; i < NS; i++) a[i] = A[row * ld + i];     // ld: row stride (a diagonal block of a larger matrix can be solved in place)
  float b = x[row], rd = 0.0f;
#pragma unroll
  for (int k = NS - 1; k > 0; k--) {
    const float pinv = grx_rcp_refined(grx_readlane_f(a[k], k));
    rd = (lane_ == k) ? pinv : rd;
    const float mi = (lane_ < k) ? -a[k] * pinv : 0.0f;
#pragma unroll
    for (int j = 0; j < k; j++) { if (!HAND || ((kGrxHandAnc[k % 24] >> j) & 1u)) a[j] = fmaf(mi, grx_readlane_f(a[j], k), a[j]); }
    b = fmaf(mi, grx_readlane_f(b, k), b);
  }
  { const float pinv = grx_rcp_refined(grx_readlane_f(a[0], 0)); rd = (lane_ == 0) ? pinv : rd; }
  float xo = 0.0f;
#pragma unroll
  for (int k = 0; k < NS; k++) {
    const float t = b * rd;                 // lane k: b_k / L_kk = x_k (b_k is final once x_0 .. x_k-1 have been applied)
    const float xk = grx_readlane_f(t, k);
    xo = (lane_ == k) ? t : xo;
    b = fmaf(-a[k], xk, b);                 // lanes i > k: b_i -= L_ik x_k ; lanes <= k are finished, their b is dead
  }
  __syncthreads();
  if (lane_ < NS) x[lane_] = xo;
  __syncthreads();
  return __ballot((lane_ < NS) && !(rd > 0.0f)) != 0ull;   // a non-positive (or NaN) pivot: the matrix was not positive definite (GRX_ST_FACTOR)
}
// Newton Hessian of the hand models, H = M + J' D J (24 hand dofs [+ NOBJ = 6 dofs of a free object]).  M has the tree pattern; a limit / tendon row
// touches dofs of one chain; a contact between a finger and the object couples that finger's chain (and the wrist) with the object dofs -- so, unless
// two DIFFERENT fingers touch each other, H's hand block keeps the tree pattern and everything else sits in the object's rows and columns.
// Eliminating the hand dofs leaf-to-root FIRST and the object dofs LAST creates no fill outside that pattern (a pivot k couples anc(k) and the
// object among themselves: ancestors of one dof form a chain), so a pivot broadcasts |anc(k)| + NOBJ columns instead of all the remaining ones:
// 83 + 24 NOBJ + NOBJ (NOBJ - 1) / 2 = 242 against 435 for the 30 dofs of hand + object; the skipped updates are exact zeros.  LINKED = false:
// no active row links hand and object (the caller looked), the object columns are skipped as well (83 + 15).
// The pattern is CHECKED on the values (every lane scans the off-pattern part of its row: 24 compares); returns -1 without touching x when an
// off-pattern entry is non-zero (finger-finger contact): the caller falls back to the dense elimination.
static constexpr unsigned kGrxHandAncTable[32] = {0x0, 0x1, 0x3, 0x7, 0xF, 0x1F, 0x3, 0x43, 0xC3, 0x1C3, 0x3, 0x403, 0xC03, 0x1C03,
                                                  0x3, 0x4003, 0xC003, 0x1C003, 0x3C003, 0x3, 0x80003, 0x180003, 0x380003, 0x780003,
                                                  0xFFFFFF, 0xFFFFFF, 0xFFFFFF, 0xFFFFFF, 0xFFFFFF, 0xFFFFFF, 0xFFFFFF, 0xFFFFFF};
template <int NOBJ, bool LINKED>
  static __device__ __forceinline__ int grx_sym_solve_hand(const float* A, int ld, float* x, int lane_) {
  constexpr int NH = 24, NS = NH + NOBJ;
  float a[NS];
  const int row = lane_ < NS ? lane_ : 0;
#pragma unroll
  for (int i = 0; i < NS; i++) a[i] = A[row * ld + i];
  {
    const unsigned anc = kGrxHandAncTable[row & 31];
    int off = 0;
#pragma unroll
    for (int j = 0; j < NH - 1; j++) off |= (j < row) && !((anc >> j) & 1u) && (a[j] != 0.0f);
    if (__ballot(off && lane_ < NH) != 0ull) return -1;
  }
  float b = x[row], rd = 0.0f;
  const bool obj = lane_ >= NH;
#pragma unroll
  for (int k = NH - 1; k >= 0; k--) {          // hand pivots, leaf to root; remaining rows: hand dofs < k (only the ancestors hold a non-zero a[k]) and the object
    const float pinv = grx_rcp_refined(grx_readlane_f(a[k], k));
    rd = (lane_ == k) ? pinv : rd;
    const float mi = (lane_ < k || (LINKED && obj)) ? -a[k] * pinv : 0.0f;
#pragma unroll
    for (int j = 0; j < k; j++) { if ((kGrxHandAnc[k] >> j) & 1u) a[j] = fmaf(mi, grx_readlane_f(a[j], k), a[j]); }
    if (LINKED) {
#pragma unroll
      for (int j = NH; j < NS; j++) a[j] = fmaf(mi, grx_readlane_f(a[j], k), a[j]);
    }
    b = fmaf(mi, grx_readlane_f(b, k), b);
  }
#pragma unroll
  for (int k = NS - 1; k >= NH; k--) {         // object pivots: a dense NOBJ x NOBJ block
    const float pinv = grx_rcp_refined(grx_readlane_f(a[k], k));
    rd = (lane_ == k) ? pinv : rd;
    const float mi = (obj && lane_ < k) ? -a[k] * pinv : 0.0f;
#pragma unroll
    for (int j = NH; j < k; j++) a[j] = fmaf(mi, grx_readlane_f(a[j], k), a[j]);
    b = fmaf(mi, grx_readlane_f(b, k), b);
  }
  // substitution in the reverse order of the elimination: object dofs first, then the hand dofs root to leaf
  float xo = 0.0f;
#pragma unroll
  for (int kk = 0; kk < NS; kk++) {
    const int k = kk < NOBJ ? NH + kk : kk - NOBJ;
    const float t = b * rd;
    const float xk = grx_readlane_f(t, k);
    xo = (lane_ == k) ? t : xo;
    if (k >= NH && !LINKED) b = obj ? fmaf(-a[k], xk, b) : b;
    else b = fmaf(-a[k], xk, b);
  }
  __syncthreads();
  if (lane_ < NS) x[lane_] = xo;
  __syncthreads();
  return __ballot((lane_ < NS) && !(rd > 0.0f)) != 0ull;
}
// In-place Gauss-Jordan inverse of a symmetric positive definite matrix, same register layout (lane i = row i, NS registers): step k broadcasts
// row k with v_readlane, every other row subtracts its multiple of it, the pivot column becomes the k-th column of the inverse.  No LDS traffic,
// no barrier: ~2 NS^2 instructions against ~NS^3 / 8 dependent LDS round trips of the factor-and-substitute route (noslip needs all of M^-1).
template <int NS>
  static __device__ __forceinline__ void grx_sym_inverse_reg(const float* A, int ld, float* out, int lane_) {
  float a[NS];
  const int row = lane_ < NS ? lane_ : 0;
#pragma unroll
  for (int i = 0; i < NS; i++) a[i] = A[row * ld + i];
#pragma unroll
  for (int k = 0; k < NS; k++) {
    const float p = grx_rcp_refined(grx_readlane_f(a[k], k));
    const bool own = (lane_ == k);
    const float f = own ? 0.0f : a[k] * p;
#pragma unroll
    for (int j = 0; j < NS; j++) {
      if (j == k) continue;
      const float akj = grx_readlane_f(a[j], k);
      a[j] = own ? akj * p : fmaf(-f, akj, a[j]);
    }
    a[k] = own ? p : -f;
  }
  __syncthreads();
  if (lane_ < NS) {
#pragma unroll
    for (int i = 0; i < NS; i++) out[lane_ * ld + i] = a[i];
  }
  __syncthreads();
}
#endif

// nsplit: the last nsplit (= 6) dofs are a free object whose block of A is decoupled from the rest (all entries between the two
// blocks are exactly zero: always true for M + h B, true for the Hessian while no contact links object and robot).  The two diagonal
// blocks are then solved one after the other -- the same arithmetic as the full elimination, in which every multiplier between the
// blocks is an exact zero, at (nr^2 + 36) / n^2 of its broadcasts (15 + 6 instead of 21: 43 % fewer).
GRX_MEM int grx_sym_solve_full(float* A, int n, float* x, int lane_, int nsplit = 0, int tree = 0) {
#if !defined(GRX_EMU)
  if (S::kFixed && S::NF == 24 && tree) {      // hand shapes (the host matched m->handtree): M / M + h B solves
    int bad_ = grx_sym_solve_reg<24, true>(A, n, x, lane_);
    if (S::NV == 30) bad_ |= grx_sym_solve_reg<6>(A + 24 * n + 24, n, x + 24, lane_);
    return bad_;
  }
  if (S::kFixed && S::NF == 24 && !tree) {     // hand shapes, Newton Hessian: structured elimination while no two fingers touch each other
    int r_;
    if (S::NV == 30) r_ = nsplit == 6 ? grx_sym_solve_hand<6, false>(A, n, x, lane_) : grx_sym_solve_hand<6, true>(A, n, x, lane_);
    else r_ = grx_sym_solve_hand<0, false>(A, n, x, lane_);
    if (r_ >= 0) return r_;
  }
  if (nsplit == 6 && n == 21) { int bad_ = grx_sym_solve_reg<15>(A, n, x, lane_); return bad_ | grx_sym_solve_reg<6>(A + 15 * n + 15, n, x + 15, lane_); }
  if (nsplit == 6 && n == 30) { int bad_ = grx_sym_solve_reg<24>(A, n, x, lane_); return bad_ | grx_sym_solve_reg<6>(A + 24 * n + 24, n, x + 24, lane_); }
  if (n == 21) return grx_sym_solve_reg<21>(A, n, x, lane_);
  if (n == 14) return grx_sym_solve_reg<14>(A, n, x, lane_);
  if (n == 15) return grx_sym_solve_reg<15>(A, n, x, lane_);
  if (n == 24) return grx_sym_solve_reg<24>(A, n, x, lane_);
  if (n == 29) return grx_sym_solve_reg<29>(A, n, x, lane_);
  if (n == 30) return grx_sym_solve_reg<30>(A, n, x, lane_);
  if (n == 33) return grx_sym_solve_reg<33>(A, n, x, lane_);
  if (n == 36) return grx_sym_solve_reg<36>(A, n, x, lane_);
#endif
#if defined(GRX_EMU)
  if (n == 21 || n == 14 || n == 15 || n == 24 || n == 29 || n == 30 || n == 33 || n == 36) {   // mirror the device: these sizes are solved without touching A
    static float copy[36 * 36];
    for (int i = 0; i < n * n; i++) copy[i] = A[i];
    int bad_ = grx_sym_factor(copy, n, lane_);
    grx_sym_solve(copy, n, x, lane_);
    return bad_;
  }
#endif
  int bad = grx_sym_factor(A, n, lane_);
  grx_sym_solve(A, n, x, lane_);
  return bad;
}

// ------------------------------------------------------------------------------------------
// velocity stage: cvel, cdof_dot, RNE bias (K2/K5), passive (K6), actuation (K7)
// ------------------------------------------------------------------------------------------
GRX_MEM void grx_velocity(const GrxModel* m, GrxCtx* c, int lane_) {
  GRX_FRESH_MODEL(m, c);
  const int nv = GRX_NVC;
  // body spatial velocities = sum over the dof chain (parallel, no tree walk)
  FOR_LANES {
    for (int it = lane; it < 6 * GRX_NBC; it += 64) {
      int b = it / 6, k = it - 6 * b;
      unsigned mlo = (unsigned)m->dof_chainmask[2 * b], mhi = (unsigned)m->dof_chainmask[2 * b + 1];
      float s = 0;
#pragma unroll 8
      for (int d = 0; d < nv; d++) { unsigned bit = d < 32 ? (mlo >> d) & 1u : (mhi >> (d - 32)) & 1u; s += bit ? c->cdof[6 * d + k] * c->qvel[d] : 0.0f; }
      c->cvel[it] = s;
    }
    // passive forces
    for (int d = lane; d < nv; d += 64) {
      float f = -m->dof_damping[d] * c->qvel[d];
      int j = m->dof_jntid[d];
      if (m->jnt_stiffness[j] != 0.0f && m->jnt_type[j] >= 2) f -= m->jnt_stiffness[j] * (c->qpos[m->jnt_qposadr[j]] - m->jnt_springref[j]);
      c->qfrc_passive[d] = f;
      c->qfrc_actuator[d] = 0;
    }
  }
  WAVE_SYNC();
  FOR_LANES {
    // cdof_dot = crossMotion(velocity just before this dof, cdof).  That velocity is the spatial velocity of the body
    // owning dof_cvelstart[d], minus the dofs of that body that come after it (only multi-dof joints have any).
    for (int d = lane; d < nv; d += 64) {
      float v[6] = {0, 0, 0, 0, 0, 0}, cd[6], r[6];
      const int e0 = m->dof_cvelstart[d];
      if (e0 >= 0) {
        const int bb = m->dof_bodyid[e0], last = m->body_dofadr[bb] + m->body_dofnum[bb] - 1;
        for (int k = 0; k < 6; k++) v[k] = c->cvel[6 * bb + k];
        for (int e = e0 + 1; e <= last; e++) { float qd = c->qvel[e]; for (int k = 0; k < 6; k++) v[k] -= c->cdof[6 * e + k] * qd; }
      }
      for (int k = 0; k < 6; k++) cd[k] = c->cdof[6 * d + k];
      int jt = m->jnt_type[m->dof_jntid[d]];
      if (jt == 0 && d - m->jnt_dofadr[m->dof_jntid[d]] < 3) { for (int k = 0; k < 6; k++) r[k] = 0; }
      else crossMotionf(r, v, cd);
      for (int k = 0; k < 6; k++) c->cdof_dot[6 * d + k] = r[k];
    }
  }
  WAVE_SYNC();
  GRX_SUBTICK(c, 6);
  FOR_LANES {
    // accelerations with qacc = 0 and per-body inertial forces
    for (int b = 1 + lane; b < GRX_NBC; b += 64) {
      float a[6] = {0, 0, 0, -m->gravity[0], -m->gravity[1], -m->gravity[2]}, v[6], Ia[6], Iv[6], t[6];
      unsigned mlo = (unsigned)m->dof_chainmask[2 * b], mhi = (unsigned)m->dof_chainmask[2 * b + 1];
#pragma unroll 4
      for (int d = 0; d < nv; d++) {
        unsigned bit = d < 32 ? (mlo >> d) & 1u : (mhi >> (d - 32)) & 1u;
        float qd = bit ? c->qvel[d] : 0.0f;
        for (int k = 0; k < 6; k++) a[k] += c->cdof_dot[6 * d + k] * qd;
      }
      for (int k = 0; k < 6; k++) v[k] = c->cvel[6 * b + k];
      inertMulf(Ia, c->cinert + 10 * b, a); inertMulf(Iv, c->cinert + 10 * b, v);
      crossForcef(t, v, Iv);
      for (int k = 0; k < 6; k++) c->cacc[6 * b + k] = Ia[k] + t[k];
    }
    // actuators (one lane each; joint transmission)
    for (int i = lane; i < GRX_NUC; i += 64) {
      int j = m->act_trnid[i]; float gear = m->act_gear[i];
      float len = gear * c->qpos[m->jnt_qposadr[j]], vel = gear * c->qvel[m->jnt_dofadr[j]];
      float u = c->ctrl[i];
      if (m->act_ctrllimited[i]) u = fminf(m->act_ctrlrange[2 * i + 1], fmaxf(m->act_ctrlrange[2 * i], u));
      float gain = m->act_gainprm[3 * i];
      if (m->act_gaintype[i] == 1) gain += m->act_gainprm[3 * i + 1] * len + m->act_gainprm[3 * i + 2] * vel;
      float bias = 0;
      if (m->act_biastype[i] == 1) bias = m->act_biasprm[3 * i] + m->act_biasprm[3 * i + 1] * len + m->act_biasprm[3 * i + 2] * vel;
      float f = gain * u + bias;
      if (m->act_forcelimited[i]) f = fminf(m->act_forcerange[2 * i + 1], fmaxf(m->act_forcerange[2 * i], f));
      c->qfrc_actuator[m->jnt_dofadr[j]] = gear * f;  // models in scope have at most one actuator per dof
    }
  }
  WAVE_SYNC();
  GRX_SUBTICK(c, 7);
  FOR_LANES {
    // subtree force sums
    for (int it = lane; it < 6 * GRX_NBC; it += 64) {
      int b = it / 6, k = it - 6 * b;
      unsigned mlo = (unsigned)m->body_submask[2 * b], mhi = (S::kFixed && S::NB <= 32) ? 0u : (unsigned)m->body_submask[2 * b + 1];
      float s = 0;
#pragma unroll 16
      for (int e = 1; e < GRX_NBC; e++) { unsigned bit = e < 32 ? (mlo >> e) & 1u : (mhi >> (e - 32)) & 1u; s += bit ? c->cacc[6 * e + k] : 0.0f; }
      c->cfrc[it] = s;
    }
  }
  WAVE_SYNC();
  GRX_SUBTICK(c, 8);
  FOR_LANES {
    for (int d = lane; d < nv; d += 64) {
      float s = 0; int b = m->dof_bodyid[d];
      for (int k = 0; k < 6; k++) s += c->cdof[6 * d + k] * c->cfrc[6 * b + k];
      c->qfrc_bias[d] = s;
      float f = c->qfrc_passive[d] - s + c->qfrc_actuator[d];
      c->qfrc_smooth[d] = f; c->qacc_smooth[d] = f;
    }
  }
  WAVE_SYNC();
}

// ------------------------------------------------------------------------------------------
// K8 collision: static candidate list -> narrow phase
// ------------------------------------------------------------------------------------------
GRX_MEM void grx_make_frame(float* f) {
  float* x = f; float* y = f + 3; float* z = f + 6;
  if (x[1] < 0.5f && x[1] > -0.5f) { y[0] = 0; y[1] = 1; y[2] = 0; } else { y[0] = 0; y[1] = 0; y[2] = 1; }
  float d = dot3f(x, y); y[0] -= d * x[0]; y[1] -= d * x[1]; y[2] -= d * x[2];
  float n = 1.0f / sqrtf(dot3f(y, y)); y[0] *= n; y[1] *= n; y[2] *= n;
  cross3f(z, x, y);
}

// contact append: slot from an LDS counter; only the normal is stored here, the tangent frame is completed by
// grx_make_constraint (one lane per contact)
GRX_MEM void grx_add_contact(GrxCtx* c, int pair, const float* pos, const float* normal, float dist) {
  int slot = GRX_ATOMIC_ADD(&c->cnt[0], 1);
  if (slot >= c->maxcon) { c->cnt[2] |= GRX_ST_CON_OVERFLOW; return; }
  c->con_dist[slot] = dist; c->con_pair[slot] = pair;
  for (int k = 0; k < 3; k++) { c->con_pos[3 * slot + k] = pos[k]; c->con_frame[3 * slot + k] = normal[k]; }
}

GRX_MEM void grx_plane_box(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, float margin) {
  const float* pp = c->gxpos + 3 * g1; const float* pm = c->gxmat + 9 * g1;
  const float* bp = c->gxpos + 3 * g2; const float* bm = c->gxmat + 9 * g2; const float* sz = m->geom_size + 3 * g2;
  float n[3] = {pm[2], pm[5], pm[8]};
  float sx = sz[0], sy = sz[1], szz = sz[2];
  int cnt = 0;
  for (int k = 0; k < 8; k++) {
    float loc[3] = {(k & 1) ? sx : -sx, (k & 2) ? sy : -sy, (k & 4) ? szz : -szz}, w[3];
    mulMatVec3f(w, bm, loc); w[0] += bp[0]; w[1] += bp[1]; w[2] += bp[2];
    float d[3] = {w[0] - pp[0], w[1] - pp[1], w[2] - pp[2]};
    float dist = dot3f(d, n);
    if (dist > margin || cnt >= 4) continue;
    float pos[3] = {w[0] - 0.5f * dist * n[0], w[1] - 0.5f * dist * n[1], w[2] - 0.5f * dist * n[2]};
    grx_add_contact(c, pair, pos, n, dist); cnt++;
  }
}

#define GRX_SEL3(a0, a1, a2, i) ((i) == 0 ? (a0) : ((i) == 1 ? (a1) : (a2)))
#define GRX_SEL6(v, i) ((i) == 0 ? (v)[0] : ((i) == 1 ? (v)[1] : ((i) == 2 ? (v)[2] : ((i) == 3 ? (v)[3] : ((i) == 4 ? (v)[4] : (v)[5])))))

GRX_MEM void grx_plane_sphere(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, float margin) {
  float n[3] = {c->gxmat[9 * g1 + 2], c->gxmat[9 * g1 + 5], c->gxmat[9 * g1 + 8]};
  const float* ce = c->gxpos + 3 * g2; float r = m->geom_size[3 * g2];
  float d[3] = {ce[0] - c->gxpos[3 * g1], ce[1] - c->gxpos[3 * g1 + 1], ce[2] - c->gxpos[3 * g1 + 2]};
  float dist = dot3f(d, n) - r;
  if (dist > margin) return;
  float pos[3] = {ce[0] - n[0] * (r + 0.5f * dist), ce[1] - n[1] * (r + 0.5f * dist), ce[2] - n[2] * (r + 0.5f * dist)};
  grx_add_contact(c, pair, pos, n, dist);
}

// sphere (geom1) vs box (geom2): closest point of the box to the sphere centre; normal from the sphere to the box
GRX_MEM void grx_sphere_box(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, float margin) {
  const float* ce = c->gxpos + 3 * g1; float r = m->geom_size[3 * g1];
  const float* bp = c->gxpos + 3 * g2; const float* bm = c->gxmat + 9 * g2; const float* sz = m->geom_size + 3 * g2;
  float dw[3] = {ce[0] - bp[0], ce[1] - bp[1], ce[2] - bp[2]}, loc[3];
  mulMatTVec3f(loc, bm, dw);
  float s0 = sz[0], s1 = sz[1], s2 = sz[2];
  float c0 = fminf(s0, fmaxf(-s0, loc[0])), c1 = fminf(s1, fmaxf(-s1, loc[1])), c2 = fminf(s2, fmaxf(-s2, loc[2]));
  float nl[3], dist;
  if (c0 != loc[0] || c1 != loc[1] || c2 != loc[2]) {
    float dv[3] = {c0 - loc[0], c1 - loc[1], c2 - loc[2]};
    float len = sqrtf(dot3f(dv, dv));
    dist = len - r;
    if (dist > margin) return;
    float li = 1.0f / len; nl[0] = dv[0] * li; nl[1] = dv[1] * li; nl[2] = dv[2] * li;
  } else {
    float d0 = s0 - fabsf(loc[0]), d1 = s1 - fabsf(loc[1]), d2 = s2 - fabsf(loc[2]);
    int ax = 0; float best = d0;
    if (d1 < best) { best = d1; ax = 1; }
    if (d2 < best) { best = d2; ax = 2; }
    float sg = (GRX_SEL3(loc[0], loc[1], loc[2], ax) >= 0) ? -1.0f : 1.0f;
    nl[0] = (ax == 0) ? sg : 0.0f; nl[1] = (ax == 1) ? sg : 0.0f; nl[2] = (ax == 2) ? sg : 0.0f;
    dist = -best - r;
  }
  float n[3]; mulMatVec3f(n, bm, nl);
  float pos[3] = {ce[0] + n[0] * (r + 0.5f * dist), ce[1] + n[1] * (r + 0.5f * dist), ce[2] + n[2] * (r + 0.5f * dist)};
  grx_add_contact(c, pair, pos, n, dist);
}

// plane vs capsule: the two end spheres
GRX_MEM void grx_plane_capsule(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, float margin) {
  float n[3] = {c->gxmat[9 * g1 + 2], c->gxmat[9 * g1 + 5], c->gxmat[9 * g1 + 8]};
  const float* ce = c->gxpos + 3 * g2; const float* R = c->gxmat + 9 * g2;
  float r = m->geom_size[3 * g2], hl = m->geom_size[3 * g2 + 1], ax[3] = {R[2], R[5], R[8]};
  for (int e = -1; e <= 1; e += 2) {
    float p[3] = {ce[0] + e * hl * ax[0], ce[1] + e * hl * ax[1], ce[2] + e * hl * ax[2]};
    float d[3] = {p[0] - c->gxpos[3 * g1], p[1] - c->gxpos[3 * g1 + 1], p[2] - c->gxpos[3 * g1 + 2]};
    float dist = dot3f(d, n) - r;
    if (dist > margin) continue;
    float pos[3] = {p[0] - n[0] * (r + 0.5f * dist), p[1] - n[1] * (r + 0.5f * dist), p[2] - n[2] * (r + 0.5f * dist)};
    grx_add_contact(c, pair, pos, n, dist);
  }
}

GRX_MEM float grx_box_point_dist2(float s0, float s1, float s2, float p0, float p1, float p2) {
  float d0 = p0 - fminf(s0, fmaxf(-s0, p0)), d1 = p1 - fminf(s1, fmaxf(-s1, p1)), d2 = p2 - fminf(s2, fmaxf(-s2, p2));
  return d0 * d0 + d1 * d1 + d2 * d2;
}
// sphere of radius r at box-frame point p against the box (normal from the sphere to the box); returns 1 if a contact was made
GRX_MEM int grx_sphere_box_local(GrxCtx* c, int pair, const float* bp, const float* bm, float s0, float s1, float s2, const float* p, float r, float margin) {
  float c0 = fminf(s0, fmaxf(-s0, p[0])), c1 = fminf(s1, fmaxf(-s1, p[1])), c2 = fminf(s2, fmaxf(-s2, p[2]));
  float nl[3], dist;
  if (c0 != p[0] || c1 != p[1] || c2 != p[2]) {
    float dv[3] = {c0 - p[0], c1 - p[1], c2 - p[2]};
    float len = sqrtf(dot3f(dv, dv));
    dist = len - r;
    if (dist > margin) return 0;
    float li = 1.0f / len; nl[0] = dv[0] * li; nl[1] = dv[1] * li; nl[2] = dv[2] * li;
  } else {
    float d0 = s0 - fabsf(p[0]), d1 = s1 - fabsf(p[1]), d2 = s2 - fabsf(p[2]);
    int ax = 0; float best = d0;
    if (d1 < best) { best = d1; ax = 1; }
    if (d2 < best) { best = d2; ax = 2; }
    float sg = (GRX_SEL3(p[0], p[1], p[2], ax) >= 0) ? -1.0f : 1.0f;
    nl[0] = (ax == 0) ? sg : 0.0f; nl[1] = (ax == 1) ? sg : 0.0f; nl[2] = (ax == 2) ? sg : 0.0f;
    dist = -best - r;
  }
  float n[3], pw[3]; mulMatVec3f(n, bm, nl); mulMatVec3f(pw, bm, p);
  float pos[3] = {pw[0] + bp[0] + n[0] * (r + 0.5f * dist), pw[1] + bp[1] + n[1] * (r + 0.5f * dist), pw[2] + bp[2] + n[2] * (r + 0.5f * dist)};
  grx_add_contact(c, pair, pos, n, dist);
  return 1;
}
// capsule vs capsule: closest points of the two axis segments (clamped), then a sphere-sphere contact (see oracle/grx_oracle.c)
GRX_MEM void grx_capsule_capsule(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, float margin) {
  const float* c1 = c->gxpos + 3 * g1; const float* R1 = c->gxmat + 9 * g1; const float* c2 = c->gxpos + 3 * g2; const float* R2 = c->gxmat + 9 * g2;
  const float r1 = m->geom_size[3 * g1], h1 = m->geom_size[3 * g1 + 1], r2 = m->geom_size[3 * g2], h2 = m->geom_size[3 * g2 + 1];
  const float a1[3] = {R1[2], R1[5], R1[8]}, a2[3] = {R2[2], R2[5], R2[8]}, w[3] = {c1[0] - c2[0], c1[1] - c2[1], c1[2] - c2[2]};
  const float b = dot3f(a1, a2), d = dot3f(a1, w), e = dot3f(a2, w), den = 1.0f - b * b;
  float x1 = den > GRX_MINVAL ? (b * e - d) / den : 0.0f;
  x1 = fminf(h1, fmaxf(-h1, x1));
  float x2 = b * x1 + e;
  if (x2 > h2) { x2 = h2; x1 = fminf(h1, fmaxf(-h1, b * x2 - d)); }
  else if (x2 < -h2) { x2 = -h2; x1 = fminf(h1, fmaxf(-h1, b * x2 - d)); }
  float p1[3], n[3];
  for (int k = 0; k < 3; k++) { p1[k] = c1[k] + x1 * a1[k]; n[k] = c2[k] + x2 * a2[k] - p1[k]; }
  const float len = sqrtf(dot3f(n, n));
  if (len < GRX_MINVAL) { n[0] = 1; n[1] = n[2] = 0; } else { const float li = 1.0f / len; n[0] *= li; n[1] *= li; n[2] *= li; }
  const float dist = len - r1 - r2;
  if (dist > margin) return;
  float pos[3] = {p1[0] + n[0] * (r1 + 0.5f * dist), p1[1] + n[1] * (r1 + 0.5f * dist), p1[2] + n[2] * (r1 + 0.5f * dist)};
  grx_add_contact(c, pair, pos, n, dist);
}
// sphere vs sphere and sphere (geom1) vs capsule (geom2): the capsule contributes the point of its axis segment closest to the sphere centre
GRX_MEM void grx_sphere_sphere_raw(GrxCtx* c, int pair, const float* c1, float r1, const float* c2, float r2, float margin) {
  float n[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]};
  const float len = sqrtf(dot3f(n, n)), dist = len - r1 - r2;
  if (dist > margin) return;
  if (len < GRX_MINVAL) { n[0] = 1; n[1] = n[2] = 0; } else { const float li = 1.0f / len; n[0] *= li; n[1] *= li; n[2] *= li; }
  float pos[3] = {c1[0] + n[0] * (r1 + 0.5f * dist), c1[1] + n[1] * (r1 + 0.5f * dist), c1[2] + n[2] * (r1 + 0.5f * dist)};
  grx_add_contact(c, pair, pos, n, dist);
}
GRX_MEM void grx_sphere_capsule(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, float margin) {
  const float* c1 = c->gxpos + 3 * g1; const float* c2 = c->gxpos + 3 * g2; const float* R2 = c->gxmat + 9 * g2;
  const float ax[3] = {R2[2], R2[5], R2[8]}, d[3] = {c1[0] - c2[0], c1[1] - c2[1], c1[2] - c2[2]};
  const float h = m->geom_size[3 * g2 + 1], x = fminf(h, fmaxf(-h, dot3f(ax, d)));
  const float p2[3] = {c2[0] + x * ax[0], c2[1] + x * ax[1], c2[2] + x * ax[2]};
  grx_sphere_sphere_raw(c, pair, c1, m->geom_size[3 * g1], p2, m->geom_size[3 * g2], margin);
}
// ------------------------------------------------------------------------------------------
// General convex pairs (ellipsoid / cylinder against sphere, capsule, ellipsoid, cylinder, box): Minkowski Portal Refinement, one
// lane per pair, one contact per pair (what MuJoCo's libccd route produces; see the oracle's header comment on the algorithm and on
// what is not restated).  Everything is computed relative to the centre of geom 1, so the fp32 support points are O(geom size)
// instead of O(world coordinates); a Minkowski point is kept with its witness on geom 1 (the witness on geom 2 is w - v).
// ------------------------------------------------------------------------------------------
// The "is it zero / are they equal" thresholds of the portal routine are part of the ALGORITHM MuJoCo runs (libccd's CCD_EPS, built in double
// precision: 2.2e-16), not a statement about this build's arithmetic: several of the tests compare triple products of portal vertices (scale
// size^3 ~ 1e-5 for centimetre geoms) against it, and with the fp32 machine epsilon (1.2e-7: what rounds 1 to 1 in THIS arithmetic) the routine
// took other branches than the reference's in 20 % of the resting egg contacts -- all of the egg / puck / door discrepancy of round 2 was this
// one constant (the fp64 oracle compiled with 1.2e-7 reproduces the round-2 error table digit for digit; tools/emu_tolerances.py).  The rounding
// noise of fp32 in the same tests only moves decisions that are ties in exact arithmetic.
#ifndef GRX_MPR_EPS
#define GRX_MPR_EPS 2.220446e-16f
#endif
// Arithmetic type of the general convex routine (portal refinement + its support functions): GRX_MPR_REAL.  The routine's branch decisions compare
// triple products of nearly coplanar portal vertices and its final triangle is the size of a resting contact's depth, so fp32 rounding inside it moves the
// contact POINT of a line / face contact by centimetres (tools/emu_trace.py); it runs for a handful of pairs per substep, which is why it can afford fp64.
#ifndef GRX_MPR_REAL
#define GRX_MPR_REAL double
#endif
typedef GRX_MPR_REAL MF;
#ifndef GRX_HULL_REAL
#define GRX_HULL_REAL float
#endif
typedef GRX_HULL_REAL HF;   // arithmetic of the hull support scan (vertex tables are fp32)
#ifndef GRX_TIE_REAL
#define GRX_TIE_REAL double
#endif
typedef GRX_TIE_REAL TF;    // arithmetic that decides between hull vertices whose fp32 projections tie (grx_mesh_support)
GRX_MEM float grx_sqrt(float x) { return sqrtf(x); }
GRX_MEM float grx_fabs(float x) { return fabsf(x); }
GRX_MEM float grx_fmin(float a, float b) { return fminf(a, b); }
GRX_MEM float grx_fmax(float a, float b) { return fmaxf(a, b); }
#ifndef GRX_EMU_FP64
GRX_MEM double grx_sqrt(double x) { return sqrt(x); }
GRX_MEM double grx_fabs(double x) { return fabs(x); }
GRX_MEM double grx_fmin(double a, double b) { return fmin(a, b); }
GRX_MEM double grx_fmax(double a, double b) { return fmax(a, b); }
#endif
struct GrxMprPt { MF v[3], w[3]; };
GRX_MEM int grx_mpr_zero(MF x) { return grx_fabs(x) < GRX_MPR_EPS; }
GRX_MEM int grx_mpr_eq(MF a, MF b) {
  MF ab = grx_fabs(a - b);
  if (ab < GRX_MPR_EPS) return 1;
  a = grx_fabs(a); b = grx_fabs(b);
  return ab < GRX_MPR_EPS * (b > a ? b : a);
}
GRX_MEM MF grx_sgn1f(MF x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
GRX_MEM void grx_normalize3f(MF* v) { MF n2 = dot3f(v, v); if (n2 > 0.0f) { MF s = 1.0f / grx_sqrt(n2); v[0] *= s; v[1] *= s; v[2] *= s; } }
// farthest point of the geom along the world direction d, relative to the geom centre
template <typename RF>
GRX_MEM void grx_geom_support(const RF* R, const RF* szf, int type, const MF* d, MF* out) {
  MF dl[3], r[3] = {0.0f, 0.0f, 0.0f};
  const MF sz[3] = {szf[0], szf[1], szf[2]};
  mulMatTVec3f(dl, R, d);
  if (type == 2) { r[0] = dl[0] * sz[0]; r[1] = dl[1] * sz[0]; r[2] = dl[2] * sz[0]; }
  else if (type == 3) { r[0] = dl[0] * sz[0]; r[1] = dl[1] * sz[0]; r[2] = dl[2] * sz[0] + grx_sgn1f(dl[2]) * sz[1]; }
  else if (type == 4) {
    MF t[3] = {dl[0] * sz[0], dl[1] * sz[1], dl[2] * sz[2]};
    grx_normalize3f(t);
    r[0] = t[0] * sz[0]; r[1] = t[1] * sz[1]; r[2] = t[2] * sz[2];
  } else if (type == 5) {
    MF h = grx_sqrt(dl[0] * dl[0] + dl[1] * dl[1]);
    if (h > GRX_MINVAL) { MF ih = sz[0] / h; r[0] = dl[0] * ih; r[1] = dl[1] * ih; }
    r[2] = grx_sgn1f(dl[2]) * sz[1];
  } else if (type == 6) { r[0] = grx_sgn1f(dl[0]) * sz[0]; r[1] = grx_sgn1f(dl[1]) * sz[1]; r[2] = grx_sgn1f(dl[2]) * sz[2]; }
  mulMatVec3f(out, R, r);
}
template <typename RF>   // storage of the two frames and sizes: MF where the caller derived them in MF (grx_geom_frame_mf), float where they are the fp32 values of the kinematics stage (hull pairs: half the registers)
struct GrxMprPairT { RF R1[9], R2[9], s1[3], s2[3]; MF c21[3], hm; int t1, t2;   // the two frames are copied into registers: ~20 support evaluations each read them twice
                    const float *v1, *v2; int n1, n2, lane; const int *aadr1, *anum1, *aadr2, *anum2, *adj;   // hull adjacency (per hull vertex: first neighbour / count into adj)
                                        // hull vertices (geom frame) of mesh geoms: only read by the wave-cooperative variant
                    GrxMprPt* pts;
                    const float *nbr1, *nbr2;   // neighbour records of the two hulls (GrxModel::mesh_nbr + 64 * first hull vertex), or null
                    const int *cell1, *cell2; const float* cellrec;   // support-candidate lists of the two hulls (GrxModel::mesh_cellhdr + 2 * geom_cellbase, mesh_cellrec), or null
                    mutable int hint, hk;       // wave-cooperative variant: lane e holds the guessed support vertices of evaluation e ((v1 + 1) | (v2 + 1) << 16); evaluations so far
#if defined(GRX_PROFILE) && !defined(GRX_EMU)
                    long long* prof;
#endif
                  };                                            // wave-cooperative variant: LDS storage of the five portal points (keeps them out of the VGPR budget)
typedef GrxMprPairT<MF> GrxMprPair;       // lane-per-pair convex routine (primitive pairs)
typedef GrxMprPairT<float> GrxMprPairW;   // wave-cooperative hull pairs
// (Round 4, measured and removed: the scan as a leaf function behind a real call or inline with 16-byte vertex records and 8 - 16 loads in flight per lane -- one memory
// round per hull instead of three -- is 12 % SLOWER on the Fetch launch, profiles/ab_r04_fetch_scan4.txt: the registers it needs are spilled by the substep loop.)
// Convex hull of a mesh: the hull vertex farthest along the (geom-frame) direction dl; the lowest vertex index wins ties, like the oracle's
// exhaustive scan.  Called from wave-uniform code: on the GPU the 64 lanes share the scan (lane l takes the vertices l, l + 64, ...; the
// loads are coalesced) and agree on the winner through two DPP reductions -- a hull of 500 vertices costs 8 loads per lane.
// fp64 support vertex from the fp32 scan's winner: the vertices whose projection is within fp32 rounding of the maximum form a connected cap of the convex
// hull (a face lying flat on a table: all of its vertices tie to ~1e-7), and the reference -- a double precision scan -- picks among them by the digits the
// fp32 products do not have; another pick moves the portal's first vertex and with it the contact normal by 0.1 rad (FetchHullContacts fixture, snapshot 93).
// Hill climbing over the hull's edge graph in MF arithmetic from the fp32 winner reaches the fp64 winner in one or two rounds of neighbour loads; the lowest
// index wins exact ties, like the reference's scan.  aadr / anum: per-vertex adjacency of THIS hull, adj: the model's neighbour table.
GRX_MEM int grx_mesh_support_refine(const float* verts, const int* aadr, const int* anum, const int* adj, const MF* dlm, int cur, int lane_) {
  if (sizeof(TF) == sizeof(HF) || aadr == nullptr) return cur;
  const TF dlt[3] = {(TF)dlm[0], (TF)dlm[1], (TF)dlm[2]};
  for (int guard = 0; guard < 64; guard++) {
    const TF tc = (TF)verts[3 * cur] * dlt[0] + (TF)verts[3 * cur + 1] * dlt[1] + (TF)verts[3 * cur + 2] * dlt[2];
    const int aa = aadr[cur], an = anum[cur];
    TF tb = tc; int nb = cur;
#if defined(GRX_EMU)
    (void)lane_;
    for (int k = 0; k < an; k++) {
      const int v = adj[aa + k];
      const TF t = (TF)verts[3 * v] * dlt[0] + (TF)verts[3 * v + 1] * dlt[1] + (TF)verts[3 * v + 2] * dlt[2];
      if (t > tb || (t == tb && v < nb)) { tb = t; nb = v; }
    }
#else
    for (int k = lane_; k < an; k += 64) {
      const int v = adj[aa + k];
      const TF t = (TF)verts[3 * v] * dlt[0] + (TF)verts[3 * v + 1] * dlt[1] + (TF)verts[3 * v + 2] * dlt[2];
      if (t > tb || (t == tb && v < nb)) { tb = t; nb = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {   // wave argmax in MF (rare path: a few times per portal search)
      const TF to = __shfl_xor(tb, o, 64); const int no = __shfl_xor(nb, o, 64);
      if (to > tb || (to == tb && no < nb)) { tb = to; nb = no; }
    }
#endif
    if (nb == cur) break;
    cur = nb;
  }
  return cur;
}
// hint / nbr: a GUESS of the support vertex (the one the same evaluation of the same pair's portal search found in the previous substep) and the hull's neighbour records.
// The guess is accepted only if its projection exceeds that of every hull neighbour by 1e-6 |d| (metres): a clear local maximum over the hull's edge graph is the unique
// global maximum (convexity), so the exhaustive scan below -- fp32 scan, fp64 decision among the near-ties -- returns the same vertex.  The margin is what makes this
// rigorous on REAL hull tables: qhull's triangulation of the float32-rounded vertices contains near-coplanar facets whose diagonals are "concave" at the 1e-9 m level, so
// a vertex can top all of its listed neighbours by up to 7e-9 m without being the maximum (tests/test_cpu_hull_hints.py measures this on every packaged hull: nothing
// above 1e-7 m over 10^5 face-normal, chord and random directions).  One coalesced fetch of 16 records instead of a scan of the whole hull; anything else (a near-tie, a
// vertex with more than 15 neighbours, a stale guess) falls through to the scan.
GRX_MEM int grx_mesh_support(const float* verts, int n, const MF* dlm, MF* r, int lane_, const int* aadr = nullptr, const int* anum = nullptr, const int* adj = nullptr, int hint = -1,
                             const float* nbr = nullptr, const int* cellhdr = nullptr, const float* cellrec = nullptr) {
  r[0] = r[1] = r[2] = 0.0f;
  if (n <= 0) return -1;
#if !defined(GRX_EMU) && defined(GRX_HULL_HINTS)
  if (hint >= 0 && hint < n && nbr != nullptr) {
    const float4 p = ((const float4*)nbr)[GRX_NBR_RECS * hint + (lane_ & (GRX_NBR_RECS - 1))];
    const int deg = (int)grx_readlane_f(p.w, 0);
    if (deg >= 1) {
      const double t = (double)p.x * (double)dlm[0] + (double)p.y * (double)dlm[1] + (double)p.z * (double)dlm[2];
      const unsigned long long tb = (unsigned long long)__double_as_longlong(t);
      const double t0 = __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(tb >> 32), 0) << 32) | (unsigned)__builtin_amdgcn_readlane((int)tb, 0)));
      const double dn = sqrt((double)dlm[0] * (double)dlm[0] + (double)dlm[1] * (double)dlm[1] + (double)dlm[2] * (double)dlm[2]);
      const bool beaten = lane_ >= 1 && lane_ <= deg && !(t0 - t > 1.0e-6 * dn);
      if (__ballot(beaten) == 0ull) {
        r[0] = grx_readlane_f(p.x, 0); r[1] = grx_readlane_f(p.y, 0); r[2] = grx_readlane_f(p.z, 0);
        return hint;
      }
    }
  }
#else
  (void)hint; (void)nbr;
#endif
  const HF dl[3] = {(HF)dlm[0], (HF)dlm[1], (HF)dlm[2]};   // the scan's own arithmetic type (GRX_HULL_REAL)
#if defined(GRX_EMU)
  HF best = -3.0e38f; int bi = 0;
  for (int v = 0; v < n; v++) { const HF t = verts[3 * v] * dl[0] + verts[3 * v + 1] * dl[1] + verts[3 * v + 2] * dl[2]; if (t > best) { best = t; bi = v; } }
  if (cellhdr) {   // the emulator scans the hull; it CHECKS that the device's candidate list of this direction's cell holds every vertex inside the tie band (what the device reads instead)
    const int cell = grx_hull_cell((float)dl[0], (float)dl[1], (float)dl[2]), off = cellhdr[2 * cell], cnt = cellhdr[2 * cell + 1];
    g_grx_cell_stats[0]++;
    if (cnt > 0) {
      g_grx_cell_stats[1]++; g_grx_cell_stats[2] += cnt;
      const HF near_ = best - 1.0e-6f * fmaxf(1.0f, fabsf(best));
      for (int v = 0; v < n; v++) {
        const HF t = verts[3 * v] * dl[0] + verts[3 * v + 1] * dl[1] + verts[3 * v + 2] * dl[2];
        if (t < near_) continue;
        int found = 0;
        for (int k = 0; k < cnt; k++) {
          const float* rec = cellrec + 4 * (size_t)(off + k); int id; memcpy(&id, rec + 3, 4);
          if (id == v) { found = (rec[0] == verts[3 * v] && rec[1] == verts[3 * v + 1] && rec[2] == verts[3 * v + 2]); break; }
        }
        if (!found) g_grx_cell_stats[3]++;   // a vertex the device would not have seen: must stay 0 (tests/test_cpu_hull_cells.py)
      }
    }
  }
#else
#ifndef GRX_HULL_INFLIGHT
#define GRX_HULL_INFLIGHT 4
#endif
  float best = -3.0e38f, second = -3.0e38f, bx = 0.0f, by = 0.0f, bz = 0.0f; int mine = 0;   // second: this lane's runner-up (is the winner unique beyond fp32 rounding?)
  int listed = 0;
  if (cellhdr) {   // the candidate list of the direction's cell: every vertex that can win or tie is in it (GrxModel::mesh_cellhdr), one record per lane
    const int cell = grx_hull_cell(dl[0], dl[1], dl[2]);
    const int off = __builtin_amdgcn_readfirstlane(cellhdr[2 * cell]), cnt = __builtin_amdgcn_readfirstlane(cellhdr[2 * cell + 1]);
    if (cnt > 0) {
      listed = 1;
      if (lane_ < cnt) {
        const float4 p = ((const float4*)cellrec)[off + lane_];
        best = p.x * dl[0] + p.y * dl[1] + p.z * dl[2]; mine = __float_as_int(p.w); bx = p.x; by = p.y; bz = p.z;
      }
    }
  }
  for (int v0 = lane_; !listed && v0 < n; v0 += 64 * GRX_HULL_INFLIGHT) {   // several independent vertex fetches in flight per lane: one memory latency per 64 * GRX_HULL_INFLIGHT vertices
    float x[GRX_HULL_INFLIGHT], y[GRX_HULL_INFLIGHT], z[GRX_HULL_INFLIGHT];
#pragma unroll
    for (int u = 0; u < GRX_HULL_INFLIGHT; u++) { const int v = v0 + 64 * u < n ? v0 + 64 * u : n - 1; x[u] = verts[3 * v]; y[u] = verts[3 * v + 1]; z[u] = verts[3 * v + 2]; }
#pragma unroll
    for (int u = 0; u < GRX_HULL_INFLIGHT; u++) {
      const float t = x[u] * dl[0] + y[u] * dl[1] + z[u] * dl[2];
#ifdef GRX_NO_SECOND   // (A/B only: misses two tied vertices of one lane)
      if (v0 + 64 * u < n && t > best) { best = t; mine = v0 + 64 * u; bx = x[u]; by = y[u]; bz = z[u]; }
#else
      // the lane's runner-up costs ONE instruction per vertex: the second largest of {best, second, t} is their median (best >= second).  (As a compare + two
      // selects it cost 9 % of the Fetch launch, profiles/ab_r04_fetch_tiebreak.txt: the scan loop is the hot spot of the worlds that end a launch.)
      const float tt = (v0 + 64 * u < n) ? t : -3.0e38f;
      second = __builtin_amdgcn_fmed3f(best, second, tt);
      if (tt > best) { best = tt; mine = v0 + 64 * u; bx = x[u]; by = y[u]; bz = z[u]; }
#endif
    }
  }
  const float mx = grx_reduce_max(best);
  int bi = (int)(-grx_reduce_max((best == mx) ? -(float)mine : -3.0e38f));   // vertex indices are far below 2^24: exact in fp32
  {
    // the winner is unique beyond the rounding of the fp32 products (|t| < 1 m: error < 3e-7) in all but face-on / edge-on directions: no refinement, and its
    // coordinates are in the registers of the lane that scanned it -- no second trip to memory
    const float near_ = mx - 1.0e-6f * fmaxf(1.0f, fabsf(mx));
#ifdef GRX_NO_HULL_REFINE   // (A/B: the fp32 winner as it is)
    const unsigned long long cand = 1ull, cand2 = 0ull;
#else
    const unsigned long long cand = __ballot(best >= near_), cand2 = __ballot(second >= near_);
#endif
    if (sizeof(TF) == sizeof(HF) || aadr == nullptr || (__builtin_popcountll(cand) <= 1 && cand2 == 0ull)) {
      const unsigned long long own = __ballot(best == mx && mine == bi);
      const int src = own ? __builtin_ctzll(own) : 0;
      r[0] = grx_readlane_f(bx, src); r[1] = grx_readlane_f(by, src); r[2] = grx_readlane_f(bz, src);
      return bi;
    }
    if (cand2 == 0ull) {
      // the tied vertices are the winners of different lanes (the common case: a face of a few vertices): their fp64 projections come from the coordinates the
      // lanes still hold -- no further memory traffic -- and one wave argmax picks the reference's vertex (lowest index on an exact tie)
      const bool c_ = best >= near_;
      const double tb = c_ ? (double)bx * (double)dlm[0] + (double)by * (double)dlm[1] + (double)bz * (double)dlm[2] : 0.0;
      // wave maximum of the fp64 projections through their order-preserving 64-bit integer images (DPP butterflies, no LDS round trips; a resting hull face ties
      // with ALL of its vertices -- tens of candidates in every support evaluation of exactly the worlds that end a Fetch launch -- so a scalar walk over the
      // candidate lanes cost 6 % of the launch); candidates get a key >= 1, everything else 0
      const unsigned long long bits = (unsigned long long)__double_as_longlong(tb);
      const unsigned long long key = c_ ? ((bits >> 63) ? ~bits : (bits | 0x8000000000000000ull)) : 0ull;
      const unsigned long long kmax = grx_reduce_max_u64(key);
      const unsigned long long top = __ballot(c_ && key == kmax);
      int src = __builtin_ctzll(top), nb = __builtin_amdgcn_readlane(mine, src);
      if (top & (top - 1ull)) {   // an exact fp64 tie: the lowest vertex index wins (the reference's scan keeps the first maximum)
        unsigned long long mk = top & (top - 1ull);
        while (mk) { const int l = __builtin_ctzll(mk); mk &= mk - 1ull; const int il = __builtin_amdgcn_readlane(mine, l); if (il < nb) { nb = il; src = l; } }
      }
      r[0] = grx_readlane_f(bx, src); r[1] = grx_readlane_f(by, src); r[2] = grx_readlane_f(bz, src);
      return nb;
    }
  }
#endif
  bi = grx_mesh_support_refine(verts, aadr, anum, adj, dlm, bi, lane_);
  r[0] = verts[3 * bi]; r[1] = verts[3 * bi + 1]; r[2] = verts[3 * bi + 2];
  return bi;
}
// The same with a guess: a hull vertex that is not lower than any of its hull neighbours along dl IS the support vertex (convexity), so a
// vertex remembered from the previous substep is verified with one round of neighbour loads instead of a scan of the whole hull.
// Returns the support vertex (hint, or the winner of the full scan).
GRX_MEM int grx_mesh_support_hint(const GrxModel* m, int adr, int n, const MF* dlm, int hint, MF* r, int lane_, const int* cellhdr = nullptr) {
  const float* verts = m->mesh_vert + 3 * adr;
  // The guess is verified in the scan's arithmetic (fp32): this routine only serves the re-check of a cached separating direction, whose test keeps 1e-6 of
  // slack -- ten times what a tie between fp32 projections can hide.  The portal search proper goes through grx_mesh_support (fp64 tie-break).
  const HF dl[3] = {(HF)dlm[0], (HF)dlm[1], (HF)dlm[2]};
  if (hint >= 0 && hint < n) {
    const int aa = m->mesh_adjadr[adr + hint], an = m->mesh_adjnum[adr + hint];
    const HF t0 = verts[3 * hint] * dl[0] + verts[3 * hint + 1] * dl[1] + verts[3 * hint + 2] * dl[2];
#if defined(GRX_EMU)
    int higher = 0;
    for (int k = 0; k < an; k++) { const int nb = m->mesh_adj[aa + k]; higher |= (verts[3 * nb] * dl[0] + verts[3 * nb + 1] * dl[1] + verts[3 * nb + 2] * dl[2] > t0); }
#else
    int hi_ = 0;
    for (int k = lane_; k < an; k += 64) { const int nb = m->mesh_adj[aa + k]; hi_ |= (verts[3 * nb] * dl[0] + verts[3 * nb + 1] * dl[1] + verts[3 * nb + 2] * dl[2] > t0); }
    const int higher = __ballot(hi_ != 0) != 0ull;
#endif
    if (!higher) { r[0] = verts[3 * hint]; r[1] = verts[3 * hint + 1]; r[2] = verts[3 * hint + 2]; return hint; }
  }
  return grx_mesh_support(verts, n, dlm, r, lane_, m->mesh_adjadr + adr, m->mesh_adjnum + adr, m->mesh_adj, -1, nullptr, cellhdr, m->mesh_cellrec);
}
// W: wave-cooperative variant (uniform control flow, every lane holds the same values; mesh geoms allowed)
template <bool W, typename Q>
GRX_MEM void grx_mpr_support(const Q* q, const MF* d, GrxMprPt* o) {
  MF nd[3] = {-d[0], -d[1], -d[2]}, b[3];
#if defined(GRX_PROFILE) && !defined(GRX_EMU)
  if (W && q->lane == 0) { q->prof[16 + 26] += 1; q->prof[16 + 27] += (q->t1 == 7 ? q->n1 : 0) + (q->t2 == 7 ? q->n2 : 0); }
#endif
#if defined(GRX_PROFILE) && !defined(GRX_EMU)
  const long long tp0_ = clock64();
#endif
  int h1 = -1, h2 = -1, f1 = -1, f2 = -1, ek = 0;
#if !defined(GRX_EMU) && defined(GRX_HULL_HINTS)
  if (W) {
    ek = __builtin_amdgcn_readfirstlane(q->hk);
    if (ek < 16) { const int pk = __builtin_amdgcn_readlane(q->hint, ek); h1 = (pk & 0xFFFF) - 1; h2 = (int)((unsigned)pk >> 16) - 1; }
    q->hk = ek + 1;
  }
#endif
  if (W && q->t1 == 7) { MF dl[3], r[3]; mulMatTVec3f(dl, q->R1, d); f1 = grx_mesh_support(q->v1, q->n1, dl, r, q->lane, q->aadr1, q->anum1, q->adj, h1, q->nbr1, q->cell1, q->cellrec); mulMatVec3f(o->w, q->R1, r); }
  else grx_geom_support(q->R1, q->s1, q->t1, d, o->w);
  if (W && q->t2 == 7) { MF dl[3], r[3]; mulMatTVec3f(dl, q->R2, nd); f2 = grx_mesh_support(q->v2, q->n2, dl, r, q->lane, q->aadr2, q->anum2, q->adj, h2, q->nbr2, q->cell2, q->cellrec); mulMatVec3f(b, q->R2, r); }
  else grx_geom_support(q->R2, q->s2, q->t2, nd, b);
#if !defined(GRX_EMU) && defined(GRX_HULL_HINTS)
  // the winners become the guesses of this evaluation in the next substep.  (The guess words live in the world's HBM row, written and read by the lanes of ONE wave without a
  // fence: a stale, torn or foreign word can never change a result, because a guess is only ever a CANDIDATE -- grx_mesh_support accepts it when it provably is the support
  // vertex (tops every hull neighbour by the margin) and scans otherwise; tests/test_gpu_fetch.py::test_hull_caches_do_not_change_the_rollout.)
  if (W && ek < 16 && q->lane == ek) q->hint = ((f1 + 1) & 0xFFFF) | ((f2 + 1) << 16);
#endif
#if defined(GRX_PROFILE) && !defined(GRX_EMU)
  if (W && q->lane == 0) q->prof[16 + 28] += clock64() - tp0_;
#endif
  for (int k = 0; k < 3; k++) { o->w[k] += d[k] * q->hm; o->v[k] = o->w[k] - (b[k] + q->c21[k] - d[k] * q->hm); }
}
// the portal is kept as four separate points (not an array): every access is to a named variable, so the 30 floats stay in registers
GRX_MEM void grx_mpr_portal_dir(const GrxMprPt& P1, const GrxMprPt& P2, const GrxMprPt& P3, MF* dir) {
  MF a[3], b[3];
  for (int k = 0; k < 3; k++) { a[k] = P2.v[k] - P1.v[k]; b[k] = P3.v[k] - P1.v[k]; }
  cross3f(dir, a, b); grx_normalize3f(dir);
}
GRX_MEM int grx_mpr_reach_tolerance(const GrxMprPt& P1, const GrxMprPt& P2, const GrxMprPt& P3, const GrxMprPt& v4, const MF* dir, MF tol) {
  MF d4 = dot3f(v4.v, dir), mn = grx_fmin(d4 - dot3f(P1.v, dir), grx_fmin(d4 - dot3f(P2.v, dir), d4 - dot3f(P3.v, dir)));
  return grx_mpr_eq(mn, tol) || mn < tol;
}
GRX_MEM void grx_mpr_set(GrxMprPt& dst, const GrxMprPt& src, int take) {
  for (int k = 0; k < 3; k++) { dst.v[k] = take ? src.v[k] : dst.v[k]; dst.w[k] = take ? src.w[k] : dst.w[k]; }
}
GRX_MEM void grx_mpr_expand(const GrxMprPt& P0, GrxMprPt& P1, GrxMprPt& P2, GrxMprPt& P3, const GrxMprPt& v4) {
  MF cr[3];
  cross3f(cr, v4.v, P0.v);
  const int s1 = dot3f(P1.v, cr) > 0.0f, s2 = dot3f(P2.v, cr) > 0.0f, s3 = dot3f(P3.v, cr) > 0.0f;
  // s1: (s2 ? P1 : P3) <- v4;   !s1: (s3 ? P2 : P1) <- v4
  const int to1 = (s1 && s2) || (!s1 && !s3), to2 = !s1 && s3, to3 = s1 && !s2;
  grx_mpr_set(P1, v4, to1); grx_mpr_set(P2, v4, to2); grx_mpr_set(P3, v4, to3);
}
GRX_MEM MF grx_mpr_seg_dist2(const MF* a, const MF* b, MF* w) {
  MF d[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, t = -dot3f(a, d), dd = dot3f(d, d);
  t = dd > 0.0f ? grx_fmin((MF)1.0f, grx_fmax((MF)0.0f, t / dd)) : (MF)0.0f;
  for (int k = 0; k < 3; k++) w[k] = a[k] + t * d[k];
  return dot3f(w, w);
}
GRX_MEM MF grx_mpr_tri_dist2(const MF* x0, const MF* b, const MF* cc, MF* w) {
  MF d1[3], d2[3];
  for (int k = 0; k < 3; k++) { d1[k] = b[k] - x0[k]; d2[k] = cc[k] - x0[k]; }
  MF u = dot3f(x0, x0), v = dot3f(d1, d1), ww = dot3f(d2, d2), p = dot3f(x0, d1), q = dot3f(x0, d2), r = dot3f(d1, d2);
  MF den = ww * v - r * r, best;
  if (!grx_mpr_zero(den)) {
    MF sp = (q * r - ww * p) / den, tp = (-sp * r - q) / ww;
    if ((grx_mpr_zero(sp) || sp > 0.0f) && (grx_mpr_eq(sp, 1.0f) || sp < 1.0f) && (grx_mpr_zero(tp) || tp > 0.0f) && (grx_mpr_eq(tp, 1.0f) || tp < 1.0f) &&
        (grx_mpr_eq(tp + sp, 1.0f) || tp + sp < 1.0f)) {
      for (int k = 0; k < 3; k++) w[k] = x0[k] + sp * d1[k] + tp * d2[k];
      // |w|^2, not the expanded quadratic form sp^2 v + tp^2 ww + 2 sp tp r + 2 sp p + 2 tp q + u of the published routine: for a portal whose vertices are
      // decimetres from an origin 0.2 mm off its plane the form's terms are ~0.1 and cancel to 4e-8, which in fp32 is rounding noise -- 25 um of depth at 0.2 mm,
      // measured by tests/test_gpu_anchors.py (a mesh cube standing on a vertex, away from the slab's centre).  The components of w cancel too, but to 1e-4
      // relative.  Same value in exact arithmetic (and in the fp64 oracle).
      best = dot3f(w, w);
      return best;
    }
  }
  MF w2[3], dist;
  best = grx_mpr_seg_dist2(x0, b, w);
  dist = grx_mpr_seg_dist2(x0, cc, w2); if (dist < best) { best = dist; w[0] = w2[0]; w[1] = w2[1]; w[2] = w2[2]; }
  dist = grx_mpr_seg_dist2(b, cc, w2); if (dist < best) { best = dist; w[0] = w2[0]; w[1] = w2[1]; w[2] = w2[2]; }
  return best;
}
// 0 = penetration (depth, dir, pos, surface witnesses w1 on geom 1 / w2 on geom 2 -- all relative to the centre of geom 1), -1 = separated
// sep (may be null): on a -1 return caused by a support point on the far side of the origin (v . d <= 0), sep[0..2] <- that direction d
// and sep[3] <- 1: d separates the two (inflated) geoms, which any later call can re-check with ONE support evaluation (grx_mesh_pairs)
#ifdef GRX_MPR_CALL   // the portal search behind a real call: its (fp64) register appetite stays out of the substep loop's allocation
#define GRX_MPR_FN GRX_MEM_CALL
#else
#define GRX_MPR_FN GRX_MEM
#endif
template <bool W, typename Q>
GRX_MPR_FN int grx_mpr_penetration(const Q* q, MF tol, int maxit, MF* depth, MF* dir, MF* pos, MF* w1, MF* w2, MF* sep = nullptr) {
#define GRX_MPR_SEP(D) do { if (W && sep) { sep[0] = (D)[0]; sep[1] = (D)[1]; sep[2] = (D)[2]; sep[3] = 1.0f; } } while (0)
  // lane-per-pair variant: the portal lives in registers; wave-cooperative variant: in LDS (every lane writes the same values)
  GrxMprPt r0_, r1_, r2_, r3_, r4_;
#ifdef GRX_MPR_PORTAL_REGS
  constexpr bool kLds = false;
#else
  constexpr bool kLds = W;
#endif
  GrxMprPt& P0 = kLds ? q->pts[0] : r0_; GrxMprPt& P1 = kLds ? q->pts[1] : r1_; GrxMprPt& P2 = kLds ? q->pts[2] : r2_; GrxMprPt& P3 = kLds ? q->pts[3] : r3_;
  GrxMprPt& v4 = kLds ? q->pts[4] : r4_;
  MF d[3], a[3], b[3], dotv;
  for (int k = 0; k < 3; k++) { P0.w[k] = 0.0f; P0.v[k] = -q->c21[k]; }
  if (grx_mpr_eq(P0.v[0], 0.0f) && grx_mpr_eq(P0.v[1], 0.0f) && grx_mpr_eq(P0.v[2], 0.0f)) P0.v[0] += GRX_MPR_EPS * 10.0f;
  for (int k = 0; k < 3; k++) d[k] = -P0.v[k];
  grx_normalize3f(d);
  grx_mpr_support<W>(q, d, &P1);
  dotv = dot3f(P1.v, d);
  if (grx_mpr_zero(dotv) || dotv < 0.0f) { GRX_MPR_SEP(d); return -1; }
  cross3f(d, P0.v, P1.v);
  if (grx_mpr_zero(dot3f(d, d))) {
    for (int k = 0; k < 3; k++) { w1[k] = P1.w[k]; w2[k] = P1.w[k] - P1.v[k]; pos[k] = 0.5f * (w1[k] + w2[k]); }
    if (grx_mpr_eq(P1.v[0], 0.0f) && grx_mpr_eq(P1.v[1], 0.0f) && grx_mpr_eq(P1.v[2], 0.0f)) { *depth = 0.0f; dir[0] = dir[1] = dir[2] = 0.0f; return 0; }
    dir[0] = P1.v[0]; dir[1] = P1.v[1]; dir[2] = P1.v[2]; *depth = grx_sqrt(dot3f(dir, dir)); grx_normalize3f(dir);
    return 0;
  }
  grx_normalize3f(d);
  grx_mpr_support<W>(q, d, &P2);
  dotv = dot3f(P2.v, d);
  if (grx_mpr_zero(dotv) || dotv < 0.0f) { GRX_MPR_SEP(d); return -1; }
  for (int k = 0; k < 3; k++) { a[k] = P1.v[k] - P0.v[k]; b[k] = P2.v[k] - P0.v[k]; }
  cross3f(d, a, b); grx_normalize3f(d);
  if (dot3f(d, P0.v) > 0.0f) { GrxMprPt t = P1; P1 = P2; P2 = t; d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2]; }
  for (int guard = 0;; guard++) {
    if (guard > 200) return -1;
    grx_mpr_support<W>(q, d, &P3);
    dotv = dot3f(P3.v, d);
    if (grx_mpr_zero(dotv) || dotv < 0.0f) { GRX_MPR_SEP(d); return -1; }
    int cont = 0;
    cross3f(a, P1.v, P3.v); dotv = dot3f(a, P0.v);
    if (dotv < 0.0f && !grx_mpr_zero(dotv)) { P2 = P3; cont = 1; }
    if (!cont) {
      cross3f(a, P3.v, P2.v); dotv = dot3f(a, P0.v);
      if (dotv < 0.0f && !grx_mpr_zero(dotv)) { P1 = P3; cont = 1; }
    }
    if (!cont) break;
    for (int k = 0; k < 3; k++) { a[k] = P1.v[k] - P0.v[k]; b[k] = P2.v[k] - P0.v[k]; }
    cross3f(d, a, b); grx_normalize3f(d);
  }
  for (int guard = 0;; guard++) {
    if (guard > 200) return -1;
    grx_mpr_portal_dir(P1, P2, P3, d);
    dotv = dot3f(d, P1.v);
    if (grx_mpr_zero(dotv) || dotv > 0.0f) break;
    grx_mpr_support<W>(q, d, &v4);
    dotv = dot3f(v4.v, d);
    if (!(grx_mpr_zero(dotv) || dotv > 0.0f)) { GRX_MPR_SEP(d); return -1; }
    if (grx_mpr_reach_tolerance(P1, P2, P3, v4, d, tol)) return -1;
    grx_mpr_expand(P0, P1, P2, P3, v4);
  }
  for (int it = 0;; it++) {
    grx_mpr_portal_dir(P1, P2, P3, d);
    grx_mpr_support<W>(q, d, &v4);
#if defined(GRX_EMU) && defined(GRX_MPR_STATS)
    if (W) { g_grx_mesh_stats[2]++; if (it > maxit) g_grx_mesh_stats[3]++; }
#endif
    if (grx_mpr_reach_tolerance(P1, P2, P3, v4, d, tol) || it > maxit) {
      MF w[3];
      *depth = grx_sqrt(grx_mpr_tri_dist2(P1.v, P2.v, P3.v, w));
      if (grx_mpr_zero(w[0]) && grx_mpr_zero(w[1]) && grx_mpr_zero(w[2])) { w[0] = d[0]; w[1] = d[1]; w[2] = d[2]; }
      grx_normalize3f(w); dir[0] = w[0]; dir[1] = w[1]; dir[2] = w[2];
      MF bc[4], cr[3], sum;
      cross3f(cr, P1.v, P2.v); bc[0] = dot3f(cr, P3.v);
      cross3f(cr, P3.v, P2.v); bc[1] = dot3f(cr, P0.v);
      cross3f(cr, P0.v, P1.v); bc[2] = dot3f(cr, P3.v);
      cross3f(cr, P2.v, P1.v); bc[3] = dot3f(cr, P0.v);
      sum = bc[0] + bc[1] + bc[2] + bc[3];
      if (grx_mpr_zero(sum) || sum < 0.0f) {
        bc[0] = 0.0f;
        cross3f(cr, P2.v, P3.v); bc[1] = dot3f(cr, d);
        cross3f(cr, P3.v, P1.v); bc[2] = dot3f(cr, d);
        cross3f(cr, P1.v, P2.v); bc[3] = dot3f(cr, d);
        sum = bc[1] + bc[2] + bc[3];
      }
      // witness on geom 2 = w - v (+ the centre offset, which cancels in the relative frame except for P0: its witnesses are the two centres)
      const MF is = 1.0f / sum;
      for (int k = 0; k < 3; k++) {
        MF p1 = 0.0f, p2 = bc[0] * q->c21[k];
        p1 += bc[1] * P1.w[k] + bc[2] * P2.w[k] + bc[3] * P3.w[k];
        p2 += bc[1] * (P1.w[k] - P1.v[k]) + bc[2] * (P2.w[k] - P2.v[k]) + bc[3] * (P3.w[k] - P3.v[k]);
        pos[k] = 0.5f * (p1 + p2) * is;
      }
      // surface witnesses: the foot of the origin on the portal plane in barycentric coordinates of the triangle alone
      cross3f(cr, P2.v, P3.v); bc[1] = dot3f(cr, d);
      cross3f(cr, P3.v, P1.v); bc[2] = dot3f(cr, d);
      cross3f(cr, P1.v, P2.v); bc[3] = dot3f(cr, d);
      const MF it3 = 1.0f / (bc[1] + bc[2] + bc[3]);
      for (int k = 0; k < 3; k++) {
        w1[k] = (bc[1] * P1.w[k] + bc[2] * P2.w[k] + bc[3] * P3.w[k]) * it3;
        w2[k] = (bc[1] * (P1.w[k] - P1.v[k]) + bc[2] * (P2.w[k] - P2.v[k]) + bc[3] * (P3.w[k] - P3.v[k])) * it3;
      }
      return 0;
    }
    grx_mpr_expand(P0, P1, P2, P3, v4);
  }
}
#undef GRX_MPR_SEP
// analytic outward normal of a smooth geom (sphere, capsule, ellipsoid) at the world point p (see the oracle: the portal direction of a
// shallow contact is ill-conditioned, MuJoCo replaces it for smooth geoms); returns 0 for the other types
template <typename RF>
GRX_MEM int grx_smooth_normal(const RF* R, const MF* ce, const RF* szf, int type, const MF* p, MF* n) {
  const MF sz[3] = {szf[0], szf[1], szf[2]};
  MF d[3] = {p[0] - ce[0], p[1] - ce[1], p[2] - ce[2]}, loc[3], nl[3];
  mulMatTVec3f(loc, R, d);
  if (type == 2) { nl[0] = loc[0]; nl[1] = loc[1]; nl[2] = loc[2]; }
  else if (type == 3) { nl[0] = loc[0]; nl[1] = loc[1]; nl[2] = loc[2] > sz[1] ? loc[2] - sz[1] : (loc[2] < -sz[1] ? loc[2] + sz[1] : 0.0f); }
  else if (type == 4) { nl[0] = loc[0] / (sz[0] * sz[0]); nl[1] = loc[1] / (sz[1] * sz[1]); nl[2] = loc[2] / (sz[2] * sz[2]); }
  else return 0;
  const MF l2 = dot3f(nl, nl);
  if (l2 < 1e-30f) return 0;
  const MF il = 1.0f / grx_sqrt(l2);
  nl[0] *= il; nl[1] *= il; nl[2] *= il;
  mulMatVec3f(n, R, nl);
  return 1;
}
// Frame of geom g for the convex routine, in MF.  A geom of a FREE ROOT body (a free joint directly under the world: the manipulated objects) gets its frame
// straight from the world's qpos in MF arithmetic -- normalised quaternion -> body frame -> geom frame, the oracle's operation order -- instead of the fp32 frames of
// the kinematics stage: an object lying flat on a table is a line / face contact whose single contact point is decided by a tilt of ~1e-6 rad, which the ~1e-7
// rounding of the fp32 frames moves by centimetres (tools/emu_mixed.py: the kinematics stage was the only fp32 stage the AdroitHammer fixtures noticed).
GRX_MEM void grx_quat2mat_mf(MF* X, const MF* q) {
  const MF w = q[0], x = q[1], y = q[2], z = q[3];
  X[0] = w * w + x * x - y * y - z * z; X[1] = 2 * (x * y - w * z); X[2] = 2 * (x * z + w * y);
  X[3] = 2 * (x * y + w * z); X[4] = w * w - x * x + y * y - z * z; X[5] = 2 * (y * z - w * x);
  X[6] = 2 * (x * z - w * y); X[7] = 2 * (y * z + w * x); X[8] = w * w - x * x - y * y + z * z;
}
GRX_MEM void grx_geom_frame_mf(const GrxModel* m, const GrxCtx* c, int g, MF* R, MF* pos) {
  const int b = m->geom_bodyid[g];
#ifndef GRX_NO_FREE_FRAMES
  // root bodies (children of the world that are not mocap bodies and not members of a shift group): the oracle's kinematics of ONE body, in MF
  // (joint types this routine restates: free 0, slide 2, hinge 3.  A BALL joint -- type 1 -- on a root body is not restated: such a body keeps the fp32 frame of the kinematics
  // stage, which is also what the Jacobians of its contacts are built from; no packaged model has one, compile_mjcf is a general compiler)
  int supported = b > 0 && m->body_parent[b] == 0 && m->body_mocapid[b] < 0 && !(S::kShift && m->nshift && (m->geom_shift[g] || m->body_shift[b]));
  if (supported) { const int jn0 = m->body_jntnum[b], ja0 = m->body_jntadr[b]; for (int kk = 0; kk < jn0; kk++) { const int ty = m->jnt_type[ja0 + kk]; if (ty != 0 && ty != 2 && ty != 3) supported = 0; } }
  if (supported) {
    const int jn = m->body_jntnum[b], ja = m->body_jntadr[b];
    MF p[3], q[4];
    if (jn == 1 && m->jnt_type[ja] == 0) {
      const int qa = m->jnt_qposadr[ja];
      for (int k = 0; k < 3; k++) p[k] = c->qpos[qa + k];
      for (int k = 0; k < 4; k++) q[k] = c->qpos[qa + 3 + k];
    } else {
      for (int k = 0; k < 3; k++) p[k] = m->body_pos[3 * b + k];
      for (int k = 0; k < 4; k++) q[k] = m->body_quat[4 * b + k];
      for (int kk = 0; kk < jn; kk++) {
        const int j = ja + kk;
        MF Rq[9]; grx_quat2mat_mf(Rq, q);
        const MF jp[3] = {m->jnt_pos[3 * j], m->jnt_pos[3 * j + 1], m->jnt_pos[3 * j + 2]}, jx[3] = {m->jnt_axis[3 * j], m->jnt_axis[3 * j + 1], m->jnt_axis[3 * j + 2]};
        MF anchor[3], axis[3];
        mulMatVec3f(anchor, Rq, jp); anchor[0] += p[0]; anchor[1] += p[1]; anchor[2] += p[2];
        mulMatVec3f(axis, Rq, jx);
        const MF dq = (MF)c->qpos[m->jnt_qposadr[j]] - (MF)m->qpos0[m->jnt_qposadr[j]];
        if (m->jnt_type[j] == 2) { p[0] += axis[0] * dq; p[1] += axis[1] * dq; p[2] += axis[2] * dq; }
        else if (m->jnt_type[j] == 3) {
          const MF sn = sin(0.5 * (double)dq), cs = cos(0.5 * (double)dq);
          const MF qr[4] = {cs, jx[0] * sn, jx[1] * sn, jx[2] * sn};
          const MF qn[4] = {q[0] * qr[0] - q[1] * qr[1] - q[2] * qr[2] - q[3] * qr[3], q[0] * qr[1] + q[1] * qr[0] + q[2] * qr[3] - q[3] * qr[2],
                            q[0] * qr[2] - q[1] * qr[3] + q[2] * qr[0] + q[3] * qr[1], q[0] * qr[3] + q[1] * qr[2] - q[2] * qr[1] + q[3] * qr[0]};
          for (int k = 0; k < 4; k++) q[k] = qn[k];
          MF Rn[9], off[3]; grx_quat2mat_mf(Rn, q); mulMatVec3f(off, Rn, jp);
          p[0] = anchor[0] - off[0]; p[1] = anchor[1] - off[1]; p[2] = anchor[2] - off[2];
        }
      }
    }
    const MF n = grx_sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n > 1e-12f) { const MF r = 1.0f / n; q[0] *= r; q[1] *= r; q[2] *= r; q[3] *= r; }
    MF X[9], L[9];
    grx_quat2mat_mf(X, q);
    const MF lq[4] = {m->geom_quat[4 * g], m->geom_quat[4 * g + 1], m->geom_quat[4 * g + 2], m->geom_quat[4 * g + 3]};
    grx_quat2mat_mf(L, lq);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) R[3 * i + j] = X[3 * i] * L[j] + X[3 * i + 1] * L[3 + j] + X[3 * i + 2] * L[6 + j];
    const MF lp[3] = {m->geom_pos[3 * g], m->geom_pos[3 * g + 1], m->geom_pos[3 * g + 2]};
    for (int i = 0; i < 3; i++) pos[i] = p[i] + (X[3 * i] * lp[0] + X[3 * i + 1] * lp[1] + X[3 * i + 2] * lp[2]);
    return;
  }
#endif
  for (int k = 0; k < 9; k++) R[k] = c->gxmat[9 * g + k];
  for (int k = 0; k < 3; k++) pos[k] = c->gxpos[3 * g + k];
}
GRX_MEM void grx_convex_pair(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, int t1, int t2, float margin) {
  GrxMprPair q;
  MF p1[3], p2[3];
  grx_geom_frame_mf(m, c, g1, q.R1, p1); grx_geom_frame_mf(m, c, g2, q.R2, p2);
  q.t1 = t1; q.t2 = t2; q.hm = 0.5f * margin;
  for (int k = 0; k < 3; k++) { q.s1[k] = m->geom_size[3 * g1 + k]; q.s2[k] = m->geom_size[3 * g2 + k]; q.c21[k] = p2[k] - p1[k]; }
  MF depth, dir[3], pos[3], w1[3], w2[3];
  q.v1 = q.v2 = nullptr; q.n1 = q.n2 = 0; q.lane = 0; q.pts = nullptr; q.aadr1 = q.anum1 = q.aadr2 = q.anum2 = q.adj = nullptr; q.nbr1 = q.nbr2 = nullptr; q.hint = q.hk = 0; q.cell1 = q.cell2 = nullptr; q.cellrec = nullptr;
  if (grx_mpr_penetration<false>(&q, m->mpr_tolerance, m->mpr_iterations, &depth, dir, pos, w1, w2) != 0) return;
#if defined(GRX_EMU) && defined(GRX_EMU_TRACE)
  if (getenv("GRX_TRACE_MPR")) {
    fprintf(stderr, "MPR pair %d g %d %d t %d %d\n R1", pair, g1, g2, t1, t2);
    for (int k = 0; k < 9; k++) fprintf(stderr, " %.17g", (double)q.R1[k]);
    fprintf(stderr, "\n R2"); for (int k = 0; k < 9; k++) fprintf(stderr, " %.17g", (double)q.R2[k]);
    fprintf(stderr, "\n c21 %.17g %.17g %.17g s1 %.9g %.9g %.9g s2 %.9g %.9g %.9g hm %.9g\n depth %.17g dir %.17g %.17g %.17g pos %.17g %.17g %.17g\n", (double)q.c21[0], (double)q.c21[1], (double)q.c21[2],
            (double)q.s1[0], (double)q.s1[1], (double)q.s1[2], (double)q.s2[0], (double)q.s2[1], (double)q.s2[2], (double)q.hm, (double)depth, (double)dir[0], (double)dir[1], (double)dir[2], (double)pos[0], (double)pos[1], (double)pos[2]);
  }
#endif
  if (dir[0] == 0.0f && dir[1] == 0.0f && dir[2] == 0.0f) return;
  // still relative to the centre of geom 1: the smooth normals are taken in that frame as well (the world offset only enters the stored contact position)
  const MF ce1[3] = {0.0f, 0.0f, 0.0f};
  MF n1[3] = {0.0f, 0.0f, 0.0f}, n2[3] = {0.0f, 0.0f, 0.0f};
  const int h1 = grx_smooth_normal(q.R1, ce1, q.s1, t1, pos, n1), h2 = grx_smooth_normal(q.R2, q.c21, q.s2, t2, pos, n2);
  if (h1 || h2) {
    MF n[3] = {n1[0] - n2[0], n1[1] - n2[1], n1[2] - n2[2]};
    const MF l2 = dot3f(n, n);
    if (l2 > 1e-30f) {
      const MF il = 1.0f / grx_sqrt(l2); dir[0] = n[0] * il; dir[1] = n[1] * il; dir[2] = n[2] * il;
      // penetration along the corrected normal: extreme point of a smooth geom, portal witness of a box / cylinder (see the oracle)
      MF nd[3] = {-dir[0], -dir[1], -dir[2]};
      if (h1) { grx_geom_support(q.R1, q.s1, t1, dir, w1); for (int k = 0; k < 3; k++) w1[k] += dir[k] * q.hm; }
      if (h2) { grx_geom_support(q.R2, q.s2, t2, nd, w2); for (int k = 0; k < 3; k++) w2[k] += q.c21[k] - dir[k] * q.hm; }
      depth = (w1[0] - w2[0]) * dir[0] + (w1[1] - w2[1]) * dir[1] + (w1[2] - w2[2]) * dir[2];
    }
  }
  const float posw[3] = {(float)(pos[0] + p1[0]), (float)(pos[1] + p1[1]), (float)(pos[2] + p1[2])}, dirf[3] = {(float)dir[0], (float)dir[1], (float)dir[2]};
  grx_add_contact(c, pair, posw, dirf, (float)(margin - depth));
}
// ------------------------------------------------------------------------------------------
// Hull-vs-convex pairs (the convex hull of a mesh against a primitive or another hull: the Fetch arm / gripper / base links, assets/fetch/
// robot.xml:16-93).  MuJoCo sends them through the same general convex routine as the ellipsoid / cylinder pairs; here the pair is
// handled by the WHOLE wavefront: the portal refinement runs in wave-uniform control flow (every lane holds the same values) and the hull
// support function is a cooperative scan over the vertices (grx_mesh_support), because a lane-private walk over a hull in global memory
// is a chain of dependent loads (~25 us per support point).  Candidates are rare -- one persistent pair per Fetch world passes the
// bounding-box filter, a contact exists in 0.04 % of the substeps -- and a separating direction found by one substep is kept for the next
// ones (c->meshcache): re-checking it costs ONE support evaluation instead of the six or seven of a fresh portal search, and a direction
// that still separates the two inflated geoms proves that the routine would report "no contact".
// ------------------------------------------------------------------------------------------
// separating-axis test of the two geoms' oriented bounding boxes (geom_aabb), each grown by margin / 2 (the oracle's obb_overlap)
// (written out with named scalars: an array indexed by a loop variable would live in scratch memory)
GRX_MEM int grx_obb_overlap(const GrxModel* m, const GrxCtx* c, int g1, int g2, float margin) {
  const float* R1 = c->gxmat + 9 * g1; const float* R2 = c->gxmat + 9 * g2; const float* a1 = m->geom_aabb + 6 * g1; const float* a2 = m->geom_aabb + 6 * g2;
  const float hm = 0.5f * margin;
  const float a10 = a1[0], a11 = a1[1], a12 = a1[2], e10 = a1[3] + hm, e11 = a1[4] + hm, e12 = a1[5] + hm;
  const float a20 = a2[0], a21 = a2[1], a22 = a2[2], e20 = a2[3] + hm, e21 = a2[4] + hm, e22 = a2[5] + hm;
  const float r100 = R1[0], r101 = R1[1], r102 = R1[2], r110 = R1[3], r111 = R1[4], r112 = R1[5], r120 = R1[6], r121 = R1[7], r122 = R1[8];
  const float r200 = R2[0], r201 = R2[1], r202 = R2[2], r210 = R2[3], r211 = R2[4], r212 = R2[5], r220 = R2[6], r221 = R2[7], r222 = R2[8];
  // centre offset in world coordinates, then in the frames of box 1 (ta) and box 2 (tb)
  const float tx = (c->gxpos[3 * g2] + r200 * a20 + r201 * a21 + r202 * a22) - (c->gxpos[3 * g1] + r100 * a10 + r101 * a11 + r102 * a12);
  const float ty = (c->gxpos[3 * g2 + 1] + r210 * a20 + r211 * a21 + r212 * a22) - (c->gxpos[3 * g1 + 1] + r110 * a10 + r111 * a11 + r112 * a12);
  const float tz = (c->gxpos[3 * g2 + 2] + r220 * a20 + r221 * a21 + r222 * a22) - (c->gxpos[3 * g1 + 2] + r120 * a10 + r121 * a11 + r122 * a12);
  const float ta0 = tx * r100 + ty * r110 + tz * r120, ta1 = tx * r101 + ty * r111 + tz * r121, ta2 = tx * r102 + ty * r112 + tz * r122;
  const float tb0 = tx * r200 + ty * r210 + tz * r220, tb1 = tx * r201 + ty * r211 + tz * r221, tb2 = tx * r202 + ty * r212 + tz * r222;
  // C_ij = A_i . B_j (columns of the two frames)
#define GRX_OBB_C(i, j) const float C##i##j = r10##i * r20##j + r11##i * r21##j + r12##i * r22##j, Q##i##j = fabsf(C##i##j);
  GRX_OBB_C(0, 0) GRX_OBB_C(0, 1) GRX_OBB_C(0, 2) GRX_OBB_C(1, 0) GRX_OBB_C(1, 1) GRX_OBB_C(1, 2) GRX_OBB_C(2, 0) GRX_OBB_C(2, 1) GRX_OBB_C(2, 2)
#undef GRX_OBB_C
  if (fabsf(ta0) > e10 + e20 * Q00 + e21 * Q01 + e22 * Q02) return 0;
  if (fabsf(ta1) > e11 + e20 * Q10 + e21 * Q11 + e22 * Q12) return 0;
  if (fabsf(ta2) > e12 + e20 * Q20 + e21 * Q21 + e22 * Q22) return 0;
  if (fabsf(tb0) > e20 + e10 * Q00 + e11 * Q10 + e12 * Q20) return 0;
  if (fabsf(tb1) > e21 + e10 * Q01 + e11 * Q11 + e12 * Q21) return 0;
  if (fabsf(tb2) > e22 + e10 * Q02 + e11 * Q12 + e12 * Q22) return 0;
  // axis A_i x B_j (unnormalised on both sides of the test; nearly parallel edges are left to the face axes)
#define GRX_OBB_EDGE(i, i1, i2, j, j1, j2) \
  if (1.0f - C##i##j * C##i##j >= 1e-6f) { \
    const float tp_ = fabsf(ta##i2 * C##i1##j - ta##i1 * C##i2##j); \
    const float ra_ = e1##i1 * Q##i2##j + e1##i2 * Q##i1##j, rb_ = e2##j1 * Q##i##j2 + e2##j2 * Q##i##j1; \
    if (tp_ > (ra_ + rb_) * 1.0001f + 1e-7f) return 0; }
  GRX_OBB_EDGE(0, 1, 2, 0, 1, 2) GRX_OBB_EDGE(0, 1, 2, 1, 2, 0) GRX_OBB_EDGE(0, 1, 2, 2, 0, 1)
  GRX_OBB_EDGE(1, 2, 0, 0, 1, 2) GRX_OBB_EDGE(1, 2, 0, 1, 2, 0) GRX_OBB_EDGE(1, 2, 0, 2, 0, 1)
  GRX_OBB_EDGE(2, 0, 1, 0, 1, 2) GRX_OBB_EDGE(2, 0, 1, 1, 2, 0) GRX_OBB_EDGE(2, 0, 1, 2, 0, 1)
#undef GRX_OBB_EDGE
  return 1;
}

// Joint-box gate of a hull pair (mjcf/pair_gates.py): the two bodies are separated by at most three hinge / slide joints, and for joint values inside the
// gate's box the compiler has PROVEN the two margin-inflated geoms disjoint (rigorous distance bound on a grid + a Lipschitz bound in between).  1 = inside
// the box: the pair cannot produce a contact in this configuration and leaves the candidate sweep -- the Fetch arm's torso / shoulder pair, 1.9 cm apart in
// every pose the tasks reach, no longer walks through the bounding-box filter and the hull routine in every substep of every world.
GRX_MEM int grx_gate_clear(const GrxModel* m, const GrxCtx* c, int gi) {
  const int* qa = m->gate_qadr + 3 * gi; const float* bx = m->gate_box + 6 * gi;
  int ok = 1;
  for (int k = 0; k < 3; k++) { const int a = qa[k]; if (a >= 0) { const float q = c->qpos[a]; ok &= (q > bx[2 * k]) & (q < bx[2 * k + 1]); } }
  return ok;
}
// the queued hull-vs-convex pairs of this pass, one after the other, all lanes on each (wave-uniform code)
GRX_MEM void grx_mesh_pairs(const GrxModel* m, GrxCtx* c, const int* queue, int nq, int lane_) {
  for (int e = 0; e < nq; e++) {
    const int pair = queue[e], g1 = m->pair_geom1[pair], g2 = m->pair_geom2[pair];
    const float margin = m->pair_margin[pair];
    GrxMprPairW q;
    for (int k = 0; k < 9; k++) { q.R1[k] = c->gxmat[9 * g1 + k]; q.R2[k] = c->gxmat[9 * g2 + k]; }
    q.t1 = m->geom_type[g1]; q.t2 = m->geom_type[g2]; q.hm = 0.5f * margin; q.lane = lane_;
    for (int k = 0; k < 3; k++) { q.s1[k] = m->geom_size[3 * g1 + k]; q.s2[k] = m->geom_size[3 * g2 + k]; q.c21[k] = (MF)c->gxpos[3 * g2 + k] - (MF)c->gxpos[3 * g1 + k]; }
    q.v1 = q.t1 == 7 ? m->mesh_vert + 3 * m->geom_hulladr[g1] : m->mesh_vert; q.n1 = q.t1 == 7 ? m->geom_hullnum[g1] : 0;
    q.v2 = q.t2 == 7 ? m->mesh_vert + 3 * m->geom_hulladr[g2] : m->mesh_vert; q.n2 = q.t2 == 7 ? m->geom_hullnum[g2] : 0;
    q.aadr1 = m->mesh_adjadr + (q.t1 == 7 ? m->geom_hulladr[g1] : 0); q.anum1 = m->mesh_adjnum + (q.t1 == 7 ? m->geom_hulladr[g1] : 0);
    q.aadr2 = m->mesh_adjadr + (q.t2 == 7 ? m->geom_hulladr[g2] : 0); q.anum2 = m->mesh_adjnum + (q.t2 == 7 ? m->geom_hulladr[g2] : 0); q.adj = m->mesh_adj;
    q.pts = (GrxMprPt*)(c->Jp + 192);
    q.nbr1 = (q.t1 == 7 && m->mesh_nbr) ? m->mesh_nbr + (size_t)4 * GRX_NBR_RECS * m->geom_hulladr[g1] : nullptr;
    q.nbr2 = (q.t2 == 7 && m->mesh_nbr) ? m->mesh_nbr + (size_t)4 * GRX_NBR_RECS * m->geom_hulladr[g2] : nullptr;
    q.hint = 0; q.hk = 0;
    q.cell1 = (q.t1 == 7 && m->mesh_cellhdr && m->geom_cellbase[g1] >= 0) ? m->mesh_cellhdr + 2 * (size_t)m->geom_cellbase[g1] : nullptr;
    q.cell2 = (q.t2 == 7 && m->mesh_cellhdr && m->geom_cellbase[g2] >= 0) ? m->mesh_cellhdr + 2 * (size_t)m->geom_cellbase[g2] : nullptr;
    q.cellrec = m->mesh_cellrec;
#if defined(GRX_PROFILE) && !defined(GRX_EMU)
    q.prof = c->prof;
#endif   // 30 words behind the pair queue: the Jacobian pool is free until the constraint stage
    GRX_SUBTICK(c, 21);   // pair set-up
    GRX_COUNT(c, 24, 1);
    // a direction kept from an earlier substep: still separating?  (entry: pair + 1, direction, (v1 + 1) + 4096 (v2 + 1) = the support vertices)
    float* mc = c->meshcache;
    const float key = (float)(pair + 1);
    const int slot = mc[0] == key ? 0 : (mc[5] == key ? 1 : (mc[10] == key ? 2 : (mc[15] == key ? 3 : -1)));
    if (slot >= 0) {
      const MF d[3] = {mc[5 * slot + 1], mc[5 * slot + 2], mc[5 * slot + 3]}, nd[3] = {-d[0], -d[1], -d[2]};
      const int hints = (int)mc[5 * slot + 4];
      int h1 = (hints & 4095) - 1, h2 = (hints >> 12) - 1;
      MF sw[3], sb[3], dl[3], r[3];
      if (q.t1 == 7) { mulMatTVec3f(dl, q.R1, d); h1 = grx_mesh_support_hint(m, m->geom_hulladr[g1], q.n1, dl, h1, r, lane_, q.cell1); mulMatVec3f(sw, q.R1, r); }
      else grx_geom_support(q.R1, q.s1, q.t1, d, sw);
      if (q.t2 == 7) { mulMatTVec3f(dl, q.R2, nd); h2 = grx_mesh_support_hint(m, m->geom_hulladr[g2], q.n2, dl, h2, r, lane_, q.cell2); mulMatVec3f(sb, q.R2, r); }
      else grx_geom_support(q.R2, q.s2, q.t2, nd, sb);
      MF sv = 0.0f;   // v . d of the Minkowski support point (see grx_mpr_support)
      for (int k = 0; k < 3; k++) sv += ((sw[k] + d[k] * q.hm) - (sb[k] + q.c21[k] - d[k] * q.hm)) * d[k];
      if (sv < -1e-6f) {   // strictly on the far side: the (inflated) geoms are disjoint
        LANE0 { mc[5 * slot + 4] = (float)((h1 + 1) + 4096 * (h2 + 1)); }
#if defined(GRX_EMU)
        g_grx_mesh_stats[0]++;
#endif
        GRX_SUBTICK(c, 22);   // cached separating direction re-checked: disjoint
        continue;
      }
    }
    GRX_SUBTICK(c, 22);
    GRX_COUNT(c, 25, 1);
#if defined(GRX_EMU)
    g_grx_mesh_stats[1]++;
#endif
    // guesses of the support vertices, one word per evaluation of this pair's search (the world's HBM row, 4 blocks of key + 16 words): a pair in persistent contact -- the
    // upper arm resting on the head link, the worlds that end a Fetch launch -- repeats its search substep after substep with almost the same directions
    int hblk = -1;
#if !defined(GRX_EMU) && defined(GRX_HULL_HINTS)
    if (c->hullhint) {
      const float hk0 = c->hullhint[0], hk1 = c->hullhint[17], hk2 = c->hullhint[34], hk3 = c->hullhint[51];
      hblk = hk0 == key ? 0 : (hk1 == key ? 1 : (hk2 == key ? 2 : (hk3 == key ? 3 : -1)));
      hblk = __builtin_amdgcn_readfirstlane(hblk);
      if (hblk >= 0 && lane_ < 16) q.hint = __float_as_int(c->hullhint[17 * hblk + 1 + lane_]);
    }
#endif
    MF depth, dir[3], pos[3], w1[3], w2[3], sep[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int rc = grx_mpr_penetration<true>(&q, m->mpr_tolerance, m->mpr_iterations, &depth, dir, pos, w1, w2, sep);
#if !defined(GRX_EMU) && defined(GRX_HULL_HINTS)
    if (c->hullhint && rc == 0) {   // in contact: this search will run again in the next substep
      int wblk = hblk;
      if (wblk < 0) { wblk = ((int)c->hullhint[68]) & 3; if (lane_ == 0) c->hullhint[68] = (float)((wblk + 1) & 3); }
      wblk = __builtin_amdgcn_readfirstlane(wblk);
      if (lane_ < 16) c->hullhint[17 * wblk + 1 + lane_] = __int_as_float(lane_ < q.hk ? q.hint : 0);
      if (lane_ == 0) c->hullhint[17 * wblk] = key;
    }
#endif
    GRX_SUBTICK(c, 23);   // portal search
#ifdef GRX_PROBE_HULL   // outcome of the searches (tools/hull_outcome_probe.py): contacts, separations with a direction, the pair searched last
    GRX_COUNT(c, 35, rc == 0 ? 1 : 0); GRX_COUNT(c, 36, (rc != 0 && sep[3] != 0.0f) ? 1 : 0); GRX_PMAX(c, 37, pair);
#endif
#if defined(GRX_EMU) && defined(GRX_MESH_DEBUG)
    fprintf(stderr, "meshpair %d (g %d %d) slot %d rc %d sep %g\n", pair, g1, g2, slot, rc, (double)sep[3]);
#endif
    WAVE_SYNC();
    if (rc != 0) {
      if (sep[3] != 0.0f) {   // keep the direction for the next substeps
        const int w = slot >= 0 ? slot : ((int)mc[20] & 3);   // the pair's own slot, else round robin over the four
        LANE0 { mc[5 * w] = key; mc[5 * w + 1] = sep[0]; mc[5 * w + 2] = sep[1]; mc[5 * w + 3] = sep[2]; mc[5 * w + 4] = 0.0f; if (slot < 0) mc[20] = (MF)((w + 1) & 3); }
      }
      WAVE_SYNC();
      continue;
    }
    if (slot >= 0) { LANE0 { mc[5 * slot] = 0.0f; } WAVE_SYNC(); }   // the pair is in contact: its old direction is useless, do not re-check it (two support evaluations) before every search of the next substeps
    if (dir[0] == 0.0f && dir[1] == 0.0f && dir[2] == 0.0f) continue;
    const MF ce1[3] = {0.0f, 0.0f, 0.0f};
    MF n1[3] = {0.0f, 0.0f, 0.0f}, n2[3] = {0.0f, 0.0f, 0.0f};
    const int h1 = grx_smooth_normal(q.R1, ce1, q.s1, q.t1, pos, n1), h2 = grx_smooth_normal(q.R2, q.c21, q.s2, q.t2, pos, n2);
    if (h1 || h2) {   // a smooth primitive against the hull: analytic normal, depth along it (see grx_convex_pair)
      MF n[3] = {n1[0] - n2[0], n1[1] - n2[1], n1[2] - n2[2]};
      const MF l2 = dot3f(n, n);
      if (l2 > 1e-30f) {
        const MF il = 1.0f / grx_sqrt(l2); dir[0] = n[0] * il; dir[1] = n[1] * il; dir[2] = n[2] * il;
        MF nd[3] = {-dir[0], -dir[1], -dir[2]};
        if (h1) { grx_geom_support(q.R1, q.s1, q.t1, dir, w1); for (int k = 0; k < 3; k++) w1[k] += dir[k] * q.hm; }
        if (h2) { grx_geom_support(q.R2, q.s2, q.t2, nd, w2); for (int k = 0; k < 3; k++) w2[k] += q.c21[k] - dir[k] * q.hm; }
        depth = (w1[0] - w2[0]) * dir[0] + (w1[1] - w2[1]) * dir[1] + (w1[2] - w2[2]) * dir[2];
      }
    }
    const float posw[3] = {(float)(pos[0] + c->gxpos[3 * g1]), (float)(pos[1] + c->gxpos[3 * g1 + 1]), (float)(pos[2] + c->gxpos[3 * g1 + 2])}, dirf[3] = {(float)dir[0], (float)dir[1], (float)dir[2]};
    LANE0 { grx_add_contact(c, pair, posw, dirf, (float)(margin - depth)); }
    WAVE_SYNC();
  }
}
// plane vs cylinder: near-cap rim point, far-cap rim point, two more corners of a triangle inscribed in the near rim (see the oracle)
GRX_MEM void grx_plane_cylinder(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, float margin) {
  const float* pm = c->gxmat + 9 * g1; const float* cm = c->gxmat + 9 * g2; const float* cp = c->gxpos + 3 * g2;
  const float r = m->geom_size[3 * g2], hl = m->geom_size[3 * g2 + 1];
  float n[3] = {pm[2], pm[5], pm[8]}, ax[3] = {cm[2], cm[5], cm[8]};
  float prjaxis = dot3f(n, ax);
  if (prjaxis > 0.0f) { ax[0] = -ax[0]; ax[1] = -ax[1]; ax[2] = -ax[2]; prjaxis = -prjaxis; }
  float dd[3] = {cp[0] - c->gxpos[3 * g1], cp[1] - c->gxpos[3 * g1 + 1], cp[2] - c->gxpos[3 * g1 + 2]};
  const float dist0 = dot3f(dd, n);
  float vec[3] = {ax[0] * prjaxis - n[0], ax[1] * prjaxis - n[1], ax[2] * prjaxis - n[2]};
  const float len2 = dot3f(vec, vec);
  if (len2 >= 1e-30f) { const float sc = r / sqrtf(len2); vec[0] *= sc; vec[1] *= sc; vec[2] *= sc; }
  else { vec[0] = cm[0] * r; vec[1] = cm[3] * r; vec[2] = cm[6] * r; }
  const float prjvec = dot3f(vec, n);
  ax[0] *= hl; ax[1] *= hl; ax[2] *= hl; prjaxis *= hl;
  float dist = dist0 + prjaxis + prjvec, pos[3];
  if (dist > margin) return;
  for (int k = 0; k < 3; k++) pos[k] = cp[k] + vec[k] + ax[k] - n[k] * dist * 0.5f;
  grx_add_contact(c, pair, pos, n, dist);
  dist = dist0 - prjaxis + prjvec;
  if (dist <= margin) {
    for (int k = 0; k < 3; k++) pos[k] = cp[k] + vec[k] - ax[k] - n[k] * dist * 0.5f;
    grx_add_contact(c, pair, pos, n, dist);
  }
  dist = dist0 + prjaxis - 0.5f * prjvec;
  if (dist <= margin) {
    float v1[3];
    cross3f(v1, vec, ax);
    const float l2 = dot3f(v1, v1);
    if (l2 > 0.0f) { const float sc = r * 0.8660254f / sqrtf(l2); v1[0] *= sc; v1[1] *= sc; v1[2] *= sc; }
    for (int sg = 0; sg < 2; sg++) {
      const float sgn = sg ? -1.0f : 1.0f;
      for (int k = 0; k < 3; k++) pos[k] = cp[k] + sgn * v1[k] + ax[k] - 0.5f * vec[k] - n[k] * dist * 0.5f;
      grx_add_contact(c, pair, pos, n, dist);
    }
  }
}
GRX_MEM void grx_plane_ellipsoid(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, float margin) {
  float n[3] = {c->gxmat[9 * g1 + 2], c->gxmat[9 * g1 + 5], c->gxmat[9 * g1 + 8]}, p[3];
  {
    const MF sz[3] = {m->geom_size[3 * g2], m->geom_size[3 * g2 + 1], m->geom_size[3 * g2 + 2]}, nd[3] = {-n[0], -n[1], -n[2]};
    MF R2[9], pm[3];
    for (int k = 0; k < 9; k++) R2[k] = c->gxmat[9 * g2 + k];
    grx_geom_support(R2, sz, 4, nd, pm);
    p[0] = (float)pm[0]; p[1] = (float)pm[1]; p[2] = (float)pm[2];
  }
  float dd[3];
  for (int k = 0; k < 3; k++) { p[k] += c->gxpos[3 * g2 + k]; dd[k] = p[k] - c->gxpos[3 * g1 + k]; }
  const float dist = dot3f(dd, n);
  if (dist > margin) return;
  float pos[3] = {p[0] - 0.5f * dist * n[0], p[1] - 0.5f * dist * n[1], p[2] - 0.5f * dist * n[2]};
  grx_add_contact(c, pair, pos, n, dist);
}
// capsule (geom1) vs box (geom2): axis point closest to the box (golden-section search, the distance is convex along the
// axis) as a sphere contact, plus the farther end sphere when it is inside the margin as well (see oracle/grx_oracle.c)
GRX_MEM void grx_capsule_box(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, float margin) {
  const float* ce = c->gxpos + 3 * g1; const float* R = c->gxmat + 9 * g1;
  const float* bp = c->gxpos + 3 * g2; const float* bm = c->gxmat + 9 * g2; const float* sz = m->geom_size + 3 * g2;
  const float r = m->geom_size[3 * g1], hl = m->geom_size[3 * g1 + 1], s0 = sz[0], s1 = sz[1], s2 = sz[2];
  float axw[3] = {R[2], R[5], R[8]}, dw[3] = {ce[0] - bp[0], ce[1] - bp[1], ce[2] - bp[2]}, cen[3], ax[3];
  mulMatTVec3f(cen, bm, dw); mulMatTVec3f(ax, bm, axw);
  // Axis point closest to the box: g(t) = dist^2(box, cen + t ax) is convex and piecewise quadratic, so g'(t)/2 = sum_k ax_k *
  // (p_k - clamp(p_k, -s_k, s_k)) is nondecreasing and piecewise linear with breakpoints where a coordinate crosses a face plane.
  // Evaluate g' at the two ends and the six breakpoints, bracket the sign change between neighbouring candidates, interpolate
  // linearly: the exact minimiser in ~10 evaluations (the oracle finds the same point by golden-section search).
#define GRX_CB_DG(T, OUT) { const float t_ = (T), p0_ = cen[0] + t_ * ax[0], p1_ = cen[1] + t_ * ax[1], p2_ = cen[2] + t_ * ax[2]; \
    OUT = ax[0] * (p0_ - fminf(s0, fmaxf(-s0, p0_))) + ax[1] * (p1_ - fminf(s1, fmaxf(-s1, p1_))) + ax[2] * (p2_ - fminf(s2, fmaxf(-s2, p2_))); }
  float ts, dlo, dhi;
  GRX_CB_DG(-hl, dlo) GRX_CB_DG(hl, dhi)
  if (dlo >= 0.0f) ts = -hl;
  else if (dhi <= 0.0f) ts = hl;
  else {
    float tlo = -hl, thi = hl;   // invariant: g'(tlo) = dlo <= 0 <= dhi = g'(thi)
#define GRX_CB_TRY(TB) { const float tb_ = (TB); if (tb_ > tlo && tb_ < thi) { float d_; GRX_CB_DG(tb_, d_) if (d_ <= 0.0f) { tlo = tb_; dlo = d_; } else { thi = tb_; dhi = d_; } } }
#define GRX_CB_AXIS(K, SK) if (fabsf(ax[K]) > 1e-12f) { const float ia_ = 1.0f / ax[K]; GRX_CB_TRY((SK - cen[K]) * ia_) GRX_CB_TRY((-SK - cen[K]) * ia_) }
    GRX_CB_AXIS(0, s0) GRX_CB_AXIS(1, s1) GRX_CB_AXIS(2, s2)
#undef GRX_CB_AXIS
#undef GRX_CB_TRY
    const float den = dhi - dlo;
    ts = den > 0.0f ? tlo - dlo * (thi - tlo) / den : 0.5f * (tlo + thi);
  }
#undef GRX_CB_DG
  {   // the axis segment passes through the box (penetration deeper than the radius): g vanishes on the whole inside stretch; take its middle
    float ta = -hl, tb = hl; int hit = 1;
#define GRX_CB_SLAB(K, SK) if (fabsf(ax[K]) < GRX_MINVAL) { if (fabsf(cen[K]) > SK) hit = 0; } else { float u_ = (-SK - cen[K]) / ax[K], v_ = (SK - cen[K]) / ax[K]; \
      if (u_ > v_) { const float w_ = u_; u_ = v_; v_ = w_; } ta = fmaxf(ta, u_); tb = fminf(tb, v_); }
    GRX_CB_SLAB(0, s0) GRX_CB_SLAB(1, s1) GRX_CB_SLAB(2, s2)
#undef GRX_CB_SLAB
    if (hit && ta < tb) ts = 0.5f * (ta + tb);
  }
  float ps[3] = {cen[0] + ts * ax[0], cen[1] + ts * ax[1], cen[2] + ts * ax[2]};
  if (!grx_sphere_box_local(c, pair, bp, bm, s0, s1, s2, ps, r, margin)) return;
  float te = (ts >= 0) ? -hl : hl;
  if (fabsf(te - ts) > 0.2f * hl) {
    float pf[3] = {cen[0] + te * ax[0], cen[1] + te * ax[1], cen[2] + te * ax[2]};
    grx_sphere_box_local(c, pair, bp, bm, s0, s1, s2, pf, r, margin);
  }
}

// plane vs a SMALL convex vertex set (<= 32 hull vertices, e.g. the compile-time pruned hulls): one lane does it all
GRX_MEM void grx_plane_mesh_small(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, float margin) {
  const float* gm = c->gxmat + 9 * g2;
  int adr = m->geom_meshadr[g2], num = m->geom_meshnum[g2];
  float n[3] = {c->gxmat[9 * g1 + 2], c->gxmat[9 * g1 + 5], c->gxmat[9 * g1 + 8]}, nl[3];
  mulMatTVec3f(nl, gm, n);
  float off = dot3f(c->gxpos + 3 * g2, n) - dot3f(c->gxpos + 3 * g1, n);
  float bd = 1e30f; int best = -1;
  // the vertex tables live in global memory: fetch four vertices per round with independent loads (one latency per round, not per vertex)
  for (int v0 = 0; v0 < num; v0 += 4) {
    float vx[4], vy[4], vz[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int v = (v0 + u < num) ? v0 + u : num - 1;
      const float* mv = m->mesh_vert + 3 * (adr + v);
      vx[u] = mv[0]; vy[u] = mv[1]; vz[u] = mv[2];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const float dd = vx[u] * nl[0] + vy[u] * nl[1] + vz[u] * nl[2] + off;
      if (v0 + u < num && dd < bd) { bd = dd; best = v0 + u; }
    }
  }
  if (best < 0 || bd > margin) return;
  // the deepest vertex, then its hull neighbours inside the margin (at most 4 contacts): neighbour indices and their vertices in two rounds
  const int aa = m->mesh_adjadr[adr + best], an = m->mesh_adjnum[adr + best];
  int nb[8]; float wx[8], wy[8], wz[8];
#pragma unroll
  for (int u = 0; u < 8; u++) nb[u] = (u < an) ? m->mesh_adj[aa + u] : best;
#pragma unroll
  for (int u = 0; u < 8; u++) { const float* mv = m->mesh_vert + 3 * (adr + nb[u]); wx[u] = mv[0]; wy[u] = mv[1]; wz[u] = mv[2]; }
  int cn = 0;
  // fully unrolled over the fetched neighbours (static register indices); hull vertices of higher degree take the tail loop
#define GRX_PM_EMIT(LX, LY, LZ, IS_BEST) { \
    const float lx_ = (LX), ly_ = (LY), lz_ = (LZ); \
    const float dd = lx_ * nl[0] + ly_ * nl[1] + lz_ * nl[2] + off; \
    if ((IS_BEST) || dd <= margin) { \
      const float lv[3] = {lx_, ly_, lz_}; float w[3], pos[3]; \
      mulMatVec3f(w, gm, lv); \
      for (int t = 0; t < 3; t++) pos[t] = w[t] + c->gxpos[3 * g2 + t] - 0.5f * dd * n[t]; \
      grx_add_contact(c, pair, pos, n, dd); cn++; \
    } }
  { const float* mv = m->mesh_vert + 3 * (adr + best); GRX_PM_EMIT(mv[0], mv[1], mv[2], 1) }
#define GRX_PM_NB(U) if ((U) < an && cn < 4) GRX_PM_EMIT(wx[U], wy[U], wz[U], 0)
  GRX_PM_NB(0) GRX_PM_NB(1) GRX_PM_NB(2) GRX_PM_NB(3) GRX_PM_NB(4) GRX_PM_NB(5) GRX_PM_NB(6) GRX_PM_NB(7)
#undef GRX_PM_NB
  for (int e = 8; e < an && cn < 4; e++) { const float* mv = m->mesh_vert + 3 * (adr + m->mesh_adj[aa + e]); GRX_PM_EMIT(mv[0], mv[1], mv[2], 0) }
#undef GRX_PM_EMIT
}


// box-box: SAT over the 15 axes, then face clipping or edge-edge (the contact set Sutherland-Hodgman clipping yields:
// (a) incident-face corners inside the reference rectangle, (b) reference corners inside the incident quad, (c) proper
// crossings of incident edges with the rectangle sides; at most 8, in that order).
// Eight lanes work on one pair: lane t evaluates the axes t and t+8, then the contact candidates t, t+8 and t+16; the
// winners are found with DPP reductions inside the octet and the surviving candidates are compacted, in candidate
// order, with wave ballots.  Up to eight pairs per pass; the pair queue is filled by grx_collision.
// Everything stays in registers (no dynamically indexed local arrays): axes are selected with GRX_SEL3.
#define GRX_BB_LOAD(PAIR) \
  const int g1 = m->pair_geom1[PAIR], g2 = m->pair_geom2[PAIR]; const float margin = m->pair_margin[PAIR]; \
  const float* p1 = c->gxpos + 3 * g1; const float* R1 = c->gxmat + 9 * g1; const float* p2 = c->gxpos + 3 * g2; const float* R2 = c->gxmat + 9 * g2; \
  const float A0[3] = {R1[0], R1[3], R1[6]}, A1[3] = {R1[1], R1[4], R1[7]}, A2[3] = {R1[2], R1[5], R1[8]}; \
  const float B0[3] = {R2[0], R2[3], R2[6]}, B1[3] = {R2[1], R2[4], R2[7]}, B2[3] = {R2[2], R2[5], R2[8]}; \
  const float a0 = m->geom_size[3 * g1], a1 = m->geom_size[3 * g1 + 1], a2 = m->geom_size[3 * g1 + 2]; \
  const float b0 = m->geom_size[3 * g2], b1 = m->geom_size[3 * g2 + 1], b2 = m->geom_size[3 * g2 + 2]; \
  const float d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
// axis T (0-2: faces of box 1, 3-5: faces of box 2, 6-14: edge i of box 1 x edge j of box 2): unit axis, projection of d, separation
#define GRX_BB_AXIS(T, AXV, TP, SEP, OK) { \
  const int T_ = (T), iu_ = T_ < 3 ? T_ : (T_ < 6 ? 0 : (T_ - 6) / 3), jv_ = T_ < 3 ? 0 : (T_ < 6 ? T_ - 3 : (T_ - 6) % 3); \
  float u_[3], v_[3], x_[3]; \
  for (int e_ = 0; e_ < 3; e_++) { u_[e_] = GRX_SEL3(A0[e_], A1[e_], A2[e_], iu_); v_[e_] = GRX_SEL3(B0[e_], B1[e_], B2[e_], jv_); } \
  cross3f(x_, u_, v_); \
  const float l_ = sqrtf(dot3f(x_, x_)), li_ = 1.0f / fmaxf(l_, 1e-12f); \
  OK = (T_ < 15) && ((T_ < 6) || (l_ >= 1e-6f)); \
  for (int e_ = 0; e_ < 3; e_++) AXV[e_] = T_ < 3 ? u_[e_] : (T_ < 6 ? v_[e_] : x_[e_] * li_); \
  TP = dot3f(d, AXV); \
  float ra_ = a0 * fabsf(dot3f(A0, AXV)) + a1 * fabsf(dot3f(A1, AXV)) + a2 * fabsf(dot3f(A2, AXV)); \
  float rb_ = b0 * fabsf(dot3f(B0, AXV)) + b1 * fabsf(dot3f(B1, AXV)) + b2 * fabsf(dot3f(B2, AXV)); \
  if (T_ < 3) ra_ = GRX_SEL3(a0, a1, a2, T_); else if (T_ < 6) rb_ = GRX_SEL3(b0, b1, b2, T_ - 3); \
  SEP = fabsf(TP) - (ra_ + rb_); }

GRX_MEM void grx_box_box_queue(const GrxModel* m, GrxCtx* c, const int* queue, int nq, int lane_) {
  for (int pb = 0; pb < nq; pb += 8) {
    // ---- separating axes
    GRX_LANEVAR(sf); GRX_LANEVAR(se); GRX_LANEVAR(sall); GRX_LANEVAR_I(cf); GRX_LANEVAR_I(ce);
    FOR_LANES {
      const int g = lane >> 3, t = lane & 7;
      float f = -1e30f, e = -1e30f; int fi = 99, ei = 99;
      if (pb + g < nq) {
        const int pair = queue[pb + g];
        GRX_BB_LOAD(pair)
        float ax[3], tp, sep; int ok;
        GRX_BB_AXIS(t, ax, tp, sep, ok)
        if (ok) { if (t < 6) { f = sep; fi = t; } else { e = sep; ei = t; } }
        GRX_BB_AXIS(t + 8, ax, tp, sep, ok)
        if (ok && sep > e) { e = sep; ei = t + 8; }
        (void)tp; (void)margin;
      }
      LV(sf) = f; LV(se) = e; LV(cf) = fi; LV(ce) = ei; LV(sall) = fmaxf(f, e);
    }
    GRX_LANEVAR(bestf); GRX_LANEVAR(beste); GRX_LANEVAR(maxall); GRX_LANEVAR_I(codef); GRX_LANEVAR_I(codee);
    GRX_OCT_MAX(sf, bestf); GRX_OCT_MAX(se, beste); GRX_OCT_MAX(sall, maxall);
    FOR_LANES { if (!(LV(sf) == LV(bestf))) LV(cf) = 99; if (!(LV(se) == LV(beste))) LV(ce) = 99; }
    GRX_OCT_MIN_I(cf, codef); GRX_OCT_MIN_I(ce, codee);
    // ---- contact candidates
    GRX_LANEVAR(nx); GRX_LANEVAR(ny); GRX_LANEVAR(nz);
    GRX_LANEVAR(cpx0); GRX_LANEVAR(cpy0); GRX_LANEVAR(cpz0); GRX_LANEVAR(ch0); GRX_LANEVAR_I(cv0);
    GRX_LANEVAR(cpx1); GRX_LANEVAR(cpy1); GRX_LANEVAR(cpz1); GRX_LANEVAR(ch1); GRX_LANEVAR_I(cv1);
    GRX_LANEVAR(cpx2); GRX_LANEVAR(cpy2); GRX_LANEVAR(cpz2); GRX_LANEVAR(ch2); GRX_LANEVAR_I(cv2);
    FOR_LANES {
      const int g = lane >> 3, t = lane & 7;
      int v0 = 0, v1 = 0, v2 = 0; float P0[3] = {0, 0, 0}, P1[3] = {0, 0, 0}, P2[3] = {0, 0, 0}, h0 = 0, h1 = 0, h2 = 0, nrm[3] = {0, 0, 0};
      if (pb + g < nq) {
        const int pair = queue[pb + g];
        GRX_BB_LOAD(pair)
        const float best = LV(bestf), ebest = LV(beste); const int code = LV(codef), ecode = LV(codee);
        if (LV(maxall) <= margin && code < 6) {
          if (ecode < 15 && ebest > best + 1e-7f + 0.02f * fabsf(best)) {
            // edge-edge: a single contact, lane 0 of the octet
            const int ei = (ecode - 6) / 3, ej = (ecode - 6) % 3;
            float en[3], tp, sep; int ok;
            GRX_BB_AXIS(ecode, en, tp, sep, ok)
            (void)sep; (void)ok;
            const float sg = tp < 0 ? -1.0f : 1.0f;
            en[0] *= sg; en[1] *= sg; en[2] *= sg;
            float pa[3] = {p1[0], p1[1], p1[2]}, pb_[3] = {p2[0], p2[1], p2[2]};
            float s0 = (ei != 0) ? (dot3f(en, A0) > 0 ? a0 : -a0) : 0.0f, s1 = (ei != 1) ? (dot3f(en, A1) > 0 ? a1 : -a1) : 0.0f, s2 = (ei != 2) ? (dot3f(en, A2) > 0 ? a2 : -a2) : 0.0f;
            float t0 = (ej != 0) ? (dot3f(en, B0) > 0 ? -b0 : b0) : 0.0f, t1 = (ej != 1) ? (dot3f(en, B1) > 0 ? -b1 : b1) : 0.0f, t2 = (ej != 2) ? (dot3f(en, B2) > 0 ? -b2 : b2) : 0.0f;
            float u[3], v[3];
            for (int e = 0; e < 3; e++) {
              pa[e] += s0 * A0[e] + s1 * A1[e] + s2 * A2[e]; pb_[e] += t0 * B0[e] + t1 * B1[e] + t2 * B2[e];
              u[e] = GRX_SEL3(A0[e], A1[e], A2[e], ei); v[e] = GRX_SEL3(B0[e], B1[e], B2[e], ej);
            }
            float w[3] = {pa[0] - pb_[0], pa[1] - pb_[1], pa[2] - pb_[2]};
            float uv = dot3f(u, v), uw = dot3f(u, w), vw = dot3f(v, w);
            float den = 1.0f - uv * uv;
            float sc = den > 1e-12f ? (uv * vw - uw) / den : 0.0f, tc = den > 1e-12f ? (vw - uv * uw) / den : 0.0f;
            for (int k = 0; k < 3; k++) { P0[k] = 0.5f * ((pa[k] + sc * u[k]) + (pb_[k] + tc * v[k])); nrm[k] = en[k]; }
            h0 = ebest; v0 = (t == 0);
          } else {
            // ---- face contact
            float bn[3], tp, sep; int ok;
            GRX_BB_AXIS(code, bn, tp, sep, ok)
            (void)sep; (void)ok;
            { const float sg = tp < 0 ? -1.0f : 1.0f; bn[0] *= sg; bn[1] *= sg; bn[2] *= sg; }
            const int ref1 = code < 3, ax = ref1 ? code : code - 3;
            float pr[3], pi[3], nr[3], Ar0[3], Ar1[3], Ar2[3], Ai0[3], Ai1[3], Ai2[3];
            for (int e = 0; e < 3; e++) {
              pr[e] = ref1 ? p1[e] : p2[e]; pi[e] = ref1 ? p2[e] : p1[e]; nr[e] = ref1 ? bn[e] : -bn[e];
              Ar0[e] = ref1 ? A0[e] : B0[e]; Ar1[e] = ref1 ? A1[e] : B1[e]; Ar2[e] = ref1 ? A2[e] : B2[e];
              Ai0[e] = ref1 ? B0[e] : A0[e]; Ai1[e] = ref1 ? B1[e] : A1[e]; Ai2[e] = ref1 ? B2[e] : A2[e];
            }
            const float sr0 = ref1 ? a0 : b0, sr1 = ref1 ? a1 : b1, sr2 = ref1 ? a2 : b2;
            const float si0 = ref1 ? b0 : a0, si1 = ref1 ? b1 : a1, si2 = ref1 ? b2 : a2;
            // incident face: the face of the other box most anti-parallel to nr
            float dd0 = dot3f(Ai0, nr), dd1 = dot3f(Ai1, nr), dd2 = dot3f(Ai2, nr);
            int iax = 0; float mind = 1e30f, isg = 1.0f;
            if (dd0 < mind) { mind = dd0; iax = 0; isg = 1.0f; } if (-dd0 < mind) { mind = -dd0; iax = 0; isg = -1.0f; }
            if (dd1 < mind) { mind = dd1; iax = 1; isg = 1.0f; } if (-dd1 < mind) { mind = -dd1; iax = 1; isg = -1.0f; }
            if (dd2 < mind) { mind = dd2; iax = 2; isg = 1.0f; } if (-dd2 < mind) { mind = -dd2; iax = 2; isg = -1.0f; }
            float Iu[3], Iv[3], In[3], Ru[3], Rv[3];
            for (int e = 0; e < 3; e++) {
              In[e] = GRX_SEL3(Ai0[e], Ai1[e], Ai2[e], iax); Iu[e] = GRX_SEL3(Ai1[e], Ai2[e], Ai0[e], iax); Iv[e] = GRX_SEL3(Ai2[e], Ai0[e], Ai1[e], iax);
              Ru[e] = GRX_SEL3(Ar1[e], Ar2[e], Ar0[e], ax); Rv[e] = GRX_SEL3(Ar2[e], Ar0[e], Ar1[e], ax);
            }
            const float sin_ = GRX_SEL3(si0, si1, si2, iax), siu = GRX_SEL3(si1, si2, si0, iax), siv = GRX_SEL3(si2, si0, si1, iax);
            const float srn = GRX_SEL3(sr0, sr1, sr2, ax), sx = GRX_SEL3(sr1, sr2, sr0, ax), sy = GRX_SEL3(sr2, sr0, sr1, ax);
            float rc[3], fcw[3];
            for (int e = 0; e < 3; e++) { rc[e] = pr[e] + srn * nr[e]; fcw[e] = pi[e] + isg * sin_ * In[e] - rc[e]; }
            // incident quad in the reference face frame: corner q = centre + su*U + sv*V, (su,sv) = (+,+),(-,+),(-,-),(+,-)
            const float cx = dot3f(fcw, Ru), cy = dot3f(fcw, Rv), chh = dot3f(fcw, nr);
            const float ux = siu * dot3f(Iu, Ru), uy = siu * dot3f(Iu, Rv), uh = siu * dot3f(Iu, nr);
            const float vx = siv * dot3f(Iv, Ru), vy = siv * dot3f(Iv, Rv), vh = siv * dot3f(Iv, nr);
            const float qx0 = cx + ux + vx, qy0 = cy + uy + vy, qh0 = chh + uh + vh;
            const float qx1 = cx - ux + vx, qy1 = cy - uy + vy, qh1 = chh - uh + vh;
            const float qx2 = cx - ux - vx, qy2 = cy - uy - vy;
            const float qx3 = cx + ux - vx, qy3 = cy + uy - vy, qh3 = chh + uh - vh;
            // height field of the incident plane over the reference frame
            const float x1 = qx1 - qx0, y1 = qy1 - qy0, x2 = qx3 - qx0, y2 = qy3 - qy0, hh1 = qh1 - qh0, hh2 = qh3 - qh0;
            const float det = x1 * y2 - x2 * y1;
            const int flat = !(fabsf(det) > 1e-14f);
            const float gu = flat ? 0.0f : (hh1 * y2 - hh2 * y1) / det, gv = flat ? 0.0f : (x1 * hh2 - x2 * hh1) / det;
            const float orient = det > 0 ? -1.0f : 1.0f;  // det > 0 <=> q0->q1->q2->q3 is counter-clockwise <=> interior has cross > 0
#define GRX_SEL4(v0_, v1_, v2_, v3_, i_) ((i_) == 0 ? (v0_) : ((i_) == 1 ? (v1_) : ((i_) == 2 ? (v2_) : (v3_))))
#define GRX_SIDE(PX, PY, AX_, AY_, BX_, BY_) (orient * (((BX_) - (AX_)) * ((PY) - (AY_)) - ((BY_) - (AY_)) * ((PX) - (AX_))))
            // candidate I: 0-3 incident corners inside the rectangle (inclusive); 4-7 rectangle corners strictly inside the
            // incident quad; 8-23 proper crossings of incident edge e = (I-8)/4 with rectangle side (I-8)%4 = +x, -x, +y, -y
            // (x-sides closed in y, y-sides open in x)
#define GRX_BB_CAND(I, VALID, POS, H) { \
              const int i_ = (I); int ok_ = 0; float X_ = 0, Y_ = 0; \
              if (i_ < 4) { X_ = GRX_SEL4(qx0, qx1, qx2, qx3, i_); Y_ = GRX_SEL4(qy0, qy1, qy2, qy3, i_); ok_ = fabsf(X_) <= sx && fabsf(Y_) <= sy; } \
              else if (!flat && i_ < 8) { \
                const int k_ = i_ - 4; X_ = (k_ == 0 || k_ == 3) ? sx : -sx; Y_ = (k_ < 2) ? sy : -sy; \
                ok_ = GRX_SIDE(X_, Y_, qx0, qy0, qx1, qy1) < 0 && GRX_SIDE(X_, Y_, qx1, qy1, qx2, qy2) < 0 && GRX_SIDE(X_, Y_, qx2, qy2, qx3, qy3) < 0 && \
                      GRX_SIDE(X_, Y_, qx3, qy3, qx0, qy0) < 0; \
              } else if (!flat && i_ < 24) { \
                const int e_ = (i_ - 8) >> 2, s_ = (i_ - 8) & 3, e1_ = (e_ + 1) & 3; \
                const float ax_ = GRX_SEL4(qx0, qx1, qx2, qx3, e_), ay_ = GRX_SEL4(qy0, qy1, qy2, qy3, e_); \
                const float bx_ = GRX_SEL4(qx0, qx1, qx2, qx3, e1_), by_ = GRX_SEL4(qy0, qy1, qy2, qy3, e1_); \
                const int hz_ = s_ < 2; const float sg_ = (s_ & 1) ? -1.0f : 1.0f; \
                const float pa_ = hz_ ? ax_ : ay_, pb2_ = hz_ ? bx_ : by_, lim_ = hz_ ? sx : sy; \
                const float da_ = sg_ * pa_ - lim_, db_ = sg_ * pb2_ - lim_; \
                const int cr_ = (da_ < 0 && db_ > 0) || (da_ > 0 && db_ < 0); \
                const float t_ = da_ / (da_ - db_), oa_ = hz_ ? ay_ : ax_, ob_ = hz_ ? by_ : bx_, o_ = oa_ + t_ * (ob_ - oa_); \
                if (hz_) { ok_ = cr_ && fabsf(o_) <= sy; X_ = sg_ * sx; Y_ = o_; } else { ok_ = cr_ && fabsf(o_) < sx; X_ = o_; Y_ = sg_ * sy; } \
              } \
              const float h_ = qh0 + gu * (X_ - qx0) + gv * (Y_ - qy0); \
              VALID = ok_ && h_ <= margin; H = h_; \
              for (int k_ = 0; k_ < 3; k_++) POS[k_] = rc[k_] + X_ * Ru[k_] + Y_ * Rv[k_] + 0.5f * h_ * nr[k_]; }
            GRX_BB_CAND(t, v0, P0, h0)
            GRX_BB_CAND(t + 8, v1, P1, h1)
            GRX_BB_CAND(t + 16, v2, P2, h2)
#undef GRX_BB_CAND
#undef GRX_SIDE
#undef GRX_SEL4
            for (int k = 0; k < 3; k++) nrm[k] = bn[k];
          }
        }
      }
      LV(nx) = nrm[0]; LV(ny) = nrm[1]; LV(nz) = nrm[2];
      LV(cpx0) = P0[0]; LV(cpy0) = P0[1]; LV(cpz0) = P0[2]; LV(ch0) = h0; LV(cv0) = v0;
      LV(cpx1) = P1[0]; LV(cpy1) = P1[1]; LV(cpz1) = P1[2]; LV(ch1) = h1; LV(cv1) = v1;
      LV(cpx2) = P2[0]; LV(cpy2) = P2[1]; LV(cpz2) = P2[2]; LV(ch2) = h2; LV(cv2) = v2;
    }
    // ---- ordered compaction: candidate order inside a pair, pair order across the octets, at most 8 contacts per pair
    const unsigned long long m0 = GRX_BALLOT(cv0), m1 = GRX_BALLOT(cv1), m2 = GRX_BALLOT(cv2);
    WAVE_SYNC();
    const int base = c->cnt[0];
    int total = 0;
    for (int g = 0; g < 8; g++) { int n = __builtin_popcountll((m0 >> (8 * g)) & 0xFFull) + __builtin_popcountll((m1 >> (8 * g)) & 0xFFull) + __builtin_popcountll((m2 >> (8 * g)) & 0xFFull); total += n < 8 ? n : 8; }
    FOR_LANES {
      const int g = lane >> 3, t = lane & 7;
      if (pb + g < nq) {
        const int pair = queue[pb + g];
        int gbase = base;
        for (int q = 0; q < g; q++) { int n = __builtin_popcountll((m0 >> (8 * q)) & 0xFFull) + __builtin_popcountll((m1 >> (8 * q)) & 0xFFull) + __builtin_popcountll((m2 >> (8 * q)) & 0xFFull); gbase += n < 8 ? n : 8; }
        const unsigned b0_ = (unsigned)((m0 >> (8 * g)) & 0xFFull), b1_ = (unsigned)((m1 >> (8 * g)) & 0xFFull), b2_ = (unsigned)((m2 >> (8 * g)) & 0xFFull), low = (1u << t) - 1u;
        const int r0 = __builtin_popcount(b0_ & low), r1 = __builtin_popcount(b0_) + __builtin_popcount(b1_ & low), r2 = __builtin_popcount(b0_) + __builtin_popcount(b1_) + __builtin_popcount(b2_ & low);
        const float nrm[3] = {LV(nx), LV(ny), LV(nz)};
#define GRX_BB_WRITE(V, R, PX, PY, PZ, H) if ((V) && (R) < 8) { const int slot = gbase + (R); \
          if (slot >= c->maxcon) c->cnt[2] |= GRX_ST_CON_OVERFLOW; \
          else { c->con_dist[slot] = (H); c->con_pair[slot] = pair; c->con_pos[3 * slot] = (PX); c->con_pos[3 * slot + 1] = (PY); c->con_pos[3 * slot + 2] = (PZ); \
                 for (int k_ = 0; k_ < 3; k_++) c->con_frame[3 * slot + k_] = nrm[k_]; } }
        GRX_BB_WRITE(LV(cv0), r0, LV(cpx0), LV(cpy0), LV(cpz0), LV(ch0))
        GRX_BB_WRITE(LV(cv1), r1, LV(cpx1), LV(cpy1), LV(cpz1), LV(ch1))
        GRX_BB_WRITE(LV(cv2), r2, LV(cpx2), LV(cpy2), LV(cpz2), LV(ch2))
#undef GRX_BB_WRITE
      }
    }
    WAVE_SYNC();
    LANE0 { c->cnt[0] = base + total; }
    WAVE_SYNC();
  }
}
#undef GRX_BB_AXIS
#undef GRX_BB_LOAD

GRX_MEM void grx_collision(const GrxModel* m, GrxCtx* c, int lane_) {
  GRX_FRESH_MODEL(m, c);
  // geom frames (they share LDS with the composite inertias of the previous stage)
  FOR_LANES {
    for (int i = lane; i < GRX_NGC; i += 64) {
      int b = m->geom_bodyid[i];
      float lpv[3] = {m->geom_pos[3 * i], m->geom_pos[3 * i + 1], m->geom_pos[3 * i + 2]}, lqv[4] = {m->geom_quat[4 * i], m->geom_quat[4 * i + 1], m->geom_quat[4 * i + 2], m->geom_quat[4 * i + 3]}, v[3], R[9], Rw[9];
      mulMatVec3f(v, c->xmat + 9 * b, lpv);
      const int sh = (S::kShift && m->nshift) ? m->geom_shift[i] : 0;
      for (int e = 0; e < 3; e++) v[e] += c->xpos[3 * b + e];
      quat2matf(R, lqv); mulMat3f(Rw, c->xmat + 9 * b, R);
      if (S::kShiftRot && sh == 2) grx_apply_group_rotation(c->shift + 3, v, Rw);
      for (int e = 0; e < 3; e++) c->gxpos[3 * i + e] = v[e] + (sh ? c->shift[e] : 0.0f);
      for (int e = 0; e < 9; e++) c->gxmat[9 * i + e] = Rw[e];
    }
  }
  LANE0 { c->cnt[0] = 0; c->cnt[7] = 0; }
  WAVE_SYNC();
  GRX_RNDINJ(7, (grx_rnd(c->gxpos, 3 * m->ngeom), grx_rnd(c->gxmat, 9 * m->ngeom)));
  GRX_SUBTICK(c, 12);
  // Wall lattice (maze layouts): a moving sphere / capsule only meets the walls of the 3 x 3 cells around its centre -- nine table lookups per
  // mover instead of one bounding-sphere test per (mover, wall) pair of the flat list; same pairs, same tests, same narrow phase.
  if (m->ngridgeom > 0) {
    FOR_LANES {
      for (int it = lane; it < 9 * m->ngridgeom; it += 64) {
        const int k = it / 9, nb = it - 9 * k, rec = m->grid_geom[k], g1 = rec & 0xFFF, t1 = rec >> 12;
        const float r = m->grid_geom_bound[2 * k], margin = m->grid_geom_bound[2 * k + 1];
        const int ix = (int)floorf((c->gxpos[3 * g1] - m->gridx0) * m->gridinv) + (nb % 3) - 1, iy = (int)floorf((c->gxpos[3 * g1 + 1] - m->gridy0) * m->gridinv) + (nb / 3) - 1;
        if (ix >= 0 && iy >= 0 && ix < m->gridnx && iy < m->gridny) {
          const int wl = m->grid_cell[iy * m->gridnx + ix];
          if (wl >= 0) {
            const int g2 = m->grid_wall_geom[wl], p = m->grid_pair[k * m->ngridwall + wl];
            float dx[3] = {c->gxpos[3 * g2] - c->gxpos[3 * g1], c->gxpos[3 * g2 + 1] - c->gxpos[3 * g1 + 1], c->gxpos[3 * g2 + 2] - c->gxpos[3 * g1 + 2]};
            if (dot3f(dx, dx) <= r * r) {
              if (t1 == 2) grx_sphere_box(m, c, p, g1, g2, margin);
              else grx_capsule_box(m, c, p, g1, g2, margin);
            }
          }
        }
      }
    }
    WAVE_SYNC();
  }
  // Models with more than one wave of candidates (the Fetch arm: 163, most of them hull pairs): a first sweep runs only the bounding-sphere /
  // plane-distance test and compacts the survivors (ballot prefix, pair order), so the narrow phases and the bounding-box tests below see
  // ONE dense pass instead of one sparse, divergent pass per 64 candidates.
  // Scenes with more candidates than the survivor list has room for (the kitchen: 3 736) are swept in chunks of that size: pair order is kept.
  const int ndp = m->ndevpair;
  const bool kGate = S::kMesh && m->ngate > 0;   // joint-box gates of hull pairs (grx_gate_clear); the skin-list sweep of the large scenes does not use them
  unsigned long long gmask = 0ull;   // the model's gates (at most 64: the compiler keeps those of the nearest pairs) evaluated once per pass, one lane each; the sweep tests a bit
  if (kGate) { GRX_LANEVAR_I(gc); FOR_LANES { LV(gc) = (lane < m->ngate) ? grx_gate_clear(m, c, lane) : 0; } gmask = GRX_BALLOT(gc); }
#define GRX_GATE_CLEAR(gi) ((int)((gmask >> ((gi) & 63)) & 1ull))
  constexpr bool kChunked = !S::kFixed || S::NG > 64;   // small scenes (every specialised shape but the kitchen): one pass, no loop around the sweep
  // Skin list (large scenes, GPU build): the kitchen has 3 736 candidate pairs of which ~170 pass the bounding-sphere test and ~250 are within 10 cm of
  // passing it.  The flat sweep of all candidates in every substep is replaced by a sweep of the pairs that passed the test with the radius inflated
  // by `skin` when the list was built; the list (pair order) and the geom positions at that moment live in HBM, one row per world, across substeps
  // AND launches.  Every substep checks the largest geom displacement since the build: while 2 * displacement <= skin no pair outside the list can pass
  // the exact test (which only reads the two geom centres; plane geoms are static, checked by the host), so the survivors -- and everything after
  // them -- are exactly those of the full sweep.  Otherwise (and for a zeroed row) the list is rebuilt first: one full sweep per ~40 substeps.
  const int* slist = nullptr; int ncand = ndp;
#if !defined(GRX_EMU)
  if (kChunked && c->skin != nullptr && ndp > 256) {
    volatile int* hdr = c->skin; float* gref = (float*)(c->skin + 4); int* list = c->skin + 4 + 3 * GRX_NGC;
    const float skin = c->skin_r;
    float d2 = 0.0f;
    for (int g = lane_; g < GRX_NGC; g += 64) {
      const float dx = c->gxpos[3 * g] - gref[3 * g], dy = c->gxpos[3 * g + 1] - gref[3 * g + 1], dz = c->gxpos[3 * g + 2] - gref[3 * g + 2];
      d2 = fmaxf(d2, dx * dx + dy * dy + dz * dz);
    }
    const int valid = hdr[1];
    const float dmax2 = grx_reduce_max(d2);
    if (!valid || !(4.0f * dmax2 <= 0.81f * skin * skin)) {   // 10 % of the skin left for the rounding of the two tests
      for (int g = lane_; g < 3 * GRX_NGC; g += 64) gref[g] = c->gxpos[g];
      int ns = 0;
      for (int base = 0; base < ndp; base += 256) {
        unsigned rec[4]; float mg[4], rb[4]; int pass[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int k = base + 64 * u + lane_, kk = k < ndp ? k : 0; rec[u] = (unsigned)m->devpair_geoms[kk]; mg[u] = m->devpair_bound[2 * kk]; rb[u] = m->devpair_bound[2 * kk + 1]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int g1 = rec[u] & 0xFFF, g2 = (rec[u] >> 12) & 0xFFF, t1 = (rec[u] >> 24) & 0xF;
          float dx[3] = {c->gxpos[3 * g2] - c->gxpos[3 * g1], c->gxpos[3 * g2 + 1] - c->gxpos[3 * g1 + 1], c->gxpos[3 * g2 + 2] - c->gxpos[3 * g1 + 2]};
          const float r = rb[u] + mg[u] + skin;
          int ps;
          if (t1 == 0) { float n[3] = {c->gxmat[9 * g1 + 2], c->gxmat[9 * g1 + 5], c->gxmat[9 * g1 + 8]}; ps = dot3f(dx, n) <= r; }
          else ps = dot3f(dx, dx) <= r * r;
          pass[u] = (base + 64 * u + lane_ < ndp) && ps;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const unsigned long long bm = __ballot(pass[u]);
          if (pass[u]) list[ns + __builtin_popcountll(bm & ((1ull << lane_) - 1ull))] = base + 64 * u + lane_;
          ns += __builtin_popcountll(bm);
        }
      }
      if (lane_ == 0) { hdr[0] = ns; hdr[1] = 1; }
      __threadfence_block();
      WAVE_SYNC();
    }
    ncand = hdr[0]; slist = list;
  }
#endif
  const int cap = c->jpool - 256, compact = ncand > 64 && cap >= 64;
  const bool classed = kChunked && ndp > 512 && ndp < 65536;   // a property of the MODEL (not of the kernel shape): the generic and the specialised kernel agree
  const int chunk = (kChunked && compact && cap < ncand) ? cap : (ncand > 0 ? ncand : 1);
  int c0 = 0;
  do {
  const int cend = c0 + chunk < ncand ? c0 + chunk : ncand;
  int nsurv = cend - c0; const int* surv = nullptr;
  if (compact) {
    int* sv = (int*)(c->Jp + 256);   // the Jacobian pool is free until the constraint stage ([0, 128) is c->red, [128, 222) the hull-pair queue + portal)
    int ns = 0;
    // four groups of 64 candidates per round: their table records (global memory, one dependent load chain per candidate) are fetched together,
    // so that a round pays one memory latency instead of four
    for (int base = c0; base < cend; base += 256) {
      GRX_LANEVAR_I(ps0); GRX_LANEVAR_I(ps1); GRX_LANEVAR_I(ps2); GRX_LANEVAR_I(ps3);
      GRX_LANEVAR_I(kp0); GRX_LANEVAR_I(kp1); GRX_LANEVAR_I(kp2); GRX_LANEVAR_I(kp3);
      FOR_LANES {
        unsigned rec[4]; float mg[4], rb[4]; int ok[4], kp[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int k = base + 64 * u + lane;
          ok[u] = k < cend;
          kp[u] = ok[u] ? k : c0;
          if (kChunked && slist) kp[u] = slist[kp[u]];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) { const int kk = kp[u]; rec[u] = (unsigned)m->devpair_geoms[kk]; mg[u] = m->devpair_bound[2 * kk]; rb[u] = m->devpair_bound[2 * kk + 1]; }
        int gate[4];
#pragma unroll
        for (int u = 0; u < 4; u++) gate[u] = kGate ? m->devpair_gate[kp[u]] : -1;
        int pass[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int g1 = rec[u] & 0xFFF, g2 = (rec[u] >> 12) & 0xFFF, t1 = (rec[u] >> 24) & 0xF;
          float dx[3] = {c->gxpos[3 * g2] - c->gxpos[3 * g1], c->gxpos[3 * g2 + 1] - c->gxpos[3 * g1 + 1], c->gxpos[3 * g2 + 2] - c->gxpos[3 * g1 + 2]};
          int ps;
          if (t1 == 0) { float n[3] = {c->gxmat[9 * g1 + 2], c->gxmat[9 * g1 + 5], c->gxmat[9 * g1 + 8]}; ps = dot3f(dx, n) <= rb[u] + mg[u]; }
          else { const float r = rb[u] + mg[u]; ps = dot3f(dx, dx) <= r * r; }
          pass[u] = ok[u] && ps;
          if (kGate && pass[u] && gate[u] >= 0 && GRX_GATE_CLEAR(gate[u])) pass[u] = 0;   // proven disjoint at these joint values
          if (classed) {
            const int t2 = rec[u] >> 28;
            // Large scenes, second filter: the bounding spheres of long thin geoms are loose (the kitchen: ~120 capsule-box candidates per substep pass
            // the sphere test, none of them passes this one) -- separating-axis test of the two oriented bounding boxes, grown by the margin plus a
            // rounding allowance: a pair it rejects cannot produce a contact within the margin, so the contact list does not change.
            if (pass[u] && t1 >= 3 && t2 >= 3) pass[u] = grx_obb_overlap(m, c, g1, g2, mg[u] + 4e-6f);
            // kind of narrow phase (see the regrouping below), carried in the bits above the pair index
            kp[u] |= ((t2 == 7 && t1 != 0) ? 2 : ((t1 == 6 && t2 == 6) ? 1 : ((S::kConvex && t1 >= 2 && t2 <= 6 && (t1 == 4 || t1 == 5 || t2 == 4 || t2 == 5)) ? 3 : 0))) << 16;
          }
        }
        LV(ps0) = pass[0]; LV(ps1) = pass[1]; LV(ps2) = pass[2]; LV(ps3) = pass[3];
        LV(kp0) = kp[0]; LV(kp1) = kp[1]; LV(kp2) = kp[2]; LV(kp3) = kp[3];
      }
#define GRX_COMPACT_GROUP(PS, KP) { const unsigned long long bm = GRX_BALLOT(PS); \
        FOR_LANES { if (LV(PS)) sv[ns + __builtin_popcountll(bm & ((1ull << lane) - 1ull))] = LV(KP); } \
        ns += __builtin_popcountll(bm); }
      GRX_COMPACT_GROUP(ps0, kp0) GRX_COMPACT_GROUP(ps1, kp1) GRX_COMPACT_GROUP(ps2, kp2) GRX_COMPACT_GROUP(ps3, kp3)
#undef GRX_COMPACT_GROUP
    }
    WAVE_SYNC();
    nsurv = ns; surv = sv;
    GRX_SUBTICK(c, 19);
    // Large scenes: the survivors (the kitchen: ~170 per substep, three rounds of 64) are regrouped by the KIND of narrow phase they need -- analytic
    // primitive tests, box-box (queued), hull pairs (bounding-box test + queue), portal refinement on a lane -- so that a round of 64 lanes runs one
    // kind instead of paying every kind's divergent branch in every round.  The contact list is put back into pair order afterwards (below).
    if (classed && ns > 64 && 2 * ns <= cap) {
      int* dst = sv + ns;
      int off[4] = {0, 0, 0, 0};
      for (int base = 0; base < ns; base += 64) {
        GRX_LANEVAR_I(cl);
        FOR_LANES { LV(cl) = base + lane < ns ? (sv[base + lane] >> 16) : -1; }
        for (int q = 0; q < 3; q++) { GRX_LANEVAR_I(hit); FOR_LANES { LV(hit) = LV(cl) == q; } off[q + 1] += __builtin_popcountll(GRX_BALLOT(hit)); }
      }
      off[3] += off[2] + off[1]; off[2] += off[1];     // counts of the classes 0 .. 2 -> start of the classes 1 .. 3
      for (int base = 0; base < ns; base += 64) {
        GRX_LANEVAR_I(cl);
        FOR_LANES { LV(cl) = base + lane < ns ? (sv[base + lane] >> 16) : -1; }
        for (int q = 0; q < 4; q++) {
          GRX_LANEVAR_I(hit);
          FOR_LANES { LV(hit) = LV(cl) == q; }
          const unsigned long long bm = GRX_BALLOT(hit);
          FOR_LANES { if (LV(hit)) dst[off[q] + __builtin_popcountll(bm & ((1ull << lane) - 1ull))] = sv[base + lane]; }
          off[q] += __builtin_popcountll(bm);
        }
      }
      WAVE_SYNC();
      surv = dst;
    }
    GRX_SUBTICK(c, 20);
  }
  for (int base = 0; base < nsurv; base += 64) {
    GRX_LANEVAR_I(boxq); GRX_LANEVAR_I(meshq); GRX_LANEVAR_I(pairq);
    FOR_LANES {
      int isbox = 0, ismesh = 0, pq = 0;
      if (base + lane < nsurv) {
        const int k = surv ? (surv[base + lane] & 0xFFFF) : ((kChunked && slist) ? slist[base + lane] : base + lane);
        // one packed record per candidate (geoms, types, margin, broad-phase radius): a single level of model-table loads
        const unsigned rec = (unsigned)m->devpair_geoms[k];
        const int g1 = rec & 0xFFF, g2 = (rec >> 12) & 0xFFF, t1 = (rec >> 24) & 0xF, t2 = rec >> 28;
        const float margin = m->devpair_bound[2 * k], rb = m->devpair_bound[2 * k + 1];
        int pass = 1;
        if (!surv) {
          float dx[3] = {c->gxpos[3 * g2] - c->gxpos[3 * g1], c->gxpos[3 * g2 + 1] - c->gxpos[3 * g1 + 1], c->gxpos[3 * g2 + 2] - c->gxpos[3 * g1 + 2]};
          if (t1 == 0) {
            float n[3] = {c->gxmat[9 * g1 + 2], c->gxmat[9 * g1 + 5], c->gxmat[9 * g1 + 8]};
            pass = dot3f(dx, n) <= rb + margin;
          } else {
            float r = rb + margin;
            pass = dot3f(dx, dx) <= r * r;
          }
          if (kGate && pass) { const int gi = m->devpair_gate[k]; if (gi >= 0 && GRX_GATE_CLEAR(gi)) pass = 0; }
        }
        pq = m->devpair[k];
        if (pass) {
          const int p = pq;
#ifdef GRX_DBG_NO_OBB
          if (t2 == 7 && t1 != 0) { if (S::kMesh) ismesh = 1; }
#else
          if (t2 == 7 && t1 != 0) { if (S::kMesh) ismesh = grx_obb_overlap(m, c, g1, g2, margin); }
#endif
          else if (t1 == 2 && t2 == 2) grx_sphere_sphere_raw(c, p, c->gxpos + 3 * g1, m->geom_size[3 * g1], c->gxpos + 3 * g2, m->geom_size[3 * g2], margin);
          else if (t1 == 2 && t2 == 3) grx_sphere_capsule(m, c, p, g1, g2, margin);
          else if (t1 == 0 && t2 == 2) grx_plane_sphere(m, c, p, g1, g2, margin);
          else if (t1 == 0 && t2 == 3) grx_plane_capsule(m, c, p, g1, g2, margin);
          else if (t1 == 3 && t2 == 6) grx_capsule_box(m, c, p, g1, g2, margin);
          else if (t1 == 3 && t2 == 3) grx_capsule_capsule(m, c, p, g1, g2, margin);
          else if (t1 == 2 && t2 == 6) grx_sphere_box(m, c, p, g1, g2, margin);
          else if (t1 == 0 && t2 == 6) grx_plane_box(m, c, p, g1, g2, margin);
          else if (t1 == 6 && t2 == 6) isbox = 1;
          else if (S::kConvex && t1 == 0 && t2 == 4) grx_plane_ellipsoid(m, c, p, g1, g2, margin);
          else if (S::kConvex && t1 == 0 && t2 == 5) grx_plane_cylinder(m, c, p, g1, g2, margin);
          else if (S::kConvex && t1 >= 2 && t2 <= 6 && (t1 == 4 || t1 == 5 || t2 == 4 || t2 == 5)) grx_convex_pair(m, c, p, g1, g2, t1, t2, margin);
          else if (t1 == 0 && t2 == 7) {
            if (m->geom_meshnum[g2] <= 32) grx_plane_mesh_small(m, c, p, g1, g2, margin);
            else { int q = GRX_ATOMIC_ADD(&c->cnt[7], 1); if (q < 32) c->ired[q] = p; }
          }
        }
      }
      LV(boxq) = isbox; LV(meshq) = ismesh; LV(pairq) = pq;
    }
    WAVE_SYNC();
    GRX_SUBTICK(c, 13);
    if (S::kMesh) {   // hull-vs-convex pairs that passed the bounding-box filter: pair order, the whole wave on each
      const unsigned long long mm = GRX_BALLOT(meshq);
      if (__builtin_expect(mm != 0ull, 0)) {   // marked cold: the register allocator then places the spill code this region needs around IT instead of inside the hot stages
        int* queue = (int*)(c->Jp + 128);   // the Jacobian pool is free until the constraint stage; [0, 128) is c->red
        FOR_LANES { if (LV(meshq)) queue[__builtin_popcountll(mm & ((1ull << lane) - 1ull))] = LV(pairq); }
        WAVE_SYNC();
#ifndef GRX_DBG_NO_MESHPAIRS
        grx_mesh_pairs(m, c, queue, __builtin_popcountll(mm), lane_);
#endif
      }
    }
    GRX_SUBTICK(c, 16);
    // box-box pairs that passed the broad phase: queue them (pair order) and let eight lanes work on each
    {
      const unsigned long long bm = GRX_BALLOT(boxq);
      if (bm) {
        int* queue = (int*)c->red;
        FOR_LANES { if (LV(boxq)) queue[__builtin_popcountll(bm & ((1ull << lane) - 1ull))] = LV(pairq); }
        WAVE_SYNC();
        grx_box_box_queue(m, c, queue, __builtin_popcountll(bm), lane_);
      }
    }
    GRX_SUBTICK(c, 11);
    // large hulls (a moving link near the plane): all lanes scan the vertices of one pair at a time
    int nbig = c->cnt[7] < 32 ? c->cnt[7] : 32;
    for (int l = 0; l < nbig; l++) {
      int p = c->ired[l];
      int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p];
      int adr = m->geom_meshadr[g2], num = m->geom_meshnum[g2];
      float margin = m->pair_margin[p];
      const float* gm = c->gxmat + 9 * g2;
      float n[3] = {c->gxmat[9 * g1 + 2], c->gxmat[9 * g1 + 5], c->gxmat[9 * g1 + 8]}, nl[3];
      mulMatTVec3f(nl, gm, n);
      float off = dot3f(c->gxpos + 3 * g2, n) - dot3f(c->gxpos + 3 * g1, n);
      FOR_LANES {
        float bd = 1e30f; int bi = -1;
        for (int v = lane; v < num; v += 64) {
          float dd = m->mesh_vert[3 * (adr + v)] * nl[0] + m->mesh_vert[3 * (adr + v) + 1] * nl[1] + m->mesh_vert[3 * (adr + v) + 2] * nl[2] + off;
          if (dd < bd) { bd = dd; bi = v; }
        }
        c->red[lane] = bd; c->red[64 + lane] = (float)bi;
      }
      WAVE_SYNC();
      LANE0 {
        float bd = 1e30f; int best = -1;
        for (int e = 0; e < 64; e++) {
          float dd = c->red[e]; int vi = (int)c->red[64 + e];
          if (vi >= 0 && (dd < bd || (dd == bd && vi < best))) { bd = dd; best = vi; }
        }
        if (best >= 0 && bd <= margin) {
          int aa = m->mesh_adjadr[adr + best], an = m->mesh_adjnum[adr + best], cn = 0;
          for (int e = -1; e < an && cn < 4; e++) {
            int v = (e < 0) ? best : m->mesh_adj[aa + e];
            float lv[3] = {m->mesh_vert[3 * (adr + v)], m->mesh_vert[3 * (adr + v) + 1], m->mesh_vert[3 * (adr + v) + 2]}, w[3], pos[3];
            float dd = lv[0] * nl[0] + lv[1] * nl[1] + lv[2] * nl[2] + off;
            if (e >= 0 && dd > margin) continue;
            mulMatVec3f(w, gm, lv);
            for (int t = 0; t < 3; t++) pos[t] = w[t] + c->gxpos[3 * g2 + t] - 0.5f * dd * n[t];
            grx_add_contact(c, p, pos, n, dd); cn++;
          }
        }
        c->cnt[7] = 0;
      }
      WAVE_SYNC();
    }
  }
  c0 += chunk;
  } while (kChunked && c0 < ncand);
  LANE0 { if (c->cnt[0] > c->maxcon) c->cnt[0] = c->maxcon; }
  WAVE_SYNC();
  // The noslip sweeps are Gauss-Seidel over the contact list: while they have not converged their iterates depend on the ORDER of the list.
  // The late queues above (box-box, hull pairs, large plane-mesh pairs) append their contacts after everything else; put the list back into
  // pair order (stable: a pair's contacts keep their order), the order of the reference's list.  One lane per contact, rank by counting.
  if ((S::kNoslip && m->noslip_iterations > 0) || classed) {
    const int nc = c->cnt[0];
    GRX_LANEVAR_I(rk); GRX_LANEVAR_I(pk); GRX_LANEVAR(dk); GRX_LANEVAR(x0); GRX_LANEVAR(x1); GRX_LANEVAR(x2); GRX_LANEVAR(f0); GRX_LANEVAR(f1); GRX_LANEVAR(f2);
    FOR_LANES {
      int r = 0, key = 0;
      if (lane < nc) {
        key = c->con_pair[lane];
        for (int j = 0; j < nc; j++) { const int kj = c->con_pair[j]; r += (kj < key) || (kj == key && j < lane); }
        LV(dk) = c->con_dist[lane];
        LV(x0) = c->con_pos[3 * lane]; LV(x1) = c->con_pos[3 * lane + 1]; LV(x2) = c->con_pos[3 * lane + 2];
        LV(f0) = c->con_frame[3 * lane]; LV(f1) = c->con_frame[3 * lane + 1]; LV(f2) = c->con_frame[3 * lane + 2];
      }
      LV(rk) = r; LV(pk) = key;
    }
    WAVE_SYNC();
    FOR_LANES {
      if (lane < nc) {
        const int r = LV(rk);
        c->con_pair[r] = LV(pk); c->con_dist[r] = LV(dk);
        c->con_pos[3 * r] = LV(x0); c->con_pos[3 * r + 1] = LV(x1); c->con_pos[3 * r + 2] = LV(x2);
        c->con_frame[3 * r] = LV(f0); c->con_frame[3 * r + 1] = LV(f1); c->con_frame[3 * r + 2] = LV(f2);
      }
    }
    WAVE_SYNC();
  }
}

// ------------------------------------------------------------------------------------------
// K9 constraint rows (equality weld, dof frictionloss, joint limits, pyramidal contacts)
// ------------------------------------------------------------------------------------------
GRX_MEM float grx_impedance(const float* solimp, float pos) {
  float dmin = fminf(GRX_MAXIMP, fmaxf(GRX_MINIMP, solimp[0])), dmax = fminf(GRX_MAXIMP, fmaxf(GRX_MINIMP, solimp[1]));
  float width = fmaxf(0.0f, solimp[2]), mid = fminf(GRX_MAXIMP, fmaxf(GRX_MINIMP, solimp[3])), power = fmaxf(1.0f, solimp[4]);
  if (dmin == dmax || width <= GRX_MINVAL) return 0.5f * (dmin + dmax);
  float x = fabsf(pos) / width;
  if (x >= 1) return dmax;
  if (x <= 0) return dmin;
  float y;
  if (power == 1.0f) y = x;
  else if (power == 2.0f) y = (x <= mid) ? x * x / mid : 1.0f - (1.0f - x) * (1.0f - x) / (1.0f - mid);
  else if (x <= mid) y = powf(x, power) / powf(mid, power - 1.0f);
  else y = 1.0f - powf(1.0f - x, power) / powf(1.0f - mid, power - 1.0f);
  return dmin + y * (dmax - dmin);
}

// column d of the translational / rotational Jacobian of a world point on body b (zero if d not in chain)
GRX_MEM void grx_jac_col(const GrxModel* m, const GrxCtx* c, int b, const float* point, int d, float* jp, float* jr) {
  unsigned lo = (unsigned)m->dof_chainmask[2 * b], hi = (unsigned)m->dof_chainmask[2 * b + 1];
  int in = d < 32 ? (lo >> d) & 1u : (hi >> (d - 32)) & 1u;
  if (!in) { jp[0] = jp[1] = jp[2] = 0; jr[0] = jr[1] = jr[2] = 0; return; }
  const float* cref = c->xpos + 3 * m->body_rootid[b];
  float off[3] = {point[0] - cref[0], point[1] - cref[1], point[2] - cref[2]}, t[3];
  const float* cd = c->cdof + 6 * d;
  float w[3] = {cd[0], cd[1], cd[2]};
  cross3f(t, w, off);
  jr[0] = w[0]; jr[1] = w[1]; jr[2] = w[2]; jp[0] = cd[3] + t[0]; jp[1] = cd[4] + t[1]; jp[2] = cd[5] + t[2];
}

// dof span [lo, lo+len) of a 64-bit dof mask
GRX_MEM void grx_mask_span(unsigned long long msk, int* lo, int* len) {
  if (!msk) { *lo = 0; *len = 0; return; }
  int l = __builtin_ctzll(msk), h = 63 - __builtin_clzll(msk);
  *lo = l; *len = h - l + 1;
}
GRX_MEM unsigned long long grx_chainmask(const GrxModel* m, int b) {
  return ((unsigned long long)(unsigned)m->dof_chainmask[2 * b + 1] << 32) | (unsigned)m->dof_chainmask[2 * b];
}
// efc_row[r] = off | lo << 14 | len << 21: 14-bit pool offsets (the large tables of the overflow lane hold up to 16 368 words), dof spans below 128
#define GRX_ROW_OFF(info) ((info) & 0x3FFF)
#define GRX_ROW_LO(info) (((info) >> 14) & 0x7F)
#define GRX_ROW_LEN(info) (((info) >> 21) & 0x7F)
#define GRX_ROW_PACK(off, lo, len) ((off) | ((lo) << 14) | ((len) << 21))
#define GRX_ROW_FROM_STATIC(x) GRX_ROW_PACK((x) & 0xFFF, ((x) >> 12) & 0xFF, ((x) >> 20) & 0xFF)   // the compiler's static rows (weld_row, jeq_row): off | lo << 12 | len << 20
// Second dof span of a row (contacts whose two body chains leave a gap of unused dofs between them): it rides in the upper bits of
// efc_id = sub | id << 4 | loB << 12 | lenB << 20, and its entries follow the first span's entries in the pool.
#define GRX_ROW_IDOF(id) (((id) >> 4) & 0xFF)
#define GRX_ROWB_LO(id) (((id) >> 12) & 0xFF)
#define GRX_ROWB_LEN(id) (((id) >> 20) & 0xFF)
// index of dof d inside the row's storage, or -1 when the row has no entry for it
GRX_MEM int grx_row_pos(int info, int id, int d) {
  const int ja = d - GRX_ROW_LO(info);
  if ((unsigned)ja < (unsigned)GRX_ROW_LEN(info)) return ja;
  if (!S::kTwoSpan) return -1;
  const int jb = d - GRX_ROWB_LO(id);
  return ((unsigned)jb < (unsigned)GRX_ROWB_LEN(id)) ? GRX_ROW_LEN(info) + jb : -1;
}
// row r of J times a dof vector
GRX_MEM float grx_row_dot(const GrxCtx* c, int r, const float* v) {
  const int info = c->efc_row[r], off = GRX_ROW_OFF(info), lo = GRX_ROW_LO(info), len = GRX_ROW_LEN(info);
  float s = 0;
#pragma unroll 8
  for (int j = 0; j < len; j++) s += c->Jp[off + j] * v[lo + j];
  if (S::kTwoSpan) {
    const int id = c->efc_id[r], lob = GRX_ROWB_LO(id), lenb = GRX_ROWB_LEN(id);
#pragma unroll 2
    for (int j = 0; j < lenb; j++) s += c->Jp[off + len + j] * v[lob + j];
  }
  return s;
}

GRX_MEM void grx_make_constraint(const GrxModel* m, GrxCtx* c, int lane_) {
#if !defined(GRX_EMU)
  // The lane index is made opaque for this stage: its cheap lane-derived values (packed row descriptors) are then recomputed here
  // instead of being hoisted out of the 20-substep loop, kept live across it and spilled to scratch (one dword per lane, but written
  // back to HBM by every wave).  Doing this for the whole pass costs more recomputation than it saves (measured -3 % on the hand models).
  asm volatile("" : "+v"(lane_));
#endif
  GRX_FRESH_MODEL(m, c);
  const int nv = GRX_NVC;
  const int ncon = c->cnt[0];
  // ---- row bookkeeping: one lane per joint (limit flags) and one lane per contact (row count, dof span), then
  // exclusive prefix sums across the wave give every limit / contact its first row and its Jacobian-pool offset.
  const int nwr = 6 * m->nweld, ne = nwr + m->njeq, nf = m->nfric, wpool = m->wpool;   // equality rows: the welds' six each, then one per joint equality
  GRX_LANEVAR_I(limc); GRX_LANEVAR_I(conr); GRX_LANEVAR_I(conw); GRX_LANEVAR_I(coni);
  GRX_LANEVAR_I(tenf); GRX_LANEVAR_I(tenc); GRX_LANEVAR_I(tenw); GRX_LANEVAR(tenl);
  FOR_LANES {
    int f = 0;
    if (lane < GRX_NJC) {
      const int j = lane;
      if (m->jnt_limited[j] && m->jnt_type[j] >= 2) {
        float q = c->qpos[m->jnt_qposadr[j]], mg = m->jnt_margin[j];
        if (q - m->jnt_range[2 * j] < mg) f |= 1;
        if (m->jnt_range[2 * j + 1] - q < mg) f |= 2;
      }
      c->ired[j] = f;
    }
    LV(limc) = (f & 1) + ((f >> 1) & 1);
    int nr = 0, slen = 0;
    if (lane < ncon) {
      const int k = lane;
      int p = c->con_pair[k], dim = m->pair_condim[p];
      int active = c->con_dist[k] < m->pair_margin[p] - m->pair_gap[p];
      nr = active ? ((dim == 1) ? 1 : 2 * (dim - 1)) : 0;
      int cb1 = m->geom_bodyid[m->pair_geom1[p]], cb2 = m->geom_bodyid[m->pair_geom2[p]];
      c->con_b1[k] = cb1; c->con_b2[k] = cb2;
      const int sp = m->pair_span[p];   // static: the two dof spans of the pair's body chains
      slen = ((sp >> 8) & 0xFF) + ((sp >> 24) & 0xFF);
      c->con_span[k] = sp;
    }
    LV(conr) = nr; LV(conw) = nr * slen; LV(coni) = nr ? slen : 0;
    // fixed-tendon limits: one lane per tendon (length = sum coef * qpos)
    int tf = 0; float tl = 0.0f;
    if (lane < m->ntendon && m->tendon_limited[lane]) {
      const int t = lane;
      for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) tl += m->wrap_coef[w] * c->qpos[m->wrap_qadr[w]];
      const float mg = m->tendon_margin[t];
      if (tl - m->tendon_range[2 * t] < mg) tf |= 1;
      if (m->tendon_range[2 * t + 1] - tl < mg) tf |= 2;
    }
    LV(tenf) = tf; LV(tenl) = tl;
    LV(tenc) = (tf & 1) + ((tf >> 1) & 1);
    LV(tenw) = LV(tenc) * (lane < m->ntendon ? (m->tendon_span[lane] >> 8) : 0);
  }
  WAVE_SYNC();
  GRX_SUBTICK(c, 0);
  GRX_LANEVAR_I(limx); GRX_LANEVAR_I(conrx); GRX_LANEVAR_I(conwx); GRX_LANEVAR_I(tenx); GRX_LANEVAR_I(tenwx);
  int nl, nc_all, pool_all, nlt = 0, tpool = 0;
  GRX_SCAN_EXCL(limc, limx, nl);
  if (m->ntendon) { GRX_SCAN_EXCL(tenc, tenx, nlt); GRX_SCAN_EXCL(tenw, tenwx, tpool); }
  const int nlj = nl;   // joint-limit rows; tendon-limit rows follow them (MuJoCo's row order)
  nl += nlt;
  GRX_SCAN_EXCL(conr, conrx, nc_all);
  GRX_SCAN_EXCL(conw, conwx, pool_all);
  // contacts come last: keep as many whole contacts as fit into the row table and the Jacobian pool
  const int rows0 = ne + nf + nl, pool0 = wpool + nf + nlj + tpool;
  const int maxefc = c->maxefc, jpool = c->jpool;
  int overflow = (rows0 > maxefc) || (pool0 > jpool), ncon_fit = ncon, nc = nc_all;
  GRX_PMAX(c, 32, rows0 + nc_all); GRX_PMAX(c, 33, pool0 + pool_all); GRX_PMAX(c, 34, c->cnt[0]);
  if (c->soft_maxefc > 0 && (rows0 + nc_all > c->soft_maxefc || pool0 + pool_all > c->soft_jpool || c->cnt[0] > c->soft_maxcon)) { LANE0 { c->cnt[2] |= GRX_ST_SOFT; } }
  if (rows0 + nc_all > maxefc || pool0 + pool_all > jpool) {  // rare: find the first contact that does not fit
    GRX_LANEVAR(failp);
    FOR_LANES {
      int fits = (lane >= ncon) || (rows0 + LV(conrx) + LV(conr) <= maxefc && pool0 + LV(conwx) + LV(conw) <= jpool);
      LV(failp) = fits ? -1000.0f : -(float)lane;
    }
    const float mx = grx_reduce_max(failp);
    if (mx > -999.0f) { ncon_fit = (int)(-mx); overflow = 1; nc = GRX_LANE_READ_I(conrx, ncon_fit); }
  }
  int nefc = rows0 + nc;
  if (nefc > maxefc) nefc = maxefc;
  // items of the contact-Jacobian pass: one per (kept contact, dof of its spans); the running item offset (con_ioff) lets an
  // item find its contact with a binary search
  GRX_LANEVAR_I(conix); int nitem;
  FOR_LANES { if (lane >= ncon_fit) LV(coni) = 0; }
  GRX_SCAN_EXCL(coni, conix, nitem);
  GRX_SUBTICK(c, 1);
  // ---- descriptors
  FOR_LANES {
    if (lane < nwr) {  // welds: spans and pool offsets are static (weld_row)
      const int r = lane, w = r / 6, sub = r - 6 * w, info0 = GRX_ROW_FROM_STATIC(m->weld_row[w]);
      c->efc_kind[r] = GRX_ROW_EQ; c->efc_id[r] = (m->weld_eq[w] << 4) | sub;
      c->efc_row[r] = info0 + sub * GRX_ROW_LEN(info0);  // the offset field is the low one: adding sub*len moves to row sub
    } else if (lane < ne) {  // joint equalities (sub 8: their invweight sits in the second eq_invweight slot too)
      const int r = lane, j = r - nwr;
      c->efc_kind[r] = GRX_ROW_EQ; c->efc_id[r] = (m->jeq_eq[j] << 4) | 8; c->efc_row[r] = GRX_ROW_FROM_STATIC(m->jeq_row[j]);
    }
    if ((S::kFixed ? S::NF > 0 : true) && nf > 0)   // compile-time dead for the shapes without friction-loss dofs
      for (int d = lane; d < nv; d += 64) {
        if (m->dof_frictionloss[d] > 0) {
          int r = ne; for (int q = 0; q < d; q++) if (m->dof_frictionloss[q] > 0) r++;
          c->efc_kind[r] = GRX_ROW_FRICTION; c->efc_id[r] = d << 4; c->efc_row[r] = GRX_ROW_PACK(wpool + (r - ne), d, 1);
        }
      }
    if (lane < GRX_NJC) {
      const int j = lane, f = c->ired[j];
      if (f) {
        int r = ne + nf + LV(limx);
        int dd = m->jnt_dofadr[j];
        if (f & 1) { if (r < nefc) { c->efc_kind[r] = GRX_ROW_LIMIT; c->efc_id[r] = j << 4; c->efc_row[r] = GRX_ROW_PACK(wpool + (r - ne), dd, 1); } r++; }
        if (f & 2) { if (r < nefc) { c->efc_kind[r] = GRX_ROW_LIMIT; c->efc_id[r] = (j << 4) | 1; c->efc_row[r] = GRX_ROW_PACK(wpool + (r - ne), dd, 1); } }
      }
    }
    if (LV(tenf)) {
      const int t = lane, f = LV(tenf), sp = m->tendon_span[t], slo = sp & 0xFF, slen = sp >> 8;
      int r = ne + nf + nlj + LV(tenx), off = wpool + nf + nlj + LV(tenwx);
      if (f & 1) { if (r < nefc) { c->efc_kind[r] = GRX_ROW_TENDON; c->efc_id[r] = t << 4; c->efc_row[r] = GRX_ROW_PACK(off, slo, slen); } r++; off += slen; }
      if (f & 2) { if (r < nefc) { c->efc_kind[r] = GRX_ROW_TENDON; c->efc_id[r] = (t << 4) | 1; c->efc_row[r] = GRX_ROW_PACK(off, slo, slen); } }
    }
    if (lane < ncon) {
      const int k = lane;
      int nr = (k < ncon_fit) ? LV(conr) : 0;
      int r = rows0 + LV(conrx), off = pool0 + LV(conwx);
      c->con_efc[k] = nr ? r : -1;
      c->con_nr[k] = nr;
      const int sp = c->con_span[k], slo = sp & 0xFF, slena = (sp >> 8) & 0xFF, slob = (sp >> 16) & 0xFF, slenb = (sp >> 24) & 0xFF, slen = slena + slenb;
      c->con_ioff[k] = LV(conix);
      for (int q = 0; q < nr; q++) {
        c->efc_kind[r + q] = GRX_ROW_CONTACT; c->efc_id[r + q] = (k << 4) | q | (slob << 12) | (slenb << 20);
        c->efc_row[r + q] = GRX_ROW_PACK(off + q * slen, slo, slena);
      }
    }
  }
  LANE0 { c->cnt[1] = nefc; c->cnt[3] = ne; c->cnt[4] = nf; c->cnt[5] = nl; if (overflow) c->cnt[2] |= GRX_ST_EFC_OVERFLOW; }
  WAVE_SYNC();
  GRX_SUBTICK(c, 2);
  // ---- Jacobian rows.  zero fill, then per (row-group, dof) items
  // (every (row group, dof) item below writes all of its entries, zeros included: no separate clear of J)
  FOR_LANES {
    // welds: one lane per (weld, dof)
    for (int it = lane; it < (nwr / 6) * nv; it += 64) {
      int w = it / nv, d = it - w * nv;
      int e = m->weld_eq[w];
      int b0 = m->eq_obj1[e], b1 = m->eq_obj2[e];
      const float* data = m->eq_data + 11 * e; const float* rel = m->eq_relpose + 14 * e;
      float bx[2][3], bq[2][4], pos[2][3];
      for (int s = 0; s < 2; s++) {
        int bb = s ? b1 : b0; float v[3], rp[3] = {rel[7 * s], rel[7 * s + 1], rel[7 * s + 2]}, rq[4] = {rel[7 * s + 3], rel[7 * s + 4], rel[7 * s + 5], rel[7 * s + 6]};
        mulMatVec3f(v, c->xmat + 9 * bb, rp);
        for (int k = 0; k < 3; k++) bx[s][k] = c->xpos[3 * bb + k] + v[k];
        mulQuatf(bq[s], c->xquat + 4 * bb, rq); normalize4f(bq[s]);
        float an[3] = {data[3 * (1 - s)], data[3 * (1 - s) + 1], data[3 * (1 - s) + 2]};
        rotVecQuatf(v, an, bq[s]);
        for (int k = 0; k < 3; k++) pos[s][k] = bx[s][k] + v[k];
      }
      float jp0[3], jr0[3], jp1[3], jr1[3];
      grx_jac_col(m, c, b0, pos[0], d, jp0, jr0); grx_jac_col(m, c, b1, pos[1], d, jp1, jr1);
      float ts = data[10];
      float relq[4] = {data[6], data[7], data[8], data[9]}, quat[4], quat1[4] = {bq[1][0], -bq[1][1], -bq[1][2], -bq[1][3]};
      mulQuatf(quat, bq[0], relq);
      float axis[4] = {0, jr0[0] - jr1[0], jr0[1] - jr1[1], jr0[2] - jr1[2]}, t1[4], t2[4];
      mulQuatf(t1, quat1, axis); mulQuatf(t2, t1, quat);
      { int info = c->efc_row[6 * w], jd = d - GRX_ROW_LO(info), len = GRX_ROW_LEN(info), off = GRX_ROW_OFF(info);
        if ((unsigned)jd < (unsigned)len)
          for (int r = 0; r < 3; r++) { c->Jp[off + r * len + jd] = jp0[r] - jp1[r]; c->Jp[off + (3 + r) * len + jd] = 0.5f * ts * t2[1 + r]; } }
      if (d == 0) {  // residuals (one lane per weld)
        float quat2[4]; mulQuatf(quat2, quat1, quat);
        for (int r = 0; r < 3; r++) { c->efc_pos[6 * w + r] = pos[0][r] - pos[1][r]; c->efc_pos[6 * w + 3 + r] = ts * quat2[1 + r]; }
      }
    }
    // joint equalities: one lane per constraint.  r = (q1 - q1_0) - poly(q2 - q2_0), J = e_dof1 - poly'(q2 - q2_0) e_dof2 (MuJoCo mjEQ_JOINT [3P])
    for (int j = lane; j < m->njeq; j += 64) {
      const int r = nwr + j, e = m->jeq_eq[j], info = c->efc_row[r], off = GRX_ROW_OFF(info), lo = GRX_ROW_LO(info), len = GRX_ROW_LEN(info);
      const float* data = m->eq_data + 11 * e;
      const float x = c->qpos[m->jeq_qadr[2 * j + 1]] - data[6];
      const float poly = data[0] + x * (data[1] + x * (data[2] + x * (data[3] + x * data[4])));
      const float deriv = data[1] + x * (2.0f * data[2] + x * (3.0f * data[3] + x * 4.0f * data[4]));
      for (int k = 0; k < len; k++) c->Jp[off + k] = 0.0f;
      c->Jp[off + m->jeq_dof[2 * j] - lo] = 1.0f;
      c->Jp[off + m->jeq_dof[2 * j + 1] - lo] += -deriv;
      c->efc_pos[r] = (c->qpos[m->jeq_qadr[2 * j]] - data[5]) - poly;
    }
    // frictionloss + limits: one lane per row
    GRX_SUBTICK(c, 3);
    for (int r = ne + lane; r < ne + nf + nlj && r < nefc; r += 64) {
      if (c->efc_kind[r] == GRX_ROW_FRICTION) {
        c->Jp[GRX_ROW_OFF(c->efc_row[r])] = 1.0f;
        c->efc_pos[r] = 0;
      } else {
        int j = GRX_ROW_IDOF(c->efc_id[r]), side = c->efc_id[r] & 15; float q = c->qpos[m->jnt_qposadr[j]];
        c->Jp[GRX_ROW_OFF(c->efc_row[r])] = side ? -1.0f : 1.0f;
        c->efc_pos[r] = side ? m->jnt_range[2 * j + 1] - q : q - m->jnt_range[2 * j];
      }
    }
    // tendon limits: one lane per tendon writes its (up to two) rows over the tendon's dof span
    if (LV(tenf)) {
      const int t = lane, f = LV(tenf);
      int r = ne + nf + nlj + LV(tenx);
      for (int side = 0; side < 2; side++) {
        if (!((f >> side) & 1)) continue;
        if (r < nefc) {
          const int info = c->efc_row[r], off = GRX_ROW_OFF(info), lo = GRX_ROW_LO(info), len = GRX_ROW_LEN(info);
          for (int j = 0; j < len; j++) c->Jp[off + j] = 0.0f;
          for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) c->Jp[off + m->wrap_dof[w] - lo] += side ? -m->wrap_coef[w] : m->wrap_coef[w];
          c->efc_pos[r] = side ? m->tendon_range[2 * t + 1] - LV(tenl) : LV(tenl) - m->tendon_range[2 * t];
        }
        r++;
      }
    }
    // contacts: one lane per (contact, dof of its span)
    GRX_SUBTICK(c, 4);
    for (int it = lane; it < nitem; it += 64) {
      int k = 0;  // largest k with item offset <= it (contacts without items share the offset of the next one)
      for (int step = GRX_MAXCON / 2; step > 0; step >>= 1) { int kk = k + step; if (kk < ncon_fit && c->con_ioff[kk] <= it) k = kk; }
      const int jd = it - c->con_ioff[k];
      int r0 = c->con_efc[k];
      const int sp = c->con_span[k], slena = (sp >> 8) & 0xFF, slen = slena + ((sp >> 24) & 0xFF);
      int d = jd < slena ? (sp & 0xFF) + jd : ((sp >> 16) & 0xFF) + jd - slena;
      int p = c->con_pair[k], nrk = c->con_nr[k], dim = (nrk == 1) ? 1 : nrk / 2 + 1;
      int b1 = c->con_b1[k], b2 = c->con_b2[k];
      float pos[3] = {c->con_pos[3 * k], c->con_pos[3 * k + 1], c->con_pos[3 * k + 2]};
      float jp1[3], jr1[3], jp2[3], jr2[3];
      grx_jac_col(m, c, b1, pos, d, jp1, jr1); grx_jac_col(m, c, b2, pos, d, jp2, jr2);
      float dp[3] = {jp2[0] - jp1[0], jp2[1] - jp1[1], jp2[2] - jp1[2]}, dr[3] = {jr2[0] - jr1[0], jr2[1] - jr1[1], jr2[2] - jr1[2]};
      float fr[9] = {c->con_frame[3 * k], c->con_frame[3 * k + 1], c->con_frame[3 * k + 2], 0, 0, 0, 0, 0, 0};
      grx_make_frame(fr);
      float jc[6];
      for (int r = 0; r < 3; r++) { jc[r] = fr[3 * r] * dp[0] + fr[3 * r + 1] * dp[1] + fr[3 * r + 2] * dp[2]; jc[3 + r] = fr[3 * r] * dr[0] + fr[3 * r + 1] * dr[1] + fr[3 * r + 2] * dr[2]; }
      const int off0 = GRX_ROW_OFF(c->efc_row[r0]);
      if (dim == 1) c->Jp[off0 + jd] = jc[0];
      else
        for (int q = 1; q < dim; q++) {
          float mu = m->pair_friction[5 * p + q - 1];
          int ro = off0 + 2 * (q - 1) * slen + jd;
          c->Jp[ro] = jc[0] + mu * jc[q];
          c->Jp[ro + slen] = jc[0] - mu * jc[q];
        }
    }
  }
  WAVE_SYNC();
  GRX_SUBTICK(c, 5);
  // ---- per-row impedance, regulariser, reference acceleration
  FOR_LANES {
    for (int r = lane; r < nefc; r += 64) {
      int kind = c->efc_kind[r], id = GRX_ROW_IDOF(c->efc_id[r]), sub = c->efc_id[r] & 15;
      float solref[2], solimp[5], pos, margin = 0, dA, floss = 0, rscale = 1.0f;
      if (kind == GRX_ROW_EQ) {
        for (int k = 0; k < 2; k++) solref[k] = m->eq_solref[2 * id + k];
        for (int k = 0; k < 5; k++) solimp[k] = m->eq_solimp[5 * id + k];
        pos = c->efc_pos[r]; dA = m->eq_invweight[2 * id + (sub >= 3)];
      } else if (kind == GRX_ROW_FRICTION) {
        for (int k = 0; k < 2; k++) solref[k] = m->dof_solref[2 * id + k];
        for (int k = 0; k < 5; k++) solimp[k] = m->dof_solimp[5 * id + k];
        pos = 0; dA = m->dof_invweight0[id]; floss = m->dof_frictionloss[id];
      } else if (kind == GRX_ROW_LIMIT) {
        for (int k = 0; k < 2; k++) solref[k] = m->jnt_solref[2 * id + k];
        for (int k = 0; k < 5; k++) solimp[k] = m->jnt_solimp[5 * id + k];
        pos = c->efc_pos[r]; margin = m->jnt_margin[id]; dA = m->dof_invweight0[m->jnt_dofadr[id]];
      } else if (kind == GRX_ROW_TENDON) {
        for (int k = 0; k < 2; k++) solref[k] = m->tendon_solref[2 * id + k];
        for (int k = 0; k < 5; k++) solimp[k] = m->tendon_solimp[5 * id + k];
        pos = c->efc_pos[r]; margin = m->tendon_margin[id]; dA = m->tendon_invweight0[id];
      } else {
        int p = c->con_pair[id], dim = m->pair_condim[p];
        for (int k = 0; k < 2; k++) solref[k] = m->pair_solref[2 * p + k];
        for (int k = 0; k < 5; k++) solimp[k] = m->pair_solimp[5 * p + k];
        pos = c->con_dist[id]; margin = m->pair_margin[p] - m->pair_gap[p];
        int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p];
        float tran = m->geom_invweight0[2 * g1] + m->geom_invweight0[2 * g2];
        if (dim == 1) dA = tran;
        else {  // every pyramid row of a contact shares R = 2 mu^2 R(first row)
          float f0 = m->pair_friction[5 * p];
          dA = tran + f0 * f0 * tran;
          float mu = f0 / sqrtf(m->impratio);
          rscale = 2.0f * mu * mu;
        }
        c->efc_pos[r] = pos;
      }
      float imp = grx_impedance(solimp, pos - margin);
      float dmax = fminf(GRX_MAXIMP, fmaxf(GRX_MINIMP, solimp[1]));
      float kk, bb;
      if (solref[0] > 0) { float tc = fmaxf(solref[0], 2.0f * m->timestep), dr = solref[1]; kk = 1.0f / (dmax * dmax * tc * tc * dr * dr); bb = 2.0f / (dmax * tc); }
      else { kk = -solref[0] / (dmax * dmax); bb = -solref[1] / dmax; }
      if (kind == GRX_ROW_FRICTION) kk = 0;
      float R = fmaxf(GRX_MINVAL, (1.0f - imp) * dA / imp) * rscale;
      const float vel = grx_row_dot(c, r, c->qvel);
      c->efc_D[r] = 1.0f / R;
      c->efc_aref[r] = -bb * vel - kk * imp * (pos - margin);
      if (m->nfric) c->efc_floss[r] = floss;
    }
  }
  WAVE_SYNC();
}

// ------------------------------------------------------------------------------------------
// K10 constraint solve: Newton on the primal problem (MuJoCo's default solver [3P]) with an
// exact line search; wave-parallel over dofs / rows / Hessian entries.
// ------------------------------------------------------------------------------------------
// Ma = M a ; jar = J a - aref ; force / active flags ; returns total cost if want_cost
// Row states: 0 = inactive (satisfied inequality), 1 = quadratic, 2 / 3 = friction-loss row saturated at -f / +f.
// Returns 1 if any row changed state with respect to the previous evaluation (bits 0-1 of efc_quad), else 0.
// carried != 0: Ma and jar were advanced along the accepted step (Ma += alpha Mv, jar += alpha Jv) by the caller, only the row states
// and forces are re-derived (no mat-vec, no row dot products).
GRX_MEM int grx_newton_eval(const GrxModel* m, GrxCtx* c, const float* a, int nefc, int carried, int lane_) {
  const int nv = GRX_NVC;
  GRX_LANEVAR(chgp);
  FOR_LANES {
    float chg = 0;
    if (!carried) {
      for (int i = lane; i < nv; i += 64) {
        float s = 0;
#pragma unroll 8
        for (int j = 0; j < nv; j++) s += c->M[i * nv + j] * a[j];
        c->Ma[i] = s;
      }
    }
    for (int r = lane; r < nefc; r += 64) {
      float x = carried ? c->efc_jar[r] : (float)(grx_row_dot(c, r, a) - c->efc_aref[r]), D = c->efc_D[r], f; int st;
      int kind = c->efc_kind[r];
      if (kind == GRX_ROW_EQ) { f = -D * x; st = 1; }
      else if (kind == GRX_ROW_FRICTION) {
        float fl = c->efc_floss[r], Rf = fl / D;
        if (x <= -Rf) { f = fl; st = 3; } else if (x >= Rf) { f = -fl; st = 2; } else { f = -D * x; st = 1; }
      } else {
        if (x < 0) { f = -D * x; st = 1; } else { f = 0; st = 0; }
      }
      const int old = c->efc_quad[r];
      if (st != (old & 3)) chg = 1.0f;
      c->efc_jar[r] = x; c->efc_force[r] = f; c->efc_quad[r] = (old & 0x30) | st;   // bits 4-5: the state this row has in the assembled Hessian
    }
    LV(chgp) = chg;
  }
  WAVE_SYNC();
  return grx_reduce_max(chgp) > 0.5f;
}

// derivative (d1) and curvature (d2) of the cost along the search direction at step alpha
// *same <- (want_same and) every row is, at step alpha, in the state it has at alpha = 0 (efc_quad): the cost is then exactly
// quadratic on [0, alpha]
GRX_MEM void grx_ls_eval(GrxCtx* c, int nefc, float alpha, float q1, float q2, float* d1, float* d2, int want_same, int* same, int lane_) {
#ifdef GRX_LS_STATS
  { extern int g_ls_calls; g_ls_calls++; }
#endif
  GRX_LANEVAR(gp); GRX_LANEVAR(hp); GRX_LANEVAR_I(difp);
  FOR_LANES {
    float g = 0, h = 0; int dif = 0;
    for (int r = lane; r < nefc; r += 64) {
      float jv = c->efc_jv[r], D = c->efc_D[r], x = c->efc_jar[r] + alpha * jv;
      int kind = c->efc_kind[r], st;
      if (kind == GRX_ROW_EQ) { g += D * x * jv; h += D * jv * jv; st = 1; }
      else if (kind == GRX_ROW_FRICTION) {
        float fl = c->efc_floss[r], Rf = fl / D;
        if (x <= -Rf) { g -= fl * jv; st = 3; } else if (x >= Rf) { g += fl * jv; st = 2; } else { g += D * x * jv; h += D * jv * jv; st = 1; }
      } else if (x < 0) { g += D * x * jv; h += D * jv * jv; st = 1; } else st = 0;
      if (want_same) dif |= (st != (c->efc_quad[r] & 3));
    }
    LV(gp) = g; LV(hp) = h; LV(difp) = dif;
  }
  float g = grx_reduce_sum(gp), h = grx_reduce_sum(hp);
  *d1 = q1 + alpha * q2 + g; *d2 = q2 + h;
  *same = want_same && (GRX_BALLOT(difp) == 0ull);
}

// H = M + J' diag(D_active) J  ->  c->A   (efc_jv is used as scratch for the masked D)
// Also returns J' f (the constraint force in joint space for the row forces of the last evaluation) in c->grad: on the
// matrix-core path it rides along as one extra output column of the same MFMA chain.
GRX_MEM void grx_hessian(const GrxModel* m, GrxCtx* c, int nefc, int lane_) {
  const int nv = GRX_NVC;
    // Hessian H = M + J' diag(D_active) J
  FOR_LANES {
    for (int r = lane; r < nefc; r += 64) {
      const int st = c->efc_quad[r] & 3;
      c->efc_jv[r] = (st == 1) ? c->efc_D[r] : 0.0f;   // efc_jv reused as scratch
      c->efc_quad[r] = st | (st << 4);                  // the Hessian now represents this row in state st (grx_hessian_update)
    }
  }
  WAVE_SYNC();
#if !defined(GRX_EMU)
  if (nv < 32 || (S::kFixed && S::NV <= 40)) {
    // matrix cores: [H | J'f] = J' [D J | f] as a chain of v_mfma_f32_32x32x2_f32 (exact f32, two constraint rows per instruction).
    // nv >= 32 (Adroit: 33): the chain forms the leading 32 x 32 block; the remaining rows / columns and J'f follow in a lane-per-dof pass.
    const int nvm = nv < 32 ? nv : 32;
    // operand maps: A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]; C: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5)
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.0f;
    const int idx = lane_ & 31, half = lane_ >> 5;
    const bool incol = idx < nvm;
    // Branch-free operand fetch (clamped addresses, selects instead of divergent paths), software-pipelined by hand over
    // four row pairs: 16 independent LDS reads (row info, second span, masked D, force), then 4 reads of the packed Jacobian, then 4 MFMAs.
    const bool isf = (idx == nv);   // the spare column carries J'f (only when nv < 32)
    for (int r0 = 0; r0 < nefc; r0 += 8) {
      int info[4], idb[4]; float dq[4], fr[4], v[4]; bool rowok[4], in[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int row = r0 + 2 * u + half;
        rowok[u] = row < nefc;
        const int rr = rowok[u] ? row : 0;
        info[u] = c->efc_row[rr]; idb[u] = S::kTwoSpan ? c->efc_id[rr] : 0; dq[u] = c->efc_jv[rr]; fr[u] = c->efc_force[rr];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int pos = grx_row_pos(info[u], idb[u], idx);
        in[u] = rowok[u] && incol && pos >= 0;
        v[u] = c->Jp[GRX_ROW_OFF(info[u]) + (in[u] ? pos : 0)];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const float a = in[u] ? v[u] : 0.0f;                                        // A[i = idx][k = row] = J[row][idx]
        const float b = in[u] ? v[u] * dq[u] : ((isf && rowok[u]) ? fr[u] : 0.0f);  // B[k = row][j = idx] = D J[row][idx]; column nv: f[row]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int i = (e & 3) + 8 * (e >> 2) + 4 * half;
      if (i < nvm) {
        if (incol) c->A[i * nv + idx] = c->M[i * nv + idx] + acc[e];
        else if (idx == nv) c->grad[i] = acc[e];
      }
    }
    if (nv >= 32) {   // rows / columns 32 .. nv-1 of H and the whole of J'f: lane j = dof j, one pass over the rows per extra dof
      if (lane_ < nv) {
        const int j = lane_;
        float gj = 0.0f;
        for (int r = 0; r < nefc; r++) {
          const int info = c->efc_row[r], idb = S::kTwoSpan ? c->efc_id[r] : 0, pj = grx_row_pos(info, idb, j);
          if (pj >= 0) gj += c->Jp[GRX_ROW_OFF(info) + pj] * c->efc_force[r];
        }
        c->grad[j] = gj;
        for (int i = 32; i < nv; i++) {
          float hij = 0.0f;
          for (int r = 0; r < nefc; r++) {
            const float dq = c->efc_jv[r];
            if (dq == 0.0f) continue;
            const int info = c->efc_row[r], idb = S::kTwoSpan ? c->efc_id[r] : 0, pi = grx_row_pos(info, idb, i), pj = grx_row_pos(info, idb, j);
            if (pi >= 0 && pj >= 0) hij += dq * c->Jp[GRX_ROW_OFF(info) + pi] * c->Jp[GRX_ROW_OFF(info) + pj];
          }
          const float v = c->M[i * nv + j] + hij;
          c->A[i * nv + j] = v; c->A[j * nv + i] = v;
        }
      }
    }
    __syncthreads();
  } else
#endif
  {
  FOR_LANES {
    const int li = lane >> 3, lj = lane & 7;
    for (int i0 = li; i0 < nv; i0 += 24)
      for (int j0 = lj; j0 < nv && j0 <= i0 + 16; j0 += 24) {
        // 3x3 register tile: rows i0, i0+8, i0+16 ; cols j0, j0+8, j0+16
        float acc[3][3];
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) acc[a][b] = 0.0f;
        const int i1 = i0 + 8, i2 = i0 + 16, j1 = j0 + 8, j2 = j0 + 16;
        const int vi1 = i1 < nv, vi2 = i2 < nv, vj1 = j1 < nv, vj2 = j2 < nv;
        for (int r = 0; r < nefc; r++) {
          const int info = c->efc_row[r], idb = c->efc_id[r], off = GRX_ROW_OFF(info);
#define GRX_JAT(dof) (grx_row_pos(info, idb, (dof)) >= 0 ? c->Jp[off + grx_row_pos(info, idb, (dof))] : 0.0f)
          float d = c->efc_jv[r];
          float a0 = GRX_JAT(i0) * d, a1 = vi1 ? GRX_JAT(i1) * d : 0.0f, a2 = vi2 ? GRX_JAT(i2) * d : 0.0f;
          float b0 = GRX_JAT(j0), b1 = vj1 ? GRX_JAT(j1) : 0.0f, b2 = vj2 ? GRX_JAT(j2) : 0.0f;
#undef GRX_JAT
          acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[0][2] += a0 * b2;
          acc[1][0] += a1 * b0; acc[1][1] += a1 * b1; acc[1][2] += a1 * b2;
          acc[2][0] += a2 * b0; acc[2][1] += a2 * b1; acc[2][2] += a2 * b2;
        }
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++) {
            int i = i0 + 8 * a, j = j0 + 8 * b;
            if (i < nv && j < nv && j <= i) { float v = c->M[i * nv + j] + acc[a][b]; c->A[i * nv + j] = v; c->A[j * nv + i] = v; }
          }
      }
    for (int i = lane; i < nv; i += 64) {
      float sacc = 0;
      for (int r = 0; r < nefc; r++) { const int info = c->efc_row[r], pos = grx_row_pos(info, c->efc_id[r], i); if (pos >= 0) sacc += c->Jp[GRX_ROW_OFF(info) + pos] * c->efc_force[r]; }
      c->grad[i] = sacc;
    }
  }
  WAVE_SYNC();
  }
}

// Incremental Hessian.  A row contributes to the problem through its state: force = -d (J a - aref) + kf, with d = D in the quadratic
// state, (d, kf) = (0, -+floss) for a saturated friction-loss row and (0, 0) when inactive; H = M + sum d J'J.  Between two Newton
// iterations of one substep only the rows whose state flipped change d: apply their rank-1 corrections to A instead of re-assembling H
// over all rows (one flip is the common case; the iterations after the first are what separates an expensive world from a cheap one).
// The GRADIENT is advanced the same way (the caller has put g_old + alpha H_old v into `gnew`: exact while no row changes state); a row
// that flipped adds J_r' (f_new - f_old-state(jar_new)) = J_r' (-(d_new - d_old) jar_new + (kf_new - kf_old)), all of it small near the
// solution.  (Round 3 formed the gradient as H a - qfrc_smooth - sum k J with k = D aref: terms of size D |aref| |J| ~ 400 cancelling to
// 1e-6 -- in fp32 a noise of 2e-5 on the puck's angular dof, whose Hessian entry is 6e-4: an acceleration error of 3e-2 rad/s^2 per
// substep, the whole FetchSlide rotation-velocity discrepancy; tools/emu_mixed.py, tools/emu_trace.py.)  Returns 0 when more than
// GRX_HUPD_MAX rows flipped (the caller re-assembles).  Uses c->ired (row list) and c->Mv (the row, expanded) as scratch.
#define GRX_HUPD_MAX 8
GRX_MEM int grx_hessian_update(const GrxModel* m, GrxCtx* c, int nefc, float* gnew, int lane_) {
  const int nv = GRX_NVC;
  int* list = c->ired;
  int nd = 0;
  for (int base = 0; base < nefc; base += 64) {
    GRX_LANEVAR_I(dirty);
    FOR_LANES { const int r = base + lane; const int q = r < nefc ? c->efc_quad[r] : 0; LV(dirty) = (r < nefc) && ((q & 3) != ((q >> 4) & 3)); }
    const unsigned long long bm = GRX_BALLOT(dirty);
    FOR_LANES { if (LV(dirty)) { const int k = nd + __builtin_popcountll(bm & ((1ull << lane) - 1ull)); if (k < GRX_HUPD_MAX) list[k] = base + lane; } }
    nd += __builtin_popcountll(bm);
  }
  WAVE_SYNC();
  if (nd > GRX_HUPD_MAX) return 0;
  for (int e = 0; e < nd; e++) {
    const int r = list[e];
    const int q = c->efc_quad[r], st = q & 3, hs = (q >> 4) & 3, info = c->efc_row[r], idb = S::kTwoSpan ? c->efc_id[r] : 0;
    const float D = c->efc_D[r];
    const float fl = (st >= 2 || hs >= 2) ? c->efc_floss[r] : 0.0f;
    const float dd = (st == 1 ? D : 0.0f) - (hs == 1 ? D : 0.0f);
    // change of the row force at the current point: -(d_new - d_old) jar + (kf_new - kf_old)
    const float df = -dd * c->efc_jar[r] + ((st == 2 ? -fl : (st == 3 ? fl : 0.0f)) - (hs == 2 ? -fl : (hs == 3 ? fl : 0.0f)));
    FOR_LANES { for (int i = lane; i < nv; i += 64) { const int pos = grx_row_pos(info, idb, i); c->Mv[i] = pos >= 0 ? c->Jp[GRX_ROW_OFF(info) + pos] : 0.0f; } }
    WAVE_SYNC();
    FOR_LANES {
      for (int i = lane; i < nv; i += 64) {
        const float vi = c->Mv[i];
        if (vi != 0.0f) {
          const float s_ = dd * vi;
          for (int j = 0; j < nv; j++) c->A[i * nv + j] += s_ * c->Mv[j];
          gnew[i] -= df * vi;   // gradient = M a - qfrc_smooth - J'f
        }
      }
    }
    WAVE_SYNC();
    LANE0 { c->efc_quad[r] = st | (st << 4); }
  }
  WAVE_SYNC();
  return 1;
}

// ------------------------------------------------------------------------------------------
// K10b noslip post-solver (MuJoCo option noslip_iterations; Adroit: assets/adroit_hand/adroit_assets.xml:3): projected Gauss-Seidel on the
// dual with the regulariser removed, over the friction-loss rows and the pairs of opposing pyramid edges of the frictional contacts (the
// oracle's solve_noslip restates the reference algorithm with an explicit A = J M^-1 J').  Here it is matrix-free, at wavefront level: the
// rows are visited one after the other, every dot product runs across the lanes (lane i = dof i):
//     t = M^-1 J_r'            (lane i: sum over the row's span of Minv[i][d] J_r[d])
//     A_rr = J_r . t ,  res_r = J_r . a - aref_r          (a = the acceleration implied by the current forces, kept in c->qacc)
//     f_r <- projected update ,  a += t * delta
// M^-1 is formed once per substep (LDL' of M, one right-hand side per lane).  Ends with M a in c->Ma (so that the caller's
// qfrc_constraint = M a - qfrc_smooth holds for the new forces).
// ------------------------------------------------------------------------------------------
GRX_MEM void grx_noslip(const GrxModel* m, GrxCtx* c, int nefc, int lane_) {
  const int nv = GRX_NVC, maxiter = m->noslip_iterations;
  const int ne = c->cnt[3], nf = c->cnt[4], ncon = c->cnt[0] < c->maxcon ? c->cnt[0] : c->maxcon;
  GRX_TICK(c, GRX_P_NEVAL);
  // ---- M^-1 into c->minv (= the Hessian's buffer: Newton is done with it).  Specialised shapes on the GPU: Gauss-Jordan in registers.
#if !defined(GRX_EMU)
  if (S::kFixed && S::NV > 0 && S::NV <= 40) grx_sym_inverse_reg<(S::NV > 0 && S::NV <= 40) ? S::NV : 1>(c->M, nv, c->minv, lane_);
  else
#endif
  {
    // in-place Gauss-Jordan in LDS (lane i = row i; the matrix is positive definite: no pivoting): step k eliminates column k from every other
    // row and turns it into the k-th column of the inverse, then row k is scaled -- the arithmetic of grx_sym_inverse_reg, through LDS
    FOR_LANES { for (int i = lane; i < nv * nv; i += 64) c->minv[i] = c->M[i]; }
    WAVE_SYNC();
    int bad = 0;
    for (int k = 0; k < nv; k++) {
      const float d = c->minv[k * nv + k];
      bad |= !(d > 0.0f);
      const float pinv = 1.0f / d;
      FOR_LANES {
        for (int i = lane; i < nv; i += 64) {
          if (i == k) continue;
          float* row = c->minv + i * nv; const float* piv = c->minv + k * nv;
          const float f = row[k] * pinv;
          for (int j = 0; j < nv; j++) if (j != k) row[j] = fmaf(-f, piv[j], row[j]);
          row[k] = -f;
        }
      }
      WAVE_SYNC();
      FOR_LANES { for (int j = lane; j < nv; j += 64) c->minv[k * nv + j] = (j == k) ? pinv : c->minv[k * nv + j] * pinv; }
      WAVE_SYNC();
    }
    if (bad) { LANE0 { c->cnt[2] |= GRX_ST_FACTOR; } }
  }
  GRX_SUBTICK(c, 17);
  const float scale = 1.0f / (m->meaninertia * (float)(nv > 1 ? nv : 1));
  float improvement0 = 0.0f;   // cost change of dropping the regulariser: 0.5 sum f^2 R (enters the first sweep's improvement)
  {
    GRX_LANEVAR(ip);
    FOR_LANES { float sacc = 0.0f; for (int r = lane; r < nefc; r += 64) { const float f = c->efc_force[r]; sacc += 0.5f * f * f / c->efc_D[r]; } LV(ip) = sacc; }
    improvement0 = grx_reduce_sum(ip);
  }
#if !defined(GRX_EMU)
  {
    // GPU: the sweep state lives in registers -- lane i holds a_i, lane r holds the r-th friction-loss row (dof, aref, bound, force, A_rr);
    // a row update is a handful of v_readlane broadcasts plus one LDS read of the M^-1 column, no barrier.  Same arithmetic, same order
    // as the plain version below (which the lane emulator runs).
    float a_l = lane_ < nv ? c->qacc[lane_] : 0.0f;
    int fr_d = 0; float fr_aref = 0.0f, fr_fl = 0.0f, fr_f = 0.0f, fr_arr = 1.0f, fr_rinv = 1.0f;
    if (lane_ < nf) {
      const int r = ne + lane_;
      fr_d = GRX_ROW_IDOF(c->efc_id[r]); fr_aref = c->efc_aref[r]; fr_fl = c->efc_floss[r]; fr_f = c->efc_force[r]; fr_arr = c->minv[fr_d * nv + fr_d];
      fr_rinv = 1.0f / fmaxf(GRX_MINVAL, fr_arr);
    }
    // Sweep-invariant part of a contact pair: t = M^-1 J' of its two rows (one word per lane each) and A00 / A01 / A11.  For the first KC pairs (sweep
    // order) they are formed once per substep and parked in the Newton scratch that is dead by now (grad, search, Mv, tmpv, efc_jar, efc_jv: contiguous);
    // a sweep then costs such a pair two LDS reads per lane and the two residual reductions instead of ~4 x len LDS reads and five reductions.
    float* const tc = c->grad;
    const int tstride = 2 * nv + 4;
    int KC = (int)(c->efc_force - c->grad) / tstride;
    if (KC > 24) KC = 24;
    {
      int pi = 0;
      for (int k = 0; k < ncon && pi < KC; k++) {
        const int r0 = c->con_efc[k], nr = c->con_nr[k];
        if (r0 < 0 || nr < 2) continue;
        for (int j = r0; j + 1 < r0 + nr && j + 1 < nefc && pi < KC; j += 2, pi++) {
          const int infoA = c->efc_row[j], idA = c->efc_id[j], infoB = c->efc_row[j + 1], idB = c->efc_id[j + 1];
          float ta = 0.0f, tb = 0.0f, ja = 0.0f, jb = 0.0f;
          if (lane_ < nv) {
            const float* mi = c->minv + lane_ * nv;
            const int offA = GRX_ROW_OFF(infoA), loA = GRX_ROW_LO(infoA), lenA = GRX_ROW_LEN(infoA), offB = GRX_ROW_OFF(infoB), loB = GRX_ROW_LO(infoB), lenB = GRX_ROW_LEN(infoB);
            for (int e = 0; e < lenA; e++) ta += c->Jp[offA + e] * mi[loA + e];
            for (int e = 0; e < lenB; e++) tb += c->Jp[offB + e] * mi[loB + e];
            if (S::kTwoSpan) {
              const int lo2A = GRX_ROWB_LO(idA), len2A = GRX_ROWB_LEN(idA), lo2B = GRX_ROWB_LO(idB), len2B = GRX_ROWB_LEN(idB);
              for (int e = 0; e < len2A; e++) ta += c->Jp[offA + lenA + e] * mi[lo2A + e];
              for (int e = 0; e < len2B; e++) tb += c->Jp[offB + lenB + e] * mi[lo2B + e];
            }
            const int pa = grx_row_pos(infoA, idA, lane_), pb = grx_row_pos(infoB, idB, lane_);
            ja = pa >= 0 ? c->Jp[offA + pa] : 0.0f; jb = pb >= 0 ? c->Jp[offB + pb] : 0.0f;
          }
          const float A00 = grx_reduce_sum(ja * ta), A01 = grx_reduce_sum(ja * tb), A11 = grx_reduce_sum(jb * tb);
          float* slot = tc + pi * tstride;
          if (lane_ < nv) { slot[lane_] = ta; slot[nv + lane_] = tb; }
          if (lane_ == 0) { slot[2 * nv] = A00; slot[2 * nv + 1] = A01; slot[2 * nv + 2] = A11; }
        }
      }
      __syncthreads();
    }
    for (int iter = 0; iter < maxiter; iter++) {
      float improvement = iter == 0 ? improvement0 : 0.0f;
      // one row after the other (Gauss-Seidel): the only LDS access of a row -- its column of M^-1 -- is fetched one row ahead, the division by A_rr
      // became a multiplication by the reciprocal formed with the row state (the plain version below divides: last-ulp difference)
      float col_next = (lane_ < nv && nf > 0) ? c->minv[lane_ * nv + __builtin_amdgcn_readlane(fr_d, 0)] : 0.0f;
      for (int r = 0; r < nf; r++) {
        const int d = __builtin_amdgcn_readlane(fr_d, r);
        const float col = col_next;
        const int dn = __builtin_amdgcn_readlane(fr_d, r + 1 < nf ? r + 1 : r);
        col_next = lane_ < nv ? c->minv[lane_ * nv + dn] : 0.0f;
        const float Arr = grx_readlane_f(fr_arr, r), rinv = grx_readlane_f(fr_rinv, r), res = grx_readlane_f(a_l, d) - grx_readlane_f(fr_aref, r), old = grx_readlane_f(fr_f, r),
                    fl = grx_readlane_f(fr_fl, r);
        float fn = old - res * rinv;
        fn = fn < -fl ? -fl : (fn > fl ? fl : fn);
        const float dl = fn - old;
        improvement -= 0.5f * dl * dl * Arr + dl * res;
        fr_f = (lane_ == r) ? fn : fr_f;
        a_l = fmaf(col, dl, a_l);
      }
      int pi = 0;
      for (int k = 0; k < ncon; k++) {
        const int r0 = c->con_efc[k], nr = c->con_nr[k];
        if (r0 < 0 || nr < 2) continue;
        for (int j = r0; j + 1 < r0 + nr && j + 1 < nefc; j += 2, pi++) {
          const int infoA = c->efc_row[j], idA = c->efc_id[j], infoB = c->efc_row[j + 1], idB = c->efc_id[j + 1];
          float ta = 0.0f, tb = 0.0f, ja = 0.0f, jb = 0.0f, A00, A01, A11;
          if (lane_ < nv) {
            const int offA = GRX_ROW_OFF(infoA), offB = GRX_ROW_OFF(infoB);
            const int pa = grx_row_pos(infoA, idA, lane_), pb = grx_row_pos(infoB, idB, lane_);
            ja = pa >= 0 ? c->Jp[offA + pa] : 0.0f; jb = pb >= 0 ? c->Jp[offB + pb] : 0.0f;
          }
          if (pi < KC) {   // parked above
            const float* slot = tc + pi * tstride;
            if (lane_ < nv) { ta = slot[lane_]; tb = slot[nv + lane_]; }
            A00 = slot[2 * nv]; A01 = slot[2 * nv + 1]; A11 = slot[2 * nv + 2];
          } else {
            if (lane_ < nv) {
              const float* mi = c->minv + lane_ * nv;
              const int offA = GRX_ROW_OFF(infoA), loA = GRX_ROW_LO(infoA), lenA = GRX_ROW_LEN(infoA), offB = GRX_ROW_OFF(infoB), loB = GRX_ROW_LO(infoB), lenB = GRX_ROW_LEN(infoB);
              for (int e = 0; e < lenA; e++) ta += c->Jp[offA + e] * mi[loA + e];
              for (int e = 0; e < lenB; e++) tb += c->Jp[offB + e] * mi[loB + e];
              if (S::kTwoSpan) {
                const int lo2A = GRX_ROWB_LO(idA), len2A = GRX_ROWB_LEN(idA), lo2B = GRX_ROWB_LO(idB), len2B = GRX_ROWB_LEN(idB);
                for (int e = 0; e < len2A; e++) ta += c->Jp[offA + lenA + e] * mi[lo2A + e];
                for (int e = 0; e < len2B; e++) tb += c->Jp[offB + lenB + e] * mi[lo2B + e];
              }
            }
            A00 = grx_reduce_sum(ja * ta); A01 = grx_reduce_sum(ja * tb); A11 = grx_reduce_sum(jb * tb);
          }
          const float res0 = grx_reduce_sum(ja * a_l) - c->efc_aref[j], res1 = grx_reduce_sum(jb * a_l) - c->efc_aref[j + 1];
          const float o0 = c->efc_force[j], o1 = c->efc_force[j + 1];
          const float bc0 = res0 - (A00 * o0 + A01 * o1), bc1 = res1 - (A01 * o0 + A11 * o1);
          const float mid = 0.5f * (o0 + o1), K1 = A00 + A11 - 2.0f * A01, K0 = mid * (A00 - A11) + bc0 - bc1;
          float f0, f1;
          if (K1 < GRX_MINVAL) { f0 = f1 = mid; }
          else {
            const float y = -K0 / K1;
            if (y < -mid) { f0 = 0.0f; f1 = 2.0f * mid; } else if (y > mid) { f0 = 2.0f * mid; f1 = 0.0f; } else { f0 = mid + y; f1 = mid - y; }
          }
          const float d0 = f0 - o0, d1 = f1 - o1;
          improvement -= 0.5f * (d0 * (A00 * d0 + A01 * d1) + d1 * (A01 * d0 + A11 * d1)) + d0 * res0 + d1 * res1;
          __syncthreads();
          if (lane_ == 0) { c->efc_force[j] = f0; c->efc_force[j + 1] = f1; }
          __syncthreads();
          a_l += ta * d0 + tb * d1;
        }
      }
      if (improvement * scale < m->noslip_tolerance) break;
    }
    __syncthreads();
    if (lane_ < nv) c->qacc[lane_] = a_l;
    if (lane_ < nf) c->efc_force[ne + lane_] = fr_f;
    __syncthreads();
  }
#else
  for (int iter = 0; iter < maxiter; iter++) {
    float improvement = iter == 0 ? improvement0 : 0.0f;
    // ---- dry friction: J = e_d, so t is a column of M^-1 and no reduction is needed
    for (int r = ne; r < ne + nf; r++) {
      const int d = GRX_ROW_IDOF(c->efc_id[r]);
      const float Arr = c->minv[d * nv + d], res = c->qacc[d] - c->efc_aref[r], old = c->efc_force[r], fl = c->efc_floss[r];
      float fn = old - res / fmaxf(GRX_MINVAL, Arr);
      fn = fn < -fl ? -fl : (fn > fl ? fl : fn);
      const float dl = fn - old;
      improvement -= 0.5f * dl * dl * Arr + dl * res;
      WAVE_SYNC();
      c->efc_force[r] = fn;
      FOR_LANES { if (lane < nv) c->qacc[lane] += c->minv[lane * nv + d] * dl; }
      WAVE_SYNC();
    }
    // ---- contact friction: pairs of opposing pyramid edges (their sum, the normal force, is kept)
    for (int k = 0; k < ncon; k++) {
      const int r0 = c->con_efc[k], nr = c->con_nr[k];
      if (r0 < 0 || nr < 2) continue;
      for (int j = r0; j + 1 < r0 + nr && j + 1 < nefc; j += 2) {
        GRX_LANEVAR(tA); GRX_LANEVAR(tB); GRX_LANEVAR(p00); GRX_LANEVAR(p01); GRX_LANEVAR(p11); GRX_LANEVAR(pr0); GRX_LANEVAR(pr1);
        const int infoA = c->efc_row[j], idA = c->efc_id[j], infoB = c->efc_row[j + 1], idB = c->efc_id[j + 1];
        FOR_LANES {
          float ta = 0.0f, tb = 0.0f, ja = 0.0f, jb = 0.0f, al = 0.0f;
          if (lane < nv) {
            const float* mi = c->minv + lane * nv;
            const int offA = GRX_ROW_OFF(infoA), loA = GRX_ROW_LO(infoA), lenA = GRX_ROW_LEN(infoA), offB = GRX_ROW_OFF(infoB), loB = GRX_ROW_LO(infoB), lenB = GRX_ROW_LEN(infoB);
            for (int e = 0; e < lenA; e++) ta += c->Jp[offA + e] * mi[loA + e];
            for (int e = 0; e < lenB; e++) tb += c->Jp[offB + e] * mi[loB + e];
            if (S::kTwoSpan) {
              const int lo2A = GRX_ROWB_LO(idA), len2A = GRX_ROWB_LEN(idA), lo2B = GRX_ROWB_LO(idB), len2B = GRX_ROWB_LEN(idB);
              for (int e = 0; e < len2A; e++) ta += c->Jp[offA + lenA + e] * mi[lo2A + e];
              for (int e = 0; e < len2B; e++) tb += c->Jp[offB + lenB + e] * mi[lo2B + e];
            }
            const int pa = grx_row_pos(infoA, idA, lane), pb = grx_row_pos(infoB, idB, lane);
            ja = pa >= 0 ? c->Jp[offA + pa] : 0.0f; jb = pb >= 0 ? c->Jp[offB + pb] : 0.0f;
            al = c->qacc[lane];
          }
          LV(tA) = ta; LV(tB) = tb; LV(p00) = ja * ta; LV(p01) = ja * tb; LV(p11) = jb * tb; LV(pr0) = ja * al; LV(pr1) = jb * al;
        }
        const float A00 = grx_reduce_sum(p00), A01 = grx_reduce_sum(p01), A11 = grx_reduce_sum(p11);
        const float res0 = grx_reduce_sum(pr0) - c->efc_aref[j], res1 = grx_reduce_sum(pr1) - c->efc_aref[j + 1];
        const float o0 = c->efc_force[j], o1 = c->efc_force[j + 1];
        const float bc0 = res0 - (A00 * o0 + A01 * o1), bc1 = res1 - (A01 * o0 + A11 * o1);
        const float mid = 0.5f * (o0 + o1), K1 = A00 + A11 - 2.0f * A01, K0 = mid * (A00 - A11) + bc0 - bc1;
        float f0, f1;
        if (K1 < GRX_MINVAL) { f0 = f1 = mid; }
        else {
          const float y = -K0 / K1;
          if (y < -mid) { f0 = 0.0f; f1 = 2.0f * mid; } else if (y > mid) { f0 = 2.0f * mid; f1 = 0.0f; } else { f0 = mid + y; f1 = mid - y; }
        }
        const float d0 = f0 - o0, d1 = f1 - o1;
        improvement -= 0.5f * (d0 * (A00 * d0 + A01 * d1) + d1 * (A01 * d0 + A11 * d1)) + d0 * res0 + d1 * res1;
        WAVE_SYNC();
        c->efc_force[j] = f0; c->efc_force[j + 1] = f1;
        FOR_LANES { if (lane < nv) c->qacc[lane] += LV(tA) * d0 + LV(tB) * d1; }
        WAVE_SYNC();
      }
    }
    if (improvement * scale < m->noslip_tolerance) break;
  }
#endif
  GRX_SUBTICK(c, 18);
  // M a for the caller (qfrc_constraint = M a - qfrc_smooth)
  FOR_LANES {
    for (int i = lane; i < nv; i += 64) {
      float sacc = 0.0f;
      for (int j = 0; j < nv; j++) sacc += c->M[i * nv + j] * c->qacc[j];
      c->Ma[i] = sacc;
    }
  }
  WAVE_SYNC();
}

// One more Newton step in the subspace of a DECOUPLED trailing free object (m->nfreeobj = 6, no active row links it to the robot), after Newton has converged by an
// exact full step.  H = M + J'DJ of a light body under a stiff contact carries the body's inertia at ~1e-4 of the contact's entries (the puck of FetchSlide:
// I = 5.8e-4 against D r^2 = 4.3), so the fp32 Hessian resolves the curvature of the body's weak mode -- rocking about the contact point -- to ~4e-4 and a full
// step of size 200 rad/s^2 leaves that mode 3e-2 rad/s^2 off the minimiser: the whole rotation-velocity discrepancy of the FetchSlide fixtures (tools/emu_mixed.py,
// tools/emu_trace.py).  The GRADIENT in that mode, taken from the rows (M a - qfrc_smooth - J'f with f from the carried J a - aref), has no such loss, so one more
// step with the same 6 x 6 Hessian block contracts the error by another 4e-4.  The step is only applied when it leaves every row of the object in its state (the
// block is then exact for the piece): cost one pass over the rows for 6 lanes and a 6 x 6 solve.
GRX_MEM void grx_refine_object_block(const GrxModel* m, GrxCtx* c, int nefc, int lane_) {
  GRX_FRESH_MODEL(m, c);
  const int nv = GRX_NVC, o0 = nv - 6;
  // J'f over the object's six dofs: one lane per row (the rows that touch the object are few), six wave sums
  GRX_LANEVAR(j0); GRX_LANEVAR(j1); GRX_LANEVAR(j2); GRX_LANEVAR(j3); GRX_LANEVAR(j4); GRX_LANEVAR(j5);
  FOR_LANES {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f, a4 = 0.0f, a5 = 0.0f;
    for (int r = lane; r < nefc; r += 64) {
      const int info = c->efc_row[r], idb = S::kTwoSpan ? c->efc_id[r] : 0;
      const int p0 = grx_row_pos(info, idb, o0), p5 = grx_row_pos(info, idb, o0 + 5);
      if (p0 < 0 && p5 < 0) continue;   // spans are contiguous dof ranges: a row that holds neither end of the object's six dofs holds none of them
      const float x = c->efc_jar[r], D = c->efc_D[r]; const int kind = c->efc_kind[r];
      float f;
      if (kind == GRX_ROW_EQ) f = -D * x;
      else if (kind == GRX_ROW_FRICTION) { const float fl = c->efc_floss[r], Rf = fl / D; f = (x <= -Rf) ? fl : ((x >= Rf) ? -fl : -D * x); }
      else f = (x < 0.0f) ? -D * x : 0.0f;
      const float* J = c->Jp + GRX_ROW_OFF(info);
      const int q1 = grx_row_pos(info, idb, o0 + 1), q2 = grx_row_pos(info, idb, o0 + 2), q3 = grx_row_pos(info, idb, o0 + 3), q4 = grx_row_pos(info, idb, o0 + 4);
      a0 += (p0 >= 0 ? J[p0] : 0.0f) * f; a1 += (q1 >= 0 ? J[q1] : 0.0f) * f; a2 += (q2 >= 0 ? J[q2] : 0.0f) * f;
      a3 += (q3 >= 0 ? J[q3] : 0.0f) * f; a4 += (q4 >= 0 ? J[q4] : 0.0f) * f; a5 += (p5 >= 0 ? J[p5] : 0.0f) * f;
    }
    LV(j0) = a0; LV(j1) = a1; LV(j2) = a2; LV(j3) = a3; LV(j4) = a4; LV(j5) = a5;
  }
  const float jf[6] = {grx_reduce_sum(j0), grx_reduce_sum(j1), grx_reduce_sum(j2), grx_reduce_sum(j3), grx_reduce_sum(j4), grx_reduce_sum(j5)};
  FOR_LANES { if (lane < 6) c->search[o0 + lane] = -(c->Ma[o0 + lane] - c->qfrc_smooth[o0 + lane] - GRX_SEL6(jf, lane)); }
  WAVE_SYNC();
#if defined(GRX_EMU)
  {
    static float blk[36];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) blk[6 * i + j] = c->A[(o0 + i) * nv + o0 + j];
    if (grx_sym_factor(blk, 6, lane_)) return;
    grx_sym_solve(blk, 6, c->search + o0, lane_);
  }
#else
  if (grx_sym_solve_reg<6>(c->A + o0 * nv + o0, nv, c->search + o0, lane_)) return;
#endif
  WAVE_SYNC();
  // the step must leave every row that touches the object in its state; jv of those rows
  const float d0 = c->search[o0], d1 = c->search[o0 + 1], d2 = c->search[o0 + 2], d3 = c->search[o0 + 3], d4 = c->search[o0 + 4], d5 = c->search[o0 + 5];
  GRX_LANEVAR_I(flipp);
  FOR_LANES {
    int flip = 0;
    for (int r = lane; r < nefc; r += 64) {
      const int info = c->efc_row[r], idb = S::kTwoSpan ? c->efc_id[r] : 0;
      const int p0 = grx_row_pos(info, idb, o0), p5 = grx_row_pos(info, idb, o0 + 5);
      float jv = 0.0f;
      if (p0 >= 0 || p5 >= 0) {
        const float* J = c->Jp + GRX_ROW_OFF(info);
        const int q1 = grx_row_pos(info, idb, o0 + 1), q2 = grx_row_pos(info, idb, o0 + 2), q3 = grx_row_pos(info, idb, o0 + 3), q4 = grx_row_pos(info, idb, o0 + 4);
        jv = (p0 >= 0 ? J[p0] : 0.0f) * d0 + (q1 >= 0 ? J[q1] : 0.0f) * d1 + (q2 >= 0 ? J[q2] : 0.0f) * d2 + (q3 >= 0 ? J[q3] : 0.0f) * d3 + (q4 >= 0 ? J[q4] : 0.0f) * d4 + (p5 >= 0 ? J[p5] : 0.0f) * d5;
        const float x0 = c->efc_jar[r], x1 = x0 + jv; const int kind = c->efc_kind[r];
        if (kind == GRX_ROW_FRICTION) { const float Rf = c->efc_floss[r] / c->efc_D[r]; flip |= ((x0 <= -Rf) != (x1 <= -Rf)) | ((x0 >= Rf) != (x1 >= Rf)); }
        else if (kind != GRX_ROW_EQ) flip |= ((x0 < 0.0f) != (x1 < 0.0f));
      }
      c->efc_jv[r] = jv;
    }
    LV(flipp) = flip;
  }
  if (GRX_BALLOT(flipp) != 0ull) return;
  WAVE_SYNC();
  FOR_LANES {
    if (lane < 6) {
      const int i = o0 + lane;
      const float* Mi = c->M + i * nv + o0;
      c->qacc[i] += c->search[i]; c->Ma[i] += Mi[0] * d0 + Mi[1] * d1 + Mi[2] * d2 + Mi[3] * d3 + Mi[4] * d4 + Mi[5] * d5;
    }
    for (int r = lane; r < nefc; r += 64) c->efc_jar[r] += c->efc_jv[r];
  }
  WAVE_SYNC();
}

// Constraint solve (Newton) + optional semi-implicit Euler step as ONE state machine, so that the three heavy
// primitives -- row evaluation, Hessian assembly and the register-resident linear solve -- each have a single call site
// in the kernel: the fused 20-substep loop has to stay inside the instruction cache.
//   phase 0: Newton iterations on the primal problem        (A = M + J' D J,      rhs = -gradient)
//   phase 2: no constraint rows at all                      (A = M,               rhs = qfrc_smooth)
//   phase 1: Euler velocity update with implicit damping    (A = M + h diag(B),   rhs = qfrc_smooth + qfrc_constraint)
GRX_MEM void grx_solve_integrate(const GrxModel* m, GrxCtx* c, int do_euler, int lane_) {
  GRX_FRESH_MODEL(m, c);
  const int nv = GRX_NVC; const float h = m->timestep;
  const int nefc = c->cnt[1];
  const float scale = 1.0f / (m->meaninertia * (float)(nv > 1 ? nv : 1));
  const int implicit_damp = (m->anydamp && m->eulerdamp);
  int phase = nefc ? 0 : 2, it = 0, done = 0, full_step = 0;
  float last_stepmax = 0.0f;   // largest component of the last accepted Newton step
  int exact_exit = 0, last_split = 0;   // converged by an exact full step (no row changed state) / the last linear solve ran on the decoupled robot | object blocks
  float alpha_prev = 0.0f;   // the step length accepted by the previous Newton iteration (gradient advance of the incremental path)
#if defined(GRX_EMU) && defined(GRX_EMU_STAGEHOOK)
  if (g_grx_solve_mode == 2 && nefc) phase = 1;
#endif
  GRX_COUNT(c, 30, 1);     // profiling build: constrained solves (substeps) of the step
#if defined(GRX_EMU)
  if (nefc) g_grx_newton_stats[0]++;
#endif
  GRX_COUNT(c, 31, nefc);  // ... and their constraint rows
  // the linear solve leaves c->A intact where it runs from registers (grx_sym_solve_full): the Hessian can then be corrected in place
  const int keepA = S::kIncrHess && (nv == 21 || nv == 14 || nv == 15 || nv == 24 || nv == 29 || nv == 30 || nv == 33 || nv == 36);
  // Newton starts from the previous solution (qacc_warmstart).  MuJoCo starts from the cheaper of (warmstart,
  // M^-1 qfrc_smooth); the minimiser of the strictly convex problem does not depend on the start, and skipping the
  // comparison saves one factorisation of M per substep.
  FOR_LANES { for (int i = lane; i < nv; i += 64) c->qacc[i] = c->qacc_ws[i]; }
  WAVE_SYNC();
  GRX_TICK(c, GRX_P_MSOLVE);
  for (;;) {
    float* rhs;
    if (phase == 0) {
      // After a step, M a and J a - aref are current (carried) and convergence has already been decided: the only thing the evaluation would
      // still produce are the row forces, which nothing reads after the solve unless the model has touch sensors.
      const int noslip = S::kNoslip && m->noslip_iterations > 0;
      const int skip_eval = done && it > 0 && (S::kFixed ? S::NT : m->ntouch) == 0 && !noslip;
      const int changed = skip_eval ? 0 : grx_newton_eval(m, c, c->qacc, nefc, it > 0, lane_);
      GRX_TICK(c, GRX_P_NEVAL);
      // a full Newton step (alpha = 1 accepted) that did not change any row state landed on the exact minimiser of the
      // piecewise-quadratic cost: no further iteration can move it beyond rounding
      if (it > 0 && full_step && !changed) { done = 1; exact_exit = 1; }
      if (done || it >= m->iterations) {   // MuJoCo's option iterations (default 100; the hand models: 20)
#ifndef GRX_NO_OBJ_REFINE
        // only models whose free object can rest on ONE contact of the general convex routine (puck, egg, pen: a flat-on-flat or line contact stands on a single point, the
        // object block of the Hessian has a weak rocking mode and fp32 resolves the full step to ~4e-4 of its size there); a box object stands on its corner contacts and
        // the refinement changes nothing at the 1e-7 level (tools/emu_tolerances.py with -DGRX_NO_OBJ_REFINE: FetchPush / PickAndPlace identical), at 3 % of the step
        const int weak_object = (S::kFixed ? S::kConvex : (m->nconvex != 0));
#if defined(GRX_EMU)
        if (exact_exit && last_split == 6 && keepA && weak_object) { g_grx_newton_stats[4]++; if (last_stepmax > GRX_OBJ_REFINE_MINSTEP) g_grx_newton_stats[5]++; }
#endif
        if (exact_exit && last_split == 6 && keepA && weak_object && last_stepmax > GRX_OBJ_REFINE_MINSTEP) grx_refine_object_block(m, c, nefc, lane_);
#endif
        if (noslip) grx_noslip(m, c, nefc, lane_);   // re-solves the friction forces without regularisation: new qacc, new M a
        // converged: at the minimiser the gradient M a - qfrc_smooth - J'f vanishes, so the joint-space constraint force
        // J'f of the final evaluation is M a - qfrc_smooth (to the solver's residual) -- no further pass over the rows
        FOR_LANES { for (int i = lane; i < nv; i += 64) { c->qfrc_constraint[i] = c->Ma[i] - c->qfrc_smooth[i]; c->qacc_ws[i] = c->qacc[i]; } }
        WAVE_SYNC();
        GRX_TICK(c, GRX_P_NFINAL);
#if defined(GRX_EMU) && defined(GRX_EMU_STAGEHOOK)
        if (g_grx_solve_mode == 1) break;
        if (do_euler) GRX_STAGE_HOOK(7);
#endif
        if (!do_euler) break;
        phase = 1;
        continue;
      }
      // Hessian of the current active set.  First iteration of a substep: assembled over all rows together with J'f of the current
      // row forces (one pass).  Later iterations: rank-1 corrections for the rows that flipped (grx_hessian_update), and the gradient is
      // ADVANCED along the accepted step instead of being re-formed: g_new = g_old + alpha H_old v is exact while no row changes state
      // (v = the step just taken, still in c->search; H_old = c->A, which the register solve leaves intact), the flipped rows add their
      // force change.  Every term is of the size of the gradient itself -- no cancellation of D |aref| |J|-sized numbers.
      int incremental = 0;
      if (it > 0 && keepA) {
        FOR_LANES {
          for (int i = lane; i < nv; i += 64) {
            float sacc = 0.0f;
#pragma unroll 8
            for (int j = 0; j < nv; j++) sacc += c->A[i * nv + j] * c->search[j];
            c->tmpv[i] = c->grad[i] + alpha_prev * sacc;
          }
        }
        WAVE_SYNC();
        incremental = grx_hessian_update(m, c, nefc, c->tmpv, lane_);
      }
#if defined(GRX_EMU)
      g_grx_newton_stats[incremental ? 3 : 2]++;
#endif
      if (!incremental) grx_hessian(m, c, nefc, lane_);
      GRX_TICK(c, GRX_P_NHESS);
      GRX_LANEVAR(gnp);
      if (incremental) {
        FOR_LANES {
          float part = 0;
          for (int i = lane; i < nv; i += 64) { const float sacc = c->tmpv[i]; c->grad[i] = sacc; c->search[i] = -sacc; part += sacc * sacc; }
          LV(gnp) = part;
        }
      } else {
        // gradient = M a - qfrc_smooth - J' f
        FOR_LANES {
          float part = 0;
          for (int i = lane; i < nv; i += 64) {
            const float sacc = c->Ma[i] - c->qfrc_smooth[i] - c->grad[i];
            c->grad[i] = sacc; c->search[i] = -sacc; part += sacc * sacc;
          }
          LV(gnp) = part;
        }
      }
      WAVE_SYNC();
      float gn = sqrtf(grx_reduce_sum(gnp));
      GRX_TICK(c, GRX_P_NGRAD);
#if defined(GRX_EMU) && defined(GRX_EMU_TRACE)
      if (getenv("GRX_TRACE_NEWTON")) fprintf(stderr, "NEWTON it %d gn %.6e scale*gn %.3e incremental %d\n", it, (double)gn, (double)(scale * gn), incremental);
#endif
      if (scale * gn < 1e-8f) { done = 1; continue; }
      rhs = c->search;
    } else if (phase == 1) {
      if (!implicit_damp) {
        FOR_LANES { for (int i = lane; i < nv; i += 64) c->tmpv[i] = c->qacc[i]; }
        WAVE_SYNC();
      } else {
        FOR_LANES {
          for (int i = lane; i < nv * nv; i += 64) c->A[i] = c->M[i];
          for (int i = lane; i < nv; i += 64) c->tmpv[i] = c->qfrc_smooth[i] + c->qfrc_constraint[i];
        }
        WAVE_SYNC();
        FOR_LANES { for (int i = lane; i < nv; i += 64) c->A[i * nv + i] += h * m->dof_damping[i]; }
        WAVE_SYNC();
      }
      rhs = c->tmpv;
    } else {
      FOR_LANES { for (int i = lane; i < nv * nv; i += 64) c->A[i] = c->M[i]; }
      WAVE_SYNC();
      rhs = c->qacc_smooth;
    }
    // ---- the one linear solve
    if (!(phase == 1 && !implicit_damp)) {
      // a trailing free object (m->nfreeobj = 6): M + h B is always block diagonal; the Hessian is while no active row links object and robot
      int nsplit = 0;
      if (m->nfreeobj == 6 && (nv == 21 || nv == 30)) {
        if (phase == 0) {
          GRX_LANEVAR_I(nzp);
          FOR_LANES {
            int nz = 0;
            for (int e = lane; e < (nv - 6) * 6; e += 64) { const int i = e / 6, j = nv - 6 + (e - 6 * i); nz |= (c->A[i * nv + j] != 0.0f) | (c->A[j * nv + i] != 0.0f); }
            LV(nzp) = nz;
          }
          nsplit = (GRX_BALLOT(nzp) == 0ull) ? 6 : 0;
        } else nsplit = 6;
      }
      if (phase == 0) last_split = nsplit;
      if (grx_sym_solve_full(c->A, nv, rhs, lane_, nsplit, phase != 0)) { LANE0 { c->cnt[2] |= GRX_ST_FACTOR; } }
    }
    if (phase == 0) {
      GRX_TICK(c, GRX_P_NFACTOR);
      // Mv, Jv, quadratic coefficients of the Gauss term along the direction
      GRX_LANEVAR(q1p); GRX_LANEVAR(q2p); GRX_LANEVAR(g0p);
      FOR_LANES {
        float p1 = 0, p2 = 0, p0 = 0;
        for (int i = lane; i < nv; i += 64) {
          p0 += c->grad[i] * c->search[i];
          float sacc = 0;
#pragma unroll 8
          for (int j = 0; j < nv; j++) sacc += c->M[i * nv + j] * c->search[j];
          c->Mv[i] = sacc;
          p1 += c->search[i] * (c->Ma[i] - c->qfrc_smooth[i]); p2 += c->search[i] * sacc;
        }
        for (int r = lane; r < nefc; r += 64) {
          c->efc_jv[r] = grx_row_dot(c, r, c->search);
        }
        LV(q1p) = p1; LV(q2p) = p2; LV(g0p) = p0;
      }
      WAVE_SYNC();
      const float q1 = grx_reduce_sum(q1p), q2 = grx_reduce_sum(q2p), dphi0 = grx_reduce_sum(g0p);
      // exact line search: root of the monotone piecewise-linear derivative, starting from the Newton step
      // phi'(0) = gradient . search (exact, from the pass above); the first row pass is at the full Newton step
      float d1, d2, alpha = 1.0f, lo = 0.0f, hi = 0.0f, dlo = dphi0, dhi = 0.0f;
      const float gtol = 1e-6f * fabsf(dphi0);
      int have_hi = 0;
      const int stop = !(dphi0 < 0);
      full_step = 0;
      for (int k = 0; k < GRX_LS_MAXIT + 1 && !stop; k++) {
        int same = 0;
        grx_ls_eval(c, nefc, alpha, q1, q2, &d1, &d2, k == 0, &same, lane_);
        // The search direction is the exact Newton step of the current active set: when no row changes state on [0, 1] the cost is
        // quadratic there and alpha = 1 is its minimiser, whatever rounding left in d1 (a difference of two numbers of size |phi'(0)|).
        if (k == 0 && same) { full_step = 2; break; }
        if (fabsf(d1) <= gtol) { full_step = (k == 0); break; }
        if (d1 < 0) { lo = alpha; dlo = d1; } else { hi = alpha; dhi = d1; have_hi = 1; }
        float na = alpha - d1 / d2;
        if (have_hi) { if (!(na > lo && na < hi)) na = lo + (hi - lo) * (dlo / (dlo - dhi)); if (!(na > lo && na < hi)) na = 0.5f * (lo + hi); }
        else if (!(na > lo)) na = 2.0f * alpha;
        alpha = na;
      }
#if defined(GRX_EMU) && defined(GRX_EMU_TRACE)
      if (getenv("GRX_TRACE_NEWTON")) fprintf(stderr, "   dphi0 %.6e stop %d alpha %.6f full_step %d\n", (double)dphi0, stop, (double)alpha, full_step);
#endif
      if (stop) { done = 1; continue; }  // not a descent direction any more: converged to rounding
      alpha_prev = alpha;
      GRX_LANEVAR(msp); GRX_LANEVAR(map_);
      FOR_LANES {
        float ms = 0, ma = 0;
        for (int i = lane; i < nv; i += 64) { float d = alpha * c->search[i]; float q = c->qacc[i] + d; c->qacc[i] = q; ms = fmaxf(ms, fabsf(d)); ma = fmaxf(ma, fabsf(q)); }
        {   // carry M a and J a - aref along the step: the next evaluation only re-derives row states and forces
          for (int i = lane; i < nv; i += 64) c->Ma[i] += alpha * c->Mv[i];
          for (int r = lane; r < nefc; r += 64) c->efc_jar[r] += alpha * c->efc_jv[r];
        }
        LV(msp) = ms; LV(map_) = ma;
      }
      WAVE_SYNC();
      const float stepmax = grx_reduce_max(msp), qmax = grx_reduce_max(map_);
      last_stepmax = stepmax;
#if defined(GRX_EMU) && defined(GRX_EMU_TRACE)
      if (getenv("GRX_TRACE_NEWTON")) { const int d_ = atoi(getenv("GRX_TRACE_NEWTON")); fprintf(stderr, "   stepmax %.6e qmax %.4e search[d] %.6e qacc[d] %.9e grad[d] %.6e\n", (double)stepmax, (double)qmax, (double)c->search[d_], (double)c->qacc[d_], (double)c->grad[d_]); }
#endif
      LANE0 { c->cnt[6] += 1; }
#if defined(GRX_EMU)
      g_grx_newton_stats[1]++;
#endif
      GRX_COUNT(c, 29, 1);   // profiling build: Newton iterations of the step
#ifdef GRX_LS_STATS
      { extern int g_ls_iters, g_ls_full; g_ls_iters++; g_ls_full += full_step; }
#endif
      GRX_TICK(c, GRX_P_NLS);
      it++;
      // converged when the accepted step is below the resolution we can hold in fp32 (quadratic convergence: the
      // step just applied is ~ the error BEFORE it, the error after it is far smaller)
      if (stepmax <= GRX_NEWTON_RTOL * qmax + GRX_NEWTON_ATOL) done = 1;
      // an exact full step (no row changes state on [0,1], decided with the very arithmetic the carried evaluation would repeat) lands on
      // the minimiser of the current piece and leaves every row in its state: converged
      if (full_step == 2) { done = 1; exact_exit = 1; }
    } else if (phase == 2) {
      FOR_LANES { for (int i = lane; i < nv; i += 64) { float q = c->qacc_smooth[i]; c->qacc[i] = q; c->qacc_ws[i] = q; c->qfrc_constraint[i] = 0; } }
      WAVE_SYNC();
      if (!do_euler) break;
      phase = 1;
    } else {
      // ---- semi-implicit Euler: velocities, then positions with the new velocities
      FOR_LANES { for (int i = lane; i < nv; i += 64) c->qvel[i] += h * c->tmpv[i]; }
      WAVE_SYNC();
      FOR_LANES {
        for (int j = lane; j < GRX_NJC; j += 64) {
          int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
          if (m->jnt_type[j] == 0) {
            for (int k = 0; k < 3; k++) c->qpos[qa + k] += h * c->qvel[da + k];
            float w[3] = {c->qvel[da + 3], c->qvel[da + 4], c->qvel[da + 5]};
            float n = sqrtf(dot3f(w, w));
            if (n > 1e-12f) {
              float sn, cs; sincosf(0.5f * h * n, &sn, &cs);
              float ri = sn / n, qr[4] = {cs, w[0] * ri, w[1] * ri, w[2] * ri}, q[4] = {c->qpos[qa + 3], c->qpos[qa + 4], c->qpos[qa + 5], c->qpos[qa + 6]}, qn[4];
              mulQuatf(qn, q, qr); normalize4f(qn);
              for (int k = 0; k < 4; k++) c->qpos[qa + 3 + k] = qn[k];
            }
          } else c->qpos[qa] += h * c->qvel[da];
        }
      }
      WAVE_SYNC();
      GRX_TICK(c, GRX_P_EULER);
      break;
    }
  }
}

// ------------------------------------------------------------------------------------------
// mj_forward (do_euler = 0) / mj_step (do_euler = 1) for one world
// ------------------------------------------------------------------------------------------
GRX_MEM void grx_forward_euler(const GrxModel* m, GrxCtx* c, int do_euler, int lane_) {
  GRX_TICK(c, GRX_P_OTHER);
  GRX_RNDINJ(6, (grx_rnd(c->qpos, m->nq), grx_rnd(c->qvel, m->nv), grx_rnd(c->qacc_ws, m->nv)));
  GRX_STAGE_HOOK(-1);
  grx_kinematics(m, c, lane_);
  GRX_STAGE_HOOK(0);
  GRX_RNDINJ(0, (grx_rnd(c->xpos, 3 * m->nbody), grx_rnd(c->xquat, 4 * m->nbody), grx_rnd(c->xmat, 9 * m->nbody), grx_rnd(c->sxpos, 3 * m->nsite), grx_rnd(c->sxmat, 9 * m->nsite), grx_rnd(c->janchor, 3 * m->njnt), grx_rnd(c->jaxis, 3 * m->njnt)));
  GRX_TICK(c, GRX_P_KIN);
  grx_inertia_cdof(m, c, lane_);
  GRX_STAGE_HOOK(1);
  GRX_RNDINJ(1, (grx_rnd(c->cinert, 10 * m->nbody), grx_rnd(c->cdof, 6 * m->nv), grx_rnd(c->M, m->nv * m->nv)));
  GRX_TICK(c, GRX_P_INERTIA);
  grx_collision(m, c, lane_);
  GRX_STAGE_HOOK(2);
  GRX_RNDINJ(2, (grx_rnd(c->con_dist, c->maxcon), grx_rnd(c->con_pos, 3 * c->maxcon), grx_rnd(c->con_frame, 3 * c->maxcon)));
  GRX_TICK(c, GRX_P_COLLIDE);
  grx_make_constraint(m, c, lane_);
  GRX_STAGE_HOOK(3);
  GRX_RNDINJ(3, (grx_rnd(c->Jp, c->jpool), grx_rnd(c->efc_D, c->maxefc), grx_rnd(c->efc_aref, c->maxefc)));
  GRX_TICK(c, GRX_P_CONSTR);
  grx_velocity(m, c, lane_);
  GRX_STAGE_HOOK(4);
  GRX_RNDINJ(4, (grx_rnd(c->qfrc_smooth, m->nv), grx_rnd(c->qacc_smooth, m->nv), grx_rnd(c->efc_aref, c->maxefc)));
#if defined(GRX_EMU) && defined(GRX_EMU_TRACE)
  grx_emu_trace(m, c, 0);   // test infrastructure (tools/emu_trace.py): contact list / rows of this pass
#endif
  grx_solve_integrate(m, c, do_euler, lane_);
  GRX_STAGE_HOOK(do_euler ? 5 : 6);
  GRX_RNDINJ(5, (grx_rnd(c->qpos, m->nq), grx_rnd(c->qvel, m->nv), grx_rnd(c->qacc_ws, m->nv)));
#if defined(GRX_EMU) && defined(GRX_EMU_TRACE)
  grx_emu_trace(m, c, 1);
#endif
}

// qpos <- q0 (+) hh * v  (mj_integratePos semantics: quaternion exponential for free joints), one lane per joint
GRX_MEM void grx_integrate_pos(const GrxModel* m, GrxCtx* c, const float* q0, const float* v, float hh, int lane_) {
  FOR_LANES {
    for (int j = lane; j < GRX_NJC; j += 64) {
      int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
      if (m->jnt_type[j] == 0) {
        for (int k = 0; k < 3; k++) c->qpos[qa + k] = q0[qa + k] + hh * v[da + k];
        float w[3] = {v[da + 3], v[da + 4], v[da + 5]};
        float n = sqrtf(dot3f(w, w));
        float q[4] = {q0[qa + 3], q0[qa + 4], q0[qa + 5], q0[qa + 6]};
        if (n > 1e-12f) {
          float sn, cs; sincosf(0.5f * hh * n, &sn, &cs);
          float ri = sn / n, qr[4] = {cs, w[0] * ri, w[1] * ri, w[2] * ri}, qn[4];
          mulQuatf(qn, q, qr); normalize4f(qn);
          for (int k = 0; k < 4; k++) q[k] = qn[k];
        }
        for (int k = 0; k < 4; k++) c->qpos[qa + 3 + k] = q[k];
      } else c->qpos[qa] = q0[qa] + hh * v[da];
    }
  }
  WAVE_SYNC();
}

// Runge-Kutta 4 (mj_RungeKutta [3P]).  Call after the forward pass of stage `stage` (0..3): records the
// stage derivative and moves the state to the next stage point (stages 0..2) or to the end of the step (stage 3).
GRX_MEM void grx_rk4_after_forward(const GrxModel* m, GrxCtx* c, int stage, int lane_) {
  const int nv = GRX_NVC; const float h = m->timestep;
  FOR_LANES {
    if (stage == 0) {
      for (int i = lane; i < GRX_NQC; i += 64) c->rk_q0[i] = c->qpos[i];
      for (int i = lane; i < nv; i += 64) c->rk_v0[i] = c->qvel[i];
    }
    for (int i = lane; i < nv; i += 64) { c->rk_Fv[stage * nv + i] = c->qvel[i]; c->rk_Fa[stage * nv + i] = c->qacc[i]; }
  }
  WAVE_SYNC();
  const float hh = (stage < 2) ? 0.5f * h : h;
  FOR_LANES {
    for (int i = lane; i < nv; i += 64) {
      float dv, da;
      if (stage < 3) { dv = c->rk_Fv[stage * nv + i]; da = c->rk_Fa[stage * nv + i]; }
      else {
        dv = (c->rk_Fv[i] + 2.0f * c->rk_Fv[nv + i] + 2.0f * c->rk_Fv[2 * nv + i] + c->rk_Fv[3 * nv + i]) * (1.0f / 6.0f);
        da = (c->rk_Fa[i] + 2.0f * c->rk_Fa[nv + i] + 2.0f * c->rk_Fa[2 * nv + i] + c->rk_Fa[3 * nv + i]) * (1.0f / 6.0f);
      }
      c->tmpv[i] = dv;
      c->qvel[i] = c->rk_v0[i] + hh * da;
    }
  }
  WAVE_SYNC();
  grx_integrate_pos(m, c, c->rk_q0, c->tmpv, hh, lane_);
}

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
