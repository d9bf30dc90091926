// grx_kernels.hip -- gfx950 kernels + the C ABI declared in include/grx_capi.h.
//
// Launch geometry: one 64-lane wavefront (= one workgroup) per world, dynamic LDS holds the
// world's whole working set (grx_ctx_words), so the 20 fused substeps of an env.step() read
// and write HBM exactly once.  Worlds are independent: no inter-workgroup communication.
//
// Build: this ONE source is compiled either as a single translation unit (no GRX_TU_* macro: the profiling variant) or, for the product
// library, four times in parallel with -DGRX_TU_FETCH / -DGRX_TU_HAND / -DGRX_TU_POINT / -DGRX_TU_ADROIT / -DGRX_TU_KITCHEN / -DGRX_TU_API (one family of kernel
// instantiations each; __graft_entry__.build links the objects).  Every unit has its own copy of the constant-memory model descriptors
// (g_grx_models is static): grx_model_create uploads a descriptor to each of them through grx_tu_*_prepare.
#if !defined(GRX_TU_FETCH) && !defined(GRX_TU_HAND) && !defined(GRX_TU_POINT) && !defined(GRX_TU_ADROIT) && !defined(GRX_TU_KITCHEN) && !defined(GRX_TU_API)
#define GRX_TU_FETCH 1
#define GRX_TU_HAND 1
#define GRX_TU_POINT 1
#define GRX_TU_ADROIT 1
#define GRX_TU_KITCHEN 1
#define GRX_TU_API 1
#endif
// The guessed support vertices of persistent hull contacts (grx_engine.h, grx_mesh_support) need a per-world HBM row, which only the Fetch buffers carry (hullcache): the
// code is compiled into the Fetch kernels only -- in the hand / kitchen kernels it would be dead weight in a register-starved routine (measured: -1 ... -2.5 %).
#if GRX_TU_FETCH && !defined(GRX_NO_HULL_HINTS)
#define GRX_HULL_HINTS 1
#endif
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/grx_capi.h"
#include "grx_fetch_task.h"
#include "grx_point_task.h"
#include "grx_hand_task.h"
#include "grx_adroit_task.h"
#include "grx_kitchen_task.h"
#include "grx_host_model.h"

static_assert(sizeof(grx_fetch_task) == sizeof(GrxFetchTask), "grx_fetch_task must mirror GrxFetchTask");
static_assert(sizeof(grx_fetch_buffers) == sizeof(GrxFetchBuffers), "grx_fetch_buffers must mirror GrxFetchBuffers");
static_assert(sizeof(grx_point_task) == sizeof(GrxPointTask), "grx_point_task must mirror GrxPointTask");
static_assert(sizeof(grx_point_buffers) == sizeof(GrxPointBuffers), "grx_point_buffers must mirror GrxPointBuffers");
static_assert(sizeof(grx_hand_task) == sizeof(GrxHandTask), "grx_hand_task must mirror GrxHandTask");
static_assert(sizeof(grx_hand_buffers) == sizeof(GrxHandBuffers), "grx_hand_buffers must mirror GrxHandBuffers");
static_assert(sizeof(grx_adroit_task) == sizeof(GrxAdroitTask), "grx_adroit_task must mirror GrxAdroitTask");
static_assert(sizeof(grx_adroit_buffers) == sizeof(GrxAdroitBuffers), "grx_adroit_buffers must mirror GrxAdroitBuffers");
static_assert(sizeof(grx_kitchen_task) == sizeof(GrxKitchenTask), "grx_kitchen_task must mirror GrxKitchenTask");
static_assert(sizeof(grx_kitchen_buffers) == sizeof(GrxKitchenBuffers), "grx_kitchen_buffers must mirror GrxKitchenBuffers");

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void grx_load_world(const GrxModel& m, const GrxFetchBuffers& b, GrxCtx& c, int w, float* lds, int words, int lane_) {
  for (int i = lane_; i < words; i += 64) lds[i] = 0.0f;  // also zeroes the structurally-zero part of M
  __syncthreads();
  for (int i = lane_; i < m.nq; i += 64) c.qpos[i] = b.qpos[(size_t)w * m.nq + i];
  for (int i = lane_; i < m.nv; i += 64) { c.qvel[i] = b.qvel[(size_t)w * m.nv + i]; c.qacc_ws[i] = b.qacc_ws[(size_t)w * m.nv + i]; }
  for (int i = lane_; i < 7 * m.nmocap; i += 64) {
    int k = i / 7, e = i - 7 * k; float v = b.mocap[(size_t)w * 7 * m.nmocap + i];
    if (e < 3) c.mocap_pos[3 * k + e] = v; else c.mocap_quat[4 * k + e - 3] = v;
  }
  __syncthreads();
}

// status word: bits 0-15 = GRX_ST_* flags of this launch, bits 16-31 = the same flags OR-accumulated over every launch since the host
// last cleared the buffer (sticky: a capacity overflow in step 17 is still visible after step 50)
__device__ __forceinline__ int grx_status_word(int old, int now) { now &= 15; return now | ((((old >> 16) | now) & 0xFFFF) << 16); }   // GRX_ST_SOFT and above are internal

// ---- the overflow lane (include/grx_capi.h, grx_overflow_lane)
// both kernels, before the simulation: the fast kernel may hand the world over at the first overflowing substep (grx_lane_claim, csrc/grx_engine.h), both watch the soft thresholds
__device__ __forceinline__ void grx_lane_setup(const GrxLane& L, GrxCtx& c, int w, bool stepping) {
  c.bail = (stepping && L.entry_count != nullptr) ? 1 : 0;
  if (c.bail) { c.lane_entry_count = L.entry_count; c.lane_entry_list = L.entry_list; c.lane_entry_cap = L.entry_cap; c.lane_world = w; c.lane_ready = L.ready; c.lane_ready_cap = L.ready_cap; }
  if (stepping && (L.list != nullptr || L.entry_count != nullptr)) { c.soft_maxefc = L.soft_maxefc; c.soft_jpool = L.soft_jpool; c.soft_maxcon = L.soft_maxcon; }
}
// fast kernel, after the simulation: true = the world claimed a re-run on the large tables: the caller returns WITHOUT writing anything of it
__device__ __forceinline__ bool grx_lane_overflowed(const GrxCtx& c) { return c.bail == 2; }
// append w to the lane of the next step (both kernels); a full list (next_cap: the grid of the next step's launch) leaves the world on the fast kernel
__device__ __forceinline__ void grx_lane_append(const GrxLane& L, int w) {
  const int idx = atomicAdd(L.next_count, 1);
  if (idx < L.next_cap) { L.next_list[idx] = w; L.next_flags[w] = 1; }
}
// fast kernel, after a step that did NOT overflow but came within the soft thresholds of a capacity: the result is committed as usual and the world moves to the
// lane for the next steps -- before it can overflow, so that entering the lane costs no serialised re-run
__device__ __forceinline__ void grx_lane_join(const GrxLane& L, const GrxCtx& c, int w, int lane_) {
  if (L.entry_count == nullptr || L.next_list == nullptr || lane_ != 0 || !(c.cnt[2] & GRX_ST_SOFT)) return;
  L.ttl[w] = (signed char)L.ttl_init;
  grx_lane_append(L, w);
}
// large-table kernel: the world's ticket (it stays in the lane while it is within the soft thresholds, and ttl_init steps longer); st < 0: the world was not part of this
// step (masked out: it waits for its reset) and keeps its place
__device__ __forceinline__ void grx_lane_ticket(const GrxLane& L, int st, int w, int lane_) {   // st: the world's status flags of this step, -1 = it was not stepped (by value: taking the context's address would keep the whole GrxCtx in scratch memory)
  if (L.list == nullptr || lane_ != 0) return;
  int t = L.ttl[w];
  if (st >= 0) { t = (st & GRX_ST_SOFT) ? L.ttl_init : (t > 0 ? t - 1 : 0); L.ttl[w] = (signed char)t; }
  else if (t <= 0) t = 1;
  if (t > 0) grx_lane_append(L, w);
}
// ---- entrants without the serialised re-run (include/grx_capi.h, grx_overflow_lane.ready / progress / poll_*).  The worlds that overflow are the heaviest of the batch and
// their re-run used to start when the fast launch had ENDED (hand + touch: 2 ms in 60 % of the steps, a hand jammed into the door 5 - 9 ms).  The standing lane launch now
// carries poll_grid extra workgroups; workgroup p sleeps until entry p of THIS step's entry list is published (ready[p] == 1), claims it (-> 2) and steps the world on the
// large tables while the fast launch is still running.  It gives up when every workgroup of the fast launch has ended (progress == progress_total) or after a bounded number
// of polls; whatever is unclaimed then is taken by the entry launch behind the fast kernel, as before.  Nothing waits for anything that is not already submitted.
__device__ __forceinline__ void grx_lane_progress(const GrxLane& L) { if (L.progress && threadIdx.x == 0) atomicAdd(L.progress, 1); }   // fast kernel: this workgroup has ended
// entry launch (list == the step's entry list): 1 = entry e was taken by a polling workgroup
__device__ __forceinline__ int grx_lane_taken(const GrxLane& L, int e) {
  if (!L.ready || L.poll_grid != 0 || e >= L.ready_cap) return 0;
  int r = 0;
  if (threadIdx.x == 0) r = atomicCAS(L.ready + e, 1, 2) != 1;
  return __builtin_amdgcn_readfirstlane(r);
}
// polling workgroup p of the standing launch: the world to step, or -1
__device__ __forceinline__ int grx_lane_poll(const GrxLane& L, int p) {
  if (!L.ready || p >= L.ready_cap) return -1;
  int w = -1;
  if (threadIdx.x == 0) {
    for (int it = 0; it < 40000; it++) {      // bounded: ~40000 x 2 us
      int r = __hip_atomic_load(L.ready + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (r == 0 && __hip_atomic_load(L.progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= L.progress_total)
        r = __hip_atomic_load(L.ready + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the fast launch has ended: one last look
      else if (r == 0) { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); continue; }
      if (r == 1 && atomicCAS(L.ready + p, 1, 2) == 1) { __threadfence(); w = ((volatile const int*)L.poll_list)[p]; }
      break;
    }
  }
  return __builtin_amdgcn_readfirstlane(w);
}
__device__ __forceinline__ void grx_store_world(const GrxModel& m, const GrxFetchTask& t, const GrxFetchBuffers& b, GrxCtx& c, int w, int lane_, int keep_outcome = 0) {
  for (int i = lane_; i < m.nq; i += 64) b.qpos[(size_t)w * m.nq + i] = c.qpos[i];
  for (int i = lane_; i < m.nv; i += 64) { b.qvel[(size_t)w * m.nv + i] = c.qvel[i]; b.qacc_ws[(size_t)w * m.nv + i] = c.qacc_ws[i]; }
  for (int i = lane_; i < 7 * m.nmocap; i += 64) {
    int k = i / 7, e = i - 7 * k;
    b.mocap[(size_t)w * 7 * m.nmocap + i] = (e < 3) ? c.mocap_pos[3 * k + e] : c.mocap_quat[4 * k + e - 3];
  }
  if (lane_ == 0) {
    const float* ag = b.achieved + (size_t)w * 3;
    const double d = grx_goal_distance3(ag, b.goal + (size_t)w * 3);
    if (!keep_outcome) {
      b.reward[w] = grx_fetch_reward(d, t.distance_threshold, t.sparse_reward);
      b.success[w] = (d < t.distance_threshold) ? 1 : 0;
    }
    b.status[w] = grx_status_word(b.status[w], c.cnt[2]);
    if (b.packed) {   // [obs | achieved | desired | reward | success] row for the cross-rank gather: no pack kernels on the host side
      float* row = b.packed + (size_t)w * (t.obs_dim + 8);
      const float* ob = b.obs + (size_t)w * t.obs_dim;
      for (int k = 0; k < t.obs_dim; k++) row[k] = ob[k];
      for (int k = 0; k < 3; k++) { row[t.obs_dim + k] = ag[k]; row[t.obs_dim + 3 + k] = b.goal[(size_t)w * 3 + k]; }
      row[t.obs_dim + 6] = b.reward[w]; row[t.obs_dim + 7] = b.success[w] ? 1.0f : 0.0f;
    }
  }
}

#ifdef GRX_PROFILE
__device__ long long g_grx_prof[GRX_NPROF];
extern "C" int grx_profile_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_grx_prof), sizeof(long long) * GRX_NPROF); }
extern "C" int grx_profile_reset() { long long z[GRX_NPROF] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_grx_prof), z, sizeof(z)); }
// per-world stage cycles of the last Fetch step launch (first 4096 worlds): which stages make a slow world slow (tools/straggler_probe.py)
__device__ int g_grx_world_prof[4096 * GRX_NPROF];
extern "C" int grx_profile_world_stages(int* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_grx_world_prof), sizeof(int) * GRX_NPROF * n); }
#endif
#if defined(GRX_PROFILE) || defined(GRX_WORLD_SPAN)
// per-world start / end timestamps of the last Fetch step launch (wall_clock64: one clock for the whole device): load balance across worlds.  -DGRX_WORLD_SPAN alone
// records them in a build that is otherwise the product build (same LDS footprint and occupancy: tools/span_probe.py)
__device__ long long g_grx_world_span[2 * 16384];
extern "C" int grx_profile_world_spans(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_grx_world_span), sizeof(long long) * 2 * n); }
#endif

// World <-> workgroup mapping.  The dispatcher deals workgroups round-robin to the 8 XCDs, each with its own L2, so with w = blockIdx.x
// the rows of neighbouring worlds (88-B qpos rows, 100-B obs rows, 4-B reward / flag entries: several worlds per 128-B line) are
// fetched by up to 8 L2s and written back as 8 partial lines.  The grid is rounded up to a multiple of 8 and XCD k takes the
// k-th contiguous slice of the worlds, so a line is read and merged in one L2 (rocprofv3 FETCH_SIZE / WRITE_SIZE: profiles/).
static inline unsigned grx_grid_for(int n_worlds) { return (unsigned)((n_worlds + 7) & ~7); }
#define GRX_LANE_GRID 64u   // default workgroups of a large-table launch of the overflow lane (grx_overflow_lane.grid): they walk the compacted list
static __device__ __forceinline__ int grx_world_of_block() { return (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)); }
// the same index re-derived after the substep loop from the (architected) workgroup id: the epilogue's addresses are then computed
// there instead of being kept -- as 64-bit VGPR pairs spilled to scratch -- across the whole simulation
static __device__ __forceinline__ unsigned grx_block_late() { unsigned bx = blockIdx.x; asm volatile("" : "+s"(bx)); return bx; }
// (split step: `parts` workgroups per world, workgroup index = part * G + slot with G = gridDim.x / parts)
static __device__ __forceinline__ int grx_world_of_slot_late(const int* order, int parts) {
  unsigned bx = blockIdx.x; asm volatile("" : "+s"(bx));
  const unsigned G = parts > 1 ? gridDim.x / (unsigned)parts : gridDim.x, slot = parts > 1 ? bx % G : bx;
  return order ? order[slot] : (int)((slot & 7u) * (G >> 3) + (slot >> 3));
}
static __device__ __forceinline__ int grx_world_of_block_late() {
  unsigned bx = blockIdx.x; asm volatile("" : "+s"(bx));
  return (int)((bx & 7u) * (gridDim.x >> 3) + (bx >> 3));
}

// Model shapes.  A step kernel specialised for a shape has all layout dims as compile-time constants: the LDS carve folds
// into immediate offsets of the ds_read/ds_write instructions (no address arithmetic, no pointer SGPRs) and the loops over
// dofs / bodies / joints unroll.  GrxShapeAny is the generic kernel (dims read from the model at run time).
template <class S> static __device__ __forceinline__ GrxDims grx_shape_dims(const GrxModel& m) {
  if (S::kFixed) return GrxDims{S::NQ, S::NV, S::NU, S::NB, S::NJ, S::NG, S::NS, S::NM, S::NF, S::INTEG, S::ME, S::JP, S::NT, S::MC, S::NMESH, S::NSHIFT, S::NOSLIP};
  return grx_dims_of(&m);
}
#ifndef GRX_MATCH_ANY_MESH
#define GRX_MATCH_ANY_MESH 0   // 1 (experiments only): a shape compiled without the hull routine also serves a model that has hull pairs (they are then never collided)
#endif
template <class S> static bool grx_shape_matches(const GrxModel& g) {
  return g.nq == S::NQ && g.nv == S::NV && g.nu == S::NU && g.nbody == S::NB && g.njnt == S::NJ && g.ngeom == S::NG && g.nsite == S::NS &&
         g.nmocap == S::NM && g.nfric == S::NF && g.integrator == S::INTEG && g.maxefc == S::ME && g.jpool == S::JP && g.ntouch == S::NT && g.maxcon == S::MC && (S::kTwoSpan || !g.twospan) && (S::kConvex == (g.nconvex != 0)) && (GRX_MATCH_ANY_MESH || S::kHullFilter == (g.nmeshpair != 0)) && (S::kShift == (g.nshift != 0)) && (S::kNoslip == (g.noslip_iterations > 0)) && (S::NF != 24 || g.handtree);
}
// last template argument: bit 0 = general convex routine for primitive pairs (ellipsoid / cylinder), bit 1 = hull-vs-convex pairs (every model with
// collidable mesh geoms next to boxes / other meshes: all Fetch and Shadow-hand models)
#ifndef GRX_FETCH_ME       // row / Jacobian-pool capacities the Fetch kernels are specialised for (envs/fetch.py FETCH_CAPACITY)
#define GRX_FETCH_ME 144
#define GRX_FETCH_JP 2032     // round 5: 1 984 -> 2 032 words paid for by 32 -> 28 contacts (observed maximum 26): the same 16 LDS granules, half as many worlds exceed the fast tables (profiles/demand_r05_fetch.txt)
#endif
#ifndef GRX_FETCH_MC
#define GRX_FETCH_MC 28
#endif
#ifndef GRX_FETCH_PICK_FLAGS   // experiments (tools/ab_fetch_3wave.sh): 0 = the FetchPickAndPlace kernels WITHOUT the hull-pair routine -- an upper bound for a fast kernel that hands hull worlds off (wrong physics for them)
#define GRX_FETCH_PICK_FLAGS 2
#endif
typedef GrxShape<22, 21, 2, 16, 16, 20, 3, 1, 0, 0, GRX_FETCH_ME, GRX_FETCH_JP, 0, GRX_FETCH_MC, 0, GRX_FETCH_PICK_FLAGS> GrxShapeFetchPick;   // FetchPickAndPlace (arm + gripper actuators + object)
typedef GrxShape<22, 21, 0, 16, 16, 20, 3, 1, 0, 0, GRX_FETCH_ME, GRX_FETCH_JP, 0, GRX_FETCH_MC, 0, 2> GrxShapeFetchObject; // FetchPush (arm + object)
typedef GrxShape<15, 15, 0, 15, 15, 19, 2, 1, 0, 0, GRX_FETCH_ME, GRX_FETCH_JP, 0, GRX_FETCH_MC, 0, 2> GrxShapeFetchArm;    // FetchReach (arm only)
typedef GrxShape<22, 21, 0, 16, 16, 20, 3, 1, 0, 0, GRX_FETCH_ME, GRX_FETCH_JP, 0, GRX_FETCH_MC, 0, 3> GrxShapeFetchPuck; // FetchSlide (arm + cylinder puck: convex narrow phase)
// the SAME models with the tables of the overflow lane (core.RERUN_CAPACITY): only the lane kernels are instantiated for them (BASELINE configs 2 / 3 / 5a; the other
// models' lanes run on the generic kernel, 2-3 x slower per world)
typedef GrxShape<22, 21, 2, 16, 16, 20, 3, 1, 0, 0, 256, 4080, 0, 64, 0, 2> GrxShapeFetchPickLane;
// Round 6: the FAST FetchPickAndPlace step kernel.  No hull routine (flag 32: GrxShape::kHandoff) -> 168 VGPRs, a third wave per SIMD; 96 rows / 1 024 pool words / 24 contacts ->
// 15.2 KB of LDS, ten worlds per CU.  A world in which a hull pair passes the bounding-box filter (8 % of the worlds of a stationary batch, profiles/hull_share_r06_fetch.txt) or
// that exceeds a table is handed off MID-STEP (grx_lane_handoff) to the standing lane, which runs GrxShapeFetchPick -- today's kernel -- as the lane kernel; what exceeds
// THAT kernel's tables goes on to GrxShapeFetchPickLane.  Step kernel only: forward / reset launches of the model use GrxShapeFetchPick.
#ifndef GRX_FETCH_FAST_ME
#define GRX_FETCH_FAST_ME 96
#define GRX_FETCH_FAST_JP 1024
#define GRX_FETCH_FAST_MC 24
#endif
typedef GrxShape<22, 21, 2, 16, 16, 20, 3, 1, 0, 0, GRX_FETCH_FAST_ME, GRX_FETCH_FAST_JP, 0, GRX_FETCH_FAST_MC, 0, 32> GrxShapeFetchPickFast;
// ant.xml + maze walls (RK4): the geom count depends on the maze layout (Large / Medium / Open / UMaze of maze/maps.py)
typedef GrxShape<15, 14, 8, 10, 9, 76, 1, 0, 0, 1, 64, 512, 0, 16> GrxShapeAntLarge;
typedef GrxShape<15, 14, 8, 10, 9, 52, 1, 0, 0, 1, 64, 512, 0, 16> GrxShapeAntMedium;
typedef GrxShape<15, 14, 8, 10, 9, 34, 1, 0, 0, 1, 64, 512, 0, 16> GrxShapeAntOpen;
typedef GrxShape<15, 14, 8, 10, 9, 32, 1, 0, 0, 1, 64, 512, 0, 16> GrxShapeAntUMaze;
typedef GrxShape<24, 24, 20, 25, 24, 23, 5, 0, 24, 0, 96, 512, 0, 16, 1, 2> GrxShapeHandReach;  // Shadow hand, reach.xml: 24 hinges, 24 friction-loss dofs, the 5 fingertip sites, 16 contact slots
typedef GrxShape<31, 30, 20, 26, 25, 24, 0, 0, 24, 0, 112, 1024, 0, 24, 1, 2> GrxShapeHandBlock;  // Shadow hand + free block (manipulate_block.xml without the visual-only target body)
typedef GrxShape<31, 30, 20, 26, 25, 24, 0, 0, 24, 0, 112, 1024, 0, 24, 1, 3> GrxShapeHandEgg;     // manipulate_egg.xml: the ellipsoid goes through the convex narrow phase
typedef GrxShape<31, 30, 20, 26, 25, 24, 0, 0, 24, 0, 112, 928, 92, 24, 1, 3> GrxShapeHandEggTouch;
typedef GrxShape<31, 30, 20, 26, 25, 24, 0, 0, 24, 0, 112, 928, 92, 24, 1, 2> GrxShapeHandBlockTouch;  // + the 92 touch zones of robot_touch_sensors_92.xml
typedef GrxShape<31, 30, 20, 26, 25, 24, 0, 0, 24, 0, 256, 4080, 92, 64, 1, 2> GrxShapeHandBlockTouchLane;   // overflow-lane tables (see GrxShapeFetchPickLane)

// waves per SIMD the Fetch kernels are compiled for (VGPR budget 168 at 3, 256 at 2): the convex narrow phase needs the full budget
#ifndef GRX_FETCH_WAVES
#define GRX_FETCH_WAVES(S) (S::kHandoff ? 3 : 2)   // the wave-cooperative hull routine keeps ~100 values live: at 168 VGPRs (3 waves) the step kernels spill 60-130 registers and run slower than at 2 waves even with 9 instead of 8 worlds per CU and the hull branch marked cold (measured 3.75 vs 3.44 ms per step)
#endif
// one world's env.step().  LANE: called from the list-walking loop of the large-table kernel (w comes from the list; see grx_overflow_lane)
// SPLIT (include/grx_capi.h, grx_fetch_buffers.split_parts): `part` of `parts` workgroups of this world, each running its share of the substeps; 0 of 1 = the whole step.
// How a part of a split step hands the world to the next one (MI355X_MICROARCH.md, workgroup dispatch / hand-off forms): the carrier row is written with write-through (volatile = sc0 sc1)
// stores, drained with s_waitcnt vmcnt(0), then the flag word is stored the same way; the reader polls the flag and reads the row with L1-bypassing (volatile) loads.  Valid for any
// workgroup -> XCD placement, and without an agent-scope release: `__threadfence()` writes back EVERY dirty line of the XCD's L2 (buffer_wbl2) -- the scratch of all resident waves --
// once per part and wave: that was 19 MB of write-back per launch of 4 096 worlds (PMC traffic 3.9x -> 10.6x algorithmic) and what made a third and fourth part cost more than they
// saved.  -DGRX_SPLIT_AGENT_FENCES restores the fences (A/B: tools/ab_split_fences.sh).
#ifdef GRX_SPLIT_AGENT_FENCES
#define GRX_SPLIT_ROW float
#define GRX_SPLIT_DRAIN() __threadfence()
#define GRX_SPLIT_ACQUIRE() __threadfence()
#else
#define GRX_SPLIT_ROW volatile float
#define GRX_SPLIT_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define GRX_SPLIT_ACQUIRE() asm volatile("" ::: "memory")
#endif
#define GRX_SPLIT_SPIN_LIMIT (1 << 22)   // polls (~64 cycles each: > 100 ms) before a part gives its predecessor up: never reached while workgroups of an XCD start in index order
template <class S, bool LANE>
__device__ __forceinline__ void grx_fetch_step_world(int mslot, const GrxFetchTask& t, const GrxFetchBuffers& b, const int w, int n_worlds, int words, float* lds, const int lane_,
                                                     const int part = 0, const int parts = 1) {
  if (w >= n_worlds) return;
  if (b.mask && !b.mask[w]) { if (LANE) grx_lane_ticket(b.lane, -1, w, lane_); return; }
  if (!LANE && b.lane.skip && b.lane.skip[w]) return;   // in the overflow lane: stepped by the large-table kernel
  const bool split = !LANE && parts > 1, last_part = part == parts - 1;
  if (split && part > 0) {   // wait for the part before this one (it was dispatched earlier on the same XCD: it is running or done)
    volatile int* st = b.split_state + 2 * (size_t)w;
    int v = 0;
    for (int spins = 0; spins < GRX_SPLIT_SPIN_LIMIT; spins++) {
      v = __builtin_amdgcn_readfirstlane(st[0]);
      if (v == part || v < 0) break;
      __builtin_amdgcn_s_sleep(8);
    }
    if (v != part) {   // the earlier part booked the world's re-run (v < 0), or never came (v < part: flagged): nothing to do here; the last part leaves the word clean
      if (lane_ == 0) { if (v >= 0) b.status[w] |= GRX_ST_BADNUM | (GRX_ST_BADNUM << 16); if (last_part) st[0] = 0; }
      return;
    }
    GRX_SPLIT_ACQUIRE();
  }
  const GrxModel& m = g_grx_models[mslot];
  GrxCtx c;
  c.mslot = mslot;
  grx_ctx_carve(&c, lds, grx_shape_dims<S>(m));
  grx_lane_setup(b.lane, c, w, true);
  if (b.handoff && c.bail && !split) { c.handoff = b.handoff; c.handoff_stride = b.handoff_stride; c.handoff_large = b.handoff_large; }   // this launch can hand a world off mid-step (it has an entry list to claim from)
#ifdef GRX_PROFILE
  __shared__ long long prof_s[GRX_NPROF + 1];
  c.prof = prof_s; c.prof_last = prof_s + GRX_NPROF;
  if (lane_ == 0) { for (int k = 0; k < GRX_NPROF; k++) prof_s[k] = 0; prof_s[GRX_NPROF] = clock64(); if (w < 16384) g_grx_world_span[2 * w] = wall_clock64(); }
#endif
#ifndef GRX_COST_MODEL
  if (b.cost && lane_ == 0) b.cost[w] = (int)wall_clock64();   // start stamp (100 MHz), parked in the cost slot itself: nothing stays live across the substep loop
#endif
#if defined(GRX_WORLD_SPAN) && !defined(GRX_PROFILE)
  if (lane_ == 0 && w < 16384) g_grx_world_span[2 * w] = wall_clock64();
#endif
  grx_load_world(m, b, c, w, lds, words, lane_);
  if (S::kMesh && m.nmeshpair > 0 && b.hullcache) c.hullhint = b.hullcache + (size_t)w * GRX_HULLCACHE_WORDS + 21;   // support-vertex guesses of the world's persistent hull contacts: used in place (HBM)
  if (S::kMesh && m.nmeshpair > 0 && b.hullcache && lane_ < 21)   // separating directions remembered from the previous step (re-verified before use); a later part of a split step reads what the part before it wrote (another CU: past the L1)
    c.meshcache[lane_] = (split && part > 0) ? ((volatile const float*)b.hullcache)[(size_t)w * GRX_HULLCACHE_WORDS + lane_] : b.hullcache[(size_t)w * GRX_HULLCACHE_WORDS + lane_];
  float aux_in[8];
  for (int k = 0; k < 8; k++) aux_in[k] = b.aux[(size_t)w * 8 + k];
  int s0 = 0;
  if ((LANE && b.handoff) || (split && part > 0)) {   // a world another kernel of this step handed off (or the part before this one of a split step): resume AT the substep it stopped before (its row was written during this very launch group, possibly
                             // behind another XCD's L2: volatile = cache-bypassing loads, ordered behind the claim of the entry by the caller's fence)
    volatile const float* row = b.handoff + (size_t)w * b.handoff_stride;
    const int hs = __builtin_amdgcn_readfirstlane(((volatile const int*)row)[0]);
    if (hs > 0) {
      GrxFetch<S>::grx_fetch_set_action(&m, &t, &c, aux_in, b.action + (size_t)w * 4, lane_);   // (everything it computes is overwritten below; kept so that the resumed path differs from the plain one by loads only)
      const int nq = S::kFixed ? S::NQ : m.nq, nv = S::kFixed ? S::NV : m.nv, nu = S::kFixed ? S::NU : m.nu, nmo = S::kFixed ? S::NM : m.nmocap;
      int o = 2;
      for (int i = lane_; i < nu; i += 64) c.ctrl[i] = row[o + i];
      o += nu;
      for (int i = lane_; i < 7 * nmo; i += 64) { const int k = i / 7, e = i - 7 * k; const float v = row[o + i]; if (e < 3) c.mocap_pos[3 * k + e] = v; else c.mocap_quat[4 * k + e - 3] = v; }
      o += 7 * nmo;
      for (int i = lane_; i < nq; i += 64) c.qpos[i] = row[o + i];
      o += nq;
      for (int i = lane_; i < nv; i += 64) { c.qvel[i] = row[o + i]; c.qacc_ws[i] = row[o + nv + i]; }
      if (lane_ == 0) { c.cnt[2] |= ((volatile const int*)row)[1]; ((volatile int*)row)[0] = 0; }   // the flags of the substeps the other kernel ran; the row is consumed
      __syncthreads();
      c.resume_first = (split && part > 0) ? 0 : 1;   // (a part of a split step ended at a substep BOUNDARY: the next substep normalises the free-joint quaternions as the plain step does)
      s0 = hs - 1;
    }
  }
  const int total_ = t.n_substeps + (t.block_gripper ? 1 : 0), s_end = (split && !last_part) ? ((part + 1) * total_) / parts : -1;
  GrxFetch<S>::grx_fetch_sim_world(&m, &t, &c, aux_in, b.action + (size_t)w * 4, lane_, s0, s_end);
  const int wl = LANE ? w : grx_world_of_slot_late(b.order, parts);
  if (split && !last_part) {   // an earlier part of a split step: the state goes to the world's row, nothing else is written
    volatile int* st = b.split_state + 2 * (size_t)wl;
    if (grx_lane_overflowed(c)) { if (lane_ == 0) st[0] = -1; }      // the re-run on the large tables is booked: the later parts return
    else {
      if (S::kMesh && m.nmeshpair > 0 && b.hullcache && lane_ < 21) ((GRX_SPLIT_ROW*)b.hullcache)[(size_t)wl * GRX_HULLCACHE_WORDS + lane_] = c.meshcache[lane_];
      const int nq = S::kFixed ? S::NQ : m.nq, nv = S::kFixed ? S::NV : m.nv, nu = S::kFixed ? S::NU : m.nu, nmo = S::kFixed ? S::NM : m.nmocap;
      GRX_SPLIT_ROW* row = b.handoff + (size_t)wl * b.handoff_stride;
      int o = 2;
      for (int i = lane_; i < nu; i += 64) row[o + i] = c.ctrl[i];
      o += nu;
      for (int i = lane_; i < 7 * nmo; i += 64) { const int k = i / 7, e = i - 7 * k; row[o + i] = (e < 3) ? c.mocap_pos[3 * k + e] : c.mocap_quat[4 * k + e - 3]; }
      o += 7 * nmo;
      for (int i = lane_; i < nq; i += 64) row[o + i] = c.qpos[i];
      o += nq;
      for (int i = lane_; i < nv; i += 64) { row[o + i] = c.qvel[i]; row[o + nv + i] = c.qacc_ws[i]; }
      if (lane_ == 0) { ((volatile int*)row)[1] = c.cnt[2]; ((volatile int*)row)[0] = s_end + 1; }
      GRX_SPLIT_DRAIN();
      __syncthreads();
      if (lane_ == 0) {
        if (b.cost) { const int t0 = ((volatile int*)b.cost)[wl]; st[1] = (part > 0 ? st[1] : 0) + (((int)wall_clock64() - t0) >> 3); }
        GRX_SPLIT_DRAIN();
        st[0] = part + 1;
      }
    }
    return;
  }
  // a capacity overflowed (wave-uniform: the flag lives in LDS): keep nothing, the world is re-run on the large tables (grx_overflow_lane)
  if (!grx_lane_overflowed(c)) {
    if (LANE) grx_lane_ticket(b.lane, c.cnt[2] & 0xFFFF, wl, lane_); else grx_lane_join(b.lane, c, wl, lane_);
    if (S::kMesh && m.nmeshpair > 0 && b.hullcache && lane_ < 21) b.hullcache[(size_t)wl * GRX_HULLCACHE_WORDS + lane_] = c.meshcache[lane_];
    GrxFetch<S>::grx_fetch_outputs(&m, &t, &c, b.aux + (size_t)wl * 8, b.obs + (size_t)wl * t.obs_dim, b.achieved + (size_t)wl * 3, lane_);
    __syncthreads();
    grx_store_world(m, t, b, c, wl, lane_);
  }
#ifdef GRX_PROFILE_ITER
  if (b.cost && lane_ == 0) b.cost[wl] = c.cnt[6] | (c.cnt[0] << 16);   // diagnostic build: Newton iterations of the step, contacts of the last substep
#else
  // The cost the next launch is ordered by = the MEASURED duration of this world (device-wide 100 MHz clock, 80 ns units).  Round 1 counted instead
  // (12 us per Newton iteration + 24 us per contact): the count knows nothing of the hull-pair refinement and of the line-search length, and a
  // world it under-estimates starts in the second round and ends the launch alone (A/B in one process, FetchPickAndPlace 4096 worlds: 3.48 ms with
  // the count, 3.16 ms with the measurement; -DGRX_COST_MODEL restores the count, whose dispatch order is reproducible from run to run).
#ifdef GRX_COST_MODEL
  if (b.cost && lane_ == 0) b.cost[wl] = 12 * c.cnt[6] + 24 * c.cnt[0];
#else
  if (b.cost && lane_ == 0) { const int t0 = ((volatile int*)b.cost)[wl]; b.cost[wl] = (((int)wall_clock64() - t0) >> 3) + (split ? ((volatile int*)b.split_state)[2 * (size_t)wl + 1] : 0); }
#endif
#endif
  if (split && lane_ == 0) ((volatile int*)b.split_state)[2 * (size_t)wl] = 0;   // the last part leaves the world's word clean for the next launch
#if defined(GRX_WORLD_SPAN) && !defined(GRX_PROFILE)
  if (lane_ == 0 && wl < 16384) g_grx_world_span[2 * wl + 1] = wall_clock64();
#endif
#ifdef GRX_PROFILE
  GRX_TICK(&c, GRX_P_OTHER);
  if (lane_ == 0) for (int k = 0; k < GRX_NPROF; k++) atomicAdd((unsigned long long*)&g_grx_prof[k], (unsigned long long)c.prof[k]);  // summed over worlds
  if (lane_ == 0 && w < 16384) g_grx_world_span[2 * w + 1] = wall_clock64();
  if (lane_ == 0 && w < 4096) for (int k = 0; k < GRX_NPROF; k++) g_grx_world_prof[w * GRX_NPROF + k] = (int)c.prof[k];
#endif
}
template <class S>
__global__ void __launch_bounds__(64, GRX_FETCH_WAVES(S))
grx_fetch_step_kernel(int mslot, GrxFetchTask t, GrxFetchBuffers b, int n_worlds, int words) {
  extern __shared__ float lds[];
  const int lane_ = threadIdx.x;
  const int parts = b.split_parts > 1 ? b.split_parts : 1;      // (one part: G = the grid, part 0, slot = the workgroup: grx_world_of_block)
  const unsigned G = gridDim.x / (unsigned)parts, part = blockIdx.x / G, slot = blockIdx.x - part * G;   // slot and slot + G, slot + 2 G ... share blockIdx.x mod 8: one XCD, one L2 for all parts of a world; ONE call site of the step (code size, compile time)
  grx_fetch_step_world<S, false>(mslot, t, b, b.order ? b.order[slot] : (int)((slot & 7u) * (G >> 3) + (slot >> 3)), n_worlds, words, lds, lane_, (int)part, parts);
  grx_lane_progress(b.lane);   // (launches with polling workgroups behind them: this workgroup has ended)
}
// the large-table kernel of the overflow lane (grx_overflow_lane): a small fixed grid walks the compacted list of worlds; generic shape only
template <class S>
__global__ void __launch_bounds__(64, 2)
grx_fetch_lane_kernel(int mslot, GrxFetchTask t, GrxFetchBuffers b, int n_worlds, int words) {   // one workgroup per entry of the compacted list (the caller sizes the grid: grx_overflow_lane.grid >= the list's cap)
  extern __shared__ float lds[];
  const int e = blockIdx.x, nstand = (int)gridDim.x - b.lane.poll_grid;
  // entries carry the world in their low 30 bits; bit 30 = "needs the LARGE tables" (claimed by a launch that already ran the middle ones, GrxCtx::handoff_large)
  if (e >= nstand) {   // a polling workgroup (grx_lane_poll): takes entry e - nstand of THIS step's entry list while the fast launch is still running
    const int ent = grx_lane_poll(b.lane, e - nstand);
    if (ent < 0) return;
    if ((ent >> 30) & 1) { if (threadIdx.x == 0) atomicExch(b.lane.ready + (e - nstand), 1); return; }   // not for these tables: give the entry back, the entry launch behind the fast kernel takes it
    grx_fetch_step_world<S, true>(mslot, t, b, ent & 0x3FFFFFFF, n_worlds, words, lds, (int)threadIdx.x);
    return;
  }
  if (e >= *b.lane.count || grx_lane_taken(b.lane, e)) return;
  grx_fetch_step_world<S, true>(mslot, t, b, b.lane.list[e] & 0x3FFFFFFF, n_worlds, words, lds, (int)threadIdx.x);
}

// reset-time mj_forward + outputs (nstep > 0: raw settle steps first, _env_setup).  Shape-specialised like the step kernel: the generic
// instantiation needs 275 VGPRs (one wave per SIMD) and took 1.6 x a whole env.step().
template <class S>
__global__ void __launch_bounds__(64, GRX_FETCH_WAVES(S))
grx_fetch_forward_kernel(int mslot, GrxFetchTask t, GrxFetchBuffers b, int n_worlds, int words, int nstep) {
  extern __shared__ float lds[];
  const int w = grx_world_of_block(), lane_ = threadIdx.x;
  if (w >= n_worlds) return;
  if (b.mask && !b.mask[w]) return;
  const GrxModel& m = g_grx_models[mslot];
  GrxCtx c;
  c.mslot = mslot;
  grx_ctx_carve(&c, lds, grx_shape_dims<S>(m));
#ifdef GRX_PROFILE
  __shared__ long long prof_s[GRX_NPROF + 1];
  c.prof = prof_s; c.prof_last = prof_s + GRX_NPROF;
#endif
  grx_load_world(m, b, c, w, lds, words, lane_);
  { const int total = nstep > 0 ? nstep : 1; for (int s = 0; s < total; s++) GrxEngine<S>::grx_forward_euler(&m, &c, nstep > 0, lane_); }
  const int wl = grx_world_of_block_late();
  GrxFetch<S>::grx_fetch_outputs(&m, &t, &c, b.aux + (size_t)wl * 8, b.obs + (size_t)wl * t.obs_dim, b.achieved + (size_t)wl * 3, lane_);
  __syncthreads();
  grx_store_world(m, t, b, c, wl, lane_);
}

// Episode reset of a COMPACTED list of worlds (fetch_env.py:375-402 _reset_sim + :153-166 _sample_goal + mj_forward + _get_obs): workgroup k
// resets world idx[k] -- initial state rows, the object position and the goal the host drew for it -- so an autoreset of 82 of 4096 worlds is
// an 82-workgroup launch instead of a masked launch over the whole grid, and nothing on the host waits for the device.
struct GrxFetchResetArgs {
  const int* idx;          // [n] worlds to reset
  const float* samples;    // [n,5] object xy, goal xyz (host PCG64 draws, grx_fetch_sample_resets)
  const float *init_qpos, *init_qvel, *init_mocap;   // [nq] [nv] [7*nmocap]: the state _env_setup left (fetch_env.py:404-428)
  int obj_qadr;            // qpos address of object0:joint, -1 without object
  int keep_outcome;        // same-step autoreset: reward / success (and the packed row's last two words) keep the finished episode's values
  float* final_packed;     // [N, obs_dim + 8] or null: the packed row of world w (the finished episode's terminal row) is parked here before the reset overwrites it
};
template <class S>
__global__ void __launch_bounds__(64, GRX_FETCH_WAVES(S))
grx_fetch_reset_kernel(int mslot, GrxFetchTask t, GrxFetchBuffers b, GrxFetchResetArgs r, int n_reset, int words) {
  extern __shared__ float lds[];
  const int k = blockIdx.x, lane_ = threadIdx.x;
  if (k >= n_reset) return;
  const int w = r.idx[k];
  const GrxModel& m = g_grx_models[mslot];
  if (r.final_packed && b.packed) { const int pw = t.obs_dim + 8; for (int i = lane_; i < pw; i += 64) r.final_packed[(size_t)w * pw + i] = b.packed[(size_t)w * pw + i]; }
  GrxCtx c;
  c.mslot = mslot;
  grx_ctx_carve(&c, lds, grx_shape_dims<S>(m));
#ifdef GRX_PROFILE
  __shared__ long long prof_s[GRX_NPROF + 1];
  c.prof = prof_s; c.prof_last = prof_s + GRX_NPROF;
#endif
  for (int i = lane_; i < words; i += 64) lds[i] = 0.0f;
  __syncthreads();
  for (int i = lane_; i < m.nq; i += 64) c.qpos[i] = r.init_qpos[i];
  for (int i = lane_; i < m.nv; i += 64) c.qvel[i] = r.init_qvel[i];   // qacc_warmstart stays 0 (mj_resetData)
  for (int i = lane_; i < 7 * m.nmocap; i += 64) {
    const int q = i / 7, e = i - 7 * q; const float v = r.init_mocap[i];
    if (e < 3) c.mocap_pos[3 * q + e] = v; else c.mocap_quat[4 * q + e - 3] = v;
  }
  __syncthreads();
  if (lane_ == 0) {
    if (r.obj_qadr >= 0) { c.qpos[r.obj_qadr] = r.samples[5 * k]; c.qpos[r.obj_qadr + 1] = r.samples[5 * k + 1]; }
    float* goal = const_cast<float*>(b.goal) + (size_t)w * 3;
    goal[0] = r.samples[5 * k + 2]; goal[1] = r.samples[5 * k + 3]; goal[2] = r.samples[5 * k + 4];
  }
  __syncthreads();
  GrxEngine<S>::grx_forward_euler(&m, &c, 0, lane_);
  GrxFetch<S>::grx_fetch_outputs(&m, &t, &c, b.aux + (size_t)w * 8, b.obs + (size_t)w * t.obs_dim, b.achieved + (size_t)w * 3, lane_);
  __syncthreads();
  grx_store_world(m, t, b, c, w, lane_, r.keep_outcome);
}

// PointMaze env.step(): one wavefront per world, same engine
template <class S>
__global__ void __launch_bounds__(64, S::kFixed ? 3 : 2)
grx_point_step_kernel(int mslot, GrxPointTask t, GrxPointBuffers b, int n_worlds, int words) {
  extern __shared__ float lds[];
  const int lane_ = threadIdx.x;
  // SPLIT STEP (include/grx_capi.h grx_point_buffers.split_parts; the Fetch family's: grx_fetch_buffers.split_parts): P workgroups per world, workgroup part * G + slot running the
  // substeps [part T / P, (part + 1) T / P) of the slot's world.  The carrier is the world's own state row: at a substep boundary qpos / qvel / warm start are the whole state.
  const int parts = b.split_parts > 1 ? b.split_parts : 1;
  const unsigned G = gridDim.x / (unsigned)parts, part = blockIdx.x / G, slot = blockIdx.x - part * G;   // slot, slot + G, ... share blockIdx.x mod 8: one XCD (one L2) for all parts of a world
  const int w = (int)((slot & 7u) * (G >> 3) + (slot >> 3));
  if (w >= n_worlds) return;
  if (b.mask && !b.mask[w]) return;
  const bool split = parts > 1, last_part = (int)part == parts - 1;
  if (split && part > 0) {   // wait for the part before this one (dispatched earlier on the same XCD: running or done)
    volatile int* st = b.split_state + 2 * (size_t)w;
    int v = 0;
    for (int spins = 0; spins < GRX_SPLIT_SPIN_LIMIT; spins++) {
      v = __builtin_amdgcn_readfirstlane(st[0]);
      if (v == (int)part) break;
      __builtin_amdgcn_s_sleep(8);
    }
    if (v != (int)part) {   // the earlier part never came: flagged; the last part leaves the words clean
      if (lane_ == 0) { b.status[w] |= GRX_ST_BADNUM | (GRX_ST_BADNUM << 16); if (last_part) { st[0] = 0; st[1] = 0; } }
      return;
    }
    GRX_SPLIT_ACQUIRE();
  }
  const GrxModel& m = g_grx_models[mslot];
  GrxCtx c;
  c.mslot = mslot;
  grx_ctx_carve(&c, lds, grx_shape_dims<S>(m));
#ifdef GRX_PROFILE
  __shared__ long long prof_s[GRX_NPROF + 1];
  c.prof = prof_s; c.prof_last = prof_s + GRX_NPROF;
  if (lane_ == 0) { for (int k = 0; k < GRX_NPROF; k++) prof_s[k] = 0; prof_s[GRX_NPROF] = clock64(); }
#endif
  for (int i = lane_; i < words; i += 64) lds[i] = 0.0f;
  __syncthreads();
  if (split && part > 0) {   // rows the part before this one wrote on another CU during this launch: cache-bypassing loads, ordered behind its flag by the fence above
    volatile const float *vq = b.qpos + (size_t)w * m.nq, *vv = b.qvel + (size_t)w * m.nv, *va = b.qacc_ws + (size_t)w * m.nv;
    for (int i = lane_; i < m.nq; i += 64) c.qpos[i] = vq[i];
    for (int i = lane_; i < m.nv; i += 64) { c.qvel[i] = vv[i]; c.qacc_ws[i] = va[i]; }
  } else {
  for (int i = lane_; i < m.nq; i += 64) c.qpos[i] = b.qpos[(size_t)w * m.nq + i];
  for (int i = lane_; i < m.nv; i += 64) { c.qvel[i] = b.qvel[(size_t)w * m.nv + i]; c.qacc_ws[i] = b.qacc_ws[(size_t)w * m.nv + i]; }
  }
  __syncthreads();
  const int s0 = split ? ((int)part * t.n_substeps) / parts : 0, s1 = split ? (((int)part + 1) * t.n_substeps) / parts : t.n_substeps;
  GrxPoint<S>::grx_point_sim_world(&m, &t, &c, b.action + (size_t)w * m.nu, lane_, s0, s1);
  unsigned bx_ = blockIdx.x; asm volatile("" : "+s"(bx_));   // (re-derived, not kept live across the simulation)
  const unsigned Gl = gridDim.x / (unsigned)parts, sl = bx_ % Gl;
  const int wl = (int)((sl & 7u) * (Gl >> 3) + (sl >> 3));
  if (split && !last_part) {   // an earlier part: the state row IS the carrier; the flags of its substeps travel in the world's second word
    __syncthreads();
    GRX_SPLIT_ROW *rq = b.qpos + (size_t)wl * m.nq, *rv = b.qvel + (size_t)wl * m.nv, *ra = b.qacc_ws + (size_t)wl * m.nv;
    for (int i = lane_; i < m.nq; i += 64) rq[i] = c.qpos[i];
    for (int i = lane_; i < m.nv; i += 64) { rv[i] = c.qvel[i]; ra[i] = c.qacc_ws[i]; }
    GRX_SPLIT_DRAIN();
    __syncthreads();
    if (lane_ == 0) {
      volatile int* st = b.split_state + 2 * (size_t)wl;
      st[1] = (part > 0 ? st[1] : 0) | c.cnt[2];
      GRX_SPLIT_DRAIN();
      st[0] = (int)part + 1;
    }
    return;
  }
  if (split && lane_ == 0) { volatile int* st = b.split_state + 2 * (size_t)wl; c.cnt[2] |= st[1]; st[0] = 0; st[1] = 0; }   // the flags of the earlier parts; the words are clean for the next launch
  if (split) __syncthreads();
  float* obs = b.obs + (size_t)wl * (m.nq + m.nv - (t.agent ? 2 : 0)); float* ach = b.achieved + (size_t)wl * 2;
  GrxPoint<S>::grx_point_outputs(&m, &t, &c, obs, ach, lane_);
  __syncthreads();
  for (int i = lane_; i < m.nq; i += 64) b.qpos[(size_t)wl * m.nq + i] = c.qpos[i];
  for (int i = lane_; i < m.nv; i += 64) { b.qvel[(size_t)wl * m.nv + i] = c.qvel[i]; b.qacc_ws[(size_t)wl * m.nv + i] = c.qacc_ws[i]; }
  if (lane_ == 0) {
    const double d = grx_goal_distance2(ach, b.goal + (size_t)wl * 2);
    int succ = d <= t.goal_radius;
    b.reward[wl] = grx_maze_reward(d, t.goal_radius, t.sparse_reward);
    b.success[wl] = succ; b.terminated[wl] = (!t.continuing_task && succ) ? 1 : 0;
    b.status[wl] = grx_status_word(b.status[wl], c.cnt[2]);
  }
  if (b.packed) {   // [obs | achieved | desired | reward | success] row for the cross-rank gather
    const int od = m.nq + m.nv - (t.agent ? 2 : 0);
    float* row = b.packed + (size_t)wl * (od + 6);
    for (int i = lane_; i < od; i += 64) row[i] = obs[i];
    if (lane_ == 0) {
      const double d = grx_goal_distance2(ach, b.goal + (size_t)wl * 2);
      row[od] = ach[0]; row[od + 1] = ach[1]; row[od + 2] = b.goal[(size_t)wl * 2]; row[od + 3] = b.goal[(size_t)wl * 2 + 1];
      row[od + 4] = grx_maze_reward(d, t.goal_radius, t.sparse_reward); row[od + 5] = (d <= t.goal_radius) ? 1.0f : 0.0f;
    }
  }
#ifdef GRX_PROFILE
  GRX_TICK(&c, GRX_P_OTHER);
  if (lane_ == 0) for (int k = 0; k < GRX_NPROF; k++) atomicAdd((unsigned long long*)&g_grx_prof[k], (unsigned long long)c.prof[k]);
#endif
}

// Shadow hand reach env.step() (or mj_forward + outputs when forward_only): one wavefront per world, same engine
template <class S>
#ifndef GRX_HANDREACH_WAVES
#define GRX_HANDREACH_WAVES 2   // with the hull-pair routine the 168-VGPR build spills 73 registers: 11.9 ms per step at 16 384 worlds against 10.95 ms at 2 waves
#endif
__device__ __forceinline__ void grx_hand_step_world(int mslot, const GrxHandTask& t, const GrxHandBuffers& b, const int w, int n_worlds, int words, int forward_only, float* lds, const int lane_, const bool in_lane,
                                                    const int part = 0, const int parts = 1) {
  if (w >= n_worlds) return;
  if (b.mask && !b.mask[w]) { if (in_lane) grx_lane_ticket(b.lane, -1, w, lane_); return; }
  if (!in_lane && forward_only != 1 && b.lane.skip && b.lane.skip[w]) return;   // in the overflow lane: stepped by the large-table kernel (reset-time forward passes cover every masked world)
  // SPLIT STEP (include/grx_capi.h grx_hand_buffers.split_parts; see grx_adroit_step_world): part `part` of `parts` workgroups of this world, each running its share of the substeps (plain step launches only)
  const bool split = parts > 1, last_part = part == parts - 1;
  if (split && part > 0) {
    volatile int* st = b.split_state + 4 * (size_t)w;
    int v = 0;
    for (int spins = 0; spins < GRX_SPLIT_SPIN_LIMIT; spins++) {
      v = __builtin_amdgcn_readfirstlane(st[0]);
      if (v == part || v < 0) break;
      __builtin_amdgcn_s_sleep(8);
    }
    if (v != part) {   // the earlier part booked the world's re-run (v < 0), or never came (flagged): nothing to do here; the last part leaves the words clean
      if (lane_ == 0) { if (v >= 0) b.status[w] |= GRX_ST_BADNUM | (GRX_ST_BADNUM << 16); if (last_part) { st[0] = 0; st[1] = 0; st[2] = 0; } }
      return;
    }
    GRX_SPLIT_ACQUIRE();
  }
  const GrxModel& m = g_grx_models[mslot];
  GrxCtx c;
  c.mslot = mslot;
  grx_ctx_carve(&c, lds, grx_shape_dims<S>(m));
#ifdef GRX_PROFILE
  __shared__ long long prof_s[GRX_NPROF + 1];
  c.prof = prof_s; c.prof_last = prof_s + GRX_NPROF;
  if (lane_ == 0) { for (int k = 0; k < GRX_NPROF; k++) prof_s[k] = 0; prof_s[GRX_NPROF] = clock64(); }
#endif
  const int nq = S::kFixed ? S::NQ : m.nq, nv = S::kFixed ? S::NV : m.nv, nu = S::kFixed ? S::NU : m.nu;
  // REPEAT launches (grx_hand_step_repeat: forward_only carries the count, >= 2): that many consecutive env.step()s of the same action rows without leaving the kernel -- each one
  // the whole body below, state rows written and read back exactly as between two launches (bit-identical to them), so a reset's ten settle steps (manipulate.py:205-224) are
  // one launch: one wait for a wave slot, one tail, instead of ten of each beside a step kernel that fills the chip
  const int nrep = forward_only > 1 ? forward_only : 1;
  forward_only = forward_only == 1;
  grx_lane_setup(b.lane, c, w, !forward_only);
  const int od = grx_hand_obs_dim(&t, nq, nv, m.ntouch), gd = grx_hand_goal_dim(&t);
  float* obs = b.obs + (size_t)w * od; float* ach = b.achieved + (size_t)w * gd; float* palm = b.palm + (size_t)w * 3;
  for (int rep = 0; rep < nrep; rep++) {
  if (rep > 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }   // the rows this wave wrote in the previous repetition are read back below: its stores have reached the L2, this CU's L1 is invalidated (no release fence: that would write back the whole L2, see GRX_SPLIT_DRAIN)
  if (b.cost && lane_ == 0) b.cost[w] = (int)wall_clock64();   // start stamp, parked in the cost slot (see grx_fetch_step_kernel)
  for (int i = lane_; i < words; i += 64) lds[i] = 0.0f;
  __syncthreads();
  if (split && part > 0) {   // the row the part before this one wrote on another CU during this launch: cache-bypassing loads
    volatile const float* row = b.split_rows + (size_t)w * b.split_stride;
    for (int i = lane_; i < nq; i += 64) c.qpos[i] = row[i];
    for (int i = lane_; i < nv; i += 64) { c.qvel[i] = row[nq + i]; c.qacc_ws[i] = row[nq + nv + i]; }
  } else {
  for (int i = lane_; i < nq; i += 64) c.qpos[i] = b.qpos[(size_t)w * nq + i];
  for (int i = lane_; i < nv; i += 64) { c.qvel[i] = b.qvel[(size_t)w * nv + i]; c.qacc_ws[i] = b.qacc_ws[(size_t)w * nv + i]; }
  }
  __syncthreads();
  if (forward_only) {
    GrxEngine<S>::grx_forward_euler(&m, &c, 0, lane_);
    GrxHand<S>::grx_hand_outputs(&m, &t, &c, obs, ach, palm, lane_);
  } else {
    const int s0 = split ? (part * t.n_substeps) / parts : 0, s1 = split ? ((part + 1) * t.n_substeps) / parts : t.n_substeps;
    GrxHand<S>::grx_hand_step_world(&m, &t, &c, b.action + (size_t)w * nu, obs, ach, palm, lane_, s0, s1, !split || last_part);
  }
  __syncthreads();
  if (split && !last_part) {   // an earlier part: the state goes to the world's carrier row (write-through + drained flag: GRX_SPLIT_DRAIN), nothing else is written
    volatile int* st = b.split_state + 4 * (size_t)w;
    if (grx_lane_overflowed(c)) { if (lane_ == 0) st[0] = -1; }      // the re-run on the large tables is booked: the later parts return
    else {
      GRX_SPLIT_ROW* row = b.split_rows + (size_t)w * b.split_stride;
      for (int i = lane_; i < nq; i += 64) row[i] = c.qpos[i];
      for (int i = lane_; i < nv; i += 64) { row[nq + i] = c.qvel[i]; row[nq + nv + i] = c.qacc_ws[i]; }
      GRX_SPLIT_DRAIN();
      __syncthreads();
      if (lane_ == 0) {
        st[1] = (part > 0 ? st[1] : 0) | c.cnt[2];
        if (b.cost) { const int t0 = ((volatile int*)b.cost)[w]; st[2] = (part > 0 ? st[2] : 0) + (((int)wall_clock64() - t0) >> 3); }
        GRX_SPLIT_DRAIN();
        st[0] = part + 1;
      }
    }
    return;
  }
  int earlier = 0;
  if (split) {   // the last part: the flags and the measured time of the earlier parts; the words are clean for the next launch
    if (lane_ == 0) { volatile int* st = b.split_state + 4 * (size_t)w; c.cnt[2] |= st[1]; earlier = st[2]; st[0] = 0; st[1] = 0; st[2] = 0; }
    __syncthreads();
  }
  if (grx_lane_overflowed(c)) {   // capacity overflow: keep nothing (obs / achieved are outputs only), re-run on the large tables
    if (lane_ == 0 && b.cost) { const int t0 = ((volatile int*)b.cost)[w]; b.cost[w] = (((int)wall_clock64() - t0) >> 3) + earlier; }
    return;
  }
  if (in_lane) grx_lane_ticket(b.lane, c.cnt[2] & 0xFFFF, w, lane_); else if (!forward_only) grx_lane_join(b.lane, c, w, lane_);
  for (int i = lane_; i < nq; i += 64) b.qpos[(size_t)w * nq + i] = c.qpos[i];
  for (int i = lane_; i < nv; i += 64) { b.qvel[(size_t)w * nv + i] = c.qvel[i]; b.qacc_ws[(size_t)w * nv + i] = c.qacc_ws[i]; }
  if (lane_ == 0) {
    if (t.kind) {
      float dp, dr;
      grx_manip_distance(ach, b.goal + (size_t)w * gd, t.ignore_position, t.ignore_rotation, t.ignore_z, &dp, &dr);
      b.reward[w] = grx_manip_reward(dp, dr, (float)t.distance_threshold, t.rotation_threshold, t.sparse_reward);
      b.success[w] = grx_manip_success(dp, dr, (float)t.distance_threshold, t.rotation_threshold);
    } else {
      const double d = grx_goal_distance_n(ach, b.goal + (size_t)w * gd, gd);
      b.reward[w] = grx_hand_reward(d, t.distance_threshold, t.sparse_reward);
      b.success[w] = (d < t.distance_threshold) ? 1 : 0;
    }
    b.status[w] = grx_status_word(b.status[w], c.cnt[2]);
    if (b.cost) { const int t0 = ((volatile int*)b.cost)[w]; b.cost[w] = (((int)wall_clock64() - t0) >> 3) + earlier; }   // measured duration of this world, 80 ns units (see grx_fetch_step_kernel)
  }
  if (b.packed) {   // [obs | achieved | desired | reward | success] row for the cross-rank gather
    float* row = b.packed + (size_t)w * (od + 2 * gd + 2);
    __syncthreads();   // reward / success of lane 0 above
    for (int i = lane_; i < od; i += 64) row[i] = obs[i];
    for (int i = lane_; i < gd; i += 64) { row[od + i] = ach[i]; row[od + gd + i] = b.goal[(size_t)w * gd + i]; }
    if (lane_ == 0) { row[od + 2 * gd] = b.reward[w]; row[od + 2 * gd + 1] = b.success[w] ? 1.0f : 0.0f; }
  }
  }   // repetitions
#ifdef GRX_PROFILE
  GRX_TICK(&c, GRX_P_OTHER);
  if (lane_ == 0) for (int k = 0; k < GRX_NPROF; k++) atomicAdd((unsigned long long*)&g_grx_prof[k], (unsigned long long)c.prof[k]);
#endif
}
template <class S>
__global__ void __launch_bounds__(64, (S::kFixed && S::JP <= 512) ? GRX_HANDREACH_WAVES : 2)   // third wave per SIMD only where the LDS footprint lets more than 8 worlds share a CU (HandReach); the object models sit at 8
grx_hand_step_kernel(int mslot, GrxHandTask t, GrxHandBuffers b, int n_worlds, int words, int forward_only) {
  extern __shared__ float lds[];
  const int lane_ = threadIdx.x;
  const int parts = (b.split_parts > 1 && forward_only == 0) ? b.split_parts : 1;      // (one part: G = the grid, part 0, slot = the workgroup: grx_world_of_block)
  const unsigned G = gridDim.x / (unsigned)parts, part = blockIdx.x / G, slot = blockIdx.x - part * G;   // slot, slot + G, ... share blockIdx.x mod 8; ONE call site of the step (code size, compile time)
  grx_hand_step_world<S>(mslot, t, b, b.order ? b.order[slot] : (int)((slot & 7u) * (G >> 3) + (slot >> 3)), n_worlds, words, forward_only, lds, lane_, false, (int)part, parts);
  grx_lane_progress(b.lane);
}
// the large-table kernel of the overflow lane (grx_overflow_lane): a small fixed grid walks the compacted list of worlds; generic shape only
template <class S>
__global__ void __launch_bounds__(64, 2)
grx_hand_lane_kernel(int mslot, GrxHandTask t, GrxHandBuffers b, int n_worlds, int words) {   // one workgroup per entry of the compacted list (see grx_fetch_lane_kernel)
  extern __shared__ float lds[];
  const int e = blockIdx.x, nstand = (int)gridDim.x - b.lane.poll_grid;
  if (e >= nstand) { const int w = grx_lane_poll(b.lane, e - nstand); if (w >= 0) grx_hand_step_world<S>(mslot, t, b, w, n_worlds, words, 0, lds, (int)threadIdx.x, true); return; }
  if (e >= *b.lane.count || grx_lane_taken(b.lane, e)) return;
  grx_hand_step_world<S>(mslot, t, b, b.lane.list[e], n_worlds, words, 0, lds, (int)threadIdx.x, true);
}

// AdroitHandHammer env.step() (or, forward_only, the reset-time mj_forward + observation): one wavefront per world, same engine + the noslip pass
// nv = 33, every dof carries a friction-loss row (class main: 0.001; the nail: 2.5), general-affine actuators, cylinder / capsule pairs through the convex routine
// capacities of the Adroit FAST kernels (rows, Jacobian-pool words, contacts): the worlds that exceed one are stepped on the large tables of the overflow lane, so these are a
// throughput choice, not a correctness one -- they decide the LDS footprint, i.e. how many worlds a CU holds, and the step kernels' throughput is nearly proportional to that
// (profiles/ab_r05_two_worlds_occupancy.txt).  Measured per task (profiles/ab_r05_adroit_capacity.txt, 16 384 worlds, MI355X): hammer 144 / 2 032 / 32 (5 worlds per CU) 1.07 M ->
// 96 / 1 024 / 24 (7 per CU) 1.31 - 1.34 M env-steps/s (8 per CU: the lane's launches end the step, 1.03 - 1.13 M); pen 1.81 M -> 112 / 1 280 / 24 (7 per CU) 2.03 M -> 80 / 896 / 24 (8 per CU) 2.22 M
// (profiles/ab_r05_adroit_capacity2.txt); door and relocate LOSE with smaller tables (their overflow lane -- hands jammed
// into the door, the ball pressed into the table -- is already what the step waits for: door 1.17 M -> 1.15 M -> 0.87 M) and keep the defaults.  envs/adroit_spec.py ADROIT_CAPACITY must agree.
#ifndef GRX_ADROIT_ME      // (-DGRX_ADROIT_ME / _JP / _MC: one capacity for all four tasks; -DGRX_ADROIT_<TASK>_CAP=rows,pool,touch,contacts: one task -- the A/B builds)
#ifndef GRX_ADROIT_HAMMER_CAP
#define GRX_ADROIT_HAMMER_CAP 96, 1024, 1, 24
#endif
#ifndef GRX_ADROIT_PEN_CAP
#define GRX_ADROIT_PEN_CAP 80, 896, 0, 24
#endif
#define GRX_ADROIT_DOOR_CAP 144, 2032, 0, 32
#define GRX_ADROIT_RELOCATE_CAP 144, 2032, 0, 32
#else
#define GRX_ADROIT_HAMMER_CAP GRX_ADROIT_ME, GRX_ADROIT_JP, 1, GRX_ADROIT_MC
#define GRX_ADROIT_PEN_CAP GRX_ADROIT_ME, GRX_ADROIT_JP, 0, GRX_ADROIT_MC
#define GRX_ADROIT_DOOR_CAP GRX_ADROIT_ME, GRX_ADROIT_JP, 0, GRX_ADROIT_MC
#define GRX_ADROIT_RELOCATE_CAP GRX_ADROIT_ME, GRX_ADROIT_JP, 0, GRX_ADROIT_MC
#endif
typedef GrxShape<33, 33, 26, 29, 33, 30, 4, 1, 33, 0, GRX_ADROIT_HAMMER_CAP, 1, 13> GrxShapeAdroitHammer;   // CONVEX bits: 1 (cylinders) | 4 (board shift group) | 8 (noslip)
// AdroitHandDoor (nv 30: 4 arm + 24 hand + hinge + latch; the door frame is the shift group), AdroitHandPen (nv 30: 24 hand + 6 pen joints; the target
// cylinder is a ROTATING shift group: bit 16), AdroitHandRelocate (nv 36: 6 arm + 24 hand + 6 ball joints; the ball's body is the shift group, no cylinders)
typedef GrxShape<30, 30, 28, 29, 30, 32, 2, 1, 30, 0, GRX_ADROIT_DOOR_CAP, 1, 13> GrxShapeAdroitDoor;
typedef GrxShape<30, 30, 24, 27, 30, 26, 5, 1, 30, 0, GRX_ADROIT_PEN_CAP, 1, 29> GrxShapeAdroitPen;
typedef GrxShape<36, 36, 30, 28, 36, 25, 1, 1, 36, 0, GRX_ADROIT_RELOCATE_CAP, 1, 12> GrxShapeAdroitRelocate;
// the door and relocate models with the tables of the overflow lane (core.RERUN_CAPACITY): on the GENERIC large-table kernel one serialised re-run of a contact-rich door world
// took 5 - 9 ms of a 12 ms step (profiles/lane_probe_r03_door.txt); hammer and pen do not overflow in 100 000 world-steps and keep the generic lane kernel
typedef GrxShape<30, 30, 28, 29, 30, 32, 2, 1, 30, 0, 256, 4080, 0, 64, 1, 13> GrxShapeAdroitDoorLane;
typedef GrxShape<36, 36, 30, 28, 36, 25, 1, 1, 36, 0, 256, 4080, 0, 64, 1, 12> GrxShapeAdroitRelocateLane;
template <class S>
__device__ __forceinline__ void grx_adroit_step_world(int mslot, const GrxAdroitTask& t, const GrxAdroitBuffers& b, const int w, int n_worlds, int words, int forward_only, float* lds, const int lane_, const bool in_lane,
                                                      const int part = 0, const int parts = 1) {
  if (w >= n_worlds) return;
  if (b.mask && !b.mask[w]) { if (in_lane) grx_lane_ticket(b.lane, -1, w, lane_); return; }
  if (!in_lane && !forward_only && b.lane.skip && b.lane.skip[w]) return;   // in the overflow lane: stepped by the large-table kernel (reset-time forward passes cover every masked world)
  // SPLIT STEP (include/grx_capi.h grx_adroit_buffers.split_parts; mechanism: grx_fetch_buffers.split_parts): part `part` of `parts` workgroups of this world, each running its share of the
  // frame_skip substeps; the carrier is the world's row of split_rows (the state rows stay untouched until the last part, so a world that exceeds a table is re-run from them as usual)
  const bool split = parts > 1, last_part = part == parts - 1;
  if (split && part > 0) {
    volatile int* st = b.split_state + 4 * (size_t)w;
    int v = 0;
    for (int spins = 0; spins < GRX_SPLIT_SPIN_LIMIT; spins++) {
      v = __builtin_amdgcn_readfirstlane(st[0]);
      if (v == part || v < 0) break;
      __builtin_amdgcn_s_sleep(8);
    }
    if (v != part) {   // the earlier part booked the world's re-run (v < 0), or never came (flagged): nothing to do here; the last part leaves the words clean
      if (lane_ == 0) { if (v >= 0) b.status[w] |= GRX_ST_BADNUM | (GRX_ST_BADNUM << 16); if (last_part) { st[0] = 0; st[1] = 0; st[2] = 0; } }
      return;
    }
    GRX_SPLIT_ACQUIRE();
  }
  const GrxModel& m = g_grx_models[mslot];
  GrxCtx c;
  c.mslot = mslot;
  grx_ctx_carve(&c, lds, grx_shape_dims<S>(m));
#ifdef GRX_PROFILE
  __shared__ long long prof_s[GRX_NPROF + 1];
  c.prof = prof_s; c.prof_last = prof_s + GRX_NPROF;
  if (lane_ == 0) { for (int k = 0; k < GRX_NPROF; k++) prof_s[k] = 0; prof_s[GRX_NPROF] = clock64(); }
#endif
  const int nq = S::kFixed ? S::NQ : m.nq, nv = S::kFixed ? S::NV : m.nv, nu = S::kFixed ? S::NU : m.nu;
  for (int i = lane_; i < words; i += 64) lds[i] = 0.0f;
  __syncthreads();
  if (split && part > 0) {   // the row the part before this one wrote on another CU during this launch: cache-bypassing loads
    volatile const float* row = b.split_rows + (size_t)w * b.split_stride;
    for (int i = lane_; i < nq; i += 64) c.qpos[i] = row[i];
    for (int i = lane_; i < nv; i += 64) { c.qvel[i] = row[nq + i]; c.qacc_ws[i] = row[nq + nv + i]; }
  } else {
  for (int i = lane_; i < nq; i += 64) c.qpos[i] = b.qpos[(size_t)w * nq + i];
  for (int i = lane_; i < nv; i += 64) { c.qvel[i] = b.qvel[(size_t)w * nv + i]; c.qacc_ws[i] = b.qacc_ws[(size_t)w * nv + i]; }
  }
  for (int i = lane_; i < 7 * m.nmocap; i += 64) { const int q = i / 7, e = i - 7 * q; if (e < 3) c.mocap_pos[3 * q + e] = m.mocap_pos0[3 * q + e]; else c.mocap_quat[4 * q + e - 3] = m.mocap_quat0[4 * q + e - 3]; }
  if (m.nshift && lane_ < 7) c.shift[lane_] = b.shift[(size_t)w * 7 + lane_];
  __syncthreads();
  grx_lane_setup(b.lane, c, w, !forward_only);
  const bool timed = b.cost && !forward_only && !in_lane && !b.compact;   // cost-ordered dispatch (include/grx_capi.h grx_adroit_buffers.order / .cost): the start stamp (100 MHz) is parked in the cost slot itself
  if (timed && lane_ == 0) b.cost[w] = (int)wall_clock64();
  const int s0 = split ? (part * t.n_substeps) / parts : 0, s1 = split ? ((part + 1) * t.n_substeps) / parts : t.n_substeps;
  if (forward_only) GrxEngine<S>::grx_forward_euler(&m, &c, 0, lane_);
  else GrxAdroit<S>::grx_adroit_sim_world(&m, &t, &c, b.action + (size_t)w * nu, b.act_mean, b.act_rng, lane_, s0, s1);
  int wl;
  if (in_lane) wl = w;
  else if (b.compact) wl = (int)b.compact[blockIdx.x];
  else { const unsigned bx = grx_block_late(), G = gridDim.x / (unsigned)parts, sl = split ? bx % G : bx; wl = b.order ? b.order[sl] : (int)((sl & 7u) * (G >> 3) + (sl >> 3)); }   // (recomputed, not kept live across the simulation)
  if (split && !last_part) {   // an earlier part: the state goes to the world's carrier row (write-through + drained flag: GRX_SPLIT_DRAIN), nothing else is written
    volatile int* st = b.split_state + 4 * (size_t)wl;
    if (grx_lane_overflowed(c)) { if (lane_ == 0) st[0] = -1; }      // the re-run on the large tables is booked: the later parts return
    else {
      GRX_SPLIT_ROW* row = b.split_rows + (size_t)wl * b.split_stride;
      __syncthreads();
      for (int i = lane_; i < nq; i += 64) row[i] = c.qpos[i];
      for (int i = lane_; i < nv; i += 64) { row[nq + i] = c.qvel[i]; row[nq + nv + i] = c.qacc_ws[i]; }
      GRX_SPLIT_DRAIN();
      __syncthreads();
      if (lane_ == 0) {
        st[1] = (part > 0 ? st[1] : 0) | c.cnt[2];
        if (timed) { const int t0 = ((volatile int*)b.cost)[wl]; st[2] = (part > 0 ? st[2] : 0) + (((int)wall_clock64() - t0) >> 3); }
        GRX_SPLIT_DRAIN();
        st[0] = part + 1;
      }
    }
    return;
  }
  int earlier = 0;
  if (split) {   // the last part: the flags and the measured time of the earlier parts; the words are clean for the next launch
    if (lane_ == 0) { volatile int* st = b.split_state + 4 * (size_t)wl; c.cnt[2] |= st[1]; earlier = st[2]; st[0] = 0; st[1] = 0; st[2] = 0; }
    __syncthreads();
  }
  if (timed && lane_ == 0) { const int t0 = ((volatile int*)b.cost)[wl]; b.cost[wl] = (((int)wall_clock64() - t0) >> 3) + earlier; }   // measured duration of this world, 80 ns units
  if (grx_lane_overflowed(c)) return;   // capacity overflow: keep nothing, re-run on the large tables
  if (in_lane) grx_lane_ticket(b.lane, c.cnt[2] & 0xFFFF, wl, lane_); else if (!forward_only) grx_lane_join(b.lane, c, wl, lane_);
  GrxAdroit<S>::grx_adroit_outputs(&m, &t, &c, b.target ? b.target + (size_t)wl * 3 : nullptr, b.obs + (size_t)wl * t.obs_dim, b.reward + wl, b.success + wl, lane_);
  __syncthreads();
  for (int i = lane_; i < nq; i += 64) b.qpos[(size_t)wl * nq + i] = c.qpos[i];
  for (int i = lane_; i < nv; i += 64) { b.qvel[(size_t)wl * nv + i] = c.qvel[i]; b.qacc_ws[(size_t)wl * nv + i] = c.qacc_ws[i]; }
  if (lane_ == 0) b.status[wl] = grx_status_word(b.status[wl], c.cnt[2]);
#ifdef GRX_PROFILE
  GRX_TICK(&c, GRX_P_OTHER);
  if (lane_ == 0) for (int k = 0; k < GRX_NPROF; k++) atomicAdd((unsigned long long*)&g_grx_prof[k], (unsigned long long)c.prof[k]);
#endif
}
template <class S>
__global__ void __launch_bounds__(64, 2)
grx_adroit_step_kernel(int mslot, GrxAdroitTask t, GrxAdroitBuffers b, int n_worlds, int words, int forward_only) {
  extern __shared__ float lds[];
  const int lane_ = threadIdx.x;
  const int parts = (b.split_parts > 1 && !b.compact && !forward_only) ? b.split_parts : 1;      // (one part: G = the grid, part 0, slot = the workgroup: grx_world_of_block)
  const unsigned G = gridDim.x / (unsigned)parts, part = blockIdx.x / G, slot = blockIdx.x - part * G;   // slot, slot + G, ... share blockIdx.x mod 8; ONE call site of the step (code size, compile time)
  const int w = b.compact ? ((int)blockIdx.x < b.n_compact ? (int)b.compact[blockIdx.x] : n_worlds) : (b.order ? b.order[slot] : (int)((slot & 7u) * (G >> 3) + (slot >> 3)));
  grx_adroit_step_world<S>(mslot, t, b, w, n_worlds, words, forward_only, lds, lane_, false, (int)part, parts);
  grx_lane_progress(b.lane);
}
// the large-table kernel of the overflow lane (grx_overflow_lane): a small fixed grid walks the compacted list of worlds; generic shape only
template <class S>
__global__ void __launch_bounds__(64, 2)
grx_adroit_lane_kernel(int mslot, GrxAdroitTask t, GrxAdroitBuffers b, int n_worlds, int words) {   // one workgroup per entry of the compacted list (see grx_fetch_lane_kernel)
  extern __shared__ float lds[];
  const int e = blockIdx.x, nstand = (int)gridDim.x - b.lane.poll_grid;
  if (e >= nstand) { const int w = grx_lane_poll(b.lane, e - nstand); if (w >= 0) grx_adroit_step_world<S>(mslot, t, b, w, n_worlds, words, 0, lds, (int)threadIdx.x, true); return; }
  if (e >= *b.lane.count || grx_lane_taken(b.lane, e)) return;
  grx_adroit_step_world<S>(mslot, t, b, b.lane.list[e], n_worlds, words, 0, lds, (int)threadIdx.x, true);
}

// FrankaKitchen-v1 env.step() (or, forward_only, the reset-time mj_forward + observation): one wavefront per world; 40 substeps; nv = 29 (9 robot dofs, 5 joint
// equalities knob <-> burner / switch <-> light, the free kettle), 124 colliding geoms / 3 736 candidate pairs, condim-6 finger pads, hull pairs
#ifndef GRX_KITCHEN_CAP      // rows, pool words, touch zones, contacts of the kitchen's FAST kernel (the overflow lane steps the worlds that exceed them): envs/kitchen_spec.py KITCHEN_CAPACITY must agree
#define GRX_KITCHEN_CAP 128, 1280, 0, 24   // 26.1 KB = 6 worlds per CU (rounds 2 - 4: 192 / 2 240 / 32 = 31.9 KB = 5): +5.7 % with the lane taking more worlds (profiles/ab_r05_kitchen_capacity.txt)
#endif
typedef GrxShape<30, 29, 9, 25, 24, 124, 0, 0, 6, 0, GRX_KITCHEN_CAP, 1, 3> GrxShapeKitchen;
typedef GrxShape<30, 29, 9, 25, 24, 124, 0, 0, 6, 0, 400, 8160, 0, 64, 1, 3> GrxShapeKitchenLane;   // overflow-lane tables (see GrxShapeFetchPickLane)
template <class S>
__device__ __forceinline__ void grx_kitchen_step_world(int mslot, const GrxKitchenTask& t, const GrxKitchenBuffers& b, const int w, int n_worlds, int words, int forward_only, float* lds, const int lane_, const bool in_lane,
                                                       const int part = 0, const int parts = 1) {
  if (w >= n_worlds) return;
  if (b.mask && !b.mask[w]) { if (in_lane) grx_lane_ticket(b.lane, -1, w, lane_); return; }
  if (!in_lane && !forward_only && b.lane.skip && b.lane.skip[w]) return;   // in the overflow lane: stepped by the large-table kernel (reset-time forward passes cover every masked world)
  // SPLIT STEP (include/grx_capi.h grx_kitchen_buffers.split_parts; see grx_adroit_step_world): part `part` of `parts` workgroups of this world, each running its share of the 40 substeps
  const bool split = parts > 1, last_part = part == parts - 1;
  if (split && part > 0) {
    volatile int* st = b.split_state + 4 * (size_t)w;
    int v = 0;
    for (int spins = 0; spins < GRX_SPLIT_SPIN_LIMIT; spins++) {
      v = __builtin_amdgcn_readfirstlane(st[0]);
      if (v == part || v < 0) break;
      __builtin_amdgcn_s_sleep(8);
    }
    if (v != part) {   // the earlier part booked the world's re-run (v < 0), or never came (flagged): nothing to do here; the last part leaves the words clean
      if (lane_ == 0) { if (v >= 0) b.status[w] |= GRX_ST_BADNUM | (GRX_ST_BADNUM << 16); if (last_part) { st[0] = 0; st[1] = 0; st[2] = 0; } }
      return;
    }
    GRX_SPLIT_ACQUIRE();
  }
  const GrxModel& m = g_grx_models[mslot];
  GrxCtx c;
  c.mslot = mslot;
  grx_ctx_carve(&c, lds, grx_shape_dims<S>(m));
#ifdef GRX_PROFILE
  __shared__ long long prof_s[GRX_NPROF + 1];
  c.prof = prof_s; c.prof_last = prof_s + GRX_NPROF;
  if (lane_ == 0) { for (int k = 0; k < GRX_NPROF; k++) prof_s[k] = 0; prof_s[GRX_NPROF] = clock64(); }
#endif
  const int nq = S::kFixed ? S::NQ : m.nq, nv = S::kFixed ? S::NV : m.nv;
  for (int i = lane_; i < words; i += 64) lds[i] = 0.0f;
  __syncthreads();
  if (split && part > 0) {   // the row the part before this one wrote on another CU during this launch: cache-bypassing loads
    volatile const float* row = b.split_rows + (size_t)w * b.split_stride;
    for (int i = lane_; i < nq; i += 64) c.qpos[i] = row[i];
    for (int i = lane_; i < nv; i += 64) { c.qvel[i] = row[nq + i]; c.qacc_ws[i] = row[nq + nv + i]; }
  } else {
  for (int i = lane_; i < nq; i += 64) c.qpos[i] = b.qpos[(size_t)w * nq + i];
  for (int i = lane_; i < nv; i += 64) { c.qvel[i] = b.qvel[(size_t)w * nv + i]; c.qacc_ws[i] = b.qacc_ws[(size_t)w * nv + i]; }
  }
  __syncthreads();
  float* last = b.last_qpos + (size_t)w * GRX_KITCHEN_NROBOT;
  if (b.skin) { c.skin = b.skin + (size_t)w * b.skin_stride; c.skin_r = b.skin_radius; }
  if (split && part > 0 && b.skin) {   // the world's skin list was (re)written by another CU during this launch with plain stores: this part does not read it, it REBUILDS it in its first substep (results are identical with any valid list or none)
    if (lane_ == 0) ((volatile int*)c.skin)[1] = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  grx_lane_setup(b.lane, c, w, !forward_only);
  const bool timed = b.cost && !forward_only && !in_lane;   // cost-ordered dispatch (include/grx_capi.h grx_kitchen_buffers.order / .cost): the start stamp (100 MHz) is parked in the cost slot itself
  if (timed && lane_ == 0) b.cost[w] = (int)wall_clock64();
  const int s0 = split ? (part * t.n_substeps) / parts : 0, s1 = split ? ((part + 1) * t.n_substeps) / parts : t.n_substeps;
  if (forward_only) GrxEngine<S>::grx_forward_euler(&m, &c, 0, lane_);
  else GrxKitchen<S>::grx_kitchen_sim_world(&m, &t, &c, b.action + (size_t)w * GRX_KITCHEN_NROBOT, last, lane_, s0, s1);
  if (split && !last_part) {   // an earlier part: the state goes to the world's carrier row (write-through + drained flag: GRX_SPLIT_DRAIN), nothing else is written
    volatile int* st = b.split_state + 4 * (size_t)w;
    if (grx_lane_overflowed(c)) { if (lane_ == 0) st[0] = -1; }      // the re-run on the large tables is booked: the later parts return
    else {
      GRX_SPLIT_ROW* row = b.split_rows + (size_t)w * b.split_stride;
      __syncthreads();
      for (int i = lane_; i < nq; i += 64) row[i] = c.qpos[i];
      for (int i = lane_; i < nv; i += 64) { row[nq + i] = c.qvel[i]; row[nq + nv + i] = c.qacc_ws[i]; }
      GRX_SPLIT_DRAIN();
      __syncthreads();
      if (lane_ == 0) {
        st[1] = (part > 0 ? st[1] : 0) | c.cnt[2];
        if (timed) { const int t0 = ((volatile int*)b.cost)[w]; st[2] = (part > 0 ? st[2] : 0) + (((int)wall_clock64() - t0) >> 3); }
        GRX_SPLIT_DRAIN();
        st[0] = part + 1;
      }
    }
    return;
  }
  int earlier = 0;
  if (split) {   // the last part: the flags and the measured time of the earlier parts; the words are clean for the next launch
    if (lane_ == 0) { volatile int* st = b.split_state + 4 * (size_t)w; c.cnt[2] |= st[1]; earlier = st[2]; st[0] = 0; st[1] = 0; st[2] = 0; }
    __syncthreads();
  }
  if (timed && lane_ == 0) { const int t0 = ((volatile int*)b.cost)[w]; b.cost[w] = (((int)wall_clock64() - t0) >> 3) + earlier; }   // measured duration of this world, 80 ns units (a world that overflowed a table too: it ran up to there)
  if (grx_lane_overflowed(c)) return;   // capacity overflow: keep nothing (last_qpos included), re-run on the large tables
  if (in_lane) grx_lane_ticket(b.lane, c.cnt[2] & 0xFFFF, w, lane_); else if (!forward_only) grx_lane_join(b.lane, c, w, lane_);
  GrxKitchen<S>::grx_kitchen_outputs(&m, &t, &c, b.noise ? b.noise + (size_t)w * t.obs_dim : nullptr, b.obs + (size_t)w * t.obs_dim, last, b.completed + w, lane_);
  __syncthreads();
  for (int i = lane_; i < nq; i += 64) b.qpos[(size_t)w * nq + i] = c.qpos[i];
  for (int i = lane_; i < nv; i += 64) { b.qvel[(size_t)w * nv + i] = c.qvel[i]; b.qacc_ws[(size_t)w * nv + i] = c.qacc_ws[i]; }
  if (lane_ == 0) b.status[w] = grx_status_word(b.status[w], c.cnt[2]);
#ifdef GRX_PROFILE
  GRX_TICK(&c, GRX_P_OTHER);
  if (lane_ == 0) for (int k = 0; k < GRX_NPROF; k++) atomicAdd((unsigned long long*)&g_grx_prof[k], (unsigned long long)c.prof[k]);
  if (lane_ == 0 && w < 4096 && !forward_only) for (int k = 0; k < GRX_NPROF; k++) g_grx_world_prof[w * GRX_NPROF + k] = (int)c.prof[k];   // per-world rows (tools/straggler_probe.py): lane worlds included
#endif
}
template <class S>
__global__ void __launch_bounds__(64, 2)
grx_kitchen_step_kernel(int mslot, GrxKitchenTask t, GrxKitchenBuffers b, int n_worlds, int words, int forward_only) {
  extern __shared__ float lds[];
  const int lane_ = threadIdx.x;
  const int parts = (b.split_parts > 1 && !forward_only) ? b.split_parts : 1;      // (one part: G = the grid, part 0, slot = the workgroup: grx_world_of_block)
  const unsigned G = gridDim.x / (unsigned)parts, part = blockIdx.x / G, slot = blockIdx.x - part * G;   // slot, slot + G, ... share blockIdx.x mod 8; ONE call site of the step (code size, compile time)
  grx_kitchen_step_world<S>(mslot, t, b, b.order ? b.order[slot] : (int)((slot & 7u) * (G >> 3) + (slot >> 3)), n_worlds, words, forward_only, lds, lane_, false, (int)part, parts);
  grx_lane_progress(b.lane);
}
// the large-table kernel of the overflow lane (grx_overflow_lane): a small fixed grid walks the compacted list of worlds; generic shape only
template <class S>
__global__ void __launch_bounds__(64, 2)
grx_kitchen_lane_kernel(int mslot, GrxKitchenTask t, GrxKitchenBuffers b, int n_worlds, int words) {   // one workgroup per entry of the compacted list (see grx_fetch_lane_kernel)
  extern __shared__ float lds[];
  const int e = blockIdx.x, nstand = (int)gridDim.x - b.lane.poll_grid;
  if (e >= nstand) { const int w = grx_lane_poll(b.lane, e - nstand); if (w >= 0) grx_kitchen_step_world<S>(mslot, t, b, w, n_worlds, words, 0, lds, (int)threadIdx.x, true); return; }
  if (e >= *b.lane.count || grx_lane_taken(b.lane, e)) return;
  grx_kitchen_step_world<S>(mslot, t, b, b.lane.list[e], n_worlds, words, 0, lds, (int)threadIdx.x, true);
}

// ------------------------------------------------------------------------------------------
// per-family translation units: shape selection, LDS limits, descriptor upload, launches
// ------------------------------------------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the kernel, not of a model: several live models share the generic
// kernels, so the attribute is only ever raised (a later, smaller model must not lower the limit of an earlier, larger one).
#include <map>
static hipError_t grx_raise_lds_limit(const void* fn, int bytes) {
  static std::map<const void*, int> limit;
  int& cur = limit[fn];
  if (bytes <= cur) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) cur = bytes;
  return e;
}
static hipError_t grx_upload_descriptor(const GrxModel* g, int slot) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_grx_models), g, sizeof(GrxModel), sizeof(GrxModel) * (size_t)slot, hipMemcpyHostToDevice);
}
#define GRX_LDS(KERNEL) do { hipError_t e_ = grx_raise_lds_limit((const void*)(KERNEL), bytes); if (e_ != hipSuccess) return (int)e_; } while (0)
// grx_tu_*_prepare: returns a hipError_t (0 = ok); *shape <- id of the shape-specialised kernels that serve the model (unchanged if none)

#if GRX_TU_FETCH
#define GRX_FETCH_SHAPES(X) X(1, GrxShapeFetchPick) X(2, GrxShapeFetchObject) X(7, GrxShapeFetchPuck) X(3, GrxShapeFetchArm)
extern "C" int grx_tu_fetch_prepare(const GrxModel* g, int bytes, int slot, int* shape) {
  GRX_LDS(grx_fetch_step_kernel<GrxShapeAny>); GRX_LDS(grx_fetch_forward_kernel<GrxShapeAny>); GRX_LDS(grx_fetch_reset_kernel<GrxShapeAny>); GRX_LDS(grx_fetch_lane_kernel<GrxShapeAny>);
  int found = 0;
#define X(ID, SHAPE) if (!found && grx_shape_matches<SHAPE>(*g)) { found = ID; GRX_LDS(grx_fetch_step_kernel<SHAPE>); GRX_LDS(grx_fetch_forward_kernel<SHAPE>); GRX_LDS(grx_fetch_reset_kernel<SHAPE>); }
  GRX_FETCH_SHAPES(X)
#undef X
  if (found == 1) GRX_LDS(grx_fetch_lane_kernel<GrxShapeFetchPick>);   // today's FetchPickAndPlace kernel as the standing lane of the fast one (round 6)
  if (!found && grx_shape_matches<GrxShapeFetchPickFast>(*g)) { found = 8; GRX_LDS(grx_fetch_step_kernel<GrxShapeFetchPickFast>); }   // step kernel only
  if (!found && grx_shape_matches<GrxShapeFetchPickLane>(*g)) { found = 101; GRX_LDS(grx_fetch_lane_kernel<GrxShapeFetchPickLane>); }   // ids >= 100: lane kernel only (everything else of such a model runs generic)
  if (found) *shape = found;
  return (int)grx_upload_descriptor(g, slot);
}
// kind 0: env.step, 1: forward (nstep), 2: compacted reset
extern "C" int grx_tu_fetch_launch(int kind, int shape, unsigned grid, size_t lds_bytes, void* stream, int slot, const GrxFetchTask* t, const GrxFetchBuffers* b,
                                   const GrxFetchResetArgs* r, int n, int words, int nstep) {
  const dim3 g(grid), blk(64); hipStream_t st = (hipStream_t)stream;
  if (kind == 0 && b->lane.list) {
    if (shape == 101) hipLaunchKernelGGL(grx_fetch_lane_kernel<GrxShapeFetchPickLane>, g, blk, lds_bytes, st, slot, *t, *b, n, words);
    else if (shape == 1) hipLaunchKernelGGL(grx_fetch_lane_kernel<GrxShapeFetchPick>, g, blk, lds_bytes, st, slot, *t, *b, n, words);
    else hipLaunchKernelGGL(grx_fetch_lane_kernel<GrxShapeAny>, g, blk, lds_bytes, st, slot, *t, *b, n, words);
    return (int)hipGetLastError();
  }
  if (shape == 8) {   // the fast FetchPickAndPlace model: its shape exists as a step kernel only
    if (kind != 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(grx_fetch_step_kernel<GrxShapeFetchPickFast>, g, blk, lds_bytes, st, slot, *t, *b, n, words);
    return (int)hipGetLastError();
  }
#define GRX_FETCH_GO(SHAPE) do { \
    if (kind == 0) hipLaunchKernelGGL(grx_fetch_step_kernel<SHAPE>, g, blk, lds_bytes, st, slot, *t, *b, n, words); \
    else if (kind == 1) hipLaunchKernelGGL(grx_fetch_forward_kernel<SHAPE>, g, blk, lds_bytes, st, slot, *t, *b, n, words, nstep); \
    else hipLaunchKernelGGL(grx_fetch_reset_kernel<SHAPE>, g, blk, lds_bytes, st, slot, *t, *b, *r, n, words); } while (0)
  switch (shape) {   // the specialised kernels are bit-identical to the generic one (same source, dims folded)
#define X(ID, SHAPE) case ID: GRX_FETCH_GO(SHAPE); break;
    GRX_FETCH_SHAPES(X)
#undef X
    default: GRX_FETCH_GO(GrxShapeAny);
  }
#undef GRX_FETCH_GO
  return (int)hipGetLastError();
}
#endif

#if GRX_TU_POINT
#define GRX_POINT_SHAPES(X) X(10, GrxShapeAntLarge) X(11, GrxShapeAntMedium) X(12, GrxShapeAntOpen) X(13, GrxShapeAntUMaze)
extern "C" int grx_tu_point_prepare(const GrxModel* g, int bytes, int slot, int* shape) {
  GRX_LDS(grx_point_step_kernel<GrxShapeAny>);
#define X(ID, SHAPE) if (grx_shape_matches<SHAPE>(*g)) { *shape = ID; GRX_LDS(grx_point_step_kernel<SHAPE>); }
  GRX_POINT_SHAPES(X)
#undef X
  return (int)grx_upload_descriptor(g, slot);
}
extern "C" int grx_tu_point_launch(int shape, unsigned grid, size_t lds_bytes, void* stream, int slot, const GrxPointTask* t, const GrxPointBuffers* b, int n, int words) {
  const dim3 g(grid), blk(64); hipStream_t st = (hipStream_t)stream;
  switch (shape) {
#define X(ID, SHAPE) case ID: hipLaunchKernelGGL(grx_point_step_kernel<SHAPE>, g, blk, lds_bytes, st, slot, *t, *b, n, words); break;
    GRX_POINT_SHAPES(X)
#undef X
    default: hipLaunchKernelGGL(grx_point_step_kernel<GrxShapeAny>, g, blk, lds_bytes, st, slot, *t, *b, n, words);
  }
  return (int)hipGetLastError();
}
#endif

#if GRX_TU_HAND
#define GRX_HAND_SHAPES(X) X(4, GrxShapeHandReach) X(6, GrxShapeHandBlockTouch) X(8, GrxShapeHandEgg) X(9, GrxShapeHandEggTouch) X(5, GrxShapeHandBlock)
extern "C" int grx_tu_hand_prepare(const GrxModel* g, int bytes, int slot, int* shape) {
  GRX_LDS(grx_hand_step_kernel<GrxShapeAny>); GRX_LDS(grx_hand_lane_kernel<GrxShapeAny>);
#define X(ID, SHAPE) if (grx_shape_matches<SHAPE>(*g)) { *shape = ID; GRX_LDS(grx_hand_step_kernel<SHAPE>); }
  GRX_HAND_SHAPES(X)
#undef X
  if (grx_shape_matches<GrxShapeHandBlockTouchLane>(*g)) { *shape = 106; GRX_LDS(grx_hand_lane_kernel<GrxShapeHandBlockTouchLane>); }   // ids >= 100: lane kernel only
  return (int)grx_upload_descriptor(g, slot);
}
extern "C" int grx_tu_hand_launch(int shape, unsigned grid, size_t lds_bytes, void* stream, int slot, const GrxHandTask* t, const GrxHandBuffers* b, int n, int words,
                                  int forward_only) {
  const dim3 g(grid), blk(64); hipStream_t st = (hipStream_t)stream;
  if (b->lane.list) {
    if (shape == 106) hipLaunchKernelGGL(grx_hand_lane_kernel<GrxShapeHandBlockTouchLane>, g, blk, lds_bytes, st, slot, *t, *b, n, words);
    else hipLaunchKernelGGL(grx_hand_lane_kernel<GrxShapeAny>, g, blk, lds_bytes, st, slot, *t, *b, n, words);
    return (int)hipGetLastError();
  }
  switch (shape) {
#define X(ID, SHAPE) case ID: hipLaunchKernelGGL(grx_hand_step_kernel<SHAPE>, g, blk, lds_bytes, st, slot, *t, *b, n, words, forward_only); break;
    GRX_HAND_SHAPES(X)
#undef X
    default: hipLaunchKernelGGL(grx_hand_step_kernel<GrxShapeAny>, g, blk, lds_bytes, st, slot, *t, *b, n, words, forward_only);
  }
  return (int)hipGetLastError();
}
#endif
#if GRX_TU_ADROIT
extern "C" int grx_tu_adroit_prepare(const GrxModel* g, int bytes, int slot, int* shape) {
  GRX_LDS(grx_adroit_step_kernel<GrxShapeAny>); GRX_LDS(grx_adroit_lane_kernel<GrxShapeAny>);
#define GRX_ADROIT_SHAPES(X) X(20, GrxShapeAdroitHammer) X(21, GrxShapeAdroitDoor) X(22, GrxShapeAdroitPen) X(23, GrxShapeAdroitRelocate)
  int found = 0;
#define X(ID, SHAPE) if (!found && grx_shape_matches<SHAPE>(*g)) { found = ID; GRX_LDS(grx_adroit_step_kernel<SHAPE>); }
  GRX_ADROIT_SHAPES(X)
#undef X
  if (found) *shape = found;
  if (grx_shape_matches<GrxShapeAdroitDoorLane>(*g)) { *shape = 121; GRX_LDS(grx_adroit_lane_kernel<GrxShapeAdroitDoorLane>); }            // ids >= 100: lane kernel only
  if (grx_shape_matches<GrxShapeAdroitRelocateLane>(*g)) { *shape = 123; GRX_LDS(grx_adroit_lane_kernel<GrxShapeAdroitRelocateLane>); }
  return (int)grx_upload_descriptor(g, slot);
}
extern "C" int grx_tu_adroit_launch(int shape, unsigned grid, size_t lds_bytes, void* stream, int slot, const GrxAdroitTask* t, const GrxAdroitBuffers* b, int n, int words,
                                    int forward_only) {
  const dim3 g(grid), blk(64); hipStream_t st = (hipStream_t)stream;
  if (b->lane.list) {
    if (shape == 121) hipLaunchKernelGGL(grx_adroit_lane_kernel<GrxShapeAdroitDoorLane>, g, blk, lds_bytes, st, slot, *t, *b, n, words);
    else if (shape == 123) hipLaunchKernelGGL(grx_adroit_lane_kernel<GrxShapeAdroitRelocateLane>, g, blk, lds_bytes, st, slot, *t, *b, n, words);
    else hipLaunchKernelGGL(grx_adroit_lane_kernel<GrxShapeAny>, g, blk, lds_bytes, st, slot, *t, *b, n, words);
    return (int)hipGetLastError();
  }
  switch (shape) {
#define X(ID, SHAPE) case ID: hipLaunchKernelGGL(grx_adroit_step_kernel<SHAPE>, g, blk, lds_bytes, st, slot, *t, *b, n, words, forward_only); break;
    GRX_ADROIT_SHAPES(X)
#undef X
    default: hipLaunchKernelGGL(grx_adroit_step_kernel<GrxShapeAny>, g, blk, lds_bytes, st, slot, *t, *b, n, words, forward_only);
  }
  return (int)hipGetLastError();
}
#endif
#if GRX_TU_KITCHEN
extern "C" int grx_tu_kitchen_prepare(const GrxModel* g, int bytes, int slot, int* shape) {
  GRX_LDS(grx_kitchen_step_kernel<GrxShapeAny>); GRX_LDS(grx_kitchen_lane_kernel<GrxShapeAny>);
  if (grx_shape_matches<GrxShapeKitchen>(*g)) { *shape = 30; GRX_LDS(grx_kitchen_step_kernel<GrxShapeKitchen>); }
  if (grx_shape_matches<GrxShapeKitchenLane>(*g)) { *shape = 130; GRX_LDS(grx_kitchen_lane_kernel<GrxShapeKitchenLane>); }   // ids >= 100: lane kernel only
  return (int)grx_upload_descriptor(g, slot);
}
extern "C" int grx_tu_kitchen_launch(int shape, unsigned grid, size_t lds_bytes, void* stream, int slot, const GrxKitchenTask* t, const GrxKitchenBuffers* b, int n, int words,
                                     int forward_only) {
  const dim3 g(grid), blk(64); hipStream_t st = (hipStream_t)stream;
  if (b->lane.list) {
    if (shape == 130) hipLaunchKernelGGL(grx_kitchen_lane_kernel<GrxShapeKitchenLane>, g, blk, lds_bytes, st, slot, *t, *b, n, words);
    else hipLaunchKernelGGL(grx_kitchen_lane_kernel<GrxShapeAny>, g, blk, lds_bytes, st, slot, *t, *b, n, words);
    return (int)hipGetLastError();
  }
  if (shape == 30) hipLaunchKernelGGL(grx_kitchen_step_kernel<GrxShapeKitchen>, g, blk, lds_bytes, st, slot, *t, *b, n, words, forward_only);
  else hipLaunchKernelGGL(grx_kitchen_step_kernel<GrxShapeAny>, g, blk, lds_bytes, st, slot, *t, *b, n, words, forward_only);
  return (int)hipGetLastError();
}
#endif
#undef GRX_LDS

#if GRX_TU_API
extern "C" int grx_tu_adroit_prepare(const GrxModel* g, int bytes, int slot, int* shape);
extern "C" int grx_tu_adroit_launch(int shape, unsigned grid, size_t lds_bytes, void* stream, int slot, const GrxAdroitTask* t, const GrxAdroitBuffers* b, int n, int words,
                                    int forward_only);
extern "C" int grx_tu_kitchen_prepare(const GrxModel* g, int bytes, int slot, int* shape);
extern "C" int grx_tu_kitchen_launch(int shape, unsigned grid, size_t lds_bytes, void* stream, int slot, const GrxKitchenTask* t, const GrxKitchenBuffers* b, int n, int words,
                                     int forward_only);
extern "C" int grx_tu_fetch_prepare(const GrxModel* g, int bytes, int slot, int* shape);
extern "C" int grx_tu_point_prepare(const GrxModel* g, int bytes, int slot, int* shape);
extern "C" int grx_tu_hand_prepare(const GrxModel* g, int bytes, int slot, int* shape);
extern "C" int grx_tu_fetch_launch(int kind, int shape, unsigned grid, size_t lds_bytes, void* stream, int slot, const GrxFetchTask* t, const GrxFetchBuffers* b,
                                   const GrxFetchResetArgs* r, int n, int words, int nstep);
extern "C" int grx_tu_point_launch(int shape, unsigned grid, size_t lds_bytes, void* stream, int slot, const GrxPointTask* t, const GrxPointBuffers* b, int n, int words);
extern "C" int grx_tu_hand_launch(int shape, unsigned grid, size_t lds_bytes, void* stream, int slot, const GrxHandTask* t, const GrxHandBuffers* b, int n, int words,
                                  int forward_only);

extern "C" __global__ void __launch_bounds__(256)
grx_goal_reward_kernel(const float* __restrict__ ag, const float* __restrict__ dg, long long B, int dim, double thr, int sparse, float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (long long)gridDim.x * blockDim.x)
    out[i] = grx_hand_reward(grx_goal_distance_n(ag + i * dim, dg + i * dim, dim), thr, sparse);
}

extern "C" __global__ void __launch_bounds__(256)
grx_manip_reward_kernel(const float* __restrict__ ag, const float* __restrict__ dg, long long B, int ignore_pos, int ignore_rot, int ignore_z, float thr_pos, float thr_rot,
                        int sparse, float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (long long)gridDim.x * blockDim.x) {
    float dp, dr;
    grx_manip_distance(ag + i * 7, dg + i * 7, ignore_pos, ignore_rot, ignore_z, &dp, &dr);
    out[i] = grx_manip_reward(dp, dr, thr_pos, thr_rot, sparse);
  }
}

extern "C" __global__ void __launch_bounds__(256)
grx_maze_reward_kernel(const float* __restrict__ ag, const float* __restrict__ dg, long long B, double radius, int sparse, float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (long long)gridDim.x * blockDim.x) {
    float a[2] = {ag[2 * i], ag[2 * i + 1]}, g[2] = {dg[2 * i], dg[2 * i + 1]};
    out[i] = grx_maze_reward(grx_goal_distance2(a, g), radius, sparse);
  }
}

// HER relabel: reward for B (achieved, desired) pairs; 16 B/lane loads where the layout allows
extern "C" __global__ void __launch_bounds__(256)
grx_fetch_reward_kernel(const float* __restrict__ ag, const float* __restrict__ dg, long long B, double thresh, int sparse, float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (long long)gridDim.x * blockDim.x) {
    float a[3] = {ag[3 * i], ag[3 * i + 1], ag[3 * i + 2]}, g[3] = {dg[3 * i], dg[3 * i + 1], dg[3 * i + 2]};
    out[i] = grx_fetch_reward(grx_goal_distance3(a, g), thresh, sparse);
  }
}

// HER relabel + replay write (the caller of compute_reward: /root/reference/README.md:72-76, core.py:45-67).  Episode storage on the device: a ring of
// R = T + 1 rows per world, rows[r] = the packed output row [obs | achieved | desired | reward | success] the step kernels write (a reset row or the
// row after a step), acts[r] = the action that LED to row r.  Row indices are absolute step counts, taken modulo R here (an episode buffer of T steps is
// the special case that never wraps).  Sample b = (row t_idx[b], world w_idx[b], goal row t_goal[b]): the transition (row t -> action -> row t + 1)
// with the desired goal replaced by the goal ACHIEVED at row t_goal (t_goal < 0: the episode's own goal), its reward recomputed with the very
// device functions compute_reward uses, written as one packed replay row
//     [obs_t | achieved_t | goal | action_t | reward | obs_t+1 | achieved_t+1 | success]          (OW = 2 obs_dim + 3 goal_dim + act_dim + 2)
// One thread per output word (coalesced stores, gathers of whole rows): the trajectories never leave HBM.
struct GrxHerArgs {
  const float* rows; const float* acts;
  int T, N, W, obs_dim, goal_dim, act_dim;
  const int *t_idx, *w_idx, *t_goal;
  int kind;            // 0: Euclidean goals with a distance threshold (Fetch: -(d > thr) / -d), 1: same with the hand's -0.0 convention (HandReach), 2: maze, 3: manipulate pose goals
  double p0, p1;       // kind 0 / 1: threshold; 2: goal radius; 3: position threshold, rotation threshold (fp64: the reference's own compare, see grx_goal_distance3)
  int sparse, ignore_pos, ignore_rot, ignore_z;
  float* out;
  const float* term_rows; const int* term_t;   // terminal rows of the episodes that ended under same-step autoreset (include/grx_capi.h)
};
static_assert(sizeof(grx_her_args) == sizeof(GrxHerArgs), "grx_her_args must mirror GrxHerArgs");
GRX_DEV void grx_her_outcome(const GrxHerArgs& a, const float* ag, const float* g, float* reward, float* success) {
  if (a.kind == 3) {
    float dp, dr;
    grx_manip_distance(ag, g, a.ignore_pos, a.ignore_rot, a.ignore_z, &dp, &dr);
    *reward = grx_manip_reward(dp, dr, (float)a.p0, (float)a.p1, a.sparse); *success = grx_manip_success(dp, dr, (float)a.p0, (float)a.p1) ? 1.0f : 0.0f;
  } else if (a.kind == 2) {
    const double d = grx_goal_distance2(ag, g);
    *reward = grx_maze_reward(d, a.p0, a.sparse); *success = (d <= a.p0) ? 1.0f : 0.0f;
  } else if (a.kind == 1) {
    const double d = grx_goal_distance_n(ag, g, a.goal_dim);
    *reward = grx_hand_reward(d, a.p0, a.sparse); *success = (d < a.p0) ? 1.0f : 0.0f;
  } else {
    const double d = grx_goal_distance3(ag, g);
    *reward = grx_fetch_reward(d, a.p0, a.sparse); *success = (d < a.p0) ? 1.0f : 0.0f;
  }
}
extern "C" __global__ void __launch_bounds__(256)
grx_her_relabel_kernel(GrxHerArgs a, long long B) {
  const int od = a.obs_dim, gd = a.goal_dim, ad = a.act_dim, OW = 2 * od + 3 * gd + ad + 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < B * OW; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / OW; const int e = (int)(i - b * OW);
    const int R = a.T + 1, t = a.t_idx[b] % R, t1 = (a.t_idx[b] + 1) % R, w = a.w_idx[b], tg = a.t_goal[b];
    const int tt = a.term_t ? a.term_t[w] : -1;   // the row index whose ring entry is the first row of a new episode; the terminal row is in term_rows[w]
    const float* r0 = a.rows + ((size_t)t * a.N + w) * a.W;
    const float* r1 = (a.t_idx[b] + 1 == tt) ? a.term_rows + (size_t)w * a.W : a.rows + ((size_t)t1 * a.N + w) * a.W;
    const float* g = tg < 0 ? r0 + od + gd : ((tg == tt ? a.term_rows + (size_t)w * a.W : a.rows + ((size_t)(tg % R) * a.N + w) * a.W) + od);   // the substituted goal: achieved at row tg
    float v;
    if (e < od + gd) v = r0[e];
    else if (e < od + 2 * gd) v = g[e - od - gd];
    else if (e < od + 2 * gd + ad) v = a.acts[((size_t)t1 * a.N + w) * ad + (e - od - 2 * gd)];
    else if (e == od + 2 * gd + ad || e == OW - 1) {
      float ag[16], gg[16], rw, sc;
      for (int k = 0; k < gd; k++) { ag[k] = r1[od + k]; gg[k] = g[k]; }
      grx_her_outcome(a, ag, gg, &rw, &sc);
      v = (e == OW - 1) ? sc : rw;
    } else v = r1[e - (od + 2 * gd + ad + 1)];
    a.out[i] = v;
  }
}

// unit-test hook for the two GPU-specific numerical primitives (register-resident solve, MFMA Hessian)
extern "C" __global__ void __launch_bounds__(64)
grx_debug_kernel(int mode, int nv, int nefc, const float* A_in, const float* b_in, const float* J_in, const float* D_in, float* out) {
  extern __shared__ float lds[];
  const int lane_ = threadIdx.x;
  GrxModel m; m.nv = nv;
  GrxCtx c;
  c.A = lds; c.M = c.A + nv * nv; c.Jp = c.M + nv * nv; c.efc_D = c.Jp + GRX_MAXEFC * nv; c.efc_jv = c.efc_D + GRX_MAXEFC;
  c.efc_quad = (int*)(c.efc_jv + GRX_MAXEFC); c.efc_row = c.efc_quad + GRX_MAXEFC; c.tmpv = (float*)(c.efc_row + GRX_MAXEFC);
  c.grad = c.tmpv + nv; c.efc_force = c.grad + nv; c.efc_id = (int*)(c.efc_force + GRX_MAXEFC);
  for (int i = lane_; i < nv * nv; i += 64) { c.A[i] = A_in[i]; c.M[i] = A_in[i]; }
  for (int i = lane_; i < nv; i += 64) c.tmpv[i] = b_in[i];
  // mode 1: every row stored over all dofs; mode 2: every row stored as two spans [0, na) and [nb, nv) (the dofs in between must be zero)
  const int na = nv / 3, nb = nv - nv / 3, two = (mode == 2), rl = two ? na + (nv - nb) : nv;
  for (int i = lane_; i < nefc * nv; i += 64) {
    const int r = i / nv, d = i - r * nv;
    if (!two) c.Jp[i] = J_in[i];
    else if (d < na) c.Jp[r * rl + d] = J_in[i];
    else if (d >= nb) c.Jp[r * rl + na + d - nb] = J_in[i];
  }
  for (int i = lane_; i < nefc; i += 64) {
    c.efc_D[i] = fabsf(D_in[i]); c.efc_quad[i] = D_in[i] > 0 ? 1 : 0; c.efc_force[i] = 0.5f + 0.01f * (float)i;
    c.efc_row[i] = GRX_ROW_PACK(i * rl, 0, two ? na : nv); c.efc_id[i] = two ? ((nb << 12) | ((nv - nb) << 20)) : 0;
  }
  __syncthreads();
  if (mode == 0) {
    GrxEngine<GrxShapeAny>::grx_sym_solve_full(c.A, nv, c.tmpv, lane_);
    for (int i = lane_; i < nv; i += 64) out[i] = c.tmpv[i];
  } else {
    GrxEngine<GrxShapeAny>::grx_hessian(&m, &c, nefc, lane_);
    for (int i = lane_; i < nv * nv; i += 64) out[i] = c.A[i];
    for (int i = lane_; i < nv; i += 64) out[nv * nv + i] = c.grad[i];  // J' f
  }
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
static int g_slot_used[GRX_MAX_MODELS];

struct grx_model {
  GrxPackedModel pm;
  float* d_f = nullptr;
  int32_t* d_i = nullptr;
  GrxModel dev;  // table pointers address d_f / d_i
  int device = 0;
  int words = 0;
  int slot = -1;  // index of the device-side descriptor in g_grx_models (constant memory)
  int shape = 0;  // 0 = generic step kernel, else index of the specialised GrxShape
  int lds_pad = 0;  // GRX_LDS_PAD_BYTES (occupancy experiments)
};

static thread_local std::string g_err;
static int fail(const std::string& msg) { g_err = msg; return -1; }
#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

extern "C" const char* grx_last_error(void) { return g_err.c_str(); }

static void grx_model_free(grx_model* m) {
  if (!m) return;
  if (m->slot >= 0) g_slot_used[m->slot] = 0;
  (void)hipSetDevice(m->device);
  if (m->d_f) (void)hipFree(m->d_f);
  if (m->d_i) (void)hipFree(m->d_i);
  delete m;
}

static int grx_model_create_impl(const int32_t* H, const int32_t* I, const double* F, grx_model* m) {
  grx_pack_model(H, I, F, &m->pm);
  HIP_OK(hipMalloc(&m->d_f, sizeof(float) * (m->pm.f.size() + 4)));
  HIP_OK(hipMalloc(&m->d_i, sizeof(int32_t) * (m->pm.i.size() + 4)));
  HIP_OK(hipMemcpy(m->d_f, m->pm.f.data(), sizeof(float) * m->pm.f.size(), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(m->d_i, m->pm.i.data(), sizeof(int32_t) * m->pm.i.size(), hipMemcpyHostToDevice));
  m->dev = grx_bind_model(m->pm, m->d_f, m->d_i);
  const GrxModel& g = m->dev;
  m->words = grx_ctx_words(grx_dims_of(&g));
  // GRX_LDS_PAD_BYTES (occupancy experiments, tools/occupancy_sweep.py): every launch of this model asks for that much dynamic LDS on top of its working set, which lowers the
  // number of worlds resident per CU without touching the code
  m->lds_pad = getenv("GRX_LDS_PAD_BYTES") ? atoi(getenv("GRX_LDS_PAD_BYTES")) : 0;
  int bytes = m->words * 4 + m->lds_pad;
  if (bytes > 160 * 1024) return fail("model working set exceeds the 160 KiB LDS of a CU");
  if (g.njnt > 64) return fail("engine limit: at most 64 joints per world (one lane per joint)");
  if (g.nv > 64) return fail("engine limit: at most 64 dofs per world (dof-chain masks are 64-bit)");
  if (g.nbody > 64) return fail("engine limit: at most 64 bodies per world (one lane per body, 64-bit subtree masks)");
  if (g.nweld > g.maxefc / 16) return fail("engine limit: too many weld constraints (weld frames are staged in the row-parameter slot)");
  for (int k = 0; k < GRX_MAX_MODELS && m->slot < 0; k++) if (!g_slot_used[k]) { g_slot_used[k] = 1; m->slot = k; }
  if (m->slot < 0) return fail("grx_model_create: all model descriptor slots are in use (destroy a model first)");
  m->shape = 0;
  // every family unit gets the descriptor (its own constant-memory copy) and raises the LDS limit of the kernels that can serve the model
  int e;
  if ((e = grx_tu_fetch_prepare(&g, bytes, m->slot, &m->shape)) != 0) return fail(std::string("grx_model_create (fetch kernels): ") + hipGetErrorString((hipError_t)e));
  if ((e = grx_tu_point_prepare(&g, bytes, m->slot, &m->shape)) != 0) return fail(std::string("grx_model_create (point kernels): ") + hipGetErrorString((hipError_t)e));
  if ((e = grx_tu_hand_prepare(&g, bytes, m->slot, &m->shape)) != 0) return fail(std::string("grx_model_create (hand kernels): ") + hipGetErrorString((hipError_t)e));
  if ((e = grx_tu_adroit_prepare(&g, bytes, m->slot, &m->shape)) != 0) return fail(std::string("grx_model_create (adroit kernels): ") + hipGetErrorString((hipError_t)e));
  if ((e = grx_tu_kitchen_prepare(&g, bytes, m->slot, &m->shape)) != 0) return fail(std::string("grx_model_create (kitchen kernels): ") + hipGetErrorString((hipError_t)e));
  HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_grx_models), &m->dev, sizeof(GrxModel), sizeof(GrxModel) * (size_t)m->slot, hipMemcpyHostToDevice));
  return 0;
}

extern "C" int grx_model_create(const int32_t* H, int nH, const int32_t* I, int nI, const double* F, int nF, int device, grx_model** out) {
  (void)nH; (void)nI; (void)nF;
  if (!H || !I || !F || !out) return fail("grx_model_create: null argument");
  int ndev = 0;
  HIP_OK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail("grx_model_create: no such HIP device");
  HIP_OK(hipSetDevice(device));
  grx_model* m = new grx_model();
  m->device = device;
  if (grx_model_create_impl(H, I, F, m)) { grx_model_free(m); return -1; }   // every failure path releases the host object, the device tables and the descriptor slot
  *out = m;
  return 0;
}

extern "C" int grx_model_destroy(grx_model* m) {
  grx_model_free(m);
  return 0;
}

extern "C" int grx_model_set_table(grx_model* m, const char* name, const double* data, int n) {
  if (!m || !name || !data) return fail("grx_model_set_table: null argument");
  int k = grx_find_table(m->pm, name);
  if (k < 0 || m->pm.kind[k] != 'f') return fail(std::string("grx_model_set_table: no fp table named ") + name);
  if (n > m->pm.cnt[k]) return fail("grx_model_set_table: too many elements");
  for (int i = 0; i < n; i++) m->pm.f[m->pm.off[k] + i] = (float)data[i];
  HIP_OK(hipSetDevice(m->device));
  HIP_OK(hipMemcpy(m->d_f + m->pm.off[k], m->pm.f.data() + m->pm.off[k], sizeof(float) * n, hipMemcpyHostToDevice));
  // the per-stage records (GrxModel::recf_* / reci_*) are gathered from the tables: refill them in place and upload their two regions
  grx_build_records(&m->pm, false);
  if (m->pm.rec_f1 > m->pm.rec_f0) HIP_OK(hipMemcpy(m->d_f + m->pm.rec_f0, m->pm.f.data() + m->pm.rec_f0, sizeof(float) * (size_t)(m->pm.rec_f1 - m->pm.rec_f0), hipMemcpyHostToDevice));
  if (m->pm.rec_i1 > m->pm.rec_i0) HIP_OK(hipMemcpy(m->d_i + m->pm.rec_i0, m->pm.i.data() + m->pm.rec_i0, sizeof(int32_t) * (size_t)(m->pm.rec_i1 - m->pm.rec_i0), hipMemcpyHostToDevice));
  return 0;
}

extern "C" int grx_model_lds_bytes(const grx_model* m) { return m ? m->words * 4 : -1; }

extern "C" int grx_model_dim(const grx_model* m, const char* name) {
  if (!m) return -1;
  const GrxModel& g = m->dev;
  if (!strcmp(name, "nq")) return g.nq;
  if (!strcmp(name, "nv")) return g.nv;
  if (!strcmp(name, "nu")) return g.nu;
  if (!strcmp(name, "nbody")) return g.nbody;
  if (!strcmp(name, "nmocap")) return g.nmocap;
  if (!strcmp(name, "ndevpair")) return g.ndevpair;
  if (!strcmp(name, "shape")) return m->shape;          // id of the shape-specialised kernel the launches use, 0 = the generic one
  if (!strcmp(name, "handtree")) return g.handtree;
  return -1;
}

static int check_buffers(const grx_fetch_buffers* b) {
  if (!b || !b->qpos || !b->qvel || !b->qacc_ws || !b->mocap || !b->aux || !b->goal || !b->obs || !b->achieved || !b->reward ||
      !b->success || !b->status)
    return fail("grx_fetch_*: null buffer");
  return 0;
}

extern "C" int grx_fetch_step(const grx_model* m, const grx_fetch_task* task, const grx_fetch_buffers* buf, int n_worlds, void* stream) {
  if (!m || !task) return fail("grx_fetch_step: null argument");
  if (check_buffers(buf)) return -1;
  if (!buf->action) return fail("grx_fetch_step: null action buffer");
  if (n_worlds <= 0) return 0;
  GrxFetchTask t; memcpy(&t, task, sizeof(t));
  GrxFetchBuffers b; memcpy(&b, buf, sizeof(b));
  if (b.lane.list && m->shape != 0 && m->shape != 1 && m->shape < 100) return fail("grx_fetch_step: a lane launch needs a model with a lane kernel (the generic kernel, FetchPickAndPlace's own shape, or its large-table shape)");
  if (m->shape == 8 && !(b.handoff && b.lane.entry_count && b.lane.entry_list && b.handoff_stride >= grx_handoff_words(m->dev.nq, m->dev.nv, m->dev.nu, m->dev.nmocap)))
    return fail("grx_fetch_step: the fast FetchPickAndPlace kernel carries no hull routine: it needs hand-off rows (handoff, handoff_stride) and an entry list to hand hull worlds to");
  if (b.split_parts > 1) {
    if (b.lane.list || m->shape == 8) return fail("grx_fetch_step: split_parts applies to the step launch of a full kernel (not to a lane launch, not to the hull-less fast kernel)");
    if (!b.split_state || !b.handoff || b.handoff_stride < grx_handoff_words(m->dev.nq, m->dev.nv, m->dev.nu, m->dev.nmocap)) return fail("grx_fetch_step: a split step needs split_state [N, 2] and hand-off rows (handoff, handoff_stride) as carrier");
    if (b.split_parts > 8) return fail("grx_fetch_step: split_parts <= 8");
  } else b.split_parts = 0;
  const unsigned split_mul = b.split_parts > 1 ? (unsigned)b.split_parts : 1u;
  const int e = grx_tu_fetch_launch(0, m->shape, (b.lane.list ? (unsigned)(b.lane.grid > 0 ? b.lane.grid : GRX_LANE_GRID) : split_mul * grx_grid_for(n_worlds)), (size_t)m->words * 4 + m->lds_pad, stream, m->slot, &t, &b, nullptr, n_worlds, m->words, 0);
  if (e) return fail(std::string("grx_fetch_step launch: ") + hipGetErrorString((hipError_t)e));
  return 0;
}

extern "C" int grx_fetch_forward(const grx_model* m, const grx_fetch_task* task, const grx_fetch_buffers* buf, int n_worlds, int nstep, void* stream) {
  if (!m || !task) return fail("grx_fetch_forward: null argument");
  if (check_buffers(buf)) return -1;
  if (n_worlds <= 0) return 0;
  GrxFetchTask t; memcpy(&t, task, sizeof(t));
  GrxFetchBuffers b; memcpy(&b, buf, sizeof(b));
  const int e = grx_tu_fetch_launch(1, m->shape, grx_grid_for(n_worlds), (size_t)m->words * 4 + m->lds_pad, stream, m->slot, &t, &b, nullptr, n_worlds, m->words, nstep);
  if (e) return fail(std::string("grx_fetch_forward launch: ") + hipGetErrorString((hipError_t)e));
  return 0;
}

static_assert(sizeof(grx_fetch_reset_args) == sizeof(GrxFetchResetArgs), "grx_fetch_reset_args must mirror GrxFetchResetArgs");
extern "C" int grx_fetch_reset(const grx_model* m, const grx_fetch_task* task, const grx_fetch_buffers* buf, const grx_fetch_reset_args* args, int n_reset, void* stream) {
  if (!m || !task || !args) return fail("grx_fetch_reset: null argument");
  if (check_buffers(buf)) return -1;
  if (n_reset <= 0) return 0;
  if (!args->idx || !args->samples || !args->init_qpos || !args->init_qvel || (m->dev.nmocap && !args->init_mocap)) return fail("grx_fetch_reset: null reset array");
  if (args->obj_qadr >= 0 && args->obj_qadr + 7 > m->dev.nq) return fail("grx_fetch_reset: object joint address out of range");
  GrxFetchTask t; memcpy(&t, task, sizeof(t));
  GrxFetchBuffers b; memcpy(&b, buf, sizeof(b));
  GrxFetchResetArgs r; memcpy(&r, args, sizeof(r));
  const int e = grx_tu_fetch_launch(2, m->shape, (unsigned)n_reset, (size_t)m->words * 4 + m->lds_pad, stream, m->slot, &t, &b, &r, n_reset, m->words, 0);
  if (e) return fail(std::string("grx_fetch_reset launch: ") + hipGetErrorString((hipError_t)e));
  return 0;
}

extern "C" int grx_point_step(const grx_model* m, const grx_point_task* task, const grx_point_buffers* buf, int n_worlds, void* stream) {
  if (!m || !task || !buf) return fail("grx_point_step: null argument");
  if (!buf->qpos || !buf->qvel || !buf->qacc_ws || !buf->goal || !buf->action || !buf->obs || !buf->achieved || !buf->reward || !buf->success ||
      !buf->terminated || !buf->status)
    return fail("grx_point_step: null buffer");
  if (n_worlds <= 0) return 0;
  GrxPointTask t; memcpy(&t, task, sizeof(t));
  GrxPointBuffers b; memcpy(&b, buf, sizeof(b));
  if (b.split_parts > 1) {
    if (!b.split_state) return fail("grx_point_step: a split step needs split_state [N, 2]");
    if (b.split_parts > t.n_substeps || b.split_parts > 8) return fail("grx_point_step: split_parts <= min(frame_skip, 8): a part runs whole substeps");
  } else b.split_parts = 0;
  const int e = grx_tu_point_launch(m->shape, (b.split_parts > 1 ? (unsigned)b.split_parts : 1u) * grx_grid_for(n_worlds), (size_t)m->words * 4 + m->lds_pad, stream, m->slot, &t, &b, n_worlds, m->words);
  if (e) return fail(std::string("grx_point_step launch: ") + hipGetErrorString((hipError_t)e));
  return 0;
}

extern "C" int grx_hand_step(const grx_model* m, const grx_hand_task* task, const grx_hand_buffers* buf, int n_worlds, int forward_only, void* stream) {
  if (!m || !task || !buf) return fail("grx_hand_step: null argument");
  if (!buf->qpos || !buf->qvel || !buf->qacc_ws || !buf->goal || !buf->obs || !buf->achieved || !buf->palm || !buf->reward || !buf->success || !buf->status)
    return fail("grx_hand_step: null buffer");
  if (!forward_only && !buf->action) return fail("grx_hand_step: null action buffer");
  if (n_worlds <= 0) return 0;
  GrxHandTask t; memcpy(&t, task, sizeof(t));
  GrxHandBuffers b; memcpy(&b, buf, sizeof(b));
  if (t.kind) {
    if (t.nq_robot <= 0 || t.nq_robot > m->dev.nv || t.obj_qadr < 0 || t.obj_qadr + 7 > m->dev.nq || t.obj_dadr < 0 || t.obj_dadr + 6 > m->dev.nv)
      return fail("grx_hand_step: object joint addresses out of range");
    if (t.touch_mode && m->dev.ntouch == 0) return fail("grx_hand_step: touch_mode set but the model has no touch sensors");
  } else
    for (int k = 0; k < GRX_HAND_NTIPS; k++) if (t.site[k] < 0 || t.site[k] >= m->dev.nsite) return fail("grx_hand_step: fingertip site out of range");
  if (t.palm_body < 0 || t.palm_body >= m->dev.nbody) return fail("grx_hand_step: palm body out of range");
  if (b.lane.list && m->shape != 0 && m->shape < 100) return fail("grx_hand_step: the large-table launch of the overflow lane needs a model that runs on the generic kernel (capacities that match no specialised shape)");
  const bool split = b.split_parts > 1 && forward_only == 0 && !b.lane.list;      // plain step launches only (not forward-only, not a repeat launch, not the lane's)
  if (split) {
    if (!b.split_state || !b.split_rows || b.split_stride < m->dev.nq + 2 * m->dev.nv) return fail("grx_hand_step: a split step needs split_state [N, 4] and carrier rows split_rows [N, split_stride >= nq + 2 nv]");
    if (b.split_parts > t.n_substeps || b.split_parts > 8) return fail("grx_hand_step: split_parts <= min(n_substeps, 8): a part runs whole substeps");
  } else b.split_parts = 0;
  const int e = grx_tu_hand_launch(m->shape, (b.lane.list ? (unsigned)(b.lane.grid > 0 ? b.lane.grid : GRX_LANE_GRID) : (split ? (unsigned)b.split_parts : 1u) * grx_grid_for(n_worlds)), (size_t)m->words * 4 + m->lds_pad, stream, m->slot, &t, &b, n_worlds, m->words, forward_only);
  if (e) return fail(std::string("grx_hand_step launch: ") + hipGetErrorString((hipError_t)e));
  return 0;
}

extern "C" int grx_hand_step_repeat(const grx_model* m, const grx_hand_task* task, const grx_hand_buffers* buf, int n_worlds, int repeat, void* stream) {
  if (!buf) return fail("grx_hand_step_repeat: null argument");
  if (repeat < 1 || repeat > 1024) return fail("grx_hand_step_repeat: repeat must be in 1 .. 1024");
  if (buf->lane.list || buf->lane.skip || buf->lane.entry_list || buf->lane.next_list) return fail("grx_hand_step_repeat: not for launches with an overflow lane (a world that exceeds a table drops the contact and flags it, as in any launch without a lane)");
  if (!buf->action) return fail("grx_hand_step_repeat: null action buffer");
  return grx_hand_step(m, task, buf, n_worlds, repeat == 1 ? 0 : repeat, stream);
}

extern "C" int grx_adroit_step(const grx_model* m, const grx_adroit_task* task, const grx_adroit_buffers* buf, int n_worlds, int forward_only, void* stream) {
  if (!m || !task || !buf) return fail("grx_adroit_step: null argument");
  if (!buf->qpos || !buf->qvel || !buf->qacc_ws || !buf->obs || !buf->reward || !buf->success || !buf->status) return fail("grx_adroit_step: null buffer");
  if (!forward_only && (!buf->action || !buf->act_mean || !buf->act_rng)) return fail("grx_adroit_step: null action buffers");
  if (m->dev.nshift && !buf->shift) return fail("grx_adroit_step: the model has a shift group but no shift buffer was given");
  if (n_worlds <= 0) return 0;
  GrxAdroitTask t; memcpy(&t, task, sizeof(t));
  GrxAdroitBuffers b; memcpy(&b, buf, sizeof(b));
  const GrxModel& g = m->dev;
  static const int nsites[4] = {4, 2, 5, 1}, tail[4] = {19, 12, 21, 9};
  if (t.kind < 0 || t.kind > 3) return fail("grx_adroit_step: unknown task kind");
  for (int k = 0; k < nsites[t.kind]; k++) if (t.site[k] < 0 || t.site[k] >= g.nsite) return fail("grx_adroit_step: site id out of range");
  if (t.kind != GRX_ADROIT_DOOR && (t.obj_body <= 0 || t.obj_body >= g.nbody || t.nq_obs != g.nq - 6 || g.nv < 6)) return fail("grx_adroit_step: object body / nq_obs do not fit the model");
  if (t.kind == GRX_ADROIT_DOOR && (t.nq_obs != g.nq - 3 || t.qadr[0] < 0 || t.qadr[0] >= g.nq || t.qadr[1] < 0 || t.qadr[1] >= g.nq)) return fail("grx_adroit_step: door qpos indices do not fit the model");
  if (t.obs_dim != t.nq_obs + tail[t.kind]) return fail("grx_adroit_step: obs_dim does not fit the task kind");
  if (t.kind == GRX_ADROIT_HAMMER && g.ntouch != 1) return fail("grx_adroit_step: the hammer task reads one touch sensor");
  if (t.kind == GRX_ADROIT_PEN && !(t.len[0] > 0.0f && t.len[1] > 0.0f)) return fail("grx_adroit_step: pen / target lengths must be positive");
  if (t.kind == GRX_ADROIT_RELOCATE && !buf->target) return fail("grx_adroit_step: the relocate task needs the target buffer");
  if (b.lane.list && m->shape != 0 && m->shape < 100) return fail("grx_adroit_step: the large-table launch of the overflow lane needs a model that runs on the generic kernel (capacities that match no specialised shape)");
  if (b.compact && (b.n_compact <= 0 || b.lane.list)) return fail("grx_adroit_step: a compacted launch needs n_compact > 0 and is not a lane launch");
  const bool split = b.split_parts > 1 && !b.compact && !forward_only && !b.lane.list;      // step launches only
  if (split) {
    if (!b.split_state || !b.split_rows || b.split_stride < m->dev.nq + 2 * m->dev.nv) return fail("grx_adroit_step: a split step needs split_state [N, 4] and carrier rows split_rows [N, split_stride >= nq + 2 nv]");
    if (b.split_parts > t.n_substeps || b.split_parts > 8) return fail("grx_adroit_step: split_parts <= min(frame_skip, 8): a part runs whole substeps");
  } else b.split_parts = 0;
  const int e = grx_tu_adroit_launch(m->shape, (b.lane.list ? (unsigned)(b.lane.grid > 0 ? b.lane.grid : GRX_LANE_GRID) : (b.compact ? (unsigned)b.n_compact : (split ? (unsigned)b.split_parts : 1u) * grx_grid_for(n_worlds))), (size_t)m->words * 4 + m->lds_pad, stream, m->slot, &t, &b, n_worlds, m->words, forward_only);
  if (e) return fail(std::string("grx_adroit_step launch: ") + hipGetErrorString((hipError_t)e));
  return 0;
}

// host tables of a model by name (the packed copy kept with the handle)
static const int32_t* grx_host_itable(const grx_model* m, const char* name, int* n) {
  for (size_t k = 0; k < m->pm.name.size(); k++)
    if (m->pm.kind[k] == 'i' && m->pm.name[k] == name) { *n = m->pm.cnt[k]; return m->pm.i.data() + m->pm.off[k]; }
  *n = 0; return nullptr;
}
// the skin list of grx_collision assumes that a plane geom never moves: every plane must sit on a body without a dof or a mocap id on its path to the world
static bool grx_planes_static(const grx_model* m) {
  int ng, nb, n;
  const int32_t *gt = grx_host_itable(m, "geom_type", &ng), *gb = grx_host_itable(m, "geom_bodyid", &n), *bp = grx_host_itable(m, "body_parent", &nb),
                *dn = grx_host_itable(m, "body_dofnum", &n), *mc = grx_host_itable(m, "body_mocapid", &n);
  if (!gt || !gb || !bp || !dn || !mc) return false;
  for (int g = 0; g < ng; g++) {
    if (gt[g] != 0) continue;
    for (int b = gb[g]; b > 0; b = bp[b]) if (dn[b] > 0 || mc[b] >= 0) return false;
  }
  return true;
}

extern "C" int grx_kitchen_step(const grx_model* m, const grx_kitchen_task* task, const grx_kitchen_buffers* buf, int n_worlds, int forward_only, void* stream) {
  if (!m || !task || !buf) return fail("grx_kitchen_step: null argument");
  if (!buf->qpos || !buf->qvel || !buf->qacc_ws || !buf->last_qpos || !buf->obs || !buf->completed || !buf->status) return fail("grx_kitchen_step: null buffer");
  if (!forward_only && !buf->action) return fail("grx_kitchen_step: null action buffer");
  if (n_worlds <= 0) return 0;
  GrxKitchenTask t; memcpy(&t, task, sizeof(t));
  GrxKitchenBuffers b; memcpy(&b, buf, sizeof(b));
  const GrxModel& g = m->dev;
  if (g.nu != GRX_KITCHEN_NROBOT || t.obs_dim != g.nq + g.nv || t.obs_dim > GRX_KITCHEN_OBS || t.n_substeps <= 0) return fail("grx_kitchen_step: the model is not the kitchen scene (nu 9, obs = nq + nv <= 59)");
  for (int j = 0; j < GRX_KITCHEN_NTASK; j++)
    if (t.task_adr[j] < 0 || t.task_num[j] < 0 || t.task_adr[j] + t.task_num[j] > g.nq) return fail("grx_kitchen_step: task qpos slice out of range");
  if (b.skin) {
    if (b.skin_stride < 4 + 3 * g.ngeom + g.ndevpair || !(b.skin_radius > 0.0f)) return fail("grx_kitchen_step: skin rows need 4 + 3 ngeom + ndevpair words and a positive radius");
    if (!grx_planes_static(m)) return fail("grx_kitchen_step: the skin list needs static plane geoms");
  }
  if (b.lane.list && m->shape != 0 && m->shape < 100) return fail("grx_kitchen_step: the large-table launch of the overflow lane needs a model that runs on the generic kernel (capacities that match no specialised shape)");
  const bool split = b.split_parts > 1 && !forward_only && !b.lane.list;      // step launches only
  if (split) {
    if (!b.split_state || !b.split_rows || b.split_stride < g.nq + 2 * g.nv) return fail("grx_kitchen_step: a split step needs split_state [N, 4] and carrier rows split_rows [N, split_stride >= nq + 2 nv]");
    if (b.split_parts > t.n_substeps || b.split_parts > 8) return fail("grx_kitchen_step: split_parts <= min(n_substeps, 8): a part runs whole substeps");
  } else b.split_parts = 0;
  const int e = grx_tu_kitchen_launch(m->shape, (b.lane.list ? (unsigned)(b.lane.grid > 0 ? b.lane.grid : GRX_LANE_GRID) : (split ? (unsigned)b.split_parts : 1u) * grx_grid_for(n_worlds)), (size_t)m->words * 4 + m->lds_pad, stream, m->slot, &t, &b, n_worlds, m->words, forward_only);
  if (e) return fail(std::string("grx_kitchen_step launch: ") + hipGetErrorString((hipError_t)e));
  return 0;
}

// Cost-ordered dispatch, device side: one workgroup per XCD slice sorts (cost, world) keys of its contiguous `per` worlds in LDS
// (bitonic, descending cost, ties by world index) and writes order[i * 8 + slice] = the i-th most expensive world of the slice --
// workgroup b of the next step launch runs on XCD b & 7 and starts in index order.
extern "C" __global__ void __launch_bounds__(256)
grx_order_kernel(const int* __restrict__ cost, float* __restrict__ ema, float alpha, int per, int npow2, int slots, int* __restrict__ order) {
  extern __shared__ unsigned long long keys[];
  const int s = blockIdx.x, base = s * per;
  for (int i = threadIdx.x; i < npow2; i += 256) {
    unsigned k = 0;
    if (i < per) {
      // the sort key: the last cost, or its exponential moving average (a world's cost has a persistent part -- is the object in contact -- and
      // a per-step part; averaging predicts the next step better than the last sample alone)
      float c = (float)cost[base + i];
      if (ema) { c = (1.0f - alpha) * ema[base + i] + alpha * c; ema[base + i] = c; }
      k = (unsigned)fminf(fmaxf(c * 16.0f, 0.0f), 4.0e9f);
    }
    keys[i] = i < per ? (((unsigned long long)k << 32) | (unsigned)(0x7FFFFFFF - i)) : 0ull;   // padding sorts last
  }
  __syncthreads();
  for (int k = 2; k <= npow2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npow2; i += 256) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long a = keys[i], b = keys[l];
          const bool desc = (i & k) == 0;
          if (desc ? (a < b) : (a > b)) { keys[i] = b; keys[l] = a; }
        }
      }
      __syncthreads();
    }
  // Two worlds per wave slot (per <= 2 * slots: BASELINE cfg 2, 4096 worlds on 2048 slots).  K predicted stragglers -- worlds that take longer than the two
  // cheapest worlds one after the other (the arm resting on the head: two hull pairs in contact) -- hold their slots for the whole launch, so M = per - 2 slots + K
  // workgroups have to be a slot's THIRD world.  With the plain descending order those are dispatched when the first slots finish their second world, two MEDIAN
  // worlds after the start, and end the launch alone (3.02 ms against 2.68 ms for the slowest world, profiles/cost_probe_r03.txt).  Here the 3 M cheapest worlds
  // are placed so that M slots run three CHEAP worlds back to back: the M cheapest at the end of the first round (they free their slots first), the next M at the
  // start of the second round (dispatched onto exactly those slots), the next M at the very end (dispatched when that second cheap world ends).
  __shared__ int sM;
  if (threadIdx.x == 0) {
    int M = 0;
    if (slots > 0 && per > slots && per <= 2 * slots) {
      const unsigned long long thr = 2ull * (keys[per - 1] >> 32);
      int K = 0, hi = per;      // the list is sorted (descending): K = the first position whose key is <= thr, by bisection (a linear walk costs 25 us when half the slice is above it)
      while (K < hi) { const int mid = (K + hi) >> 1; if ((keys[mid] >> 32) > thr) K = mid + 1; else hi = mid; }
      M = per - 2 * slots + K;
      if (M < 0 || 3 * M > per - slots || M > slots / 4) M = 0;
    }
    sM = M;
  }
  __syncthreads();
  const int M = sM;
  for (int pos = threadIdx.x; pos < per; pos += 256) {
    int i = pos;   // index into the descending list
    if (M > 0) {
      if (pos < slots - M) i = pos;
      else if (pos < slots) i = per - M + (pos - (slots - M));
      else if (pos < slots + M) i = per - 2 * M + (pos - slots);
      else if (pos < per - M) i = pos - 2 * M;
      else i = per - 3 * M + (pos - (per - M));
    }
    order[pos * 8 + s] = base + (0x7FFFFFFF - (int)(unsigned)(keys[i] & 0xFFFFFFFFull));
  }
}

extern "C" int grx_order_by_cost_slots(const int* cost, float* ema, float alpha, int n_worlds, int slots_per_xcd, int* order, void* stream);
extern "C" int grx_order_by_cost(const int* cost, float* ema, float alpha, int n_worlds, int* order, void* stream) { return grx_order_by_cost_slots(cost, ema, alpha, n_worlds, 0, order, stream); }
extern "C" int grx_order_by_cost_slots(const int* cost, float* ema, float alpha, int n_worlds, int slots_per_xcd, int* order, void* stream) {
  if (!cost || !order) return fail("grx_order_by_cost: null argument");
  if (ema && !(alpha > 0.0f && alpha <= 1.0f)) return fail("grx_order_by_cost: alpha must be in (0, 1]");
  if (n_worlds <= 0 || (n_worlds & 7)) return fail("grx_order_by_cost: the number of worlds must be a positive multiple of 8 (one contiguous slice per XCD)");
  const int per = n_worlds >> 3;
  int npow2 = 1;
  while (npow2 < per) npow2 <<= 1;
  if ((size_t)npow2 * 8 > 64 * 1024) return fail("grx_order_by_cost: more than 65536 worlds per launch are not supported");
  hipLaunchKernelGGL(grx_order_kernel, dim3(8), dim3(256), (size_t)npow2 * 8, (hipStream_t)stream, cost, ema, alpha, per, npow2, slots_per_xcd < 0 ? 0 : slots_per_xcd, order);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int grx_goal_compute_reward(const float* achieved, const float* desired, int64_t batch, int dim, double distance_threshold, int sparse,
                                       float* reward_out, void* stream) {
  if (!achieved || !desired || !reward_out || dim <= 0) return fail("grx_goal_compute_reward: bad argument");
  if (batch <= 0) return 0;
  long long blocks = (batch + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(grx_goal_reward_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, achieved, desired, (long long)batch, dim,
                     distance_threshold, sparse, reward_out);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int grx_manip_compute_reward(const float* achieved, const float* desired, int64_t batch, int ignore_position, int ignore_rotation,
                                        int ignore_z, float distance_threshold, float rotation_threshold, int sparse, float* reward_out, void* stream) {
  if (!achieved || !desired || !reward_out) return fail("grx_manip_compute_reward: null argument");
  if (batch <= 0) return 0;
  long long blocks = (batch + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(grx_manip_reward_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, achieved, desired, (long long)batch, ignore_position,
                     ignore_rotation, ignore_z, distance_threshold, rotation_threshold, sparse, reward_out);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int grx_maze_compute_reward(const float* achieved, const float* desired, int64_t batch, double goal_radius, int sparse, float* reward_out,
                                       void* stream) {
  if (!achieved || !desired || !reward_out) return fail("grx_maze_compute_reward: null argument");
  if (batch <= 0) return 0;
  long long blocks = (batch + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(grx_maze_reward_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, achieved, desired, (long long)batch, goal_radius,
                     sparse, reward_out);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int grx_her_relabel(const grx_her_args* args, int64_t batch, void* stream) {
  if (!args) return fail("grx_her_relabel: null argument");
  GrxHerArgs a; memcpy(&a, args, sizeof(a));
  if (!a.rows || !a.acts || !a.t_idx || !a.w_idx || !a.t_goal || !a.out) return fail("grx_her_relabel: null buffer");
  if ((a.term_rows == nullptr) != (a.term_t == nullptr)) return fail("grx_her_relabel: term_rows and term_t go together");
  if (a.T <= 0 || a.N <= 0 || a.obs_dim <= 0 || a.goal_dim <= 0 || a.goal_dim > 16 || a.act_dim <= 0 || a.W < a.obs_dim + 2 * a.goal_dim)
    return fail("grx_her_relabel: dimensions out of range (goal_dim <= 16, W >= obs_dim + 2 goal_dim)");
  if (a.kind < 0 || a.kind > 3 || (a.kind == 0 && a.goal_dim != 3) || (a.kind == 2 && a.goal_dim != 2) || (a.kind == 3 && a.goal_dim != 7))
    return fail("grx_her_relabel: reward kind does not fit goal_dim");
  if (batch <= 0) return 0;
  const long long words = (long long)batch * (2 * a.obs_dim + 3 * a.goal_dim + a.act_dim + 2);
  long long blocks = (words + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(grx_her_relabel_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, (long long)batch);
  HIP_OK(hipGetLastError());
  return 0;
}

// HER index draws (include/grx_capi.h): one thread per sample, splitmix64 stream keyed by (seed, call, sample)
static __device__ __forceinline__ unsigned long long grx_splitmix(unsigned long long& s) {
  unsigned long long z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
extern "C" __global__ void __launch_bounds__(256)
grx_her_sample_kernel(const int* __restrict__ start, const int* __restrict__ prev_start, const int* __restrict__ term_t, int N, int t_now, int T, int k_future,
                      unsigned long long seed, unsigned long long call, long long B, int* __restrict__ t_idx, int* __restrict__ w_idx, int* __restrict__ t_goal) {
  const int lo_min = t_now - T > 0 ? t_now - T : 0;
  // first row of the episode world w is sampled from: its current one, or -- when that one began in this very step -- the one that has just ended
#define GRX_HER_LO(W_) ((term_t && term_t[W_] == t_now) ? (prev_start[W_] > lo_min ? prev_start[W_] : lo_min) : (start[W_] > lo_min ? start[W_] : lo_min))
  for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (long long)gridDim.x * blockDim.x) {
    unsigned long long s = seed * 0xD1342543DE82EF95ull + call * 0x2545F4914F6CDD1Dull + (unsigned long long)b;
    (void)grx_splitmix(s);
    int w = 0, lo = t_now;
    for (int attempt = 0; attempt < 64 && lo >= t_now; attempt++) {          // uniform over the worlds that have a transition (the caller made sure one exists)
      w = (int)(((grx_splitmix(s) >> 32) * (unsigned long long)N) >> 32);
      lo = GRX_HER_LO(w);
    }
    for (int probe = 0; probe < N && lo >= t_now; probe++) { w = w + 1 < N ? w + 1 : 0; lo = GRX_HER_LO(w); }
    const unsigned long long r = grx_splitmix(s), r2 = grx_splitmix(s);
    const float u0 = (float)(r >> 40) * (1.0f / 16777216.0f), u1 = (float)((r >> 16) & 0xFFFFFF) * (1.0f / 16777216.0f), u2 = (float)(r2 >> 40) * (1.0f / 16777216.0f);
    int t = lo + (int)(u0 * (float)(t_now - lo));
    if (t > t_now - 1) t = t_now - 1;
    int fut = t + 1 + (int)(u1 * (float)(t_now - t));
    if (fut > t_now) fut = t_now;
    t_idx[b] = t; w_idx[b] = w;
    t_goal[b] = (u2 >= (float)k_future / ((float)k_future + 1.0f)) ? -1 : fut;
  }
#undef GRX_HER_LO
}
extern "C" int grx_her_sample_final(const int* episode_start, const int* prev_start, const int* term_t, int n_worlds, int t_now, int T, int k_future, uint64_t seed,
                                    uint64_t call, int64_t batch, int* t_idx, int* w_idx, int* t_goal, void* stream) {
  if (!episode_start || !t_idx || !w_idx || !t_goal) return fail("grx_her_sample: null argument");
  if ((prev_start == nullptr) != (term_t == nullptr)) return fail("grx_her_sample_final: prev_start and term_t go together");
  if (n_worlds <= 0 || T <= 0 || t_now <= 0 || k_future < 0) return fail("grx_her_sample: n_worlds, T and t_now must be positive, k_future >= 0");
  if (batch <= 0) return 0;
  long long blocks = (batch + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(grx_her_sample_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, episode_start, prev_start, term_t, n_worlds, t_now, T, k_future,
                     (unsigned long long)seed, (unsigned long long)call, (long long)batch, t_idx, w_idx, t_goal);
  HIP_OK(hipGetLastError());
  return 0;
}
extern "C" int grx_her_sample(const int* episode_start, int n_worlds, int t_now, int T, int k_future, uint64_t seed, uint64_t call, int64_t batch,
                              int* t_idx, int* w_idx, int* t_goal, void* stream) {
  return grx_her_sample_final(episode_start, nullptr, nullptr, n_worlds, t_now, T, k_future, seed, call, batch, t_idx, w_idx, t_goal, stream);
}
extern "C" __global__ void __launch_bounds__(256)
grx_her_mark_kernel(const unsigned char* __restrict__ mask, int N, int t, int* __restrict__ start, int* __restrict__ prev_start, int* __restrict__ term_t) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= N || !mask[w]) return;
  if (prev_start) { prev_start[w] = start[w]; term_t[w] = t; }
  start[w] = t;
}
extern "C" int grx_her_mark_resets(const unsigned char* reset_mask, int n_worlds, int t, int* episode_start, int* prev_start, int* term_t, void* stream) {
  if (!reset_mask || !episode_start || n_worlds <= 0) return fail("grx_her_mark_resets: null argument");
  if ((prev_start == nullptr) != (term_t == nullptr)) return fail("grx_her_mark_resets: prev_start and term_t go together");
  hipLaunchKernelGGL(grx_her_mark_kernel, dim3((unsigned)((n_worlds + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reset_mask, n_worlds, t, episode_start, prev_start, term_t);
  HIP_OK(hipGetLastError());
  return 0;
}

// maze episode reset for a compacted list of worlds (include/grx_capi.h): one 64-thread workgroup per listed world
extern "C" __global__ void __launch_bounds__(64)
grx_maze_reset_kernel(grx_maze_reset_args a, int n_reset) {
  const int k = blockIdx.x, l = threadIdx.x;
  if (k >= n_reset) return;
  const int w = a.idx[k];
  const float sx = a.stage[4 * k], sy = a.stage[4 * k + 1], gx = a.stage[4 * k + 2], gy = a.stage[4 * k + 3];
  const int nobs_q = a.nq - a.obs_skip;
  float* row = a.packed ? a.packed + (size_t)w * (a.obs_dim + 6) : nullptr;
  for (int i = l; i < a.nq; i += 64) a.qpos[(size_t)w * a.nq + i] = i == 0 ? sx : (i == 1 ? sy : a.qpos0[i]);
  for (int i = l; i < a.nv; i += 64) { a.qvel[(size_t)w * a.nv + i] = 0.0f; a.qacc_ws[(size_t)w * a.nv + i] = 0.0f; }
  for (int i = l; i < a.obs_dim; i += 64) {
    const int q = i + a.obs_skip;
    const float v = i < nobs_q ? (q == 0 ? sx : (q == 1 ? sy : a.qpos0[q])) : 0.0f;
    a.obs[(size_t)w * a.obs_dim + i] = v;
    if (row) row[i] = v;
  }
  if (l == 0) {
    a.goal[2 * (size_t)w] = gx; a.goal[2 * (size_t)w + 1] = gy;
    a.achieved[2 * (size_t)w] = sx; a.achieved[2 * (size_t)w + 1] = sy;
    const float sg[2] = {sx, sy}, gg[2] = {gx, gy};
    const int succ = grx_goal_distance2(sg, gg) <= a.goal_radius;
    a.success[w] = (unsigned char)succ;
    if (row) {
      row[a.obs_dim] = sx; row[a.obs_dim + 1] = sy; row[a.obs_dim + 2] = gx; row[a.obs_dim + 3] = gy;
      if (!a.keep_outcome) { row[a.obs_dim + 4] = 0.0f; row[a.obs_dim + 5] = succ ? 1.0f : 0.0f; }
    }
    if (!a.keep_outcome) a.reward[w] = 0.0f;     // a reset step reports reward 0 (next-step autoreset): reward[] and the packed row agree
  }
}
extern "C" int grx_maze_reset_rows(const grx_maze_reset_args* args, int n_reset, void* stream) {
  if (!args) return fail("grx_maze_reset_rows: null argument");
  const grx_maze_reset_args& a = *args;
  if (!a.idx || !a.stage || !a.qpos0 || !a.qpos || !a.qvel || !a.qacc_ws || !a.goal || !a.obs || !a.achieved || !a.reward || !a.success) return fail("grx_maze_reset_rows: null buffer");
  if (a.nq < 2 || a.nv <= 0 || a.obs_skip < 0 || a.obs_skip > 2 || a.obs_dim != a.nq + a.nv - a.obs_skip) return fail("grx_maze_reset_rows: obs_dim must be nq + nv - obs_skip");
  if (n_reset <= 0) return 0;
  hipLaunchKernelGGL(grx_maze_reset_kernel, dim3((unsigned)n_reset), dim3(64), 0, (hipStream_t)stream, a, n_reset);
  HIP_OK(hipGetLastError());
  return 0;
}

// commit of an overlapped hand-manipulate reset (include/grx_capi.h): one 64-thread workgroup per row
extern "C" __global__ void __launch_bounds__(64)
grx_hand_commit_kernel(grx_hand_commit_args a) {
  const int j = blockIdx.x, l = threadIdx.x;
  if (j >= a.k) return;
  const size_t w = (size_t)a.idx[j], s = (size_t)j;
  const int od = a.obs_dim, gd = a.goal_dim, pw = od + 2 * gd + 2;
  for (int i = l; i < a.nq; i += 64) a.qpos[w * a.nq + i] = a.s_qpos[s * a.nq + i];
  for (int i = l; i < a.nv; i += 64) { a.qvel[w * a.nv + i] = a.s_qvel[s * a.nv + i]; a.qacc_ws[w * a.nv + i] = a.s_qacc_ws[s * a.nv + i]; }
  for (int i = l; i < od; i += 64) { const float v = a.s_obs[s * od + i]; a.obs[w * od + i] = v; }
  for (int i = l; i < gd; i += 64) { a.achieved[w * gd + i] = a.s_achieved[s * gd + i]; a.goal[w * gd + i] = a.s_goal[s * gd + i]; }
  if (l < 3) a.palm[w * 3 + l] = a.s_palm[s * 3 + l];
  for (int i = l; i < od + gd; i += 64) a.packed[w * pw + i] = a.s_packed[s * pw + i];               // [obs | achieved] of the reset state
  for (int i = l; i < gd; i += 64) a.packed[w * pw + od + gd + i] = a.s_goal[s * gd + i];           // the new goal
  if (l == 0) a.status[w] |= a.s_status[s] & (int)0xFFFF0000;
}
extern "C" int grx_hand_commit_rows(const grx_hand_commit_args* args, void* stream) {
  if (!args) return fail("grx_hand_commit_rows: null argument");
  const grx_hand_commit_args& a = *args;
  if (!a.idx || !a.s_qpos || !a.s_qvel || !a.s_qacc_ws || !a.s_obs || !a.s_achieved || !a.s_palm || !a.s_goal || !a.s_packed || !a.s_status || !a.qpos || !a.qvel ||
      !a.qacc_ws || !a.obs || !a.achieved || !a.palm || !a.goal || !a.packed || !a.status) return fail("grx_hand_commit_rows: null buffer");
  if (a.nq <= 0 || a.nv <= 0 || a.obs_dim <= 0 || a.goal_dim <= 0) return fail("grx_hand_commit_rows: bad dimensions");
  if (a.k <= 0) return 0;
  hipLaunchKernelGGL(grx_hand_commit_kernel, dim3((unsigned)a.k), dim3(64), 0, (hipStream_t)stream, a);
  HIP_OK(hipGetLastError());
  return 0;
}

// commit of an overlapped Fetch reset (include/grx_capi.h): one 64-thread workgroup per listed world; staged rows are indexed by WORLD (the reset kernel wrote them through a
// second grx_fetch_buffers)
extern "C" __global__ void __launch_bounds__(64)
grx_fetch_commit_kernel(grx_fetch_commit_args a) {
  const int j = blockIdx.x, l = threadIdx.x;
  if (j >= a.k) return;
  const size_t w = (size_t)a.idx[j];
  const int od = a.obs_dim, pw = od + 8;
  for (int i = l; i < a.nq; i += 64) a.qpos[w * a.nq + i] = a.s_qpos[w * a.nq + i];
  for (int i = l; i < a.nv; i += 64) { a.qvel[w * a.nv + i] = a.s_qvel[w * a.nv + i]; a.qacc_ws[w * a.nv + i] = a.s_qacc_ws[w * a.nv + i]; }
  for (int i = l; i < a.mocap_words; i += 64) a.mocap[w * a.mocap_words + i] = a.s_mocap[w * a.mocap_words + i];
  if (l < 8) a.aux[w * 8 + l] = a.s_aux[w * 8 + l];
  if (l < 3) { a.goal[w * 3 + l] = a.s_goal[w * 3 + l]; a.achieved[w * 3 + l] = a.s_achieved[w * 3 + l]; }
  for (int i = l; i < od; i += 64) a.obs[w * od + i] = a.s_obs[w * od + i];
  for (int i = l; i < pw; i += 64) {      // the thread that parks a word of the terminal row is the one that overwrites it
    const float old = a.packed[w * pw + i];
    if (a.final_packed) a.final_packed[w * pw + i] = old;
    a.packed[w * pw + i] = i < od ? a.s_obs[w * od + i] : (i < od + 3 ? a.s_achieved[w * 3 + i - od] : (i < od + 6 ? a.s_goal[w * 3 + i - od - 3] : old));
  }
  if (l == 0) a.status[w] = grx_status_word(a.status[w], a.s_status[w]);
}
extern "C" int grx_fetch_commit_rows(const grx_fetch_commit_args* args, void* stream) {
  if (!args) return fail("grx_fetch_commit_rows: null argument");
  const grx_fetch_commit_args& a = *args;
  if (!a.idx || !a.s_qpos || !a.s_qvel || !a.s_qacc_ws || !a.s_aux || !a.s_goal || !a.s_obs || !a.s_achieved || !a.s_status || !a.qpos || !a.qvel || !a.qacc_ws || !a.aux ||
      !a.goal || !a.obs || !a.achieved || !a.packed || !a.status || (a.mocap_words > 0 && (!a.s_mocap || !a.mocap))) return fail("grx_fetch_commit_rows: null buffer");
  if (a.nq <= 0 || a.nv <= 0 || a.obs_dim <= 0 || a.mocap_words < 0) return fail("grx_fetch_commit_rows: bad dimensions");
  if (a.k <= 0) return 0;
  hipLaunchKernelGGL(grx_fetch_commit_kernel, dim3((unsigned)a.k), dim3(64), 0, (hipStream_t)stream, a);
  HIP_OK(hipGetLastError());
  return 0;
}

// commit of an overlapped Adroit reset (include/grx_capi.h): one 64-thread workgroup per listed world, staged rows indexed by WORLD
extern "C" __global__ void __launch_bounds__(64)
grx_adroit_commit_kernel(grx_adroit_commit_args a) {
  const int j = blockIdx.x, l = threadIdx.x;
  if (j >= a.k) return;
  const size_t w = (size_t)a.idx[j];
  for (int i = l; i < a.nq; i += 64) a.qpos[w * a.nq + i] = a.s_qpos[w * a.nq + i];
  for (int i = l; i < a.nv; i += 64) { a.qvel[w * a.nv + i] = a.s_qvel[w * a.nv + i]; a.qacc_ws[w * a.nv + i] = a.s_qacc_ws[w * a.nv + i]; }
  for (int i = l; i < a.obs_dim; i += 64) a.obs[w * a.obs_dim + i] = a.s_obs[w * a.obs_dim + i];
  if (a.shift && l < 7) a.shift[w * 7 + l] = a.s_shift[w * 7 + l];
  if (a.target && l < 3) a.target[w * 3 + l] = a.s_target[w * 3 + l];
  if (l == 0) a.status[w] = grx_status_word(a.status[w], a.s_status[w]);   // as the in-line reset (and grx_fetch_commit_kernel): the low half reports the reset's forward pass, its flags join the sticky half
}
extern "C" int grx_adroit_commit_rows(const grx_adroit_commit_args* args, void* stream) {
  if (!args) return fail("grx_adroit_commit_rows: null argument");
  const grx_adroit_commit_args& a = *args;
  if (!a.idx || !a.s_qpos || !a.s_qvel || !a.s_qacc_ws || !a.s_obs || !a.s_status || !a.qpos || !a.qvel || !a.qacc_ws || !a.obs || !a.status) return fail("grx_adroit_commit_rows: null buffer");
  if ((a.shift != nullptr) != (a.s_shift != nullptr) || (a.target != nullptr) != (a.s_target != nullptr)) return fail("grx_adroit_commit_rows: shift / target need both the staged and the live buffer");
  if (a.nq <= 0 || a.nv <= 0 || a.obs_dim <= 0) return fail("grx_adroit_commit_rows: bad dimensions");
  if (a.k <= 0) return 0;
  hipLaunchKernelGGL(grx_adroit_commit_kernel, dim3((unsigned)a.k), dim3(64), 0, (hipStream_t)stream, a);
  HIP_OK(hipGetLastError());
  return 0;
}

// test-only entry point (tests/test_gpu_primitives.py); all pointers are device pointers
extern "C" int grx_debug_primitive(int mode, int nv, int nefc, const float* A, const float* b, const float* J, const float* D, float* out, void* stream) {
  int bytes = (2 * nv * nv + GRX_MAXEFC * nv + 6 * GRX_MAXEFC + 2 * nv + 64) * 4;
  hipLaunchKernelGGL(grx_debug_kernel, dim3(1), dim3(64), bytes, (hipStream_t)stream, mode, nv, nefc, A, b, J, D, out);
  HIP_OK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// Host-side episode-boundary sampling.  The reference draws object/goal positions from numpy's PCG64
// (Generator.uniform) in a data-dependent rejection loop (envs/fetch/fetch_env.py:153-166,388-391).  The per-world
// generator states are created by numpy (SeedSequence seeding stays in Python) and advanced here, bit-exactly:
//   PCG64 = 128-bit LCG (multiplier 0x2360ED051FC65DA44385DF649FCCF645) with XSL-RR output,
//   uniform(lo, hi) = lo + (hi - lo) * ((next64 >> 11) * 2^-53).
// ------------------------------------------------------------------------------------------
typedef unsigned __int128 grx_u128;
static inline uint64_t grx_pcg64_next(uint64_t* st /* state_hi, state_lo, inc_hi, inc_lo */) {
  const grx_u128 mult = ((grx_u128)0x2360ED051FC65DA4ULL << 64) | 0x4385DF649FCCF645ULL;
  grx_u128 state = ((grx_u128)st[0] << 64) | st[1], inc = ((grx_u128)st[2] << 64) | st[3];
  state = state * mult + inc;
  st[0] = (uint64_t)(state >> 64); st[1] = (uint64_t)state;
  uint64_t hi = st[0], lo = st[1], x = hi ^ lo; unsigned rot = (unsigned)(hi >> 58);
  return (x >> rot) | (x << ((64 - rot) & 63));
}
static inline double grx_pcg64_uniform(uint64_t* st, double lo, double hi) {
  return lo + (hi - lo) * ((double)(grx_pcg64_next(st) >> 11) * (1.0 / 9007199254740992.0));
}

// states: [n_total, 4] uint64 (state_hi, state_lo, inc_hi, inc_lo); idx: worlds to sample (n of them)
// out_oxy [n,2] (only if has_object), out_goal [n,3]; target_offset[3]
extern "C" int grx_fetch_sample_resets(uint64_t* states, const int64_t* idx, int n, int has_object, int target_in_the_air, double obj_range,
                                       double target_range, const double* target_offset, const double* gripper_xpos, double height_offset,
                                       double* out_oxy, double* out_goal) {
  if (!states || !idx || !out_goal || !gripper_xpos || !target_offset) return fail("grx_fetch_sample_resets: null argument");
  for (int k = 0; k < n; k++) {
    uint64_t* st = states + 4 * idx[k];
    if (has_object) {
      double ox = gripper_xpos[0], oy = gripper_xpos[1];
      for (;;) {
        double dx = ox - gripper_xpos[0], dy = oy - gripper_xpos[1];
        if (!(sqrt(dx * dx + dy * dy) < 0.1)) break;
        ox = gripper_xpos[0] + grx_pcg64_uniform(st, -obj_range, obj_range);
        oy = gripper_xpos[1] + grx_pcg64_uniform(st, -obj_range, obj_range);
      }
      out_oxy[2 * k] = ox; out_oxy[2 * k + 1] = oy;
    }
    double g[3];
    for (int e = 0; e < 3; e++) g[e] = gripper_xpos[e] + grx_pcg64_uniform(st, -target_range, target_range);
    if (has_object) {
      for (int e = 0; e < 3; e++) g[e] += target_offset[e];
      g[2] = height_offset;
      if (target_in_the_air && grx_pcg64_uniform(st, 0.0, 1.0) < 0.5) g[2] += grx_pcg64_uniform(st, 0.0, 0.45);
    }
    for (int e = 0; e < 3; e++) out_goal[3 * k + e] = g[e];
  }
  return 0;
}

// ---- the same draws ON THE DEVICE (include/grx_capi.h, grx_fetch_sample_resets_device): the worlds' numpy PCG64 streams live in HBM ([N,4] uint64), one thread per
// world to reset walks its stream through the data-dependent loops of _reset_sim / _sample_goal (fetch_env.py:153-166, 388-391) in the reference's fp64 arithmetic
// (explicitly rounded multiplies / adds: no fused contraction, the host routine and numpy have none) and writes the float32 sample row the reset kernel reads.
// Bit-equal to the host routine and therefore to Generator.uniform (tests/test_gpu_fetch.py); nothing is drawn, staged or uploaded by the host.
// (`__dmul_rn` / `__dadd_rn` are plain `*` / `+` in HIP, and hipcc's default -ffp-contract=fast fuses them -- ignoring `#pragma clang fp contract(off)` -- into one fma: the
// fused a + (b - a) d differs from numpy's two roundings in ~12 % of the draws, in the last bit: invisible in the float32 Fetch rows, visible in the float64 Adroit rows.  The
// rounded product is therefore passed through an empty asm statement the optimiser cannot look through.)
__device__ __forceinline__ double grx_rounded(double x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ double grx_pcg64_uniform_dev(unsigned long long& hi, unsigned long long& lo, unsigned long long ihi, unsigned long long ilo, double a, double b) {
  const unsigned long long mhi = 0x2360ED051FC65DA4ULL, mlo = 0x4385DF649FCCF645ULL;
  const unsigned long long plo = lo * mlo, phi = __umul64hi(lo, mlo) + hi * mlo + lo * mhi;
  lo = plo + ilo;
  hi = phi + ihi + (lo < plo ? 1ULL : 0ULL);
  const unsigned long long x = hi ^ lo; const unsigned rot = (unsigned)(hi >> 58);
  const unsigned long long r = (x >> rot) | (x << ((64 - rot) & 63));
  const double d = __dmul_rn((double)(r >> 11), 1.0 / 9007199254740992.0);
  return __dadd_rn(a, grx_rounded(__dmul_rn(grx_rounded(__dsub_rn(b, a)), d)));
}
__global__ void __launch_bounds__(64)
grx_fetch_sample_kernel(unsigned long long* __restrict__ states, const int* __restrict__ idx, int n, int has_object, int in_air, double obj_range, double target_range,
                        double t0, double t1, double t2, double g0, double g1, double g2, double height_offset, float* __restrict__ samples) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= n) return;
  const int w = idx[k];
  unsigned long long hi = states[4 * w], lo = states[4 * w + 1];
  const unsigned long long ihi = states[4 * w + 2], ilo = states[4 * w + 3];
  double ox = g0, oy = g1;
  if (has_object) {
    int guard = 0;
    for (; guard < 65536; guard++) {      // (bounded: a kernel must not spin; the acceptance probability of the reference's ranges is 0.65)
      const double dx = __dsub_rn(ox, g0), dy = __dsub_rn(oy, g1);
      if (!(__dsqrt_rn(__dadd_rn(grx_rounded(__dmul_rn(dx, dx)), grx_rounded(__dmul_rn(dy, dy)))) < 0.1)) break;
      ox = __dadd_rn(g0, grx_pcg64_uniform_dev(hi, lo, ihi, ilo, -obj_range, obj_range));
      oy = __dadd_rn(g1, grx_pcg64_uniform_dev(hi, lo, ihi, ilo, -obj_range, obj_range));
    }
    // the loop gave up (an obj_range that can never clear the 0.1 m ring: the reference would spin forever): the sample is poisoned, NOT silently accepted -- the world's
    // state turns NaN and the engine raises its sticky GRX_STATUS_BADNUM bit at the next step
    if (guard == 65536) ox = oy = __longlong_as_double(0x7FF8000000000000LL);
  }
  double g[3];
  g[0] = __dadd_rn(g0, grx_pcg64_uniform_dev(hi, lo, ihi, ilo, -target_range, target_range));
  g[1] = __dadd_rn(g1, grx_pcg64_uniform_dev(hi, lo, ihi, ilo, -target_range, target_range));
  g[2] = __dadd_rn(g2, grx_pcg64_uniform_dev(hi, lo, ihi, ilo, -target_range, target_range));
  if (has_object) {
    g[0] = __dadd_rn(g[0], t0); g[1] = __dadd_rn(g[1], t1); g[2] = __dadd_rn(g[2], t2);
    g[2] = height_offset;
    if (in_air && grx_pcg64_uniform_dev(hi, lo, ihi, ilo, 0.0, 1.0) < 0.5) g[2] = __dadd_rn(g[2], grx_pcg64_uniform_dev(hi, lo, ihi, ilo, 0.0, 0.45));
  }
  states[4 * w] = hi; states[4 * w + 1] = lo;
  float* s = samples + 5 * (size_t)k;
  s[0] = (float)ox; s[1] = (float)oy; s[2] = (float)g[0]; s[3] = (float)g[1]; s[4] = (float)g[2];
}
extern "C" int grx_fetch_sample_resets_device(uint64_t* states, const int* idx, int n, int has_object, int target_in_the_air, double obj_range, double target_range,
                                              const double* target_offset, const double* gripper_xpos, double height_offset, float* samples, void* stream) {
  if (!states || !idx || !samples || !gripper_xpos || !target_offset) return fail("grx_fetch_sample_resets_device: null argument");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(grx_fetch_sample_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)states, idx, n, has_object, target_in_the_air, obj_range,
                     target_range, target_offset[0], target_offset[1], target_offset[2], gripper_xpos[0], gripper_xpos[1], gripper_xpos[2], height_offset, samples);
  HIP_OK(hipGetLastError());
  return 0;
}

// ---- reset_model's draws of AdroitHandHammer / Door / Relocate ON THE DEVICE (include/grx_capi.h, grx_adroit_sample_resets_device; adroit_hammer.py:374-376,
// adroit_door.py:362-370, adroit_relocate.py:353-372): one thread per world to reset draws the task's uniforms from the world's PCG64 stream in HBM (the reference's
// fp64 arithmetic, explicitly rounded) and rewrites the world's model edit: the fp64 body_pos row get_env_state reports, the fp32 shift pose the engine applies
// (translation = body_pos - the XML pose, identity rotation) and, for relocate, the target site.  reset_model rewrites only SOME components of body_pos (hammer: z,
// relocate: x / y): the others keep what a set_env_state wrote, as in the reference.  (The pen's draws stay on the host: euler2quat goes through sin / cos.)
__global__ void __launch_bounds__(64)
grx_adroit_sample_kernel(unsigned long long* __restrict__ states, const long long* __restrict__ idx, int n, int kind, double p0x, double p0y, double p0z,
                         double* __restrict__ edit, double* __restrict__ target64, float* __restrict__ shift, float* __restrict__ target) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= n) return;
  const int w = (int)idx[k];
  unsigned long long hi = states[4 * w], lo = states[4 * w + 1];
  const unsigned long long ihi = states[4 * w + 2], ilo = states[4 * w + 3];
  double ex = edit[3 * w], ey = edit[3 * w + 1], ez = edit[3 * w + 2];
  if (kind == 0) ez = grx_pcg64_uniform_dev(hi, lo, ihi, ilo, 0.1, 0.25);
  else if (kind == 1) { ex = grx_pcg64_uniform_dev(hi, lo, ihi, ilo, -0.3, -0.2); ey = grx_pcg64_uniform_dev(hi, lo, ihi, ilo, 0.25, 0.35); ez = grx_pcg64_uniform_dev(hi, lo, ihi, ilo, 0.252, 0.35); }
  else {
    ex = grx_pcg64_uniform_dev(hi, lo, ihi, ilo, -0.15, 0.15); ey = grx_pcg64_uniform_dev(hi, lo, ihi, ilo, -0.15, 0.3);
    const double tx = grx_pcg64_uniform_dev(hi, lo, ihi, ilo, -0.2, 0.2), ty = grx_pcg64_uniform_dev(hi, lo, ihi, ilo, -0.2, 0.2), tz = grx_pcg64_uniform_dev(hi, lo, ihi, ilo, 0.15, 0.35);
    target64[3 * w] = tx; target64[3 * w + 1] = ty; target64[3 * w + 2] = tz;
    target[3 * w] = (float)tx; target[3 * w + 1] = (float)ty; target[3 * w + 2] = (float)tz;
  }
  states[4 * w] = hi; states[4 * w + 1] = lo;
  edit[3 * w] = ex; edit[3 * w + 1] = ey; edit[3 * w + 2] = ez;
  float* s = shift + 7 * (size_t)w;
  s[0] = (float)__dsub_rn(ex, p0x); s[1] = (float)__dsub_rn(ey, p0y); s[2] = (float)__dsub_rn(ez, p0z); s[3] = 1.0f; s[4] = 0.0f; s[5] = 0.0f; s[6] = 0.0f;
}
extern "C" int grx_adroit_sample_resets_device(uint64_t* states, const int64_t* idx, int n, int kind, const double* shift_pos0, double* edit, double* target64, float* shift,
                                               float* target, void* stream) {
  if (!states || !idx || !shift_pos0 || !edit || !shift) return fail("grx_adroit_sample_resets_device: null argument");
  if (kind != 0 && kind != 1 && kind != 3) return fail("grx_adroit_sample_resets_device: kind must be 0 (hammer), 1 (door) or 3 (relocate)");
  if (kind == 3 && (!target || !target64)) return fail("grx_adroit_sample_resets_device: relocate needs the target rows");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(grx_adroit_sample_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)states, (const long long*)idx, n, kind, shift_pos0[0],
                     shift_pos0[1], shift_pos0[2], edit, target64, shift, target);
  HIP_OK(hipGetLastError());
  return 0;
}

// ---- MazeEnv.reset's draws ON THE DEVICE (include/grx_capi.h, grx_maze_sample_resets_device; maze_v4.py:299-358): goal cell, xy noise, the reset cell drawn until it is
// farther than half a cell from the goal, xy noise -- integers() and uniform() of the world's numpy PCG64 stream.  Generator.integers(0, K) for K <= 2^32 is Lemire's
// multiply-shift on 32-bit draws with rejection, and the bit generator hands out a 64-bit output in two 32-bit halves (low first, the high half buffered): the buffer is part of
// the stream state (states[5 w + 4] = has << 32 | value).  integers(0, 1) consumes nothing.  Validated draw for draw against numpy on the host twin of this routine
// (tests/test_cpu_host.py) and on the device (tests/test_gpu_maze.py).
__device__ __forceinline__ unsigned long long grx_pcg64_next64_dev(unsigned long long& hi, unsigned long long& lo, unsigned long long ihi, unsigned long long ilo) {
  const unsigned long long mhi = 0x2360ED051FC65DA4ULL, mlo = 0x4385DF649FCCF645ULL;
  const unsigned long long plo = lo * mlo, phi = __umul64hi(lo, mlo) + hi * mlo + lo * mhi;
  lo = plo + ilo;
  hi = phi + ihi + (lo < plo ? 1ULL : 0ULL);
  const unsigned long long x = hi ^ lo; const unsigned rot = (unsigned)(hi >> 58);
  return (x >> rot) | (x << ((64 - rot) & 63));
}
__device__ __forceinline__ unsigned grx_pcg64_next32_dev(unsigned long long& hi, unsigned long long& lo, unsigned long long ihi, unsigned long long ilo, unsigned long long& buf) {
  if (buf >> 32) { const unsigned v = (unsigned)buf; buf = 0ULL; return v; }
  const unsigned long long n = grx_pcg64_next64_dev(hi, lo, ihi, ilo);
  buf = (1ULL << 32) | (n >> 32);
  return (unsigned)n;
}
__device__ __forceinline__ int grx_pcg64_integers_dev(unsigned long long& hi, unsigned long long& lo, unsigned long long ihi, unsigned long long ilo, unsigned long long& buf, unsigned count) {
  if (count <= 1u) return 0;
  const unsigned rng = count - 1u;
  unsigned long long m = (unsigned long long)grx_pcg64_next32_dev(hi, lo, ihi, ilo, buf) * count;
  unsigned left = (unsigned)m;
  if (left < count) {
    const unsigned thr = (0xFFFFFFFFu - rng) % count;
    while (left < thr) { m = (unsigned long long)grx_pcg64_next32_dev(hi, lo, ihi, ilo, buf) * count; left = (unsigned)m; }
  }
  return (int)(m >> 32);
}
__global__ void __launch_bounds__(64)
grx_maze_sample_kernel(unsigned long long* __restrict__ states, const int* __restrict__ idx, int n, const double* __restrict__ goal_xy, int n_goal, const double* __restrict__ reset_xy,
                       int n_reset, double noise, double scaling, int fixed_goal, double fgx, double fgy, int fixed_reset, double frx, double fry, float* __restrict__ stage) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= n) return;
  const int w = idx[k];
  unsigned long long hi = states[5 * w], lo = states[5 * w + 1], buf = states[5 * w + 4];
  const unsigned long long ihi = states[5 * w + 2], ilo = states[5 * w + 3];
  double gx = fgx, gy = fgy;
  if (!fixed_goal) { const int c = grx_pcg64_integers_dev(hi, lo, ihi, ilo, buf, (unsigned)n_goal); gx = goal_xy[2 * c]; gy = goal_xy[2 * c + 1]; }
  gx = __dadd_rn(gx, grx_rounded(__dmul_rn(grx_pcg64_uniform_dev(hi, lo, ihi, ilo, -noise, noise), scaling)));
  gy = __dadd_rn(gy, grx_rounded(__dmul_rn(grx_pcg64_uniform_dev(hi, lo, ihi, ilo, -noise, noise), scaling)));
  double rx = frx, ry = fry;
  if (!fixed_reset) {
    rx = gx; ry = gy;
    const double far = __dmul_rn(0.5, scaling);
    // (bounded: with ONE reset cell that is also the goal cell the reference's loop never ends -- integers(0, 1) draws nothing; a kernel must not spin)
    int guard = 0;
    for (; guard < 65536; guard++) {
      const double dx = __dsub_rn(rx, gx), dy = __dsub_rn(ry, gy);
      if (!(__dsqrt_rn(__dadd_rn(grx_rounded(__dmul_rn(dx, dx)), grx_rounded(__dmul_rn(dy, dy)))) <= far)) break;
      const int c = grx_pcg64_integers_dev(hi, lo, ihi, ilo, buf, (unsigned)n_reset);
      rx = reset_xy[2 * c]; ry = reset_xy[2 * c + 1];
    }
    if (guard == 65536) rx = ry = __longlong_as_double(0x7FF8000000000000LL);   // never silently: a poisoned reset position raises GRX_STATUS_BADNUM (the host refuses the degenerate maze up front, envs/point_maze.py)
  }
  rx = __dadd_rn(rx, grx_rounded(__dmul_rn(grx_pcg64_uniform_dev(hi, lo, ihi, ilo, -noise, noise), scaling)));
  ry = __dadd_rn(ry, grx_rounded(__dmul_rn(grx_pcg64_uniform_dev(hi, lo, ihi, ilo, -noise, noise), scaling)));
  states[5 * w] = hi; states[5 * w + 1] = lo; states[5 * w + 4] = buf;
  float* s = stage + 4 * (size_t)k;
  s[0] = (float)rx; s[1] = (float)ry; s[2] = (float)gx; s[3] = (float)gy;
}
extern "C" int grx_maze_sample_resets_device(uint64_t* states, const int* idx, int n, const double* goal_xy, int n_goal, const double* reset_xy, int n_reset, double noise_range,
                                             double scaling, const double* fixed_goal_xy, const double* fixed_reset_xy, float* stage, void* stream) {
  if (!states || !idx || !goal_xy || !reset_xy || !stage) return fail("grx_maze_sample_resets_device: null argument");
  if (n_goal < 1 || n_reset < 1) return fail("grx_maze_sample_resets_device: empty cell list");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(grx_maze_sample_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)states, idx, n, goal_xy, n_goal, reset_xy, n_reset, noise_range,
                     scaling, fixed_goal_xy ? 1 : 0, fixed_goal_xy ? fixed_goal_xy[0] : 0.0, fixed_goal_xy ? fixed_goal_xy[1] : 0.0, fixed_reset_xy ? 1 : 0,
                     fixed_reset_xy ? fixed_reset_xy[0] : 0.0, fixed_reset_xy ? fixed_reset_xy[1] : 0.0, stage);
  HIP_OK(hipGetLastError());
  return 0;
}

// count consecutive Generator.uniform(-1, 1) draws of each listed world's numpy PCG64 stream, as float32 rows (FrankaKitchen's observation noise:
// franka_env.py:118-127 + kitchen_env.py:361-369 draw 9 + 9 + 21 + 20 per observation).  states as in grx_fetch_sample_resets; HOST pointers.
extern "C" int grx_sample_uniform_rows(uint64_t* states, const int64_t* idx, int n, int count, float* out) {
  if (!states || !out) return fail("grx_sample_uniform_rows: null argument");
  for (int k = 0; k < n; k++) {
    uint64_t* st = states + 4 * (idx ? idx[k] : k);
    for (int e = 0; e < count; e++) out[(size_t)k * count + e] = (float)grx_pcg64_uniform(st, -1.0, 1.0);
  }
  return 0;
}

// ---- the same streams advanced ON THE DEVICE (include/grx_capi.h, grx_uniform_rows_device): one thread per world walks its numpy PCG64 stream -- the 128-bit
// LCG step in 64-bit halves (mul-hi for the carry), the XSL-RR output, numpy's 53-bit double and uniform(-1, 1) = -1 + 2 d in fp64 (2 d is exact, so an fma
// contraction rounds like the two operations), rounded to float32 like the host routine: bit-equal rows (tests/test_gpu_kitchen.py), no upload, no host loop.
__global__ void __launch_bounds__(256)
grx_uniform_rows_kernel(unsigned long long* __restrict__ states, const unsigned char* __restrict__ mask, int n, int count, float* __restrict__ out) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= n || (mask && !mask[w])) return;
  unsigned long long hi = states[4 * w], lo = states[4 * w + 1];
  const unsigned long long ihi = states[4 * w + 2], ilo = states[4 * w + 3];
  const unsigned long long mhi = 0x2360ED051FC65DA4ULL, mlo = 0x4385DF649FCCF645ULL;
  for (int e = 0; e < count; e++) {
    const unsigned long long plo = lo * mlo, phi = __umul64hi(lo, mlo) + hi * mlo + lo * mhi;   // (hi:lo) * (mhi:mlo) mod 2^128
    lo = plo + ilo;
    hi = phi + ihi + (lo < plo ? 1ULL : 0ULL);
    const unsigned long long x = hi ^ lo; const unsigned rot = (unsigned)(hi >> 58);
    const unsigned long long r = (x >> rot) | (x << ((64 - rot) & 63));
    const double d = (double)(r >> 11) * (1.0 / 9007199254740992.0);
    out[(size_t)w * count + e] = (float)(-1.0 + 2.0 * d);
  }
  states[4 * w] = hi; states[4 * w + 1] = lo;
}
extern "C" int grx_uniform_rows_device(uint64_t* states, const unsigned char* mask, int n, int count, float* out, void* stream) {
  if (!states || !out) return fail("grx_uniform_rows_device: null argument");
  if (n <= 0 || count <= 0) return 0;
  hipLaunchKernelGGL(grx_uniform_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)states, mask, n, count, out);
  HIP_OK(hipGetLastError());
  return 0;
}

// ---- KitchenEnv.step's task bookkeeping on the device (kitchen_env.py:386-423; include/grx_capi.h, grx_kitchen_bookkeeping): one thread per world
struct GrxKitchenBook {
  const int* completed; const unsigned char* stepped;   // completion bits of this observation (step kernel); worlds this step covered (null = all)
  int *tasks_to_complete, *episode_completions, *elapsed, *step_completions;
  float* reward; unsigned char *terminated, *truncated, *needs_reset, *reset_now;
  float *qpos, *qvel, *qacc_ws; const float* init_qpos;
  int nq, nv, all_mask, max_steps, remove_when_completed, terminate_when_completed, mode;   // mode 0 disabled, 1 next_step, 2 same_step
  int* final_info;   // [N,3] or null: (tasks_to_complete, step_task_completions, episode_task_completions) of a same-step reset world's FINISHED episode (info["final_info"])
};
static_assert(sizeof(grx_kitchen_book) == sizeof(GrxKitchenBook), "grx_kitchen_book must mirror GrxKitchenBook");
__global__ void __launch_bounds__(256)
grx_kitchen_book_kernel(GrxKitchenBook a, int n) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= n) return;
  const int stepped = a.stepped ? a.stepped[w] != 0 : 1;
  const int pending = a.mode == 1 && a.needs_reset[w];     // next_step: the world finished its episode in the previous step and resets in this one
  int ttc = a.tasks_to_complete[w], epi = a.episode_completions[w], el = a.elapsed[w];
  const int step_done = stepped ? (a.completed[w] & ttc) : 0;
  if (a.remove_when_completed) ttc &= ~step_done;
  epi |= step_done;
  const int term = a.terminate_when_completed && stepped && epi == a.all_mask;
  if (stepped) el++;
  const int trunc = a.max_steps > 0 && stepped && el >= a.max_steps;
  const int done = term | trunc;
  const int reset_now = pending || (a.mode == 2 && done);
  a.reward[w] = pending ? 0.0f : (float)__popc((unsigned)step_done);
  a.terminated[w] = (unsigned char)term; a.truncated[w] = (unsigned char)trunc;
  a.step_completions[w] = (a.mode == 2 && done) ? 0 : step_done;      // a same-step reset world reports its NEW episode (the finished one is in the final_* rows)
  if (a.mode == 1) a.needs_reset[w] = (unsigned char)(pending ? 0 : (a.needs_reset[w] | done));
  a.reset_now[w] = (unsigned char)reset_now;
  if (a.final_info && a.mode == 2 && done) { a.final_info[3 * w] = ttc; a.final_info[3 * w + 1] = step_done; a.final_info[3 * w + 2] = epi; }
  if (reset_now) {      // FrankaRobot.reset_model (franka_env.py:133-139): init_qpos, zero velocity; mj_resetData zeroes the warm start
    ttc = a.all_mask; epi = 0; el = 0;
    for (int i = 0; i < a.nq; i++) a.qpos[(size_t)w * a.nq + i] = a.init_qpos[i];
    for (int i = 0; i < a.nv; i++) { a.qvel[(size_t)w * a.nv + i] = 0.0f; a.qacc_ws[(size_t)w * a.nv + i] = 0.0f; }
  }
  a.tasks_to_complete[w] = ttc; a.episode_completions[w] = epi; a.elapsed[w] = el;
}
extern "C" int grx_kitchen_bookkeeping(const grx_kitchen_book* args, int n_worlds, void* stream) {
  if (!args) return fail("grx_kitchen_bookkeeping: null argument");
  GrxKitchenBook a; memcpy(&a, args, sizeof(a));
  if (!a.completed || !a.tasks_to_complete || !a.episode_completions || !a.elapsed || !a.step_completions || !a.reward || !a.terminated || !a.truncated || !a.needs_reset ||
      !a.reset_now || !a.qpos || !a.qvel || !a.qacc_ws || !a.init_qpos) return fail("grx_kitchen_bookkeeping: null buffer");
  if (a.mode < 0 || a.mode > 2 || a.nq <= 0 || a.nv <= 0) return fail("grx_kitchen_bookkeeping: bad mode / dimensions");
  if (n_worlds <= 0) return 0;
  hipLaunchKernelGGL(grx_kitchen_book_kernel, dim3((unsigned)((n_worlds + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, n_worlds);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int grx_fetch_compute_reward(const float* achieved, const float* desired, int64_t batch, double distance_threshold, int sparse,
                                        float* reward_out, void* stream) {
  if (!achieved || !desired || !reward_out) return fail("grx_fetch_compute_reward: null argument");
  if (batch <= 0) return 0;
  long long blocks = (batch + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(grx_fetch_reward_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, achieved, desired, (long long)batch,
                     distance_threshold, sparse, reward_out);
  HIP_OK(hipGetLastError());
  return 0;
}

#endif  // GRX_TU_API
