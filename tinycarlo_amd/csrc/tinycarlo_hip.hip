// tinycarlo_hip.hip -- libtinycarlo_hip.so: kernels + C ABI (include/tinycarlo_hip.h).
//
// Build (csrc/Makefile): hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared
//
// One tc_step = two launches, one 64-lane wavefront (= one workgroup) per env in both:
//   tc_env_kernel<K>      phase A  kinematics + lanepath tracking + CTE/heading   (car.py:70-148, 46-53)
//                         phase B  nearest lane-line edge per layer + distance    (layer.py:33-44, car.py:55-64)
//                         phase C  camera: transform -> 4 clip passes -> project -> visibility -> draw list
//                                  (camera.py:52-110), handed over through a small global buffer
//   tc_raster_kernel<..>  cv2.polylines of the draw list into LDS bit-planes (renderer.py:36-51), expanded to
//                         uint8 and stored with 16-byte-per-lane coalesced stores (zeros included: a class-mask
//                         frame is written exactly once)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/tinycarlo_hip.h"
#include "tc_device.h"
#include "tc_rng.h"

// Timing build only (make timing -> libtinycarlo_hip_timing.so, used by tools/phase_clock.py): every wavefront stores
// the shader clock at its phase boundaries.  The shipped library is compiled without TC_TIMING and contains none of it.
#if defined(TC_TIMING) && defined(TC_TIMING_LOOP)
// variant of the timing build (make dev-timing-loop): no phase probes, only the period of every step of a K-step launch
// (slot k = clocks from the top of step k-1 to the top of step k, k < 32; slot 0 = launch entry -> top of step 0)
__device__ long long* tc_tstamp = nullptr;
#define TSTAMP(i) \
  do {            \
  } while (0)
#define TSTAMP_REAL(i) \
  do {                 \
  } while (0)
#define TSTAMP_LOOP(k, nsteps, tprev)                                                      \
  do {                                                                                     \
    long long _t = clock64();                                                              \
    long long* _tp = tc_tstamp;                                                            \
    if (_tp && threadIdx.x == 0 && (k) < 31) _tp[(size_t)env * 32 + (k)] = _t - (tprev);  \
    (tprev) = _t;                                                                          \
  } while (0)
#define TSTAMP_END(tprev)                                                                  \
  do {                                                                                     \
    long long* _tp = tc_tstamp;                                                            \
    if (_tp && threadIdx.x == 0) {                                                         \
      _tp[(size_t)env * 32 + 31] = clock64() - (tprev);                                    \
      _tp[(size_t)env * 32 + 30] = (long long)(unsigned)__builtin_amdgcn_s_getreg(63492) | /* HW_REG_HW_ID */ \
                                   ((long long)(unsigned)__builtin_amdgcn_s_getreg(63508) << 32); /* XCC_ID */ \
    }                                                                                      \
  } while (0)
#elif defined(TC_TIMING) && defined(TC_TIMING_LDS)
// Variant for the frame / step kernels (make dev-timing-lds): the stamps go to 256 bytes of LDS -- no vector-memory
// operation, nothing drained -- and one lane copies them out when the frame is done.  A stamp written with a global store
// (the variant below) puts that store's round trip into the NEXT s_waitcnt vmcnt(0) of the wavefront, i.e. the probes
// themselves create the longest "phase" of a wavefront that otherwise has no store in flight; this variant shows the
// latencies the shipped kernel has.  (s_memtime returns through the scalar-memory counter, so a stamp still waits for the
// LDS queue: almost every probe sits behind an lds_sync anyway.)
__device__ long long* tc_tstamp = nullptr;  // [N][32]
__shared__ long long tc_ts_lds[32];
#define TSTAMP(i)                                              \
  do {                                                         \
    if (threadIdx.x == 0) tc_ts_lds[(i)] = clock64();          \
  } while (0)
#define TSTAMP_REAL(i)                                         \
  do {                                                         \
    if (threadIdx.x == 0) tc_ts_lds[(i)] = wall_clock64();     \
  } while (0)
#define TSTAMP_LOOP(k, nsteps, tprev) \
  do {                                \
  } while (0)
#define TSTAMP_END(tprev) \
  do {                    \
  } while (0)
#define TSTAMP_DUMP(env_)                                                                  \
  do {                                                                                     \
    long long* _tp = tc_tstamp;                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                     \
    if (_tp && threadIdx.x < 32) _tp[(size_t)(env_) * 32 + threadIdx.x] = tc_ts_lds[threadIdx.x]; \
  } while (0)
#define TSTAMP_CLEAR()                                         \
  do {                                                         \
    if (threadIdx.x < 32) tc_ts_lds[threadIdx.x] = 0;          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         \
  } while (0)
#elif defined(TC_TIMING)
__device__ long long* tc_tstamp = nullptr;  // [N][32]
// (outstanding memory operations are drained first, so a phase is charged with the latencies it started)
#define TSTAMP(i)                                                                          \
  do {                                                                                     \
    long long* _tp = tc_tstamp;                                                            \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                            \
    if (_tp && threadIdx.x == 0) _tp[(size_t)env * 32 + (i)] = clock64();                  \
  } while (0)
// the constant 100 MHz counter next to the shader clock: (stamp 13 - stamp 0) / (stamp 31 - stamp 30) x 100 MHz is the
// clock the chip actually holds while the kernel runs
// loop period of the step loop: slot 28 = clocks from the top of the previous step to the top of the last one
#define TSTAMP_LOOP(k, nsteps, tprev)                                                      \
  do {                                                                                     \
    long long _t = clock64();                                                              \
    long long* _tp = tc_tstamp;                                                            \
    if (_tp && threadIdx.x == 0 && (k) == (nsteps)-1 && (k) > 0) _tp[(size_t)env * 32 + 28] = _t - (tprev); \
    (tprev) = _t;                                                                          \
  } while (0)
#define TSTAMP_REAL(i)                                                                     \
  do {                                                                                     \
    long long* _tp = tc_tstamp;                                                            \
    if (_tp && threadIdx.x == 0) _tp[(size_t)env * 32 + (i)] = wall_clock64();            \
  } while (0)
#define TSTAMP_END(tprev) \
  do {                    \
  } while (0)
#elif defined(TC_MARKERS)
// tools/kernel_stats.py --sections: the probes become assembler comments, so that the static instruction mix between two
// probes can be read off the compiler's output (no code is emitted for them)
#define TC_STR2(x) #x
#define TC_STR(x) TC_STR2(x)
#define TSTAMP(i) asm volatile("; @@TS " TC_STR(i) ::: "memory")
#define TSTAMP_REAL(i) \
  do {                 \
  } while (0)
#define TSTAMP_LOOP(k, nsteps, tprev) \
  do {                                \
  } while (0)
#define TSTAMP_END(tprev) \
  do {                    \
  } while (0)
#else
#define TSTAMP(i) \
  do {            \
  } while (0)
#define TSTAMP_REAL(i) \
  do {                 \
  } while (0)
#define TSTAMP_LOOP(k, nsteps, tprev) \
  do {                                \
  } while (0)
#define TSTAMP_END(tprev) \
  do {                    \
  } while (0)
#endif
#ifndef TSTAMP_DUMP
#define TSTAMP_DUMP(env_) \
  do {                    \
  } while (0)
#define TSTAMP_CLEAR() \
  do {                 \
  } while (0)
#endif
// finer section marks for tools/kernel_stats.py --sections (nothing in any other build)
#ifdef TC_MARKERS
#define MARK(name) asm volatile("; @@MK " name ::: "memory")
#else
#define MARK(name) \
  do {             \
  } while (0)
#endif

#define TC_PROF_RING 64
#define TC_RING_SLOTS 3  // chunks of a K-step call whose pose rows / draw lists exist at once (tc_env_reserve_steps)
#define MODE_STEP 0
#define MODE_RESET 1
#define MODE_RENDER 2

// ablation switches for profiling (not part of the ABI contract; results are wrong when set)
#define DBG_SKIP_RASTER 0x100u
#define DBG_SKIP_STORE 0x200u
#define DBG_SKIP_CAMERA 0x400u
#define DBG_SKIP_DIST 0x800u
#define DBG_SKIP_R2 0x1000u
#define DBG_SKIP_R3 0x2000u
#define DBG_SKIP_R4 0x4000u
#define DBG_SKIP_EVENTS 0x10000u
#define DBG_SKIP_LINESETUP 0x20000u
#define DBG_SKIP_SLOPE 0x40000u
#define DBG_SKIP_QUAD 0x80000u
#define DBG_SKIP_CLIP 0x100000u
#define DBG_SKIP_DRAWLIST 0x200000u
// the kernels test them only in the ablation build (make dev-ablate): in the shipped library every test folds to false,
// so the switches cost no scalar registers there (they were ~25 live conditions at the head of the raster stage)
#ifdef TC_ABLATE
#define DBG_ON(word, f) (((word) & (f)) != 0)
#else
#define DBG_ON(word, f) false
#endif

// Every workgroup of the stage kernels is ONE wavefront, and the LDS executes a wavefront's operations in program order:
// between phases that hand data from lane to lane through LDS nothing has to be waited for except the LDS queue itself
// (and the compiler kept from moving memory operations across).  __syncthreads() would also drain the vector-memory
// counter -- the frame's stores still in flight at the end of a band, the next batch's draw-list loads.
// The same goes for the simulate stage: a wait for vmcnt there would sit out the round trip of the step's rollout-row /
// pose-row stores (vector-memory operations retire in issue order).
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// A reference into the kernel-argument segment, passed through an empty asm: the compiler can no longer tell that two
// reads go to the same block, so values loaded behind this point are not merged with (and kept alive from) earlier
// loads, nor hoisted out of the loop the call sits in -- they are s_load'ed where they are used.
template <class T>
__device__ __forceinline__ const T& kernarg_again(const T& r) {
  unsigned long long p = (unsigned long long)&r;
  asm volatile("" : "+s"(p));
  return *(const T*)(const __attribute__((address_space(4))) T*)p;
}

// Segments rasterised per batch: 32 (64 * 106 bytes of tables per workgroup), or 16 in the kernels of the big-map
// variants (K >= 8: knuffingen), where the tables' 6.6 KB are what decides how many workgroups a CU holds.
#define RB_MAX 32
#define RB_OF_K(K) ((K) >= 8 ? 16 : 32)
// raster-stage LDS layout (compile-time: folds into instruction offsets and frees SGPRs)
#define R_OFF_TAB 0
#define R_TAB_BYTES_OF(RB) (((RB) * 4 * 5 + (RB) * 4 + 5 * (RB) + 8 * (RB)) * 4 + 2 * 4 * (RB) * 8)
#define R_OFF_BITS_OF(RB) ((R_TAB_BYTES_OF(RB) + 15) / 16 * 16)
#define LCH 8        // outline steps per chunk

#define TC_MAX_GROUPS 8
struct LdsLayout {
  int off_p, off_flg, off_list, off_cnt;
  int total;
  int off_live;  // tc_step_multi: the env's state between two steps (beyond both stages' buffers, never aliased)
};

// The env's state and step outputs live in this LDS record for the whole launch: live_in() fills it from the caller's
// buffers, every step reads and rewrites it (one wavefront, program order), live_out() stores it back.  The step body
// therefore has ONE form whether it is the only step of a launch (tc_step) or one of many (tc_step_multi), and the
// raster stage keeps none of it in registers.
struct LiveLds {
  double d[10];  // x, y, theta, velocity, steering, radius, front_x, front_y, cos(theta), sin(theta)
  double cte, he, reward;
  double dist[TC_MAX_LAYERS];
  int lp[8];
  int lp_len, last_maneuver;
  int have_trig;    // d[8], d[9] hold cos / sin of d[2] (set by every update of the front axle)
  int needs_reset;  // TC_F_AUTORESET: re-spawn at the start of the next step
  int cursor;       // spawn_cursor: re-spawns of this env so far
  int cursor0;      // its value in the caller's buffer (stored back only when it changed)
  int trunc, status, terminated, pad[3];
  int cnt[TC_MAX_TERMS];  // steps_true of the consecutive-step terms
  int ne[TC_MAX_LAYERS];
};
#define TC_LIVE_BYTES 416
static_assert(sizeof(LiveLds) <= TC_LIVE_BYTES, "LiveLds must fit its LDS slot");

// per-step rollout outputs of tc_step_multi, already advanced to the current step (each [N] or NULL)
struct RollStep {
  double *reward, *cte, *heading_error;
  unsigned char *terminated, *truncated;
  int* status;
  double *x, *y, *theta, *velocity;
  double* dist;  // [N][C]
  int* ne;       // [N][C]
  int* lp;       // [N][8]
  int* lp_len;
  size_t row0;   // first row of the step in the [K][N] arrays above (k * N)
};

struct KArgs {
  DevMap m;
  DevCar car;
  DevCam cam;
  tc_buffers b;
  LdsLayout lds;
  int N;
  int env0;    // first env of this launch (a step may be issued as several sub-batches)
  const double* cam_E;  // optional per-env extrinsics [N][12] (NULL: cam.E for every env)
  const double* cam_K;  // optional per-env intrinsics [N][9]
  int* seg_g;  // [N][seg_cap][5] draw list of the current frame (library owned)
  int* seg_n;  // [N]
  int seg_cap;
  const tc_term* terms;  // device table [TC_MAX_TERMS] (library owned); reward / termination wrappers
  int n_terms;
  int* term_counters;    // [N][TC_MAX_TERMS] (caller owned)
  // camera layer groups: phase C handles lane-line layers grp_layer[g] .. grp_layer[g+1]-1 together (camera.py's
  // per-layer loop makes the layers independent); the LDS node buffer holds cap_nodes nodes, the largest group
  // Group g = nodes grp_n0[g] .. grp_n0[g+1]-1 and edges grp_e0[g] .. grp_e0[g+1]-1, whose layers are grp_l0[g] ..
  // grp_l1[g]-1.  Either whole layers of the map's own arrays (cam_nodes == NULL), or -- maps that need groups -- whole
  // CONNECTED COMPONENTS of the lane-line graph in the env's camera copy of the map (cam_nodes / cam_edges_g: the same
  // nodes and edges, every layer's edges still contiguous and in their original relative order, components side by
  // side): camera.py moves a node only along its own edges, so components are as independent as layers are, and a
  // group no longer has to hold the largest layer (knuffingen: 517 nodes) -- see tc_env_create.
  int n_grp;
  int grp_layer[TC_MAX_GROUPS + 1];
  int grp_n0[TC_MAX_GROUPS + 1], grp_e0[TC_MAX_GROUPS + 1];
  int grp_l0[TC_MAX_GROUPS], grp_l1[TC_MAX_GROUPS];
  const double2* cam_nodes;
  const int2* cam_edges_g;
  int cap_nodes;
  const int* spawn_tab;  // TC_F_DEVICE_SPAWN: spawnable candidate nodes (library owned), spawn_n > 0 entries
  int spawn_n;
  unsigned long long spawn_seed;
  unsigned int dbg;  // DBG_* ablation switches for the camera stage of tc_frame_kernel (0 in every other launch)
  // tc_frame_kernel: the first seg_lds_cap entries of the frame's draw list stay in LDS (byte offset seg_lds_off of the
  // workgroup's block, behind both stages' buffers) instead of going through global memory; 0 in every other launch
  int seg_lds_off, seg_lds_cap;
};

// ---------------------------------------------------------------------------------------------
// Reward / termination wrappers (tinycarlo/wrapper/*.py) as an epilogue of the step.  Every lane runs the same
// scalar code; dist_l is lane l's laneline distance (lanes >= C hold 0) and is broadcast layer by layer so that
// the additions happen in the dict order of info["laneline_distances"] (env.py:85, reward.py:20,40).
__device__ inline double d_linear_reward(double x, double max_x, double max_reward, double min_reward) {
  double y = (-max_reward / max_x) * tc_fabs(x) + max_reward;  // utils.py:33
  if (max_reward > 0) return (min_reward > y) ? min_reward : y;  // python max(y, min_reward)
  return (min_reward < y) ? min_reward : y;                      // python min(y, min_reward)
}

// value of lane `l` (wave-uniform index): v_readlane instead of a shuffle through the LDS crossbar
__device__ __forceinline__ double d_readlane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// wave-uniform value as something the compiler knows to be uniform (first active lane's copy, held in SGPRs)
__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni_d(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// (scalars by value: a reference to the kernel-argument struct would force a copy of it into scratch)
// my_cnt: lane t < TC_MAX_TERMS holds steps_true of term slot t (loaded at kernel start, stored by the caller)
__device__ __forceinline__ void d_apply_terms(const tc_term* terms, int n_terms, int& my_cnt, double tw, int tid, int C,
                                              double cte, double vel, double dist_l, double& reward, int& terminated) {
  const double half = tw / 2;
  for (int t = 0; t < n_terms; t++) {
    const tc_term* T = terms + t;
    const int kind = T->kind;
    const unsigned mask = T->layer_mask;
    if (kind == TC_T_LANELINE_SPARSE_REWARD) {  // reward.py:20-21, utils.py:15-19
      double local = 0.0;
      for (int l = 0; l < C; l++) {
        const double d = d_readlane(dist_l, l);
        if (((mask >> l) & 1u) && d < half) local += T->per_layer[l];
      }
      reward = reward + local;
    } else if (kind == TC_T_LANELINE_LINEAR_REWARD) {  // reward.py:40-41
      for (int l = 0; l < C; l++) {
        const double d = d_readlane(dist_l, l);
        reward = reward + d_linear_reward(d, tw, T->per_layer[l], 0.0);
      }
    } else if (kind == TC_T_CTE_SPARSE_REWARD) {  // reward.py:60
      double local = 0.0;
      if (tc_fabs(cte) <= T->p[0]) local += T->p[1];
      reward = reward + local;
    } else if (kind == TC_T_CTE_LINEAR_REWARD) {  // reward.py:83
      reward = reward + d_linear_reward(cte, T->p[0], T->p[1], T->p[2]);
    } else if (kind == TC_T_LANELINE_CROSSING_TERMINATION) {  // termination.py:19-21
      for (int l = 0; l < C; l++) {
        const double d = d_readlane(dist_l, l);
        if (((mask >> l) & 1u) && d <= half) terminated = 1;
      }
    } else if (kind == TC_T_CTE_TERMINATION || kind == TC_T_CRASH_TERMINATION) {  // termination.py:39-47,61-69
      const bool cond = kind == TC_T_CTE_TERMINATION ? (tc_fabs(cte) > T->p[0]) : (tc_fabs(vel) < T->p[0]);
      int c = __builtin_amdgcn_readlane(my_cnt, t);
      if (cond) {
        c += 1;
        if (c >= T->number_of_steps) {
          terminated = 1;
          c = 0;
        }
      } else {
        c = 0;
      }
      if (tid == t) my_cnt = c;
    }
  }
}

// The same stack for the grouped simulate kernel, where every lane of an env's group runs it: distances and counters
// of the env sit in a small LDS record (all lanes of the group read and write the same words with the same values).
__device__ __forceinline__ void d_apply_terms_mem(const tc_term* terms, int n_terms, int* cnt, double tw, int C, double cte,
                                                  double vel, const double* dist, double& reward, int& terminated) {
  const double half = tw / 2;
  for (int t = 0; t < n_terms; t++) {
    const tc_term* T = terms + t;
    const int kind = T->kind;
    const unsigned mask = T->layer_mask;
    if (kind == TC_T_LANELINE_SPARSE_REWARD) {  // reward.py:20-21, utils.py:15-19
      double local = 0.0;
      for (int l = 0; l < C; l++)
        if (((mask >> l) & 1u) && dist[l] < half) local += T->per_layer[l];
      reward = reward + local;
    } else if (kind == TC_T_LANELINE_LINEAR_REWARD) {  // reward.py:40-41
      for (int l = 0; l < C; l++) reward = reward + d_linear_reward(dist[l], tw, T->per_layer[l], 0.0);
    } else if (kind == TC_T_CTE_SPARSE_REWARD) {  // reward.py:60
      double local = 0.0;
      if (tc_fabs(cte) <= T->p[0]) local += T->p[1];
      reward = reward + local;
    } else if (kind == TC_T_CTE_LINEAR_REWARD) {  // reward.py:83
      reward = reward + d_linear_reward(cte, T->p[0], T->p[1], T->p[2]);
    } else if (kind == TC_T_LANELINE_CROSSING_TERMINATION) {  // termination.py:19-21
      for (int l = 0; l < C; l++)
        if (((mask >> l) & 1u) && dist[l] <= half) terminated = 1;
    } else if (kind == TC_T_CTE_TERMINATION || kind == TC_T_CRASH_TERMINATION) {  // termination.py:39-47,61-69
      const bool cond = kind == TC_T_CTE_TERMINATION ? (tc_fabs(cte) > T->p[0]) : (tc_fabs(vel) < T->p[0]);
      int c = cnt[t];
      if (cond) {
        c += 1;
        if (c >= T->number_of_steps) {
          terminated = 1;
          c = 0;
        }
      } else {
        c = 0;
      }
      cnt[t] = c;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Per-lane register cache of the map: lane `tid` keeps nodes / edges (w*K + k)*64 + tid, k < K, of window w.
// Maps up to 64*K nodes and edges (all bundled ones with K = 5 or 13) are a single window that is loaded once at
// kernel start -- behind the latency of phase A -- and then serves every pass of phases B and C; larger maps
// reload window by window inside each pass.
template <int K>
struct MapCache {
  double2 nd[K];
  int2 ed[K];
};

// slot k of lane tid <-> node (edge) base + k*64 + tid, valid below `end`
template <int K>
__device__ inline void cache_nodes(MapCache<K>& c, const DevMap& m, int base, int end, int tid) {
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int i = base + k * TC_NT + tid;
    c.nd[k] = i < end ? m.nodes[i] : make_double2(0.0, 0.0);
  }
}
template <int K>
__device__ inline void cache_nodes_from(MapCache<K>& c, const double2* nodes, int base, int end, int tid) {
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int i = base + k * TC_NT + tid;
    c.nd[k] = i < end ? nodes[i] : make_double2(0.0, 0.0);
  }
}
template <int K>
__device__ inline void cache_edges_from(MapCache<K>& c, const int2* edges_g, int base, int end, int nbase, int tid) {
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int e = base + k * TC_NT + tid;
    int2 ed = e < end ? edges_g[e] : make_int2(nbase, nbase);
    c.ed[k] = make_int2(ed.x - nbase, ed.y - nbase);
  }
}
// node ids are stored relative to `nbase` (the first node of the camera group; 0 for whole-map use)
template <int K>
__device__ inline void cache_edges(MapCache<K>& c, const DevMap& m, int base, int end, int nbase, int tid) {
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int e = base + k * TC_NT + tid;
    int2 ed = e < end ? m.edges_g[e] : make_int2(nbase, nbase);
    c.ed[k] = make_int2(ed.x - nbase, ed.y - nbase);
  }
}

// camera.py:70-86: one of the four fix-up loops.  `bit` selects the membership flag (1 = idx_front,
// 2 = idx_in_range).  The reference builds the edge list first and then mutates nodes in list
// (= edge) order; updates of different target nodes are independent (a target is never read as
// the "other" end within one pass), so each target node's chain is replayed in ascending edge
// index by the lane that owns the chain's first edge.  List entries carry (edge, target | other << 16)
// so the chain loops touch LDS only.
// Works on one camera group: edges ge0 .. ge0+ne-1 (ids below are relative to ge0), node ids relative to gn0.
// camera.py:112-122 __point_on_line_at_z(p0 = P[o], p1 = P[t], tz), stored into P[t]
__device__ __forceinline__ void cam_move_to_plane(double* Px, double* Py, double* Pz, int o, int t, double tz) {
  double d0 = Px[o] - Px[t], d1 = Py[o] - Py[t], d2 = Pz[o] - Pz[t];
  if (d2 == 0) {
    double qn = __longlong_as_double(0x7ff8000000000000LL);
    Px[t] = qn;
    Py[t] = qn;
    Pz[t] = qn;
  } else {
    double tt = (tz - Pz[t]) / d2;
    double aa = Px[t] + tt * d0, bb = Py[t] + tt * d1, cc = Pz[t] + tt * d2;
    Px[t] = aa;
    Py[t] = bb;
    Pz[t] = cc;
  }
}

// The exact replay of one fix-up pass from its list of n straddling edges (edge id, target | other << 16): every target
// node's chain of moves in ascending edge index, run by the lane that holds the chain's first edge.  range_too: the
// target's idx_in_range bit is refreshed from its new depth (camera.py:80 reads the depths passes 1-2 left behind).
__device__ inline void cam_chain_replay(double* Px, double* Py, double* Pz, unsigned char* flg, int bit, double tz,
                                        const int* list, int n, double max_range, bool range_too, int tid) {
  for (int k = tid; k < n; k += TC_NT) {
    const int e = list[2 * k];
    const int t = list[2 * k + 1] & 0xffff;
    bool first = true;
    for (int j = 0; j < n; j++)
      if ((list[2 * j + 1] & 0xffff) == t && list[2 * j] < e) first = false;
    if (!first) continue;
    int cur = e, o = (unsigned)list[2 * k + 1] >> 16;  // the end that stays
    for (int guard = 0; guard < n; guard++) {
      cam_move_to_plane(Px, Py, Pz, o, t, tz);
      int nxt = 0x7fffffff, no = 0;
      for (int j = 0; j < n; j++) {
        const int ej = list[2 * j], pj = list[2 * j + 1];
        if ((pj & 0xffff) == t && ej > cur && ej < nxt) {
          nxt = ej;
          no = (unsigned)pj >> 16;
        }
      }
      if (nxt == 0x7fffffff) break;
      cur = nxt;
      o = no;
    }
    unsigned int f = flg[t] | (unsigned)bit;
    if (range_too) f = (f & ~2u) | (Pz[t] > -max_range ? 2u : 0u);
    flg[t] = (unsigned char)f;
  }
}

__device__ inline int2 cam_project(const double* K, double X, double Y, double Z, double& u, double& v);

typedef __attribute__((address_space(3))) int* LdsIntPtr;  // (explicitly LDS: a select between it and a global pointer
                                                            // cannot be folded into one flat access)

// rank of this lane among the set bits of a wave ballot
__device__ __forceinline__ int wave_rank(unsigned long long m) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// Phase C of one camera group whose nodes and edges sit in the wavefront's registers (one window: every bundled map).
// camera.py:52-110 reads the whole node / edge arrays in each of its steps; here the flags of an edge's two ends are
// carried in a register per edge slot (`ff`) and refreshed only after a pass that moved something, so a fix-up pass in
// which no edge straddles the plane -- or only a handful, the usual case -- costs a few compares per slot instead of a
// round of LDS traffic, and the straddling edges are compacted (ballot + mbcnt, no atomics) so that the move code runs
// ONCE per pass, on the low lanes, instead of once per register slot.
//   idx_in_range (camera.py:80) is evaluated on the depths left by passes 1-2: it is set here from the transformed depth
// and refreshed for exactly the nodes those passes move -- the same values, without a pass over all nodes.
// Appends the group's visible edges to the draw list at segg[5 * nseg ...] and advances nseg (wave-uniform).
template <int K>
__device__ __forceinline__ void cam_group_regs(const KArgs& a, const MapCache<K>& mc, const double* pose, const double* Kc,
                                               const int l0, const int l1, const int ge0, const int nn, const int ne,
                                               double* Px, double* Py, double* Pz, unsigned char* flg, int* list,
                                               int* segg, LdsIntPtr seg_lds, int& nseg, unsigned int& my_layers, const int env,
                                               const int tid) {
  const DevMap& m = a.m;
  const DevCam& cam = a.cam;
  const double max_range = cam.max_range;
#pragma unroll
  for (int k = 0; k < K; k++) {  // camera.py:124-131, 70, 80
    const int i = k * TC_NT + tid;
    if (i < nn) {
      double h[4] = {mc.nd[k].x, mc.nd[k].y, 0.0, 1.0};
      double p[3];
      d_matmul<3, 4, 1>(pose, h, p);
      Px[i] = p[0];
      Py[i] = p[1];
      Pz[i] = p[2];
      flg[i] = (unsigned char)((p[2] < 0 ? 1 : 0) | (p[2] > -max_range ? 2 : 0));
    }
  }
  lds_sync();
  TSTAMP(4);
  int ff[K];  // flags of the two ends of edge slot k: flg[ed.x] | flg[ed.y] << 8 (0 for an empty slot)
#pragma unroll
  for (int k = 0; k < K; k++) ff[k] = k * TC_NT + tid < ne ? (int)flg[mc.ed[k].x] | ((int)flg[mc.ed[k].y] << 8) : 0;
  // camera.py:71-74, 75-77 (plane z = -1e-7, flag idx_front), then 81-83, 84-86 (plane z = -max_range, flag
  // idx_in_range): one copy of the pass, looped
#pragma nounroll
  for (int pass = 0; pass < 4 && !DBG_ON(a.dbg, DBG_SKIP_CLIP); pass++) {
    const int bit = pass < 2 ? 1 : 2;
    const bool target_e0 = (pass & 1) == 0;
    const double tz = pass < 2 ? -0.0000001 : -max_range;
    int sel = 0;  // bit k: slot k straddles the plane in the direction of this pass
#pragma unroll
    for (int k = 0; k < K; k++) {
      const bool fa = ff[k] & bit, fb = (ff[k] >> 8) & bit;
      sel |= (target_e0 ? (!fa && fb) : (fa && !fb)) ? 1 << k : 0;
    }
    if (__ballot(sel != 0) != 0) {
      int n = 0;
#pragma unroll
      for (int k = 0; k < K; k++) {
        const bool s_k = (sel >> k) & 1;
        const unsigned long long mk = __ballot(s_k);
        if (s_k) {
          const int j = n + wave_rank(mk);
          const int t = target_e0 ? mc.ed[k].x : mc.ed[k].y, o = target_e0 ? mc.ed[k].y : mc.ed[k].x;
          list[2 * j] = k * TC_NT + tid;
          list[2 * j + 1] = t | (o << 16);
        }
        n += __popcll(mk);
      }
      lds_sync();
      // the usual case: every straddling edge has a target node of its own -> one move each, order irrelevant
      bool dup = n > TC_NT;
      int mine = 0;
      if (n <= TC_NT) {
        // lane j < n takes list entry j; the target nodes are compared lane against lane through v_readlane (the loop
        // over the list in LDS was one dependent LDS round trip per entry, four passes per frame)
        mine = tid < n ? list[2 * tid + 1] : -1;
        const int my_t = mine & 0xffff;
        for (int j = 0; j < n; j++) {
          const int tj = __builtin_amdgcn_readlane(my_t, j);
          dup |= tid < n && j != tid && tj == my_t;
        }
      }
      if (__ballot(dup) == 0) {
        if (tid < n) {
          const int t = mine & 0xffff, o = (unsigned)mine >> 16;
          cam_move_to_plane(Px, Py, Pz, o, t, tz);
          unsigned int f = flg[t] | (unsigned)bit;
          if (pass < 2) f = (f & ~2u) | (Pz[t] > -max_range ? 2u : 0u);
          flg[t] = (unsigned char)f;
        }
      } else {
        cam_chain_replay(Px, Py, Pz, flg, bit, tz, list, n, max_range, pass < 2, tid);
      }
      lds_sync();
#pragma unroll
      for (int k = 0; k < K; k++) ff[k] = k * TC_NT + tid < ne ? (int)flg[mc.ed[k].x] | ((int)flg[mc.ed[k].y] << 8) : 0;
    }
    if (pass < 3) TSTAMP(17 + pass);
  }
  TSTAMP(5);
  // Only nodes in front AND in range can be "visible" (camera.py:92-93), and an edge is drawn when one of its ends is
  // (camera.py:95) -- with the pixel coordinates of BOTH ends.  So the nodes worth projecting (two f64 divisions each)
  // are the ends of edges that have an end in front and in range: mark them, compact them, project them in one
  // lane-parallel pass.
  int cand = 0;  // bit k: edge slot k has an end in front and in range
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int fa = ff[k] & 0xff, fb = ff[k] >> 8;
    if ((fa & 3) == 3 || (fb & 3) == 3) {  // (lanes sharing a node write the same value: nothing else changes here)
      cand |= 1 << k;
      flg[mc.ed[k].x] = (unsigned char)(fa | 16);
      flg[mc.ed[k].y] = (unsigned char)(fb | 16);
    }
  }
  lds_sync();
  int ncand = 0;
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int i = k * TC_NT + tid;
    const bool c = i < nn && (flg[i] & 16);
    const unsigned long long mk = __ballot(c);
    if (c) list[ncand + wave_rank(mk)] = i;
    ncand += __popcll(mk);
  }
  lds_sync();
  for (int k = tid; k < ncand; k += TC_NT) {  // camera.py:133-142, 90
    const int i = list[k];
    double u, v;
    int2 q = cam_project(Kc, Px[i], Py[i], Pz[i], u, v);
    const bool vis = (u > 0) && (u < cam.W) && (v > 0) && (v < cam.H) && (flg[i] & 3) == 3;
    ((int2*)Px)[i] = q;  // renderer.py:43,50 np.int32(...)
    if (vis) flg[i] |= 4;
  }
  lds_sync();
  TSTAMP(6);
  // camera.py:95: the edges with a visible end, compacted first (edge id, end nodes) so that the code that emits a
  // segment runs once over a dense list instead of once per register slot with a few lanes active in each
  int ndraw = 0;
#pragma unroll
  for (int k = 0; k < K; k++) {
    bool draw = false;
    if ((cand >> k) & 1) draw = ((flg[mc.ed[k].x] | flg[mc.ed[k].y]) & 4) != 0;  // a visible end
    const unsigned long long mk = __ballot(draw);
    if (draw) {
      const int j = ndraw + wave_rank(mk);
      list[2 * j] = k * TC_NT + tid;
      list[2 * j + 1] = mc.ed[k].x | (mc.ed[k].y << 16);
    }
    ndraw += __popcll(mk);
  }
  lds_sync();
  for (int j = tid; j < ndraw; j += TC_NT) {  // both ends were marked above and hold pixel coordinates
    const int e = list[2 * j], xy = list[2 * j + 1];
    const int2 pa = ((int2*)Px)[xy & 0xffff], pb = ((int2*)Px)[(unsigned)xy >> 16];
    int layer = l0;
    for (int c = l0 + 1; c < l1; c++) layer += (ge0 + e) >= m.edge_off[c];
    my_layers |= 1u << layer;
    const int js = nseg + j;  // < seg_cap == total edge count
    if (js < a.seg_lds_cap) {  // the raster stage of this wavefront reads it back from LDS: no store -> load round trip
      const LdsIntPtr o = seg_lds + 5 * js;
      o[0] = layer;
      o[1] = pa.x;
      o[2] = pa.y;
      o[3] = pb.x;
      o[4] = pb.y;
    } else {
      int* o = segg + 5 * js;
      o[0] = layer;
      o[1] = pa.x;
      o[2] = pa.y;
      o[3] = pb.x;
      o[4] = pb.y;
    }
  }
  nseg += ndraw;
  lds_sync();  // the next group reuses the node buffer
}

template <int K>
__device__ inline void cam_fixup_pass(MapCache<K>& mc, const DevMap& m, bool reload, int ge0, int ne, int gn0, int nwin,
                                      double* Px, double* Py, double* Pz, unsigned char* flg, int bit, bool target_e0,
                                      double tz, int* list, int* cnt, int tid) {
  // *cnt was zeroed (and a barrier passed) before the call
  if (!reload) {
    // The group's edges sit in this wavefront's registers (one window).  Typical frame: a handful of edges straddle
    // the plane, each with a target node of its own.  Then no list is needed at all: the lane holding a straddling
    // edge moves that edge's target itself.  Only when two straddling edges share a target does the order of
    // camera.py's loop matter -- detected by every selecting lane writing its edge id into claim[target] and reading
    // it back (one wavefront: LDS operations complete in program order) -- and then the exact list replay below runs.
    int sel = 0;  // bit k: slot k of this lane straddles the plane in the direction of this pass
    int* claim = list;
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int e = k * TC_NT + tid;
      if (e < ne) {
        const int2 ed = mc.ed[k];
        const bool fa = flg[ed.x] & bit, fb = flg[ed.y] & bit;
        if (target_e0 ? (!fa && fb) : (fa && !fb)) {
          sel |= 1 << k;
          claim[target_e0 ? ed.x : ed.y] = e;
        }
      }
    }
    if (__ballot(sel != 0) == 0) return;  // nothing straddles: flags, nodes, list and counter untouched
    bool dup = false;
#pragma unroll
    for (int k = 0; k < K; k++)
      if ((sel >> k) & 1) dup |= claim[target_e0 ? mc.ed[k].x : mc.ed[k].y] != k * TC_NT + tid;
    if (__ballot(dup) == 0) {
#pragma unroll
      for (int k = 0; k < K; k++)
        if ((sel >> k) & 1) {
          const int t = target_e0 ? mc.ed[k].x : mc.ed[k].y, o = target_e0 ? mc.ed[k].y : mc.ed[k].x;
          cam_move_to_plane(Px, Py, Pz, o, t, tz);
          flg[t] |= (unsigned char)bit;  // targets are distinct: no two lanes touch the same byte
        }
      __syncthreads();
      return;
    }
    __syncthreads();  // claim[] aliases the list the exact path is about to build
  }
  for (int w = 0; w < nwin; w++) {
    if (reload) cache_edges(mc, m, ge0 + w * K * TC_NT, ge0 + ne, gn0, tid);
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int e = (w * K + k) * TC_NT + tid;
      if (e < ne) {
        const int2 ed = mc.ed[k];
        const bool fa = flg[ed.x] & bit, fb = flg[ed.y] & bit;
        const bool sel = target_e0 ? (!fa && fb) : (fa && !fb);
        if (sel) {
          const int j = atomicAdd(cnt, 1);
          list[2 * j] = e;
          list[2 * j + 1] = target_e0 ? (ed.x | (ed.y << 16)) : (ed.y | (ed.x << 16));
        }
      }
    }
  }
  __syncthreads();
  cam_chain_replay(Px, Py, Pz, flg, bit, tz, list, *cnt, 0.0, false, tid);
  __syncthreads();
}

// A spawn node must exist and have an out-edge (map.py:62-64 re-draws otherwise; the host RNG mirror
// does that).  Anything else would index out of bounds, so it is replaced and flagged.
__device__ inline int checked_spawn(const DevMap& m, int node, int& status) {
  if ((unsigned)node < (unsigned)m.lpN && m.lp_fat[node].nnext > 0) return node;
  status |= TC_S_BAD_SPAWN;
  return m.first_spawnable;
}

__device__ inline unsigned int spread4(unsigned int x) {  // x < 16: 4 bits -> 4 bytes of 0x00/0xFF
  const unsigned int m = (x * 0x00204081u) & 0x01010101u;  // 24-bit multiply: full rate
  return (m << 8) - m;                                     // * 255 without v_mul_lo_u32
}

// camera.py:133-142 for one node + the np.int32 cast of renderer.py:43,50
__device__ inline int2 cam_project(const double* K, double X, double Y, double Z, double& u, double& v) {
  double P3[3] = {X, Y, Z};
  double h[3];
  d_matmul<3, 3, 1>(K, P3, h);
  u = h[0] / h[2];
  v = h[1] / h[2];
  return make_int2(d_np_int32(u), d_np_int32(v));
}

// Inclusive prefix sum over the 64 lanes with DPP moves (register to register): Hillis-Steele inside each row of 16
// (row_shr 1, 2, 4, 8, zeros shifted in), then lane 15 of a row added to the next row and lane 31 to the upper half
// (row_bcast 15 / 31) -- instead of six ds_bpermute round trips through the LDS crossbar.  All 64 lanes must be active.
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
  (void)lane;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2 and 3
  return v;
}

// Caller's buffers -> LiveLds, one item per lane (one vector-memory round trip for the ~30 scalars of an env).
__device__ __forceinline__ void live_in(const KArgs& a, unsigned char* smem, int env, int mode, unsigned int flags,
                                        const int tid = threadIdx.x) {
  const tc_buffers& b = a.b;
  LiveLds* lv = (LiveLds*)(smem + a.lds.off_live);
  if (tid < 8) {
    const double* p = b.x;
    p = tid == 1 ? b.y : p;
    p = tid == 2 ? b.theta : p;
    p = tid == 3 ? b.velocity : p;
    p = tid == 4 ? b.steering : p;
    p = tid == 5 ? b.radius : p;
    p = tid == 6 ? b.front_x : p;
    p = tid == 7 ? b.front_y : p;
    lv->d[tid] = p[env];
  } else if (tid < 16) {
    lv->lp[tid - 8] = b.local_path[env * 8 + (tid - 8)];
  } else if (tid == 16) {
    lv->lp_len = b.lp_len[env];
  } else if (tid == 17) {
    lv->last_maneuver = b.last_maneuver[env];
  } else if (tid == 18) {
    lv->needs_reset = (mode == MODE_STEP && (flags & TC_F_AUTORESET)) ? b.needs_reset[env] : 0;
  } else if (tid == 19) {
    const int c = (mode == MODE_STEP && (flags & TC_F_AUTORESET)) ? b.spawn_cursor[env] : 0;
    lv->cursor = c;
    lv->cursor0 = c;
  } else if (tid < 28) {
    const int t = tid - 20;
    lv->cnt[t] = (a.n_terms > 0 && a.term_counters) ? a.term_counters[(size_t)env * TC_MAX_TERMS + t] : 0;
  } else if (tid == 28) {
    lv->d[8] = 0;
    lv->d[9] = 0;
    lv->have_trig = 0;
  }
  lds_sync();
}

// LiveLds -> caller's buffers after the last step of the launch (not for MODE_RENDER, which changes nothing).
__device__ __forceinline__ void live_out(const KArgs& a, unsigned char* smem, int env, const int tid = threadIdx.x) {
  const tc_buffers& b = a.b;
  const LiveLds* lv = (const LiveLds*)(smem + a.lds.off_live);
  const int C = a.m.C;
  lds_sync();
  if (tid < 8) {
    double* p = b.x;
    p = tid == 1 ? b.y : p;
    p = tid == 2 ? b.theta : p;
    p = tid == 3 ? b.velocity : p;
    p = tid == 4 ? b.steering : p;
    p = tid == 5 ? b.radius : p;
    p = tid == 6 ? b.front_x : p;
    p = tid == 7 ? b.front_y : p;
    p[env] = lv->d[tid];
  } else if (tid < 16) {
    b.local_path[env * 8 + (tid - 8)] = lv->lp[tid - 8];
  } else if (tid == 16) {
    b.lp_len[env] = lv->lp_len;
  } else if (tid == 17) {
    b.last_maneuver[env] = lv->last_maneuver;
  } else if (tid == 18) {
    if (b.needs_reset) b.needs_reset[env] = (unsigned char)lv->needs_reset;
  } else if (tid == 19) {
    if (lv->cursor != lv->cursor0) b.spawn_cursor[env] = lv->cursor;
  } else if (tid < 28) {
    const int t = tid - 20;
    if (a.term_counters && t < a.n_terms) a.term_counters[(size_t)env * TC_MAX_TERMS + t] = lv->cnt[t];
  } else if (tid == 28) {
    b.cte[env] = lv->cte;
  } else if (tid == 29) {
    b.heading_error[env] = lv->he;
  } else if (tid == 30) {
    b.reward[env] = lv->reward;
  } else if (tid == 31) {
    b.terminated[env] = (unsigned char)lv->terminated;
  } else if (tid == 32) {
    b.truncated[env] = (unsigned char)lv->trunc;
  } else if (tid == 33) {
    b.status[env] = lv->status;
  } else if (tid < 34 + TC_MAX_LAYERS - 4) {  // lanes 34..45: layers 0..11
    const int l = tid - 34;
    if (l < C) {
      b.laneline_distances[(size_t)env * C + l] = lv->dist[l];
      b.nearest_edge[(size_t)env * C + l] = lv->ne[l];
    }
  } else if (tid < 50) {  // lanes 46..49: layers 12..15
    const int l = tid - 34;
    if (l < C) {
      b.laneline_distances[(size_t)env * C + l] = lv->dist[l];
      b.nearest_edge[(size_t)env * C + l] = lv->ne[l];
    }
  }
}

#ifndef TC_MIN_WAVES
#define TC_MIN_WAVES 4
#endif
// Stage 1 (simulate) for one env by one wavefront.  Returns false when nothing is to be rasterised for this env.
// always inlined: out of line, `a` would be a pointer into private memory and the whole kernel-argument struct
// (~1 KB) would be copied to scratch by every lane (the inliner's cost threshold is close: measured 47.9 k vs 47.6 k)
// One step of one env: reads and rewrites the env's LiveLds record; the caller's buffers are only touched by live_in /
// live_out (and the per-step rollout rows of `roll`).
// what the camera needs of an env's state: rear-axle position and cos / sin of MINUS the heading (car.py:159-165)
struct FramePose {
  double x, y, cth, sth;
};

template <int K>
__device__ __forceinline__ void sim_body(const KArgs& a, unsigned char* smem, int env, int mode,
                                const void* car_control, int cdtype,
                                const int* maneuver, const int* spawn_nodes, const unsigned char* mask,
                                unsigned int flags, const RollStep& roll, const int tid, MapCache<K>& mc, FramePose& fp) {

  TSTAMP(0);
  TSTAMP_REAL(30);
  const DevMap& m = a.m;
  const tc_buffers& b = a.b;
  // map windows of 64*K nodes / edges; one window (the usual case) is fetched now and kept in registers
  const int nwin_n = (m.total_nodes + TC_NT * K - 1) / (TC_NT * K), nwin_e = (m.total_edges + TC_NT * K - 1) / (TC_NT * K);
  const bool single = a.n_grp == 1 && !a.cam_nodes && a.grp_layer[0] == 0 && a.grp_layer[1] == m.C && nwin_n <= 1 && nwin_e <= 1;
  double* dn = (double*)(smem + a.lds.off_p);  // phase B: one double per lane-line node (aliases the camera's node buffer)

  // ---- state: all lanes read the same LDS words (broadcast)
  LiveLds* lv = (LiveLds*)(smem + a.lds.off_live);
  // (readfirstlane: the values are wave-uniform, and the compiler has to KNOW it -- read straight from LDS they count
  // as divergent, every branch of the scalar phases below becomes an exec-mask region and the kernel triples in size)
  CarState s;
  s.x = uni_d(lv->d[0]);
  s.y = uni_d(lv->d[1]);
  s.theta = uni_d(lv->d[2]);
  s.velocity = uni_d(lv->d[3]);
  s.steering = uni_d(lv->d[4]);
  s.radius = uni_d(lv->d[5]);
  s.front_x = uni_d(lv->d[6]);
  s.front_y = uni_d(lv->d[7]);
  s.cth = uni_d(lv->d[8]);
  s.sth = uni_d(lv->d[9]);
#pragma unroll
  for (int i = 0; i < 8; i++) s.lp[i] = uni_i(lv->lp[i]);
  s.lp_len = uni_i(lv->lp_len);
  s.last_maneuver = uni_i(lv->last_maneuver);
  bool have_trig = uni_i(lv->have_trig) != 0;  // s.cth / s.sth hold cos / sin of the current heading
  const int nr = uni_i(lv->needs_reset);
  int cursor = uni_i(lv->cursor);
  int my_cnt = tid < TC_MAX_TERMS ? lv->cnt[tid] : 0;  // lane t: steps_true of term slot t

  int status = 0, trunc = 0;
  PathInfo pinfo;
  pinfo.ax = pinfo.ay = pinfo.bx = pinfo.by = pinfo.ori = 0;
  pinfo.valid = 0;
  bool fresh = false;  // env was (re)spawned in this launch: info is empty (car.py:47-51)
  if (mode == MODE_RESET) {
    d_reset(m, a.car, s, checked_spawn(m, spawn_nodes[env], status));
    fresh = true;
    have_trig = true;
  } else if (mode == MODE_STEP) {
    if ((flags & TC_F_AUTORESET) && nr) {
      const int cur = cursor;
      int node;
      if ((flags & TC_F_DEVICE_SPAWN) && a.spawn_n > 0) {
        node = a.spawn_tab[tc_spawn_index(a.spawn_seed, (uint32_t)env, (uint32_t)cur, (uint32_t)a.spawn_n)];
      } else {
        if ((unsigned)cur >= (unsigned)b.spawn_queue_len) status |= TC_S_SPAWN_WRAPPED;  // replaying the queue
        node = b.spawn_queue[(size_t)env * b.spawn_queue_len + ((unsigned)cur % (unsigned)b.spawn_queue_len)];
      }
      d_reset(m, a.car, s, checked_spawn(m, node, status));
      fresh = true;
      have_trig = true;
      cursor = cur + 1;
    } else if ((unsigned)s.lp[0] >= (unsigned)m.lpN || (unsigned)s.lp[1] >= (unsigned)m.lpN) {
      status |= TC_S_NOT_RESET;  // stepping an env that was never reset: no valid lanepath edge to index with
      trunc = 1;
    } else {
      double v, st;
      if (cdtype == TC_F32) {
        v = (double)((const float*)car_control)[2 * env];
        st = (double)((const float*)car_control)[2 * env + 1];
      } else {
        v = ((const double*)car_control)[2 * env];
        st = ((const double*)car_control)[2 * env + 1];
      }
      v = d_np_clip(v, -1.0, 1.0);  // env.py:118
      st = d_np_clip(st, -1.0, 1.0);
      const int man = maneuver[env];
      d_car_kinematics(a.car, s, v, st, have_trig);
      TSTAMP(23);
      const FatGlobal fg = {m.lp_fat, m.lp_nodes};
      trunc = d_find_local_path(m, fg, s, man, status, pinfo, tid);
      have_trig = true;
    }
  }

  TSTAMP(1);
  if (mode != MODE_RENDER) {
    // ---- write the state back + scalar info (lane 0)
    const bool have_info = !fresh && s.lp_len >= 2;
    double cte = 0, he = 0;
    if (have_info && !pinfo.valid) {  // path kept from an earlier step (truncated before it was rebuilt)
      double2 n1 = m.lp_nodes[s.lp[2]], n2 = m.lp_nodes[s.lp[3]];
      pinfo.ax = n1.x;
      pinfo.ay = n1.y;
      pinfo.bx = n2.x;
      pinfo.by = n2.y;
      pinfo.ori = d_lp_edge_ori(m, s.lp[2], s.lp[3]);
    }
    if (have_info) {  // car.py:52-53 (nodes and orientation of local_path[1] were captured while tracking)
      cte = d_distance_to_edge(pinfo.ax, pinfo.ay, pinfo.bx, pinfo.by, s.front_x, s.front_y);
      he = d_clip_angle(pinfo.ori - s.theta);
    }
    double reward = 0;
    int terminated = 0;
    if (!(flags & TC_F_WRAPPED) && !fresh) {  // env.py:93,99
      double r = (-1 / a.car.track_width) * cte + 1;
      reward = (0 > r) ? 0 : r;
      terminated = cte > (a.car.track_width * 10);
    }
    const bool late = a.n_terms > 0;  // reward / terminated depend on phase B: written after it
    if (tid == 0) {
      lv->d[0] = s.x;
      lv->d[1] = s.y;
      lv->d[2] = s.theta;
      lv->d[3] = s.velocity;
      lv->d[4] = s.steering;
      lv->d[5] = s.radius;
      lv->d[6] = s.front_x;
      lv->d[7] = s.front_y;
      lv->d[8] = s.cth;
      lv->d[9] = s.sth;
      lv->lp_len = s.lp_len;
      lv->last_maneuver = s.last_maneuver;
      lv->have_trig = have_trig ? 1 : 0;
      lv->cursor = cursor;
      lv->cte = cte;
      lv->he = he;
      lv->trunc = trunc;
      lv->status = status;
      if (roll.cte) roll.cte[roll.row0 + env] = cte;
      if (roll.heading_error) roll.heading_error[roll.row0 + env] = he;
      if (roll.truncated) roll.truncated[roll.row0 + env] = (unsigned char)trunc;
      if (roll.status) roll.status[roll.row0 + env] = status;
      if (roll.x) roll.x[roll.row0 + env] = s.x;
      if (roll.y) roll.y[roll.row0 + env] = s.y;
      if (roll.theta) roll.theta[roll.row0 + env] = s.theta;
      if (roll.velocity) roll.velocity[roll.row0 + env] = s.velocity;
      if (roll.lp_len) roll.lp_len[roll.row0 + env] = s.lp_len;
      if (!late) {
        lv->reward = reward;
        lv->terminated = terminated;
        lv->needs_reset = (flags & TC_F_AUTORESET) ? (terminated || trunc) : 0;
        if (roll.reward) roll.reward[roll.row0 + env] = reward;
        if (roll.terminated) roll.terminated[roll.row0 + env] = (unsigned char)terminated;
      }
    }
    if (tid < 8) {  // register-resident select (a runtime-indexed s.lp[tid] would live in scratch)
      int v = s.lp[0];
#pragma unroll
      for (int i = 1; i < 8; i++) v = tid == i ? s.lp[i] : v;
      lv->lp[tid] = v;
      if (roll.lp) roll.lp[(roll.row0 + env) * 8 + tid] = v;
    }

    TSTAMP(2);
    // The map window is fetched here, behind phase A, and not at the top of the step: across the scalar phase its 30
    // registers per lane were what pushed the fused kernel over 128 VGPRs (the allocator answered by spilling the node
    // half to scratch and reloading it here).  The lines are L1 / L2 resident after the first step of a launch.
    if (single) {  // (the camera stage behind this one finds the window loaded)
      cache_nodes(mc, m, 0, m.total_nodes, tid);
      cache_edges(mc, m, 0, m.total_edges, 0, tid);
    }
    // ---- phase B: lane-line distances (car.py:55-64)
    const int C = m.C;
    double dist_l = 0;  // lane l < C: distance to lane-line layer l (0 while the info is empty, car.py:47-51)
    if (have_info && !DBG_ON(flags, DBG_SKIP_DIST)) {
      int my_e = -1;
      const int cell = uni_i(d_grid_cell(m, s.x, s.y));  // (every lane holds the same state)
      if (cell >= 0) {
        // layer.py:43 over the cell's candidate edges (DevMap: every edge that can be the first minimum for a point of
        // the cell, all layers, ascending): one lane per candidate, two distances straight from the edge record
        const int o0 = m.cand_off[cell * C], o1 = m.cand_off[cell * C + C];
        TSTAMP(14);
        for (int l0 = 0; l0 < C; l0 += TC_AG) {
          double bd[TC_AG];
          int best[TC_AG], lo[TC_AG], hi[TC_AG];
#pragma unroll
          for (int g = 0; g < TC_AG; g++) {
            bd[g] = 0;
            best[g] = -1;
            lo[g] = l0 + g < C ? m.edge_off[l0 + g] : 0x7fffffff;
            hi[g] = l0 + g < C ? m.edge_off[l0 + g + 1] : 0x7fffffff;
          }
          const int e_lo = m.edge_off[l0], e_hi = m.edge_off[l0 + TC_AG < C ? l0 + TC_AG : C];
          for (int i = o0 + tid; i < o1; i += TC_NT) {
            const int e = m.cand_idx[i];
            if (e >= e_lo && e < e_hi) {
              const double4 q = m.edge_xy[e];
              const double d = tc_fabs(d_dist(s.x, s.y, q.x, q.y) + d_dist(s.x, s.y, q.z, q.w));
#pragma unroll
              for (int g = 0; g < TC_AG; g++) {
                const bool take = e >= lo[g] && e < hi[g] && (best[g] < 0 || d < bd[g]);
                bd[g] = take ? d : bd[g];
                best[g] = take ? e - lo[g] : best[g];
              }
            }
          }
          wave_argmin_group(bd, best);
#pragma unroll
          for (int g = 0; g < TC_AG; g++)
            if (tid == l0 + g) my_e = best[g];
        }
      } else {
      for (int w = 0; w < nwin_n; w++) {
        if (!single) cache_nodes(mc, m, w * K * TC_NT, m.total_nodes, tid);
#pragma unroll
        for (int k = 0; k < K; k++) {
          const int i = (w * K + k) * TC_NT + tid;
          if (i < m.total_nodes) dn[i] = d_dist(s.x, s.y, mc.nd[k].x, mc.nd[k].y);
        }
      }
      lds_sync();
      TSTAMP(14);
      // layer.py:43 for every layer: each lane scans its edges once (ascending index) and keeps one partial per layer
      // of the current group of TC_AG layers; the group's reductions then run together (wave_argmin_group).  A layer's
      // edges are contiguous, so edge e belongs to group layer g iff edge_off[l0 + g] <= e < edge_off[l0 + g + 1].
      for (int l0 = 0; l0 < C; l0 += TC_AG) {
        double bd[TC_AG];
        int best[TC_AG], lo[TC_AG], hi[TC_AG];
#pragma unroll
        for (int g = 0; g < TC_AG; g++) {
          bd[g] = 0;
          best[g] = -1;
          lo[g] = l0 + g < C ? m.edge_off[l0 + g] : 0x7fffffff;
          hi[g] = l0 + g < C ? m.edge_off[l0 + g + 1] : 0x7fffffff;
        }
        const int e_lo = m.edge_off[l0], e_hi = m.edge_off[l0 + TC_AG < C ? l0 + TC_AG : C];
        for (int w = 0; w < nwin_e; w++) {
          const int w0 = w * K * TC_NT;
          if (w0 >= e_hi || w0 + K * TC_NT <= e_lo) continue;  // no edge of this layer group in the window
          if (!single) cache_edges(mc, m, w0, m.total_edges, 0, tid);
#pragma unroll
          for (int k = 0; k < K; k++) {
            const int e = (w * K + k) * TC_NT + tid;
            if (e >= e_lo && e < e_hi) {
              const double d = tc_fabs(dn[mc.ed[k].x] + dn[mc.ed[k].y]);
#pragma unroll
              for (int g = 0; g < TC_AG; g++) {
                const bool take = e >= lo[g] && e < hi[g] && (best[g] < 0 || d < bd[g]);
                bd[g] = take ? d : bd[g];
                best[g] = take ? e - lo[g] : best[g];
              }
            }
          }
        }
        wave_argmin_group(bd, best);
#pragma unroll
        for (int g = 0; g < TC_AG; g++)
          if (tid == l0 + g) my_e = best[g];
      }
      }
      TSTAMP(15);
      if (tid < C) {
        const int l = tid;
        if (my_e >= 0) {
          const int ge = m.edge_off[l] + my_e;
          int2 ed = m.edges_g[ge];
          double2 n0 = m.nodes[ed.x], n1 = m.nodes[ed.y];
          // layer.py:126-142 by the sign of two dot products; the literal atan2 form only for the lanes (if any) whose
          // angle is within 1e-6 rad of the pi/2 threshold
          bool certain;
          bool inb = d_within_bounds_filter(n0.x, n0.y, n1.x, n1.y, s.x, s.y, certain);
          if (!certain) inb = d_within_bounds(n0.x, n0.y, n1.x, n1.y, m.ori_fwd[ge], m.ori_rev[ge], s.x, s.y);
          if (inb) {
            dist_l = tc_fabs(d_distance_to_edge(n0.x, n0.y, n1.x, n1.y, s.x, s.y));
          } else {
            double da = d_dist(s.x, s.y, n0.x, n0.y);
            double db = d_dist(s.front_x, s.front_y, n1.x, n1.y);  // FRONT axle for n1 (car.py:64)
            dist_l = db < da ? db : da;
          }
        }
        lv->dist[l] = dist_l;
        lv->ne[l] = my_e;
        if (roll.dist) roll.dist[(roll.row0 + env) * C + l] = dist_l;
        if (roll.ne) roll.ne[(roll.row0 + env) * C + l] = my_e;
      }
      TSTAMP(16);
      lds_sync();  // dn aliases the camera's node buffer
    } else if (tid < C) {
      lv->dist[tid] = 0;
      lv->ne[tid] = -1;
      if (roll.dist) roll.dist[(roll.row0 + env) * C + tid] = 0;
      if (roll.ne) roll.ne[(roll.row0 + env) * C + tid] = -1;
    }
    if (late) {
      // a re-spawned env did not go through Wrapper.step (the reference's reset() bypasses the wrappers)
      if (!fresh) {
        d_apply_terms(a.terms, a.n_terms, my_cnt, a.car.track_width, tid, C, cte, have_info ? s.velocity : 0.0, dist_l,
                      reward, terminated);
        if (tid < TC_MAX_TERMS) lv->cnt[tid] = my_cnt;
      }
      if (tid == 0) {
        lv->reward = reward;
        lv->terminated = terminated;
        lv->needs_reset = (flags & TC_F_AUTORESET) ? (terminated || trunc) : 0;
        if (roll.reward) roll.reward[roll.row0 + env] = reward;
        if (roll.terminated) roll.terminated[roll.row0 + env] = (unsigned char)terminated;
      }
    }
  }

  TSTAMP(3);
  TSTAMP_REAL(29);
  // what the camera stage needs.  car.py:159-165 takes cos(-theta), sin(-theta): tc_cos is exactly even and tc_sin
  // exactly odd (their kernels are built from x*x and x*y terms only), so the values of the front-axle update are
  // reused bit for bit.
  fp.x = s.x;
  fp.y = s.y;
  if (have_trig) {
    fp.cth = s.cth;
    fp.sth = -s.sth;
  } else {
    fp.cth = tc_cos(-s.theta);
    fp.sth = tc_sin(-s.theta);
  }
}

// Phase C, the camera (camera.py:52-110) of one frame: transform -> 4 clip passes -> project -> visibility -> draw list
// (row `seg_row` of the per-env lists).  Depends on the env's state only through `fp`, so a frame can be produced by the
// wavefront that simulated the step or by any other one later.  mc_loaded: `mc` already holds the whole map's window
// (the simulate stage of the same wavefront loaded it).
// camera.py:62 `pose = E @ car3d` of one frame: the 3x4 matrix every node of the map is transformed with.  It depends on
// the env's state only through fp (and on the env's camera): the simulate stage computes it once per (step, env) and
// hands it over in the pose row, so the 64 lanes of a frame wavefront do not each redo the two matrix products.
#define TC_POSE_ROW 16  // doubles per (step, env) pose row: the 12 entries, padded to 128 bytes
// "not written yet" in a pose row of a streamed call (a NaN no arithmetic produces: all ones); and the marker a frame
// workgroup that gave up waiting leaves in place of its draw-list length
#define TC_POSE_EMPTY 0xFFFFFFFFFFFFFFFFull
#define TC_FRAME_SKIPPED (-2)
__device__ __forceinline__ void cam_pose12(const KArgs& a, int env, const FramePose& fp, double* pose) {
  double Ec[12];  // this env's camera (camera.py:23-24,48-50): shared, or its own after tc_env_set_camera_per_env
  if (a.cam_E) {
#pragma unroll
    for (int i = 0; i < 12; i++) Ec[i] = a.cam_E[(size_t)env * 12 + i];
  } else {
    const __attribute__((address_space(4))) double* ee = (const __attribute__((address_space(4))) double*)(unsigned long long)a.cam.E;
#pragma unroll
    for (int i = 0; i < 12; i++) Ec[i] = ee[i];
  }
  const double cth = fp.cth, sth = fp.sth;
  double R[16] = {cth, -sth, 0, 0, sth, cth, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  double Tm[16] = {1, 0, 0, -fp.x, 0, 1, 0, -fp.y, 0, 0, 1, 0, 0, 0, 0, 1};
  double car3d[16];
  d_matmul<4, 4, 4>(R, Tm, car3d);
  d_matmul<3, 4, 4>(Ec, car3d, pose);  // camera.py:62
}

template <int K>
__device__ __forceinline__ void cam_body(const KArgs& a, unsigned char* smem, int env, const double* pose_in, MapCache<K>& mc,
                                         const bool mc_loaded, const int tid, const int seg_row, int& nseg_out,
                                         unsigned int& used_out, const bool store_n = true) {
  unsigned int my_layers = 0;  // layers this lane put a segment into the draw list for
  const DevMap& m = a.m;
  const int nwin_n = (m.total_nodes + TC_NT * K - 1) / (TC_NT * K), nwin_e = (m.total_edges + TC_NT * K - 1) / (TC_NT * K);
  const bool single = a.n_grp == 1 && !a.cam_nodes && a.grp_layer[0] == 0 && a.grp_layer[1] == m.C && nwin_n <= 1 && nwin_e <= 1;
  if (single && !mc_loaded) {
    cache_nodes(mc, m, 0, m.total_nodes, tid);
    cache_edges(mc, m, 0, m.total_edges, 0, tid);
  }
  double* Px = (double*)(smem + a.lds.off_p);
  double* Py = Px + a.cap_nodes;
  double* Pz = Py + a.cap_nodes;
  unsigned char* flg = smem + a.lds.off_flg;
  int* list = (int*)(smem + a.lds.off_list);
  int* cnt = (int*)(smem + a.lds.off_cnt);
  const DevCam& cam = a.cam;
  double pose[12], Kc[9];
  if (a.cam_K) {  // (a branch, not a select of pointers: the shared camera's K then comes in scalar loads)
#pragma unroll
    for (int i = 0; i < 9; i++) Kc[i] = a.cam_K[(size_t)env * 9 + i];
  } else {
    const __attribute__((address_space(4))) double* kk = (const __attribute__((address_space(4))) double*)(unsigned long long)cam.K;
#pragma unroll
    for (int i = 0; i < 9; i++) Kc[i] = kk[i];  // (explicitly the kernarg segment: cannot be merged with the other branch)
  }
  // the 12 entries are the same in every lane: kept in scalar registers through the node loop (24 VGPRs less)
#pragma unroll
  for (int i = 0; i < 12; i++) pose[i] = uni_d(pose_in[i]);
  // The lane-line layers of a camera group are processed together: node ids are made global (edges_g) and then
  // relative to the group, so each of the passes below is ONE loop over the group's nodes / edges instead of one
  // per layer (the layers never share nodes, so camera.py's per-layer loop and this are the same computation).
  // Small maps are a single group; larger ones are split by the host so that the node buffer (and with it the
  // number of workgroups a CU can hold) is sized by the largest group instead of the whole map.
  // counters: cnt[0..3] fix-up passes, cnt[4] projection candidates, cnt[5] draw list
  int* seg_cnt = cnt + 5;
  const size_t seg_slot = (size_t)seg_row * a.N + env;
  int* segg = a.seg_g + seg_slot * a.seg_cap * 5;  // [seg_cap][5]: layer, x0, y0, x1, y1
  int nseg = 0;  // draw-list length so far (wave-uniform)
  for (int g = 0; g < a.n_grp; g++) {
    const int l0 = a.grp_l0[g], l1 = a.grp_l1[g];
    const int gn0 = a.grp_n0[g], ge0 = a.grp_e0[g];
    const int nn = a.grp_n0[g + 1] - gn0, ne = a.grp_e0[g + 1] - ge0;
    const int nwn = (nn + TC_NT * K - 1) / (TC_NT * K), nwe = (ne + TC_NT * K - 1) / (TC_NT * K);
    const bool one = nwn <= 1 && nwe <= 1;  // the group fits the register cache: loaded once, serves every pass
    const bool reload = !one;
    if (one && !single) {
      if (a.cam_nodes) {  // (component groups: always `one`, tc_env_create sees to it)
        cache_nodes_from(mc, a.cam_nodes, gn0, gn0 + nn, tid);
        cache_edges_from(mc, a.cam_edges_g, ge0, ge0 + ne, gn0, tid);
      } else {
        cache_nodes(mc, m, gn0, gn0 + nn, tid);
        cache_edges(mc, m, ge0, ge0 + ne, gn0, tid);
      }
    }
    if (one) {
      cam_group_regs<K>(a, mc, pose, Kc, l0, l1, ge0, nn, ne, Px, Py, Pz, flg, list, segg, (LdsIntPtr)(smem + a.seg_lds_off), nseg,
                        my_layers, env, tid);
      continue;
    }
    // a group larger than the register window (no bundled map): window by window, lists and counters in LDS
    if (tid < 5) cnt[tid] = 0;
    if (tid == 5) cnt[5] = nseg;
    for (int w = 0; w < nwn; w++) {  // camera.py:124-131
      if (reload) cache_nodes(mc, m, gn0 + w * K * TC_NT, gn0 + nn, tid);
#pragma unroll
      for (int k = 0; k < K; k++) {
        const int i = (w * K + k) * TC_NT + tid;
        if (i < nn) {
          double h[4] = {mc.nd[k].x, mc.nd[k].y, 0.0, 1.0};
          double p[3];
          d_matmul<3, 4, 1>(pose, h, p);
          Px[i] = p[0];
          Py[i] = p[1];
          Pz[i] = p[2];
          flg[i] = p[2] < 0 ? 1 : 0;  // camera.py:70
        }
      }
    }
    __syncthreads();
    TSTAMP(4);
    // camera.py:71-74, 75-77 (plane z = -1e-7, flag idx_front), then 81-83, 84-86 (plane z = -max_range, flag
    // idx_in_range): ONE inlined copy of the pass, looped -- four copies were ~4 k instructions of the kernel
#pragma nounroll
    for (int pass = 0; pass < 4 && !DBG_ON(a.dbg, DBG_SKIP_CLIP); pass++) {
      if (pass == 2) {
        for (int i = tid; i < nn; i += TC_NT)
          if (Pz[i] > -cam.max_range) flg[i] |= 2;  // camera.py:80, on the mutated depths
        __syncthreads();
      }
      cam_fixup_pass(mc, m, reload, ge0, ne, gn0, nwe, Px, Py, Pz, flg, pass < 2 ? 1 : 2, (pass & 1) == 0,
                     pass < 2 ? -0.0000001 : -cam.max_range, list, cnt + pass, tid);
      if (pass < 3) TSTAMP(17 + pass);
    }
    TSTAMP(5);
    // Only nodes in front AND in range can be "visible" (camera.py:92-93), and an edge is drawn when one of its ends is
    // (camera.py:95) -- with the pixel coordinates of BOTH ends.  So the nodes worth projecting (two f64 divisions each)
    // are the ends of edges that have an end in front and in range: mark them, compact them, project them in one
    // lane-parallel pass.  (Projecting the far ends inside the draw-list loop instead made the whole wavefront run the
    // projection code in every edge slot in which a single lane needed it.)
    for (int w = 0; w < nwe; w++) {
      if (reload) cache_edges(mc, m, ge0 + w * K * TC_NT, ge0 + ne, gn0, tid);
#pragma unroll
      for (int k = 0; k < K; k++) {
        const int e = (w * K + k) * TC_NT + tid;
        if (e < ne) {
          const int2 ed = mc.ed[k];
          const int fa = flg[ed.x], fb = flg[ed.y];
          if ((fa & 3) == 3 || (fb & 3) == 3) {  // (lanes sharing a node write the same value: nothing else changes here)
            flg[ed.x] = (unsigned char)(fa | 16);
            flg[ed.y] = (unsigned char)(fb | 16);
          }
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < nn; i += TC_NT)
      if (flg[i] & 16) list[atomicAdd(cnt + 4, 1)] = i;
    __syncthreads();
    const int ncand = cnt[4];
    for (int k = tid; k < ncand; k += TC_NT) {  // camera.py:133-142, 90
      const int i = list[k];
      double u, v;
      int2 q = cam_project(Kc, Px[i], Py[i], Pz[i], u, v);
      const bool vis = (u > 0) && (u < cam.W) && (v > 0) && (v < cam.H) && (flg[i] & 3) == 3;
      ((int2*)Px)[i] = q;  // renderer.py:43,50 np.int32(...)
      if (vis) flg[i] |= 4;
    }
    __syncthreads();
    TSTAMP(6);
    for (int w = 0; w < nwe; w++) {  // camera.py:95
      if (reload) cache_edges(mc, m, ge0 + w * K * TC_NT, ge0 + ne, gn0, tid);
#pragma unroll
      for (int k = 0; k < K; k++) {
        const int e = (w * K + k) * TC_NT + tid;
        if (e < ne) {
          const int2 ed = mc.ed[k];
          if ((flg[ed.x] | flg[ed.y]) & 4) {  // a visible end: both ends were marked above and hold pixel coordinates
            const int2 pa = ((int2*)Px)[ed.x], pb = ((int2*)Px)[ed.y];
            int layer = l0;
            for (int c = l0 + 1; c < l1; c++) layer += (ge0 + e) >= m.edge_off[c];
            my_layers |= 1u << layer;
            int j = atomicAdd(seg_cnt, 1);
            int* o = segg + 5 * j;  // j < seg_cap == total edge count
            o[0] = layer;
            o[1] = pa.x;
            o[2] = pa.y;
            o[3] = pb.x;
            o[4] = pb.y;
          }
        }
      }
    }
    __syncthreads();  // the next group reuses the node buffer and the counters
    nseg = uni_i(*seg_cnt);
    __syncthreads();
  }
  TSTAMP(7);
  nseg_out = nseg;
  // The list length goes to memory for a raster launch of its own.  When the SAME wavefront rasterises the frame it gets
  // the length in a register, and must not have this store in flight: the raster stage's first s_waitcnt vmcnt(0) (the
  // compiler guards the registers of its draw-list loads with one) would sit out the store's whole round trip behind
  // the other wavefronts' frame stores -- measured with the phase clock: 12.9 k of a frame's 83 k clocks.  Those kernels
  // store the length for tc_env_draw_list_stats after their last frame store instead.
  if (store_n && tid == 0) a.seg_n[seg_slot] = nseg_out;
  used_out = 0;
  for (int c = 0; c < m.C; c++)
    if (__ballot((my_layers >> c) & 1u)) used_out |= 1u << c;
}

// Step k of a launch reads row k of the action arrays and writes row k of the rollout arrays ([nsteps][N] each).
struct MultiArgs {
  int nsteps;
  int seg_rows;  // > 1: step k writes its draw list to row k of seg_g / seg_n ([seg_rows][N] lists) for a raster launch
                 // over all (step, env) frames behind this kernel; 1: row 0 every step (same wavefront rasterises it)
  int cam_here;  // tc_env_kernel: 1 = the camera stage (draw list) runs in this kernel, 0 = not (no observation wanted,
                 // or tc_frame_kernel produces the frames from pose_rows)
  double* pose_rows;  // [nsteps][N][TC_POSE_ROW] camera.py:62 pose matrix of every (step, env) for tc_frame_kernel, or NULL
  unsigned int* resident;  // streamed call (see launch()): every workgroup counts itself in here when it starts, and the
                           // pose rows are written with device-scope stores (a frame kernel that runs BESIDE this launch
                           // polls them); NULL otherwise
  int map_lds;        // tc_envg_kernel: 1 = the lane-line edge records (end points + orientations, 48 bytes per edge) are
                      // copied into the workgroup's LDS at kernel start and phase B reads them from there
  int fat_lds;        // tc_envg_kernel: 1 = the lanepath's fat node records (96 bytes per node) sit in LDS too, behind the
                      // edge records (or at offset 0 without them), and lanepath tracking reads them there
  tc_rollout roll;
};
// (base pointers and the row: the address of a row is formed where it is stored, under its null test -- forming all 15 row
// pointers at the top of every step was ~60 scalar instructions per step of every wavefront, 4 % of cfg2)
__device__ __forceinline__ RollStep roll_at(const tc_rollout& r, size_t row0, int C) {
  RollStep q;
  q.reward = r.reward;
  q.cte = r.cte;
  q.heading_error = r.heading_error;
  q.terminated = r.terminated;
  q.truncated = r.truncated;
  q.status = r.status;
  q.x = r.x;
  q.y = r.y;
  q.theta = r.theta;
  q.velocity = r.velocity;
  q.dist = r.laneline_distances;
  q.ne = r.nearest_edge;
  q.lp = r.local_path;
  q.lp_len = r.lp_len;
  q.row0 = row0;
  return q;
}

// ---------------------------------------------------------------------------------------------
// Kernel 2: renderer.py:36-51 -- cv2.polylines of every segment into LDS bit-planes, then the frame is
// written to HBM once with 16-byte stores.  One wavefront per env; small argument block, small LDS.
struct RCam {
  int H, W, wpr, band_rows, n_bands, thickness, format;
  int cap_r;                  // round-cap radius of ThickLine: (thickness*32768 + 32768) >> 16
  unsigned char cap_hw[32];   // half width of the cap per |row offset| (midpoint circle of drawing.cpp)
  unsigned int cap_hw4;       // cap_hw[0..3] packed, for radii <= 3 (thickness <= 6): a scalar, where indexing the table
                              // is a vector load from the kernarg segment per cap row (three dependent ~1 us loads per frame)
};
struct RArgs {
  int N, C;
  int env0;
  RCam cam;
  unsigned char colors[16][3];
  const int* seg_g;  // [N][seg_cap][5]
  const int* seg_n;  // [N]
  int seg_cap;
  unsigned char* obs;
  const unsigned char* mask;  // reset mask (NULL = all envs)
  int off_tab, off_bits;
  unsigned int flags;
  // raster launches over several steps' frames (grid.y = frames rows): row blockIdx.y reads draw list / pose row
  // seg_row0 + blockIdx.y of the library's scratch ring, writes its frame obs_row_stride bytes behind the previous
  // row's, and is step noise_row0 + blockIdx.y of the call (position in the blob stream)
  int seg_row0;
  int noise_row0;
  long long obs_row_stride;
  // NoiseObservationWrapper fused into the raster stage (class masks only): blobs per plane (0 = off), radius bound,
  // the per-radius span table and the blob stream (position of frame row 0 = *noise_step)
  int noise_blobs, noise_max_radius;
  const unsigned char* noise_hw;
  unsigned long long noise_seed;
  const unsigned int* noise_step;
  int seg_lds_off, seg_lds_cap;  // see KArgs
  unsigned int colors_packed[16];  // colors[c] as r | g << 8 | b << 16: read with scalar loads (uniform class index)
};

#ifndef TC_RASTER_WAVES
#define TC_RASTER_WAVES 4
#endif
// LDSONLY (tc_frame_kernel): the stage reads draw-list entries from LDS only, and its common path contains no vector load
// at all.  That matters beyond the loads themselves: vector memory operations retire in issue order, and the compiler
// guards a load that MAY have been issued (the k >= seg_lds_cap branch of seg_get) with s_waitcnt vmcnt(0) at the join --
// which, executed by a wavefront that took the LDS branch, waits for nothing but the wavefront's own earlier STORES: the
// zero planes stored at the head of the stage, the previous band's pixels.  Without the branch the stores drain behind
// the raster work (cfg5: band b's 38 KB land while band b+1 is rasterised).
//   A frame whose list outgrew the LDS head (nseg_in > a.seg_lds_cap: the caller has then put the WHOLE list into the
// global array, head included) is rasterised batch by batch all the same: each batch of RB entries is first copied from
// the global list to the start of the LDS region -- loads and their wait inside a branch the common path never enters.
// Flow control of a banded frame's stores (cfg5: 19 bands of 38 KB): before a band's stores are issued, the stores of the
// band before it must have landed.  With no wait at all a wavefront runs up to 63 store instructions ahead and the chip
// as a whole writes SLOWER (cfg5, stores only: 2.81 ms per dispatch against 2.62); with the wait at the head of the band
// (where the compiler used to put one) the raster work of band b+1 cannot overlap the stores of band b (3.36 ms -> see
// DESIGN.md 6); here the raster work overlaps and at most one band is in flight.
#ifndef TC_BAND_VMCNT
#define TC_BAND_VMCNT 0
#endif
#define TC_BAND_THROTTLE() asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TC_BAND_VMCNT) : "memory")
template <bool THICK, int FMT, bool LDSONLY = false, int RB = RB_MAX>
__device__ __forceinline__ void raster_body(const RArgs& a0, unsigned char* smem, int env, unsigned char* obs_base,
                                            const int tid, const size_t seg_slot0, const int nseg_in,
                                            const unsigned int used_in, const int frame_row) {
  const RArgs& a = a0;
  const RCam& cam = a.cam;
  unsigned int* bits = (unsigned int*)(smem + R_OFF_BITS_OF(RB));
  const int* segg = a.seg_g + (seg_slot0 + env) * a.seg_cap * 5;
  // draw-list entry k: from LDS when the camera stage of this wavefront left it there (tc_frame_kernel), else global
  const LdsIntPtr seg_lds = (LdsIntPtr)(smem + a.seg_lds_off);
  const int seg_lds_cap = a.seg_lds_cap;
  struct SegV {
    int v[5];
    __device__ __forceinline__ int operator[](int i) const { return v[i]; }
  };
  // LDSONLY: entry k sits at LDS slot k - lds_base (lds_base = 0, or the first entry of the batch just copied in)
  int lds_base = 0;
  const bool refill = LDSONLY && nseg_in > seg_lds_cap;  // (wave-uniform)
  const int rb_step = (refill && seg_lds_cap < RB) ? seg_lds_cap : RB;  // batch size (the LDS region holds >= 8 entries: tc_env_create)
  auto seg_refill = [&](int base, int nb) {  // global entries [base, base + nb) -> LDS slots [0, nb); nb <= rb_step <= seg_lds_cap
    lds_sync();  // the batch before has been read
    for (int i = threadIdx.x; i < 5 * nb; i += TC_NT) seg_lds[i] = segg[5 * base + i];
    lds_base = base;
    lds_sync();
  };
  auto seg_get = [&](int k) {
    SegV r;
    if (LDSONLY) {
#pragma unroll
      for (int i = 0; i < 5; i++) r.v[i] = seg_lds[5 * (k - lds_base) + i];
    } else if (k < seg_lds_cap) {
#pragma unroll
      for (int i = 0; i < 5; i++) r.v[i] = seg_lds[5 * k + i];
    } else {
#pragma unroll
      for (int i = 0; i < 5; i++) r.v[i] = segg[5 * k + i];
    }
    return r;
  };
  auto seg_layer = [&](int k) { return LDSONLY ? seg_lds[5 * (k - lds_base)] : k < seg_lds_cap ? seg_lds[5 * k] : segg[5 * k]; };
  // draw-list length and the layers that have a segment in this frame (wave-uniform): handed over in registers by the
  // camera stage of the same wavefront, or read back when this is a launch of its own (nseg_in < 0)
  int nseg = nseg_in;
  unsigned int used_layers = used_in;
  TSTAMP(8);
  if (!LDSONLY && nseg_in < 0) {
    nseg = a.seg_n[seg_slot0 + env];
    used_layers = 0;
    for (int k = tid; k < nseg; k += TC_NT) used_layers |= 1u << segg[5 * k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) used_layers |= __shfl_xor(used_layers, off);
  }

  const int H = cam.H, W = cam.W, wpr = cam.wpr, C = a.C;
  unsigned char* out = obs_base + (size_t)env * ((size_t)H * W * (FMT == TC_FMT_CLASSES ? C : 3));
  int* lt = (int*)(smem + R_OFF_TAB);     // [RB*4][5] outline-edge parameters
  int* lc = lt + RB * 4 * 5;              // [RB*4] chunks per outline edge, then exclusive prefix
  int* fl = lc + RB * 4;                  // [RB] first fill row of the segment in this band
  int* fc = fl + RB;                      // [RB] fill rows, then exclusive prefix
  int* fm = fc + RB;                      // [RB] pieces | walker mask << 8 | quad valid << 16
  int* fd = fm + RB;                      // [RB][2] ThickLine's (dp.x, dp.y)
  int* fpy = fd + 2 * RB;                 // [RB][4] fill pieces: start row
  int* fpv = fpy + 4 * RB;                // [RB][4] fill pieces: polygon vertices idx0 | idx << 2
  long long* fpx = (long long*)(fpv + 4 * RB);  // [RB][4] x at the start row (16.16)
  long long* fpd = fpx + 4 * RB;          // [RB][4] dx per row
  // Class masks, one band: the planes of layers that have no segment in this frame are all zeros (unless noise blobs are
  // going to copy into them) and are stored NOW, at the head of the stage -- their 16-byte stores drain while the used
  // planes are rasterised instead of queueing behind each other at the end of the wavefront's life (the store phase
  // is 15 % of the kernel's time for 9 % of its instructions).  A frame with no segment at all (37-57 % of the
  // benchmark's frames once cars have wandered off the road) is done here: no planes to clear, no tables, no expansion.
  unsigned int early_zero = 0;
  if (FMT == TC_FMT_CLASSES && cam.n_bands == 1 && (W & 15) == 0 && a.noise_blobs == 0 && !DBG_ON(a.flags, DBG_SKIP_STORE)) {
    const int per_plane = H * (W >> 4);
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    for (int c = 0; c < C; c++) {
      if ((used_layers >> c) & 1u) continue;
      uint4* po = (uint4*)(out + (size_t)c * H * W);
      for (int q = tid; q < per_plane; q += TC_NT) po[q] = zero4;
    }
    early_zero = ~used_layers;
    if (nseg == 0) {
      TSTAMP(9);
      TSTAMP(10);
      TSTAMP(11);
      TSTAMP(12);
      TSTAMP(13);
      TSTAMP_REAL(31);
      return;
    }
  }
  for (int band = 0; band < cam.n_bands; band++) {
    const int y0 = band * cam.band_rows;
    const int y1 = (y0 + cam.band_rows < H) ? y0 + cam.band_rows : H;
    const int rows = y1 - y0;
    const int nwords = C * cam.band_rows * wpr;
    if (cam.n_bands > 1 && (W & 15) == 0) {
      // A frame taller than one LDS band (cfg5: 480 x 640 in 19 bands): most bands see no lane line at all.  Such a band
      // is all zeros whatever the format (and whatever the noise blobs copy or erase), so it is written as a plain
      // stream of 16-byte zero stores: no bit-planes to clear, nothing to scan, and no wait for the zeros to land before
      // pixel stores that never come.
      bool touched = refill;  // (a list that is not in LDS as a whole is not worth a pass of its own)
      const long long mg = (long long)cam.thickness + 2;  // ThickLine paints within thickness / 2 + 1 rows of the segment
      for (int k0 = 0; k0 < nseg && !touched; k0 += TC_NT) {
        bool t = false;
        if (k0 + tid < nseg) {
          const SegV sg = seg_get(k0 + tid);
          const long long ya = sg[2], yb = sg[4];
          t = (ya < yb ? yb : ya) + mg >= y0 && (ya < yb ? ya : yb) - mg < y1;
        }
        touched = __ballot(t) != 0;
      }
      if (!touched && DBG_ON(a.flags, DBG_SKIP_STORE)) continue;
      if (!touched) {
        const uint4 z = make_uint4(0, 0, 0, 0);
        TC_BAND_THROTTLE();
        if (FMT == TC_FMT_CLASSES) {
          const int per_plane = rows * (W >> 4);
          for (int c = 0; c < C; c++) {
            uint4* dst = (uint4*)(out + ((size_t)c * H + y0) * W);
            for (int q = tid; q < per_plane; q += TC_NT) dst[q] = z;
          }
        } else {
          const int n16 = rows * W * 3 / 16;
          uint4* dst = (uint4*)(out + (size_t)y0 * W * 3);
          for (int q = tid; q < n16; q += TC_NT) dst[q] = z;
        }
        continue;
      }
    }
    {  // (R_OFF_BITS_OF(RB) is a multiple of 16: whole 16-byte writes, then the odd words)
      const int n4 = nwords >> 2;
      for (int i = tid; i < n4; i += TC_NT) ((uint4*)bits)[i] = make_uint4(0, 0, 0, 0);
      for (int i = (n4 << 2) + tid; i < nwords; i += TC_NT) bits[i] = 0;
    }
    lds_sync();
    TSTAMP(9);
#ifdef TC_TIMING_LDS
    TSTAMP(26);  // back to back with probe 9: what a probe costs
#endif
    Ras r;
    r.W = W;
    r.H = H;
    r.wpr = wpr;
    r.y0 = y0;
    r.y1 = y1;
    const int plane = cam.band_rows * wpr;
    if (DBG_ON(a.flags, DBG_SKIP_RASTER)) {
    } else if (!THICK) {
      for (int base = 0; base < nseg; base += refill ? rb_step : nseg) {
        const int nb = refill ? (nseg - base < rb_step ? nseg - base : rb_step) : nseg;
        if (refill) seg_refill(base, nb);
        for (int k = base + tid; k < base + nb; k += TC_NT) {
          const SegV sg = seg_get(k);
          r.bits = bits + sg[0] * plane;
          r_line_bresenham(r, sg[1], sg[2], sg[3], sg[4]);  // ThickLine with thickness <= 1 is a plain Line()
        }
      }
    } else {
      // ThickLine = FillConvexPoly(quad) [4 Line2 outline edges + scanline fill] + 2 round caps.
      // Segments are taken RB at a time; their pixel work is cut into small uniform items that are dealt
      // to the 64 lanes through prefix sums, so one long line does not serialise the wave.
      for (int base = 0; base < nseg; base += rb_step) {
        // (the loops usually run once: arguments are re-read inside them instead of being hoisted in front and held --
        // or spilled -- across the whole stage)
        const RArgs& a = kernarg_again(a0);
        const RCam& cam = a.cam;
        const int nb = nseg - base < rb_step ? nseg - base : rb_step;
        if (refill) seg_refill(base, nb);
#ifdef TC_TIMING_LDS
        if (cam.n_bands >= 0) TSTAMP(27);  // first value of the re-read argument block has arrived
#endif
        if (cam.n_bands > 1) {
          // Frames taller than one LDS band run this loop once per band (cfg5: 19 bands).  A batch none of whose
          // segments can reach the band's rows is skipped whole -- set-up, prefix sums, pixel loops: most bands of a
          // 480 x 640 frame see no lane line at all and are then a pure zero-store stream.
          bool t = false;
          if (tid < nb) {
            const SegV sg = seg_get(base + tid);
            const long long ya = sg[2], yb = sg[4], mg = (long long)cam.thickness + 2;
            t = (ya < yb ? yb : ya) + mg >= y0 && (ya < yb ? ya : yb) - mg < y1;
          }
          if (__ballot(t) == 0) continue;
        }
#ifdef TC_TIMING_LDS
        if (cam.thickness > 0) TSTAMP(24);  // (depends on a kernel argument re-read above: the scalar loads have returned)
#endif
        if (tid < RB) {  // per segment: ThickLine's dp, fill walker pieces, fill rows of this band
          int nrow = 0, lo = 0, np = 0, wm = 0, dpx = 0, dpy = 0, okq = 0;
#ifdef TC_TIMING_LDS
          TSTAMP(25);
#endif
          if (tid < nb) {
            const SegV sg = seg_get(base + tid);
            long long qx0, qx1, qx2, qx3, qy0, qy1, qy2, qy3;
            TSTAMP(20);
            // Frames taller than one LDS band are rasterised band by band, and every band runs this set-up again.  A
            // segment whose rows cannot reach the band is dropped before any of it: ThickLine paints within
            // thickness / 2 + 1 rows of the segment (quad corners p +- dp with |dp| <= thickness / 2 + 1 px, cap radius
            // (thickness + 1) / 2), and `thickness + 2` rows are allowed here.  On cfg5 (12 bands) a segment survives
            // this in 1-3 bands instead of 12.
            bool touch = true;
            if (cam.n_bands > 1) {
              const long long ya = sg[2], yb = sg[4], mg = (long long)cam.thickness + 2;
              const long long ymin = (ya < yb ? ya : yb) - mg, ymax = (ya < yb ? yb : ya) + mg;
              touch = ymax >= y0 && ymin < y1;
            }
            if (touch && !DBG_ON(a.flags, DBG_SKIP_QUAD) &&
                r_quad(sg[1], sg[2], sg[3], sg[4], cam.thickness, qx0, qx1, qx2, qx3, qy0, qy1, qy2, qy3)) {
              TSTAMP(21);
              okq = 1;
              dpx = (int)(qx0 - (long long)sg[1] * TC_XY_ONE);
              dpy = (int)(qy0 - (long long)sg[2] * TC_XY_ONE);
              int hi = -1;
              if (!DBG_ON(a.flags, DBG_SKIP_EVENTS))
              np = r_fill_events(W, H, qx0, qx1, qx2, qx3, qy0, qy1, qy2, qy3, fpy + 4 * tid, fpv + 4 * tid, wm, lo,
                                 hi);
              TSTAMP(22);
              if (lo < y0) lo = y0;
              if (hi > y1 - 1) hi = y1 - 1;
              if (np > 0 && hi >= lo) nrow = hi - lo + 1;
            }
          }
          fl[tid] = lo;
          fc[tid] = nrow;
          fm[tid] = np | (wm << 8) | (okq << 16);
          fd[2 * tid] = dpx;
          fd[2 * tid + 1] = dpy;
        }
        lds_sync();
        TSTAMP(10);
        MARK("outline setup");
        for (int t = tid; t < RB * 4; t += TC_NT) {  // outline edges: clip + DDA parameters
          int nchunk = 0;
          if (t < nb * 4 && ((fm[t >> 2] >> 16) & 1)) {
            const int j = t >> 2, e = t & 3;
            const SegV sg = seg_get(base + j);
            const long long p0x = (long long)sg[1] * TC_XY_ONE, p0y = (long long)sg[2] * TC_XY_ONE;
            const long long p1x = (long long)sg[3] * TC_XY_ONE, p1y = (long long)sg[4] * TC_XY_ONE;
            const long long dpx = fd[2 * j], dpy = fd[2 * j + 1];
            const long long qx0 = p0x + dpx, qx1 = p0x - dpx, qx2 = p1x - dpx, qx3 = p1x + dpx;
            const long long qy0 = p0y + dpy, qy1 = p0y - dpy, qy2 = p1y - dpy, qy3 = p1y + dpy;
            // FillConvexPoly walks p0 = v[3]; Line2(p0, v[i]); p0 = v[i]: outline edge e runs v[e-1] -> v[e] with
            // v0 = p0+dp, v1 = p0-dp, v2 = p1-dp, v3 = p1+dp
            const bool a_is_p1 = e == 0 || e == 3, b_is_p1 = e >= 2;
            const bool a_minus = e == 2 || e == 3, b_minus = e == 1 || e == 2;
            long long ax = (a_is_p1 ? p1x : p0x) + (a_minus ? -dpx : dpx), ay = (a_is_p1 ? p1y : p0y) + (a_minus ? -dpy : dpy);
            long long bx = (b_is_p1 ? p1x : p0x) + (b_minus ? -dpx : dpx), by = (b_is_p1 ? p1y : p0y) + (b_minus ? -dpy : dpy);
            if (e < (fm[j] & 0xff) && !DBG_ON(a.flags, DBG_SKIP_SLOPE)) {  // fill piece e of this segment: x at its start row and slope
              long long xs, dxs;
              r_fill_slope(qx0, qx1, qx2, qx3, qy0, qy1, qy2, qy3, fpy[t], fpv[t], xs, dxs);
              fpx[t] = xs;
              fpd[t] = dxs;
            }
            LineP L;
            L.ecount = -1;
            if (!DBG_ON(a.flags, DBG_SKIP_LINESETUP)) L = r_line2_setup(W, H, ax, ay, bx, by);
            if (L.ecount >= 0) {
              r.bits = bits + sg[0] * plane;
              r_put(r, L.ex, L.ey);
              int* o = lt + 5 * t;
              o[0] = L.a;
              o[1] = L.b;
              o[2] = L.step;
              o[3] = L.ecount | (L.xmajor << 30);
              o[4] = sg[0];
              nchunk = (L.ecount + LCH) / LCH;
            }
          }
          lc[t] = nchunk;
        }
        lds_sync();
        MARK("prefix sums");
        int tot_l, tot_f;
        {  // exclusive prefix sums (RB*4 = 2 entries per lane, or 1 when RB = 16; RB entries on the low lanes)
          static_assert(RB * 4 == 2 * TC_NT || RB * 4 == TC_NT, "outline-edge chunk counts: one or two per lane");
          constexpr bool TWO = RB * 4 == 2 * TC_NT;
          int v0 = TWO ? lc[2 * tid] : lc[tid], v1 = TWO ? lc[2 * tid + 1] : 0;
          int inc = wave_incl_scan(v0 + v1, tid);
          tot_l = __builtin_amdgcn_readlane(inc, TC_NT - 1);
          int f = tid < RB ? fc[tid] : 0;
          int finc = wave_incl_scan(f, tid);
          tot_f = __builtin_amdgcn_readlane(finc, TC_NT - 1);
          if (TWO) {
            lc[2 * tid] = inc - v0 - v1;  // each lane rewrites only the entries it read
            lc[2 * tid + 1] = inc - v1;
          } else {
            lc[tid] = inc - v0;
          }
          if (tid < RB) fc[tid] = finc - f;
        }
        lds_sync();
        TSTAMP(11);
        for (int c = tid; c < tot_l && !DBG_ON(a.flags, DBG_SKIP_R2); c += TC_NT) {  // outline pixels, LCH steps per chunk
          int lo = 0, hi = RB * 4;
          while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (lc[mid] <= c) lo = mid; else hi = mid;
          }
          const int* o = lt + 5 * lo;
          const int ecount = o[3] & 0xffff, xmajor = (o[3] >> 30) & 1;
          const int k0 = (c - lc[lo]) * LCH;
          const int k1 = k0 + LCH - 1 < ecount ? k0 + LCH - 1 : ecount;
          r.bits = bits + o[4] * plane;
          r_line2_pixels(r, o[0], o[1], o[2], xmajor, k0, k1);
        }
        MARK("fill rows");
        for (int c = tid; c < tot_f && !DBG_ON(a.flags, DBG_SKIP_R3); c += TC_NT) {  // scanline fill, one row per item
          int lo = 0, hi = RB;
          while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (fc[mid] <= c) lo = mid; else hi = mid;
          }
          const int row = fl[lo] + (c - fc[lo]);
          const int mm = fm[lo];
          r.bits = bits + seg_layer(base + lo) * plane;
          r_fill_row(r, row, mm & 0xff, (mm >> 8) & 0xff, fpy + 4 * lo, fpx + 4 * lo, fpd + 4 * lo);
        }
        MARK("caps");
        for (int t = tid; t < nb * 2 && !DBG_ON(a.flags, DBG_SKIP_R4); t += TC_NT) {  // round caps (flags = 3: both ends)
          const int k = base + (t >> 1);
          const SegV sg = seg_get(k);
          r.bits = bits + sg[0] * plane;
          const int cx = (t & 1) ? sg[3] : sg[1], cy = (t & 1) ? sg[4] : sg[2];
          if (cam.n_bands > 1 && ((long long)cy + cam.cap_r < y0 || (long long)cy - cam.cap_r >= y1)) continue;  // cap outside the band
          if (cam.cap_r < 32)
            r_cap(r, cx, cy, cam.cap_r, cam.cap_hw, cam.cap_hw4);
          else
            r_circle_fill(r, cx, cy, cam.cap_r);
        }
        lds_sync();
      }
    }
    lds_sync();
    if (FMT == TC_FMT_CLASSES && a.noise_blobs > 0) {
      // NoiseObservationWrapper (wrapper/observation.py:15-27) on the bit-planes, before they are expanded: the frame
      // never makes the extra round trip through HBM a pass of its own costs (71 us per step on cfg3 in round 1).
      // Every operation of the reference is per pixel -- plane c |= plane src & circle, or plane c &= ~circle, blob
      // after blob, plane after plane -- so one lane owns one (row, 32-pixel word) of ALL planes and replays the whole
      // blob sequence on it: no lane ever reads a word another lane writes, no barrier between blobs.  Plane c's word
      // lives in a register while its n_blobs are applied; planes < c are read back noised, planes > c as rasterised,
      // exactly the sequential order.  A filled cv2.circle with its centre inside the image is, row by row, the span
      // centre +- hw[radius][|row - cy|] clipped to the image (see tc_noise_kernel).
      const int nbl = a.noise_blobs, nb = C * nbl, mr = a.noise_max_radius;
      int* bp = (int*)(smem + R_OFF_TAB);                 // [nb][2] (x | y << 16), (r | mode << 12 | src << 16): the
      unsigned char* bh = (unsigned char*)(bp + 2 * nb);  // raster tables are dead here;  [nb][rows] half widths
      const bool staged = nb * 8 + nb * rows <= R_TAB_BYTES_OF(RB);
      const unsigned int nstep = *a.noise_step + (unsigned int)frame_row;
      for (int k = tid; k < nb; k += TC_NT) {
        const tc_blob bl = tc_noise_blob(a.noise_seed, (uint32_t)env, nstep, (uint32_t)k, W, H, mr, C);
        bp[2 * k] = bl.x | (bl.y << 16);
        bp[2 * k + 1] = bl.r | (bl.mode << 12) | (bl.src << 16);
      }
      lds_sync();
      if (staged) {  // half width of every (blob, row of this band): one table byte each, fetched with the loads in flight
        for (int row = tid; row < rows; row += TC_NT) {
#pragma unroll 5
          for (int k = 0; k < nb; k++) {
            const int by = bp[2 * k] >> 16, rr = bp[2 * k + 1] & 0xfff;
            int t = y0 + row - by;
            t = t < 0 ? -t : t;
            bh[k * rows + row] = t <= rr ? a.noise_hw[rr * mr + t] : (unsigned char)0;
          }
        }
        lds_sync();
      }
      const int items = rows * wpr;
      for (int it = tid; it < items; it += TC_NT) {
        const int row = it / wpr, w = it - row * wpr;
        const int yy = y0 + row, x_lo = w << 5;
        for (int c = 0; c < C; c++) {
          unsigned int cur = bits[(c * cam.band_rows + row) * wpr + w];
          for (int j = 0; j < nbl; j++) {
            const int k = c * nbl + j;
            const int p0 = bp[2 * k], p1 = bp[2 * k + 1];
            const int bx = p0 & 0xffff, by = p0 >> 16, rr = p1 & 0xfff, src = p1 >> 16;
            int t = yy - by;
            t = t < 0 ? -t : t;
            if (t <= rr) {
              const int hwv = staged ? (int)bh[k * rows + row] : (int)a.noise_hw[rr * mr + t];
              int xl = bx - hwv, xr = bx + hwv;
              xl = xl < 0 ? 0 : xl;
              xr = xr > W - 1 ? W - 1 : xr;
              const int b0 = xl > x_lo ? xl - x_lo : 0, b1 = xr < x_lo + 31 ? xr - x_lo : 31;
              if (b1 >= b0) {
                const unsigned int mask = (0xffffffffu >> (31 - b1)) & (0xffffffffu << b0);
                if ((p1 >> 12) & 1) {  // observation.py:22-24
                  const unsigned int sw = src == c ? cur : bits[(src * cam.band_rows + row) * wpr + w];
                  cur |= sw & mask;
                } else {
                  cur &= ~mask;  // observation.py:26
                }
              }
            }
          }
          bits[(c * cam.band_rows + row) * wpr + w] = cur;
        }
      }
      lds_sync();
      used_layers = C >= 32 ? 0xffffffffu : ((1u << C) - 1u);  // a copy may have filled a plane that had no segment
    }
    TSTAMP(12);
    const RArgs& a = kernarg_again(a0);
    const RCam& cam = a.cam;
    if (cam.n_bands > 1) TC_BAND_THROTTLE();
    if (DBG_ON(a.flags, DBG_SKIP_STORE)) {
    } else if (FMT == TC_FMT_CLASSES) {
      if ((W & 15) == 0) {
        // 16 pixels -> one 16-byte store per lane, consecutive lanes on consecutive addresses
        // (two 16-byte stores per lane at a 32-byte lane stride were tried: 2x slower, half-line writes)
        const int g = W >> 4;  // 16-pixel groups per row
        const int per_plane = rows * g;
        const bool dense = (W & 31) == 0;  // then 16-bit half h of plane word q>>1 is group q
        const uint4 zero4 = make_uint4(0, 0, 0, 0);
        for (int c = 0; c < C; c++) {
          const unsigned int* pl = bits + c * cam.band_rows * wpr;
          unsigned char* po = out + ((size_t)c * H + y0) * W;
          if ((early_zero >> c) & 1u) continue;  // stored as zeros at the head of the stage
          if (!((used_layers >> c) & 1u)) {  // no segment of this layer in the frame: the plane is all zeros
            for (int q = tid; q < per_plane; q += TC_NT) *(uint4*)(po + (size_t)q * 16) = zero4;
            continue;
          }
          if (dense) {
            // four groups per lane and round: the four plane words are fetched together, so the wavefront waits for the
            // LDS once per round instead of once per store (the phase is a chain of read -> expand -> store; measured
            // with the stores switched off it is 15 % of the kernel's time for 9 % of its instructions)
            for (int q0 = tid; q0 < per_plane; q0 += 4 * TC_NT) {
              unsigned int w4[4];
#pragma unroll
              for (int j = 0; j < 4; j++) {
                const int q = q0 + j * TC_NT;
                w4[j] = q < per_plane ? pl[q >> 1] : 0u;
              }
#pragma unroll
              for (int j = 0; j < 4; j++) {
                const int q = q0 + j * TC_NT;
                const unsigned int b16 = (w4[j] >> ((q & 1) << 4)) & 0xffffu;
                uint4 o = zero4;
                if (__ballot(b16 != 0)) {  // wave-uniform: skip the bit spreading when all 64 groups are empty
                  o.x = spread4(b16 & 15u);
                  o.y = spread4((b16 >> 4) & 15u);
                  o.z = spread4((b16 >> 8) & 15u);
                  o.w = spread4(b16 >> 12);
                }
                if (q < per_plane) *(uint4*)(po + (size_t)q * 16) = o;  // rows of a plane are contiguous: group q sits at byte 16 q
              }
            }
            continue;
          }
          for (int q = tid; q < per_plane; q += TC_NT) {
            const int yy = q / g;
            const int xx = (q - yy * g) << 4;
            const unsigned int b16 = (pl[yy * wpr + (xx >> 5)] >> (xx & 31)) & 0xffffu;
            uint4 o = zero4;
            if (__ballot(b16 != 0)) {
              o.x = spread4(b16 & 15u);
              o.y = spread4((b16 >> 4) & 15u);
              o.z = spread4((b16 >> 8) & 15u);
              o.w = spread4(b16 >> 12);
            }
            *(uint4*)(po + (size_t)q * 16) = o;
          }
        }
      } else {
        const int per_plane = rows * W;
        const int total = C * per_plane;
        for (int q = tid; q < total; q += TC_NT) {
          int c = q / per_plane, p = q - c * per_plane;
          int yy = p / W, xx = p - yy * W;
          unsigned int word = bits[(c * cam.band_rows + yy) * wpr + (xx >> 5)];
          out[((size_t)c * H + y0 + yy) * W + xx] = ((word >> (xx & 31)) & 1u) ? 255 : 0;
        }
      }
    } else {
      // rgb: painter's order (renderer.py:41-43): the highest layer covering a pixel wins
      if ((W & 15) == 0) {
        // this band's rows as zeros first, then only the pixels that have a bit set in some plane.  The byte stores
        // follow the zero stores of the same wavefront to the same addresses in program order, and the memory system
        // keeps stores of one wavefront to one address in that order (the compiler relies on the same guarantee: it
        // never waits between two stores that may alias) -- no wait for the zeros to land: that wait was a full store
        // round trip per band, the largest single item of a touched band (cfg5 ablation, profiles/r03).
        // (Zeroing the whole frame before the first band was measured: 16 % slower)
        {
          const int n16 = rows * W * 3 / 16;
          uint4* dst = (uint4*)(out + (size_t)y0 * W * 3);
          const uint4 z = make_uint4(0, 0, 0, 0);
          for (int q = tid; q < n16; q += TC_NT) dst[q] = z;
        }
        // Class by class, highest first; a class paints the pixels it covers that no higher class covers.  The colour of a
        // class is then ONE value for the whole wavefront, fetched with a scalar load.  (Looked up per pixel --
        // colors[top][k] with `top` differing between lanes -- it was a vector load from the kernarg segment, and vector
        // memory operations retire in order: the wavefront sat out the round trip of this band's zero stores at every
        // such load, which is why raster and store time of cfg5 added up instead of overlapping -- ablation)
        const int nw = rows * wpr;
        const int pstride = cam.band_rows * wpr;
        const __attribute__((address_space(4))) unsigned int* pk =
            (const __attribute__((address_space(4))) unsigned int*)(unsigned long long)&a.colors_packed[0];
        for (int c = C - 1; c >= 0; c--) {
          const unsigned int col = pk[c];
          const unsigned char c0 = (unsigned char)col, c1 = (unsigned char)(col >> 8), c2 = (unsigned char)(col >> 16);
          for (int q = tid; q < nw; q += TC_NT) {
            unsigned int mine = bits[c * pstride + q];
            if (mine == 0) continue;
            for (int h = c + 1; h < C; h++) mine &= ~bits[h * pstride + q];
            const int yy = q / wpr, xw = (q - yy * wpr) << 5;
            unsigned char* row = out + ((size_t)(y0 + yy) * W + xw) * 3;
            while (mine) {
              const int bpos = __ffs(mine) - 1;
              mine &= mine - 1;
              unsigned char* o = row + bpos * 3;
              o[0] = c0;
              o[1] = c1;
              o[2] = c2;
            }
          }
        }
      } else if ((W & 3) == 0) {
        const int total = rows * W / 4;
        for (int g = tid; g < total; g += TC_NT) {
          int pix = g * 4;
          int yy = pix / W, xx = pix - yy * W;
          unsigned int top4 = 0;  // per pixel: 1 + index of the highest layer set, 4 pixels in 4 bytes
          for (int c = 0; c < C; c++) {
            unsigned int word = bits[(c * cam.band_rows + yy) * wpr + (xx >> 5)];
            unsigned int m4 = spread4((word >> (xx & 31)) & 15u);  // 0xFF where the layer covers the pixel
            top4 = (top4 & ~m4) | (m4 & (0x01010101u * (unsigned)(c + 1)));
          }
          unsigned char px[12];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            int top = (int)((top4 >> (8 * k)) & 255u) - 1;
            px[3 * k] = top >= 0 ? a.colors[top][0] : 0;
            px[3 * k + 1] = top >= 0 ? a.colors[top][1] : 0;
            px[3 * k + 2] = top >= 0 ? a.colors[top][2] : 0;
          }
          unsigned int* o = (unsigned int*)(out + ((size_t)(y0 + yy) * W + xx) * 3);
          o[0] = px[0] | (px[1] << 8) | (px[2] << 16) | ((unsigned)px[3] << 24);
          o[1] = px[4] | (px[5] << 8) | (px[6] << 16) | ((unsigned)px[7] << 24);
          o[2] = px[8] | (px[9] << 8) | (px[10] << 16) | ((unsigned)px[11] << 24);
        }
      } else {
        const int total = rows * W;
        for (int p = tid; p < total; p += TC_NT) {
          int yy = p / W, xx = p - yy * W;
          int top = -1;
          for (int c = 0; c < C; c++) {
            unsigned int word = bits[(c * cam.band_rows + yy) * wpr + (xx >> 5)];
            if ((word >> (xx & 31)) & 1u) top = c;
          }
          unsigned char* o = out + ((size_t)(y0 + yy) * W + xx) * 3;
          o[0] = top >= 0 ? a.colors[top][0] : 0;
          o[1] = top >= 0 ? a.colors[top][1] : 0;
          o[2] = top >= 0 ? a.colors[top][2] : 0;
        }
      }
    }
    lds_sync();
  }
  TSTAMP(13);
  TSTAMP_REAL(31);
}

template <bool THICK, int FMT>
__global__ __launch_bounds__(TC_NT, TC_RASTER_WAVES) void tc_raster_kernel(RArgs a_unused) {
  extern __shared__ __align__(16) unsigned char smem[];
  // raster_body re-reads its arguments through kernarg_again(): it must be handed the block in the kernarg segment
  // itself, not a by-value copy the compiler is free to keep in registers or scratch
  const RArgs& a = *(const RArgs*)(const __attribute__((address_space(4))) RArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  const int env = a.env0 + blockIdx.x;
  if (env >= a.N) return;
  if (a.mask && !a.mask[env]) return;
  // one workgroup per (frame row, env): with more frames than resident workgroups the dispatcher hands the next frame
  // to whichever slot frees up first, so light and heavy frames balance out across the chip
  const size_t slot0 = (size_t)(a.seg_row0 + blockIdx.y) * a.N;
  raster_body<THICK, FMT>(a, smem, env, a.obs + (size_t)blockIdx.y * a.obs_row_stride, threadIdx.x, slot0, -1, 0u, a.noise_row0 + (int)blockIdx.y);
}

// ---------------------------------------------------------------------------------------------
// NoiseObservationWrapper (wrapper/observation.py:15-27) as a pass over a finished class-mask observation.
// Every operation of the reference is per pixel -- obs[c] |= obs[src] & circle, or obs[c] &= ~circle -- so rows are
// independent: the frame is taken through LDS as bit-planes band by band (same layout as the rasteriser), the blobs
// are applied to the band in order, and the band is written back.  A filled cv2.circle whose centre lies inside the
// image is, row by row, the span centre +- hw[radius][|row - cy|] clipped to the image (Circle() of drawing.cpp paints
// symmetric spans and its bounding-box tests never reject a row when the centre is inside); hw is tabulated on the
// host by running that midpoint algorithm once per radius.
struct NArgs {
  int N, C, H, W, wpr, band_rows, n_bands;
  unsigned char* obs;
  const int* blobs;  // [N][C * n_blobs][5] or NULL: drawn here (tc_rng.h)
  int n_blobs, max_radius;
  const unsigned char* hw;  // [max_radius][max_radius]: half width of row offset t of the circle of radius r
  unsigned long long seed;
  const unsigned int* step;  // device counter of noise passes so far (advanced by tc_noise_tick on the same stream:
                             // a kernel argument would be frozen into a captured HIP graph and replay the same blobs)
  unsigned int inv_cpr;  // 2^32 / (W / 16) + 1, for fdiv by the 16-pixel chunks per row
};

__global__ void tc_noise_tick(unsigned int* step, unsigned int n) { *step += n; }

// q / d for q < 2^31 with inv = 2^32 / d + 1 (host): one multiply-high and a fix-up instead of a ~30-instruction
// runtime division (as first written this kernel spent most of its ~12 k instructions per wavefront dividing indices)
__device__ __forceinline__ int fdiv(int q, int d, unsigned int inv) {
  int e = (int)__umulhi((unsigned int)q, inv);
  e -= (e * d > q);
  e += ((e + 1) * d <= q);
  return e;
}

// LDS: bit-planes of one band | blob rows [nb][5] | per blob the span-table row of its radius [nb][max_radius]
__global__ __launch_bounds__(TC_NT) void tc_noise_kernel(NArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int env = blockIdx.x, tid = threadIdx.x;
  if (env >= a.N) return;
  const int C = a.C, H = a.H, W = a.W, wpr = a.wpr;
  const int nb = C * a.n_blobs;
  unsigned int* bits = (unsigned int*)smem;
  int* lb = (int*)(smem + (size_t)C * a.band_rows * wpr * 4);
  unsigned char* lhw = (unsigned char*)(lb + nb * 5);
  unsigned char* out = a.obs + (size_t)env * ((size_t)C * H * W);

  // All blobs of this env at once, one per lane: the draws (or the caller's rows) and, behind them, each blob's row of
  // the span table, so that the blob loop touches nothing but LDS.
  for (int k = tid; k < nb; k += TC_NT) {
    tc_blob bl;
    if (a.blobs) {
      const int* p = a.blobs + ((size_t)env * nb + k) * 5;
      bl.x = p[0];
      bl.y = p[1];
      bl.r = p[2];
      bl.mode = p[3];
      bl.src = p[4];
    } else {
      bl = tc_noise_blob(a.seed, (uint32_t)env, *a.step, (uint32_t)k, W, H, a.max_radius, C);
    }
    // caller-provided lists are not trusted with LDS indices: an invalid row becomes a blob that touches nothing.
    // src only matters on the copy branch: erase blobs carry src = -1 in the reference's draw order (nothing is drawn
    // for them, observation.py:25-26) and must stay erasers.
    const bool ok = bl.r >= 1 && bl.r < a.max_radius && (unsigned)bl.x < (unsigned)W && (unsigned)bl.y < (unsigned)H &&
                    (bl.mode == 0 || (unsigned)bl.src < (unsigned)C);
    lb[5 * k] = bl.x;
    lb[5 * k + 1] = bl.y;
    lb[5 * k + 2] = ok ? bl.r : -1;
    lb[5 * k + 3] = bl.mode;
    lb[5 * k + 4] = (ok && bl.mode != 0) ? bl.src : 0;
  }
  __syncthreads();
  for (int k = 0; k < nb; k++) {  // row k of lhw = row r_k of the table, entries 0 .. r_k
    const int r = lb[5 * k + 2];
    for (int t = tid; t <= r; t += TC_NT) lhw[k * a.max_radius + t] = a.hw[r * a.max_radius + t];
  }

  // this lane's place in a pass over (row, 32-bit word) and over (row, 16-pixel chunk): fixed for the whole kernel
  const int lw = wpr <= TC_NT ? wpr : TC_NT;           // words of a row handled side by side
  const int w_lane = tid % lw, r_lane = tid / lw, r_step = TC_NT / lw;
  const bool w16 = (W & 15) == 0;                      // rows are whole 16-byte chunks: one lane per chunk, coalesced
  const int cpr = W >> 4;
  for (int band = 0; band < a.n_bands; band++) {
    const int y0 = band * a.band_rows;
    const int rows = (y0 + a.band_rows < H ? y0 + a.band_rows : H) - y0;
    if (w16) {  // bytes (0 / 255) -> bits, 16 pixels per lane
      const int per_plane = rows * cpr;
      for (int c = 0; c < C; c++) {
        const unsigned char* pl = out + ((size_t)c * H + y0) * W;  // the band's rows of a plane are contiguous
        for (int q = tid; q < per_plane; q += TC_NT) {
          const int row = fdiv(q, cpr, a.inv_cpr), ch = q - row * cpr;
          const uint4 v = *(const uint4*)(pl + (size_t)q * 16);
          const unsigned int h = ((((v.x & 0x01010101u) * 0x10204080u) >> 28) & 0xfu) |
                                 (((((v.y & 0x01010101u) * 0x10204080u) >> 28) & 0xfu) << 4) |
                                 (((((v.z & 0x01010101u) * 0x10204080u) >> 28) & 0xfu) << 8) |
                                 (((((v.w & 0x01010101u) * 0x10204080u) >> 28) & 0xfu) << 12);
          ((unsigned short*)bits)[((c * a.band_rows + row) * wpr) * 2 + ch] = (unsigned short)h;
        }
      }
    } else {
      for (int c = 0; c < C; c++)
        for (int row = r_lane; row < rows; row += r_step)
          for (int w = w_lane; w < wpr; w += lw) {
            const unsigned char* src = out + ((size_t)c * H + y0 + row) * W + w * 32;
            unsigned int word = 0;
            for (int j = 0; j < 32 && w * 32 + j < W; j++) word |= (src[j] ? 1u : 0u) << j;
            bits[(c * a.band_rows + row) * wpr + w] = word;
          }
    }
    __syncthreads();
    int c = 0, left = a.n_blobs;  // plane of blob k = k / n_blobs, kept by counting
    for (int k = 0; k < nb; k++) {  // wrapper/observation.py:16-26, blob after blob
      const int bx = lb[5 * k], by = lb[5 * k + 1], br = lb[5 * k + 2], mode = lb[5 * k + 3], bsrc = lb[5 * k + 4];
      const int ylo = by - br > y0 ? by - br : y0;
      const int yhi = br >= 1 ? (by + br < y0 + rows - 1 ? by + br : y0 + rows - 1) : ylo - 1;
      const unsigned char* hwk = lhw + k * a.max_radius;
      for (int row = ylo + r_lane; row <= yhi; row += r_step) {
        const int t = row > by ? row - by : by - row;
        const int hw = hwk[t];
        int xl = bx - hw, xr = bx + hw;
        xl = xl < 0 ? 0 : xl;
        xr = xr > W - 1 ? W - 1 : xr;
        for (int w = w_lane; w < wpr; w += lw) {
          const int b0 = xl > w * 32 ? xl - w * 32 : 0, b1 = xr < w * 32 + 31 ? xr - w * 32 : 31;
          if (b1 < b0) continue;
          const unsigned int mask = (b1 - b0 == 31) ? 0xffffffffu : (((1u << (b1 - b0 + 1)) - 1u) << b0);
          const int ic = (c * a.band_rows + row - y0) * wpr + w;
          if (mode)
            bits[ic] |= bits[(bsrc * a.band_rows + row - y0) * wpr + w] & mask;  // observation.py:22-24
          else
            bits[ic] &= ~mask;  // observation.py:26
        }
      }
      __syncthreads();  // the next blob may read what this one wrote
      if (--left == 0) {
        c++;
        left = a.n_blobs;
      }
    }
    if (w16) {  // bits -> bytes, one 16-byte store per lane
      const int per_plane = rows * cpr;
      for (int cc = 0; cc < C; cc++) {
        unsigned char* pl = out + ((size_t)cc * H + y0) * W;
        for (int q = tid; q < per_plane; q += TC_NT) {
          const int row = fdiv(q, cpr, a.inv_cpr), ch = q - row * cpr;
          const unsigned int h = ((const unsigned short*)bits)[((cc * a.band_rows + row) * wpr) * 2 + ch];
          uint4 o;
          o.x = spread4(h & 15u);
          o.y = spread4((h >> 4) & 15u);
          o.z = spread4((h >> 8) & 15u);
          o.w = spread4((h >> 12) & 15u);
          *(uint4*)(pl + (size_t)q * 16) = o;
        }
      }
    } else {
      for (int cc = 0; cc < C; cc++)
        for (int row = r_lane; row < rows; row += r_step)
          for (int w = w_lane; w < wpr; w += lw) {
            unsigned char* dst = out + ((size_t)cc * H + y0 + row) * W + w * 32;
            const unsigned int word = bits[(cc * a.band_rows + row) * wpr + w];
            for (int j = 0; j < 32 && w * 32 + j < W; j++) dst[j] = (word >> j) & 1u ? 255 : 0;
          }
    }
    __syncthreads();
  }
}

// Everything a launch is given, as ONE kernel parameter (so that it sits at offset 0 of the kernarg segment).
struct StepArgs {
  KArgs a;
  RArgs r;
  MultiArgs ma;
  int mode, cdtype;
  unsigned int flags;
  const void* car_control;
  const int* maneuver;
  const int* spawn_nodes;
  const unsigned char* mask;
  const int* env_order;  // tc_step_kernel: workgroup w works on env env_order[w] (a permutation of 0..N-1), or NULL = env w
};

// The launch arguments, read through a pointer the optimiser cannot see through.  Inside the step loop of tc_step_multi
// every argument (and every address computed from one) is loop invariant; LLVM hoists all of them out of the loop and
// keeps them alive across the whole ~20 k-instruction body -- measured: 817 SGPRs spilled into VGPR lanes, which in turn
// pushed 223 VGPRs to scratch.  Re-deriving the pointer per step (an empty asm the compiler must assume changes it)
// makes each use reload its argument with a scalar load from the constant address space where it needs it, exactly as
// in a single-step kernel.
// The lane id, likewise: everything computed from it (lane predicates, LDS addresses) is loop invariant too and was
// hoisted and then spilled (-mllvm -disable-machine-licm in the Makefile does the same for the constants the instruction
// selector materialises).
__device__ __forceinline__ int step_lane() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}
typedef const __attribute__((address_space(4))) StepArgs* StepArgsConst;
__device__ __forceinline__ const StepArgs& step_args() {
  unsigned long long p = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return *(const StepArgs*)(StepArgsConst)p;
}

__device__ __forceinline__ bool wants_frame(const StepArgs& sa) {
  return !(sa.flags & (TC_F_NO_OBSERVATION | DBG_SKIP_CAMERA)) && (sa.a.b.obs != nullptr || sa.ma.roll.obs != nullptr);
}

// Simulate stage alone: the steps' state recurrence (phases A + B) with the env's state parked in LDS between steps.
// What becomes of the frames is the launch's choice: nothing (no observation), the camera stage here and a raster
// launch behind (cam_here; maps of the K = 13 variant, TC_FUSE=0), or -- the K-step default -- only the poses, from which
// tc_frame_kernel produces every (step, env) frame as a workgroup of its own.
template <int K, bool CAM>
__global__ __launch_bounds__(TC_NT, (K <= 5 ? TC_MIN_WAVES : K <= 9 ? 3 : 2)) void tc_env_kernel(StepArgs sa_unused) {
  extern __shared__ __align__(16) unsigned char smem[];
  // Touching v127 makes the kernel descriptor ask for 128 VGPRs, i.e. caps the SIMD at the 4 wavefronts the launch
  // needs (N = 4096 one-wavefront workgroups = 4 per SIMD).  The camera-less variant uses 74 registers and would fit 6,
  // and the dispatcher does fill SIMDs that unevenly: measured, wavefronts on the crowded SIMDs took 54 k clocks per
  // step against 27 k on the sparse ones, and a K-step launch lasts as long as its slowest wavefront (580 us, median
  // wavefront 420 us).  (amdgpu_waves_per_eu(4, 4) next to __launch_bounds__ did not raise the allocation.)
  asm volatile("v_mov_b32 v127, 0" ::: "v127");
  const StepArgs& s0 = step_args();
  const int env = s0.a.env0 + blockIdx.x;
  if (env >= s0.a.N) return;
  if (s0.mode == MODE_RESET && s0.mask && !s0.mask[env]) return;  // whole workgroup skips
  live_in(s0.a, smem, env, s0.mode, s0.flags);
  const int nsteps = s0.ma.nsteps;
  long long t_prev = 0;
#ifdef TC_TIMING
  t_prev = clock64();
#endif
  (void)t_prev;
  // The 4 wavefronts of a SIMD compete for its vector issue slots, and at equal priority the arbiter favours the OLDEST
  // one: measured over a 30-step launch, a step took the oldest wavefront 27 k clocks and the youngest 54 k, so the
  // launch lasted 550 us while the median wavefront was done after 420.  Rotating the priority with the step index
  // (each wavefront is on top every 4th step) lets the four progress at the same average rate and finish together.
  const int wave_slot = (int)(__builtin_amdgcn_s_getreg(63492) & 15u);  // HW_REG_HW_ID.wave_id: distinct on a SIMD
  for (int k = 0; k < nsteps; k++) {
    TSTAMP_LOOP(k, nsteps, t_prev);
    switch ((k + wave_slot) & 3) {  // (s_setprio takes an immediate)
      case 0: __builtin_amdgcn_s_setprio(0); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      default: __builtin_amdgcn_s_setprio(3); break;
    }
    const StepArgs& sa = step_args();  // re-read per step: see step_args()
    const size_t esz = sa.cdtype == TC_F32 ? 4 : 8;
    const size_t row0 = (size_t)k * sa.a.N;
    const int seg_row = sa.ma.seg_rows > 1 ? k : 0;
    const int tid = step_lane();
    MapCache<K> mc = {};  // (initialised: a path that leaves it unset would otherwise make it a loop-carried value -- 30
                          // registers per lane held across the whole step body, raster stage included)
    FramePose fp;
    sim_body<K>(sa.a, smem, env, sa.mode, (const char*)sa.car_control + row0 * 2 * esz, sa.cdtype, sa.maneuver + row0,
                sa.spawn_nodes, sa.mask, sa.flags, roll_at(sa.ma.roll, row0, sa.a.m.C), tid, mc, fp);
    if (sa.ma.pose_rows) {
      double pose[12];
      cam_pose12(sa.a, env, fp, pose);
      if (tid == 0) {
        double2* o = (double2*)(step_args().ma.pose_rows + (row0 + env) * TC_POSE_ROW);
#pragma unroll
        for (int i = 0; i < 6; i++) o[i] = make_double2(pose[2 * i], pose[2 * i + 1]);
      }
    }
    if (CAM && sa.ma.cam_here) {
      if (wants_frame(sa)) {
        int nseg;
        unsigned int used;
        double pose[12];
        cam_pose12(sa.a, env, fp, pose);
        cam_body<K>(sa.a, smem, env, pose, mc, sa.mode != MODE_RENDER, tid, seg_row, nseg, used);  // (tc_render skips phase B's fetch)
      }
      else if (tid == 0)
        step_args().a.seg_n[(size_t)seg_row * sa.a.N + env] = 0;
    }
    lds_sync();  // the next step reads the LiveLds record this one wrote
  }
  TSTAMP_END(t_prev);
  const StepArgs& s1 = step_args();
  if (s1.mode != MODE_RENDER) live_out(s1.a, smem, env);
}

// ---------------------------------------------------------------------------------------------
// Grouped simulate kernel for K-step calls: TC_EL lanes per env, 64 / TC_EL envs per wavefront.
// Phases A and B of a step are ~900 + ~900 wave-instructions in the one-wavefront-per-env kernel above, and A is scalar
// work that all 64 lanes repeat.  The chip is bound by vector issue (counters: the frame kernel keeps the VALUs 96 %
// busy, the per-env simulate kernel 71 %), so what a step COSTS is its instruction count.  Here an instruction of phase
// A serves 64 / TC_EL envs at once (control flow may differ between envs: plain SIMT divergence), and phase B gives each
// lane of a group every TC_EL-th edge of a layer -- two distances per edge straight from the map (L1-resident), no
// staging of node distances, an xor butterfly over the group for the argmin -- with the per-layer tail evaluated by
// every lane of the group, so nothing has to be passed around afterwards.  State lives in registers for the whole
// launch.  Per env-step: ~2 300 wave-instructions / 8 envs instead of ~1 700 / 1 env.
// The price is latency (a group of 8 lanes walks 33 edges per layer pass one after the other), which nobody waits for:
// the kernel leaves most of the chip's issue slots free for the frame kernel of the previous call.
#define TC_EL 8
#ifndef TC_ENVG_NT
#define TC_ENVG_NT 256  // threads per workgroup: 4 wavefronts = 32 envs share one LDS copy of the edge records
#endif
#ifndef TC_ENVG_PRIO
#define TC_ENVG_PRIO 3
#endif
struct GroupLds {  // per env of the wavefront: what the reward / termination terms read and count
  double dist[TC_MAX_LAYERS];
  int cnt[TC_MAX_TERMS];
};

// Where the edge records are read from matters when the kernel runs beside the frame kernel: a frame wavefront ends
// with ~20 KB of 16-byte stores, and a global load of this kernel issued on the same CU queues behind those bursts in
// the CU's vector-memory pipeline (measured: 22 us per step alone, 35 us beside the frame kernel, 23 us beside a frame
// kernel with its stores switched off).  The edge scan is ~33 loads per lane and step: from LDS it neither waits for
// the stores nor pays an L2 round trip per batch of loads.
typedef const __attribute__((address_space(3))) double* LdsDouble;
__global__ __launch_bounds__(TC_ENVG_NT) void tc_envg_kernel(StepArgs sa_unused) {
  __shared__ GroupLds glds[TC_ENVG_NT / TC_EL];
  extern __shared__ __align__(16) unsigned char gsm[];
  // Highest issue priority: when this kernel shares the chip with the frame kernel of the previous chunk it is the
  // critical path (a serial chain per step, few instructions), and the frame wavefronts would otherwise crowd it out
  // of the vector issue slots by sheer number (measured: 36 us per step beside them, 21 us alone).
  __builtin_amdgcn_s_setprio(TC_ENVG_PRIO);
  const StepArgs& s0 = step_args();
  const int lane = threadIdx.x, sub = lane & (TC_EL - 1), grp = lane / TC_EL;
  if (s0.ma.resident && lane == 0) __hip_atomic_fetch_add(s0.ma.resident, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int N = s0.a.N;
  int env = s0.a.env0 + blockIdx.x * (TC_ENVG_NT / TC_EL) + grp;
  const bool map_lds = s0.ma.map_lds != 0;
  const LdsDouble l_exy = (LdsDouble)gsm;  // [total_edges][4]: n0.x, n0.y, n1.x, n1.y
  const LdsDouble l_ofw = (LdsDouble)(gsm + (size_t)s0.a.m.total_edges * 32);
  const LdsDouble l_orv = l_ofw + s0.a.m.total_edges;
  const bool fat_lds = s0.ma.fat_lds != 0;
  const size_t fat_off = map_lds ? ((size_t)s0.a.m.total_edges * 48 + 15) / 16 * 16 : 0;
  const FatLds fl = {(const __attribute__((address_space(3))) int*)(gsm + fat_off)};
  if (fat_lds) {  // the table word by word, consecutive lanes on consecutive words (once per launch)
    const int* src = (const int*)s0.a.m.lp_fat;
    __attribute__((address_space(3))) int* dst = (__attribute__((address_space(3))) int*)(gsm + fat_off);
    const int nw = s0.a.m.lpN * (int)(sizeof(LpNode) / 4);
    for (int i = lane; i < nw; i += TC_ENVG_NT) dst[i] = src[i];
    static_assert(sizeof(LpNode) % 16 == 0, "fat node records keep 16-byte alignment in LDS");
  }
  if (map_lds) {
    const DevMap& m0 = s0.a.m;
    __attribute__((address_space(3))) double* w4 = (__attribute__((address_space(3))) double*)gsm;
    __attribute__((address_space(3))) double* wd = w4 + (size_t)m0.total_edges * 4;
    for (int i = lane; i < m0.total_edges; i += TC_ENVG_NT) {
      const double4 q = m0.edge_xy[i];
      w4[4 * i] = q.x;
      w4[4 * i + 1] = q.y;
      w4[4 * i + 2] = q.z;
      w4[4 * i + 3] = q.w;
      wd[i] = m0.ori_fwd[i];
      wd[m0.total_edges + i] = m0.ori_rev[i];
    }
  }
  if (map_lds || fat_lds) __syncthreads();
  const bool live = env < N;  // groups past the last env compute on a copy of it and store nothing
  env = live ? env : N - 1;
  GroupLds& gl = glds[grp];
  CarState s;
  int nr = 0, cursor = 0, cursor0 = 0;
  bool have_trig = false;
  {
    const tc_buffers& b = s0.a.b;
    s.x = b.x[env];
    s.y = b.y[env];
    s.theta = b.theta[env];
    s.velocity = b.velocity[env];
    s.steering = b.steering[env];
    s.radius = b.radius[env];
    s.front_x = b.front_x[env];
    s.front_y = b.front_y[env];
    s.cth = s.sth = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s.lp[i] = b.local_path[env * 8 + i];
    s.lp_len = b.lp_len[env];
    s.last_maneuver = b.last_maneuver[env];
    if (s0.flags & TC_F_AUTORESET) {
      nr = b.needs_reset[env];
      cursor = cursor0 = b.spawn_cursor[env];
    }
    if (sub < TC_MAX_TERMS) gl.cnt[sub] = (s0.a.n_terms > 0 && s0.a.term_counters) ? s0.a.term_counters[(size_t)env * TC_MAX_TERMS + sub] : 0;
    static_assert(TC_EL >= TC_MAX_TERMS, "one lane per term counter");
  }
  double cte = 0, he = 0, reward = 0;
  int terminated = 0, trunc = 0, status = 0;
  const int nsteps = s0.ma.nsteps;
  // The action of step k+1 is fetched at the top of step k.  Vector-memory operations retire in issue order, so a load
  // issued at the top of a step would come back behind the rollout / pose-row stores of the step before -- each a round
  // trip through a memory pipeline the frame kernel keeps full of 16-byte stores.  Issued a whole step early, the load
  // has long returned when it is used, and the stores behind it are not waited for (raw bits are kept: converting at
  // once would be a wait).
  double2 act_d = make_double2(0.0, 0.0);
  float2 act_f = make_float2(0.f, 0.f);
  int act_m = 0;
  auto fetch_action = [&](int kk) {
    const StepArgs& sq = step_args();
    const size_t r = (size_t)kk * sq.a.N + env;
    if (sq.cdtype == TC_F32)
      act_f = ((const float2*)sq.car_control)[r];
    else
      act_d = ((const double2*)sq.car_control)[r];
    act_m = sq.maneuver[r];
  };
  fetch_action(0);
  for (int k = 0; k < nsteps; k++) {
    const StepArgs& sa = step_args();  // re-read per step: see step_args()
    const KArgs& a = sa.a;
    const DevMap& m = a.m;
    const tc_buffers& b = a.b;
    const unsigned int flags = sa.flags;
    const size_t row0 = (size_t)k * a.N;
    const RollStep roll = roll_at(sa.ma.roll, row0, a.m.C);
    status = 0;
    trunc = 0;
    // this step's action (fetched a step ago), and the next step's on its way (the last step re-reads its own row)
    const double2 cur_d = act_d;
    const float2 cur_f = act_f;
    const int cur_m = act_m;
    fetch_action(k + 1 < nsteps ? k + 1 : k);
    TSTAMP(24);
    PathInfo pinfo;
    pinfo.ax = pinfo.ay = pinfo.bx = pinfo.by = pinfo.ori = 0;
    pinfo.valid = 0;
    bool fresh = false, track = false;
    int track_man = 0;
    if ((flags & TC_F_AUTORESET) && nr) {
      const int cur = cursor;
      int node;
      if ((flags & TC_F_DEVICE_SPAWN) && a.spawn_n > 0) {
        node = a.spawn_tab[tc_spawn_index(a.spawn_seed, (uint32_t)env, (uint32_t)cur, (uint32_t)a.spawn_n)];
      } else {
        if ((unsigned)cur >= (unsigned)b.spawn_queue_len) status |= TC_S_SPAWN_WRAPPED;
        node = b.spawn_queue[(size_t)env * b.spawn_queue_len + ((unsigned)cur % (unsigned)b.spawn_queue_len)];
      }
      d_reset(m, a.car, s, checked_spawn(m, node, status));
      fresh = true;
      have_trig = true;
      cursor = cur + 1;
    } else if ((unsigned)s.lp[0] >= (unsigned)m.lpN || (unsigned)s.lp[1] >= (unsigned)m.lpN) {
      status |= TC_S_NOT_RESET;
      trunc = 1;
    } else {
      double v, st;
      if (sa.cdtype == TC_F32) {
        v = (double)cur_f.x;
        st = (double)cur_f.y;
      } else {
        v = cur_d.x;
        st = cur_d.y;
      }
      v = d_np_clip(v, -1.0, 1.0);  // env.py:118
      st = d_np_clip(st, -1.0, 1.0);
      const int man = cur_m;
      TSTAMP(25);
      d_car_kinematics(a.car, s, v, st, have_trig);
      TSTAMP(26);
      have_trig = true;
      track = true;
      track_man = man;
    }
    // ---- this step's pose row, as early as the pose is final (nothing below moves the car): in a streamed call the frame
    // workgroups of this (step, env) are waiting for it, and what follows -- lanepath tracking, distances, reward, rollout
    // rows -- is most of the step
    if (sa.ma.pose_rows) {
      // car.py:159-165 takes cos(-theta), sin(-theta): tc_cos is exactly even and tc_sin exactly odd, so the values of
      // the front-axle update are reused bit for bit (as in sim_body)
      FramePose fp;
      fp.x = s.x;
      fp.y = s.y;
      fp.cth = have_trig ? s.cth : tc_cos(-s.theta);
      fp.sth = have_trig ? -s.sth : tc_sin(-s.theta);
      double pose[12];
      cam_pose12(sa.a, live ? env : 0, fp, pose);
      if (live && sub == 0) {
        if (sa.ma.resident) {
          // streamed call: the frame workgroup of this (step, env) may already be polling the row.  Each entry is one
          // 8-byte device-scope store (sc1: written through to where every XCD sees it) and validates itself -- the
          // reader waits until none of the twelve holds TC_POSE_EMPTY any more -- so the stores need no order among
          // themselves and this wavefront waits for none of them.
          unsigned long long* o = (unsigned long long*)(sa.ma.pose_rows + (row0 + env) * TC_POSE_ROW);
#pragma unroll
          for (int i = 0; i < 12; i++)
            __hip_atomic_store(o + i, (unsigned long long)__double_as_longlong(pose[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          double2* o = (double2*)(sa.ma.pose_rows + (row0 + env) * TC_POSE_ROW);
#pragma unroll
          for (int i = 0; i < 6; i++) o[i] = make_double2(pose[2 * i], pose[2 * i + 1]);
        }
      }
    }
    if (track) {  // car.py:127-148 (the env went through the kinematic update above)
      if (fat_lds) {
        trunc = d_find_local_path<TC_EL>(m, fl, s, track_man, status, pinfo, sub);
      } else {
        const FatGlobal fg = {m.lp_fat, m.lp_nodes};
        trunc = d_find_local_path<TC_EL>(m, fg, s, track_man, status, pinfo, sub);
      }
    }
    TSTAMP(27);
    // ---- info (car.py:46-53), default reward / termination (env.py:93,99)
    const bool have_info = !fresh && s.lp_len >= 2;
    cte = 0;
    he = 0;
    if (have_info && !pinfo.valid) {  // path kept from an earlier step (truncated before it was rebuilt)
      double2 n1 = m.lp_nodes[s.lp[2]], n2 = m.lp_nodes[s.lp[3]];
      pinfo.ax = n1.x;
      pinfo.ay = n1.y;
      pinfo.bx = n2.x;
      pinfo.by = n2.y;
      pinfo.ori = d_lp_edge_ori(m, s.lp[2], s.lp[3]);
    }
    if (have_info) {
      cte = d_distance_to_edge(pinfo.ax, pinfo.ay, pinfo.bx, pinfo.by, s.front_x, s.front_y);
      he = d_clip_angle(pinfo.ori - s.theta);
    }
    reward = 0;
    terminated = 0;
    if (!(flags & TC_F_WRAPPED) && !fresh) {
      double r = (-1 / a.car.track_width) * cte + 1;
      reward = (0 > r) ? 0 : r;
      terminated = cte > (a.car.track_width * 10);
    }
    // ---- phase B: nearest lane-line edge and distance per layer (car.py:55-64, layer.py:33-44,126-164)
    const int C = m.C;
    const bool last = k == nsteps - 1;
    const int cell = have_info ? d_grid_cell(m, s.x, s.y) : -1;
    TSTAMP(0);
    static_assert(TC_EL == 8 && TC_AG <= TC_EL, "group8_argmin_multi, one lane per layer of a block");
    if (have_info) {
      // The cell's candidate edges (DevMap: every edge that can be the first minimum for a point of the cell, layer by
      // layer, ascending -- a handful per layer), a block of TC_AG layers at a time.  Memory first, arithmetic after: the
      // block's list offsets are fetched together, then each lane's first candidate of every layer together (two load
      // latencies for the whole block, whatever the vector-memory pipeline's queue looks like beside the frame kernel),
      // then the layers are scanned one after the other from the LDS edge records; lane g of the group finally evaluates
      // layer g's tail, so the bounds test and the distance run once per block, not once per layer.
      //   A car outside the grid (cell < 0) runs the SAME code with the identity list -- every edge of the layer -- so a
      // wavefront with one such env executes longer loops, not a second copy of the phase.
      const bool far = cell < 0;
      for (int lb = 0; lb < C; lb += TC_AG) {
        double bd[TC_AG];
        int best[TC_AG], lo[TC_AG], off[TC_AG + 1], e0[TC_AG];
#pragma unroll
        for (int g = 0; g <= TC_AG; g++) {
          const int lg = lb + g < C ? lb + g : C;
          off[g] = far ? m.edge_off[lg] : m.cand_off[cell * C + lg];
        }
#pragma unroll
        for (int g = 0; g < TC_AG; g++) {
          lo[g] = lb + g < C ? m.edge_off[lb + g] : 0;
          e0[g] = off[g] + sub < off[g + 1] ? (far ? off[g] + sub : m.cand_idx[off[g] + sub]) : -1;
        }
#pragma unroll
        for (int g = 0; g < TC_AG; g++) {
          double bdg = 0;
          int bg = -1, e = e0[g];
          for (int i = off[g] + sub; i < off[g + 1]; i += TC_EL) {
            double4 q;
            if (map_lds)
              q = make_double4(l_exy[4 * e], l_exy[4 * e + 1], l_exy[4 * e + 2], l_exy[4 * e + 3]);
            else
              q = m.edge_xy[e];
            const double d = tc_fabs(d_dist(s.x, s.y, q.x, q.y) + d_dist(s.x, s.y, q.z, q.w));
            if (bg < 0 || d < bdg) {
              bg = e - lo[g];
              bdg = d;
            }
            if (i + TC_EL < off[g + 1]) e = far ? i + TC_EL : m.cand_idx[i + TC_EL];
          }
          bd[g] = bdg;
          best[g] = bg;
        }
        if (lb == 0) TSTAMP(16);
        group8_argmin_multi(bd, best);
        if (lb == 0) TSTAMP(1);
        int ne = -1, eo = 0;
#pragma unroll
        for (int g = 0; g < TC_AG; g++) {
          ne = sub == g ? best[g] : ne;
          eo = sub == g ? lo[g] : eo;
        }
        const int l = lb + sub;
        if (sub < TC_AG && l < C) {  // lane g: layer lb + g
          double dist_l = 0;
          if (ne >= 0) {
            const int ge = eo + ne;
            double4 q;
            double o_fw, o_rv;
            if (map_lds) {
              q = make_double4(l_exy[4 * ge], l_exy[4 * ge + 1], l_exy[4 * ge + 2], l_exy[4 * ge + 3]);
              o_fw = l_ofw[ge];
              o_rv = l_orv[ge];
            } else {
              q = m.edge_xy[ge];
              o_fw = m.ori_fwd[ge];
              o_rv = m.ori_rev[ge];
            }
            const double2 n0 = make_double2(q.x, q.y), n1 = make_double2(q.z, q.w);
            bool certain;
            bool inb = d_within_bounds_filter(n0.x, n0.y, n1.x, n1.y, s.x, s.y, certain);
            if (!certain) inb = d_within_bounds(n0.x, n0.y, n1.x, n1.y, o_fw, o_rv, s.x, s.y);
            if (inb) {
              dist_l = tc_fabs(d_distance_to_edge(n0.x, n0.y, n1.x, n1.y, s.x, s.y));
            } else {
              const double da = d_dist(s.x, s.y, n0.x, n0.y);
              const double db = d_dist(s.front_x, s.front_y, n1.x, n1.y);  // FRONT axle for n1 (car.py:64)
              dist_l = db < da ? db : da;
            }
          }
          gl.dist[l] = dist_l;
          if (live) {
            if (roll.dist) roll.dist[(roll.row0 + env) * C + l] = dist_l;
            if (roll.ne) roll.ne[(roll.row0 + env) * C + l] = ne;
          }
          if (last && live) {
            b.laneline_distances[(size_t)env * C + l] = dist_l;
            b.nearest_edge[(size_t)env * C + l] = ne;
          }
        }
        if (lb == 0) TSTAMP(2);
      }
    } else {
      for (int l = sub; l < C; l += TC_EL) {  // no info this step (car.py:47-51): zero distances, no nearest edge
        gl.dist[l] = 0;
        if (live) {
          if (roll.dist) roll.dist[(roll.row0 + env) * C + l] = 0;
          if (roll.ne) roll.ne[(roll.row0 + env) * C + l] = -1;
        }
        if (last && live) {
          b.laneline_distances[(size_t)env * C + l] = 0;
          b.nearest_edge[(size_t)env * C + l] = -1;
        }
      }
    }
    TSTAMP(14);
    // ---- reward / termination wrappers (a re-spawned env did not go through Wrapper.step)
    if (a.n_terms > 0 && !fresh)
      d_apply_terms_mem(a.terms, a.n_terms, gl.cnt, a.car.track_width, C, cte, have_info ? s.velocity : 0.0, gl.dist, reward,
                        terminated);
    nr = (flags & TC_F_AUTORESET) ? (terminated || trunc) : 0;
    // ---- this step's rollout rows and pose row
    if (live && sub == 0) {
      if (roll.cte) roll.cte[roll.row0 + env] = cte;
      if (roll.heading_error) roll.heading_error[roll.row0 + env] = he;
      if (roll.truncated) roll.truncated[roll.row0 + env] = (unsigned char)trunc;
      if (roll.reward) roll.reward[roll.row0 + env] = reward;
      if (roll.terminated) roll.terminated[roll.row0 + env] = (unsigned char)terminated;
      if (roll.status) roll.status[roll.row0 + env] = status;
      if (roll.x) roll.x[roll.row0 + env] = s.x;
      if (roll.y) roll.y[roll.row0 + env] = s.y;
      if (roll.theta) roll.theta[roll.row0 + env] = s.theta;
      if (roll.velocity) roll.velocity[roll.row0 + env] = s.velocity;
      if (roll.lp_len) roll.lp_len[roll.row0 + env] = s.lp_len;
      if (roll.lp) {
        int4* o = (int4*)(roll.lp + (roll.row0 + env) * 8);
        o[0] = make_int4(s.lp[0], s.lp[1], s.lp[2], s.lp[3]);
        o[1] = make_int4(s.lp[4], s.lp[5], s.lp[6], s.lp[7]);
      }
    }
    TSTAMP(15);
  }
  // ---- state and the last step's outputs back to the caller's buffers
  const StepArgs& s1 = step_args();
  const tc_buffers& b = s1.a.b;
  if (live && sub == 0) {
    b.x[env] = s.x;
    b.y[env] = s.y;
    b.theta[env] = s.theta;
    b.velocity[env] = s.velocity;
    b.steering[env] = s.steering;
    b.radius[env] = s.radius;
    b.front_x[env] = s.front_x;
    b.front_y[env] = s.front_y;
#pragma unroll
    for (int i = 0; i < 8; i++) b.local_path[env * 8 + i] = s.lp[i];
    b.lp_len[env] = s.lp_len;
    b.last_maneuver[env] = s.last_maneuver;
    b.cte[env] = cte;
    b.heading_error[env] = he;
    b.reward[env] = reward;
    b.terminated[env] = (unsigned char)terminated;
    b.truncated[env] = (unsigned char)trunc;
    b.status[env] = status;
    if (b.needs_reset) b.needs_reset[env] = (unsigned char)nr;
    if (cursor != cursor0) b.spawn_cursor[env] = cursor;
  }
  if (live && s1.a.term_counters && sub < s1.a.n_terms) s1.a.term_counters[(size_t)env * TC_MAX_TERMS + sub] = gl.cnt[sub];
}

// One (step, env) frame per workgroup: camera stage from the pose the simulate launch left, then the raster stage.
// Frames do not depend on each other, a launch has steps x N of them -- many more than the chip holds at once -- and
// their cost varies 8-fold with what is in view, so the dispatcher's hand-out of the next frame to the next free slot is
// what balances the load (a wavefront that keeps an env for the whole launch inherits that env's view instead).
struct FrameArgs {
  KArgs a;
  RArgs r;
  const double* pose_rows;  // [rows][N][TC_POSE_ROW]
  const int* order;         // workgroup x of every grid row draws env order[x] (heaviest frames first: see launch()), or NULL
  // Streamed call (launch()): this kernel runs BESIDE the simulate launch that produces its pose rows.
  int gate;                 // 0: the rows are complete (written by an earlier launch); 1: wait for the row (bounded: gate_ticks);
                            // 2: tc_frame_recover_kernel, behind the simulate launch: draws what a gate-1 workgroup gave up on
  int recover_rows;         // gate 2: rows of the call
  unsigned int* abort_word; // gate 1: set by the first workgroup whose wait ran out; the others then give up at once
  unsigned int* resident;   // gate 2: workgroup 0 clears this and abort_word for the next call
  long long gate_ticks;     // gate 1: how long a workgroup waits for its row, in 100 MHz ticks
  int gate_test;            // tests only: gate-1 workgroups with (row + env) % gate_test == 0 give up without waiting
};
// The argument block is read through a pointer the compiler cannot see through, once per stage: the stage's values are
// then loaded (s_load from the kernarg segment) where they are used instead of all being fetched at kernel entry and
// kept in -- or spilled from -- scalar registers across both stages.
typedef const __attribute__((address_space(4))) FrameArgs* FrameArgsConst;
__device__ __forceinline__ const FrameArgs& frame_args() {
  unsigned long long p = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return *(const FrameArgs*)(FrameArgsConst)p;
}
// The pose row of frame (slot0 + env).  In a streamed call (fa.gate) the row may not exist yet: lanes 0..11 read one entry
// each with device-scope loads (sc1: neither this CU's caches nor this XCD's L2 may answer with an older copy; the scalar
// cache is out of the question) until none holds TC_POSE_EMPTY; the values then move to scalar registers.  Rows are
// dispatched in step order and the simulate launch is resident before the frame kernel starts (tc_gate_kernel), so the
// wait ends -- but nothing relies on that: a wait that outlasts gate_ticks marks the frame skipped, tells the others and
// returns false, and tc_frame_recover_kernel, behind the simulate launch, draws the frame.  `wait` false: the row is known
// to be there (gate 2, or read before by this workgroup) -- one read.
__device__ __forceinline__ bool frame_pose(const size_t slot0, const int env, const int row, const bool wait, double* pose) {
  const FrameArgs& fa = frame_args();
  const int tid = threadIdx.x;
  if (fa.gate) {
    unsigned long long* prow = (unsigned long long*)(fa.pose_rows + (slot0 + env) * TC_POSE_ROW);
    unsigned long long v = 0;
    bool give_up = wait && fa.gate == 1 && fa.gate_test > 0 && (row + env) % fa.gate_test == 0;
    long long t0 = 0;
    while (!give_up) {
      v = tid < 12 ? __hip_atomic_load(prow + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
      if (__ballot(v == TC_POSE_EMPTY) == 0 || fa.gate == 2 || !wait) break;
      const long long now = wall_clock64();
      if (t0 == 0) t0 = now;
      if (now - t0 > fa.gate_ticks || __hip_atomic_load(fa.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
        give_up = true;
      else
        __builtin_amdgcn_s_sleep(8);
    }
    if (give_up) {
      if (tid == 0) {
        if (t0 != 0) __hip_atomic_store(fa.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (not the test's skips)
        fa.a.seg_n[slot0 + env] = TC_FRAME_SKIPPED;
      }
      return false;
    }
    const unsigned int vlo = (unsigned int)v, vhi = (unsigned int)(v >> 32);
#pragma unroll
    for (int i = 0; i < 12; i++) {
      const unsigned long long q = (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)vlo, i) |
                                   ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)vhi, i) << 32);
      pose[i] = __longlong_as_double((long long)q);
    }
  } else {
    // written by an earlier launch, complete and visible before this kernel started: one address for the whole wavefront,
    // read through the scalar cache
    const __attribute__((address_space(4))) double* pr =
        (const __attribute__((address_space(4))) double*)(unsigned long long)(fa.pose_rows + (slot0 + env) * TC_POSE_ROW);
#pragma unroll
    for (int i = 0; i < 12; i++) pose[i] = pr[i];
  }
  return true;
}

// One frame: `rowy` is the frame's row in the launch (its step), `env` its env.
// HOT (tc_frame_kernel): the raster stage in its LDS-only form (see raster_body); !HOT (the recover kernel): the generic form.
template <int K, bool THICK, int FMT, bool HOT, int RBT>
__device__ __forceinline__ void frame_one(unsigned char* smem, const int env, const int rowy) {
  const int tid = threadIdx.x;
  int nseg;
  unsigned int used;
  {
    const FrameArgs& fa = frame_args();
    const int row = fa.r.seg_row0 + rowy;
    const size_t slot0 = (size_t)row * fa.a.N;
    double pose[12];
    if (!frame_pose(slot0, env, row, true, pose)) return;
    MapCache<K> mc;
    TSTAMP_CLEAR();
    TSTAMP(3);
    TSTAMP_REAL(30);
    cam_body<K>(fa.a, smem, env, pose, mc, false, tid, row, nseg, used, false);
  }
  const FrameArgs& fr = frame_args();
  const size_t slot0 = (size_t)(fr.r.seg_row0 + rowy) * fr.a.N;
  if (__builtin_expect(nseg > fr.a.seg_lds_cap, 0)) {  // (wave-uniform; cfg3: more than 56 segments)
    if (HOT) {
      // the list outgrew the LDS head: its tail is in the global array already, the head follows, and the raster stage
      // takes the whole list from there batch by batch
      lds_sync();
      const LdsIntPtr sl = (LdsIntPtr)(smem + fr.a.seg_lds_off);
      int* sg = fr.a.seg_g + (slot0 + env) * fr.a.seg_cap * 5;
      for (int i = tid; i < 5 * fr.a.seg_lds_cap; i += TC_NT) sg[i] = sl[i];
    }
    __syncthreads();  // draw-list entries that went through global memory are visible to this wavefront (vmcnt(0) + barrier)
  } else {
    lds_sync();
  }
  raster_body<THICK, FMT, HOT, RBT>(fr.r, smem, env, fr.r.obs + (size_t)rowy * fr.r.obs_row_stride, tid, slot0, nseg, used,
                                           fr.r.noise_row0 + rowy);
  {
    const FrameArgs& fe = frame_args();
    const size_t slot = (size_t)(fe.r.seg_row0 + rowy) * fe.a.N + env;
    if (tid == 0) fe.a.seg_n[slot] = nseg;  // workload statistics, the next dispatch's order, "drawn" for the recover pass
    // a streamed call's row goes back to "not written yet" for the next call (which starts behind this kernel)
    if (fe.gate && tid < 12)
      __hip_atomic_store((unsigned long long*)(fe.pose_rows + slot * TC_POSE_ROW) + tid, TC_POSE_EMPTY, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
  }
  TSTAMP_DUMP(env);
}

// K: slots of the camera stage's register cache per lane (a camera group has at most 64 K nodes / edges); RBT: segments per
// raster batch.  K = 5 with RBT = 16 is the variant of maps whose camera groups are packs of connected components (knuffingen:
// the map needs K = 9 as a whole or layer by layer, its component groups fit K = 5).
template <int K, bool THICK, int FMT, int RBT = RB_OF_K(K)>
__global__ __launch_bounds__(TC_NT, (K <= 5 ? 4 : K <= 9 ? 3 : 2)) void tc_frame_kernel(FrameArgs fa_unused) {
  extern __shared__ __align__(16) unsigned char smem[];
  const FrameArgs& fa = frame_args();
  if ((int)blockIdx.x >= fa.a.N) return;
  const int env = fa.order ? uni_i(((const __attribute__((address_space(4))) int*)(unsigned long long)fa.order)[blockIdx.x])
                           : fa.a.env0 + (int)blockIdx.x;
  frame_one<K, THICK, FMT, true, RBT>(smem, env, (int)blockIdx.y);
}

// The pass behind a streamed call's simulate launch: one workgroup per env looks through the call's rows for frames a
// gated workgroup gave up on (TC_FRAME_SKIPPED; normally none: one strided read per row block and out) and draws them
// itself, one after the other, with gate = 2 (the row is there: read once, no waiting) and the generic form of both
// stages.  Workgroup 0 resets the call's two words.
// Never on the hot path, so what the step loop around a ~10 k-instruction body costs in registers does not matter.
template <int K, bool THICK, int FMT, int RBT = RB_OF_K(K)>
__global__ __launch_bounds__(TC_NT) void tc_frame_recover_kernel(FrameArgs fa_unused) {
  extern __shared__ __align__(16) unsigned char smem[];
  const FrameArgs& fa = frame_args();
  const int env = (int)blockIdx.x, tid = threadIdx.x;
  if (env >= fa.a.N) return;
  if (fa.gate && env == 0 && tid == 0) {
    __hip_atomic_store(fa.abort_word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(fa.resident, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int rows = fa.recover_rows;
  bool any = false;  // (TC_FRAME_SKIPPED: the only negative length)
  for (int r = tid; r < rows; r += TC_NT) any |= ((const volatile int*)fa.a.seg_n)[(size_t)(fa.r.seg_row0 + r) * fa.a.N + env] < 0;
  if (__ballot(any) == 0) return;
  for (int r = 0; r < rows; r++) {
    const FrameArgs& fb = frame_args();
    if (uni_i(((const volatile int*)fb.a.seg_n)[(size_t)(fb.r.seg_row0 + r) * fb.a.N + env]) >= 0) continue;
    frame_one<K, THICK, FMT, false, RBT>(smem, env, r);
    __syncthreads();  // the next frame reuses the LDS
  }
}

// All stages in one launch: the same wavefront simulates its env, runs the camera and rasterises the frame.  The form
// of tc_step / tc_reset / tc_render (one step: nothing to balance, and one kernel boundary less), and of tc_step_multi
// under TC_MULTI_SPLIT=0.
template <int K, bool THICK, int FMT, int RBT = RB_OF_K(K)>
__global__ __launch_bounds__(TC_NT, (K <= 5 ? 4 : K <= 9 ? 3 : 2)) void tc_step_kernel(StepArgs sa_unused) {
  extern __shared__ __align__(16) unsigned char smem[];
  const StepArgs& s0 = step_args();
  if ((int)blockIdx.x >= s0.a.N) return;
  // which env this workgroup works on: see tc_order_kernel (a scalar load: one address for the whole wavefront)
  const int env = s0.env_order ? uni_i(((const __attribute__((address_space(4))) int*)(unsigned long long)s0.env_order)[blockIdx.x])
                               : s0.a.env0 + (int)blockIdx.x;
  if (s0.mode == MODE_RESET && s0.mask && !s0.mask[env]) return;  // whole workgroup skips
  live_in(s0.a, smem, env, s0.mode, s0.flags);
  const int nsteps = s0.ma.nsteps;
  long long t_prev = 0;
#ifdef TC_TIMING
  t_prev = clock64();
#endif
  (void)t_prev;
  for (int k = 0; k < nsteps; k++) {
    TSTAMP_LOOP(k, nsteps, t_prev);
    const StepArgs& sa = step_args();  // re-read per step: see step_args()
    const size_t esz = sa.cdtype == TC_F32 ? 4 : 8;
    const size_t row0 = (size_t)k * sa.a.N;
    const int tid = step_lane();
    MapCache<K> mc = {};  // (initialised: a path that leaves it unset would otherwise make it a loop-carried value -- 30
                          // registers per lane held across the whole step body, raster stage included)
    FramePose fp;
    sim_body<K>(sa.a, smem, env, sa.mode, (const char*)sa.car_control + row0 * 2 * esz, sa.cdtype, sa.maneuver + row0,
                sa.spawn_nodes, sa.mask, sa.flags, roll_at(sa.ma.roll, row0, sa.a.m.C), tid, mc, fp);
    if (wants_frame(sa)) {
      int nseg;
      unsigned int used;
      double pose[12];
      cam_pose12(sa.a, env, fp, pose);
      cam_body<K>(sa.a, smem, env, pose, mc, sa.mode != MODE_RENDER, tid, 0, nseg, used, false);  // (tc_render skips phase B's fetch)
      if (nseg > sa.a.seg_lds_cap)
        __syncthreads();  // draw-list entries that went through global memory are visible to this wavefront (vmcnt(0) + barrier)
      else
        lds_sync();
      const StepArgs& sb = step_args();
      const size_t obs_step = sb.ma.roll.obs ? (size_t)sb.a.N * ((size_t)sb.r.cam.H * sb.r.cam.W * (FMT == TC_FMT_CLASSES ? sb.r.C : 3)) : 0;
      unsigned char* obs_base = sb.ma.roll.obs ? sb.ma.roll.obs : sb.r.obs;
      raster_body<THICK, FMT, false, RBT>(sb.r, smem, env, obs_base + (size_t)k * obs_step, tid, 0, nseg, used, k);
      if (tid == 0) step_args().a.seg_n[env] = nseg;  // workload statistics only
    } else {
      lds_sync();  // the next step reads the LiveLds record this one wrote
    }
  }
  TSTAMP_END(t_prev);
  const StepArgs& s1 = step_args();
  if (s1.mode != MODE_RENDER) live_out(s1.a, smem, env);
}

// ---------------------------------------------------------------------------------------------
// Which env each workgroup of a single-step launch works on.  One tc_step is N one-wavefront workgroups, exactly one
// resident round on the chip, so it lasts as long as its busiest SIMD -- and the hardware puts workgroups w, w + G,
// w + 2G, ... (G = number of SIMDs: 1024) on the SAME SIMD, whichever SIMD that is in a given launch (measured,
// tools/hw_map.py: every SIMD holds env ids that differ by exactly 1024; sum of the four wavefront lives per SIMD:
// busiest / mean = 1.4-1.6, because an env's frame costs between ~10 k and ~130 k clocks and keeps its kind of view for
// many steps).  The cost of a frame is known well enough from the frame before it (its draw-list length; the wavefront's
// measured life as the key was tried and is worse than no re-deal at all: it carries the contention of the previous deal), so the envs are
// sorted by that and dealt to the G groups in serpentine order -- heaviest with lightest -- and workgroup w takes env
// order[w].  Any permutation gives the same results (envs are independent); only the launch time changes.
// One workgroup: counting sort of the N lengths (descending), then rank r -> workgroup (r / G) * G + (r / G even ? r % G :
// G - 1 - r % G).
#define TC_ORDER_NT 1024
#define TC_ORDER_BINS 256
// (the keys are read ONCE, into LDS: the lengths may belong to a frame row another stream is about to redraw, and a key
// that changed between the counting and the placing pass would break the permutation)
__global__ __launch_bounds__(TC_ORDER_NT) void tc_order_kernel(const int* cost, int N, int G, int* order) {
  __shared__ int hist[TC_ORDER_BINS], cursor[TC_ORDER_BINS];
  extern __shared__ unsigned char okeys[];  // [N]
  const int t = threadIdx.x;
  for (int b = t; b < TC_ORDER_BINS; b += TC_ORDER_NT) hist[b] = 0;
  __syncthreads();
  for (int i = t; i < N; i += TC_ORDER_NT) {
    int c = cost[i];
    c = c < 0 ? 0 : (c > TC_ORDER_BINS - 1 ? TC_ORDER_BINS - 1 : c);
    okeys[i] = (unsigned char)(TC_ORDER_BINS - 1 - c);  // bin 0 = the heaviest
    atomicAdd(&hist[TC_ORDER_BINS - 1 - c], 1);
  }
  __syncthreads();
  {  // exclusive prefix sum over the bins: a DPP scan inside each of the first four wavefronts, their totals through LDS
    __shared__ int wtot[TC_ORDER_BINS / TC_NT];
    static_assert(TC_ORDER_BINS % TC_NT == 0 && TC_ORDER_BINS <= TC_ORDER_NT, "one thread per bin, whole wavefronts");
    int v = 0, inc = 0;
    if (t < TC_ORDER_BINS) {
      v = hist[t];
      inc = wave_incl_scan(v, t & (TC_NT - 1));
      if ((t & (TC_NT - 1)) == TC_NT - 1) wtot[t / TC_NT] = inc;
    }
    __syncthreads();
    if (t < TC_ORDER_BINS) {
      int base = 0;
      for (int w = 0; w < t / TC_NT; w++) base += wtot[w];
      cursor[t] = base + inc - v;
    }
  }
  __syncthreads();
  for (int i = t; i < N; i += TC_ORDER_NT) {
    const int r = atomicAdd(&cursor[okeys[i]], 1);  // rank among the envs, heaviest first
    const int row = r / G, j = r - row * G;
    order[row * G + ((row & 1) ? G - 1 - j : j)] = i;
  }
}

// Streamed call: holds the frame stream until every workgroup of the simulate launch on the caller's stream has started
// (they count themselves into *resident).  One wavefront, so it cannot keep them off the chip itself; the frame
// kernel behind it in stream order then finds its producers resident whatever it fills the chip with.  Bounded: after
// `ticks` (100 MHz) it lets the frame kernel go regardless, whose workgroups have a bound of their own.
__global__ __launch_bounds__(64) void tc_gate_kernel(const unsigned int* resident, unsigned int want, long long ticks) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(resident, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && wall_clock64() - t0 < ticks)
    __builtin_amdgcn_s_sleep(2);
}

typedef void (*fused_kern_t)(StepArgs);
// TC_DEV_FAST (make dev): only the K = 5 / thick / classes variants are instantiated -- a compile of seconds instead of
// minutes for kernel work on cfg3 (make dev DEVK=9 DEVFMT=TC_FMT_RGB: the variants of cfg5; DEVK=9 alone: cfg4).  Never
// shipped: the default build has no such macro.
#ifndef TC_DEV_KV
#define TC_DEV_KV 5
#endif
#ifndef TC_DEV_FMTV
#define TC_DEV_FMTV TC_FMT_CLASSES
#endif
#ifndef TC_DEV_SK   // K / batch size of the one fused step kernel variant compiled (cfg4, cfg5: 5 and 16, like TC_DEV_FK / TC_DEV_FRB)
#define TC_DEV_SK TC_DEV_KV
#endif
#ifndef TC_DEV_SRB
#define TC_DEV_SRB RB_OF_K(TC_DEV_SK)
#endif
template <int K, int RBT = RB_OF_K(K)>
static fused_kern_t pick_fused(bool thick, bool cls) {
#ifdef TC_DEV_FAST
  return tc_step_kernel<TC_DEV_SK, true, TC_DEV_FMTV, TC_DEV_SRB>;
#else
  return thick ? (cls ? tc_step_kernel<K, true, TC_FMT_CLASSES, RBT> : tc_step_kernel<K, true, TC_FMT_RGB, RBT>)
               : (cls ? tc_step_kernel<K, false, TC_FMT_CLASSES, RBT> : tc_step_kernel<K, false, TC_FMT_RGB, RBT>);
#endif
}

typedef void (*frame_kern_t)(FrameArgs);
#ifndef TC_DEV_FK   // dev builds: K / batch size of the one frame kernel variant compiled (cfg4, cfg5: DEVFK=5 DEVFRB=16)
#define TC_DEV_FK TC_DEV_KV
#endif
#ifndef TC_DEV_FRB
#define TC_DEV_FRB RB_OF_K(TC_DEV_FK)
#endif
template <int K, int RBT = RB_OF_K(K)>
static frame_kern_t pick_frame(bool thick, bool cls) {
#ifdef TC_DEV_FAST
  return tc_frame_kernel<TC_DEV_FK, true, TC_DEV_FMTV, TC_DEV_FRB>;
#else
  return thick ? (cls ? tc_frame_kernel<K, true, TC_FMT_CLASSES, RBT> : tc_frame_kernel<K, true, TC_FMT_RGB, RBT>)
               : (cls ? tc_frame_kernel<K, false, TC_FMT_CLASSES, RBT> : tc_frame_kernel<K, false, TC_FMT_RGB, RBT>);
#endif
}

template <int K, int RBT = RB_OF_K(K)>
static frame_kern_t pick_recover(bool thick, bool cls) {
#ifdef TC_DEV_FAST
  return tc_frame_recover_kernel<TC_DEV_FK, true, TC_DEV_FMTV, TC_DEV_FRB>;
#else
  return thick ? (cls ? tc_frame_recover_kernel<K, true, TC_FMT_CLASSES, RBT> : tc_frame_recover_kernel<K, true, TC_FMT_RGB, RBT>)
               : (cls ? tc_frame_recover_kernel<K, false, TC_FMT_CLASSES, RBT> : tc_frame_recover_kernel<K, false, TC_FMT_RGB, RBT>);
#endif
}
// by tc_env::kframe: the simulate variant's K, or 516 = K 5 with batches of 16
static frame_kern_t frame_kernel_of(int kframe, bool thick, bool cls) {
  return kframe == 516 ? pick_frame<5, 16>(thick, cls) : kframe == 5 ? pick_frame<5>(thick, cls) : kframe == 8 ? pick_frame<8>(thick, cls) : pick_frame<9>(thick, cls);
}
static frame_kern_t recover_kernel_of(int kframe, bool thick, bool cls) {
  return kframe == 516 ? pick_recover<5, 16>(thick, cls) : kframe == 5 ? pick_recover<5>(thick, cls) : kframe == 8 ? pick_recover<8>(thick, cls) : pick_recover<9>(thick, cls);
}

// =============================================================================================
// Host side: C ABI
// =============================================================================================
static thread_local std::string g_err;
static void set_err(const std::string& s) { g_err = s; }
extern "C" const char* tc_last_error(void) { return g_err.c_str(); }
extern "C" int tc_abi_version(void) { return TC_ABI_VERSION; }

#define HIP_TRY(expr)                                                                    \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      set_err(std::string(#expr) + ": " + hipGetErrorString(_e));                        \
      return TC_E_HIP;                                                                   \
    }                                                                                    \
  } while (0)

struct tc_map {
  DevMap d;
  std::vector<void*> allocs;
  int device;
  std::vector<double2> h_nodes;   // host copies of d.nodes / d.edges_g (tc_env_create groups the camera's copy by them)
  std::vector<int2> h_edges_g;
};

struct tc_env {
  const tc_map* map;
  KArgs k;
  bool bound;
  int64_t obs_bytes;
  int r_off_tab, r_off_bits, r_lds;
  int fuse;  // 1: simulate + raster in one launch (tc_step_kernel); 0: two launches
  // tc_step_multi with observations, split form (default; TC_MULTI_SPLIT=0 selects the fused K-step kernel): ONE
  // simulate launch loops over the K steps and leaves K draw lists per env, ONE raster launch of K x N workgroups
  // draws them.  Why: a frame costs between ~10 k clocks (nothing in view) and ~80 k (60 segments) to rasterise and an
  // env keeps its kind of view for many steps, so in the fused K-step kernel the launch lasts as long as its heaviest
  // env (measured on cfg3: 70.9 us per step against a mean of 41 us per wavefront-step).  Frames are independent of
  // each other, and K x N workgroups are many more than the chip holds at once, so the dispatcher balances them.
  int multi_split;
  int chunk;        // K-step calls with a rollout: steps per chunk (TC_CHUNK, default 16)
  int pipe;         // 1: the frame launches of a chunk run on an internal stream beside the next chunk's simulate launch
                    // (TC_CHUNK=0 switches that off: chunks of the default size follow each other on the caller's stream)
  hipStream_t frame_stream, frame_stream2;
  hipEvent_t sim_ev, frames_ev, frames_ev2;
  hipEvent_t slot_ev[TC_RING_SLOTS];  // frames of the chunk that last used ring slot s are done
  hipEvent_t call_ev;                 // end of the last K-step call on its stream
  hipStream_t last_stream;            // stream of that call
  bool have_call;
  // where the draw-list lengths of the most recent frames are (tc_env_draw_list_stats): up to TC_RING_SLOTS (first ring
  // row, rows) pairs of the last K-step call, or the per-env list of the last single step (draw_n < 0)
  int draw_rows[TC_RING_SLOTS][2];
  int draw_n;
  const int* draw_base;  // the [rows][N] array the pairs index: the ring's, or the streamed call's
  // Streamed K-step calls (TC_STREAM=0: chunked as above): ONE simulate launch for all steps of the call and ONE frame
  // launch beside it whose workgroups wait for their pose row -- see launch().  Scratch for st_rows steps
  // (tc_env_reserve_steps); longer calls run as segments of st_rows steps.
  int stream;            // 1 = on
  int st_rows;
  int* st_segm_g;        // [st_rows][N][seg_cap][5]
  int* st_segm_n;        // [st_rows][N]
  double* st_pose;       // [st_rows][N][TC_POSE_ROW], every entry TC_POSE_EMPTY between calls
  unsigned int* st_words;  // [0] simulate workgroups resident, [1] "a frame workgroup gave up waiting"
  long long gate_ticks;  // bound of a frame workgroup's wait (100 MHz ticks; TC_STREAM_WAIT_US)
  int gate_test;         // TC_STREAM_TEST_SKIP=m (tests): frames with (row + env) % m == 0 are left to the gate-2 pass
  hipEvent_t start_ev;
  int step_lds;  // tc_step_kernel: LDS bytes per workgroup (lds.total grown like frame_lds)
  // cost-aware env order of single-step launches (tc_order_kernel): refreshed every order_every-th tc_step
  // heaviest-first order of the frame workgroups of a K-step call (TC_FRAME_ORDER=0: env order), one buffer per frame stream
  int* frame_order[2];       // [N] each, device
  const int* cost_row[2];    // draw-list lengths of the last frame row launched on that stream (library scratch), or NULL
  bool frame_order_valid[2]; // the buffer holds a permutation (a sort has run on it)
  int* env_order;   // [N] device, a permutation (identity until the first refresh); NULL = off (TC_STEP_ORDER=0, N % G != 0)
  int order_g;      // G = SIMDs of the device
  int order_every, order_calls;
  int seg_lds_limit;  // TC_SEG_LDS_CAP
  int frame_lds, seg_lds_off, seg_lds_cap;  // tc_frame_kernel: LDS bytes per workgroup, draw-list region (see KArgs)
  void *cam_nodes_dev, *cam_edges_dev;  // the camera's copy of the map, grouped by connected components (or NULL)
  int kframe;  // tc_frame_kernel variant: kvar, or 516 (K = 5, batches of 16) when every component group fits 320 nodes / edges
  int frame_seg_cap;  // tc_frame_kernel's own capacity: seg_lds_cap, but never below 8 (its raster stage reads the list from LDS only)
  int frame_streams;  // short K-step calls: frame launches of consecutive chunks alternate between two streams (TC_FRAME_STREAMS)
  int prof_piped[TC_PROF_RING];
  int envg_map_lds; // tc_envg_kernel keeps the edge records in LDS when they fit (TC_ENVG_MAP_LDS=0: always from global)
  int first_per_env;  // the first chunk of a pipelined call goes through tc_env_kernel (TC_FIRST_CHUNK_PER_ENV=0: grouped too)
  int env_grouped;  // K-step calls: simulate with tc_envg_kernel (TC_EL lanes per env); TC_ENV_GROUPED=0 keeps one wavefront per env
  // scratch ring of K-step calls (tc_env_reserve_steps): TC_RING_SLOTS chunks of ring_rows steps
  int *segm_g, *segm_n;  // [TC_RING_SLOTS * ring_rows][N][seg_cap][5], [..][N]
  double* pose_rows;     // [TC_RING_SLOTS * ring_rows][N][TC_POSE_ROW]
  int ring_rows;
  int kvar;  // register-cache slots of the simulate stage: 5, 8 (whole map in one window), 9 (camera layer groups), 13
  // optional per-kernel timing: a ring of (start, mid, end) HIP events recorded on the caller's stream
  int prof;    // 0 = off, n = record every n-th tc_step
  int prof_n;  // launches recorded so far
  int prof_calls;
  // NoiseObservationWrapper: blobs per plane (0 = off), radius bound, the per-radius span table, launch counter
  int noise_blobs, noise_max_radius;
  unsigned char* noise_hw;
  unsigned long long noise_seed;
  unsigned int* noise_step;  // device counter
  hipEvent_t ev[4][TC_PROF_RING];  // start, simulate done, all done, first frame launch (pipelined calls)
};

template <typename T>
static int upload(tc_map* m, const std::vector<T>& h, const T** out) {
  void* p = nullptr;
  size_t bytes = (h.size() ? h.size() : 1) * sizeof(T);
  HIP_TRY(hipMalloc(&p, bytes));
  m->allocs.push_back(p);
  if (h.size()) HIP_TRY(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (const T*)p;
  return TC_OK;
}

extern "C" int tc_map_destroy(tc_map* m) {
  if (!m) return TC_OK;
  for (void* p : m->allocs) (void)hipFree(p);
  delete m;
  return TC_OK;
}

extern "C" int tc_map_create(const tc_map_desc* desc, tc_map** out) {
  if (!desc || !out) return TC_E_INVALID;
  *out = nullptr;
  const int C = desc->n_layers;
  {
    long long tn = 0;
    for (int l = 0; l < C && l < TC_MAX_LAYERS; l++) tn += desc->node_count[l];
    if (tn > 65535) {
      set_err("tc_map_create: at most 65535 lane-line nodes (node ids are packed in 16 bits in the clip lists)");
      return TC_E_INVALID;
    }
  }
  if (C < 1 || C > TC_MAX_LAYERS || desc->lanepath_node_count < 1 || desc->lanepath_edge_count < 0) {
    set_err("tc_map_create: need 1..16 lane-line layers and a non-empty lanepath");
    return TC_E_INVALID;
  }
  tc_map* m = new tc_map();
  DevMap& d = m->d;
  memset(&d, 0, sizeof(d));
  d.C = C;
  for (int l = 0; l < C; l++) {
    if (desc->node_count[l] < 0 || desc->edge_count[l] < 0) {
      delete m;
      return TC_E_INVALID;
    }
    d.node_off[l + 1] = d.node_off[l] + desc->node_count[l];
    d.edge_off[l + 1] = d.edge_off[l] + desc->edge_count[l];
    if (desc->node_count[l] > d.max_nodes) d.max_nodes = desc->node_count[l];
    if (desc->edge_count[l] > d.max_edges) d.max_edges = desc->edge_count[l];
    memcpy(d.colors[l], desc->colors + 3 * l, 3);
  }
  d.total_nodes = d.node_off[C];
  d.total_edges = d.edge_off[C];
  const int TN = d.total_nodes, TE = d.total_edges;
  // validate indices: the kernels index LDS/global with them
  for (int l = 0; l < C; l++)
    for (int e = d.edge_off[l]; e < d.edge_off[l + 1]; e++)
      for (int k = 0; k < 2; k++) {
        int v = desc->edges[2 * e + k];
        if (v < 0 || v >= desc->node_count[l]) {
          set_err("tc_map_create: lane-line edge references a node outside its layer");
          delete m;
          return TC_E_INVALID;
        }
      }
  const int lpN = desc->lanepath_node_count, lpE = desc->lanepath_edge_count;
  for (int e = 0; e < lpE; e++)
    for (int k = 0; k < 2; k++) {
      int v = desc->lanepath_edges[2 * e + k];
      if (v < 0 || v >= lpN) {
        set_err("tc_map_create: lanepath edge references a missing node");
        delete m;
        return TC_E_INVALID;
      }
    }
  std::vector<double2> nodes(TN), lpn(lpN);
  std::vector<int2> edges(TE), lpe(lpE), edges_g(TE);
  std::vector<int> edge_layer(TE);
  std::vector<double> of(TE), orv(TE), lpo(lpE), nori(lpE), pori(lpE);
  std::vector<double4> exy(TE);
  std::vector<int> noff(lpN + 1, 0), poff(lpN + 1, 0), nnode(lpE), pnode(lpE);
  for (int i = 0; i < TN; i++) nodes[i] = make_double2(desc->nodes[2 * i], desc->nodes[2 * i + 1]);
  for (int l = 0; l < C; l++)
    for (int e = d.edge_off[l]; e < d.edge_off[l + 1]; e++) {
      edges[e] = make_int2(desc->edges[2 * e], desc->edges[2 * e + 1]);
      edges_g[e] = make_int2(d.node_off[l] + edges[e].x, d.node_off[l] + edges[e].y);
      edge_layer[e] = l;
      double2 a = nodes[d.node_off[l] + edges[e].x], b = nodes[d.node_off[l] + edges[e].y];
      double evx = b.x - a.x, evy = b.y - a.y;
      // static halves of layer.py:140-141, evaluated with the host libm like the reference does
      of[e] = atan2(evy, evx);
      orv[e] = atan2(-evy, -evx);
      exy[e] = make_double4(a.x, a.y, b.x, b.y);
    }
  for (int i = 0; i < lpN; i++) lpn[i] = make_double2(desc->lanepath_nodes[2 * i], desc->lanepath_nodes[2 * i + 1]);
  for (int e = 0; e < lpE; e++) {
    lpe[e] = make_int2(desc->lanepath_edges[2 * e], desc->lanepath_edges[2 * e + 1]);
    double2 a = lpn[lpe[e].x], b = lpn[lpe[e].y];
    lpo[e] = atan2(b.y - a.y, b.x - a.x);  // layer.py:179-181
    noff[lpe[e].x + 1]++;
    poff[lpe[e].y + 1]++;
  }
  for (int i = 0; i < lpN; i++) {
    noff[i + 1] += noff[i];
    poff[i + 1] += poff[i];
  }
  {  // get_next_nodes / get_prev_nodes (layer.py:183-185) as CSR, edge-list order preserved
    std::vector<int> nf(lpN, 0), pf(lpN, 0);
    for (int e = 0; e < lpE; e++) {
      int a = lpe[e].x, b = lpe[e].y;
      int s = noff[a] + nf[a]++;
      nnode[s] = b;
      nori[s] = atan2(lpn[b].y - lpn[a].y, lpn[b].x - lpn[a].x);  // layer.py:122 seen from a
      int p = poff[b] + pf[b]++;
      pnode[p] = a;
      pori[p] = atan2(lpn[a].y - lpn[b].y, lpn[a].x - lpn[b].x);  // layer.py:122 seen from b
    }
  }
  std::vector<LpNode> fat(lpN);
  for (int i = 0; i < lpN; i++) {
    LpNode& F = fat[i];
    memset(&F, 0, sizeof(F));
    F.x = lpn[i].x;
    F.y = lpn[i].y;
    F.nnext = noff[i + 1] - noff[i];
    F.nprev = poff[i + 1] - poff[i];
    for (int k = 0; k < 3; k++) {
      F.next[k] = k < F.nnext ? nnode[noff[i] + k] : -1;
      F.next_ori[k] = k < F.nnext ? nori[noff[i] + k] : 0.0;
      F.prev[k] = k < F.nprev ? pnode[poff[i] + k] : -1;
      F.prev_ori[k] = k < F.nprev ? pori[poff[i] + k] : 0.0;
    }
  }
  d.lpN = lpN;
  d.lpE = lpE;
  d.first_spawnable = -1;
  for (int i = 0; i < lpN && d.first_spawnable < 0; i++)
    if (noff[i + 1] > noff[i]) d.first_spawnable = i;
  if (d.first_spawnable < 0) {
    set_err("tc_map_create: lanepath has no edge, nothing can spawn");
    delete m;
    return TC_E_INVALID;
  }
  // ---- candidate grid (see DevMap): cells of >= 4 cm, at most ~64 k of them, over the lane lines + 4 m
  std::vector<int> coff(1, 0), cidx;
  {
    bool finite = TN > 0 && TE > 0;
    double bx0 = 0, by0 = 0, bx1 = 0, by1 = 0;
    for (int i = 0; i < TN && finite; i++) {
      const double x = nodes[i].x, y = nodes[i].y;
      if (!(fabs(x) < 1e12) || !(fabs(y) < 1e12)) finite = false;
      if (i == 0 || x < bx0) bx0 = x;
      if (i == 0 || x > bx1) bx1 = x;
      if (i == 0 || y < by0) by0 = y;
      if (i == 0 || y > by1) by1 = y;
    }
    if (const char* g = getenv("TC_CAND_GRID")) finite = finite && atoi(g) != 0;
    std::vector<double> f(d.max_edges > 0 ? d.max_edges : 1);
    // appends the lists of an nx x ny grid of `cell`-sized cells with origin (x0, y0); returns the id of its first cell
    auto add_level = [&](double x0, double y0, double cell, int nx, int ny) {
      const int base = (int)((coff.size() - 1) / C);
      const double r = 0.5 * sqrt(2.0) * cell * 1.001 + 1e-12;
      coff.resize(coff.size() + (size_t)nx * ny * C, 0);
      for (int iy = 0; iy < ny; iy++)
        for (int ix = 0; ix < nx; ix++) {
          const double cx = x0 + (ix + 0.5) * cell, cy = y0 + (iy + 0.5) * cell;
          for (int l = 0; l < C; l++) {
            const int e0 = d.edge_off[l], e1 = d.edge_off[l + 1];
            double fmin = 0;
            for (int e = e0; e < e1; e++) {
              const double4 q = exy[e];
              const double a0 = q.x - cx, a1 = q.y - cy, b0 = q.z - cx, b1 = q.w - cy;
              f[e - e0] = sqrt(a0 * a0 + a1 * a1) + sqrt(b0 * b0 + b1 * b1);
              if (e == e0 || f[e - e0] < fmin) fmin = f[e - e0];
            }
            const double thr = fmin + 4.0 * r + 1e-9 * (1.0 + fmin);
            for (int e = e0; e < e1; e++)
              if (f[e - e0] <= thr) cidx.push_back(e);
            coff[((size_t)base + (size_t)iy * nx + ix) * C + l + 1] = (int)cidx.size();
          }
          if (cidx.size() > (size_t)48 << 20) return -1;  // a map on which the lists do not thin out (192 MB of ids): no grid
        }
      return base;
    };
    if (finite) {
      const double margin = 4.0, x0 = bx0 - margin, y0 = by0 - margin, x1 = bx1 + margin, y1 = by1 + margin;
      double cell = sqrt((x1 - x0) * (y1 - y0) / 65536.0);
      if (cell < 0.04) cell = 0.04;
      const int nx = (int)ceil((x1 - x0) / cell), ny = (int)ceil((y1 - y0) / cell);
      if (nx >= 1 && ny >= 1 && (long long)nx * ny <= 200000 && add_level(x0, y0, cell, nx, ny) >= 0) {
        d.grid_nx = nx;
        d.grid_ny = ny;
        d.grid_x0 = x0;
        d.grid_y0 = y0;
        d.grid_inv = 1.0 / cell;
      }
    }
    if (d.grid_nx == 0) {  // no grid (or abandoned): the kernels take the identity lists
      coff.assign(1, 0);
      cidx.clear();
    }
    if (cidx.empty()) cidx.push_back(0);
  }
  int rc = TC_OK;
  HIP_TRY(hipGetDevice(&m->device));
#define UP(vec, field)                                             \
  if (rc == TC_OK) rc = upload(m, vec, &d.field);
  UP(fat, lp_fat) UP(nodes, nodes) UP(edges, edges) UP(edges_g, edges_g) UP(edge_layer, edge_layer) UP(of, ori_fwd) UP(orv, ori_rev) UP(exy, edge_xy) UP(lpn, lp_nodes) UP(lpe, lp_edges)
  UP(lpo, lp_ori) UP(noff, next_off) UP(nnode, next_node) UP(nori, next_ori) UP(poff, prev_off)
  UP(pnode, prev_node) UP(pori, prev_ori) UP(coff, cand_off) UP(cidx, cand_idx)
#undef UP
  m->h_nodes = nodes;
  m->h_edges_g = edges_g;
  if (rc != TC_OK) {
    tc_map_destroy(m);
    return rc;
  }
  *out = m;
  return TC_OK;
}

static int align_up(int v, int a) { return (v + a - 1) / a * a; }

static int fill_camera(tc_env* e, const tc_camera_params* cam) {
  if (cam->height < 1 || cam->width < 1 || cam->height > 16384 || cam->width > 16384 || cam->line_thickness < 1 ||
      cam->line_thickness > 255 || !(cam->max_range > 0) ||
      (cam->format != TC_FMT_RGB && cam->format != TC_FMT_CLASSES)) {
    set_err("camera: need height,width,line_thickness >= 1, max_range > 0, format rgb|classes");
    return TC_E_INVALID;
  }
  DevCam& c = e->k.cam;
  c.H = cam->height;
  c.W = cam->width;
  memcpy(c.E, cam->E, sizeof(c.E));
  memcpy(c.K, cam->K, sizeof(c.K));
  c.max_range = cam->max_range;
  c.thickness = cam->line_thickness;
  c.format = cam->format;
  c.wpr = (c.W + 31) / 32;
  return TC_OK;
}

extern "C" int tc_env_create(const tc_map* map, const tc_car_params* car, const tc_camera_params* cam,
                             int32_t num_envs, tc_env** out) {
  if (!map || !car || !cam || !out || num_envs < 1) return TC_E_INVALID;
  *out = nullptr;
  tc_env* e = new tc_env();
  memset(&e->k, 0, sizeof(e->k));
  e->map = map;
  e->bound = false;
  e->prof = 0;
  e->prof_n = 0;
  e->prof_calls = 0;
  memset(e->ev, 0, sizeof(e->ev));
  e->fuse = 1;
  if (const char* fu = getenv("TC_FUSE")) e->fuse = atoi(fu) != 0;
  e->multi_split = 1;
  if (const char* ms = getenv("TC_MULTI_SPLIT")) e->multi_split = atoi(ms) != 0;
  e->env_grouped = 1;
  if (const char* eg = getenv("TC_ENV_GROUPED")) e->env_grouped = atoi(eg) != 0;
  e->envg_map_lds = 1;
  if (const char* ml = getenv("TC_ENVG_MAP_LDS")) e->envg_map_lds = atoi(ml) != 0;
  e->chunk = 16;
  e->pipe = 1;
  if (const char* ch = getenv("TC_CHUNK")) {
    if (atoi(ch) > 0)
      e->chunk = atoi(ch);
    else
      e->pipe = 0;
  }
  e->frame_stream = e->frame_stream2 = nullptr;
  e->sim_ev = e->frames_ev = e->frames_ev2 = e->call_ev = nullptr;
  memset(e->slot_ev, 0, sizeof(e->slot_ev));
  e->last_stream = nullptr;
  e->have_call = false;
  e->first_per_env = 1;
  if (const char* fp = getenv("TC_FIRST_CHUNK_PER_ENV")) e->first_per_env = atoi(fp) != 0;
  e->frame_streams = 2;
  if (const char* fsn = getenv("TC_FRAME_STREAMS")) e->frame_streams = atoi(fsn) == 1 ? 1 : 2;
  memset(e->prof_piped, 0, sizeof(e->prof_piped));
  e->stream = 1;
  if (const char* st = getenv("TC_STREAM")) e->stream = atoi(st) != 0;
  e->st_rows = 0;
  e->st_segm_g = e->st_segm_n = nullptr;
  e->st_pose = nullptr;
  e->st_words = nullptr;
  e->draw_base = nullptr;
  e->start_ev = nullptr;
  e->gate_ticks = 5000 * 100LL;  // 5 ms
  if (const char* gw = getenv("TC_STREAM_WAIT_US")) e->gate_ticks = (long long)(atof(gw) * 100.0);
  e->gate_test = 0;
  if (const char* gt = getenv("TC_STREAM_TEST_SKIP")) e->gate_test = atoi(gt) > 0 ? atoi(gt) : 0;
  if (hipStreamCreateWithFlags(&e->frame_stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&e->frame_stream2, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&e->frames_ev2, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->sim_ev, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->start_ev, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->frames_ev, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->call_ev, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->slot_ev[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->slot_ev[1], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->slot_ev[2], hipEventDisableTiming) != hipSuccess) {
    set_err("tc_env_create: cannot create the internal frame streams / events");
    tc_env_destroy(e);
    return TC_E_HIP;
  }
  static_assert(TC_RING_SLOTS == 3, "slot events are created one by one above");
  e->segm_g = e->segm_n = nullptr;
  e->pose_rows = nullptr;
  e->ring_rows = 0;
  e->k.m = map->d;
  e->k.N = num_envs;
  DevCar& c = e->k.car;
  c.T = car->T;
  c.wheelbase = car->wheelbase;
  c.track_width = car->track_width;
  c.max_velocity = car->max_velocity;
  c.max_steering_angle = car->max_steering_angle;
  c.steering_speed = car->steering_speed;
  c.max_acceleration = car->max_acceleration;
  c.max_deceleration = car->max_deceleration;
  c.has_steering_speed = car->has_steering_speed;
  c.has_max_acceleration = car->has_max_acceleration;
  int rc = fill_camera(e, cam);
  if (rc != TC_OK) {
    delete e;
    return rc;
  }
  DevCam& dc = e->k.cam;
  const DevMap& m = map->d;
  // bit-plane band: keep one band's planes within a budget so several envs share a CU's 160 KiB LDS
  int budget = 16384;
  if (const char* s = getenv("TC_BAND_BYTES")) {
    int v = atoi(s);
    if (v >= 1024) budget = v;
  }
  int row_bytes = m.C * dc.wpr * 4;
  int band_rows = budget / row_bytes;
  if (band_rows < 1) band_rows = 1;
  if (band_rows > dc.H) band_rows = dc.H;
  dc.band_rows = band_rows;
  dc.n_bands = (dc.H + band_rows - 1) / band_rows;
  // Camera layer groups.  Up to 512 nodes / edges the whole map is one group held in the K = 5 / 8 register cache.
  // Beyond that the layers are packed greedily, in order, into groups no larger than the largest single layer:
  // the LDS node buffer shrinks from the whole map to that layer (knuffingen: 827 -> 517 nodes, 26.6 -> 17.4 KB per
  // env, 6 -> 8 workgroups per CU) and a K = 9 cache (576 slots) covers a group.  Maps whose largest layer exceeds
  // 576 nodes or edges, or that would need more than TC_MAX_GROUPS groups, stay one group on the K = 13 windowed path
  // (env var TC_GROUPS=0 forces that).
  const int big = m.total_nodes > m.total_edges ? m.total_nodes : m.total_edges;
  int cap_n = m.total_nodes, cap_e = m.total_edges;
  e->k.n_grp = 1;
  e->k.grp_layer[0] = 0;
  e->k.grp_layer[1] = m.C;
  e->kvar = big <= 5 * TC_NT ? 5 : big <= 8 * TC_NT ? 8 : 13;
  bool want_groups = big > 8 * TC_NT;
  if (const char* sg = getenv("TC_GROUPS")) want_groups = want_groups && atoi(sg) != 0;
  if (want_groups) {
    int cap = 0;
    for (int l = 0; l < m.C; l++) {
      int nl = m.node_off[l + 1] - m.node_off[l], el = m.edge_off[l + 1] - m.edge_off[l];
      cap = nl > cap ? nl : cap;
      cap = el > cap ? el : cap;
    }
    if (cap <= 9 * TC_NT) {
      int lay[TC_MAX_GROUPS + 1], ng = 0, l = 0;
      bool ok = true;
      lay[0] = 0;
      while (l < m.C) {
        if (ng == TC_MAX_GROUPS) {
          ok = false;
          break;
        }
        int first = l;
        while (l < m.C && m.node_off[l + 1] - m.node_off[first] <= cap && m.edge_off[l + 1] - m.edge_off[first] <= cap) l++;
        lay[++ng] = l;  // l > first: a single layer always fits `cap`
      }
      if (ok && ng >= 2) {
        e->k.n_grp = ng;
        cap_n = cap_e = 0;
        for (int g = 0; g <= ng; g++) e->k.grp_layer[g] = lay[g];
        for (int g = 0; g < ng; g++) {
          int nl = m.node_off[lay[g + 1]] - m.node_off[lay[g]], el = m.edge_off[lay[g + 1]] - m.edge_off[lay[g]];
          cap_n = nl > cap_n ? nl : cap_n;
          cap_e = el > cap_e ? el : cap_e;
        }
        e->kvar = 9;
      }
    }
  }
  // group bounds of the layer scheme (or of the single group)
  e->k.cam_nodes = nullptr;
  e->k.cam_edges_g = nullptr;
  e->cam_nodes_dev = e->cam_edges_dev = nullptr;
  for (int g = 0; g <= e->k.n_grp; g++) {
    e->k.grp_n0[g] = m.node_off[e->k.grp_layer[g]];
    e->k.grp_e0[g] = m.edge_off[e->k.grp_layer[g]];
  }
  for (int g = 0; g < e->k.n_grp; g++) {
    e->k.grp_l0[g] = e->k.grp_layer[g];
    e->k.grp_l1[g] = e->k.grp_layer[g + 1];
  }
  // Component groups.  The layer scheme leaves the LDS node buffer at the size of the largest LAYER (knuffingen: 517 of
  // 827 nodes, 12.4 KB of a workgroup's 17.4 KB), and that buffer is what decides how many frame workgroups a CU holds
  // (9).  A layer is not the unit of independence, though: camera.py's fix-up passes move a node along its own edges
  // only, so every connected component of the lane-line graph (a dash of a dashed line is one) can be processed on its
  // own.  The camera stage therefore works from a copy of the map in which the components of each layer stand side by
  // side -- nodes renumbered, every layer's edges still contiguous and in their original relative order (the replay of a
  // fix-up chain follows edge order, and a chain never leaves its component) -- packed greedily into groups of at most
  // TC_CAM_GROUP nodes / edges (default 320: the buffer shrinks to 7.7 KB).  Nodes without an edge are never drawn and
  // are left out.  Falls back to the layer scheme when a component is larger than that or the groups are too many.
  if (e->k.n_grp >= 2 && e->kvar == 9 && (int)map->h_nodes.size() == m.total_nodes && (int)map->h_edges_g.size() == m.total_edges) {
    int T = 5 * TC_NT;
    if (const char* cg = getenv("TC_CAM_GROUP")) T = atoi(cg);
    if (T >= TC_NT && T <= 9 * TC_NT) {
      const int TN = m.total_nodes, TE = m.total_edges;
      std::vector<int> parent(TN);
      for (int i = 0; i < TN; i++) parent[i] = i;
      auto find = [&](int x) {
        while (parent[x] != x) x = parent[x] = parent[parent[x]];
        return x;
      };
      for (int ed = 0; ed < TE; ed++) {
        const int a0 = find(map->h_edges_g[ed].x), b0 = find(map->h_edges_g[ed].y);
        if (a0 != b0) parent[b0 > a0 ? b0 : a0] = b0 > a0 ? a0 : b0;
      }
      // components in order of their first edge (edges are layer by layer, so components are too)
      std::vector<int> comp_of_root(TN, -1), comp_first_edge, comp_nn, comp_ne;
      std::vector<int> edge_comp(TE);
      for (int ed = 0; ed < TE; ed++) {
        const int r = find(map->h_edges_g[ed].x);
        if (comp_of_root[r] < 0) {
          comp_of_root[r] = (int)comp_first_edge.size();
          comp_first_edge.push_back(ed);
          comp_nn.push_back(0);
          comp_ne.push_back(0);
        }
        edge_comp[ed] = comp_of_root[r];
        comp_ne[edge_comp[ed]]++;
      }
      std::vector<int> node_comp(TN, -1);
      for (int i = 0; i < TN; i++) {
        const int c = comp_of_root[find(i)];
        node_comp[i] = c;  // -1: a node without an edge
        if (c >= 0) comp_nn[c]++;
      }
      const int NC = (int)comp_first_edge.size();
      bool ok = NC > 0;
      for (int c = 0; c < NC && ok; c++) ok = comp_nn[c] <= T && comp_ne[c] <= T;
      // a component's edges must all belong to one layer (they do: edges join nodes of their own layer) and components
      // must appear layer by layer, so that an edge's position in the new order is its position in the old one's layer
      std::vector<int> grp_first_comp;
      if (ok) {
        int gn = 0, ge = 0;
        grp_first_comp.push_back(0);
        for (int c = 0; c < NC; c++) {
          if (gn + comp_nn[c] > T || ge + comp_ne[c] > T) {
            grp_first_comp.push_back(c);
            gn = ge = 0;
          }
          gn += comp_nn[c];
          ge += comp_ne[c];
        }
        grp_first_comp.push_back(NC);
        ok = (int)grp_first_comp.size() - 1 <= TC_MAX_GROUPS;
      }
      if (ok) {
        // new node ids: components in order, nodes of a component in their old order; new edge order: by component,
        // old order inside -- which is the old order within each layer as long as a layer's components are visited in
        // the order of their first edges AND no component's edges interleave with another's inside a layer.  They may
        // (two dashes drawn alternately), so the edges are NOT moved: an edge keeps its index and only its node ids
        // change; a group's edges are then the index range from its first component's first edge to its last
        // component's last edge, which must not contain edges of other groups' components -- checked below.
        std::vector<int> new_id(TN, -1);
        std::vector<int> comp_n0(NC + 1, 0);
        for (int c = 0; c < NC; c++) comp_n0[c + 1] = comp_n0[c] + comp_nn[c];
        std::vector<int> fill(comp_n0.begin(), comp_n0.end() - 1);
        for (int i = 0; i < TN; i++)
          if (node_comp[i] >= 0) new_id[i] = fill[node_comp[i]]++;
        std::vector<int> comp_last_edge(NC, -1);
        for (int ed = 0; ed < TE; ed++) comp_last_edge[edge_comp[ed]] = ed;
        const int NG = (int)grp_first_comp.size() - 1;
        std::vector<int> ge0(NG + 1), gn0(NG + 1);
        for (int g = 0; g < NG && ok; g++) {
          const int c0 = grp_first_comp[g], c1 = grp_first_comp[g + 1];
          int lo = TE, hi = -1;
          for (int c = c0; c < c1; c++) {
            lo = comp_first_edge[c] < lo ? comp_first_edge[c] : lo;
            hi = comp_last_edge[c] > hi ? comp_last_edge[c] : hi;
          }
          ge0[g] = lo;
          gn0[g] = comp_n0[c0];
          for (int ed = lo; ed <= hi && ok; ed++) ok = edge_comp[ed] >= c0 && edge_comp[ed] < c1;
          if (g > 0) ok = ok && lo == ge0[g - 1] + [&] { int n = 0; for (int c = grp_first_comp[g - 1]; c < c0; c++) n += comp_ne[c]; return n; }();
        }
        ge0[NG] = TE;
        gn0[NG] = comp_n0[NC];
        ok = ok && ge0[0] == 0;
        if (ok) {
          std::vector<double2> cn((size_t)comp_n0[NC]);
          std::vector<int2> ce((size_t)TE);
          for (int i = 0; i < TN; i++)
            if (new_id[i] >= 0) cn[new_id[i]] = map->h_nodes[i];
          for (int ed = 0; ed < TE; ed++) ce[ed] = make_int2(new_id[map->h_edges_g[ed].x], new_id[map->h_edges_g[ed].y]);
          void *dn = nullptr, *de = nullptr;
          if (hipMalloc(&dn, cn.size() * sizeof(double2)) == hipSuccess && hipMalloc(&de, ce.size() * sizeof(int2)) == hipSuccess &&
              hipMemcpy(dn, cn.data(), cn.size() * sizeof(double2), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(de, ce.data(), ce.size() * sizeof(int2), hipMemcpyHostToDevice) == hipSuccess) {
            e->cam_nodes_dev = dn;
            e->cam_edges_dev = de;
            e->k.cam_nodes = (const double2*)dn;
            e->k.cam_edges_g = (const int2*)de;
            e->k.n_grp = NG;
            cap_n = cap_e = 0;
            for (int g = 0; g <= NG; g++) {
              e->k.grp_n0[g] = gn0[g];
              e->k.grp_e0[g] = ge0[g];
            }
            for (int g = 0; g < NG; g++) {
              const int nl = gn0[g + 1] - gn0[g], el = ge0[g + 1] - ge0[g];
              cap_n = nl > cap_n ? nl : cap_n;
              cap_e = el > cap_e ? el : cap_e;
              int la = 0, lb = 0;
              while (la + 1 < m.C && ge0[g] >= m.edge_off[la + 1]) la++;
              while (lb + 1 < m.C && ge0[g + 1] - 1 >= m.edge_off[lb + 1]) lb++;
              e->k.grp_l0[g] = la;
              e->k.grp_l1[g] = lb + 1;
            }
          } else {
            if (dn) (void)hipFree(dn);
            if (de) (void)hipFree(de);
          }
        }
      }
    }
  }
  e->kframe = e->kvar;
  if (e->k.cam_nodes && cap_n <= 5 * TC_NT && cap_e <= 5 * TC_NT) e->kframe = 516;
  e->k.cap_nodes = cap_n > 0 ? cap_n : 1;
  LdsLayout& L = e->k.lds;
  int off = 0;
  L.off_p = off;  // node buffer: 3 doubles per node of the largest camera group; phase B aliases it with one
                  // double per lane-line node of the whole map
  int pbytes = 3 * e->k.cap_nodes * 8;
  if (m.total_nodes * 8 > pbytes) pbytes = m.total_nodes * 8;
  off += align_up(pbytes, 16);
  L.off_flg = off;
  off += align_up(e->k.cap_nodes, 16);
  L.off_list = off;  // fix-up edge list / projection candidate list
  off += align_up((2 * cap_e > cap_n ? 2 * cap_e : cap_n) * 4 + 16, 16);
  L.off_cnt = off;
  off += 64;
  L.total = off;
  // raster kernel: tables + bit-planes of one band
  // A frame taller than one band: the workgroup's LDS is the larger of the two stages' needs, and how many workgroups a
  // CU holds decides the rate of the big frames (cfg5: 6 per CU at 23.2 KB, 9 at 17.4 KB: 3.10 -> 3.82 M env-steps/s).
  // So the band shrinks until the raster stage needs no more than the camera stage does anyway -- not further: more
  // bands only repeat the per-band set-up (measured: 8 KB and 6 KB bands are slower again).
  // (tables of the batch size the fused / frame kernels of this variant are compiled with; tc_raster_kernel: RB_MAX)
  const int r_off_bits_k = e->kvar == 13 ? R_OFF_BITS_OF(RB_MAX) : (e->kvar >= 8 ? R_OFF_BITS_OF(RB_OF_K(8)) : R_OFF_BITS_OF(RB_OF_K(5)));
  if (dc.n_bands > 1 && !getenv("TC_BAND_BYTES")) {
    const int fit = (L.total - r_off_bits_k) / row_bytes;
    if (fit >= 8 && fit < dc.band_rows) {
      dc.band_rows = band_rows = fit;
      dc.n_bands = (dc.H + band_rows - 1) / band_rows;
    }
  }
  e->r_off_tab = R_OFF_TAB;
  e->r_off_bits = R_OFF_BITS_OF(RB_MAX);
  e->r_lds = e->r_off_bits + align_up(m.C * band_rows * dc.wpr * 4, 16);
  // the env's parked state (tc_step_multi) sits behind whichever stage needs more, so that neither aliases it
  const int r_lds_k = r_off_bits_k + align_up(m.C * band_rows * dc.wpr * 4, 16);
  L.off_live = align_up(L.total > r_lds_k ? L.total : r_lds_k, 16);
  L.total = L.off_live + TC_LIVE_BYTES;
  {
    // tc_frame_kernel keeps no env state in LDS, so the bytes behind both stages' buffers -- the LiveLds slot and whatever
    // the workgroup can grow without costing the CU a workgroup (160 KB / workgroups per CU, in 1280-byte steps) --
    // hold the head of the frame's draw list: the raster stage reads what the camera stage of the same wavefront just
    // wrote without a round trip through global memory (cfg3: 44 of a frame's ~20-30 segments).
    const int cu_lds = 160 * 1024;
    const int w = cu_lds / L.total > 0 ? cu_lds / L.total : 1;
    int grown = cu_lds / w / 1280 * 1280;
    if (grown > L.total + 4096) grown = L.total + 4096;  // (200 segments are plenty)
    if (grown < L.total) grown = L.total;
#ifdef TC_TIMING_LDS
    if (grown - 256 >= L.total) grown -= 256;  // the stamp buffer (static LDS) comes out of the draw-list head: same workgroups per CU
#endif
    e->frame_lds = e->step_lds = grown;
    e->seg_lds_off = L.off_live;
    e->seg_lds_cap = (grown - L.off_live) / 20;
    e->seg_lds_limit = 1 << 30;
    if (const char* sl = getenv("TC_SEG_LDS")) {
      if (atoi(sl) == 0) {
        e->seg_lds_cap = 0;
        e->frame_lds = e->step_lds = L.total;
      }
    }
    // TC_SEG_LDS_CAP=n: at most n segments of a frame's draw list stay in LDS (tests: a small n makes every frame take
    // the mixed LDS + global list that otherwise only frames with more than ~44 segments reach)
    if (const char* sc = getenv("TC_SEG_LDS_CAP")) {
      const int v = atoi(sc);
      if (v >= 0) e->seg_lds_limit = v;
      if (e->seg_lds_cap > e->seg_lds_limit) e->seg_lds_cap = e->seg_lds_limit;
    }
    // tc_frame_kernel reads draw lists from LDS only (longer ones batch by batch through it): it keeps a region of at least
    // 8 entries whatever the switches above say (they still rule tc_step_kernel, which has the generic form)
    e->frame_seg_cap = e->seg_lds_cap;
    if (e->frame_seg_cap < 8) {
      e->frame_seg_cap = 8;
      if (e->frame_lds < e->seg_lds_off + 8 * 20) e->frame_lds = e->seg_lds_off + 8 * 20;
    }
#ifdef TC_ABLATE
    if (getenv("TC_PRINT_LDS"))
      fprintf(stderr, "tc_env_create: lds.total %d off_live %d frame_lds %d seg_lds_cap %d r_lds %d band_rows %d n_bands %d\n", (int)L.total,
              (int)L.off_live, e->frame_lds, e->seg_lds_cap, e->r_lds, e->k.cam.band_rows, e->k.cam.n_bands);
#endif
#ifdef TC_EXPERIMENT
    // occupancy experiments (make dev-exp, never shipped): unused LDS bytes per frame workgroup -> fewer workgroups per CU
    if (const char* pd = getenv("TC_LDS_PAD")) e->frame_lds += atoi(pd);
    if (const char* pd = getenv("TC_STEP_LDS_PAD")) e->step_lds += atoi(pd);
#endif
  }
  if (L.total > 160 * 1024 || e->r_lds > 160 * 1024) {
    set_err("tc_env_create: map too large for one workgroup's LDS");
    delete e;
    return TC_E_LDS;
  }
  {
    int lds = L.total > e->r_lds ? L.total : e->r_lds;
    if (e->frame_lds > lds) lds = e->frame_lds;
    if (lds > 48 * 1024) {
      for (int t = 0; t < 2; t++)
        for (int c = 0; c < 2; c++) {
          (void)hipFuncSetAttribute((const void*)pick_fused<5>(t, c), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
          (void)hipFuncSetAttribute((const void*)pick_fused<5, 16>(t, c), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
          (void)hipFuncSetAttribute((const void*)pick_fused<8>(t, c), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
          (void)hipFuncSetAttribute((const void*)pick_fused<9>(t, c), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
          (void)hipFuncSetAttribute((const void*)pick_frame<5>(t, c), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
          (void)hipFuncSetAttribute((const void*)pick_frame<5, 16>(t, c), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
          (void)hipFuncSetAttribute((const void*)pick_frame<8>(t, c), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
          (void)hipFuncSetAttribute((const void*)pick_frame<9>(t, c), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        }
    }
  }
  {  // tc_envg_kernel's LDS copies of the map (edge records + fat lanepath nodes, see launch()) can exceed the 48 KB default
    const size_t envg_lds = ((size_t)m.total_edges * 48 + 15) / 16 * 16 + (size_t)m.lpN * sizeof(LpNode);
    if (envg_lds > 40 * 1024)
      (void)hipFuncSetAttribute((const void*)tc_envg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(envg_lds < 150 * 1024 ? envg_lds : 150 * 1024));
  }
#ifndef TC_DEV_FAST
  if (e->r_lds > 48 * 1024) {
    (void)hipFuncSetAttribute((const void*)tc_raster_kernel<true, TC_FMT_CLASSES>, hipFuncAttributeMaxDynamicSharedMemorySize, e->r_lds);
    (void)hipFuncSetAttribute((const void*)tc_raster_kernel<true, TC_FMT_RGB>, hipFuncAttributeMaxDynamicSharedMemorySize, e->r_lds);
    (void)hipFuncSetAttribute((const void*)tc_raster_kernel<false, TC_FMT_CLASSES>, hipFuncAttributeMaxDynamicSharedMemorySize, e->r_lds);
    (void)hipFuncSetAttribute((const void*)tc_raster_kernel<false, TC_FMT_RGB>, hipFuncAttributeMaxDynamicSharedMemorySize, e->r_lds);
  }
  if (L.total > 48 * 1024) {
    hipError_t he = hipSuccess;
    const void* ek[8] = {(const void*)tc_env_kernel<5, true>,  (const void*)tc_env_kernel<8, true>,  (const void*)tc_env_kernel<9, true>,
                         (const void*)tc_env_kernel<13, true>, (const void*)tc_env_kernel<5, false>, (const void*)tc_env_kernel<8, false>,
                         (const void*)tc_env_kernel<9, false>, (const void*)tc_env_kernel<13, false>};
    for (int i = 0; i < 8 && he == hipSuccess; i++) he = hipFuncSetAttribute(ek[i], hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
    if (he != hipSuccess) {
      set_err(std::string("hipFuncSetAttribute: ") + hipGetErrorString(he));
      delete e;
      return TC_E_HIP;
    }
  }
#endif
  e->obs_bytes = (int64_t)dc.H * dc.W * (dc.format == TC_FMT_CLASSES ? m.C : 3);
  e->k.seg_cap = m.total_edges > 0 ? m.total_edges : 1;
  {
    void *p = nullptr, *q = nullptr;
    hipError_t he = hipMalloc(&p, (size_t)num_envs * e->k.seg_cap * 5 * sizeof(int));
    if (he == hipSuccess) he = hipMalloc(&q, (size_t)num_envs * sizeof(int));
    if (he == hipSuccess) he = hipMemset(q, 0, (size_t)num_envs * sizeof(int));
    if (he != hipSuccess) {
      set_err(std::string("hipMalloc(draw list): ") + hipGetErrorString(he));
      if (p) (void)hipFree(p);
      delete e;
      return TC_E_NOMEM;
    }
    e->k.seg_g = (int*)p;
    e->k.seg_n = (int*)q;
  }
  {
    // cost-aware env order of single-step launches (tc_order_kernel): when the N workgroups are whole rows of G = SIMDs
    // of the device and all resident at once (N / G <= 4 wavefronts per SIMD).  TC_STEP_ORDER=n refreshes the order every
    // n-th tc_step (default 8), 0 switches it off.
    e->env_order = nullptr;
    e->frame_order[0] = e->frame_order[1] = nullptr;
    e->cost_row[0] = e->cost_row[1] = nullptr;
    e->frame_order_valid[0] = e->frame_order_valid[1] = false;
    if (!(getenv("TC_FRAME_ORDER") && atoi(getenv("TC_FRAME_ORDER")) == 0) && num_envs <= 60000) {
      void *f0 = nullptr, *f1 = nullptr;
      if (hipMalloc(&f0, (size_t)num_envs * sizeof(int)) == hipSuccess && hipMalloc(&f1, (size_t)num_envs * sizeof(int)) == hipSuccess) {
        e->frame_order[0] = (int*)f0;
        e->frame_order[1] = (int*)f1;
      } else {
        if (f0) (void)hipFree(f0);
        (void)hipGetLastError();
      }
    }
    e->order_calls = 0;
    e->order_every = 8;
    if (const char* so = getenv("TC_STEP_ORDER")) e->order_every = atoi(so) > 0 ? atoi(so) : 0;
    hipDeviceProp_t prop;
    int dev = 0;
    e->order_g = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) e->order_g = prop.multiProcessorCount * 4;
    if (e->order_every > 0 && e->order_g > 0 && num_envs % e->order_g == 0 && num_envs / e->order_g <= 4 && num_envs / e->order_g >= 2) {
      std::vector<int> ident((size_t)num_envs);
      for (int i = 0; i < num_envs; i++) ident[(size_t)i] = i;
      void* o = nullptr;
      if (hipMalloc(&o, (size_t)num_envs * sizeof(int)) == hipSuccess &&
          hipMemcpy(o, ident.data(), (size_t)num_envs * sizeof(int), hipMemcpyHostToDevice) == hipSuccess) {
        e->env_order = (int*)o;
      } else {
        if (o) (void)hipFree(o);
        (void)hipGetLastError();
      }
    }
  }
  *out = e;
  return TC_OK;
}

extern "C" int tc_env_profile(tc_env* e, int32_t enable) {
  if (!e) return TC_E_INVALID;
  if (enable && !e->ev[0][0]) {
    for (int k = 0; k < 4; k++)
      for (int i = 0; i < TC_PROF_RING; i++) HIP_TRY(hipEventCreate(&e->ev[k][i]));
  }
  e->prof = enable > 0 ? enable : 0;
  e->prof_n = 0;
  e->prof_calls = 0;
  return TC_OK;
}

extern "C" int tc_env_profile_read(tc_env* e, double* sim_us, double* raster_us, int32_t* launches) {
  if (!e || !sim_us || !raster_us || !launches) return TC_E_INVALID;
  int n = e->prof_n < TC_PROF_RING ? e->prof_n : TC_PROF_RING;
  double a = 0, b = 0;
  for (int i = 0; i < n; i++) {
    float t0 = 0, t1 = 0;
    HIP_TRY(hipEventSynchronize(e->ev[2][i]));
    HIP_TRY(hipEventElapsedTime(&t0, e->ev[0][i], e->ev[1][i]));
    // pipelined K-step call: the frame launches run beside the simulate launches, from their own start event
    HIP_TRY(hipEventElapsedTime(&t1, e->prof_piped[i] ? e->ev[3][i] : e->ev[1][i], e->ev[2][i]));
    a += t0;
    b += t1;
  }
  *launches = n;
  *sim_us = n ? a / n * 1e3 : 0;
  *raster_us = n ? b / n * 1e3 : 0;
  return TC_OK;
}

extern "C" int tc_env_destroy(tc_env* e) {
  if (e && e->ev[0][0])
    for (int k = 0; k < 4; k++)
      for (int i = 0; i < TC_PROF_RING; i++) (void)hipEventDestroy(e->ev[k][i]);
  if (e) {
    for (int sl = 0; sl < TC_RING_SLOTS; sl++)
      if (e->slot_ev[sl]) (void)hipEventDestroy(e->slot_ev[sl]);
    if (e->call_ev) (void)hipEventDestroy(e->call_ev);
    if (e->frame_stream) (void)hipStreamDestroy(e->frame_stream);
    if (e->frame_stream2) (void)hipStreamDestroy(e->frame_stream2);
    if (e->frames_ev2) (void)hipEventDestroy(e->frames_ev2);
    if (e->sim_ev) (void)hipEventDestroy(e->sim_ev);
    if (e->start_ev) (void)hipEventDestroy(e->start_ev);
    if (e->st_segm_g) (void)hipFree(e->st_segm_g);
    if (e->st_segm_n) (void)hipFree(e->st_segm_n);
    if (e->st_pose) (void)hipFree(e->st_pose);
    if (e->st_words) (void)hipFree(e->st_words);
    if (e->cam_nodes_dev) (void)hipFree(e->cam_nodes_dev);
    if (e->cam_edges_dev) (void)hipFree(e->cam_edges_dev);
    if (e->frames_ev) (void)hipEventDestroy(e->frames_ev);
  }
  if (e && e->segm_g) (void)hipFree(e->segm_g);
  if (e && e->segm_n) (void)hipFree(e->segm_n);
  if (e && e->pose_rows) (void)hipFree(e->pose_rows);
  if (e && e->k.seg_g) (void)hipFree(e->k.seg_g);
  if (e && e->k.seg_n) (void)hipFree(e->k.seg_n);
  if (e && e->env_order) (void)hipFree(e->env_order);
  if (e && e->frame_order[0]) (void)hipFree(e->frame_order[0]);
  if (e && e->frame_order[1]) (void)hipFree(e->frame_order[1]);
  if (e && e->k.terms) (void)hipFree((void*)e->k.terms);
  if (e && e->k.spawn_tab) (void)hipFree((void*)e->k.spawn_tab);
  if (e && e->noise_hw) (void)hipFree(e->noise_hw);
  if (e && e->noise_step) (void)hipFree(e->noise_step);
  delete e;
  return TC_OK;
}

extern "C" int64_t tc_env_obs_bytes(const tc_env* e) { return e ? e->obs_bytes : TC_E_INVALID; }
extern "C" int64_t tc_env_lds_bytes(const tc_env* e) { return e ? (e->k.lds.total > e->r_lds ? e->k.lds.total : e->r_lds) : TC_E_INVALID; }

extern "C" int tc_env_bind(tc_env* e, const tc_buffers* b) {
  if (!e || !b) return TC_E_INVALID;
  if (!b->x || !b->y || !b->theta || !b->velocity || !b->steering || !b->radius || !b->front_x || !b->front_y ||
      !b->local_path || !b->lp_len || !b->last_maneuver || !b->cte || !b->heading_error || !b->reward ||
      !b->terminated || !b->truncated || !b->status || !b->laneline_distances || !b->nearest_edge) {
    set_err("tc_env_bind: a required buffer pointer is NULL");
    return TC_E_INVALID;
  }
  if (b->needs_reset && (!b->spawn_queue || !b->spawn_cursor || b->spawn_queue_len < 1)) {
    set_err("tc_env_bind: needs_reset given without spawn_queue/spawn_cursor/spawn_queue_len");
    return TC_E_INVALID;
  }
  e->k.b = *b;
  e->bound = true;
  return TC_OK;
}

extern "C" int tc_env_set_camera_per_env(tc_env* e, const double* E, const double* K) {
  if (!e || ((E == nullptr) != (K == nullptr))) return TC_E_INVALID;
  e->k.cam_E = E;
  e->k.cam_K = K;
  return TC_OK;
}

static int noise_lds_bytes(const tc_env* e) {  // bit-planes of a band + blob rows + one span-table row per blob
  const int nb = e->k.m.C * e->noise_blobs;
  return e->k.m.C * e->k.cam.band_rows * e->k.cam.wpr * 4 + nb * 5 * 4 + align_up(nb * e->noise_max_radius, 16);
}

#ifdef TC_TIMING
// timing build only: a [n_envs][32] buffer of shader-clock stamps (see TSTAMP); read() copies it to the host
// The buffer must cover every env of every launch made while it is installed: install it BEFORE the first launch of an
// env handle and remove it (n_envs = 0) before using a larger one -- a stale smaller buffer is an out-of-bounds write.
static long long* g_tstamp = nullptr;
static int g_tstamp_n = 0;
extern "C" int tc_debug_tstamp_alloc(int n_envs) {
  HIP_TRY(hipDeviceSynchronize());
  long long* none = nullptr;
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(tc_tstamp), &none, sizeof(none)));
  if (g_tstamp) HIP_TRY(hipFree(g_tstamp));
  g_tstamp = nullptr;
  g_tstamp_n = 0;
  if (n_envs <= 0) return TC_OK;
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, (size_t)n_envs * 32 * sizeof(long long)));
  HIP_TRY(hipMemset(p, 0, (size_t)n_envs * 32 * sizeof(long long)));
  g_tstamp = (long long*)p;
  g_tstamp_n = n_envs;
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(tc_tstamp), &g_tstamp, sizeof(g_tstamp)));
  return TC_OK;
}
// copies the stamps of the last launch to the host and zeroes them (a probe a wavefront skipped then reads 0)
extern "C" int tc_debug_tstamp_read(long long* out, int n_envs) {
  if (!g_tstamp || n_envs != g_tstamp_n) return TC_E_INVALID;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, g_tstamp, (size_t)n_envs * 32 * sizeof(long long), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(g_tstamp, 0, (size_t)n_envs * 32 * sizeof(long long)));
  return TC_OK;
}
#endif

extern "C" int tc_env_set_noise(tc_env* e, int32_t n_blobs, int32_t max_radius, uint64_t seed) {
  if (!e || n_blobs < 0) return TC_E_INVALID;
  if (n_blobs == 0) {
    e->noise_blobs = 0;
    return TC_OK;
  }
  if (e->k.cam.format != TC_FMT_CLASSES) {
    set_err("tc_env_set_noise: only class-mask observations (wrapper/observation.py:7)");
    return TC_E_INVALID;
  }
  if (max_radius < 2 || max_radius > 256) {
    set_err("tc_env_set_noise: max_radius must be in [2, 256] (numpy's randint(1, max_radius) needs >= 2)");
    return TC_E_INVALID;
  }
  // Circle(center, radius, fill) of drawing.cpp run once per radius on the host: per-row half widths
  std::vector<unsigned char> hw((size_t)max_radius * max_radius, 0);
  for (int radius = 1; radius < max_radius; radius++) {
    unsigned char* row = hw.data() + (size_t)radius * max_radius;
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
      if (row[dy] < dx) row[dy] = (unsigned char)dx;
      if (row[dx] < dy) row[dx] = (unsigned char)dy;
      dy++;
      err += plus;
      plus += 2;
      int mask2 = (err <= 0) - 1;
      err -= minus & mask2;
      dx += mask2;
      minus -= mask2 & 2;
    }
  }
  HIP_TRY(hipDeviceSynchronize());
  if (e->noise_hw) {
    HIP_TRY(hipFree(e->noise_hw));
    e->noise_hw = nullptr;
  }
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, hw.size()));
  HIP_TRY(hipMemcpy(p, hw.data(), hw.size(), hipMemcpyHostToDevice));
  e->noise_hw = (unsigned char*)p;
  e->noise_blobs = n_blobs;
  e->noise_max_radius = max_radius;
  e->noise_seed = seed;
  if (!e->noise_step) {
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, sizeof(unsigned int)));
    e->noise_step = (unsigned int*)q;
  }
  HIP_TRY(hipMemset(e->noise_step, 0, sizeof(unsigned int)));
  const int lds = noise_lds_bytes(e);
  if (lds > 160 * 1024) {
    set_err("tc_env_set_noise: blob tables do not fit one workgroup's LDS");
    e->noise_blobs = 0;
    return TC_E_LDS;
  }
  if (lds > 48 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void*)tc_noise_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  return TC_OK;
}

static int launch_noise(tc_env* e, const int32_t* blobs, void* stream, uint8_t* obs = nullptr) {
  NArgs n;
  memset(&n, 0, sizeof(n));
  const DevCam& c = e->k.cam;
  n.N = e->k.N;
  n.C = e->k.m.C;
  n.H = c.H;
  n.W = c.W;
  n.wpr = c.wpr;
  n.band_rows = c.band_rows;
  n.n_bands = c.n_bands;
  n.obs = obs ? obs : e->k.b.obs;
  n.blobs = blobs;
  n.n_blobs = e->noise_blobs;
  n.max_radius = e->noise_max_radius;
  n.hw = e->noise_hw;
  n.seed = e->noise_seed;
  n.step = e->noise_step;
  n.inv_cpr = (c.W >> 4) > 0 ? (unsigned int)((1ull << 32) / (unsigned)(c.W >> 4) + 1ull) : 0u;
  hipLaunchKernelGGL(tc_noise_kernel, dim3(n.N), dim3(TC_NT), noise_lds_bytes(e), (hipStream_t)stream, n);
  HIP_TRY(hipGetLastError());
  if (!blobs) {  // device-drawn blobs consumed one position of the stream
    hipLaunchKernelGGL(tc_noise_tick, dim3(1), dim3(1), 0, (hipStream_t)stream, e->noise_step, 1u);
    HIP_TRY(hipGetLastError());
  }
  return TC_OK;
}

extern "C" int tc_noise(tc_env* e, const int32_t* blobs, void* stream) {
  if (!e) return TC_E_INVALID;
  if (!e->bound || !e->k.b.obs) {
    set_err("tc_noise: no observation buffer bound");
    return TC_E_UNBOUND;
  }
  if (e->noise_blobs < 1) {
    set_err("tc_noise: call tc_env_set_noise first");
    return TC_E_INVALID;
  }
  return launch_noise(e, blobs, stream);
}

extern "C" int tc_env_set_spawn_table(tc_env* e, const int32_t* nodes, int32_t n, uint64_t seed) {
  if (!e || n < 0 || (n > 0 && !nodes)) {
    set_err("tc_env_set_spawn_table: need n >= 0 and a node array");
    return TC_E_INVALID;
  }
  for (int i = 0; i < n; i++)
    if (nodes[i] < 0 || nodes[i] >= e->k.m.lpN) {
      set_err("tc_env_set_spawn_table: node id out of range");
      return TC_E_INVALID;
    }
  HIP_TRY(hipDeviceSynchronize());  // launches in flight still read the old table
  if (e->k.spawn_tab) {
    HIP_TRY(hipFree((void*)e->k.spawn_tab));
    e->k.spawn_tab = nullptr;
  }
  e->k.spawn_n = 0;
  if (n > 0) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, sizeof(int32_t) * (size_t)n));
    HIP_TRY(hipMemcpy(p, nodes, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
    e->k.spawn_tab = (const int*)p;
    e->k.spawn_n = n;
  }
  e->k.spawn_seed = seed;
  return TC_OK;
}

extern "C" int tc_env_set_terms(tc_env* e, const tc_term* terms, int32_t n_terms, int32_t* counters) {
  if (!e || n_terms < 0 || n_terms > TC_MAX_TERMS || (n_terms > 0 && !terms)) {
    set_err("tc_env_set_terms: n_terms must be 0..TC_MAX_TERMS with a term array");
    return TC_E_INVALID;
  }
  const int C = e->k.m.C;
  const uint32_t all = C >= 32 ? 0xffffffffu : ((1u << C) - 1u);
  for (int t = 0; t < n_terms; t++) {
    const tc_term& T = terms[t];
    if (T.kind < TC_T_LANELINE_SPARSE_REWARD || T.kind > TC_T_CRASH_TERMINATION) {
      set_err("tc_env_set_terms: unknown term kind");
      return TC_E_INVALID;
    }
    if (T.layer_mask & ~all) {
      set_err("tc_env_set_terms: layer_mask names a layer the map does not have");
      return TC_E_INVALID;
    }
    if ((T.kind == TC_T_CTE_TERMINATION || T.kind == TC_T_CRASH_TERMINATION) && !counters) {
      set_err("tc_env_set_terms: consecutive-step terms need the counters buffer");
      return TC_E_INVALID;
    }
  }
  HIP_TRY(hipDeviceSynchronize());  // launches in flight still read the old table
  if (!e->k.terms) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, sizeof(tc_term) * TC_MAX_TERMS));
    e->k.terms = (const tc_term*)p;
  }
  if (n_terms > 0) HIP_TRY(hipMemcpy((void*)e->k.terms, terms, sizeof(tc_term) * n_terms, hipMemcpyHostToDevice));
  e->k.n_terms = n_terms;
  e->k.term_counters = counters;
  return TC_OK;
}

extern "C" int tc_env_set_camera(tc_env* e, const tc_camera_params* cam) {
  if (!e || !cam) return TC_E_INVALID;
  if (cam->height != e->k.cam.H || cam->width != e->k.cam.W || cam->format != e->k.cam.format) {
    set_err("tc_env_set_camera: resolution and format are fixed at tc_env_create");
    return TC_E_INVALID;
  }
  int band_rows = e->k.cam.band_rows, n_bands = e->k.cam.n_bands;
  int rc = fill_camera(e, cam);
  e->k.cam.band_rows = band_rows;
  e->k.cam.n_bands = n_bands;
  return rc;
}

static RArgs make_rargs(tc_env* e, const int* seg_g, const int* seg_n, int seg_cap, const uint8_t* mask, uint32_t flags,
                        int env0, uint8_t* obs = nullptr, bool with_noise = false) {
  RArgs r;
  memset(&r, 0, sizeof(r));
  r.N = e->k.N;
  r.env0 = env0;
  r.C = e->k.m.C;
  const DevCam& c = e->k.cam;
  r.cam.H = c.H; r.cam.W = c.W; r.cam.wpr = c.wpr; r.cam.band_rows = c.band_rows; r.cam.n_bands = c.n_bands;
  r.cam.thickness = c.thickness; r.cam.format = c.format;
  {  // Circle(center, radius, fill) of drawing.cpp run once on the host: per-row half widths
    int radius = (int)((((long long)c.thickness << 15) + 32768) >> 16);
    r.cam.cap_r = radius;
    if (radius < 32) {
      int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
      while (dx >= dy) {
        if (r.cam.cap_hw[dy] < dx) r.cam.cap_hw[dy] = (unsigned char)dx;
        if (r.cam.cap_hw[dx] < dy) r.cam.cap_hw[dx] = (unsigned char)dy;
        dy++;
        err += plus;
        plus += 2;
        int mask2 = (err <= 0) - 1;
        err -= minus & mask2;
        dx += mask2;
        minus -= mask2 & 2;
      }
    }
  }
  r.cam.cap_hw4 = r.cam.cap_hw[0] | (r.cam.cap_hw[1] << 8) | (r.cam.cap_hw[2] << 16) | ((unsigned)r.cam.cap_hw[3] << 24);
  memcpy(r.colors, e->k.m.colors, sizeof(r.colors));
  for (int c = 0; c < 16; c++) r.colors_packed[c] = r.colors[c][0] | (r.colors[c][1] << 8) | ((unsigned)r.colors[c][2] << 16);
  r.seg_g = seg_g;
  r.seg_n = seg_n;
  r.seg_cap = seg_cap;
  r.obs = obs ? obs : e->k.b.obs;
  r.mask = mask;
  r.off_tab = e->r_off_tab;
  r.off_bits = e->r_off_bits;
  r.flags = flags;
  r.seg_row0 = 0;
  r.noise_row0 = 0;
  r.obs_row_stride = 0;
  if (with_noise && e->noise_blobs > 0 && c.format == TC_FMT_CLASSES) {
    r.noise_blobs = e->noise_blobs;
    r.noise_max_radius = e->noise_max_radius;
    r.noise_hw = e->noise_hw;
    r.noise_seed = e->noise_seed;
    r.noise_step = e->noise_step;
  }
  return r;
}

static int launch_raster(tc_env* e, const int* seg_g, const int* seg_n, int seg_cap, const uint8_t* mask, uint32_t flags,
                         void* stream, int env0 = 0, int count = -1, uint8_t* obs = nullptr, bool with_noise = false) {
#ifdef TC_TIMING
  if (g_tstamp && e->k.N > g_tstamp_n) {
    set_err("timing build: the installed stamp buffer is smaller than this env batch");
    return TC_E_INVALID;
  }
#endif
  RArgs r = make_rargs(e, seg_g, seg_n, seg_cap, mask, flags, env0, obs, with_noise);
  if (count < 0) count = e->k.N;
#ifdef TC_DEV_FAST
  auto kern = tc_raster_kernel<true, TC_DEV_FMTV>;
#else
  const bool thick = r.cam.thickness > 1, cls = r.cam.format == TC_FMT_CLASSES;
  auto kern = thick ? (cls ? tc_raster_kernel<true, TC_FMT_CLASSES> : tc_raster_kernel<true, TC_FMT_RGB>)
                    : (cls ? tc_raster_kernel<false, TC_FMT_CLASSES> : tc_raster_kernel<false, TC_FMT_RGB>);
#endif
  hipLaunchKernelGGL(kern, dim3(count), dim3(TC_NT), e->r_lds, (hipStream_t)stream, r);
  HIP_TRY(hipGetLastError());
  return TC_OK;
}

static int noise_advance(tc_env* e, int mode, bool rendered, int nsteps, void* stream) {
  // the fused noise consumed one position of the blob stream per step of this launch
  if (rendered && mode == MODE_STEP && e->noise_blobs > 0 && e->k.cam.format == TC_FMT_CLASSES) {
    hipLaunchKernelGGL(tc_noise_tick, dim3(1), dim3(1), 0, (hipStream_t)stream, e->noise_step, (unsigned int)nsteps);
    HIP_TRY(hipGetLastError());
  }
  return TC_OK;
}

// steps per simulate / frame dispatch of a K-step call that renders (see launch()): TC_CHUNK (16), or a quarter of the
// call when that is less, so that a short call (the 20 steps of a smoke benchmark) still has several chunks in flight;
// never more than a ring slot holds.  Calls that are not pipelined run in chunks as large as the ring allows.
static int chunk_steps(const tc_env* e, int nsteps, bool frames, bool all) {
  const bool can_pipe = frames && e->env_grouped && e->pipe && all && nsteps > 1;
  int c = nsteps;
  if (can_pipe) {
    const int q = (nsteps + 3) / 4;
    c = e->chunk < q ? e->chunk : (q < 2 ? 2 : q);
  }
  if (e->ring_rows > 0 && c > e->ring_rows) c = e->ring_rows;
  return c;
}

static bool fused_path(const tc_env* e, uint32_t flags) {
  const bool do_raster = !(flags & (TC_F_NO_OBSERVATION | DBG_SKIP_CAMERA)) && e->k.b.obs;
  return do_raster && e->fuse && e->kvar != 13;
}

#define TC_STREAM_MAX_ROWS 128
// scratch of streamed calls: rows for min(max_call_steps, TC_STREAM_MAX_ROWS) steps
static int reserve_stream(tc_env* e, int max_call_steps) {
  int rows = max_call_steps < TC_STREAM_MAX_ROWS ? max_call_steps : TC_STREAM_MAX_ROWS;
  // the scratch of a row is N x (pose row + draw-list length + room for a whole draw list: 20 B x lane-line edges of the
  // map -- knuffingen: 15 KB per frame); with many envs on a big map the rows are cut so that it stays within a budget
  // (TC_STREAM_SCRATCH_MB, default 16 GiB), and a longer call runs as more, shorter segments
  {
    const double per_row = (double)e->k.N * ((double)e->k.seg_cap * 5 * sizeof(int) + sizeof(int) + TC_POSE_ROW * sizeof(double));
    double budget = 16384.0 * 1048576.0;
    if (const char* sb = getenv("TC_STREAM_SCRATCH_MB")) budget = atof(sb) * 1048576.0;
    const double fit = budget / per_row;
    if (fit < (double)rows) rows = (int)fit;
  }
  if (rows < 2) rows = 2;
  if (e->st_rows >= rows) return TC_OK;
  HIP_TRY(hipDeviceSynchronize());  // earlier launches may still use the old arrays
  if (e->st_segm_g) (void)hipFree(e->st_segm_g);
  if (e->st_segm_n) (void)hipFree(e->st_segm_n);
  if (e->st_pose) (void)hipFree(e->st_pose);
  e->st_segm_g = e->st_segm_n = nullptr;
  e->st_pose = nullptr;
  e->st_rows = 0;
  e->cost_row[0] = e->cost_row[1] = nullptr;
  e->draw_n = 0;
  const size_t R = (size_t)rows * e->k.N;
  void *p = nullptr, *q = nullptr, *pr = nullptr;
  hipError_t he = hipMalloc(&p, R * e->k.seg_cap * 5 * sizeof(int));
  if (he == hipSuccess) he = hipMalloc(&q, R * sizeof(int));
  if (he == hipSuccess) he = hipMalloc(&pr, R * TC_POSE_ROW * sizeof(double));
  if (he == hipSuccess && !e->st_words) {
    void* w = nullptr;
    he = hipMalloc(&w, 256);
    if (he == hipSuccess) he = hipMemset(w, 0, 256);
    e->st_words = (unsigned int*)w;
  }
  if (he == hipSuccess) he = hipMemset(pr, 0xFF, R * TC_POSE_ROW * sizeof(double));  // every entry TC_POSE_EMPTY
  if (he == hipSuccess) he = hipMemset(q, 0, R * sizeof(int));
  if (he == hipSuccess) he = hipDeviceSynchronize();
  if (he != hipSuccess) {
    if (p) (void)hipFree(p);
    if (q) (void)hipFree(q);
    if (pr) (void)hipFree(pr);
    set_err(std::string("hipMalloc(scratch of streamed K-step calls): ") + hipGetErrorString(he));
    return TC_E_NOMEM;
  }
  e->st_segm_g = (int*)p;
  e->st_segm_n = (int*)q;
  e->st_pose = (double*)pr;
  e->st_rows = rows;
  return TC_OK;
}

extern "C" int tc_env_reserve_steps(tc_env* e, int32_t max_call_steps) {
  if (!e || max_call_steps < 1) return TC_E_INVALID;
  if (e->stream && e->env_grouped && e->pipe) {
    int rc = reserve_stream(e, max_call_steps);
    if (rc != TC_OK) return rc;
  }
  int rows = max_call_steps < e->chunk ? max_call_steps : e->chunk;
  if (rows < 2) rows = 2;  // (a pipelined call never uses chunks of fewer than 2 steps)
  if (e->ring_rows >= rows) return TC_OK;
  HIP_TRY(hipDeviceSynchronize());  // earlier launches may still read the old ring
  if (e->segm_g) (void)hipFree(e->segm_g);
  if (e->segm_n) (void)hipFree(e->segm_n);
  if (e->pose_rows) (void)hipFree(e->pose_rows);
  e->segm_g = e->segm_n = nullptr;
  e->pose_rows = nullptr;
  e->ring_rows = 0;
  e->cost_row[0] = e->cost_row[1] = nullptr;
  const size_t R = (size_t)TC_RING_SLOTS * rows * e->k.N;
  void *p = nullptr, *q = nullptr, *pr = nullptr;
  hipError_t he = hipMalloc(&p, R * e->k.seg_cap * 5 * sizeof(int));
  if (he == hipSuccess) he = hipMalloc(&q, R * sizeof(int));
  if (he == hipSuccess) he = hipMalloc(&pr, R * TC_POSE_ROW * sizeof(double));
  if (he != hipSuccess) {
    if (p) (void)hipFree(p);
    if (q) (void)hipFree(q);
    set_err(std::string("hipMalloc(scratch ring of K-step calls): ") + hipGetErrorString(he));
    return TC_E_NOMEM;
  }
  e->segm_g = (int*)p;
  e->segm_n = (int*)q;
  e->pose_rows = (double*)pr;
  e->ring_rows = rows;
  return TC_OK;
}

static int launch(tc_env* e, int mode, const void* cc, int cdtype, const int32_t* man, const int32_t* spawn,
                  const uint8_t* mask, uint32_t flags, void* stream, int nsteps = 1, const tc_rollout* roll = nullptr) {
  if (!e) return TC_E_INVALID;
  if (!e->bound) {
    set_err("tc_env_bind has not been called");
    return TC_E_UNBOUND;
  }
  if ((flags & TC_F_AUTORESET) && !e->k.b.needs_reset) {
    set_err("TC_F_AUTORESET needs needs_reset/spawn_queue/spawn_cursor buffers");
    return TC_E_INVALID;
  }
  if ((flags & TC_F_DEVICE_SPAWN) && e->k.spawn_n < 1) {
    set_err("TC_F_DEVICE_SPAWN needs a table (tc_env_set_spawn_table)");
    return TC_E_INVALID;
  }
#ifdef TC_TIMING
  if (g_tstamp && e->k.N > g_tstamp_n) {
    set_err("timing build: the installed stamp buffer is smaller than this env batch");
    return TC_E_INVALID;
  }
#endif
  const bool prof = e->prof > 0 && mode == MODE_STEP && (e->prof_calls++ % e->prof) == 0;
  const int slot = e->prof_n % TC_PROF_RING;
  const int kv = e->kvar;
  // simulate stage alone, with or without the camera stage compiled in (without: fewer registers, half the code)
#ifdef TC_DEV_FAST
  auto kern = tc_env_kernel<TC_DEV_KV, true>;
  auto kern_nocam = tc_env_kernel<TC_DEV_KV, false>;
#else
  auto kern = kv == 5 ? tc_env_kernel<5, true> : kv == 8 ? tc_env_kernel<8, true> : kv == 9 ? tc_env_kernel<9, true> : tc_env_kernel<13, true>;
  auto kern_nocam = kv == 5 ? tc_env_kernel<5, false> : kv == 8 ? tc_env_kernel<8, false> : kv == 9 ? tc_env_kernel<9, false> : tc_env_kernel<13, false>;
#endif
  const bool do_raster = !(flags & (TC_F_NO_OBSERVATION | DBG_SKIP_CAMERA)) && (e->k.b.obs || (roll && roll->obs));
  const int N = e->k.N;
  MultiArgs ma;
  memset(&ma, 0, sizeof(ma));
  ma.nsteps = nsteps;
  if (roll) ma.roll = *roll;
  hipStream_t main = (hipStream_t)stream;
  if (do_raster && nsteps > 1 && (e->multi_split || !e->fuse || kv == 13)) {
    // K steps, split form: per chunk ONE simulate launch over its steps and ONE launch over the frames wanted.  The pose
    // rows / draw lists of a chunk live in slot (chunk index mod TC_RING_SLOTS) of the scratch ring.
    if (e->ring_rows < 1) {
      set_err("tc_step_multi with observations needs the scratch ring: call tc_env_reserve_steps(env, n) once before "
              "(tc_step_multi itself never allocates)");
      return TC_E_INVALID;
    }
    if (prof) HIP_TRY(hipEventRecord(e->ev[0][slot], main));
    // one stream at a time per handle: a call on another stream than the previous K-step call waits for that call
    if (e->have_call && e->last_stream != main) HIP_TRY(hipStreamWaitEvent(main, e->call_ev, 0));
    const bool frames = e->fuse && kv != 13;  // camera + raster per frame (tc_frame_kernel); else camera in the simulate
                                              // launch (the register-hungry K = 13 stage, TC_FUSE=0) and a raster launch
    // Pipelining inside the call.  The steps are issued in chunks: the simulate launch of chunk c+1 runs on the caller's
    // stream while the frames of chunk c are produced on an internal stream (joined back before the call's work ends
    // on the caller's stream, so the caller still sees everything complete in stream order).  The two kernels suit
    // each other: the frame kernel is bound by vector issue, the grouped simulate kernel by latency (512 wavefronts
    // for 4096 envs), so the second hides in the first's shadow instead of adding its 21 us per step.
    // (Measured and dropped, DESIGN.md section 4: chunk sizes ramping up 1, 2, 4, ...; a short or half first chunk; other
    // divisors than 4 for short calls.)
    const bool all = roll && roll->obs;  // with a rollout every step's frame is wanted; else only the last survives
    const bool piped = frames && e->env_grouped && e->pipe && all && nsteps > 1;
    const int chunk = chunk_steps(e, nsteps, frames, all);
    hipStream_t fs = piped ? e->frame_stream : main;
    // Short calls (chunks of fewer than 8 steps): a frame launch of 5 x N workgroups spends a good part of its life
    // ramping up and draining (5 rows: 34.6 us per row alone, 16 rows: 30.8), and a 20-step call is four of them.  There
    // the frame launches of consecutive chunks go to two streams alternately -- each still behind its own chunk's
    // simulate launch -- so that chunk c+1's workgroups fill the slots chunk c's tail leaves empty (cfg3: 8-step call
    // 66.2 -> 58.3 us per step, 20-step call 46.9 -> 44.0; with chunks of 10 steps it costs 7 %, so longer calls keep one
    // stream and their frame launches follow one another, as a kernel trace of the default command shows them).
    const bool two_fs = piped && e->frame_streams == 2 && chunk < 8;
    const size_t esz = cdtype == TC_F32 ? 4 : 8;
    const bool thick = e->k.cam.thickness > 1, cls = e->k.cam.format == TC_FMT_CLASSES;
    (void)thick;
    (void)cls;
    bool first_frames = true;
    bool used_fs[2] = {false, false};
    e->draw_n = 0;
    e->draw_base = e->segm_n;
    const int C = e->k.m.C;
    // Streamed form (default when every step's frame is wanted).  The chunked pipeline below pays for its structure: the
    // first chunk's simulate launch overlaps with nothing, and every frame dispatch ends in a tail with the chip half empty
    // -- a quarter of a 20-step call.  Here the whole call (or a segment of st_rows steps) is ONE simulate launch on the
    // caller's stream and ONE frame launch of steps x N workgroups on the internal stream that starts right away, beside
    // it: a frame workgroup polls its pose row until the simulate launch has written it (device-scope loads; every entry
    // validates itself, tc_frame_kernel).  Only the first step is exposed, and there is one tail per call.
    //   frame stream: [order kernel] [tc_gate_kernel: until the simulate workgroups are resident] [tc_frame_kernel, gate 1]
    //                 (wait: simulate launch done) [tc_frame_recover_kernel: whatever a gate-1 workgroup gave up on]
    // Every wait on the device is bounded (gate_ticks), and what a bound cuts short the recover pass completes: the
    // result never depends on how the two launches were scheduled.
    const bool streamed = piped && e->stream && e->st_rows >= 2 && e->st_words;
    if (streamed) {
      e->draw_base = e->st_segm_n;
      const int seg_steps = e->st_rows;
      int si = 0;
      for (int c0 = 0, cn = 0; c0 < nsteps; c0 += cn, si++) {
        cn = nsteps - c0 < seg_steps ? nsteps - c0 : seg_steps;
        const size_t r0 = (size_t)c0 * N;
        // a later segment reuses the scratch rows: the frames of the one before must be drawn (and its rows cleared)
        if (si > 0) HIP_TRY(hipStreamWaitEvent(main, e->slot_ev[0], 0));
        HIP_TRY(hipEventRecord(e->start_ev, main));
        StepArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.a = e->k;
        sa.a.env0 = 0;
        sa.a.seg_g = e->st_segm_g;
        sa.a.seg_n = e->st_segm_n;
        sa.ma = ma;
        sa.ma.nsteps = cn;
        sa.ma.seg_rows = cn > 1 ? cn : 2;
        sa.ma.cam_here = 0;
        sa.ma.pose_rows = e->st_pose;
        sa.ma.resident = e->st_words;
        if (roll) {
          tc_rollout& q = sa.ma.roll;
          q.obs = roll->obs + r0 * (size_t)e->obs_bytes;
          q.reward = roll->reward ? roll->reward + r0 : nullptr;
          q.terminated = roll->terminated ? roll->terminated + r0 : nullptr;
          q.truncated = roll->truncated ? roll->truncated + r0 : nullptr;
          q.cte = roll->cte ? roll->cte + r0 : nullptr;
          q.heading_error = roll->heading_error ? roll->heading_error + r0 : nullptr;
          q.status = roll->status ? roll->status + r0 : nullptr;
          q.x = roll->x ? roll->x + r0 : nullptr;
          q.y = roll->y ? roll->y + r0 : nullptr;
          q.theta = roll->theta ? roll->theta + r0 : nullptr;
          q.velocity = roll->velocity ? roll->velocity + r0 : nullptr;
          q.laneline_distances = roll->laneline_distances ? roll->laneline_distances + r0 * C : nullptr;
          q.nearest_edge = roll->nearest_edge ? roll->nearest_edge + r0 * C : nullptr;
          q.local_path = roll->local_path ? roll->local_path + r0 * 8 : nullptr;
          q.lp_len = roll->lp_len ? roll->lp_len + r0 : nullptr;
        }
        sa.mode = mode;
        sa.cdtype = cdtype;
        sa.flags = flags;
        sa.car_control = (const char*)cc + r0 * 2 * esz;
        sa.maneuver = man + r0;
        sa.spawn_nodes = spawn;
        sa.mask = mask;
        const size_t map_bytes = ((size_t)e->k.m.total_edges * 48 + 15) / 16 * 16, fat_bytes = (size_t)e->k.m.lpN * sizeof(LpNode);
        sa.ma.map_lds = (e->envg_map_lds && map_bytes <= 40 * 1024) ? 1 : 0;
        sa.ma.fat_lds = (e->envg_map_lds && fat_bytes <= 56 * 1024) ? 1 : 0;
        const unsigned int sim_wgs = (unsigned int)((N + TC_ENVG_NT / TC_EL - 1) / (TC_ENVG_NT / TC_EL));
        hipLaunchKernelGGL(tc_envg_kernel, dim3(sim_wgs), dim3(TC_ENVG_NT), (sa.ma.map_lds ? map_bytes : 0) + (sa.ma.fat_lds ? fat_bytes : 0),
                           main, sa);
        HIP_TRY(hipGetLastError());
        if (prof && c0 + cn >= nsteps) HIP_TRY(hipEventRecord(e->ev[1][slot], main));
        HIP_TRY(hipEventRecord(e->sim_ev, main));
        // ---- the frame stream
        hipStream_t fs = e->frame_stream;
        // heaviest frames first, by the last row of the call before (see the chunked form below).  Depends on nothing but
        // this stream's own past, so it runs while the simulate launch starts
        if (e->frame_order[0] && e->cost_row[0]) {
          hipLaunchKernelGGL(tc_order_kernel, dim3(1), dim3(TC_ORDER_NT), (size_t)(N + 15) / 16 * 16, fs, e->cost_row[0], N, N, e->frame_order[0]);
          HIP_TRY(hipGetLastError());
          e->frame_order_valid[0] = true;
        }
        HIP_TRY(hipStreamWaitEvent(fs, e->start_ev, 0));
        RArgs r = make_rargs(e, e->st_segm_g, e->st_segm_n, e->k.seg_cap, nullptr, flags, 0, roll->obs + r0 * (size_t)e->obs_bytes,
                             mode == MODE_STEP);
        r.seg_row0 = 0;
        r.noise_row0 = c0;
        r.obs_row_stride = (long long)N * (long long)e->obs_bytes;
        FrameArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.a = e->k;
        fa.a.dbg = flags;
        fa.a.env0 = 0;
        fa.a.seg_g = e->st_segm_g;
        fa.a.seg_n = e->st_segm_n;
        fa.r = r;
        fa.pose_rows = e->st_pose;
        fa.a.seg_lds_off = fa.r.seg_lds_off = e->seg_lds_off;
        fa.a.seg_lds_cap = fa.r.seg_lds_cap = e->frame_seg_cap;
        fa.abort_word = e->st_words + 1;
        fa.resident = e->st_words;
        fa.gate_ticks = e->gate_ticks;
        fa.gate_test = e->gate_test;
        if (e->frame_order[0] && e->frame_order_valid[0]) fa.order = e->frame_order[0];
        // (ten times the patience of a frame workgroup: one idle wavefront costs nothing, and when another env's frame
        // launch has the chip -- two handles stepping on one GPU -- the simulate workgroups only get on as that launch drains)
        hipLaunchKernelGGL(tc_gate_kernel, dim3(1), dim3(64), 0, fs, (const unsigned int*)e->st_words, sim_wgs, 10 * e->gate_ticks);
        HIP_TRY(hipGetLastError());
        if (prof && si == 0) HIP_TRY(hipEventRecord(e->ev[3][slot], fs));
        frame_kern_t fk = frame_kernel_of(e->kframe, thick, cls);
        fa.gate = 1;
        hipLaunchKernelGGL(fk, dim3(N, cn), dim3(TC_NT), e->frame_lds, fs, fa);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamWaitEvent(fs, e->sim_ev, 0));
        fa.gate = 2;
        fa.recover_rows = cn;
        fa.order = nullptr;
        frame_kern_t rk = recover_kernel_of(e->kframe, thick, cls);
        hipLaunchKernelGGL(rk, dim3(N), dim3(TC_NT), e->frame_lds, fs, fa);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(e->slot_ev[0], fs));
        if (e->frame_order[0]) e->cost_row[0] = e->st_segm_n + (size_t)(cn - 1) * N;
        used_fs[0] = true;
        const int keep = cn < 3 * 16 ? cn : 3 * 16;  // the statistics cover the same span as a chunked call's ring
        e->draw_rows[0][0] = cn - keep;
        e->draw_rows[0][1] = keep;
        e->draw_n = 1;
      }
    }
    int ci = 0;
    for (int c0 = 0, cn = 0; !streamed && c0 < nsteps; c0 += cn, ci++) {
      cn = nsteps - c0 < chunk ? nsteps - c0 : chunk;
      const size_t r0 = (size_t)c0 * N;                 // first [step][env] row of this chunk in the caller's arrays
      const int rslot = ci % TC_RING_SLOTS;
      const size_t rb = (size_t)rslot * e->ring_rows;   // first row of this chunk in the scratch ring
      // the frames that read this ring slot TC_RING_SLOTS chunks ago must be done before it is rewritten
      if (piped && ci >= TC_RING_SLOTS) HIP_TRY(hipStreamWaitEvent(main, e->slot_ev[rslot], 0));
      StepArgs sa;
      memset(&sa, 0, sizeof(sa));
      sa.a = e->k;
      sa.a.env0 = 0;
      sa.a.seg_g = e->segm_g + rb * N * e->k.seg_cap * 5;
      sa.a.seg_n = e->segm_n + rb * N;
      sa.ma = ma;
      sa.ma.nsteps = cn;
      sa.ma.seg_rows = cn > 1 ? cn : 2;  // (> 1: row k of the chunk's lists; a one-step chunk still writes row 0 of them)
      sa.ma.cam_here = frames ? 0 : 1;
      sa.ma.pose_rows = frames ? e->pose_rows + rb * N * TC_POSE_ROW : nullptr;
      if (roll) {
        tc_rollout& q = sa.ma.roll;
        q.obs = roll->obs ? roll->obs + r0 * (size_t)e->obs_bytes : nullptr;
        q.reward = roll->reward ? roll->reward + r0 : nullptr;
        q.terminated = roll->terminated ? roll->terminated + r0 : nullptr;
        q.truncated = roll->truncated ? roll->truncated + r0 : nullptr;
        q.cte = roll->cte ? roll->cte + r0 : nullptr;
        q.heading_error = roll->heading_error ? roll->heading_error + r0 : nullptr;
        q.status = roll->status ? roll->status + r0 : nullptr;
        q.x = roll->x ? roll->x + r0 : nullptr;
        q.y = roll->y ? roll->y + r0 : nullptr;
        q.theta = roll->theta ? roll->theta + r0 : nullptr;
        q.velocity = roll->velocity ? roll->velocity + r0 : nullptr;
        q.laneline_distances = roll->laneline_distances ? roll->laneline_distances + r0 * C : nullptr;
        q.nearest_edge = roll->nearest_edge ? roll->nearest_edge + r0 * C : nullptr;
        q.local_path = roll->local_path ? roll->local_path + r0 * 8 : nullptr;
        q.lp_len = roll->lp_len ? roll->lp_len + r0 : nullptr;
      }
      sa.mode = mode;
      sa.cdtype = cdtype;
      sa.flags = flags;
      sa.car_control = (const char*)cc + r0 * 2 * esz;
      sa.maneuver = man + r0;
      sa.spawn_nodes = spawn;
      sa.mask = mask;
      // The first chunk's simulate launch overlaps with nothing, so it should be SHORT rather than cheap: it goes through
      // the one-wavefront-per-env kernel (4096 wavefronts, bound by throughput: ~15 us per step) instead of the grouped
      // one (512 wavefronts, a latency chain: 13-23 us per step).  20-step call 48.4 -> 47.2 us per step, 128-step calls
      // 38.2 -> 37.6.  Both kernels read and leave the env's state in the caller's buffers, bit for bit the same.
      if (frames && e->env_grouped && !(e->first_per_env && c0 == 0 && piped && nsteps > chunk)) {  // (a one-chunk call: grouped)
        // LDS copies of the lane-line edge records (48 B per edge) and of the lanepath's fat node records (96 B per node):
        // simple_layout 12.4 + 17.5 KB, knuffingen 34.6 + 40.2 KB per workgroup of 32 envs (a CU holds one or two such
        // workgroups: 128 of them cover 4096 envs)
        const size_t map_bytes = ((size_t)e->k.m.total_edges * 48 + 15) / 16 * 16, fat_bytes = (size_t)e->k.m.lpN * sizeof(LpNode);
        sa.ma.map_lds = (e->envg_map_lds && map_bytes <= 40 * 1024) ? 1 : 0;
        sa.ma.fat_lds = (e->envg_map_lds && fat_bytes <= 56 * 1024) ? 1 : 0;
        hipLaunchKernelGGL(tc_envg_kernel, dim3((N + TC_ENVG_NT / TC_EL - 1) / (TC_ENVG_NT / TC_EL)), dim3(TC_ENVG_NT),
                           (sa.ma.map_lds ? map_bytes : 0) + (sa.ma.fat_lds ? fat_bytes : 0), main, sa);
      }
      else
        hipLaunchKernelGGL(frames ? kern_nocam : kern, dim3(N), dim3(TC_NT), e->k.lds.total, main, sa);
      HIP_TRY(hipGetLastError());
      const bool last_chunk = c0 + cn >= nsteps;
      if (prof && last_chunk) HIP_TRY(hipEventRecord(e->ev[1][slot], main));
      if (!all && !last_chunk) continue;  // only the last step's frame is wanted
      if (two_fs) fs = (ci & 1) ? e->frame_stream2 : e->frame_stream;
      if (piped) {
        HIP_TRY(hipEventRecord(e->sim_ev, main));
        HIP_TRY(hipStreamWaitEvent(fs, e->sim_ev, 0));
        used_fs[fs == e->frame_stream2 ? 1 : 0] = true;
      }
      if (prof && first_frames && piped) HIP_TRY(hipEventRecord(e->ev[3][slot], fs));
      first_frames = false;
      RArgs r = make_rargs(e, e->segm_g, e->segm_n, e->k.seg_cap, nullptr, flags, 0, all ? roll->obs + r0 * (size_t)e->obs_bytes : nullptr,
                           mode == MODE_STEP);
      // scratch row of the first frame drawn / its step index in the call (position in the blob stream)
      r.seg_row0 = (int)rb + (all ? 0 : cn - 1);
      r.noise_row0 = all ? c0 : nsteps - 1;
      r.obs_row_stride = all ? (long long)N * (long long)e->obs_bytes : 0;
      const int rows = all ? cn : 1;  // (<= ring_rows <= 16: one grid)
      if (frames) {
        FrameArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.a = e->k;
        fa.a.dbg = flags;
        fa.a.env0 = 0;
        fa.a.seg_g = e->segm_g;
        fa.a.seg_n = e->segm_n;
        fa.r = r;
        fa.pose_rows = e->pose_rows;
        frame_kern_t fk = frame_kernel_of(e->kframe, thick, cls);
        fa.a.seg_lds_off = fa.r.seg_lds_off = e->seg_lds_off;
        fa.a.seg_lds_cap = fa.r.seg_lds_cap = e->frame_seg_cap;
        // Heaviest frames first.  A dispatch ends with a tail -- the chip half empty while the last workgroups finish, ~36 us
        // of a 16-step dispatch, a fifth of a 5-step one -- and the dispatcher hands workgroups out in index order, so the
        // envs are sorted by the draw-list lengths of the last frame row drawn on this stream (tc_order_kernel with one
        // group: plain descending order): the last workgroups to start are then the cheap, mostly empty frames.
        const int fsi = fs == e->frame_stream2 ? 1 : 0;
        // (a re-sort costs ~5 us of the frame stream -- kernel + launch -- so short dispatches, whose tails already overlap on
        // two streams, re-sort only once per call and otherwise keep the stream's last order)
        if (e->frame_order[fsi] && e->cost_row[fsi] && (rows >= 8 || ci < 2)) {
          hipLaunchKernelGGL(tc_order_kernel, dim3(1), dim3(TC_ORDER_NT), (size_t)(N + 15) / 16 * 16, fs, e->cost_row[fsi], N, N, e->frame_order[fsi]);
          HIP_TRY(hipGetLastError());
          e->frame_order_valid[fsi] = true;
        }
        if (e->frame_order[fsi] && e->frame_order_valid[fsi]) fa.order = e->frame_order[fsi];
        hipLaunchKernelGGL(fk, dim3(N, rows), dim3(TC_NT), e->frame_lds, fs, fa);
        if (e->frame_order[fsi]) e->cost_row[fsi] = e->segm_n + ((size_t)r.seg_row0 + (size_t)rows - 1) * N;
      } else {
#ifdef TC_DEV_FAST
        auto rk = tc_raster_kernel<true, TC_DEV_FMTV>;
#else
        auto rk = thick ? (cls ? tc_raster_kernel<true, TC_FMT_CLASSES> : tc_raster_kernel<true, TC_FMT_RGB>)
                        : (cls ? tc_raster_kernel<false, TC_FMT_CLASSES> : tc_raster_kernel<false, TC_FMT_RGB>);
#endif
        hipLaunchKernelGGL(rk, dim3(N, rows), dim3(TC_NT), e->r_lds, fs, r);
      }
      HIP_TRY(hipGetLastError());
      if (piped) HIP_TRY(hipEventRecord(e->slot_ev[rslot], fs));
      {  // (the ring keeps the last TC_RING_SLOTS chunks: remember where their draw-list lengths are)
        const int w = e->draw_n < TC_RING_SLOTS ? e->draw_n++ : (memmove(e->draw_rows[0], e->draw_rows[1], sizeof(int) * 2 * (TC_RING_SLOTS - 1)), TC_RING_SLOTS - 1);
        e->draw_rows[w][0] = r.seg_row0;
        e->draw_rows[w][1] = rows;
      }
    }
    // join the frame stream(s) back; the end-of-frames probe then sits on the caller's stream behind the join
    if (piped) {
      if (used_fs[0]) {
        HIP_TRY(hipEventRecord(e->frames_ev, e->frame_stream));
        HIP_TRY(hipStreamWaitEvent(main, e->frames_ev, 0));
      }
      if (used_fs[1]) {
        HIP_TRY(hipEventRecord(e->frames_ev2, e->frame_stream2));
        HIP_TRY(hipStreamWaitEvent(main, e->frames_ev2, 0));
      }
    }
    if (prof) {
      HIP_TRY(hipEventRecord(e->ev[2][slot], main));
      e->prof_piped[slot] = piped ? 1 : 0;
      e->prof_n++;
    }
    int rc = noise_advance(e, mode, true, nsteps, stream);
    if (rc != TC_OK) return rc;
    HIP_TRY(hipEventRecord(e->call_ev, main));
    e->last_stream = main;
    e->have_call = true;
    return TC_OK;
  }
  if (prof) HIP_TRY(hipEventRecord(e->ev[0][slot], main));
  if (do_raster && e->fuse && kv != 13) {  // one launch: simulate + raster by the same wavefront
    // (the register-hungry K = 13 simulate stage spills when fused, so it stays two launches)
    const bool thick = e->k.cam.thickness > 1, cls = e->k.cam.format == TC_FMT_CLASSES;
    // (component groups that fit the K = 5 register cache: the K = 5 kernel with 16-segment batches, as for the frame kernel --
    // its simulate stage then walks the map's lane-line nodes in windows of 320 instead of 576)
    fused_kern_t fk = e->kframe == 516 ? pick_fused<5, 16>(thick, cls)
                      : kv == 5        ? pick_fused<5>(thick, cls)
                      : kv == 8        ? pick_fused<8>(thick, cls)
                                       : pick_fused<9>(thick, cls);
    KArgs k = e->k;
    k.env0 = 0;
    RArgs r = make_rargs(e, e->k.seg_g, e->k.seg_n, e->k.seg_cap, nullptr, flags, 0, nullptr, mode == MODE_STEP);
    // covers both stages and the parked state; behind it, up to the size that costs the CU no workgroup, the head of the
    // frame's draw list (see tc_env_create)
    const int lds = e->step_lds;
    k.seg_lds_off = r.seg_lds_off = e->k.lds.total;
    k.seg_lds_cap = r.seg_lds_cap = e->seg_lds_cap ? (e->step_lds - e->k.lds.total) / 20 : 0;
    if (k.seg_lds_cap > e->seg_lds_limit) k.seg_lds_cap = r.seg_lds_cap = e->seg_lds_limit;
    StepArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.a = k;
    sa.r = r;
    sa.ma = ma;
    sa.mode = mode;
    sa.cdtype = cdtype;
    sa.flags = flags;
    sa.car_control = cc;
    sa.maneuver = man;
    sa.spawn_nodes = spawn;
    sa.mask = mask;
    if (e->env_order && nsteps == 1) {
      // every order_every-th step the envs are re-dealt to the workgroups by the draw-list lengths of the step before
      // (tc_order_kernel: a 1-workgroup launch of a few microseconds on the caller's stream)
      if (mode == MODE_STEP && e->order_calls++ % e->order_every == 0 && e->order_calls > 1) {
        hipLaunchKernelGGL(tc_order_kernel, dim3(1), dim3(TC_ORDER_NT), (size_t)(N + 15) / 16 * 16, main, (const int*)e->k.seg_n, N, e->order_g, e->env_order);
        HIP_TRY(hipGetLastError());
      }
      sa.env_order = e->env_order;
    }
    hipLaunchKernelGGL(fk, dim3(N), dim3(TC_NT), lds, main, sa);
    HIP_TRY(hipGetLastError());
    e->draw_n = -1;
    if (prof) {  // one kernel: its whole duration is reported as the first interval, the second is empty
      HIP_TRY(hipEventRecord(e->ev[1][slot], main));
      HIP_TRY(hipEventRecord(e->ev[2][slot], main));
      e->prof_n++;
    }
    return noise_advance(e, mode, true, nsteps, stream);
  }
  {
    KArgs k = e->k;
    k.env0 = 0;
    StepArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.a = k;
    sa.ma = ma;
    sa.ma.cam_here = do_raster ? 1 : 0;
    sa.mode = mode;
    sa.cdtype = cdtype;
    sa.flags = flags;
    sa.car_control = cc;
    sa.maneuver = man;
    sa.spawn_nodes = spawn;
    sa.mask = mask;
    // (K steps without observations run on the one-wavefront-per-env kernel: with nothing to share the chip with, its
    // shorter serial chain wins -- cfg2: 16.5 us per step against 20.7 for the grouped kernel)
    hipLaunchKernelGGL(do_raster ? kern : kern_nocam, dim3(N), dim3(TC_NT), k.lds.total, main, sa);
    HIP_TRY(hipGetLastError());
    if (prof) HIP_TRY(hipEventRecord(e->ev[1][slot], main));
    if (do_raster) {
      int rc = launch_raster(e, e->k.seg_g, e->k.seg_n, e->k.seg_cap, mode == MODE_RESET ? mask : nullptr, flags, main, 0, N,
                             roll ? roll->obs : nullptr, mode == MODE_STEP);
      if (rc != TC_OK) return rc;
      e->draw_n = -1;
    }
  }
  if (prof) {
    HIP_TRY(hipEventRecord(e->ev[2][slot], main));
    e->prof_n++;
  }
  return noise_advance(e, mode, do_raster, nsteps, stream);
}

extern "C" int tc_reset(tc_env* e, const int32_t* spawn_nodes, const uint8_t* mask, uint32_t flags, void* stream) {
  if (!spawn_nodes) return TC_E_INVALID;
  return launch(e, MODE_RESET, nullptr, TC_F32, nullptr, spawn_nodes, mask, flags, stream);
}

extern "C" int tc_step(tc_env* e, const void* car_control, int32_t control_dtype, const int32_t* maneuver,
                       uint32_t flags, void* stream) {
  if (!car_control || !maneuver || (control_dtype != TC_F32 && control_dtype != TC_F64)) return TC_E_INVALID;
  return launch(e, MODE_STEP, car_control, control_dtype, maneuver, nullptr, nullptr, flags, stream);
}

extern "C" int tc_step_multi(tc_env* e, const void* car_control, int32_t control_dtype, const int32_t* maneuver,
                             int32_t n_steps, uint32_t flags, const tc_rollout* rollout, void* stream) {
  if (!e || !car_control || !maneuver || (control_dtype != TC_F32 && control_dtype != TC_F64) || n_steps < 1) return TC_E_INVALID;
  if (!e->bound) {
    set_err("tc_env_bind has not been called");
    return TC_E_UNBOUND;
  }
  const bool want_obs = !(flags & (TC_F_NO_OBSERVATION | DBG_SKIP_CAMERA)) && (e->k.b.obs || (rollout && rollout->obs));
  (void)want_obs;
  return launch(e, MODE_STEP, car_control, control_dtype, maneuver, nullptr, nullptr, flags, stream, n_steps, rollout);
}

// Workload descriptor for benchmark lines: what the most recent frames actually drew.  Waits for the device and copies
// the draw-list lengths (one int per frame) to the host: of the last single step's N frames, or of the frames of the
// last (up to TC_RING_SLOTS) chunks of the last K-step call.
extern "C" int tc_env_draw_list_stats(tc_env* e, double* mean_segments, double* empty_frac, int32_t* max_segments,
                                      int64_t* frames) {
  if (!e || !mean_segments || !empty_frac || !max_segments || !frames) return TC_E_INVALID;
  *mean_segments = *empty_frac = 0;
  *max_segments = 0;
  *frames = 0;
  if (e->draw_n == 0) return TC_OK;
  HIP_TRY(hipDeviceSynchronize());
  const int N = e->k.N;
  std::vector<int> h;
  long long tot = 0, empty = 0, cnt = 0;
  int mx = 0;
  auto take = [&](const int* dev, size_t n) -> int {
    h.resize(n);
    HIP_TRY(hipMemcpy(h.data(), dev, n * sizeof(int), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) {
      tot += h[i];
      empty += h[i] == 0;
      mx = h[i] > mx ? h[i] : mx;
    }
    cnt += (long long)n;
    return TC_OK;
  };
  if (e->draw_n < 0) {
    int rc = take(e->k.seg_n, (size_t)N);
    if (rc != TC_OK) return rc;
  } else {
    for (int i = 0; i < e->draw_n; i++) {
      int rc = take(e->draw_base + (size_t)e->draw_rows[i][0] * N, (size_t)e->draw_rows[i][1] * N);
      if (rc != TC_OK) return rc;
    }
  }
  *frames = cnt;
  *mean_segments = cnt ? (double)tot / (double)cnt : 0.0;
  *empty_frac = cnt ? (double)empty / (double)cnt : 0.0;
  *max_segments = mx;
  return TC_OK;
}

extern "C" int tc_env_launch_info(const tc_env* e, uint32_t flags, int32_t n_steps, int32_t* fused, int32_t* kvar,
                                  int32_t* steps_per_dispatch, char* name, int32_t name_cap) {
  if (!e) return TC_E_INVALID;
  const bool do_raster = !(flags & (TC_F_NO_OBSERVATION | DBG_SKIP_CAMERA)) && e->k.b.obs;
  const bool f = fused_path(e, flags) && !(n_steps > 1 && e->multi_split);
  if (fused) *fused = f ? 1 : 0;
  if (kvar) *kvar = (f && e->kframe == 516) ? 5 : e->kvar;  // (the fused kernel of a map with component groups: K = 5)
  const bool frames = n_steps > 1 && do_raster && !f && e->fuse && e->kvar != 13;
  const bool streamed = frames && e->env_grouped && e->pipe && e->stream && e->st_rows >= 2;  // (as in launch())
  if (steps_per_dispatch)
    *steps_per_dispatch = f ? n_steps : streamed ? (n_steps < e->st_rows ? n_steps : e->st_rows) : chunk_steps(e, n_steps, frames, true);
  if (name && name_cap > 0)
    snprintf(name, (size_t)name_cap, "%s",
             f ? "tc_step_kernel"
               : frames ? (e->env_grouped ? "tc_envg_kernel+tc_frame_kernel" : "tc_env_kernel+tc_frame_kernel")
               : do_raster ? "tc_env_kernel+tc_raster_kernel"
                           : "tc_env_kernel");
  return TC_OK;
}

extern "C" int tc_render_segments(tc_env* e, const int32_t* segments, const int32_t* counts, int32_t capacity,
                                  void* stream) {
  if (!e || !segments || !counts || capacity < 1) return TC_E_INVALID;
  if (!e->bound || !e->k.b.obs) {
    set_err("tc_render_segments needs bound buffers with an observation tensor");
    return TC_E_UNBOUND;
  }
  return launch_raster(e, segments, counts, capacity, nullptr, 0, stream);
}

extern "C" int tc_render(tc_env* e, uint32_t flags, void* stream) {
  return launch(e, MODE_RENDER, nullptr, TC_F32, nullptr, nullptr, nullptr, flags & ~TC_F_NO_OBSERVATION, stream);
}
