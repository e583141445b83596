/* tinycarlo_hip.h -- C ABI of libtinycarlo_hip.so: batched tinycarlo step()/reset() on one MI355X.
 *
 * The reference (emrullahArkun/tinycarlo) is pure Python and has no FFI; its only seam is the
 * gymnasium API of tinycarlo/env.py.  This header is what a Python `TinyCarloEnv` binds through
 * ctypes to move the per-step hot path onto the GPU (binding shown in INTEGRATION.md):
 *
 *   reference interface replaced                                  entry point here
 *   ------------------------------------------------------------  -------------------------
 *   Map.__init__ / __change_scale      tinycarlo/map.py:9-37       tc_map_create
 *   Car.__init__, Camera.__init__      car.py:10-19, camera.py:12-24  tc_env_create
 *   Camera.update_params               camera.py:48-50             tc_env_set_camera
 *   TinyCarloEnv.reset                 env.py:101-113              tc_reset
 *     (Car.reset car.py:34-44, Map.sample_spawn map.py:51-69 with the node index drawn by the host RNG)
 *   TinyCarloEnv.step                  env.py:115-147              tc_step
 *     (Car.step car.py:70-125, Car.find_local_path 127-148, Layer.* layer.py:33-187,
 *      Camera.capture_frame camera.py:52-110, Renderer.render_camera_frame_{rgb,classes}
 *      renderer.py:36-51, Car.get_info car.py:46-68, default reward/termination env.py:87-99)
 *
 * Conventions
 *   - N independent envs live as structure-of-arrays in device memory OWNED BY THE CALLER
 *     (e.g. torch tensors); the library stores the pointers given to tc_env_bind and never frees them.
 *   - every pointer inside tc_buffers, and the action pointers of tc_step / tc_reset, are DEVICE
 *     pointers on the device that was current at tc_map_create time.
 *   - kernels are enqueued on the hipStream_t passed as `stream` (void*, NULL = default stream) and complete in
 *     stream order; tc_reset / tc_step / tc_step_multi / tc_render* / tc_noise neither allocate nor synchronise
 *     (they can be captured into a HIP graph).  The calls that DO wait for the device: tc_map_create,
 *     tc_env_create, tc_env_reserve_steps (when it has to re-allocate), tc_env_set_terms, tc_env_set_noise,
 *     tc_env_set_spawn_table, tc_env_profile_read and the destroy calls.
 *   - one host thread per env handle at a time, and ONE STREAM AT A TIME per handle: the library-owned scratch of a
 *     handle (draw lists, pose rows) is reused by consecutive calls, which is safe because they follow each other in
 *     stream order.  A K-step call issued on a different stream than the handle's previous K-step call is ordered
 *     behind it by the library (an event wait); for every other combination of streams the caller orders them.
 *   - every function returns 0 on success or a negative TC_E_* code; nothing throws.
 *   - arithmetic is IEEE double like the reference (python floats / numpy float64); pixel
 *     coordinates are int32, observations uint8.
 */
#ifndef TINYCARLO_HIP_H
#define TINYCARLO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TC_ABI_VERSION 5
#define TC_MAX_LAYERS 16

/* error codes */
#define TC_OK 0
#define TC_E_INVALID (-1)  /* bad argument / inconsistent sizes */
#define TC_E_HIP (-2)      /* a HIP runtime call failed (tc_last_error() has the text) */
#define TC_E_NOMEM (-3)
#define TC_E_UNBOUND (-4)  /* tc_step/tc_reset before tc_env_bind */
#define TC_E_LDS (-5)      /* map/camera too large for the 160 KiB LDS budget of one workgroup */

/* observation formats (sim.observation_space_format, env.py:42,67-72) */
#define TC_FMT_RGB 0     /* uint8 [N][H][W][3] */
#define TC_FMT_CLASSES 1 /* uint8 [N][C][H][W], values 0/255 */

/* action dtypes for tc_step */
#define TC_F32 0
#define TC_F64 1

/* step / reset flags */
#define TC_F_NO_OBSERVATION 1u /* env.no_observation (env.py:60,78-81): obs left untouched, no camera work */
#define TC_F_WRAPPED 2u        /* env.wrapped (env.py:56,137-138): reward = 0, terminated = false */
#define TC_F_AUTORESET 4u      /* envs whose needs_reset flag is set are re-spawned from spawn_queue at the
                                  start of the step (their action is ignored, reward 0, info empty) and the
                                  flag is set again from terminated|truncated at the end of the step */

#define TC_F_DEVICE_SPAWN 8u    /* with TC_F_AUTORESET: re-spawn nodes are drawn on the device from the table of
                                  tc_env_set_spawn_table (counter-based, tinycarlo_amd/csrc/tc_rng.h) instead of
                                  being read from spawn_queue; spawn_cursor[i] counts env i's re-spawns */

/* per-env status bits (tc_buffers.status), situations where the reference raises a Python exception */
#define TC_S_UTURN_NO_EDGE 1 /* U-turn found no lanepath edge within +-30 deg (TypeError at car.py:143): truncated */
#define TC_S_PICK_EMPTY 2    /* neighbour list made only of self-loops (ValueError at layer.py:123): truncated */
#define TC_S_BAD_SPAWN 4     /* spawn node out of range or without out-edge: first spawnable node used instead */
#define TC_S_NOT_RESET 8     /* tc_step on an env that was never reset: truncated, state untouched */
#define TC_S_SPAWN_WRAPPED 16 /* TC_F_AUTORESET without TC_F_DEVICE_SPAWN: this re-spawn read spawn_queue past its end
                                 (cursor >= spawn_queue_len, the queue is consumed cyclically): from here on the env
                                 replays spawn nodes it already used -- refill the queue and clear spawn_cursor */

/* Reward / termination terms: the wrappers of tinycarlo/wrapper/reward.py and termination.py, evaluated in the
 * epilogue of the step kernel in the order given (= the order the wrappers are stacked, innermost first; reward
 * additions are floating point, so the order is part of the result). */
#define TC_MAX_TERMS 8
#define TC_T_LANELINE_SPARSE_REWARD 1       /* reward.py:5-23   per_layer = sparse_rewards[name], layer_mask = names present */
#define TC_T_LANELINE_LINEAR_REWARD 2       /* reward.py:25-42  per_layer = max_rewards[name] (every layer) */
#define TC_T_CTE_SPARSE_REWARD 3            /* reward.py:44-62  p = {min_cte, sparse_reward} */
#define TC_T_CTE_LINEAR_REWARD 4            /* reward.py:64-84  p = {min_cte, max_reward, min_reward} */
#define TC_T_LANELINE_CROSSING_TERMINATION 5 /* termination.py:4-22   layer_mask = lanelines */
#define TC_T_CTE_TERMINATION 6              /* termination.py:24-48  p = {max_cte}, number_of_steps */
#define TC_T_CRASH_TERMINATION 7            /* termination.py:50-70  p = {velocity threshold}, number_of_steps */

typedef struct {
  int32_t kind;            /* TC_T_* */
  int32_t number_of_steps; /* consecutive-step terms */
  uint32_t layer_mask;     /* bit l = lane-line layer l takes part */
  int32_t reserved;
  double p[4];
  double per_layer[TC_MAX_LAYERS];
} tc_term;

typedef struct tc_map tc_map;
typedef struct tc_env tc_env;

/* Host-side description of a map, coordinates already in metres (map.py:28-37).
 * Lane-line layers are concatenated in JSON key order (map.py:23); edge node ids are layer-local. */
typedef struct {
  int32_t n_layers;
  const int32_t* node_count; /* [n_layers] */
  const int32_t* edge_count; /* [n_layers] */
  const double* nodes;       /* [sum node_count][2] */
  const int32_t* edges;      /* [sum edge_count][2] */
  const uint8_t* colors;     /* [n_layers][3], written to channels 0,1,2 as given (renderer.py:43) */
  int32_t lanepath_node_count;
  int32_t lanepath_edge_count;
  const double* lanepath_nodes;  /* [lanepath_node_count][2] */
  const int32_t* lanepath_edges; /* [lanepath_edge_count][2] */
} tc_map_desc;

/* car.py:10-19 */
typedef struct {
  double T; /* 1/fps, env.py:40-41 */
  double wheelbase, track_width, max_velocity, max_steering_angle;
  double steering_speed;                  /* used when has_steering_speed */
  double max_acceleration, max_deceleration; /* used when has_max_acceleration */
  int32_t has_steering_speed, has_max_acceleration;
} tc_car_params;

/* camera.py:12-24; E and K are computed on the host exactly as camera.py:145-178 does */
typedef struct {
  int32_t height, width;
  double E[12]; /* 3x4 row major */
  double K[9];  /* 3x3 row major */
  double max_range;
  int32_t line_thickness;
  int32_t format; /* TC_FMT_* */
} tc_camera_params;

/* Device buffers, all [N] unless noted.  State is in/out, the rest is written by tc_step/tc_reset. */
typedef struct {
  /* --- car state (car.py:25-32) */
  double *x, *y, *theta;  /* rear-axle position (m), heading (rad) */
  double *velocity;       /* m/s */
  double *steering;       /* deg */
  double *radius;         /* m, 0 when driving straight */
  double *front_x, *front_y;
  int32_t* local_path;    /* [N][8]: up to 4 lanepath edges (n0,n1), -1 padded */
  int32_t* lp_len;        /* 1 after reset, 1..4 after a step */
  int32_t* last_maneuver;
  /* --- per-step outputs (env.py:83-85,136-147) */
  double *cte, *heading_error, *reward;
  uint8_t *terminated, *truncated;
  int32_t* status;             /* TC_S_* bits */
  double* laneline_distances;  /* [N][n_layers] */
  int32_t* nearest_edge;       /* [N][n_layers] layer-local edge index of Layer.get_nearest_edge(rear), -1 if info empty */
  uint8_t* obs;                /* [N][C][H][W] or [N][H][W][3]; may be NULL if every call passes TC_F_NO_OBSERVATION */
  /* --- auto-reset (TC_F_AUTORESET) */
  uint8_t* needs_reset;        /* [N] */
  const int32_t* spawn_queue;  /* [N][spawn_queue_len] spawnable lanepath node ids drawn by the host RNG */
  int32_t* spawn_cursor;       /* [N] */
  int32_t spawn_queue_len;
} tc_buffers;

int tc_abi_version(void);
const char* tc_last_error(void);

int tc_map_create(const tc_map_desc* desc, tc_map** out);
int tc_map_destroy(tc_map* map);

int tc_env_create(const tc_map* map, const tc_car_params* car, const tc_camera_params* cam, int32_t num_envs,
                  tc_env** out);
int tc_env_destroy(tc_env* env);
int tc_env_bind(tc_env* env, const tc_buffers* buffers);
/* New E/K (and range / thickness) for all envs; takes effect for launches enqueued afterwards. */
int tc_env_set_camera(tc_env* env, const tc_camera_params* cam);
/* Per-env cameras (domain randomisation: examples/train_stanley_il.py:53-57 changes camera.orientation / fov and
 * calls update_params() once per episode; batched, that is one E and K per env).  E: device double [N][12],
 * K: device double [N][9], caller owned and read by every launch until replaced; (NULL, NULL) returns to the shared
 * camera of tc_env_create / tc_env_set_camera.  Resolution, max_range, thickness and format stay shared. */
int tc_env_set_camera_per_env(tc_env* env, const double* E, const double* K);
/* Installs n_terms (0..TC_MAX_TERMS) reward / termination terms; they apply to every tc_step enqueued afterwards
 * (the call waits for earlier launches).  Each term starts from the reward / terminated value left by the one
 * before it, the first from the base values of env.py:136-138 (0 / false under TC_F_WRAPPED, which the reference
 * wrappers always set).  counters: device int32 [N][TC_MAX_TERMS], caller owned, zero-initialised: steps_true of
 * the consecutive-step terms (termination.py:37,59), one per env and term slot; like the reference's attribute it
 * is NOT cleared when an env is reset.  May be NULL when no such term is installed.  Envs re-spawned by
 * TC_F_AUTORESET in a step skip the terms for that step (the reference's reset() does not pass through
 * Wrapper.step): reward 0, terminated 0, counters untouched.  n_terms = 0 removes all terms. */
int tc_env_set_terms(tc_env* env, const tc_term* terms, int32_t n_terms, int32_t* counters);
/* Device-side spawn sampling (TC_F_DEVICE_SPAWN).  nodes: HOST array of n > 0 lanepath node ids, copied by the call:
 * the candidates of Map.sample_spawn (map.py:61: spawn_points, or 0..len(nodes)-2) that have an out-edge, duplicates
 * kept -- a uniform draw from it is distributed like the reference's draw-again-on-sinks loop (map.py:62-64).
 * Re-spawn number k of env i uses SplitMix64 output (i << 32 | k) of the stream `seed`.  Not seed-compatible with
 * the reference's numpy generator; the host-drawn spawn_queue remains the seed-parity mode.  n = 0 removes the table. */
int tc_env_set_spawn_table(tc_env* env, const int32_t* nodes, int32_t n, uint64_t seed);
/* NoiseObservationWrapper (wrapper/observation.py:5-33) for class-mask observations: per class plane n_blobs blobs,
 * each a filled circle (centre inside the frame, radius in [1, max_radius)) that either erases the plane inside the
 * circle or ORs in the circle-masked content of a random plane, applied in order.  With n_blobs > 0 every tc_step /
 * tc_step_multi that renders an observation applies the blobs inside the raster stage, on the frame's bit-planes in
 * LDS before they are expanded to bytes (no extra pass over the observation in HBM), blobs drawn on the device
 * (tinycarlo_amd/csrc/tc_rng.h; the reference draws from the global numpy generator, which cannot be reproduced
 * for a batch).  Every rendered step consumes one position of the blob stream; the position is a counter in device
 * memory advanced on the stream behind the launch, so a call captured into a HIP graph draws new blobs on every
 * replay.  tc_reset / tc_render frames carry no noise (the reference's reset() does not pass through the wrapper's
 * step()).  n_blobs = 0 switches the noise off.  Needs TC_FMT_CLASSES and max_radius in [2, 256]. */
int tc_env_set_noise(tc_env* env, int32_t n_blobs, int32_t max_radius, uint64_t seed);
/* The noise pass alone, on the currently bound observation.  blobs: device int32 [N][n_layers * n_blobs][5] rows
 * (x, y, radius, mode, src) -- blob k belongs to plane k / n_blobs, mode 1 = copy from plane src, 0 = erase -- or
 * NULL to draw them on the device as tc_step does. */
int tc_noise(tc_env* env, const int32_t* blobs, void* stream);
/* bytes of one env's observation */
int64_t tc_env_obs_bytes(const tc_env* env);
/* dynamic LDS bytes one workgroup of the step kernel uses (for occupancy reporting) */
int64_t tc_env_lds_bytes(const tc_env* env);

/* Per-kernel timing for benchmarks: with enable = n > 0 every n-th tc_step records HIP events on the caller's
 * stream around its two kernels ("simulate": kinematics + tracking + distances + camera geometry; "raster":
 * cv2.polylines + observation store) into a ring of the last 64 launches.  tc_env_profile_read waits for
 * those launches and returns their mean durations in microseconds.  When a step is issued as ONE fused kernel
 * (the default unless a lane-line layer has more than 576 nodes / edges; env var TC_FUSE=0 forces two launches), simulate_us is
 * the duration of that kernel and raster_us is ~0. */
int tc_env_profile(tc_env* env, int32_t enable);
int tc_env_profile_read(tc_env* env, double* simulate_us, double* raster_us, int32_t* launches);

/* env.py:101-113 for the envs selected by mask (NULL = all): pose from spawn_nodes[i], zeroed info,
 * observation rendered unless TC_F_NO_OBSERVATION. */
int tc_reset(tc_env* env, const int32_t* spawn_nodes, const uint8_t* mask, uint32_t flags, void* stream);

/* env.py:115-147 for all envs.  car_control: [N][2] (velocity, steering in [-1,1], clipped like env.py:118),
 * dtype TC_F32 or TC_F64; maneuver: [N] in {0,1,2,3}.
 * One launch (tc_step_kernel: a wavefront simulates its env, runs the camera and rasterises the frame).  Which env a
 * workgroup works on is the library's choice -- envs are independent, any assignment gives the same results: when N is
 * 2-4 rows of one workgroup per SIMD of the device, the envs are re-dealt every TC_STEP_ORDER-th call (default 8; a
 * one-workgroup tc_order_kernel launch in front of the step) so that frames with long and short draw lists share a
 * SIMD (workgroups w, w + #SIMDs, ... land on the same SIMD). */
int tc_step(tc_env* env, const void* car_control, int32_t control_dtype, const int32_t* maneuver, uint32_t flags,
            void* stream);

/* K steps in ONE call: the caller's `for k in range(K): env.step(action[k])` loop (env.py:115-147 called K times,
 * e.g. examples/stanley_control.py:50-60 with the actions known in advance: action repeat, open-loop rollouts,
 * scripted policies).  Envs are independent, so no grid-wide synchronisation exists between steps.
 *   car_control: [K][N][2], maneuver: [K][N] (step k uses row k).
 *   Results are bit-identical to K calls of tc_step with the same flags: the bound buffers (tc_env_bind) hold the
 *   state and the outputs of step K-1 afterwards (status included: the TC_S_* bits of step K-1 only -- the per-step
 *   bits are in rollout->status); TC_F_AUTORESET re-spawns and fused terms (tc_env_set_terms) act per step exactly
 *   as there.
 *   rollout (may be NULL): per-step copies of everything env.step() returns (env.py:83-85,131-147), each member
 *   [K][N] (shapes below) or NULL.  With rollout->obs the observation of step k goes to rollout->obs[k] and the
 *   bound obs buffer is left untouched; without it every step stores its observation into the bound buffer (which
 *   ends up holding the last).
 * What is launched (tc_env_launch_info reports it):
 *   - no observation wanted: ONE launch of tc_env_kernel, one wavefront per env looping over the K steps with the
 *     env's state parked in LDS.
 *   - observations into rollout->obs (every step's frame wanted), STREAMED -- the default: per call (per segment of
 *     128 steps of a longer call) ONE simulate launch on the caller's stream -- tc_envg_kernel: 8 lanes per env, 8 envs
 *     per wavefront, state in registers, looping over the steps and leaving one 128-byte pose row (camera.py:62) per
 *     (step, env) -- and, on an internal stream, ONE frame launch that starts at once and runs BESIDE it --
 *     tc_frame_kernel: one workgroup per (step, env), camera stage + raster stage + the observation store; a workgroup
 *     first polls its pose row with device-scope loads until the simulate launch has written it (each of the twelve
 *     entries validates itself: a row is "not written yet" while any entry holds an all-ones NaN, and the frame
 *     workgroup puts that pattern back when it is done).  Only the call's first step is exposed, and the chip drains
 *     once per call instead of once per chunk.  Around the two launches, on the internal stream: tc_order_kernel
 *     (heaviest frames first, see TC_FRAME_ORDER), tc_gate_kernel, one wavefront that holds the frame launch back until
 *     every simulate workgroup has started, so the frame workgroups can fill the chip without keeping their producers
 *     off it, and, behind the simulate launch, tc_frame_recover_kernel.  EVERY wait on the device is bounded
 *     (TC_STREAM_WAIT_US, default 5000): a frame workgroup whose wait runs out marks its frame skipped, tells the others
 *     to stop waiting and leaves, and the recover kernel -- one workgroup per env, normally one look at the call's
 *     rows and out -- draws the skipped frames from the then complete rows.  The result therefore never depends on
 *     how the two launches were scheduled, only the time does.  The internal stream is joined back into the caller's
 *     before the call's work there ends, so everything completes in stream order.
 *   - the same, CHUNKED (TC_STREAM=0, and the form of calls without rollout->obs, where only the last step's frame
 *     is drawn): the steps are issued in chunks of at most 16 steps (a quarter of the call if that is less).  Per
 *     chunk a simulate launch (tc_envg_kernel; the first chunk of a multi-chunk call goes through tc_env_kernel: short
 *     rather than cheap, nothing overlaps it) and a frame launch (tc_frame_kernel, pose rows complete: no polling)
 *     on an internal stream behind it, so chunk c+1 is simulated while chunk c is drawn.  Calls whose chunks are
 *     shorter than 8 steps alternate their frame launches between two internal streams.
 *   - maps whose largest lane-line layer exceeds 576 nodes / edges (K = 13 variant) and TC_FUSE=0: the camera stage
 *     runs inside the simulate launch and tc_raster_kernel draws the frames.
 * The pose rows and draw lists in flight live in library-owned scratch sized by tc_env_reserve_steps, which must have
 * been called once before the first K-step call that renders (n_steps > 1 with an observation): tc_step_multi itself
 * never allocates and never waits for the device; without the scratch it returns TC_E_INVALID.
 * Env vars (all result-neutral, read at tc_env_create): TC_STREAM=0 chunked form; TC_STREAM_WAIT_US=t bound of a frame
 * workgroup's wait; TC_STREAM_SCRATCH_MB=m budget of the streamed form's scratch (default 16384: the rows of a segment
 * are cut to fit, tc_env_reserve_steps); TC_STREAM_TEST_SKIP=m (tests) frames with (step + env) % m == 0 are left to the recover kernel;
 * TC_CHUNK=n steps per chunk (0: chunks follow each other on the caller's stream, no overlap, no streaming),
 * TC_ENV_GROUPED=0 one wavefront per env in every simulate launch (no overlap, no streaming),
 * TC_FIRST_CHUNK_PER_ENV=0, TC_FRAME_STREAMS=1 (chunked form), TC_ENVG_MAP_LDS=0, TC_MULTI_SPLIT=0 a single fused launch
 * in which the same wavefront simulates its env and rasterises each of its frames, TC_SEG_LDS=0 / TC_SEG_LDS_CAP=n draw
 * lists through global memory only / beyond the first n segments, TC_FRAME_ORDER=0 frame workgroups in env order (default:
 * the envs whose last frame had the longest draw list first, so that a dispatch ends with its cheapest frames),
 * TC_STEP_ORDER=n (tc_step: the envs are re-dealt to the workgroups every n-th step so that heavy and light frames share
 * a SIMD, default 8; 0 = workgroup w works on env w). */
typedef struct {
  uint8_t* obs;          /* [K][N][tc_env_obs_bytes] */
  double* reward;        /* [K][N] */
  uint8_t* terminated;   /* [K][N] */
  uint8_t* truncated;    /* [K][N] */
  double* cte;           /* [K][N] */
  double* heading_error; /* [K][N] */
  /* --- ABI 5: the rest of env.step()'s info dict (env.py:83-85) and the status bits, per step */
  int32_t* status;             /* [K][N] TC_S_* bits raised in step k */
  double *x, *y, *theta;       /* [K][N] info["position"], info["orientation"] (state after step k) */
  double* velocity;            /* [K][N] car velocity after step k (info["velocity"] is this, or 0 while lp_len < 2) */
  double* laneline_distances;  /* [K][N][n_layers] */
  int32_t* nearest_edge;       /* [K][N][n_layers] */
  int32_t* local_path;         /* [K][N][8] */
  int32_t* lp_len;             /* [K][N] */
} tc_rollout;
int tc_step_multi(tc_env* env, const void* car_control, int32_t control_dtype, const int32_t* maneuver, int32_t n_steps,
                  uint32_t flags, const tc_rollout* rollout, void* stream);

/* Sizes the scratch of K-step calls that render observations, for calls of up to max_call_steps steps: per (step, env)
 * a pose row (128 B), a draw-list length and room for a draw list (20 B x lane-line edges of the map; only a frame's
 * overflow beyond the LDS-resident head travels through it).  Streamed form: min(max_call_steps, 128) steps x N envs, fewer when that would exceed TC_STREAM_SCRATCH_MB (16 GiB)
 * (cfg3: 22 MB per step); chunked form: a ring of TC_RING_SLOTS (3) chunks of min(max_call_steps, 16 or TC_CHUNK) steps.
 * A call of ANY n_steps then runs in segments / chunks that fit.  Re-allocating waits for the device first (an earlier
 * launch may still use the old arrays); a request the current scratch already covers returns at once.
 * max_call_steps < 1 is TC_E_INVALID. */
int tc_env_reserve_steps(tc_env* env, int32_t max_call_steps);

/* What the library launches for a call of n_steps steps (1 = tc_step) with the current settings -- for benchmark
 * labels, not for control flow: fused = 1 when simulate + raster run as one kernel; kvar = register-cache variant of
 * the simulate stage (5, 8, 9, 13); steps_per_dispatch = steps one kernel dispatch of the call covers when every
 * step's frame goes to a rollout (the whole call when it is streamed, else a chunk: see tc_step_multi); name receives the
 * kernel symbols ("tc_step_kernel", "tc_envg_kernel+tc_frame_kernel", "tc_env_kernel+tc_raster_kernel", "tc_env_kernel"),
 * at most name_cap bytes including the terminator. */
int tc_env_launch_info(const tc_env* env, uint32_t flags, int32_t n_steps, int32_t* fused, int32_t* kvar,
                       int32_t* steps_per_dispatch, char* name, int32_t name_cap);

/* Workload descriptor for benchmark lines -- what the most recent frames drew: mean / max length of the frames' draw
 * lists (segments handed to cv2.polylines, camera.py:95-106) and the fraction of frames with none, over the N frames of
 * the last tc_step or, of the last tc_step_multi call, the frames of its last (up to 48) steps.  Waits for the device. */
int tc_env_draw_list_stats(tc_env* env, double* mean_segments, double* empty_frac, int32_t* max_segments, int64_t* frames);

/* Renderer.render_camera_frame_{rgb,classes} alone (renderer.py:36-51): rasterise caller-provided segment lists
 * into the bound observation tensor.  segments: device int32 [N][capacity][5] rows of (layer, x0, y0, x1, y1) --
 * the np.int32 end points handed to cv2.polylines; counts: device int32 [N], each <= capacity. */
int tc_render_segments(tc_env* env, const int32_t* segments, const int32_t* counts, int32_t capacity, void* stream);

/* Re-render the observation of the current state without stepping (Camera.capture_frame, camera.py:52). */
int tc_render(tc_env* env, uint32_t flags, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TINYCARLO_HIP_H */
