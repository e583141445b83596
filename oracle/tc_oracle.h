/* tc_oracle.h -- CPU ORACLE for the tinycarlo step() hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A scalar, double-precision, plain-C restatement of the reference algorithm
 * (emrullahArkun/tinycarlo: tinycarlo/car.py, layer.py, map.py, camera.py, renderer.py, env.py,
 * helper.py; every function below cites the lines it follows).  Nothing in the shipped package
 * (tinycarlo_amd/) may import, link or call this; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do, and only as the checker / the timed CPU baseline.
 *
 * Pinning: kinematics, lanepath tracking, CTE/heading, lane-line distances, nearest-edge ids and
 * the int32 segment lists handed to cv2.polylines are pinned against golden vectors generated
 * from the imported reference (tests/golden/, tests/test_oracle_golden.py) and against the
 * reference's own unit tests restated as vectors (tests/golden/unit_vectors.json).
 * The PIXELS painted by cv2.polylines are third-party OpenCV arithmetic that is not available in
 * the build container: the rasteriser here restates OpenCV 4.x drawing.cpp (opencv-python>=4.5.5.62,
 * reference setup.py:19; call sites renderer.py:43,50) from its published algorithm and is
 * "raster parity unpinned" (DESIGN.md).
 *
 * Two math modes (orc_set_math_mode):
 *   ORC_MATH_LIBM     (default) host libm sin/cos/tan/atan2/pow -- what CPython's math.* calls;
 *   ORC_MATH_PORTABLE tinycarlo_amd/csrc/tc_trig.h and x*x -- the exact operation sequence the
 *                     HIP kernels execute, so GPU results can be compared bit-for-bit.
 */
#ifndef TC_ORACLE_H
#define TC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAXC 16
#define ORC_MATH_LIBM 0
#define ORC_MATH_PORTABLE 1

#define ORC_FMT_RGB 0
#define ORC_FMT_CLASSES 1

/* step flags (same bit values as include/tinycarlo_hip.h) */
#define ORC_F_NO_OBSERVATION 1u /* env.py:60,78-81 */
#define ORC_F_WRAPPED 2u        /* env.py:56,137-138: default reward/termination disabled */
#define ORC_F_AUTORESET 4u      /* batched extension: reset finished envs at the start of the next step */
#define ORC_F_DEVICE_SPAWN 8u   /* with AUTORESET: spawn node from the counter-based table draw (csrc/tc_rng.h) */

/* status bits */
#define ORC_S_UTURN_NO_EDGE 1  /* reference raises TypeError at car.py:143 */
#define ORC_S_PICK_EMPTY 2     /* reference raises ValueError at layer.py:123 (all-self-loop neighbour list) */

typedef struct orc_map orc_map;

typedef struct {
  double T, wheelbase, track_width, max_velocity, max_steering_angle;
  double steering_speed, max_acceleration, max_deceleration;
  int32_t has_steering_speed, has_max_acceleration;
} orc_car;

typedef struct {
  int32_t H, W;
  double E[12]; /* 3x4 row major, camera.py:145-156 */
  double K[9];  /* 3x3 row major, camera.py:158-178 */
  double max_range;
  int32_t line_thickness;
  int32_t format; /* ORC_FMT_* */
} orc_cam;

typedef struct {
  double x, y, theta, velocity, steering, radius, front_x, front_y;
  int32_t lp[8]; /* up to 4 lanepath edges (n0,n1) */
  int32_t lp_len;
  int32_t last_maneuver;
} orc_state;

typedef struct {
  double cte, heading_error, reward, velocity;
  double dist[ORC_MAXC];
  double lp_coords[8];
  int32_t nearest_edge[ORC_MAXC];
  int32_t n_lp_coords;
  int32_t terminated, truncated, status;
} orc_info;

/* tinycarlo/wrapper/reward.py + termination.py as an ordered list of terms (same layout and kind values as
 * tc_term in include/tinycarlo_hip.h) */
#define ORC_MAX_TERMS 8
#define ORC_T_LANELINE_SPARSE_REWARD 1
#define ORC_T_LANELINE_LINEAR_REWARD 2
#define ORC_T_CTE_SPARSE_REWARD 3
#define ORC_T_CTE_LINEAR_REWARD 4
#define ORC_T_LANELINE_CROSSING_TERMINATION 5
#define ORC_T_CTE_TERMINATION 6
#define ORC_T_CRASH_TERMINATION 7
typedef struct {
  int32_t kind, number_of_steps;
  uint32_t layer_mask;
  int32_t reserved;
  double p[4];
  double per_layer[ORC_MAXC];
} orc_term;

void orc_set_math_mode(int mode);
int orc_get_math_mode(void);

/* scalar helpers exposed for the unit-vector tests and tests/test_trig.py */
double orc_clip_angle(double a);
double orc_trig(int fn, double a, double b, int mode); /* fn: 0 sin 1 cos 2 tan 3 atan2(a,b) */

/* Layer queries on a bare node/edge list (layer.py) */
int orc_layer_nearest_edge(const double* nodes, const int32_t* edges, int n_edges, double px, double py);
int orc_layer_nearest_node(const double* nodes, int n_nodes, double px, double py);
int orc_layer_nearest_edge_with_orientation(const double* nodes, const int32_t* edges, int n_edges, double px,
                                            double py, double orientation, double margin_deg);
int orc_layer_within_bounds(const double* nodes, const int32_t* edge, double px, double py);
double orc_layer_distance_to_edge(const double* nodes, const int32_t* edge, double px, double py);

/* Map (map.py:9-37): nodes already divided by pixel_per_meter by the caller */
orc_map* orc_map_create(int32_t n_layers, const int32_t* node_count, const int32_t* edge_count, const double* nodes,
                        const int32_t* edges, const uint8_t* colors, int32_t lp_nodes_n, int32_t lp_edges_n,
                        const double* lp_nodes, const int32_t* lp_edges);
void orc_map_free(orc_map*);
int orc_map_has_next(const orc_map*, int node); /* map.py:62-64: spawn nodes without out-edge are re-drawn */

/* car.py:34-44 + map.py:51-69 with the node index already drawn */
void orc_reset(const orc_map*, const orc_car*, orc_state*, int spawn_node);
/* car.py:70-125 (+127-148); returns truncated; *status gets ORC_S_* bits */
int orc_car_step(const orc_map*, const orc_car*, orc_state*, double v_in, double s_in, int maneuver, int* status);
/* car.py:46-68 + env.py:93,99 */
void orc_get_info(const orc_map*, const orc_car*, const orc_state*, uint32_t flags, orc_info*);
/* camera.py:52-110 up to the np.int32 cast in renderer.py:43,50.
 * seg_i: rows of (layer,x0,y0,x1,y1); seg_f: rows of (u0,v0,u1,v1) doubles (may be NULL). returns count (<= max) */
int orc_capture_segments(const orc_map*, const orc_cam*, const orc_state*, int32_t* seg_i, double* seg_f, int max);
/* renderer.py:36-51: zero frame + cv2.polylines per segment. frame: [C][H][W] (classes) or [H][W][3] (rgb) */
void orc_render(const orc_map*, const orc_cam*, const int32_t* seg_i, int nseg, uint8_t* frame);
/* a single cv2.polylines(img, int32[[p0,p1]], False, color, thickness) on a 1- or 3-channel image */
void orc_polyline2(uint8_t* img, int W, int H, int channels, int x0, int y0, int x1, int y1, const uint8_t* color,
                   int thickness);

/* env.py:115-147 for a batch of independent envs.  car_control: [N][2] doubles (f32-valued is fine),
 * obs: [N][obs_bytes] or NULL.  needs_reset/spawn_queue/spawn_cursor only used with ORC_F_AUTORESET.
 * n_threads > 1 uses OpenMP over envs.  */
void orc_step_batch(const orc_map*, const orc_car*, const orc_cam*, int N, orc_state* st, const double* car_control,
                    const int32_t* maneuver, uint32_t flags, orc_info* info, uint8_t* obs, uint8_t* needs_reset,
                    const int32_t* spawn_queue, int spawn_queue_len, int32_t* spawn_cursor, int n_threads);
/* env.py:101-113 for a batch: reset + observation + empty info */
void orc_reset_batch(const orc_map*, const orc_car*, const orc_cam*, int N, orc_state* st, const int32_t* spawn_node,
                     const uint8_t* mask, uint32_t flags, orc_info* info, uint8_t* obs, int n_threads);
/* wrapper/utils.py:21-37 */
double orc_linear_reward(double x, double max_x, double max_reward, double min_reward);
/* One Wrapper.step() pass of the stacked wrappers over the info of one env (innermost term first): updates
 * info->reward / info->terminated and the steps_true counters [ORC_MAX_TERMS] of this env. */
void orc_apply_terms(const orc_term* terms, int n_terms, int n_layers, double track_width, orc_info* info,
                     int32_t* counters);
/* orc_step_batch followed by orc_apply_terms on every env that was stepped (re-spawned envs skip the terms, as
 * the reference's reset() bypasses Wrapper.step); counters: [N][ORC_MAX_TERMS] */
void orc_step_batch_terms(const orc_map*, const orc_car*, const orc_cam*, int N, orc_state* st,
                          const double* car_control, const int32_t* maneuver, uint32_t flags, orc_info* info,
                          uint8_t* obs, uint8_t* needs_reset, const int32_t* spawn_queue, int spawn_queue_len,
                          int32_t* spawn_cursor, int n_threads, const orc_term* terms, int n_terms,
                          int32_t* counters);
/* everything optional a batched step can carry; NULL / zero fields switch the feature off */
typedef struct {
  const orc_term* terms;
  int32_t n_terms;
  int32_t* counters;          /* [N][ORC_MAX_TERMS] */
  const int32_t* spawn_table; /* ORC_F_DEVICE_SPAWN: spawnable candidate nodes */
  int32_t spawn_n;
  uint64_t spawn_seed;
} orc_step_ext;
void orc_step_batch_ext(const orc_map*, const orc_car*, const orc_cam*, int N, orc_state* st, const double* car_control,
                        const int32_t* maneuver, uint32_t flags, orc_info* info, uint8_t* obs, uint8_t* needs_reset,
                        const int32_t* spawn_queue, int spawn_queue_len, int32_t* spawn_cursor, int n_threads,
                        const orc_step_ext* ext);
/* wrapper/observation.py:15-27 (add_blob_noise_classes) on one class-mask frame [C][H][W] with the random draws given:
 * blobs [C * n_blobs][5] rows (x, y, radius, mode, src); blob k belongs to plane k / n_blobs; mode 1 = the "True"
 * branch (copy the circle-masked content of plane src in), 0 = erase the circle.  cv2.circle is this file's
 * restatement of Circle(..., fill) -- pixels unpinned like the rest of the raster. */
void orc_noise_classes(uint8_t* frame, int C, int H, int W, const int32_t* blobs, int n_blobs);
/* the blobs the device draws for (env, step): tinycarlo_amd/csrc/tc_rng.h (no reference counterpart) */
void orc_noise_blobs(uint64_t seed, uint32_t env, uint32_t step, int n_blobs, int C, int H, int W, int max_radius,
                     int32_t* out);
/* tinycarlo_amd/csrc/tc_rng.h, exported for the known-answer tests */
uint64_t orc_splitmix64_at(uint64_t seed, uint64_t n);
uint32_t orc_spawn_index(uint64_t seed, uint32_t env, uint32_t cursor, uint32_t count);
int64_t orc_obs_bytes(const orc_map*, const orc_cam*);

#ifdef __cplusplus
}
#endif
#endif
