// tinycarlo_hip.hip -- libtinycarlo_hip.so: kernels + C ABI (include/tinycarlo_hip.h).
//
// Build (see build.py): hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared
//
// Kernel layout: one 64-lane wavefront (= one workgroup) per env, N workgroups per launch.
//   phase A  kinematics + lanepath tracking + CTE/heading   (car.py:70-148, 46-53)   wave-uniform scalar math
//   phase B  nearest lane-line edge per layer + distance    (layer.py:33-44, car.py:55-64)
//            node distances staged in LDS once, edges strided over lanes, shuffle argmin
//   phase C  camera: transform -> 4 clip passes -> project -> visibility -> draw list   (camera.py:52-110)
//            rasterise cv2.polylines into LDS bit-planes, expand to uint8 and store with
//            16-byte-per-lane coalesced stores (zeros included: the frame is written exactly once)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/tinycarlo_hip.h"
#include "tc_device.h"

#define MODE_STEP 0
#define MODE_RESET 1
#define MODE_RENDER 2

// ablation switches for profiling (not part of the ABI contract; results are wrong when set)
#define DBG_SKIP_RASTER 0x100u
#define DBG_SKIP_STORE 0x200u
#define DBG_SKIP_CAMERA 0x400u
#define DBG_SKIP_DIST 0x800u

struct LdsLayout {
  int off_p, off_flg, off_list, off_seg, off_bits, off_cnt;
  int seg_cap;
  int total;
};

struct KArgs {
  DevMap m;
  DevCar car;
  DevCam cam;
  tc_buffers b;
  LdsLayout lds;
  int N;
};

// ---------------------------------------------------------------------------------------------
// camera.py:70-86: one of the four fix-up loops.  `bit` selects the membership flag (1 = idx_front,
// 2 = idx_in_range).  The reference builds the edge list first and then mutates nodes in list
// (= edge) order; updates of different target nodes are independent (a target is never read as
// the "other" end within one pass), so each target node's chain is replayed in ascending edge
// index by the lane that owns the chain's first edge.
__device__ inline void cam_fixup_pass(double* Px, double* Py, double* Pz, unsigned char* flg, int bit, const int2* LE,
                                      int ne, bool target_e0, double tz, int* list, int* cnt) {
  const int tid = threadIdx.x;
  if (tid == 0) *cnt = 0;
  __syncthreads();
  for (int e = tid; e < ne; e += TC_NT) {
    int2 ed = LE[e];
    bool fa = flg[ed.x] & bit, fb = flg[ed.y] & bit;
    bool sel = target_e0 ? (!fa && fb) : (fa && !fb);
    if (sel) list[atomicAdd(cnt, 1)] = e;
  }
  __syncthreads();
  const int n = *cnt;
  for (int k = tid; k < n; k += TC_NT) {
    const int e = list[k];
    const int t = target_e0 ? LE[e].x : LE[e].y;
    bool first = true;
    for (int j = 0; j < n; j++) {
      int ej = list[j];
      int tj = target_e0 ? LE[ej].x : LE[ej].y;
      if (tj == t && ej < e) {
        first = false;
        break;
      }
    }
    if (!first) continue;
    int cur = e;
    for (int guard = 0; guard < n; guard++) {
      const int o = target_e0 ? LE[cur].y : LE[cur].x;  // the end that stays
      // camera.py:112-122 __point_on_line_at_z(p0 = P[o], p1 = P[t], tz)
      double d0 = Px[o] - Px[t], d1 = Py[o] - Py[t], d2 = Pz[o] - Pz[t];
      if (d2 == 0) {
        double qn = __longlong_as_double(0x7ff8000000000000LL);
        Px[t] = qn;
        Py[t] = qn;
        Pz[t] = qn;
      } else {
        double tt = (tz - Pz[t]) / d2;
        double a = Px[t] + tt * d0, b = Py[t] + tt * d1, c = Pz[t] + tt * d2;
        Px[t] = a;
        Py[t] = b;
        Pz[t] = c;
      }
      int nxt = 0x7fffffff;
      for (int j = 0; j < n; j++) {
        int ej = list[j];
        int tj = target_e0 ? LE[ej].x : LE[ej].y;
        if (tj == t && ej > cur && ej < nxt) nxt = ej;
      }
      if (nxt == 0x7fffffff) break;
      cur = nxt;
    }
    flg[t] |= (unsigned char)bit;
  }
  __syncthreads();
}

// A spawn node must exist and have an out-edge (map.py:62-64 re-draws otherwise; the host RNG mirror
// does that).  Anything else would index out of bounds, so it is replaced and flagged.
__device__ inline int checked_spawn(const DevMap& m, int node, int& status) {
  if ((unsigned)node < (unsigned)m.lpN && m.next_off[node + 1] > m.next_off[node]) return node;
  status |= TC_S_BAD_SPAWN;
  return m.first_spawnable;
}

__device__ inline unsigned int spread4(unsigned int x) {  // 4 bits -> 4 bytes of 0x00/0xFF
  return ((x & 1u) | ((x & 2u) << 7) | ((x & 4u) << 14) | ((x & 8u) << 21)) * 255u;
}

__global__ __launch_bounds__(TC_NT) void tc_env_kernel(KArgs a, int mode, const void* car_control, int cdtype,
                                                       const int* maneuver, const int* spawn_nodes,
                                                       const unsigned char* mask, unsigned int flags) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int env = blockIdx.x;
  const int tid = threadIdx.x;
  if (env >= a.N) return;
  if (mode == MODE_RESET && mask && !mask[env]) return;  // whole workgroup leaves: no barrier below is reached

  const DevMap& m = a.m;
  const tc_buffers& b = a.b;
  double* Px = (double*)(smem + a.lds.off_p);
  double* Py = Px + m.max_nodes;
  double* Pz = Py + m.max_nodes;
  double* dn = Px;  // phase B alias
  unsigned char* flg = smem + a.lds.off_flg;
  int* list = (int*)(smem + a.lds.off_list);
  int* seg = (int*)(smem + a.lds.off_seg);
  unsigned int* bits = (unsigned int*)(smem + a.lds.off_bits);
  int* cnt = (int*)(smem + a.lds.off_cnt);

  // ---- state (wave-uniform loads)
  CarState s;
  s.x = b.x[env];
  s.y = b.y[env];
  s.theta = b.theta[env];
  s.velocity = b.velocity[env];
  s.steering = b.steering[env];
  s.radius = b.radius[env];
  s.front_x = b.front_x[env];
  s.front_y = b.front_y[env];
#pragma unroll
  for (int i = 0; i < 8; i++) s.lp[i] = b.local_path[env * 8 + i];
  s.lp_len = b.lp_len[env];
  s.last_maneuver = b.last_maneuver[env];

  int status = 0, trunc = 0;
  bool fresh = false;  // env was (re)spawned in this launch: info is empty (car.py:47-51)
  if (mode == MODE_RESET) {
    d_reset(m, a.car, s, checked_spawn(m, spawn_nodes[env], status));
    fresh = true;
  } else if (mode == MODE_STEP) {
    if ((flags & TC_F_AUTORESET) && b.needs_reset[env]) {
      int cur = b.spawn_cursor[env];
      int node = b.spawn_queue[(size_t)env * b.spawn_queue_len + ((unsigned)cur % (unsigned)b.spawn_queue_len)];
      d_reset(m, a.car, s, checked_spawn(m, node, status));
      fresh = true;
      if (tid == 0) b.spawn_cursor[env] = cur + 1;
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
      trunc = d_car_step(m, a.car, s, v, st, maneuver[env], status);
    }
  }

  if (mode != MODE_RENDER) {
    // ---- write the state back + scalar info (lane 0)
    const bool have_info = !fresh && s.lp_len >= 2;
    double cte = 0, he = 0;
    if (have_info) {  // car.py:52-53
      double2 n1 = m.lp_nodes[s.lp[2]], n2 = m.lp_nodes[s.lp[3]];
      cte = d_distance_to_edge(n1.x, n1.y, n2.x, n2.y, s.front_x, s.front_y);
      he = d_clip_angle(d_lp_edge_ori(m, s.lp[2], s.lp[3]) - s.theta);
    }
    double reward = 0;
    int terminated = 0;
    if (!(flags & TC_F_WRAPPED) && !fresh) {  // env.py:93,99
      double r = (-1 / a.car.track_width) * cte + 1;
      reward = (0 > r) ? 0 : r;
      terminated = cte > (a.car.track_width * 10);
    }
    if (tid == 0) {
      b.x[env] = s.x;
      b.y[env] = s.y;
      b.theta[env] = s.theta;
      b.velocity[env] = s.velocity;
      b.steering[env] = s.steering;
      b.radius[env] = s.radius;
      b.front_x[env] = s.front_x;
      b.front_y[env] = s.front_y;
      b.lp_len[env] = s.lp_len;
      b.last_maneuver[env] = s.last_maneuver;
      b.cte[env] = cte;
      b.heading_error[env] = he;
      b.reward[env] = reward;
      b.terminated[env] = (unsigned char)terminated;
      b.truncated[env] = (unsigned char)trunc;
      b.status[env] = status;
      if (b.needs_reset) b.needs_reset[env] = (flags & TC_F_AUTORESET) ? (unsigned char)(terminated || trunc) : 0;
    }
    if (tid < 8) {  // register-resident select (a runtime-indexed s.lp[tid] would live in scratch)
      int v = s.lp[0];
#pragma unroll
      for (int i = 1; i < 8; i++) v = tid == i ? s.lp[i] : v;
      b.local_path[env * 8 + tid] = v;
    }

    // ---- phase B: lane-line distances (car.py:55-64)
    const int C = m.C;
    if (have_info && !(flags & DBG_SKIP_DIST)) {
      for (int i = tid; i < m.total_nodes; i += TC_NT) {
        double2 n = m.nodes[i];
        dn[i] = d_dist(s.x, s.y, n.x, n.y);
      }
      __syncthreads();
      int my_e = -1;
      for (int l = 0; l < C; l++) {
        const int no = m.node_off[l], eo = m.edge_off[l], ne = m.edge_off[l + 1] - eo;
        int best = -1;
        double bd = 0;
        for (int e = tid; e < ne; e += TC_NT) {  // layer.py:43
          int2 ed = m.edges[eo + e];
          double d = tc_fabs(dn[no + ed.x] + dn[no + ed.y]);
          if (best < 0 || d < bd) {
            best = e;
            bd = d;
          }
        }
        wave_argmin(bd, best);
        if (tid == l) my_e = best;
      }
      if (tid < C) {
        const int l = tid;
        double dist_l = 0;
        if (my_e >= 0) {
          const int no = m.node_off[l], ge = m.edge_off[l] + my_e;
          int2 ed = m.edges[ge];
          double2 n0 = m.nodes[no + ed.x], n1 = m.nodes[no + ed.y];
          if (d_within_bounds(n0.x, n0.y, n1.x, n1.y, m.ori_fwd[ge], m.ori_rev[ge], s.x, s.y)) {
            dist_l = tc_fabs(d_distance_to_edge(n0.x, n0.y, n1.x, n1.y, s.x, s.y));
          } else {
            double da = d_dist(s.x, s.y, n0.x, n0.y);
            double db = d_dist(s.front_x, s.front_y, n1.x, n1.y);  // FRONT axle for n1 (car.py:64)
            dist_l = db < da ? db : da;
          }
        }
        b.laneline_distances[(size_t)env * C + l] = dist_l;
        b.nearest_edge[(size_t)env * C + l] = my_e;
      }
      __syncthreads();  // dn aliases the camera's node buffer
    } else if (tid < C) {
      b.laneline_distances[(size_t)env * C + tid] = 0;
      b.nearest_edge[(size_t)env * C + tid] = -1;
    }
  }

  // ---- phase C: camera (camera.py:52-110) + raster (renderer.py:36-51)
  if ((flags & TC_F_NO_OBSERVATION) || b.obs == nullptr) return;
  if (flags & DBG_SKIP_CAMERA) return;
  const DevCam& cam = a.cam;
  double pose[12];
  {
    double cth = tc_cos(-s.theta), sth = tc_sin(-s.theta);  // car.py:159-165
    double R[16] = {cth, -sth, 0, 0, sth, cth, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double Tm[16] = {1, 0, 0, -s.x, 0, 1, 0, -s.y, 0, 0, 1, 0, 0, 0, 0, 1};
    double car3d[16];
    d_matmul<4, 4, 4>(R, Tm, car3d);
    d_matmul<3, 4, 4>(cam.E, car3d, pose);  // camera.py:62
  }
  int* seg_cnt = cnt + 1;
  if (tid == 0) *seg_cnt = 0;
  for (int l = 0; l < m.C; l++) {
    const int no = m.node_off[l], nn = m.node_off[l + 1] - no;
    const int eo = m.edge_off[l], ne = m.edge_off[l + 1] - eo;
    const int2* LE = m.edges + eo;
    for (int i = tid; i < nn; i += TC_NT) {  // camera.py:124-131
      double2 n = m.nodes[no + i];
      double h[4] = {n.x, n.y, 0.0, 1.0};
      double p[3];
      d_matmul<3, 4, 1>(pose, h, p);
      Px[i] = p[0];
      Py[i] = p[1];
      Pz[i] = p[2];
      flg[i] = p[2] < 0 ? 1 : 0;  // camera.py:70
    }
    __syncthreads();
    cam_fixup_pass(Px, Py, Pz, flg, 1, LE, ne, true, -0.0000001, list, cnt);   // camera.py:71-74
    cam_fixup_pass(Px, Py, Pz, flg, 1, LE, ne, false, -0.0000001, list, cnt);  // camera.py:75-77
    for (int i = tid; i < nn; i += TC_NT)
      if (Pz[i] > -cam.max_range) flg[i] |= 2;  // camera.py:80, on the mutated depths
    __syncthreads();
    cam_fixup_pass(Px, Py, Pz, flg, 2, LE, ne, true, -cam.max_range, list, cnt);   // camera.py:81-83
    cam_fixup_pass(Px, Py, Pz, flg, 2, LE, ne, false, -cam.max_range, list, cnt);  // camera.py:84-86
    for (int i = tid; i < nn; i += TC_NT) {  // camera.py:133-142, 90-93
      double P3[3] = {Px[i], Py[i], Pz[i]};
      double h[3];
      d_matmul<3, 3, 1>(cam.K, P3, h);
      double u = h[0] / h[2], v = h[1] / h[2];
      bool vis = (u > 0) && (u < cam.W) && (v > 0) && (v < cam.H) && ((flg[i] & 3) == 3);
      ((int2*)Px)[i] = make_int2(d_np_int32(u), d_np_int32(v));  // renderer.py:43,50 np.int32(...)
      flg[i] = vis ? 4 : 0;
    }
    __syncthreads();
    for (int e = tid; e < ne; e += TC_NT) {  // camera.py:95
      int2 ed = LE[e];
      if ((flg[ed.x] | flg[ed.y]) & 4) {
        int k = atomicAdd(seg_cnt, 1);
        if (k < a.lds.seg_cap) {
          int2 pa = ((int2*)Px)[ed.x], pb = ((int2*)Px)[ed.y];
          int* o = seg + 5 * k;
          o[0] = l;
          o[1] = pa.x;
          o[2] = pa.y;
          o[3] = pb.x;
          o[4] = pb.y;
        }
      }
    }
    __syncthreads();
  }
  int nseg = *seg_cnt;
  if (nseg > a.lds.seg_cap) nseg = a.lds.seg_cap;  // cannot happen: seg_cap == total edge count

  const int H = cam.H, W = cam.W, wpr = cam.wpr, C = m.C;
  unsigned char* out = b.obs + (size_t)env * ((size_t)H * W * (cam.format == TC_FMT_CLASSES ? C : 3));
  for (int band = 0; band < cam.n_bands; band++) {
    const int y0 = band * cam.band_rows;
    const int y1 = (y0 + cam.band_rows < H) ? y0 + cam.band_rows : H;
    const int rows = y1 - y0;
    const int nwords = C * cam.band_rows * wpr;
    for (int i = tid; i < nwords; i += TC_NT) bits[i] = 0;
    __syncthreads();
    for (int k = tid; k < nseg && !(flags & DBG_SKIP_RASTER); k += TC_NT) {
      const int* sg = seg + 5 * k;
      Ras r;
      r.bits = bits + sg[0] * cam.band_rows * wpr;
      r.W = W;
      r.H = H;
      r.wpr = wpr;
      r.y0 = y0;
      r.y1 = y1;
      r_thick_line(r, sg[1], sg[2], sg[3], sg[4], cam.thickness);
    }
    __syncthreads();
    if (flags & DBG_SKIP_STORE) {
    } else if (cam.format == TC_FMT_CLASSES) {
      if ((W & 15) == 0) {
        // 16 pixels -> one 16-byte store per lane, consecutive lanes on consecutive addresses
        const int per_plane = rows * W / 16;
        const int total = C * per_plane;
        for (int q = tid; q < total; q += TC_NT) {
          int c = q / per_plane, r16 = q - c * per_plane;
          int pix = r16 * 16;
          int yy = pix / W, xx = pix - yy * W;
          unsigned int word = bits[(c * cam.band_rows + yy) * wpr + (xx >> 5)];
          unsigned int b16 = (word >> (xx & 31)) & 0xffffu;
          uint4 o;
          o.x = spread4(b16);
          o.y = spread4(b16 >> 4);
          o.z = spread4(b16 >> 8);
          o.w = spread4(b16 >> 12);
          *(uint4*)(out + ((size_t)c * H + y0 + yy) * W + xx) = o;
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
      if ((W & 3) == 0) {
        const int total = rows * W / 4;
        for (int g = tid; g < total; g += TC_NT) {
          int pix = g * 4;
          int yy = pix / W, xx = pix - yy * W;
          unsigned char px[12];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            int top = -1;
            for (int c = 0; c < C; c++) {
              unsigned int word = bits[(c * cam.band_rows + yy) * wpr + ((xx + k) >> 5)];
              if ((word >> ((xx + k) & 31)) & 1u) top = c;
            }
            px[3 * k] = top >= 0 ? m.colors[top][0] : 0;
            px[3 * k + 1] = top >= 0 ? m.colors[top][1] : 0;
            px[3 * k + 2] = top >= 0 ? m.colors[top][2] : 0;
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
          o[0] = top >= 0 ? m.colors[top][0] : 0;
          o[1] = top >= 0 ? m.colors[top][1] : 0;
          o[2] = top >= 0 ? m.colors[top][2] : 0;
        }
      }
    }
    __syncthreads();
  }
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
};

struct tc_env {
  const tc_map* map;
  KArgs k;
  bool bound;
  int64_t obs_bytes;
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
  std::vector<int2> edges(TE), lpe(lpE);
  std::vector<double> of(TE), orv(TE), lpo(lpE), nori(lpE), pori(lpE);
  std::vector<int> noff(lpN + 1, 0), poff(lpN + 1, 0), nnode(lpE), pnode(lpE);
  for (int i = 0; i < TN; i++) nodes[i] = make_double2(desc->nodes[2 * i], desc->nodes[2 * i + 1]);
  for (int l = 0; l < C; l++)
    for (int e = d.edge_off[l]; e < d.edge_off[l + 1]; e++) {
      edges[e] = make_int2(desc->edges[2 * e], desc->edges[2 * e + 1]);
      double2 a = nodes[d.node_off[l] + edges[e].x], b = nodes[d.node_off[l] + edges[e].y];
      double evx = b.x - a.x, evy = b.y - a.y;
      // static halves of layer.py:140-141, evaluated with the host libm like the reference does
      of[e] = atan2(evy, evx);
      orv[e] = atan2(-evy, -evx);
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
  int rc = TC_OK;
  HIP_TRY(hipGetDevice(&m->device));
#define UP(vec, field)                                             \
  if (rc == TC_OK) rc = upload(m, vec, &d.field);
  UP(nodes, nodes) UP(edges, edges) UP(of, ori_fwd) UP(orv, ori_rev) UP(lpn, lp_nodes) UP(lpe, lp_edges)
  UP(lpo, lp_ori) UP(noff, next_off) UP(nnode, next_node) UP(nori, next_ori) UP(poff, prev_off)
  UP(pnode, prev_node) UP(pori, prev_ori)
#undef UP
  if (rc != TC_OK) {
    tc_map_destroy(m);
    return rc;
  }
  *out = m;
  return TC_OK;
}

static int align_up(int v, int a) { return (v + a - 1) / a * a; }

static int fill_camera(tc_env* e, const tc_camera_params* cam) {
  if (cam->height < 1 || cam->width < 1 || cam->line_thickness < 1 || !(cam->max_range > 0) ||
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
  LdsLayout& L = e->k.lds;
  int off = 0;
  L.off_p = off;
  int pcount = 3 * m.max_nodes > m.total_nodes ? 3 * m.max_nodes : m.total_nodes;
  off += align_up(pcount * 8, 16);
  L.off_flg = off;
  off += align_up(m.max_nodes, 16);
  L.off_list = off;
  off += align_up(m.max_edges * 4, 16);
  L.off_seg = off;
  L.seg_cap = m.total_edges;
  off += align_up(L.seg_cap * 5 * 4, 16);
  L.off_bits = off;
  off += align_up(m.C * band_rows * dc.wpr * 4, 16);
  L.off_cnt = off;
  off += 16;
  L.total = off;
  if (L.total > 160 * 1024) {
    set_err("tc_env_create: map too large for one workgroup's LDS");
    delete e;
    return TC_E_LDS;
  }
  if (L.total > 48 * 1024) {
    hipError_t he = hipFuncSetAttribute((const void*)tc_env_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
    if (he != hipSuccess) {
      set_err(std::string("hipFuncSetAttribute: ") + hipGetErrorString(he));
      delete e;
      return TC_E_HIP;
    }
  }
  e->obs_bytes = (int64_t)dc.H * dc.W * (dc.format == TC_FMT_CLASSES ? m.C : 3);
  *out = e;
  return TC_OK;
}

extern "C" int tc_env_destroy(tc_env* e) {
  delete e;
  return TC_OK;
}

extern "C" int64_t tc_env_obs_bytes(const tc_env* e) { return e ? e->obs_bytes : TC_E_INVALID; }
extern "C" int64_t tc_env_lds_bytes(const tc_env* e) { return e ? e->k.lds.total : TC_E_INVALID; }

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

static int launch(tc_env* e, int mode, const void* cc, int cdtype, const int32_t* man, const int32_t* spawn,
                  const uint8_t* mask, uint32_t flags, void* stream) {
  if (!e) return TC_E_INVALID;
  if (!e->bound) {
    set_err("tc_env_bind has not been called");
    return TC_E_UNBOUND;
  }
  if ((flags & TC_F_AUTORESET) && !e->k.b.needs_reset) {
    set_err("TC_F_AUTORESET needs needs_reset/spawn_queue/spawn_cursor buffers");
    return TC_E_INVALID;
  }
  hipLaunchKernelGGL(tc_env_kernel, dim3(e->k.N), dim3(TC_NT), e->k.lds.total, (hipStream_t)stream, e->k, mode, cc,
                     cdtype, man, spawn, mask, flags);
  HIP_TRY(hipGetLastError());
  return TC_OK;
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

extern "C" int tc_render(tc_env* e, uint32_t flags, void* stream) {
  return launch(e, MODE_RENDER, nullptr, TC_F32, nullptr, nullptr, nullptr, flags & ~TC_F_NO_OBSERVATION, stream);
}
