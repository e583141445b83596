// tc_device.h -- device-side building blocks of the batched tinycarlo step (gfx950 / CDNA4).
//
// Everything here is double precision with the reference's own formulation (no FMA contraction:
// the library is built with -ffp-contract=off), transcendental functions from tc_trig.h, and static
// edge orientations read from tables computed on the host with libm (they are constants of the
// map; the reference evaluates math.atan2 on them at run time, layer.py:122,140-141,181).
//
// Execution model: ONE WAVEFRONT (64 lanes) PER ENV.  Scalar per-env work (kinematics, lanepath
// tracking) is computed redundantly by all lanes (same wave-instruction count as one active lane,
// no LDS broadcast needed); data-parallel work (argmin over edges, node transforms, segment
// rasterisation, observation stores) is strided over the 64 lanes.
#ifndef TC_DEVICE_H
#define TC_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tc_trig.h"

#define TC_NT 64  // threads per workgroup == wavefront size
#define TC_PI 3.141592653589793

// Lanepath node with its adjacency inlined, so that lanepath tracking (car.py:127-148) costs one memory
// round trip per visited node instead of one per CSR array touched.  Lists longer than 3 keep their
// true length in nnext / nprev and are then read from the CSR arrays instead.
struct LpNode {
  double x, y;
  int nnext, nprev;
  int next[3];
  int prev[3];
  double next_ori[3];  // atan2(nodes[next[k]] - this)  (layer.py:122 / 179-181)
  double prev_ori[3];  // atan2(nodes[prev[k]] - this)
};

struct DevMap {
  int C;
  int node_off[17];
  int edge_off[17];
  int max_nodes, max_edges, total_nodes, total_edges;
  const double2* nodes;   // lane-line nodes, all layers concatenated
  const int2* edges;      // lane-line edges (layer-local node ids)
  const int2* edges_g;    // the same edges with global node ids (node_off[layer] + local id)
  const int* edge_layer;  // layer of each edge
  const double* ori_fwd;  // per lane-line edge: atan2(evy, evx)
  const double* ori_rev;  // per lane-line edge: atan2(-evy, -evx)
  const double4* edge_xy; // per lane-line edge: both end points (n0.x, n0.y, n1.x, n1.y) in one 32-byte record, so that a
                          // scan over edges is one independent load per edge instead of edge -> two dependent node loads
  // Candidate grid for layer.py:33-44 (nearest edge = first minimum of |d(p, n0) + d(p, n1)| over a layer's edges).
  // A uniform grid of square cells over the map's bounding box + margin; for cell c and layer l the host lists, in
  // ascending edge order, every edge that can attain the minimum for SOME point of the cell: with f_c(e) the value at the
  // cell centre and r the half diagonal, |f_p(e) - f_c(e)| <= 2 r for every p in the cell (each of the two distances
  // moves by at most |p - centre|), so the minimising edge of p satisfies f_c(e) <= min_e f_c(e) + 4 r.  Scanning the
  // list with the same strict `<` as the full scan therefore returns the same edge, ties included (every edge that
  // attains the minimum is in the list).  Points outside the grid (or non-finite) take the full scan.
  // The grid reaches 4 m beyond the lane lines (under random actions ~2 % of the cars are more than 1 m off the road at any
  // time and the furthest about 3.3 m: beyond 10 track widths of cross-track error the default env terminates).  It has to
  // cover them: in the grouped kernel a wavefront whose envs took different paths -- candidates for seven, the full scan
  // for one -- pays for both, and a launch lasts as long as its slowest wavefront.
  int grid_nx, grid_ny;          // 0: no grid (empty map, non-finite coordinates)
  double grid_x0, grid_y0, grid_inv;  // cell (ix, iy) = floor((p - origin) * grid_inv)
  const int* cand_off;           // [grid_nx * grid_ny * C + 1]: entries of (cell, layer) are cand_idx[off[cell*C+l] .. off[cell*C+l+1])
  const int* cand_idx;           // global lane-line edge ids (edge_off[l] + local id)
  unsigned char colors[16][3];
  int lpN, lpE;
  int first_spawnable;    // a lanepath node with an out-edge (fallback for invalid spawn requests)
  const struct LpNode* lp_fat;  // one record per lanepath node: position + first 3 next/prev neighbours
  const double2* lp_nodes;
  const int2* lp_edges;
  const double* lp_ori;   // per lanepath edge orientation
  const int* next_off;    // CSR of get_next_nodes (edge-list order kept)
  const int* next_node;
  const double* next_ori;
  const int* prev_off;    // CSR of get_prev_nodes
  const int* prev_node;
  const double* prev_ori;
};

struct DevCar {
  double T, wheelbase, track_width, max_velocity, max_steering_angle;
  double steering_speed, max_acceleration, max_deceleration;
  int has_steering_speed, has_max_acceleration;
};

struct DevCam {
  int H, W;
  double E[12];
  double K[9];
  double max_range;
  int thickness;
  int format;
  int wpr;        // 32-bit words per bit-plane row
  int band_rows;  // rows rasterised per pass (bit-planes of one band live in LDS)
  int n_bands;
};

struct CarState {
  double x, y, theta, velocity, steering, radius, front_x, front_y;
  double cth, sth;  // cos(theta), sin(theta) of the current heading (not stored: recomputed when state is loaded)
  int lp[8];
  int lp_len, last_maneuver;
};

// ------------------------------------------------------------------ scalar helpers
__device__ inline double d_clip_angle(double a) {  // helper.py:11-19 (bounded, see oracle)
  int guard = 0;
#pragma nounroll
  while (a > TC_PI && guard++ < 64) a -= 2 * TC_PI;
#pragma nounroll
  while (a < -TC_PI && guard++ < 128) a += 2 * TC_PI;
  return a;
}

__device__ inline double d_np_clip(double x, double lo, double hi) {  // numpy clip ufunc
  double m = (x != x) ? x : (x > lo ? x : lo);
  return (m != m) ? m : (m < hi ? m : hi);
}

__device__ inline double d_dist(double ax, double ay, double bx, double by) {  // layer.py:187
  double dx = ax - bx, dy = ay - by;
  return sqrt(dx * dx + dy * dy);
}

__device__ inline double d_radians(double d) { return d * (TC_PI / 180.0); }

// lowest-index argmin across the wave; idx < 0 means "no candidate"
__device__ inline void wave_argmin(double& v, int& idx) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double ov = __shfl_down(v, off);
    int oi = __shfl_down(idx, off);
    bool take = (oi >= 0) && (idx < 0 || ov < v || (ov == v && oi < idx));
    if (take) {
      v = ov;
      idx = oi;
    }
  }
  v = __shfl(v, 0);
  idx = __shfl(idx, 0);
}

// The same for TC_AG layers at once: the TC_AG reductions are independent, so their shuffles are issued together and each
// of the 6 stages waits for the LDS crossbar once instead of once per layer (one reduction after the other was 5 x ~2 k
// clocks of a wavefront's step on simple_layout, almost all of it latency).  bd / best are indexed by unrolled constants
// only and stay in registers.
#define TC_AG 5
__device__ __forceinline__ void wave_argmin_group(double (&bd)[TC_AG], int (&best)[TC_AG]) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double ov[TC_AG];
    int oi[TC_AG];
#pragma unroll
    for (int g = 0; g < TC_AG; g++) {
      ov[g] = __shfl_down(bd[g], off);
      oi[g] = __shfl_down(best[g], off);
    }
#pragma unroll
    for (int g = 0; g < TC_AG; g++) {
      const bool take = (oi[g] >= 0) && (best[g] < 0 || ov[g] < bd[g] || (ov[g] == bd[g] && oi[g] < best[g]));
      bd[g] = take ? ov[g] : bd[g];
      best[g] = take ? oi[g] : best[g];
    }
  }
#pragma unroll
  for (int g = 0; g < TC_AG; g++) best[g] = __builtin_amdgcn_readfirstlane(best[g]);  // lane 0 holds the result
}

// ------------------------------------------------------------------ layer.py
// layer.py:144-164
__device__ inline double d_distance_to_edge(double n1x, double n1y, double n2x, double n2y, double px, double py) {
  double lvx = n2x - n1x, lvy = n2y - n1y;
  double pvx = px - n1x, pvy = py - n1y;
  if (lvx == 0) {
    if (lvy > 0) return px - n1x;
    return n1x - px;
  }
  return (pvx * lvy - pvy * lvx) / sqrt(lvx * lvx + lvy * lvy);
}

// layer.py:126-142
__device__ inline bool d_within_bounds(double n0x, double n0y, double n1x, double n1y, double ori_fwd, double ori_rev,
                                       double px, double py) {
  if (px == n0x && py == n0y) return true;
  if (px == n1x && py == n1y) return true;
  double a0 = tc_fabs(d_clip_angle(tc_atan2(py - n0y, px - n0x) - ori_fwd));
  double a1 = tc_fabs(d_clip_angle(tc_atan2(py - n1y, px - n1x) - ori_rev));
  return a0 <= TC_PI / 2 && a1 <= TC_PI / 2;
}

// The same decision without the two atan2 whenever it is not close: |angle(p - n, +-edge)| <= pi/2 is the sign of a dot
// product.  The reference's a0 / a1 carry at most a few 1e-16 of rounding (two atan2 of <= 1-2 ulp, one subtraction,
// clip_angle's +-2 pi), so its comparison with pi/2 can only disagree with exact arithmetic when the true angle is
// within ~1e-14 of pi/2; here the dot product decides only when |cos| of that angle exceeds 1e-6 (i.e. the angle is
// at least 1e-6 rad away from pi/2), which also dwarfs the dot product's own rounding error (~1e-16 relative to
// |v||e|).  Everything else -- angles within 1e-6 of pi/2, zero-length edges, NaNs -- reports `certain = false` and
// the caller evaluates d_within_bounds itself.  v = p - n and e = n1 - n0 are the very (rounded) vectors the reference
// hands to atan2 (layer.py:140-141), so both routes judge the same angle.
__device__ inline bool d_within_bounds_filter(double n0x, double n0y, double n1x, double n1y, double px, double py,
                                              bool& certain) {
  certain = true;
  if (px == n0x && py == n0y) return true;
  if (px == n1x && py == n1y) return true;
  const double ex = n1x - n0x, ey = n1y - n0y;
  const double ax = px - n0x, ay = py - n0y, bx = px - n1x, by = py - n1y;
  const double dot0 = ax * ex + ay * ey;     // angle between (p - n0) and the edge direction
  const double dot1 = -(bx * ex + by * ey);  // angle between (p - n1) and the reversed edge
  const double ee = ex * ex + ey * ey;
  const double lim0 = 1e-12 * ((ax * ax + ay * ay) * ee), lim1 = 1e-12 * ((bx * bx + by * by) * ee);
  certain = (dot0 * dot0 > lim0) && (dot1 * dot1 > lim1);  // false for NaNs and zero-length vectors
  return dot0 > 0 && dot1 > 0;
}

// layer.py:179-181 for a lanepath edge (a,b): table entry of the first a->b edge
__device__ inline double d_lp_edge_ori(const DevMap& m, int a, int b) {
  int s0 = m.next_off[a], s1 = m.next_off[a + 1];
  for (int s = s0; s < s1; s++)
    if (m.next_node[s] == b) return m.next_ori[s];
  double2 na = m.lp_nodes[a], nb = m.lp_nodes[b];
  return tc_atan2(nb.y - na.y, nb.x - na.x);
}

// layer.py:105-124 on a CSR slice; returns node id or -1 (None)
__device__ inline int d_pick_node(const int* lst, const double* ori, int s0, int s1, int node_idx, double orientation,
                                  int& status) {
  int n = s1 - s0;
  if (n == 0) return -1;
  if (n <= 1) return lst[s0];
  int best = -1, k = 0;
  double bk = 0;
  for (int s = s0; s < s1; s++) {
    if (lst[s] == node_idx) continue;
    double key = tc_fabs(d_clip_angle(ori[s] - orientation));
    if (best < 0 || key < bk) {
      best = k;
      bk = key;
    }
    k++;
  }
  if (best < 0) {
    status |= 2;  // TC_S_PICK_EMPTY
    return -1;
  }
  return lst[s0 + best];
}

// ---- the same two queries on an inlined LpNode record (fallback to CSR when a list has > 3 entries)
__device__ inline int sel3i(int a0, int a1, int a2, int i) {
  int r = a0;
  r = i == 1 ? a1 : r;
  r = i == 2 ? a2 : r;
  return r;
}

__device__ inline double d_edge_ori_f(const DevMap& m, const LpNode& A, int a, int b) {
  if (A.nnext <= 3) {
    if (A.nnext > 0 && A.next[0] == b) return A.next_ori[0];
    if (A.nnext > 1 && A.next[1] == b) return A.next_ori[1];
    if (A.nnext > 2 && A.next[2] == b) return A.next_ori[2];
  }
  return d_lp_edge_ori(m, a, b);
}

// layer.py:105-124 on up to 3 inlined neighbours
__device__ inline int d_pick3(int l0, int l1, int l2, double o0, double o1, double o2, int n, int node_idx,
                              double orientation, int& status) {
  if (n == 0) return -1;
  if (n <= 1) return l0;
  int best = -1, k = 0;
  double bk = 0;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int lj = j == 0 ? l0 : j == 1 ? l1 : l2;
    const double oj = j == 0 ? o0 : j == 1 ? o1 : o2;
    if (j < n && lj != node_idx) {
      double key = tc_fabs(d_clip_angle(oj - orientation));
      if (best < 0 || key < bk) {
        best = k;
        bk = key;
      }
      k++;
    }
  }
  if (best < 0) {
    status |= 2;  // TC_S_PICK_EMPTY
    return -1;
  }
  return sel3i(l0, l1, l2, best);  // index into the UNFILTERED list (layer.py:122-124)
}

__device__ inline int d_pick_next_f(const DevMap& m, const LpNode& A, int node_idx, double orientation, int& status) {
  if (A.nnext <= 3)
    return d_pick3(A.next[0], A.next[1], A.next[2], A.next_ori[0], A.next_ori[1], A.next_ori[2], A.nnext, node_idx,
                   orientation, status);
  return d_pick_node(m.next_node, m.next_ori, m.next_off[node_idx], m.next_off[node_idx + 1], node_idx, orientation, status);
}

__device__ inline int d_pick_prev_f(const DevMap& m, const LpNode& A, int node_idx, double orientation, int& status) {
  if (A.nprev <= 3)
    return d_pick3(A.prev[0], A.prev[1], A.prev[2], A.prev_ori[0], A.prev_ori[1], A.prev_ori[2], A.nprev, node_idx,
                   orientation, status);
  return d_pick_node(m.prev_node, m.prev_ori, m.prev_off[node_idx], m.prev_off[node_idx + 1], node_idx, orientation, status);
}

// layer.py:59-74 over the lanepath, all 64 lanes cooperating; returns edge index or -1
__device__ inline int d_nearest_edge_with_orientation(const DevMap& m, double px, double py, double orientation,
                                                      double margin_deg, const int tid) {
  double lim = d_radians(margin_deg);
  int best = -1;
  double bd = 0;
  for (int e = tid; e < m.lpE; e += TC_NT) {
    if (!(tc_fabs(d_clip_angle(m.lp_ori[e] - orientation)) <= lim)) continue;
    int2 ed = m.lp_edges[e];
    double2 a = m.lp_nodes[ed.x], b = m.lp_nodes[ed.y];
    double d = tc_fabs(d_dist(px, py, a.x, a.y) + d_dist(px, py, b.x, b.y));
    if (best < 0 || d < bd) {
      best = e;
      bd = d;
    }
  }
  wave_argmin(bd, best);
  return best;
}

// cell of the candidate grid holding p, or -1 (no grid, p outside it or not finite): see DevMap
__device__ __forceinline__ int d_grid_cell(const DevMap& m, double px, double py) {
  const double gx = (px - m.grid_x0) * m.grid_inv, gy = (py - m.grid_y0) * m.grid_inv;
  const bool in = m.grid_nx > 0 && gx >= 0.0 && gx < (double)m.grid_nx && gy >= 0.0 && gy < (double)m.grid_ny;
  return in ? (int)gy * m.grid_nx + (int)gx : -1;
}

// lowest-index argmin among the EL lanes of an aligned lane group (EL a power of two < 64): xor butterfly, after which
// every lane of the group holds the result.  All lanes of a group must be active (group-uniform control flow).
template <int EL>
__device__ __forceinline__ void group_argmin(double& v, int& idx) {
#pragma unroll
  for (int off = EL / 2; off > 0; off >>= 1) {
    const double ov = __shfl_xor(v, off);
    const int oi = __shfl_xor(idx, off);
    const bool take = (oi >= 0) && (idx < 0 || ov < v || (ov == v && oi < idx));
    v = take ? ov : v;
    idx = take ? oi : idx;
  }
}

// The same reduction for groups of 8 lanes and TC_AG independent (value, index) pairs, with DPP moves -- register to
// register inside the SIMD (quad_perm xor 1, quad_perm xor 2, row_half_mirror i <-> 7 - i) -- instead of ds_bpermute
// round trips through the LDS crossbar (~1.1 k clocks per reduction in the grouped simulate kernel, five per step).
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  const int lo = dpp_i<CTRL>(__double2loint(v)), hi = dpp_i<CTRL>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ void group8_argmin_stage(double (&bd)[TC_AG], int (&best)[TC_AG]) {
#pragma unroll
  for (int g = 0; g < TC_AG; g++) {
    const double ov = dpp_d<CTRL>(bd[g]);
    const int oi = dpp_i<CTRL>(best[g]);
    const bool take = (oi >= 0) && (best[g] < 0 || ov < bd[g] || (ov == bd[g] && oi < best[g]));
    bd[g] = take ? ov : bd[g];
    best[g] = take ? oi : best[g];
  }
}
__device__ __forceinline__ void group8_argmin_multi(double (&bd)[TC_AG], int (&best)[TC_AG]) {
  group8_argmin_stage<0xB1>(bd, best);   // quad_perm [1,0,3,2]
  group8_argmin_stage<0x4E>(bd, best);   // quad_perm [2,3,0,1]
  group8_argmin_stage<0x141>(bd, best);  // row_half_mirror: the other quad of the 8
}

// layer.py:59-74 over the lanepath by the EL lanes of one env's group (sub = lane within the group)
template <int EL>
__device__ inline int d_nearest_edge_with_orientation_g(const DevMap& m, double px, double py, double orientation,
                                                        double margin_deg, const int sub) {
  double lim = d_radians(margin_deg);
  int best = -1;
  double bd = 0;
  for (int e = sub; e < m.lpE; e += EL) {
    if (!(tc_fabs(d_clip_angle(m.lp_ori[e] - orientation)) <= lim)) continue;
    int2 ed = m.lp_edges[e];
    double2 a = m.lp_nodes[ed.x], b = m.lp_nodes[ed.y];
    double d = tc_fabs(d_dist(px, py, a.x, a.y) + d_dist(px, py, b.x, b.y));
    if (best < 0 || d < bd) {
      best = e;
      bd = d;
    }
  }
  group_argmin<EL>(bd, best);
  return best;
}

// ------------------------------------------------------------------ car.py
__device__ inline void d_update_front(const DevCar& c, CarState& s) {  // car.py:167-168
  s.cth = tc_cos(s.theta);
  s.sth = tc_sin(s.theta);
  s.front_x = s.x + c.wheelbase * s.cth;
  s.front_y = s.y + c.wheelbase * s.sth;
}

// car.py:34-44 + map.py:62-69 with the spawn node already drawn
__device__ inline void d_reset(const DevMap& m, const DevCar& c, CarState& s, int node) {
  const LpNode F = m.lp_fat[node];
  s.x = F.x;
  s.y = F.y;
  s.theta = F.next_ori[0];
#pragma unroll
  for (int i = 0; i < 8; i++) s.lp[i] = -1;
  s.lp[0] = node;
  s.lp[1] = F.next[0];
  s.lp_len = 1;
  d_update_front(c, s);
  s.steering = 0.0;
  s.radius = 0.0;
  s.velocity = 0.0;
  s.last_maneuver = 0;
}

// What car.get_info needs about local_path[1] (car.py:52-53), captured while the records are in registers
struct PathInfo {
  double ax, ay, bx, by;  // nodes of local_path[1]
  double ori;             // orientation of local_path[1]
  int valid;              // 0: local_path was left untouched (early truncation) -> read it from the tables
};

// Where lanepath tracking reads the fat node records from: the map's table in global memory, or a copy of it in the
// workgroup's LDS (the grouped simulate kernel: its tracking chain is 3-5 DEPENDENT record reads per step, and beside
// the frame kernel a global load queues behind that kernel's store bursts on the same CU).
struct FatGlobal {
  const LpNode* fat;
  const double2* nodes;
  __device__ __forceinline__ LpNode get(int i) const { return fat[i]; }
  __device__ __forceinline__ double2 pos(int i) const { return nodes[i]; }
};
struct FatLds {
  const __attribute__((address_space(3))) int* base;  // records of sizeof(LpNode) bytes, 16-byte aligned
  __device__ __forceinline__ LpNode get(int i) const {
    union {
      LpNode n;
      int w[sizeof(LpNode) / 4];
    } u;
    const __attribute__((address_space(3))) int* p = base + i * (int)(sizeof(LpNode) / 4);
#pragma unroll
    for (int j = 0; j < (int)(sizeof(LpNode) / 4); j++) u.w[j] = p[j];  // (consecutive words: merged into 16-byte LDS reads)
    return u.n;
  }
  __device__ __forceinline__ double2 pos(int i) const {
    const __attribute__((address_space(3))) int* p = base + i * (int)(sizeof(LpNode) / 4);
    return make_double2(__hiloint2double(p[1], p[0]), __hiloint2double(p[3], p[2]));
  }
};

// car.py:127-148.  EL = lanes that work on this env together: 64 (one wavefront per env, tid = lane) or a lane group
// of the grouped simulate kernel (tid = lane within the group); it only matters for the U-turn search.
template <int EL = 64, class Fat = FatGlobal>
__device__ inline int d_find_local_path(const DevMap& m, const Fat& fat, CarState& s, int maneuver, int& status, PathInfo& pi,
                                        const int tid) {
  double fx = s.front_x, fy = s.front_y;
  int e0 = s.lp[0], e1 = s.lp[1];
  const LpNode N0 = fat.get(e0), N1 = fat.get(e1);
  double mdir = d_clip_angle(d_edge_ori_f(m, N0, e0, e1) + (maneuver * TC_PI) / 2);
  int ne0, ne1;
  if (maneuver == 2 && s.last_maneuver != 2) {  // wave-uniform branch
    int e = EL == 64 ? d_nearest_edge_with_orientation(m, fx, fy, mdir, 30.0, tid)
                     : d_nearest_edge_with_orientation_g<(EL == 64 ? 32 : EL)>(m, fx, fy, mdir, 30.0, tid);
    mdir = d_clip_angle(mdir + TC_PI);
    if (e < 0) {
      status |= 1;  // TC_S_UTURN_NO_EDGE
      return 1;
    }
    int2 ed = m.lp_edges[e];
    ne0 = ed.x;
    ne1 = ed.y;
  } else {  // layer.py:77-103
    int nx = d_pick_next_f(m, N1, e1, mdir, status);
    int pv = d_pick_prev_f(m, N0, e0, mdir, status);
    if (nx < 0 || pv < 0) return 1;
    double2 nn = fat.pos(nx), np = fat.pos(pv);
    double d0 = d_dist(fx, fy, N0.x, N0.y), d1 = d_dist(fx, fy, N1.x, N1.y);
    double dn = d_dist(fx, fy, nn.x, nn.y), dp = d_dist(fx, fy, np.x, np.y);
    if (dn < d0 && dn < d1) {
      ne0 = e1;
      ne1 = nx;
    } else if (dp < d0 && dp < d1) {
      ne0 = pv;
      ne1 = e0;
    } else {
      ne0 = e0;
      ne1 = e1;
    }
  }
  s.last_maneuver = maneuver;
  s.lp[0] = ne0;
  s.lp[1] = ne1;
  s.lp_len = 1;
  const bool fwd = s.velocity > 0;  // car.py:143
  int node = fwd ? ne1 : ne0;
  LpNode A = fat.get(node);
#pragma unroll
  for (int i = 0; i < 3; i++) {
    int nn = d_pick_next_f(m, A, node, mdir, status);
    if (nn < 0) return 1;
    s.lp[2 * (i + 1)] = node;
    s.lp[2 * (i + 1) + 1] = nn;
    s.lp_len = i + 2;
    if (i == 0) {
      pi.valid = 1;
      pi.ax = A.x;
      pi.ay = A.y;
      pi.ori = d_edge_ori_f(m, A, node, nn);
      if (!fwd) {  // the next record loaded is `node` again: fetch nn's position on its own
        double2 q = fat.pos(nn);
        pi.bx = q.x;
        pi.by = q.y;
      }
    }
    if (fwd) {  // velocity > 0: continue from the new end node; otherwise from the same start node
      node = nn;
      if (i < 2) A = fat.get(node);
      if (i == 0) {
        pi.bx = A.x;
        pi.by = A.y;
      }
    }
  }
  return 0;
}

// car.py:70-125; returns truncated
// have_trig: s.cth / s.sth already hold cos / sin of s.theta (left there by the front-axle update of the previous step
// of the same launch): the same function of the same argument, so reusing them changes no bit
__device__ inline void d_car_kinematics(const DevCar& c, CarState& s, double v_in, double s_in, bool have_trig) {
  double dt = c.T;
  double nv = v_in * c.max_velocity;
  if (c.has_max_acceleration)
    nv = d_np_clip(nv, s.velocity - c.max_deceleration * dt, s.velocity + c.max_acceleration * dt);
  s.velocity = nv;
  double ns = s_in * c.max_steering_angle;
  if (c.has_steering_speed) ns = d_np_clip(ns, s.steering - c.steering_speed * dt, s.steering + c.steering_speed * dt);
  s.steering = ns;
  double vxn, vyn;
  if (have_trig) {
    vxn = s.cth;
    vyn = s.sth;
  } else {
    vxn = tc_cos(s.theta);
    vyn = tc_sin(s.theta);
  }
  if (tc_fabs(s.steering) < 0.0001) {
    s.radius = 0;
    s.x = s.x + s.velocity * vxn * dt;
    s.y = s.y + s.velocity * vyn * dt;
    // heading unchanged: car.py:167-168 evaluates cos/sin of the same angle again
    s.cth = vxn;
    s.sth = vyn;
    s.front_x = s.x + c.wheelbase * vxn;
    s.front_y = s.y + c.wheelbase * vyn;
  } else {
    s.radius = c.wheelbase / tc_tan(d_radians(s.steering));
    double ang_vel = s.velocity / s.radius;
    double dyaw = ang_vel * dt;
    double nx = vyn, ny = -vxn;
    double tx = nx * s.radius, ty = ny * s.radius;
    double cd = tc_cos(dyaw), sd = tc_sin(dyaw);
    // R_M.dot([tx, ty]) (car.py:111-113) is numpy's dgemv: row i = fma(R[i][0], tx, R[i][1] * ty) on x86-64 with FMA
    // (tools/numpy_matmul_probe.py: 5 000 of 5 000 random cases; the unfused form matches 67 %)
    double r0 = __builtin_fma(cd, tx, (-sd) * ty);
    double r1 = __builtin_fma(sd, tx, cd * ty);
    s.x = s.x - tx + r0;
    s.y = s.y - ty + r1;
    s.theta += dyaw;
    if (s.theta > TC_PI)
      s.theta -= 2 * TC_PI;
    else if (s.theta < -TC_PI)
      s.theta += 2 * TC_PI;
    d_update_front(c, s);
  }
}

// car.py:70-125 = kinematics (above), then lanepath tracking; returns truncated
__device__ inline int d_car_step(const DevMap& m, const DevCar& c, CarState& s, double v_in, double s_in, int maneuver,
                                 int& status, PathInfo& pi, bool have_trig, const int tid) {
  d_car_kinematics(c, s, v_in, s_in, have_trig);
  const FatGlobal fg = {m.lp_fat, m.lp_nodes};
  return d_find_local_path(m, fg, s, maneuver, status, pi, tid);  // ONE call site: the function is ~2 k instructions inlined
}

// ------------------------------------------------------------------ camera.py helpers
// C[i][j] = one chain of fused multiply-adds over ascending t from a zero accumulator: the association of numpy's
// `A @ B` (OpenBLAS dgemm on x86-64 with FMA) that the reference's camera runs on (camera.py:62,131,138; isolated with
// exact rational arithmetic by tools/numpy_matmul_probe.py: R@T, E@car3d, pose@points and K@P reproduce bit for bit
// with this form and with no other tried).
// The fma here is an explicit operation of the algorithm, not a contraction (the library is built -ffp-contract=off).
template <int N, int K, int P>
__device__ inline void d_matmul(const double* A, const double* B, double* C) {
#pragma unroll
  for (int i = 0; i < N; i++)
#pragma unroll
    for (int j = 0; j < P; j++) {
      double acc = 0.0;
#pragma unroll
      for (int t = 0; t < K; t++) acc = __builtin_fma(A[i * K + t], B[t * P + j], acc);
      C[i * P + j] = acc;
    }
}

// np.int32(float64) as x86-64 does it: out of range / NaN -> INT_MIN
__device__ inline int d_np_int32(double v) {
  if (!(v > -2147483649.0 && v < 2147483648.0)) return (int)0x80000000;
  return (int)v;
}

// ------------------------------------------------------------------ cv2.polylines restated on LDS bit-planes
// One bit per pixel and layer; `bits` points at the plane of the segment's layer for the current band.
#define TC_XY_SHIFT 16
#define TC_XY_ONE 65536

struct Ras {
  unsigned int* bits;
  int W, H, wpr;
  int y0, y1;  // band rows [y0, y1)
};

__device__ inline int d_wrap32(long long v) { return (int)(unsigned int)(unsigned long long)v; }

// Branch-free on purpose: divergent `if`s inside the pixel loops cost more exec-mask bookkeeping than the
// work they skip; an out-of-window pixel ORs 0 into word 0 of the plane instead.
__device__ inline void r_put(const Ras& r, int x, int y) {
  const bool ok = (unsigned)x < (unsigned)r.W && (unsigned)(y - r.y0) < (unsigned)(r.y1 - r.y0);
  const int idx = ok ? __mul24(y - r.y0, r.wpr) + (x >> 5) : 0;  // rows, words per row < 2^24: full-rate multiply
  atomicOr(&r.bits[idx], ok ? 1u << (x & 31) : 0u);
}

// inclusive span [xl, xr] on row y, clamped to the image and the band
__device__ inline void r_hline(const Ras& r, int y, int xl, int xr) {
  xl = xl < 0 ? 0 : xl;
  xr = xr > r.W - 1 ? r.W - 1 : xr;
  const bool ok = (unsigned)(y - r.y0) < (unsigned)(r.y1 - r.y0) && xl <= xr;
  const int w0 = xl >> 5;
  const int w1 = ok ? xr >> 5 : w0 - 1;
  unsigned int* row = r.bits + (ok ? __mul24(y - r.y0, r.wpr) : 0);
  for (int w = w0; w <= w1; w++) {
    const int lo = w == w0 ? (xl & 31) : 0;
    const int hi = w == w1 ? (xr & 31) : 31;
    atomicOr(&row[w], (0xffffffffu << lo) & (0xffffffffu >> (31 - hi)));
  }
}

// The same for a span that touches at most two consecutive words of its row -- any span when a row HAS only two words
// (frames up to 64 pixels wide), or a span of at most 32 pixels: no loop, two ORs (the second with an empty mask when the
// span stays inside one word)
__device__ inline void r_hline_2w(const Ras& r, int y, int xl, int xr) {
  xl = xl < 0 ? 0 : xl;
  xr = xr > r.W - 1 ? r.W - 1 : xr;
  const bool ok = (unsigned)(y - r.y0) < (unsigned)(r.y1 - r.y0) && xl <= xr;
  const int w0 = xl >> 5, w1 = xr >> 5;
  unsigned int* row = r.bits + (ok ? __mul24(y - r.y0, r.wpr) + w0 : 0);
  const unsigned int lo = 0xffffffffu << (xl & 31), hi = 0xffffffffu >> (31 - (xr & 31));
  const bool two = ok && w1 != w0;
  atomicOr(&row[0], ok ? (two ? lo : (lo & hi)) : 0u);
  atomicOr(&row[two ? 1 : 0], two ? hi : 0u);
}

// Filled circle (round cap) from its per-row half widths hw[|dy|], dy = -rad..rad: the pixel set of
// Circle(center, rad, fill) is the union of centred spans, so per row only the widest one matters.
// A centre more than 2^20 pixels away cannot reach a frame of at most 16384 x 16384 with rad < 32, whatever Circle()'s
// 64-bit arithmetic makes of it; everything nearer is plain 32-bit arithmetic.
// hw[] is indexed by the row offset, which is the same in every lane: read with scalar loads (a vector load here -- it
// used to be one, from the kernarg segment -- makes the wavefront wait for every store it has in flight: vector memory
// operations retire in order).  The table is 4-byte aligned (RCam) and constant for the kernel's lifetime.
__device__ __forceinline__ int r_cap_hw(const unsigned char* hw, int ady) {
  const unsigned int w = ((const __attribute__((address_space(4))) unsigned int*)(unsigned long long)hw)[ady >> 2];
  return (int)((w >> ((ady & 3) << 3)) & 255u);
}
__device__ inline void r_cap(const Ras& r, int cx, int cy, int rad, const unsigned char* hw, unsigned int hw4) {
  const bool near = (unsigned)cx + (1u << 20) <= (2u << 20) && (unsigned)cy + (1u << 20) <= (2u << 20);
  if (rad <= 15) {  // spans of at most 31 pixels
    for (int dy = -rad; dy <= rad; dy++) {
      const int ady = dy < 0 ? -dy : dy;
      const int h = rad <= 3 ? (int)((hw4 >> (8 * ady)) & 255u) : r_cap_hw(hw, ady);
      const int y = cy + dy, xl = cx - h, xr = cx + h;
      const bool ok = near && y >= 0 && y < r.H && xr >= 0 && xl < r.W;
      r_hline_2w(r, ok ? y : -1, ok ? xl : 1, ok ? xr : 0);  // (at most 31 pixels: two words at most)
    }
    return;
  }
  for (int dy = -rad; dy <= rad; dy++) {
    const int ady = dy < 0 ? -dy : dy;
    const int h = r_cap_hw(hw, ady);
    const int y = cy + dy, xl = cx - h, xr = cx + h;
    const bool ok = near && y >= 0 && y < r.H && xr >= 0 && xl < r.W;
    r_hline(r, ok ? y : -1, ok ? (xl < 0 ? 0 : xl) : 1, ok ? (xr > r.W - 1 ? r.W - 1 : xr) : 0);
  }
}

// clipLine(Size2l, Point2l&, Point2l&)
__device__ inline bool r_clip_line(long long width, long long height, long long& x1, long long& y1, long long& x2,
                                   long long& y2) {
  // clipLine() clips end 1 against the y range, then end 2 (against the already clipped end 1), then the same for x.
  // Under SIMT each of those four blocks (an f64 multiply + divide between int64 conversions) would be executed by
  // the whole wave as soon as one lane needs it.  Here ONE instance per axis serves whichever end needs it -- roles
  // are swapped for lanes where only end 2 does; (a-y2)*(x1-x2)/(y1-y2) equals (a-y2)*(x2-x1)/(y2-y1) bit for bit
  // because IEEE multiplication and division are sign-symmetric -- and a second instance runs only for lanes where
  // both ends need clipping on that axis (rare; skipped by the whole wave otherwise).
  int c1, c2;
  long long right = width - 1, bottom = height - 1;
  if (width <= 0 || height <= 0) return false;
  c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
  c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
  if ((c1 & c2) == 0 && (c1 | c2) != 0) {
    const bool n1 = (c1 & 12) != 0, n2 = (c2 & 12) != 0;
    if (n1 || n2) {
      const bool sw = !n1;  // only end 2 needs it: treat it as "the end to clip"
      long long xa = sw ? x2 : x1, ya = sw ? y2 : y1, xb = sw ? x1 : x2, yb = sw ? y1 : y2;
      const int ca = sw ? c2 : c1;
      const long long a = ca < 8 ? 0 : bottom;
      xa += (long long)((double)(a - ya) * (double)(xb - xa) / (double)(yb - ya));
      const int cn = (xa < 0) + (xa > right) * 2;
      if (sw) {
        x2 = xa;
        y2 = a;
        c2 = cn;
      } else {
        x1 = xa;
        y1 = a;
        c1 = cn;
      }
      if (n1 && n2) {  // both ends: end 2 against the clipped end 1
        const long long a2 = c2 < 8 ? 0 : bottom;
        x2 += (long long)((double)(a2 - y2) * (double)(x2 - x1) / (double)(y2 - y1));
        y2 = a2;
        c2 = (x2 < 0) + (x2 > right) * 2;
      }
    }
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
      const bool m1 = c1 != 0, m2 = c2 != 0;
      const bool sw = !m1;
      long long xa = sw ? x2 : x1, ya = sw ? y2 : y1, xb = sw ? x1 : x2, yb = sw ? y1 : y2;
      const int ca = sw ? c2 : c1;
      const long long a = ca == 1 ? 0 : right;
      ya += (long long)((double)(a - xa) * (double)(yb - ya) / (double)(xb - xa));
      if (sw) {
        x2 = a;
        y2 = ya;
        c2 = 0;
      } else {
        x1 = a;
        y1 = ya;
        c1 = 0;
      }
      if (m1 && m2) {
        const long long a2 = c2 == 1 ? 0 : right;
        y2 += (long long)((double)(a2 - x2) * (double)(y2 - y1) / (double)(x2 - x1));
        x2 = a2;
        c2 = 0;
      }
    }
  }
  return (c1 | c2) == 0;
}

// Line(): LineIterator(..., 8, leftToRight=true) -- thickness <= 1
__device__ inline void r_line_bresenham(const Ras& r, long long x1, long long y1, long long x2, long long y2) {
  if ((unsigned long long)x1 >= (unsigned long long)r.W || (unsigned long long)x2 >= (unsigned long long)r.W ||
      (unsigned long long)y1 >= (unsigned long long)r.H || (unsigned long long)y2 >= (unsigned long long)r.H) {
    if (!r_clip_line(r.W, r.H, x1, y1, x2, y2)) return;
  }
  long long dx = x2 - x1, dy = y2 - y1;
  int sy = 1;
  if (dx < 0) {
    dx = -dx;
    dy = -dy;
    x1 = x2;
    y1 = y2;
  }
  if (dy < 0) {
    dy = -dy;
    sy = -1;
  }
  bool vert = dy > dx;
  if (vert) {
    long long t = dx;
    dx = dy;
    dy = t;
  }
  long long err = dx - (dy + dy), plus = dx + dx, minus = -(dy + dy);
  int count = (int)dx + 1;
  int x = (int)x1, y = (int)y1;
  for (int i = 0; i < count; i++) {
    r_put(r, x, y);
    bool mask = err < 0;
    err += minus + (mask ? plus : 0);
    if (vert) {
      y += sy;
      if (mask) x += 1;
    } else {
      x += 1;
      if (mask) y += sy;
    }
  }
}

// Truncating int64 division n / d (C semantics) for operands below 2^50 (always the case after clipping):
// several times cheaper than the 64-bit integer division sequence.
__device__ inline long long d_sdiv(long long n, long long d) {
  long long an = n < 0 ? -n : n, ad = d < 0 ? -d : d;
  if (an < (1LL << 50) && ad < (1LL << 50)) {
    // quotient estimate from a hardware reciprocal (relative error ~2^-50: off by at most one for an < 2^50),
    // made exact by the remainder test -- the result is the true integer quotient, no rounding mode involved.
    // The remainder an - q * ad is taken with ONE fused multiply-add in double precision instead of a 64 x 64-bit
    // integer multiply (four quarter-rate v_mul / v_mad_u64_u32): an, ad and q are integers below 2^53, exact as doubles;
    // the fma forms q * ad exactly and rounds an - q * ad once, and that value is an integer of magnitude <= 2 ad < 2^51,
    // so the rounding is exact too.
    const double dn = (double)an, dd = (double)ad;
    double qd = __builtin_trunc(dn * __builtin_amdgcn_rcp(dd));
    double rd = __builtin_fma(-qd, dd, dn);
    if (rd < 0) {
      qd -= 1.0;
      rd += dd;
    }
    if (rd >= dd) qd += 1.0;
    const long long q = (long long)qd;
    return ((n < 0) != (d < 0)) ? -q : q;
  }
  return n / d;
}

// One polygon-outline edge after clipping, ready for random access by step index k (0..ecount):
//   x-major: pixel (a + k, (b + k*step) >> 16)        y-major: pixel ((b + k*step) >> 16, a + k)
// plus the far end point pixel (ex, ey) that Line2 writes first.
struct LineP {
  int a, b, step;
  int ecount;  // -1: edge invisible
  int ex, ey;
  int xmajor;
};

__device__ inline LineP r_line2_setup(int W, int H, long long p1x, long long p1y, long long p2x, long long p2y) {
  LineP L;
  L.a = L.b = L.step = 0;
  L.ecount = -1;
  L.ex = L.ey = -1;
  L.xmajor = 0;
  if (!r_clip_line((long long)W << TC_XY_SHIFT, (long long)H << TC_XY_SHIFT, p1x, p1y, p2x, p2y)) return L;
  long long dx = p2x - p1x, dy = p2y - p1y;
  long long j = dx < 0 ? -1 : 0;
  long long ax = (dx ^ j) - j;
  long long i = dy < 0 ? -1 : 0;
  long long ay = (dy ^ i) - i;
  bool xmajor = ax > ay;
  long long step;
  if (xmajor) {
    dy = (dy ^ j) - j;
    if (j) {
      long long t = p1x; p1x = p2x; p2x = t;
      t = p1y; p1y = p2y; p2y = t;
    }
    step = d_sdiv(dy * TC_XY_ONE, ax | 1);
    L.ecount = (int)((p2x - p1x) >> TC_XY_SHIFT);
  } else {
    dx = (dx ^ i) - i;
    if (i) {
      long long t = p1x; p1x = p2x; p2x = t;
      t = p1y; p1y = p2y; p2y = t;
    }
    step = d_sdiv(dx * TC_XY_ONE, ay | 1);
    L.ecount = (int)((p2y - p1y) >> TC_XY_SHIFT);
  }
  p1x += (TC_XY_ONE >> 1);
  p1y += (TC_XY_ONE >> 1);
  L.ex = (int)((p2x + (TC_XY_ONE >> 1)) >> TC_XY_SHIFT);
  L.ey = (int)((p2y + (TC_XY_ONE >> 1)) >> TC_XY_SHIFT);
  L.xmajor = xmajor;
  L.step = (int)step;
  if (xmajor) {
    L.a = (int)(p1x >> TC_XY_SHIFT);
    L.b = (int)p1y;
  } else {
    L.a = (int)(p1y >> TC_XY_SHIFT);
    L.b = (int)p1x;
  }
  return L;
}

// steps k0..k1 (inclusive) of an outline edge
__device__ inline void r_line2_pixels(const Ras& r, int a, int b, int step, int xmajor, int k0, int k1) {
  int major = a + k0;
  int minor = b + __mul24(k0, step);  // k0 <= 16384 (frame size limit), |step| <= 65536: exact in 24x24 bits
  for (int k = k0; k <= k1; k++) {
    const int mn = minor >> TC_XY_SHIFT;
    r_put(r, xmajor ? major : mn, xmajor ? mn : major);  // selects, not a divergent branch per pixel
    major++;
    minor += step;
  }
}

// The four polygon vertices travel as BY-VALUE scalars.  Any aggregate (array or struct behind a
// reference) lets LLVM fold "select of loads" into "load of selected pointer", which pins the
// aggregate in scratch memory -- and scratch write-back shows up as HBM traffic.
__device__ inline long long sel4(long long a0, long long a1, long long a2, long long a3, int i) {
  long long r = a0;
  r = i == 1 ? a1 : r;
  r = i == 2 ? a2 : r;
  r = i == 3 ? a3 : r;
  return r;
}

// FillConvexPoly(v[4], shift = 16, LINE_8) = 4 outline edges (Line2, above) + the scanline fill below.
// FillConvexPoly's two edge walkers as closed-form pieces.  The scanline loop of drawing.cpp is a
// sequence of "events" (a walker reaches the end row of its polygon edge and picks the next one, with a
// shared budget of npts edges) between which both walkers just add dx per row.  This runs the event
// part literally -- only at event rows -- and records every new piece (start row, x at that row, dx per
// row, walker id); row r of walker w is then xs + (r - y_start) * dx of its latest piece.  At most 4
// pieces (every successful update consumes at least one of the 4 edges).
//   py/px/pdx: LDS tables [4]; returns number of pieces; wmask bit s = walker of piece s;
//   rows [y_first, y_last] are the ones the fill draws (empty when y_last < y_first).
__device__ inline int r_fill_events(int W, int H, long long qx0, long long qx1, long long qx2, long long qx3,
                                    long long qy0, long long qy1, long long qy2, long long qy3, int* py, int* pv,
                                    int& wmask, int& y_first, int& y_last) {
  // Integer-only part: which polygon edge each walker switches to at which row.  pv[s] = idx0 | idx << 2
  // (xs = vx[idx0], xe = vx[idx], end row = ty[idx]); the slope of piece s is computed by r_fill_slope.
  const int npts = 4, shift = TC_XY_SHIFT;
  const int delta = 1 << shift >> 1;
  wmask = 0;
  y_first = 0;
  y_last = -1;
  int imin = 0;
  long long xmin = qx0, xmax = qx0, ymin = qy0, ymax = qy0;
#pragma unroll
  for (int i = 1; i < npts; i++) {
    long long x = sel4(qx0, qx1, qx2, qx3, i), y = sel4(qy0, qy1, qy2, qy3, i);
    if (y < ymin) {
      ymin = y;
      imin = i;
    }
    if (y > ymax) ymax = y;
    if (x > xmax) xmax = x;
    if (x < xmin) xmin = x;
  }
  xmin = (xmin + delta) >> shift;
  xmax = (xmax + delta) >> shift;
  ymin = (ymin + delta) >> shift;
  ymax = (ymax + delta) >> shift;
  if (d_wrap32(xmax) < 0 || d_wrap32(ymax) < 0 || d_wrap32(xmin) >= W || d_wrap32(ymin) >= H) return 0;
  if (ymax > H - 1) ymax = H - 1;
  const int ty0 = d_wrap32((qy0 + delta) >> shift), ty1 = d_wrap32((qy1 + delta) >> shift);
  const int ty2 = d_wrap32((qy2 + delta) >> shift), ty3 = d_wrap32((qy3 + delta) >> shift);
  int y = d_wrap32(ymin);
  int e_idx0 = imin, e_idx1 = imin, e_ye0 = y, e_ye1 = y;
  int edges = npts, np = 0;
  int y_end = (int)ymax + 1;
  for (int guard = 0; guard < 8; guard++) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int ye = i ? e_ye1 : e_ye0;
      if (y >= ye) {
        int idx0 = i ? e_idx1 : e_idx0;
        const int di = i ? npts - 1 : 1;
        int idx = idx0 + di;
        if (idx >= npts) idx -= npts;
        for (; edges-- > 0;) {  // (a straight-line 4-candidate version of this scan was tried: 25 % more instructions)
          int ty = idx == 0 ? ty0 : idx == 1 ? ty1 : idx == 2 ? ty2 : ty3;
          if (ty > y) {
            py[np] = y;
            pv[np] = idx0 | (idx << 2);
            wmask |= i << np;
            np++;
            if (i) {
              e_ye1 = ty;
              e_idx1 = idx;
            } else {
              e_ye0 = ty;
              e_idx0 = idx;
            }
            break;
          }
          idx0 = idx;
          idx += di;
          if (idx >= npts) idx -= npts;
        }
      }
    }
    if (edges < 0) {
      y_end = y;
      break;
    }
    int ynext = e_ye0 < e_ye1 ? e_ye0 : e_ye1;
    if (ynext > (int)ymax) break;
    y = ynext;
  }
  y_first = d_wrap32(ymin) > 0 ? d_wrap32(ymin) : 0;
  y_last = y_end - 1 < (int)ymax ? y_end - 1 : (int)ymax;
  return np;
}

// slope of one recorded piece (drawing.cpp: edge[i].dx = ((xe - xs)*2 + (ty - y)) / (2*(ty - y)), edge[i].x = xs)
__device__ inline void r_fill_slope(long long qx0, long long qx1, long long qx2, long long qx3, long long qy0,
                                    long long qy1, long long qy2, long long qy3, int y_u, int v, long long& xs,
                                    long long& dx) {
  const int idx0 = v & 3, idx = (v >> 2) & 3;
  const int ty = d_wrap32((sel4(qy0, qy1, qy2, qy3, idx) + (TC_XY_ONE >> 1)) >> TC_XY_SHIFT);
  xs = sel4(qx0, qx1, qx2, qx3, idx0);
  const long long xe = sel4(qx0, qx1, qx2, qx3, idx);
  dx = d_sdiv((xe - xs) * 2 + ((long long)ty - y_u), 2 * ((long long)ty - y_u));
}

// one fill row from the piece table
__device__ inline void r_fill_row(const Ras& r, int row, int np, int wmask, const int* py, const long long* px,
                                  const long long* pdx) {
  long long xa = -TC_XY_ONE, xb = -TC_XY_ONE;
  for (int s = 0; s < np; s++) {
    const int ys = py[s];
    if (ys <= row) {
      // (row - ys) < 2^15; a slope that fits 32 bits (every edge that is not within a hair of horizontal) takes one
      // 32 x 32 -> 64-bit multiply instead of the three quarter-rate instructions of the 32 x 64-bit product
      // (the choice is made for the whole wavefront: a per-lane select would evaluate both products)
      const long long sl = pdx[s];
      long long x = px[s];
      if (__ballot(sl != (long long)(int)sl) == 0)
        x += (long long)(row - ys) * (long long)(int)sl;
      else
        x += (long long)(row - ys) * sl;
      if ((wmask >> s) & 1)
        xb = x;
      else
        xa = x;
    }
  }
  long long xl = xa < xb ? xa : xb, xr = xa < xb ? xb : xa;
  int xx1 = d_wrap32((xl + (TC_XY_ONE >> 1)) >> TC_XY_SHIFT);
  int xx2 = d_wrap32((xr + (TC_XY_ONE >> 1)) >> TC_XY_SHIFT);
  const bool ok = xx2 >= 0 && xx1 < r.W;
  if (r.wpr <= 2)  // (wave-uniform) a row of at most 64 pixels: a clamped span touches at most its two words
    r_hline_2w(r, ok ? row : -1, xx1, xx2);
  else
    r_hline(r, ok ? row : -1, xx1, xx2);
}

// Circle(center, radius, fill)
__device__ inline void r_circle_fill(const Ras& r, int cx, int cy, int radius) {
  int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
  const int W = r.W, H = r.H;
  bool inside = cx >= radius && cx < W - radius && cy >= radius && cy < H - radius;
  while (dx >= dy) {
    long long y11 = (long long)cy - dy, y12 = (long long)cy + dy, y21 = (long long)cy - dx, y22 = (long long)cy + dx;
    long long x11 = (long long)cx - dx, x12 = (long long)cx + dx, x21 = (long long)cx - dy, x22 = (long long)cx + dy;
    if (inside) {
      r_hline(r, (int)y11, (int)x11, (int)x12);
      r_hline(r, (int)y12, (int)x11, (int)x12);
      r_hline(r, (int)y21, (int)x21, (int)x22);
      r_hline(r, (int)y22, (int)x21, (int)x22);
    } else if (x11 < W && x12 >= 0 && y21 < H && y22 >= 0) {
      if (x11 < 0) x11 = 0;
      if (x12 > W - 1) x12 = W - 1;
      if (y11 >= 0 && y11 < H) r_hline(r, (int)y11, (int)x11, (int)x12);
      if (y12 >= 0 && y12 < H) r_hline(r, (int)y12, (int)x11, (int)x12);
      if (x21 < W && x22 >= 0) {
        if (x21 < 0) x21 = 0;
        if (x22 > W - 1) x22 = W - 1;
        if (y21 >= 0 && y21 < H) r_hline(r, (int)y21, (int)x21, (int)x22);
        if (y22 >= 0 && y22 < H) r_hline(r, (int)y22, (int)x21, (int)x22);
      }
    }
    dy++;
    err += plus;
    plus += 2;
    int mask = (err <= 0) - 1;
    err -= minus & mask;
    dx += mask;
    minus -= mask & 2;
  }
}

// ThickLine's quad for thickness > 1: returns false when the segment is degenerate (|r| <= DBL_EPSILON)
__device__ inline bool r_quad(int x0, int y0, int x1, int y1, int thickness, long long& qx0, long long& qx1,
                              long long& qx2, long long& qx3, long long& qy0, long long& qy1, long long& qy2,
                              long long& qy3) {
  long long p0x = (long long)x0 * TC_XY_ONE, p0y = (long long)y0 * TC_XY_ONE;
  long long p1x = (long long)x1 * TC_XY_ONE, p1y = (long long)y1 * TC_XY_ONE;
  const double INV_XY_ONE = 1. / TC_XY_ONE;
  double dx = (double)(p0x - p1x) * INV_XY_ONE, dy = (double)(p1y - p0y) * INV_XY_ONE;
  double rr = dx * dx + dy * dy;
  int odd = thickness & 1;
  long long th = (long long)thickness << (TC_XY_SHIFT - 1);
  if (!(tc_fabs(rr) > 2.2204460492503131e-16)) return false;
  rr = ((double)th + odd * TC_XY_ONE * 0.5) / sqrt(rr);
  long long dpx = __double2int_rn(dy * rr), dpy = __double2int_rn(dx * rr);
  qx0 = p0x + dpx; qx1 = p0x - dpx; qx2 = p1x - dpx; qx3 = p1x + dpx;
  qy0 = p0y + dpy; qy1 = p0y - dpy; qy2 = p1y - dpy; qy3 = p1y + dpy;
  return true;
}

#endif  // TC_DEVICE_H
