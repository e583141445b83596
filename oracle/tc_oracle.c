/* tc_oracle.c -- CPU ORACLE (test infrastructure, see tc_oracle.h).  Plain C11, double precision.
 * Build: see oracle/Makefile (-O2 -ffp-contract=off -fno-builtin-pow -fopenmp).
 * Every function cites the reference lines (relative to the reference repo root) it restates. */
#include "tc_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../tinycarlo_amd/csrc/tc_trig.h" /* ORC_MATH_PORTABLE only */
#include "../tinycarlo_amd/csrc/tc_rng.h"  /* ORC_F_DEVICE_SPAWN only (no reference counterpart) */

#ifdef _OPENMP
#include <omp.h>
#endif

#define PI 3.141592653589793 /* == math.pi */

static int g_mode = ORC_MATH_LIBM;
void orc_set_math_mode(int mode) { g_mode = mode; }
int orc_get_math_mode(void) { return g_mode; }

/* ---- math.* as CPython calls them (libm) or the portable kernels the GPU runs ---- */
static double m_sin(double x) { return g_mode ? tc_sin(x) : sin(x); }
static double m_cos(double x) { return g_mode ? tc_cos(x) : cos(x); }
static double m_tan(double x) { return g_mode ? tc_tan(x) : tan(x); }
static double m_atan2(double y, double x) { return g_mode ? tc_atan2(y, x) : atan2(y, x); }
/* python `x**2` is libm pow(x, 2.0) (not always == x*x) */
static double m_sq(double x) { return g_mode ? x * x : pow(x, 2.0); }
static double m_radians(double d) { return d * (PI / 180.0); } /* math.radians: x * (pi/180) */

double orc_trig(int fn, double a, double b, int mode) {
  int save = g_mode;
  double r;
  g_mode = mode;
  switch (fn) {
    case 0: r = m_sin(a); break;
    case 1: r = m_cos(a); break;
    case 2: r = m_tan(a); break;
    default: r = m_atan2(a, b); break;
  }
  g_mode = save;
  return r;
}

/* helper.py:11-19.  The reference loops forever on +-inf; bounded here (angles on this path stay
 * within a few multiples of pi). */
double orc_clip_angle(double a) {
  int guard = 0;
  while (a > PI && guard++ < 64) a -= 2 * PI;
  while (a < -PI && guard++ < 128) a += 2 * PI;
  return a;
}

/* numpy clip ufunc (np.clip at car.py:82,88, env.py:118): NaN in x propagates */
static double np_clip(double x, double lo, double hi) {
  double m = isnan(x) ? x : (x > lo ? x : lo);
  return isnan(m) ? m : (m < hi ? m : hi);
}

/* layer.py:187 */
static double dist(double ax, double ay, double bx, double by) { return sqrt(m_sq(ax - bx) + m_sq(ay - by)); }

/* ============================================================== Layer queries (layer.py) */

/* layer.py:33-44: argmin_e |d(p,n0)+d(p,n1)|, list.index(min) -> first minimum */
int orc_layer_nearest_edge(const double* nodes, const int32_t* edges, int n_edges, double px, double py) {
  int best = -1;
  double bd = 0;
  for (int e = 0; e < n_edges; e++) {
    const double* a = nodes + 2 * edges[2 * e];
    const double* b = nodes + 2 * edges[2 * e + 1];
    double d = fabs(dist(px, py, a[0], a[1]) + dist(px, py, b[0], b[1]));
    if (best < 0 || d < bd) {
      best = e;
      bd = d;
    }
  }
  return best;
}

/* layer.py:46-57 */
int orc_layer_nearest_node(const double* nodes, int n_nodes, double px, double py) {
  int best = -1;
  double bd = 0;
  for (int i = 0; i < n_nodes; i++) {
    double d = dist(px, py, nodes[2 * i], nodes[2 * i + 1]);
    if (best < 0 || d < bd) {
      best = i;
      bd = d;
    }
  }
  return best;
}

/* layer.py:179-181 */
static double edge_orientation(const double* nodes, int a, int b) {
  return m_atan2(nodes[2 * b + 1] - nodes[2 * a + 1], nodes[2 * b] - nodes[2 * a]);
}

/* layer.py:59-74 (orientation table optional) */
static int nearest_edge_with_orientation(const double* nodes, const int32_t* edges, const double* ori_tab, int n_edges,
                                         double px, double py, double orientation, double margin_deg) {
  int best = -1;
  double bd = 0;
  double lim = m_radians(margin_deg);
  for (int e = 0; e < n_edges; e++) {
    int a = edges[2 * e], b = edges[2 * e + 1];
    double ori = ori_tab ? ori_tab[e] : edge_orientation(nodes, a, b);
    if (!(fabs(orc_clip_angle(ori - orientation)) <= lim)) continue;
    double d = fabs(dist(px, py, nodes[2 * a], nodes[2 * a + 1]) + dist(px, py, nodes[2 * b], nodes[2 * b + 1]));
    if (best < 0 || d < bd) {
      best = e;
      bd = d;
    }
  }
  return best;
}
int orc_layer_nearest_edge_with_orientation(const double* nodes, const int32_t* edges, int n_edges, double px,
                                            double py, double orientation, double margin_deg) {
  return nearest_edge_with_orientation(nodes, edges, NULL, n_edges, px, py, orientation, margin_deg);
}

/* layer.py:126-142.  ori_fwd = angle(*edge_vector), ori_rev = angle(-ev[0], -ev[1]) */
static int within_bounds(double n0x, double n0y, double n1x, double n1y, double ori_fwd, double ori_rev, double px,
                         double py) {
  if (px == n0x && py == n0y) return 1;
  if (px == n1x && py == n1y) return 1;
  double a0 = fabs(orc_clip_angle(m_atan2(py - n0y, px - n0x) - ori_fwd));
  double a1 = fabs(orc_clip_angle(m_atan2(py - n1y, px - n1x) - ori_rev));
  return a0 <= PI / 2 && a1 <= PI / 2;
}
int orc_layer_within_bounds(const double* nodes, const int32_t* edge, double px, double py) {
  const double* a = nodes + 2 * edge[0];
  const double* b = nodes + 2 * edge[1];
  double evx = b[0] - a[0], evy = b[1] - a[1];
  return within_bounds(a[0], a[1], b[0], b[1], m_atan2(evy, evx), m_atan2(-evy, -evx), px, py);
}

/* layer.py:144-164 */
static double distance_to_edge(double n1x, double n1y, double n2x, double n2y, double px, double py) {
  double lvx = n2x - n1x, lvy = n2y - n1y;
  double pvx = px - n1x, pvy = py - n1y;
  if (lvx == 0) {
    if (lvy > 0) return px - n1x;
    return n1x - px;
  }
  return (pvx * lvy - pvy * lvx) / sqrt(m_sq(lvx) + m_sq(lvy));
}
double orc_layer_distance_to_edge(const double* nodes, const int32_t* edge, double px, double py) {
  const double* a = nodes + 2 * edge[0];
  const double* b = nodes + 2 * edge[1];
  return distance_to_edge(a[0], a[1], b[0], b[1], px, py);
}

/* ============================================================== Map (map.py) */
struct orc_map {
  int C;
  int node_off[ORC_MAXC + 1], edge_off[ORC_MAXC + 1];
  double* nodes;   /* [sum nodes][2] */
  int32_t* edges;  /* [sum edges][2], layer-local node ids */
  double* ori_fwd; /* per lane-line edge: atan2(evy, evx)      (static part of layer.py:140) */
  double* ori_rev; /* per lane-line edge: atan2(-evy, -evx)    (static part of layer.py:141) */
  uint8_t colors[ORC_MAXC][3];
  int lpN, lpE;
  double* lp_nodes;
  int32_t* lp_edges;
  double* lp_ori; /* per lanepath edge (layer.py:179-181) */
  /* get_next_nodes / get_prev_nodes (layer.py:183-185) as CSR in edge-list order */
  int32_t *next_off, *next_node, *prev_off, *prev_node;
  double *next_ori, *prev_ori; /* atan2(nodes[nn]-nodes[n]) of layer.py:122 */
};

static double host_atan2(double y, double x) { return atan2(y, x); }

orc_map* orc_map_create(int32_t n_layers, const int32_t* node_count, const int32_t* edge_count, const double* nodes,
                        const int32_t* edges, const uint8_t* colors, int32_t lpN, int32_t lpE, const double* lp_nodes,
                        const int32_t* lp_edges) {
  if (n_layers < 0 || n_layers > ORC_MAXC) return NULL;
  orc_map* m = (orc_map*)calloc(1, sizeof(orc_map));
  m->C = n_layers;
  for (int l = 0; l < n_layers; l++) {
    m->node_off[l + 1] = m->node_off[l] + node_count[l];
    m->edge_off[l + 1] = m->edge_off[l] + edge_count[l];
    memcpy(m->colors[l], colors + 3 * l, 3);
  }
  int tn = m->node_off[n_layers], te = m->edge_off[n_layers];
  m->nodes = (double*)malloc(sizeof(double) * 2 * (tn + 1));
  m->edges = (int32_t*)malloc(sizeof(int32_t) * 2 * (te + 1));
  m->ori_fwd = (double*)malloc(sizeof(double) * (te + 1));
  m->ori_rev = (double*)malloc(sizeof(double) * (te + 1));
  memcpy(m->nodes, nodes, sizeof(double) * 2 * tn);
  memcpy(m->edges, edges, sizeof(int32_t) * 2 * te);
  /* static orientation tables are always host-libm values: that is what the reference evaluates
   * at run time (math.atan2 of constant inputs), and what tc_map_create() uploads to the GPU. */
  for (int l = 0; l < n_layers; l++)
    for (int e = m->edge_off[l]; e < m->edge_off[l + 1]; e++) {
      const double* a = m->nodes + 2 * (m->node_off[l] + m->edges[2 * e]);
      const double* b = m->nodes + 2 * (m->node_off[l] + m->edges[2 * e + 1]);
      double evx = b[0] - a[0], evy = b[1] - a[1];
      m->ori_fwd[e] = host_atan2(evy, evx);
      m->ori_rev[e] = host_atan2(-evy, -evx);
    }
  m->lpN = lpN;
  m->lpE = lpE;
  m->lp_nodes = (double*)malloc(sizeof(double) * 2 * (lpN + 1));
  m->lp_edges = (int32_t*)malloc(sizeof(int32_t) * 2 * (lpE + 1));
  m->lp_ori = (double*)malloc(sizeof(double) * (lpE + 1));
  memcpy(m->lp_nodes, lp_nodes, sizeof(double) * 2 * lpN);
  memcpy(m->lp_edges, lp_edges, sizeof(int32_t) * 2 * lpE);
  m->next_off = (int32_t*)calloc(lpN + 2, sizeof(int32_t));
  m->prev_off = (int32_t*)calloc(lpN + 2, sizeof(int32_t));
  m->next_node = (int32_t*)malloc(sizeof(int32_t) * (lpE + 1));
  m->prev_node = (int32_t*)malloc(sizeof(int32_t) * (lpE + 1));
  m->next_ori = (double*)malloc(sizeof(double) * (lpE + 1));
  m->prev_ori = (double*)malloc(sizeof(double) * (lpE + 1));
  for (int e = 0; e < lpE; e++) {
    int a = lp_edges[2 * e], b = lp_edges[2 * e + 1];
    m->lp_ori[e] = host_atan2(lp_nodes[2 * b + 1] - lp_nodes[2 * a + 1], lp_nodes[2 * b] - lp_nodes[2 * a]);
    m->next_off[a + 1]++;
    m->prev_off[b + 1]++;
  }
  for (int i = 0; i < lpN; i++) {
    m->next_off[i + 1] += m->next_off[i];
    m->prev_off[i + 1] += m->prev_off[i];
  }
  int32_t* nf = (int32_t*)calloc(lpN + 1, sizeof(int32_t));
  int32_t* pf = (int32_t*)calloc(lpN + 1, sizeof(int32_t));
  for (int e = 0; e < lpE; e++) { /* stable: keeps edge-list order inside each bucket */
    int a = lp_edges[2 * e], b = lp_edges[2 * e + 1];
    int s = m->next_off[a] + nf[a]++;
    m->next_node[s] = b;
    m->next_ori[s] = host_atan2(lp_nodes[2 * b + 1] - lp_nodes[2 * a + 1], lp_nodes[2 * b] - lp_nodes[2 * a]);
    int p = m->prev_off[b] + pf[b]++;
    m->prev_node[p] = a;
    m->prev_ori[p] = host_atan2(lp_nodes[2 * a + 1] - lp_nodes[2 * b + 1], lp_nodes[2 * a] - lp_nodes[2 * b]);
  }
  free(nf);
  free(pf);
  return m;
}

void orc_map_free(orc_map* m) {
  if (!m) return;
  free(m->nodes); free(m->edges); free(m->ori_fwd); free(m->ori_rev);
  free(m->lp_nodes); free(m->lp_edges); free(m->lp_ori);
  free(m->next_off); free(m->prev_off); free(m->next_node); free(m->prev_node);
  free(m->next_ori); free(m->prev_ori);
  free(m);
}

int orc_map_has_next(const orc_map* m, int node) {
  if (node < 0 || node >= m->lpN) return 0;
  return m->next_off[node + 1] > m->next_off[node];
}

/* orientation of lanepath edge (a,b) (layer.py:179-181) = the table entry of the first a->b edge */
static double lp_edge_ori(const orc_map* m, int a, int b) {
  for (int s = m->next_off[a]; s < m->next_off[a + 1]; s++)
    if (m->next_node[s] == b) return m->next_ori[s];
  return m_atan2(m->lp_nodes[2 * b + 1] - m->lp_nodes[2 * a + 1], m->lp_nodes[2 * b] - m->lp_nodes[2 * a]);
}

/* layer.py:105-124.  list = CSR slice [s0,s1) of (node, ori).  Returns node id or -1 (None).
 * Quirk kept: self-loops are dropped from the angle list but the argmin index is applied to the
 * UNFILTERED list (layer.py:122-124). */
static int pick_node(const int32_t* lst, const double* ori, int s0, int s1, int node_idx, double orientation,
                     int* status) {
  int n = s1 - s0;
  if (n == 0) return -1;
  if (n <= 1) return lst[s0];
  int best = -1, k = 0;
  double bk = 0;
  for (int s = s0; s < s1; s++) {
    if (lst[s] == node_idx) continue;
    double key = fabs(orc_clip_angle(ori[s] - orientation));
    if (best < 0 || key < bk) {
      best = k;
      bk = key;
    }
    k++;
  }
  if (best < 0) { /* reference: min() of an empty range -> ValueError */
    *status |= ORC_S_PICK_EMPTY;
    return -1;
  }
  return lst[s0 + best];
}

/* ============================================================== Car (car.py) */

static void update_front(const orc_car* c, orc_state* s) { /* car.py:167-168 */
  s->front_x = s->x + c->wheelbase * m_cos(s->theta);
  s->front_y = s->y + c->wheelbase * m_sin(s->theta);
}

/* car.py:34-44, map.py:51-69 (node index already drawn and known to have an out-edge) */
void orc_reset(const orc_map* m, const orc_car* c, orc_state* s, int node) {
  int nn = m->next_node[m->next_off[node]];
  s->x = m->lp_nodes[2 * node];
  s->y = m->lp_nodes[2 * node + 1];
  s->theta = m->next_ori[m->next_off[node]]; /* atan2(next-pos), map.py:68 */
  for (int i = 0; i < 8; i++) s->lp[i] = -1;
  s->lp[0] = node;
  s->lp[1] = nn;
  s->lp_len = 1;
  update_front(c, s);
  s->steering = 0.0;
  s->radius = 0.0;
  s->velocity = 0.0;
  s->last_maneuver = 0;
}

/* car.py:127-148 */
static int find_local_path(const orc_map* m, orc_state* s, int maneuver, int* status) {
  const double* N = m->lp_nodes;
  double fx = s->front_x, fy = s->front_y;
  int e0 = s->lp[0], e1 = s->lp[1];
  double mdir = orc_clip_angle(lp_edge_ori(m, e0, e1) + (maneuver * PI) / 2);
  int ne0, ne1;
  if (maneuver == 2 && s->last_maneuver != 2) {
    int e = nearest_edge_with_orientation(N, m->lp_edges, m->lp_ori, m->lpE, fx, fy, mdir, 30.0);
    mdir = orc_clip_angle(mdir + PI);
    if (e < 0) { /* reference: local_path=[None] then TypeError at car.py:143 */
      *status |= ORC_S_UTURN_NO_EDGE;
      return 1;
    }
    ne0 = m->lp_edges[2 * e];
    ne1 = m->lp_edges[2 * e + 1];
  } else {
    /* layer.py:77-103 */
    int nx = pick_node(m->next_node, m->next_ori, m->next_off[e1], m->next_off[e1 + 1], e1, mdir, status);
    int pv = pick_node(m->prev_node, m->prev_ori, m->prev_off[e0], m->prev_off[e0 + 1], e0, mdir, status);
    if (nx < 0 || pv < 0) return 1;
    double d0 = dist(fx, fy, N[2 * e0], N[2 * e0 + 1]), d1 = dist(fx, fy, N[2 * e1], N[2 * e1 + 1]);
    double dn = dist(fx, fy, N[2 * nx], N[2 * nx + 1]), dp = dist(fx, fy, N[2 * pv], N[2 * pv + 1]);
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
  s->last_maneuver = maneuver;
  s->lp[0] = ne0;
  s->lp[1] = ne1;
  s->lp_len = 1;
  for (int i = 0; i < 3; i++) {
    int node = s->velocity > 0 ? s->lp[2 * i + 1] : s->lp[2 * i]; /* car.py:143 */
    int nn = pick_node(m->next_node, m->next_ori, m->next_off[node], m->next_off[node + 1], node, mdir, status);
    if (nn < 0) return 1;
    s->lp[2 * (i + 1)] = node;
    s->lp[2 * (i + 1) + 1] = nn;
    s->lp_len = i + 2;
  }
  return 0;
}

/* car.py:70-125 */
int orc_car_step(const orc_map* m, const orc_car* c, orc_state* s, double v_in, double s_in, int maneuver,
                 int* status) {
  double dt = c->T;
  double nv = v_in * c->max_velocity;
  if (c->has_max_acceleration) nv = np_clip(nv, s->velocity - c->max_deceleration * dt, s->velocity + c->max_acceleration * dt);
  s->velocity = nv;
  double ns = s_in * c->max_steering_angle;
  if (c->has_steering_speed) ns = np_clip(ns, s->steering - c->steering_speed * dt, s->steering + c->steering_speed * dt);
  s->steering = ns;
  double vxn = m_cos(s->theta), vyn = m_sin(s->theta);
  if (fabs(s->steering) < 0.0001) {
    s->radius = 0;
    s->x = s->x + s->velocity * vxn * dt;
    s->y = s->y + s->velocity * vyn * dt;
  } else {
    s->radius = c->wheelbase / m_tan(m_radians(s->steering));
    double ang_vel = s->velocity / s->radius;
    double dyaw = ang_vel * dt;
    double nx = vyn, ny = -vxn;
    double tx = nx * s->radius, ty = ny * s->radius;
    double cd = m_cos(dyaw), sd = m_sin(dyaw);
    /* R_M.dot([tx,ty]) car.py:111-113: a 2x2 matrix times a vector is cblas_dgemv, whose x86-64 FMA kernel forms
     * row i as fma(R[i][0], tx, R[i][1] * ty) (tools/numpy_matmul_probe.py: 20 000 of 20 000 random cases, the unfused
     * form matches 67 %) */
    double r0 = __builtin_fma(cd, tx, (-sd) * ty);
    double r1 = __builtin_fma(sd, tx, cd * ty);
    s->x = s->x - tx + r0;
    s->y = s->y - ty + r1;
    s->theta += dyaw;
    if (s->theta > PI)
      s->theta -= 2 * PI;
    else if (s->theta < -PI)
      s->theta += 2 * PI;
  }
  update_front(c, s);
  return find_local_path(m, s, maneuver, status);
}

/* car.py:46-68 + env.py:84-99,136-138 */
void orc_get_info(const orc_map* m, const orc_car* c, const orc_state* s, uint32_t flags, orc_info* o) {
  int status_keep = o->status, trunc_keep = o->truncated;
  memset(o, 0, sizeof(*o));
  o->status = status_keep;
  o->truncated = trunc_keep;
  for (int l = 0; l < ORC_MAXC; l++) o->nearest_edge[l] = -1;
  if (s->lp_len >= 2) {
    const double* N = m->lp_nodes;
    int a = s->lp[2], b = s->lp[3];
    o->cte = distance_to_edge(N[2 * a], N[2 * a + 1], N[2 * b], N[2 * b + 1], s->front_x, s->front_y);
    o->heading_error = orc_clip_angle(lp_edge_ori(m, a, b) - s->theta);
    for (int l = 0; l < m->C; l++) {
      const double* LN = m->nodes + 2 * m->node_off[l];
      const int32_t* LE = m->edges + 2 * m->edge_off[l];
      int ne = m->edge_off[l + 1] - m->edge_off[l];
      int e = orc_layer_nearest_edge(LN, LE, ne, s->x, s->y);
      o->nearest_edge[l] = e;
      if (e < 0) { /* empty layer: reference raises on min([]) */
        o->dist[l] = 0;
        continue;
      }
      const double* n0 = LN + 2 * LE[2 * e];
      const double* n1 = LN + 2 * LE[2 * e + 1];
      int ge = m->edge_off[l] + e;
      if (within_bounds(n0[0], n0[1], n1[0], n1[1], m->ori_fwd[ge], m->ori_rev[ge], s->x, s->y)) {
        o->dist[l] = fabs(distance_to_edge(n0[0], n0[1], n1[0], n1[1], s->x, s->y));
      } else {
        double da = dist(s->x, s->y, n0[0], n0[1]);
        double db = dist(s->front_x, s->front_y, n1[0], n1[1]); /* FRONT for n1: car.py:64 */
        o->dist[l] = db < da ? db : da;
      }
    }
    o->n_lp_coords = s->lp_len;
    for (int i = 0; i < s->lp_len; i++) {
      o->lp_coords[2 * i] = N[2 * s->lp[2 * i + 1]];
      o->lp_coords[2 * i + 1] = N[2 * s->lp[2 * i + 1] + 1];
    }
    o->velocity = s->velocity;
  }
  if (!(flags & ORC_F_WRAPPED)) {
    double r = (-1 / c->track_width) * o->cte + 1; /* env.py:93 */
    o->reward = (0 > r) ? 0 : r;                   /* python max(r, 0) */
    o->terminated = o->cte > (c->track_width * 10); /* env.py:99 */
  }
}

/* ============================================================== Camera (camera.py) */

/* numpy's `A @ B` on float64 matrices (camera.py:62,131,138; car.py:165) is cblas_dgemm of the OpenBLAS bundled with
 * numpy.  On every x86-64 CPU with FMA its micro-kernels accumulate C[i][j] as ONE chain of fused multiply-adds over
 * ascending k, starting from a zero accumulator -- isolated in round 2 by replaying the reference's matrices with exact
 * rational arithmetic (tools/numpy_matmul_probe.py: R@T, E@car3d, pose@points and K@P all reproduce bit for bit with
 * this form and with no other tried: unfused, reversed, two accumulators).  A one-ulp difference here is what moved the
 * np.int32 of far off-screen end points (|u| ~ 1e8) by one in 5 of 4 520 recorded frames. */
static void matmul(const double* A, const double* B, double* C, int n, int k, int p) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < p; j++) {
      double acc = 0.0;
      for (int t = 0; t < k; t++) acc = __builtin_fma(A[i * k + t], B[t * p + j], acc);
      C[i * p + j] = acc;
    }
}

/* camera.py:112-122 */
static void point_on_line_at_z(const double* p0, const double* p1, double tz, double* out) {
  double d0 = p0[0] - p1[0], d1 = p0[1] - p1[1], d2 = p0[2] - p1[2];
  if (d2 == 0) { /* returns None -> the row assignment stores NaN */
    out[0] = out[1] = out[2] = NAN;
    return;
  }
  double t = (tz - p1[2]) / d2;
  double a = p1[0] + t * d0, b = p1[1] + t * d1, c = p1[2] + t * d2;
  out[0] = a;
  out[1] = b;
  out[2] = c;
}

/* np.int32(float64) on x86-64 (cvttsd2si): out-of-range / NaN -> INT_MIN */
static int32_t np_int32(double v) {
  if (!(v > -2147483649.0 && v < 2147483648.0)) return INT_MIN;
  return (int32_t)v;
}

/* The four fix-up loops of camera.py:70-86, literally: list first, then mutate in list order. */
static void fixup_pass(double* P, const int32_t* LE, int ne, uint8_t* flag, int target_is_e0, double tz, int32_t* list) {
  int n = 0;
  for (int e = 0; e < ne; e++) {
    int a = LE[2 * e], b = LE[2 * e + 1];
    int sel = target_is_e0 ? (!flag[a] && flag[b]) : (flag[a] && !flag[b]);
    if (sel) list[n++] = e;
  }
  for (int i = 0; i < n; i++) {
    int a = LE[2 * list[i]], b = LE[2 * list[i] + 1];
    double out[3];
    if (target_is_e0) {
      point_on_line_at_z(P + 3 * b, P + 3 * a, tz, out);
      memcpy(P + 3 * a, out, sizeof(out));
      flag[a] = 1;
    } else {
      point_on_line_at_z(P + 3 * a, P + 3 * b, tz, out);
      memcpy(P + 3 * b, out, sizeof(out));
      flag[b] = 1;
    }
  }
}

int orc_capture_segments(const orc_map* m, const orc_cam* cam, const orc_state* s, int32_t* seg_i, double* seg_f,
                         int max) {
  /* car.py:159-165 */
  double cth = m_cos(-s->theta), sth = m_sin(-s->theta);
  double R[16] = {cth, -sth, 0, 0, sth, cth, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  double Tm[16] = {1, 0, 0, -s->x, 0, 1, 0, -s->y, 0, 0, 1, 0, 0, 0, 0, 1};
  double car3d[16], pose[12];
  matmul(R, Tm, car3d, 4, 4, 4);
  matmul(cam->E, car3d, pose, 3, 4, 4); /* camera.py:62 */
  int count = 0;
  int maxn = 0, maxe = 0;
  for (int l = 0; l < m->C; l++) {
    int nn = m->node_off[l + 1] - m->node_off[l], ne = m->edge_off[l + 1] - m->edge_off[l];
    if (nn > maxn) maxn = nn;
    if (ne > maxe) maxe = ne;
  }
  double* P = (double*)malloc(sizeof(double) * 3 * (maxn + 1));
  double* pp = (double*)malloc(sizeof(double) * 2 * (maxn + 1));
  uint8_t* front = (uint8_t*)malloc(maxn + 1);
  uint8_t* inrange = (uint8_t*)malloc(maxn + 1);
  uint8_t* vis = (uint8_t*)malloc(maxn + 1);
  int32_t* list = (int32_t*)malloc(sizeof(int32_t) * (maxe + 1));
  for (int l = 0; l < m->C; l++) {
    const double* LN = m->nodes + 2 * m->node_off[l];
    const int32_t* LE = m->edges + 2 * m->edge_off[l];
    int nn = m->node_off[l + 1] - m->node_off[l], ne = m->edge_off[l + 1] - m->edge_off[l];
    /* camera.py:124-131: (pose @ [x,y,0,1]^T) */
    for (int i = 0; i < nn; i++) {
      double h[4] = {LN[2 * i], LN[2 * i + 1], 0.0, 1.0};
      matmul(pose, h, P + 3 * i, 3, 4, 1);
    }
    for (int i = 0; i < nn; i++) front[i] = P[3 * i + 2] < 0; /* camera.py:70 */
    fixup_pass(P, LE, ne, front, 1, -0.0000001, list);        /* camera.py:71-74 */
    fixup_pass(P, LE, ne, front, 0, -0.0000001, list);        /* camera.py:75-77 */
    for (int i = 0; i < nn; i++) inrange[i] = P[3 * i + 2] > -cam->max_range; /* camera.py:80 (mutated z) */
    fixup_pass(P, LE, ne, inrange, 1, -cam->max_range, list); /* camera.py:81-83 */
    fixup_pass(P, LE, ne, inrange, 0, -cam->max_range, list); /* camera.py:84-86 */
    /* camera.py:133-142 */
    for (int i = 0; i < nn; i++) {
      double h[3];
      matmul(cam->K, P + 3 * i, h, 3, 3, 1);
      pp[2 * i] = h[0] / h[2];
      pp[2 * i + 1] = h[1] / h[2];
      vis[i] = (pp[2 * i] > 0) && (pp[2 * i] < cam->W) && (pp[2 * i + 1] > 0) && (pp[2 * i + 1] < cam->H) &&
               front[i] && inrange[i]; /* camera.py:90-93 */
    }
    for (int e = 0; e < ne; e++) { /* camera.py:95 */
      int a = LE[2 * e], b = LE[2 * e + 1];
      if (!(vis[a] || vis[b])) continue;
      if (count < max) {
        if (seg_i) {
          int32_t* o = seg_i + 5 * count;
          o[0] = l;
          o[1] = np_int32(pp[2 * a]);
          o[2] = np_int32(pp[2 * a + 1]);
          o[3] = np_int32(pp[2 * b]);
          o[4] = np_int32(pp[2 * b + 1]);
        }
        if (seg_f) {
          double* o = seg_f + 4 * count;
          o[0] = pp[2 * a];
          o[1] = pp[2 * a + 1];
          o[2] = pp[2 * b];
          o[3] = pp[2 * b + 1];
        }
      }
      count++;
    }
  }
  free(P); free(pp); free(front); free(inrange); free(vis); free(list);
  return count;
}

/* ============================================================== cv2.polylines restated (OpenCV 4.x drawing.cpp)
 * UNPINNED: OpenCV is not available in the build container; see header comment. */
#define XY_SHIFT 16
#define XY_ONE (1 << XY_SHIFT)

typedef struct {
  uint8_t* data;
  int W, H, ch;
  uint8_t color[3];
} img_t;

static int32_t wrap32(int64_t v) { return (int32_t)(uint32_t)(uint64_t)v; } /* (int) cast of an int64 */

static void put_px(img_t* im, int x, int y) {
  if (x < 0 || x >= im->W || y < 0 || y >= im->H) return;
  uint8_t* p = im->data + ((size_t)y * im->W + x) * im->ch;
  for (int c = 0; c < im->ch; c++) p[c] = im->color[c];
}
/* ICV_HLINE: inclusive [xl, xr] on row y (caller guarantees the row and the clamp) */
static void hline(img_t* im, int y, int xl, int xr) {
  for (int x = xl; x <= xr; x++) put_px(im, x, y);
}

/* clipLine(Size2l, Point2l&, Point2l&) */
static int clip_line(int64_t width, int64_t height, int64_t* x1, int64_t* y1, int64_t* x2, int64_t* y2) {
  int c1, c2;
  int64_t right = width - 1, bottom = height - 1;
  if (width <= 0 || height <= 0) return 0;
  c1 = (*x1 < 0) + (*x1 > right) * 2 + (*y1 < 0) * 4 + (*y1 > bottom) * 8;
  c2 = (*x2 < 0) + (*x2 > right) * 2 + (*y2 < 0) * 4 + (*y2 > bottom) * 8;
  if ((c1 & c2) == 0 && (c1 | c2) != 0) {
    int64_t a;
    if (c1 & 12) {
      a = c1 < 8 ? 0 : bottom;
      *x1 += (int64_t)((double)(a - *y1) * (double)(*x2 - *x1) / (double)(*y2 - *y1));
      *y1 = a;
      c1 = (*x1 < 0) + (*x1 > right) * 2;
    }
    if (c2 & 12) {
      a = c2 < 8 ? 0 : bottom;
      *x2 += (int64_t)((double)(a - *y2) * (double)(*x2 - *x1) / (double)(*y2 - *y1));
      *y2 = a;
      c2 = (*x2 < 0) + (*x2 > right) * 2;
    }
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
      if (c1) {
        a = c1 == 1 ? 0 : right;
        *y1 += (int64_t)((double)(a - *x1) * (double)(*y2 - *y1) / (double)(*x2 - *x1));
        *x1 = a;
        c1 = 0;
      }
      if (c2) {
        a = c2 == 1 ? 0 : right;
        *y2 += (int64_t)((double)(a - *x2) * (double)(*y2 - *y1) / (double)(*x2 - *x1));
        *x2 = a;
        c2 = 0;
      }
    }
  }
  return (c1 | c2) == 0;
}

/* Line(): LineIterator(img, pt1, pt2, 8, leftToRight=true), thickness<=1 path */
static void line_bresenham(img_t* im, int64_t x1, int64_t y1, int64_t x2, int64_t y2) {
  if ((uint64_t)x1 >= (uint64_t)im->W || (uint64_t)x2 >= (uint64_t)im->W || (uint64_t)y1 >= (uint64_t)im->H ||
      (uint64_t)y2 >= (uint64_t)im->H) {
    if (!clip_line(im->W, im->H, &x1, &y1, &x2, &y2)) return;
  }
  int64_t dx = x2 - x1, dy = y2 - y1;
  int sx = 1, sy = 1;
  if (dx < 0) { /* leftToRight */
    dx = -dx;
    dy = -dy;
    x1 = x2;
    y1 = y2;
  }
  if (dy < 0) {
    dy = -dy;
    sy = -1;
  }
  int vert = dy > dx;
  if (vert) {
    int64_t t = dx;
    dx = dy;
    dy = t;
  }
  int64_t err = dx - (dy + dy), plus = dx + dx, minus = -(dy + dy);
  int64_t count = dx + 1;
  int64_t x = x1, y = y1;
  for (int64_t i = 0; i < count; i++) {
    put_px(im, (int)x, (int)y);
    int mask = err < 0;
    err += minus + (mask ? plus : 0);
    if (vert) {
      y += sy;
      if (mask) x += sx;
    } else {
      x += sx;
      if (mask) y += sy;
    }
  }
}

/* Line2(): fixed-point DDA used for the polygon outline when shift != 0 */
static void line2(img_t* im, int64_t p1x, int64_t p1y, int64_t p2x, int64_t p2y) {
  if (!clip_line((int64_t)im->W << XY_SHIFT, (int64_t)im->H << XY_SHIFT, &p1x, &p1y, &p2x, &p2y)) return;
  int64_t dx = p2x - p1x, dy = p2y - p1y;
  int64_t j = dx < 0 ? -1 : 0;
  int64_t ax = (dx ^ j) - j;
  int64_t i = dy < 0 ? -1 : 0;
  int64_t ay = (dy ^ i) - i;
  int64_t x_step, y_step;
  int ecount;
  if (ax > ay) {
    dy = (dy ^ j) - j;
    if (j) { /* the xor-swap of both points */
      int64_t t = p1x; p1x = p2x; p2x = t;
      t = p1y; p1y = p2y; p2y = t;
    }
    x_step = XY_ONE;
    y_step = (dy * XY_ONE) / (ax | 1); /* (dy << XY_SHIFT) / (ax | 1), C truncating division */
    ecount = (int)((p2x - p1x) >> XY_SHIFT);
  } else {
    dx = (dx ^ i) - i;
    if (i) {
      int64_t t = p1x; p1x = p2x; p2x = t;
      t = p1y; p1y = p2y; p2y = t;
    }
    x_step = (dx * XY_ONE) / (ay | 1);
    y_step = XY_ONE;
    ecount = (int)((p2y - p1y) >> XY_SHIFT);
  }
  p1x += (XY_ONE >> 1);
  p1y += (XY_ONE >> 1);
  put_px(im, (int)((p2x + (XY_ONE >> 1)) >> XY_SHIFT), (int)((p2y + (XY_ONE >> 1)) >> XY_SHIFT));
  if (ax > ay) {
    p1x >>= XY_SHIFT;
    while (ecount >= 0) {
      put_px(im, (int)p1x, (int)(p1y >> XY_SHIFT));
      p1x++;
      p1y += y_step;
      ecount--;
    }
  } else {
    p1y >>= XY_SHIFT;
    while (ecount >= 0) {
      put_px(im, (int)(p1x >> XY_SHIFT), (int)p1y);
      p1x += x_step;
      p1y++;
      ecount--;
    }
  }
}

/* FillConvexPoly(img, v[4], 4, color, LINE_8, shift = XY_SHIFT) */
static void fill_convex_poly4(img_t* im, const int64_t* vx, const int64_t* vy) {
  const int npts = 4, shift = XY_SHIFT;
  struct {
    int idx, di;
    int64_t x, dx;
    int ye;
  } edge[2];
  int delta = 1 << shift >> 1;
  int i, y, imin = 0;
  int edges = npts;
  int64_t xmin, xmax, ymin, ymax;
  int delta1 = XY_ONE >> 1, delta2 = XY_ONE >> 1;
  int64_t p0x = vx[npts - 1], p0y = vy[npts - 1];
  xmin = xmax = vx[0];
  ymin = ymax = vy[0];
  for (i = 0; i < npts; i++) {
    int64_t px = vx[i], py = vy[i];
    if (py < ymin) {
      ymin = py;
      imin = i;
    }
    if (py > ymax) ymax = py;
    if (px > xmax) xmax = px;
    if (px < xmin) xmin = px;
    line2(im, p0x, p0y, px, py);
    p0x = px;
    p0y = py;
  }
  xmin = (xmin + delta) >> shift;
  xmax = (xmax + delta) >> shift;
  ymin = (ymin + delta) >> shift;
  ymax = (ymax + delta) >> shift;
  if (wrap32(xmax) < 0 || wrap32(ymax) < 0 || wrap32(xmin) >= im->W || wrap32(ymin) >= im->H) return;
  if (ymax > im->H - 1) ymax = im->H - 1;
  edge[0].idx = edge[1].idx = imin;
  edge[0].ye = edge[1].ye = y = wrap32(ymin);
  edge[0].di = 1;
  edge[1].di = npts - 1;
  edge[0].x = edge[1].x = -XY_ONE;
  edge[0].dx = edge[1].dx = 0;
  do {
    for (i = 0; i < 2; i++) {
      if (y >= edge[i].ye) {
        int idx0 = edge[i].idx, di = edge[i].di;
        int idx = idx0 + di;
        if (idx >= npts) idx -= npts;
        int ty = 0;
        for (; edges-- > 0;) {
          ty = wrap32((vy[idx] + delta) >> shift);
          if (ty > y) {
            int64_t xs = vx[idx0], xe = vx[idx];
            edge[i].ye = ty;
            edge[i].dx = ((xe - xs) * 2 + ((int64_t)ty - y)) / (2 * ((int64_t)ty - y));
            edge[i].x = xs;
            edge[i].idx = idx;
            break;
          }
          idx0 = idx;
          idx += di;
          if (idx >= npts) idx -= npts;
        }
      }
    }
    if (edges < 0) break;
    if (y >= 0) {
      int left = 0, right = 1;
      if (edge[0].x > edge[1].x) {
        left = 1;
        right = 0;
      }
      int xx1 = wrap32((edge[left].x + delta1) >> XY_SHIFT);
      int xx2 = wrap32((edge[right].x + delta2) >> XY_SHIFT);
      if (xx2 >= 0 && xx1 < im->W) {
        if (xx1 < 0) xx1 = 0;
        if (xx2 >= im->W) xx2 = im->W - 1;
        hline(im, y, xx1, xx2);
      }
      edge[0].x += edge[0].dx;
      edge[1].x += edge[1].dx;
    } else {
      /* OpenCV walks every negative row one by one (only x += dx happens there).  Equivalent
       * closed form: jump to just before the next event row (edge end, row 0, or ymax+1). */
      int64_t nxt = 0;
      if (edge[0].ye < nxt) nxt = edge[0].ye;
      if (edge[1].ye < nxt) nxt = edge[1].ye;
      if (ymax + 1 < nxt) nxt = ymax + 1;
      int64_t k = nxt - y; /* rows y .. nxt-1 are event free; k >= 1 */
      if (k < 1) k = 1;
      edge[0].x += edge[0].dx * k;
      edge[1].x += edge[1].dx * k;
      y += (int)(k - 1);
    }
  } while (++y <= (int)ymax);
}

/* Circle(img, center, radius, color, fill=1) */
static void circle_fill(img_t* im, int cx, int cy, int radius) {
  int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
  int W = im->W, H = im->H;
  int inside = cx >= radius && cx < W - radius && cy >= radius && cy < H - radius;
  while (dx >= dy) {
    int mask;
    int64_t y11 = (int64_t)cy - dy, y12 = (int64_t)cy + dy, y21 = (int64_t)cy - dx, y22 = (int64_t)cy + dx;
    int64_t x11 = (int64_t)cx - dx, x12 = (int64_t)cx + dx, x21 = (int64_t)cx - dy, x22 = (int64_t)cx + dy;
    if (inside) {
      hline(im, (int)y11, (int)x11, (int)x12);
      hline(im, (int)y12, (int)x11, (int)x12);
      hline(im, (int)y21, (int)x21, (int)x22);
      hline(im, (int)y22, (int)x21, (int)x22);
    } else if (x11 < W && x12 >= 0 && y21 < H && y22 >= 0) {
      if (x11 < 0) x11 = 0;
      if (x12 > W - 1) x12 = W - 1;
      if (y11 >= 0 && y11 < H) hline(im, (int)y11, (int)x11, (int)x12);
      if (y12 >= 0 && y12 < H) hline(im, (int)y12, (int)x11, (int)x12);
      if (x21 < W && x22 >= 0) {
        if (x21 < 0) x21 = 0;
        if (x22 > W - 1) x22 = W - 1;
        if (y21 >= 0 && y21 < H) hline(im, (int)y21, (int)x21, (int)x22);
        if (y22 >= 0 && y22 < H) hline(im, (int)y22, (int)x21, (int)x22);
      }
    }
    dy++;
    err += plus;
    plus += 2;
    mask = (err <= 0) - 1;
    err -= minus & mask;
    dx += mask;
    minus -= mask & 2;
  }
}

/* cvRound: round half to even (lrint under the default rounding mode) */
static int cv_round(double v) { return (int)lrint(v); }

/* PolyLine(open, 2 points) -> ThickLine(p0, p1, color, thickness, LINE_8, flags=3, shift=0) */
static void thick_line(img_t* im, int64_t p0x, int64_t p0y, int64_t p1x, int64_t p1y, int thickness) {
  p0x *= XY_ONE; p0y *= XY_ONE; p1x *= XY_ONE; p1y *= XY_ONE;
  if (thickness <= 1) {
    line_bresenham(im, (p0x + (XY_ONE >> 1)) >> XY_SHIFT, (p0y + (XY_ONE >> 1)) >> XY_SHIFT,
                   (p1x + (XY_ONE >> 1)) >> XY_SHIFT, (p1y + (XY_ONE >> 1)) >> XY_SHIFT);
    return;
  }
  const double INV_XY_ONE = 1. / XY_ONE;
  double dx = (p0x - p1x) * INV_XY_ONE, dy = (p1y - p0y) * INV_XY_ONE;
  double r = dx * dx + dy * dy;
  int odd = thickness & 1;
  int64_t th = (int64_t)thickness << (XY_SHIFT - 1);
  if (fabs(r) > DBL_EPSILON) {
    r = (th + odd * XY_ONE * 0.5) / sqrt(r);
    int64_t dpx = cv_round(dy * r), dpy = cv_round(dx * r);
    int64_t vx[4] = {p0x + dpx, p0x - dpx, p1x - dpx, p1x + dpx};
    int64_t vy[4] = {p0y + dpy, p0y - dpy, p1y - dpy, p1y + dpy};
    fill_convex_poly4(im, vx, vy);
  }
  for (int i = 0; i < 2; i++) { /* flags = 3: round caps on both ends */
    int cx = wrap32((p0x + (XY_ONE >> 1)) >> XY_SHIFT);
    int cy = wrap32((p0y + (XY_ONE >> 1)) >> XY_SHIFT);
    circle_fill(im, cx, cy, (int)((th + (XY_ONE >> 1)) >> XY_SHIFT));
    p0x = p1x;
    p0y = p1y;
  }
}

void orc_polyline2(uint8_t* img, int W, int H, int channels, int x0, int y0, int x1, int y1, const uint8_t* color,
                   int thickness) {
  img_t im = {img, W, H, channels, {color[0], channels > 1 ? color[1] : 0, channels > 2 ? color[2] : 0}};
  thick_line(&im, x0, y0, x1, y1, thickness);
}

int64_t orc_obs_bytes(const orc_map* m, const orc_cam* cam) {
  return (int64_t)cam->H * cam->W * (cam->format == ORC_FMT_CLASSES ? m->C : 3);
}

/* renderer.py:36-51 */
void orc_render(const orc_map* m, const orc_cam* cam, const int32_t* seg_i, int nseg, uint8_t* frame) {
  memset(frame, 0, (size_t)orc_obs_bytes(m, cam));
  for (int k = 0; k < nseg; k++) {
    const int32_t* s = seg_i + 5 * k;
    if (cam->format == ORC_FMT_CLASSES) {
      uint8_t col[3] = {255, 0, 0};
      orc_polyline2(frame + (size_t)s[0] * cam->H * cam->W, cam->W, cam->H, 1, s[1], s[2], s[3], s[4], col,
                    cam->line_thickness);
    } else {
      orc_polyline2(frame, cam->W, cam->H, 3, s[1], s[2], s[3], s[4], m->colors[s[0]], cam->line_thickness);
    }
  }
}

/* ============================================================== env.py for a batch */
static void observe(const orc_map* m, const orc_cam* cam, const orc_state* s, uint32_t flags, uint8_t* obs) {
  if (!obs) return;
  if (flags & ORC_F_NO_OBSERVATION) { /* env.py:78-81 */
    memset(obs, 0, (size_t)orc_obs_bytes(m, cam));
    return;
  }
  int cap = 4096;
  int32_t* seg = (int32_t*)malloc(sizeof(int32_t) * 5 * cap);
  int n = orc_capture_segments(m, cam, s, seg, NULL, cap);
  if (n > cap) {
    cap = n;
    seg = (int32_t*)realloc(seg, sizeof(int32_t) * 5 * cap);
    n = orc_capture_segments(m, cam, s, seg, NULL, cap);
  }
  orc_render(m, cam, seg, n, obs);
  free(seg);
}

void orc_reset_batch(const orc_map* m, const orc_car* c, const orc_cam* cam, int N, orc_state* st,
                     const int32_t* spawn_node, const uint8_t* mask, uint32_t flags, orc_info* info, uint8_t* obs,
                     int n_threads) {
  int64_t ob = orc_obs_bytes(m, cam);
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads > 0 ? n_threads : 1) schedule(static)
#endif
  for (int i = 0; i < N; i++) {
    if (mask && !mask[i]) continue;
    orc_reset(m, c, &st[i], spawn_node[i]);
    if (info) {
      memset(&info[i], 0, sizeof(orc_info));
      orc_get_info(m, c, &st[i], flags | ORC_F_WRAPPED, &info[i]); /* lp_len==1 -> empty_info, car.py:47-51 */
    }
    observe(m, cam, &st[i], flags, obs ? obs + (size_t)i * ob : NULL);
  }
}

/* ---------------------------------------------------------------------------------------------
 * Reward / termination wrappers (tinycarlo/wrapper/reward.py, termination.py, utils.py)
 * ------------------------------------------------------------------------------------------- */
double orc_linear_reward(double x, double max_x, double max_reward, double min_reward) {
  double y = (-max_reward / max_x) * fabs(x) + max_reward; /* utils.py:33 */
  if (max_reward > 0) return (min_reward > y) ? min_reward : y; /* utils.py:34-35, python max(y, min_reward) */
  return (min_reward < y) ? min_reward : y;                     /* utils.py:36-37, python min(y, min_reward) */
}

void orc_apply_terms(const orc_term* terms, int n_terms, int C, double tw, orc_info* o, int32_t* counters) {
  const double half = tw / 2;
  double reward = o->reward;
  int terminated = o->terminated;
  for (int t = 0; t < n_terms; t++) {
    const orc_term* T = &terms[t];
    switch (T->kind) {
      case ORC_T_LANELINE_SPARSE_REWARD: { /* reward.py:20-21; utils.py:15-19 sums into its own 0.0 first */
        double local = 0.0;
        for (int l = 0; l < C; l++)
          if (((T->layer_mask >> l) & 1u) && o->dist[l] < half) local += T->per_layer[l];
        reward = reward + local;
      } break;
      case ORC_T_LANELINE_LINEAR_REWARD: /* reward.py:40-41: one += per layer, in layer order */
        for (int l = 0; l < C; l++) reward = reward + orc_linear_reward(o->dist[l], tw, T->per_layer[l], 0.0);
        break;
      case ORC_T_CTE_SPARSE_REWARD: { /* reward.py:60 */
        double local = 0.0;
        if (fabs(o->cte) <= T->p[0]) local += T->p[1];
        reward = reward + local;
      } break;
      case ORC_T_CTE_LINEAR_REWARD: /* reward.py:83 */
        reward = reward + orc_linear_reward(o->cte, T->p[0], T->p[1], T->p[2]);
        break;
      case ORC_T_LANELINE_CROSSING_TERMINATION: /* termination.py:19-21 */
        for (int l = 0; l < C; l++)
          if (((T->layer_mask >> l) & 1u) && o->dist[l] <= half) terminated = 1;
        break;
      case ORC_T_CTE_TERMINATION:   /* termination.py:39-47 */
      case ORC_T_CRASH_TERMINATION: /* termination.py:61-69 */ {
        int cond = T->kind == ORC_T_CTE_TERMINATION ? (fabs(o->cte) > T->p[0]) : (fabs(o->velocity) < T->p[0]);
        if (cond) {
          counters[t] += 1;
          if (counters[t] >= T->number_of_steps) {
            terminated = 1;
            counters[t] = 0;
          }
        } else {
          counters[t] = 0;
        }
      } break;
      default: break;
    }
  }
  o->reward = reward;
  o->terminated = terminated;
}

void orc_step_batch(const orc_map* m, const orc_car* c, const orc_cam* cam, int N, orc_state* st,
                    const double* car_control, const int32_t* maneuver, uint32_t flags, orc_info* info, uint8_t* obs,
                    uint8_t* needs_reset, const int32_t* spawn_queue, int spawn_queue_len, int32_t* spawn_cursor,
                    int n_threads) {
  orc_step_batch_terms(m, c, cam, N, st, car_control, maneuver, flags, info, obs, needs_reset, spawn_queue,
                       spawn_queue_len, spawn_cursor, n_threads, NULL, 0, NULL);
}

void orc_step_batch_terms(const orc_map* m, const orc_car* c, const orc_cam* cam, int N, orc_state* st,
                          const double* car_control, const int32_t* maneuver, uint32_t flags, orc_info* info,
                          uint8_t* obs, uint8_t* needs_reset, const int32_t* spawn_queue, int spawn_queue_len,
                          int32_t* spawn_cursor, int n_threads, const orc_term* terms, int n_terms,
                          int32_t* counters) {
  orc_step_ext ext;
  memset(&ext, 0, sizeof(ext));
  ext.terms = terms;
  ext.n_terms = n_terms;
  ext.counters = counters;
  orc_step_batch_ext(m, c, cam, N, st, car_control, maneuver, flags, info, obs, needs_reset, spawn_queue,
                     spawn_queue_len, spawn_cursor, n_threads, &ext);
}

void orc_noise_classes(uint8_t* frame, int C, int H, int W, const int32_t* blobs, int n_blobs) {
  uint8_t* mask = (uint8_t*)malloc((size_t)H * W);
  for (int k = 0; k < C * n_blobs; k++) {
    const int c = k / n_blobs; /* observation.py:16-17: planes outer, blobs inner */
    const int x = blobs[5 * k], y = blobs[5 * k + 1], radius = blobs[5 * k + 2], mode = blobs[5 * k + 3],
              src = blobs[5 * k + 4];
    uint8_t* plane = frame + (size_t)c * H * W;
    if (mode) { /* observation.py:21-24 */
      img_t im = {mask, W, H, 1, {255, 255, 255}};
      memset(mask, 0, (size_t)H * W);
      circle_fill(&im, x, y, radius);
      const uint8_t* other = frame + (size_t)src * H * W;
      for (size_t i = 0; i < (size_t)H * W; i++) {
        uint8_t m = mask[i] ? (uint8_t)(other[i] & mask[i]) : 0; /* bitwise_and(src1, mask, mask=mask) */
        plane[i] = plane[i] | m;                                   /* bitwise_or */
      }
    } else { /* observation.py:26 */
      img_t im = {plane, W, H, 1, {0, 0, 0}};
      circle_fill(&im, x, y, radius);
    }
  }
  free(mask);
}

void orc_noise_blobs(uint64_t seed, uint32_t env, uint32_t step, int n_blobs, int C, int H, int W, int max_radius,
                     int32_t* out) {
  for (int k = 0; k < C * n_blobs; k++) {
    tc_blob b = tc_noise_blob(seed, env, step, (uint32_t)k, W, H, max_radius, C);
    out[5 * k] = b.x;
    out[5 * k + 1] = b.y;
    out[5 * k + 2] = b.r;
    out[5 * k + 3] = b.mode;
    out[5 * k + 4] = b.src;
  }
}

uint64_t orc_splitmix64_at(uint64_t seed, uint64_t n) { return tc_splitmix64_at(seed, n); }
uint32_t orc_spawn_index(uint64_t seed, uint32_t env, uint32_t cursor, uint32_t count) {
  return tc_spawn_index(seed, env, cursor, count);
}

void orc_step_batch_ext(const orc_map* m, const orc_car* c, const orc_cam* cam, int N, orc_state* st,
                        const double* car_control, const int32_t* maneuver, uint32_t flags, orc_info* info,
                        uint8_t* obs, uint8_t* needs_reset, const int32_t* spawn_queue, int spawn_queue_len,
                        int32_t* spawn_cursor, int n_threads, const orc_step_ext* ext) {
  const orc_term* terms = ext ? ext->terms : NULL;
  const int n_terms = ext ? ext->n_terms : 0;
  int32_t* counters = ext ? ext->counters : NULL;
  int64_t ob = orc_obs_bytes(m, cam);
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads > 0 ? n_threads : 1) schedule(static)
#endif
  for (int i = 0; i < N; i++) {
    orc_info* o = &info[i];
    uint8_t* oi = obs ? obs + (size_t)i * ob : NULL;
    if ((flags & ORC_F_AUTORESET) && needs_reset && needs_reset[i]) {
      int node;
      if ((flags & ORC_F_DEVICE_SPAWN) && ext && ext->spawn_n > 0)
        node = ext->spawn_table[tc_spawn_index(ext->spawn_seed, (uint32_t)i, (uint32_t)spawn_cursor[i],
                                               (uint32_t)ext->spawn_n)];
      else
        node = spawn_queue[(size_t)i * spawn_queue_len + (spawn_cursor[i] % spawn_queue_len)];
      spawn_cursor[i]++;
      orc_reset(m, c, &st[i], node);
      memset(o, 0, sizeof(*o));
      orc_get_info(m, c, &st[i], flags | ORC_F_WRAPPED, o);
      observe(m, cam, &st[i], flags, oi);
      needs_reset[i] = 0;
      continue;
    }
    /* env.py:118 */
    double v = np_clip(car_control[2 * i], -1.0, 1.0), s = np_clip(car_control[2 * i + 1], -1.0, 1.0);
    int status = 0;
    int trunc = orc_car_step(m, c, &st[i], v, s, maneuver[i], &status);
    observe(m, cam, &st[i], flags, oi);
    memset(o, 0, sizeof(*o));
    o->status = status;
    o->truncated = trunc;
    orc_get_info(m, c, &st[i], flags, o);
    if (n_terms > 0) 
      orc_apply_terms(terms, n_terms, m->C, c->track_width, o, counters + (size_t)i * ORC_MAX_TERMS);
    if ((flags & ORC_F_AUTORESET) && needs_reset) needs_reset[i] = (o->terminated || o->truncated) ? 1 : 0;
  }
}
