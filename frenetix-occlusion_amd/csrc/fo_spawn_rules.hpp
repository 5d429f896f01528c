// fo_spawn_rules.hpp -- the reference's three spawn rule families on the per-step cell classes, on the device.
// Included at the end of fo_scene.hip (one translation unit: the kernels read the static map), compiled with
// -ffp-contract=off like the rest of the scene stage.
//
// Replaces SpawnLocator.find_spawn_points' rule functions (ref: spawn_locator.py:80-139):
//   pedestrian behind a visible static obstacle   spawn_locator.py:323-476
//   pedestrian behind a turn                       spawn_locator.py:481-578
//   Car / Bicycle behind a visible dynamic obstacle  spawn_locator.py:145-317, rectangle fit :695-726
// The reference asks shapely for intersections of lines, buffers and polygons with the visible / occluded AREAS; here
// the same predicates are asked of the cell classes of fo_scene_visibility ("Rule families on cells"):
//   line.intersects(area)             -> a sample of the line (every cs/8) lies in a cell of that class
//   point.buffer(r).intersects(area)  -> the disc touches a cell square of that class
//   point.buffer(r).within(road)      -> every cell square the disc touches is road
//   area.buffer(b).exterior & line    -> samples where "disc of radius b touches the area" flips along the line
// The curvilinear frame is the polyline frame of the ego's reference path (utils/curvilinear.PolylineCS; the table
// [n][6] = x, y, s, segment length, unit tangent is built on the host once per reference path).
// Checked against oracle/fo_spawn_rules_ref.py (an independent NumPy restatement of the same definitions).
//
// Launch shape: one workgroup for the turn rule + one per obstacle (static rule: a wave; dynamic rule: 1 024 threads and
// 100 KB of LDS for the 97 x 97 candidate lattice), then one small workgroup that applies what depends on the order of
// the obstacles (sorted by distance, the maxima of the YAML, 5 m between pedestrians) and writes the spawn points.

namespace {

#ifndef FO_RULE_RUNS
#define FO_RULE_RUNS 1    // 0: tuning / test builds -- the dynamic rule's connected parts on the lattice nodes (rounds 4-5) instead of on the row runs
#endif
#ifndef FO_RULE_TRACE
#define FO_RULE_TRACE 0
#endif
#if FO_RULE_TRACE
#define RL_TICK(i) do { __syncthreads(); if (threadIdx.x == 0) rec[16 + (i)] = (double)wall_clock64(); } while (0)
#else
#define RL_TICK(i) do { } while (0)
#endif
// (per-workgroup stamps of the whole rule kernel: RL_WTICK(i), i < 8, row blockIdx of g_rule_wticks -- fo_debug_rule_wticks)
#if FO_RULE_TRACE
__device__ long long g_rule_wticks[8 * 1024];
#define RL_WTICK(i) do { if ((threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == (i >= 8 ? 1 : 0) && blockIdx.x < 1024) g_rule_wticks[8 * blockIdx.x + ((i) & 7)] = wall_clock64(); } while (0)
#else
#define RL_WTICK(i) do { } while (0)
#endif
#if FO_RULE_TRACE == 3   // tuning: stamps inside the dynamic rule's SET-UP (slots 2..7 of the workgroup's row)
#define RL_STICK(i) RL_WTICK(i)
#else
#define RL_STICK(i) do { } while (0)
#endif
// (thread 0's own clock inside the first rectangle fit, no barrier: trace build -DFO_RULE_TRACE=4)
#if FO_RULE_TRACE == 4
#define RL_FTICK(i) do { if (ftr && blockIdx.x < 1024) g_rule_wticks[8 * blockIdx.x + (i)] = wall_clock64(); } while (0)
#else
#define RL_FTICK(i) do { } while (0)
#endif
#if FO_RULE_TRACE == 2   // tuning: stamps INSIDE the first rectangle fit instead of after the two fits (slots 5, 6, 7)
#define RL_TICKF(i) do { __syncthreads(); if (threadIdx.x == 0 && rec[16 + (i)] == 0.0) rec[16 + (i)] = (double)wall_clock64(); } while (0)
#else
#define RL_TICKF(i) do { } while (0)
#endif

constexpr double RL_MAX_DIST_OBST = 30.0;          // spawn_locator.py:69
constexpr double RL_MIN_DIST_PED = 5.0;            // :73
constexpr double RL_TOL_SAME_DIR = 20.0 / 180.0 * 3.14159265358979323846;   // :68
constexpr double RL_BUFFER_SIDE = 12.0;            // :70
constexpr double RL_MIN_AREA = 10.0;               // :71
constexpr double RL_AREA_CAR = 9.0, RL_AREA_BIKE = 1.7;   // :72
constexpr int RL_LAT = 97;                         // nodes per side of the 0.25 m candidate lattice (2 x 12 m + 1)
constexpr int RL_TURNW = 1536;                      // vertices of the reference window the turn rule holds (40 m of path; fo_scene_spawn_rules refuses more)
constexpr int RL_FIFTHV = 512;                     // every-fifth-vertex queries of the window (dynamic rule outside an intersection)
constexpr int RL_MAXSAMP = 1024;                   // samples of a rule polyline (cs/8 steps; 40 m at cs = 0.5 -> 641)
constexpr int RL_REC = 24;                         // doubles per per-workgroup record
constexpr int RL_PATHV = 512;                      // vertices of the reference path table held in LDS (longer paths: read from HBM)
constexpr int RL_PARTS = 16;                       // workgroups that share a dynamic obstacle's candidate lattice
constexpr int RL_PVERT = 1024;                     // vertices of a dynamic obstacle's <= 8 candidate lanelet polygons held in LDS

enum { RL_TYPE_CAR = 0, RL_TYPE_BICYCLE = 3, RL_TYPE_PED = 4 };
enum { RL_SRC_DYNAMIC = 1, RL_SRC_STATIC = 2, RL_SRC_LEFT = 3, RL_SRC_RIGHT = 4 };

struct RuleView {
  const uint8_t *cls;       // [ny][nx] class bits of the step (1 road, 2 visible, 4 occluded)
  int ix0, iy0, nx, ny;     // window inside the raster
  double x0, y0, cs;        // raster origin, cell size
  const double *lane_yaw;   // [rny][rnx] or null
  int rnx, rny;
  int P;
  const int32_t *poly_off;
  const double *poly_xy, *poly_box;
  const double *left0;      // [P][2] or null
  const int32_t *pred0, *adj_left;
  int n_inter;
  const int32_t *inter_off, *inter_lanelet;
  const uint8_t *inter_kind;
  const double *path;       // [n_path][6] x, y, s, segment length, tangent x, tangent y
  int n_path;
};

__device__ inline int rl_class_at(const RuleView &v, double x, double y) {
  const int ix = (int)floor((x - v.x0) / v.cs) - v.ix0, iy = (int)floor((y - v.y0) / v.cs) - v.iy0;
  return (ix >= 0 && ix < v.nx && iy >= 0 && iy < v.ny) ? (int)v.cls[(size_t)iy * v.nx + ix] : 0;
}

// classes of the cell squares a disc touches: any has `bit` / all have `bit` (cells outside the window count as class 0)
__device__ inline void rl_disc(const RuleView &v, double x, double y, double rad, int bit, bool &any, bool &all) {
  const int ix0 = (int)floor((x - rad - v.x0) / v.cs) - v.ix0, iy0 = (int)floor((y - rad - v.y0) / v.cs) - v.iy0;
  const int ix1 = (int)floor((x + rad - v.x0) / v.cs) - v.ix0, iy1 = (int)floor((y + rad - v.y0) / v.cs) - v.iy0;
  any = false;
  all = true;
  for (int iy = iy0; iy <= iy1; ++iy)
    for (int ix = ix0; ix <= ix1; ++ix) {
      const double xl = v.x0 + (double)(v.ix0 + ix) * v.cs, yl = v.y0 + (double)(v.iy0 + iy) * v.cs;
      const double qx = fmin(fmax(x, xl), xl + v.cs), qy = fmin(fmax(y, yl), yl + v.cs);
      if ((qx - x) * (qx - x) + (qy - y) * (qy - y) <= rad * rad) {
        const int c = (ix >= 0 && ix < v.nx && iy >= 0 && iy < v.ny) ? (int)v.cls[(size_t)iy * v.nx + ix] : 0;
        if (c & bit) any = true; else all = false;
      }
    }
}
__device__ inline bool rl_disc_touches(const RuleView &v, double x, double y, double rad, int bit) {
  bool any, all;
  rl_disc(v, x, y, rad, bit, any, all);
  return any;
}

__device__ inline bool rl_lane_yaw_at(const RuleView &v, double x, double y, double &yaw) {
  if (!v.lane_yaw) return false;
  const int ix = (int)floor((x - v.x0) / v.cs), iy = (int)floor((y - v.y0) / v.cs);
  if (ix < 0 || ix >= v.rnx || iy < 0 || iy >= v.rny) return false;
  yaw = v.lane_yaw[(size_t)iy * v.rnx + ix];
  return yaw == yaw;
}

// crossing-number test, the rule of the road raster (half-open in y).
// rl_crossing_parity: vertices b .. e-1 of one ring through `get(k)`; the edge arithmetic of the plain loop (xc = xi + (y - yi)
// (xj - xi) / (yj - yi) on the edges that straddle y), with the vertices fetched EIGHT at a time in front of their tests: the
// loop used to be a chain of dependent round trips -- a load, a test, a branch per vertex, ~30 of them per lanelet polygon at
// 0.2-0.5 us each from the L2 -- and every "which lanelet holds this point" of the rule families waited for it (round 6)
// x < xi + (y - yi) (xj - xi) / (yj - yi) for an edge that straddles y (yi != yj) -- the quotient form is the checker's and
// decides whenever it is close; everywhere else the sign of s = (x - xi) d - (y - yi)(xj - xi) against the sign of d = yj - yi
// says the same without the ~40 instructions of a float64 quotient (the listed edges of a band are nearly all straddled by
// some lane of a wave).  The bound: the computed crossing differs from the real one of the rounded differences by
// < 2.01 u |m / d| + u |xc| (u = 2^-53; product, quotient and sum round once each), i.e. |s_real| > 3.01 u |m| + u |d xi|
// decides, and the computed s is within 3.02 u (|t2| + |m|) of s_real; 2^-48 (|t2| + |m| + |d xi|) = 32 u (...) covers both
// with room.  NaNs fail the comparison and take the quotient.
__device__ __forceinline__ bool rl_left_of_crossing(double x, double y, double xi, double yi, double xj, double yj) {
  const double d = yj - yi, m = (y - yi) * (xj - xi), t2 = (x - xi) * d, s = t2 - m;
  if (fabs(s) > 0x1p-48 * (fabs(t2) + fabs(m) + fabs(d * xi))) return (s < 0.0) != (d < 0.0);
  return x < xi + m / d;
}
template <class GET>
__device__ __forceinline__ int rl_crossing_parity(int b, int e, double x, double y, GET get) {
  int c = 0;
  if (e <= b) return 0;   // (an empty ring holds nothing -- and has no last vertex to start from)
  double2 pj = get(e - 1);
  for (int i0 = b; i0 < e; i0 += 8) {
    double2 pv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) pv[u] = get(i0 + u < e ? i0 + u : e - 1);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u < e) {
        const double xi = pv[u].x, yi = pv[u].y, xj = pj.x, yj = pj.y;
        if ((yi > y) != (yj > y)) {
          const double xc = xi + (y - yi) * (xj - xi) / (yj - yi);
          if (x < xc) c ^= 1;
        }
        pj = pv[u];
      }
  }
  return c;
}
__device__ inline bool rl_in_polygon(const RuleView &v, int p, double x, double y) {
  const double *bb = v.poly_box + 4 * (size_t)p;
  if (x < bb[0] || x > bb[2] || y < bb[1] || y > bb[3]) return false;
  const int b = v.poly_off[p], e = v.poly_off[p + 1];
  const double2 *xy = (const double2 *)v.poly_xy;
  return rl_crossing_parity(b, e, x, y, [&](int k) { return xy[k]; }) != 0;
}
// the same test by a whole wave (uniform arguments): a lane per edge -- bounding box, offsets and vertices are one round trip each,
// where a thread on its own walks the ring in chunks (1 + 1 + 1 + ceil(n / 8) trips); the edge arithmetic is that of
// rl_crossing_parity, the parity comes from a ballot
__device__ inline bool rl_in_polygon_wave(const RuleView &v, int p, double x, double y) {
  const int lane = threadIdx.x & 63;
  const int b = v.poly_off[p], e = v.poly_off[p + 1];
  const double2 *xy = (const double2 *)v.poly_xy;
  int c = 0;
  for (int i0 = b; i0 < e; i0 += 64) {
    const int i = i0 + lane;
    bool cross = false;
    if (i < e) {
      const double2 pi = xy[i], pj = xy[i == b ? e - 1 : i - 1];
      if ((pi.y > y) != (pj.y > y)) {
        const double xc = pi.x + (y - pi.y) * (pj.x - pi.x) / (pj.y - pi.y);
        cross = x < xc;
      }
    }
    c ^= (int)(__popcll(__ballot(cross)) & 1);
  }
  return c != 0;
}

// "Which lanelets hold these points?" for a workgroup (round 6): nq query points x P lanelets.  Pass A, a thread per (point,
// lanelet): the bounding box -- one round trip -- and the survivors (a handful: a point lies in two or three boxes) go to a list
// in LDS; pass B, a WAVE per survivor: the crossing-number test a lane per edge.  Three round trips and a barrier, where the
// thread-per-pair form took seven trips per test and ran the tests of one thread one after the other (the dynamic rule's set-up
// spent 7 of its 9 us there, stamps of the trace build -DFO_RULE_TRACE=3).  `keep(q, p)`: pairs worth asking at all;
// `act(q, p)`: called by lane 0 of the wave that found point q inside lanelet p -- the callers combine with atomics, so the
// order of the list does not matter.  *n_hits must be 0 on entry (and a barrier passed since); more survivors than the list
// holds: the thread-per-pair form.  Every thread of the workgroup must call; ends with a barrier.
template <class PT, class KEEP, class ACT>
__device__ __forceinline__ void rl_which_lanelets(const RuleView &v, int nq, PT point, KEEP keep, ACT act, int *hits, int cap, int *n_hits) {
  const int tid = threadIdx.x, nth = blockDim.x, wave = tid >> 6, lane = tid & 63, nw = nth >> 6;
  const unsigned total = (unsigned)nq * (unsigned)v.P;
  for (unsigned w = tid; w < total; w += nth) {
    const int q = (int)(w / (unsigned)v.P), p = (int)(w - (unsigned)q * (unsigned)v.P);
    if (!keep(q, p)) continue;
    double x, y;
    point(q, x, y);
    const double *bb = v.poly_box + 4 * (size_t)p;
    if (x < bb[0] || x > bb[2] || y < bb[1] || y > bb[3]) continue;
    const int h = atomicAdd(n_hits, 1);
    if (h < cap) hits[h] = (q << 16) | p;
  }
  __syncthreads();
  const int nh = *n_hits;
  if (nh <= cap && v.P < 65536) {
    for (int h = wave; h < nh; h += nw) {
      const int q = hits[h] >> 16, p = hits[h] & 0xffff;
      double x, y;
      point(q, x, y);
      const bool in = rl_in_polygon_wave(v, p, x, y);
      if (in && lane == 0) act(q, p);
    }
  } else {
    for (unsigned w = tid; w < total; w += nth) {
      const int q = (int)(w / (unsigned)v.P), p = (int)(w - (unsigned)q * (unsigned)v.P);
      if (!keep(q, p)) continue;
      double x, y;
      point(q, x, y);
      if (rl_in_polygon(v, p, x, y)) act(q, p);
    }
  }
  __syncthreads();
}

__device__ inline int rl_lanelet_of(const RuleView &v, double x, double y) {   // first lanelet (list order) holding the point
  for (int p = 0; p < v.P; ++p)
    if (rl_in_polygon(v, p, x, y)) return p;
  return -1;
}

// the same by a whole wave (every lane calls with the same point): lane l tests the lanelets l, l + 64, ...; the first group
// with a hit decides, its lowest lane = the first lanelet in list order.  (One lane walking the list is a chain of dependent
// round trips: bounding box after bounding box.)
__device__ inline int rl_lanelet_of_wave(const RuleView &v, double x, double y) {
  const int lane = threadIdx.x & 63;
  for (int p0 = 0; p0 < v.P; p0 += 64) {
    const int p = p0 + lane;
    const unsigned long long hit = __ballot(p < v.P && rl_in_polygon(v, p, x, y));
    if (hit) return p0 + __builtin_ctzll(hit);
  }
  return -1;
}

// ---- polyline frame (utils/curvilinear.PolylineCS): d positive to the left; false outside the projection domain
__device__ inline bool rl_to_curv(const RuleView &v, double x, double y, double &s, double &d) {
  const int ns = v.n_path - 1;
  double best = INFINITY, bt = 0.0, btc = 0.0;
  int k = 0;
  for (int i = 0; i < ns; ++i) {
    const double *q = v.path + 6 * (size_t)i;
    const double t = (x - q[0]) * q[4] + (y - q[1]) * q[5];
    const double tc = fmin(fmax(t, 0.0), q[3]);
    const double fx = q[0] + tc * q[4], fy = q[1] + tc * q[5];
    const double d2 = (x - fx) * (x - fx) + (y - fy) * (y - fy);
    if (d2 < best) { best = d2; k = i; bt = t; btc = tc; }
  }
  const double *q = v.path + 6 * (size_t)k;
  if ((k == 0 && bt < 0.0) || (k == ns - 1 && bt > q[3])) return false;
  const double fx = q[0] + btc * q[4], fy = q[1] + btc * q[5];
  s = q[2] + btc;
  d = (x - fx) * (-q[5]) + (y - fy) * q[4];
  return true;
}
// the same projection by a whole wave (every lane must call it with the same point): lane l takes the segments l, l + 64,
// ...; the wave keeps the smallest (distance, segment index) -- the first minimum of the sequential scan -- and every
// lane returns it.  (The sequential form is a chain of ~n_path dependent trips to the table in HBM.)
__device__ inline bool rl_to_curv_wave(const RuleView &v, double x, double y, double &s, double &d) {
  const int ns = v.n_path - 1, lane = threadIdx.x & 63;
  double best = INFINITY, bt = 0.0, btc = 0.0;
  int k = 0x7fffffff;
  for (int i = lane; i < ns; i += 64) {
    const double *q = v.path + 6 * (size_t)i;
    const double t = (x - q[0]) * q[4] + (y - q[1]) * q[5];
    const double tc = fmin(fmax(t, 0.0), q[3]);
    const double fx = q[0] + tc * q[4], fy = q[1] + tc * q[5];
    const double d2 = (x - fx) * (x - fx) + (y - fy) * (y - fy);
    if (d2 < best) { best = d2; k = i; bt = t; btc = tc; }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double b2 = __shfl_xor(best, off), t2 = __shfl_xor(bt, off), c2 = __shfl_xor(btc, off);
    const int k2 = __shfl_xor(k, off);
    if (b2 < best || (b2 == best && k2 < k)) { best = b2; k = k2; bt = t2; btc = c2; }
  }
  if (k == 0x7fffffff) return false;
  const double *q = v.path + 6 * (size_t)k;
  if ((k == 0 && bt < 0.0) || (k == ns - 1 && bt > q[3])) return false;
  const double fx = q[0] + btc * q[4], fy = q[1] + btc * q[5];
  s = q[2] + btc;
  d = (x - fx) * (-q[5]) + (y - fy) * q[4];
  return true;
}
// N points at once (every lane calls with the same points): the N searches share the pass over the segments and their
// exchanges overlap -- the arithmetic per point is rl_to_curv_wave's (same bits); ok bit n = point n projects onto the path.
// (A projection alone is six exchange steps of LDS-crossbar latency: one after the other, the five of an obstacle's centre
// and corners cost the static rule 10 us.)
template <int N>
__device__ inline unsigned rl_to_curv_wave_n(const RuleView &v, const double *x, const double *y, double *s, double *d) {
  const int ns = v.n_path - 1, lane = threadIdx.x & 63;
  double best[N], bt[N], btc[N];
  int k[N];
#pragma unroll
  for (int n = 0; n < N; ++n) { best[n] = INFINITY; bt[n] = 0.0; btc[n] = 0.0; k[n] = 0x7fffffff; }
  for (int i = lane; i < ns; i += 64) {
    const double *q = v.path + 6 * (size_t)i;
    const double q0 = q[0], q1 = q[1], q3 = q[3], q4 = q[4], q5 = q[5];
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const double t = (x[n] - q0) * q4 + (y[n] - q1) * q5;
      const double tc = fmin(fmax(t, 0.0), q3);
      const double fx = q0 + tc * q4, fy = q1 + tc * q5;
      const double d2 = (x[n] - fx) * (x[n] - fx) + (y[n] - fy) * (y[n] - fy);
      if (d2 < best[n]) { best[n] = d2; k[n] = i; bt[n] = t; btc[n] = tc; }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const double b2 = __shfl_xor(best[n], off);
      const int k2 = __shfl_xor(k[n], off);
      if (b2 < best[n] || (b2 == best[n] && k2 < k[n])) { best[n] = b2; k[n] = k2; }
    }
  }
  unsigned ok = 0u;
#pragma unroll
  for (int n = 0; n < N; ++n) {
    if (k[n] == 0x7fffffff) continue;
    // (segment i lives in lane i mod 64: its parameters from there instead of through the six exchange steps)
    const double t_ = __shfl(bt[n], k[n] & 63), tc_ = __shfl(btc[n], k[n] & 63);
    const double *q = v.path + 6 * (size_t)k[n];
    if ((k[n] == 0 && t_ < 0.0) || (k[n] == ns - 1 && t_ > q[3])) continue;
    const double fx = q[0] + tc_ * q[4], fy = q[1] + tc_ * q[5];
    s[n] = q[2] + tc_;
    d[n] = (x[n] - fx) * (-q[5]) + (y[n] - fy) * q[4];
    ok |= 1u << n;
  }
  return ok;
}
__device__ inline bool rl_to_cart(const RuleView &v, double s, double d, double &x, double &y) {
  const int n = v.n_path;
  if (s < v.path[2] || s > v.path[6 * (size_t)(n - 1) + 2]) return false;
  int lo = 0, hi = n;   // searchsorted(s_table, s, side = 'right'): first index with table > s
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (v.path[6 * (size_t)mid + 2] <= s) lo = mid + 1; else hi = mid;
  }
  const int k = min(lo - 1, n - 2);
  const double *q = v.path + 6 * (size_t)k;
  x = q[0] + (s - q[2]) * q[4] + d * (-q[5]);
  y = q[1] + (s - q[2]) * q[5] + d * q[4];
  return true;
}

// distance between segment ab and a convex quadrilateral c [4][2] (0 if they touch or the segment starts / ends inside)
__device__ inline double rl_pt_seg(double px, double py, double ax, double ay, double bx, double by) {
  const double dx = bx - ax, dy = by - ay, l2 = dx * dx + dy * dy;
  double t = 0.0;
  if (l2 != 0.0) t = fmin(1.0, fmax(0.0, ((px - ax) * dx + (py - ay) * dy) / l2));
  const double qx = px - (ax + t * dx), qy = py - (ay + t * dy);
  return sqrt(qx * qx + qy * qy);
}
__device__ inline bool rl_inside_quad(double px, double py, const double *c) {
  int sgn = 0;
  for (int i = 0; i < 4; ++i) {
    const int j = (i + 1) & 3;
    const double cr = (c[2 * j] - c[2 * i]) * (py - c[2 * i + 1]) - (c[2 * j + 1] - c[2 * i + 1]) * (px - c[2 * i]);
    if (fabs(cr) > 1e-12) {
      if (sgn == 0) sgn = cr > 0 ? 1 : -1;
      else if ((cr > 0) != (sgn > 0)) return false;
    }
  }
  return true;
}
__device__ inline double rl_seg_rect_distance(double ax, double ay, double bx, double by, const double *c) {
  if (rl_inside_quad(ax, ay, c) || rl_inside_quad(bx, by, c)) return 0.0;
  double best = INFINITY;
  for (int i = 0; i < 4; ++i) {
    const int j = (i + 1) & 3;
    const double p3x = c[2 * i], p3y = c[2 * i + 1], p4x = c[2 * j], p4y = c[2 * j + 1];
    const double d1x = bx - ax, d1y = by - ay, d2x = p4x - p3x, d2y = p4y - p3y;
    const double den = d1x * d2y - d1y * d2x;
    if (fabs(den) > 1e-14) {
      const double wx = p3x - ax, wy = p3y - ay;
      const double t = (wx * d2y - wy * d2x) / den, u = (wx * d1y - wy * d1x) / den;
      if (t >= 0.0 && t <= 1.0 && u >= 0.0 && u <= 1.0) return 0.0;
    }
    best = fmin(best, fmin(fmin(rl_pt_seg(ax, ay, p3x, p3y, p4x, p4y), rl_pt_seg(bx, by, p3x, p3y, p4x, p4y)),
                           fmin(rl_pt_seg(p3x, p3y, ax, ay, bx, by), rl_pt_seg(p4x, p4y, ax, ay, bx, by))));
  }
  return best;
}

struct RuleParams {
  double ego_x, ego_y, ego_yaw, ego_s, ego_d, s_threshold;
  double ped_width, ped_length;
  int intention;                 // 0 straight ahead, 1 left turn, 2 right turn
  int win_i0, win_i1;            // reference window = path vertices [i0, i1)
  int behind_static, behind_turn, behind_dynamic, max_static, max_dynamic;
  int label_nodes;               // tests (FO_SCENE_RULE_NODES=1): the dynamic rule's connected parts on the lattice nodes, not on the row runs
};

// sample i of a polyline with cumulative lengths cum[] (np.interp on both coordinates); n_s samples, step apart, the
// last one clamped to the end
__device__ inline void rl_sample(const double *px, const double *py, const double *cum, int n, double q, double &x, double &y) {
  if (q >= cum[n - 1]) { x = px[n - 1]; y = py[n - 1]; return; }
  int lo = 0, hi = n - 1;   // largest j with cum[j] <= q
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (cum[mid] <= q) lo = mid; else hi = mid - 1;
  }
  const double w = cum[lo + 1] - cum[lo];
  x = (px[lo + 1] - px[lo]) / w * (q - cum[lo]) + px[lo];
  y = (py[lo + 1] - py[lo]) / w * (q - cum[lo]) + py[lo];
}

// ---------------------------------------------------------------- pedestrian behind a turn (one wave)
// rec: [0] valid, [1] x, [2] y, [3] s_ph, [4] d (the lateral phantom offset), [5] source
__device__ __forceinline__ void rl_turn_rule(const RuleView &v, const RuleParams &pr, double *rec, double *lx, double *ly, double *cum,
                             unsigned char *inside) {
  const int lane = threadIdx.x & 63;
  if (lane == 0) rec[0] = 0.0;
  const int nw = pr.win_i1 - pr.win_i0;
  if (nw < 2) return;
  if (nw > RL_TURNW) { if (lane == 0) rec[0] = -1.0; return; }   // (refused by the host entry already; -1: out of table space, see the selection kernel)
  const bool left = pr.intention == 1;
  // the line: the reference window, for a left turn shifted 3 m to the left (spawn_locator.py:510-518)
  bool ok = true;
  for (int i = lane; i < nw; i += 64) {
    const double *q = v.path + 6 * (size_t)(pr.win_i0 + i);
    double x = q[0], y = q[1];
    if (left) ok = rl_to_cart(v, q[2], 3.0, x, y) && ok;
    lx[i] = x;
    ly[i] = y;
  }
  if (__ballot(!ok)) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (lane == 0) {
    cum[0] = 0.0;
    for (int i = 1; i < nw; ++i) cum[i] = cum[i - 1] + sqrt((lx[i] - lx[i - 1]) * (lx[i] - lx[i - 1]) + (ly[i] - ly[i - 1]) * (ly[i] - ly[i - 1]));
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const double total = cum[nw - 1], step = v.cs / 8.0;
  if (!(total > 0.0)) return;
  const int ns = (int)ceil((total + 0.5 * step) / step);
  if (ns > RL_MAXSAMP) { if (lane == 0) rec[0] = -1.0; return; }   // a line longer than the sample table (cells below 0.32 m at a 40 m window)
  for (int i = lane; i < ns; i += 64) {
    double x, y;
    rl_sample(lx, ly, cum, nw, fmin((double)i * step, total), x, y);
    inside[i] = (rl_class_at(v, x, y) & 4) ? 1 : 0;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  // runs of consecutive samples in occluded cells: the first point of the only run, of the LAST run when there are several (:528)
  int first_of_last = -1;
  for (int i = lane; i < ns; i += 64)
    if (inside[i] && (i == 0 || !inside[i - 1])) first_of_last = i;      // ascending per lane: its last run start
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) first_of_last = max(first_of_last, __shfl_xor(first_of_last, off));
  if (first_of_last < 0) return;
  double fx, fy;
  rl_sample(lx, ly, cum, nw, fmin((double)first_of_last * step, total), fx, fy);
  double s_int, d_int;
  if (!rl_to_curv_wave(v, fx, fy, s_int, d_int)) return;
  if (lane != 0) return;
  double s_ph = s_int + (left ? -0.5 : 0.0);
  if (s_ph > pr.s_threshold || s_ph < pr.ego_s + 3.0) return;                    // :542
  const double d_ph = left ? 1.0 : -1.0, d_off = d_ph + (left ? 3.0 : 0.0);     // :546
  double x, y;
  if (!rl_to_cart(v, s_ph, d_off, x, y)) return;
  while (rl_disc_touches(v, x, y, 0.5, 2)) {                                     // :552-554
    s_ph += 0.5;
    if (!rl_to_cart(v, s_ph, d_off, x, y)) return;
  }
  rec[1] = x; rec[2] = y; rec[3] = s_ph; rec[4] = d_ph; rec[5] = left ? RL_SRC_LEFT : RL_SRC_RIGHT;
  rec[0] = 1.0;   // (the obstacle and heading conditions, :557-572, are applied by the selection workgroup)
}

// ---------------------------------------------------------------- pedestrian behind a static obstacle (two waves)
// rec: [0] distance to the ego, [1] role (1 static candidate, 2 dynamic candidate, 0 nothing), per line li = 0, 1:
// [2 + 6 li] valid, x, y, s, d, yaw
// Wave li of the workgroup takes cross line li (the rear and the front end of the obstacle's extent along the path): the two
// lines are independent chains of projections and class look-ups, a wave's worth of latency each.  Both waves work out the
// obstacle's extent for themselves (the same arithmetic: no exchange); sx, sy, near: this wave's scratch.
__device__ __forceinline__ void rl_static_rule(const RuleView &v, const RuleParams &pr, int o, int O, const double *ocorn, const double *ocen,
                               const uint8_t *oflags, const uint8_t *ovis, double *rec, double *sx, double *sy,
                               unsigned char *near, int li) {
  const int lane = threadIdx.x & 63;
  const double cx = ocen[2 * o], cy = ocen[2 * o + 1];
  const double *oc = ocorn + 8 * (size_t)o;
  if (lane == 0) rec[2 + 6 * li] = 0.0;
  if (sqrt((pr.ego_x - cx) * (pr.ego_x - cx) + (pr.ego_y - cy) * (pr.ego_y - cy)) > RL_MAX_DIST_OBST) return;   // :369
  // the centre and the four corners in one pass over the path (rl_to_curv_wave_n)
  const double p5x[5] = {cx, oc[0], oc[2], oc[4], oc[6]}, p5y[5] = {cy, oc[1], oc[3], oc[5], oc[7]};
  double p5s[5], p5d[5];
  RL_WTICK(2);
  const unsigned ok5 = rl_to_curv_wave_n<5>(v, p5x, p5y, p5s, p5d);
  RL_WTICK(3);
  if (!(ok5 & 1u)) return;
  const double ob_s = p5s[0];
  // :380 compares with ego s + s_threshold although s_threshold already contains ego s (kept as in the reference)
  if (pr.ego_s + pr.s_threshold < ob_s || ob_s < pr.ego_s + 3.0) return;
  if (ok5 != 31u) return;
  double s_min = INFINITY, s_max = -INFINITY, d_min = INFINITY, d_max = -INFINITY;
  for (int i = 1; i < 5; ++i) {
    s_min = fmin(s_min, p5s[i]); s_max = fmax(s_max, p5s[i]); d_min = fmin(d_min, p5d[i]); d_max = fmax(d_max, p5d[i]);
  }
  s_min -= 0.8; s_max += 0.8; d_min -= 0.8; d_max += 0.8;                          // :384-390
  double yaw_l = 0.0;
  const bool have_yaw = rl_lane_yaw_at(v, cx, cy, yaw_l);
  RL_WTICK(4);
  for (int once = 0; once < 1; ++once) {   // (this wave's line; `continue` = no point on it)
    const double s_line = li == 0 ? s_min : s_max;
    double ax, ay, bx, by;
    if (!rl_to_cart(v, s_line, d_min, ax, ay) || !rl_to_cart(v, s_line, d_max, bx, by)) continue;
    const double total = sqrt((bx - ax) * (bx - ax) + (by - ay) * (by - ay)), step = v.cs / 8.0;
    if (!(total > 0.0)) continue;
    const int ns = (int)ceil((total + 0.5 * step) / step);
    if (ns > RL_MAXSAMP) { if (lane == 0) rec[2 + 6 * li] = -1.0; continue; }       // (a cross line of > 64 m at 0.5 m cells: out of table space)
    const double b = pr.ped_length / 2.0 * 1.3;                                    // :414
    bool t_occ = false, t_vis = false;
    for (int i = lane; i < ns; i += 64) {
      const double q = fmin((double)i * step, total);
      double x = bx, y = by;
      if (q < total) { x = (bx - ax) / total * q + ax; y = (by - ay) / total * q + ay; }
      sx[i] = x; sy[i] = y;
      const int c = rl_class_at(v, x, y);
      t_occ = t_occ || (c & 4);
      t_vis = t_vis || (c & 2);
      near[i] = rl_disc_touches(v, x, y, b, 2) ? 1 : 0;
    }
    RL_WTICK(5);
    if (!__ballot(t_occ) || !__ballot(t_vis)) continue;                           // :406-408
    bool blocked = false;                                                         // :409-411: any VISIBLE obstacle within half a pedestrian width
    for (int j = lane; j < O; j += 64)
      if ((oflags[j] & 1) && ovis[j] && rl_seg_rect_distance(ax, ay, bx, by, ocorn + 8 * (size_t)j) <= pr.ped_width / 2.0) blocked = true;
    if (__ballot(blocked)) continue;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // candidates: where "the disc touches the visible area" flips, the sample just outside (:414-415) -- a lane per sample
    // pair; one candidate: that one; several (MultiPoint, :419-433): the one nearest to the lanelet's first left vertex among
    // those inside the occluded area (the first of equally near ones: smallest (distance, index) over the wave)
    double spx = 0.0, spy = 0.0;
    bool okp = false;
    {
      int n_c = 0, only = -1;
      for (int i = lane; i + 1 < ns; i += 64)
        if (near[i] != near[i + 1]) { ++n_c; only = near[i] ? i + 1 : i; }
      int n_all = n_c, only_all = only;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) { n_all += __shfl_xor(n_all, off); only_all = max(only_all, __shfl_xor(only_all, off)); }
      bool found = false;
      if (n_all == 1) {
        spx = sx[only_all]; spy = sy[only_all]; found = true;
      } else if (n_all > 1) {
        const int ll = rl_lanelet_of_wave(v, cx, cy);
        const double anx = (ll >= 0 && v.left0) ? v.left0[2 * ll] : cx, any_ = (ll >= 0 && v.left0) ? v.left0[2 * ll + 1] : cy;
        double bestd = INFINITY;
        int bc = 0x7fffffff, bi = 0x7fffffff;   // candidate sample, and the sample pair it came from (orders ties)
        for (int i = lane; i + 1 < ns; i += 64)
          if (near[i] != near[i + 1]) {
            const int c = near[i] ? i + 1 : i;
            const double dd = sqrt((anx - sx[c]) * (anx - sx[c]) + (any_ - sy[c]) * (any_ - sy[c]));
            if ((rl_class_at(v, sx[c], sy[c]) & 4) && dd < bestd) { bestd = dd; bc = c; bi = i; }   // ascending per lane: first minimum
          }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
          const double d2 = __shfl_xor(bestd, off);
          const int c2 = __shfl_xor(bc, off), i2 = __shfl_xor(bi, off);
          if (d2 < bestd || (d2 == bestd && i2 < bi)) { bestd = d2; bc = c2; bi = i2; }
        }
        if (bc != 0x7fffffff) { spx = sx[bc]; spy = sy[bc]; found = true; }
      }
      okp = found;
      if (okp && lane == 0) {
        bool any, all;
        rl_disc(v, spx, spy, 0.15, 2, any, all);
        if (any) okp = false;                                                     // :440
        rl_disc(v, spx, spy, 0.15, 1, any, all);
        if (!all) okp = false;                                                    // :444
      }
    }
    // the point's curvilinear position: projected by the whole wave (lane 0 holds the point)
    const bool okw = __shfl((int)okp, 0) != 0;
    spx = __shfl(spx, 0);
    spy = __shfl(spy, 0);
    double ss = 0.0, sd = 0.0;
    const bool okc = okw && rl_to_curv_wave(v, spx, spy, ss, sd);
    RL_WTICK(6);
    if (lane == 0) {
      double *r = rec + 2 + 6 * li;
      r[0] = (okc && have_yaw) ? 1.0 : 0.0; r[1] = spx; r[2] = spy; r[3] = ss; r[4] = sd; r[5] = yaw_l + 1.5707963267948966;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}

// ---------------------------------------------------------------- Car / Bicycle behind a dynamic obstacle (a workgroup)
struct RlFit { double area, cx, cy, jac; bool any; };
constexpr int RL_THREADS_DYN = 1024;   // threads of the workgroup that runs the rule (= RL_THREADS below)

// rec: [0] distance, [1] role = 2, [2] car valid, [3] car x, [4] car y, [5] bicycle valid, [6] x, [7] y
__device__ __forceinline__ void rl_dynamic_rule(const RuleView &v, const RuleParams &pr, int o, const double *ocorn, const double *ocen,
                                const double *oyaw, const double *odims, double *rec, int *lab, double *red, int *ired,
                                unsigned char *fitok, double *polyv, int part, int *g_lab, int *g_cnt) {
  const int tid = threadIdx.x, nth = blockDim.x;
  const double cx = ocen[2 * o], cy = ocen[2 * o + 1], oy = oyaw[o], olen = odims[2 * o], owid = odims[2 * o + 1];
  const double *oc = ocorn + 8 * (size_t)o;
  __shared__ int s_pol[8], s_npol, s_go, s_changed, s_best, s_bestn, s_ego_ll, s_inter, s_nin, s_in[16], s_vll[RL_FIFTHV], s_relc, s_curv_ok;
  __shared__ int s_poff[9], s_plds, s_inter_first, s_nhit, s_ecnt[1];
  __shared__ double s_c[2], s_yaw, s_pbox[32], s_obsd[2], s_oc[8];
  // relevant lanelets (:171-202): the other incomings / inner lanelets of the intersection the ego is in, else the
  // oncoming neighbours (adj_left) of the lanelets under every fifth vertex of the reference window.  Flags per lanelet
  // in ired[0, P): bit0 relevant, bit1 inner, bit2 holds the obstacle's centre.  Every "which lanelet holds this point" below is asked of all lanelets at
  // once, a thread per (point, lanelet) -- the first lanelet in list order by atomicMin -- instead of one thread walking
  // the polygon table in HBM
  // out of table space (-1 in the Car slot's validity word; every part of the obstacle decides the same and leaves before the
  // hand-off): more lanelets than the flag array holds (refused by the host entry already), or more fifth vertices of the window
  if (v.P > RL_LAT * RL_LAT || (pr.win_i1 - pr.win_i0 + 4) / 5 > RL_FIFTHV) {
    if (part == 0 && tid == 0) { rec[2] = -1.0; rec[5] = 0.0; }
    return;
  }
  for (int p = tid; p < v.P; p += nth) ired[p] = 0;
  if (tid == 0) {
    rec[2] = 0.0; rec[5] = 0.0;
    s_go = 0; s_npol = 0; s_inter = -1; s_inter_first = 0x7fffffff; s_ego_ll = 0x7fffffff; s_nin = 0; s_relc = 0; s_curv_ok = 0; s_nhit = 0;
  }
  if (tid == 0) s_ecnt[0] = 0;   // (edge_band's count, far below)
  if (tid >= 64 && tid < 72) s_oc[tid - 64] = oc[tid - 64];   // (the obstacle's corners for the shadow test: LDS instead of a load from HBM's caches per edge and point)
  if (tid < RL_FIFTHV) s_vll[tid] = 0x7fffffff;
  __syncthreads();
  RL_STICK(2);
  // (measured and dropped, round 6: the lanelets under every fifth vertex of the reference window -- needed when the ego turns
  // out to be in no intersection, two barriers further down -- asked in this same pass: +4 us in front of the lattice where
  // there IS an intersection, the usual case of the rule)
  // the lanelets under the ego (the first in list order) and under the obstacle's centre (all of them; also flagged: more than
  // sixteen are re-collected in list order below); the hit list borrows the lattice array, which is idle until the hand-off
  rl_which_lanelets(v, 2, [&](int q, double &x, double &y) { x = q == 0 ? pr.ego_x : cx; y = q == 0 ? pr.ego_y : cy; },
                    [](int, int) { return true; },
                    [&](int q, int p) {
                      if (q == 0) { atomicMin(&s_ego_ll, p); return; }
                      atomicOr(&ired[p], 4);
                      const int k = atomicAdd(&s_nin, 1);
                      if (k < 16) s_in[k] = p;
                    }, lab, 2048, &s_nhit);
  RL_STICK(3);
  if (tid < 64) {   // the obstacle's curvilinear position (wave 0)
    double ob_s, ob_d;
    const bool okc = rl_to_curv_wave(v, cx, cy, ob_s, ob_d);
    if (tid == 0) { s_curv_ok = okc ? 1 : 0; s_obsd[0] = ob_s; s_obsd[1] = ob_d; s_nhit = 0; }
  }
  RL_STICK(4);
  __syncthreads();
  if (s_ego_ll == 0x7fffffff) return;
  // the first intersection (list order) that lists the ego's lanelet: a thread per table entry and an atomicMin on the
  // intersection's index (one thread walking the table was a chain of dependent loads)
  {
    const int n_ent = v.n_inter > 0 ? v.inter_off[v.n_inter] : 0;
    for (int e = tid; e < n_ent; e += nth)
      if (v.inter_lanelet[e] == s_ego_ll) {
        int it = 0;
        while (it + 1 < v.n_inter && v.inter_off[it + 1] <= e) ++it;
        atomicMin(&s_inter_first, it);
      }
  }
  __syncthreads();
  if (tid == 0) s_inter = s_inter_first == 0x7fffffff ? -1 : s_inter_first;
  __syncthreads();
  RL_STICK(5);
  if (s_inter >= 0) {
    for (int e = v.inter_off[s_inter] + tid; e < v.inter_off[s_inter + 1]; e += nth) {
      const int p = v.inter_lanelet[e];
      atomicOr(&ired[p], (p != s_ego_ll ? 1 : 0) | (v.inter_kind[e] == 1 ? 2 : 0));
    }
  } else if (v.adj_left) {
    const int nv = (pr.win_i1 - pr.win_i0 + 4) / 5;   // every fifth vertex of the reference window (40 m: a dozen; <= RL_FIFTHV, above)
    rl_which_lanelets(v, nv, [&](int q, double &x, double &y) { const double *w_ = v.path + 6 * (size_t)(pr.win_i0 + 5 * q); x = w_[0]; y = w_[1]; },
                      [](int, int) { return true; }, [&](int q, int p) { atomicMin(&s_vll[q], p); }, lab, 2048, &s_nhit);
    if (tid < nv) {
      const int ll = s_vll[tid];
      if (ll != 0x7fffffff && v.adj_left[ll] >= 0) atomicOr(&ired[v.adj_left[ll]], 1);
    }
  }
  __syncthreads();
  RL_STICK(6);
  if (tid == 0) {
    s_nhit = 0;   // (the next lanelet query -- the centroid's -- finds its list empty)
    do {
      if (sqrt((pr.ego_x - cx) * (pr.ego_x - cx) + (pr.ego_y - cy) * (pr.ego_y - cy)) > RL_MAX_DIST_OBST) break;   // :215
      // the obstacle's lanelets (all that hold its centre) in list order.  Up to sixteen arrived through the atomic counter
      // in any order and are sorted; MORE than sixteen (a centre on a pile of overlapping lanelets) would leave a subset that
      // depends on the arrival order -- and the sixteen workgroups of an obstacle must take identical decisions before their
      // hand-off ticket below -- so the first sixteen in list order are collected from the flags instead
      // (and more than sixteen is more than the rule holds: the step's list is refused, below)
      int n_ob = min(s_nin, 16);
      bool short_of_space = s_nin > 16;
      if (s_nin > 16) {
        n_ob = 0;
        for (int p = 0; p < v.P && n_ob < 16; ++p)
          if (ired[p] & 4) s_in[n_ob++] = p;
      }
      for (int i = 1; i < n_ob; ++i) {   // (a point lies on a handful of lanelets: insertion sort)
        const int key = s_in[i];
        int j = i - 1;
        while (j >= 0 && s_in[j] > key) { s_in[j + 1] = s_in[j]; --j; }
        s_in[j + 1] = key;
      }
      int first_rel = -1;
      bool any_rel = false, all_inner = true;
      for (int i = 0; i < n_ob; ++i) {
        const int p = s_in[i];
        if (ired[p] & 1) {
          any_rel = true;
          if (first_rel < 0) first_rel = p;
          if (s_npol < 7) s_pol[s_npol++] = p; else short_of_space = true;
        }
        if (!(ired[p] & 2)) all_inner = false;
      }
      // more relevant lanelets under the obstacle's centre than candidate polygons are held (seven + the predecessor): out of
      // table space, -1 in the Car slot's validity word (the selection kernel refuses the step's list).  Every part of the
      // obstacle decides the same and writes the same -- whichever clears the word last at its own start writes it again here.
      if (short_of_space) { rec[2] = -1.0; break; }
      if (!any_rel) break;                                                        // :222
      if (!s_curv_ok) break;
      if (s_obsd[0] < pr.ego_s + 3.0 || fabs(s_obsd[1]) > 15.0) break;            // :234
      if (s_inter >= 0 && n_ob > 0 && all_inner && v.pred0 && v.pred0[first_rel] >= 0 && s_npol < 8) s_pol[s_npol++] = v.pred0[first_rel];   // :249-252
      s_go = 1;
    } while (false);
  }
  __syncthreads();
  RL_STICK(7);
  if (!s_go) return;
  RL_TICK(0);
  const int npol = s_npol;
  // where the candidate polygons' vertices go in LDS (member() below); too many vertices: read from HBM as before.  (Round 6:
  // a thread per polygon asks for its offsets, then ONE flat copy of all vertices -- thread 0 used to walk the offset table,
  // sixteen loads one after the other, and the copy ran polygon by polygon behind a load of its own each)
  __shared__ int s_pb0[8], s_plen[8];
  if (tid < npol) {
    const int p = s_pol[tid], b0 = v.poly_off[p];
    s_pb0[tid] = b0;
    s_plen[tid] = v.poly_off[p + 1] - b0;
  }
  if (tid < 4 * npol) s_pbox[tid] = v.poly_box[4 * (size_t)s_pol[tid >> 2] + (tid & 3)];
  __syncthreads();
  if (tid == 0) {
    int tot = 0;
    for (int i = 0; i < npol; ++i) { s_poff[i] = tot; tot += s_plen[i]; }
    s_poff[npol] = tot;
    s_plds = tot <= RL_PVERT;
  }
  __syncthreads();
  const bool plds = s_plds != 0;
#if FO_RULE_TRACE < 3
  RL_WTICK(2);
#endif
  if (plds) {
    const int n2 = 2 * s_poff[npol];
    for (int k = tid; k < n2; k += nth) {
      int i = 0;
      while (i + 1 < npol && k >= 2 * s_poff[i + 1]) ++i;
      polyv[k] = v.poly_xy[2 * (size_t)s_pb0[i] + (k - 2 * s_poff[i])];
    }
  }
  // (the relevance flags move to the end of `lab`'s companion array later; keep a compact copy for the centroid test)
  unsigned char *relflag = fitok + 1536;   // [P] bit0: relevant -- only consulted for the few lanelets holding the centroid
  const bool rel_fits = v.P <= 512;
  if (rel_fits)
    for (int p = tid; p < v.P; p += nth) relflag[p] = (unsigned char)(ired[p] & 1);
  __syncthreads();
#if FO_RULE_TRACE < 3
  RL_WTICK(3);
#endif
  // membership of a point in the candidate region's defining sets (:254-277)
  const double diff = fmod(fabs(oy - pr.ego_yaw), 6.283185307179586);
  const bool wedge = 3.141592653589793 - RL_TOL_SAME_DIR <= diff && diff <= 3.141592653589793 + RL_TOL_SAME_DIR;
  const double oc_c = cos(oy), oc_s = sin(oy);
  // (the tests are a conjunction: cheapest first -- distance, the obstacle grown by 1 m, shadow / occluded class -- and
  // the lanelet polygons, the dear ones, last)
  // returns 0 (not a member) or 1 + the slot of a candidate polygon that holds the point; `hint`: the slot asked first
  int *const el = (int *)(red + 64);   // [<= RL_PVERT] edge_band's list
  bool el_on = false;
  int ftr = 0;   // (trace build 4: thread 0's first point of the first fit)
  (void)ftr;
  auto member_idx = [&](double x, double y, int hint) -> int {
    const double rx = x - cx, ry = y - cy;
    // sqrt(d2) <= 12 exactly when d2 <= 144: the midpoint between 12 and the next double squares to 144 + 2.1e-14, below the
    // double that follows 144 (144 + 2.8e-14) -- no square root needed
    static_assert(RL_BUFFER_SIDE == 12.0, "the squared form of the distance test is derived for 12 m");
    if (!(rx * rx + ry * ry <= 144.0)) return 0;
    const double lx_ = oc_c * rx + oc_s * ry, ly_ = -oc_s * rx + oc_c * ry;
    const double ex_ = fmax(fabs(lx_) - olen / 2.0, 0.0), ey_ = fmax(fabs(ly_) - owid / 2.0, 0.0);
    // minus the obstacle grown by 1 m: sqrt(e2) > 1 exactly when e2 > 1 + 2^-52 (sqrt(1 + 2^-52) = 1 + 2^-53 - ... rounds to 1)
    if (!(ex_ * ex_ + ey_ * ey_ > 1.0000000000000002)) return 0;
    RL_FTICK(4);
    if (wedge) {   // the obstacle's own shadow: the sight line ego -> point crosses the rectangle (:264)
      bool hit = false;
      const double dx = x - pr.ego_x, dy = y - pr.ego_y;
      for (int i = 0; i < 4 && !hit; ++i) {
        const int j = (i + 1) & 3;
        const double ex = s_oc[2 * j] - s_oc[2 * i], ey = s_oc[2 * j + 1] - s_oc[2 * i + 1];
        const double den = dx * ey - dy * ex, wx = s_oc[2 * i] - pr.ego_x, wy = s_oc[2 * i + 1] - pr.ego_y;
        if (fabs(den) > 1e-14) {
          // t = tn / den and u = un / den in [0, 1] without the divisions: a correctly rounded quotient is <= 1 exactly
          // when |tn| <= |den| and >= 0 exactly when the signs agree (or tn = 0)
          const double tn = wx * ey - wy * ex, un = wx * dy - wy * dx;
          hit = den > 0.0 ? (tn >= 0.0 && tn <= den && un >= 0.0 && un <= den) : (tn <= 0.0 && tn >= den && un <= 0.0 && un >= den);
        }
      }
      if (!hit) return 0;
    } else if (!(rl_class_at(v, x, y) & 4)) {   // the global occluded area (:272)
      return 0;
    }
    RL_FTICK(5);
    // possible_polygon (:255): the union of the candidate lanelet polygons -- any order of asking gives the same answer;
    // the polygon that held the nearest lattice node goes first (it holds most points around that node as well)
    if (el_on) {
      // ONE pass over the edges listed for the band the point lies in (edge_band below: only those can straddle its y), all
      // polygons at once: a crossing flips the bit of the edge's polygon -- the crossing number is a parity, the order of the
      // edges does not matter, each edge is tested by the arithmetic of the ring walk -- and a polygon with an odd count
      // holds the point if its bounding box does (rl_in_polygon asks the box first; kept, so that the answer is the ring
      // walk's in every rounding case).  The walk polygon by polygon was a chain of four dependent LDS round trips per
      // polygon, and a wave walks every polygon one of its lanes needs: 1.3 of the 2.2 us a wave spent per point.
      const double2 *pv2 = (const double2 *)polyv;
      const int n = s_ecnt[0];
      int par = 0;
      for (int h0 = 0; h0 < n; h0 += 4) {
        int en[4];
        double2 pi[4], pj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) en[u] = el[h0 + u < n ? h0 + u : n - 1];
#pragma unroll
        for (int u = 0; u < 4; ++u) { pi[u] = pv2[en[u] & 1023]; pj[u] = pv2[(en[u] >> 10) & 1023]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (h0 + u < n && (pi[u].y > y) != (pj[u].y > y) && rl_left_of_crossing(x, y, pi[u].x, pi[u].y, pj[u].x, pj[u].y)) par ^= 1 << (en[u] >> 20);
      }
      for (int i = 0; par != 0 && i < npol; ++i)
        if ((par >> i) & 1) {
          const double *bb = s_pbox + 4 * i;
          if (!(x < bb[0] || x > bb[2] || y < bb[1] || y > bb[3])) return i + 1;
        }
      return 0;
    }
    for (int q = 0; q < npol; ++q) {
      const int i = q == 0 ? hint : (q <= hint ? q - 1 : q);
      if (!plds) {
        if (rl_in_polygon(v, s_pol[i], x, y)) return i + 1;
        continue;
      }
      const double *bb = s_pbox + 4 * i;            // rl_in_polygon on the copy in LDS (the same arithmetic)
      if (x < bb[0] || x > bb[2] || y < bb[1] || y > bb[3]) continue;
      const double2 *pv2 = (const double2 *)polyv;
      if (rl_crossing_parity(s_poff[i], s_poff[i + 1], x, y, [&](int k) { return pv2[k]; }) != 0) return i + 1;
    }
    return 0;
  };
  // edge_band(ylo, yhi): the edges of the candidate polygons that some y in [ylo, yhi] can straddle (min(yi, yj) <= yhi and
  // max(yi, yj) > ylo: a straddled edge has min <= y < max) -- round 6.  A lanelet polygon has 50-100 vertices; a band of lattice
  // rows or a fit's rectangle is crossed by a handful of its edges.  An entry: the edge's vertex | its predecessor in the ring
  // << 10 | the polygon's slot << 20 (RL_PVERT = 1024 vertices in LDS).  The list borrows the tail of `red` (the lanelet
  // queries' hit list, idle here) and holds every edge if it must.  The count is zero on entry (cleared behind a barrier
  // after its last reader); ends with a barrier.
  static_assert(RL_PVERT <= 1024, "edge_band packs two vertex indices of ten bits");
  auto edge_band = [&](double ylo, double yhi) {
    const double2 *pv2 = (const double2 *)polyv;
    const int tot = s_poff[npol];
    for (int k = tid; k < tot; k += nth) {
      int i = 0;
      while (i + 1 < npol && k >= s_poff[i + 1]) ++i;
      const int kj = k == s_poff[i] ? s_poff[i + 1] - 1 : k - 1;
      const double yi = pv2[k].y, yj = pv2[kj].y;
      if (fmin(yi, yj) <= yhi && fmax(yi, yj) > ylo) el[atomicAdd(&s_ecnt[0], 1)] = k | (kj << 10) | (i << 20);
    }
    __syncthreads();
  };
  // the 0.25 m lattice around the obstacle; label = linear index where the node is a member, INT_MAX elsewhere
  const double h = 0.25;
  constexpr int NL = RL_LAT * RL_LAT;
  // RL_PARTS workgroups (on as many CUs) share the lattice: each decides its slice of the nodes -- membership is arithmetic,
  // ~300 float64 operations per node, and one CU's four SIMDs are the limit -- and writes it to the obstacle's lattice in
  // HBM; the workgroup that finishes LAST (a counter per obstacle) loads the whole lattice and goes on alone, the others
  // are done.  (Every workgroup took the same decisions up to here: they read the same inputs.)
  {
    const int chunk = (NL + RL_PARTS - 1) / RL_PARTS, i1 = min((part + 1) * chunk, NL);
    if (plds && part * chunk < i1) {   // the rows of this workgroup's slice (the nodes' y by the expression of the loop below: monotone in the row)
      edge_band(cy + (-RL_BUFFER_SIDE + (double)((part * chunk) / RL_LAT) * h), cy + (-RL_BUFFER_SIDE + (double)((i1 - 1) / RL_LAT) * h));
      el_on = true;
    }
    for (int i = part * chunk + tid; i < i1; i += nth) {
      const int ix = i % RL_LAT, iy = i / RL_LAT;
      const int mi = member_idx(cx + (-RL_BUFFER_SIDE + (double)ix * h), cy + (-RL_BUFFER_SIDE + (double)iy * h), 0);
      g_lab[i] = mi ? (i | ((mi - 1) << 16)) : 0x7fffffff;      // (+ which polygon held the node: the fits' hint)
    }
    el_on = false;
    __shared__ int s_ticket;
#if FO_RULE_TRACE < 3
    RL_WTICK(4);
#endif
    // Hand-off with ONE release and ONE acquire per workgroup (round 6).  The fences are whole-cache operations -- the release
    // writes the XCD's L2 back, the acquire invalidates the CU's L1 and the L2's non-local lines -- and sixteen waves issuing
    // them one after the other cost 5 us on the releasing and 3 us on the acquiring side (stamps of the trace build;
    // tools/microbench/grid_barrier.hip: the same finding for a grid barrier).  The workgroup barrier in front orders every
    // wave's stores before thread 0's release (its cumulativity carries them to agent scope), the one behind holds the
    // other waves' loads back until thread 0's acquire has been executed for the CU they share.
    __syncthreads();
    if (tid == 0) {
      __threadfence();
      s_ticket = atomicAdd(g_cnt, 1);
    }
    __syncthreads();
#if FO_RULE_TRACE < 3
    RL_WTICK(5);
#endif
    if (s_ticket != RL_PARTS - 1) return;
    if (tid == 0) {
      *g_cnt = 0;   // for the next planning step (launches on a stream are ordered)
      __threadfence();
    }
    if (tid == 64) s_ecnt[0] = 0;   // (edge_band's count: every reader is past the barriers above)
    __syncthreads();
    const volatile int *gl = g_lab;
    for (int i = tid; i < NL; i += nth) { const int w = gl[i]; lab[i] = w == 0x7fffffff ? w : (w & 0xffff); }
  }
  __syncthreads();
  RL_TICK(1);
  // connected parts (4-neighbourhood, scipy.ndimage.label's default), their sizes and the largest one (first maximum in label
  // order, :279-281).  Round 6: on the RUNS of the lattice rows (maximal stretches of member nodes in a row: a few per row, a
  // couple of hundred in all) instead of on its 9 409 nodes -- the sixteen waves cut the rows into runs with two ballots per
  // row, ONE wave then labels the runs by the same label equivalence as before (a run's label = its index, runs are numbered
  // row-major, so the smallest index of a part is the run that holds the part's smallest linear node index = scipy's numbering
  // order; runs of neighbouring rows touch where their column intervals overlap), adds up the run lengths per root and picks
  // the largest part; the nodes of that part are then stamped with its label.  A wave's LDS traffic is ordered: its rounds need
  // no workgroup barrier, where the node form paid three barriers of sixteen waves per round and two more passes over the
  // lattice for the sizes (7.8 + 8.5 us -> see DESIGN section 5).  More runs than the arrays hold: the node form below.
  constexpr int RL_MAXRUN = 3072;
  int *run_rec = ired, *run_lab = ired + RL_MAXRUN, *run_size = ired + 2 * RL_MAXRUN;   // (ired: 9 409 ints)
  __shared__ int s_rowoff[RL_LAT + 1], s_nrun;
  __shared__ unsigned long long s_rowbits[RL_LAT][2];
  bool by_runs = FO_RULE_RUNS != 0 && !pr.label_nodes;
  if (by_runs) {
    const int wave = tid >> 6, lane = tid & 63, nw = nth >> 6;
    for (int r = wave; r < RL_LAT; r += nw) {
      const unsigned long long b0 = __ballot(lab[r * RL_LAT + lane] != 0x7fffffff);
      const unsigned long long b1 = __ballot(lane < RL_LAT - 64 && lab[r * RL_LAT + 64 + (lane < RL_LAT - 64 ? lane : 0)] != 0x7fffffff);
      if (lane == 0) {
        const unsigned long long s0 = b0 & ~(b0 << 1), s1 = b1 & ~((b1 << 1) | (b0 >> 63));
        s_rowbits[r][0] = b0; s_rowbits[r][1] = b1;
        s_rowoff[r + 1] = __popcll(s0) + __popcll(s1);
      }
    }
    if (tid == 0) s_rowoff[0] = 0;
    __syncthreads();
    if (tid < 64) {   // inclusive prefix of the row counts (97 rows: two per lane)
      const int r0 = 2 * tid + 1, r1 = 2 * tid + 2;
      const int c0 = r0 <= RL_LAT ? s_rowoff[r0] : 0, c1 = r1 <= RL_LAT ? s_rowoff[r1] : 0;
      int incl = c0 + c1;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if (tid >= off) incl += t; }
      if (r0 <= RL_LAT) s_rowoff[r0] = incl - c1;
      if (r1 <= RL_LAT) s_rowoff[r1] = incl;
      if (tid == 63) s_nrun = incl;
    }
    __syncthreads();
    by_runs = s_nrun <= RL_MAXRUN;   // (uniform)
  }
  if (by_runs) {
    const int wave = tid >> 6, lane = tid & 63, nw = nth >> 6;
    const int NR = s_nrun;
    for (int r = wave; r < RL_LAT; r += nw) {   // the runs of row r: a lane per start column
      const unsigned long long b0 = s_rowbits[r][0], b1 = s_rowbits[r][1];
      const unsigned long long s0 = b0 & ~(b0 << 1), s1 = b1 & ~((b1 << 1) | (b0 >> 63));
      const int base = s_rowoff[r];
      if ((s0 >> lane) & 1ull) {
        const unsigned long long z = ~(b0 >> lane);          // first column past the run, relative to `lane`
        int len = z ? __builtin_ctzll(z) : 64;
        if (lane + len == 64) len += b1 == ~0ull ? 64 : __builtin_ctzll(~b1);   // (the run goes on in the second word)
        const int idx = base + __popcll(s0 & ((1ull << lane) - 1ull));
        run_rec[idx] = (r << 16) | (lane << 8) | (lane + len - 1);
        run_lab[idx] = idx;
        run_size[idx] = 0;
      }
      if ((s1 >> lane) & 1ull) {
        const int len = __builtin_ctzll(~(b1 >> lane));
        const int idx = base + __popcll(s0) + __popcll(s1 & ((1ull << lane) - 1ull));
        run_rec[idx] = (r << 16) | ((64 + lane) << 8) | (64 + lane + len - 1);
        run_lab[idx] = idx;
        run_size[idx] = 0;
      }
    }
    __syncthreads();
    if (tid < 64) {   // wave 0 alone: label equivalence on the runs, sizes, the largest part
      for (int round = 0; round < 4096; ++round) {
        bool ch = false;
        for (int i = lane; i < NR; i += 64) {
          const int rec_ = run_rec[i], r = rec_ >> 16, c0 = (rec_ >> 8) & 255, c1 = rec_ & 255;
          const int l = run_lab[i];
          int m = l;
          if (r > 0)
            for (int j = s_rowoff[r - 1]; j < s_rowoff[r]; ++j) {
              const int q = run_rec[j];
              if (((q >> 8) & 255) <= c1 && (q & 255) >= c0) m = min(m, run_lab[j]);
            }
          if (r + 1 < RL_LAT)
            for (int j = s_rowoff[r + 1]; j < s_rowoff[r + 2]; ++j) {
              const int q = run_rec[j];
              if (((q >> 8) & 255) <= c1 && (q & 255) >= c0) m = min(m, run_lab[j]);
            }
          if (m < l) { atomicMin(&run_lab[l], m); ch = true; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int i = lane; i < NR; i += 64) {
          int r0 = run_lab[i];
          while (true) {
            const int q = run_lab[r0];
            if (q == r0) break;
            r0 = q;
          }
          run_lab[i] = r0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (!__ballot(ch)) break;
      }
      for (int i = lane; i < NR; i += 64) {
        const int rec_ = run_rec[i];
        atomicAdd(&run_size[run_lab[i]], (rec_ & 255) - ((rec_ >> 8) & 255) + 1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // the largest part, the smallest label among equals: (size, -index) as one 64-bit key
      unsigned long long key = 0ull;
      for (int i = lane; i < NR; i += 64)
        if (run_lab[i] == i) {
          const unsigned long long k_ = ((unsigned long long)(unsigned)run_size[i] << 32) | (unsigned)(0x7fffffff - i);
          key = k_ > key ? k_ : key;
        }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o_ = __shfl_xor(key, off);
        key = o_ > key ? o_ : key;
      }
      if (tid == 0) {
        if (key >> 32) {
          const int bi = 0x7fffffff - (int)(unsigned)(key & 0xffffffffull), rec_ = run_rec[bi];
          s_bestn = (int)(key >> 32);
          s_best = (rec_ >> 16) * RL_LAT + ((rec_ >> 8) & 255);   // the part's smallest linear node index = its label
          s_changed = bi;                                          // (the root run, for the stamping below)
        } else { s_bestn = 0; s_best = -1; s_changed = -1; }
      }
    }
    __syncthreads();
    RL_TICK(2);
    {   // stamp the nodes of the largest part (every other member node keeps its own index, which is not the part's label)
      const int root = s_changed, bl = s_best;
      for (int i = tid; i < NR; i += nth)
        if (run_lab[i] == root) {
          const int rec_ = run_rec[i], r = rec_ >> 16;
          for (int c = (rec_ >> 8) & 255; c <= (rec_ & 255); ++c) lab[r * RL_LAT + c] = bl;
        }
    }
    __syncthreads();
  } else {
  // connected parts (4-neighbourhood, scipy.ndimage.label's default) by label equivalence (Hawick et al.): every member node
  // starts as its own root (label = linear index); a round links the root of every node whose neighbourhood holds a smaller
  // label to that label (atomicMin), then flattens every node to its root by pointer jumping; labels only ever decrease
  // and stay inside their part, so every part ends up carrying its smallest linear index (= scipy's numbering order) after
  // a handful of rounds, whatever its shape -- and every thread of the workgroup works in every round
  for (int round = 0; round < 256; ++round) {
    if (tid == 0) s_changed = 0;
    __syncthreads();
    bool ch = false;
    for (int i = tid; i < NL; i += nth) {
      const int l = lab[i];
      if (l == 0x7fffffff) continue;
      const int ix = i % RL_LAT, iy = i / RL_LAT;
      int m = l;
      if (ix > 0) m = min(m, lab[i - 1]);
      if (ix + 1 < RL_LAT) m = min(m, lab[i + 1]);
      if (iy > 0) m = min(m, lab[i - RL_LAT]);
      if (iy + 1 < RL_LAT) m = min(m, lab[i + RL_LAT]);
      if (m < l) { atomicMin(&lab[l], m); ch = true; }
    }
    if (ch) s_changed = 1;
    __syncthreads();
    for (int i = tid; i < NL; i += nth) {
      int r = lab[i];
      if (r == 0x7fffffff) continue;
      while (true) {
        const int q = lab[r];
        if (q == r) break;
        r = q;
      }
      lab[i] = r;
    }
    __syncthreads();
    if (!s_changed) break;
    __syncthreads();
  }
  RL_TICK(2);
  // the largest part (first maximum in label order, :279-281): sizes by the roots' labels.  A thread counts a contiguous
  // stretch of nodes and adds a run of equal labels with one atomic (neighbours in a row mostly share their part)
  if (tid == 0) { s_best = -1; s_bestn = 0; }
  for (int i = tid; i < NL; i += nth) ired[i] = 0;
  __syncthreads();
  {
    const int per = (NL + nth - 1) / nth, i0 = tid * per, i1 = min(i0 + per, NL);
    int run_l = 0x7fffffff, run_n = 0;
    for (int i = i0; i < i1; ++i) {
      const int l = lab[i];
      if (l == run_l) { ++run_n; continue; }
      if (run_n > 0 && run_l != 0x7fffffff) atomicAdd(&ired[run_l], run_n);
      run_l = l; run_n = 1;
    }
    if (run_n > 0 && run_l != 0x7fffffff) atomicAdd(&ired[run_l], run_n);
  }
  __syncthreads();
  {   // first maximum in label order: per-thread (count, smallest label), then thread 0 over the partials
    int bn = 0, bi = -1;
    for (int i = tid; i < NL; i += nth)
      if (ired[i] > bn || (ired[i] == bn && bn > 0 && i < bi)) { bn = ired[i]; bi = i; }
    red[2 * tid] = (double)bn; red[2 * tid + 1] = (double)bi;
    __syncthreads();
    // (two levels: 32 threads fold nth / 32 partials each, thread 0 folds those -- the order is fixed, the rule associative)
    const int grp = nth / 32;
    if (tid < 32) {
      int gn = 0, gl = -1;
      for (int i = tid * grp; i < (tid + 1) * grp; ++i) {
        const int n_ = (int)red[2 * i], l_ = (int)red[2 * i + 1];
        if (n_ > gn || (n_ == gn && n_ > 0 && l_ < gl)) { gn = n_; gl = l_; }
      }
      red[2 * tid * grp] = (double)gn; red[2 * tid * grp + 1] = (double)gl;
    }
    __syncthreads();
    if (tid == 0)
      for (int t = 0; t < 32; ++t) {
        const int n_ = (int)red[2 * t * grp], l_ = (int)red[2 * t * grp + 1];
        if (n_ > s_bestn || (n_ == s_bestn && n_ > 0 && l_ < s_best)) { s_bestn = n_; s_best = l_; }
      }
    __syncthreads();
  }
  }
  const int best = s_best;
  if (best < 0 || (double)s_bestn * h * h < RL_MIN_AREA) return;                  // :282-284
  RL_TICK(3);
  // (the polygon slot that held each lattice node -- the hint of member_idx, for the fits -- is asked for here, under the
  // centroid's arithmetic, and parked in `ired`, which is free from here on)
  int hint_w[(RL_LAT * RL_LAT + RL_THREADS_DYN - 1) / RL_THREADS_DYN];
  {
    const volatile int *gl = g_lab;
#pragma unroll
    for (int u = 0; u < (RL_LAT * RL_LAT + RL_THREADS_DYN - 1) / RL_THREADS_DYN; ++u) {
      const int i = tid + u * RL_THREADS_DYN;
      hint_w[u] = i < NL ? gl[i] : 0x7fffffff;
    }
  }
  // centroid of the part (mean of its nodes; fixed summation order: per-thread partials, then thread 0)
  double ax = 0.0, ay = 0.0;
  for (int i = tid; i < NL; i += nth)
    if (lab[i] == best) { ax += cx + (-RL_BUFFER_SIDE + (double)(i % RL_LAT) * h); ay += cy + (-RL_BUFFER_SIDE + (double)(i / RL_LAT) * h); }
  // fixed summation order: per thread, a butterfly over the wave (every lane ends with the same total), the waves in order
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { ax += __shfl_xor(ax, off); ay += __shfl_xor(ay, off); }
  if ((tid & 63) == 0) { red[2 * (tid >> 6)] = ax; red[2 * (tid >> 6) + 1] = ay; }
#pragma unroll
  for (int u = 0; u < (RL_LAT * RL_LAT + RL_THREADS_DYN - 1) / RL_THREADS_DYN; ++u) {
    const int i = tid + u * RL_THREADS_DYN;
    if (i < NL) ired[i] = hint_w[u] == 0x7fffffff ? 0 : (hint_w[u] >> 16);
  }
  __syncthreads();
  if (tid == 0) {
    double sx_ = 0.0, sy_ = 0.0;
    for (int w = 0; w < nth / 64; ++w) { sx_ += red[2 * w]; sy_ += red[2 * w + 1]; }
    s_c[0] = sx_ / (double)s_bestn; s_c[1] = sy_ / (double)s_bestn;
  }
  __syncthreads();
  auto in_region = [&](double x, double y) {
    const int ix = (int)rint((x - (cx - RL_BUFFER_SIDE)) / h), iy = (int)rint((y - (cy - RL_BUFFER_SIDE)) / h);
    if (ix < 0 || ix >= RL_LAT || iy < 0 || iy >= RL_LAT) return false;
    const int l_ = lab[iy * RL_LAT + ix], hn_ = ired[iy * RL_LAT + ix];
    RL_FTICK(3);
    return l_ == best && member_idx(x, y, hn_) != 0;
  };
  // three conditions, independent of each other (:287-300): the centroid on a relevant lanelet -- every lanelet asked at once
  // --, no region in front of the obstacle, a lane heading at the centroid.  Round 6: the last two are taken by the LAST thread
  // of the workgroup (another wave) while the others ask the lanelets, instead of by thread 0 behind them (a membership test
  // and a raster look-up by one thread: ~3 us of the chain)
  __shared__ int s_front, s_yawok;
  if (tid == nth - 1) {
    s_front = in_region(cx + 4.0 * oc_c, cy + 4.0 * oc_s) ? 1 : 0;                  // the region in front of the obstacle (:297)
    double yw = 0.0;
    s_yawok = rl_lane_yaw_at(v, s_c[0], s_c[1], yw) ? 1 : 0;
    s_yaw = yw;
  }
  if (rel_fits) {   // the centroid must lie on a relevant lanelet (:287-291): every relevant lanelet asked at once
    // (s_nhit: zero since the set-up's last query -- the hand-off and four barriers lie between; the list borrows the tail of `red`)
    rl_which_lanelets(v, 1, [&](int, double &x, double &y) { x = s_c[0]; y = s_c[1]; }, [&](int, int p) { return (relflag[p] & 1) != 0; },
                      [&](int, int) { s_relc = 1; }, (int *)(red + 64), 2048, &s_nhit);
  } else {
    __syncthreads();
  }
  if (tid == 0) {
    s_go = 0;
    do {
      // the centroid must lie on a relevant lanelet (:287-291)
      bool rel_c = rel_fits && s_relc;
      for (int p = 0; !rel_fits && p < v.P && !rel_c; ++p)
        if (rl_in_polygon(v, p, s_c[0], s_c[1])) {
          if (s_inter >= 0) {
            if (p != s_ego_ll)
              for (int e = v.inter_off[s_inter]; e < v.inter_off[s_inter + 1]; ++e)
                if (v.inter_lanelet[e] == p) rel_c = true;
          } else {
            for (int i = pr.win_i0; i < pr.win_i1 && !rel_c; i += 5) {
              const double *q = v.path + 6 * (size_t)i;
              const int ll = rl_lanelet_of(v, q[0], q[1]);
              if (ll >= 0 && v.adj_left && v.adj_left[ll] == p) rel_c = true;
            }
          }
        }
      if (!rel_c) break;
      if (s_front) break;
      if (!s_yawok) break;
      s_go = 1;
    } while (false);
  }
  __syncthreads();
  if (!s_go) return;
  RL_TICK(4);
  // rectangle fits on a 0.1 m lattice (:695-726): lane-aligned rectangle clipped to the region -> area, centroid, Jaccard
  // similarity with the minimum rotated rectangle of the clipped part
  int fitno = 0;
  (void)fitno;
  const double fc = cos(s_yaw), fs = sin(s_yaw);
  __shared__ double s_fit[4];   // area, cx, cy, jaccard
  __shared__ int s_fitany;
  __shared__ int s_a0[32], s_a1[32], s_nv, s_np2, s_pr[64], s_pc[64];
  __shared__ unsigned long long s_bestA;
  auto fit = [&](auto nx_c, auto ny_c, double ccx, double ccy, double length, double width) {
    const double fh = 0.1;
    // (rint(length / fh), rint(width / fh) as constants: the index split below is a multiplication instead of two divisions per point)
    constexpr int nx_ = decltype(nx_c)::value, ny_ = decltype(ny_c)::value, np_ = nx_ * ny_;   // (ny_ <= 32)
    int cnt = 0;
    double fx = 0.0, fy = 0.0;
    if (plds) {   // the rectangle's extent in y (+ a micrometre for the roundings of the points' expression below)
      const double ext = fabs(fs) * (length / 2.0) + fabs(fc) * (width / 2.0) + 1e-6;
      edge_band(ccy - ext, ccy + ext);
      el_on = true;
    }
    for (int i = tid; i < np_; i += nth) {
      const double u = ((double)(i % nx_) + 0.5) * fh - length / 2.0, w_ = ((double)(i / nx_) + 0.5) * fh - width / 2.0;
      const double x = ccx + fc * u - fs * w_, y = ccy + fs * u + fc * w_;
#if FO_RULE_TRACE == 4
      ftr = tid < 64 && fitno == 0 && i < 64;
#endif
      RL_FTICK(2);
      const bool ok = in_region(x, y);
      fitok[i] = ok ? 1 : 0;
      RL_FTICK(6);
      if (ok) { ++cnt; fx += x; fy += y; }
      RL_FTICK(7);
#if FO_RULE_TRACE == 4
      ftr = 0;
#endif
    }
    el_on = false;
    double fn = (double)cnt;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { fx += __shfl_xor(fx, off); fy += __shfl_xor(fy, off); fn += __shfl_xor(fn, off); }
    if ((tid & 63) == 0) { red[3 * (tid >> 6)] = fx; red[3 * (tid >> 6) + 1] = fy; red[3 * (tid >> 6) + 2] = fn; }
    __syncthreads();
    RL_TICKF(5);
    if (tid == 128) s_ecnt[0] = 0;   // (edge_band's count, read by the clipping above; barriers follow)
    // the clipped part's convex hull needs only the first and last clipped point of every lattice row (the rest of a row
    // lies between them): a thread per row finds them while thread 0 adds up the partial sums
    if (tid >= 64 && tid < 64 + ny_) {
      const int r = tid - 64;
      int a0 = -1, a1 = -1;
      for (int c = 0; c < nx_; ++c)
        if (fitok[r * nx_ + c]) { if (a0 < 0) a0 = c; a1 = c; }
      s_a0[r] = a0; s_a1[r] = a1;
    }
    if (tid == 0) {
      double sx_ = 0.0, sy_ = 0.0, n = 0.0;
      for (int w = 0; w < nth / 64; ++w) { sx_ += red[3 * w]; sy_ += red[3 * w + 1]; n += red[3 * w + 2]; }
      s_fitany = n > 0.0;
      s_fit[0] = n * fh * fh; s_fit[1] = n > 0.0 ? sx_ / n : 0.0; s_fit[2] = n > 0.0 ? sy_ / n : 0.0;
      s_fit[3] = ((int)n == np_) ? 1.0 : 0.0;
    }
    __syncthreads();
    RL_TICKF(6);
    if (!s_fitany || s_fit[3] == 1.0) return;   // nothing clipped in, or nothing clipped off (Jaccard 1)
    // The smallest rectangle over the edge directions of the clipped part's convex hull (:716-724), without building the
    // hull: the enclosing rectangle of smallest area has a side along a hull edge (Freeman & Shapira), so the minimum over
    // the directions of ALL point pairs is the minimum over the hull's edge directions -- a superset of directions cannot
    // undercut the global optimum, and it contains the hull's.  <= 64 points (first and last clipped point of every lattice
    // row, integer lattice coordinates), <= 2 016 pairs over the workgroup, each spanning the extents of all points; the
    // smallest area is kept by atomicMin on its bit pattern (positive doubles order like their bits).  Fewer than three
    // points, or all on one line (QHull raises there): Jaccard 0.
    if (tid < 64) {
      // the candidate list (wave 0: a lane per lattice row): first and last clipped point of the row, but only where the left
      // (first points) or right (last points) chain turns strictly outwards against its neighbours in the rows below and
      // above -- every vertex of the convex hull does; points on straight stretches (most: the part is a clipped
      // rectangle) and in dents do not, they neither span an extent nor define a hull edge
      const int r = tid;
      const int a0 = r < ny_ ? s_a0[r] : -1, a1 = r < ny_ ? s_a1[r] : -1;
      int rp = -1, rn = -1;
      if (a0 >= 0) {
        for (int q = r - 1; q >= 0 && rp < 0; --q) if (s_a0[q] >= 0) rp = q;
        for (int q = r + 1; q < ny_ && rn < 0; ++q) if (s_a0[q] >= 0) rn = q;
      }
      bool k0 = a0 >= 0, k1 = a0 >= 0 && a1 != a0;
      if (a0 >= 0 && rp >= 0 && rn >= 0) {
        const int zl = (r - rp) * (s_a0[rn] - a0) - (a0 - s_a0[rp]) * (rn - r);   // > 0: the left chain bulges to smaller columns here
        const int zr = (r - rp) * (s_a1[rn] - a1) - (a1 - s_a1[rp]) * (rn - r);   // < 0: the right chain bulges to larger columns
        if (a1 != a0) { k0 = zl > 0; k1 = zr < 0; }
        else k0 = zl > 0 || zr < 0;
      }
      const int cnt = (k0 ? 1 : 0) + (k1 ? 1 : 0);
      int incl = cnt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if (tid >= off) incl += t; }
      int pos = incl - cnt;
      if (k0) { s_pr[pos] = r; s_pc[pos] = a0; ++pos; }
      if (k1) { s_pr[pos] = r; s_pc[pos] = a1; }
      if (tid == 63) s_np2 = incl;
      // a proper polygon?  fewer than three clipped points, or all of them on one line (QHull raises there): Jaccard 0
      const int n_all = __popcll(__ballot(a0 >= 0)) + __popcll(__ballot(a0 >= 0 && a1 != a0));
      const int rf = __ffsll((long long)__ballot(a0 >= 0)) - 1;      // first row that holds a point
      bool off = false;
      if (rf >= 0 && a0 >= 0) {
        const int c0 = s_a0[rf];
        const int rl = 63 - __clzll((long long)__ballot(a0 >= 0));    // last such row; the line through (rf, c0) and (rl, its last point)
        const int dc = s_a1[rl] - c0, dr = rl - rf;
        off = (dr * (a0 - c0) - dc * (r - rf) != 0) || (dr * (a1 - c0) - dc * (r - rf) != 0);
      }
      const bool proper = n_all >= 3 && __ballot(off) != 0;
      if (tid == 0) { s_nv = proper ? 3 : 0; s_bestA = 0x7ff0000000000000ull; }   // +inf
    }
    __syncthreads();
    const int np2 = s_np2;
    for (int w = tid; w < np2 * np2; w += nth) {
      const int i = w / np2, j = w % np2;
      if (i >= j) continue;
      double ex = (double)(s_pc[j] - s_pc[i]), ey = (double)(s_pr[j] - s_pr[i]);
      const double nn = sqrt(ex * ex + ey * ey);
      ex /= nn; ey /= nn;
      double a1n = INFINITY, a1x = -INFINITY, a2n = INFINITY, a2x = -INFINITY;
      for (int q = 0; q < np2; ++q) {
        const double p1 = (double)s_pc[q] * ex + (double)s_pr[q] * ey, p2 = (double)s_pc[q] * (-ey) + (double)s_pr[q] * ex;
        a1n = fmin(a1n, p1); a1x = fmax(a1x, p1); a2n = fmin(a2n, p2); a2x = fmax(a2x, p2);
      }
      const double area = ((a1x - a1n) * fh + fh) * ((a2x - a2n) * fh + fh);
      atomicMin(&s_bestA, (unsigned long long)__double_as_longlong(area));
    }
    __syncthreads();
    RL_TICKF(7);
    if (s_nv < 3) return;   // degenerate (QHull raises): Jaccard 0
    if (tid == 0) s_fit[3] = fmin(1.0, s_fit[0] / __longlong_as_double((long long)s_bestA));
    __syncthreads();
  };
  fit(std::integral_constant<int, 55>{}, std::integral_constant<int, 25>{}, s_c[0], s_c[1], 5.5, 2.5);
  ++fitno;
#if FO_RULE_TRACE != 2
  RL_TICK(5);
#endif
  if (!s_fitany) return;
  const double car_a = s_fit[0], car_x = s_fit[1], car_y = s_fit[2], car_j = s_fit[3];
  __syncthreads();
  fit(std::integral_constant<int, 20>{}, std::integral_constant<int, 10>{}, car_x, car_y, 2.0, 1.0);
#if FO_RULE_TRACE != 2
  RL_TICK(6);
#endif
  if (tid == 0) {
    if (car_a >= RL_AREA_CAR && car_j > 0.98) { rec[2] = 1.0; rec[3] = car_x; rec[4] = car_y; }
    if (s_fitany && s_fit[0] >= RL_AREA_BIKE && s_fit[3] > 0.98) { rec[5] = 1.0; rec[6] = s_fit[1]; rec[7] = s_fit[2]; }
  }
}

// flags of an obstacle at this step: bit0 present, bit1 occludes (not a bicycle), bit2 dynamic role, bit3 type bicycle or
// pedestrian (never triggers the dynamic rule, :209-210)
constexpr int RL_THREADS = 1024;   // the dynamic rule's lattice work spreads over sixteen waves (the other rules use one)
static_assert(RL_THREADS == RL_THREADS_DYN, "rl_dynamic_rule sizes its per-thread lattice slices for the kernel's block");
__global__ __launch_bounds__(RL_THREADS) void fo_spawn_rules_kernel(RuleView v, RuleParams pr, int O, const double *__restrict__ ocorn,
                                                             const double *__restrict__ ocen, const double *__restrict__ oyaw,
                                                             const double *__restrict__ odims, const uint8_t *__restrict__ oflags,
                                                             const uint8_t *__restrict__ ovis, double *__restrict__ recs,
                                                             int *__restrict__ g_lab, int *__restrict__ g_cnt, int all_obstacles, int n_helped) {
  // one LDS arena, carved per rule (the dynamic rule needs the two lattice arrays: 2 x 37.6 KB)
  __shared__ int lab[RL_LAT * RL_LAT];
  __shared__ int ired[RL_LAT * RL_LAT];
  __shared__ double red[3 * RL_THREADS];
  __shared__ unsigned char bytes[2048];
  __shared__ __align__(16) double polyv[2 * RL_PVERT];   // dynamic rule: the vertices of the candidate region's lanelet polygons (read as double2)
  // the reference path table into LDS: the projections and the arc-length searches of every rule are chains of dependent
  // reads of it (a binary search in HBM costs eight round trips of ~0.6 us; in LDS, of ~30 ns)
  __shared__ double pathv[6 * RL_PATHV];
  RL_WTICK(0);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // Only the dynamic rule needs the whole workgroup.  The others run on its first wave or two, which do not wait for the other
  // fourteen to be launched (sixteen waves of this size arrive over ~6 us): every wave that has nothing to do leaves at once,
  // the working waves copy the path table into LDS for themselves and meet no workgroup barrier.
  auto own_path = [&](double *dst) {   // this wave's copy of the path table (a wave's LDS accesses are ordered)
    if (v.n_path <= RL_PATHV) {
      for (int i = lane; i < 6 * v.n_path; i += 64) dst[i] = v.path[i];
      v.path = dst;
    }
  };
  if (blockIdx.x == 0) {   // the turn rule's record
    if (wave > 0) return;
    double *rec = recs;
    for (int i = lane; i < RL_REC; i += 64) rec[i] = 0.0;
    if (pr.behind_turn && pr.intention != 0) {
      own_path(pathv);
      static_assert(3 * RL_TURNW * sizeof(double) <= sizeof(lab), "the turn rule's three window arrays borrow the lattice array");
      double *lx = (double *)lab, *ly = lx + RL_TURNW, *cum = ly + RL_TURNW;
      rl_turn_rule(v, pr, rec, lx, ly, cum, bytes);
    }
    return;
  }
  // blocks 1 .. O: an obstacle each (its record, the static rule, part 0 of the dynamic rule's lattice); blocks beyond: the
  // other RL_PARTS - 1 parts of the dynamic rule's lattice -- of the c-th obstacle whose HOST-known flags allow the rule at
  // all (present, dynamic role, no bicycle / pedestrian), c = (b - 1 - O) / (RL_PARTS - 1): the caller says how many there
  // are (fo_spawn_rule_params_t::n_dynamic_plus1), and the launch dispatches helper workgroups -- sixteen waves and 144 KB of
  // LDS each, a CU apiece -- for those only instead of for every obstacle (scenario 1: 23 workgroups instead of 113; whether
  // such an obstacle is visible and the rule applies stays a decision of the device).  `all_obstacles`: the caller did not say.
  const bool helper = (int)blockIdx.x > O;
  int o = (int)blockIdx.x - 1;
  const int part = helper ? 1 + ((int)blockIdx.x - 1 - O) % (RL_PARTS - 1) : 0;
  if (helper) {
    int want = ((int)blockIdx.x - 1 - O) / (RL_PARTS - 1);
    if (all_obstacles) o = want;
    else {
      o = -1;
      for (int i = 0; i < O; ++i)
        if ((oflags[i] & 13) == 5 && want-- == 0) { o = i; break; }
      if (o < 0) return;      // (the caller counted more candidates than the flags hold: nothing to do)
    }
  }
  double *rec = recs + (size_t)(1 + o) * RL_REC;
  const bool vis = (oflags[o] & 1) && ovis[o];
  // (one call site for the dynamic rule, inlined: a call would put the kernel's RuleView on a stack in scratch memory -- and a
  // kernel with a private segment is dispatched noticeably later than one without, measured ~11 us here)
  bool dyn_rule = vis && (oflags[o] & 4) && !(oflags[o] & 8) && pr.behind_dynamic && (pr.intention == 0 || pr.intention == 1);
  if (dyn_rule && !helper && !all_obstacles) {
    // (a caller that counted fewer candidates than its flags hold: an obstacle beyond the count has no helper workgroups --
    // its lattice would never be handed over -- and is treated like a dynamic obstacle the rule does not apply to)
    int rank = 0;
    for (int i = 0; i < o; ++i) rank += (oflags[i] & 13) == 5;
    if (rank >= n_helped) dyn_rule = false;
  }
  if (helper && !dyn_rule) return;
  if (!dyn_rule) {
    // the obstacle's own workgroup without the dynamic rule: wave 0 keeps the record's head and the first cross line of the
    // static rule, wave 1 the second (helper workgroups of an obstacle WITHOUT the dynamic rule have returned above; with it,
    // every part clears the two validity words rec[2] / rec[5] in rl_dynamic_rule before its hand-off ticket -- the same value
    // from sixteen writers, ordered before the last part's results by the fence in front of the ticket; the selection kernel
    // is the next launch)
    if (wave >= 2) return;
    for (int i = lane; i < RL_REC; i += 64)
      if ((i >= 8 && i < 14) == (wave == 1)) rec[i] = 0.0;
    if (wave == 0 && lane == 0) {
      const double dx = pr.ego_x - ocen[2 * o], dy = pr.ego_y - ocen[2 * o + 1];
      rec[0] = sqrt(dx * dx + dy * dy);
    }
    if (!vis) return;
    if (oflags[o] & 4) {                                       // a dynamic obstacle the rule does not apply to
      if (!(oflags[o] & 8) && wave == 0 && lane == 0) rec[1] = 2.0;   // (bicycles and pedestrians, :209-210: no role)
      return;
    }
    if (wave == 0 && lane == 0) rec[1] = 1.0;
    if (pr.behind_static) {
      own_path(wave == 0 ? pathv : red);
      rl_static_rule(v, pr, o, O, ocorn, ocen, oflags, ovis, rec, (double *)lab + 2 * RL_MAXSAMP * wave, (double *)lab + 2 * RL_MAXSAMP * wave + RL_MAXSAMP,
                     bytes + RL_MAXSAMP * wave, wave);
    }
    return;
  }
  // the dynamic rule (straight ahead or left turn, :124-126): all sixteen waves, sixteen workgroups per obstacle
  if (v.n_path <= RL_PATHV) {
    for (int i = threadIdx.x; i < 6 * v.n_path; i += blockDim.x) pathv[i] = v.path[i];
    v.path = pathv;
  }
  if (!helper) {
    // (no fence behind the clear: the only other writer of the record is the rule's last part, which takes its ticket after this
    // workgroup has released its own -- rl_dynamic_rule fences before the ticket)
    if (threadIdx.x < RL_REC) rec[threadIdx.x] = 0.0;
  }
  __syncthreads();
  RL_WTICK(1);
  if (!helper && threadIdx.x == 0) {
    const double dx = pr.ego_x - ocen[2 * o], dy = pr.ego_y - ocen[2 * o + 1];
    rec[0] = sqrt(dx * dx + dy * dy);
    rec[1] = 2.0;
  }
  rl_dynamic_rule(v, pr, o, ocorn, ocen, oyaw, odims, rec, lab, red, ired, bytes, polyv, part, g_lab + (size_t)o * (RL_LAT * RL_LAT), g_cnt + o);
}

// what depends on the order of the obstacles: both lists sorted by distance (stable), the maxima of the YAML compared
// with '>' BEFORE appending (Q11), 5 m in s between pedestrians; output order = the reference's (dynamic, static, turn).
// out [max_out][8]: type code, x, y, yaw (NaN = none), s, d (NaN = none), source code, obstacle index (-1 = none)
__global__ void fo_spawn_rules_select_kernel(RuleView v, RuleParams pr, int O, const double *__restrict__ ocorn,
                                             const uint8_t *__restrict__ oflags, const uint8_t *__restrict__ ovis,
                                             const double *__restrict__ recs, int max_out, double *__restrict__ out,
                                             int32_t *__restrict__ n_out) {
  // (round 6, measured and dropped: the records and flags copied into LDS by the wave, the deciding thread reading them there
  // instead of in HBM -- the rules step does not move, same box, two passes: 0.0845 / 0.0831 against 0.0830 / 0.0850 ms)
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  // a rule that ran out of table space left -1 in its validity word: the step's list would be short of a point the reference
  // finds, so there is NO list -- the count says -1 and the callers refuse it (fo_hip.h)
  bool over = pr.behind_turn && pr.intention != 0 && recs[0] < 0.0;
  for (int o = 0; o < O; ++o) {   // (no short cuts: the loads of all records go out together)
    const double *r = recs + (size_t)(1 + o) * RL_REC;
    const double role = r[1], v0 = r[2], v1 = r[8];
    over = over | ((role == 2.0) & (v0 < 0.0)) | ((role == 1.0) & ((v0 < 0.0) | (v1 < 0.0)));
  }
  if (over) { *n_out = -1; return; }
  int n = 0;
  auto put = [&](double type, double x, double y, double yaw, double s, double d, double src, double ob) {
    if (n < max_out) {
      double *q = out + 8 * (size_t)n;
      q[0] = type; q[1] = x; q[2] = y; q[3] = yaw; q[4] = s; q[5] = d; q[6] = src; q[7] = ob;
    }
    ++n;
  };
  // visit the obstacles of one role in ascending distance, ties in list order (a stable sort)
  auto next_by_distance = [&](double role, double last_d, int last_o) {
    int best = -1;
    double bd = INFINITY;
    for (int o = 0; o < O; ++o) {
      const double *r = recs + (size_t)(1 + o) * RL_REC;
      if (r[1] != role) continue;
      const bool after = r[0] > last_d || (r[0] == last_d && o > last_o);
      if (after && (r[0] < bd)) { bd = r[0]; best = o; }
    }
    return best;
  };
  if (pr.behind_dynamic && (pr.intention == 0 || pr.intention == 1)) {
    int n_dyn = 0, last_o = -1;
    double last_d = -1.0;
    for (;;) {
      const int o = next_by_distance(2.0, last_d, last_o);
      if (o < 0) break;
      const double *r = recs + (size_t)(1 + o) * RL_REC;
      last_d = r[0]; last_o = o;
      if (n_dyn > pr.max_dynamic) break;                                           // :212
      if (r[2] != 0.0) { put(RL_TYPE_CAR, r[3], r[4], NAN, NAN, NAN, RL_SRC_DYNAMIC, o); ++n_dyn; }
      if (r[5] != 0.0) { put(RL_TYPE_BICYCLE, r[6], r[7], NAN, NAN, NAN, RL_SRC_DYNAMIC, o); ++n_dyn; }
    }
  }
  if (pr.behind_static) {
    double s_pos[16];
    int n_st = 0, last_o = -1;
    double last_d = -1.0;
    for (;;) {
      const int o = next_by_distance(1.0, last_d, last_o);
      if (o < 0) break;
      const double *r = recs + (size_t)(1 + o) * RL_REC;
      last_d = r[0]; last_o = o;
      if (n_st > pr.max_static) break;                                             // :365
      for (int li = 0; li < 2; ++li) {
        const double *q = r + 2 + 6 * li;
        if (q[0] == 0.0) continue;
        bool close = false;
        for (int i = 0; i < n_st && i < 16; ++i) close = close || fabs(s_pos[i] - q[3]) <= RL_MIN_DIST_PED;   // :453
        if (close) continue;
        put(RL_TYPE_PED, q[1], q[2], q[5], q[3], q[4], RL_SRC_STATIC, o);
        if (n_st < 16) s_pos[n_st] = q[3];
        ++n_st;
        break;                                                                     // one per obstacle
      }
    }
  }
  if (pr.behind_turn && pr.intention != 0) {
    const double *r = recs;
    if (r[0] != 0.0) {
      bool ok = true;
      for (int o = 0; o < O && ok; ++o)                                            // :557: no visible obstacle within the 0.5 m disc
        if ((oflags[o] & 1) && ovis[o] && rl_seg_rect_distance(r[1], r[2], r[1], r[2], ocorn + 8 * (size_t)o) <= 0.5) ok = false;
      double ye, yp;
      if (ok && (!rl_lane_yaw_at(v, pr.ego_x, pr.ego_y, ye) || !rl_lane_yaw_at(v, r[1], r[2], yp))) ok = false;
      if (ok) {                                                                    // :561-572: the lanelet there must head elsewhere (>= 45 deg)
        double m = fmod(fabs(yp - ye), 6.283185307179586);
        if (m < 45.0 / 180.0 * 3.141592653589793) ok = false;
      }
      if (ok) put(RL_TYPE_PED, r[1], r[2], NAN, r[3], r[4], r[5], -1.0);
    }
  }
#if FO_RULE_TRACE
  for (int o = 0; o < O; ++o)
    if (recs[(size_t)(1 + o) * RL_REC + 1] == 2.0 && recs[(size_t)(1 + o) * RL_REC + 16] != 0.0)
      for (int i = 0; i < 8; ++i) out[8 * (size_t)(max_out - 1) + i] = recs[(size_t)(1 + o) * RL_REC + 16 + i];
#endif
  *n_out = n < max_out ? n : max_out;
}


// ---------------------------------------------------------------- rule points -> phantom agents, on the device
// Replaces the loop of FOInterface.evaluate_scenario over the spawn points (interface.py:186-198) with
// FOAgentManager.add_agent (agent.py:46-141) behind it: one wave per prediction slot (point i, route r).  Reads the records
// fo_spawn_rules_select_kernel wrote -- type, x, y, orientation (NaN = derive), s, d, source, obstacle -- in HBM.
struct RuleAgentTypes { double speed[3], raw_l[3], raw_w[3], infl_l[3], infl_w[3]; };   // 0 Car, 1 Bicycle, 2 Pedestrian

__global__ __launch_bounds__(64) void fo_spawn_rule_predict_kernel(
    RuleView v, int max_points, const double *__restrict__ points, const int32_t *__restrict__ n_points, int R, RuleAgentTypes ty,
    int n_path, const double *__restrict__ path, const int32_t *__restrict__ center_off, const double *__restrict__ center_xy,
    RouteView rv, int T, double dt, double var0, double factor, int slot0, int agent0, double *__restrict__ pos0,
    double *__restrict__ yaw0, PredOut o, int table_on, fo_agent_table_t at) {
  const int lane = threadIdx.x;
  const int i = blockIdx.x / R, r = blockIdx.x % R, slot = slot0 + blockIdx.x;
  const int n = min(max(*n_points, 0), max_points);
  const double vpow = pow(factor, (double)lane);   // (spawn_write_slot; here: under the first round trip)
  const bool on = i < n;
  const double *rec = points + 8 * (size_t)i;
  const int type = on ? (int)rec[0] : RL_TYPE_PED;
  const int ti = type == RL_TYPE_CAR ? 0 : type == RL_TYPE_BICYCLE ? 1 : 2;
  const double px = on ? rec[1] : 0.0, py = on ? rec[2] : 0.0;
  const int src = on ? (int)rec[6] : 0;
  double a0 = on ? rec[3] : 0.0;
  int ll = -1;
  if (on && ti == 2) {                                           // OAPPedestrianAgent._create_ped_trajectory (agent.py:451-481)
    if (a0 != a0) {
      const double *curve = path;
      int nc = n_path;
      if ((src == RL_SRC_LEFT || src == RL_SRC_RIGHT) && center_off) {       // mode 'lane_center' (interface.py:194)
        const int lc = rl_lanelet_of_wave(v, px, py);
        if (lc >= 0 && center_off[lc + 1] - center_off[lc] >= 2) { curve = center_xy + 2 * (size_t)center_off[lc]; nc = center_off[lc + 1] - center_off[lc]; }
      }
      a0 = heading_to_curve(lane, nc, curve, px, py);
    }
  } else if (on) {                                               // OAPVehicleAgent (agent.py:283-312): the lanelet under the point
    ll = rv.RT > 0 ? rl_lanelet_of_wave(v, px, py) : -1;
    if (ll >= 0 && rv.count[(size_t)ll * rv.RT] >= 2) {          // heading of the record: first segment of route 0
      const double *q = rv.xy + 2 * (size_t)rv.first[(size_t)ll * rv.RT];
      a0 = atan2(q[3] - q[1], q[2] - q[0]);
    } else {
      ll = -1;
      a0 = heading_to_curve(lane, n_path, path, px, py);
    }
  }
  if (r == 0 && lane == 0) { pos0[2 * (agent0 + i)] = px; pos0[2 * (agent0 + i) + 1] = py; yaw0[agent0 + i] = a0; }
  spawn_write_slot(lane, slot, r, on, px, py, a0, type, ty.speed[ti], ty.raw_l[ti], ty.raw_w[ti], ty.infl_l[ti], ty.infl_w[ti], ll, rv, T,
                   dt, var0, factor, o, table_on, at, vpow, -1.0);
}

}  // namespace

extern "C" {

#if FO_RULE_TRACE
int fo_debug_rule_wticks(long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rule_wticks), sizeof(long long) * 8 * 1024); }
#endif

int fo_scene_set_topology(fo_ctx *ctx, int P, const double *h_left0, const int32_t *h_pred0, const int32_t *h_adj_left,
                          int n_inter, const int32_t *h_inter_off, const int32_t *h_inter_lanelet,
                          const uint8_t *h_inter_kind) {
  if (!ctx || !ctx->scene) return fo_fail(ctx, FO_E_STATE, "fo_scene_set_topology: call fo_scene_set_map first");
  Scene *sc = (Scene *)ctx->scene;
  if (P != sc->map->P || !h_left0 || !h_pred0 || !h_adj_left || n_inter < 0 || (n_inter > 0 && (!h_inter_off || !h_inter_lanelet || !h_inter_kind)))
    return fo_fail(ctx, FO_E_ARG, "fo_scene_set_topology: bad arguments (P=%d, the map has %d lanelets)", P, sc->map->P);
  if (sc->map->refs.load() > 1)
    return fo_fail(ctx, FO_E_STATE, "fo_scene_set_topology: the static map is shared (fo_scene_share_map); set the topology on the owner before sharing");
  for (int p = 0; p < P; ++p)
    if (h_pred0[p] < -1 || h_pred0[p] >= P || h_adj_left[p] < -1 || h_adj_left[p] >= P)
      return fo_fail(ctx, FO_E_ARG, "fo_scene_set_topology: lanelet index out of range at %d", p);
  const int n_e = n_inter > 0 ? h_inter_off[n_inter] : 0;
  for (int e = 0; e < n_e; ++e)
    if (h_inter_lanelet[e] < 0 || h_inter_lanelet[e] >= P) return fo_fail(ctx, FO_E_ARG, "fo_scene_set_topology: intersection entry %d out of range", e);
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  StaticMap *m = sc->map;
  for (void **p : {(void **)&m->d_left0, (void **)&m->d_pred0, (void **)&m->d_adj_left, (void **)&m->d_inter_off,
                   (void **)&m->d_inter_lanelet, (void **)&m->d_inter_kind}) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
  }
  FO_HIP_TRY(ctx, hipMalloc((void **)&m->d_left0, sizeof(double) * 2 * P));
  FO_HIP_TRY(ctx, hipMalloc((void **)&m->d_pred0, sizeof(int32_t) * P));
  FO_HIP_TRY(ctx, hipMalloc((void **)&m->d_adj_left, sizeof(int32_t) * P));
  FO_HIP_TRY(ctx, hipMemcpy(m->d_left0, h_left0, sizeof(double) * 2 * P, hipMemcpyHostToDevice));
  FO_HIP_TRY(ctx, hipMemcpy(m->d_pred0, h_pred0, sizeof(int32_t) * P, hipMemcpyHostToDevice));
  FO_HIP_TRY(ctx, hipMemcpy(m->d_adj_left, h_adj_left, sizeof(int32_t) * P, hipMemcpyHostToDevice));
  m->n_inter = n_inter;
  if (n_inter > 0) {
    FO_HIP_TRY(ctx, hipMalloc((void **)&m->d_inter_off, sizeof(int32_t) * (n_inter + 1)));
    FO_HIP_TRY(ctx, hipMalloc((void **)&m->d_inter_lanelet, sizeof(int32_t) * (n_e > 0 ? n_e : 1)));
    FO_HIP_TRY(ctx, hipMalloc((void **)&m->d_inter_kind, (size_t)(n_e > 0 ? n_e : 1)));
    FO_HIP_TRY(ctx, hipMemcpy(m->d_inter_off, h_inter_off, sizeof(int32_t) * (n_inter + 1), hipMemcpyHostToDevice));
    if (n_e > 0) {
      FO_HIP_TRY(ctx, hipMemcpy(m->d_inter_lanelet, h_inter_lanelet, sizeof(int32_t) * n_e, hipMemcpyHostToDevice));
      FO_HIP_TRY(ctx, hipMemcpy(m->d_inter_kind, h_inter_kind, (size_t)n_e, hipMemcpyHostToDevice));
    }
  }
  return FO_OK;
}

int fo_scene_spawn_rules(fo_ctx *ctx, const uint8_t *d_cls, int win_ix0, int win_iy0, int win_nx, int win_ny, int n_path,
                         const double *d_path6, int O, const double *d_ocorn, const double *d_ocen, const double *d_oyaw,
                         const double *d_odims, const uint8_t *d_oflags, const uint8_t *d_obst_vis,
                         const fo_spawn_rule_params_t *params, int max_out, double *d_out, int32_t *d_n_out, void *stream) {
  if (!ctx || !ctx->scene) return fo_fail(ctx, FO_E_STATE, "fo_scene_spawn_rules: call fo_scene_set_map first");
  Scene *sc = (Scene *)ctx->scene;
  StaticMap *m = sc->map;
  if (!d_cls || !params || !d_out || !d_n_out || max_out < 1 || win_nx < 1 || win_ny < 1 || n_path < 2 || !d_path6 || O < 0 ||
      (O > 0 && (!d_ocorn || !d_ocen || !d_oyaw || !d_odims || !d_oflags || !d_obst_vis)))
    return fo_fail(ctx, FO_E_ARG, "fo_scene_spawn_rules: bad arguments (n_path=%d O=%d max_out=%d)", n_path, O, max_out);
  if (!m->d_poly_off) return fo_fail(ctx, FO_E_STATE, "fo_scene_spawn_rules: the map holds no lanelet polygons");
  if (params->win_i0 < 0 || params->win_i1 > n_path || params->win_i1 < params->win_i0)
    return fo_fail(ctx, FO_E_ARG, "fo_scene_spawn_rules: reference window [%d, %d) outside the path", params->win_i0, params->win_i1);
  // table space the host can see (what only the device can -- a sampled line longer than RL_MAXSAMP cells / 8 -- comes back as
  // *d_n_out = -1): a rule that ran short would leave out a point the reference finds, unnoticed
  if (params->behind_turn && params->intention != 0 && params->win_i1 - params->win_i0 > RL_TURNW)
    return fo_fail(ctx, FO_E_ARG, "fo_scene_spawn_rules: the reference window holds %d path vertices, the turn rule %d "
                   "(thin the path out: the window is 40 m)", params->win_i1 - params->win_i0, RL_TURNW);
  if (params->behind_static && params->max_static > 15)
    return fo_fail(ctx, FO_E_ARG, "fo_scene_spawn_rules: max_static = %d, the selection compares the pedestrians of at most 16 obstacles", params->max_static);
  if (params->behind_dynamic && (params->intention == 0 || params->intention == 1)) {
    if (m->P > RL_LAT * RL_LAT)
      return fo_fail(ctx, FO_E_ARG, "fo_scene_spawn_rules: %d lanelets, the dynamic-obstacle rule holds flags for %d", m->P, RL_LAT * RL_LAT);
    if ((params->win_i1 - params->win_i0 + 4) / 5 > RL_FIFTHV)
      return fo_fail(ctx, FO_E_ARG, "fo_scene_spawn_rules: the reference window holds %d path vertices, the dynamic-obstacle rule "
                     "asks every fifth of at most %d", params->win_i1 - params->win_i0, 5 * RL_FIFTHV);
  }
  {
    // the select kernel compares the maxima BEFORE appending and a dynamic obstacle can yield two points (Q11): what the three
    // families can emit.  A smaller buffer would silently lose the last points in the reference's order (the turn rule's
    // pedestrian first) -- phantoms the sweep then never sees.
    const int can = (params->behind_dynamic ? (params->max_dynamic > 0 ? params->max_dynamic : 0) + 2 : 0) +
                    (params->behind_static ? (params->max_static > 0 ? params->max_static : 0) + 1 : 0) + (params->behind_turn ? 1 : 0);
    if (max_out < can)
      return fo_fail(ctx, FO_E_ARG, "fo_scene_spawn_rules: max_out = %d cannot hold the %d spawn points max_dynamic = %d / "
                     "max_static = %d allow", max_out, can, params->max_dynamic, params->max_static);
  }
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if ((rc = fo_reserve(ctx, &sc->d_rule_rec, &sc->cap_rule_rec, (size_t)(O + 1) * RL_REC))) return rc;
  RuleView v{};
  v.cls = d_cls; v.ix0 = win_ix0; v.iy0 = win_iy0; v.nx = win_nx; v.ny = win_ny;
  v.x0 = m->x0; v.y0 = m->y0; v.cs = m->cs; v.lane_yaw = m->d_lane_yaw; v.rnx = m->rnx; v.rny = m->rny;
  v.P = m->P; v.poly_off = m->d_poly_off; v.poly_xy = m->d_poly_xy; v.poly_box = m->d_poly_box;
  v.left0 = m->d_left0; v.pred0 = m->d_pred0; v.adj_left = m->d_adj_left;
  v.n_inter = m->n_inter; v.inter_off = m->d_inter_off; v.inter_lanelet = m->d_inter_lanelet; v.inter_kind = m->d_inter_kind;
  v.path = d_path6; v.n_path = n_path;
  RuleParams pr{};
  pr.ego_x = params->ego_x; pr.ego_y = params->ego_y; pr.ego_yaw = params->ego_yaw; pr.ego_s = params->ego_s; pr.ego_d = params->ego_d;
  pr.s_threshold = params->s_threshold; pr.ped_width = params->ped_width; pr.ped_length = params->ped_length;
  pr.intention = params->intention; pr.win_i0 = params->win_i0; pr.win_i1 = params->win_i1;
  pr.behind_static = params->behind_static; pr.behind_turn = params->behind_turn; pr.behind_dynamic = params->behind_dynamic;
  pr.max_static = params->max_static; pr.max_dynamic = params->max_dynamic;
  {  // (looked at on every call, like the other knobs: a test runs both labelling forms in one process)
    const char *e = fo_getenv(fo_env_any("FO_SCENE_"), "FO_SCENE_RULE_NODES");
    pr.label_nodes = e && e[0] == '1';
  }
  {  // lattice hand-off of the dynamic rule: [O][97 x 97] labels + a counter per obstacle (zero between launches)
    const size_t cap0 = sc->cap_rule_cnt;
    if ((rc = fo_reserve(ctx, &sc->d_rule_lab, &sc->cap_rule_lab, (size_t)(O > 0 ? O : 1) * RL_LAT * RL_LAT))) return rc;
    if ((rc = fo_reserve(ctx, &sc->d_rule_cnt, &sc->cap_rule_cnt, (size_t)(O > 0 ? O : 1)))) return rc;
    if (sc->cap_rule_cnt != cap0) FO_HIP_TRY(ctx, hipMemsetAsync(sc->d_rule_cnt, 0, sc->cap_rule_cnt * sizeof(int), s));
  }
  // helper workgroups of the dynamic rule for the obstacles that MAY take it (n_dynamic_plus1 - 1 of them by the caller's flags;
  // 0 = not told: every obstacle) -- none when the rule is off or the ego turns right
  const bool told = params->n_dynamic_plus1 > 0;
  int n_dyn = told ? (params->n_dynamic_plus1 - 1 < O ? params->n_dynamic_plus1 - 1 : O) : O;
  if (!(pr.behind_dynamic && (pr.intention == 0 || pr.intention == 1))) n_dyn = 0;
  hipLaunchKernelGGL(fo_spawn_rules_kernel, dim3(1 + O + n_dyn * (RL_PARTS - 1)), dim3(RL_THREADS), 0, s, v, pr, O, d_ocorn, d_ocen,
                     d_oyaw, d_odims, d_oflags, d_obst_vis, sc->d_rule_rec, sc->d_rule_lab, sc->d_rule_cnt, told ? 0 : 1, n_dyn);
  hipLaunchKernelGGL(fo_spawn_rules_select_kernel, dim3(1), dim3(64), 0, s, v, pr, O, d_ocorn, d_oflags, d_obst_vis,
                     sc->d_rule_rec, max_out, d_out, d_n_out);
  FO_HIP_TRY(ctx, hipGetLastError());
  return FO_OK;
}


int fo_scene_set_centerlines(fo_ctx *ctx, int P, const int32_t *h_off, const double *h_xy) {
  if (!ctx || !ctx->scene) return fo_fail(ctx, FO_E_STATE, "fo_scene_set_centerlines: call fo_scene_set_map first");
  Scene *sc = (Scene *)ctx->scene;
  StaticMap *m = sc->map;
  if (P != m->P || !h_off || h_off[0] != 0) return fo_fail(ctx, FO_E_ARG, "fo_scene_set_centerlines: bad arguments (P=%d, the map has %d lanelets)", P, m->P);
  for (int p = 0; p < P; ++p)
    if (h_off[p + 1] < h_off[p]) return fo_fail(ctx, FO_E_ARG, "fo_scene_set_centerlines: offsets must not decrease (lanelet %d)", p);
  const int NV = h_off[P];
  if (NV > 0 && !h_xy) return fo_fail(ctx, FO_E_ARG, "fo_scene_set_centerlines: null vertex table");
  if (m->refs.load() > 1)
    return fo_fail(ctx, FO_E_STATE, "fo_scene_set_centerlines: the static map is shared (fo_scene_share_map); set the centre lines on the owner before sharing");
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  for (void **q : {(void **)&m->d_center_off, (void **)&m->d_center_xy})
    if (*q) { (void)hipFree(*q); *q = nullptr; }
  FO_HIP_TRY(ctx, hipMalloc((void **)&m->d_center_off, sizeof(int32_t) * (P + 1)));
  FO_HIP_TRY(ctx, hipMalloc((void **)&m->d_center_xy, sizeof(double) * 2 * (size_t)(NV > 0 ? NV : 1)));
  FO_HIP_TRY(ctx, hipMemcpy(m->d_center_off, h_off, sizeof(int32_t) * (P + 1), hipMemcpyHostToDevice));
  if (NV > 0) FO_HIP_TRY(ctx, hipMemcpy(m->d_center_xy, h_xy, sizeof(double) * 2 * (size_t)NV, hipMemcpyHostToDevice));
  return FO_OK;
}

// fo_scene_spawn_rule_agents; slot0 / agent0 / at (fo_step_run): the rule agents' slots follow the cell sampler's in the
// same arrays, and the kernel writes its slots' rows of the sweep's agent table
int fo_scene_rule_agents_(fo_ctx *ctx, int max_points, const double *d_points, const int32_t *d_n_points, int routes,
                          const fo_rule_agent_types_t *types, int n_path, const double *d_path, int T, double dt, double var0,
                          double var_factor, int slot0, int agent0, double *d_pos0, double *d_yaw0, double *d_pos, double *d_yaw,
                          double *d_v, double *d_cov, double *d_shape, double *d_raw_dims, int32_t *d_type, int32_t *d_len,
                          void *stream, const fo_agent_table_t *at) {
  if (!ctx || !ctx->scene) return fo_fail(ctx, FO_E_STATE, "fo_scene_spawn_rule_agents: call fo_scene_set_map first");
  Scene *sc = (Scene *)ctx->scene;
  StaticMap *m = sc->map;
  if (max_points < 1 || !d_points || !d_n_points || !types || n_path < 2 || !d_path || T < 1 || !d_pos0 || !d_yaw0 || !d_pos ||
      !d_yaw || !d_v || !d_cov || !d_shape || !d_raw_dims || !d_type || !d_len || slot0 < 0 || agent0 < 0)
    return fo_fail(ctx, FO_E_ARG, "fo_scene_spawn_rule_agents: bad arguments (max_points=%d n_path=%d T=%d)", max_points, n_path, T);
  if (!m->d_poly_off) return fo_fail(ctx, FO_E_STATE, "fo_scene_spawn_rule_agents: the map holds no lanelet polygons");
  if (routes < 0 || (routes > 0 && !m->d_route_first))
    return fo_fail(ctx, FO_E_STATE, "fo_scene_spawn_rule_agents: routes = %d needs fo_scene_set_routes first", routes);
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  RuleView v{};
  v.x0 = m->x0; v.y0 = m->y0; v.cs = m->cs; v.P = m->P; v.poly_off = m->d_poly_off; v.poly_xy = m->d_poly_xy; v.poly_box = m->d_poly_box;
  RuleAgentTypes ty;
  for (int i = 0; i < 3; ++i) {
    ty.speed[i] = types->speed[i]; ty.raw_l[i] = types->raw_l[i]; ty.raw_w[i] = types->raw_w[i];
    ty.infl_l[i] = types->infl_l[i]; ty.infl_w[i] = types->infl_w[i];
  }
  RouteView rv;
  if (routes > 0) { rv.RT = m->R; rv.first = m->d_route_first; rv.count = m->d_route_count; rv.xy = m->d_route_xy; rv.s = m->d_route_s; }
  const int R = routes > 0 ? routes : 1;
  PredOut po{d_pos, d_yaw, d_v, d_cov, d_shape, d_raw_dims, d_type, d_len};
  hipLaunchKernelGGL(fo_spawn_rule_predict_kernel, dim3(max_points * R), dim3(64), 0, (hipStream_t)stream, v, max_points, d_points,
                     d_n_points, R, ty, n_path, d_path, m->d_center_off, m->d_center_xy, rv, T, dt, var0, var_factor, slot0, agent0,
                     d_pos0, d_yaw0, po, at ? 1 : 0, at ? *at : fo_agent_table_t());
  FO_HIP_TRY(ctx, hipGetLastError());
  return FO_OK;
}

int fo_scene_spawn_rule_agents(fo_ctx *ctx, int max_points, const double *d_points, const int32_t *d_n_points, int routes,
                               const fo_rule_agent_types_t *types, int n_path, const double *d_path, int T, double dt,
                               double var0, double var_factor, double *d_pos0, double *d_yaw0, double *d_pos, double *d_yaw,
                               double *d_v, double *d_cov, double *d_shape, double *d_raw_dims, int32_t *d_type,
                               int32_t *d_len, void *stream) {
  return fo_scene_rule_agents_(ctx, max_points, d_points, d_n_points, routes, types, n_path, d_path, T, dt, var0, var_factor, 0, 0,
                               d_pos0, d_yaw0, d_pos, d_yaw, d_v, d_cov, d_shape, d_raw_dims, d_type, d_len, stream, nullptr);
}

}  // extern "C"
