// fo_scene.hip -- visibility ray fan, occluded-cell grid, phantom spawn sampling and constant-velocity predictions
// for gfx950.  Replaces, per planning step, SensorModel.calc_visible_and_occluded_area (ref: sensor_model.py:41-193),
// the cell-based core of SpawnLocator.find_spawn_points (ref: spawn_locator.py:80-139) and the pedestrian-style
// prediction generator (ref: agent.py:451-536).  The reference does this with GEOS polygon algebra (one
// `difference` per boundary vertex); there is no ray or cell in it -- the discretisation is defined in
// DESIGN.md and is the same one oracle/ restates on the CPU.
//
// Compiled with -ffp-contract=off: every integer output (hit ids, cell classes, occluded-cell indices, spawn cells)
// must be bit-identical to the CPU restatement, so only + - * / sqrt and comparisons on float64 are used and no
// FMA is formed.
//
// Kernels (all latency-bound at one ego; launch count matters more than bandwidth here -- eight launches per step):
//   fo_raster_kernel       one-off: world-aligned road raster (cell centre inside any lanelet polygon)
//   fo_fan_kernel          ray directions, footprint range per ray, half fan of the occluded area
//   fo_rays_kernel         ray fan + obstacle-visibility probes in one launch: a workgroup per ray / per probe; the
//                          boundary pieces come in 64-piece chunks with bounding boxes, culled a lane per box;
//                          lexicographic (t, id) minimum by cross-lane shuffles = first hit
//   fo_grid_kernel         one thread per cell: fan sector by binary search on cross products, inside-the-chord test,
//                          half-fan test -> class bits; cells the fan cannot decide go to a list; also the per-block
//                          counts of the occluded-cell compaction
//   fo_settle_kernel       a workgroup per undecided cell (does an occluder cross the segment ego -> centre) and a
//                          workgroup per obstacle (5 mm skin)
//   fo_flag_compact_kernel deterministic stream compaction in one launch (ballot prefix inside a block, every block
//                          sums the counts before it; fo_flag_scan/scatter for very large windows)
//   fo_spawn_flag_kernel   candidate cells (+ block counts), fo_spawn_predict_kernel (evenly spaced pick + heading +
//                          predictions in the sweep's agent layout)
#include <hip/hip_runtime.h>
#include <atomic>
#include <math.h>
#include "fo_ctx.hpp"
#include "fo_prep_traj.hpp"
#include "fo_agent_rows.hpp"

namespace {


// the static map of a scenario: uploaded once (fo_scene_set_map / _set_routes / _set_edge_lines), read-only afterwards,
// and shared by reference between the contexts of several egos on one GPU (fo_scene_share_map: BASELINE configs[4],
// "shared occlusion map in HBM")
struct StaticMap {
  std::atomic<int> refs{1};        // contexts reading this map (fo_scene_share_map / fo_destroy may run on other host threads)
  int P = 0, E = 0;
  double cs = 0.5, x0 = 0, y0 = 0;
  int rnx = 0, rny = 0;
  double *d_edges = nullptr;      // [E][4]
  int32_t *d_edge_line = nullptr; // [E] straight-line chain of each piece (optional)
  double *d_chunk_box = nullptr;  // [ceil(E/64)][4] xmin, ymin, xmax, ymax of 64 consecutive pieces
  double *d_sub_box = nullptr;    // [ceil(E/64)][4][4] the same for the four 16-piece quarters of each chunk
  uint8_t *d_raster = nullptr;    // [rny][rnx]
  double *d_lane_yaw = nullptr;   // [rny][rnx] or null
  // phantom vehicle routes (optional): routes r < R of lanelet p = vertices route_first[p*R+r] .. +route_count[p*R+r]
  int R = 0, n_lanelets = 0;
  int32_t *d_route_first = nullptr, *d_route_count = nullptr, *d_lanelet_raster = nullptr;
  double *d_route_xy = nullptr, *d_route_s = nullptr;
  // lanelet polygons as uploaded (exact point-in-lanelet tests of the spawn rule families, fo_spawn_rules.hpp)
  int32_t *d_poly_off = nullptr;  // [P + 1]
  double *d_poly_xy = nullptr;    // [V][2]
  double *d_poly_box = nullptr;   // [P][4] xmin, ymin, xmax, ymax
  // lanelet topology the rule families read (fo_scene_set_topology; optional)
  double *d_left0 = nullptr;      // [P][2] first vertex of the left bound
  int32_t *d_pred0 = nullptr, *d_adj_left = nullptr;   // [P] index of predecessors[0] / adj_left, -1 = none
  int n_inter = 0;
  int32_t *d_inter_off = nullptr, *d_inter_lanelet = nullptr;   // intersection i: entries [off[i], off[i+1]) of
  uint8_t *d_inter_kind = nullptr;                              // (lanelet index, kind: 0 incoming, 1 inner)
  // lanelet centre lines (fo_scene_set_centerlines; optional): vertices center_xy[center_off[p] .. center_off[p+1])
  int32_t *d_center_off = nullptr;
  double *d_center_xy = nullptr;
};

void map_release(StaticMap *m) {
  if (!m || m->refs.fetch_sub(1) > 1) return;
  void *ptrs[] = {m->d_edges, m->d_edge_line, m->d_chunk_box, m->d_sub_box, m->d_raster, m->d_lane_yaw, m->d_route_first,
                  m->d_route_count, m->d_lanelet_raster, m->d_route_xy, m->d_route_s, m->d_poly_off, m->d_poly_xy,
                  m->d_poly_box, m->d_left0, m->d_pred0, m->d_adj_left, m->d_inter_off, m->d_inter_lanelet, m->d_inter_kind,
                  m->d_center_off, m->d_center_xy};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  delete m;
}

struct Scene {
  StaticMap *map = new StaticMap();
  // per-step workspace
  size_t cap_cand = 0, cap_vis32 = 0;
  int32_t *d_vis32 = nullptr;     // [O] probe results (zero between steps)
  uint8_t *d_flags = nullptr;     // [cells]
  int32_t *d_blk = nullptr;       // block counts / offsets
  size_t cap_cells = 0, cap_blk = 0;
  uint8_t *d_flags2 = nullptr;    // the same pair for the candidate flags a fused step writes during the first compaction
  int32_t *d_blk2 = nullptr;
  size_t cap_cells2 = 0, cap_blk2 = 0;
  bool cand_flags_ready = false;  // d_flags2 / d_blk2 hold this step's candidate flags (set by the fused visibility call)
  int32_t *d_cand = nullptr;      // candidate cell list
  int32_t *d_ncand = nullptr;
  int32_t *d_amb = nullptr;       // [cells] window indices of the cells the fan cannot decide
  int32_t *d_namb = nullptr;      // [1]; zeroed by the ray kernel of the step
  size_t cap_amb = 0;
  double *d_rule_rec = nullptr;   // [1 + O][24] per-workgroup records of the spawn rule families (fo_spawn_rules.hpp)
  size_t cap_rule_rec = 0;
  double shadow_length = 100.0;   // where an obstacle's shadow wedge ends (helper_functions.py:145-146); <= 0 or inf: nowhere
  double *d_ofar = nullptr;       // [O][3] per step: half-plane beyond the end of each obstacle's wedge
  size_t cap_ofar = 0;
  int *d_rule_lab = nullptr, *d_rule_cnt = nullptr;   // dynamic rule: [O][97 x 97] lattice labels, [O] arrival counters
  size_t cap_rule_lab = 0, cap_rule_cnt = 0;
};

Scene *scene_of(fo_ctx *ctx) {
  if (!ctx->scene) ctx->scene = new Scene();
  return (Scene *)ctx->scene;
}

// ------------------------------------------------------------------------------------------------ road raster
__global__ void fo_raster_kernel(int P, const int32_t *__restrict__ poly_off, const double *__restrict__ poly_xy,
                                 const double *__restrict__ pbox, double x0, double y0, double cs, int nx, int ny,
                                 uint8_t *__restrict__ mask) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nx * ny) return;
  const int ix = idx % nx, iy = idx / nx;
  const double px = x0 + ((double)ix + 0.5) * cs, py = y0 + ((double)iy + 0.5) * cs;
  int inside_any = 0;
  for (int p = 0; p < P && !inside_any; ++p) {
    const double *bb = pbox + 4 * (size_t)p;  // xmin, ymin, xmax, ymax: pure early-out, cannot change the result
    if (px < bb[0] || px > bb[2] || py < bb[1] || py > bb[3]) continue;
    const int b = poly_off[p], e = poly_off[p + 1];
    int c = 0;
    for (int i = b, j = e - 1; i < e; j = i++) {
      const double xi = poly_xy[2 * i], yi = poly_xy[2 * i + 1], xj = poly_xy[2 * j], yj = poly_xy[2 * j + 1];
      if ((yi > py) != (yj > py)) {
        const double xc = xi + (py - yi) * (xj - xi) / (yj - yi);
        if (px < xc) c ^= 1;
      }
    }
    inside_any = c;
  }
  mask[idx] = (uint8_t)inside_any;
}

// ------------------------------------------------------------------------------------------------ ray casting
// Ray o + t d against segment a + u (b - a):  denom = d x e,  tn = w x e,  un = w x d  (w = a - o).
// Hit iff denom != 0 and 0 <= tn/denom and 0 <= un/denom <= 1, decided on the signs of the numerators (no division
// on the rejection path); t = tn / denom is formed for hits only.  Same predicate, same operation order as the CPU
// restatement used by the tests, compiled without FMA contraction on both sides.
__device__ __forceinline__ double ray_segment(double ox, double oy, double dx, double dy, double ax, double ay,
                                              double bx, double by) {
  const double ex = bx - ax, ey = by - ay;
  const double denom = dx * ey - dy * ex;
  if (denom == 0.0) return INFINITY;
  const double wx = ax - ox, wy = ay - oy;
  const double tn = wx * ey - wy * ex;
  const double un = wx * dy - wy * dx;
  const bool hit = denom > 0.0 ? (tn >= 0.0 && un >= 0.0 && un <= denom) : (tn <= 0.0 && un <= 0.0 && un >= denom);
  return hit ? tn / denom : INFINITY;
}

// lexicographic (t, id) minimum across the wave
__device__ __forceinline__ void wave_min_hit(double &t, int &id) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double t2 = __shfl_xor(t, off);
    const int id2 = __shfl_xor(id, off);
    if (t2 < t || (t2 == t && id2 < id)) { t = t2; id = id2; }
  }
}

// A wave's share of the occluder soup.  Static pieces come in chunks of 64 consecutive table entries (a lane each, 2 KB
// contiguous per chunk, L2 resident); wave w takes chunks [64 w, 64 w + 64), [64 (w + n_waves), ...).  A chunk is skipped
// -- decided a lane per chunk box, then shared by ballot -- when the box misses the bounding box of the ray segment [o, o + tmax d] or lies
// entirely on one side of the ray's line; both tests carry a margin far above rounding (1e-7 m), so culling never
// changes a result, it only saves the reads (the maps are hundreds of metres wide, a ray reaches tens).
// Obstacle o contributes its four sides with id E + o when it is present and occludes (bicycles do not, Q10).
// SKIP: eskip[e] != 0 marks boundary pieces that cast no shadow this step (rings of the road union enclosed by the
// sensor footprint: the reference walks exterior rings of road ∩ footprint only, sensor_model.py:126-131).
__device__ __forceinline__ bool chunk_culled(const double *__restrict__ box, double ox, double oy, double dx, double dy,
                                             double tmax) {
  const double m = 1e-7;
  const double px = ox + tmax * dx, py = oy + tmax * dy;
  const double sx0 = fmin(ox, px) - m, sx1 = fmax(ox, px) + m, sy0 = fmin(oy, py) - m, sy1 = fmax(oy, py) + m;
  const double bx0 = box[0], by0 = box[1], bx1 = box[2], by1 = box[3];
  if (bx0 > sx1 || bx1 < sx0 || by0 > sy1 || by1 < sy0) return true;
  // signed offsets of the four box corners from the ray's line, scaled by |d| (<= 1.5 r for the settle kernel)
  const double mm = m * (fabs(dx) + fabs(dy)) * 1.0e2;
  const double c00 = dx * (by0 - oy) - dy * (bx0 - ox), c10 = dx * (by0 - oy) - dy * (bx1 - ox);
  const double c01 = dx * (by1 - oy) - dy * (bx0 - ox), c11 = dx * (by1 - oy) - dy * (bx1 - ox);
  if (c00 > mm && c10 > mm && c01 > mm && c11 > mm) return true;
  if (c00 < -mm && c10 < -mm && c01 < -mm && c11 < -mm) return true;
  return false;
}

#ifndef FO_PRED_TRACE
#define FO_PRED_TRACE 0   // tuning builds: wall-clock stamps of one prediction workgroup's phases (fo_debug_pred_ticks, tools/pred_trace.py)
#endif
#if FO_PRED_TRACE
__device__ long long g_pred_ticks[16], g_ray_ticks[16];
#define PRED_TICK(i) do { if (blockIdx.x == FO_PRED_TRACE && threadIdx.x == 0) g_pred_ticks[i] = wall_clock64(); } while (0)
#define RAY_TICK(i) do { if (blockIdx.x == FO_PRED_TRACE && threadIdx.x == 0) g_ray_ticks[i] = wall_clock64(); } while (0)
#define FO_PRED_TRACE_BLOCK FO_PRED_TRACE
#define RAY_NOTE(i, v) do { g_ray_ticks[i] = (long long)(v); } while (0)
#else
#define PRED_TICK(i) do { } while (0)
#define RAY_TICK(i) do { } while (0)
#define FO_PRED_TRACE_BLOCK (-1)
#define RAY_NOTE(i, v) do { } while (0)
#endif

template <bool SKIP>
__device__ __forceinline__ void scan_soup(int E, const double *__restrict__ edges, const double *__restrict__ chunk_box,
                                          const uint8_t *__restrict__ eskip, int O, const double *__restrict__ ocorn,
                                          const uint8_t *__restrict__ oflags, int wave, int n_waves, int lane, double ox,
                                          double oy, double dx, double dy, double tmax, int skip_id, double &best,
                                          int &best_id) {
  const int nc = (E + 63) >> 6;
  for (int cb = wave * 64; cb < nc; cb += n_waves * 64) {
    // a lane per chunk box: one round trip culls 64 chunks; the survivors are then scanned a lane per piece
    const int cc = cb + lane;
    unsigned long long live = __ballot(cc < nc && !chunk_culled(chunk_box + 4 * (size_t)(cc < nc ? cc : 0), ox, oy, dx, dy, tmax));
    if (cb == 0) { RAY_TICK(5); if (threadIdx.x == 0 && blockIdx.x == FO_PRED_TRACE_BLOCK) RAY_NOTE(12, __popcll(live)); }
    while (live) {
      const int c = cb + __builtin_ctzll(live);
      live &= live - 1;
      const int gi = (c << 6) + lane;
      if (gi >= E) continue;
      if (SKIP && eskip[gi]) continue;
      const double *p = edges + 4 * (size_t)gi;
      const double t = ray_segment(ox, oy, dx, dy, p[0], p[1], p[2], p[3]);
      if (t < best || (t == best && gi < best_id)) { best = t; best_id = gi; }
    }
  }
  RAY_TICK(6);
  // obstacle sides, interleaved over the whole workgroup
  for (int k = wave * 64 + lane; k < 4 * O; k += 64 * n_waves) {
    const int o = k >> 2, sd = k & 3, s2 = (sd + 1) & 3;
    if (!((oflags[o] & 1) && (oflags[o] & 2)) || E + o == skip_id) continue;
    const double *c = ocorn + 8 * (size_t)o;
    const int id = E + o;
    const double t = ray_segment(ox, oy, dx, dy, c[2 * sd], c[2 * sd + 1], c[2 * s2], c[2 * s2 + 1]);
    if (t < best || (t == best && id < best_id)) { best = t; best_id = id; }
  }
}

// ------------------------------------------------------------------------------------------------ fan sector
// (DIR: where the unit direction of ray i comes from -- the table the fan kernel wrote, or, inside the launch that is
// still writing that table, the fan's own arithmetic: FanDirs below)
struct TableDirs {
  const double *__restrict__ dirs;
  __device__ __forceinline__ void get(int i, double &cx, double &cy) const { cx = dirs[2 * (size_t)i]; cy = dirs[2 * (size_t)i + 1]; }
};
template <class DIR>
__device__ __forceinline__ int fan_ccw_t(int n_rays, const DIR &D, int i, double rx, double ry) {
  double d0, d1;
  D.get(i == n_rays ? 0 : i, d0, d1);
  const double c = d0 * ry - d1 * rx;
  if (c > 0.0) return 1;
  if (c < 0.0) return 0;
  return (d0 * rx + d1 * ry) > 0.0;
}
template <class DIR>
__device__ int fan_search_t(int n_rays, const DIR &D, int a, int b, double rx, double ry) {
  if (!fan_ccw_t(n_rays, D, a, rx, ry) || fan_ccw_t(n_rays, D, b, rx, ry)) return -1;
  int lo = a, hi = b;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (fan_ccw_t(n_rays, D, mid, rx, ry)) lo = mid; else hi = mid;
  }
  return lo;
}
template <class DIR>
__device__ int fan_sector_t(int n_rays, const DIR &D, int full, double rx, double ry) {
  if (full) {  // thirds: each part spans < pi for every n >= 4 (halves exceed pi by a ray pitch when n is odd)
    const int a = n_rays / 3, b = (2 * n_rays) / 3;
    int s = fan_search_t(n_rays, D, 0, a, rx, ry);
    if (s >= 0) return s;
    s = fan_search_t(n_rays, D, a, b, rx, ry);
    if (s >= 0) return s;
    return fan_search_t(n_rays, D, b, n_rays, rx, ry);
  }
  const int m = (n_rays - 1) / 2;
  const int s = fan_search_t(n_rays, D, 0, m, rx, ry);
  if (s >= 0) return s;
  return fan_search_t(n_rays, D, m, n_rays - 1, rx, ry);
}
__device__ __forceinline__ int fan_ccw(int n_rays, const double *__restrict__ dirs, int i, double rx, double ry) {
  return fan_ccw_t(n_rays, TableDirs{dirs}, i, rx, ry);
}
__device__ int fan_search(int n_rays, const double *__restrict__ dirs, int a, int b, double rx, double ry) {
  return fan_search_t(n_rays, TableDirs{dirs}, a, b, rx, ry);
}
__device__ int fan_sector(int n_rays, const double *__restrict__ dirs, int full, double rx, double ry) {
  return fan_sector_t(n_rays, TableDirs{dirs}, full, rx, ry);
}

// ------------------------------------------------------------------------------------------------ ray fan
// Directions and footprint ranges of the fan about the ego heading, written where the ray and cell kernels read them
// (no host trigonometry, no per-step upload).  Full circle: angle_i = yaw + 2 pi i / n; open fan: n rays from
// yaw - fov/2 to yaw + fov/2 inclusive (sensor_model.py:115-124).  rmax: range of the reference's polygonal footprint
// along the ray -- regular 64-gon with a vertex at world angle 0 (Point.buffer(r)) or the 100-point fan of
// _calc_relevant_sector (:201-209): r cos(d/2) / cos(rel mod d - d/2) with d the angular pitch of the arc points.
// ray i of the fan: unit direction and (want_rmax) the footprint range along it
__device__ __forceinline__ void fan_ray(int i, int n, double yaw, double fov, int full, double r, int polygon, double &cs,
                                        double &sn, double &rm) {
  const double two_pi = 6.283185307179586476925286766559;
  double ang, rel, d;
  if (full) {
    ang = yaw + two_pi * (double)i / (double)n;
    rel = ang;
    d = two_pi / 64.0;
  } else {
    rel = i == n - 1 ? fov : fov * (double)i / (double)(n - 1);
    ang = yaw - 0.5 * fov + rel;
    d = fov / 99.0;
  }
  sincos(ang, &sn, &cs);
  rm = r;
  if (polygon) {
    const double m = rel - d * floor(rel / d);
    rm = r * cos(0.5 * d) / cos(m - 0.5 * d);
  }
}
// unit direction i < 100 of the 100-point half fan (radius 1.5 r) of sensor_model.py:85-87
__device__ __forceinline__ void fan_half_dir(int i, double yaw, double &cs, double &sn) {
  const double two_pi = 6.283185307179586476925286766559;
  const double a = i == 99 ? yaw + 0.25 * two_pi : yaw - 0.25 * two_pi + 0.5 * two_pi * (double)i / 99.0;
  sincos(a, &sn, &cs);
}
__global__ void fo_fan_kernel(int n, double yaw, double fov, int full, double r, int polygon,
                              double *__restrict__ dirs, double *__restrict__ rmax, double *__restrict__ half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (half && i < 100) {
    double sn, cs;
    fan_half_dir(i, yaw, cs, sn);
    half[2 * i] = cs;
    half[2 * i + 1] = sn;
  }
  if (i >= n) return;
  double sn, cs, rm;
  fan_ray(i, n, yaw, fov, full, r, polygon, cs, sn, rm);
  dirs[2 * i] = cs;
  dirs[2 * i + 1] = sn;
  if (rmax) rmax[i] = rm;
}
// the fan computed inside the ray kernel (fo_step_run: one launch less): every ray workgroup works out its own direction
// and range and leaves them where the later kernels of the step read them
struct FanArgs {
  int on = 0, full = 0, polygon = 0;
  double yaw = 0, fov = 0;
  double *dirs = nullptr, *rmax = nullptr, *half = nullptr;
  // fo_step_t::h_mirror as the device sees it (pinned host memory is mapped): the hit ids and the obstacles' visibility
  // flags are ALSO stored there by the launches that produce them -- posted writes over PCIe, no copy command behind the step
  // (a device-to-host copy of 3 KB is a blit launch of ~5 us); null: no mirror, or one the step copies
  int32_t *hit_host = nullptr;
  uint8_t *vis_host = nullptr;
};
struct FanDirs {   // the direction of ray i by the fan's arithmetic (bit-identical to the table entry)
  int n;
  double yaw, fov;
  int full;
  __device__ __forceinline__ void get(int i, double &cx, double &cy) const {
    double rm;
    fan_ray(i, n, yaw, fov, full, 1.0, 0, cx, cy, rm);
  }
};

// Sector of a uniform full fan (ray i at angle yaw + 2 pi i / n, ray 0 = dirs[0]): a float atan2 of the direction
// rotated back by ray 0 proposes the index, the exact predicate of fan_sector (ccw(i) and not ccw(i + 1)) confirms it
// or moves it by a step -- same answer as the binary search, a fraction of its dependent loads.
__device__ __forceinline__ int fan_sector_uniform(int n_rays, const double *__restrict__ dirs, double rx, double ry) {
  const float c0 = (float)dirs[0], s0 = (float)dirs[1];
  const float fx = (float)rx, fy = (float)ry;
  const float ang = atan2f(fy * c0 - fx * s0, fx * c0 + fy * s0);
  int i = (int)floorf(ang * ((float)n_rays * 0.15915494309189535f));
  if (i < 0) i += n_rays;
  if (i >= n_rays) i -= n_rays;
#pragma unroll 1
  for (int it = 0; it < 3; ++it) {
    const int j = i + 1 == n_rays ? 0 : i + 1;
    const bool a = fan_ccw(n_rays, dirs, i, rx, ry), b = fan_ccw(n_rays, dirs, j, rx, ry);
    if (a && !b) return i;
    if (!a) i = i == 0 ? n_rays - 1 : i - 1; else i = j;
  }
  return fan_sector(n_rays, dirs, 1, rx, ry);
}

// ------------------------------------------------------------------------------------------------ rays + probes
// One launch for the ray fan and the obstacle-visibility probes.  Workgroups [0, n_rays): one ray each, its five
// waves scan interleaved fifths of the soup and the (t, id) minima are combined through LDS.  Workgroups
// [n_rays, n_rays + 5 O): one visibility probe each (obstacle o, probe p: 4 corners + centre; sensor_model.py:59-76
// restated) against the soup with the obstacle itself left out; a visible probe sets vis32[o], which the cell-grid
// kernel turns into the byte flag and clears again for the next step.
constexpr int RAY_WAVES = 5;
constexpr int SETTLE_BLOCKS = 1024;  // grid of the settle kernel (grid-stride over the undecided cells)
// NW: waves per workgroup -- RAY_WAVES, or 1 for maps whose boundary soup is a single group of chunk boxes (<= 64 chunks = 4 096
// pieces: only the first wave of five would have pieces to scan; one-wave workgroups are dispatched five times faster and
// meet no barrier)
template <bool SKIP, int NW>
__global__ __launch_bounds__(64 * NW) void fo_rays_kernel(int E, const double *__restrict__ edges,
                                                                 const double *__restrict__ chunk_box,
                                                                 const uint8_t *__restrict__ eskip, int O,
                                                                 const double *__restrict__ ocorn,
                                                                 const double *__restrict__ ocen,
                                                                 const uint8_t *__restrict__ oflags, double ex, double ey,
                                                                 int n_rays, const double *__restrict__ dirs, double r,
                                                                 const double *__restrict__ rmax, int full,
                                                                 double *__restrict__ range,
                                                                 int32_t *__restrict__ hit_id, double *__restrict__ ring,
                                                                 int32_t *__restrict__ vis32,
                                                                 int32_t *__restrict__ n_amb, FanArgs fan,
                                                                 const fo_prep_args_t prep) {
  __shared__ double sh_t[NW];
  __shared__ int sh_id[NW];
  // Workgroups past the rays and probes (fo_step_run): the sweep's tile table of the candidate trajectories -- independent
  // of the scene, written while this launch leaves most of the chip idle instead of by a launch of its own before the sweep.
  if ((int)blockIdx.x >= n_rays + (vis32 ? 5 * O : 0)) {
    __shared__ double prep_sh[2 * FO_PREP_TZ * (FO_PREP_TILE + 1)];
    const int e = (int)blockIdx.x - (n_rays + (vis32 ? 5 * O : 0));
    fo_prep_traj_block(prep, e % prep.n_tiles, (e / prep.n_tiles) & 1, e / (2 * prep.n_tiles), prep_sh);
    return;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (blockIdx.x == 0 && threadIdx.x == 0 && n_amb) *n_amb = 0;  // list of undecided cells of this step (grid kernel)
  if ((int)blockIdx.x < n_rays) {
    const int i = blockIdx.x;
    double dx, dy, rm;
    RAY_TICK(0);
    if (fan.on) {   // (wave-uniform; the same arithmetic as fo_fan_kernel)
      fan_ray(i, n_rays, fan.yaw, fan.fov, fan.full, r, fan.polygon, dx, dy, rm);
      if (!fan.rmax) rm = r;
      if (threadIdx.x == 0) {
        fan.dirs[2 * i] = dx;
        fan.dirs[2 * i + 1] = dy;
        if (fan.rmax) fan.rmax[i] = rm;
      }
      if (fan.half && i == 0)
        for (int k = threadIdx.x; k < 100; k += 64 * NW) {   // (a workgroup may be a single wave)
          double hs, hc;
          fan_half_dir(k, fan.yaw, hc, hs);
          fan.half[2 * k] = hc;
          fan.half[2 * k + 1] = hs;
        }
    } else {
      dx = dirs[2 * i];
      dy = dirs[2 * i + 1];
      rm = rmax ? rmax[i] : r;  // range of the sensor footprint along this ray
    }
    double best = INFINITY;
    int id = 0x7fffffff;
    RAY_TICK(1);
    scan_soup<SKIP>(E, edges, chunk_box, eskip, O, ocorn, oflags, wave, NW, lane, ex, ey, dx, dy, rm, -3, best, id);
    RAY_TICK(2);
    wave_min_hit(best, id);
    RAY_TICK(3);
    if (lane == 0) { sh_t[wave] = best; sh_id[wave] = id; }
    __syncthreads();
    RAY_TICK(4);
    if (threadIdx.x == 0) {
      for (int w = 1; w < NW; ++w)
        if (sh_t[w] < best || (sh_t[w] == best && sh_id[w] < id)) { best = sh_t[w]; id = sh_id[w]; }
      if (!(best <= rm)) { best = rm; id = -1; }
      range[i] = best;
      hit_id[i] = id;
      if (fan.hit_host) fan.hit_host[i] = id;
      // a ray that stops at an obstacle has reached a lit point of its boundary: the obstacle touches the visible area
      // (sensor_model.py:59-76); the probe workgroups below add the obstacles that slip between two rays
      if (id >= E && vis32) atomicOr(&vis32[id - E], 1);
      if (ring) {
        ring[2 * i] = ex + best * dx;
        ring[2 * i + 1] = ey + best * dy;
      }
      RAY_TICK(7);
    }
    return;
  }
  // probe workgroup: obstacle o, probe p (4 corners + centre); all five waves share the soup like a ray workgroup
  const int o = (blockIdx.x - n_rays) / 5, p = (blockIdx.x - n_rays) % 5;
  const bool exists = oflags[o] & 1;
  const double qx = p < 4 ? ocorn[8 * (size_t)o + 2 * p] : ocen[2 * o];
  const double qy = p < 4 ? ocorn[8 * (size_t)o + 2 * p + 1] : ocen[2 * o + 1];
  const double rx = qx - ex, ry = qy - ey;
  const double dist = sqrt(rx * rx + ry * ry);
  bool cand = exists && !(dist > r + 0.01);
  if (cand && dist == 0.0) {
    if (threadIdx.x == 0) atomicOr(&vis32[o], 1);
    return;
  }
  if (cand) {   // (in the fused launch the table of directions is still being written by the ray workgroups)
    const int sec = fan.on ? fan_sector_t(n_rays, FanDirs{n_rays, fan.yaw, fan.fov, fan.full}, full, rx, ry)
                           : fan_sector(n_rays, dirs, full, rx, ry);
    if (sec < 0) cand = false;
  }
  if (!cand) return;  // uniform over the workgroup
  const double dx = rx / dist, dy = ry / dist;
  double best = INFINITY;
  int id = 0x7fffffff;
  scan_soup<SKIP>(E, edges, chunk_box, eskip, O, ocorn, oflags, wave, NW, lane, ex, ey, dx, dy, dist, E + o, best, id);
  wave_min_hit(best, id);
  if (lane == 0) sh_t[wave] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < NW; ++w) best = fmin(best, sh_t[w]);
    double t = best;
    if (!(t <= dist)) t = dist;  // first_hit(..., rmax = dist)
    if (t >= dist - 0.01) atomicOr(&vis32[o], 1);
  }
}

// The reference's 1.5 r half disc is the 100-point fan of _calc_relevant_sector (sensor_model.py:85-87,201-209).  A
// centre in the thin rim between that polygon and the circle (d2 above 0.9994 ro2; the polygon's inscribed radius
// squared is 0.99975 ro2) is tested against the chord of its sector; half = its 100 unit directions or null.
__device__ __forceinline__ int in_half_fan(const double *__restrict__ half, double r, double rx, double ry, double d2,
                                           double ro2) {
  if (!half || !(d2 > 0.9994 * ro2)) return 1;
  const int k = fan_search(100, half, 0, 99, rx, ry);
  if (k < 0) return 1;
  const double R = 1.5 * r;
  const double ax = R * half[2 * k], ay = R * half[2 * k + 1];
  const double bx = R * half[2 * k + 2], by = R * half[2 * k + 3];
  return ((bx - ax) * (ry - ay) - (by - ay) * (rx - ax)) >= 0.0;
}

// sensor_model.py:183: the obstacle rectangle grown by 5 mm (mitred corners) is taken out of the visible area.  Inside
// iff the signed distance to each side is <= 5 mm: cross(e, q - a) against 0.005 |e|, either ring orientation;
// present, non-bicycle obstacles only.
__device__ __forceinline__ int in_obstacle_skin(int O, const double *__restrict__ ocorn,
                                                const uint8_t *__restrict__ oflags, double px, double py) {
  for (int o = 0; o < O; ++o) {
    if (!(oflags[o] & 1) || !(oflags[o] & 2)) continue;
    const double *c = ocorn + 8 * (size_t)o;
    {  // pure early-out: farther from the rectangle's centre than its grown half diagonal (hd + 0.005 sqrt 2)
      const double mx = 0.5 * (c[0] + c[4]), my = 0.5 * (c[1] + c[5]);
      const double hd2 = (c[0] - mx) * (c[0] - mx) + (c[1] - my) * (c[1] - my);
      const double d2c = (px - mx) * (px - mx) + (py - my) * (py - my);
      if (d2c > hd2 + 0.0071 * (1.0 + hd2) + 1e-4) continue;
    }
    const double area2 = (c[2] - c[0]) * (c[5] - c[1]) - (c[3] - c[1]) * (c[4] - c[0]);
    const double sg = area2 >= 0.0 ? 1.0 : -1.0;
    int inside = 1;
    for (int sd = 0; sd < 4 && inside; ++sd) {
      const int s2 = (sd + 1) & 3;
      const double ex = c[2 * s2] - c[2 * sd], ey = c[2 * s2 + 1] - c[2 * sd + 1];
      const double cr = ex * (py - c[2 * sd + 1]) - ey * (px - c[2 * sd]);
      if (-(sg * cr) > 0.005 * sqrt(ex * ex + ey * ey)) inside = 0;
    }
    if (inside) return 1;
  }
  return 0;
}

// Where an obstacle's shadow ENDS in the reference (helper_functions.py:139-176): the occlusion polygon is the quad
// [c1, c2, c2 + L u(c2 - ego), c1 + L u(c1 - ego)], L = 100 m, with (c1, c2) the corner pair that subtends the largest angle
// at the ego (_identify_projection_points: all 4 x 4 ordered pairs, arccos of the clipped dot product of the unit vectors,
// strictly greater wins, first in loop order).  Beyond the chord between the two end points the obstacle hides nothing.
// out[3] = (a, b, c): a point lies beyond that chord iff a x + b y + c > 0 (the ego on the other side); a = b = 0, c = -1
// when there is no such chord (length <= 0 or infinite: shadows without end, or a degenerate view).
// Sixteen consecutive lanes per obstacle (q = lane & 15 = the ordered corner pair i = q >> 2, j = q & 3): one arccos per lane
// instead of a chain of sixteen; the largest angle with the smallest q among equals = the reference's "strictly greater, first
// in loop order".  Every lane of the group must call; lane q == 0 writes.
__device__ inline void wedge_far_halfplane(int q, double ex, double ey, const double *__restrict__ c, double length, double *out) {
  const bool on = length > 0.0 && length < INFINITY;
  const int i = q >> 2, j = q & 3;
  double ang;
  {
    const double r1x = c[2 * i] - ex, r1y = c[2 * i + 1] - ey, r2x = c[2 * j] - ex, r2y = c[2 * j + 1] - ey;
    const double n1 = sqrt(r1x * r1x + r1y * r1y), n2 = sqrt(r2x * r2x + r2y * r2y);
    const double u1x = r1x / n1, u1y = r1y / n1, u2x = r2x / n2, u2y = r2y / n2;
    ang = acos(fmin(fmax(u1x * u2x + u1y * u2y, -1.0), 1.0));
  }
  // (an angle that is not > 0 -- zero or NaN -- never replaces the initial "none": key -1)
  double best = ang > 0.0 ? ang : -1.0;
  int bq = ang > 0.0 ? q : 16;
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) {
    const double b2 = __shfl_xor(best, off, 16);
    const int q2 = __shfl_xor(bq, off, 16);
    if (b2 > best || (b2 == best && q2 < bq)) { best = b2; bq = q2; }
  }
  if (q != 0) return;
  out[0] = 0.0; out[1] = 0.0; out[2] = -1.0;
  if (!on || bq >= 16) return;
  const int i1 = bq >> 2, i2 = bq & 3;
  const double r1x = c[2 * i1] - ex, r1y = c[2 * i1 + 1] - ey, r2x = c[2 * i2] - ex, r2y = c[2 * i2 + 1] - ey;
  const double n1 = sqrt(r1x * r1x + r1y * r1y), n2 = sqrt(r2x * r2x + r2y * r2y);
  const double c4x = c[2 * i1] + r1x / n1 * length, c4y = c[2 * i1 + 1] + r1y / n1 * length;   // c1 + L u(c1 - ego)
  const double c3x = c[2 * i2] + r2x / n2 * length, c3y = c[2 * i2 + 1] + r2y / n2 * length;   // c2 + L u(c2 - ego)
  double a = -(c4y - c3y), b = c4x - c3x;
  double cc = -(a * c3x + b * c3y);
  const double ge = a * ex + b * ey + cc;
  if (ge == 0.0 || ge != ge) return;
  if (ge > 0.0) { a = -a; b = -b; cc = -cc; }
  out[0] = a; out[1] = b; out[2] = cc;
}

// ------------------------------------------------------------------------------------------------ cell grid
__global__ void fo_grid_kernel(const uint8_t *__restrict__ raster, int rnx, int rny, double rx0, double ry0, double cs,
                               int ix0, int iy0, int nx, int ny, double ex, double ey, double hx, double hy, double r,
                               int full, int n_rays, const double *__restrict__ dirs,
                               const double *__restrict__ range, uint8_t *__restrict__ cls,
                               uint8_t *__restrict__ occ_flag, int32_t *__restrict__ blk, int O,
                               int32_t *__restrict__ vis32, uint8_t *__restrict__ vis, int exact, int E,
                               const int32_t *__restrict__ hit_id, const double *__restrict__ rmax,
                               int32_t *__restrict__ amb, int32_t *__restrict__ n_amb,
                               const double *__restrict__ half, const int32_t *__restrict__ edge_line,
                               const double *__restrict__ ocorn, const uint8_t *__restrict__ oflags, double shadow_length,
                               double *__restrict__ ofar, int n_obst, uint8_t *__restrict__ vis_host) {
  __shared__ int wsum[4];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  // where the obstacles' shadows end (read by the settle kernel, the next launch): sixteen lanes per obstacle
  if (ofar && ocorn && (idx >> 4) < n_obst) {   // (uniform over each group of sixteen lanes: blocks are multiples of 16)
    const int o = idx >> 4;
    if ((oflags[o] & 1) && (oflags[o] & 2)) wedge_far_halfplane(idx & 15, ex, ey, ocorn + 8 * (size_t)o, shadow_length, ofar + 3 * (size_t)o);
    else if ((idx & 15) == 0) { ofar[3 * o] = 0.0; ofar[3 * o + 1] = 0.0; ofar[3 * o + 2] = -1.0; }
  }
  if (vis && idx < O) {  // obstacle-visibility flags of the probe workgroups (previous launch); self-cleaning
    vis[idx] = vis32[idx] ? 1 : 0;
    if (vis_host) vis_host[idx] = vis32[idx] ? 1 : 0;
    vis32[idx] = 0;
  }
  const bool in = idx < nx * ny;
  uint8_t c = 0;
  bool pending = false;
  int visible = 0;
  double px = 0.0, py = 0.0, rx = 0.0, ry = 0.0, d2 = 0.0;
  const double r2 = r * r, ro2 = (1.5 * r) * (1.5 * r);
  if (in) {
    const int ix = idx % nx, iy = idx / nx;
    const int wx = ix0 + ix, wy = iy0 + iy;
    if (wx >= 0 && wx < rnx && wy >= 0 && wy < rny && raster[(size_t)wy * rnx + wx]) c |= 1;
    px = rx0 + ((double)wx + 0.5) * cs;
    py = ry0 + ((double)wy + 0.5) * cs;
    rx = px - ex;
    ry = py - ey;
    d2 = rx * rx + ry * ry;
    if ((c & 1) && d2 <= r2) {
      // (the zero vector never confirms a proposal and falls through to fan_sector's -1)
      const int i = full ? fan_sector_uniform(n_rays, dirs, rx, ry) : fan_sector(n_rays, dirs, full, rx, ry);
      if (rx == 0.0 && ry == 0.0) {
        visible = 1;
      } else if (i >= 0) {
        const int j = (i + 1 == n_rays) ? 0 : i + 1;
        const double hix = range[i] * dirs[2 * i], hiy = range[i] * dirs[2 * i + 1];
        const double hjx = range[j] * dirs[2 * j], hjy = range[j] * dirs[2 * j + 1];
        const double cr = (hjx - hix) * (ry - hiy) - (hjy - hiy) * (rx - hix);
        visible = cr >= 0.0;
        if (exact) {
          // the two enclosing rays stop at different occluders (or at an obstacle) and the centre is not nearer than
          // the shorter of them by more than a cell: the fan cannot decide (at grazing incidence the centre's own
          // ray may reach well past both); inside the footprint chord the settle kernel does
          int idi = hit_id[i], idj = hit_id[j];
          if (edge_line && idi >= 0 && idi < E && idj >= 0 && idj < E) {  // same straight chain = one occluder
            idi = edge_line[idi];
            idj = edge_line[idj];
          }
          if (idi != idj || idi >= E) {
            const double lo = range[i] < range[j] ? range[i] : range[j];
            double lom = lo - cs;
            if (lom < 0.0) lom = 0.0;
            if (d2 >= lom * lom) {
              const double fi = rmax ? rmax[i] : r, fj = rmax ? rmax[j] : r;
              const double fix = fi * dirs[2 * i], fiy = fi * dirs[2 * i + 1];
              const double fjx = fj * dirs[2 * j], fjy = fj * dirs[2 * j + 1];
              const double cf = (fjx - fix) * (ry - fiy) - (fjy - fiy) * (rx - fix);
              visible = 0;
              pending = cf >= 0.0;
            }
          }
        }
      }
    }
  }
  if (in) {
    if (visible) c |= 2;
    if ((c & 1) && !visible && !pending && d2 <= ro2 && (rx * hx + ry * hy) >= 0.0 &&
        in_half_fan(half, r, rx, ry, d2, ro2))
      c |= 4;
    cls[idx] = c;
    occ_flag[idx] = (c & 4) ? 1 : 0;
  }
  if (exact) {  // append the undecided cells (wave-aggregated; order is irrelevant, each cell is settled on its own)
    const unsigned long long pb = __ballot(pending);
    if (pb) {
      const int lane = threadIdx.x & 63;
      int base = 0;
      if (lane == 0) base = atomicAdd(n_amb, __popcll(pb));
      base = __shfl(base, 0);
      if (pending) amb[base + __popcll(pb & ((1ull << lane) - 1ull))] = idx;
    }
  }
  // block count of the occluded cells (first stage of the compaction, saves a launch)
  const unsigned long long b = __ballot(in && (c & 4));
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) blk[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// ------------------------------------------------------------------------------------------------ settle
// The cells the fan could not decide, settled by the reference's own rule at the cell centre: the shadow quads
// [v1, v2, v2 + 100 (v2 - ego), v1 + 100 (v1 - ego)] (helper_functions.py:79-96) and the obstacle occlusion polygons
// (:133-141) contain a point iff an occluding piece crosses the open segment ego -> point.  A workgroup per cell
// (grid-stride over the list), its threads share the soup like a ray workgroup; "t < 1" along the unnormalised
// direction is decided on tn and denom, no division.  Thread 0 writes the class and keeps the per-block counts of
// the occluded-cell compaction in step.
template <bool SKIP, int NW>
__global__ __launch_bounds__(64 * NW) void fo_settle_kernel(
    int E, const double *__restrict__ edges, const double *__restrict__ chunk_box, const uint8_t *__restrict__ eskip,
    int O, const double *__restrict__ ocorn, const uint8_t *__restrict__ oflags, double rx0, double ry0, double cs, int ix0, int iy0, int nx, double ex,
    double ey, double hx, double hy, double r, const double *__restrict__ half, const int32_t *__restrict__ amb,
    const int32_t *__restrict__ n_amb, uint8_t *__restrict__ cls, uint8_t *__restrict__ occ_flag,
    int32_t *__restrict__ blk, int ny, const double *__restrict__ ofar) {
  if ((int)blockIdx.x >= SETTLE_BLOCKS) {
    // Workgroups past the cell part: one per obstacle.  sensor_model.py:183 takes the obstacle grown by 5 mm out of
    // the visible area, so a centre the grid kernel found visible inside that skin loses the bit here (the undecided
    // cells get the same test below).  The bit is cleared with a 32-bit atomic on the word holding the class byte, so
    // two obstacles with overlapping skins cannot both count the cell.
    const int o = blockIdx.x - SETTLE_BLOCKS;
    if (!((oflags[o] & 1) && (oflags[o] & 2))) return;
    const double *q = ocorn + 8 * (size_t)o;
    const double xa = fmin(fmin(q[0], q[2]), fmin(q[4], q[6])) - 0.0072, xb = fmax(fmax(q[0], q[2]), fmax(q[4], q[6])) + 0.0072;
    const double ya = fmin(fmin(q[1], q[3]), fmin(q[5], q[7])) - 0.0072, yb = fmax(fmax(q[1], q[3]), fmax(q[5], q[7])) + 0.0072;
    int ixa = (int)floor((xa - rx0) / cs - 0.5) - ix0 - 1, ixb = (int)ceil((xb - rx0) / cs - 0.5) - ix0 + 1;
    int iya = (int)floor((ya - ry0) / cs - 0.5) - iy0 - 1, iyb = (int)ceil((yb - ry0) / cs - 0.5) - iy0 + 1;
    ixa = ixa < 0 ? 0 : ixa; iya = iya < 0 ? 0 : iya;
    ixb = ixb > nx - 1 ? nx - 1 : ixb; iyb = iyb > ny - 1 ? ny - 1 : iyb;
    if (ixb < ixa || iyb < iya) return;
    const int w = ixb - ixa + 1, h = iyb - iya + 1;
    unsigned int *words = (unsigned int *)cls;
    for (int t = threadIdx.x; t < w * h; t += 64 * NW) {
      const int ix = ixa + t % w, iy = iya + t / w;
      const int idx = iy * nx + ix;
      const int sh = 8 * (idx & 3);
      if (!((words[idx >> 2] >> sh) & 2u)) continue;
      const double px = rx0 + ((double)(ix0 + ix) + 0.5) * cs, py = ry0 + ((double)(iy0 + iy) + 0.5) * cs;
      if (!in_obstacle_skin(1, q, oflags + o, px, py)) continue;
      const unsigned int old = atomicAnd(&words[idx >> 2], ~(2u << sh));
      if (!((old >> sh) & 2u)) continue;  // another obstacle's workgroup took it first
      const double rx = px - ex, ry = py - ey;
      const double d2 = rx * rx + ry * ry, ro2 = (1.5 * r) * (1.5 * r);
      if (d2 <= ro2 && (rx * hx + ry * hy) >= 0.0 && in_half_fan(half, r, rx, ry, d2, ro2)) {
        atomicOr(&words[idx >> 2], 4u << sh);
        occ_flag[idx] = 1;
        atomicAdd(&blk[idx >> 8], 1);
      }
    }
    return;
  }
  const int n = *n_amb;
  for (int k = blockIdx.x; k < n; k += SETTLE_BLOCKS) {
    const int idx = amb[k];
    const int ix = idx % nx, iy = idx / nx;
    const int wx = ix0 + ix, wy = iy0 + iy;
    const double px = rx0 + ((double)wx + 0.5) * cs, py = ry0 + ((double)wy + 0.5) * cs;
    const double rx = px - ex, ry = py - ey;
    int hit = 0;
    constexpr int stride = 64 * NW;
    auto crosses = [&](double ax, double ay, double bx, double by) -> int {
      const double sx = bx - ax, sy = by - ay;
      const double denom = rx * sy - ry * sx;
      if (denom == 0.0) return 0;
      const double wx_ = ax - ex, wy_ = ay - ey;
      const double tn = wx_ * sy - wy_ * sx;
      const double un = wx_ * ry - wy_ * rx;
      return denom > 0.0 ? (tn >= 0.0 && un >= 0.0 && un <= denom && tn < denom)
                         : (tn <= 0.0 && un <= 0.0 && un >= denom && tn > denom);
    };
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nc = (E + 63) >> 6;
    for (int cb = wave * 64; cb < nc; cb += NW * 64) {  // culled like the ray scan (segment ego -> centre)
      const int cc = cb + lane;
      unsigned long long live = __ballot(cc < nc && !chunk_culled(chunk_box + 4 * (size_t)(cc < nc ? cc : 0), ex, ey, rx, ry, 1.0));
      while (live) {
        const int c = cb + __builtin_ctzll(live);
        live &= live - 1;
        const int gi = (c << 6) + lane;
        if (gi >= E) continue;
        if (SKIP && eskip[gi]) continue;
        const double *p = edges + 4 * (size_t)gi;
        hit |= crosses(p[0], p[1], p[2], p[3]);
      }
    }
    // a centre within 5 mm of an obstacle is not visible either (sensor_model.py:183): an obstacle per thread
    for (int o = threadIdx.x; o < O; o += stride) hit |= in_obstacle_skin(1, ocorn + 8 * (size_t)o, oflags + o, px, py);
    for (int gi = threadIdx.x; gi < 4 * O; gi += stride) {
      const int o = gi >> 2, sd = gi & 3, s2 = (sd + 1) & 3;
      if (!((oflags[o] & 1) && (oflags[o] & 2))) continue;
      const double *c = ocorn + 8 * (size_t)o;
      // the obstacle hides the centre unless the centre lies beyond the end of its shadow wedge (wedge_far_halfplane)
      if (ofar && ofar[3 * o] * px + ofar[3 * o + 1] * py + ofar[3 * o + 2] > 0.0) continue;
      hit |= crosses(c[2 * sd], c[2 * sd + 1], c[2 * s2], c[2 * s2 + 1]);
    }
    int blocked = __syncthreads_or(hit);
    if (threadIdx.x == 0) {
      uint8_t c = 1;
      if (!blocked) c |= 2;
      const double d2 = rx * rx + ry * ry;
      const double ro2 = (1.5 * r) * (1.5 * r);
      if (blocked && d2 <= ro2 && (rx * hx + ry * hy) >= 0.0 && in_half_fan(half, r, rx, ry, d2, ro2)) c |= 4;
      cls[idx] = c;
      if (c & 4) {
        occ_flag[idx] = 1;
        atomicAdd(&blk[idx >> 8], 1);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ future visibility
// An extension, NOT part of the reference: how much of the currently occluded area a candidate trajectory
// will come to see.  A workgroup per pose (trajectory m, every t_stride-th sample k): (1) the 64-piece chunks whose box
// lies within r of the pose are listed in LDS; (2) a thread per ray of a world-aligned full fan walks that list --
// per-ray box culling, and a wave whose rays all miss a chunk skips it -- keeping the first hit; (3) shoelace area of
// the polygon of hit points; (4) the cells of the current occluded set are tested against the fan (chord rule of the
// cell-grid kernel) and counted.  Ranges never leave LDS.
constexpr int FV_THREADS = 256;   // threads per pose; a thread walks RPT rays (tid, tid + 256, ...): RPT = ceil(n_rays / 256)
constexpr int FV_MAX_RPT = 3;     // <= 768 rays: the 720-ray fan of BASELINE configs[2] (0.5 deg) fits (round 6; 256 before)
constexpr int FV_BATCH = 48;      // 16-piece quarters staged in LDS at a time (24 KB)
template <int RPT>
__global__ __launch_bounds__(FV_THREADS) void fo_future_visibility_kernel(
    int T, const double *__restrict__ x, const double *__restrict__ y, int t_stride, int K, int n_rays,
    const double *__restrict__ dirs, double r, int E, const double *__restrict__ edges,
    const double *__restrict__ sub_box, int O, const double *__restrict__ ocorn, const uint8_t *__restrict__ oflags,
    const int32_t *__restrict__ occ_idx, const int32_t *__restrict__ n_occ_ptr, double rx0, double ry0, double cs,
    int ix0, int iy0, int nx, int32_t *__restrict__ revealed, double *__restrict__ area) {
  __shared__ double s_dir[2 * FV_THREADS * RPT];
  __shared__ double s_rng[FV_THREADS * RPT];
  __shared__ double s_seg[FV_BATCH][64];    // 16 pieces x (ax, ay, bx, by) per staged quarter
  __shared__ float s_box[FV_BATCH][4];      // their boxes relative to the pose (float, grown by 1 mm)
  __shared__ int s_ch[FV_BATCH];
  __shared__ int s_nob;
  __shared__ double s_ob[64][8];            // corner rows of the obstacles within reach (64 at a time)
  __shared__ double s_red[FV_THREADS / 64];
  __shared__ int s_cnt[FV_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = blockIdx.x / K, k = blockIdx.x % K;
  const double px = x[(size_t)m * T + (size_t)k * t_stride], py = y[(size_t)m * T + (size_t)k * t_stride];
  for (int i = tid; i < n_rays; i += FV_THREADS) { s_dir[2 * i] = dirs[2 * i]; s_dir[2 * i + 1] = dirs[2 * i + 1]; }
  __syncthreads();
  // ray u of this thread: index tid + 256 u
  bool ray[RPT];
  double dx[RPT], dy[RPT], best[RPT];
#pragma unroll
  for (int u = 0; u < RPT; ++u) {
    const int i = tid + u * FV_THREADS;
    ray[u] = i < n_rays;
    dx[u] = ray[u] ? s_dir[2 * i] : 1.0;
    dy[u] = ray[u] ? s_dir[2 * i + 1] : 0.0;
    best[u] = INFINITY;
  }
  // (1) + (2): the 16-piece quarters whose box lies within r of the pose are listed FV_BATCH at a time, staged in LDS
  // by the whole workgroup (one exposed round trip per batch), and every ray walks the staged list: box culling per
  // ray, 16 segment tests per surviving quarter, all operands LDS broadcasts
  const int nq = (E + 15) >> 4;
  const double rr = r + 1e-7;
  const size_t n_dbl = 4 * (size_t)E;
  auto in_reach = [&](int c) {
    const double *b = sub_box + 4 * (size_t)c;
    const double ddx = fmax(fmax(b[0] - px, px - b[2]), 0.0), ddy = fmax(fmax(b[1] - py, py - b[3]), 0.0);
    return ddx * ddx + ddy * ddy <= rr * rr;   // an empty box (inf, -inf) is never in reach
  };
  // rank of every quarter in reach (thread-major order): per-thread count, then an exclusive prefix over the workgroup
  int mine = 0;
  for (int c = tid; c < nq; c += FV_THREADS) mine += in_reach(c) ? 1 : 0;
  int incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  if (lane == 63) s_cnt[wave] = incl;
  __syncthreads();
  int offset = incl - mine, n_total = 0;
  for (int w = 0; w < FV_THREADS / 64; ++w) {
    if (w < wave) offset += s_cnt[w];
    n_total += s_cnt[w];
  }
  __syncthreads();
  for (int b0 = 0; b0 < n_total; b0 += FV_BATCH) {
    int rk = offset;
    for (int c = tid; c < nq; c += FV_THREADS)
      if (in_reach(c)) {
        if (rk >= b0 && rk < b0 + FV_BATCH) s_ch[rk - b0] = c;
        ++rk;
      }
    __syncthreads();
    const int nch = n_total - b0 < FV_BATCH ? n_total - b0 : FV_BATCH;
    // stage: four quarters per pass (thread -> quarter tid / 64, double tid % 64), loads back to back
    for (int base = 0; base < nch; base += 4) {
      const int slot = base + (tid >> 6);
      if (slot < nch) {
        const size_t g = 64 * (size_t)s_ch[slot] + (tid & 63);
        s_seg[slot][tid & 63] = g < n_dbl ? edges[g] : 0.0;
        if ((tid & 63) < 4) {  // box relative to the pose, in float, grown by 1 mm (>> float rounding at map scale)
          const int u = tid & 63;
          const double v = sub_box[4 * (size_t)s_ch[slot] + u] - ((u & 1) ? py : px);
          s_box[slot][u] = (float)v + (u < 2 ? -1e-3f : 1e-3f);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < RPT; ++u)
    if (ray[u]) {
      // the ray segment [0, r d] against the staged boxes, all in pose-relative float: bounding boxes, then "all four
      // corners on one side of the ray's line" (1 mm margins; culling is conservative, it never changes a result)
      const float fdx = (float)dx[u], fdy = (float)dy[u], fr = (float)r * 1.000001f;
      const float ex_ = fr * fdx, ey_ = fr * fdy;
      const float sx0 = fminf(0.0f, ex_) - 1e-3f, sx1 = fmaxf(0.0f, ex_) + 1e-3f;
      const float sy0 = fminf(0.0f, ey_) - 1e-3f, sy1 = fmaxf(0.0f, ey_) + 1e-3f;
      for (int q = 0; q < nch; ++q) {
        const float bx0 = s_box[q][0], by0 = s_box[q][1], bx1 = s_box[q][2], by1 = s_box[q][3];
        if (bx0 > sx1 || bx1 < sx0 || by0 > sy1 || by1 < sy0) continue;
        const float c00 = fdx * by0 - fdy * bx0, c10 = fdx * by0 - fdy * bx1;
        const float c01 = fdx * by1 - fdy * bx0, c11 = fdx * by1 - fdy * bx1;
        const float mm = 2e-3f;
        if ((c00 > mm && c10 > mm && c01 > mm && c11 > mm) || (c00 < -mm && c10 < -mm && c01 < -mm && c11 < -mm)) continue;
        const int e_first = s_ch[q] << 4;
        const int cnt = E - e_first < 16 ? E - e_first : 16;
        const double *buf = s_seg[q];
        for (int e = 0; e < cnt; ++e) {
          const double t = ray_segment(px, py, dx[u], dy[u], buf[4 * e], buf[4 * e + 1], buf[4 * e + 2], buf[4 * e + 3]);
          best[u] = t < best[u] ? t : best[u];
        }
      }
    }
    __syncthreads();
  }
  // obstacles within reach: corner rows staged in LDS (64 at a time)
  for (int base = 0; base < O; base += 64) {
    __syncthreads();
    if (tid == 0) s_nob = 0;
    __syncthreads();
    const int o = base + tid;
    if (tid < 64 && o < O && (oflags[o] & 1) && (oflags[o] & 2)) {
      const double *c = ocorn + 8 * (size_t)o;
      const double mx = 0.5 * (c[0] + c[4]), my = 0.5 * (c[1] + c[5]);
      const double hd2 = (c[0] - mx) * (c[0] - mx) + (c[1] - my) * (c[1] - my);
      const double d2c = (px - mx) * (px - mx) + (py - my) * (py - my);
      // nearer than r + half diagonal ((r + hd)^2 <= r^2 + r (1 + hd2) + hd2; a pure early-out)
      if (d2c <= r * r + r * (1.0 + hd2) + hd2 + 1e-6) {
        const int slot = atomicAdd(&s_nob, 1);
#pragma unroll
        for (int u = 0; u < 8; ++u) s_ob[slot][u] = c[u];
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < RPT; ++u)
    if (ray[u]) {
      for (int q = 0; q < s_nob; ++q) {
        const double *c = s_ob[q];
        {  // per-ray early-out: the obstacle's circumscribed circle misses the ray segment (margin as for the boxes)
          const double mx = 0.5 * (c[0] + c[4]) - px, my = 0.5 * (c[1] + c[5]) - py;
          const double hd2 = (c[0] - px - mx) * (c[0] - px - mx) + (c[1] - py - my) * (c[1] - py - my);
          const double cr = dx[u] * my - dy[u] * mx, al = dx[u] * mx + dy[u] * my;      // offset from the line, position along it
          const double lim = hd2 + 1e-6 * (1.0 + hd2);
          if (cr * cr > lim || (al < 0.0 && al * al > lim) || (al > r && (al - r) * (al - r) > lim)) continue;
        }
#pragma unroll
        for (int sd = 0; sd < 4; ++sd) {
          const int s2 = (sd + 1) & 3;
          const double t = ray_segment(px, py, dx[u], dy[u], c[2 * sd], c[2 * sd + 1], c[2 * s2], c[2 * s2 + 1]);
          best[u] = t < best[u] ? t : best[u];
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < RPT; ++u) {
    if (!(best[u] <= r)) best[u] = r;
    if (ray[u]) s_rng[tid + u * FV_THREADS] = best[u];
  }
  __syncthreads();
  // (3) shoelace area of the polygon of hit points: per-thread terms (its rays in ascending order), fixed-order tree sum
  double term = 0.0;
#pragma unroll
  for (int u = 0; u < RPT; ++u)
    if (ray[u]) {
      const int i = tid + u * FV_THREADS, j = (i + 1 == n_rays) ? 0 : i + 1;
      const double hix = s_rng[i] * s_dir[2 * i], hiy = s_rng[i] * s_dir[2 * i + 1];
      const double hjx = s_rng[j] * s_dir[2 * j], hjy = s_rng[j] * s_dir[2 * j + 1];
      term += hix * hjy - hjx * hiy;
    }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) term += __shfl_xor(term, off);
  if (lane == 0) s_red[wave] = term;
  // (4) occluded cells inside the fan
  const int n_occ = *n_occ_ptr;
  const double r2 = r * r;
  int cnt = 0;
  auto inside_fan = [&](int idx) -> int {
    const int wx = ix0 + idx % nx, wy = iy0 + idx / nx;
    const double cx = rx0 + ((double)wx + 0.5) * cs, cy = ry0 + ((double)wy + 0.5) * cs;
    const double qx = cx - px, qy = cy - py;
    if (qx * qx + qy * qy > r2) return 0;
    if (qx == 0.0 && qy == 0.0) return 1;
    const int i = fan_sector_uniform(n_rays, s_dir, qx, qy);
    if (i < 0) return 0;
    const int j = (i + 1 == n_rays) ? 0 : i + 1;
    const double hix = s_rng[i] * s_dir[2 * i], hiy = s_rng[i] * s_dir[2 * i + 1];
    const double hjx = s_rng[j] * s_dir[2 * j], hjy = s_rng[j] * s_dir[2 * j + 1];
    return ((hjx - hix) * (qy - hiy) - (hjy - hiy) * (qx - hix) >= 0.0) ? 1 : 0;
  };
  // four cell indices per thread in flight (the list is read once per pose; the loads are what the loop waits for)
  int ci = tid;
  for (; ci + 3 * FV_THREADS < n_occ; ci += 4 * FV_THREADS) {
    const int i0 = occ_idx[ci], i1 = occ_idx[ci + FV_THREADS], i2 = occ_idx[ci + 2 * FV_THREADS],
              i3 = occ_idx[ci + 3 * FV_THREADS];
    cnt += inside_fan(i0) + inside_fan(i1) + inside_fan(i2) + inside_fan(i3);
  }
  for (; ci < n_occ; ci += FV_THREADS) cnt += inside_fan(occ_idx[ci]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
  if (lane == 0) s_cnt[wave] = cnt;
  __syncthreads();
  if (tid == 0) {
    double a2 = 0.0;
    int total = 0;
    for (int w = 0; w < FV_THREADS / 64; ++w) { a2 += s_red[w]; total += s_cnt[w]; }
    area[blockIdx.x] = 0.5 * a2;
    revealed[blockIdx.x] = total;
  }
}

// ------------------------------------------------------------------------------------------------ compaction
// flags[n] (+ per-256 block counts from the producing kernel) -> ascending index list + count; two launches, no
// atomics (deterministic order)
__global__ __launch_bounds__(1024) void fo_flag_scan_kernel(int32_t *__restrict__ blk, int nb,
                                                            int32_t *__restrict__ total) {
  __shared__ int sh[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nb ? blk[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int add = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
      __syncthreads();
      sh[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < nb) blk[i] = carry + sh[threadIdx.x] - v;  // exclusive
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256) void fo_flag_scatter_kernel(const uint8_t *__restrict__ flags, int n,
                                                              const int32_t *__restrict__ blk,
                                                              int32_t *__restrict__ out) {
  __shared__ int wsum[4];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const bool f = idx < n && flags[idx];
  const unsigned long long b = __ballot(f);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int before = __popcll(b & ((1ull << lane) - 1ull));
  if (lane == 0) wsum[w] = __popcll(b);
  __syncthreads();
  int off = blk[blockIdx.x];
  for (int k = 0; k < w; ++k) off += wsum[k];
  if (f) out[off + before] = idx;
}

// candidate cells of the phantom sampler: occluded, on the front towards the visible area (or anywhere when
// all_occluded), ahead of the ego and within max_dist -- flags + per-block counts for the compaction that follows
struct SpawnFlagArgs {
  int on = 0;
  const uint8_t *cls = nullptr;
  int nx = 0, ny = 0, ix0 = 0, iy0 = 0, all_occluded = 0;
  double rx0 = 0, ry0 = 0, cs = 0, ex = 0, ey = 0, hx = 0, hy = 0, min_ahead = 0, max_dist = 0;
  uint8_t *flag = nullptr;
  int32_t *blk = nullptr;
};
// (whole 256-thread block; wsum: four ints of LDS)
__device__ __forceinline__ void spawn_flag_block(const SpawnFlagArgs &a, int *wsum) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const bool in = idx < a.nx * a.ny;
  const int ix = in ? idx % a.nx : 0, iy = in ? idx / a.nx : 0;
  const uint8_t *__restrict__ cls = a.cls;
  uint8_t f = 0;
  if (in && (cls[idx] & 4)) {
    int front = a.all_occluded;
    if (ix > 0 && (cls[idx - 1] & 2)) front = 1;
    if (ix + 1 < a.nx && (cls[idx + 1] & 2)) front = 1;
    if (iy > 0 && (cls[idx - a.nx] & 2)) front = 1;
    if (iy + 1 < a.ny && (cls[idx + a.nx] & 2)) front = 1;
    if (front) {
      const double px = a.rx0 + ((double)(a.ix0 + ix) + 0.5) * a.cs, py = a.ry0 + ((double)(a.iy0 + iy) + 0.5) * a.cs;
      const double rx = px - a.ex, ry = py - a.ey;
      if (!(rx * a.hx + ry * a.hy < a.min_ahead) && !(rx * rx + ry * ry > a.max_dist * a.max_dist)) f = 1;
    }
  }
  if (in) a.flag[idx] = f;
  const unsigned long long b = __ballot(f != 0);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) a.blk[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// One-launch variant for the usual window sizes (a few hundred blocks): every block sums the counts of the blocks
// before it itself (a few hundred L2-resident ints) instead of waiting for a separate scan launch; same output.
// sf.on (fo_step_run): the same launch also flags the phantom sampler's candidate cells for the compaction after this
// one -- into buffers of their own, this compaction's flags and counts are still being read by other blocks.
__global__ __launch_bounds__(256) void fo_flag_compact_kernel(const uint8_t *__restrict__ flags, int n,
                                                              const int32_t *__restrict__ cnt,
                                                              int32_t *__restrict__ out, int32_t *__restrict__ total,
                                                              SpawnFlagArgs sf) {
  __shared__ int wsum[4], psum[4];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const bool f = idx < n && flags[idx];
  const unsigned long long b = __ballot(f);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int before = __popcll(b & ((1ull << lane) - 1ull));
  int part = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) part += cnt[i];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o);
  if (lane == 0) { wsum[w] = __popcll(b); psum[w] = part; }
  __syncthreads();
  int off = psum[0] + psum[1] + psum[2] + psum[3];
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = off + wsum[0] + wsum[1] + wsum[2] + wsum[3];
  for (int k = 0; k < w; ++k) off += wsum[k];
  if (f) out[off + before] = idx;
  if (sf.on) {
    __syncthreads();   // (wsum is used again)
    spawn_flag_block(sf, wsum);
  }
}

// ------------------------------------------------------------------------------------------------ spawn sampling
__global__ __launch_bounds__(256) void fo_spawn_flag_kernel(SpawnFlagArgs a) {
  __shared__ int wsum[4];
  spawn_flag_block(a, wsum);
}

struct SpawnTypes {  // per pattern slot (j % 4): type code, speed, raw dims, inflated dims
  int32_t type[4];
  double speed[4], raw_l[4], raw_w[4], infl_l[4], infl_w[4];
};

// unit normal from (px, py) towards the closest point of the polyline path [N][2], as an angle in [0, 2 pi)
// (agent.py:475-481 + helper_functions.py:38-64); the whole wave calls this, the lanes share the search for the closest
// segment (per lane ascending i, first minimum; across lanes the smallest (d2, i)), the result is wave-uniform
__device__ __forceinline__ double heading_to_curve(int lane, int N, const double *__restrict__ path, double px, double py) {
  double best = INFINITY, qx = px, qy = py;
  int bi = 0x7fffffff;
  for (int i = lane; i + 1 < N; i += 64) {
    const double ax = path[2 * i], ay = path[2 * i + 1], bx = path[2 * i + 2], by = path[2 * i + 3];
    const double ex = bx - ax, ey = by - ay;
    const double l2 = ex * ex + ey * ey;
    double t = 0.0;
    if (l2 > 0.0) {
      t = ((px - ax) * ex + (py - ay) * ey) / l2;
      if (t < 0.0) t = 0.0;
      if (t > 1.0) t = 1.0;
    }
    const double cx = ax + t * ex, cy = ay + t * ey;
    const double d2 = (px - cx) * (px - cx) + (py - cy) * (py - cy);
    if (d2 < best) { best = d2; qx = cx; qy = cy; bi = i; }   // per lane: ascending i, first minimum
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {                    // across lanes: smallest (d2, i) = first minimum
    const double b2 = __shfl_xor(best, off);
    const int i2 = __shfl_xor(bi, off);
    if (b2 < best || (b2 == best && i2 < bi)) { best = b2; bi = i2; }
  }
  // (the winner's closest point from the lane that holds it -- segment i lives in lane i mod 64 -- instead of carrying it
  // through the six exchange steps)
  qx = __shfl(qx, bi & 63);
  qy = __shfl(qy, bi & 63);
  const double vx = qx - px, vy = qy - py;
  const double nn = sqrt(vx * vx + vy * vy);
  double ux = 1.0, uy = 0.0;
  if (nn > 0.0) { ux = vx / nn; uy = vy / nn; }
  double a = atan2(uy, ux);
  if (a < 0.0) a += 2.0 * M_PI;
  return a;
}

// evenly spaced pick of the candidates + heading per phantom: pedestrians -> unit vector to the closest point of the
// ego reference path (agent.py:475-481 + helper_functions.py:38-76); vehicles -> lane heading raster at their cell
// Phantom slot j of the step (the whole wave calls this; every result is wave-uniform): which candidate cell it takes
// -- the candidates at ranks floor(j n / max_agents) when there are more than slots -- its centre and its heading:
// vehicles on a lane follow the lane-heading raster, everything else heads for the closest point of the reference path
// (agent.py:475-481), the lanes sharing the search for the closest path segment.  Returns false for an unused slot.
__device__ __forceinline__ bool spawn_pick(int j, int lane, const int32_t *__restrict__ cand, int n, int nx, double rx0,
                                           double ry0, double cs, int ix0, int iy0, int max_agents, const SpawnTypes &st,
                                           int N, const double *__restrict__ path,
                                           const double *__restrict__ lane_yaw, int rnx, int rny, int &ci, double &px,
                                           double &py, double &a) {
  const int m = n < max_agents ? n : max_agents;
  ci = -1; px = 0.0; py = 0.0; a = 0.0;
  if (j >= m) return false;
  const int pick = (n <= max_agents) ? j : (int)(((long long)j * n) / max_agents);
  ci = cand[pick];
  const int wx = ix0 + ci % nx, wy = iy0 + ci / nx;
  px = rx0 + ((double)wx + 0.5) * cs;
  py = ry0 + ((double)wy + 0.5) * cs;
  const int type = st.type[j & 3];
  a = NAN;
  if (type != FO_TYPE_PEDESTRIAN && lane_yaw && wx >= 0 && wx < rnx && wy >= 0 && wy < rny)
    a = lane_yaw[(size_t)wy * rnx + wx];
  if (isnan(a)) a = heading_to_curve(lane, N, path, px, py);  // wave-uniform
  return true;
}

// route tables of the static map as the prediction kernels read them (fo_scene_set_routes)
struct RouteView {
  int RT = 0;                       // routes per lanelet in the table (0 = no table)
  const int32_t *first = nullptr, *count = nullptr;
  const double *xy = nullptr, *s = nullptr;
};

// where the prediction kernels write: the arrays fo_sweep_set_agents consumes (whole arrays; `slot` indexes them)
struct PredOut {
  double *pos, *yaw, *v, *cov, *shape, *raw;
  int32_t *type, *len;
};

// One prediction slot, written by one wave (every argument wave-uniform):
//   on && ll >= 0 && the lanelet ll has routes -> route r of that lanelet: the reference's min-var(v) Frenet sample (speed
//     held along the route, quintic lateral move to the nearest of d1 in {-0.5, 0, 0.5}; replaces route_planner.py:31-90
//     + frenetix_handler.py + agent.py:283-426); the prediction ends where the route ends;
//   on, otherwise -> r = 0: straight constant velocity along heading a0 (agent.py:451-536), r > 0 empty;
//   !on -> inactive (len = 0).
// table_on (fo_step_run): the slot's rows of the sweep's agent table as well, instead of a launch of fo_prep_agents_kernel
// (same function, same bits: fo_agent_rows.hpp).
__device__ __forceinline__ void spawn_write_slot(int lane, int slot, int r, bool on, double p0x, double p0y, double a0,
                                                 int atype, double spd, double raw_l, double raw_w, double infl_l,
                                                 double infl_w, int ll, const RouteView &rv, int T, double dt, double var0,
                                                 double factor, const PredOut &o, int table_on, const fo_agent_table_t &at, double vpow, double m_obs) {
  double *P = o.pos + (size_t)slot * T * 2, *Y = o.yaw + (size_t)slot * T, *V = o.v + (size_t)slot * T;
  double *C = o.cov + (size_t)slot * T * 4;
  // (what lane k < 64 writes for sample k, kept for the table rows at the end: r_*)
  double r_var = 0.0, r_px = 0.0, r_py = 0.0, r_yaw = 0.0, r_v = 0.0;
  for (int k = lane; k < T; k += 64) {
    const double var = var0 * (k == lane ? vpow : pow(factor, (double)k));  // agent.py:273; vpow = pow(factor, lane), worked out by the caller
    C[4 * k] = var; C[4 * k + 1] = 0.0; C[4 * k + 2] = 0.0; C[4 * k + 3] = var;
    if (k == lane) r_var = var;
  }
  if (lane == 0) {
    o.shape[2 * slot] = infl_l; o.shape[2 * slot + 1] = infl_w;
    o.raw[2 * slot] = raw_l; o.raw[2 * slot + 1] = raw_w;
    o.type[slot] = atype;
  }
  const int RT = rv.RT;
  const bool routed = on && ll >= 0 && r < RT && rv.count[(size_t)ll * RT] > 0;
  int L = 0;
  if (on && !routed && r == 0) {  // straight constant velocity
    const double a = a0;
    const double vx = __builtin_rint(spd * cos(a) * 1000.0) / 1000.0;  // round(v cos psi, 3)  (agent.py:492, Q12)
    const double vy = __builtin_rint(spd * sin(a) * 1000.0) / 1000.0;
    for (int k = lane; k < T; k += 64) {
      const double t = (double)k * dt;
      const double x_ = p0x + t * vx, y_ = p0y + t * vy;
      P[2 * k] = x_; P[2 * k + 1] = y_; Y[k] = a; V[k] = spd;
      if (k == lane) { r_px = x_; r_py = y_; r_yaw = a; r_v = spd; }
    }
    L = T;
  } else if (routed && rv.count[(size_t)ll * RT + r] >= 2) {
    const int nv = rv.count[(size_t)ll * RT + r];
    const double *qg = rv.xy + 2 * (size_t)rv.first[(size_t)ll * RT + r];
    const double *sg = rv.s + rv.first[(size_t)ll * RT + r];
    // a route of up to ROUTE_LDS vertices is read once, into LDS: the per-sample binary search below is then a chain of LDS
    // reads instead of global ones (the kernel is one chain of dependent round trips; this one had six links)
    constexpr int ROUTE_LDS = 256;
    __shared__ double rt_q[2 * ROUTE_LDS], rt_s[ROUTE_LDS];
    const bool staged = nv <= ROUTE_LDS;
    if (staged) {
      for (int i = lane; i < nv; i += 64) { rt_q[2 * i] = qg[2 * i]; rt_q[2 * i + 1] = qg[2 * i + 1]; rt_s[i] = sg[i]; }
      __syncthreads();
    }
    const double px = p0x, py = p0y;
    double s0 = 0.0, d0 = 0.0, d1 = -0.5, s_end = 0.0;
    const double t1 = 3.0;
    auto follow = [&](const double *q, const double *sq) {
    double best = INFINITY;
    int bi = 0x7fffffff;
    for (int i = lane; i + 1 < nv; i += 64) {  // closest point of the route: per lane ascending i, first minimum
      const double ax = q[2 * i], ay = q[2 * i + 1], ex = q[2 * i + 2] - ax, ey = q[2 * i + 3] - ay;
      const double l2 = ex * ex + ey * ey;
      double t = ((px - ax) * ex + (py - ay) * ey) / l2;
      if (t < 0.0) t = 0.0;
      if (t > 1.0) t = 1.0;
      const double cx = ax + t * ex, cy = ay + t * ey;
      const double d2 = (px - cx) * (px - cx) + (py - cy) * (py - cy);
      if (d2 < best) {
        const double l = sqrt(l2);
        best = d2; bi = i;
        s0 = sq[i] + t * l;
        d0 = ((px - cx) * (-ey) + (py - cy) * ex) / l;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {  // across lanes: smallest (d2, i)
      const double b2 = __shfl_xor(best, off);
      const int i2 = __shfl_xor(bi, off);
      if (b2 < best || (b2 == best && i2 < bi)) { best = b2; bi = i2; }
    }
    s0 = __shfl(s0, bi & 63);   // (from the lane that holds the winning segment)
    d0 = __shfl(d0, bi & 63);
    // the Frenet sample the reference keeps (agent.py:349-379 on the nine samples of frenetix_handler.py:82-105): end speed
    // v0, lateral target d1 = the one of {-0.5, 0, 0.5} nearest to d0 (first of equally near ones), quintic d(t) over 3 s
    d1 = -0.5;
    if (fabs(0.0 - d0) < fabs(d1 - d0)) d1 = 0.0;
    if (fabs(0.5 - d0) < fabs(d1 - d0)) d1 = 0.5;
    s_end = sq[nv - 1];
    for (int k = lane; k < T; k += 64) {
      const double tk = (double)k * dt, sk = s0 + spd * tk;
      if (sk > s_end) continue;
      int lo = 0, hi = nv - 2;  // largest m <= nv-2 with sq[m] <= sk
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (sq[mid] <= sk) lo = mid; else hi = mid - 1;
      }
      const int m = lo;
      const double ex = q[2 * m + 2] - q[2 * m], ey = q[2 * m + 3] - q[2 * m + 1];
      const double l = sqrt(ex * ex + ey * ey), ux = ex / l, uy = ey / l, loc = sk - sq[m];
      const double tau = tk < t1 ? tk / t1 : 1.0;
      const double dk = d0 + (d1 - d0) * (tau * tau * tau * (10.0 + tau * (-15.0 + 6.0 * tau)));
      const double dd = (d1 - d0) * (30.0 * tau * tau * (1.0 + tau * (-2.0 + tau))) / t1;
      const double x_ = q[2 * m] + loc * ux + dk * (-uy), y_ = q[2 * m + 1] + loc * uy + dk * ux;
      const double yw_ = atan2(uy, ux) + atan2(dd, spd), v_ = sqrt(spd * spd + dd * dd);
      P[2 * k] = x_; P[2 * k + 1] = y_; Y[k] = yw_; V[k] = v_;
      if (k == lane) { r_px = x_; r_py = y_; r_yaw = yw_; r_v = v_; }
    }
    };
    PRED_TICK(6);
    if (staged) follow(rt_q, rt_s);
    else follow(qg, sg);
    // number of samples on the route: sk is non-decreasing in k, so the valid samples are a prefix
    int cnt = 0;
    for (int k = 0; k < T; ++k) cnt += (s0 + spd * ((double)k * dt) > s_end) ? 0 : 1;
    L = cnt;
  }
  for (int k = lane; k < T; k += 64)
    if (k >= L) { P[2 * k] = 0.0; P[2 * k + 1] = 0.0; Y[k] = 0.0; V[k] = 0.0; }
  if (lane == 0) o.len[slot] = L;
  PRED_TICK(7);
  if (table_on && T <= 64) {
    // the slot's rows of the sweep's agent table from the values just written, a lane per sample (no read-back through
    // memory, one atomic for the agent's longest step)
    fo_agent_sample_t q;
    q.px = r_px; q.py = r_py; q.ppx = __shfl_up(r_px, 1); q.ppy = __shfl_up(r_py, 1); q.yaw = r_yaw; q.v = r_v;
    q.sxx = r_var; q.sxy = 0.0; q.syx = 0.0; q.syy = r_var;
    fo_agent_row_core<true>(lane < T, slot, lane, T, L, q, infl_l, infl_w, raw_l, raw_w, atype, at.ego_mass, at.hlA, at.hwA, at.hc, at.tab,
                            at.cst, at.aint, at.status, at.gen, m_obs);
  } else if (table_on) {
    __threadfence_block();
    __syncthreads();
    for (int k = lane; k < T; k += 64)
      fo_agent_row(slot * T + k, T, o.pos, o.yaw, o.v, o.cov, o.shape, o.raw, o.type, o.len, at.ego_mass, at.hlA, at.hwA, at.hc, at.tab,
                   at.cst, at.aint, at.status, at.gen);
  }
}

// Phantoms sampled in the occluded cells: one wave per prediction slot (j, r), r < R.  A vehicle whose cell lies on a
// lanelet with routes gets one prediction per candidate route; pedestrians / off-lane vehicles / no route table: one
// straight prediction in r = 0.  Slots of agents j >= n are inactive (len = 0).
__global__ __launch_bounds__(64) void fo_spawn_predict_kernel(
    int max_agents, int R, const int32_t *__restrict__ cand, const int32_t *__restrict__ n_cand, double rx0, double ry0,
    double cs, int n_path, const double *__restrict__ path, const double *__restrict__ lane_yaw, SpawnTypes st, int T,
    double dt, double var0, double factor, int nx, int ix0, int iy0, int rnx, int rny,
    const int32_t *__restrict__ lanelet_raster, RouteView rv, int32_t *__restrict__ cell, double *__restrict__ pos0,
    double *__restrict__ yaw0, int32_t *__restrict__ n_out, PredOut o, int table_on, fo_agent_table_t at) {
  const int lane = threadIdx.x;
  const int slot = blockIdx.x, j = slot / R, r = slot % R;
  // the pick of agent j (repeated by each of its R route slots: a few dozen path segments; saves a launch)
  PRED_TICK(0);
  const int n_c = *n_cand;
  // (the covariance growth factor of this lane's sample: a page of arithmetic with no input from memory -- here, under the
  // first round trip of the chain that follows)
  const double vpow = pow(factor, (double)lane);
  const double m_obs = table_on ? fo_obstacle_mass(st.type[j & 3], st.infl_l[j & 3] * st.infl_w[j & 3]) : -1.0;   // (as well)
  int ci;
  double p0x, p0y, a0;
  const bool on = spawn_pick(j, lane, cand, n_c, nx, rx0, ry0, cs, ix0, iy0, max_agents, st, n_path, path, lane_yaw, rnx,
                             rny, ci, p0x, p0y, a0);
  PRED_TICK(4);
  if (r == 0 && lane == 0) {
    cell[j] = ci; pos0[2 * j] = p0x; pos0[2 * j + 1] = p0y; yaw0[j] = a0;
    if (j == 0) *n_out = n_c < max_agents ? n_c : max_agents;
  }
  const int sdx = j & 3;
  int ll = -1;
  if (on && lanelet_raster && st.type[sdx] != FO_TYPE_PEDESTRIAN) {
    const int wx = ix0 + ci % nx, wy = iy0 + ci / nx;
    if (wx >= 0 && wx < rnx && wy >= 0 && wy < rny) ll = lanelet_raster[(size_t)wy * rnx + wx];
  }
  PRED_TICK(5);
  spawn_write_slot(lane, slot, r, on, p0x, p0y, a0, st.type[sdx], st.speed[sdx], st.raw_l[sdx], st.raw_w[sdx], st.infl_l[sdx],
                   st.infl_w[sdx], ll, rv, T, dt, var0, factor, o, table_on, at, vpow, m_obs);
  PRED_TICK(9);
}

int ensure_cells(fo_ctx *ctx, Scene *sc, size_t cells) {
  int rc;
  if ((rc = fo_reserve(ctx, &sc->d_flags, &sc->cap_cells, cells))) return rc;
  const size_t nb = (cells + 255) / 256 + 1;
  if ((rc = fo_reserve(ctx, &sc->d_blk, &sc->cap_blk, nb))) return rc;
  if (!sc->d_ncand) FO_HIP_TRY(ctx, hipMalloc((void **)&sc->d_ncand, sizeof(int32_t)));
  if ((rc = fo_reserve(ctx, &sc->d_amb, &sc->cap_amb, cells))) return rc;
  if (!sc->d_namb) {
    FO_HIP_TRY(ctx, hipMalloc((void **)&sc->d_namb, sizeof(int32_t)));
    FO_HIP_TRY(ctx, hipMemset(sc->d_namb, 0, sizeof(int32_t)));
  }
  return FO_OK;
}

// flags -> ascending indices (out) + count (d_total)
// (sf: candidate flags of the phantom sampler in the same launch, fo_step_run; *sf_done says whether that happened)
int compact(fo_ctx *ctx, Scene *sc, const uint8_t *flags, const int32_t *blk, int n, int32_t *out, int32_t *d_total,
            hipStream_t s, const SpawnFlagArgs *sf = nullptr, bool *sf_done = nullptr) {
  const int nb = (n + 255) / 256;  // block counts were written by the kernel that produced the flags
  if (sf_done) *sf_done = false;
  if (nb <= 2048) {
    SpawnFlagArgs a;
    if (sf) { a = *sf; if (sf_done) *sf_done = true; }
    hipLaunchKernelGGL(fo_flag_compact_kernel, dim3(nb), dim3(256), 0, s, flags, n, blk, out, d_total, a);
  } else {
    hipLaunchKernelGGL(fo_flag_scan_kernel, dim3(1), dim3(1024), 0, s, const_cast<int32_t *>(blk), nb, d_total);
    hipLaunchKernelGGL(fo_flag_scatter_kernel, dim3(nb), dim3(256), 0, s, flags, n, blk, out);
  }
  FO_HIP_TRY(ctx, hipGetLastError());
  return FO_OK;
}

}  // namespace

extern "C" {

void fo_scene_destroy_(fo_ctx *ctx) {
  if (!ctx || !ctx->scene) return;
  Scene *sc = (Scene *)ctx->scene;
  void *ptrs[] = {sc->d_vis32, sc->d_flags, sc->d_blk, sc->d_flags2, sc->d_blk2, sc->d_cand, sc->d_ncand, sc->d_amb, sc->d_namb, sc->d_rule_rec, sc->d_rule_lab, sc->d_rule_cnt, sc->d_ofar};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  map_release(sc->map);
  delete sc;
  ctx->scene = nullptr;
}

// Let `ctx` read the static map another context of the same device has uploaded (edge soup, chunk boxes, road and
// lane-heading rasters, route table) instead of holding a copy: several egos planning on one scenario on one GPU
// (BASELINE configs[4]).  The per-step workspace stays per context.  The map is reference counted; a later
// fo_scene_set_map on either context gives that context a map of its own again.
int fo_scene_share_map(fo_ctx *ctx, fo_ctx *owner) {
  if (!ctx || !owner || !owner->scene) return fo_fail(ctx, FO_E_ARG, "fo_scene_share_map: the owner has no map");
  if (ctx->device != owner->device) return fo_fail(ctx, FO_E_ARG, "fo_scene_share_map: contexts live on different devices");
  Scene *dst = scene_of(ctx), *src = (Scene *)owner->scene;
  if (dst->map == src->map) return FO_OK;
  if (src->map->P < 1) return fo_fail(ctx, FO_E_STATE, "fo_scene_share_map: the owner has not called fo_scene_set_map");
  map_release(dst->map);
  dst->map = src->map;
  dst->map->refs.fetch_add(1);
  return FO_OK;
}

int fo_scene_set_map(fo_ctx *ctx, int P, const int32_t *h_poly_off, const double *h_poly_xy, int E,
                     const double *h_edges, double cs, double margin, const double *h_lane_yaw_or_null,
                     const double *h_raster_origin_or_null, const int32_t *h_raster_dims_or_null) {
  if (!ctx) return FO_E_ARG;
  if (P < 1 || !h_poly_off || !h_poly_xy || E < 0 || (E > 0 && !h_edges) || !(cs > 0))
    return fo_fail(ctx, FO_E_ARG, "fo_scene_set_map: bad arguments (P=%d E=%d cs=%g)", P, E, cs);
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  Scene *sc = scene_of(ctx);
  if (sc->map->refs > 1) {  // shared with other contexts: this one gets a map of its own
    map_release(sc->map);
    sc->map = new StaticMap();
  }
  const int V = h_poly_off[P];
  double xmin = INFINITY, ymin = INFINITY, xmax = -INFINITY, ymax = -INFINITY;
  double *pbox = new double[4 * (size_t)P];
  for (int p = 0; p < P; ++p) {
    double bx0 = INFINITY, by0 = INFINITY, bx1 = -INFINITY, by1 = -INFINITY;
    for (int i = h_poly_off[p]; i < h_poly_off[p + 1]; ++i) {
      bx0 = fmin(bx0, h_poly_xy[2 * i]); bx1 = fmax(bx1, h_poly_xy[2 * i]);
      by0 = fmin(by0, h_poly_xy[2 * i + 1]); by1 = fmax(by1, h_poly_xy[2 * i + 1]);
    }
    pbox[4 * p] = bx0; pbox[4 * p + 1] = by0; pbox[4 * p + 2] = bx1; pbox[4 * p + 3] = by1;
    xmin = fmin(xmin, bx0); ymin = fmin(ymin, by0); xmax = fmax(xmax, bx1); ymax = fmax(ymax, by1);
  }
  if (h_raster_origin_or_null && h_raster_dims_or_null) {
    sc->map->x0 = h_raster_origin_or_null[0]; sc->map->y0 = h_raster_origin_or_null[1];
    sc->map->rnx = h_raster_dims_or_null[0]; sc->map->rny = h_raster_dims_or_null[1];
  } else {  // origin snapped to whole cells so that windows of different steps share cell boundaries
    sc->map->x0 = floor((xmin - margin) / cs) * cs;
    sc->map->y0 = floor((ymin - margin) / cs) * cs;
    sc->map->rnx = (int)ceil((xmax + margin - sc->map->x0) / cs);
    sc->map->rny = (int)ceil((ymax + margin - sc->map->y0) / cs);
  }
  if (sc->map->rnx < 1 || sc->map->rny < 1 || (long)sc->map->rnx * sc->map->rny > (1L << 28)) {
    delete[] pbox;
    return fo_fail(ctx, FO_E_ARG, "fo_scene_set_map: raster %d x %d out of range", sc->map->rnx, sc->map->rny);
  }
  sc->map->P = P; sc->map->E = E; sc->map->cs = cs;
  sc->map->R = 0;  // a new raster invalidates the route table
  if (sc->map->d_lanelet_raster) { (void)hipFree(sc->map->d_lanelet_raster); sc->map->d_lanelet_raster = nullptr; }
  for (void **p : {(void **)&sc->map->d_edges, (void **)&sc->map->d_raster, (void **)&sc->map->d_lane_yaw, (void **)&sc->map->d_chunk_box,
                   (void **)&sc->map->d_edge_line, (void **)&sc->map->d_sub_box, (void **)&sc->map->d_poly_off,
                   (void **)&sc->map->d_poly_xy, (void **)&sc->map->d_poly_box, (void **)&sc->map->d_left0, (void **)&sc->map->d_pred0,
                   (void **)&sc->map->d_adj_left, (void **)&sc->map->d_inter_off, (void **)&sc->map->d_inter_lanelet,
                   (void **)&sc->map->d_inter_kind, (void **)&sc->map->d_center_off, (void **)&sc->map->d_center_xy}) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
  }
  int32_t *d_off = nullptr;
  double *d_xy = nullptr, *d_box = nullptr;
  const size_t cells = (size_t)sc->map->rnx * sc->map->rny;
  FO_HIP_TRY(ctx, hipMalloc((void **)&d_off, sizeof(int32_t) * (P + 1)));
  FO_HIP_TRY(ctx, hipMalloc((void **)&d_xy, sizeof(double) * 2 * V));
  FO_HIP_TRY(ctx, hipMalloc((void **)&d_box, sizeof(double) * 4 * P));
  FO_HIP_TRY(ctx, hipMalloc((void **)&sc->map->d_raster, cells));
  FO_HIP_TRY(ctx, hipMalloc((void **)&sc->map->d_edges, sizeof(double) * 4 * (size_t)(E > 0 ? E : 1)));
  FO_HIP_TRY(ctx, hipMemcpy(d_off, h_poly_off, sizeof(int32_t) * (P + 1), hipMemcpyHostToDevice));
  FO_HIP_TRY(ctx, hipMemcpy(d_xy, h_poly_xy, sizeof(double) * 2 * V, hipMemcpyHostToDevice));
  FO_HIP_TRY(ctx, hipMemcpy(d_box, pbox, sizeof(double) * 4 * P, hipMemcpyHostToDevice));
  delete[] pbox;
  if (E > 0) FO_HIP_TRY(ctx, hipMemcpy(sc->map->d_edges, h_edges, sizeof(double) * 4 * (size_t)E, hipMemcpyHostToDevice));
  {  // bounding boxes of the 64-piece chunks the scans cull by (tight when the caller's order is spatially coherent)
    const int nc = (E + 63) / 64;
    double *cb = new double[4 * (size_t)(nc > 0 ? nc : 1)];
    for (int c = 0; c < nc; ++c) {
      double bx0 = INFINITY, by0 = INFINITY, bx1 = -INFINITY, by1 = -INFINITY;
      for (int e = 64 * c; e < E && e < 64 * (c + 1); ++e) {
        const double *q = h_edges + 4 * (size_t)e;
        bx0 = fmin(bx0, fmin(q[0], q[2])); bx1 = fmax(bx1, fmax(q[0], q[2]));
        by0 = fmin(by0, fmin(q[1], q[3])); by1 = fmax(by1, fmax(q[1], q[3]));
      }
      cb[4 * c] = bx0; cb[4 * c + 1] = by0; cb[4 * c + 2] = bx1; cb[4 * c + 3] = by1;
    }
    hipError_t e1 = hipMalloc((void **)&sc->map->d_chunk_box, sizeof(double) * 4 * (size_t)(nc > 0 ? nc : 1));
    if (e1 == hipSuccess && nc > 0)
      e1 = hipMemcpy(sc->map->d_chunk_box, cb, sizeof(double) * 4 * (size_t)nc, hipMemcpyHostToDevice);
    delete[] cb;
    FO_HIP_TRY(ctx, e1);
    double *sb = new double[16 * (size_t)(nc > 0 ? nc : 1)];
    for (int c = 0; c < 4 * nc; ++c) {  // quarter c of the table: pieces [16 c, 16 c + 16); empty quarters get an empty box
      double bx0 = INFINITY, by0 = INFINITY, bx1 = -INFINITY, by1 = -INFINITY;
      for (int e = 16 * c; e < E && e < 16 * (c + 1); ++e) {
        const double *q = h_edges + 4 * (size_t)e;
        bx0 = fmin(bx0, fmin(q[0], q[2])); bx1 = fmax(bx1, fmax(q[0], q[2]));
        by0 = fmin(by0, fmin(q[1], q[3])); by1 = fmax(by1, fmax(q[1], q[3]));
      }
      sb[4 * c] = bx0; sb[4 * c + 1] = by0; sb[4 * c + 2] = bx1; sb[4 * c + 3] = by1;
    }
    e1 = hipMalloc((void **)&sc->map->d_sub_box, sizeof(double) * 16 * (size_t)(nc > 0 ? nc : 1));
    if (e1 == hipSuccess && nc > 0)
      e1 = hipMemcpy(sc->map->d_sub_box, sb, sizeof(double) * 16 * (size_t)nc, hipMemcpyHostToDevice);
    delete[] sb;
    FO_HIP_TRY(ctx, e1);
  }
  hipLaunchKernelGGL(fo_raster_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, 0, P, d_off, d_xy, d_box,
                     sc->map->x0, sc->map->y0, cs, sc->map->rnx, sc->map->rny, sc->map->d_raster);
  FO_HIP_TRY(ctx, hipGetLastError());
  FO_HIP_TRY(ctx, hipDeviceSynchronize());
  sc->map->d_poly_off = d_off; sc->map->d_poly_xy = d_xy; sc->map->d_poly_box = d_box;   // kept: the rule families test points against them
  sc->map->n_inter = 0;
  if (h_lane_yaw_or_null) {
    FO_HIP_TRY(ctx, hipMalloc((void **)&sc->map->d_lane_yaw, sizeof(double) * cells));
    FO_HIP_TRY(ctx, hipMemcpy(sc->map->d_lane_yaw, h_lane_yaw_or_null, sizeof(double) * cells, hipMemcpyHostToDevice));
  }
  return FO_OK;
}

int fo_scene_set_routes(fo_ctx *ctx, int P, int R, const int32_t *h_first, const int32_t *h_count, int NV,
                        const double *h_xy, const double *h_s, const int32_t *h_lanelet_raster) {
  if (!ctx || !ctx->scene) return fo_fail(ctx, FO_E_STATE, "fo_scene_set_routes: call fo_scene_set_map first");
  Scene *sc = (Scene *)ctx->scene;
  if (P < 1 || R < 1 || !h_first || !h_count || NV < 0 || (NV > 0 && (!h_xy || !h_s)) || !h_lanelet_raster)
    return fo_fail(ctx, FO_E_ARG, "fo_scene_set_routes: bad arguments (P=%d R=%d NV=%d)", P, R, NV);
  for (int i = 0; i < P * R; ++i)
    if (h_count[i] < 0 || h_first[i] < 0 || (long)h_first[i] + h_count[i] > NV)
      return fo_fail(ctx, FO_E_ARG, "fo_scene_set_routes: route %d leaves the vertex table", i);
  // the tables belong to the map: while other contexts read it, freeing them here would pull them away under their
  // kernels (and change the route count R for every ego)
  if (sc->map->refs.load() > 1)
    return fo_fail(ctx, FO_E_STATE, "fo_scene_set_routes: the static map is shared (fo_scene_share_map); set the routes on "
                                    "the owner before sharing, or give this context its own map with fo_scene_set_map");
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  for (void **p : {(void **)&sc->map->d_route_first, (void **)&sc->map->d_route_count, (void **)&sc->map->d_lanelet_raster,
                   (void **)&sc->map->d_route_xy, (void **)&sc->map->d_route_s}) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
  }
  const size_t cells = (size_t)sc->map->rnx * sc->map->rny, nvs = (size_t)(NV > 0 ? NV : 1);
  FO_HIP_TRY(ctx, hipMalloc((void **)&sc->map->d_route_first, sizeof(int32_t) * P * R));
  FO_HIP_TRY(ctx, hipMalloc((void **)&sc->map->d_route_count, sizeof(int32_t) * P * R));
  FO_HIP_TRY(ctx, hipMalloc((void **)&sc->map->d_lanelet_raster, sizeof(int32_t) * cells));
  FO_HIP_TRY(ctx, hipMalloc((void **)&sc->map->d_route_xy, sizeof(double) * 2 * nvs));
  FO_HIP_TRY(ctx, hipMalloc((void **)&sc->map->d_route_s, sizeof(double) * nvs));
  FO_HIP_TRY(ctx, hipMemcpy(sc->map->d_route_first, h_first, sizeof(int32_t) * P * R, hipMemcpyHostToDevice));
  FO_HIP_TRY(ctx, hipMemcpy(sc->map->d_route_count, h_count, sizeof(int32_t) * P * R, hipMemcpyHostToDevice));
  FO_HIP_TRY(ctx, hipMemcpy(sc->map->d_lanelet_raster, h_lanelet_raster, sizeof(int32_t) * cells, hipMemcpyHostToDevice));
  if (NV > 0) {
    FO_HIP_TRY(ctx, hipMemcpy(sc->map->d_route_xy, h_xy, sizeof(double) * 2 * NV, hipMemcpyHostToDevice));
    FO_HIP_TRY(ctx, hipMemcpy(sc->map->d_route_s, h_s, sizeof(double) * NV, hipMemcpyHostToDevice));
  }
  sc->map->R = R;
  sc->map->n_lanelets = P;
  return FO_OK;
}

int fo_scene_set_shadow_length(fo_ctx *ctx, double length) {
  if (!ctx) return FO_E_ARG;
  if (length != length) return fo_fail(ctx, FO_E_ARG, "fo_scene_set_shadow_length: NaN");
  scene_of(ctx)->shadow_length = length;
  return FO_OK;
}

int fo_scene_map_info(fo_ctx *ctx, double *x0, double *y0, double *cs, int *nx, int *ny, int *n_edges) {
  if (!ctx || !ctx->scene) return fo_fail(ctx, FO_E_STATE, "fo_scene_map_info: no map set");
  Scene *sc = (Scene *)ctx->scene;
  if (x0) *x0 = sc->map->x0;
  if (y0) *y0 = sc->map->y0;
  if (cs) *cs = sc->map->cs;
  if (nx) *nx = sc->map->rnx;
  if (ny) *ny = sc->map->rny;
  if (n_edges) *n_edges = sc->map->E;
  return FO_OK;
}

int fo_scene_copy_raster(fo_ctx *ctx, uint8_t *h_out) {
  if (!ctx || !ctx->scene || !h_out) return fo_fail(ctx, FO_E_STATE, "fo_scene_copy_raster: no map set");
  Scene *sc = (Scene *)ctx->scene;
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  FO_HIP_TRY(ctx, hipMemcpy(h_out, sc->map->d_raster, (size_t)sc->map->rnx * sc->map->rny, hipMemcpyDeviceToHost));
  return FO_OK;
}

int fo_scene_set_edge_lines(fo_ctx *ctx, int E, const int32_t *h_line) {
  if (!ctx || !ctx->scene) return fo_fail(ctx, FO_E_STATE, "fo_scene_set_edge_lines: call fo_scene_set_map first");
  Scene *sc = (Scene *)ctx->scene;
  if (E != sc->map->E || (E > 0 && !h_line)) return fo_fail(ctx, FO_E_ARG, "fo_scene_set_edge_lines: E must match the map");
  if (sc->map->refs.load() > 1)
    return fo_fail(ctx, FO_E_STATE, "fo_scene_set_edge_lines: the static map is shared (fo_scene_share_map); set the labels "
                                    "on the owner before sharing, or give this context its own map with fo_scene_set_map");
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (sc->map->d_edge_line) { (void)hipFree(sc->map->d_edge_line); sc->map->d_edge_line = nullptr; }
  if (E == 0) return FO_OK;
  for (int e = 0; e < E; ++e)
    if (h_line[e] < 0 || h_line[e] >= E) return fo_fail(ctx, FO_E_ARG, "fo_scene_set_edge_lines: label out of [0, E)");
  FO_HIP_TRY(ctx, hipMalloc((void **)&sc->map->d_edge_line, sizeof(int32_t) * (size_t)E));
  FO_HIP_TRY(ctx, hipMemcpy(sc->map->d_edge_line, h_line, sizeof(int32_t) * (size_t)E, hipMemcpyHostToDevice));
  return FO_OK;
}

int fo_scene_fan(fo_ctx *ctx, int n_rays, double ego_yaw, double fov_deg, double r, int polygon_footprint,
                 double *d_dirs, double *d_rmax, double *d_half, void *stream) {
  if (!ctx) return FO_E_ARG;
  if (n_rays < 4 || !d_dirs || !(r > 0) || !(fov_deg > 0))
    return fo_fail(ctx, FO_E_ARG, "fo_scene_fan: bad arguments");
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int full = fov_deg >= 359.9;
  hipLaunchKernelGGL(fo_fan_kernel, dim3((n_rays + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_rays, ego_yaw,
                     fov_deg * (3.14159265358979323846 / 180.0), full, r, polygon_footprint, d_dirs, d_rmax, d_half);
  FO_HIP_TRY(ctx, hipGetLastError());
  return FO_OK;
}

// fo_scene_visibility; fan / sf (fo_step_run): the ray fan of fo_scene_fan inside the ray kernel, the candidate flags of
// fo_scene_spawn inside the compaction -- two launches less, the same bits
static int scene_visibility(fo_ctx *ctx, double ego_x, double ego_y, double head_x, double head_y, double r, int full_circle,
                            int exact_cells, int n_rays, const double *d_dirs, const double *d_rmax, const double *d_half,
                            const uint8_t *d_edge_skip, int O, const double *d_ocorn, const double *d_ocen,
                            const uint8_t *d_oflags, int win_ix0, int win_iy0, int win_nx, int win_ny, double *d_range,
                            int32_t *d_hit_id, double *d_ring, uint8_t *d_obst_vis, uint8_t *d_cls, int32_t *d_occ_idx,
                            int32_t *d_n_occ, void *stream, const FanArgs *fan_in, const SpawnFlagArgs *sf_in,
                            const fo_prep_args_t *prep_in = nullptr) {
  if (!ctx || !ctx->scene) return fo_fail(ctx, FO_E_STATE, "fo_scene_visibility: call fo_scene_set_map first");
  Scene *sc = (Scene *)ctx->scene;
  sc->cand_flags_ready = false;
  FanArgs fan;
  if (fan_in) fan = *fan_in;
  if (n_rays < 4 || !d_dirs || !d_range || !d_hit_id || O < 0 || (O > 0 && (!d_ocorn || !d_ocen || !d_oflags)) ||
      win_nx < 1 || win_ny < 1 || !d_cls || !d_occ_idx || !d_n_occ || !(r > 0))
    return fo_fail(ctx, FO_E_ARG, "fo_scene_visibility: bad arguments");
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = (hipStream_t)stream;
  int rc;
  const bool probes = O > 0 && d_obst_vis;
  if (probes && (size_t)O > sc->cap_vis32) {
    if ((rc = fo_reserve(ctx, &sc->d_vis32, &sc->cap_vis32, (size_t)O))) return rc;
    FO_HIP_TRY(ctx, hipMemsetAsync(sc->d_vis32, 0, sizeof(int32_t) * sc->cap_vis32, s));
  }
  const int cells = win_nx * win_ny;
  if (probes && O > cells) return fo_fail(ctx, FO_E_ARG, "fo_scene_visibility: more obstacles than window cells");
  if ((rc = ensure_cells(ctx, sc, (size_t)cells))) return rc;
  fo_prep_args_t prep = prep_in ? *prep_in : fo_prep_args_t();
  // one wave per ray / probe / undecided cell where the soup is a single group of chunk boxes (see fo_rays_kernel) and the
  // obstacle sides fit a wave; the tile table's spare workgroups then take horizon slices of two samples (a lane handles two
  // elements, as in the 256-thread shape)
  const bool one_wave = (sc->map->E + 63) / 64 <= 64 && 4 * O <= 64 && !fo_getenv(fo_env_any("FO_SCENE_"), "FO_SCENE_FIVE_WAVES");
  if (one_wave && prep.on) { prep.tz = prep.T > 2 ? 2 : prep.T; prep.nz = (prep.T + prep.tz - 1) / prep.tz; }
  const dim3 rgrid(n_rays + (probes ? 5 * O : 0) + prep.blocks()), rblock(64 * (one_wave ? 1 : RAY_WAVES));
#define FO_LAUNCH_RAYS(SK, NW_)                                                                                                       \
  hipLaunchKernelGGL((fo_rays_kernel<SK, NW_>), rgrid, rblock, 0, s, sc->map->E, sc->map->d_edges, sc->map->d_chunk_box, d_edge_skip, O, \
                     d_ocorn, d_ocen, d_oflags, ego_x, ego_y, n_rays, d_dirs, r, d_rmax, full_circle, d_range, d_hit_id, d_ring,        \
                     probes ? sc->d_vis32 : nullptr, sc->d_namb, fan, prep)
  if (d_edge_skip) { if (one_wave) FO_LAUNCH_RAYS(true, 1); else FO_LAUNCH_RAYS(true, RAY_WAVES); }
  else { if (one_wave) FO_LAUNCH_RAYS(false, 1); else FO_LAUNCH_RAYS(false, RAY_WAVES); }
#undef FO_LAUNCH_RAYS
  // where the obstacles' shadows end: worked out by the grid kernel (a thread per obstacle), read by the settle kernel
  double *far = nullptr;
  if (exact_cells && O > 0 && sc->shadow_length > 0.0 && sc->shadow_length < INFINITY) {
    if ((size_t)16 * O > (size_t)(cells + 255) / 256 * 256)
      return fo_fail(ctx, FO_E_ARG, "fo_scene_visibility: more than a sixteenth as many obstacles as window cells");
    if ((rc = fo_reserve(ctx, &sc->d_ofar, &sc->cap_ofar, (size_t)3 * O))) return rc;
    far = sc->d_ofar;
  }
  hipLaunchKernelGGL(fo_grid_kernel, dim3((cells + 255) / 256), dim3(256), 0, s, sc->map->d_raster, sc->map->rnx, sc->map->rny, sc->map->x0,
                     sc->map->y0, sc->map->cs, win_ix0, win_iy0, win_nx, win_ny, ego_x, ego_y, head_x, head_y, r, full_circle,
                     n_rays, d_dirs, d_range, d_cls, sc->d_flags, sc->d_blk, probes ? O : 0, sc->d_vis32,
                     probes ? d_obst_vis : nullptr, exact_cells ? 1 : 0, sc->map->E, d_hit_id, d_rmax, sc->d_amb,
                     sc->d_namb, d_half, sc->map->d_edge_line, d_ocorn, d_oflags, sc->shadow_length, far, O,
                     probes ? fan.vis_host : nullptr);
  if (exact_cells) {
    if ((uintptr_t)d_cls & 3) return fo_fail(ctx, FO_E_ARG, "fo_scene_visibility: d_cls must be 4-byte aligned");
    const dim3 sgrid(SETTLE_BLOCKS + O), sblock(64 * (one_wave ? 1 : RAY_WAVES));
#define FO_LAUNCH_SETTLE(SK, NW_)                                                                                                       \
  hipLaunchKernelGGL((fo_settle_kernel<SK, NW_>), sgrid, sblock, 0, s, sc->map->E, sc->map->d_edges, sc->map->d_chunk_box, d_edge_skip, O, \
                     d_ocorn, d_oflags, sc->map->x0, sc->map->y0, sc->map->cs, win_ix0, win_iy0, win_nx, ego_x, ego_y, head_x, head_y, r, \
                     d_half, sc->d_amb, sc->d_namb, d_cls, sc->d_flags, sc->d_blk, win_ny, far)
    if (d_edge_skip) { if (one_wave) FO_LAUNCH_SETTLE(true, 1); else FO_LAUNCH_SETTLE(true, RAY_WAVES); }
    else { if (one_wave) FO_LAUNCH_SETTLE(false, 1); else FO_LAUNCH_SETTLE(false, RAY_WAVES); }
#undef FO_LAUNCH_SETTLE
  }
  FO_HIP_TRY(ctx, hipGetLastError());
  if (sf_in) {
    if ((rc = fo_reserve(ctx, &sc->d_flags2, &sc->cap_cells2, (size_t)cells))) return rc;
    if ((rc = fo_reserve(ctx, &sc->d_blk2, &sc->cap_blk2, (size_t)(cells + 255) / 256 + 1))) return rc;
    SpawnFlagArgs sf = *sf_in;
    sf.on = 1; sf.cls = d_cls; sf.nx = win_nx; sf.ny = win_ny; sf.ix0 = win_ix0; sf.iy0 = win_iy0;
    sf.rx0 = sc->map->x0; sf.ry0 = sc->map->y0; sf.cs = sc->map->cs; sf.ex = ego_x; sf.ey = ego_y; sf.hx = head_x; sf.hy = head_y;
    sf.flag = sc->d_flags2; sf.blk = sc->d_blk2;
    return compact(ctx, sc, sc->d_flags, sc->d_blk, cells, d_occ_idx, d_n_occ, s, &sf, &sc->cand_flags_ready);
  }
  return compact(ctx, sc, sc->d_flags, sc->d_blk, cells, d_occ_idx, d_n_occ, s);
}

int fo_scene_visibility(fo_ctx *ctx, double ego_x, double ego_y, double head_x, double head_y, double r, int full_circle,
                        int exact_cells, int n_rays, const double *d_dirs, const double *d_rmax, const double *d_half,
                        const uint8_t *d_edge_skip, int O, const double *d_ocorn, const double *d_ocen,
                        const uint8_t *d_oflags, int win_ix0, int win_iy0, int win_nx, int win_ny, double *d_range,
                        int32_t *d_hit_id, double *d_ring, uint8_t *d_obst_vis, uint8_t *d_cls, int32_t *d_occ_idx,
                        int32_t *d_n_occ, void *stream) {
  return scene_visibility(ctx, ego_x, ego_y, head_x, head_y, r, full_circle, exact_cells, n_rays, d_dirs, d_rmax, d_half, d_edge_skip,
                          O, d_ocorn, d_ocen, d_oflags, win_ix0, win_iy0, win_nx, win_ny, d_range, d_hit_id, d_ring, d_obst_vis,
                          d_cls, d_occ_idx, d_n_occ, stream, nullptr, nullptr);
}

int fo_scene_future_visibility(fo_ctx *ctx, int M, int T, const double *d_x, const double *d_y, int t_stride, int n_rays,
                               const double *d_dirs, double r, int O, const double *d_ocorn, const uint8_t *d_oflags,
                               const int32_t *d_occ_idx, const int32_t *d_n_occ, int win_ix0, int win_iy0, int win_nx,
                               int32_t *d_revealed, double *d_area, void *stream) {
  if (!ctx || !ctx->scene) return fo_fail(ctx, FO_E_STATE, "fo_scene_future_visibility: call fo_scene_set_map first");
  Scene *sc = (Scene *)ctx->scene;
  if (M < 0 || T < 1 || !d_x || !d_y || t_stride < 1 || n_rays < 4 || n_rays > FV_THREADS * FV_MAX_RPT || !d_dirs || !(r > 0) || O < 0 ||
      (O > 0 && (!d_ocorn || !d_oflags)) || !d_occ_idx || !d_n_occ || win_nx < 1 || !d_revealed || !d_area)
    return fo_fail(ctx, FO_E_ARG, "fo_scene_future_visibility: bad arguments (4 <= n_rays <= %d)", FV_THREADS * FV_MAX_RPT);
  if (M == 0) return FO_OK;
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int K = (T + t_stride - 1) / t_stride;
  // a thread per ray up to 256 rays (the form of rounds 1-5, unchanged), two or three rays per thread beyond
#define FO_LAUNCH_FV(RPT_)                                                                                                          \
  hipLaunchKernelGGL(fo_future_visibility_kernel<RPT_>, dim3((unsigned)((size_t)M * K)), dim3(FV_THREADS), 0, (hipStream_t)stream, \
                     T, d_x, d_y, t_stride, K, n_rays, d_dirs, r, sc->map->E, sc->map->d_edges, sc->map->d_sub_box, O, d_ocorn,        \
                     d_oflags, d_occ_idx, d_n_occ, sc->map->x0, sc->map->y0, sc->map->cs, win_ix0, win_iy0, win_nx, d_revealed, d_area)
  const int rpt = (n_rays + FV_THREADS - 1) / FV_THREADS;
  if (rpt == 1) FO_LAUNCH_FV(1); else if (rpt == 2) FO_LAUNCH_FV(2); else FO_LAUNCH_FV(3);
#undef FO_LAUNCH_FV
  FO_HIP_TRY(ctx, hipGetLastError());
  return FO_OK;
}

// fo_scene_spawn; at (fo_step_run): the prediction kernel also writes its slots' rows of the sweep's agent table, and the
// candidate flags may already be there (scene_visibility with sf)
static int scene_spawn(fo_ctx *ctx, const uint8_t *d_cls, int win_ix0, int win_iy0, int win_nx, int win_ny, double ego_x,
                   double ego_y, double head_x, double head_y, double min_ahead, double max_dist, int all_occluded,
                   int max_agents, int routes, const int32_t *type4, const double *speed4, const double *raw_l4, const double *raw_w4,
                   const double *infl_l4, const double *infl_w4, int n_path, const double *d_path, int T, double dt,
                   double var0, double var_factor, int32_t *d_cell, double *d_pos0, double *d_yaw0, int32_t *d_n,
                   double *d_pos, double *d_yaw, double *d_v, double *d_cov, double *d_shape, double *d_raw_dims,
                   int32_t *d_type, int32_t *d_len, void *stream, const fo_agent_table_t *at) {
  if (!ctx || !ctx->scene) return fo_fail(ctx, FO_E_STATE, "fo_scene_spawn: call fo_scene_set_map first");
  Scene *sc = (Scene *)ctx->scene;
  if (!d_cls || max_agents < 1 || !type4 || !speed4 || !raw_l4 || !raw_w4 || !infl_l4 || !infl_w4 || n_path < 2 ||
      !d_path || T < 1 || !d_cell || !d_pos0 || !d_yaw0 || !d_n || !d_pos || !d_yaw || !d_v || !d_cov || !d_shape ||
      !d_raw_dims || !d_type || !d_len)
    return fo_fail(ctx, FO_E_ARG, "fo_scene_spawn: bad arguments");
  if (routes < 0 || (routes > 0 && !sc->map->d_lanelet_raster))
    return fo_fail(ctx, FO_E_STATE, "fo_scene_spawn: routes = %d needs fo_scene_set_routes first", routes);
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = (hipStream_t)stream;
  const int cells = win_nx * win_ny;
  int rc;
  if ((rc = ensure_cells(ctx, sc, (size_t)cells))) return rc;
  if ((rc = fo_reserve(ctx, &sc->d_cand, &sc->cap_cand, (size_t)cells))) return rc;
  if (at && sc->cand_flags_ready) {   // flagged during the compaction of the visibility stage
    sc->cand_flags_ready = false;
    if ((rc = compact(ctx, sc, sc->d_flags2, sc->d_blk2, cells, sc->d_cand, sc->d_ncand, s))) return rc;
  } else {
    SpawnFlagArgs sf;
    sf.on = 1; sf.cls = d_cls; sf.nx = win_nx; sf.ny = win_ny; sf.ix0 = win_ix0; sf.iy0 = win_iy0; sf.all_occluded = all_occluded ? 1 : 0;
    sf.rx0 = sc->map->x0; sf.ry0 = sc->map->y0; sf.cs = sc->map->cs; sf.ex = ego_x; sf.ey = ego_y; sf.hx = head_x; sf.hy = head_y;
    sf.min_ahead = min_ahead; sf.max_dist = max_dist; sf.flag = sc->d_flags; sf.blk = sc->d_blk;
    hipLaunchKernelGGL(fo_spawn_flag_kernel, dim3((cells + 255) / 256), dim3(256), 0, s, sf);
    if ((rc = compact(ctx, sc, sc->d_flags, sc->d_blk, cells, sc->d_cand, sc->d_ncand, s))) return rc;
  }
  SpawnTypes st;
  for (int i = 0; i < 4; ++i) {
    st.type[i] = type4[i]; st.speed[i] = speed4[i]; st.raw_l[i] = raw_l4[i]; st.raw_w[i] = raw_w4[i];
    st.infl_l[i] = infl_l4[i]; st.infl_w[i] = infl_w4[i];
  }
  const int R = routes > 0 ? routes : 1;
  RouteView rv;
  if (routes > 0) { rv.RT = sc->map->R; rv.first = sc->map->d_route_first; rv.count = sc->map->d_route_count; rv.xy = sc->map->d_route_xy; rv.s = sc->map->d_route_s; }
  PredOut po{d_pos, d_yaw, d_v, d_cov, d_shape, d_raw_dims, d_type, d_len};
  hipLaunchKernelGGL(fo_spawn_predict_kernel, dim3(max_agents * R), dim3(64), 0, s, max_agents, R, sc->d_cand, sc->d_ncand,
                     sc->map->x0, sc->map->y0, sc->map->cs, n_path, d_path, sc->map->d_lane_yaw, st, T, dt, var0, var_factor, win_nx, win_ix0,
                     win_iy0, sc->map->rnx, sc->map->rny, routes > 0 ? sc->map->d_lanelet_raster : nullptr, rv, d_cell, d_pos0, d_yaw0,
                     d_n, po, at ? 1 : 0, at ? *at : fo_agent_table_t());
  FO_HIP_TRY(ctx, hipGetLastError());
  return FO_OK;
}

int fo_scene_spawn(fo_ctx *ctx, const uint8_t *d_cls, int win_ix0, int win_iy0, int win_nx, int win_ny, double ego_x,
                   double ego_y, double head_x, double head_y, double min_ahead, double max_dist, int all_occluded,
                   int max_agents, int routes, const int32_t *type4, const double *speed4, const double *raw_l4, const double *raw_w4,
                   const double *infl_l4, const double *infl_w4, int n_path, const double *d_path, int T, double dt,
                   double var0, double var_factor, int32_t *d_cell, double *d_pos0, double *d_yaw0, int32_t *d_n,
                   double *d_pos, double *d_yaw, double *d_v, double *d_cov, double *d_shape, double *d_raw_dims,
                   int32_t *d_type, int32_t *d_len, void *stream) {
  return scene_spawn(ctx, d_cls, win_ix0, win_iy0, win_nx, win_ny, ego_x, ego_y, head_x, head_y, min_ahead, max_dist, all_occluded,
                     max_agents, routes, type4, speed4, raw_l4, raw_w4, infl_l4, infl_w4, n_path, d_path, T, dt, var0, var_factor,
                     d_cell, d_pos0, d_yaw0, d_n, d_pos, d_yaw, d_v, d_cov, d_shape, d_raw_dims, d_type, d_len, stream, nullptr);
}

// the scene stages of a planning step in their fused form (fo_step_run, fo_api.hip): fan inside the ray kernel, candidate
// flags inside the first compaction, the sweep's agent table written by the prediction kernels
int fo_scene_rule_agents_(fo_ctx *ctx, int max_points, const double *d_points, const int32_t *d_n_points, int routes,
                          const fo_rule_agent_types_t *types, int n_path, const double *d_path, int T, double dt, double var0,
                          double var_factor, int slot0, int agent0, double *d_pos0, double *d_yaw0, double *d_pos, double *d_yaw,
                          double *d_v, double *d_cov, double *d_shape, double *d_raw_dims, int32_t *d_type, int32_t *d_len,
                          void *stream, const fo_agent_table_t *at);   // fo_spawn_rules.hpp

// h_mirror as a device pointer if the kernels can fill it themselves (see FanArgs::hit_host), else null (fo_api.hip copies)
void *fo_step_direct_mirror_(fo_ctx *ctx, const fo_step_t *p) {
  if (!p->h_mirror || !p->d_mirror || p->O < 1 || !p->d_obst_vis) return nullptr;
  if (p->d_mirror != (const void *)p->d_hit_id || (const uint8_t *)p->d_obst_vis != (const uint8_t *)p->d_hit_id + sizeof(int32_t) * (size_t)p->n_rays ||
      p->mirror_bytes != (int64_t)(sizeof(int32_t) * (size_t)p->n_rays + (size_t)p->O))
    return nullptr;
  // (FO_STEP_MIRROR_COPY=1: the copy command instead, for A/B runs and the test that compares the two paths -- looked at on
  // every call like the other knobs, fo_ctx.hpp)
  const char *e = fo_getenv(fo_env_any("FO_STEP_"), "FO_STEP_MIRROR_COPY");
  if (e && e[0] == '1') return nullptr;
  // the device-side address of the pinned block, cached by (address, size): a block freed and registered again at the same
  // address with another size is another mapping
  if (ctx->mirror_host != p->h_mirror || ctx->mirror_bytes != p->mirror_bytes) {
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, p->h_mirror, 0) != hipSuccess) { (void)hipGetLastError(); d = nullptr; }
    ctx->mirror_host = p->h_mirror;
    ctx->mirror_bytes = p->mirror_bytes;
    ctx->mirror_dev = d;
  }
  return ctx->mirror_dev;
}

int fo_scene_step_(fo_ctx *ctx, const fo_step_t *p, const fo_agent_table_t *at, const fo_prep_args_t *prep, void *stream) {
  if (!ctx || !p) return FO_E_ARG;
  if (p->n_rays < 4 || !p->d_dirs || !(p->r > 0) || !(p->fov_deg > 0)) return fo_fail(ctx, FO_E_ARG, "fo_scene_fan: bad arguments");
  const bool cells = p->spawn_mode != FO_SPAWN_RULES, rules = p->spawn_mode != FO_SPAWN_CELLS;
  FanArgs fan;
  fan.on = 1; fan.full = p->fov_deg >= 359.9; fan.polygon = p->polygon_footprint; fan.yaw = p->ego_yaw;
  fan.fov = p->fov_deg * (3.14159265358979323846 / 180.0);
  fan.dirs = p->d_dirs; fan.rmax = p->d_rmax; fan.half = p->d_half;
  // the step's mirror, when it is exactly the pair (hit ids | visibility flags) the interface allocates back to back: stored
  // by the kernels themselves (fo_api.hip then only records the event behind the step)
  if (void *hm = fo_step_direct_mirror_(ctx, p)) {
    fan.hit_host = (int32_t *)hm;
    fan.vis_host = (uint8_t *)hm + sizeof(int32_t) * (size_t)p->n_rays;
  }
  SpawnFlagArgs sf;
  sf.all_occluded = p->all_occluded ? 1 : 0; sf.min_ahead = p->min_ahead; sf.max_dist = p->max_dist;
  int rc;
  if ((rc = scene_visibility(ctx, p->ego_x, p->ego_y, p->head_x, p->head_y, p->r, p->full_circle, p->exact_cells, p->n_rays, p->d_dirs,
                             p->d_rmax, p->d_half, p->d_edge_skip, p->O, p->d_ocorn, p->d_ocen, p->d_oflags, p->win_ix0, p->win_iy0,
                             p->win_nx, p->win_ny, p->d_range, p->d_hit_id, p->d_ring, p->d_obst_vis, p->d_cls, p->d_occ_idx,
                             p->d_n_occ, stream, &fan, cells ? &sf : nullptr, prep))) return rc;
  if (cells && (rc = scene_spawn(ctx, p->d_cls, p->win_ix0, p->win_iy0, p->win_nx, p->win_ny, p->ego_x, p->ego_y, p->head_x, p->head_y,
                                 p->min_ahead, p->max_dist, p->all_occluded, p->max_agents, p->routes, p->type4, p->speed4, p->raw_l4,
                                 p->raw_w4, p->infl_l4, p->infl_w4, p->n_path, p->d_path, p->T_agents, p->dt, p->var0, p->var_factor,
                                 p->d_cell, p->d_pos0, p->d_yaw0, p->d_n, p->d_pos, p->d_yaw, p->d_v, p->d_cov, p->d_shape,
                                 p->d_raw_dims, p->d_type, p->d_len, stream, at))) return rc;
  if (rules) {
    const int R = p->routes > 0 ? p->routes : 1, cell_agents = cells ? p->max_agents : 0;
    if ((rc = fo_scene_spawn_rules(ctx, p->d_cls, p->win_ix0, p->win_iy0, p->win_nx, p->win_ny, p->n_path6, p->d_path6, p->O, p->d_ocorn,
                                   p->d_ocen, p->d_oyaw, p->d_odims, p->d_oflags, p->d_obst_vis, &p->rule, p->max_rule_points,
                                   p->d_rule_points, p->d_n_rule_points, stream))) return rc;
    if ((rc = fo_scene_rule_agents_(ctx, p->max_rule_points, p->d_rule_points, p->d_n_rule_points, p->routes, &p->rule_types, p->n_path,
                                    p->d_path, p->T_agents, p->dt, p->var0, p->var_factor, cell_agents * R, cell_agents, p->d_pos0,
                                    p->d_yaw0, p->d_pos, p->d_yaw, p->d_v, p->d_cov, p->d_shape, p->d_raw_dims, p->d_type, p->d_len,
                                    stream, at))) return rc;
  }
  return FO_OK;
}

#if FO_PRED_TRACE
int fo_debug_pred_ticks(long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pred_ticks), sizeof(long long) * 16); }
int fo_debug_ray_ticks(long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ray_ticks), sizeof(long long) * 16); }
#endif

int fo_scene_candidate_count(fo_ctx *ctx, int32_t *h_n, void *stream) {
  if (!ctx || !ctx->scene || !h_n) return FO_E_ARG;
  Scene *sc = (Scene *)ctx->scene;
  FO_HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
  FO_HIP_TRY(ctx, hipMemcpy(h_n, sc->d_ncand, sizeof(int32_t), hipMemcpyDeviceToHost));
  return FO_OK;
}

}  // extern "C"

#include "fo_spawn_rules.hpp"   // the reference's three spawn rule families on the cell classes (same translation unit: they read the map)
