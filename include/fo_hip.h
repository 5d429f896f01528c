/* fo_hip.h -- C ABI of the MI355X (gfx950) implementation of the Frenetix-Occlusion per-timestep hot path.
 *
 * Shared library: frenetix-occlusion_amd/lib/libfo_hip.so  (built by __graft_entry__.build()).
 * All pointers named d_* are DEVICE pointers (HBM), caller-owned, and must stay valid until the work queued on
 * `stream` (a hipStream_t passed as void*; NULL = the default stream) has completed.  No torch types, no C++
 * types, no exceptions cross this boundary: every entry point returns 0 or a negative FO_E_* code and
 * fo_last_error(ctx) gives the message.  One context per ego vehicle per GPU; a context is not re-entrant
 * (same ownership model as one FOInterface instance, /root/reference/frenetix_occlusion/interface.py:85-90).
 *
 * Citations "ref:" are relative to /root/reference/frenetix_occlusion/.
 *
 * Array layouts (float64 unless noted):
 *   trajectories  x,y,theta,v,a : [M][T]   row = one candidate trajectory (trajectory.cartesian.*, ref: interface.py:216)
 *   agent predictions           : pos [A][Ta][2], yaw [A][Ta], v [A][Ta], cov [A][Ta][4] (xx,xy,yx,yy),
 *                                 shape [A][2] = prediction['shape'] (inflated length,width; ref: agent.py:404-409,523-524),
 *                                 raw_dims [A][2] = agent.shape (ref: agent.py:216), type int32 [A] (FO_TYPE_*),
 *                                 len int32 [A] = number of valid samples of that prediction (1..Ta; 0 = inactive slot)
 *   outputs, trajectory index fastest (lane = trajectory):
 *     cost   [M][FO_NC]                 per-trajectory cost vector (the unit all-gathered across GPUs)
 *     safe   uint8 [M]                  safety_assessment of metric.py:50-100
 *     pair_f [FO_NPF][A][M]             per (trajectory, agent) scalars
 *     pair_i int32 [FO_NPI][A][M]
 *     lists  FO_NL A (T-1) M doubles    per-timestep lists of hr.py:87-98 (NaN past the reference's list length), n = A (T-1) M
 *                                       entries each, in three blocks: collision probability [A][T-1][M] at offset 0;
 *                                       (ego harm, obstacle harm) pairs [A][T-1][M][2] at offset n; (ego risk, obstacle risk)
 *                                       pairs [A][T-1][M][2] at offset 3 n -- a GPU lane writes one sample with three
 *                                       stores of 8 + 16 + 16 bytes, each contiguous across the trajectories of a wave
 */
#ifndef FO_HIP_H
#define FO_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FO_ABI_VERSION 12

enum { FO_OK = 0, FO_E_ARG = -1, FO_E_UNSUPPORTED_COV = -2, FO_E_HIP = -3, FO_E_NOMEM = -4, FO_E_STATE = -5 };

/* commonroad ObstacleType strings -> codes (ref: metrics/utils/harm_model.py:15-32) */
enum {
  FO_TYPE_CAR = 0, FO_TYPE_TRUCK = 1, FO_TYPE_BUS = 2, FO_TYPE_BICYCLE = 3, FO_TYPE_PEDESTRIAN = 4,
  FO_TYPE_PRIORITY_VEHICLE = 5, FO_TYPE_PARKED_VEHICLE = 6, FO_TYPE_TRAIN = 7, FO_TYPE_MOTORCYCLE = 8,
  FO_TYPE_TAXI = 9, FO_TYPE_UNKNOWN = 10, FO_TYPE_STRUCTURE = 11
};

/* activated_metrics bits (ref: metrics/metric.py:109-117; dependency closure :125-147 is applied inside) */
enum { FO_M_DCE = 1, FO_M_CP = 2, FO_M_TTC = 4, FO_M_TTCE = 8, FO_M_WTTC = 16, FO_M_BE = 32, FO_M_HR = 64 };

/* vehicle_params attributes the path reads (ref: convert_dynamic_obstacle.py:60,78; collision_probability.py:35;
 * harm_model.py:96-97; be.py:56) */
typedef struct { double length, width, wb_rear_axle, mass, a_max; } fo_vehicle_t;

/* coefficients read from config/harm_params.json (ref: harm_params.json:26-29,40-45,98-101) */
typedef struct {
  double lr4s_const, lr4s_speed, lr4s_side, lr4s_rear;
  double lr1s_const, lr1s_speed;
  double ped_const, ped_speed;
} fo_harm_coeff_t;

/* metrics.metric_thresholds of the YAML (ref: config/config.yaml:14-22); NaN = null = check disabled */
typedef struct { double harm, risk, be, cp, ttc, dce; } fo_thresholds_t;

enum { FO_PF_DCE = 0, FO_PF_TTC, FO_PF_TTCE, FO_PF_MAX_EGO_RISK, FO_PF_MAX_OBST_RISK, FO_PF_HARM_WITH_CP,
       FO_PF_MAX_EGO_HARM, FO_PF_MAX_OBST_HARM, FO_PF_MAX_CP, FO_PF_BE_DECEL, FO_PF_BE_BTN, FO_PF_SPARE, FO_NPF = 12 };
enum { FO_PI_TIME_DCE = 0, FO_PI_RISK_INDEX, FO_PI_CP_ARGMAX, FO_PI_HR_VALID, FO_NPI = 4 };
enum { FO_L_CP = 0, FO_L_EGO_HARM, FO_L_OBST_HARM, FO_L_EGO_RISK, FO_L_OBST_RISK, FO_NL = 5 };
enum { FO_C_WTTC = 0, FO_C_MIN_DCE, FO_C_MAX_EGO_RISK, FO_C_MAX_OBST_RISK, FO_C_MAX_EGO_HARM, FO_C_MAX_OBST_HARM,
       FO_C_MAX_CP, FO_C_HARM_WITH_CP, FO_C_MIN_TTCE, FO_C_ARGMIN_DCE, FO_C_ARGMIN_TTC, FO_C_ARGMAX_RISK,
       FO_C_SAFE, FO_C_MAX_BTN, FO_C_RES0, FO_C_RES1, FO_NC = 16 };

typedef struct fo_ctx fo_ctx;

/* ---- context -------------------------------------------------------------------------------------------- */
int fo_create(fo_ctx **out, int device);            /* replaces: FOInterface.__init__ native state (interface.py:69-131) */
void fo_destroy(fo_ctx *ctx);
const char *fo_last_error(const fo_ctx *ctx);       /* replaces: Python exceptions (SURVEY 8b "error conventions") */
int fo_abi_version(void);
/* SHA-256 (hex) of the sources and flags this library was built from (__graft_entry__.source_id()): build(), smoke()
 * and bench.py compare it with the tree they run in, so a stale prebuilt library cannot pass for the current one */
const char *fo_build_id(void);

/* ---- metric sweep: replaces M calls of FOInterface.trajectory_safety_assessment (interface.py:216-219 ->
 *      metrics/metric.py:35-100 -> dce.py, ttc.py, ttce.py, wttc.py, cp.py, hr.py) ------------------------ */
int fo_sweep_configure(fo_ctx *ctx, const fo_vehicle_t *veh, const fo_harm_coeff_t *hc, const fo_thresholds_t *thr,
                       uint32_t metric_mask, double dt);        /* Metric.__init__ (metric.py:22-33), HR._load_param */

/* pre-size the context's HBM workspace so that fo_sweep_run never allocates (needed before hipGraph capture) -- for EVERY
 * batch of at most max_M trajectories, max_A agent slots and horizons up to max_T (a smaller batch may take the kernel's
 * horizon-split form, which keeps more partial rows per trajectory: the reservation covers that case as well) */
int fo_sweep_reserve(fo_ctx *ctx, int max_M, int max_T, int max_A, int max_Ta);

/* Element type of the per-timestep lists (d_lists of fo_sweep_run).  FO_LISTS_F64 (default): float64, the reference's
 * own numpy dtype (hr.py:87-98) -- what the parity tests hold to 1e-9.  FO_LISTS_F32: the same three blocks with float32
 * elements (cp float [n], harm float2 [n], risk float2 [n]; n = A (T-1) M), the storage SURVEY 8d prices at 648 B per
 * pair.  Collision probabilities and risks are the float64 results rounded at the store; harm entries away from the 5 m
 * gate are evaluated in float32 (hardware exp / rcp, |error| < 4e-7 -- north_star's tolerance is 1e-5).  cost, safe,
 * pair_f and pair_i are float64 / exact and bit-identical in both formats. */
/* FO_LISTS_F32_EXACT: float32 elements in the same layout as FO_LISTS_F32, every entry the float64 result rounded at the
 * store (the arithmetic of FO_LISTS_F64, the bytes of FO_LISTS_F32; bench.py times it beside FO_LISTS_F32 so that the
 * price of the float32 harm entries is a number). */
enum { FO_LISTS_F64 = 0, FO_LISTS_F32 = 1, FO_LISTS_F32_EXACT = 2 };
int fo_sweep_set_list_format(fo_ctx *ctx, int format);

/* replaces: agent_manager.predictions (agent.py:179-183) as read by every metric */
int fo_sweep_set_agents(fo_ctx *ctx, int A, int Ta, const double *d_pos, const double *d_yaw, const double *d_v,
                        const double *d_cov, const double *d_shape, const double *d_raw_dims, const int32_t *d_type,
                        const int32_t *d_len, void *stream);

/* d_pair_f / d_pair_i / d_lists may be NULL (reduced output mode); d_cost and d_safe are required.
 * d_a (acceleration profile) may be NULL unless FO_M_BE is active (ref: metrics/be.py:66-68 reads cartesian.a).
 * FO_M_BE implies FO_M_TTC and FO_M_DCE (be.py:39,49 read results['ttc']); its per-pair outputs are
 * pair_f[FO_PF_BE_DECEL], pair_f[FO_PF_BE_BTN] (0 for pairs that do not collide, be.py:45-46) and the per-trajectory
 * maximum goes to cost[FO_C_MAX_BTN]; thresholds.be is applied like metric.py:54-61. */
int fo_sweep_run(fo_ctx *ctx, int M, int T, const double *d_x, const double *d_y, const double *d_theta,
                 const double *d_v, const double *d_a, double *d_cost, uint8_t *d_safe, double *d_pair_f,
                 int32_t *d_pair_i, double *d_lists, void *stream);

/* The sweep kernel gives every wave `agents per wave` agents of a tile; the best value (1, 2, 4 or 8) depends on the batch
 * shape (how many workgroups the grid has against the chip's 768 slots).  fo_sweep_run picks one by a static rule;
 * fo_sweep_autotune MEASURES the four settings on the caller's own batch -- `reps` launches each, HIP events on `stream` --
 * and remembers the fastest for every later fo_sweep_run / fo_step_run on this context with the same shape (number of
 * 64-trajectory tiles, agent slots, horizon, output mode, list format).  Arguments as for fo_sweep_run (agents set
 * before); the outputs hold a complete result afterwards.  best_apw / ms4 [4] (ms per launch for 1, 2, 4, 8) may be
 * NULL.  Synchronises the stream; call it once when a planning loop starts.  (No counterpart in the reference.) */
int fo_sweep_autotune(fo_ctx *ctx, int M, int T, const double *d_x, const double *d_y, const double *d_theta,
                      const double *d_v, const double *d_a, double *d_cost, uint8_t *d_safe, double *d_pair_f,
                      int32_t *d_pair_i, double *d_lists, int reps, int *best_apw, double *ms4, void *stream);

/* Blocks until `stream` has drained, then returns FO_E_UNSUPPORTED_COV if the last fo_sweep_set_agents met a matrix
 * that is no usable covariance -- asymmetric, not positive, or |correlation| > 0.99 (those agents' collision
 * probabilities are NaN and no trajectory reads as safe) -- else FO_OK.  Diagonal covariances take the closed form,
 * symmetric ones with correlation are integrated numerically (the reference hands any matrix to scipy's mvnun,
 * collision_probability.py:117).  Not capturable in a hipGraph. */
int fo_sweep_check(fo_ctx *ctx, void *stream);

/* HIP-event timing of the sweep kernel alone (events recorded on the launch stream around that one kernel):
 * enable, run K times (at most 1024 timed launches), then read the summed duration and the number of timed launches.
 * enable = k > 1 puts the event pair around every k-th launch only (the two event records cost a few microseconds of
 * stream time each).  fo_sweep_timing_read synchronises. */
int fo_sweep_timing(fo_ctx *ctx, int enable);
int fo_sweep_timing_read(fo_ctx *ctx, double *total_ms, int *launches);
/* the same read-out launch by launch: each_ms[i] = duration of the i-th timed launch, i < min(*launches, cap) */
int fo_sweep_timing_read_each(fo_ctx *ctx, double *each_ms, int cap, int *launches);

/* launch geometry of the last sweep launch (for profiling scripts) */
int fo_sweep_last_launch(const fo_ctx *ctx, int *grid, int *block, int *agents_per_wave);

/* ---- scene: visibility / occlusion / phantom spawn ------------------------------------------------------------
 * The reference computes these with GEOS polygon algebra; this library uses a polar ray fan + a cell grid (the
 * discretisation is specified in DESIGN.md).  Occluder ids: 0..E-1 = map boundary edge, E + o = obstacle o, -1 = none
 * within range.  Cell class bits: 1 road, 2 visible, 4 occluded.  Cells are indexed iy * win_nx + ix inside the
 * window whose lower-left raster cell is (win_ix0, win_iy0). */

/* One-off (replaces SensorModel.__init__ / _convert_lanelet_network, sensor_model.py:17-39,195-199): HOST arrays.
 * polygons: P lanelet polygons, vertices poly_xy[poly_off[p] .. poly_off[p+1]); edges [E][4] = occluding boundary
 * segments (ax, ay, bx, by); cs = cell size; margin = raster margin around the polygons' bounding box.
 * lane_yaw (optional, [rny][rnx], NaN off-lane) must match the raster geometry, which the caller may fix with
 * raster_origin[2] / raster_dims[2] (else it is derived from the bounding box and reported by fo_scene_map_info). */
int fo_scene_set_map(fo_ctx *ctx, int P, const int32_t *h_poly_off, const double *h_poly_xy, int E,
                     const double *h_edges, double cs, double margin, const double *h_lane_yaw_or_null,
                     const double *h_raster_origin_or_null, const int32_t *h_raster_dims_or_null);
/* Instead of fo_scene_set_map: let `ctx` read the static map (boundary pieces and their chunk boxes, road and
 * lane-heading rasters, chain labels, route table) that `owner` -- another context on the same device -- has uploaded,
 * by reference.  Several egos planning on one scenario on one GPU (BASELINE configs[4]: "shared occlusion map in HBM")
 * then hold the map once; every per-step buffer stays per context.  Reference counted: destroying the owner does not
 * pull the map away, and a later fo_scene_set_map gives the calling context a map of its own again.  (No counterpart in
 * the reference, which builds one shapely road polygon per SensorModel, sensor_model.py:195-199.)
 * The route table and the edge-line labels are part of the map: upload them through the owner BEFORE sharing.  While a
 * map has more than one reader, fo_scene_set_routes / fo_scene_set_edge_lines on any of them return FO_E_STATE (they
 * would free tables other contexts' kernels are reading); fo_scene_set_map detaches the caller first and is always
 * allowed.  The reference count is atomic: sharing and destroying from different host threads is safe. */
int fo_scene_share_map(fo_ctx *ctx, fo_ctx *owner);
/* One-off, after fo_scene_set_map (replaces FORoutePlanner, route_planner.py:15-90, evaluated for every lanelet up
 * front): HOST arrays.  Lanelet index p (order of the polygons given to fo_scene_set_map) has up to R candidate routes;
 * route r occupies vertices [first[p*R+r], first[p*R+r] + count[p*R+r]) of xy [NV][2] / s [NV] (arc length from the
 * route's first vertex); count 0 = no such route.  lanelet_raster int32 [rny][rnx]: lanelet index per raster cell, -1
 * off-lane. */
int fo_scene_set_routes(fo_ctx *ctx, int P, int R, const int32_t *h_first, const int32_t *h_count, int NV,
                        const double *h_xy, const double *h_s, const int32_t *h_lanelet_raster);
/* One-off, optional, after fo_scene_set_map: HOST int32 [E], label in [0, E) of the straight-line chain each boundary
 * piece belongs to (pieces that continue one another in a straight line; lanelet bounds are sampled polylines).  Two
 * rays that stop on the same chain see one occluder, which keeps the exact settlement of fo_scene_visibility to the
 * cells at real corners and depth discontinuities.  Without it every piece is its own chain. */
int fo_scene_set_edge_lines(fo_ctx *ctx, int E, const int32_t *h_line);
int fo_scene_map_info(fo_ctx *ctx, double *x0, double *y0, double *cs, int *nx, int *ny, int *n_edges);
/* Where an obstacle's shadow ends.  The reference builds an obstacle's occlusion polygon from the two silhouette corners
 * and the points `length` = 100 m beyond them along the sight lines (helper_functions.py:139-176: get_polygon_from_obstacle_
 * occlusion, _identify_projection_points): behind the chord between those two end points the obstacle hides nothing.  Seen
 * from close by (a 9 m truck within ~1.5 m) that chord falls inside the sensor range.  Default 100.0 = the reference's;
 * <= 0 or infinity = shadows without end.  Applied where cells are settled exactly (exact_cells of fo_scene_visibility:
 * every cell behind an obstacle is); the first-hit ranges / hit ids / the visible polygon's ring are unaffected.  The
 * reference's boundary shadows end at 101 x the vertex distance (:91-92), which the ray fan does not represent: inside
 * 1.5 r = 75 m that takes a boundary vertex within 0.74 m of the ego. */
int fo_scene_set_shadow_length(fo_ctx *ctx, double length);
int fo_scene_copy_raster(fo_ctx *ctx, uint8_t *h_out);

/* Ray fan about the ego heading, written on the device (replaces the angle bookkeeping of
 * _calc_visible_area_from_lanelet_geometry / _calc_relevant_sector, sensor_model.py:115-124,201-209).
 * d_dirs [n_rays][2]: full circle (fov_deg >= 359.9) angle_i = yaw + 2 pi i / n, else n rays from yaw - fov/2 to
 * yaw + fov/2 inclusive.  d_rmax [n_rays] (may be NULL): range of the sensor footprint along each ray; with
 * polygon_footprint != 0 the reference's inscribed polygon (Point.buffer(r) = regular 64-gon with a vertex at world
 * angle 0 [ext shapely default], or ego + 100 arc points), else r.  d_half [100][2] (may be NULL): unit directions of
 * the 100-point half fan about the heading whose radius-1.5 r polygon bounds the occluded area (sensor_model.py:85-93);
 * fo_scene_visibility takes it as d_half (NULL there = the true half disc). */
int fo_scene_fan(fo_ctx *ctx, int n_rays, double ego_yaw, double fov_deg, double r, int polygon_footprint,
                 double *d_dirs, double *d_rmax, double *d_half, void *stream);

/* Per step (replaces SensorModel.calc_visible_and_occluded_area, sensor_model.py:41-101).  d_dirs [n_rays][2] unit
 * ray directions in counter-clockwise order (full_circle: ray n_rays == ray 0); d_rmax [n_rays] (or NULL = r) range of
 * the sensor footprint along each ray -- the reference's footprint is a polygon (Point.buffer(r): regular 64-gon,
 * or the 100-point fan of _calc_relevant_sector, sensor_model.py:118-124,201-209); d_edge_skip [E] (or NULL = none)
 * marks boundary pieces that cast no shadow this step: rings of the road union enclosed by the footprint are interior
 * rings of road ∩ footprint and the reference walks exterior rings only (sensor_model.py:126-131).  Obstacles at this step:
 * d_ocorn [O][4][2] corner points (fo_obstacle.py:79-93), d_ocen [O][2], d_oflags [O] (bit0 present, bit1 occludes --
 * clear for bicycles, sensor_model.py:177).  Outputs: d_range/d_hit_id [n_rays], d_ring [n_rays][2] (vertices of the
 * visible polygon = what evaluate_scenario returns), d_obst_vis [O] (visible_objects_timestep), d_cls [ny][nx],
 * d_occ_idx (ascending, capacity nx*ny) + d_n_occ [1].  d_cls: 4-byte aligned, capacity rounded up to whole 32-bit words
 * (class bits are cleared with word atomics).
 * Cell classes: a road cell within r is visible iff its centre lies on the ego side of the chord between the hit points
 * of the two rays enclosing it.  exact_cells != 0: where those two rays stop at different occluders (or at an
 * obstacle) and the centre is not nearer than the shorter of the two by more than a cell, the fan cannot decide and
 * the cell is settled by the reference's own set algebra at the centre -- visible iff inside the footprint and no
 * occluding piece crosses the open segment ego -> centre (= the centre is in none of the shadow quads of
 * helper_functions.py:79-96 / occlusion polygons of :133-141); and no visible cell's centre lies within 5 mm of a
 * present non-bicycle obstacle (the buffered obstacle the reference subtracts, sensor_model.py:183). */
int fo_scene_visibility(fo_ctx *ctx, double ego_x, double ego_y, double head_x, double head_y, double r, int full_circle,
                        int exact_cells, int n_rays, const double *d_dirs, const double *d_rmax, const double *d_half,
                        const uint8_t *d_edge_skip, int O, const double *d_ocorn, const double *d_ocen,
                        const uint8_t *d_oflags, int win_ix0, int win_iy0, int win_nx, int win_ny, double *d_range,
                        int32_t *d_hit_id, double *d_ring, uint8_t *d_obst_vis, uint8_t *d_cls, int32_t *d_occ_idx,
                        int32_t *d_n_occ, void *stream);

/* EXTENSION, not part of the reference: how much of the currently occluded area each candidate trajectory
 * will come to see.  d_x / d_y [M][T] as for fo_sweep_run; pose (m, k) = sample k * t_stride, K = ceil(T / t_stride).
 * From every pose a full fan of n_rays (<= 768: the 720-ray fan of the visibility stage fits) rays of length r along d_dirs [n_rays][2] (world-aligned, counter-
 * clockwise, e.g. fo_scene_fan with yaw 0) is cast against the static map and the obstacles where they stand now
 * (d_ocorn / d_oflags as for fo_scene_visibility).  d_revealed [M][K] = number of cells of the current occluded set
 * (d_occ_idx / d_n_occ and the window of the fo_scene_visibility call that produced them) whose centre lies within r
 * of the pose and on its side of the chord between the hit points of the two rays enclosing it; d_area [M][K] = area
 * of the polygon of hit points. */
int fo_scene_future_visibility(fo_ctx *ctx, int M, int T, const double *d_x, const double *d_y, int t_stride, int n_rays,
                               const double *d_dirs, double r, int O, const double *d_ocorn, const uint8_t *d_oflags,
                               const int32_t *d_occ_idx, const int32_t *d_n_occ, int win_ix0, int win_iy0, int win_nx,
                               int32_t *d_revealed, double *d_area, void *stream);

/* Phantom sampling in the occluded cells + constant-velocity predictions (replaces the cell-based core of
 * SpawnLocator.find_spawn_points, spawn_locator.py:80-139, and agent.py:451-536).  Candidates: occluded cells at least
 * min_ahead ahead of the ego and within max_dist, on the visible/occluded frontier (all_occluded = 0) or anywhere in
 * the occluded set (all_occluded = 1); evenly spaced ranks are kept.  Prediction slots: R = max(routes, 1) per
 * agent, slot j * R + r; with routes > 0 (needs fo_scene_set_routes) a vehicle on a lanelet gets one prediction per
 * candidate route r (constant speed along the route, lateral offset kept; replaces route_planner.py:31-90 +
 * utils/frenetix_handler.py + agent.py:283-426), everything else one straight constant-velocity prediction in r = 0.
 * The per-prediction outputs (d_pos ... d_len) therefore hold max_agents * R slots.  Agent j takes pattern slot j % 4
 * (type4/speed4/raw/inflated dims: HOST arrays of 4).  d_path [n_path][2] = ego reference path.  Outputs for
 * max_agents slots, directly in the layout fo_sweep_set_agents consumes (slots >= *d_n have len 0 = inactive). */
int fo_scene_spawn(fo_ctx *ctx, const uint8_t *d_cls, int win_ix0, int win_iy0, int win_nx, int win_ny, double ego_x,
                   double ego_y, double head_x, double head_y, double min_ahead, double max_dist, int all_occluded,
                   int max_agents, int routes, const int32_t *type4, const double *speed4, const double *raw_l4, const double *raw_w4,
                   const double *infl_l4, const double *infl_w4, int n_path, const double *d_path, int T, double dt,
                   double var0, double var_factor, int32_t *d_cell, double *d_pos0, double *d_yaw0, int32_t *d_n,
                   double *d_pos, double *d_yaw, double *d_v, double *d_cov, double *d_shape, double *d_raw_dims,
                   int32_t *d_type, int32_t *d_len, void *stream);
int fo_scene_candidate_count(fo_ctx *ctx, int32_t *h_n, void *stream);

/* ---- the reference's three spawn rule families on the cell classes (replaces SpawnLocator.find_spawn_points' rule
 *      functions: pedestrian behind a visible static obstacle, spawn_locator.py:323-476; pedestrian behind a turn,
 *      :481-578; Car / Bicycle behind a visible dynamic obstacle, :145-317 with the rectangle fit of :695-726).  The
 *      reference asks shapely for intersections with the visible / occluded polygons; here the same predicates are asked
 *      of the cell classes.  The curvilinear frame is the polyline frame of the ego's reference path. */

/* One-off, after fo_scene_set_map (HOST arrays, lanelet index = order of the polygons): first vertex of every lanelet's
 * left bound [P][2] (spawn_locator.py:424), index of predecessors[0] and of adj_left per lanelet (-1 = none; :249-252,
 * :196-202), and the intersections: entries [off[i], off[i+1]) of (lanelet index, kind: 0 incoming, 1 inner = the left /
 * right / straight successors of an incoming element) for intersection i (:171-195). */
int fo_scene_set_topology(fo_ctx *ctx, int P, const double *h_left0, const int32_t *h_pred0, const int32_t *h_adj_left,
                          int n_inter, const int32_t *h_inter_off, const int32_t *h_inter_lanelet,
                          const uint8_t *h_inter_kind);

/* per-step scalars of the rules: ego pose, its curvilinear position, s_threshold = s_ego + max(4 v_ego, 25)
 * (spawn_locator.py:65-66,113), the ego's intention (0 straight ahead, 1 left turn, 2 right turn: curvature of the next
 * 40 m of the reference path, :678-693,729-741) and that window as vertex range [win_i0, win_i1) of the path table, the
 * switches and maxima of the YAML (spawn_locator section), the pedestrian's width / length (agent_manager section).
 * n_dynamic_plus1 (ABI 11): 1 + the number of this step's obstacles whose flags allow the dynamic-obstacle rule at all -- present
 * (bit0), dynamic role (bit2), neither bicycle nor pedestrian (bit3 clear) -- or 0 = not told (every obstacle is assumed to).  The
 * rule's helper workgroups (fifteen per obstacle, a CU each) are launched for that many obstacles only -- the first ones in list
 * order whose flags qualify; one beyond the count is treated as if the rule did not apply to it -- and for none at all when no
 * obstacle qualifies: the flags are the caller's own data, whether such an obstacle is visible stays a decision of the device. */
typedef struct {
  double ego_x, ego_y, ego_yaw, ego_s, ego_d, s_threshold;
  double ped_width, ped_length;
  int32_t intention, win_i0, win_i1;
  int32_t behind_static, behind_turn, behind_dynamic, max_static, max_dynamic;
  int32_t n_dynamic_plus1, reserved_;
} fo_spawn_rule_params_t;

/* Per step, after fo_scene_visibility on the same stream.  d_cls + window: that call's cell classes.  d_path6 [n_path][6]:
 * reference path table x, y, arc length, segment length, unit tangent (last row: tangent unused).  Obstacles at this
 * step: d_ocorn [O][4][2], d_ocen [O][2], d_oyaw [O], d_odims [O][2] (length, width), d_oflags [O] (bit0 present, bit1
 * occludes, bit2 dynamic role, bit3 type bicycle or pedestrian), d_obst_vis [O] = visible_objects_timestep of
 * fo_scene_visibility.  Output d_out [max_out][8]: type (FO_TYPE_*), x, y, orientation (NaN = to be derived,
 * agent.py:475-481), curvilinear s, d (NaN = none), source (1 behind dynamic obstacle, 2 behind static obstacle, 3 left
 * turn, 4 right turn), obstacle index (-1 = none), in the reference's order (dynamic, static, turn); d_n_out [1].
 * max_out must hold what the maxima allow -- (max_dynamic + 2) + (max_static + 1) + 1 points with all three families
 * switched on: the reference compares its maxima before it appends, and one dynamic obstacle can yield a Car and a Bicycle
 * (spawn_locator.py:212,304-309,365) -- else FO_E_ARG (a short buffer would drop the last points unnoticed).
 * Table space: the turn rule holds a reference window of <= 1 536 path vertices, the dynamic-obstacle rule flags for <= 9 409
 * lanelets and every fifth of <= 2 560 window vertices, max_static <= 15 -- beyond: FO_E_ARG.  What only the device can tell: an
 * obstacle's centre on more than sixteen lanelets or on more than seven relevant ones, a line a rule samples (cell size / 8
 * apart: the 40 m window, an obstacle's cross line) of more than 1 024 samples -- then *d_n_out = -1 and there is no list
 * (fo_scene_spawn_rule_agents and fo_step_run add no rule agents; whoever reads the count must refuse it). */
int fo_scene_spawn_rules(fo_ctx *ctx, const uint8_t *d_cls, int win_ix0, int win_iy0, int win_nx, int win_ny, int n_path,
                         const double *d_path6, int O, const double *d_ocorn, const double *d_ocen, const double *d_oyaw,
                         const double *d_odims, const uint8_t *d_oflags, const uint8_t *d_obst_vis,
                         const fo_spawn_rule_params_t *params, int max_out, double *d_out, int32_t *d_n_out, void *stream);

/* One-off, optional, after fo_scene_set_map (HOST arrays, lanelet index = order of the polygons): the centre line of every
 * lanelet, vertices xy[off[p] .. off[p+1]).  Read by fo_scene_spawn_rule_agents: a pedestrian spawned behind a turn heads
 * for the centre of the lanelet it stands on (mode 'lane_center', agent.py:459-467; interface.py:194).  Part of the
 * static map (FO_E_STATE while the map is shared, like fo_scene_set_routes). */
int fo_scene_set_centerlines(fo_ctx *ctx, int P, const int32_t *h_off, const double *h_xy);

/* per agent type the rule families spawn -- index 0 Car, 1 Bicycle, 2 Pedestrian: default speed, un-inflated and inflated
 * dimensions (YAML agent_manager section; agent.py:69-140,402-409,523-524) */
typedef struct { double speed[3], raw_l[3], raw_w[3], infl_l[3], infl_w[3]; } fo_rule_agent_types_t;

/* Per step, after fo_scene_spawn_rules on the same stream: the rule families' spawn points become phantom agents ON THE
 * DEVICE (replaces the loop over spawn points of FOInterface.evaluate_scenario, interface.py:186-198, with
 * FOAgentManager.add_agent, agent.py:46-141, OAPPedestrianAgent, :429-536, and OAPVehicleAgent, :283-426, behind it).
 * d_points [max_points][8] / d_n_points [1]: the records fo_scene_spawn_rules wrote (read in HBM, never on the host).
 * Point i owns R = max(routes, 1) prediction slots i * R + r of the outputs, which have the layout
 * fo_sweep_set_agents consumes (slots of points >= *d_n_points and unused route slots get len 0):
 *   Pedestrian -> heading = the record's (static-obstacle rule: lanelet heading + 90 deg, spawn_locator.py:459-460) or,
 *     when the record carries none, the unit normal towards a curve (agent.py:475-481): the centre line of the first
 *     lanelet that holds the point for the turn rules (mode 'lane_center'; the ego reference path when the point is on
 *     no lanelet -- the reference's latent None there is not reproduced, Q12), the ego reference path d_path
 *     [n_path][2] otherwise (mode 'ref_path'); straight constant-velocity prediction in slot r = 0 (agent.py:483-505);
 *   Car / Bicycle -> the first lanelet that holds the point; with routes > 0 (fo_scene_set_routes) one prediction per
 *     candidate route of that lanelet (the min-var(v) Frenet sample, see fo_scene_spawn), heading of record = first
 *     segment of route 0; off-lanelet or without a route table: heading towards d_path, one straight prediction.
 * d_pos0 [max_points][2] / d_yaw0 [max_points]: spawn position and initial heading per point (host views read them
 * lazily). */
int fo_scene_spawn_rule_agents(fo_ctx *ctx, int max_points, const double *d_points, const int32_t *d_n_points, int routes,
                               const fo_rule_agent_types_t *types, int n_path, const double *d_path, int T, double dt,
                               double var0, double var_factor, double *d_pos0, double *d_yaw0, double *d_pos, double *d_yaw,
                               double *d_v, double *d_cov, double *d_shape, double *d_raw_dims, int32_t *d_type,
                               int32_t *d_len, void *stream);

/* ---- one planning step in one call: fo_scene_fan -> fo_scene_visibility -> fo_scene_spawn -> fo_sweep_set_agents ->
 *      fo_sweep_run on one stream, with the arguments of those five entry points (same names, same meaning) in one
 *      structure.  What it replaces is the reference's FOInterface.evaluate_scenario + M x trajectory_safety_assessment
 *      (interface.py:148-219) for a host that calls through an FFI: a planning step of the reference's own size (2 000
 *      candidates x 32 phantoms) is twelve kernel launches of a few microseconds each, and five FFI crossings with
 *      ~90 arguments cost the host more than the GPU needs for the step.  The structure is filled once (every pointer
 *      and size of a planning loop is stable); per step the caller updates the ego pose, the window origin and the
 *      spawn range.  NULL-able members are the NULL-able arguments of the single calls.  Stops at the first failing
 *      stage and returns its code.  Every output buffer ends up with the bits the five calls would have written; the
 *      step itself runs in eight launches instead of twelve (the ray fan is worked out inside the ray kernel, whose spare
 *      workgroups also write the sweep's tile table of the candidates; the sampler's candidate cells are flagged during
 *      the compaction of the occluded cells; the prediction kernel writes its slots' rows of the sweep's agent table).
 *      FO_STEP_STAGES=1 in the environment: the plain stage calls.
 *      With spawn_mode FO_SPAWN_RULES / FO_SPAWN_BOTH the rule families run between the visibility and the sweep:
 *      fo_scene_spawn_rules (two launches) and fo_scene_spawn_rule_agents (one, which also writes its slots' rows of the
 *      agent table) -- the reference's find_spawn_points -> add_agent flow (interface.py:186-198) without leaving HBM. */
typedef struct {
  /* fo_scene_fan */
  int32_t n_rays, polygon_footprint;
  double ego_yaw, fov_deg, r;
  double *d_dirs, *d_rmax, *d_half;
  /* fo_scene_visibility (d_dirs / d_rmax / d_half as above) */
  double ego_x, ego_y, head_x, head_y;
  int32_t full_circle, exact_cells, O;
  const uint8_t *d_edge_skip;
  const double *d_ocorn, *d_ocen;
  const uint8_t *d_oflags;
  int32_t win_ix0, win_iy0, win_nx, win_ny;
  double *d_range;
  int32_t *d_hit_id;
  double *d_ring;
  uint8_t *d_obst_vis, *d_cls;
  int32_t *d_occ_idx, *d_n_occ;
  /* fo_scene_spawn (d_cls + window as above; its per-prediction outputs are the agent arrays of the sweep) */
  double min_ahead, max_dist;
  int32_t all_occluded, max_agents, routes, n_path, T_agents;
  int32_t type4[4];
  double speed4[4], raw_l4[4], raw_w4[4], infl_l4[4], infl_w4[4];
  const double *d_path;
  double dt, var0, var_factor;
  int32_t *d_cell;
  double *d_pos0, *d_yaw0;
  int32_t *d_n;
  double *d_pos, *d_yaw, *d_v, *d_cov, *d_shape, *d_raw_dims;
  int32_t *d_type, *d_len;
  /* fo_sweep_set_agents takes max_agents * max(routes, 1) prediction slots from the arrays above; fo_sweep_run: */
  int32_t M, T;
  const double *d_x, *d_y, *d_theta, *d_vel, *d_acc;
  double *d_cost;
  uint8_t *d_safe;
  double *d_pair_f;
  int32_t *d_pair_i;
  double *d_lists;
  /* element type of d_lists for THIS run (FO_LISTS_F64 / FO_LISTS_F32 / FO_LISTS_F32_EXACT); the step sets the context's format to it before
   * the sweep, so a stale fo_sweep_set_list_format of another caller cannot make the kernel write the other width */
  int32_t list_format;
  /* Which spawn stage feeds the sweep (interface.py:186-198).  FO_SPAWN_CELLS: fo_scene_spawn, the build's own sampling in
   * the occluded cells.  FO_SPAWN_RULES: the reference's three rule families, fo_scene_spawn_rules ->
   * fo_scene_spawn_rule_agents, device resident (no read-back, no host add_agent).  FO_SPAWN_BOTH: cells first, then
   * rules.  Slot layout of the prediction arrays d_pos ... d_len: [max_agents * R] cell slots (FO_SPAWN_RULES: none),
   * then [max_rule_points * R] rule slots, R = max(routes, 1); d_pos0 / d_yaw0 hold max_agents (+ max_rule_points)
   * entries the same way. */
  int32_t spawn_mode;
  /* fo_scene_spawn_rules (d_cls + window, obstacles as above; d_oflags then carries the rule bits 2 and 3 as well) and
   * fo_scene_spawn_rule_agents (d_path, T_agents, dt, var0, var_factor, routes as above) */
  int32_t n_path6, max_rule_points;
  const double *d_path6, *d_oyaw, *d_odims;
  fo_spawn_rule_params_t rule;
  fo_rule_agent_types_t rule_types;
  double *d_rule_points;
  int32_t *d_n_rule_points;
  /* ABI 12 -- the step's two host transfers, queued by the step itself (NULL / 0: none; what the reference does on the host
   * around sensor_model.py:41-101 without noticing: its obstacles and its visible-object bookkeeping live in host memory).
   * h_obstacles: the obstacle rows of this step in host memory, any kind (d_ocorn / d_ocen / d_oflags / d_oyaw / d_odims
   * point into d_obstacles); copied into a pinned staging slot of the context and from there to d_obstacles in front of the
   * first launch -- the caller's buffer is free again when fo_step_run returns; at most 64 KB.
   * h_mirror: PINNED, device-mapped host memory (hipHostMalloc / a pinned torch tensor) that receives mirror_bytes from
   * d_mirror by the end of the step (the interface mirrors d_hit_id and d_obst_vis, which it allocates back to back: that
   * pair is stored into the mirror by the kernels that produce it -- posted writes, no copy command; any other region is
   * copied behind the last launch); complete when fo_step_mirror_wait returns.  The next fo_step_run on the same stream
   * overwrites it.  COHERENCE: the direct stores need host-coherent pinned memory -- hipHostMalloc with the default flags
   * (or hipHostMallocCoherent / a pinned torch tensor); the event the wait synchronises on releases to system scope
   * (hipEventReleaseToSystem).  Memory from hipHostMallocNonCoherent or hipHostRegister carries no such guarantee: pass it
   * with FO_STEP_MIRROR_COPY=1 in the environment (the copy command instead of the direct stores) or not at all. */
  const void *h_obstacles;
  void *d_obstacles;
  int64_t obstacles_bytes;
  void *h_mirror;
  const void *d_mirror;
  int64_t mirror_bytes;
} fo_step_t;
enum { FO_SPAWN_CELLS = 0, FO_SPAWN_RULES = 1, FO_SPAWN_BOTH = 2 };
int fo_step_run(fo_ctx *ctx, const fo_step_t *step, void *stream);
/* blocks until the mirror copy of the latest fo_step_run with h_mirror has landed (returns at once when there was none) */
int fo_step_mirror_wait(fo_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
