/* fo_oracle.h -- CPU restatement of the Frenetix-Occlusion per-timestep hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product path (frenetix-occlusion_amd/) never links, imports or calls it.
 *
 * Everything is float64 like the reference (numpy).  Citations are relative to /root/reference/.
 *
 * Pinning status:
 *   CP (diagonal and full covariances), harm incl. the impact-angle bins on their boundaries, HR, TTC, TTCE, WTTC,
 *   the safety decision (Metric thresholds + dependency closure), pedestrian predictions
 *                                  -- pinned against golden vectors produced by the reference's own code
 *                                     (tests/golden/gen_golden.py).
 *   DCE, sensor model, spawn       -- PARITY UNPINNED: the reference delegates to shapely/GEOS + commonroad, absent
 *                                     here; pinned by analytic known-answer tests (tests/test_oracle_kat.py,
 *                                     test_scene_kat.py) and to sympy's exact rational geometry (test_dce_sympy.py,
 *                                     test_scene_sympy.py).
 */
#ifndef FO_ORACLE_H
#define FO_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* agent type codes (strings of commonroad ObstacleType, harm_model.py:15-32) */
enum {
  FO_TYPE_CAR = 0, FO_TYPE_TRUCK = 1, FO_TYPE_BUS = 2, FO_TYPE_BICYCLE = 3, FO_TYPE_PEDESTRIAN = 4,
  FO_TYPE_PRIORITY_VEHICLE = 5, FO_TYPE_PARKED_VEHICLE = 6, FO_TYPE_TRAIN = 7, FO_TYPE_MOTORCYCLE = 8,
  FO_TYPE_TAXI = 9, FO_TYPE_UNKNOWN = 10, FO_TYPE_STRUCTURE = 11 /* roadBoundary/pillar/... : protection None */
};

/* metric bits (metric.py:109-117) */
enum { FO_M_DCE = 1, FO_M_CP = 2, FO_M_TTC = 4, FO_M_TTCE = 8, FO_M_WTTC = 16, FO_M_BE = 32, FO_M_HR = 64 };

typedef struct { double length, width, wb_rear_axle, mass, a_max; } fo_vehicle_t;

/* harm_params.json: log_reg.reduced_sym_angle_areas{const,speed,side,rear}, log_reg.ignore_angle{const,speed},
 * pedestrian{const,speed} -- the only entries read (harm_model.py:109-155, logistic_regression.py) */
typedef struct {
  double lr4s_const, lr4s_speed, lr4s_side, lr4s_rear;
  double lr1s_const, lr1s_speed;
  double ped_const, ped_speed;
} fo_harm_coeff_t;

/* metric_thresholds (config.yaml:14-22); NaN = null = deactivated */
typedef struct { double harm, risk, be, cp, ttc, dce; } fo_thresholds_t;

/* pair scalar slots */
enum { FO_PF_DCE = 0, FO_PF_TTC, FO_PF_TTCE, FO_PF_MAX_EGO_RISK, FO_PF_MAX_OBST_RISK, FO_PF_HARM_WITH_CP,
       FO_PF_MAX_EGO_HARM, FO_PF_MAX_OBST_HARM, FO_PF_MAX_CP, FO_PF_BE_DECEL, FO_PF_BE_BTN, FO_PF_SPARE, FO_NPF = 12 };
enum { FO_PI_TIME_DCE = 0, FO_PI_RISK_INDEX, FO_PI_CP_ARGMAX, FO_PI_HR_VALID, FO_NPI = 4 };
/* per-timestep lists */
enum { FO_L_CP = 0, FO_L_EGO_HARM, FO_L_OBST_HARM, FO_L_EGO_RISK, FO_L_OBST_RISK, FO_NL = 5 };
/* per-trajectory cost vector */
enum { FO_C_WTTC = 0, FO_C_MIN_DCE, FO_C_MAX_EGO_RISK, FO_C_MAX_OBST_RISK, FO_C_MAX_EGO_HARM, FO_C_MAX_OBST_HARM,
       FO_C_MAX_CP, FO_C_HARM_WITH_CP, FO_C_MIN_TTCE, FO_C_ARGMIN_DCE, FO_C_ARGMIN_TTC, FO_C_ARGMAX_RISK,
       FO_C_SAFE, FO_C_MAX_BTN, FO_C_RES0, FO_C_RES1, FO_NC = 16 };

/* polygon(quad)-polygon(quad) distance the way GEOS 3.11 DistanceOp computes it (shapely 2.0.2
 * Polygon.distance, called at metrics/dce.py:79).  q = 4 vertices x,y interleaved. */
double fo_oracle_quad_distance(const double *qa, const double *qb);

/* vertices of a commonroad Rectangle(length,width) at centre (cx,cy), yaw: (-l/2,-w/2),(-l/2,w/2),(l/2,w/2),(l/2,-w/2)
 * rotated then translated (convert_dynamic_obstacle.py:41,78 + commonroad Rectangle semantics [ext]) */
void fo_oracle_rect_vertices(double cx, double cy, double yaw, double length, double width, double *q8);

double fo_oracle_round3(double v); /* np.round(v, 3) */

/* P(lower <= X <= upper), X ~ N(mu, diag(sxx, syy)) -- closed form equal to scipy mvnun for diagonal cov */
double fo_oracle_box_prob(const double lo[2], const double hi[2], const double mu[2], double sxx, double syy);
/* the same for a covariance with correlation (NaN when the matrix is not a usable covariance) */
double fo_oracle_box_prob_corr(const double lo[2], const double hi[2], const double mu[2], double sxx, double sxy,
                               double syx, double syy);

/* The sweep.  Inputs: trajectories [M][T]; agent predictions [A][Ta](...) with per-agent valid length alen[k]
 * (1..Ta).  acov = [A][Ta][4] (xx,xy,yx,yy); ashape = inflated (prediction dict 'shape'), araw = agent.shape.
 * Outputs may be NULL: pair_f [M][A][FO_NPF], pair_i [M][A][FO_NPI], lists [M][A][FO_NL][T-1] (NaN beyond the
 * reference's list length), cost [M][FO_NC], safe [M].  Returns 0, or a negative error code.
 * nthreads <= 1: scalar loop; > 1: OpenMP over trajectories (used only for the timed CPU baseline). */
int fo_oracle_sweep(int M, int T, const double *x, const double *y, const double *theta, const double *v,
                    const double *a, int A, int Ta, const double *apos, const double *ayaw, const double *av,
                    const double *acov, const double *ashape, const double *araw, const int32_t *atype,
                    const int32_t *alen, const fo_vehicle_t *veh, const fo_harm_coeff_t *hc, double dt,
                    const fo_thresholds_t *thr, uint32_t metric_mask, double *pair_f, int32_t *pair_i,
                    double *lists, double *cost, uint8_t *safe, int nthreads);

/* ---- scene half (fo_oracle_scene.c): discretisation defined in DESIGN.md, "parity unpinned" vs the reference ---- */
int fo_oracle_road_raster(int P, const int32_t *poly_off, const double *poly_xy, double x0, double y0, double cs,
                          int nx, int ny, uint8_t *mask);
int fo_oracle_raycast(int E, const double *edges, const uint8_t *edge_skip, int O, const double *ocorn,
                      const uint8_t *oflags, const double *ego, int n_rays, const double *dirs, double r,
                      const double *rmax, double *range, int32_t *hit_id, double *ring);
/* inputs of the exact settlement of the cells the fan cannot decide (NULL = fan rule only) */
typedef struct {
  const int32_t *hit_id; /* [n_rays] from fo_oracle_raycast */
  const double *rmax;    /* [n_rays] or NULL */
  int E;
  const double *edges;
  const uint8_t *edge_skip; /* or NULL */
  int O;
  const double *ocorn;
  const uint8_t *oflags;
  const double *half_dirs; /* [100][2] unit directions of the reference's 1.5 r half fan, or NULL = true half disc */
  const int32_t *edge_line; /* [E] straight-line chain of each boundary piece, or NULL = every piece on its own */
  double shadow_length;     /* where an obstacle's occlusion polygon ends (helper_functions.py:145-146: 100 m along the two
                             * silhouette sight lines); <= 0 or infinity: nowhere */
} fo_oracle_exact_t;
/* the silhouette corner pair of a rectangle seen from ego (helper_functions.py:150-176, _identify_projection_points) and the
 * half-plane beyond the chord between the two wedge end points: c12[4] = c1, c2; abc[3]: a x + b y + c > 0 <=> beyond.
 * Returns 1 when such a chord exists (finite positive length, non-degenerate view), else 0 (abc = 0, 0, -1). */
int fo_oracle_wedge_far(const double *ego, const double *corners, double length, double *c12, double *abc);
int fo_oracle_grid(const uint8_t *raster, int rnx, int rny, double rx0, double ry0, double cs, int ix0, int iy0,
                   int nx, int ny, const double *ego, const double *hdir, double r, int full, int n_rays,
                   const double *dirs, const double *range, uint8_t *cls, int32_t *occ_idx, int32_t *n_occ,
                   const fo_oracle_exact_t *exact, int32_t *n_exact);
int fo_oracle_obstacle_visibility(int E, const double *edges, const uint8_t *edge_skip, int O, const double *ocorn,
                                  const double *ocen, const uint8_t *oflags, const double *ego, double r, int full,
                                  int n_rays, const double *dirs, const int32_t *hit_id, uint8_t *vis);
/* extension (not in the reference): see fo_oracle_scene.c */
int fo_oracle_future_visibility(int M, int T, const double *x, const double *y, int t_stride, int n_rays,
                                const double *dirs, double r, int E, const double *edges, int O, const double *ocorn,
                                const uint8_t *oflags, int n_occ, const int32_t *occ_idx, double rx0, double ry0,
                                double cs, int ix0, int iy0, int nx, int32_t *revealed, double *area);
int fo_oracle_spawn_cells(const uint8_t *cls, int nx, int ny, double rx0, double ry0, double cs, int ix0, int iy0,
                          const double *ego, const double *hdir, double min_ahead, double max_dist, int max_agents,
                          int all_occluded, int32_t *cell, double *pos, int32_t *n_out, int32_t *n_cand_out);
void fo_oracle_normal_to_polyline(int N, const double *path, double px, double py, double *nx_, double *ny_);
int fo_oracle_spawn_headings(int n, const double *pos, const int32_t *type, int N, const double *path,
                             const double *lane_yaw_at, double *yaw);
int fo_oracle_cv_predictions(int n, const double *pos0, const double *yaw, const double *speed, int T, double dt,
                             double var0, double factor, double *pos, double *yaw_l, double *v_l, double *cov);

int fo_oracle_route_predictions(int n, const double *pos0, const int32_t *type, const double *speed,
                                const int32_t *lanelet, int R, const int32_t *first, const int32_t *count,
                                const double *xy, const double *sarr, const double *yaw_fallback, int T, double dt,
                                double var0, double factor, double *pos, double *yaw_l, double *v_l, double *cov,
                                int32_t *len);

/* metric.py:125-147 dependency closure on the activated-metric bit mask */
uint32_t fo_oracle_required_metrics(uint32_t mask);

#ifdef __cplusplus
}
#endif
#endif
