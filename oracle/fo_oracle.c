/* fo_oracle.c -- CPU restatement (float64, scalar, reference loop order) of the metric sweep of
 * Frenetix-Occlusion.  TEST INFRASTRUCTURE ONLY (see fo_oracle.h).  Not used by the product path.
 *
 * Loop structure follows the reference: per trajectory -> per metric -> per agent prediction -> per timestep
 * (interface.py:216-219 -> metrics/metric.py:35-100).
 *
 * Pinned to outputs of the reference's own code (tests/golden/*.npz, written by tests/golden/gen_golden.py with the reference
 * imported from /root/reference): CP, harm, risk, HR, TTC, TTCE, WTTC, the safety decision -- and, since round 6, the DCE walk
 * over the time steps (dce_loop.npz) and BE's bisection (be_bisection.npz), whose reference classes run unmodified over this
 * file's rectangle primitives.  Restated from the published algorithm, not pinned: the polygon distance itself (GEOS, below).
 */
#define _GNU_SOURCE
#include "fo_oracle.h"
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ numpy helpers */
double fo_oracle_round3(double v) { return rint(v * 1000.0) / 1000.0; } /* np.round(v,3): rint(v*10^3)/10^3 */

/* ------------------------------------------------------------------ GEOS 3.11 distance (published algorithm)
 * algorithm/Distance.cpp pointToSegment / segmentToSegment; operation/distance/DistanceOp.cpp
 * computeContainmentDistance + computeFacetDistance.  shapely 2.0.2 (poetry.lock:1379) bundles GEOS 3.11.x. */
static double pt_dist(double px, double py, double qx, double qy) {
  double dx = px - qx, dy = py - qy;
  return sqrt(dx * dx + dy * dy);
}

static double point_to_segment(double px, double py, double ax, double ay, double bx, double by) {
  if (ax == bx && ay == by) return pt_dist(px, py, ax, ay);
  double len2 = (bx - ax) * (bx - ax) + (by - ay) * (by - ay);
  double r = ((px - ax) * (bx - ax) + (py - ay) * (by - ay)) / len2;
  if (r <= 0.0) return pt_dist(px, py, ax, ay);
  if (r >= 1.0) return pt_dist(px, py, bx, by);
  double s = ((ay - py) * (bx - ax) - (ax - px) * (by - ay)) / len2;
  return fabs(s) * sqrt(len2);
}

static int env_intersects(double ax, double ay, double bx, double by, double cx, double cy, double dx, double dy) {
  double minq = fmin(cx, dx), maxq = fmax(cx, dx), minp = fmin(ax, bx), maxp = fmax(ax, bx);
  if (minp > maxq) return 0;
  if (maxp < minq) return 0;
  minq = fmin(cy, dy); maxq = fmax(cy, dy); minp = fmin(ay, by); maxp = fmax(ay, by);
  if (minp > maxq) return 0;
  if (maxp < minq) return 0;
  return 1;
}

static double segment_to_segment(double ax, double ay, double bx, double by, double cx, double cy, double dx, double dy) {
  if (ax == bx && ay == by) return point_to_segment(ax, ay, cx, cy, dx, dy);
  if (cx == dx && cy == dy) return point_to_segment(dx, dy, ax, ay, bx, by);
  int no_intersection = 0;
  if (!env_intersects(ax, ay, bx, by, cx, cy, dx, dy)) {
    no_intersection = 1;
  } else {
    double denom = (bx - ax) * (dy - cy) - (by - ay) * (dx - cx);
    if (denom == 0.0) {
      no_intersection = 1;
    } else {
      double r_num = (ay - cy) * (dx - cx) - (ax - cx) * (dy - cy);
      double s_num = (ay - cy) * (bx - ax) - (ax - cx) * (by - ay);
      double s = s_num / denom, r = r_num / denom;
      if (r < 0.0 || r > 1.0 || s < 0.0 || s > 1.0) no_intersection = 1;
    }
  }
  if (no_intersection) {
    double d = point_to_segment(ax, ay, cx, cy, dx, dy);
    d = fmin(d, point_to_segment(bx, by, cx, cy, dx, dy));
    d = fmin(d, point_to_segment(cx, cy, ax, ay, bx, by));
    d = fmin(d, point_to_segment(dx, dy, ax, ay, bx, by));
    return d;
  }
  return 0.0;
}

/* point not EXTERIOR to a convex quad (boundary counts as inside, DistanceOp::computeInside) */
static int point_in_or_on_quad(double px, double py, const double *q) {
  int pos = 0, neg = 0;
  for (int i = 0; i < 4; ++i) {
    int j = (i + 1) & 3;
    double cr = (q[2 * j] - q[2 * i]) * (py - q[2 * i + 1]) - (q[2 * j + 1] - q[2 * i + 1]) * (px - q[2 * i]);
    if (cr > 0.0) pos = 1;
    if (cr < 0.0) neg = 1;
  }
  return !(pos && neg);
}

double fo_oracle_quad_distance(const double *qa, const double *qb) {
  /* containment: first vertex of one inside the other -> 0 */
  if (point_in_or_on_quad(qa[0], qa[1], qb)) return 0.0;
  if (point_in_or_on_quad(qb[0], qb[1], qa)) return 0.0;
  double best = INFINITY;
  for (int i = 0; i < 4; ++i) {
    int i2 = (i + 1) & 3;
    for (int j = 0; j < 4; ++j) {
      int j2 = (j + 1) & 3;
      double d = segment_to_segment(qa[2 * i], qa[2 * i + 1], qa[2 * i2], qa[2 * i2 + 1], qb[2 * j], qb[2 * j + 1],
                                    qb[2 * j2], qb[2 * j2 + 1]);
      if (d < best) best = d;
      if (best <= 0.0) return 0.0;
    }
  }
  return best;
}

void fo_oracle_rect_vertices(double cx, double cy, double yaw, double length, double width, double *q) {
  const double lx[4] = {-0.5 * length, -0.5 * length, 0.5 * length, 0.5 * length};
  const double ly[4] = {-0.5 * width, 0.5 * width, 0.5 * width, -0.5 * width};
  double c = cos(yaw), s = sin(yaw);
  for (int i = 0; i < 4; ++i) {
    q[2 * i] = c * lx[i] - s * ly[i] + cx;
    q[2 * i + 1] = s * lx[i] + c * ly[i] + cy;
  }
}

/* ------------------------------------------------------------------ DCE  (metrics/dce.py:52-99)
 * ego rectangle: rear-axle reference shifted by wb_rear_axle along heading (convert_dynamic_obstacle.py:69-78)
 * agent rectangle: un-inflated agent.shape at pos_list[t], orientation_list[t] (convert_dynamic_obstacle.py:27-47) */
static void dce_pair(int T, const double *x, const double *y, const double *th, const fo_vehicle_t *veh, int L,
                     const double *pos, const double *yaw, double raw_l, double raw_w, double *dce_out,
                     int *time_out) {
  double dce = INFINITY;
  int time_dce = 0;
  for (int i = 0; i < T; ++i) {
    if (i >= L) break; /* occupancy_at_time(i) is None  (dce.py:72,87-88) */
    double qa[8], qb[8];
    double cx = x[i] + veh->wb_rear_axle * cos(th[i]);
    double cy = y[i] + veh->wb_rear_axle * sin(th[i]);
    fo_oracle_rect_vertices(cx, cy, th[i], veh->length, veh->width, qa);
    fo_oracle_rect_vertices(pos[2 * i], pos[2 * i + 1], yaw[i], raw_l, raw_w, qb);
    double distance = fo_oracle_round3(fo_oracle_quad_distance(qa, qb)); /* dce.py:79 */
    if (distance < dce) { time_dce = i; dce = distance; }                /* dce.py:82-84 */
    if (dce == 0.0) break;                                               /* dce.py:85-86 */
  }
  *dce_out = dce;
  *time_out = time_dce;
}

/* ------------------------------------------------------------------ CP  (collision_probability.py:14-126) */
/* scipy.stats.mvn.mvnun == MVNDST (Genz).  For d = 2 MVNDNT reduces the box to BVNMVN(lower, upper, infin=(2,2), r):
 *   BVU(l0,l1) - BVU(u0,l1) - BVU(l0,u1) + BVU(u0,u1),   BVU(h,k) = P(X>h, Y>k)
 * and for r = 0 (diagonal covariance, the only kind the reference's phantom agents carry, agent.py:260-280)
 * BVU(h,k) = MVNPHI(-h) * MVNPHI(-k): products of upper-tail normal probabilities.  Restated literally so that
 * the far-tail behaviour (relative precision in the upper tails, absolute in the lower) follows the reference. */
static double upper_tail(double z) { return 0.5 * erfc(z / M_SQRT2); } /* MVNPHI(-z) */

double fo_oracle_box_prob(const double lo[2], const double hi[2], const double mu[2], double sxx, double syy) {
  double sx = sqrt(sxx), sy = sqrt(syy);
  double l0 = (lo[0] - mu[0]) / sx, u0 = (hi[0] - mu[0]) / sx;
  double l1 = (lo[1] - mu[1]) / sy, u1 = (hi[1] - mu[1]) / sy;
  double ql0 = upper_tail(l0), qu0 = upper_tail(u0), ql1 = upper_tail(l1), qu1 = upper_tail(u1);
  return ql0 * ql1 - qu0 * ql1 - ql0 * qu1 + qu0 * qu1;
}

/* Box probability under a covariance WITH correlation (collision_probability.py:117: mvnun -> MVNDST, which for two
 * dimensions evaluates Genz's BVU).  Restated here as a one-dimensional integral over x instead (not a line-by-line
 * restatement; pinned to the reference's own outputs by tests/golden/correlated_cov.npz):  with
 * X = (x - mu_x)/(s_x sqrt 2),
 *   P = 1/sqrt(pi) * Int_A^B exp(-X^2) * 1/2 [ erf((D - rho X) q) - erf((C - rho X) q) ] dX,   q = 1/sqrt(1 - rho^2),
 * A, B (C, D) the box edges in those units along x (y).  Composite Gauss-Legendre: the range is cut to where the
 * integrand is not below 1e-30 (|X| < 8.5 and the erf difference alive), then split into panels of half the width of
 * the integrand's narrowest feature, min(1, sqrt(1 - rho^2)/|rho|), 16 nodes each: 1e-15 against mvnun whatever the
 * variances are (tight covariances make boxes dozens of standard deviations wide).  |rho| > 0.99 counts as degenerate.
 * The HIP side integrates a different formula (over the correlation angle, fo_sweep.hip fo_corr_corners): the two
 * check each other. */
#define FO_GL_N 16
static double gl_x[FO_GL_N], gl_w[FO_GL_N];
static int gl_ready = 0;
static void gl_init(void) {
  const int n = FO_GL_N;
  for (int i = 0; i < (n + 1) / 2; ++i) {
    double z = cos(M_PI * (i + 0.75) / (n + 0.5)), pp = 1.0;
    for (int it = 0; it < 100; ++it) {
      double p1 = 1.0, p2 = 0.0;
      for (int j = 0; j < n; ++j) { double p3 = p2; p2 = p1; p1 = ((2.0 * j + 1.0) * z * p2 - j * p3) / (j + 1.0); }
      pp = n * (z * p1 - p2) / (z * z - 1.0);
      double dz = p1 / pp;
      z -= dz;
      if (fabs(dz) < 1e-16) break;
    }
    gl_x[i] = -z; gl_x[n - 1 - i] = z;
    gl_w[i] = gl_w[n - 1 - i] = 2.0 / ((1.0 - z * z) * pp * pp);
  }
  gl_ready = 1;
}

/* returns NaN for a matrix that is not a usable covariance (asymmetric, not positive, |rho| > 0.99) */
double fo_oracle_box_prob_corr(const double lo[2], const double hi[2], const double mu[2], double sxx, double sxy,
                               double syx, double syy) {
  if (!(sxx > 0.0) || !(syy > 0.0) || fabs(sxy - syx) > 1e-12 * sqrt(sxx * syy)) return NAN;
  const double rho = 0.5 * (sxy + syx) / sqrt(sxx * syy);
  if (!(fabs(rho) <= 0.99)) return NAN;
  if (!gl_ready) gl_init();
  const double ix = 1.0 / (sqrt(sxx) * M_SQRT2), iy = 1.0 / (sqrt(syy) * M_SQRT2), q = 1.0 / sqrt(1.0 - rho * rho);
  double A = (lo[0] - mu[0]) * ix, B = (hi[0] - mu[0]) * ix;
  const double Cc = (lo[1] - mu[1]) * iy, D = (hi[1] - mu[1]) * iy;
  if (A < -8.5) A = -8.5;
  if (B > 8.5) B = 8.5;
  double feat = 1.0;
  if (rho != 0.0) { /* the erf difference is below 1e-19 unless (C - rho X) q < 6.5 and (D - rho X) q > -6.5 */
    double u1 = (Cc - 6.5 / q) / rho, u2 = (D + 6.5 / q) / rho;
    if (u1 > u2) { const double t = u1; u1 = u2; u2 = t; }
    if (A < u1) A = u1;
    if (B > u2) B = u2;
    const double f = sqrt(1.0 - rho * rho) / fabs(rho);
    if (f < feat) feat = f;
  }
  if (!(B > A)) return 0.0;
  const int panels = (int)ceil((B - A) / (0.5 * feat));
  const double half = 0.5 * (B - A) / panels;
  double acc = 0.0;
  for (int p = 0; p < panels; ++p) {
    const double mid = A + (2 * p + 1) * half;
    for (int i = 0; i < FO_GL_N; ++i) {
      const double X = mid + half * gl_x[i];
      acc += gl_w[i] * exp(-X * X) * (erf((D - rho * X) * q) - erf((Cc - rho * X) * q));
    }
  }
  return acc * half * 0.5 / sqrt(M_PI);
}

/* probs[T-1]; returns -2 when a matrix is met that is not a usable covariance (see fo_oracle_box_prob_corr). */
static int cp_pair(int T, const double *x, const double *y, const double *th, const fo_vehicle_t *veh, int L,
                   const double *pos, const double *yaw, const double *cov, double len_infl, double *probs) {
  const double off0 = veh->length / 6.0, off1 = veh->width / 2.0; /* :35 */
  for (int i = 1; i < T; ++i) {
    double prob = 0.0;
    if (i < L) { /* :69 */
      /* means: agent position i-1, heading i  (Q1; :49-53) */
      double devx = cos(yaw[i]) * len_infl / 2.0, devy = sin(yaw[i]) * len_infl / 2.0;
      double mx[3] = {pos[2 * (i - 1)], pos[2 * (i - 1)] + devx, pos[2 * (i - 1)] - devx};
      double my[3] = {pos[2 * (i - 1) + 1], pos[2 * (i - 1) + 1] + devy, pos[2 * (i - 1) + 1] - devy};
      double mind = INFINITY;
      for (int j = 0; j < 3; ++j) {
        double ddx = mx[j] - x[i], ddy = my[j] - y[i];
        double d = sqrt(ddx * ddx + ddy * ddy);
        if (d < mind) mind = d;
      }
      if (!(mind > 5.0)) { /* :67,75 */
        const double *c4 = cov + 4 * (i - 1);
        double sxx = c4[0], sxy = c4[1], syx = c4[2], syy = c4[3];
        if (sxx == 0.0 && sxy == 0.0 && syx == 0.0 && syy == 0.0) { sxx = 0.1; syy = 0.1; } /* :84-86 */
        const int corr = sxy != 0.0 || syx != 0.0;
        /* three axis-aligned boxes around the REAR-AXLE point (Q2; :94-107,129-164) */
        double r_x = veh->length / 2.0;
        double ax = cos(th[i]), ay = sin(th[i]);
        double ccx[3] = {x[i], x[i] + r_x * (2.0 / 3.0) * ax, x[i] - r_x * (2.0 / 3.0) * ax};
        double ccy[3] = {y[i], y[i] + r_x * (2.0 / 3.0) * ay, y[i] - r_x * (2.0 / 3.0) * ay};
        for (int j = 0; j < 3; ++j) {
          for (int b = 0; b < 3; ++b) {
            double lo[2] = {ccx[b] - off0, ccy[b] - off1}, hi[2] = {ccx[b] + off0, ccy[b] + off1};
            double mu[2] = {mx[j], my[j]};
            if (corr) {
              const double pb = fo_oracle_box_prob_corr(lo, hi, mu, sxx, sxy, syx, syy);
              if (pb != pb) return -2;
              prob += pb;
            } else {
              prob += fo_oracle_box_prob(lo, hi, mu, sxx, syy);
            }
          }
        }
      }
    }
    probs[i - 1] = prob / 3.0; /* :122 */
  }
  return 0;
}

/* ------------------------------------------------------------------ harm (harm_model.py:35-190) */
static double obstacle_mass(int type, double size) { /* harm_model.py:158-190 */
  switch (type) {
    case FO_TYPE_CAR: case FO_TYPE_PRIORITY_VEHICLE: case FO_TYPE_PARKED_VEHICLE: case FO_TYPE_TAXI:
      return -1333.5 + 526.9 * pow(size, 0.8);
    case FO_TYPE_TRUCK: return 25000.0;
    case FO_TYPE_BUS: return 13000.0;
    case FO_TYPE_BICYCLE: return 90.0;
    case FO_TYPE_PEDESTRIAN: return 75.0;
    case FO_TYPE_TRAIN: return 118800.0;
    case FO_TYPE_MOTORCYCLE: return 250.0;
    default: return 0.0;
  }
}

/* 1 protected, 0 unprotected, 2 = None (harm_model.py:15-32) */
static int obstacle_protection(int type) {
  switch (type) {
    case FO_TYPE_CAR: case FO_TYPE_TRUCK: case FO_TYPE_BUS: case FO_TYPE_PRIORITY_VEHICLE:
    case FO_TYPE_PARKED_VEHICLE: case FO_TYPE_TRAIN: case FO_TYPE_TAXI: return 1;
    case FO_TYPE_BICYCLE: case FO_TYPE_PEDESTRIAN: case FO_TYPE_MOTORCYCLE: case FO_TYPE_UNKNOWN: return 0;
    default: return 2;
  }
}

static double lr4s(double velocity, double angle, const fo_harm_coeff_t *hc) { /* logistic_regression.py:11-52 */
  const double t_a = 45.0 / 180.0 * M_PI, t_b = 3.0 * t_a;
  double coef;
  if (-t_a < angle && angle < t_a) coef = 0.0;
  else if (t_a <= angle && angle < t_b) coef = hc->lr4s_side;
  else if (-t_a >= angle && angle > -t_b) coef = hc->lr4s_side;
  else coef = hc->lr4s_rear; /* incl. everything outside (-3pi/4, 3pi/4) on the UN-WRAPPED angle (Q5) */
  return 1.0 / (1.0 + exp(-hc->lr4s_const - hc->lr4s_speed * velocity - coef));
}

static double lr1s(double velocity, const fo_harm_coeff_t *hc) { /* logistic_regression.py:55-75 */
  return 1.0 / (1.0 + exp(-hc->lr1s_const - hc->lr1s_speed * velocity));
}

/* ego_harm[Lh], obst_harm[Lh], Lh = min(T-1, L) (harm_model.py:65-66) */
static void harm_pair(int Lh, const double *x, const double *y, const double *th, const double *v,
                      const fo_vehicle_t *veh, const double *pos, const double *yaw, const double *av, int type,
                      double size_infl, const fo_harm_coeff_t *hc, double *ego_harm, double *obst_harm) {
  double m_obs = obstacle_mass(type, size_infl); /* inflated footprint (Q8; :73,78) */
  int prot = obstacle_protection(type);
  for (int t = 0; t < Lh; ++t) {
    double pdof = yaw[t] - th[t] + M_PI;                               /* :81 */
    double rel = atan2(pos[2 * t + 1] - y[t], pos[2 * t] - x[t]);      /* :82-83 */
    double ego_angle = rel - th[t];                                    /* :86 */
    double obs_angle = M_PI + rel - yaw[t];                            /* :88 */
    double dv = sqrt(pow(v[t], 2) + pow(av[t], 2) + 2 * v[t] * av[t] * cos(pdof)); /* :91-95 */
    double ego_dv = m_obs / (veh->mass + m_obs) * dv;                  /* :96 */
    double obs_dv = veh->mass / (veh->mass + m_obs) * dv;              /* :97 */
    if (prot == 1) {
      ego_harm[t] = lr4s(ego_dv, ego_angle, hc);
      obst_harm[t] = lr4s(obs_dv, obs_angle, hc);
    } else if (prot == 0) {
      ego_harm[t] = lr1s(ego_dv, hc);
      obst_harm[t] = 1.0 / (1.0 + exp(hc->ped_const - hc->ped_speed * obs_dv)); /* :140-146 */
    } else {
      ego_harm[t] = 1.0; obst_harm[t] = 1.0;                            /* :148-149 */
    }
  }
}

/* ------------------------------------------------------------------ be (metrics/be.py:31-193)
 * Minimum constant deceleration (bisection, <= 10 iterations, be.py:66-82) that avoids the collision with one agent
 * prediction.  For a candidate deceleration the ego follows its own path with the speed profile
 * v_new = [v0, max(v1 - decel * j dt, 0) ...] (be.py:110-111), positions are re-sampled by linear interpolation over the
 * travelled chord length (scipy interp1d, be.py:114-126), and the rectangles are tested for intersection at every
 * step the agent exists (be.py:148-193).  Deviation: where the re-sampled arc length exceeds the original path
 * length scipy raises ValueError; here the arc length is clamped to the end of the path. */
static double interp_lin(int T, const double *xs, const double *ys, double s) {
  int idx = 0; /* np.searchsorted(xs, s) (left): first index with xs[idx] >= s */
  while (idx < T && xs[idx] < s) ++idx;
  if (idx < 1) idx = 1;
  if (idx > T - 1) idx = T - 1;
  const double xlo = xs[idx - 1], xhi = xs[idx], ylo = ys[idx - 1], yhi = ys[idx];
  if (xhi == xlo) return ylo; /* stationary stretch: scipy would divide by zero */
  const double slope = (yhi - ylo) / (xhi - xlo);
  return slope * (s - xlo) + ylo;
}

static int be_collides(int T, const double *x, const double *y, const double *th, const double *v, const double *dist,
                       double dt, double decel, const fo_vehicle_t *veh, int L, const double *pos, const double *yaw,
                       double raw_l, double raw_w) {
  double s = 0.0; /* dist_new[i] = sum_{j<i} v_new[j] dt */
  for (int i = 0; i < T; ++i) {
    const double sc = s > dist[T - 1] ? dist[T - 1] : s;
    if (i < L) {
      const double xn = interp_lin(T, dist, x, sc), yn = interp_lin(T, dist, y, sc), tn = interp_lin(T, dist, th, sc);
      double qa[8], qb[8];
      fo_oracle_rect_vertices(xn + veh->wb_rear_axle * cos(tn), yn + veh->wb_rear_axle * sin(tn), tn, veh->length,
                              veh->width, qa);
      fo_oracle_rect_vertices(pos[2 * i], pos[2 * i + 1], yaw[i], raw_l, raw_w, qb);
      if (fo_oracle_quad_distance(qa, qb) == 0.0) return 1; /* shapely intersects */
    }
    const double vn = (i == 0) ? v[0] : fmax(v[1] - decel * ((double)(i - 1) * dt), 0.0);
    s += vn * dt;
  }
  return 0;
}

static double be_pair(int T, const double *x, const double *y, const double *th, const double *v, const double *a,
                      double dt, const fo_vehicle_t *veh, int L, const double *pos, const double *yaw, double raw_l,
                      double raw_w, double *dist) {
  dist[0] = 0.0;
  for (int i = 1; i < T; ++i) {
    const double dx = x[i] - x[i - 1], dy = y[i] - y[i - 1];
    dist[i] = dist[i - 1] + sqrt(dx * dx + dy * dy);
  }
  double mina = 0.0;
  for (int i = 0; i < T; ++i) if (a[i] < mina) mina = a[i];
  double min_d = nearbyint(fabs(mina) * 100.0) / 100.0, max_d = 5.0, cur = 0.0; /* np.round(abs(min(min(a), 0)), 2) */
  for (int it = 0; it < 10; ++it) {
    cur = (min_d + max_d) / 2.0;
    if (!be_collides(T, x, y, th, v, dist, dt, cur, veh, L, pos, yaw, raw_l, raw_w)) max_d = cur; else min_d = cur;
    if (max_d - min_d < 0.1) break;
  }
  return cur;
}

/* ------------------------------------------------------------------ metric ordering (metric.py:125-147) */
uint32_t fo_oracle_required_metrics(uint32_t m) {
  if (m & FO_M_WTTC) m |= FO_M_TTC;
  if (m & FO_M_BE) m |= FO_M_TTC; /* be.py:39 reads results['ttc']: the reference raises KeyError without it; here it is implied */
  if (m & (FO_M_TTC | FO_M_TTCE | FO_M_BE)) m |= FO_M_DCE;
  if (m & FO_M_HR) m |= FO_M_CP;
  return m;
}

/* ------------------------------------------------------------------ one trajectory (metric.py:35-100) */
static int eval_trajectory(int T, const double *x, const double *y, const double *th, const double *v, const double *acc,
                           int A, int Ta,
                           const double *apos, const double *ayaw, const double *av, const double *acov,
                           const double *ashape, const double *araw, const int32_t *atype, const int32_t *alen,
                           const fo_vehicle_t *veh, const fo_harm_coeff_t *hc, double dt, const fo_thresholds_t *thr,
                           uint32_t mask, double *pair_f, int32_t *pair_i, double *lists, double *cost,
                           uint8_t *safe, double *scratch) {
  const int Tm1 = T - 1 > 0 ? T - 1 : 0;
  double *cp = scratch, *eh = scratch + Tm1, *oh = scratch + 2 * Tm1, *dist = scratch + 3 * Tm1;
  double c[FO_NC];
  for (int i = 0; i < FO_NC; ++i) c[i] = 0.0;
  c[FO_C_WTTC] = INFINITY; c[FO_C_MIN_DCE] = INFINITY; c[FO_C_MIN_TTCE] = INFINITY;
  c[FO_C_ARGMIN_DCE] = -1; c[FO_C_ARGMIN_TTC] = -1; c[FO_C_ARGMAX_RISK] = -1;
  int ok = 1, dce_flag = 0;
  double min_ttc = INFINITY;
  for (int k = 0; k < A; ++k) {
    int L = alen[k];
    const double *pos = apos + (size_t)k * Ta * 2, *yaw = ayaw + (size_t)k * Ta, *vv = av + (size_t)k * Ta;
    const double *cov = acov + (size_t)k * Ta * 4;
    double pf[FO_NPF];
    int32_t pi[FO_NPI] = {0, 0, 0, 0};
    for (int i = 0; i < FO_NPF; ++i) pf[i] = NAN;
    if (L <= 0) { /* inactive slot of a partly filled spawn buffer (extension of this build): contributes nothing */
      if (pair_f) memcpy(pair_f + (size_t)k * FO_NPF, pf, sizeof pf);
      if (pair_i) memcpy(pair_i + (size_t)k * FO_NPI, pi, sizeof pi);
      continue;
    }
    /* --- cp (cp.py:25-42) */
    if (mask & FO_M_CP) {
      int rc = cp_pair(T, x, y, th, veh, L, pos, yaw, cov, ashape[2 * k], cp);
      if (rc) return rc;
    }
    /* --- dce, ttc, ttce, wttc */
    if (mask & FO_M_DCE) {
      double dce; int tdce;
      dce_pair(T, x, y, th, veh, L, pos, yaw, araw[2 * k], araw[2 * k + 1], &dce, &tdce);
      pf[FO_PF_DCE] = dce; pi[FO_PI_TIME_DCE] = tdce;
      if (dce < c[FO_C_MIN_DCE]) { c[FO_C_MIN_DCE] = dce; c[FO_C_ARGMIN_DCE] = k; }
      if (!isnan(thr->dce) && dce < thr->dce) dce_flag = 1; /* metric.py:93-98 */
      if (mask & FO_M_TTC) {
        /* ttc.py:43: np.isclose(dce, 0.0) == |dce| <= 1e-8 */
        double ttc = (fabs(dce) <= 1e-8) ? fo_oracle_round3(tdce * dt) : INFINITY;
        pf[FO_PF_TTC] = ttc;
        if (ttc < min_ttc) { min_ttc = ttc; c[FO_C_ARGMIN_TTC] = k; }
      }
      if (mask & FO_M_TTCE) {
        double ttce = fo_oracle_round3(tdce * dt); /* ttce.py:39 */
        pf[FO_PF_TTCE] = ttce;
        if (ttce < c[FO_C_MIN_TTCE]) c[FO_C_MIN_TTCE] = ttce;
      }
      if (mask & FO_M_BE) { /* be.py:31-62 */
        double decel = 0.0, btn = 0.0;
        const double ttc = pf[FO_PF_TTC];
        if (isfinite(ttc) && ttc > 0.0 && T >= 2) {
          decel = be_pair(T, x, y, th, v, acc, dt, veh, L, pos, yaw, araw[2 * k], araw[2 * k + 1], dist);
          btn = decel / veh->a_max;
        }
        pf[FO_PF_BE_DECEL] = decel; pf[FO_PF_BE_BTN] = btn;
        if (btn > c[FO_C_MAX_BTN]) c[FO_C_MAX_BTN] = btn;
      }
    }
    /* --- hr (hr.py:43-116) */
    int Lh = Tm1 < L ? Tm1 : L; /* harm_model.py:66 */
    if ((mask & FO_M_HR) && Lh > 0) {
      harm_pair(Lh, x, y, th, v, veh, pos, yaw, vv, atype[k], ashape[2 * k] * ashape[2 * k + 1], hc, eh, oh);
      double max_er = -INFINITY, max_or = -INFINITY, max_eh = -INFINITY, max_oh = -INFINITY, max_cp = -INFINITY;
      int idx_or = 0, idx_cp = 0;
      for (int t = 0; t < Lh; ++t) {
        double er = eh[t] * cp[t], orr = oh[t] * cp[t]; /* hr.py:78-79 (Q6) */
        if (er > max_er) max_er = er;
        if (orr > max_or) { max_or = orr; idx_or = t; }
        if (eh[t] > max_eh) max_eh = eh[t];
        if (oh[t] > max_oh) max_oh = oh[t];
        if (lists) {
          double *l = lists + (size_t)k * FO_NL * Tm1;
          l[FO_L_EGO_HARM * Tm1 + t] = eh[t]; l[FO_L_OBST_HARM * Tm1 + t] = oh[t];
          l[FO_L_EGO_RISK * Tm1 + t] = er;    l[FO_L_OBST_RISK * Tm1 + t] = orr;
        }
      }
      for (int t = 0; t < Tm1; ++t) if (cp[t] > max_cp) { max_cp = cp[t]; idx_cp = t; } /* np.max / np.argmax */
      double hwc = (max_cp > 0.01) ? oh[idx_cp] : 0.0; /* hr.py:81-84 */
      pf[FO_PF_MAX_EGO_RISK] = max_er; pf[FO_PF_MAX_OBST_RISK] = max_or; pf[FO_PF_HARM_WITH_CP] = hwc;
      pf[FO_PF_MAX_EGO_HARM] = max_eh; pf[FO_PF_MAX_OBST_HARM] = max_oh; pf[FO_PF_MAX_CP] = max_cp;
      pi[FO_PI_RISK_INDEX] = idx_or; pi[FO_PI_CP_ARGMAX] = idx_cp; pi[FO_PI_HR_VALID] = 1;
      if (max_er > c[FO_C_MAX_EGO_RISK]) c[FO_C_MAX_EGO_RISK] = max_er; /* hr.py:101-106 */
      if (max_or > c[FO_C_MAX_OBST_RISK]) { c[FO_C_MAX_OBST_RISK] = max_or; c[FO_C_ARGMAX_RISK] = k; }
      if (max_eh > c[FO_C_MAX_EGO_HARM]) c[FO_C_MAX_EGO_HARM] = max_eh;
      if (max_oh > c[FO_C_MAX_OBST_HARM]) c[FO_C_MAX_OBST_HARM] = max_oh;
      if (hwc > c[FO_C_HARM_WITH_CP]) c[FO_C_HARM_WITH_CP] = hwc;
      if (max_cp > c[FO_C_MAX_CP]) c[FO_C_MAX_CP] = max_cp;
    }
    if (lists && (mask & FO_M_CP)) {
      double *l = lists + (size_t)k * FO_NL * Tm1;
      for (int t = 0; t < Tm1; ++t) l[FO_L_CP * Tm1 + t] = cp[t];
    }
    if (pair_f) memcpy(pair_f + (size_t)k * FO_NPF, pf, sizeof pf);
    if (pair_i) memcpy(pair_i + (size_t)k * FO_NPI, pi, sizeof pi);
  }
  c[FO_C_WTTC] = (mask & FO_M_WTTC) || (mask & FO_M_TTC) ? min_ttc : INFINITY;
  /* thresholds (metric.py:50-98); no agents -> ({}, True) (metric.py:44-45) */
  if (A > 0) {
    if ((mask & FO_M_HR) && !isnan(thr->harm) && c[FO_C_HARM_WITH_CP] > thr->harm) ok = 0;
    if ((mask & FO_M_HR) && !isnan(thr->risk) && c[FO_C_MAX_OBST_RISK] > thr->risk) ok = 0;
    if ((mask & FO_M_HR) && !isnan(thr->cp) && c[FO_C_MAX_CP] > thr->cp) ok = 0;
    if ((mask & FO_M_TTC) && !isnan(thr->ttc) && min_ttc < thr->ttc) ok = 0;
    if ((mask & FO_M_DCE) && dce_flag) ok = 0;
    if ((mask & FO_M_BE) && !isnan(thr->be) && c[FO_C_MAX_BTN] > thr->be) ok = 0; /* metric.py:54-61 */
  }
  c[FO_C_SAFE] = ok;
  if (cost) memcpy(cost, c, sizeof c);
  if (safe) *safe = (uint8_t)ok;
  return 0;
}

int fo_oracle_sweep(int M, int T, const double *x, const double *y, const double *theta, const double *v,
                    const double *a, int A, int Ta, const double *apos, const double *ayaw, const double *av,
                    const double *acov, const double *ashape, const double *araw, const int32_t *atype,
                    const int32_t *alen, const fo_vehicle_t *veh, const fo_harm_coeff_t *hc, double dt,
                    const fo_thresholds_t *thr, uint32_t metric_mask, double *pair_f, int32_t *pair_i,
                    double *lists, double *cost, uint8_t *safe, int nthreads) {
  if (M < 0 || T < 1 || A < 0) return -1;
  if ((metric_mask & FO_M_BE) && !a) return -1;
  for (int k = 0; k < A; ++k) if (alen[k] < 0 || alen[k] > Ta) return -1;
  uint32_t mask = fo_oracle_required_metrics(metric_mask);
  const int Tm1 = T - 1;
  int err = 0;
  if (!gl_ready) gl_init(); /* once, before the worker threads exist (they only read the table) */
#ifdef _OPENMP
  if (nthreads > 1) omp_set_num_threads(nthreads);
#pragma omp parallel if (nthreads > 1)
#endif
  {
    double *scratch = (double *)malloc(sizeof(double) * (3 * (size_t)(Tm1 > 0 ? Tm1 : 1) + (size_t)T));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 8)
#endif
    for (int m = 0; m < M; ++m) {
      if (lists) {
        double *lm = lists + (size_t)m * A * FO_NL * Tm1;
        for (size_t i = 0; i < (size_t)A * FO_NL * Tm1; ++i) lm[i] = NAN;
      }
      int rc = eval_trajectory(T, x + (size_t)m * T, y + (size_t)m * T, theta + (size_t)m * T, v + (size_t)m * T,
                               a ? a + (size_t)m * T : NULL, A, Ta, apos, ayaw, av, acov, ashape, araw, atype, alen, veh, hc, dt, thr, mask,
                               pair_f ? pair_f + (size_t)m * A * FO_NPF : NULL,
                               pair_i ? pair_i + (size_t)m * A * FO_NPI : NULL,
                               lists ? lists + (size_t)m * A * FO_NL * Tm1 : NULL,
                               cost ? cost + (size_t)m * FO_NC : NULL, safe ? safe + m : NULL, scratch);
      if (rc) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
        err = rc;
      }
    }
    free(scratch);
  }
  return err;
}
