/* fo_oracle_scene.c -- CPU restatement of the scene half of the hot path: visibility, occluded cells, phantom
 * spawn sampling, pedestrian predictions.  TEST INFRASTRUCTURE ONLY (see fo_oracle.h).
 *
 * PARITY UNPINNED against the reference: the reference computes visibility by exact polygon algebra in
 * shapely/GEOS (sensor_model.py:103-193) -- there are no rays and no cells in it and GEOS is not
 * available here.  What this file restates is the discretisation DESIGN.md defines (polar ray fan + cell grid)
 * whose continuum limit is the reference's construction:
 *   visible  = road  ∩  sensor footprint  −  shadow of every road-boundary edge  −  obstacles and their shadow wedges
 *              (sensor_model.py:112-157,159-193; helper_functions.py:79-96,139-176); bicycles never occlude (Q10)
 *   occluded = road  ∩  half-disc(heading ± 90°, 1.5 r)  −  visible                    (sensor_model.py:85-93)
 * and is pinned by analytic known-answer tests (tests/test_scene_kat.py).  The HIP kernels must agree with this
 * file bit-for-bit on every integer output; both sides use only + - * / sqrt and comparisons on float64 and are
 * compiled without FMA contraction.
 */
#define _GNU_SOURCE
#include "fo_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- static map: road raster
 * mask[iy*nx+ix] = 1 iff the centre of world cell (ix,iy) lies in at least one lanelet polygon
 * (= inside road_polygon, the union of sensor_model.py:195-199).  Crossing-number rule, half-open in y. */
int fo_oracle_road_raster(int P, const int32_t *poly_off, const double *poly_xy, double x0, double y0, double cs,
                          int nx, int ny, uint8_t *mask) {
  /* per-polygon bounding boxes: a pure early-out (a point outside the box is outside the polygon) */
  double *bb = (double *)malloc(sizeof(double) * 4 * (size_t)(P > 0 ? P : 1));
  for (int p = 0; p < P; ++p) {
    double bx0 = INFINITY, by0 = INFINITY, bx1 = -INFINITY, by1 = -INFINITY;
    for (int i = poly_off[p]; i < poly_off[p + 1]; ++i) {
      bx0 = fmin(bx0, poly_xy[2 * i]); bx1 = fmax(bx1, poly_xy[2 * i]);
      by0 = fmin(by0, poly_xy[2 * i + 1]); by1 = fmax(by1, poly_xy[2 * i + 1]);
    }
    bb[4 * p] = bx0; bb[4 * p + 1] = by0; bb[4 * p + 2] = bx1; bb[4 * p + 3] = by1;
  }
  for (int iy = 0; iy < ny; ++iy) {
    for (int ix = 0; ix < nx; ++ix) {
      const double px = x0 + ((double)ix + 0.5) * cs, py = y0 + ((double)iy + 0.5) * cs;
      int inside_any = 0;
      for (int p = 0; p < P && !inside_any; ++p) {
        if (px < bb[4 * p] || px > bb[4 * p + 2] || py < bb[4 * p + 1] || py > bb[4 * p + 3]) continue;
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
      mask[(size_t)iy * nx + ix] = (uint8_t)inside_any;
    }
  }
  free(bb);
  return 0;
}

/* ---------------------------------------------------------------- first hit of one ray against the occluder soup
 * Ray o + t d (d unit), segment a + u (b - a):  denom = d x e,  tn = w x e,  un = w x d,  w = a - o.
 * Hit iff denom != 0, 0 <= tn/denom and 0 <= un/denom <= 1 -- decided on the signs of the numerators, so the
 * rejection path has no division; t = tn / denom for hits.  First hit = lexicographic minimum of (t, id).
 * Returns t (or +inf). */
static double ray_segment(double ox, double oy, double dx, double dy, double ax, double ay, double bx, double by) {
  const double ex = bx - ax, ey = by - ay;
  const double denom = dx * ey - dy * ex;
  if (denom == 0.0) return INFINITY;
  const double wx = ax - ox, wy = ay - oy;
  const double tn = wx * ey - wy * ex;
  const double un = wx * dy - wy * dx;
  const int hit = denom > 0.0 ? (tn >= 0.0 && un >= 0.0 && un <= denom) : (tn <= 0.0 && un <= 0.0 && un >= denom);
  return hit ? tn / denom : INFINITY;
}

/* occluder ids: 0..E-1 map boundary edges, E + o for obstacle o (any of its four sides); -1 = nothing within range.
 * skip_obst >= 0 leaves that obstacle out (used by the visibility probes of that obstacle).  edge_skip (may be NULL)
 * marks boundary pieces that cast no shadow this step: rings of the road union that lie entirely inside the sensor
 * footprint are interior rings of road ∩ footprint, and the reference walks exterior rings only
 * (sensor_model.py:126-131). */
static void first_hit(int E, const double *edges, const uint8_t *edge_skip, int O, const double *ocorn,
                      const uint8_t *oflags, double ox, double oy, double dx, double dy, double rmax, int skip_obst,
                      double *t_out, int *id_out) {
  double best = INFINITY;
  int id = -1;
  for (int e = 0; e < E; ++e) {
    if (edge_skip && edge_skip[e]) continue;
    const double *s = edges + 4 * (size_t)e;
    const double t = ray_segment(ox, oy, dx, dy, s[0], s[1], s[2], s[3]);
    if (t < best) { best = t; id = e; }
  }
  for (int o = 0; o < O; ++o) {
    if (!(oflags[o] & 1) || !(oflags[o] & 2) || o == skip_obst) continue; /* absent, or a bicycle (Q10) */
    const double *c = ocorn + 8 * (size_t)o;
    for (int s = 0; s < 4; ++s) {
      const int s2 = (s + 1) & 3;
      const double t = ray_segment(ox, oy, dx, dy, c[2 * s], c[2 * s + 1], c[2 * s2], c[2 * s2 + 1]);
      if (t < best) { best = t; id = E + o; }
    }
  }
  if (!(best <= rmax)) { best = rmax; id = -1; }
  *t_out = best;
  *id_out = id;
}

/* the polar fan: range[i] = distance to the first occluder along dirs[i] (clamped to the footprint range: rmax[i]
 * when given -- the sensor footprint is a polygon in the reference, sensor_model.py:118-121 -- else r), hit_id[i] as
 * above, ring[i] = ego + range[i] * dirs[i] (vertices of the visible-area polygon handed back by evaluate_scenario) */
int fo_oracle_raycast(int E, const double *edges, const uint8_t *edge_skip, int O, const double *ocorn,
                      const uint8_t *oflags, const double *ego, int n_rays, const double *dirs, double r,
                      const double *rmax, double *range, int32_t *hit_id, double *ring) {
  for (int i = 0; i < n_rays; ++i) {
    double t;
    int id;
    first_hit(E, edges, edge_skip, O, ocorn, oflags, ego[0], ego[1], dirs[2 * i], dirs[2 * i + 1], rmax ? rmax[i] : r,
              -1, &t, &id);
    range[i] = t;
    hit_id[i] = id;
    if (ring) {
      ring[2 * i] = ego[0] + t * dirs[2 * i];
      ring[2 * i + 1] = ego[1] + t * dirs[2 * i + 1];
    }
  }
  return 0;
}

/* sector of the fan that contains direction (rx, ry): the i with rel in [ray i, ray i+1) counter-clockwise.
 * ccw(i) = d_i x rel > 0, or = 0 with d_i . rel > 0.  The fan is searched in parts that span < pi (so that ccw()
 * is monotone inside a part): full fan -> thirds [0, n/3], [n/3, 2n/3], [2n/3, n] with ray n == ray 0 (halves would
 * exceed pi by a ray pitch when n is odd); open fan -> [0, m] and
 * [m, n-1], m = (n-1)/2.  Returns -1 when rel is outside an open fan (or is the zero vector). */
static int fan_ccw(int n_rays, const double *dirs, int i, double rx, double ry) {
  const double *d = dirs + 2 * (size_t)(i == n_rays ? 0 : i);
  const double c = d[0] * ry - d[1] * rx;
  if (c > 0.0) return 1;
  if (c < 0.0) return 0;
  return (d[0] * rx + d[1] * ry) > 0.0;
}

static int fan_search(int n_rays, const double *dirs, int a, int b, double rx, double ry) {
  if (!fan_ccw(n_rays, dirs, a, rx, ry) || fan_ccw(n_rays, dirs, b, rx, ry)) return -1;
  int lo = a, hi = b;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (fan_ccw(n_rays, dirs, mid, rx, ry)) lo = mid; else hi = mid;
  }
  return lo;
}

static int fan_sector(int n_rays, const double *dirs, int full, double rx, double ry) {
  if (full) { /* three parts of at most 2 pi / 3 + one ray pitch: ccw() is monotone inside each for every n >= 4 */
    const int a = n_rays / 3, b = (2 * n_rays) / 3;
    int s = fan_search(n_rays, dirs, 0, a, rx, ry);
    if (s >= 0) return s;
    s = fan_search(n_rays, dirs, a, b, rx, ry);
    if (s >= 0) return s;
    return fan_search(n_rays, dirs, b, n_rays, rx, ry);
  }
  const int m = (n_rays - 1) / 2;
  const int s = fan_search(n_rays, dirs, 0, m, rx, ry);
  if (s >= 0) return s;
  return fan_search(n_rays, dirs, m, n_rays - 1, rx, ry);
}

/* "is there an occluder strictly before the point ego + (rx, ry)": the reference's shadow quads
 * [v1, v2, v2 + 100 (v2 - ego), v1 + 100 (v1 - ego)] (helper_functions.py:79-96) and obstacle occlusion polygons
 * (:133-141) contain a point iff the open segment ego -> point crosses the occluding piece.  Same predicate as
 * ray_segment with the unnormalised direction (rx, ry); "t < 1" is decided on tn and denom without the division. */
/* Where an obstacle's shadow ends in the reference (helper_functions.py:139-176, restated: every ordered corner pair, the
 * arccos of the clipped dot product of the unit sight lines, the strictly largest angle first met wins; the occlusion polygon
 * runs from the two corners to the points `length` beyond them along their sight lines). */
int fo_oracle_wedge_far(const double *ego, const double *c, double length, double *c12, double *abc) {
  abc[0] = 0.0; abc[1] = 0.0; abc[2] = -1.0;
  double best = 0.0;
  int i1 = -1, i2 = -1;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      const double r1x = c[2 * i] - ego[0], r1y = c[2 * i + 1] - ego[1], r2x = c[2 * j] - ego[0], r2y = c[2 * j + 1] - ego[1];
      const double n1 = sqrt(r1x * r1x + r1y * r1y), n2 = sqrt(r2x * r2x + r2y * r2y);
      const double u1x = r1x / n1, u1y = r1y / n1, u2x = r2x / n2, u2y = r2y / n2;
      double d = u1x * u2x + u1y * u2y;
      d = d < -1.0 ? -1.0 : (d > 1.0 ? 1.0 : d);
      const double ang = acos(d);
      if (ang > best) { best = ang; i1 = i; i2 = j; }
    }
  if (i1 < 0) return 0;
  if (c12) { c12[0] = c[2 * i1]; c12[1] = c[2 * i1 + 1]; c12[2] = c[2 * i2]; c12[3] = c[2 * i2 + 1]; }
  if (!(length > 0.0) || !(length < INFINITY)) return 0;
  const double r1x = c[2 * i1] - ego[0], r1y = c[2 * i1 + 1] - ego[1], r2x = c[2 * i2] - ego[0], r2y = c[2 * i2 + 1] - ego[1];
  const double n1 = sqrt(r1x * r1x + r1y * r1y), n2 = sqrt(r2x * r2x + r2y * r2y);
  const double c4x = c[2 * i1] + r1x / n1 * length, c4y = c[2 * i1 + 1] + r1y / n1 * length;
  const double c3x = c[2 * i2] + r2x / n2 * length, c3y = c[2 * i2 + 1] + r2y / n2 * length;
  double a = -(c4y - c3y), b = c4x - c3x;
  double cc = -(a * c3x + b * c3y);
  const double ge = a * ego[0] + b * ego[1] + cc;
  if (ge == 0.0 || ge != ge) return 0;
  if (ge > 0.0) { a = -a; b = -b; cc = -cc; }
  abc[0] = a; abc[1] = b; abc[2] = cc;
  return 1;
}

static int blocked_before(int E, const double *edges, const uint8_t *edge_skip, int O, const double *ocorn,
                          const uint8_t *oflags, double ox, double oy, double rx, double ry, const double *ofar, double px,
                          double py) {
  for (int e = 0; e < E + 4 * O; ++e) {
    double ax, ay, bx, by;
    if (e < E) {
      if (edge_skip && edge_skip[e]) continue;
      const double *s = edges + 4 * (size_t)e;
      ax = s[0]; ay = s[1]; bx = s[2]; by = s[3];
    } else {
      const int o = (e - E) >> 2, sd = (e - E) & 3, s2 = (sd + 1) & 3;
      if (!(oflags[o] & 1) || !(oflags[o] & 2)) continue;
      /* beyond the end of the obstacle's occlusion polygon it hides nothing */
      if (ofar && ofar[3 * o] * px + ofar[3 * o + 1] * py + ofar[3 * o + 2] > 0.0) continue;
      const double *c = ocorn + 8 * (size_t)o;
      ax = c[2 * sd]; ay = c[2 * sd + 1]; bx = c[2 * s2]; by = c[2 * s2 + 1];
    }
    const double ex = bx - ax, ey = by - ay;
    const double denom = rx * ey - ry * ex;
    if (denom == 0.0) continue;
    const double wx = ax - ox, wy = ay - oy;
    const double tn = wx * ey - wy * ex;
    const double un = wx * ry - wy * rx;
    const int hit = denom > 0.0 ? (tn >= 0.0 && un >= 0.0 && un <= denom && tn < denom)
                                : (tn <= 0.0 && un <= 0.0 && un >= denom && tn > denom);
    if (hit) return 1;
  }
  return 0;
}

/* sensor_model.py:183: `visible_area.difference(obst.current_polygon.buffer(0.005, join_style=2))` -- the obstacle
 * rectangle grown by 5 mm with mitred (sharp) corners is taken out of the visible area: a point is inside iff its
 * signed distance to each of the four sides is <= 5 mm.  cross(e, q - a) <= 0.005 |e| per side (corner order of
 * fo_obstacle / helper_functions.py:99-112, either orientation); present, non-bicycle obstacles only (:177). */
static int in_obstacle_skin(int O, const double *ocorn, const uint8_t *oflags, double px, double py) {
  for (int o = 0; o < O; ++o) {
    if (!(oflags[o] & 1) || !(oflags[o] & 2)) continue;
    const double *c = ocorn + 8 * (size_t)o;
    /* orientation of the ring */
    const double area2 = (c[2] - c[0]) * (c[5] - c[1]) - (c[3] - c[1]) * (c[4] - c[0]);
    const double sg = area2 >= 0.0 ? 1.0 : -1.0;
    int inside = 1;
    for (int sd = 0; sd < 4 && inside; ++sd) {
      const int s2 = (sd + 1) & 3;
      const double ex = c[2 * s2] - c[2 * sd], ey = c[2 * s2 + 1] - c[2 * sd + 1];
      const double cr = ex * (py - c[2 * sd + 1]) - ey * (px - c[2 * sd]); /* > 0: left of the side */
      /* inside a counter-clockwise ring = left of every side; outside distance = -sg * cr / |e| */
      if (-(sg * cr) > 0.005 * sqrt(ex * ex + ey * ey)) inside = 0;
    }
    if (inside) return 1;
  }
  return 0;
}

/* cell classes: bit0 road, bit1 visible, bit2 occluded.  Window of nx x ny cells whose lower-left world cell is
 * (ix0, iy0) in the raster (rnx x rny, origin rx0, ry0, cell cs).  occ_idx = ascending window indices of the
 * occluded cells.
 *
 * visible = road, within r, and on the ego side of the chord between the hit points of the two rays enclosing the
 * cell centre.  With `exact` given, cells the fan cannot decide are settled by the reference's own rule at the cell
 * centre: where the two enclosing rays stop at different occluders (or at an obstacle) and the centre's distance lies
 * in [min range - cs, max range + cs], the cell is visible iff the centre lies inside the footprint chord of that
 * sector and no occluder crosses the open segment ego -> centre (blocked_before); and no visible cell's centre may
 * lie within 5 mm of an obstacle (in_obstacle_skin). */
int fo_oracle_grid(const uint8_t *raster, int rnx, int rny, double rx0, double ry0, double cs, int ix0, int iy0,
                   int nx, int ny, const double *ego, const double *hdir, double r, int full, int n_rays,
                   const double *dirs, const double *range, uint8_t *cls, int32_t *occ_idx, int32_t *n_occ,
                   const fo_oracle_exact_t *exact, int32_t *n_exact) {
  int cnt = 0, n_ex = 0;
  const double r2 = r * r, ro2 = (1.5 * r) * (1.5 * r);
  double *ofar = NULL;
  if (exact && exact->O > 0 && exact->shadow_length > 0.0 && exact->shadow_length < INFINITY) {
    ofar = (double *)malloc(sizeof(double) * 3 * (size_t)exact->O);
    for (int o = 0; o < exact->O; ++o) {
      ofar[3 * o] = 0.0; ofar[3 * o + 1] = 0.0; ofar[3 * o + 2] = -1.0;
      if ((exact->oflags[o] & 1) && (exact->oflags[o] & 2)) fo_oracle_wedge_far(ego, exact->ocorn + 8 * (size_t)o, exact->shadow_length, NULL, ofar + 3 * o);
    }
  }
  for (int iy = 0; iy < ny; ++iy) {
    for (int ix = 0; ix < nx; ++ix) {
      const int wx = ix0 + ix, wy = iy0 + iy;
      uint8_t c = 0;
      if (wx >= 0 && wx < rnx && wy >= 0 && wy < rny && raster[(size_t)wy * rnx + wx]) c |= 1;
      const double px = rx0 + ((double)wx + 0.5) * cs, py = ry0 + ((double)wy + 0.5) * cs;
      const double rx = px - ego[0], ry = py - ego[1];
      const double d2 = rx * rx + ry * ry;
      int vis = 0;
      if ((c & 1) && d2 <= r2) {
        const int i = fan_sector(n_rays, dirs, full, rx, ry);
        if (rx == 0.0 && ry == 0.0) {
          vis = 1; /* the ego's own cell centre */
        } else if (i >= 0) {
          const int j = (i + 1 == n_rays) ? 0 : i + 1;
          const double hix = range[i] * dirs[2 * i], hiy = range[i] * dirs[2 * i + 1];
          const double hjx = range[j] * dirs[2 * j], hjy = range[j] * dirs[2 * j + 1];
          /* inside the triangle (ego, h_i, h_j): on the ego side of the chord h_i -> h_j */
          const double cr = (hjx - hix) * (ry - hiy) - (hjy - hiy) * (rx - hix);
          vis = cr >= 0.0;
          if (exact) {
            int idi = exact->hit_id[i], idj = exact->hit_id[j];
            /* two rays stopping on the same straight chain of boundary pieces see one occluder */
            if (exact->edge_line && idi >= 0 && idi < exact->E && idj >= 0 && idj < exact->E) {
              idi = exact->edge_line[idi];
              idj = exact->edge_line[idj];
            }
            if (idi != idj || idi >= exact->E) {
              const double lo = range[i] < range[j] ? range[i] : range[j];
              double lom = lo - cs;
              if (lom < 0.0) lom = 0.0;
              if (d2 >= lom * lom) {
                const double fi = exact->rmax ? exact->rmax[i] : r, fj = exact->rmax ? exact->rmax[j] : r;
                const double fix = fi * dirs[2 * i], fiy = fi * dirs[2 * i + 1];
                const double fjx = fj * dirs[2 * j], fjy = fj * dirs[2 * j + 1];
                const double cf = (fjx - fix) * (ry - fiy) - (fjy - fiy) * (rx - fix);
                ++n_ex;
                vis = cf >= 0.0 && !blocked_before(exact->E, exact->edges, exact->edge_skip, exact->O, exact->ocorn,
                                                   exact->oflags, ego[0], ego[1], rx, ry, ofar, px, py);
              }
            }
          }
        }
      }
      if (vis && exact && in_obstacle_skin(exact->O, exact->ocorn, exact->oflags, px, py)) vis = 0;
      if (vis) c |= 2;
      /* sensor_model.py:85-93: half disc about the heading, radius 1.5 r, on the road, not visible.  The reference's
       * half disc is the 100-point fan of _calc_relevant_sector (:201-209): with its unit directions given
       * (exact->half_dirs), a centre in the thin rim between that polygon and the circle (d2 above 0.9994 ro2; the
       * polygon's inscribed radius squared is 0.99975 ro2) is tested against the chord of its sector. */
      if ((c & 1) && !vis && d2 <= ro2 && (rx * hdir[0] + ry * hdir[1]) >= 0.0) {
        int in_half = 1;
        if (exact && exact->half_dirs && d2 > 0.9994 * ro2) {
          const double *hd = exact->half_dirs;
          const int k = fan_search(100, hd, 0, 99, rx, ry);
          if (k >= 0) {
            const double R = 1.5 * r;
            const double ax = R * hd[2 * k], ay = R * hd[2 * k + 1];
            const double bx = R * hd[2 * k + 2], by = R * hd[2 * k + 3];
            in_half = ((bx - ax) * (ry - ay) - (by - ay) * (rx - ax)) >= 0.0;
          }
        }
        if (in_half) c |= 4;
      }
      cls[(size_t)iy * nx + ix] = c;
      if (c & 4) {
        if (occ_idx) occ_idx[cnt] = iy * nx + ix;
        ++cnt;
      }
    }
  }
  free(ofar);
  *n_occ = cnt;
  if (n_exact) *n_exact = n_ex;
  return 0;
}

/* obstacle visibility (sensor_model.py:59-76 restated: the obstacle polygon touches the visible area grown by 1 cm):
 * an obstacle that exists is visible iff a fan ray stops at it, or one of its five probe points (4 corners + centre)
 * lies within r + 1 cm of the ego, inside the fan, and the first occluder on the way (the obstacle itself excluded) is
 * not nearer than the probe by more than 1 cm. */
int fo_oracle_obstacle_visibility(int E, const double *edges, const uint8_t *edge_skip, int O, const double *ocorn,
                                  const double *ocen, const uint8_t *oflags, const double *ego, double r, int full,
                                  int n_rays, const double *dirs, const int32_t *hit_id, uint8_t *vis) {
  for (int o = 0; o < O; ++o) {
    vis[o] = 0;
    if (!(oflags[o] & 1)) continue;
    for (int p = 0; p < 5 && !vis[o]; ++p) {
      const double qx = p < 4 ? ocorn[8 * (size_t)o + 2 * p] : ocen[2 * o];
      const double qy = p < 4 ? ocorn[8 * (size_t)o + 2 * p + 1] : ocen[2 * o + 1];
      const double rx = qx - ego[0], ry = qy - ego[1];
      const double dist = sqrt(rx * rx + ry * ry);
      if (dist > r + 0.01) continue;
      if (dist == 0.0) { vis[o] = 1; break; }
      if (fan_sector(n_rays, dirs, full, rx, ry) < 0) continue;
      double t;
      int id;
      first_hit(E, edges, edge_skip, O, ocorn, oflags, ego[0], ego[1], rx / dist, ry / dist, dist, o, &t, &id);
      if (t >= dist - 0.01) vis[o] = 1;
    }
  }
  /* a fan ray that stops at an obstacle has reached a lit point of its boundary (hit_id from fo_oracle_raycast) */
  if (hit_id)
    for (int i = 0; i < n_rays; ++i)
      if (hit_id[i] >= E && hit_id[i] < E + O) vis[hit_id[i] - E] = 1;
  return 0;
}

/* ---------------------------------------------------------------- phantom spawn sampling in the occluded cells
 * Candidates = occluded cells with a visible 4-neighbour (the frontier a hidden road user would emerge from; with
 * all_occluded != 0: every occluded cell -- "sampling in the occluded cells" of the BASELINE configs),
 * at least `min_ahead` metres ahead of the ego along its heading and not farther than `max_dist`
 * (spawn_locator.py:113,234,381: s_threshold = max(4 v, 25), "ahead by >= 3 m").  Candidates are taken in
 * ascending cell order; if there are more than max_agents, the ones at ranks floor(j * n / max_agents) are kept.
 * Agent j gets type pattern[j % 4]; position = cell centre. */
int fo_oracle_spawn_cells(const uint8_t *cls, int nx, int ny, double rx0, double ry0, double cs, int ix0, int iy0,
                          const double *ego, const double *hdir, double min_ahead, double max_dist, int max_agents,
                          int all_occluded, int32_t *cell, double *pos, int32_t *n_out, int32_t *n_cand_out) {
  int n = 0;
  int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)nx * ny);
  for (int iy = 0; iy < ny; ++iy) {
    for (int ix = 0; ix < nx; ++ix) {
      const uint8_t c = cls[(size_t)iy * nx + ix];
      if (!(c & 4)) continue;
      int front = 0;
      if (ix > 0 && (cls[(size_t)iy * nx + ix - 1] & 2)) front = 1;
      if (ix + 1 < nx && (cls[(size_t)iy * nx + ix + 1] & 2)) front = 1;
      if (iy > 0 && (cls[(size_t)(iy - 1) * nx + ix] & 2)) front = 1;
      if (iy + 1 < ny && (cls[(size_t)(iy + 1) * nx + ix] & 2)) front = 1;
      if (!front && !all_occluded) continue; /* all_occluded: every occluded cell in range is a candidate */
      const double px = rx0 + ((double)(ix0 + ix) + 0.5) * cs, py = ry0 + ((double)(iy0 + iy) + 0.5) * cs;
      const double rx = px - ego[0], ry = py - ego[1];
      if (rx * hdir[0] + ry * hdir[1] < min_ahead) continue;
      if (rx * rx + ry * ry > max_dist * max_dist) continue;
      cand[n++] = iy * nx + ix;
    }
  }
  int m = n < max_agents ? n : max_agents;
  for (int j = 0; j < m; ++j) {
    const int pick = (n <= max_agents) ? j : (int)(((long long)j * n) / max_agents);
    const int ci = cand[pick];
    cell[j] = ci;
    pos[2 * j] = rx0 + ((double)(ix0 + ci % nx) + 0.5) * cs;
    pos[2 * j + 1] = ry0 + ((double)(iy0 + ci / nx) + 0.5) * cs;
  }
  *n_out = m;
  if (n_cand_out) *n_cand_out = n;
  free(cand);
  return 0;
}

/* unit vector from p to the closest point of the polyline (helper_functions.py:38-58; first closest segment) */
void fo_oracle_normal_to_polyline(int N, const double *path, double px, double py, double *nx_, double *ny_) {
  double best = INFINITY, qx = px, qy = py;
  for (int i = 0; i + 1 < N; ++i) {
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
    if (d2 < best) { best = d2; qx = cx; qy = cy; }
  }
  const double vx = qx - px, vy = qy - py;
  const double n = sqrt(vx * vx + vy * vy);
  if (n > 0.0) { *nx_ = vx / n; *ny_ = vy / n; } else { *nx_ = 1.0; *ny_ = 0.0; }
}

/* phantom headings: pedestrians walk towards the ego reference path (agent.py:475-481, mode 'ref_path');
 * vehicles / bicycles follow the lane heading raster at their cell (NaN -> fall back to the pedestrian rule).
 * yaw in [0, 2 pi) like helper_functions.py:61-70. */
int fo_oracle_spawn_headings(int n, const double *pos, const int32_t *type, int N, const double *path,
                             const double *lane_yaw_at, double *yaw) {
  for (int j = 0; j < n; ++j) {
    double a;
    if (type[j] != FO_TYPE_PEDESTRIAN && lane_yaw_at && !isnan(lane_yaw_at[j])) {
      a = lane_yaw_at[j];
    } else {
      double ux, uy;
      fo_oracle_normal_to_polyline(N, path, pos[2 * j], pos[2 * j + 1], &ux, &uy);
      a = atan2(uy, ux); /* det([1,0],[ux,uy]) = uy, dot = ux */
      if (a < 0.0) a += 2.0 * M_PI;
    }
    yaw[j] = a;
  }
  return 0;
}

/* constant-velocity straight-line predictions (agent.py:451-536): T = int(horizon/dt)+1 samples,
 * pos_k = p0 + (k dt) (round(v cos yaw, 3), round(v sin yaw, 3)), v_k = v, yaw_k = yaw,
 * cov_k = var0 * factor^k * I (agent.py:260-280) */
int fo_oracle_cv_predictions(int n, const double *pos0, const double *yaw, const double *speed, int T, double dt,
                             double var0, double factor, double *pos, double *yaw_l, double *v_l, double *cov) {
  for (int j = 0; j < n; ++j) {
    const double vx = fo_oracle_round3(speed[j] * cos(yaw[j])), vy = fo_oracle_round3(speed[j] * sin(yaw[j]));
    for (int k = 0; k < T; ++k) {
      const double t = (double)k * dt;
      pos[((size_t)j * T + k) * 2] = pos0[2 * j] + t * vx;
      pos[((size_t)j * T + k) * 2 + 1] = pos0[2 * j + 1] + t * vy;
      yaw_l[(size_t)j * T + k] = yaw[j];
      v_l[(size_t)j * T + k] = speed[j];
      const double var = var0 * pow(factor, (double)k);
      double *c = cov + ((size_t)j * T + k) * 4;
      c[0] = var; c[1] = 0.0; c[2] = 0.0; c[3] = var;
    }
  }
  return 0;
}

/* ---------------------------------------------------------------- phantom vehicle predictions along lanelet routes
 * Replaces route_planner.py:31-90 + utils/frenetix_handler.py + agent.py:283-426 (C++ frenetix sampler, un-vendored):
 * per candidate route of the start lanelet one prediction that keeps the initial speed and the initial lateral offset
 * to the route's centre line -- what the reference's min-var(v) Frenet sample (d1 = d0, ss1 = v0) amounts to.
 * Slot (j, r), r < R:  vehicle on a lanelet with routes -> route r of that lanelet (len 0 if it has fewer);
 * pedestrian, or vehicle off-lane / without routes -> r = 0 is the straight constant-velocity prediction with
 * yaw_fallback[j], r > 0 empty.  A prediction ends (len < T) where the route polyline ends.  PARITY UNPINNED for the
 * trajectory itself (frenetix is absent); the sample SET is pinned (tests/golden/sampling_matrix.npz). */
int fo_oracle_route_predictions(int n, const double *pos0, const int32_t *type, const double *speed,
                                const int32_t *lanelet, int R, const int32_t *first, const int32_t *count,
                                const double *xy, const double *sarr, const double *yaw_fallback, int T, double dt,
                                double var0, double factor, double *pos, double *yaw_l, double *v_l, double *cov,
                                int32_t *len) {
  for (int j = 0; j < n; ++j) {
    const int ll = lanelet[j];
    const int routed = type[j] != FO_TYPE_PEDESTRIAN && ll >= 0 && count[(size_t)ll * R] > 0;
    for (int r = 0; r < R; ++r) {
      const size_t slot = (size_t)j * R + r;
      double *P = pos + slot * T * 2, *Y = yaw_l + slot * T, *V = v_l + slot * T, *C = cov + slot * T * 4;
      for (int k = 0; k < T; ++k) {
        P[2 * k] = P[2 * k + 1] = Y[k] = V[k] = 0.0;
        const double var = var0 * pow(factor, (double)k);
        C[4 * k] = var; C[4 * k + 1] = 0.0; C[4 * k + 2] = 0.0; C[4 * k + 3] = var;
      }
      len[slot] = 0;
      if (!routed) {
        if (r > 0) continue;
        const double a = yaw_fallback[j];
        const double vx = fo_oracle_round3(speed[j] * cos(a)), vy = fo_oracle_round3(speed[j] * sin(a));
        for (int k = 0; k < T; ++k) {
          const double t = (double)k * dt;
          P[2 * k] = pos0[2 * j] + t * vx; P[2 * k + 1] = pos0[2 * j + 1] + t * vy; Y[k] = a; V[k] = speed[j];
        }
        len[slot] = T;
        continue;
      }
      const int nv = count[(size_t)ll * R + r];
      if (nv < 2) continue;
      const double *q = xy + 2 * (size_t)first[(size_t)ll * R + r], *sq = sarr + first[(size_t)ll * R + r];
      const double px = pos0[2 * j], py = pos0[2 * j + 1];
      double best = INFINITY, s0 = 0.0, d0 = 0.0;
      for (int i = 0; i + 1 < nv; ++i) { /* closest point of the route (first minimum) */
        const double ax = q[2 * i], ay = q[2 * i + 1], ex = q[2 * i + 2] - ax, ey = q[2 * i + 3] - ay;
        const double l2 = ex * ex + ey * ey;
        double t = ((px - ax) * ex + (py - ay) * ey) / l2;
        if (t < 0.0) t = 0.0;
        if (t > 1.0) t = 1.0;
        const double cx = ax + t * ex, cy = ay + t * ey;
        const double d2 = (px - cx) * (px - cx) + (py - cy) * (py - cy);
        if (d2 < best) {
          const double l = sqrt(l2);
          best = d2;
          s0 = sq[i] + t * l;
          d0 = ((px - cx) * (-ey) + (py - cy) * ex) / l; /* offset along the left normal */
        }
      }
      /* The Frenet sample the reference keeps (agent.py:349-379: smallest variance of the speed among the nine samples
       * of frenetix_handler.py:82-105, d1 in {-0.5, 0, 0.5} x end speed in {0.8, 1, 1.2} v0, t1 = 3 s): the one that holds
       * the speed (end speed v0: s' = v0 throughout) and moves least sideways -- d1 = the lateral target nearest to d0
       * (the first of equally near ones, in sampling order).  d(t) is the quintic with zero lateral velocity and
       * acceleration at both ends; on a straight piece of the route the Cartesian speed is sqrt(s'^2 + d'^2) and the
       * heading leaves the route's by atan(d'/s'). */
      double d1 = -0.5;
      if (fabs(0.0 - d0) < fabs(d1 - d0)) d1 = 0.0;
      if (fabs(0.5 - d0) < fabs(d1 - d0)) d1 = 0.5;
      const double t1 = 3.0; /* frenetix_handler.py:84 */
      const double s_end = sq[nv - 1];
      int m = 0, k = 0;
      for (; k < T; ++k) {
        const double tk = (double)k * dt, sk = s0 + speed[j] * tk;
        if (sk > s_end) break;
        while (m + 2 < nv && sq[m + 1] <= sk) ++m; /* segment with s[m] <= sk (last segment at the very end) */
        const double ex = q[2 * m + 2] - q[2 * m], ey = q[2 * m + 3] - q[2 * m + 1];
        const double l = sqrt(ex * ex + ey * ey), ux = ex / l, uy = ey / l, loc = sk - sq[m];
        const double tau = tk < t1 ? tk / t1 : 1.0;
        const double dk = d0 + (d1 - d0) * (tau * tau * tau * (10.0 + tau * (-15.0 + 6.0 * tau)));
        const double dd = (d1 - d0) * (30.0 * tau * tau * (1.0 + tau * (-2.0 + tau))) / t1; /* d'(t) */
        P[2 * k] = q[2 * m] + loc * ux + dk * (-uy);
        P[2 * k + 1] = q[2 * m + 1] + loc * uy + dk * ux;
        Y[k] = atan2(uy, ux) + atan2(dd, speed[j]);
        V[k] = sqrt(speed[j] * speed[j] + dd * dd);
      }
      len[slot] = k;
    }
  }
  return 0;
}

/* ---------------------------------------------------------------- future visibility (an extension)
 * NOT part of the reference: how much of the currently occluded area a candidate trajectory will come to see.  For
 * every trajectory m and every t_stride-th sample k (pose = (x, y)[m][k t_stride]) a world-aligned full fan of n_rays
 * rays of length r is cast against the static soup (every boundary piece + the occluding obstacles where they stand
 * now); revealed[m][k] = number of cells of the current occluded set (window indices occ_idx) whose centre lies within
 * r of the pose and on the pose side of the chord between the hit points of the two rays enclosing it (the fan rule of
 * fo_oracle_grid); area[m][k] = area of the polygon of hit points (shoelace).  K = (T + t_stride - 1) / t_stride. */
int fo_oracle_future_visibility(int M, int T, const double *x, const double *y, int t_stride, int n_rays,
                                const double *dirs, double r, int E, const double *edges, int O, const double *ocorn,
                                const uint8_t *oflags, int n_occ, const int32_t *occ_idx, double rx0, double ry0,
                                double cs, int ix0, int iy0, int nx, int32_t *revealed, double *area) {
  const int K = (T + t_stride - 1) / t_stride;
  double *rng = (double *)malloc(sizeof(double) * (size_t)n_rays);
  if (!rng) return -1;
  const double r2 = r * r;
  for (int m = 0; m < M; ++m) {
    for (int k = 0; k < K; ++k) {
      const double px = x[(size_t)m * T + (size_t)k * t_stride], py = y[(size_t)m * T + (size_t)k * t_stride];
      for (int i = 0; i < n_rays; ++i) {
        double t;
        int id;
        first_hit(E, edges, NULL, O, ocorn, oflags, px, py, dirs[2 * i], dirs[2 * i + 1], r, -1, &t, &id);
        rng[i] = t;
      }
      double a2 = 0.0;
      for (int i = 0; i < n_rays; ++i) {
        const int j = (i + 1 == n_rays) ? 0 : i + 1;
        const double hix = rng[i] * dirs[2 * i], hiy = rng[i] * dirs[2 * i + 1];
        const double hjx = rng[j] * dirs[2 * j], hjy = rng[j] * dirs[2 * j + 1];
        a2 += hix * hjy - hjx * hiy;
      }
      int cnt = 0;
      for (int c = 0; c < n_occ; ++c) {
        const int idx = occ_idx[c];
        const int wx = ix0 + idx % nx, wy = iy0 + idx / nx;
        const double cx = rx0 + ((double)wx + 0.5) * cs, cy = ry0 + ((double)wy + 0.5) * cs;
        const double qx = cx - px, qy = cy - py;
        if (qx * qx + qy * qy > r2) continue;
        if (qx == 0.0 && qy == 0.0) { ++cnt; continue; }
        const int i = fan_sector(n_rays, dirs, 1, qx, qy);
        if (i < 0) continue;
        const int j = (i + 1 == n_rays) ? 0 : i + 1;
        const double hix = rng[i] * dirs[2 * i], hiy = rng[i] * dirs[2 * i + 1];
        const double hjx = rng[j] * dirs[2 * j], hjy = rng[j] * dirs[2 * j + 1];
        if ((hjx - hix) * (qy - hiy) - (hjy - hiy) * (qx - hix) >= 0.0) ++cnt;
      }
      revealed[(size_t)m * K + k] = cnt;
      area[(size_t)m * K + k] = 0.5 * a2;
    }
  }
  free(rng);
  return 0;
}
