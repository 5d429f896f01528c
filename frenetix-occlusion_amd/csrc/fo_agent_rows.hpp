// fo_agent_rows.hpp -- one row of the sweep's agent table, shared by fo_prep_agents_kernel (fo_sweep.hip) and by the
// phantom prediction kernel of the fused planning step (fo_scene.hip, fo_step_run), which writes the table for its own
// slot and saves the launch.  Floating-point contraction is switched off inside so that the two translation units
// (fo_scene.hip is built with -ffp-contract=off, fo_sweep.hip is not) produce the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "fo_hip.h"

constexpr int NAF = 12;    // agent fields per (k, t): px, py, cos, sin, yaw, v, 1/(sx*sqrt2), 1/(sy*sqrt2), v cos, v sin, rho, asin rho
constexpr int NAC = 16;    // per-agent constants (fo_agent_row below)

// where fo_sweep_set_agents keeps the agent set of a context: what a kernel needs to write it
struct fo_agent_table_t {
  double *tab = nullptr, *cst = nullptr;   // [A][Ta][NAF], [A][NAC]
  int32_t *aint = nullptr;                 // [A][2]
  int *status = nullptr;                   // [2] generation tags (unusable covariance, correlated covariance)
  int gen = 0;
  double ego_mass = 0, hlA = 0, hwA = 0;
  fo_harm_coeff_t hc{};
};

// obstacle mass / protection class by type (ref: harm_model.py:15-32,158-190)
__device__ inline double fo_obstacle_mass(int type, double size) {
#pragma clang fp contract(off)   // (see the header comment: both translation units must produce the same bits)
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
__device__ inline int fo_obstacle_protection(int type) {
  switch (type) {
    case FO_TYPE_CAR: case FO_TYPE_TRUCK: case FO_TYPE_BUS: case FO_TYPE_PRIORITY_VEHICLE:
    case FO_TYPE_PARKED_VEHICLE: case FO_TYPE_TRAIN: case FO_TYPE_TAXI: return 1;
    case FO_TYPE_BICYCLE: case FO_TYPE_PEDESTRIAN: case FO_TYPE_MOTORCYCLE: case FO_TYPE_UNKNOWN: return 0;
    default: return 2;
  }
}

// one sample of an agent's prediction as fo_agent_row_core reads it (ppx, ppy: the mean of the sample before, t >= 1)
struct fo_agent_sample_t {
  double px, py, ppx, ppy, yaw, v, sxx, sxy, syx, syy;
};

// Row (k, t) of the agent table + (t == 0) the agent's constants, from values.  L: the agent's valid length, already clamped
// to [0, Ta].  WAVE_KEY: every lane of the wave calls (valid = false for lanes without a row) and all rows belong to agent k
// -- the longest step of the agent's mean is then reduced across the wave and raised by one atomic instead of one per lane.
template <bool WAVE_KEY>
__device__ __forceinline__ void fo_agent_row_core(bool valid, int k, int t, int Ta, int L, const fo_agent_sample_t &q,
                                                  double shape_l, double shape_w, double raw_l, double raw_w, int type,
                                                  double ego_mass, double hlA, double hwA, const fo_harm_coeff_t &hc,
                                                  double *__restrict__ tab, double *__restrict__ cst,
                                                  int32_t *__restrict__ aint, int *__restrict__ status, int gen,
                                                  double m_obs_known = -1.0) {
#pragma clang fp contract(off)
  unsigned long long key = 0ull;
  if (valid) {
  const size_t i = (size_t)k * Ta + t;
  double sn, cs;
  sincos(q.yaw, &sn, &cs);
  double sxx = q.sxx, sxy = q.sxy, syx = q.syx, syy = q.syy;
  if (sxx == 0.0 && sxy == 0.0 && syx == 0.0 && syy == 0.0) { sxx = 0.1; syy = 0.1; }  // collision_probability.py:84-86
  double isx = 1.0 / (sqrt(sxx) * M_SQRT2), isy = 1.0 / (sqrt(syy) * M_SQRT2);
  double rho = 0.0;
  if (sxy != 0.0 || syx != 0.0) {  // a covariance with correlation (real agents from a prediction module): the sweep
    // integrates the bivariate normal over every box (fo_corr_corners; status[1] tells the sweep kernel which of its two
    // bodies to run); a matrix that is no usable covariance -- asymmetric, not positive, |rho| > 0.99 -- poisons the
    // row and raises status[0] (fo_sweep_check)
    rho = 0.5 * (sxy + syx) / sqrt(sxx * syy);
    if (!(sxx > 0.0) || !(syy > 0.0) || fabs(sxy - syx) > 1e-12 * sqrt(sxx * syy) || !(fabs(rho) <= 0.99)) {
      if (t < L) atomicMax(status, gen);
      isx = NAN;
      isy = NAN;
      rho = 0.0;
    } else if (t < L) {
      atomicMax(status + 1, gen);
    }
  }
  double *o = tab + i * NAF;
  o[0] = q.px; o[1] = q.py; o[2] = cs; o[3] = sn; o[4] = q.yaw; o[5] = q.v;
  const double vc = fmin(fmax(q.v, -5.0e3), 5.0e3);   // (see fo_prep_traj_kernel)
  o[6] = isx; o[7] = isy; o[8] = vc * cs; o[9] = vc * sn; o[10] = rho; o[11] = rho == 0.0 ? rho : asin(rho);   // (asin(+-0) = +-0)
  // Coarse gate test of the sweep: the gate of sample t-1 takes the ego reference point of sample t against the agent
  // mean of sample t-1; the sweep tests the distance it has anyway -- shifted ego centre t to agent mean t -- against
  // 5 m + half the inflated length (c[14]) + the longest step of the agent's mean + the centre shift:
  // |e_t - p_(t-1)| <= |e_t - c_t| + |c_t - p_t| + |p_t - p_(t-1)|.  The longest step is a maximum over the threads of
  // an agent: float32 rounded up, tagged with the generation of this agent set in the high word so that the slot needs
  // no reset (an older set's key is always smaller; the slot is zero when the buffer is allocated), c[15].
  if (t >= 1 && t < L) {
    const double sx_ = q.px - q.ppx, sy_ = q.py - q.ppy;
    const double st = sqrt(sx_ * sx_ + sy_ * sy_);
    if (st == st)   // (a NaN mean is in no gate, whatever the coarse test says)
      key = ((unsigned long long)(unsigned)gen << 32) | (unsigned long long)__float_as_uint(__double2float_ru(st));
  }
  if (!WAVE_KEY && key) atomicMax((unsigned long long *)(cst + (size_t)k * NAC + 15), key);
  if (t == 0) {
    // inflated footprint (Q8); m_obs_known: the caller has worked it out already (the same call, earlier: a single wave's
    // dependent float64 chain is what a prediction workgroup's time consists of)
    const double m_obs = m_obs_known >= 0.0 ? m_obs_known : fo_obstacle_mass(type, shape_l * shape_w);
    double *c = cst + (size_t)k * NAC;
    c[0] = 0.5 * raw_l; c[1] = 0.5 * raw_w; c[2] = shape_l / 2.0;
    c[3] = m_obs / (ego_mass + m_obs); c[4] = ego_mass / (ego_mass + m_obs);
    c[5] = (double)fo_obstacle_protection(type); c[6] = (double)L; c[7] = (double)type;
    // what every wave that takes this agent would otherwise recompute
    c[8] = sqrt(hlA * hlA + hwA * hwA) + sqrt(c[0] * c[0] + c[1] * c[1]);   // circumradii: centre distance - c[8] <= distance
    c[9] = (5.0 + c[2] + 1e-6) * (5.0 + c[2] + 1e-6);   // beyond 5 m + half the inflated length no mean is in the gate
    const bool lr4s = fo_obstacle_protection(type) == 1;
    // logistic arguments as one fma of dv: the speed coefficient times the mass split is folded per agent
    // (harm_model.py:96-97: ego_dv = m_obs/(m_ego+m_obs) dv, obs_dv = m_ego/(m_ego+m_obs) dv)
    c[10] = lr4s ? -hc.lr4s_speed * c[3] : -hc.lr1s_speed * c[3];
    c[11] = lr4s ? -hc.lr4s_speed * c[4] : -hc.ped_speed * c[4];
    c[12] = -hc.lr1s_const;
    c[13] = hc.ped_const;
    c[14] = 5.0 + c[2] + 1e-6;   // coarse gate radius without the agent's longest step (below)
    aint[2 * k] = fo_obstacle_protection(type);
    aint[2 * k + 1] = L;
  }
  }
  if (WAVE_KEY) {   // all 64 lanes are here: the wave's largest key, one atomic
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned hi = __shfl_xor((unsigned)(key >> 32), off), lo = __shfl_xor((unsigned)key, off);
      const unsigned long long k2 = ((unsigned long long)hi << 32) | lo;
      key = k2 > key ? k2 : key;
    }
    if ((threadIdx.x & 63) == 0 && key) atomicMax((unsigned long long *)(cst + (size_t)k * NAC + 15), key);
  }
}

// row (k, t) = index i of the agent table + (t == 0) the agent's constants, from the arrays fo_sweep_set_agents takes
__device__ __forceinline__ void fo_agent_row(int i, int Ta, const double *__restrict__ pos, const double *__restrict__ yaw,
                                             const double *__restrict__ v, const double *__restrict__ cov,
                                             const double *__restrict__ shape, const double *__restrict__ raw,
                                             const int32_t *__restrict__ type, const int32_t *__restrict__ len,
                                             double ego_mass, double hlA, double hwA, const fo_harm_coeff_t &hc,
                                             double *__restrict__ tab, double *__restrict__ cst,
                                             int32_t *__restrict__ aint, int *__restrict__ status, int gen) {
  const int k = i / Ta, t = i % Ta;
  // valid length clamped to the table: the sweep indexes rows with min(t, L-1) and assumes L <= Ta; a longer claim
  // would read the next agent's rows
  const int L = min(max(len[k], 0), Ta);
  fo_agent_sample_t q;
  q.px = pos[2 * (size_t)i]; q.py = pos[2 * (size_t)i + 1];
  q.ppx = t >= 1 ? pos[2 * (size_t)i - 2] : 0.0; q.ppy = t >= 1 ? pos[2 * (size_t)i - 1] : 0.0;
  q.yaw = yaw[i]; q.v = v[i];
  q.sxx = cov[4 * (size_t)i]; q.sxy = cov[4 * (size_t)i + 1]; q.syx = cov[4 * (size_t)i + 2]; q.syy = cov[4 * (size_t)i + 3];
  const bool c0 = t == 0;   // (the agent's constants are written by its first row)
  fo_agent_row_core<false>(true, k, t, Ta, L, q, c0 ? shape[2 * k] : 0.0, c0 ? shape[2 * k + 1] : 0.0, c0 ? raw[2 * k] : 0.0, c0 ? raw[2 * k + 1] : 0.0,
                           c0 ? type[k] : 0, ego_mass, hlA, hwA, hc,
                           tab, cst, aint, status, gen);
}
