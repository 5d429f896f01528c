// fo_sweep.hip -- the trajectory x agent criticality sweep for gfx950 (MI355X, CDNA4).
//
// One launch evaluates DCE -> TTC/TTCE/WTTC, CP and harm/risk for M candidate trajectories x A agent predictions
// x T timesteps; it replaces M calls of FOInterface.trajectory_safety_assessment
// (ref: interface.py:216-219 -> metrics/metric.py:35-100 -> metrics/{dce,ttc,ttce,wttc,cp,hr}.py).
//
// Mapping (wave64): lane = trajectory (64 consecutive trajectories per wave, trajectory-fastest SoA tile
// [T][6][Mp] so that every per-timestep load/store of a wave is one contiguous 512-byte segment); the agent
// prediction a wave works on is wave-uniform, so its samples come through the scalar cache (s_load) and live
// in SGPRs.  The four waves of a workgroup share one trajectory tile (L1/L2 reuse) and take different agents;
// workgroups that share a tile are placed on the same XCD (blockIdx % 8) so the tile stays in that XCD's L2.
// No MFMA: this is branchy fp64 geometry + transcendentals, not a contraction.
//
// Arithmetic is float64 throughout (the reference is numpy float64; np.round(d,3) at dce.py:79 turns 1e-7
// errors into 1e-3 jumps, see DESIGN.md "Why fp64").
#include <hip/hip_runtime.h>
#include <math.h>
#include <cstdlib>
#include <type_traits>
#include "fo_ctx.hpp"
#include "fo_agent_rows.hpp"
#include "fo_prep_traj.hpp"

namespace {

constexpr int TILE = 64;   // trajectories per wave
constexpr int WAVES = 4;   // waves per workgroup
static_assert(TILE == FO_PREP_TILE, "fo_prep_traj.hpp");
constexpr int NEF = 8;     // ego fields per (t, trajectory): x, y, cos, sin, theta, v, v cos, v sin -- stored as four
                           // pairs per trajectory, [t][pair][trajectory][2]: one 16-byte load per lane fetches two
                           // fields (a vector-memory instruction costs the CU ~10 cycles whatever its width)
// Gauss-Legendre rules of the correlation integral (fo_corr_term): node counts by the largest |rho| they serve, and where
// each rule starts in the table ([t, w] pairs, t = (x + 1)/2, w = weight/(4 pi); host, fo_sweep_init_)
constexpr int GL_NR = 5;
__host__ __device__ constexpr int gl_nodes(int r) { return r == 0 ? 6 : r == 1 ? 8 : r == 2 ? 12 : r == 3 ? 20 : 24; }
__host__ __device__ constexpr int gl_first(int r) { return r == 0 ? 0 : r == 1 ? 6 : r == 2 ? 14 : r == 3 ? 26 : 46; }
constexpr int GL_TOTAL = 70;
// rule r serves asin|rho| up to GL_ASR[r] = asin(0.5, 0.7, 0.9, 0.97); the last rule the rest, |rho| <= 0.99
constexpr double GL_ASR0 = 0.5235987755982989, GL_ASR1 = 0.775397496610753, GL_ASR2 = 1.1197695149986342,
                 GL_ASR3 = 1.3252308092796046;
typedef const double __attribute__((address_space(4))) *cdp_gl_t;
                           // (96-byte rows: the 32-byte and 16-byte groups the scalar loads fetch stay naturally aligned)
// NAF = 12 agent fields per (k, t): px, py, cos, sin, yaw, v, 1/(sx*sqrt2), 1/(sy*sqrt2), v cos, v sin, rho, asin rho
// NAC = 16 per-agent constants: hl_raw, hw_raw, half_len_infl, f_ego, f_obs, prot, len, type, sum of the circumradii,
// far-gate radius^2, logistic slopes (ego, obstacle) and offsets, coarse gate radius, longest step (tagged) -- fo_agent_rows.hpp
constexpr int NPS = 14;    // partial-reduction slots
enum { PS_MIN_DCE = 0, PS_ARG_DCE, PS_MIN_TTC, PS_ARG_TTC, PS_MIN_TTCE, PS_MAX_ER, PS_MAX_OR, PS_ARG_OR, PS_MAX_EH,
       PS_MAX_OH, PS_MAX_CP, PS_MAX_HWC, PS_DCE_FLAG, PS_MAX_BTN };

// erf by table + 5th-order Taylor step.  Nodes x0 = i/128, i = 0..768 (|u| < 6; erf(6) == 1 in float64); each entry
// holds erf(x0) and g(x0) = 2/sqrt(pi) exp(-x0^2).  |delta| <= 1/256, remainder f^(6)/720 * delta^6 < 3e-16:
// the same absolute accuracy as libm erf/erfc for the box probabilities, at ~25 VALU ops + one 16-byte LDS gather
// instead of ~300 for the branchy ocml erfc (which dominated the first version of this kernel, profiles/r01_a_*).
constexpr int ERF_N = 769;
constexpr double ERF_SCALE = 128.0;

__global__ void fo_erf_table_kernel(double2 *tab) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ERF_N) return;
  const double x0 = (double)i / ERF_SCALE;
  tab[i] = make_double2(erf(x0), 1.1283791670955125738961589031 * exp(-x0 * x0));
}

// The queue kernel's erf: same nodes, fewer VALU operations.  Its copy of the table in LDS holds (erf(x0), g(x0)/128) and the
// caller hands over the argument already multiplied by 128 (folded into 1/(sigma sqrt 2), once per sample): the node index is
// the low word of |v| + 1.5 * 2^52 (no rint / convert instruction), d = |v| - node is the offset in table steps, and the
// Taylor step is evaluated in y = x0 d_true (= node * d * 2^-14) and s = d^2:
//   erf(x0 + d_true) = e + g d_true P,   P = (1 - s/3) + y (-1 + 2 y/3) + [y s/2 - y^3/3] + [s^2/10 - 2 s y^2/5 + 2 y^4/15] + ...
//   evaluated as  P = a0 + y (-1 + 2 y / 3),  a0 = 1 - s/3.
// The bracket is never evaluated: it contributes g(x0) d^5 (1/10 - 2 x0^2/5 + 2 x0^4/15) <= 1.13 * 0.1 * 256^-5 = 1.0e-13.
// FO_ERF_ORDER 3 (default) also leaves out the two cubic terms  y s/2 - y^3/3 = d^3 x0 (1/2 - x0^2/3):  what is dropped is
// g(x0) d^4 x0 (1/2 - x0^2/3), at most 0.18 * 256^-4 = 4.3e-11 per erf (at x0 = 0.6, |d| = 1/256; a fifth of that on average
// over d), i.e. <= 2.6e-10 on a collision probability (nine products of two differences of erf, / 12; measured on the bench
// batch against the oracle: see parity.float_max_abs_err of the bench line) -- a quarter of the 1e-9 every float output of
// this library is tested to, four orders inside the 1e-5 the task allows, and far below what the float32 list storage keeps.
// The price of the two terms is three instructions per erf, and the 36 erf of an in-gate sample are the one part of the
// sweep kernel whose instructions count three times (the waves that hold the few agents next to the candidates' path carry
// all of it, and their workgroups wait for them): 19 -> 16 -> 14 operations per erf took 6.5 % off the kernel (round 5).
// FO_ERF_ORDER 4: the cubic terms kept, P = a0 + y (a1 + u/3), a1 = -1 + s/2, u = y (2 - y): 1.0e-13 per erf.
// Either way the polynomial is grouped so that every fma has at most ONE constant that is not an inline operand (1.0, 2.0): a
// VOP3 instruction of this chip reads one literal / SGPR pair, and a second constant costs two v_mov_b32 per erf to park it.
#ifndef FO_ERF_ORDER
#define FO_ERF_ORDER 3
#endif
// (Round 5, measured and dropped: the scale of y folded into the two constants of the inner fma, both parked in vector
// registers by the caller -- one multiplication less per erf -- 0.529 ms against 0.515: four registers more across the box
// loops of a kernel that sits at its register cap cost thirteen more spilled ones.)
__device__ __forceinline__ double fo_erf_fast128(const double2 *__restrict__ tab, double v) {
  constexpr double S = 0x1p-14;
  const double av = fmin(fabs(v), 768.0);
  const double MAGIC = 6755399441055744.0;  // 1.5 * 2^52
  const double tm = av + MAGIC;
  const double fi = tm - MAGIC;               // rint(|v|), exact
  const int i = __double2loint(tm);
  const double d = av - fi;
  const double2 e = tab[i];
  const double sq = d * d;
  const double a0 = fma(sq, -S / 3.0, 1.0);
  const double y = fi * (d * S);
#if FO_ERF_ORDER >= 4
  const double a1 = fma(sq, S / 2.0, -1.0);
  const double u = y * (2.0 - y);
  const double p = fma(y, fma(u, 1.0 / 3.0, a1), a0);
#else
  const double p = fma(fma(y, 2.0 / 3.0, -1.0), y, a0);
#endif
  return copysign(fma(e.y * d, p, e.x), v);
}

// offset of ego field f from a row pointer that already points at the lane's first pair (row base + 2 lane)
#define EF(f) ((((f) >> 1) * 2 * TILE) + ((f) & 1))
typedef double fo_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fo_d2 fo_ld2(const double *p) { return *(const fo_d2 *)p; }

constexpr int EXP_N = 256;
__global__ void fo_exp_table_kernel(double *tab) {
  if (threadIdx.x < EXP_N) tab[threadIdx.x] = exp2((double)threadIdx.x / (double)EXP_N);
}

__device__ __forceinline__ double fo_erf_lds(const double2 *__restrict__ tab, double u) {
  const double au = fmin(fabs(u), 6.0);
  const double fi = __builtin_rint(au * ERF_SCALE);
  const double x0 = fi * (1.0 / ERF_SCALE);
  const double d = au - x0;
  const double2 e = tab[(int)fi];
  const double q = x0 * x0;
  const double a2 = (2.0 * q - 1.0) * (1.0 / 3.0);
  const double a3 = -x0 * (2.0 * q - 3.0) * (1.0 / 6.0);
  const double a4 = (4.0 * q * q - 12.0 * q + 3.0) * (1.0 / 30.0);
  const double p = 1.0 + d * (-x0 + d * (a2 + d * (a3 + d * a4)));
  return copysign(e.x + e.y * d * p, u);
}

// sqrt by one Goldschmidt step on v_rsq_f64 (relative error ~1e-14 instead of the correctly rounded ~25-instruction
// expansion of sqrt()); x >= 0, x = 0 -> 0.  Consumers are rounded to 1e-3 / compared at 1e-9.
#ifndef FO_DIET
#define FO_DIET 1   // 0: tuning builds -- the scalar-instruction diet of round 3 switched off (clamped row addresses, literals)
#endif
__device__ __forceinline__ double fo_sqrt(double x) {
#if FO_DIET
  // x = 0: rsq gives +inf, the Goldschmidt step NaN, and v_max_f64(NaN, 0) = 0 -- a guard that needs no float64 literal
  // (1e-300 costs two s_mov per use: a scalar instruction is as dear to its wave as a vector one)
  const double g = __builtin_amdgcn_rsq(x);
  double y = x * g;
  const double h = 0.5 * g;
  const double r = fma(-h, y, 0.5);
  y = fma(y, r, y);
  double z;
  asm("v_max_f64 %0, %1, 0" : "=v"(z) : "v"(y));
  return z;
#else
  const double g = __builtin_amdgcn_rsq(fmax(x, 1e-300));
  double y = x * g;
  const double h = 0.5 * g;
  const double r = fma(-h, y, 0.5);
  y = fma(y, r, y);
  return y;
#endif
}
// Round 5: the same for x > 0 -- the squared relative speeds of the ring, which pass 1 writes with the smallest denormal added
// (fo_sq_sum_pos: an inline integer constant 1 in a float64 operand IS that number, no literal, no extra instruction), so that
// the guard of fo_sqrt is not needed where pass 2 takes the root: one instruction per list entry.
#ifndef FO_P2_TINY
#define FO_P2_TINY 1
#endif
__device__ __forceinline__ double fo_sqrt_pos(double x) {
#if FO_P2_TINY
  const double g = __builtin_amdgcn_rsq(x);
  const double y = x * g;
  const double h = 0.5 * g;
  const double r = fma(-h, y, 0.5);
  return fma(y, r, y);
#else
  return fo_sqrt(x);
#endif
}
__device__ __forceinline__ double fo_sq_sum_pos(double a, double b) {   // a^2 + b^2 (+ 4.9e-324)
#if FO_P2_TINY
  double t;
  asm("v_fma_f64 %0, %1, %1, 1" : "=v"(t) : "v"(b));
  return fma(a, a, t);
#else
  return fma(a, a, b * b);
#endif
}
// value with its three lowest mantissa bits replaced by u (0..7)
__device__ __forceinline__ double fo_pack_low(double v, int u) {
  const unsigned lo = ((unsigned)__double2loint(v) & ~7u) | (unsigned)u;   // (v_and_or_b32 with two inline constants)
  return __hiloint2double(__double2hiint(v), (int)lo);
}
// lanes of `mask`: b, the others a -- v_cndmask_b32 with the mask in a scalar pair (not vcc)
__device__ __forceinline__ int fo_sel_b32(unsigned long long mask, int a, int b) {
  int r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
  return r;
}
// the same for a float64 whose LOW word may stay (b = NaN or +-inf or 1.0 over a value with a zero low word, or a NaN over
// anything: a NaN is a NaN whatever its payload) -- one v_cndmask_b32 on the high word
__device__ __forceinline__ double fo_sel_hi(unsigned long long mask, double a, double b) {
  return __hiloint2double(fo_sel_b32(mask, __double2hiint(a), __double2hiint(b)), __double2loint(a));
}
#ifndef FO_EPI_SEL
#define FO_EPI_SEL 1   // 0: tuning builds -- the per-pair epilogue as plain C (compare + v_cndmask chains on vcc): +1 % on the headline kernel (measured with the product build's flags on both sides; an earlier comparison across two flag sets had the sign wrong)
#endif
// a * b + c with three distinct register operands (the compiler prefers v_mov_b64 + v_fmac_f64 when c outlives the result)
#ifndef FO_P2_FMA3
#define FO_P2_FMA3 1
#endif
#ifndef FO_PROBE_PACK
#define FO_PROBE_PACK 1
#endif
#ifndef FO_P2_RUNS
#define FO_P2_RUNS 1
#endif
#ifndef FO_P2_UNROLL
#define FO_P2_UNROLL 1
#endif
__device__ __forceinline__ double fo_fma3(double a, double b, double c) {
#if FO_P2_FMA3
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
#else
  return fma(a, b, c);
#endif
}

__device__ __forceinline__ double fo_round3(double v) { return __builtin_rint(v * 1000.0) / 1000.0; }  // np.round(v,3)
// r / 1000.0, correctly rounded, for finite r: q = r RN(1/1000), one fma for the exact remainder, one for the correction
// (three operations instead of the ~30 of a float64 division; checked against true division for every integer below 2e7)
__device__ __forceinline__ double fo_div1000(double r) {
  const double q = r * 0.001;
  return fma(fma(-q, 1000.0, r), 0.001, q);
}
__device__ __forceinline__ double fo_round3_fast(double v) { return fo_div1000(__builtin_rint(v * 1000.0)); }

// ------------------------------------------------------------------------------------------------ prep kernels
// trajectories [M][T] (row per trajectory) -> tile table [tile][T][NEF][64], transposed through LDS so that both the
// HBM read (along T) and the HBM write (along trajectories) are contiguous; sincos(theta) is taken once here.
// Within a tile every (t, field) row is 512 contiguous bytes = one wave-wide load, and field / timestep strides
// are compile-time constants (immediate offsets in the sweep's loads).
__global__ __launch_bounds__(256) void fo_prep_traj_kernel(const fo_prep_args_t p) {
  extern __shared__ double sh[];  // [2][tz][TILE+1]
  fo_prep_traj_block(p, blockIdx.x, blockIdx.y, blockIdx.z, sh);   // fo_prep_traj.hpp
}

// agent predictions -> [A][Ta][NAF] table + [A][NAC] constants (one thread per (k, t); fo_agent_rows.hpp)
__global__ void fo_prep_agents_kernel(int A, int Ta, const double *__restrict__ pos, const double *__restrict__ yaw,
                                      const double *__restrict__ v, const double *__restrict__ cov,
                                      const double *__restrict__ shape, const double *__restrict__ raw,
                                      const int32_t *__restrict__ type, const int32_t *__restrict__ len,
                                      double ego_mass, double hlA, double hwA, fo_harm_coeff_t hc,
                                      double *__restrict__ tab, double *__restrict__ cst,
                                      int32_t *__restrict__ aint, int *__restrict__ status, int gen) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A * Ta) return;
  fo_agent_row(i, Ta, pos, yaw, v, cov, shape, raw, type, len, ego_mass, hlA, hwA, hc, tab, cst, aint, status, gen);
}

// The five per-timestep lists of hr.py:87-98 for sample index i = (k (T-1) + t) M + m, n = A (T-1) M entries per list
// (layout of include/fo_hip.h): cp alone, the two harms and the two risks as interleaved pairs -- a lane writes 8 + 16 +
// 16 bytes with three store instructions, each covering one contiguous run of the wave (512 B / 1 KB / 1 KB).
template <bool NT = true>
__device__ __forceinline__ void fo_store_lists(double *lists, size_t n, size_t i, double cp, double eh, double oh,
                                               double er, double orr) {
  fo_d2 *h = (fo_d2 *)(lists + n) + i, *r = (fo_d2 *)(lists + 3 * n) + i;
  if (NT) {
    __builtin_nontemporal_store(cp, lists + i);
    __builtin_nontemporal_store(fo_d2{eh, oh}, h);
    __builtin_nontemporal_store(fo_d2{er, orr}, r);
  } else {
    lists[i] = cp;
    *h = fo_d2{eh, oh};
    *r = fo_d2{er, orr};
  }
}

// The same three blocks with float32 elements (fo_sweep_set_list_format(FO_LISTS_F32): the storage SURVEY 8d prices,
// 648 B per pair): cp float [n], (ego harm, obstacle harm) float2 [n], (ego risk, obstacle risk) float2 [n] -- a lane
// writes 4 + 8 + 8 bytes, each store one contiguous run of the wave (256 B / 512 B / 512 B).
typedef float fo_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fo_store_lists_f32(float *lists, size_t n, size_t i, float cp, float eh, float oh, float er,
                                                   float orr) {
  __builtin_nontemporal_store(cp, lists + i);
  __builtin_nontemporal_store(fo_f2{eh, oh}, (fo_f2 *)(lists + n) + i);
  __builtin_nontemporal_store(fo_f2{er, orr}, (fo_f2 *)(lists + 3 * n) + i);
}
// 1 / (1 + exp(nz)) in float32 on the hardware transcendentals (v_exp_f32, v_rcp_f32: 8 cycles each against ~70 for the
// float64 table route): for the float32 list entries only -- every maximum, risk and cost entry stays float64.  |error|
// < 4e-7 absolute (argument rounding 6e-8 |nz| times the slope <= 1/4, one ulp each for exp2 and rcp).
__device__ __forceinline__ float fo_logistic_neg_f32(double nz) {
  const float e = __builtin_amdgcn_exp2f((float)nz * 1.44269504f);   // +inf for large nz -> rcp gives 0; 0 for very negative nz -> 1
  return __builtin_amdgcn_rcpf(1.0f + e);
}

// ------------------------------------------------------------------------------------------------ the sweep
enum { LST_NONE = 0, LST_F64 = 1, LST_F32 = 2, LST_F32X = 3 };   // per-timestep list output of a sweep instantiation
// LST_F32X (FO_LISTS_F32_EXACT): float32 elements like LST_F32, but every entry is the float64 result rounded at the store --
// the arithmetic of LST_F64, the bytes of LST_F32; what the float32 shortcut of LST_F32 saves is the difference of the two
__host__ __device__ constexpr bool lst_is32(int l) { return l == LST_F32 || l == LST_F32X; }
// (Round 5, measured and dropped for FO_LISTS_F32_EXACT, all within +-0.5 % of this form: the shape of the float32-list
// instantiation -- running minima of the logistic arguments, harm maxima from the epilogue -- with float64 list entries from a
// table exponential of degree 2 on the rows without a gate lane; both logistic values of a sample through one reciprocal; the
// square root of sample t+1 taken beside the exponentials of sample t.  The instantiation stays pass 2 of the float64 lists
// with conversions at the store: every entry is the float64-list mode's entry, rounded.)
__host__ __device__ constexpr bool lst_exact(int l) { return l == LST_F64 || l == LST_F32X; }
struct SweepArgs {
  int M, Mp, T, A, Ta, n_tiles, nt8, apw;  // apw = agents per wave
  const double *traj;    // [n_tiles][T][NEF][64]
  const double *atab;    // [A][Ta][NAF]
  const double *acst;    // [A][NAC]
  const double2 *erf_tab;  // [ERF_N]
  const double *exp_tab;   // [EXP_N]  2^(j/EXP_N)
  const double *gl;        // [GL_TOTAL][2] Gauss-Legendre nodes and weights (correlated covariances)
  const int *status;       // [2] generation tags of fo_prep_agents_kernel: [0] unusable covariance, [1] correlated one
  int gen;                 // generation of the current agent set
  const int32_t *aint;     // [A][2] protection class, valid length
  double *partial;       // [n_chunks][NPS][Mp]
  double *pair_f;        // [NPF][A][M] or null
  int32_t *pair_i;       // [NPI][A][M] or null
  double *lists;         // [NL][A][T-1][M] or null
  signed char *be_mask;  // [A][Mp] 1 where the pair collides at ttc > 0 (only with FO_M_BE), else null
  double hlA, hwA, wb, len3, off_x, off_y;  // ego half dims, rear-axle offset, L/3, L/6, W/2
  fo_harm_coeff_t hc;
  double dt, thr_dce;
  uint32_t mask;
  uint32_t ablate;  // debug only (env FO_SWEEP_ABLATE): 1 skip DCE, 2 skip CP box sums, 4 skip harm -- wrong results, timing aid
  // Tapered grid: the chunks of a tile shrink towards the end of the launch (workgroups are dispatched in blockIdx order,
  // chunk-major): ph_n[0] chunks of 4 x ph_a[0] agents, then ph_n[1] of 4 x ph_a[1], ..., the rest of 4 x ph_a[3].
  // A workgroup lives ~40 us per agent of its waves; at the end of a launch the chip drains for about half a workgroup
  // life (tools/wg_trace.py: with 16 agents per workgroup throughout, the last fifth of the launch runs half empty) --
  // short workgroups there cut the drain, long ones before keep the per-workgroup start-up (table fill, cross-wave
  // fold) off most of the work.
  // The decode sits in a table, [chunk] -> (first agent of wave 0, agents per wave), which fo_prep_traj_kernel writes
  // before every sweep: two scalar loads here (a decode loop over the phases in this kernel tipped its register
  // allocation over: SGPR spills through scratch memory, twice the run time).
  const int *chunk_tab;
#if FO_TRACE
  long long *trace;  // tuning builds (-DFO_TRACE=1): per workgroup start / end wall clock (100 MHz) + hardware id
#endif
};

__device__ __forceinline__ double fo_lr4s_coef(double ang, double side, double rear) {
  const double t_a = 45.0 / 180.0 * M_PI, t_b = 3.0 * t_a;  // logistic_regression.py:28-29
  if (-t_a < ang && ang < t_a) return 0.0;
  if (t_a <= ang && ang < t_b) return side;
  if (-t_a >= ang && ang > -t_b) return side;
  return rear;  // un-wrapped angle: everything else is "rear" (Q5)
}

// LR4S angle classes without atan2 (queue kernel).  The reference bins the UN-wrapped angle ang = rel - heading,
// rel = atan2(dy, dx) in (-pi, pi]  (logistic_regression.py:28-42, Q5): front |ang| < pi/4, side pi/4 <= |ang| < 3pi/4,
// rear otherwise.  Write ang = phi + 2 pi k with phi the wrapped angle: k != 0 implies |ang| >= pi, i.e. rear, and for
// k == 0 the class of phi follows exactly from the signs of  S = d x h  and  C = d . h  (h = unit heading):
// front  C > |S|,  rear  -C >= |S|,  side otherwise.  k != 0 <=> |rel - heading| > pi only has to be decided when phi is
// not rear, where |rel - heading| is either < 3pi/4 or > 5pi/4 -- a float32 atan2 estimate (error < 0.01) is enough.
#ifndef FO_ATAN2_DIAMOND
#define FO_ATAN2_DIAMOND 1
#endif
#ifndef FO_ATAN2_GUARD
#define FO_ATAN2_GUARD 1   // round 6: a NaN of the float32 estimate (offsets of ~1e-40 m: both casts flush to zero) sends the sample to the float64 route
#endif
__device__ __forceinline__ float fo_atan2_crude(float y, float x) {
#if FO_ATAN2_DIAMOND
  // Round 5: the "diamond angle" -- pi/2 (1 - x / (|x| + |y|)) with the sign of y: monotonic in the true angle, exact on the
  // axes and the diagonals, 0.071 rad off at worst (the decision it feeds has pi/4 of room, see above).  No comparison, no
  // select: the three v_cmp + v_cndmask pairs of the octant form each held the SIMD for ten cycles beyond their own issue.
  // (x = y = 0 does not get here: the caller puts dx = 1 for coincident centres.  Offsets that are nonzero in float64 but vanish
  // -- or overflow, or are denormal -- as float32 give 0 * inf = NaN or +-inf here, never a value in [-pi, pi]: the callers hand
  // such a sample to the reference's own float64 route, FO_ATAN2_GUARD.)
  const float q = x * __builtin_amdgcn_rcpf(fabsf(x) + fabsf(y));
  return copysignf(fmaf(q, -1.57079633f, 1.57079633f), y);
#else
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  const float a = mx > 0.0f ? mn * __builtin_amdgcn_rcpf(mx) : 0.0f;
  float r = a * (0.78539816f + 0.273f * (1.0f - a));
  if (ay > ax) r = 1.57079633f - r;
  if (x < 0.0f) r = 3.14159265f - r;
  return copysignf(r, y);
#endif
}

// coefficient of the class: 0 (front), side, rear.  (dx, dy): from the vehicle whose occupants are rated to the other
// party, as seen by atan2; `turn` = what is added to rel before the heading is subtracted (0 for the ego, pi for the
// obstacle: obs_ang = pi + rel - yaw, harm_model.py:89-90).
__device__ __forceinline__ double fo_lr4s_coef_dir(double dx, double dy, double hc, double hs, float rel_crude, float turn,
                                                   double heading, double side, double rear) {
  const double sg = (turn != 0.0f) ? -1.0 : 1.0;  // direction of angle rel + turn
  const double S = sg * (dy * hc - dx * hs), Cc = sg * (dx * hc + dy * hs);
  const double aS = fabs(S);
  const bool unwrapped_far = fabsf(turn + rel_crude - (float)heading) > 3.14159265f;
  if (unwrapped_far || -Cc >= aS) return rear;
  if (Cc > aS) return 0.0;
  return side;
}

// the same decision as an index (0 front, 1 side, 2 rear) -- the queue kernel takes it in pass 1, where the poses are
// in registers anyway, and keeps two bits per sample until pass 2 looks the logistic offset up.  flip = obstacle side
// (angle rel + pi: both S and C change sign).
// `band` is raised where the sample may sit on a class boundary to within rounding (|C| = |S| up to 1e-13 relative): there
// the reference's own floating-point route -- atan2, the subtraction, the comparison with 45/180 pi -- decides, and may
// decide either way (a heading of exactly -pi/4 with the other party exactly on the x axis gives ang == t_a: "side",
// although cos of that heading is one ulp above |sin|); the caller re-rates those samples with fo_lr4s_class_ref.
__device__ __forceinline__ unsigned fo_lr4s_class(double dx, double dy, double hc, double hs, float rel_crude, float turn,
                                                  float heading, bool flip, bool &band) {
  double S = dy * hc - dx * hs, Cc = dx * hc + dy * hs;
  if (flip) Cc = -Cc;  // |S| is all that is used of S
  const double aS = fabs(S), aC = fabs(Cc);
  // coarse and cheap here (high words within one of each other: |C| = |S| to ~1e-6); the exact 1e-13 test is taken
  // by fo_lr4s_on_boundary where the flagged samples are re-rated
  band = (unsigned)(__double2hiint(aC) - __double2hiint(aS) + 1) <= 2u;
  const bool unwrapped_far = fabsf(turn + rel_crude - heading) > 3.14159265f;
  unsigned c = (Cc > aS) ? 0u : 1u;
  if (unwrapped_far || -Cc >= aS) c = 2u;
  return c;
}
__device__ __forceinline__ bool fo_lr4s_on_boundary(double dx, double dy, double hc, double hs) {
  const double aS = fabs(dy * hc - dx * hs), aC = fabs(dx * hc + dy * hs);
  return fabs(aC - aS) <= 1e-13 * (aC + aS);
}
// the reference's binning of the un-wrapped angle itself (logistic_regression.py:28-42) as a class index
__device__ __forceinline__ unsigned fo_lr4s_class_ref(double ang) {
  const double t_a = 45.0 / 180.0 * M_PI, t_b = 3.0 * t_a;
  if (-t_a < ang && ang < t_a) return 0u;
  if ((t_a <= ang && ang < t_b) || (-t_a >= ang && ang > -t_b)) return 1u;
  return 2u;
}
// both classes of one sample by the reference's route (harm_model.py:86-90): ego class | obstacle class << 2.  A real
// call on purpose: inlined, the float64 atan2 raises the register demand of the whole kernel (measured +7 % / +34 %).
__device__ __attribute__((noinline)) unsigned fo_lr4s_classes_ref(double ddx, double ddy, double theta, double yaw) {
  const double rel = atan2(ddy, ddx);
  return fo_lr4s_class_ref(rel - theta) | (fo_lr4s_class_ref(M_PI + rel - yaw) << 2);
}

// squared distance from point (px,py) to the axis-aligned box [-hl,hl]x[-hw,hw]
__device__ __forceinline__ double fo_pt_box2(double px, double py, double hl, double hw) {
  const double qx = fmax(fabs(px) - hl, 0.0), qy = fmax(fabs(py) - hw, 0.0);
  return qx * qx + qy * qy;
}

// 1-D normal box probability  P(lo <= X <= hi)  with arguments already divided by sigma*sqrt(2)
__device__ __forceinline__ double fo_phi_diff(const double2 *__restrict__ tab, double lo, double hi) {
  return 0.5 * (fo_erf_lds(tab, hi) - fo_erf_lds(tab, lo));
}

// the correlation integral of one box (see fo_corr_corners) for the generic kernel: libm, always the 24-node rule
__device__ __forceinline__ double fo_corr_term_plain(const double *__restrict__ gl, double A, double B, double Cc, double D,
                                                     double asr) {
  gl += 2 * gl_first(GL_NR - 1);
  double acc = 0.0;
#pragma unroll 1
  for (int i = 0; i < gl_nodes(GL_NR - 1); ++i) {
    const double sn = sin(asr * gl[2 * i]), c2 = 1.0 / (1.0 - sn * sn);
    const double f = exp(-c2 * (A * A + Cc * Cc - 2.0 * sn * A * Cc)) - exp(-c2 * (B * B + Cc * Cc - 2.0 * sn * B * Cc)) -
                     exp(-c2 * (A * A + D * D - 2.0 * sn * A * D)) + exp(-c2 * (B * B + D * D - 2.0 * sn * B * D));
    acc = fma(gl[2 * i + 1], f, acc);
  }
  return acc * asr;
}

template <bool PAIR, int LISTS>
__global__ __launch_bounds__(TILE *WAVES) void fo_sweep_generic_kernel(const SweepArgs a) {
  __shared__ double red[(WAVES - 1) * NPS * TILE];
  __shared__ double2 erf_tab[ERF_N];
  for (int i = threadIdx.x; i < ERF_N; i += TILE * WAVES) erf_tab[i] = a.erf_tab[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware decode: blocks b and b+8 share an XCD (and its L2); keep every chunk of one tile on one XCD
  const int r = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int tile = (j % a.nt8) * 8 + r;
  const int chunk = j / a.nt8;
  if (tile >= a.n_tiles) return;
  const int m = tile * TILE + lane;
  const bool valid = m < a.M;
  const int T = a.T, Tm1 = a.T - 1, M = a.M, A = a.A;
  const size_t Mp = TILE;  // field stride inside a tile
  const double *tj = a.traj + (size_t)tile * T * NEF * TILE + 2 * lane;
  const bool do_dce = a.mask & FO_M_DCE, do_cp = a.mask & FO_M_CP, do_hr = a.mask & FO_M_HR;
  const bool do_ttc = a.mask & FO_M_TTC, do_ttce = a.mask & FO_M_TTCE;

  // reductions over this wave's agents (metric.py / hr.py "all" values)
  double w_min_dce = INFINITY, w_min_ttc = INFINITY, w_min_ttce = INFINITY;
  double w_max_er = 0.0, w_max_or = 0.0, w_max_eh = 0.0, w_max_oh = 0.0, w_max_cp = 0.0, w_max_hwc = 0.0;
  double w_arg_dce = -1.0, w_arg_ttc = -1.0, w_arg_or = -1.0, w_dce_flag = 0.0;

  const int k0 = (chunk * WAVES + wave) * a.apw;
  for (int kk = 0; kk < a.apw; ++kk) {
    const int k = k0 + kk;
    if (k >= A) break;
    const double *G = a.atab + (size_t)k * a.Ta * NAF;
    const double *C = a.acst + (size_t)k * NAC;
    const double hlB = C[0], hwB = C[1], hdev = C[2], f_ego = C[3], f_obs = C[4];
    const int prot = (int)C[5], L = (int)C[6];
    const int Lh = min(Tm1, L);

    if (L <= 0) {  // inactive slot (a spawn buffer that is only partly filled): no outputs enter any reduction
      if (a.be_mask) a.be_mask[(size_t)k * a.Mp + m] = 0;
      if (PAIR && valid) {
        const size_t ps_ = (size_t)A * M;
        for (int f = 0; f < FO_NPF; ++f) a.pair_f[(size_t)f * ps_ + (size_t)k * M + m] = NAN;
        for (int f = 0; f < FO_NPI; ++f) a.pair_i[(size_t)f * ps_ + (size_t)k * M + m] = 0;
      }
      if (LISTS && valid) {
        const size_t ls = (size_t)A * Tm1 * M;
        for (int t = 0; t < Tm1; ++t) {
          if (lst_is32(LISTS)) fo_store_lists_f32((float *)a.lists, ls, ((size_t)k * Tm1 + t) * M + m, NAN, NAN, NAN, NAN, NAN);
          else fo_store_lists<false>(a.lists, ls, ((size_t)k * Tm1 + t) * M + m, NAN, NAN, NAN, NAN, NAN);
        }
      }
      continue;
    }

    double dce = INFINITY;
    int tdce = 0;
    bool done = false;
    double max_er = -INFINITY, max_or = -INFINITY, max_eh = -INFINITY, max_oh = -INFINITY, max_cp = -INFINITY;
    double oh_at_cp = 0.0;
    int idx_or = 0, idx_cp = 0;

    // Software pipeline.  Ego samples: E(t) and E(t+1) are in registers when iteration t starts (CP reads t+1 early),
    // the loads of E(t+2) are issued at the top of the iteration and first touched at its bottom.  Agent rows
    // (wave-uniform -> scalar loads): row t in SGPRs, row t+1 requested at the top and first read by the CP step.
    double ex = tj[EF(0)], ey = tj[EF(1)], ec = tj[EF(2)], es = tj[EF(3)], eth = tj[EF(4)], ev = tj[EF(5)];
    const double *tj1 = tj + (size_t)min(1, T - 1) * NEF * Mp;
    double ex1 = tj1[EF(0)], ey1 = tj1[EF(1)], ec1 = tj1[EF(2)], es1 = tj1[EF(3)], eth1 = tj1[EF(4)],
           ev1 = tj1[EF(5)];
    double px = G[0], py = G[1], pc = G[2], ps = G[3], pth = G[4], pv = G[5], isx = G[6], isy = G[7];
    double asr = G[11];   // asin of the covariance's correlation (0: the box probabilities factorise)
    for (int t = 0; t < T; ++t) {
      const double *tj2 = tj + (size_t)min(t + 2, T - 1) * NEF * Mp;
      const double ex2 = tj2[EF(0)], ey2 = tj2[EF(1)], ec2 = tj2[EF(2)], es2 = tj2[EF(3)], eth2 = tj2[EF(4)],
                   ev2 = tj2[EF(5)];
      const double *gn = G + (size_t)min(t + 1, L - 1) * NAF;
      const double npx = gn[0], npy = gn[1], pc1 = gn[2], ps1 = gn[3], npth = gn[4], npv = gn[5], nisx = gn[6],
                   nisy = gn[7], nasr = gn[11];
      const double cr = pc * ec + ps * es;  // cos(yaw - theta)
      const double sr = ps * ec - pc * es;  // sin(yaw - theta)

      // ---------------- DCE (dce.py:69-88): oriented rectangle distance, rounded to 1e-3, first minimum, stop at 0
      if (do_dce && t < L && !(a.ablate & 1)) {
        const double ccx = ex + a.wb * ec, ccy = ey + a.wb * es;  // convert_dynamic_obstacle.py:73
        const double dx = px - ccx, dy = py - ccy;
        // agent centre / half axes in the ego frame
        const double ax = ec * dx + es * dy, ay = ec * dy - es * dx;
        const double ux = hlB * cr, uy = hlB * sr, wx = -hwB * sr, wy = hwB * cr;
        // ego centre / half axes in the agent frame
        const double bx = -(pc * dx + ps * dy), by = -(pc * dy - ps * dx);
        const double vx = a.hlA * cr, vy = -a.hlA * sr, zx = a.hwA * sr, zy = a.hwA * cr;
        const bool sep = (fabs(ax) > a.hlA + fabs(ux) + fabs(wx)) || (fabs(ay) > a.hwA + fabs(uy) + fabs(wy)) ||
                         (fabs(bx) > hlB + fabs(vx) + fabs(zx)) || (fabs(by) > hwB + fabs(vy) + fabs(zy));
        double d2 = 0.0;
        if (sep) {
          d2 = fo_pt_box2(ax + ux + wx, ay + uy + wy, a.hlA, a.hwA);
          d2 = fmin(d2, fo_pt_box2(ax + ux - wx, ay + uy - wy, a.hlA, a.hwA));
          d2 = fmin(d2, fo_pt_box2(ax - ux + wx, ay - uy + wy, a.hlA, a.hwA));
          d2 = fmin(d2, fo_pt_box2(ax - ux - wx, ay - uy - wy, a.hlA, a.hwA));
          d2 = fmin(d2, fo_pt_box2(bx + vx + zx, by + vy + zy, hlB, hwB));
          d2 = fmin(d2, fo_pt_box2(bx + vx - zx, by + vy - zy, hlB, hwB));
          d2 = fmin(d2, fo_pt_box2(bx - vx + zx, by - vy + zy, hlB, hwB));
          d2 = fmin(d2, fo_pt_box2(bx - vx - zx, by - vy - zy, hlB, hwB));
        }
        const double dist = fo_round3(sqrt(d2));
        if (!done && dist < dce) { dce = dist; tdce = t; }
        if (dce == 0.0) done = true;
      }

      if (t < Tm1 && (do_cp || do_hr)) {
        // ---------------- CP (collision_probability.py:69-122): ego sample t+1, agent mean/cov t, agent yaw t+1 (Q1)
        double cp = 0.0;
        if (t + 1 < L) {
          const double devx = pc1 * hdev, devy = ps1 * hdev;
          const double rx = ex1 - px, ry = ey1 - py;  // ego(t+1) - mean
          const double d0 = rx * rx + ry * ry;
          const double dp = (rx - devx) * (rx - devx) + (ry - devy) * (ry - devy);
          const double dm = (rx + devx) * (rx + devx) + (ry + devy) * (ry + devy);
          if (!(sqrt(fmin(d0, fmin(dp, dm))) > 5.0) && !(a.ablate & 2)) {  // :67,75
            const double bxs = a.len3 * ec1, bys = a.len3 * es1;  // box centre step (L/3 along heading), rear-axle based (Q2)
            double acc = 0.0;
#pragma unroll
            for (int jm = -1; jm <= 1; ++jm) {    // three means
              const double qx = rx - jm * devx, qy = ry - jm * devy;  // ego - mean_j
#pragma unroll
              for (int b = -1; b <= 1; ++b) {     // three boxes
                const double cx = qx + b * bxs, cy = qy + b * bys;
                if (asr != 0.0)   // correlated covariance (wave-uniform: one agent per wave)
                  acc += fo_corr_term_plain(a.gl, (cx - a.off_x) * isx, (cx + a.off_x) * isx, (cy - a.off_y) * isy,
                                            (cy + a.off_y) * isy, asr);
                const double fx = fo_phi_diff(erf_tab, (cx - a.off_x) * isx, (cx + a.off_x) * isx);
                const double fy = fo_phi_diff(erf_tab, (cy - a.off_y) * isy, (cy + a.off_y) * isy);
                acc += fx * fy;
              }
            }
            cp = acc / 3.0;  // :122
          }
        }
        // ---------------- harm (harm_model.py:80-107) + risk (hr.py:78-79), same index on both sides
        double eh = NAN, oh = NAN, er = NAN, orr = NAN;
        if (do_hr && t < Lh && !(a.ablate & 4)) {
          const double dv = sqrt(fmax(ev * ev + pv * pv - 2.0 * ev * pv * cr, 0.0));  // cos(pdof) = -cos(yaw-theta)
          const double ego_dv = f_ego * dv, obs_dv = f_obs * dv;
          if (prot == 1) {
            const double rel = atan2(py - ey, px - ex);  // the impact angles only enter the LR4S model
            const double ego_ang = rel - eth;
            const double obs_ang = M_PI + rel - pth;
            eh = 1.0 / (1.0 + exp(-a.hc.lr4s_const - a.hc.lr4s_speed * ego_dv -
                                  fo_lr4s_coef(ego_ang, a.hc.lr4s_side, a.hc.lr4s_rear)));
            oh = 1.0 / (1.0 + exp(-a.hc.lr4s_const - a.hc.lr4s_speed * obs_dv -
                                  fo_lr4s_coef(obs_ang, a.hc.lr4s_side, a.hc.lr4s_rear)));
          } else if (prot == 0) {
            eh = 1.0 / (1.0 + exp(-a.hc.lr1s_const - a.hc.lr1s_speed * ego_dv));
            oh = 1.0 / (1.0 + exp(a.hc.ped_const - a.hc.ped_speed * obs_dv));
          } else {
            eh = 1.0;
            oh = 1.0;
          }
          er = eh * cp;
          orr = oh * cp;
          max_er = fmax(max_er, er);
          if (orr > max_or) { max_or = orr; idx_or = t; }
          max_eh = fmax(max_eh, eh);
          max_oh = fmax(max_oh, oh);
        }
        if (cp > max_cp) { max_cp = cp; idx_cp = t; oh_at_cp = oh; }
        if (LISTS == LST_F32 && valid)   // (this kernel converts at the store; the queue kernel has a float32 harm path)
          fo_store_lists_f32((float *)a.lists, (size_t)A * Tm1 * M, ((size_t)k * Tm1 + t) * M + m, (float)cp, (float)eh,
                             (float)oh, (float)er, (float)orr);
        else if (LISTS && valid)
          fo_store_lists(a.lists, (size_t)A * Tm1 * M, ((size_t)k * Tm1 + t) * M + m, cp, eh, oh, er, orr);
      }
      ex = ex1; ey = ey1; ec = ec1; es = es1; eth = eth1; ev = ev1;
      ex1 = ex2; ey1 = ey2; ec1 = ec2; es1 = es2; eth1 = eth2; ev1 = ev2;
      px = npx; py = npy; pc = pc1; ps = ps1; pth = npth; pv = npv; isx = nisx; isy = nisy; asr = nasr;
    }

    // ---------------- per-pair scalars
    const double ttc = (fabs(dce) <= 1e-8) ? fo_round3((double)tdce * a.dt) : INFINITY;  // ttc.py:43-46
    const double ttce = fo_round3((double)tdce * a.dt);                                   // ttce.py:39
    if (a.be_mask) a.be_mask[(size_t)k * a.Mp + m] = (do_ttc && ttc < INFINITY && ttc > 0.0) ? 1 : 0;  // be.py:49-50
    const bool hr_valid = do_hr && Lh > 0;
    const double hwc = (max_cp > 0.01) ? oh_at_cp : 0.0;                                   // hr.py:81-84
    if (PAIR && valid) {
      const size_t ps_ = (size_t)A * M;
      double *pf = a.pair_f + (size_t)k * M + m;
      pf[FO_PF_DCE * ps_] = do_dce ? dce : NAN;
      pf[FO_PF_TTC * ps_] = do_ttc ? ttc : NAN;
      pf[FO_PF_TTCE * ps_] = do_ttce ? ttce : NAN;
      pf[FO_PF_MAX_EGO_RISK * ps_] = hr_valid ? max_er : NAN;
      pf[FO_PF_MAX_OBST_RISK * ps_] = hr_valid ? max_or : NAN;
      pf[FO_PF_HARM_WITH_CP * ps_] = hr_valid ? hwc : NAN;
      pf[FO_PF_MAX_EGO_HARM * ps_] = hr_valid ? max_eh : NAN;
      pf[FO_PF_MAX_OBST_HARM * ps_] = hr_valid ? max_oh : NAN;
      pf[FO_PF_MAX_CP * ps_] = hr_valid ? max_cp : NAN;
      pf[FO_PF_BE_DECEL * ps_] = NAN;
      pf[FO_PF_BE_BTN * ps_] = NAN;
      pf[FO_PF_SPARE * ps_] = NAN;
      int32_t *pi = a.pair_i + (size_t)k * M + m;
      pi[FO_PI_TIME_DCE * ps_] = do_dce ? tdce : 0;
      pi[FO_PI_RISK_INDEX * ps_] = hr_valid ? idx_or : 0;
      pi[FO_PI_CP_ARGMAX * ps_] = hr_valid ? idx_cp : 0;
      pi[FO_PI_HR_VALID * ps_] = hr_valid ? 1 : 0;
    }
    // ---------------- fold into the wave's running "all agents" values (first-wins on ties = ascending k)
    if (do_dce) {
      if (dce < w_min_dce) { w_min_dce = dce; w_arg_dce = (double)k; }
      if (dce < a.thr_dce) w_dce_flag = 1.0;  // thr NaN -> never
      if (do_ttc && ttc < w_min_ttc) { w_min_ttc = ttc; w_arg_ttc = (double)k; }
      if (do_ttce) w_min_ttce = fmin(w_min_ttce, ttce);
    }
    if (hr_valid) {
      w_max_er = fmax(w_max_er, max_er);
      if (max_or > w_max_or) { w_max_or = max_or; w_arg_or = (double)k; }
      w_max_eh = fmax(w_max_eh, max_eh);
      w_max_oh = fmax(w_max_oh, max_oh);
      w_max_cp = fmax(w_max_cp, max_cp);
      w_max_hwc = fmax(w_max_hwc, hwc);
    }
  }

  // ---------------- combine the four waves (ascending agent order) and write one partial per (chunk, trajectory)
  if (wave > 0) {
    double *rp = red + (size_t)(wave - 1) * NPS * TILE + lane;
    rp[PS_MIN_DCE * TILE] = w_min_dce; rp[PS_ARG_DCE * TILE] = w_arg_dce; rp[PS_MIN_TTC * TILE] = w_min_ttc;
    rp[PS_ARG_TTC * TILE] = w_arg_ttc; rp[PS_MIN_TTCE * TILE] = w_min_ttce; rp[PS_MAX_ER * TILE] = w_max_er;
    rp[PS_MAX_OR * TILE] = w_max_or; rp[PS_ARG_OR * TILE] = w_arg_or; rp[PS_MAX_EH * TILE] = w_max_eh;
    rp[PS_MAX_OH * TILE] = w_max_oh; rp[PS_MAX_CP * TILE] = w_max_cp; rp[PS_MAX_HWC * TILE] = w_max_hwc;
    rp[PS_DCE_FLAG * TILE] = w_dce_flag; rp[PS_MAX_BTN * TILE] = 0.0;
  }
  __syncthreads();
  if (wave == 0) {
    for (int w = 0; w < WAVES - 1; ++w) {
      const double *rp = red + (size_t)w * NPS * TILE + lane;
      if (rp[PS_MIN_DCE * TILE] < w_min_dce) { w_min_dce = rp[PS_MIN_DCE * TILE]; w_arg_dce = rp[PS_ARG_DCE * TILE]; }
      if (rp[PS_MIN_TTC * TILE] < w_min_ttc) { w_min_ttc = rp[PS_MIN_TTC * TILE]; w_arg_ttc = rp[PS_ARG_TTC * TILE]; }
      w_min_ttce = fmin(w_min_ttce, rp[PS_MIN_TTCE * TILE]);
      w_max_er = fmax(w_max_er, rp[PS_MAX_ER * TILE]);
      if (rp[PS_MAX_OR * TILE] > w_max_or) { w_max_or = rp[PS_MAX_OR * TILE]; w_arg_or = rp[PS_ARG_OR * TILE]; }
      w_max_eh = fmax(w_max_eh, rp[PS_MAX_EH * TILE]);
      w_max_oh = fmax(w_max_oh, rp[PS_MAX_OH * TILE]);
      w_max_cp = fmax(w_max_cp, rp[PS_MAX_CP * TILE]);
      w_max_hwc = fmax(w_max_hwc, rp[PS_MAX_HWC * TILE]);
      w_dce_flag = fmax(w_dce_flag, rp[PS_DCE_FLAG * TILE]);
    }
    const size_t PM = (size_t)a.Mp;
    double *pp = a.partial + (size_t)chunk * NPS * PM + (size_t)tile * TILE + lane;
    pp[PS_MIN_DCE * PM] = w_min_dce; pp[PS_ARG_DCE * PM] = w_arg_dce; pp[PS_MIN_TTC * PM] = w_min_ttc;
    pp[PS_ARG_TTC * PM] = w_arg_ttc; pp[PS_MIN_TTCE * PM] = w_min_ttce; pp[PS_MAX_ER * PM] = w_max_er;
    pp[PS_MAX_OR * PM] = w_max_or; pp[PS_ARG_OR * PM] = w_arg_or; pp[PS_MAX_EH * PM] = w_max_eh;
    pp[PS_MAX_OH * PM] = w_max_oh; pp[PS_MAX_CP * PM] = w_max_cp; pp[PS_MAX_HWC * PM] = w_max_hwc;
    pp[PS_DCE_FLAG * PM] = w_dce_flag; pp[PS_MAX_BTN * PM] = 0.0;
  }
}

// ================================================================================================ queue kernel
// Same arithmetic as the generic kernel, restructured around what the first profiles showed (profiles/r01_*):
// the kernel is fp64-VALU bound and 40 % of its instructions were the 36 erf evaluations of the CP box sums,
// executed by whole waves although only ~7 % of the (trajectory, agent, t) samples are inside the 5 m gate.
//   pass 1 (t loop)  DCE + gate test; in-gate (lane, t) samples are appended to a per-wave LDS queue with
//                    ballot/mbcnt; whenever 64 samples are queued the wave evaluates them with all lanes busy
//                    (each lane fetches "its" sample's ego/agent rows by index) and scatters cp into cpbuf[t][lane];
//   pass 2 (t loop)  harm + risk + running maxima + coalesced list stores, cp read back from cpbuf.
// exp() for the logistic models is a 64-entry 2^(j/64) table + degree-5 polynomial (~15 VALU ops).
// Supports T-1 <= TQ; longer horizons take the generic kernel.
#ifndef FO_TC
#define FO_TC 8      // timesteps per chunk of the two-pass scheme (rows of the per-wave cp buffer)
#endif
#ifndef FO_QWAVES
#define FO_QWAVES 4  // waves per workgroup of the queue kernel (45 KB of LDS -> three workgroups per CU)
#endif
#ifndef FO_MINW
#define FO_MINW 3    // waves per SIMD the register allocation has to allow (<= 168 VGPRs)
#endif
#ifndef FO_TRACE
#define FO_TRACE 0
#endif
#ifndef FO_DYN
#define FO_DYN 0     // 1: the waves of a workgroup draw the chunk's agents one by one from an LDS counter (tuning builds)
#endif
#ifndef FO_BOX_UNROLL
#define FO_BOX_UNROLL 1   // 0: tuning builds -- the three boxes of a mean one after the other outside the horizon-split form
#endif
#ifndef FO_POOL
#define FO_POOL 1    // in-gate samples pooled over the workgroup's four waves at the end of every pass 1 (0: tuning builds -- every wave evaluates its own, inline)
#endif
#ifndef FO_X
#define FO_X 0       // timing experiments only (tools/build_variant.sh x1 -DFO_X=1 ...): 1 no pass 2, 2 no probe, 4 no harm
#endif               // geometry in pass 1, 8 pass 2 without its arithmetic, 32 no DCE in pass 1, 64 no gate, 128 no second (correlated) body -- WRONG results
enum { HM_LR4S = 0, HM_DVMAX = 1, HM_GENERIC = 2 };   // pass-2 bodies by harm model (see dvmax_mode in the kernel)
constexpr int TC = FO_TC;
constexpr int DVR = TC + 1;          // rows of the per-wave ring of relative speeds: samples t0-1 .. t1-1 are live at once
constexpr int WROWS = TC + DVR;      // LDS rows (64 doubles each) per wave
constexpr int QWAVES = FO_QWAVES;
constexpr int QCAP = 128;
static_assert(TC <= 16, "two class bits per sample are kept in 32-bit lanes, 16 samples deep");

// v_max_f64 / v_min_f64 without the canonicalisation fmax()/fmin() add for loop-carried operands (IEEE quieting of
// signalling NaNs: two extra instructions per call); operands here are results of arithmetic, never signalling
__device__ __forceinline__ double fo_vmax(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double fo_vmin(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double fo_vmin_neg(double a, double b) {   // min(a, -b), the sign as a source modifier
  double r;
  asm("v_min_f64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// exp(z) = 2^(k/256) * e^r, k = rint(256 z / ln 2), |r| <= ln2/512: 256-entry table of 2^(j/256) in LDS (2 KB) and a
// degree-3 polynomial (remainder r^4/24 < 1.5e-13 relative).  One-step argument reduction: ln2/256 cut to 43
// significant bits, so k * hi is exact for |k| < 2^10 and the dropped tail costs |k| * 2.1e-16 (< 1e-12 relative over
// the arguments the logistic models produce, z in [-5e3, 6]; a logistic value moves by a quarter of that).  Few distinct
// float64 constants on purpose: every one of them occupies an SGPR pair for the whole loop.
#ifndef FO_EXP_EARLY_SCALE
#define FO_EXP_EARLY_SCALE 1
#endif
template <bool CLAMP = true, int DEG = 3>
__device__ __forceinline__ double fo_exp_tab(const double *__restrict__ tab2, double z) {
  if (CLAMP) z = fmin(fmax(z, -700.0), 700.0);  // CLAMP = false: the caller bounds the argument
  const double MAGIC = 6755399441055744.0;                          // 1.5 * 2^52
  const double tm = fma(z, 369.3299304675746, MAGIC);               // 256 / ln 2
  const double kf = tm - MAGIC;
  const int k = __double2loint(tm);
  const double r = fma(kf, -0x1.62e42fefa3800p-9, z);               // ln2/256, 43 significant bits
  const double tv = tab2[k & (EXP_N - 1)];
  double p;
  if (DEG >= 3) {
    p = fma(r, 1.0 / 6.0, 0.5);
    p = fma(p, r, 1.0);
  } else {
    p = fma(r, 0.5, 1.0);   // degree 2: remainder r^3/6 < 4.2e-10 relative (a logistic value moves by a quarter of that)
  }
  p = fma(p, r, 1.0);
#if FO_EXP_EARLY_SCALE
  return ldexp(tv, k >> 8) * p;   // the scaling beside the polynomial, not behind it (exact either way)
#else
  return ldexp(tv * p, k >> 8);
#endif
}
// 1 + exp(z), the denominator of the logistic models: the table entry is scaled while the polynomial is evaluated, and
// the product and the 1 are one fma -- mul, ldexp, add in a row became ldexp and fma (one instruction less per logistic).
template <bool CLAMP = true, int DEG = 3>
__device__ __forceinline__ double fo_exp1p_tab(const double *__restrict__ tab2, double z) {
#if FO_EXP_EARLY_SCALE
  if (CLAMP) z = fmin(fmax(z, -700.0), 700.0);
  const double MAGIC = 6755399441055744.0;
  const double tm = fma(z, 369.3299304675746, MAGIC);
  const double kf = tm - MAGIC;
  const int k = __double2loint(tm);
  const double r = fma(kf, -0x1.62e42fefa3800p-9, z);
  const double tv = ldexp(tab2[k & (EXP_N - 1)], k >> 8);
  double p;
  if (DEG >= 3) {
    p = fma(r, 1.0 / 6.0, 0.5);
    p = fma(p, r, 1.0);
  } else {
    p = fma(r, 0.5, 1.0);
  }
  p = fma(p, r, 1.0);
  return fma(tv, p, 1.0);
#else
  return 1.0 + fo_exp_tab<CLAMP, DEG>(tab2, z);
#endif
}

// 1 / (1 + exp(nz)); v_rcp_f64 (measured ~3e-8 relative) + one Newton step (1.6e-14 against the oracle)
#ifndef FO_RCP_NR
#define FO_RCP_NR 1
#endif
template <bool CLAMP = true, int DEG = 3>
__device__ __forceinline__ double fo_logistic_neg(const double *__restrict__ tab2, double nz) {
  const double d = fo_exp1p_tab<CLAMP, DEG>(tab2, nz);
  double y = __builtin_amdgcn_rcp(d);
#pragma unroll
  for (int i = 0; i < FO_RCP_NR; ++i) y = fma(fma(-d, y, 1.0), y, y);
  return y;
}

// (Round 5, measured and dropped: both logistic values of a sample through ONE reciprocal -- y = 1/(d1 d2), s1 = y d2,
// s2 = y d1: a quarter-rate v_rcp_f64 and a Newton step less for three multiplications -- 0.5398 against 0.5410 ms: the chain
// add -> mul -> rcp -> fma -> fma -> mul is two operations longer than add -> rcp -> fma -> fma, and the chain is what counts.)

// Box probabilities under a CORRELATED covariance (collision_probability.py:117 hands any 2x2 matrix to mvnun).  With
// L(h, k) = P(X > h, Y > k) for the standardised pair, Drezner & Wesolowsky / Genz write
//   L(h, k; rho) = Phi(-h) Phi(-k) + 1/(2 pi) Int_0^asin(rho) exp(-(h^2 + k^2 - 2 h k sin th) / (2 cos^2 th)) dth,
// and P(box) = L(a1,a2) - L(b1,a2) - L(a1,b2) + L(b1,b2): the Phi products add up to the diagonal box probability the
// kernel computes anyway, the integrals to a correction that vanishes with rho.  The integrand is smooth in th whatever
// the box and the variances are: Gauss-Legendre with 6 / 8 / 12 / 20 / 24 nodes for |rho| <= 0.5 / 0.7 / 0.9 / 0.97 /
// 0.99 is exact to 1e-11 (tools/corr_nodes.py), a tenth of what the erf table leaves.  Arguments here are in units of 1/(sigma sqrt 2), which cancels the 2 of the
// denominator.  sin over |th| <= asin(0.99) = 1.43: Taylor through th^21 (remainder 1e-18).
__device__ __forceinline__ double fo_sin_halfpi(double x) {
  const double z = x * x;
  double p = -1.0 / 51090942171709440000.0;            // 1/21!
  p = fma(p, z, 1.0 / 121645100408832000.0);           // 19!
  p = fma(p, z, -1.0 / 355687428096000.0);             // 17!
  p = fma(p, z, 1.0 / 1307674368000.0);                // 15!
  p = fma(p, z, -1.0 / 6227020800.0);                  // 13!
  p = fma(p, z, 1.0 / 39916800.0);                     // 11!
  p = fma(p, z, -1.0 / 362880.0);                      // 9!
  p = fma(p, z, 1.0 / 5040.0);                         // 7!
  p = fma(p, z, -1.0 / 120.0);                         // 5!
  p = fma(p, z, 1.0 / 6.0);                            // 3!  (sign below)
  return fma(-x * z, p, x);
}
// the four corner terms of one box at one node: s2 = 2 sin th, c2 = 1/cos^2 th  (four table exponentials in flight:
// serialising them to save registers was measured 40 % slower)
__device__ __forceinline__ double fo_corr_corners(const double *__restrict__ exp_tab, double A, double B, double Cc, double D,
                                                  double s2, double c2) {
  const double a2 = A * A, b2 = B * B;
  const double eAC = fo_exp_tab(exp_tab, -c2 * fma(-s2 * A, Cc, fma(Cc, Cc, a2)));
  const double eBC = fo_exp_tab(exp_tab, -c2 * fma(-s2 * B, Cc, fma(Cc, Cc, b2)));
  const double eAD = fo_exp_tab(exp_tab, -c2 * fma(-s2 * A, D, fma(D, D, a2)));
  const double eBD = fo_exp_tab(exp_tab, -c2 * fma(-s2 * B, D, fma(D, D, b2)));
  return (eAC - eBC) - (eAD - eBD);
}

// Rounded distance (whole millimetres, rint(1000 d) = 1000 np.round(d, 3), dce.py:79) between the ego rectangle at
// rear-axle pose (ex, ey, heading (ec, es)) and the agent rectangle at (px, py, heading (pc, ps)): four-axis SAT
// (overlap -> 0), otherwise the minimum over the eight corner-to-box distances.
__device__ __forceinline__ double fo_rect_mm(double ex, double ey, double ec, double es, double px, double py, double pc,
                                             double ps, double hlA, double hwA, double wb, double hlB, double hwB) {
  const double cr = pc * ec + ps * es, sr = ps * ec - pc * es;
  const double ccx = ex + wb * ec, ccy = ey + wb * es;  // convert_dynamic_obstacle.py:73
  const double dx = px - ccx, dy = py - ccy;
  const double ax = ec * dx + es * dy, ay = ec * dy - es * dx;
  const double ux = hlB * cr, uy = hlB * sr, wx = -hwB * sr, wy = hwB * cr;
  const double bx = -(pc * dx + ps * dy), by = -(pc * dy - ps * dx);
  const double vx = hlA * cr, vy = -hlA * sr, zx = hwA * sr, zy = hwA * cr;
  const double s1 = fabs(ax) - (hlA + fabs(ux) + fabs(wx)), s2 = fabs(ay) - (hwA + fabs(uy) + fabs(wy));
  const double s3 = fabs(bx) - (hlB + fabs(vx) + fabs(zx)), s4 = fabs(by) - (hwB + fabs(vy) + fabs(zy));
  if (!(fmax(fmax(s1, s2), fmax(s3, s4)) > 0.0)) return 0.0;
  double d2 = fo_pt_box2(ax + ux + wx, ay + uy + wy, hlA, hwA);
  d2 = fmin(d2, fo_pt_box2(ax + ux - wx, ay + uy - wy, hlA, hwA));
  d2 = fmin(d2, fo_pt_box2(ax - ux + wx, ay - uy + wy, hlA, hwA));
  d2 = fmin(d2, fo_pt_box2(ax - ux - wx, ay - uy - wy, hlA, hwA));
  d2 = fmin(d2, fo_pt_box2(bx + vx + zx, by + vy + zy, hlB, hwB));
  d2 = fmin(d2, fo_pt_box2(bx + vx - zx, by + vy - zy, hlB, hwB));
  d2 = fmin(d2, fo_pt_box2(bx - vx + zx, by - vy + zy, hlB, hwB));
  d2 = fmin(d2, fo_pt_box2(bx - vx - zx, by - vy - zy, hlB, hwB));
  return __builtin_rint(fo_sqrt(d2) * 1000.0);
}

// wave-uniform tables are read through the constant address space: the loads become s_load (scalar cache, results in
// SGPRs) instead of 64-lane broadcasts through the vector memory path.  The tables are written by an earlier launch
// (fo_prep_agents_kernel), so the scalar cache is coherent with them.
typedef const double __attribute__((address_space(4))) *cdp_t;
typedef const int32_t __attribute__((address_space(4))) *cip_t;
__device__ __forceinline__ cdp_t fo_const(const double *p) { return (cdp_t)(unsigned long long)p; }
__device__ __forceinline__ cip_t fo_const(const int32_t *p) { return (cip_t)(unsigned long long)p; }

// tuning builds (-DFO_TRACE=1): four more wall-clock stamps per workgroup, by wave 0, in rows [32768 + blockIdx] of the trace
// buffer (tools/split_trace.py): 0 the agent's constants and first rows resident, 1 pass 1 of the (last) chunk done, 2 pass 2
// done, 3 the agent's horizon segments folded
#if FO_TRACE
#define SW_STAMP(i) do { if (a.trace && threadIdx.x == 0) a.trace[4 * (size_t)(32768 + blockIdx.x) + (i)] = wall_clock64(); } while (0)
#else
#define SW_STAMP(i) do { } while (0)
#endif
// ALLM: the default metric set (dce, cp, ttc, ttce, hr all active, no debug ablation) is compiled with the flags as
// constants -- fewer wave-uniform masks to keep in SGPRs, fewer branches; any other selection takes the generic copy.
// SPLIT (small batches, where one agent per wave leaves most SIMDs with a single wave): the four waves of a workgroup
// take the SAME agent and a quarter of the horizon each (time chunk `wave`); every per-pair result is a minimum or a
// first maximum over time, so the segments are folded in time order through LDS at the end.  Needs T <= QWAVES * TC.
// CORR: the agent set holds a covariance with correlation (status[1] of fo_prep_agents_kernel): in-gate samples then
// add the correlation integral to their box probabilities (fo_corr_corners).  The kernel below carries both bodies and picks one at
// its start, so that the usual diagonal case keeps the registers and the code it had.
template <bool PAIR, int LISTS, bool ALLM, bool SPLIT, bool CORR, int TC_>
__device__ __forceinline__ void fo_sweep_queue_body(const SweepArgs a, const double2 *__restrict__ erf_tab,
                                                    const double *__restrict__ exp_tab, const double *__restrict__ zc_tab,
                                                    double *__restrict__ hk_all, double *__restrict__ cpbuf_all,
                                                    unsigned short *__restrict__ queue_all, int *__restrict__ next_agent,
                                                    int *__restrict__ pool_i, double *__restrict__ pool_hd) {
  constexpr int TC = TC_, DVR = TC + 1, WROWS = TC + DVR;   // this instantiation's chunk length (see fo_sweep_queue_kernel)
  constexpr bool POOL = FO_POOL != 0 && SPLIT;            // (measured: lock step costs the full grid 9 %, see pool_round)
  constexpr int QCAPX = POOL ? TILE * TC : QCAP;            // queue entries per wave: a chunk's worth with the pool
  static_assert(!SPLIT || WROWS >= 10, "the horizon-split fold parks ten values per lane in the wave's rows");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int tile = (j % a.nt8) * 8 + r;
  const int chunk = j / a.nt8;
  if (tile >= a.n_tiles) return;
  // Lanes past the last trajectory hold copies of trajectory M-1 (fo_prep_traj_kernel pads the tile that way) and
  // run as its duplicates: they compute the same values and store them to the same addresses.  No output store is
  // predicated, on purpose: a skipped store path makes the compiler's s_waitcnt for the prefetched loads assume
  // that no store lies between issue and use, which drains the store queue in every iteration of pass 2.
  const int m = min(tile * TILE + lane, a.M - 1);
  const int T = a.T, Tm1 = a.T - 1, M = a.M, A = a.A;
  const double *tjb = a.traj + (size_t)tile * T * NEF * TILE;  // uniform tile base
  const double *tj = tjb + 2 * lane;
  double *cpw = cpbuf_all + wave * (WROWS * TILE);
  double *dvw = cpw + TC * TILE;
  double *hk = hk_all + wave * 4;
  unsigned short *q = queue_all + wave * QCAPX;
  const bool do_dce = ALLM || (a.mask & FO_M_DCE), do_cp = ALLM || (a.mask & FO_M_CP), do_hr = ALLM || (a.mask & FO_M_HR);
  const bool do_ttc = ALLM || (a.mask & FO_M_TTC), do_ttce = ALLM || (a.mask & FO_M_TTCE);
  const uint32_t ablate = ALLM ? 0u : a.ablate;
  const double hlA = a.hlA, hwA = a.hwA;

  double w_min_dce = INFINITY;
  // ttc / ttce are round3(time_dce * dt), monotone in time_dce: the minima over agents are kept as integer steps
  int w_min_tttc = 0x7fffffff, w_min_tttce = 0x7fffffff;
  double w_max_er = 0.0, w_max_or = 0.0, w_max_eh = 0.0, w_max_oh = 0.0, w_max_cp = 0.0, w_max_hwc = 0.0;
  int w_arg_dce = -1, w_arg_ttc = -1, w_arg_or = -1;  // agent indices as integers: three VGPRs less than as doubles
  bool w_dce_flag = false;


  // Evaluates queued in-gate samples, one per lane (collision_probability.py:77-122).  `item` = lane | row << 6 of a queue
  // entry of agent kq (half inflated length hdq) in the chunk whose buffer row 0 holds gate sample gbq; the probability goes
  // to row `row` of the cp rows at cpq.  With the workgroup-wide pool (below) the lanes of one call hold entries of up to
  // four agents -- whichever wave evaluates them.
  auto gate_items = [&](bool valid, int item, int kq, double hdq, int gbq, double *cpq) {
    if (valid) {
      const int src = item & 63, row = item >> 6, ti = gbq + row;
      const double *e = tjb + (size_t)(ti + 1) * NEF * TILE + 2 * src;  // ego sample ti+1 of trajectory `src`
      const fo_d2 qxy = fo_ld2(e), qcs = fo_ld2(e + EF(2));
      const double qex = qxy.x, qey = qxy.y, qec = qcs.x, qes = qcs.y;
      const double *g0 = a.atab + ((size_t)kq * a.Ta + ti) * NAF;      // agent mean / covariance: sample ti
      const double qpx = g0[0], qpy = g0[1], qisx = g0[6] * ERF_SCALE, qisy = g0[7] * ERF_SCALE;
      const double qc1 = g0[NAF + 2], qs1 = g0[NAF + 3];             // agent heading: sample ti+1 (Q1); ti+1 < L
      const double devx = qc1 * hdq, devy = qs1 * hdq;
      const double rx = qex - qpx, ry = qey - qpy;
      const double bxs = a.len3 * qec, bys = a.len3 * qes;           // rear-axle based boxes (Q2)
      double acc = 0.0;
      // The 36 erf arguments are affine in (mean j, box b, side): in units of the table spacing,
      //   X(j, b, +-) = (rx - j devx + b bxs +- off_x) 128 / (sigma_x sqrt 2)
      // -- scaled once per sample, then two running sums and one add per argument instead of an add and a multiplication
      // (four operations less per box; the arguments move by ~1e-13 of a table step)
      const double DX = devx * qisx, DY = devy * qisy, BX = bxs * qisx, BY = bys * qisy;
      const double ox = a.off_x * qisx, oy = a.off_y * qisy;
      double qx = fma(rx, qisx, DX), qy = fma(ry, qisy, DY);   // j = -1
#pragma unroll 1
      for (int jm = 0; jm < 3; ++jm) {
        double cx = qx - BX, cy = qy - BY;                      // b = -1
        // the three boxes of a mean side by side (twelve table reads in flight; round 4 kept that to the horizon-split form,
        // whose heavy waves are bound by the latency of this loop -- with the shorter erf step the registers are there in
        // every form: 0.514 -> 0.508 ms on the headline; all nine boxes in a row: 0.520)
#pragma unroll (FO_BOX_UNROLL || (SPLIT && !PAIR) ? 3 : 1)
        for (int b = 0; b < 3; ++b) {
          const double fx = fo_erf_fast128(erf_tab, cx + ox) - fo_erf_fast128(erf_tab, cx - ox);
          const double fy = fo_erf_fast128(erf_tab, cy + oy) - fo_erf_fast128(erf_tab, cy - oy);
          acc = fma(fx, fy, acc);
          cx += BX; cy += BY;
        }
        qx -= DX; qy -= DY;
      }
      // (1/2)(1/2) of the two Phi differences, /3 (:122).  A row poisoned by fo_prep_agents_kernel (no usable
      // covariance: 1/sigma = NaN) must read NaN: the table erf clamps its argument, which would turn the NaN into
      // erf(+-6) and the probability into 0
      cpq[row * TILE + src] = (qisx != qisx || qisy != qisy) ? NAN : acc * (0.25 / 3.0);
    }
    if (CORR) {
      // Covariances with correlation: a second walk over the same queued samples adds the correlation integral of
      // the nine boxes to the value stored above.  It re-reads its operands (nothing of the evaluation above stays
      // live: this body shares the kernel's register budget with the usual one); whole batches without a
      // correlated sample skip it, and asin(rho) = 0 makes it vanish lane by lane.
      __asm__ volatile("" ::: "memory");
      const double asr = valid ? a.atab[((size_t)kq * a.Ta + gbq + (item >> 6)) * NAF + 11] : 0.0;
      if (__ballot(asr != 0.0)) {
        const double ar = fabs(asr);   // asin is monotonic: the rule thresholds are compared as angles
        const int rule = __ballot(ar > GL_ASR3) ? 4 : __ballot(ar > GL_ASR2) ? 3 : __ballot(ar > GL_ASR1) ? 2
                         : __ballot(ar > GL_ASR0) ? 1 : 0;
        const cdp_gl_t gl = (cdp_gl_t)(unsigned long long)(a.gl + 2 * gl_first(rule));
        const int nn = gl_nodes(rule);
        if (valid) {
          const int src = item & 63, row = item >> 6, ti = gbq + row;
          const double *e = tjb + (size_t)(ti + 1) * NEF * TILE + 2 * src;
          const double *g0 = a.atab + ((size_t)kq * a.Ta + ti) * NAF;
          const fo_d2 qxy = fo_ld2(e), qcs = fo_ld2(e + EF(2));
          // everything in units of the standard deviations (times sqrt 2) along x and y
          const double ix0 = g0[6], iy0 = g0[7];
          const double rx = (qxy.x - g0[0]) * ix0, ry = (qxy.y - g0[1]) * iy0;
          const double devx = g0[NAF + 2] * hdq * ix0, devy = g0[NAF + 3] * hdq * iy0;
          const double bxs = a.len3 * qcs.x * ix0, bys = a.len3 * qcs.y * iy0;
          const double ox = a.off_x * ix0, oy = a.off_y * iy0;
          double csum = 0.0;
#pragma unroll 1
          for (int i = 0; i < nn; ++i) {
            const double sn = fo_sin_halfpi(asr * gl[2 * i]);   // node and weight are wave-uniform: scalar loads
            const double c2 = 1.0 / fma(-sn, sn, 1.0);
            double S = 0.0;
#pragma unroll 1
            for (int jm = -1; jm <= 1; ++jm) {
              const double qx = rx - jm * devx, qy = ry - jm * devy;
#pragma unroll 1
              for (int b = -1; b <= 1; ++b) {
                const double cx = qx + b * bxs, cy = qy + b * bys;
                S += fo_corr_corners(exp_tab, cx - ox, cx + ox, cy - oy, cy + oy, 2.0 * sn, c2);
              }
            }
            csum = fma(gl[2 * i + 1], S, csum);
          }
          cpq[row * TILE + src] = fma(asr * (1.0 / 3.0), csum, cpq[row * TILE + src]);
        }
      }
    }
  };
  // Workgroup-wide pool (POOL).  The gate work is the one part of the sweep that is NOT spread evenly: on the bench batch 56 of
  // the 256 agents have any sample inside the 5 m gate and 26 of them hold 84 % of the 1.4 million in-gate samples -- the wave
  // that holds such an agent evaluates up to seventeen batches of 36 x 64 erf for it while its three siblings have none, and
  // the workgroup lives as long as that wave (tools/sweep_stats.py; the model in DESIGN.md section 3.1 puts 5-20 % of the
  // wave slots of a launch into waiting for it).  So pass 1 only QUEUES its in-gate samples (a chunk's worth: up to 64 x TC per
  // wave), and at the end of pass 1 the four waves of the workgroup meet (the chunk loop runs in step for that: every wave
  // takes part in every round, with an empty queue where its agent slot is unused), pool their queues and deal the batches
  // of 64 round robin: every wave evaluates a quarter of the workgroup's samples, whoever queued them, and writes the
  // probabilities into the owner's rows.  A second barrier, then pass 2 as before.  Fuller batches come with it (one
  // remainder per workgroup and chunk instead of four).
  auto pool_round = [&](int qn_, int k_, double hd_, int gb_) {
    static_assert(!POOL || QWAVES == 4, "the pool's prefix over the waves' queue lengths is written for four waves");
    if (lane == 0) { pool_i[wave] = qn_; pool_i[QWAVES + wave] = k_; pool_i[2 * QWAVES + wave] = gb_; pool_hd[wave] = hd_; }
    __syncthreads();
    const int n0 = __builtin_amdgcn_readfirstlane(pool_i[0]), n1 = __builtin_amdgcn_readfirstlane(pool_i[1]);
    const int n2 = __builtin_amdgcn_readfirstlane(pool_i[2]), n3 = __builtin_amdgcn_readfirstlane(pool_i[3]);
    const int c1 = n0 + n1, c2 = c1 + n2, total = c2 + n3;
#pragma unroll 1
    for (int b = wave; (b << 6) < total; b += QWAVES) {
      const int i = (b << 6) + lane;
      const bool valid = i < total;
      const int o = valid ? (i >= n0) + (i >= c1) + (i >= c2) : 0;
      const int li = i - (o == 0 ? 0 : o == 1 ? n0 : o == 2 ? c1 : c2);
      int item = 0, kq = 0, gbq = 0;
      double hdq = 0.0;
      if (valid) { item = queue_all[o * QCAPX + li]; kq = pool_i[QWAVES + o]; gbq = pool_i[2 * QWAVES + o]; hdq = pool_hd[o]; }
      gate_items(valid, item, kq, hdq, gbq, cpbuf_all + o * (WROWS * TILE));
    }
    __syncthreads();
  };

  // agents of this wave: chunk -> (first agent, agents per wave), see SweepArgs::chunk_tab
  const int apw_ = SPLIT ? a.apw : fo_const(a.chunk_tab)[2 * chunk + 1];   // SPLIT: agents per WORKGROUP, one after the other
  const int k0 = SPLIT ? chunk * a.apw : fo_const(a.chunk_tab)[2 * chunk] + wave * apw_;
  // the samples this wave owns: everything, or time chunk `wave` of the agent the workgroup shares
  const int seg0 = SPLIT ? wave * TC : 0, seg1 = SPLIT ? min(seg0 + TC, a.T) : a.T;
  const int gfirst_ = max(seg0 - 1, 0);  // first harm / cp sample this wave evaluates for an agent
  constexpr bool DYN = FO_DYN && !SPLIT && !POOL;
  // POOL: the rounds a wave without an agent in this slot still takes part in (the barriers are the workgroup's)
  auto dead_rounds = [&]() {
    for (int t0 = seg0; t0 < seg1; t0 += TC) pool_round(0, 0, 0.0, 0);
  };
  const int kbase = k0 - wave * apw_, kcount = apw_ * QWAVES;   // the chunk's agents [kbase, kbase + kcount)
  for (int kk = 0;; ++kk) {
    int k;
    if (DYN) {   // whichever wave is free takes the chunk's next agent (ascending within a wave)
      int t_ = 0;
      if (lane == 0) t_ = atomicAdd(next_agent, 1);
      t_ = __builtin_amdgcn_readfirstlane(t_);
      if (t_ >= kcount) break;
      k = kbase + t_;
    } else {
      if (kk >= apw_) break;
      k = k0 + kk;
    }
    if (k >= A) {
      if (POOL && !SPLIT) { dead_rounds(); continue; }   // (the other waves' slots may be in use)
      break;
    }
    const cdp_t G = fo_const(a.atab) + (size_t)k * a.Ta * NAF;
    const cdp_t C = fo_const(a.acst) + (size_t)k * NAC;
    const double hlB = C[0], hwB = C[1], hdev = C[2], Rsum = C[8];
    // coarse gate radius around the agent mean of the same sample, squared (fo_prep_agents_kernel, c[14]); wave-uniform
    double gate_far2;
    {
      const unsigned long long key = *(const __attribute__((address_space(4))) unsigned long long *)(C + 15);
      #ifdef FO_NO_SMAX   // (test-the-test builds: tests/test_sweep_gpu.py::test_gate_of_agents_that_jump_between_samples must fail)
      const double smax = 0.0 * (double)(unsigned)key;
#else
      const double smax = ((unsigned)(key >> 32) == (unsigned)a.gen) ? (double)__uint_as_float((unsigned)key) : 0.0;   // no key of this set: no step
#endif
      const double gf = (C[14] + smax + fabs(a.wb)) * (1.0 + 1e-9);
      const double gf2 = gf * gf;
      gate_far2 = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(gf2)), __builtin_amdgcn_readfirstlane(__double2loint(gf2)));
    }
    // Horizon-split form: the operands of the DCE probe (below) are asked for here, together with the agent's constants and
    // its length -- one round trip instead of two at the head of a workgroup that lives for ~15 us (the probe's choice of
    // samples only seeds a threshold; any sample of the segment serves).  Four samples, every second one of the segment.
    double sp_vx[4], sp_vy[4], sp_gx[4], sp_gy[4];
    if constexpr (SPLIT) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = min(seg0 + 2 * u, T - 1);
        const fo_d2 xy = fo_ld2(tj + (size_t)t * NEF * TILE);
        sp_vx[u] = xy.x; sp_vy[u] = xy.y;
        const cdp_t g = G + (size_t)t * NAF;
        sp_gx[u] = g[0]; sp_gy[u] = g[1];
      }
    }
    const int prot = fo_const(a.aint)[2 * k], L = fo_const(a.aint)[2 * k + 1];
    const int Lh = min(Tm1, L);

    if (L <= 0) {  // inactive slot (a spawn buffer that is only partly filled): no outputs enter any reduction
      if (SPLIT && wave > 0) continue;
      if (a.be_mask) a.be_mask[(size_t)k * a.Mp + m] = 0;
      if (PAIR) {
        const size_t ps_ = (size_t)A * M;
        for (int f = 0; f < FO_NPF; ++f) a.pair_f[(size_t)f * ps_ + (size_t)k * M + m] = NAN;
        for (int f = 0; f < FO_NPI; ++f) a.pair_i[(size_t)f * ps_ + (size_t)k * M + m] = 0;
      }
      if (LISTS) {
        const size_t ls = (size_t)A * Tm1 * M;
        for (int t = 0; t < Tm1; ++t) {
          if (lst_is32(LISTS)) fo_store_lists_f32((float *)a.lists, ls, ((size_t)k * Tm1 + t) * M + m, NAN, NAN, NAN, NAN, NAN);
          else fo_store_lists<false>(a.lists, ls, ((size_t)k * Tm1 + t) * M + m, NAN, NAN, NAN, NAN, NAN);
        }
      }
      if (POOL && !SPLIT) dead_rounds();
      continue;
    }

    // Per-agent state that lives across the time chunks.
    // DCE (dce.py:69-99) = the minimum over t of the rounded rectangle distance and the EARLIEST t that attains it (the
    // reference's early stop at 0 only cuts samples after the first zero) -- a result that does not depend on the
    // order in which the samples are visited.  So a probe phase first finds, per lane, the sample where the reference
    // points are closest (a cheap loop over t) and evaluates the exact distance there; the time-ordered loop below
    // then only pays for the exact geometry of samples whose lower bound (centre distance minus circumradii, then
    // the four SAT separations) can still reach the running minimum or tie it -- a wave-level skip otherwise.
    // dce is kept in whole millimetres; thr2 = ((dce + 0.51) mm)^2 and thrR2 = ((dce + 0.51) mm + R)^2 are what the SAT
    // bound and the centre distance have to undercut (0.51: a sample that rounds to the same millimetre may still
    // win the tie on t).
    double dce = INFINITY, thr2 = INFINITY, thrR2 = INFINITY;
    int tdce = 0;
    if (do_dce && !(ablate & 1) && !(FO_X & 2) && seg0 < min(L, seg1)) {
      const int Ld = min(L, seg1);
      double bestc = INFINITY;
      int tb = seg0;
      // latency-bound by construction (two loads, five operations per sample): eight samples in flight at a time.
      // Every second sample is enough for a seed (on the bench workload the exact geometry runs as rarely as with all
      // of them; stride 4 would cost a quarter more) -- and halves the loads of this phase.
      constexpr int PS = 2;
#if FO_PROBE_PACK
      // Round 5: the running minimum carries its sample number in the low mantissa bits (v_bfi_b32 + v_min_f64: the earlier
      // sample wins a tie, a repeat of the last sample never does, as with the strict comparison) -- a compare and three
      // v_cndmask_b32 on vcc per sample before, and a v_cndmask on vcc holds the SIMD for 14 cycles where an add holds it
      // for 4 (tools/microbench/valu_rate.hip).  The probe only SEEDS the bound: the 2^-47 it moves a squared distance by
      // cannot change a result.
      if constexpr (SPLIT) {
        asm volatile("; probe operands resident" ::"s"(sp_gx[0]), "s"(sp_gy[0]), "s"(sp_gx[1]), "s"(sp_gy[1]), "s"(sp_gx[2]),
                     "s"(sp_gy[2]), "s"(sp_gx[3]), "s"(sp_gy[3]));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int t = seg0 + 2 * u;
          const double rx = sp_gx[u] - sp_vx[u], ry = sp_gy[u] - sp_vy[u];
          const double c2 = fo_pack_low(fma(rx, rx, ry * ry), u);
          if (t < Ld) bestc = fo_vmin(bestc, c2);   // (wave-uniform)
        }
        tb = seg0 + 2 * (int)(__double2loint(bestc) & 7);   // (bestc = inf -- NaN positions only: sample 0 of the segment)
      } else {
      int slot = 0;
#pragma unroll 1
      for (int t8 = 0; seg0 + t8 * PS < Ld; t8 += 8) {
        double vx[8], vy[8], gpx[8], gpy[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int t = min(seg0 + (t8 + u) * PS, Ld - 1);
          const fo_d2 xy = fo_ld2(tj + (size_t)t * NEF * TILE);
          vx[u] = xy.x;
          vy[u] = xy.y;
          const cdp_t g = G + (size_t)t * NAF;   // eight scalar loads in flight as well (one lgkmcnt wait for all)
          gpx[u] = g[0];
          gpy[u] = g[1];
        }
        asm volatile("; probe operands resident" ::"s"(gpx[0]), "s"(gpy[0]), "s"(gpx[1]), "s"(gpy[1]), "s"(gpx[2]),
                     "s"(gpy[2]), "s"(gpx[3]), "s"(gpy[3]), "s"(gpx[4]), "s"(gpy[4]), "s"(gpx[5]), "s"(gpy[5]),
                     "s"(gpx[6]), "s"(gpy[6]), "s"(gpx[7]), "s"(gpy[7]));
        double blk = INFINITY;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const double rx = gpx[u] - vx[u], ry = gpy[u] - vy[u];
          blk = fo_vmin(blk, fo_pack_low(fma(rx, rx, ry * ry), u));
        }
        // (block against block: one comparison per eight samples -- the mask in a scalar pair, not in vcc)
        const unsigned long long lt = __builtin_amdgcn_fcmp(blk, bestc, 4 /* olt */);
        bestc = fo_vmin(bestc, blk);
        slot = fo_sel_b32(lt, slot, t8);
      }
      tb = min(seg0 + (slot + (int)(__double2loint(bestc) & 7)) * PS, Ld - 1);
      }
#else
      if constexpr (SPLIT) {
        asm volatile("; probe operands resident" ::"s"(sp_gx[0]), "s"(sp_gy[0]), "s"(sp_gx[1]), "s"(sp_gy[1]), "s"(sp_gx[2]),
                     "s"(sp_gy[2]), "s"(sp_gx[3]), "s"(sp_gy[3]));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int t = seg0 + 2 * u;
          const double rx = sp_gx[u] - sp_vx[u], ry = sp_gy[u] - sp_vy[u];
          const double c2 = rx * rx + ry * ry;
          if (t < Ld && c2 < bestc) { bestc = c2; tb = t; }
        }
      } else
#pragma unroll 1
      for (int t8 = 0; seg0 + t8 * PS < Ld; t8 += 8) {
        double vx[8], vy[8], gpx[8], gpy[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int t = min(seg0 + (t8 + u) * PS, Ld - 1);
          const fo_d2 xy = fo_ld2(tj + (size_t)t * NEF * TILE);
          vx[u] = xy.x;
          vy[u] = xy.y;
          const cdp_t g = G + (size_t)t * NAF;   // eight scalar loads in flight as well (one lgkmcnt wait for all)
          gpx[u] = g[0];
          gpy[u] = g[1];
        }
        asm volatile("; probe operands resident" ::"s"(gpx[0]), "s"(gpy[0]), "s"(gpx[1]), "s"(gpy[1]), "s"(gpx[2]),
                     "s"(gpy[2]), "s"(gpx[3]), "s"(gpy[3]), "s"(gpx[4]), "s"(gpy[4]), "s"(gpx[5]), "s"(gpy[5]),
                     "s"(gpx[6]), "s"(gpy[6]), "s"(gpx[7]), "s"(gpy[7]));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int t = min(seg0 + (t8 + u) * PS, Ld - 1);  // repeats of the last sample cannot win (strict <)
          const double rx = gpx[u] - vx[u], ry = gpy[u] - vy[u];
          const double c2 = rx * rx + ry * ry;
          if (c2 < bestc) { bestc = c2; tb = t; }
        }
      }
#endif
      const double *e = tj + (size_t)tb * NEF * TILE;                   // per-lane sample: gathers
      const double *g = a.atab + ((size_t)k * a.Ta + tb) * NAF;
      const fo_d2 exy = fo_ld2(e), ecs = fo_ld2(e + EF(2));
      dce = fo_rect_mm(exy.x, exy.y, ecs.x, ecs.y, g[0], g[1], g[2], g[3], hlA, hwA, a.wb, hlB, hwB);
      tdce = tb;
      const double thr = (dce + 0.51) * 1e-3;
      thr2 = thr * thr;
      thrR2 = (thr + Rsum) * (thr + Rsum);
    }
    double max_er = -INFINITY, max_or = -INFINITY, max_eh = -INFINITY, max_oh = -INFINITY, max_cp = -INFINITY;
    double oh_at_cp = 0.0;
    int idx_or = 0, idx_cp = 0;
    // Without the per-sample lists the harm values are needed at the gate samples only (risk = harm x cp); their maxima
    // are the logistic of the smallest argument -- 1/(1 + exp(nz)) falls with nz -- so the other samples keep a running
    // minimum of the two arguments and the logistic is taken once per pair.
    double nze_min = INFINITY, nzo_min = INFINITY;
    // Without lists, for the agents of the two-coefficient models (pedestrian, LR1S: prot == 0) with the usual signs of
    // the speed coefficients (both slopes <= 0): the smallest logistic argument belongs to the LARGEST relative speed,
    // fma(k, dv, c) is monotonic in dv and so is its rounding -- pass 1 keeps the running maximum of dv (as -dv in
    // nze_min, no new register) and pass 2 visits the gate rows only.  Wave-uniform.
    // (round 4: in EVERY output mode -- the per-wave ring holds the SQUARED relative speed, pass 1 takes no square root, and
    // pass 2 runs one of three bodies chosen once per agent: HM_DVMAX for these agents, HM_LR4S, HM_GENERIC for the rest
    // -- agents without a harm model, speed coefficients of unusual sign.  With the lists the running maximum is kept by
    // pass 2, which walks every sample anyway; without them by pass 1, and pass 2 visits the gate rows only.  The maxima are
    // the same arithmetic in all three output modes: logistic at sqrt(max dv^2).)
    const bool dvmax_mode = FO_DIET && prot == 0 && C[10] <= 0.0 && C[11] <= 0.0 && !(FO_X & 8);
    // List stores: the three blocks (cp | harm pairs | risk pairs) from per-agent scalar bases plus two running 32-bit
    // lane offsets (element size 1x and 2x) -- no 64-bit address arithmetic per sample (fo_sweep_run sends batches whose
    // (T-1) M pair elements pass 4 GB to the generic kernel)
    const size_t ls = (size_t)A * Tm1 * M;
    constexpr unsigned LE = lst_is32(LISTS) ? 4u : 8u;   // list element size
    char *const lb0 = (char *)a.lists + (size_t)k * Tm1 * M * LE;
    char *const lb1 = (char *)a.lists + (ls + (size_t)k * Tm1 * M * 2) * LE;
    char *const lb2 = (char *)a.lists + (3 * ls + (size_t)k * Tm1 * M * 2) * LE;
    unsigned lo1 = (unsigned)(gfirst_ * M + m) * LE, lo2 = (unsigned)(gfirst_ * M + m) * (2u * LE);
    // logistic arguments as one fma of dv: the speed coefficient times the mass split is folded per agent
    // (harm_model.py:96-97: ego_dv = m_obs/(m_ego+m_obs) dv, obs_dv = m_ego/(m_ego+m_obs) dv)
    const bool lr4s = prot == 1;
    // The logistic slopes and offsets (fo_prep_agents_kernel) are wave-uniform, but the scalar registers are taken:
    // parked in LDS, pass 2 reads them back into vector registers that are free by then (held across pass 1 they would
    // cost eight VGPRs at its register peak).
    if (lane < 4) hk[lane] = a.acst[(size_t)k * NAC + 10 + lane];
    // LR4S impact classes (0 front, 1 side, 2 rear) of the ego's and the obstacle's occupants: two bits per sample,
    // slot t & 15 (a chunk and its predecessor's last sample are live at once: TC + 1 <= 16 slots)
    unsigned cls_e = 0u, cls_o = 0u;

    // The horizon is walked in chunks of TC iterations, two passes per chunk.
    //  pass 1, iteration t: everything that needs the poses -- DCE(t); the relative speed of sample t (harm_model.py:
    //          92-94) into the wave's ring in LDS and, for LR4S agents, the impact classes of sample t; the gate of
    //          sample t-1 (ego t, agent mean t-1, agent heading t: Q1), so chunk [t0, t1) owns the gate samples
    //          [t0-1, t1-1), whose collision probabilities go to row (g - t0 + 1) of the wave's cp buffer.
    //  pass 2, samples [t0-1, t1-1): logistic models, risk, maxima, lists -- from LDS and registers only: no vector
    //          or scalar load shares a counter with the list stores (vmcnt retires loads and stores in issue order, so
    //          a load behind five stores per iteration used to wait for their acknowledgement).
    const int gfirst = gfirst_;
#ifndef FO_CARRY
#define FO_CARRY 0   // 1: tuning builds -- the first ego row of a chunk is carried over pass 2 instead of re-loaded (measured: float32 lists 0.562 against 0.552, float64 lists 0.637 against 0.650: the 14 live registers cost more than the wait)
#endif
    // (Also measured and not kept, round 3: two register sets for the current / next rows that swap roles, the loop
    // unrolled by two, instead of one set rotated by seven v_mov_b64 and ten s_mov per sample -- 0.552 against 0.541 ms:
    // 30 spilled registers instead of 8 and a quarter more code cost more than the copies.)
    // the ego row a chunk starts with: pass 1 of the chunk before has already fetched it (its last iteration prefetches
    // row t1); carried over pass 2 in registers, the chunk's first loads do not queue behind that pass's list stores
    // (vmcnt retires in issue order: a load issued after 24 stores waits for their acknowledgement)
    fo_d2 nxy, ncs, nvv;
    double nth_ = 0.0;
    for (int t0 = seg0; t0 < seg1; t0 += TC) {
      const int t1 = min(t0 + TC, T);
      // A segment other than the first also needs the relative speed and the impact classes of the sample before it
      // (pass 2 covers the samples [t0-1, t1-1)): its pass 1 starts one sample early, for that part only.
      const int tl = (SPLIT && t0 > 0) ? t0 - 1 : t0;
      const int gbase = t0 - 1;  // gate sample of buffer row 0

      // (without the workgroup-wide pool: evaluates this wave's n (<= 64) queued samples)
      auto process = [&](int n) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        gate_items(lane < n, lane < n ? (int)q[lane] : 0, k, hdev, gbase, cpw);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      };

      // ---------------------------------------------------------------- pass 1: DCE + harm geometry + gate -> queue
      // Every operand of iteration t was requested one iteration earlier: the ego row t+1 (vector loads) and the
      // agent row t+1 (scalar loads) are issued at the top and first used at the top of the next iteration.
      unsigned gmask = 0u;  // bit row: gate sample gbase + row is inside the 5 m gate for this lane
      unsigned wgate = 0u;  // the same for the whole wave (uniform): some lane is inside the gate.  Bits 16 + (t & 15)
                            // of the same scalar: some lane's impact angle of sample t may sit on a class boundary
      int qn = 0;
      // Relative speeds: sample t sits in row t - gbase of the wave's DVR rows.  Pass 2 of this chunk starts one sample
      // before it (row 0), which the chunk before left in its last row.
      if (!(SPLIT && t0 > 0) && t0 > seg0) dvw[lane] = dvw[TC * TILE + lane];
      // the sample ranges of the DCE and of the gate as one unsigned comparison each (scalar instructions are not free:
      // DESIGN.md section 3.1): DCE on [t0, L), gate on [max(t0, 1), L)
      const bool dce_on = do_dce && !(ablate & 1) && !(FO_X & 32), gate_on = do_cp && !(ablate & 2) && !(FO_X & 64);
      const int rng_n = (dce_on || gate_on) ? max(L - t0, 0) : 0;
      const double gate_far2c = gate_on ? gate_far2 : -1.0;   // (no distance is below -1: the test never passes)
      // sample 0 has no gate (there is no sample -1): the radius of the chunk's first sample is -1 there, the loop sets
      // the real one from its second sample on -- cheaper than a test of t per sample
      double gate_far2t = (tl == 0) ? -1.0 : gate_far2c;
      if (!dce_on) thrR2 = -1.0;
      const bool geo = do_hr && !(ablate & 4);
      if (!FO_CARRY || SPLIT || t0 == seg0) {
        const double *e0_ = tj + (size_t)tl * NEF * TILE;
        nxy = fo_ld2(e0_); ncs = fo_ld2(e0_ + EF(2)); nvv = fo_ld2(e0_ + EF(6));
        if (lr4s) nth_ = e0_[EF(4)];
      }
      // (FO_DIET: rows are addressed without clamping -- rows past an agent's length are read but never used, every use sits
      // behind t < L; the tables end in spare rows, fo_sweep_set_agents / fo_sweep_run)
      const cdp_t gr0 = G + (size_t)(FO_DIET ? tl : min(tl, L - 1)) * NAF;
      double px = gr0[0], py = gr0[1], npx = px, npy = py;
      // Only the mean of the next row is fetched a sample ahead (the first thing a sample needs); heading and velocity
      // are re-loaded IN PLACE right after their last use in a sample -- no second register set, no copies
      double pc = gr0[2], ps = gr0[3], pyaw = gr0[4], pvx = gr0[8], pvy = gr0[9];
      // Scalar loads return out of order, so any use of an s_load result waits for lgkmcnt(0).  Pinning the per-agent
      // constants and the first rows here (an empty asm that names them as SGPR inputs) drains the counter before the
      // loop, which leaves the in-loop wait to cover only the row that was prefetched one iteration ago.
      asm volatile("; scalar operands resident" ::"s"(hlB), "s"(hwB), "s"(hdev), "s"(Rsum), "s"(gate_far2), "s"(px),
                   "s"(py), "s"(pc), "s"(ps), "s"(pvx), "s"(pvy), "s"(pyaw));
      SW_STAMP(0);
      // rows t+1 as running 32-bit byte offsets from the (uniform) bases of this tile's and this agent's rows: one add
      // each per sample instead of a 64-bit multiply-add, and the loads take the base from scalar registers
      unsigned eoff = (unsigned)((tl * NEF * TILE + 2 * lane) * sizeof(double));
      unsigned goff = (unsigned)(tl * NAF * sizeof(double));
      for (int t = tl; t < t1; ++t) {
        // (Round 5, measured and dropped: the ego's velocity -- and heading -- of sample t asked for at the top of iteration t
        // instead of one iteration ahead with the pose -- three register pairs and three v_mov_b64 less per sample -- and no row
        // fetched ahead at all: 0.536-0.540 ms against 0.542, inside the noise, and 2 % slower together with the shorter erf step.)
        const double ex = nxy.x, ey = nxy.y, ec = ncs.x, es = ncs.y, evx = nvv.x, evy = nvv.y, eth = nth_;
        {
          eoff += (unsigned)(NEF * TILE * sizeof(double));
          goff += (unsigned)(NAF * sizeof(double));
          const double *e1 = (const double *)((const char *)tjb + eoff);
          nxy = fo_ld2(e1); ncs = fo_ld2(e1 + EF(2)); nvv = fo_ld2(e1 + EF(6));
          const cdp_t g1 = (cdp_t)((const __attribute__((address_space(4))) char *)G + goff);
          npx = g1[0]; npy = g1[1];
          if (lr4s) nth_ = e1[EF(4)];   // the headings only enter the LR4S model
        }
        const cdp_t g1 = (cdp_t)((const __attribute__((address_space(4))) char *)G + goff);
        if ((unsigned)(t - t0) < (unsigned)rng_n) {
          const double ccx = ex + a.wb * ec, ccy = ey + a.wb * es;  // convert_dynamic_obstacle.py:73
          const double dx = px - ccx, dy = py - ccy;
          const double dd = dx * dx + dy * dy;   // shared by the DCE and the gate: both start from a coarse distance test
          // the centres must be close enough.  (Nothing is to be gained after the earliest zero: the block below sets
          // thrR2 to -1 at the sample that holds it -- a zero found here, or the probe's, whose sample always passes this
          // test: overlapping rectangles have their centres within the sum of the circumradii -- so that one comparison
          // per sample serves both conditions.)
          const bool near = dd < thrR2;
          if (__ballot(near)) {
            const double cr = pc * ec + ps * es, sr = ps * ec - pc * es;
            const double ax = ec * dx + es * dy, ay = ec * dy - es * dx;   // agent centre in the ego frame
            const double ux = hlB * cr, uy = hlB * sr, wx = -hwB * sr, wy = hwB * cr;
            const double bx = -(pc * dx + ps * dy), by = -(pc * dy - ps * dx);  // ego centre in the agent frame
            const double vx = hlA * cr, vy = -hlA * sr, zx = hwA * sr, zy = hwA * cr;
            // separations along the four face normals: each is a lower bound of the distance, all <= 0 iff overlapping
            const double s1 = fabs(ax) - (hlA + fabs(ux) + fabs(wx)), s2 = fabs(ay) - (hwA + fabs(uy) + fabs(wy));
            const double s3 = fabs(bx) - (hlB + fabs(vx) + fabs(zx)), s4 = fabs(by) - (hwB + fabs(vy) + fabs(zy));
            const double lb = fmax(fmax(s1, s2), fmax(s3, s4));
            const bool overlap = !(lb > 0.0);
            bool need = near && (overlap || lb * lb < thr2);
            if (__ballot(need)) {
              // Second, tighter bound before the eight corner distances: separated along BOTH axes of one frame, the
              // rectangles are at least the diagonal of the two gaps apart (the other one's bounding box in that frame
              // misses the corner).  On the bench workload this takes a third off the exact evaluations.
              const double g1 = fmax(s1, 0.0), g2 = fmax(s2, 0.0), g3 = fmax(s3, 0.0), g4 = fmax(s4, 0.0);
              const double q = fmax(fma(g1, g1, g2 * g2), fma(g3, g3, g4 * g4));
              need = need && (overlap || q < thr2);
            }
            if (__ballot(need)) {
              double nmm = 0.0;
              if (__ballot(need && !overlap)) {
                double d2 = fo_pt_box2(ax + ux + wx, ay + uy + wy, hlA, hwA);
                d2 = fmin(d2, fo_pt_box2(ax + ux - wx, ay + uy - wy, hlA, hwA));
                d2 = fmin(d2, fo_pt_box2(ax - ux + wx, ay - uy + wy, hlA, hwA));
                d2 = fmin(d2, fo_pt_box2(ax - ux - wx, ay - uy - wy, hlA, hwA));
                d2 = fmin(d2, fo_pt_box2(bx + vx + zx, by + vy + zy, hlB, hwB));
                d2 = fmin(d2, fo_pt_box2(bx + vx - zx, by + vy - zy, hlB, hwB));
                d2 = fmin(d2, fo_pt_box2(bx - vx + zx, by - vy + zy, hlB, hwB));
                d2 = fmin(d2, fo_pt_box2(bx - vx - zx, by - vy - zy, hlB, hwB));
                if (!overlap) nmm = __builtin_rint(fo_sqrt(d2) * 1000.0);
              }
              if (need && (nmm < dce || (nmm == dce && t < tdce))) {
                dce = nmm;
                tdce = t;
                const double thr = (nmm + 0.51) * 1e-3;
                thr2 = thr * thr;
                thrR2 = (thr + Rsum) * (thr + Rsum);
              }
            }
            if (near && dce == 0.0 && t >= tdce) thrR2 = -1.0;   // the earliest zero is in: no later sample can beat it
          }
          // gate of sample t-1 (collision_probability.py:44-67,75): ego sample t, agent mean t-1, agent heading t.  The two
          // displaced means are hdev away from the mean: beyond 5 m + hdev none of the three can be in the gate, and the
          // mean of sample t-1 is at most the agent's longest step from the one of sample t (gate_far2)
          if (__ballot(dd <= gate_far2t)) {
            // (scalar loads on the rare path; the row was read a sample ago)
            const cdp_t gq = (cdp_t)((const __attribute__((address_space(4))) char *)G + (goff - 2u * (unsigned)(NAF * sizeof(double))));
            const double rx = ex - gq[0], ry = ey - gq[1];
            const double d0 = rx * rx + ry * ry;
            const double devx = pc * hdev, devy = ps * hdev;
            const double dp = (rx - devx) * (rx - devx) + (ry - devy) * (ry - devy);
            const double dm = (rx + devx) * (rx + devx) + (ry + devy) * (ry + devy);
            const double m2 = fmin(d0, fmin(dp, dm));
            // the reference tests the ROUNDED distance, !(sqrt(m2) > 5.0) (collision_probability.py:67,75).  The
            // correctly rounded square root of m2 is 5.0 up to and including m2 = 25 + one ulp (sqrt(25 (1 + d)) = 5 (1 + d/2),
            // half an ulp of 5.0 is 4.4e-16, one ulp of 25 is 3.6e-15): no square root needed
            const bool ing = m2 <= 25.000000000000004;
            const unsigned long long bal = __ballot(ing);
            if (bal) {
              const int row = t - t0;  // = (t - 1) - gbase
              const int pos = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
              if (ing) {
                q[pos] = (unsigned short)(lane | (row << 6));
                gmask |= 1u << row;
              }
              wgate |= 1u << row;
              qn += __popcll(bal);
              if (!POOL && qn >= 64) {
                process(64);
                const int rest = qn - 64;
                unsigned short tmp = 0;
                if (lane < rest) tmp = q[64 + lane];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (lane < rest) q[lane] = tmp;
                qn = rest;
              }
            }
          }
        }
        // relative speed of sample t (harm_model.py:92-94): sqrt(ve^2 + va^2 + 2 ve va cos(pdof)), pdof = yaw - theta
        // + pi, is the length of the difference of the two velocity vectors; capped (1e4 m/s) so that the logistic
        // arguments of pass 2 stay in the range of the table exp without a clamp of their own
        double dvx = evx - pvx, dvy = evy - pvy;
        {
          // (last use of this sample's velocity above: the next row's takes its place.  The empty asm orders the load
          // behind the subtraction -- issued earlier it would need registers of its own and a copy)
          unsigned gb = __builtin_amdgcn_readfirstlane(goff);
#if !FO_TRACE   // (the time-line build does without the ordering: its extra kernel argument upsets the uniformity analysis)
          asm volatile("" : "+v"(dvx), "+v"(dvy), "+s"(gb));
#endif
          const cdp_t g2 = (cdp_t)((const __attribute__((address_space(4))) char *)G + gb);
          pvx = g2[8]; pvy = g2[9];
        }
        if (geo && t < Lh && !(FO_X & 4)) {
          // squared (<= 1e8: the prep kernels cap the speeds at 5e3 m/s); pass 2 takes the root where it needs the speed
          const double dv2_ = fo_sq_sum_pos(dvx, dvy);
          dvw[(t - gbase) * TILE + lane] = dv2_;
          if (LISTS == LST_NONE && dvmax_mode) nze_min = fo_vmin_neg(nze_min, dv2_);
          if (lr4s) {
            // the impact angles only enter the LR4S model, and only through their class (front / side / rear)
            double ddx = px - ex, ddy = py - ey;
#if FO_ATAN2_DIAMOND
            // atan2(0, 0) = 0: dx = 1 for coincident centres -- |dx| + |dy| == 0, and only the high word of dx has to change
            ddx = __hiloint2double(fabs(ddx) + fabs(ddy) == 0.0 ? 0x3ff00000 : __double2hiint(ddx), __double2loint(ddx));
#else
            if (ddx == 0.0 && ddy == 0.0) ddx = 1.0;  // atan2(0, 0) = 0
#endif
            const float relc = fo_atan2_crude((float)ddy, (float)ddx);
            bool be_, bo_;
            const unsigned ce = fo_lr4s_class(ddx, ddy, ec, es, relc, 0.0f, (float)eth, false, be_);
            const unsigned co = fo_lr4s_class(ddx, ddy, pc, ps, relc, 3.14159265f, (float)pyaw, true, bo_);
#if FO_ATAN2_GUARD
            if (__ballot(be_ || bo_ || !(fabsf(relc) <= 4.0f))) wgate |= 0x10000u << (t & 15);   // re-rated after the loop (rare; see there)
#else
            if (__ballot(be_ || bo_)) wgate |= 0x10000u << (t & 15);   // re-rated after the loop (rare; see there)
#endif
            const int sh = (t & 15) * 2;
            cls_e = (cls_e & ~(3u << sh)) | (ce << sh);
            cls_o = (cls_o & ~(3u << sh)) | (co << sh);
          }
        }
        {
          // the mean fetched at the top of this sample becomes the current one BEFORE the next loads are issued: scalar
          // loads return out of order, so the wait in front of these copies would otherwise cover the loads below
          px = npx; py = npy;
          unsigned gb = __builtin_amdgcn_readfirstlane(goff);
#if !FO_TRACE
          asm volatile("" : "+s"(px), "+s"(py), "+s"(gb));
#endif
          const cdp_t g2 = (cdp_t)((const __attribute__((address_space(4))) char *)G + gb);
          pc = g2[2]; ps = g2[3];
          if (lr4s) pyaw = g2[4];
        }
        gate_far2t = gate_far2c;
      }
      if (POOL) pool_round(qn, k, hdev, gbase);
      else if (qn > 0) process(qn);
      wgate = __builtin_amdgcn_readfirstlane(wgate);  // uniform by construction; says so to the register allocator
      const unsigned wband = wgate >> 16;
      if (lr4s && wband) {
        // Impact angles on a class boundary to within rounding: the reference's own floating-point route (float64 atan2,
        // the subtraction, the comparison with 45/180 pi; harm_model.py:86-90, logistic_regression.py:28-42) decides
        // those samples -- here, outside the loop whose registers a float64 atan2 does not fit into.
        for (unsigned wb = wband; wb; wb &= wb - 1u) {
          const int slot = __builtin_ctz(wb), t = tl + ((slot - tl) & 15);
          const double *e0 = tj + (size_t)t * NEF * TILE;
          const fo_d2 xy = fo_ld2(e0), cs = fo_ld2(e0 + EF(2));
          const double th0 = e0[EF(4)];
          const cdp_t g0 = G + (size_t)min(t, L - 1) * NAF;
          double ddx = g0[0] - xy.x, ddy = g0[1] - xy.y;
          if (ddx == 0.0 && ddy == 0.0) ddx = 1.0;
#if FO_ATAN2_GUARD
          // (an offset whose float32 casts under- or overflow -- the estimate above was NaN and "far" read false: both classes
          // by the float64 route)
          const float crude_ = fo_atan2_crude((float)ddy, (float)ddx);
          const bool nf_ = !(fabsf(crude_) <= 4.0f);   // NaN (0 * inf) or +-inf (a float32 denormal times the reciprocal of one)
          const bool be_ = nf_ || fo_lr4s_on_boundary(ddx, ddy, cs.x, cs.y), bo_ = nf_ || fo_lr4s_on_boundary(ddx, ddy, g0[2], g0[3]);
#else
          const bool be_ = fo_lr4s_on_boundary(ddx, ddy, cs.x, cs.y), bo_ = fo_lr4s_on_boundary(ddx, ddy, g0[2], g0[3]);
#endif
          if (be_ || bo_) {
            const unsigned both = fo_lr4s_classes_ref(ddx, ddy, th0, g0[4]);
            const int sh = slot * 2;
            if (be_) cls_e = (cls_e & ~(3u << sh)) | ((both & 3u) << sh);
            if (bo_) cls_o = (cls_o & ~(3u << sh)) | ((both >> 2) << sh);
          }
        }
      }

      SW_STAMP(1);
      // ---------------------------------------------------------------- pass 2: harm, risk, maxima, lists
      // of the gate samples g in [max(t0-1, 0), t1-1) -- harm index g, cp index g (Q6)
      const int g0s = max(gbase, 0), g1s = t1 - 1;
      if ((do_cp || do_hr) && g0s < g1s && !(FO_X & 1)) {
        // one instantiation per harm model: the LR4S path (impact classes -> logistic offsets) and the pedestrian /
        // LR1S path keep separate register and constant sets
        auto pass2 = [&](auto hm_tag) {
          constexpr int HM = decltype(hm_tag)::value;
          constexpr bool LR4S = HM == HM_LR4S, DVMAX = HM == HM_DVMAX;
          const double ke_ = hk[0], ko_ = hk[1], ce_ = hk[2], co_ = hk[3];
          // float32 list entries of the two-coefficient models: logistic arguments in units of ln 2 (v_exp_f32 is 2^x)
          const float kef_ = (float)(ke_ * 1.4426950408889634), kof_ = (float)(ko_ * 1.4426950408889634);
          const float cef_ = (float)(ce_ * 1.4426950408889634), cof_ = (float)(co_ * 1.4426950408889634);
          // LDS reads of sample t+1 are issued while sample t is evaluated
          double dvn = dvw[(g0s - gbase) * TILE + lane];
          double zen = 0.0, zon = 0.0;
          if (LR4S) {
            const int sh = (g0s & 15) * 2;
            zen = zc_tab[(cls_e >> sh) & 3u];
            zon = zc_tab[(cls_o >> sh) & 3u];
          }
          // Rows that take the long way (wave-uniform mask, one bit per buffer row): some lane of the wave is inside the
          // gate, the wave's first sample (it seeds the running maxima and indices), and the samples past the harm
          // length.  On every other row -- 97 % of the samples of the bench workload -- every probability is zero, so are
          // the risks, and none of the maxima or indices can move.
          unsigned slow = wgate & 0xffffu;
          if (gfirst >= g0s) slow |= 1u << (gfirst - gbase);
          unsigned hvrows = geo ? ~0u : 0u;   // bit row: the sample lies inside the harm length
          if (geo && Lh < g1s) hvrows = ~(~0u << max(Lh - gbase, 0));
          slow = __builtin_amdgcn_readfirstlane(slow | ~hvrows);
          hvrows = __builtin_amdgcn_readfirstlane(hvrows);
          if (LISTS == LST_NONE && DVMAX) {
            // only the rows that take the long way; the harm maxima come from the running maximum of dv^2 (epilogue)
            unsigned todo = slow & (~0u << (g0s - gbase)) & ~(~0u << (g1s - gbase));
            while (todo) {
              const int row = __builtin_ctz(todo), t = gbase + row;
              todo &= todo - 1u;
              const double dv = fo_sqrt_pos(dvw[row * TILE + lane]);
              double cp = 0.0;
              if ((gmask >> row) & 1u) cp = cpw[row * TILE + lane];
              if ((hvrows >> row) & 1u) {
                const double eh = fo_logistic_neg<false>(exp_tab, fma(ke_, dv, ce_));
                const double oh = fo_logistic_neg<false>(exp_tab, fma(ko_, dv, co_));
                const double er = eh * cp, orr = oh * cp;
                if (er > max_er || er != er) max_er = er;
                if (orr > max_or) { max_or = orr; idx_or = t; }
                if (cp > max_cp) { max_cp = cp; idx_cp = t; oh_at_cp = oh; }
              } else if (cp > max_cp) {
                max_cp = cp; idx_cp = t; oh_at_cp = NAN;
              }
            }
          } else {
          auto row_step = [&](const int t, auto fast_tag) {
            // FASTROW (compile time): the row lies inside the harm length and no lane of the wave is inside the gate -- the
            // two mask tests, the branch on them and the long way's code are not in this copy of the body (FO_P2_RUNS)
            constexpr bool FASTROW = decltype(fast_tag)::value;
            const int row = t - gbase;
            const double dv = dvn, ze = zen, zo = zon;
            dvn = dvw[(row + 1) * TILE + lane];
            if (LR4S) {
              const int sh = ((t + 1) & 15) * 2;
              zen = zc_tab[(cls_e >> sh) & 3u];
              zon = zc_tab[(cls_o >> sh) & 3u];
            }
            double eh = NAN, oh = NAN, er = NAN, orr = NAN, cp = 0.0;
            float ehf = NAN, ohf = NAN;   // float32 lists: the harm entries
            // harm of a sample inside the harm length (wave-uniform)
            auto harm = [&]() {
              if (FO_X & 8) {
                eh = dv; oh = ze + zo;
                return;
              }
              if (DVMAX) {
                // two-coefficient model, the usual signs: the maxima come from the running maximum of dv^2 (epilogue); what
                // is left per sample is the list entry -- float64: root + two table logistics; float32: root, two fmas
                // and two logistics on the hardware transcendentals (|error| < 4e-7: v_sqrt_f32 and the float32 fma add
                // 1e-7 |nz| to the argument, the slope of the logistic is <= 1/4)
                if (LISTS != LST_NONE) nze_min = fo_vmin_neg(nze_min, dv);
                if (lst_exact(LISTS)) {
                  const double dvs = fo_sqrt_pos(dv);
                  eh = fo_logistic_neg<false>(exp_tab, fo_fma3(ke_, dvs, ce_));
                  oh = fo_logistic_neg<false>(exp_tab, fo_fma3(ko_, dvs, co_));
                } else if (LISTS == LST_F32) {
                  const float dvf = __builtin_amdgcn_sqrtf((float)dv);
                  ehf = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(kef_, dvf, cef_)));
                  ohf = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(kof_, dvf, cof_)));
                }
                return;
              }
              const bool model = LR4S || prot == 0;   // wave-uniform; otherwise harm is 1 on both sides
              const double dvs = fo_sqrt_pos(dv);      // (the ring holds dv^2)
              const double nze = LR4S ? fma(ke_, dvs, ze) : fo_fma3(ke_, dvs, ce_), nzo = LR4S ? fma(ko_, dvs, zo) : fo_fma3(ko_, dvs, co_);
              if (lst_exact(LISTS) || !model) {
                eh = model ? fo_logistic_neg<false>(exp_tab, nze) : 1.0;
                oh = model ? fo_logistic_neg<false>(exp_tab, nzo) : 1.0;
                max_eh = fo_vmax(max_eh, eh);
                max_oh = fo_vmax(max_oh, oh);
                if (LISTS == LST_F32) { ehf = 1.0f; ohf = 1.0f; }
              } else {
                nze_min = fo_vmin(nze_min, nze);   // (neither is ever NaN: no canonicalising pair of v_max around it)
                nzo_min = fo_vmin(nzo_min, nzo);
                if (LISTS == LST_F32) {   // float32 list entries: hardware exp / rcp (the maxima above stay float64)
                  ehf = fo_logistic_neg_f32(nze);
                  ohf = fo_logistic_neg_f32(nzo);
                }
              }
            };
            const bool hv = FASTROW ? true : (hvrows >> row) & 1u;  // wave-uniform: geo && t < Lh
            const bool slow_row = FASTROW ? false : (slow >> row) & 1u;
            if (hv) harm();
            float cpf = 0.0f, erf_ = 0.0f, orf = 0.0f;   // float32 lists: what they get (constants on the short branch)
            if (!slow_row) {
              er = 0.0;
              orr = 0.0;
            } else {
              if ((gmask >> row) & 1u) cp = cpw[row * TILE + lane];
              if (!lst_exact(LISTS) && hv && !(FO_X & 8) && (LR4S || DVMAX || prot == 0)) {   // the harm values themselves, where a risk may need them
                const double dvs = fo_sqrt_pos(dv);
                eh = fo_logistic_neg<false>(exp_tab, LR4S ? fma(ke_, dvs, ze) : fma(ke_, dvs, ce_));
                oh = fo_logistic_neg<false>(exp_tab, LR4S ? fma(ko_, dvs, zo) : fma(ko_, dvs, co_));
                if (LISTS == LST_F32) { ehf = (float)eh; ohf = (float)oh; }   // so that risk = harm x cp holds in the lists too
              }
              if (hv) {
                er = eh * cp;
                orr = oh * cp;
                // (a NaN probability -- an agent row without a usable covariance -- sticks in max_er, from where the
                // pair outputs below pick it up; v_max would drop it)
                if (er > max_er || er != er) max_er = er;
                if (orr > max_or) { max_or = orr; idx_or = t; }
              }
              if (cp > max_cp) { max_cp = cp; idx_cp = t; oh_at_cp = oh; }
              if (lst_is32(LISTS)) { cpf = (float)cp; erf_ = (float)er; orf = (float)orr; }
            }
            // FO_LISTS_F32_EXACT: the float64 harm values, rounded at the store.  (The probability and the risks are converted on
            // the rows that have them, above; on the others they are the float32 constants 0 -- converted behind the branches, the
            // zeros cost three v_mov_b64 and three v_cvt_f32_f64 per sample on 97 % of the rows.  Round 5: the stores moved INTO
            // the two branches, the short one with a single v_mov_b64 for its three zeros, made the allocator rotate the six
            // running maxima through copies in every iteration -- thirteen moves for two saved.)
            if (LISTS == LST_F32X) { ehf = (float)eh; ohf = (float)oh; }
            if (LISTS == LST_F64) {
              __builtin_nontemporal_store(cp, (double *)(lb0 + lo1));
              __builtin_nontemporal_store(fo_d2{eh, oh}, (fo_d2 *)(lb1 + lo2));
              __builtin_nontemporal_store(fo_d2{er, orr}, (fo_d2 *)(lb2 + lo2));
            } else if (lst_is32(LISTS)) {
              __builtin_nontemporal_store(cpf, (float *)(lb0 + lo1));
              __builtin_nontemporal_store(fo_f2{ehf, ohf}, (fo_f2 *)(lb1 + lo2));
              __builtin_nontemporal_store(fo_f2{erf_, orf}, (fo_f2 *)(lb2 + lo2));
            }
            lo1 += (unsigned)M * LE;
            lo2 += (unsigned)M * (2u * LE);
          };
          // Round 5: two rows per trip -- the values read ahead for row t+1 (relative speed, LR4S offsets) change registers
          // instead of being copied into row t's at the end of every trip (one to three v_mov_b64 per list row)
          int t = g0s;
          // Round 5: runs of rows that take the short way (97 % of the rows of the bench workload, usually the whole chunk) in
          // a loop of their own: per row two scalar shifts, two ands, two compares and two branches less -- 71 -> 53
          // instructions per row of the two-coefficient models.  Measured per list format, same flags on both sides: float32
          // arithmetic -1.9 % (0.4538 / 0.4564 -> 0.4463 / 0.4456 ms), float64 arithmetic with float32 stores +1.4 %, float64
          // lists +5 % (23 / 48 spilled VGPRs instead of 19 / 17, and those two are not bound by pass 2's issue): on for the
          // first only.
          if constexpr (FO_P2_RUNS != 0 && LISTS == LST_F32) {
          const unsigned fastrows = hvrows & ~slow;
          while (t < g1s) {
            const int row = t - gbase;
            const int run = min(__builtin_ctz(~(fastrows >> row) | 0x80000000u), g1s - t);
            if (run > 0) {
              const int te = t + run;
              for (; t < te; ++t) row_step(t, std::true_type{});
            } else {
              row_step(t, std::false_type{});
              ++t;
            }
          }
          }
          if (FO_P2_UNROLL == 2)
            for (; t + 1 < g1s; t += 2) { row_step(t, std::false_type{}); row_step(t + 1, std::false_type{}); }
          for (; t < g1s; ++t) row_step(t, std::false_type{});
          }
        };
        if (lr4s) pass2(std::integral_constant<int, HM_LR4S>{});
        else if (dvmax_mode) pass2(std::integral_constant<int, HM_DVMAX>{});
        else pass2(std::integral_constant<int, HM_GENERIC>{});
      }
      SW_STAMP(2);
    }
    // (horizon-split form, a wave whose segment lies beyond the horizon: it still takes part in the agent's pool round)
    if (POOL && SPLIT && !(seg0 < seg1)) pool_round(0, 0, 0.0, 0);
    if (dvmax_mode) {
      if (nze_min < INFINITY) {   // nze_min = -(largest squared relative speed)
        const double dvm = fo_sqrt(-nze_min);
        max_eh = fo_vmax(max_eh, fo_logistic_neg<false>(exp_tab, fma(hk[0], dvm, hk[2])));
        max_oh = fo_vmax(max_oh, fo_logistic_neg<false>(exp_tab, fma(hk[1], dvm, hk[3])));
      }
    } else
    if (!lst_exact(LISTS) && nze_min < INFINITY) {   // (a wave whose samples carry no harm keeps -inf, as the lists path does)
      max_eh = fo_vmax(max_eh, fo_logistic_neg<false>(exp_tab, nze_min));
      max_oh = fo_vmax(max_oh, fo_logistic_neg<false>(exp_tab, nzo_min));
    }

    if (SPLIT) {
      // ---------------------------------------------------------------- fold the four time segments, in time order
      // (minimum with the earliest t for the DCE, first maximum for the risks and probabilities): each wave parks its
      // values in its own LDS rows, wave 0 folds them and goes on to the outputs alone
      if (wave > 0) {
        double *sp = cpw + lane;
        sp[0 * TILE] = dce; sp[1 * TILE] = (double)tdce; sp[2 * TILE] = max_er; sp[3 * TILE] = max_or;
        sp[4 * TILE] = (double)idx_or; sp[5 * TILE] = max_eh; sp[6 * TILE] = max_oh; sp[7 * TILE] = max_cp;
        sp[8 * TILE] = (double)idx_cp; sp[9 * TILE] = oh_at_cp;
      }
      __syncthreads();
      if (wave == 0)
      for (int w = 1; w < QWAVES; ++w) {
        const double *sp = cpbuf_all + w * (WROWS * TILE) + lane;
        const double d_w = sp[0 * TILE];
        const int t_w = (int)sp[1 * TILE];
        if (d_w < dce || (d_w == dce && t_w < tdce)) { dce = d_w; tdce = t_w; }
        { const double e_ = sp[2 * TILE]; if (e_ > max_er || e_ != e_) max_er = e_; }
        if (sp[3 * TILE] > max_or) { max_or = sp[3 * TILE]; idx_or = (int)sp[4 * TILE]; }
        max_eh = fmax(max_eh, sp[5 * TILE]);
        max_oh = fmax(max_oh, sp[6 * TILE]);
        if (sp[7 * TILE] > max_cp) { max_cp = sp[7 * TILE]; idx_cp = (int)sp[8 * TILE]; oh_at_cp = sp[9 * TILE]; }
      }
      // more agents to come: the other waves' rows are theirs again once wave 0 has read them
      if (kk + 1 < apw_ && k + 1 < A) __syncthreads();
      if (wave > 0) continue;
    }
    SW_STAMP(3);
    // ------------------------------------------------------------------ per-pair scalars
    const double dce_m = (dce < INFINITY) ? fo_div1000(dce) : dce;                          // np.round(d, 3)
    const double ttce = fo_round3_fast((double)tdce * a.dt);                                // ttce.py:39
    const double ttc = (dce == 0.0) ? ttce : INFINITY;                                      // ttc.py:43-46
    if (a.be_mask) a.be_mask[(size_t)k * a.Mp + m] = (do_ttc && ttc < INFINITY && ttc > 0.0) ? 1 : 0;  // be.py:49-50
    const bool hr_valid = do_hr && Lh > 0;
    const double hwc = (max_cp > 0.01) ? oh_at_cp : 0.0;                                    // hr.py:81-84
    if (PAIR) {
      const size_t ps_ = (size_t)A * M;
      double *pf = a.pair_f + (size_t)k * M + m;
      pf[FO_PF_DCE * ps_] = do_dce ? dce_m : NAN;
      pf[FO_PF_TTC * ps_] = do_ttc ? ttc : NAN;
      pf[FO_PF_TTCE * ps_] = do_ttce ? ttce : NAN;
#if FO_EPI_SEL
      if (hr_valid) {   // (wave-uniform)
        // a collision probability of this pair was NaN (see pass 2): NaN where the probability enters -- the high word alone
        const unsigned long long bad = __builtin_amdgcn_fcmp(max_er, max_er, 8 /* uno */);
        pf[FO_PF_MAX_EGO_RISK * ps_] = max_er;
        pf[FO_PF_MAX_OBST_RISK * ps_] = fo_sel_hi(bad, max_or, NAN);
        pf[FO_PF_HARM_WITH_CP * ps_] = fo_sel_hi(bad, hwc, NAN);
        pf[FO_PF_MAX_EGO_HARM * ps_] = max_eh;
        pf[FO_PF_MAX_OBST_HARM * ps_] = max_oh;
        pf[FO_PF_MAX_CP * ps_] = fo_sel_hi(bad, max_cp, NAN);
      } else {
        pf[FO_PF_MAX_EGO_RISK * ps_] = NAN; pf[FO_PF_MAX_OBST_RISK * ps_] = NAN; pf[FO_PF_HARM_WITH_CP * ps_] = NAN;
        pf[FO_PF_MAX_EGO_HARM * ps_] = NAN; pf[FO_PF_MAX_OBST_HARM * ps_] = NAN; pf[FO_PF_MAX_CP * ps_] = NAN;
      }
#else
      const bool cp_ok = hr_valid && max_er == max_er;   // false: a collision probability of this pair was NaN (see pass 2)
      pf[FO_PF_MAX_EGO_RISK * ps_] = hr_valid ? max_er : NAN;
      pf[FO_PF_MAX_OBST_RISK * ps_] = cp_ok ? max_or : NAN;
      pf[FO_PF_HARM_WITH_CP * ps_] = cp_ok ? hwc : NAN;
      pf[FO_PF_MAX_EGO_HARM * ps_] = hr_valid ? max_eh : NAN;
      pf[FO_PF_MAX_OBST_HARM * ps_] = hr_valid ? max_oh : NAN;
      pf[FO_PF_MAX_CP * ps_] = cp_ok ? max_cp : NAN;
#endif
      pf[FO_PF_BE_DECEL * ps_] = NAN;
      pf[FO_PF_BE_BTN * ps_] = NAN;
      pf[FO_PF_SPARE * ps_] = NAN;
      int32_t *pi = a.pair_i + (size_t)k * M + m;
      pi[FO_PI_TIME_DCE * ps_] = do_dce ? tdce : 0;
      pi[FO_PI_RISK_INDEX * ps_] = hr_valid ? idx_or : 0;
      pi[FO_PI_CP_ARGMAX * ps_] = hr_valid ? idx_cp : 0;
      pi[FO_PI_HR_VALID * ps_] = hr_valid ? 1 : 0;
    }
#if FO_EPI_SEL
    // Round 5: the running extrema over the wave's agents as v_min / v_max plus ONE select of the index on a scalar-pair mask
    // (a compare followed by three v_cndmask on vcc holds the SIMD for ~25 cycles beyond the instructions' own issue --
    // tools/microbench/valu_rate.hip, "v_cmp_f64 + v_cndmask"); fmax() with its canonicalising v_max x, x in front
    // replaced by the bare instruction (operands are results of arithmetic, never signalling NaNs).
    if (do_dce) {
      const unsigned long long lt = __builtin_amdgcn_fcmp(dce_m, w_min_dce, 4 /* olt */);
      w_min_dce = fo_vmin(w_min_dce, dce_m);
      w_arg_dce = fo_sel_b32(lt, w_arg_dce, k);
      if (dce_m < a.thr_dce) w_dce_flag = true;
      if (do_ttc) {
        const unsigned long long z = __builtin_amdgcn_ballot_w64(dce == 0.0 && tdce < w_min_tttc);
        w_min_tttc = fo_sel_b32(z, w_min_tttc, tdce);
        w_arg_ttc = fo_sel_b32(z, w_arg_ttc, k);
      }
      if (do_ttce) w_min_tttce = min(w_min_tttce, tdce);
    }
    if (hr_valid) {
      w_max_er = fo_vmax(w_max_er, max_er);
      const unsigned long long gt = __builtin_amdgcn_fcmp(max_or, w_max_or, 2 /* ogt */);
      w_max_or = fo_vmax(w_max_or, max_or);
      w_arg_or = fo_sel_b32(gt, w_arg_or, k);
      w_max_eh = fo_vmax(w_max_eh, max_eh);
      w_max_oh = fo_vmax(w_max_oh, max_oh);
      w_max_cp = fo_vmax(w_max_cp, max_cp);
      w_max_hwc = fo_vmax(w_max_hwc, hwc);
    }
#else
    if (do_dce) {
      if (dce_m < w_min_dce) { w_min_dce = dce_m; w_arg_dce = k; }
      if (dce_m < a.thr_dce) w_dce_flag = true;
      if (do_ttc && dce == 0.0 && tdce < w_min_tttc) { w_min_tttc = tdce; w_arg_ttc = k; }
      if (do_ttce) w_min_tttce = min(w_min_tttce, tdce);
    }
    if (hr_valid) {
      w_max_er = fmax(w_max_er, max_er);
      if (max_or > w_max_or) { w_max_or = max_or; w_arg_or = k; }
      w_max_eh = fmax(w_max_eh, max_eh);
      w_max_oh = fmax(w_max_oh, max_oh);
      w_max_cp = fmax(w_max_cp, max_cp);
      w_max_hwc = fmax(w_max_hwc, hwc);
    }
#endif
  }

  // ---------------- combine the waves of the workgroup (ascending agent order); scratch aliases the cp buffers
  // (horizon-split form: wave 0 has folded every agent's segments and holds the workgroup's values -- no exchange, the other
  // waves are done)
  if (SPLIT && wave > 0) return;
  if (!SPLIT) __syncthreads();
  double *red = cpbuf_all;
  double w_min_ttc = w_min_tttc == 0x7fffffff ? INFINITY : fo_round3_fast((double)w_min_tttc * a.dt);
  double w_min_ttce = w_min_tttce == 0x7fffffff ? INFINITY : fo_round3_fast((double)w_min_tttce * a.dt);
  if (!SPLIT && wave > 0) {
    double *rp = red + (size_t)(wave - 1) * NPS * TILE + lane;
    rp[PS_MIN_DCE * TILE] = w_min_dce; rp[PS_ARG_DCE * TILE] = (double)w_arg_dce; rp[PS_MIN_TTC * TILE] = w_min_ttc;
    rp[PS_ARG_TTC * TILE] = (double)w_arg_ttc; rp[PS_MIN_TTCE * TILE] = w_min_ttce; rp[PS_MAX_ER * TILE] = w_max_er;
    rp[PS_MAX_OR * TILE] = w_max_or; rp[PS_ARG_OR * TILE] = (double)w_arg_or; rp[PS_MAX_EH * TILE] = w_max_eh;
    rp[PS_MAX_OH * TILE] = w_max_oh; rp[PS_MAX_CP * TILE] = w_max_cp; rp[PS_MAX_HWC * TILE] = w_max_hwc;
    rp[PS_DCE_FLAG * TILE] = w_dce_flag ? 1.0 : 0.0; rp[PS_MAX_BTN * TILE] = 0.0;
  }
  if (!SPLIT) __syncthreads();
  if (wave == 0) {
    if (!SPLIT)
    for (int w = 0; w < QWAVES - 1; ++w) {
      const double *rp = red + (size_t)w * NPS * TILE + lane;
      // (ties keep the smaller agent index: with FO_DYN the waves' agents interleave, without it wave order = agent order
      // and the index test never fires)
      const int ad_ = (int)rp[PS_ARG_DCE * TILE], at_ = (int)rp[PS_ARG_TTC * TILE], ao_ = (int)rp[PS_ARG_OR * TILE];
      if (rp[PS_MIN_DCE * TILE] < w_min_dce || (DYN && rp[PS_MIN_DCE * TILE] == w_min_dce && ad_ >= 0 && (w_arg_dce < 0 || ad_ < w_arg_dce))) { w_min_dce = rp[PS_MIN_DCE * TILE]; w_arg_dce = ad_; }
      if (rp[PS_MIN_TTC * TILE] < w_min_ttc || (DYN && rp[PS_MIN_TTC * TILE] == w_min_ttc && at_ >= 0 && (w_arg_ttc < 0 || at_ < w_arg_ttc))) { w_min_ttc = rp[PS_MIN_TTC * TILE]; w_arg_ttc = at_; }
      w_min_ttce = fmin(w_min_ttce, rp[PS_MIN_TTCE * TILE]);
      w_max_er = fmax(w_max_er, rp[PS_MAX_ER * TILE]);
      if (rp[PS_MAX_OR * TILE] > w_max_or || (DYN && rp[PS_MAX_OR * TILE] == w_max_or && ao_ >= 0 && (w_arg_or < 0 || ao_ < w_arg_or))) { w_max_or = rp[PS_MAX_OR * TILE]; w_arg_or = ao_; }
      w_max_eh = fmax(w_max_eh, rp[PS_MAX_EH * TILE]);
      w_max_oh = fmax(w_max_oh, rp[PS_MAX_OH * TILE]);
      w_max_cp = fmax(w_max_cp, rp[PS_MAX_CP * TILE]);
      w_max_hwc = fmax(w_max_hwc, rp[PS_MAX_HWC * TILE]);
      w_dce_flag = w_dce_flag || rp[PS_DCE_FLAG * TILE] > 0.0;
    }
    const size_t PM = (size_t)a.Mp;
    double *pp = a.partial + (size_t)chunk * NPS * PM + (size_t)tile * TILE + lane;
    pp[PS_MIN_DCE * PM] = w_min_dce; pp[PS_ARG_DCE * PM] = (double)w_arg_dce; pp[PS_MIN_TTC * PM] = w_min_ttc;
    pp[PS_ARG_TTC * PM] = (double)w_arg_ttc; pp[PS_MIN_TTCE * PM] = w_min_ttce; pp[PS_MAX_ER * PM] = w_max_er;
    pp[PS_MAX_OR * PM] = w_max_or; pp[PS_ARG_OR * PM] = (double)w_arg_or; pp[PS_MAX_EH * PM] = w_max_eh;
    pp[PS_MAX_OH * PM] = w_max_oh; pp[PS_MAX_CP * PM] = w_max_cp; pp[PS_MAX_HWC * PM] = w_max_hwc;
    pp[PS_DCE_FLAG * PM] = w_dce_flag ? 1.0 : 0.0; pp[PS_MAX_BTN * PM] = 0.0;
  }
}

// Chunk length and occupancy by output mode: with the per-sample lists the hot loops need 164 VGPRs (three waves per
// SIMD, chunks of eight); without them -- and not in the horizon-split form, whose four waves must cover T <= 32 with
// one chunk each -- chunks of four fit 128 VGPRs and 37 KB of LDS: four waves per SIMD (measured -5 % on those modes).
#ifndef FO_WIDE_LISTS
#define FO_WIDE_LISTS 0   // 1: tuning builds -- the full-output instantiation in the four-wave shape as well
#endif
#ifndef FO_WIDE_NONE
#define FO_WIDE_NONE 0    // 1: the instantiations without lists in the four-wave shape (round 2 and early round 3: faster then; since the scalar diet of round 3 the three-wave shape wins, reduced outputs 0.417 against 0.438 ms)
#endif
#ifndef FO_WIDE_F32
#define FO_WIDE_F32 0     // 1: the float32-list instantiation in the four-wave shape
#endif
template <int LISTS, bool SPLIT>
struct SweepShape {
  static constexpr bool wide = ((LISTS == LST_NONE && FO_WIDE_NONE) || FO_WIDE_LISTS || (LISTS == LST_F32 && FO_WIDE_F32)) && !SPLIT &&
                               FO_MINW == 3 && FO_TC == 8;   // tuning builds override both macros
  static constexpr int tc = wide ? 4 : FO_TC, minw = wide ? 4 : FO_MINW;
};
template <bool PAIR, int LISTS, bool ALLM, bool SPLIT = false>
__global__ __launch_bounds__(TILE *QWAVES)
__attribute__((amdgpu_waves_per_eu(SweepShape<LISTS, SPLIT>::minw, SweepShape<LISTS, SPLIT>::minw)))
void fo_sweep_queue_kernel(const SweepArgs a) {
  constexpr int TCK = SweepShape<LISTS, SPLIT>::tc, WROWS = TCK + TCK + 1;
  __shared__ double2 erf_tab[ERF_N];
  __shared__ double exp_tab[EXP_N];
  __shared__ double zc_tab[4];                      // LR4S logistic offsets by impact class: front, side, rear
  __shared__ double hk_all[QWAVES * 4];             // per wave: the current agent's logistic slopes and offsets
  constexpr int BUFROWS = QWAVES * WROWS > (QWAVES - 1) * NPS ? QWAVES * WROWS : (QWAVES - 1) * NPS;
  __shared__ double cpbuf_all[BUFROWS * TILE];  // per wave: TC rows of collision probabilities, DVR rows of
                                                       // relative speeds; also the cross-wave reduction scratch
  __shared__ unsigned short queue_all[QWAVES * (FO_POOL && SPLIT ? TILE * TCK : QCAP)];   // per wave: in-gate samples (lane | row << 6)
  __shared__ int pool_i[3 * QWAVES];       // FO_POOL, per wave: queue length, agent, gate sample of buffer row 0
  __shared__ double pool_hd[QWAVES];       //          half the agent's inflated length
  __shared__ int next_agent;   // FO_DYN: agents of the chunk handed out so far
  if (threadIdx.x == 0) next_agent = 0;
  {
    // The two tables into LDS.  All of a thread's loads are issued before the first store (written as a loop the copy
    // compiles to five dependent round trips: load, wait, store, ...).  (A build WITHOUT the copy is no measure of its
    // cost: the compiler then knows the tables are never written and deletes the code that reads them.)
    static_assert(TILE * QWAVES == 256 && EXP_N == 256 && ERF_N > 768 && ERF_N <= 1024, "table copy written for 256 threads");
    const int tt = threadIdx.x;
    const double2 v0 = a.erf_tab[tt], v1 = a.erf_tab[tt + 256], v2 = a.erf_tab[tt + 512];
    const double2 v3 = a.erf_tab[min(tt + 768, ERF_N - 1)];
    const double x0 = a.exp_tab[tt];
    // (g / 128: fo_erf_fast128 measures the offset from a node in table steps)
    erf_tab[tt] = make_double2(v0.x, v0.y * 0x1p-7); erf_tab[tt + 256] = make_double2(v1.x, v1.y * 0x1p-7);
    erf_tab[tt + 512] = make_double2(v2.x, v2.y * 0x1p-7);
    if (tt + 768 < ERF_N) erf_tab[tt + 768] = make_double2(v3.x, v3.y * 0x1p-7);
    exp_tab[tt] = x0;
  }
  if (threadIdx.x < 4)
    zc_tab[threadIdx.x] = -a.hc.lr4s_const - (threadIdx.x == 0 ? 0.0 : threadIdx.x == 1 ? a.hc.lr4s_side : a.hc.lr4s_rear);
  // (FO_X & 128: register / timing experiments without the second body -- WRONG results for correlated covariances)
  const bool corr = !(FO_X & 128) && a.status[1] == a.gen;   // scalar load; written by fo_prep_agents_kernel on this stream
#if FO_TRACE
  if (a.trace && threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    a.trace[4 * (size_t)blockIdx.x + 0] = wall_clock64();
    a.trace[4 * (size_t)blockIdx.x + 2] = (long long)hw | ((long long)xcc << 32);
  }
#endif
  __syncthreads();
#if FO_TRACE
  if (a.trace && threadIdx.x == 0) a.trace[4 * (size_t)blockIdx.x + 3] = wall_clock64();   // tables in LDS
#endif
  if (__builtin_expect(!corr, 1))
    fo_sweep_queue_body<PAIR, LISTS, ALLM, SPLIT, false, TCK>(a, erf_tab, exp_tab, zc_tab, hk_all, cpbuf_all, queue_all, &next_agent, pool_i, pool_hd);
  else
    fo_sweep_queue_body<PAIR, LISTS, ALLM, SPLIT, true, TCK>(a, erf_tab, exp_tab, zc_tab, hk_all, cpbuf_all, queue_all, &next_agent, pool_i, pool_hd);
#if FO_TRACE
  if (a.trace && threadIdx.x == 0) a.trace[4 * (size_t)blockIdx.x + 1] = wall_clock64();
#endif
}


// ================================================================================================ BE (optional)
// Brake evaluation (metrics/be.py:31-193), active only with FO_M_BE: for every pair that collides at ttc > 0 the minimum
// constant deceleration found by the reference's bisection (<= 10 iterations on [round(|min(a_min, 0)|, 2), 5] m/s^2,
// stop below 0.1) and the brake threat number decel / a_max.  For one candidate deceleration the ego keeps its path,
// the speed profile becomes [v0, max(v1 - decel j dt, 0) ...], poses are re-sampled by linear interpolation over the
// travelled chord length (scipy interp1d semantics: searchsorted-left segment, clipped), rectangles are tested for
// intersection (SAT, touching counts) at every step the agent exists.  Where the re-sampled arc length exceeds the
// path length the reference raises ValueError; here it is clamped to the end of the path.
__global__ void fo_be_prep_kernel(int M, int Mp, int T, const double *__restrict__ x, const double *__restrict__ y,
                                  const double *__restrict__ acc, double *__restrict__ dist, double *__restrict__ mina) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= Mp) return;
  const int ms = min(m, M - 1);
  const double *xs = x + (size_t)ms * T, *ys = y + (size_t)ms * T, *as = acc + (size_t)ms * T;
  double d = 0.0, mn = 0.0;
  dist[m] = 0.0;
  for (int i = 0; i < T; ++i) {
    if (i > 0) {
      const double dx = xs[i] - xs[i - 1], dy = ys[i] - ys[i - 1];
      d += sqrt(dx * dx + dy * dy);
      dist[(size_t)i * Mp + m] = d;
    }
    mn = fmin(mn, as[i]);
  }
  mina[m] = mn;
}

__global__ __launch_bounds__(256) void fo_be_kernel(int M, int Mp, int T, int A, int Ta, const double *__restrict__ traj,
                                                    const double *__restrict__ dist, const double *__restrict__ mina,
                                                    const double *__restrict__ atab, const double *__restrict__ acst,
                                                    const int32_t *__restrict__ aint,
                                                    const signed char *__restrict__ be_mask, double hlA, double hwA,
                                                    double wb, double a_max, double dt, double *__restrict__ be_btn,
                                                    double *__restrict__ pair_f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x, k = blockIdx.y * 4 + wave;
  if (k >= A) return;
  const int m = tile * TILE + lane;
  const int L = aint[2 * k + 1];
  const double hlB = acst[(size_t)k * NAC + 0], hwB = acst[(size_t)k * NAC + 1];
  const double *tjl = traj + (size_t)tile * T * NEF * TILE + 2 * lane;  // this lane's pairs of the tile
  const double *G = atab + (size_t)k * Ta * NAF;
  const bool active = m < M && L > 0 && T >= 2 && be_mask[(size_t)k * Mp + m];
  double decel = 0.0, btn = 0.0;
  if (active) {
    const double v0 = tjl[EF(5)], v1 = tjl[(size_t)NEF * TILE + EF(5)];
    const double dend = dist[(size_t)(T - 1) * Mp + m];
    double min_d = __builtin_rint(fabs(mina[m]) * 100.0) / 100.0, max_d = 5.0;  // np.round(abs(min(min(a), 0)), 2)
    for (int it = 0; it < 10; ++it) {
      const double cur = (min_d + max_d) / 2.0;
      decel = cur;
      bool hit = false;
      double s = 0.0;
      int j = 0;
      for (int i = 0; i < T && !hit; ++i) {
        if (i < L) {
          const double sc = fmin(s, dend);
          while (j < T && dist[(size_t)j * Mp + m] < sc) ++j;  // searchsorted (left); s never decreases
          const int idx = min(max(j, 1), T - 1);
          const double xlo = dist[(size_t)(idx - 1) * Mp + m], xhi = dist[(size_t)idx * Mp + m];
          const double *r0 = tjl + (size_t)(idx - 1) * NEF * TILE, *r1 = tjl + (size_t)idx * NEF * TILE;
          double xn = r0[EF(0)], yn = r0[EF(1)], tn = r0[EF(4)];
          if (xhi != xlo) {
            const double w = sc - xlo, inv = xhi - xlo;
            xn = (r1[EF(0)] - xn) / inv * w + xn;
            yn = (r1[EF(1)] - yn) / inv * w + yn;
            tn = (r1[EF(4)] - tn) / inv * w + tn;
          }
          double es, ec;
          sincos(tn, &es, &ec);
          const double *g = G + (size_t)i * NAF;
          const double px = g[0], py = g[1], pc = g[2], ps = g[3];
          const double cr = pc * ec + ps * es, sr = ps * ec - pc * es;
          const double dx = px - (xn + wb * ec), dy = py - (yn + wb * es);
          const double ax = ec * dx + es * dy, ay = ec * dy - es * dx;
          const double bx = -(pc * dx + ps * dy), by = -(pc * dy - ps * dx);
          const double s1 = fabs(ax) - (hlA + fabs(hlB * cr) + fabs(hwB * sr)), s2 = fabs(ay) - (hwA + fabs(hlB * sr) + fabs(hwB * cr));
          const double s3 = fabs(bx) - (hlB + fabs(hlA * cr) + fabs(hwA * sr)), s4 = fabs(by) - (hwB + fabs(hlA * sr) + fabs(hwA * cr));
          if (!(fmax(fmax(s1, s2), fmax(s3, s4)) > 0.0)) hit = true;  // shapely intersects
        }
        const double vn = (i == 0) ? v0 : fmax(v1 - cur * ((double)(i - 1) * dt), 0.0);
        s += vn * dt;
      }
      if (!hit) max_d = cur; else min_d = cur;
      if (max_d - min_d < 0.1) break;
    }
    btn = decel / a_max;
  }
  if (m < Mp) be_btn[(size_t)k * Mp + m] = btn;
  if (pair_f && m < M && L > 0) {
    const size_t ps_ = (size_t)A * M;
    pair_f[FO_PF_BE_DECEL * ps_ + (size_t)k * M + m] = decel;
    pair_f[FO_PF_BE_BTN * ps_ + (size_t)k * M + m] = btn;
  }
}

// fold the per-chunk partials into the cost vector + safety flag (metric.py:50-100, hr.py:101-114, wttc.py:32-42)
// 64 trajectories per workgroup, eight waves: wave w folds its eighth of the chunk rows (in chunk order), the eight
// partial results meet in LDS and wave 0 folds them in the same order -- ties keep the earliest chunk, exactly like one
// sequential pass, with an eighth of the dependent-load chain.
constexpr int RED_WAVES = 8;
__global__ __launch_bounds__(64 * RED_WAVES) void fo_reduce_kernel(int M, int Mp, int A, int n_chunks,
                                                                   const double *__restrict__ partial,
                                                                   fo_thresholds_t thr, uint32_t mask,
                                                                   const double *__restrict__ be_btn,
                                                                   double *__restrict__ cost,
                                                                   uint8_t *__restrict__ safe,
                                                                   const int *__restrict__ status, int gen) {
  __shared__ double sh[RED_WAVES][NPS + 1][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x * 64 + lane;
  const bool live = m < M;
  double max_btn = 0.0;
  double min_dce = INFINITY, arg_dce = -1, min_ttc = INFINITY, arg_ttc = -1, min_ttce = INFINITY;
  double max_er = 0, max_or = 0, arg_or = -1, max_eh = 0, max_oh = 0, max_cp = 0, max_hwc = 0, flag = 0;
  if (live) {
    if (be_btn)
      for (int k = wave; k < A; k += RED_WAVES) max_btn = fmax(max_btn, be_btn[(size_t)k * Mp + m]);
    const int per = (n_chunks + RED_WAVES - 1) / RED_WAVES;
    const int c0 = wave * per, c1 = c0 + per < n_chunks ? c0 + per : n_chunks;
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
      const double *p = partial + (size_t)c * NPS * Mp + m;
      if (p[PS_MIN_DCE * (size_t)Mp] < min_dce) { min_dce = p[PS_MIN_DCE * (size_t)Mp]; arg_dce = p[PS_ARG_DCE * (size_t)Mp]; }
      if (p[PS_MIN_TTC * (size_t)Mp] < min_ttc) { min_ttc = p[PS_MIN_TTC * (size_t)Mp]; arg_ttc = p[PS_ARG_TTC * (size_t)Mp]; }
      min_ttce = fmin(min_ttce, p[PS_MIN_TTCE * (size_t)Mp]);
      max_er = fmax(max_er, p[PS_MAX_ER * (size_t)Mp]);
      if (p[PS_MAX_OR * (size_t)Mp] > max_or) { max_or = p[PS_MAX_OR * (size_t)Mp]; arg_or = p[PS_ARG_OR * (size_t)Mp]; }
      max_eh = fmax(max_eh, p[PS_MAX_EH * (size_t)Mp]);
      max_oh = fmax(max_oh, p[PS_MAX_OH * (size_t)Mp]);
      max_cp = fmax(max_cp, p[PS_MAX_CP * (size_t)Mp]);
      max_hwc = fmax(max_hwc, p[PS_MAX_HWC * (size_t)Mp]);
      flag = fmax(flag, p[PS_DCE_FLAG * (size_t)Mp]);
    }
  }
  double *q = &sh[wave][0][lane];
  q[PS_MIN_DCE * 64] = min_dce; q[PS_ARG_DCE * 64] = arg_dce; q[PS_MIN_TTC * 64] = min_ttc; q[PS_ARG_TTC * 64] = arg_ttc;
  q[PS_MIN_TTCE * 64] = min_ttce; q[PS_MAX_ER * 64] = max_er; q[PS_MAX_OR * 64] = max_or; q[PS_ARG_OR * 64] = arg_or;
  q[PS_MAX_EH * 64] = max_eh; q[PS_MAX_OH * 64] = max_oh; q[PS_MAX_CP * 64] = max_cp; q[PS_MAX_HWC * 64] = max_hwc;
  q[PS_DCE_FLAG * 64] = flag; q[NPS * 64] = max_btn;
  __syncthreads();
  if (wave != 0 || !live) return;
  for (int w = 1; w < RED_WAVES; ++w) {
    const double *p = &sh[w][0][lane];
    if (p[PS_MIN_DCE * 64] < min_dce) { min_dce = p[PS_MIN_DCE * 64]; arg_dce = p[PS_ARG_DCE * 64]; }
    if (p[PS_MIN_TTC * 64] < min_ttc) { min_ttc = p[PS_MIN_TTC * 64]; arg_ttc = p[PS_ARG_TTC * 64]; }
    min_ttce = fmin(min_ttce, p[PS_MIN_TTCE * 64]);
    max_er = fmax(max_er, p[PS_MAX_ER * 64]);
    if (p[PS_MAX_OR * 64] > max_or) { max_or = p[PS_MAX_OR * 64]; arg_or = p[PS_ARG_OR * 64]; }
    max_eh = fmax(max_eh, p[PS_MAX_EH * 64]);
    max_oh = fmax(max_oh, p[PS_MAX_OH * 64]);
    max_cp = fmax(max_cp, p[PS_MAX_CP * 64]);
    max_hwc = fmax(max_hwc, p[PS_MAX_HWC * 64]);
    flag = fmax(flag, p[PS_DCE_FLAG * 64]);
    max_btn = fmax(max_btn, p[NPS * 64]);
  }
  bool ok = true;
  if (A > 0) {  // no agents -> ({}, True)  (metric.py:44-45)
    if ((mask & FO_M_HR) && max_hwc > thr.harm) ok = false;  // NaN thresholds compare false = disabled
    if ((mask & FO_M_HR) && max_or > thr.risk) ok = false;
    if ((mask & FO_M_HR) && max_cp > thr.cp) ok = false;
    if ((mask & FO_M_TTC) && min_ttc < thr.ttc) ok = false;
    if ((mask & FO_M_DCE) && flag > 0.0) ok = false;
    if ((mask & FO_M_BE) && max_btn > thr.be) ok = false;  // metric.py:54-61
    // The current agent set holds an off-diagonal covariance (fo_prep_agents_kernel poisoned those rows and tagged
    // the status word with this generation): fmax() above drops the NaNs, so say it here -- nothing that depends
    // on a collision probability may read as "safe", whether or not the caller runs fo_sweep_check.
    if ((mask & (FO_M_CP | FO_M_HR)) && gen > 0 && *status == gen) {
      ok = false;
      max_cp = max_er = max_or = max_hwc = NAN;
    }
  }
  double *c = cost + (size_t)m * FO_NC;
  c[FO_C_WTTC] = min_ttc; c[FO_C_MIN_DCE] = min_dce; c[FO_C_MAX_EGO_RISK] = max_er; c[FO_C_MAX_OBST_RISK] = max_or;
  c[FO_C_MAX_EGO_HARM] = max_eh; c[FO_C_MAX_OBST_HARM] = max_oh; c[FO_C_MAX_CP] = max_cp;
  c[FO_C_HARM_WITH_CP] = max_hwc; c[FO_C_MIN_TTCE] = min_ttce; c[FO_C_ARGMIN_DCE] = arg_dce;
  c[FO_C_ARGMIN_TTC] = arg_ttc; c[FO_C_ARGMAX_RISK] = arg_or; c[FO_C_SAFE] = ok ? 1.0 : 0.0; c[FO_C_MAX_BTN] = max_btn;
  c[FO_C_RES0] = 0.0; c[FO_C_RES1] = 0.0;
  safe[m] = ok ? 1 : 0;
}

uint32_t required_metrics(uint32_t m) {  // metric.py:125-147
  if (m & FO_M_WTTC) m |= FO_M_TTC;
  if (m & FO_M_BE) m |= FO_M_TTC;  // be.py:39 reads results['ttc'] (the reference raises KeyError without it)
  if (m & (FO_M_TTC | FO_M_TTCE | FO_M_BE)) m |= FO_M_DCE;
  if (m & FO_M_HR) m |= FO_M_CP;
  return m;
}

inline int round_up(int v, int q) { return (v + q - 1) / q * q; }
constexpr int AGENT_PAD_ROWS = 256;   // spare rows behind the agent table (unclamped row addresses of the queue kernel)

// one instantiation of the queue kernel per output mode: cost vectors only / + pair scalars / + float64 or float32 lists
// false: no instantiation for this combination (fo_sweep_run sends those to the generic kernel BEFORE it plans the grid; a
// change of its conditions must not end in a launch that silently writes nothing)
template <bool ALLM, bool SPLIT>
bool launch_queue(int lst, bool pair, dim3 g, dim3 b, hipStream_t s, const SweepArgs &a) {
  if (lst == LST_F64) hipLaunchKernelGGL((fo_sweep_queue_kernel<true, LST_F64, ALLM, SPLIT>), g, b, 0, s, a);
  else if (lst == LST_F32) hipLaunchKernelGGL((fo_sweep_queue_kernel<true, LST_F32, ALLM, SPLIT>), g, b, 0, s, a);
  // (round 6: the headline format has every instantiation the other list formats have -- horizon-split for small batches,
  // metric subsets -- instead of dropping to the generic kernel at the reference's own batch sizes, metric.py:125-147)
  else if (lst == LST_F32X) hipLaunchKernelGGL((fo_sweep_queue_kernel<true, LST_F32X, ALLM, SPLIT>), g, b, 0, s, a);
  else if (pair) hipLaunchKernelGGL((fo_sweep_queue_kernel<true, LST_NONE, ALLM, SPLIT>), g, b, 0, s, a);
  else hipLaunchKernelGGL((fo_sweep_queue_kernel<false, LST_NONE, ALLM, SPLIT>), g, b, 0, s, a);
  return true;
}

// agents per wave in the first phase of the (tapered) grid: long workgroups keep the per-workgroup start-up (table fill,
// cross-wave fold) small, the taper takes care of the end of the launch.  Measured at steady clocks on 10 000 x 256 with
// float32 lists and the default taper: 2 -> 0.585 ms, 3 -> 0.556, 4 -> 0.546, 6 -> 0.552, 8 -> 0.550 (bench.py re-checks
// 1 / 2 / 4 / 8 per batch shape at set-up).
int pick_apw(int n_tiles, int A, int wpb) {
  int apw = 8;
  while (apw > 1 && (long)n_tiles * ((A + wpb * apw - 1) / (wpb * apw)) * wpb < 8192) apw >>= 1;
  return apw;
}

}  // namespace

extern "C" {

// called from fo_create (fo_api.hip)
int fo_sweep_init_(fo_ctx *ctx) {
  FO_HIP_TRY(ctx, hipMalloc((void **)&ctx->d_erf_tab, sizeof(double2) * ERF_N));
  hipLaunchKernelGGL(fo_erf_table_kernel, dim3((ERF_N + 255) / 256), dim3(256), 0, 0, (double2 *)ctx->d_erf_tab);
  FO_HIP_TRY(ctx, hipMalloc((void **)&ctx->d_exp_tab, sizeof(double) * EXP_N));
  hipLaunchKernelGGL(fo_exp_table_kernel, dim3(1), dim3(EXP_N), 0, 0, (double *)ctx->d_exp_tab);
  FO_HIP_TRY(ctx, hipGetLastError());
  {  // Gauss-Legendre rules of the correlation integral: Newton on the Legendre recurrence; stored as
     // t = (x + 1)/2 and w/(4 pi)  (Int_0^asr f = asr/2 Sum w f(asr t), times the 1/(2 pi) of the formula)
    static double gl[2 * GL_TOTAL];
    double *o = gl;
    for (int r = 0; r < GL_NR; ++r) {
      const int n = gl_nodes(r);
      for (int i = 0; i < n; ++i) {
        double x = -cos(M_PI * (i + 0.75) / (n + 0.5)), dp = 1.0;
        for (int it = 0; it < 100; ++it) {
          double p0 = 1.0, p1 = x;
          for (int k = 2; k <= n; ++k) { const double p2 = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k; p0 = p1; p1 = p2; }
          dp = n * (x * p1 - p0) / (x * x - 1.0);
          const double dx = p1 / dp;
          x -= dx;
          if (fabs(dx) < 1e-16) break;
        }
        *o++ = 0.5 * (x + 1.0);
        *o++ = 2.0 / ((1.0 - x * x) * dp * dp) / (4.0 * M_PI);
      }
    }
    FO_HIP_TRY(ctx, hipMalloc((void **)&ctx->d_gl_tab, sizeof gl));
    FO_HIP_TRY(ctx, hipMemcpy(ctx->d_gl_tab, gl, sizeof gl, hipMemcpyHostToDevice));
  }
  FO_HIP_TRY(ctx, hipDeviceSynchronize());
  return FO_OK;
}

int fo_sweep_configure(fo_ctx *ctx, const fo_vehicle_t *veh, const fo_harm_coeff_t *hc, const fo_thresholds_t *thr,
                       uint32_t metric_mask, double dt) {
  if (!ctx || !veh || !hc || !thr) return fo_fail(ctx, FO_E_ARG, "fo_sweep_configure: null argument");
  if (!(veh->length > 0) || !(veh->width > 0) || !(dt > 0)) return fo_fail(ctx, FO_E_ARG, "fo_sweep_configure: bad vehicle/dt");
  ctx->veh = *veh; ctx->hc = *hc; ctx->thr = *thr; ctx->dt = dt;
  ctx->mask = required_metrics(metric_mask);
  ctx->configured = true;
  return FO_OK;
}

int fo_sweep_reserve(fo_ctx *ctx, int max_M, int max_T, int max_A, int max_Ta) {
  if (!ctx || max_M < 0 || max_T < 1 || max_A < 0 || max_Ta < 0) return fo_fail(ctx, FO_E_ARG, "fo_sweep_reserve: bad sizes");
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int Mp = round_up(max_M > 0 ? max_M : 1, TILE);
  int rc;
  if ((rc = fo_reserve(ctx, &ctx->d_traj_tab, &ctx->cap_traj_tab, (size_t)(max_T + 1) * NEF * Mp))) return rc;   // (+ a spare row, see fo_sweep_run)
  // worst case number of chunks: one agent per wave
  // one agent per wave, or (small batches) one workgroup per agent -- but then n_tiles * A < 3 072
  // partial rows: (chunks + 1) x NPS x Mp doubles, for EVERY batch of at most max_M trajectories and max_A agents --
  // a full grid has <= ceil(A / 4) + 1 chunks of one agent per wave at worst; a batch below 3 072 (tile, agent) pairs takes
  // the horizon-split form (when max_T allows it) with one chunk per agent, but then tiles x A < 3 072 bounds the product
  // (tiles x (A + 2) <= 3 072 + 2 tiles)
  const size_t tiles = (size_t)Mp / TILE;
  size_t partial = ((size_t)(max_A + WAVES - 1) / WAVES + 2) * NPS * Mp;
  if (max_T <= QWAVES * TC) {
    const size_t split_cells = (3072 + 2 * tiles) < tiles * ((size_t)max_A + 2) ? (3072 + 2 * tiles) : tiles * ((size_t)max_A + 2);
    if (split_cells * NPS * TILE > partial) partial = split_cells * NPS * TILE;
  }
  if ((rc = fo_reserve(ctx, &ctx->d_partial, &ctx->cap_partial, partial))) return rc;
  if ((rc = fo_reserve(ctx, &ctx->d_chunk_tab, &ctx->cap_chunk_tab, 2 * ((size_t)max_A + 3)))) return rc;
  if ((rc = fo_reserve(ctx, &ctx->d_agent_tab, &ctx->cap_agent_tab, ((size_t)(max_A > 0 ? max_A : 1) * (max_Ta > 0 ? max_Ta : 1) + AGENT_PAD_ROWS) * NAF))) return rc;
  {
    const size_t cap0 = ctx->cap_agent_const;
    if ((rc = fo_reserve(ctx, &ctx->d_agent_const, &ctx->cap_agent_const, (size_t)(max_A > 0 ? max_A : 1) * NAC))) return rc;
    if (ctx->cap_agent_const != cap0) FO_HIP_TRY(ctx, hipMemset(ctx->d_agent_const, 0, ctx->cap_agent_const * sizeof(double)));   // (generation-tagged slots, fo_prep_agents_kernel)
  }
  if ((rc = fo_reserve(ctx, &ctx->d_agent_int, &ctx->cap_agent_int, (size_t)(max_A > 0 ? max_A : 1) * 2))) return rc;
  return FO_OK;
}

int fo_sweep_set_list_format(fo_ctx *ctx, int format) {
  if (!ctx) return FO_E_ARG;
  if (format != FO_LISTS_F64 && format != FO_LISTS_F32 && format != FO_LISTS_F32_EXACT)
    return fo_fail(ctx, FO_E_ARG, "fo_sweep_set_list_format: unknown format %d", format);
  ctx->list_format = format;
  return FO_OK;
}

// Book-keeping of a new agent set without the launch that fills it: buffers, sizes, generation tag.  `out` says where
// the rows go -- fo_sweep_set_agents launches fo_prep_agents_kernel on it, the fused planning step (fo_step_run) hands it to
// the phantom prediction kernel, which writes the rows of its own slots.
int fo_sweep_agents_begin_(fo_ctx *ctx, int A, int Ta, void *stream, fo_agent_table_t *out) {
  if (!ctx || !out) return FO_E_ARG;
  if (!ctx->configured) return fo_fail(ctx, FO_E_STATE, "fo_sweep_set_agents: call fo_sweep_configure first");
  if (A < 0 || (A > 0 && Ta < 1)) return fo_fail(ctx, FO_E_ARG, "fo_sweep_set_agents: bad arguments (A=%d Ta=%d)", A, Ta);
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = (hipStream_t)stream;
  int rc;
  // (+ AGENT_PAD_ROWS spare rows: the sweep reads row t + 1 of an agent without clamping, up to the trajectory horizon)
  if ((rc = fo_reserve(ctx, &ctx->d_agent_tab, &ctx->cap_agent_tab, ((size_t)(A > 0 ? A : 1) * Ta + AGENT_PAD_ROWS) * NAF))) return rc;
  {
    const size_t cap0 = ctx->cap_agent_const;
    if ((rc = fo_reserve(ctx, &ctx->d_agent_const, &ctx->cap_agent_const, (size_t)(A > 0 ? A : 1) * NAC))) return rc;
    if (ctx->cap_agent_const != cap0) FO_HIP_TRY(ctx, hipMemsetAsync(ctx->d_agent_const, 0, ctx->cap_agent_const * sizeof(double), s));   // (generation-tagged slots, fo_agent_rows.hpp)
  }
  if ((rc = fo_reserve(ctx, &ctx->d_agent_int, &ctx->cap_agent_int, (size_t)(A > 0 ? A : 1) * 2))) return rc;
  ctx->A = A;
  ctx->Ta = Ta;
  if (ctx->status_gen >= (1 << 30)) {  // generations never run out in practice; start over cleanly if they do
    FO_HIP_TRY(ctx, hipMemsetAsync(ctx->d_status, 0, 2 * sizeof(int), s));
    FO_HIP_TRY(ctx, hipMemsetAsync(ctx->d_agent_const, 0, ctx->cap_agent_const * sizeof(double), s));   // (the tagged slots as well)
    ctx->status_gen = 0;
  }
  out->tab = ctx->d_agent_tab; out->cst = ctx->d_agent_const; out->aint = ctx->d_agent_int; out->status = ctx->d_status;
  out->gen = ++ctx->status_gen;
  out->ego_mass = ctx->veh.mass; out->hlA = 0.5 * ctx->veh.length; out->hwA = 0.5 * ctx->veh.width; out->hc = ctx->hc;
  return FO_OK;
}

int fo_sweep_set_agents(fo_ctx *ctx, int A, int Ta, const double *d_pos, const double *d_yaw, const double *d_v,
                        const double *d_cov, const double *d_shape, const double *d_raw_dims, const int32_t *d_type,
                        const int32_t *d_len, void *stream) {
  if (!ctx) return FO_E_ARG;
  if (!ctx->configured) return fo_fail(ctx, FO_E_STATE, "fo_sweep_set_agents: call fo_sweep_configure first");
  if (A < 0 || (A > 0 && (Ta < 1 || !d_pos || !d_yaw || !d_v || !d_cov || !d_shape || !d_raw_dims || !d_type || !d_len)))
    return fo_fail(ctx, FO_E_ARG, "fo_sweep_set_agents: bad arguments (A=%d Ta=%d)", A, Ta);
  fo_agent_table_t at;
  int rc;
  if ((rc = fo_sweep_agents_begin_(ctx, A, Ta, stream, &at))) return rc;
  if (A > 0) {
    const int n = A * Ta;
    hipLaunchKernelGGL(fo_prep_agents_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, A, Ta, d_pos, d_yaw, d_v, d_cov,
                       d_shape, d_raw_dims, d_type, d_len, at.ego_mass, at.hlA, at.hwA, at.hc, at.tab, at.cst, at.aint, at.status, at.gen);
    FO_HIP_TRY(ctx, hipGetLastError());
  }
  return FO_OK;
}

}  // extern "C"

namespace {
// fo_sweep_run.  plan_only: everything up to the first launch -- argument checks, the grid plan, the work buffers -- and
// *plan_only = what the tile-table launch would have been given (fo_step_run hands it to the scene stage's ray kernel, whose
// extra workgroups write the table); prepped: that has happened on this stream, skip the launch.
int sweep_run(fo_ctx *ctx, int M, int T, const double *d_x, const double *d_y, const double *d_theta,
              const double *d_v, const double *d_a, double *d_cost, uint8_t *d_safe, double *d_pair_f,
              int32_t *d_pair_i, double *d_lists, void *stream, fo_prep_args_t *plan_only, bool prepped) {
  if (plan_only) *plan_only = fo_prep_args_t();
  if (!ctx) return FO_E_ARG;
  if (!ctx->configured) return fo_fail(ctx, FO_E_STATE, "fo_sweep_run: call fo_sweep_configure first");
  if (M < 0 || T < 1 || (M > 0 && (!d_x || !d_y || !d_theta || !d_v || !d_cost || !d_safe)))
    return fo_fail(ctx, FO_E_ARG, "fo_sweep_run: bad arguments (M=%d T=%d)", M, T);
  if ((d_pair_f == nullptr) != (d_pair_i == nullptr)) return fo_fail(ctx, FO_E_ARG, "fo_sweep_run: pair_f and pair_i go together");
  if (d_lists && !d_pair_f) return fo_fail(ctx, FO_E_ARG, "fo_sweep_run: lists output requires the pair outputs");
  const bool do_be = (ctx->mask & FO_M_BE) != 0;
  if (do_be && M > 0 && !d_a) return fo_fail(ctx, FO_E_ARG, "fo_sweep_run: metric 'be' needs the acceleration profile d_a");
  if (M == 0) return FO_OK;
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = (hipStream_t)stream;
  const int A = ctx->A, Ta = ctx->Ta;
  if (!plan_only && d_lists && A > 0 && T > 1 && !(ctx->mask & (FO_M_CP | FO_M_HR)))  // nothing will write them: all-ones = NaN
    FO_HIP_TRY(ctx, hipMemsetAsync(d_lists, 0xFF, (ctx->list_format != FO_LISTS_F64 ? sizeof(float) : sizeof(double)) *
                                                     FO_NL * (size_t)A * (T - 1) * M, s));
  const int Mp = round_up(M, TILE);
  const int n_tiles = Mp / TILE;
  const bool knobs = fo_env_any("FO_SWEEP_");   // (any tuning / test knob of this family in the environment at all?)
  const char *force_generic = fo_getenv(knobs, "FO_SWEEP_GENERIC");  // debug / A-B aid
  // (the queue kernel reads agent rows up to index T without clamping: horizons far beyond the predictions' take the generic kernel)
  // (the queue kernel addresses one agent's list rows by 32-bit byte offsets: (T-1) M pairs of float64 must stay under 4 GB)
  bool use_queue = !(force_generic && force_generic[0] == '1') && (!FO_DIET || T <= Ta + AGENT_PAD_ROWS - 1 || A == 0) &&
                         (size_t)(T > 1 ? T - 1 : 1) * (size_t)M * 16u < ((size_t)1 << 32);
  const int lst_mode = !d_lists ? LST_NONE : ctx->list_format == FO_LISTS_F32 ? LST_F32 : ctx->list_format == FO_LISTS_F32_EXACT ? LST_F32X : LST_F64;
  const int wpb = use_queue ? QWAVES : WAVES;  // waves per workgroup of the kernel that will run
  int apw = pick_apw(n_tiles, A, wpb);
  // a setting fo_sweep_autotune measured for this shape on this context wins over the static choice
  for (int i = 0; i < ctx->n_tuned; ++i) {
    const fo_ctx::Tuned &tu = ctx->tuned[i];
    if (tu.n_tiles == n_tiles && tu.A == A && tu.T == T && tu.lst == lst_mode && tu.pair == (d_pair_f != nullptr)) apw = tu.apw;
  }
  if (ctx->force_apw > 0) apw = ctx->force_apw;   // (fo_sweep_autotune while it measures)
  if (const char *e = fo_getenv(knobs, "FO_SWEEP_APW")) { const int v = atoi(e); if (v >= 1 && v <= 64) apw = v; }  // tuning aid
  // Small batches: with one agent per wave the grid is n_tiles x A waves; below the 3 072 wave slots of the chip the
  // horizon of every agent is split over the four waves of a workgroup instead (one workgroup per tile and agent).
  bool split = use_queue && T <= QWAVES * TC && (long)n_tiles * A < 3072;
  if (const char *e = fo_getenv(knobs, "FO_SWEEP_SPLIT")) split = use_queue && T <= QWAVES * TC && e[0] == '1';  // tests, A/B runs
  if (split) {
    // agents per workgroup of the horizon-split form, one after the other: 1.  (Measured on 2 000 x 32, 1 024 (tile, agent)
    // pairs on 768 resident workgroups: 2 / 3 / 4 agents per workgroup -- one round instead of two -- take 64 / 56 / 81 us
    // against 39: the launch lasts as long as its heaviest workgroup, the agents next to the candidates' path, and those
    // come in pairs.  FO_SWEEP_SPLIT_APW: tests, A/B runs.)
    apw = 1;
    if (const char *e = fo_getenv(knobs, "FO_SWEEP_SPLIT_APW")) { const int v = atoi(e); if (v >= 1 && v <= 16) apw = v; }
  }
  // Tapered grid (queue kernel, grids beyond one round of the chip): agents per wave halve from phase to phase down to
  // one -- see SweepArgs::ph_n.  f[]: fraction of the agents per phase; FO_SWEEP_TAPER="f0,f1,f2" overrides them
  // ("0" = no taper), a tuning aid.
  int ph_n[3] = {0, 0, 0}, ph_a[4] = {apw, apw, apw, apw};
  int n_chunks = A > 0 ? (split ? (A + apw - 1) / apw : (A + wpb * apw - 1) / (wpb * apw)) : 0;
  ph_n[0] = n_chunks;   // one phase unless tapered below
  if (use_queue && !split && apw >= 2 && A > 0) {
    double f[3] = {0.85, 0.10, 0.0};
    if (apw >= 8) { f[0] = 0.55; f[1] = 0.25; f[2] = 0.12; }
    if (const char *e = fo_getenv(knobs, "FO_SWEEP_TAPER")) {
      f[0] = 1.0; f[1] = f[2] = 0.0;
      sscanf(e, "%lf,%lf,%lf", &f[0], &f[1], &f[2]);
      if (f[0] <= 0.0) f[0] = 1.0;
    }
    if ((long)n_tiles * n_chunks >= 768 && f[0] < 1.0) {
      int left = A, ap = apw;
      n_chunks = 0;
      for (int ph = 0; ph < 3; ++ph) {
        ph_a[ph] = ap;
        ph_n[ph] = (int)(f[ph] * A) / (wpb * ap);
        if (ph_n[ph] * wpb * ap > left) ph_n[ph] = left / (wpb * ap);
        left -= ph_n[ph] * wpb * ap;
        n_chunks += ph_n[ph];
        ap = ap >= 2 ? ap / 2 : 1;
      }
      ph_a[3] = 1;
      n_chunks += (left + wpb - 1) / wpb;
    }
  }
  int rc;
  // (T + 1 rows per tile's worth: the sweep prefetches row t + 1 without clamping, the last tile's last prefetch lands in the spare)
  if ((rc = fo_reserve(ctx, &ctx->d_traj_tab, &ctx->cap_traj_tab, (size_t)(T + 1) * NEF * Mp))) return rc;
  if ((rc = fo_reserve(ctx, &ctx->d_partial, &ctx->cap_partial, (size_t)(n_chunks + 1) * NPS * Mp))) return rc;
  if ((rc = fo_reserve(ctx, &ctx->d_chunk_tab, &ctx->cap_chunk_tab, (size_t)2 * (n_chunks + 1)))) return rc;
  if (do_be && A > 0) {
    if ((rc = fo_reserve(ctx, &ctx->d_be_dist, &ctx->cap_be_dist, (size_t)(T + 1) * Mp))) return rc;  // [T][Mp] + min(a) [Mp]
    if ((rc = fo_reserve(ctx, &ctx->d_be_btn, &ctx->cap_be_btn, (size_t)A * Mp))) return rc;
    if ((rc = fo_reserve(ctx, &ctx->d_be_mask, &ctx->cap_be_mask, (size_t)A * Mp))) return rc;
  }

  if (A > 0) {
    const int tz = T > FO_PREP_TZ ? FO_PREP_TZ : T;   // horizon slice per block
    fo_prep_args_t pa;
    pa.on = 1; pa.M = M; pa.T = T; pa.tz = tz; pa.n_tiles = n_tiles; pa.nz = (T + tz - 1) / tz;
    pa.x = d_x; pa.y = d_y; pa.th = d_theta; pa.v = d_v; pa.tab = ctx->d_traj_tab; pa.chunk_tab = ctx->d_chunk_tab;
    pa.n_chunks = n_chunks; pa.wpb = wpb; pa.n0 = ph_n[0]; pa.n1 = ph_n[1]; pa.n2 = ph_n[2];
    pa.a0 = ph_a[0]; pa.a1 = ph_a[1]; pa.a2 = ph_a[2]; pa.a3 = ph_a[3];
    if (plan_only) { *plan_only = pa; return FO_OK; }
    if (!prepped) {
      hipLaunchKernelGGL(fo_prep_traj_kernel, dim3(n_tiles, 2, pa.nz), dim3(256), (size_t)2 * tz * (TILE + 1) * sizeof(double), s, pa);
      FO_HIP_TRY(ctx, hipGetLastError());
    }
    SweepArgs a{};
    a.M = M; a.Mp = Mp; a.T = T; a.A = A; a.Ta = Ta; a.n_tiles = n_tiles; a.nt8 = (n_tiles + 7) / 8; a.apw = apw;
    a.chunk_tab = ctx->d_chunk_tab;
    a.erf_tab = (const double2 *)ctx->d_erf_tab;
    a.exp_tab = (const double *)ctx->d_exp_tab;
    a.gl = (const double *)ctx->d_gl_tab;
    a.status = ctx->d_status;
    a.gen = ctx->status_gen;
    a.aint = ctx->d_agent_int;
    a.traj = ctx->d_traj_tab; a.atab = ctx->d_agent_tab; a.acst = ctx->d_agent_const; a.partial = ctx->d_partial;
    a.pair_f = d_pair_f; a.pair_i = d_pair_i; a.lists = d_lists;
    a.be_mask = do_be ? ctx->d_be_mask : nullptr;
    a.hlA = 0.5 * ctx->veh.length; a.hwA = 0.5 * ctx->veh.width; a.wb = ctx->veh.wb_rear_axle;
    a.len3 = ctx->veh.length / 2.0 * (2.0 / 3.0);  // r_x * (2/3)  (collision_probability.py:160-161)
    a.off_x = ctx->veh.length / 6.0; a.off_y = ctx->veh.width / 2.0;
    a.hc = ctx->hc; a.dt = ctx->dt; a.thr_dce = ctx->thr.dce; a.mask = ctx->mask;
    {
      const char *ab = fo_getenv(knobs, "FO_SWEEP_ABLATE");
      a.ablate = ab ? (uint32_t)atoi(ab) : 0u;
    }
    const int grid = a.nt8 * 8 * n_chunks;
    ctx->last_grid = grid; ctx->last_block = TILE * wpb; ctx->last_apw = apw;
    const bool timed = ctx->timing && ctx->n_timed < fo_ctx::kMaxTimed && (ctx->n_launch++ % ctx->timing_stride) == 0;
    if (timed) FO_HIP_TRY(ctx, hipEventRecord(ctx->ev_start[ctx->n_timed], s));
    const dim3 g(grid), b(TILE * wpb);
#if FO_TRACE
    static long long *d_trace = nullptr;
    const char *trace_path = fo_getenv(knobs, "FO_SWEEP_TRACE");
    if (trace_path && !d_trace) (void)hipMalloc((void **)&d_trace, sizeof(long long) * 4 * 65536);
    a.trace = trace_path ? d_trace : nullptr;
#endif
    const int lst = lst_mode;
    if (use_queue) {
      const uint32_t all5 = FO_M_DCE | FO_M_CP | FO_M_TTC | FO_M_TTCE | FO_M_HR;
      const bool allm = (a.mask & all5) == all5 && a.ablate == 0;
      const bool launched = allm && split ? launch_queue<true, true>(lst, d_pair_f != nullptr, g, b, s, a)
                            : split       ? launch_queue<false, true>(lst, d_pair_f != nullptr, g, b, s, a)
                            : allm        ? launch_queue<true, false>(lst, d_pair_f != nullptr, g, b, s, a)
                                          : launch_queue<false, false>(lst, d_pair_f != nullptr, g, b, s, a);
      if (!launched)
        return fo_fail(ctx, FO_E_STATE, "fo_sweep_run: no queue-kernel instantiation for list format %d with allm=%d split=%d "
                       "(internal: such batches belong to the generic kernel)", lst, (int)allm, (int)split);
    } else {
      if (lst == LST_F64) hipLaunchKernelGGL((fo_sweep_generic_kernel<true, LST_F64>), g, b, 0, s, a);
      else if (lst_is32(lst)) hipLaunchKernelGGL((fo_sweep_generic_kernel<true, LST_F32>), g, b, 0, s, a);   // (converts at the store: exact)
      else if (d_pair_f) hipLaunchKernelGGL((fo_sweep_generic_kernel<true, LST_NONE>), g, b, 0, s, a);
      else hipLaunchKernelGGL((fo_sweep_generic_kernel<false, LST_NONE>), g, b, 0, s, a);
    }
    FO_HIP_TRY(ctx, hipGetLastError());
    if (timed) FO_HIP_TRY(ctx, hipEventRecord(ctx->ev_stop[ctx->n_timed++], s));
#if FO_TRACE
    if (a.trace && fo_getenv(knobs, "FO_SWEEP_TRACE_DUMP")) {   // (set for the one launch that is to be dumped)
      (void)hipStreamSynchronize(s);
      // (FO_SWEEP_TRACE_PHASES: the per-workgroup phase stamps as well, rows [32768, 32768 + grid) -- tools/split_trace.py)
      const size_t rows = fo_getenv(knobs, "FO_SWEEP_TRACE_PHASES") && grid <= 32768 ? (size_t)32768 + grid : (size_t)grid;
      long long *h = (long long *)malloc(sizeof(long long) * 4 * rows);
      (void)hipMemcpy(h, d_trace, sizeof(long long) * 4 * rows, hipMemcpyDeviceToHost);
      if (FILE *f = fopen(trace_path, "wb")) { fwrite(h, sizeof(long long), 4 * rows, f); fclose(f); }
      free(h);
    }
#endif
  }
  if (plan_only) return FO_OK;   // (no agents: nothing to prepare)
  const double *be_btn = nullptr;
  if (do_be && A > 0 && T >= 1) {
    double *dist = ctx->d_be_dist, *mina = ctx->d_be_dist + (size_t)T * Mp;
    hipLaunchKernelGGL(fo_be_prep_kernel, dim3((Mp + 255) / 256), dim3(256), 0, s, M, Mp, T, d_x, d_y, d_a, dist, mina);
    hipLaunchKernelGGL(fo_be_kernel, dim3(n_tiles, (A + 3) / 4), dim3(256), 0, s, M, Mp, T, A, Ta, ctx->d_traj_tab, dist,
                       mina, ctx->d_agent_tab, ctx->d_agent_const, ctx->d_agent_int, ctx->d_be_mask, 0.5 * ctx->veh.length,
                       0.5 * ctx->veh.width, ctx->veh.wb_rear_axle, ctx->veh.a_max, ctx->dt, ctx->d_be_btn, d_pair_f);
    FO_HIP_TRY(ctx, hipGetLastError());
    be_btn = ctx->d_be_btn;
  }
  hipLaunchKernelGGL(fo_reduce_kernel, dim3((M + 63) / 64), dim3(64 * RED_WAVES), 0, s, M, Mp, A, n_chunks, ctx->d_partial,
                     ctx->thr, ctx->mask, be_btn, d_cost, d_safe, ctx->d_status, ctx->status_gen);
  FO_HIP_TRY(ctx, hipGetLastError());
  return FO_OK;
}

}  // namespace

extern "C" {

int fo_sweep_run(fo_ctx *ctx, int M, int T, const double *d_x, const double *d_y, const double *d_theta,
                 const double *d_v, const double *d_a, double *d_cost, uint8_t *d_safe, double *d_pair_f,
                 int32_t *d_pair_i, double *d_lists, void *stream) {
  return sweep_run(ctx, M, T, d_x, d_y, d_theta, d_v, d_a, d_cost, d_safe, d_pair_f, d_pair_i, d_lists, stream, nullptr, false);
}

// the two halves of fo_sweep_run for the fused planning step (fo_step_run, fo_api.hip): plan -> *prep; [the scene stage writes
// the tile table] -> run without the table launch
int fo_sweep_plan_(fo_ctx *ctx, int M, int T, const double *d_x, const double *d_y, const double *d_theta,
                   const double *d_v, const double *d_a, double *d_cost, uint8_t *d_safe, double *d_pair_f,
                   int32_t *d_pair_i, double *d_lists, void *stream, fo_prep_args_t *prep) {
  if (!prep) return FO_E_ARG;
  return sweep_run(ctx, M, T, d_x, d_y, d_theta, d_v, d_a, d_cost, d_safe, d_pair_f, d_pair_i, d_lists, stream, prep, false);
}
int fo_sweep_run_prepped_(fo_ctx *ctx, int M, int T, const double *d_x, const double *d_y, const double *d_theta,
                          const double *d_v, const double *d_a, double *d_cost, uint8_t *d_safe, double *d_pair_f,
                          int32_t *d_pair_i, double *d_lists, void *stream) {
  return sweep_run(ctx, M, T, d_x, d_y, d_theta, d_v, d_a, d_cost, d_safe, d_pair_f, d_pair_i, d_lists, stream, nullptr, true);
}

// Per-shape choice of the sweep kernel's agents-per-wave, measured on the caller's own batch (include/fo_hip.h).
int fo_sweep_autotune(fo_ctx *ctx, int M, int T, const double *d_x, const double *d_y, const double *d_theta, const double *d_v,
                      const double *d_a, double *d_cost, uint8_t *d_safe, double *d_pair_f, int32_t *d_pair_i, double *d_lists,
                      int reps, int *best_apw, double *ms4, void *stream) {
  if (!ctx) return FO_E_ARG;
  if (reps < 1) return fo_fail(ctx, FO_E_ARG, "fo_sweep_autotune: reps must be positive");
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = (hipStream_t)stream;
  const int A = ctx->A;
  const int n_tiles = round_up(M > 0 ? M : 1, TILE) / TILE;
  const int lst = !d_lists ? LST_NONE : ctx->list_format == FO_LISTS_F32 ? LST_F32 : ctx->list_format == FO_LISTS_F32_EXACT ? LST_F32X : LST_F64;
  static const int cand[4] = {1, 2, 4, 8};
  hipEvent_t e0, e1;
  FO_HIP_TRY(ctx, hipEventCreate(&e0));
  FO_HIP_TRY(ctx, hipEventCreate(&e1));
  int rc = FO_OK, best = 0;
  double best_ms = INFINITY;
  const bool was_timing = ctx->timing;
  ctx->timing = false;
  for (int c = 0; c < 4 && rc == FO_OK; ++c) {
    ctx->force_apw = cand[c];
    for (int i = 0; i < (reps + 3) / 4 && rc == FO_OK; ++i)
      rc = fo_sweep_run(ctx, M, T, d_x, d_y, d_theta, d_v, d_a, d_cost, d_safe, d_pair_f, d_pair_i, d_lists, stream);
    if (rc == FO_OK && hipEventRecord(e0, s) != hipSuccess) rc = FO_E_HIP;
    for (int i = 0; i < reps && rc == FO_OK; ++i)
      rc = fo_sweep_run(ctx, M, T, d_x, d_y, d_theta, d_v, d_a, d_cost, d_safe, d_pair_f, d_pair_i, d_lists, stream);
    float ms = 0.f;
    if (rc == FO_OK && (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                        hipEventElapsedTime(&ms, e0, e1) != hipSuccess)) rc = FO_E_HIP;
    if (rc == FO_OK) {
      if (ms4) ms4[c] = (double)ms / reps;
      if ((double)ms / reps < best_ms) { best_ms = (double)ms / reps; best = cand[c]; }
    }
  }
  ctx->force_apw = 0;
  ctx->timing = was_timing;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc != FO_OK) return rc == FO_E_HIP ? fo_fail(ctx, FO_E_HIP, "fo_sweep_autotune: HIP event failure") : rc;
  // remember (replace an entry of the same shape; the table is small and round-robin)
  int slot = -1;
  for (int i = 0; i < ctx->n_tuned; ++i) {
    const fo_ctx::Tuned &tu = ctx->tuned[i];
    if (tu.n_tiles == n_tiles && tu.A == A && tu.T == T && tu.lst == lst && tu.pair == (d_pair_f != nullptr)) slot = i;
  }
  if (slot < 0) {
    if (ctx->n_tuned < fo_ctx::kMaxTuned) slot = ctx->n_tuned++;
    else slot = ctx->next_tuned++ % fo_ctx::kMaxTuned;
  }
  ctx->tuned[slot] = fo_ctx::Tuned{n_tiles, A, T, lst, d_pair_f != nullptr, best};
  if (best_apw) *best_apw = best;
  // leave the outputs of a run with the chosen setting behind
  return fo_sweep_run(ctx, M, T, d_x, d_y, d_theta, d_v, d_a, d_cost, d_safe, d_pair_f, d_pair_i, d_lists, stream);
}

int fo_sweep_timing(fo_ctx *ctx, int enable) {
  if (!ctx) return FO_E_ARG;
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (enable && !ctx->ev_start) {
    ctx->ev_start = new hipEvent_t[fo_ctx::kMaxTimed];
    ctx->ev_stop = new hipEvent_t[fo_ctx::kMaxTimed];
    for (int i = 0; i < fo_ctx::kMaxTimed; ++i) {
      FO_HIP_TRY(ctx, hipEventCreate(&ctx->ev_start[i]));
      FO_HIP_TRY(ctx, hipEventCreate(&ctx->ev_stop[i]));
    }
  }
  ctx->timing = enable != 0;
  ctx->timing_stride = enable > 1 ? enable : 1;  // enable = k > 1: every k-th launch carries the event pair
  ctx->n_launch = 0;
  ctx->n_timed = 0;
  return FO_OK;
}

int fo_sweep_timing_read(fo_ctx *ctx, double *total_ms, int *launches) {
  if (!ctx || !total_ms || !launches) return FO_E_ARG;
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  double sum = 0.0;
  for (int i = 0; i < ctx->n_timed; ++i) {
    float ms = 0.f;
    FO_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_stop[i]));
    FO_HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_start[i], ctx->ev_stop[i]));
    sum += ms;
  }
  *total_ms = sum;
  *launches = ctx->n_timed;
  ctx->n_timed = 0;
  return FO_OK;
}

int fo_sweep_timing_read_each(fo_ctx *ctx, double *each_ms, int cap, int *launches) {
  if (!ctx || !each_ms || !launches || cap < 0) return FO_E_ARG;
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  for (int i = 0; i < ctx->n_timed && i < cap; ++i) {
    float ms = 0.f;
    FO_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_stop[i]));
    FO_HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_start[i], ctx->ev_stop[i]));
    each_ms[i] = ms;
  }
  *launches = ctx->n_timed;
  ctx->n_timed = 0;
  return FO_OK;
}

int fo_sweep_last_launch(const fo_ctx *ctx, int *grid, int *block, int *agents_per_wave) {
  if (!ctx) return FO_E_ARG;
  if (grid) *grid = ctx->last_grid;
  if (block) *block = ctx->last_block;
  if (agents_per_wave) *agents_per_wave = ctx->last_apw;
  return FO_OK;
}

}  // extern "C"
