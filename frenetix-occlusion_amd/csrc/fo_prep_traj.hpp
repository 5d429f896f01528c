// fo_prep_traj.hpp -- the candidates' tile table (trajectories [M][T] -> [tile][T][4 pairs][64 lanes]) and the sweep launch's
// chunk table, as a device function: fo_prep_traj_kernel (fo_sweep.hip) is one caller, the ray kernel of the fused planning
// step (fo_scene.hip, fo_step_run) the other -- there the table is written by extra workgroups of a launch that leaves most
// of the chip idle, instead of by a launch of its own in front of the sweep.  No contractable arithmetic inside (one
// product per output), so the two translation units produce the same bits whatever their -ffp-contract.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

constexpr int FO_PREP_TILE = 64;   // = TILE of fo_sweep.hip (static_assert there)
constexpr int FO_PREP_NEF = 8;     // = NEF
constexpr int FO_PREP_TZ = 8;      // longest horizon slice of a block
typedef double fo_prep_d2 __attribute__((ext_vector_type(2)));

struct fo_prep_args_t {
  int on = 0;                       // 0: nothing to do (no agents / no candidates)
  int M = 0, T = 0, tz = 1;
  int n_tiles = 0, nz = 0;          // blocks: n_tiles x 2 (positions | heading + speed) x nz horizon slices
  const double *x = nullptr, *y = nullptr, *th = nullptr, *v = nullptr;
  double *tab = nullptr;
  int *chunk_tab = nullptr;         // SweepArgs::chunk_tab of the sweep launch that follows: phases of n0 / n1 / n2 / the
  int n_chunks = 0, wpb = 0;        // remaining chunks with a0 / a1 / a2 / a3 agents per wave
  int n0 = 0, n1 = 0, n2 = 0, a0 = 1, a1 = 1, a2 = 1, a3 = 1;
  __host__ __device__ int blocks() const { return on ? n_tiles * 2 * nz : 0; }
};

// block (bx, by, bz) of the table; sh: [2][tz][FO_PREP_TILE + 1] doubles of LDS; any block size
__device__ __forceinline__ void fo_prep_traj_block(const fo_prep_args_t &p, int bx, int by, int bz, double *sh) {
#pragma clang fp contract(off)
  constexpr int TILE = FO_PREP_TILE, NEF = FO_PREP_NEF;
  const int M = p.M, T = p.T, tz = p.tz;
  if (p.chunk_tab && bx == 0 && by == 0 && bz == 0)
    for (int c = threadIdx.x; c < p.n_chunks; c += blockDim.x) {
      int k0, ap;
      if (c < p.n0) { ap = p.a0; k0 = c * p.a0; }
      else if (c < p.n0 + p.n1) { ap = p.a1; k0 = p.n0 * p.a0 + (c - p.n0) * p.a1; }
      else if (c < p.n0 + p.n1 + p.n2) { ap = p.a2; k0 = p.n0 * p.a0 + p.n1 * p.a1 + (c - p.n0 - p.n1) * p.a2; }
      else { ap = p.a3; k0 = p.n0 * p.a0 + p.n1 * p.a1 + p.n2 * p.a2 + (c - p.n0 - p.n1 - p.n2) * p.a3; }
      p.chunk_tab[2 * c] = k0 * p.wpb;
      p.chunk_tab[2 * c + 1] = ap;
    }
  const int m0 = bx * TILE;
  const int n = min(TILE, M - m0);
  const int ld = TILE + 1;
  const int f = by;  // 0: positions (x, y); 1: heading and speed -> (cos, sin), (theta, v), (v cos, v sin)
  const int t0 = bz * tz, nt = min(tz, T - t0);   // this block's slice of the horizon (latency: short blocks)
  if (nt <= 0) return;
  const double *s0 = (f == 0 ? p.x : p.th) + (size_t)m0 * T + t0, *s1 = (f == 0 ? p.y : p.v) + (size_t)m0 * T + t0;
  double *sh1 = sh + (size_t)tz * ld;
  for (int i = threadIdx.x; i < n * nt; i += blockDim.x) {
    const int ml = i / nt, tl = i - ml * nt;
    sh[tl * ld + ml] = s0[(size_t)ml * T + tl];
    sh1[tl * ld + ml] = s1[(size_t)ml * T + tl];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nt * TILE; i += blockDim.x) {
    const int tl = i / TILE, ml = i % TILE;
    const int src = tl * ld + min(ml, n - 1);  // pad lanes replicate the last trajectory of the tile
    const double a0 = sh[src], a1 = sh1[src];
    fo_prep_d2 *dst = (fo_prep_d2 *)(p.tab + ((size_t)bx * T + t0 + tl) * NEF * TILE) + ml;  // pair q of lane ml: dst[q * TILE]
    if (f == 0) {
      dst[0 * TILE] = fo_prep_d2{a0, a1};
    } else {
      double sn, cs;
      sincos(a0, &sn, &cs);
      // (speeds beyond 5 km/s are capped in the velocity components the harm model reads: the relative speed then stays
      // below 1e4 m/s, inside the range of the sweep's table exp, without a clamp per sample)
      const double vc = fmin(fmax(a1, -5.0e3), 5.0e3);
      dst[1 * TILE] = fo_prep_d2{cs, sn};
      dst[2 * TILE] = fo_prep_d2{a0, a1};
      dst[3 * TILE] = fo_prep_d2{vc * cs, vc * sn};
    }
  }
}
