// Store-throughput microbenchmark: the sweep kernel's output pattern without any of its arithmetic.
// lists [5][A][T-1][M] float64; workgroup = (tile of 64 trajectories, chunk of 16 agents), 4 waves x 4 agents each;
// a wave writes, per agent and sample, five 512-byte rows that lie A*(T-1)*M*8 bytes apart.
// Variants: nontemporal or plain stores; the sweep's layout or one linear stream per wave; a dependent-arithmetic
// filler between samples (to see how much of the store time hides under VALU work).
//   hipcc -O3 --offload-arch=gfx950 -o store_pattern store_pattern.hip && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <bool NT, bool LINEAR, int FILL>
__global__ __launch_bounds__(256) void k_store(double *__restrict__ out, int A, int Tm1, int M, int n_tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x % n_tiles, chunk = blockIdx.x / n_tiles;
  const int m = tile * 64 + lane;
  if (m >= M) return;
  const size_t ls = (size_t)A * Tm1 * M;
  double acc = (double)lane;
  for (int j = 0; j < 4; ++j) {
    const int k = chunk * 16 + wave * 4 + j;
    if (k >= A) break;
    double *p = LINEAR ? out + (((size_t)k * n_tiles + tile) * Tm1) * 5 * 64 + lane : out + (size_t)k * Tm1 * M + m;
    for (int t = 0; t < Tm1; ++t) {
#pragma unroll
      for (int f = 0; f < FILL; ++f) acc = acc * 1.0000001 + 1e-9;   // dependent f64 chain
#pragma unroll
      for (int l = 0; l < 5; ++l) {
        double *q = LINEAR ? p + l * 64 : p + l * ls;
        if (NT) __builtin_nontemporal_store(acc + l, q); else *q = acc + l;
      }
      p += LINEAR ? 5 * 64 : M;
    }
  }
}

template <bool NT, bool LINEAR, int FILL>
float run(double *d, int A, int Tm1, int M) {
  const int n_tiles = (M + 63) / 64, chunks = (A + 15) / 16;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_store<NT, LINEAR, FILL>), dim3(n_tiles * chunks), dim3(256), 0, 0, d, A, Tm1, M, n_tiles);
  hipEventRecord(e0);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((k_store<NT, LINEAR, FILL>), dim3(n_tiles * chunks), dim3(256), 0, 0, d, A, Tm1, M, n_tiles);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 10;
}

int main() {
  const int A = 256, Tm1 = 30, M = 10000;
  const size_t n = 5ull * A * Tm1 * (((size_t)M + 63) / 64 * 64);
  double *d; hipMalloc(&d, n * 8);
  const double gb = 5.0 * A * Tm1 * M * 8 / 1e9;
#define ROW(NT, LIN, F) { float ms = run<NT, LIN, F>(d, A, Tm1, M); printf("%-12s %-8s filler %3d: %.3f ms  %.2f TB/s\n", NT ? "nontemporal" : "plain", LIN ? "linear" : "strided", F, ms, gb / ms); }
  ROW(true, false, 0) ROW(false, false, 0) ROW(true, true, 0) ROW(false, true, 0)
  ROW(true, false, 32) ROW(false, false, 32) ROW(true, false, 128) ROW(false, false, 128)
  hipFree(d);
  return 0;
}
