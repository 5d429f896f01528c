// Issue-rate microbenchmark for the VALU instructions the sweep kernel is made of (gfx950).
// Each kernel runs N iterations of 8 independent dependency chains of one instruction type; every CU gets
// `waves` waves per SIMD.  Prints cycles per wave-instruction per SIMD (s_memtime deltas, median over waves).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int N = 4096;

template <int OP>
__global__ void k(double *out, unsigned long long *cyc, double seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double m = 1.0000001, c = 1e-9;
  float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3, f4 = (float)a4, f5 = (float)a5, f6 = (float)a6, f7 = (float)a7;
  int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < N; ++it) {
#define R8(EXPR) { auto &x = a0; EXPR; } { auto &x = a1; EXPR; } { auto &x = a2; EXPR; } { auto &x = a3; EXPR; } \
                 { auto &x = a4; EXPR; } { auto &x = a5; EXPR; } { auto &x = a6; EXPR; } { auto &x = a7; EXPR; }
#define F8(EXPR) { auto &x = f0; EXPR; } { auto &x = f1; EXPR; } { auto &x = f2; EXPR; } { auto &x = f3; EXPR; } \
                 { auto &x = f4; EXPR; } { auto &x = f5; EXPR; } { auto &x = f6; EXPR; } { auto &x = f7; EXPR; }
#define I8(EXPR) { auto &x = i0; EXPR; } { auto &x = i1; EXPR; } { auto &x = i2; EXPR; } { auto &x = i3; EXPR; } \
                 { auto &x = i4; EXPR; } { auto &x = i5; EXPR; } { auto &x = i6; EXPR; } { auto &x = i7; EXPR; }
    if (OP == 0) { R8(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c))) }
    if (OP == 1) { R8(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(m))) }
    if (OP == 2) { R8(asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(c))) }
    if (OP == 3) { R8(asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(c))) }
    if (OP == 4) { R8(asm volatile("v_rcp_f64 %0, %0" : "+v"(x))) }
    if (OP == 5) { R8(asm volatile("v_rsq_f64 %0, %0" : "+v"(x))) }
    if (OP == 6) { R8(asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(x))) }
    if (OP == 7) { R8(asm volatile("v_rndne_f64 %0, %0" : "+v"(x))) }
    if (OP == 8) { F8(asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x))) }
    if (OP == 9) { I8(asm volatile("v_add_u32 %0, %0, %0" : "+v"(x))) }
    if (OP == 10) { I8(asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(x))) }
    if (OP == 11) { R8(asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(x), "v"(c) : "vcc")) }
    if (OP == 12) { R8(asm volatile("v_fma_f64 %0, %0, s[4:5], %1" : "+v"(x) : "v"(c) : "s4", "s5")) }
    if (OP == 13) { R8(asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i0) : "v"(x))) }
    if (OP == 14) { R8(asm volatile("v_sqrt_f64 %0, %0" : "+v"(x))) }
    if (OP == 15) { I8(asm volatile("v_readlane_b32 s6, %0, 3\n v_writelane_b32 %0, s6, 4" : "+v"(x) :: "s6")) }
    // mixed streams (round 3): what a float64 instruction leaves room for in the issue of its own wave
    if (OP == 16) { R8(asm volatile("v_fma_f64 %0, %0, %1, %2\n s_add_u32 s6, s6, 1" : "+v"(x) : "v"(m), "v"(c) : "s6", "scc")) }
    if (OP == 17) { R8(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f32 %1, %1, %1, %1" : "+v"(x), "+v"(f0) : "v"(m), "v"(c))) }
    if (OP == 18) { R8(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f32 %1, %1, %1, %1\n s_add_u32 s6, s6, 1" : "+v"(x), "+v"(f0) : "v"(m), "v"(c) : "s6", "scc")) }
    if (OP == 19) { I8(asm volatile("s_add_u32 s6, s6, 1\n s_add_u32 s7, s7, 1" ::: "s6", "s7", "scc")) }
    if (OP == 20) { R8(asm volatile("v_fma_f64 %0, %0, %1, %2\n s_add_u32 s6, s6, 1\n s_add_u32 s7, s7, 1\n s_add_u32 s8, s8, 1" : "+v"(x) : "v"(m), "v"(c) : "s6", "s7", "s8", "scc")) }
    if (OP == 21) { F8(asm volatile("v_fma_f32 %0, %0, %0, %0\n s_add_u32 s6, s6, 1" : "+v"(x) :: "s6", "scc")) }
    // v_cndmask variants (round 3: the plain form above measured 14 cycles -- which part of it?)
    if (OP == 22) { I8(asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(x) : "v"(i7))) }
    if (OP == 23) { I8(asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[4:5]" : "+v"(x) : "v"(i7) : "s4", "s5")) }
    if (OP == 24) { I8(asm volatile("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(i7) : "vcc")) }
    if (OP == 25) { R8(asm volatile("v_cmp_lt_f64 vcc, %0, %2\n v_cndmask_b32 %1, %1, %3, vcc" : "+v"(x), "+v"(i0) : "v"(c), "v"(i7) : "vcc")) }
    if (OP == 26) { I8(asm volatile("v_max_i32 %0, %0, %1" : "+v"(x) : "v"(i7))) }
    // round 5: is it the encoding (VOP2, implicit vcc) or the register (vcc named in a VOP3)?
    if (OP == 27) { I8(asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(x) : "v"(i7))) }
    if (OP == 28) { R8(asm volatile("v_mov_b64 %0, %1" : "=v"(x) : "v"(c))) }
    if (OP == 29) { R8(asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f0) : "v"(x))) }
    if (OP == 30) { I8(asm volatile("v_and_or_b32 %0, %0, -8, 3" : "+v"(x))) }
    if (OP == 31) { R8(asm volatile("v_cmp_lt_f64_e64 s[4:5], %0, %2\n v_cndmask_b32_e64 %1, %1, %3, s[4:5]" : "+v"(x), "+v"(i0) : "v"(c), "v"(i7) : "s4", "s5")) }
    if (OP == 32) { R8(asm volatile("v_min_f64 %0, %0, %1" : "+v"(x) : "v"(c))) }
    if (OP == 33) { I8(asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x) : "v"(i7) : "vcc")) }
    if (OP == 34) { I8(asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[4:5]\n s_nop 0" : "+v"(x) : "v"(i7) : "s4", "s5")) }
    if (OP == 35) { I8(asm volatile("v_cndmask_b32_e64 %0, %0, 1, s[4:5]" : "+v"(x) :: "s4", "s5")) }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 +
                                               i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
}

template <int OP>
int run(const char *name, int waves_per_simd) {
  const int threads = 64 * 4 * waves_per_simd;  // one block per CU fills 4 SIMDs with `waves_per_simd` waves each
  const int blocks = 256;
  double *out; unsigned long long *cyc;
  CHECK(hipMalloc(&out, sizeof(double) * blocks * threads));
  CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * blocks * threads / 64));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.5);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.5);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(blocks * threads / 64);
  CHECK(hipMemcpy(h.data(), cyc, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2];
  const int per_iter = (OP == 15 || OP == 19) ? 16 : 8;   // (mixed streams: per GROUP of instructions)
  // cycles of SIMD time per wave-instruction = wave time / (instructions per wave * waves sharing the SIMD)
  printf("%-22s waves/SIMD=%d  wave-cycles/instr=%6.2f  SIMD-cycles/instr=%6.2f\n", name, waves_per_simd,
         med / (N * per_iter), med / (N * per_iter) / waves_per_simd);
  hipFree(out); hipFree(cyc);
  return 0;
}

int main() {
  for (int w : {1, 3}) {
    run<16>("f64 + salu (group)", w); run<17>("f64 + f32 (group)", w); run<18>("f64 + f32 + salu (group)", w);
    run<19>("s_add_u32", w); run<20>("f64 + 3 salu (group)", w); run<21>("f32 + salu (group)", w);
    run<0>("v_fma_f64", w); run<1>("v_mul_f64", w); run<2>("v_add_f64", w); run<3>("v_max_f64", w);
    run<12>("v_fma_f64 (sgpr src)", w); run<11>("v_cmp_lt_f64", w);
    run<4>("v_rcp_f64", w); run<5>("v_rsq_f64", w); run<14>("v_sqrt_f64", w); run<6>("v_ldexp_f64", w); run<7>("v_rndne_f64", w);
    run<13>("v_cvt_i32_f64", w); run<8>("v_fma_f32", w); run<9>("v_add_u32", w); run<10>("v_cndmask_b32", w);
    run<15>("v_readlane+writelane", w);
    run<22>("v_cndmask vcc, distinct src", w); run<23>("v_cndmask_e64 sgpr mask", w); run<24>("v_cmp_i32 + v_cndmask (group)", w);
    run<25>("v_cmp_f64 + v_cndmask (group)", w); run<26>("v_max_i32", w);
    run<27>("v_cndmask_e64 vcc", w); run<28>("v_mov_b64", w); run<29>("v_cvt_f32_f64", w); run<30>("v_and_or_b32", w);
    run<31>("v_cmp_f64_e64 + v_cndmask_e64 sgpr (group)", w); run<32>("v_min_f64", w); run<33>("v_addc_co_u32 vcc", w);
    run<34>("v_cndmask_e64 sgpr + s_nop (group)", w); run<35>("v_cndmask_e64 sgpr, inline const", w);
  }
  return 0;
}
