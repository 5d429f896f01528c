// What a grid-wide barrier costs on MI355X (8 XCDs, 8 L2s) against the kernel launch it would replace (round 6: the planning
// step's scene stage is a chain of six dependent launches of 4-10 us).
//
// A persistent kernel of G workgroups x B threads runs N rounds of: every workgroup writes W doubles of its own slice
// (plain stores: dirty lines in its XCD's L2), BARRIER, reads the slice of the workgroup G/2 + 1 further (another XCD:
// workgroups are dealt round-robin over the XCDs) and checks the round number.  Barrier variants:
//   flat   one agent-scope counter (atomic at the memory side), everybody spins on it
//   hier   a counter per XCD (workgroup-scope atomic = executed in that XCD's L2; the XCD is read from HW_REG_XCC_ID), the
//          last arrival of an XCD adds to the agent-scope counter, everybody spins on that
//   nofence  = flat without the release / acquire fences (protocol cost alone; the data check is then expected to fail)
//   lean   flat, but ONE release (buffer_wbl2 sc1) and ONE acquire (buffer_inv sc1) per workgroup, by thread 0 between the two
//          workgroup barriers -- the write-back / invalidate act on the whole L2 / the CU's L1, not on the issuing wave's lines
//   hlean  hier + the fences of lean
//   xcd0   the persistent part lives on ONE XCD: the launch has 8 x G workgroups, those that find themselves on XCD 0
//          (HW_REG_XCC_ID) stay, the others return at once.  All traffic then goes through one L2: release = s_waitcnt vmcnt(0)
//          (stores are write-through in the L1 and complete in the L2), arrival = an L2 atomic, spin on sc0 loads (miss the L1),
//          acquire = buffer_inv sc0 (this CU's L1) + s_dcache_inv -- no L2 write-back, no L2 invalidate
// against `chain`: the same rounds as N launches of a kernel that does one round (stream order = the barrier).
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench/grid_barrier.hip -o tools/microbench/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Sync {
  unsigned int count;        // agent scope
  unsigned int pad0[31];
  unsigned int xcd[8][32];   // one line per XCD
  unsigned int xcd_n[8];     // workgroups per XCD (written by a census pass)
};

// phase stamps of ONE barrier (round g_trace_round, -1 = off): per workgroup its XCD, the wall clock (100 MHz) when thread 0 has
// released its stores and is about to arrive, when it has seen the count complete, and when its acquire is done
__device__ long long *g_trace = nullptr;
__device__ int g_trace_round = -1;

__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15;
}

enum { FLAT = 0, HIER = 1, NOFENCE = 2, LEAN = 3, HLEAN = 4, XCD0 = 5, XCD0A = 6, XCD0B = 7 };
__host__ __device__ constexpr bool is_xcd0(int k) { return k == XCD0 || k == XCD0A || k == XCD0B; }

template <int KIND>
__device__ __forceinline__ void grid_barrier(Sync *s, unsigned int round, unsigned int G, int xcd) {
  constexpr bool hier = KIND == HIER || KIND == HLEAN;
  constexpr bool fenced = KIND != NOFENCE && !is_xcd0(KIND);
  if (is_xcd0(KIND)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // every wave: its stores have completed in the L2
  __syncthreads();
  if (threadIdx.x == 0) {
    const bool tr = g_trace && (int)round == g_trace_round;
    if (fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (tr) { g_trace[4 * blockIdx.x] = xcc_id(); g_trace[4 * blockIdx.x + 1] = wall_clock64(); }
    if (is_xcd0(KIND)) {
      __hip_atomic_fetch_add(&s->xcd[0][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (hier) {
      const unsigned int nx = s->xcd_n[xcd];
      const unsigned int old = __hip_atomic_fetch_add(&s->xcd[xcd][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (old == nx * (round + 1) - 1) __hip_atomic_fetch_add(&s->count, nx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      __hip_atomic_fetch_add(&s->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned int want = G * (round + 1);
    const long long t_in = wall_clock64();   // (100 MHz; a barrier that has not completed after 0.5 s gives up: no hung box)
    if (is_xcd0(KIND)) {
      unsigned int v;
      do {
        // how the spinning wave looks at the counter: xcd0 = a load of workgroup scope (sc0: found to HIT the L1 outside
        // threadgroup-split mode -- the wave never sees the count move), xcd0a = an atomic OR of 0 with return (executes in
        // the L2), xcd0b = a load of agent scope (sc1)
        if (KIND == XCD0) asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(&s->xcd[0][0]) : "memory");
        else if (KIND == XCD0A) v = __hip_atomic_fetch_or(&s->xcd[0][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(&s->xcd[0][0]) : "memory");
        if (v >= want) break;
        __builtin_amdgcn_s_sleep(1);
        const long long waited = wall_clock64() - t_in;
        if (waited > 50000000ll) { __hip_atomic_store(&s->pad0[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        // (somebody else has given up already: every later barrier of the launch falls through after 0.1 ms)
        if (waited > 10000ll && __hip_atomic_load(&s->pad0[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      } while (true);
      if (KIND == XCD0) asm volatile("buffer_inv sc0\n s_dcache_inv\n s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("buffer_inv sc1\n s_dcache_inv\n s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
      while (__hip_atomic_load(&s->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        const long long waited = wall_clock64() - t_in;
        if (waited > 50000000ll) { __hip_atomic_store(&s->pad0[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        if (waited > 10000ll && __hip_atomic_load(&s->pad0[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      }
      if (tr) g_trace[4 * blockIdx.x + 2] = wall_clock64();
      if (fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (tr) g_trace[4 * blockIdx.x + 3] = wall_clock64();
    }
  }
  __syncthreads();
}

__global__ void census(Sync *s) {
  if (threadIdx.x == 0) atomicAdd(&s->xcd_n[xcc_id()], 1u);
}

template <int KIND>
__global__ void persistent(Sync *s, double *buf, int W, int N, int *errors, long long *ticks, int members) {
  int G = gridDim.x, b = blockIdx.x;
  const int xcd = (KIND == HIER || KIND == HLEAN || is_xcd0(KIND)) ? xcc_id() : 0;
  if (is_xcd0(KIND)) {
    if (xcd != 0) return;
    __shared__ int rank;
    if (threadIdx.x == 0) rank = (int)atomicAdd(&s->xcd[1][0], 1u);   // a ticket: my index among the members
    __syncthreads();
    b = rank;
    G = members;
  }
  constexpr bool every_thread_fences = KIND == FLAT || KIND == HIER;
  const long long t0 = wall_clock64();
  int err = 0;
  for (int r = 0; r < N; ++r) {
    for (int i = threadIdx.x; i < W; i += blockDim.x) buf[(size_t)b * W + i] = (double)(r + 1);
    if (every_thread_fences) __threadfence();   // every thread releases its own stores (the first, naive form)
    grid_barrier<KIND>(s, 2u * (unsigned)r, (unsigned)G, xcd);
    if (every_thread_fences) __threadfence();
    const int o = (b + G / 2 + 1) % G;
    const volatile double *vb = buf;    // (not hoisted over the barrier)
    for (int i = threadIdx.x; i < W; i += blockDim.x) err += vb[(size_t)o * W + i] != (double)(r + 1);
    // (second barrier of a round: nobody may overwrite a slice somebody is still reading -- counts as a barrier of its own)
    grid_barrier<KIND>(s, 2u * (unsigned)r + 1u, (unsigned)G, xcd);
  }
  if (err) atomicAdd(errors, err);
  if (threadIdx.x == 0 && b == 0) ticks[0] = wall_clock64() - t0;
}

__global__ void one_round_write(double *buf, int W, int r) {
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < W; i += blockDim.x) buf[(size_t)b * W + i] = (double)(r + 1);
}
__global__ void one_round_read(const double *buf, int W, int r, int *errors) {
  const int G = gridDim.x, b = blockIdx.x;
  const int o = (b + G / 2 + 1) % G;
  int err = 0;
  for (int i = threadIdx.x; i < W; i += blockDim.x) err += buf[(size_t)o * W + i] != (double)(r + 1);
  if (err) atomicAdd(errors, err);
}

template <int KIND>
static int run_persistent(const char *name, int G, int B, int W, int N, Sync *d_s, double *d_buf, int *d_err, long long *d_ticks) {
  CHECK(hipMemset(d_s, 0, sizeof(Sync)));
  CHECK(hipMemset(d_err, 0, sizeof(int)));
  census<<<(is_xcd0(KIND) ? 8 * G : G), B>>>(d_s);
  CHECK(hipDeviceSynchronize());
  // occupancy: the grid has to be resident at once
  int per_cu = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, persistent<KIND>, B, 0));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int launch_G = is_xcd0(KIND) ? 8 * G : G;     // (xcd0: G members on one XCD = 32 CUs)
  if ((long long)per_cu * (is_xcd0(KIND) ? prop.multiProcessorCount / 8 : prop.multiProcessorCount) < G) { printf("%-8s G=%d B=%d: grid not resident (%d per CU)\n", name, G, B, per_cu); return 0; }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  long long ticks = 0;
  for (int rep = 0; rep < 5; ++rep) {
    unsigned int zero[40] = {0};
    CHECK(hipMemcpy(d_s, zero, sizeof(unsigned int) * 32, hipMemcpyHostToDevice));   // count
    for (int x = 0; x < 8; ++x) CHECK(hipMemcpy(&d_s->xcd[x][0], zero, sizeof(unsigned int), hipMemcpyHostToDevice));
    CHECK(hipEventRecord(e0));
    unsigned int members = 0;
    CHECK(hipMemcpy(&members, &d_s->xcd_n[0], sizeof(unsigned int), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&d_s->xcd[1][0], zero, sizeof(unsigned int), hipMemcpyHostToDevice));   // (xcd0: the ticket counter)
    persistent<KIND><<<launch_G, B>>>(d_s, d_buf, W, N, d_err, d_ticks, (int)members);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) { best = ms; CHECK(hipMemcpy(&ticks, d_ticks, sizeof(long long), hipMemcpyDeviceToHost)); }
    unsigned int g = 0;
    CHECK(hipMemcpy(&g, &d_s->pad0[0], sizeof(unsigned int), hipMemcpyDeviceToHost));
    if (g) { printf("%-8s G=%d B=%d W=%d: A BARRIER GAVE UP (0.5 s) in repetition %d -- no figure\n", name, G, B, W, rep); return 0; }
  }
  int err = 0;
  CHECK(hipMemcpy(&err, d_err, sizeof(int), hipMemcpyDeviceToHost));
  unsigned int xn[8];
  CHECK(hipMemcpy(xn, d_s->xcd_n, sizeof(xn), hipMemcpyDeviceToHost));
  printf("%-8s G=%4d B=%4d W=%6d: %7.2f us per barrier (kernel %8.1f us / %d barriers; in-kernel clock %7.2f us), stale reads %d, wg/XCD %u %u %u %u %u %u %u %u\n",
         name, G, B, W, 1e3 * best / (2.0 * N), 1e3 * best, 2 * N, ticks / 100.0 / (2.0 * N), err, xn[0], xn[1], xn[2], xn[3], xn[4], xn[5], xn[6], xn[7]);
  return 0;
}

static int run_chain(int G, int B, int W, int N, double *d_buf, int *d_err) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipMemset(d_err, 0, sizeof(int)));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < N; ++r) {
      one_round_write<<<G, B>>>(d_buf, W, r);
      one_round_read<<<G, B>>>(d_buf, W, r, d_err);
    }
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  int err = 0;
  CHECK(hipMemcpy(&err, d_err, sizeof(int), hipMemcpyDeviceToHost));
  printf("chain    G=%4d B=%4d W=%6d: %7.2f us per launch  (%d launches %8.1f us), stale reads %d\n", G, B, W, 1e3 * best / (2.0 * N), 2 * N, 1e3 * best, err);
  return 0;
}

int main(int argc, char **argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);   // (a run that is cut off still leaves its lines)
  const int N = argc > 1 ? atoi(argv[1]) : 200;
  const char *only = argc > 2 ? argv[2] : "";   // "xcd0": that part alone
  Sync *d_s;
  double *d_buf;
  int *d_err;
  long long *d_ticks;
  const size_t max_elems = (size_t)2048 * 65536;
  CHECK(hipMalloc((void **)&d_s, sizeof(Sync)));
  CHECK(hipMalloc((void **)&d_buf, sizeof(double) * max_elems));
  CHECK(hipMalloc((void **)&d_err, sizeof(int)));
  CHECK(hipMalloc((void **)&d_ticks, sizeof(long long)));
  CHECK(hipMemset(d_buf, 0, sizeof(double) * max_elems));
  const int Gs[] = {64, 256, 512, 1024, 2048};
  const int Ws[] = {64, 8192};   // 512 B and 64 KB per workgroup and round
  if (!strcmp(only, "trace")) {
    // one barrier of the best cross-XCD variant (counter per XCD, one fence pair per workgroup) under the stamps: who arrives
    // when, who is let go when, per XCD
    for (int G : {64, 256}) {
      long long *d_tr;
      CHECK(hipMalloc((void **)&d_tr, sizeof(long long) * 4 * G));
      CHECK(hipMemset(d_tr, 0, sizeof(long long) * 4 * G));
      const int round = N;   // (the first barrier of round N / 2 of the kernel's loop: warm)
      CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &d_tr, sizeof(d_tr)));
      CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_round), &round, sizeof(round)));
      if (run_persistent<HLEAN>("hlean", G, 256, 64, N, d_s, d_buf, d_err, d_ticks)) return 1;
      std::vector<long long> h(4 * G);
      CHECK(hipMemcpy(h.data(), d_tr, sizeof(long long) * 4 * G, hipMemcpyDeviceToHost));
      long long t0 = h[1], last_arr = h[1];
      for (int b = 0; b < G; ++b) { if (h[4 * b + 1] < t0) t0 = h[4 * b + 1]; if (h[4 * b + 1] > last_arr) last_arr = h[4 * b + 1]; }
      printf("trace    G=%4d: one barrier, us from the first arrival; the last workgroup arrives at %.2f\n", G, (last_arr - t0) * 0.01);
      for (int x = 0; x < 8; ++x) {
        long long a0 = 1ll << 62, a1 = 0, r0 = 1ll << 62, r1 = 0, q1 = 0;
        int n = 0;
        for (int b = 0; b < G; ++b)
          if (h[4 * b] == x) {
            ++n;
            a0 = h[4 * b + 1] < a0 ? h[4 * b + 1] : a0; a1 = h[4 * b + 1] > a1 ? h[4 * b + 1] : a1;
            r0 = h[4 * b + 2] < r0 ? h[4 * b + 2] : r0; r1 = h[4 * b + 2] > r1 ? h[4 * b + 2] : r1;
            q1 = h[4 * b + 3] > q1 ? h[4 * b + 3] : q1;
          }
        if (n) printf("   XCD %d (%3d workgroups): arrivals %.2f .. %.2f, count seen complete %.2f .. %.2f, acquire done by %.2f\n", x, n,
                      (a0 - t0) * 0.01, (a1 - t0) * 0.01, (r0 - t0) * 0.01, (r1 - t0) * 0.01, (q1 - t0) * 0.01);
      }
      const int off = -1;
      CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_round), &off, sizeof(off)));
      CHECK(hipFree(d_tr));
    }
    return 0;
  }
  if (!strcmp(only, "chain")) {
    for (int G : {64, 256, 1024}) if (run_chain(G, 256, 64, N, d_buf, d_err)) return 1;
    return 0;
  }
  if (!strcmp(only, "xcd0")) {
    for (int B : {256, 512, 1024})
      for (int G : {16, 32, 64, 128})
        for (int W : {64, 8192})
        {
          if (run_persistent<XCD0A>("xcd0a", G, B, W, N, d_s, d_buf, d_err, d_ticks)) return 1;
          if (run_persistent<XCD0B>("xcd0b", G, B, W, N, d_s, d_buf, d_err, d_ticks)) return 1;
        }
    if (run_persistent<XCD0>("xcd0", 64, 256, 64, N, d_s, d_buf, d_err, d_ticks)) return 1;
    return 0;
  }
  for (int W : Ws)
    for (int G : Gs) {
      const int B = 256;
      if (run_chain(G, B, W, N, d_buf, d_err)) return 1;
      if (run_persistent<FLAT>("flat", G, B, W, N, d_s, d_buf, d_err, d_ticks)) return 1;
      if (run_persistent<HIER>("hier", G, B, W, N, d_s, d_buf, d_err, d_ticks)) return 1;
      if (run_persistent<NOFENCE>("nofence", G, B, W, N, d_s, d_buf, d_err, d_ticks)) return 1;
      if (run_persistent<LEAN>("lean", G, B, W, N, d_s, d_buf, d_err, d_ticks)) return 1;
      if (run_persistent<HLEAN>("hlean", G, B, W, N, d_s, d_buf, d_err, d_ticks)) return 1;
      if (G <= 256 && run_persistent<XCD0>("xcd0", G, B, W, N, d_s, d_buf, d_err, d_ticks)) return 1;
    }
  // workgroup shapes of the scene stage: one wave, and 1 024 threads
  for (int B : {64, 1024})
    for (int G : {256, 1024}) {
      if (run_chain(G, B, 64, N, d_buf, d_err)) return 1;
      if (run_persistent<FLAT>("flat", G, B, 64, N, d_s, d_buf, d_err, d_ticks)) return 1;
      if (run_persistent<HIER>("hier", G, B, 64, N, d_s, d_buf, d_err, d_ticks)) return 1;
      if (run_persistent<HLEAN>("hlean", G, B, 64, N, d_s, d_buf, d_err, d_ticks)) return 1;
      if (G <= 256 && run_persistent<XCD0>("xcd0", G, B, 64, N, d_s, d_buf, d_err, d_ticks)) return 1;
    }
  // one XCD, the shapes a scene stage would use there: 32 ... 128 workgroups of 256 ... 1 024 threads
  for (int B : {256, 512, 1024})
    for (int G : {16, 32, 64, 128})
      for (int W : {64, 8192})
        if (run_persistent<XCD0>("xcd0", G, B, W, N, d_s, d_buf, d_err, d_ticks)) return 1;
  return 0;
}
