// fo_ctx.hpp -- context object behind the C ABI (include/fo_hip.h).  Internal to libfo_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "fo_hip.h"

struct fo_ctx {
  int device = 0;
  char err[512] = {0};

  // ---- sweep configuration (Metric.__init__)
  fo_vehicle_t veh{};
  fo_harm_coeff_t hc{};
  fo_thresholds_t thr{};
  uint32_t mask = 0;   // after dependency closure
  double dt = 0.1;
  bool configured = false;
  int list_format = FO_LISTS_F64;   // element type of the per-timestep lists fo_sweep_run writes

  // ---- agents (prepared form, HBM)
  int A = 0, Ta = 0;
  double *d_agent_tab = nullptr;    // [A][Ta][NAF]
  double *d_agent_const = nullptr;  // [A][NAC]
  void *d_erf_tab = nullptr;        // erf lookup table (fo_sweep.hip)
  void *d_exp_tab = nullptr;        // 2^(j/256) table
  void *d_gl_tab = nullptr;         // Gauss-Legendre nodes / weights (box probabilities under correlated covariances)
  int32_t *d_agent_int = nullptr;   // [A][2] protection class, valid length
  size_t cap_agent_int = 0;
  int *d_status = nullptr;          // [2] device status words: generation of the last fo_sweep_set_agents call that met
                                    // [0] an unusable covariance, [1] a correlated one (compared with status_gen;
                                    // never cleared, so no per-step memset)
  int status_gen = 0;
  size_t cap_agent_tab = 0, cap_agent_const = 0;

  // ---- trajectory tile buffer + partial reductions (HBM workspace)
  double *d_traj_tab = nullptr;     // [T][NEF][Mp]
  double *d_partial = nullptr;      // [n_chunks][NPS][Mp]
  size_t cap_traj_tab = 0, cap_partial = 0;
  int *d_chunk_tab = nullptr;       // [n_chunks][2] first agent / agents per wave of every chunk (tapered grid)
  size_t cap_chunk_tab = 0;
  // ---- BE metric workspace (only allocated when FO_M_BE is active)
  double *d_be_dist = nullptr, *d_be_btn = nullptr;
  signed char *d_be_mask = nullptr;
  size_t cap_be_dist = 0, cap_be_btn = 0, cap_be_mask = 0;

  // ---- last launch (profiling aid) + optional HIP-event timing of the sweep kernel alone
  int last_grid = 0, last_block = 0, last_apw = 0;
  bool timing = false;
  int timing_stride = 1, n_launch = 0;  // every timing_stride-th launch is timed
  static constexpr int kMaxTimed = 1024;
  hipEvent_t *ev_start = nullptr, *ev_stop = nullptr;
  int n_timed = 0;

  // ---- agents-per-wave of the sweep kernel measured per batch shape (fo_sweep_autotune)
  struct Tuned { int n_tiles, A, T, lst; bool pair; int apw; };
  static constexpr int kMaxTuned = 16;
  Tuned tuned[kMaxTuned];
  int n_tuned = 0, next_tuned = 0, force_apw = 0;

  // ---- the two host transfers of a planning step (fo_step_t::h_obstacles / h_mirror, ABI 12): a ring of pinned staging
  // slots for the obstacle rows on their way to HBM (the caller's buffer is free again when fo_step_run returns; a slot is
  // written again only after the copy queued from it four steps ago has completed) and the event behind the step's
  // device-to-host mirror (fo_step_mirror_wait)
  static constexpr int kRing = 4;
  static constexpr size_t kRingSlot = 64 << 10;
  char *h_ring = nullptr;            // [kRing][kRingSlot], hipHostMalloc
  hipEvent_t ev_ring[kRing] = {};
  bool ring_used[kRing] = {};
  int ring_next = 0;
  hipEvent_t ev_mirror = nullptr;
  bool mirror_queued = false;
  void *mirror_host = nullptr, *mirror_dev = nullptr;   // the last h_mirror and its device-side address (hipHostGetDevicePointer)
  int64_t mirror_bytes = 0;                              // ... and its size (the cache is keyed on both)

  // ---- scene (ray-cast / grid) state lives in fo_scene.hip
  void *scene = nullptr;
};

// Tuning / test knobs come from the environment and are looked at on every call (tests switch them at run time).  One pass
// over `environ` for the family's prefix decides whether any getenv() is worth making: none is set in production, and eight
// look-ups per planning step were 2.5 us of the host's 45.
extern char **environ;
inline bool fo_env_any(const char *prefix) {
  const size_t n = strlen(prefix);
  for (char **e = environ; e && *e; ++e)
    if (strncmp(*e, prefix, n) == 0) return true;
  return false;
}
inline const char *fo_getenv(bool any, const char *name) { return any ? getenv(name) : nullptr; }

inline int fo_fail(fo_ctx *ctx, int code, const char *fmt, ...) {
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
    va_end(ap);
  }
  return code;
}

#define FO_HIP_TRY(ctx, expr)                                                                   \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return fo_fail(ctx, FO_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// grow-only device buffer
template <typename T>
inline int fo_reserve(fo_ctx *ctx, T **ptr, size_t *cap, size_t need_elems) {
  if (need_elems <= *cap) return FO_OK;
  if (*ptr) FO_HIP_TRY(ctx, hipFree(*ptr));
  *ptr = nullptr;
  *cap = 0;
  FO_HIP_TRY(ctx, hipMalloc((void **)ptr, need_elems * sizeof(T)));
  *cap = need_elems;
  return FO_OK;
}
