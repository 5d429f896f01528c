// fo_api.hip -- context management of the C ABI (include/fo_hip.h).
#include <hip/hip_runtime.h>
#include <new>
#include <cstdlib>
#include "fo_ctx.hpp"
#include "fo_agent_rows.hpp"
#include "fo_prep_traj.hpp"

extern "C" void fo_scene_destroy_(fo_ctx *ctx);  // fo_scene.hip
extern "C" void *fo_step_direct_mirror_(fo_ctx *ctx, const fo_step_t *p);   // fo_scene.hip
extern "C" int fo_scene_step_(fo_ctx *ctx, const fo_step_t *p, const fo_agent_table_t *at, const fo_prep_args_t *prep, void *stream);  // fo_scene.hip
extern "C" int fo_sweep_agents_begin_(fo_ctx *ctx, int A, int Ta, void *stream, fo_agent_table_t *out);   // fo_sweep.hip
extern "C" int fo_sweep_init_(fo_ctx *ctx);      // fo_sweep.hip
extern "C" int fo_sweep_plan_(fo_ctx *ctx, int M, int T, const double *d_x, const double *d_y, const double *d_theta, const double *d_v,
                              const double *d_a, double *d_cost, uint8_t *d_safe, double *d_pair_f, int32_t *d_pair_i, double *d_lists,
                              void *stream, fo_prep_args_t *prep);   // fo_sweep.hip
extern "C" int fo_sweep_run_prepped_(fo_ctx *ctx, int M, int T, const double *d_x, const double *d_y, const double *d_theta,
                                     const double *d_v, const double *d_a, double *d_cost, uint8_t *d_safe, double *d_pair_f,
                                     int32_t *d_pair_i, double *d_lists, void *stream);   // fo_sweep.hip

static int fo_step_stage_obstacles_(fo_ctx *ctx, const fo_step_t *p, hipStream_t stream);
static int fo_step_queue_mirror_(fo_ctx *ctx, const fo_step_t *p, hipStream_t stream, bool direct);
static int fo_step_body_(fo_ctx *ctx, const fo_step_t *p, void *stream, bool stages, bool cells, bool rules, int slots, int slot0,
                         int cell_agents);

extern "C" {

int fo_abi_version(void) { return FO_ABI_VERSION; }

#ifndef FO_BUILD_ID
#define FO_BUILD_ID "unstamped"
#endif
const char *fo_build_id(void) { return FO_BUILD_ID; }

void fo_destroy(fo_ctx *ctx);

int fo_create(fo_ctx **out, int device) {
  if (!out) return FO_E_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return FO_E_HIP;  // no GPU: fail loudly
  if (hipSetDevice(device) != hipSuccess) return FO_E_HIP;
  fo_ctx *ctx = new (std::nothrow) fo_ctx();
  if (!ctx) return FO_E_NOMEM;
  ctx->device = device;
  if (hipMalloc((void **)&ctx->d_status, 2 * sizeof(int)) != hipSuccess) { delete ctx; return FO_E_NOMEM; }
  if (hipMemset(ctx->d_status, 0, 2 * sizeof(int)) != hipSuccess) { (void)hipFree(ctx->d_status); delete ctx; return FO_E_HIP; }
  if (fo_sweep_init_(ctx) != FO_OK) { fo_destroy(ctx); return FO_E_HIP; }
  *out = ctx;
  return FO_OK;
}

void fo_destroy(fo_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  fo_scene_destroy_(ctx);
  if (ctx->d_agent_tab) (void)hipFree(ctx->d_agent_tab);
  if (ctx->d_agent_const) (void)hipFree(ctx->d_agent_const);
  if (ctx->d_traj_tab) (void)hipFree(ctx->d_traj_tab);
  if (ctx->d_partial) (void)hipFree(ctx->d_partial);
  if (ctx->d_chunk_tab) (void)hipFree(ctx->d_chunk_tab);
  if (ctx->d_be_dist) (void)hipFree(ctx->d_be_dist);
  if (ctx->d_be_btn) (void)hipFree(ctx->d_be_btn);
  if (ctx->d_be_mask) (void)hipFree(ctx->d_be_mask);
  if (ctx->d_status) (void)hipFree(ctx->d_status);
  if (ctx->d_erf_tab) (void)hipFree(ctx->d_erf_tab);
  if (ctx->d_exp_tab) (void)hipFree(ctx->d_exp_tab);
  if (ctx->d_gl_tab) (void)hipFree(ctx->d_gl_tab);
  if (ctx->d_agent_int) (void)hipFree(ctx->d_agent_int);
  if (ctx->h_ring) (void)hipHostFree(ctx->h_ring);
  for (int i = 0; i < fo_ctx::kRing; ++i)
    if (ctx->ev_ring[i]) (void)hipEventDestroy(ctx->ev_ring[i]);
  if (ctx->ev_mirror) (void)hipEventDestroy(ctx->ev_mirror);
  if (ctx->ev_start) {
    for (int i = 0; i < fo_ctx::kMaxTimed; ++i) { (void)hipEventDestroy(ctx->ev_start[i]); (void)hipEventDestroy(ctx->ev_stop[i]); }
    delete[] ctx->ev_start;
    delete[] ctx->ev_stop;
  }
  delete ctx;
}

const char *fo_last_error(const fo_ctx *ctx) { return ctx ? ctx->err : "null context"; }

// blocks until `stream` drained, then reports data-dependent failures recorded on the device
int fo_sweep_check(fo_ctx *ctx, void *stream) {
  if (!ctx) return FO_E_ARG;
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  FO_HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
  int st = 0;
  FO_HIP_TRY(ctx, hipMemcpy(&st, ctx->d_status, sizeof(int), hipMemcpyDeviceToHost));
  if (st != 0 && st == ctx->status_gen)  // recorded by the latest fo_sweep_set_agents
    return fo_fail(ctx, FO_E_UNSUPPORTED_COV,
                   "an agent's covariance is no usable matrix (asymmetric, not positive, or |correlation| > 0.99): "
                   "the affected collision probabilities are NaN");
  return FO_OK;
}

// One planning step on one stream (include/fo_hip.h, fo_step_t): the results of the stage calls fo_scene_fan ->
// fo_scene_visibility -> fo_scene_spawn [-> fo_scene_spawn_rules -> fo_scene_spawn_rule_agents] -> fo_sweep_set_agents ->
// fo_sweep_run, bit for bit, in four launches less -- the ray fan is worked out inside the ray kernel (whose spare workgroups
// also write the sweep's tile table of the candidates), the sampler's
// candidate cells are flagged inside the compaction of the occluded cells, and the phantom prediction kernels write their
// slots' rows of the sweep's agent table themselves.
// (FO_STEP_STAGES=1 in the environment: the plain sequence of stage calls, for A/B runs.)
int fo_step_run(fo_ctx *ctx, const fo_step_t *p, void *stream) {
  if (!ctx || !p) return fo_fail(ctx, FO_E_ARG, "fo_step_run: null argument");
  int rc;
  const char *e_stages = fo_getenv(fo_env_any("FO_STEP_"), "FO_STEP_STAGES");   // (every call: tests switch it at run time)
  const bool stages = e_stages && e_stages[0] == '1';
  if (p->spawn_mode != FO_SPAWN_CELLS && p->spawn_mode != FO_SPAWN_RULES && p->spawn_mode != FO_SPAWN_BOTH)
    return fo_fail(ctx, FO_E_ARG, "fo_step_run: unknown spawn_mode %d", p->spawn_mode);
  const bool cells = p->spawn_mode != FO_SPAWN_RULES, rules = p->spawn_mode != FO_SPAWN_CELLS;
  const int R = p->routes > 0 ? p->routes : 1;
  const int cell_agents = cells ? p->max_agents : 0, rule_points = rules ? p->max_rule_points : 0;
  const int slots = (cell_agents + rule_points) * R, slot0 = cell_agents * R;
  if (!p->d_pos || !p->d_yaw || !p->d_v || !p->d_cov || !p->d_shape || !p->d_raw_dims || !p->d_type || !p->d_len || slots < 1 ||
      (cells && p->max_agents < 1) || p->T_agents < 1 ||
      (rules && (p->max_rule_points < 1 || !p->d_rule_points || !p->d_n_rule_points || !p->d_path6 || !p->d_pos0 || !p->d_yaw0)))
    return fo_fail(ctx, FO_E_ARG, "fo_step_run: bad arguments");
  if ((rc = fo_sweep_set_list_format(ctx, p->list_format))) return rc;   // the format is an argument of the run
  if ((rc = fo_step_stage_obstacles_(ctx, p, (hipStream_t)stream))) return rc;
  rc = fo_step_body_(ctx, p, stream, stages, cells, rules, slots, slot0, cell_agents);
  if (rc == FO_OK) rc = fo_step_queue_mirror_(ctx, p, (hipStream_t)stream, !stages && fo_step_direct_mirror_(ctx, p) != nullptr);
  return rc;
}

int fo_step_mirror_wait(fo_ctx *ctx) {
  if (!ctx) return FO_E_ARG;
  if (!ctx->mirror_queued) return FO_OK;
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  FO_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_mirror));
  return FO_OK;
}

}  // extern "C"

// the obstacle rows of the step: caller's host buffer -> pinned ring slot -> HBM, queued in front of the first launch
static int fo_step_stage_obstacles_(fo_ctx *ctx, const fo_step_t *p, hipStream_t stream) {
  if (!p->h_obstacles || p->obstacles_bytes <= 0) return FO_OK;
  if (!p->d_obstacles) return fo_fail(ctx, FO_E_ARG, "fo_step_run: h_obstacles without d_obstacles");
  if ((size_t)p->obstacles_bytes > fo_ctx::kRingSlot)
    return fo_fail(ctx, FO_E_ARG, "fo_step_run: %lld bytes of obstacle rows exceed the staging slot (%zu): copy them yourself and "
                   "pass h_obstacles = NULL", (long long)p->obstacles_bytes, fo_ctx::kRingSlot);
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!ctx->h_ring) {
    FO_HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_ring, fo_ctx::kRing * fo_ctx::kRingSlot, hipHostMallocDefault));
    for (int i = 0; i < fo_ctx::kRing; ++i) FO_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_ring[i], hipEventDisableTiming));
  }
  const int i = ctx->ring_next;
  ctx->ring_next = (i + 1) % fo_ctx::kRing;
  if (ctx->ring_used[i]) FO_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_ring[i]));   // (four steps back: long done)
  char *slot = ctx->h_ring + (size_t)i * fo_ctx::kRingSlot;
  memcpy(slot, p->h_obstacles, (size_t)p->obstacles_bytes);
  FO_HIP_TRY(ctx, hipMemcpyAsync(p->d_obstacles, slot, (size_t)p->obstacles_bytes, hipMemcpyHostToDevice, stream));
  FO_HIP_TRY(ctx, hipEventRecord(ctx->ev_ring[i], stream));
  ctx->ring_used[i] = true;
  return FO_OK;
}

// the step's mirror (hit ids and visibility flags for the host views of the reference's side effects): behind the last launch
static int fo_step_queue_mirror_(fo_ctx *ctx, const fo_step_t *p, hipStream_t stream, bool direct) {
  if (!p->h_mirror || p->mirror_bytes <= 0) return FO_OK;
  if (!p->d_mirror) return fo_fail(ctx, FO_E_ARG, "fo_step_run: h_mirror without d_mirror");
  FO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  // (release to SYSTEM scope: the direct mirror is written by kernel stores into mapped host memory, and the host reads it
  // after hipEventSynchronize on this event -- the default release scope of an event is the runtime's choice)
  if (!ctx->ev_mirror) FO_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_mirror, hipEventDisableTiming | hipEventReleaseToSystem));
  // (the fused step stores the mirror from the kernels that produce its contents -- fo_scene.hip, FanArgs::hit_host -- when the
  // mirror is the interface's (hit ids | visibility flags) pair; the event behind the step is all that is left to queue)
  if (!direct)
    FO_HIP_TRY(ctx, hipMemcpyAsync(p->h_mirror, p->d_mirror, (size_t)p->mirror_bytes, hipMemcpyDeviceToHost, stream));
  FO_HIP_TRY(ctx, hipEventRecord(ctx->ev_mirror, stream));
  ctx->mirror_queued = true;
  return FO_OK;
}

static int fo_step_body_(fo_ctx *ctx, const fo_step_t *p, void *stream, bool stages, bool cells, bool rules, int slots, int slot0,
                         int cell_agents) {
  int rc;
  if (stages) {
    if ((rc = fo_scene_fan(ctx, p->n_rays, p->ego_yaw, p->fov_deg, p->r, p->polygon_footprint, p->d_dirs, p->d_rmax, p->d_half, stream))) return rc;
    if ((rc = fo_scene_visibility(ctx, p->ego_x, p->ego_y, p->head_x, p->head_y, p->r, p->full_circle, p->exact_cells, p->n_rays,
                                  p->d_dirs, p->d_rmax, p->d_half, p->d_edge_skip, p->O, p->d_ocorn, p->d_ocen, p->d_oflags,
                                  p->win_ix0, p->win_iy0, p->win_nx, p->win_ny, p->d_range, p->d_hit_id, p->d_ring, p->d_obst_vis,
                                  p->d_cls, p->d_occ_idx, p->d_n_occ, stream))) return rc;
    if (cells && (rc = fo_scene_spawn(ctx, p->d_cls, p->win_ix0, p->win_iy0, p->win_nx, p->win_ny, p->ego_x, p->ego_y, p->head_x, p->head_y,
                                      p->min_ahead, p->max_dist, p->all_occluded, p->max_agents, p->routes, p->type4, p->speed4, p->raw_l4,
                                      p->raw_w4, p->infl_l4, p->infl_w4, p->n_path, p->d_path, p->T_agents, p->dt, p->var0, p->var_factor,
                                      p->d_cell, p->d_pos0, p->d_yaw0, p->d_n, p->d_pos, p->d_yaw, p->d_v, p->d_cov, p->d_shape,
                                      p->d_raw_dims, p->d_type, p->d_len, stream))) return rc;
    if (rules) {
      if ((rc = fo_scene_spawn_rules(ctx, p->d_cls, p->win_ix0, p->win_iy0, p->win_nx, p->win_ny, p->n_path6, p->d_path6, p->O, p->d_ocorn,
                                     p->d_ocen, p->d_oyaw, p->d_odims, p->d_oflags, p->d_obst_vis, &p->rule, p->max_rule_points,
                                     p->d_rule_points, p->d_n_rule_points, stream))) return rc;
      const size_t T = (size_t)p->T_agents, s0 = (size_t)slot0;
      if ((rc = fo_scene_spawn_rule_agents(ctx, p->max_rule_points, p->d_rule_points, p->d_n_rule_points, p->routes, &p->rule_types,
                                           p->n_path, p->d_path, p->T_agents, p->dt, p->var0, p->var_factor, p->d_pos0 + 2 * (size_t)cell_agents,
                                           p->d_yaw0 + cell_agents, p->d_pos + s0 * T * 2, p->d_yaw + s0 * T, p->d_v + s0 * T,
                                           p->d_cov + s0 * T * 4, p->d_shape + 2 * s0, p->d_raw_dims + 2 * s0, p->d_type + s0,
                                           p->d_len + s0, stream))) return rc;
    }
    if ((rc = fo_sweep_set_agents(ctx, slots, p->T_agents, p->d_pos, p->d_yaw, p->d_v, p->d_cov, p->d_shape, p->d_raw_dims, p->d_type,
                                  p->d_len, stream))) return rc;
  } else {
    fo_agent_table_t at;
    fo_prep_args_t prep;
    if ((rc = fo_sweep_agents_begin_(ctx, slots, p->T_agents, stream, &at))) return rc;
    // the sweep's plan for this batch (argument checks, grid, work buffers): its tile table of the candidates is written by
    // extra workgroups of the scene stage's first launch
    if ((rc = fo_sweep_plan_(ctx, p->M, p->T, p->d_x, p->d_y, p->d_theta, p->d_vel, p->d_acc, p->d_cost, p->d_safe, p->d_pair_f,
                             p->d_pair_i, p->d_lists, stream, &prep))) {
      ctx->A = 0;
      return rc;
    }
    if ((rc = fo_scene_step_(ctx, p, &at, &prep, stream))) {
      ctx->A = 0;   // the agent set was announced but not written: a sweep after this failure evaluates no agents
      return rc;
    }
    return fo_sweep_run_prepped_(ctx, p->M, p->T, p->d_x, p->d_y, p->d_theta, p->d_vel, p->d_acc, p->d_cost, p->d_safe, p->d_pair_f,
                                 p->d_pair_i, p->d_lists, stream);
  }
  return fo_sweep_run(ctx, p->M, p->T, p->d_x, p->d_y, p->d_theta, p->d_vel, p->d_acc, p->d_cost, p->d_safe, p->d_pair_f,
                      p->d_pair_i, p->d_lists, stream);
}
