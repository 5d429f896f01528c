/* The metric sweep through the C ABI alone (include/fo_hip.h): no Python, no torch -- what a host written in any
 * language with a C FFI does.  64 straight candidate trajectories fanned out from the origin against two phantom
 * predictions (a pedestrian crossing at x = 15 m, a car ahead in the lane); prints one line per trajectory:
 *   m  safe  wttc  min_dce  max_obst_risk_all  max_obst_harm_all  max_collision_probability_all
 * with %.17g, which tests/test_c_abi_example_gpu.py compares with the oracle on the same inputs.
 *
 * build:  hipcc -x c examples/sweep_from_c.c -Iinclude -Lfrenetix-occlusion_amd/lib -lfo_hip -Wl,-rpath,... -o sweep_from_c
 * (hipcc only for the HIP runtime's include and library paths; the file is plain C) */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "fo_hip.h"

#define M 64
#define T 31
#define A 2
#define DT 0.1

#define CHECK_FO(call)                                                                         \
  do {                                                                                         \
    int rc_ = (call);                                                                          \
    if (rc_ != FO_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, fo_last_error(ctx)); return 1; } \
  } while (0)
#define CHECK_HIP(call)                                                                        \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } \
  } while (0)

static void *to_device(const void *h, size_t bytes) {
  void *d = NULL;
  if (hipMalloc(&d, bytes) != hipSuccess || hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
  return d;
}

int main(void) {
  static double x[M][T], y[M][T], th[M][T], v[M][T];
  static double pos[A][T][2], yaw[A][T], av[A][T], cov[A][T][2][2], shape[A][2], raw[A][2];
  int32_t type[A] = {FO_TYPE_PEDESTRIAN, FO_TYPE_CAR}, len[A] = {T, T};
  for (int m = 0; m < M; ++m) {              /* headings from -0.2 to +0.2 rad, speeds from 6 to 12 m/s */
    const double psi = -0.2 + 0.4 * m / (M - 1), sp = 6.0 + 6.0 * ((m * 7) % M) / (M - 1);
    for (int t = 0; t < T; ++t) {
      x[m][t] = sp * DT * t * cos(psi); y[m][t] = sp * DT * t * sin(psi); th[m][t] = psi; v[m][t] = sp;
    }
  }
  for (int t = 0; t < T; ++t) {
    pos[0][t][0] = 15.0; pos[0][t][1] = -3.0 + 1.4 * DT * t; yaw[0][t] = 1.5707963267948966; av[0][t] = 1.4;   /* pedestrian */
    pos[1][t][0] = 22.0 + 3.0 * DT * t; pos[1][t][1] = 0.4; yaw[1][t] = 0.0; av[1][t] = 3.0;          /* slow car */
    for (int k = 0; k < A; ++k) {
      const double var = 0.1 * pow(1.05, t);   /* agent.py:260-280 */
      cov[k][t][0][0] = var; cov[k][t][1][1] = var; cov[k][t][0][1] = cov[k][t][1][0] = 0.0;
    }
  }
  shape[0][0] = 0.6; shape[0][1] = 0.65; raw[0][0] = 0.5; raw[0][1] = 0.5;     /* inflated / un-inflated (config.yaml) */
  shape[1][0] = 5.4; shape[1][1] = 2.34; raw[1][0] = 4.5; raw[1][1] = 1.8;

  fo_ctx *ctx = NULL;
  if (fo_create(&ctx, 0) != FO_OK) { fprintf(stderr, "fo_create failed: no GPU?\n"); return 1; }
  const fo_vehicle_t veh = {4.508, 1.610, 1.4227, 1093.3, 11.5};
  const fo_harm_coeff_t hc = {-4.457, 0.177, 0.244, -0.431, -4.591, 0.185, 3.164, 0.288};   /* harm_params.json */
  const fo_thresholds_t thr = {0.1, 1.0, NAN, NAN, NAN, NAN};                               /* harm, risk */
  CHECK_FO(fo_sweep_configure(ctx, &veh, &hc, &thr, FO_M_HR | FO_M_TTC | FO_M_TTCE | FO_M_DCE | FO_M_WTTC | FO_M_CP, DT));

  double *d_x = to_device(x, sizeof x), *d_y = to_device(y, sizeof y), *d_th = to_device(th, sizeof th),
         *d_v = to_device(v, sizeof v);
  double *d_pos = to_device(pos, sizeof pos), *d_yaw = to_device(yaw, sizeof yaw), *d_av = to_device(av, sizeof av),
         *d_cov = to_device(cov, sizeof cov), *d_shape = to_device(shape, sizeof shape), *d_raw = to_device(raw, sizeof raw);
  int32_t *d_type = to_device(type, sizeof type), *d_len = to_device(len, sizeof len);
  double *d_cost = NULL;
  uint8_t *d_safe = NULL;
  CHECK_HIP(hipMalloc((void **)&d_cost, sizeof(double) * M * FO_NC));
  CHECK_HIP(hipMalloc((void **)&d_safe, M));
  if (!d_x || !d_y || !d_th || !d_v || !d_pos || !d_yaw || !d_av || !d_cov || !d_shape || !d_raw || !d_type || !d_len) return 1;

  CHECK_FO(fo_sweep_set_agents(ctx, A, T, d_pos, d_yaw, d_av, d_cov, d_shape, d_raw, d_type, d_len, NULL));
  CHECK_FO(fo_sweep_run(ctx, M, T, d_x, d_y, d_th, d_v, NULL, d_cost, d_safe, NULL, NULL, NULL, NULL));  /* reduced outputs */
  CHECK_FO(fo_sweep_check(ctx, NULL));

  static double cost[M][FO_NC];
  static uint8_t safe[M];
  CHECK_HIP(hipMemcpy(cost, d_cost, sizeof cost, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(safe, d_safe, sizeof safe, hipMemcpyDeviceToHost));
  for (int m = 0; m < M; ++m)
    printf("%d %d %.17g %.17g %.17g %.17g %.17g\n", m, (int)safe[m], cost[m][FO_C_WTTC], cost[m][FO_C_MIN_DCE],
           cost[m][FO_C_MAX_OBST_RISK], cost[m][FO_C_MAX_OBST_HARM], cost[m][FO_C_MAX_CP]);
  fo_destroy(ctx);
  return 0;
}
