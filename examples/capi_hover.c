/* The C ABI of include/qd.h used from plain C with nothing but the HIP runtime: no Python, no PyTorch.
 *
 * 4096 drones with hanging loads (BASELINE config 3's observation / reward) are reset and then flown by the on-device PID
 * cascade (qd_rollout_pid) for 300 steps in ONE launch; the observations of the last step come back to the host and the
 * program checks that every drone is still within the truncation radius of its waypoint and no episode ended.
 *
 * Build (what tests/test_gpu_capi_example.py does):
 *   gcc -std=c99 -I include -I /opt/rocm/include examples/capi_hover.c -o capi_hover -L mujoco-drone_amd -lqd \
 *       -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/mujoco-drone_amd -Wl,-rpath,/opt/rocm/lib
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qd.h"

#define HIP_OK(x)                                                                          \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
  } while (0)
#define QD_OK_(x)                                                                    \
  do {                                                                               \
    int rc_ = (x);                                                                   \
    if (rc_ != QD_OK) { fprintf(stderr, "%s: %d %s\n", #x, rc_, qd_last_error()); return 3; } \
  } while (0)

int main(void) {
  const int n = 4096, T = 300;
  if (qd_version() != QD_VERSION) { fprintf(stderr, "header / library version mismatch\n"); return 1; }

  qd_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.num_envs = n; cfg.model = QD_MODEL_LOAD;
  cfg.obs_kind = QD_OBS_RPY_PARAMS; cfg.reward_kind = QD_REW_DISTANCE_ENERGY;
  cfg.frame_skip = 1; cfg.max_steps = 1024;
  cfg.ctrl_map = QD_CTRL_AFFINE; cfg.term_kind = QD_TERM_DEFAULT;
  cfg.random_start = QD_START_FIXED; cfg.random_params = 1; cfg.auto_reset = 1;
  cfg.timestep = 0.01; cfg.max_distance = 4.0;
  const double ref[4] = {0, 0, 15, 0}, center[6] = {1, 0.17, 7, 0.01, 1.2, 0.3}, width[6] = {0.1, 0.02, 1, 0.0025, 0.2, 0.05};
  memcpy(cfg.reference, ref, sizeof ref); memcpy(cfg.start_pos, ref, sizeof ref);
  memcpy(cfg.param_center, center, sizeof center); memcpy(cfg.param_width, width, sizeof width);
  cfg.param_difficulty = 1.0; cfg.seed = 42;
  const int D = qd_obs_dim(cfg.obs_kind, cfg.model);

  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  void* arena = NULL;
  float *obs = NULL, *reward = NULL, *obs0 = NULL;
  unsigned char* truncated = NULL;
  const size_t arena_bytes = qd_arena_bytes(n);
  HIP_OK(hipMalloc(&arena, arena_bytes));
  HIP_OK(hipMalloc((void**)&obs, (size_t)T * n * D * sizeof(float)));
  HIP_OK(hipMalloc((void**)&reward, (size_t)T * n * sizeof(float)));
  HIP_OK(hipMalloc((void**)&truncated, (size_t)T * n));
  HIP_OK(hipMalloc((void**)&obs0, (size_t)n * D * sizeof(float)));

  qd_env* env = NULL;
  QD_OK_(qd_create(&cfg, arena, arena_bytes, &env));
  QD_OK_(qd_init(env, stream));
  QD_OK_(qd_reset(env, NULL, obs0, stream));
  QD_OK_(qd_pid_reset(env, NULL, stream));
  QD_OK_(qd_rollout_pid(env, T, obs, reward, truncated, NULL, stream));
  HIP_OK(hipStreamSynchronize(stream));

  float* last = (float*)malloc((size_t)n * D * sizeof(float));
  unsigned char* tr = (unsigned char*)malloc((size_t)T * n);
  HIP_OK(hipMemcpy(last, obs + (size_t)(T - 1) * n * D, (size_t)n * D * sizeof(float), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(tr, truncated, (size_t)T * n, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int i = 0; i < n; i++) {  /* obs[0:3] = position error in the drone's frame (observation_wrappers.py) */
    const float* o = last + (size_t)i * D;
    const double d = sqrt((double)o[0] * o[0] + (double)o[1] * o[1] + (double)o[2] * o[2]);
    if (!(d == d)) { fprintf(stderr, "env %d: NaN\n", i); return 4; }
    if (d > worst) worst = d;
  }
  long ended = 0;
  for (size_t k = 0; k < (size_t)T * n; k++) ended += tr[k];
  printf("capi_hover: %d envs x %d steps, D=%d, worst |position error| %.3f m, episodes ended %ld\n", n, T, D, worst, ended);

  /* error behaviour of the reference: a wrong action count is refused with its message (mujoco_env_custom.py:200-201) */
  const int rc = qd_step(env, obs0, 4 * n - 1, obs0, reward, truncated, stream);
  if (rc != QD_ERR_SHAPE) { fprintf(stderr, "expected QD_ERR_SHAPE, got %d\n", rc); return 5; }
  printf("capi_hover: wrong action count -> %d \"%s\"\n", rc, qd_last_error());

  QD_OK_(qd_destroy(env));
  free(last); free(tr);
  hipFree(obs0); hipFree(truncated); hipFree(reward); hipFree(obs); hipFree(arena);
  hipStreamDestroy(stream);
  return (worst < cfg.max_distance && ended == 0) ? 0 : 6;
}
