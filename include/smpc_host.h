/*
 * smpc_host.h — C face of the C++ host optimizer (mpcholonavigation_amd/host/
 * optimizer.hpp, class sortham::Optimizer).  It exists so that tests (and
 * non-C++ callers) can drive the host logic that stays on the CPU in the
 * reference — Optimizer::evalControl and friends [ref src/optimizer.cpp:116-225,
 * 396-453] — through plain C.  A Nav2 build links the C++ class directly
 * (INTEGRATION.md).  Exceptions the reference throws become
 * SORTHAM_ERR_THROWN with the message in sortham_optimizer_last_error().
 */
#ifndef SMPC_HOST_H_
#define SMPC_HOST_H_

#include "smpc.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SORTHAM_ERR_THROWN -10 /* std::runtime_error where the reference throws */

typedef struct sortham_optimizer sortham_optimizer;

typedef struct sortham_optimizer_config {
  smpc_config base;             /* batch/horizon/iterations, dt, temperature, gamma,
                                   base constraints, sampling std, device          */
  double controller_frequency;  /* setOffset [ref src/optimizer.cpp:95-114]         */
  uint32_t retry_attempt_limit; /* [ref :82]                                        */
  int32_t regenerate_noises;    /* [ref src/noise_generator.cpp:35]                 */
  int32_t visualize;            /* keep generated trajectories readable             */
  uint64_t noise_seed;
  /* YAML `critics` list: up to 16 names without the namespace, e.g.
   * "ObstaclesCritic" [ref src/critic_manager.cpp:36-65]                          */
  const char* critics[16];
  uint32_t n_critics;
  float cost_scaling_factor, inflation_radius; /* ObstaclesCritic params           */
  const char* motion_model;     /* "Omni" [ref :84]                                 */
} sortham_optimizer_config;

int sortham_optimizer_create(const sortham_optimizer_config* cfg, const smpc_critic_params* critics,
                             sortham_optimizer** out);
void sortham_optimizer_destroy(sortham_optimizer* o);
const char* sortham_optimizer_last_error(const sortham_optimizer* o);

int sortham_optimizer_set_costmap(sortham_optimizer* o, const uint8_t* cells, uint32_t width,
                                  uint32_t height, double origin_x, double origin_y,
                                  double resolution, int track_unknown, float inscribed_radius,
                                  int has_inflation_layer);
int sortham_optimizer_set_noise(sortham_optimizer* o, const float* nvx, const float* nvy,
                                const float* nwz);
/* Optimizer::evalControl: in carries pose / speed / plan / goal (fail_flag_in ignored);
 * twist_out = {linear.x, linear.y, angular.z}. */
int sortham_optimizer_eval_control(sortham_optimizer* o, const smpc_tick_in* in, double* twist_out,
                                   smpc_tick_out* out);
int sortham_optimizer_set_speed_limit(sortham_optimizer* o, double speed_limit, int percentage);
int sortham_optimizer_reset(sortham_optimizer* o);
/* control_sequence_ as {vx[T], vy[T], wz[T]} */
int sortham_optimizer_get_control_sequence(sortham_optimizer* o, float* u);
int sortham_optimizer_set_control_sequence(sortham_optimizer* o, const float* u);
/* settings_.constraints = {vx_max, vx_min, vy, wz}; shift flag */
int sortham_optimizer_get_constraints(sortham_optimizer* o, float* c4, int32_t* shift_control_sequence);
/* getOptimizedTrajectory: xyyaw = T x 3 */
int sortham_optimizer_get_optimized_trajectory(sortham_optimizer* o, float* xyyaw);

/* utils::savitskyGolayFilter [ref tools/utils.hpp:442-605] on u = {vx[T], vy[T], wz[T]};
 * history = 4 x {vx, vy, wz}, oldest first.  No GPU involved. */
void sortham_utils_savitsky_golay(float* u, uint32_t T, float* history, int shift_control_sequence);

#ifdef __cplusplus
}
#endif
#endif
