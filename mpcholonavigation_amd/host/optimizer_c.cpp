// optimizer_c.cpp — extern "C" face of sortham_ns::Optimizer (include/smpc_host.h).
#include <cstring>
#include <exception>
#include <new>
#include <string>

#include "../../include/smpc_host.h"
#include "optimizer.hpp"

namespace sortham_ns = SORTHAM_HOST_NS;

struct sortham_optimizer
{
  sortham_ns::Optimizer opt;
  std::string err;
};

namespace
{
thread_local std::string g_err;

template<typename F>
int guarded(sortham_optimizer * o, F && f)
{
  try {
    f();
    return SMPC_OK;
  } catch (const std::exception & e) {
    if (o) {
      o->err = e.what();
    } else {
      g_err = e.what();
    }
    return SORTHAM_ERR_THROWN;
  }
}
}  // namespace

namespace {
// the parameter reads of the plugin's Optimizer::getParams() + reset() (nav2_plugin/src/optimizer.cpp)
void initialize_from(sortham_optimizer * o, const sortham_optimizer_config * cfg, const smpc_critic_params * critics)
{
  sortham_ns::models::OptimizerSettings s;
  const smpc_config & b = cfg->base;
  s.base_constraints = {b.vx_max, b.vx_min, b.vy_max, b.wz_max};
  s.sampling_std = {b.vx_std, b.vy_std, b.wz_std};
  s.model_dt = b.model_dt;
  s.temperature = b.temperature;
  s.gamma = b.gamma;
  s.batch_size = b.batch_size;
  s.time_steps = b.time_steps;
  s.iteration_count = b.iteration_count;
  s.retry_attempt_limit = cfg->retry_attempt_limit;
  sortham_ns::CriticsConfig cc;
  cc.params = *critics;
  for (uint32_t i = 0; i < cfg->n_critics && i < 16; ++i) {
    cc.critics.emplace_back(cfg->critics[i] ? cfg->critics[i] : "");
  }
  cc.cost_scaling_factor = cfg->cost_scaling_factor;
  cc.inflation_radius = cfg->inflation_radius;
  o->opt.setVisualize(cfg->visualize != 0);
  o->opt.setAckermannMinTurningRadius(b.ackermann_min_turning_r);
  o->opt.initialize(
    s, cfg->motion_model ? cfg->motion_model : "DiffDrive", cfg->controller_frequency, cc,
    cfg->regenerate_noises != 0, cfg->noise_seed, b.device);
}
}  // namespace

extern "C" {

int sortham_optimizer_create(
  const sortham_optimizer_config * cfg, const smpc_critic_params * critics,
  sortham_optimizer ** out)
{
  if (!cfg || !critics || !out) {
    return SMPC_ERR_INVALID;
  }
  *out = nullptr;
  auto * o = new (std::nothrow) sortham_optimizer();
  if (!o) {
    return SMPC_ERR_NOMEM;
  }
  int rc = guarded(nullptr, [&]() {initialize_from(o, cfg, critics);});
  if (rc != SMPC_OK) {
    delete o;
    return rc;
  }
  *out = o;
  return SMPC_OK;
}

int sortham_optimizer_initialize(
  sortham_optimizer * o, const sortham_optimizer_config * cfg, const smpc_critic_params * critics)
{
  if (!o || !cfg || !critics) {
    return SMPC_ERR_INVALID;
  }
  return guarded(o, [&]() {initialize_from(o, cfg, critics);});
}

void sortham_optimizer_destroy(sortham_optimizer * o) {delete o;}

const char * sortham_optimizer_last_error(const sortham_optimizer * o)
{
  return o ? o->err.c_str() : g_err.c_str();
}

int sortham_optimizer_set_costmap(
  sortham_optimizer * o, const uint8_t * cells, uint32_t width, uint32_t height, double origin_x,
  double origin_y, double resolution, int track_unknown, float inscribed_radius,
  int has_inflation_layer)
{
  if (!o) {return SMPC_ERR_INVALID;}
  return guarded(
    o, [&]() {
      sortham_ns::CostmapView m;
      m.cells = cells;
      m.size_x = width;
      m.size_y = height;
      m.origin_x = origin_x;
      m.origin_y = origin_y;
      m.resolution = resolution;
      m.track_unknown = track_unknown != 0;
      m.inscribed_radius = inscribed_radius;
      m.has_inflation_layer = has_inflation_layer != 0;
      o->opt.setCostmap(m);
    });
}

int sortham_optimizer_set_noise(
  sortham_optimizer * o, const float * nvx, const float * nvy, const float * nwz)
{
  if (!o) {return SMPC_ERR_INVALID;}
  return guarded(o, [&]() {o->opt.setNoise(nvx, nvy, nwz);});
}

int sortham_optimizer_eval_control(
  sortham_optimizer * o, const smpc_tick_in * in, double * twist_out, smpc_tick_out * out)
{
  if (!o || !in || !twist_out) {return SMPC_ERR_INVALID;}
  return guarded(
    o, [&]() {
      sortham_ns::Pose2D pose{in->pose_x, in->pose_y, static_cast<double>(in->pose_yaw)};
      sortham_ns::Pose2D goal{in->goal_x, in->goal_y, 0.0};
      sortham_ns::Twist2D speed{in->speed_vx, in->speed_vy, in->speed_wz};
      sortham_ns::models::Path plan;
      plan.x.assign(in->path_x, in->path_x + in->path_len);
      plan.y.assign(in->path_y, in->path_y + in->path_len);
      plan.yaws.assign(in->path_yaw, in->path_yaw + in->path_len);
      const sortham_ns::Twist2D t =
        o->opt.evalControl(pose, speed, plan, goal, in->goal_checker_xy_tolerance);
      twist_out[0] = t.vx;
      twist_out[1] = t.vy;
      twist_out[2] = t.wz;
      if (out) {*out = o->opt.lastTick();}
    });
}

int sortham_optimizer_set_speed_limit(sortham_optimizer * o, double speed_limit, int percentage)
{
  if (!o) {return SMPC_ERR_INVALID;}
  return guarded(o, [&]() {o->opt.setSpeedLimit(speed_limit, percentage != 0);});
}

int sortham_optimizer_reset(sortham_optimizer * o)
{
  if (!o) {return SMPC_ERR_INVALID;}
  return guarded(o, [&]() {o->opt.reset();});
}

int sortham_optimizer_get_control_sequence(sortham_optimizer * o, float * u)
{
  if (!o || !u) {return SMPC_ERR_INVALID;}
  const auto & cs = o->opt.controlSequence();
  const size_t T = cs.vx.size();
  std::memcpy(u, cs.vx.data(), T * sizeof(float));
  std::memcpy(u + T, cs.vy.data(), T * sizeof(float));
  std::memcpy(u + 2 * T, cs.wz.data(), T * sizeof(float));
  return SMPC_OK;
}

int sortham_optimizer_set_control_sequence(sortham_optimizer * o, const float * u)
{
  if (!o || !u) {return SMPC_ERR_INVALID;}
  auto & cs = o->opt.controlSequence();
  const size_t T = cs.vx.size();
  std::memcpy(cs.vx.data(), u, T * sizeof(float));
  std::memcpy(cs.vy.data(), u + T, T * sizeof(float));
  std::memcpy(cs.wz.data(), u + 2 * T, T * sizeof(float));
  return SMPC_OK;
}

int sortham_optimizer_get_constraints(sortham_optimizer * o, float * c4, int32_t * shift)
{
  if (!o || !c4) {return SMPC_ERR_INVALID;}
  const auto & s = o->opt.settings();
  c4[0] = s.constraints.vx_max;
  c4[1] = s.constraints.vx_min;
  c4[2] = s.constraints.vy;
  c4[3] = s.constraints.wz;
  if (shift) {*shift = s.shift_control_sequence ? 1 : 0;}
  return SMPC_OK;
}

int sortham_optimizer_get_optimized_trajectory(sortham_optimizer * o, float * xyyaw)
{
  if (!o || !xyyaw) {return SMPC_ERR_INVALID;}
  return guarded(
    o, [&]() {
      const auto tr = o->opt.getOptimizedTrajectory();
      for (size_t t = 0; t < tr.size(); ++t) {
        xyyaw[3 * t] = tr[t][0];
        xyyaw[3 * t + 1] = tr[t][1];
        xyyaw[3 * t + 2] = tr[t][2];
      }
    });
}

void sortham_utils_savitsky_golay(float * u, uint32_t T, float * history, int shift)
{
  sortham_ns::models::ControlSequence cs;
  cs.vx.assign(u, u + T);
  cs.vy.assign(u + T, u + 2 * T);
  cs.wz.assign(u + 2 * T, u + 3 * T);
  std::array<sortham_ns::models::Control, 4> h;
  for (int i = 0; i < 4; ++i) {
    h[i] = {history[3 * i], history[3 * i + 1], history[3 * i + 2]};
  }
  sortham_ns::models::OptimizerSettings s;
  s.shift_control_sequence = shift != 0;
  sortham_ns::utils::savitskyGolayFilter(cs, h, s);
  std::memcpy(u, cs.vx.data(), T * sizeof(float));
  std::memcpy(u + T, cs.vy.data(), T * sizeof(float));
  std::memcpy(u + 2 * T, cs.wz.data(), T * sizeof(float));
  for (int i = 0; i < 4; ++i) {
    history[3 * i] = h[i].vx;
    history[3 * i + 1] = h[i].vy;
    history[3 * i + 2] = h[i].wz;
  }
}

}  // extern "C"
