// smpc_oracle_host.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the host half of sortham::Optimizer — evalControl, prepare,
// fallback, reset, shiftControlSequence, getControlFromSequenceAsTwist,
// setSpeedLimit, setOffset, getOptimizedTrajectory (reference src/optimizer.cpp:
// 95-225, 345-360, 396-453) — on top of the oracle core (smpc_oracle.cpp).  The
// product's C++ host (mpcholonavigation_amd/host/optimizer.cpp) is checked
// against it tick by tick in tests/test_gpu_host_optimizer.py.
#include <cmath>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "smpc_oracle.h"

struct smpc_oracle_opt
{
  smpc_oracle * core = nullptr;
  smpc_config cfg{};
  float base[4] = {0, 0, 0, 0};         // base_constraints
  float cur[4] = {0, 0, 0, 0};          // constraints
  bool shift_control_sequence = false;
  size_t retry_attempt_limit = 0;
  size_t counter = 0;                   // static in the reference (H8)
  bool fail_flag = false;
  std::vector<float> u;                 // control_sequence_ {vx[T], vy[T], wz[T]}
  float history[12] = {0};              // control_history_
  float pose_yaw = 0;
  double pose_x = 0, pose_y = 0;
  smpc_tick_out last{};
  std::string err;
};

namespace
{
// Optimizer::reset (optimizer.cpp:116-132)
void opt_reset(smpc_oracle_opt * o)
{
  std::fill(o->u.begin(), o->u.end(), 0.0f);
  std::memset(o->history, 0, sizeof(o->history));
  std::memcpy(o->cur, o->base, sizeof(o->cur));
  smpc_oracle_reset(o->core);
}
}  // namespace

extern "C" {

int smpc_oracle_opt_create(
  const smpc_config * base, const smpc_critic_params * critics, double controller_frequency,
  uint32_t retry_attempt_limit, smpc_oracle_opt ** out)
{
  if (!base || !critics || !out) {return SMPC_ERR_INVALID;}
  auto * o = new (std::nothrow) smpc_oracle_opt();
  if (!o) {return SMPC_ERR_NOMEM;}
  o->cfg = *base;
  o->base[0] = base->vx_max;
  o->base[1] = base->vx_min;
  o->base[2] = base->vy_max;
  o->base[3] = base->wz_max;
  std::memcpy(o->cur, o->base, sizeof(o->cur));
  o->retry_attempt_limit = retry_attempt_limit;
  const int off = smpc_oracle_set_offset(controller_frequency, base->model_dt);   // :95-114
  if (off < 0) {
    delete o;
    return -10;  // "Controller period more then model dt, set it equal to model dt"
  }
  o->shift_control_sequence = off == 1;
  int rc = smpc_oracle_create(base, &o->core);
  if (rc != SMPC_OK) {
    delete o;
    return rc;
  }
  smpc_oracle_set_critics(o->core, critics);
  o->u.assign(3 * static_cast<size_t>(base->time_steps), 0.0f);
  *out = o;
  return SMPC_OK;
}

void smpc_oracle_opt_destroy(smpc_oracle_opt * o)
{
  if (o) {
    smpc_oracle_destroy(o->core);
    delete o;
  }
}

smpc_oracle * smpc_oracle_opt_core(smpc_oracle_opt * o) {return o ? o->core : nullptr;}
const char * smpc_oracle_opt_last_error(const smpc_oracle_opt * o) {return o ? o->err.c_str() : "";}

// Optimizer::evalControl (optimizer.cpp:134-155)
int smpc_oracle_opt_eval_control(
  smpc_oracle_opt * o, const smpc_tick_in * in, double * twist, smpc_tick_out * out)
{
  if (!o || !in || !twist) {return SMPC_ERR_INVALID;}
  const uint32_t T = o->cfg.time_steps;
  // prepare (:185-204)
  o->fail_flag = false;
  o->pose_x = in->pose_x;
  o->pose_y = in->pose_y;
  o->pose_yaw = in->pose_yaw;
  smpc_tick_in tick = *in;
  // do { optimize(); } while (fallback(fail_flag));
  for (;; ) {
    tick.fail_flag_in = o->fail_flag ? 1 : 0;
    smpc_oracle_set_constraints(o->core, o->cur[0], o->cur[1], o->cur[2], o->cur[3]);
    int rc = smpc_oracle_optimize(o->core, &tick, o->u.data(), &o->last);
    if (rc != SMPC_OK) {
      o->err = smpc_oracle_last_error(o->core);
      return rc;
    }
    o->fail_flag = o->last.fail_flag != 0;
    // fallback (:166-183)
    if (!o->fail_flag) {
      o->counter = 0;
      break;
    }
    opt_reset(o);
    if (++o->counter > o->retry_attempt_limit) {
      o->counter = 0;
      o->err = "Optimizer fail to compute path";
      if (out) {*out = o->last;}
      return -10;
    }
  }
  smpc_oracle_savitsky_golay(o->u.data(), T, o->history, o->shift_control_sequence ? 1 : 0);
  // getControlFromSequenceAsTwist (:396-410)
  const uint32_t offset = o->shift_control_sequence ? 1 : 0;
  twist[0] = o->u[offset];
  const bool holonomic = o->cfg.motion_model == SMPC_MODEL_OMNI;   // isHolonomic() (:235)
  twist[1] = holonomic ? o->u[T + offset] : 0.0;   // toTwistStamped(vx, wz, ...) (:409)
  twist[2] = o->u[2 * T + offset];
  if (o->shift_control_sequence) {
    // shiftControlSequence (:206-225) rolls vy only if holonomic
    std::vector<float> vy_row(o->u.begin() + T, o->u.begin() + 2 * T);
    smpc_oracle_shift_control_sequence(o->u.data(), T);
    if (!holonomic) {
      std::copy(vy_row.begin(), vy_row.end(), o->u.begin() + T);
    }
  }
  if (out) {*out = o->last;}
  return SMPC_OK;
}

int smpc_oracle_opt_set_speed_limit(smpc_oracle_opt * o, double speed_limit, int percentage)
{
  if (!o) {return SMPC_ERR_INVALID;}
  smpc_oracle_speed_limit(o->base, speed_limit, percentage, o->cur);
  return SMPC_OK;
}

int smpc_oracle_opt_reset(smpc_oracle_opt * o)
{
  if (!o) {return SMPC_ERR_INVALID;}
  opt_reset(o);
  return SMPC_OK;
}

int smpc_oracle_opt_get_control_sequence(smpc_oracle_opt * o, float * u)
{
  if (!o || !u) {return SMPC_ERR_INVALID;}
  std::memcpy(u, o->u.data(), o->u.size() * sizeof(float));
  return SMPC_OK;
}

int smpc_oracle_opt_set_control_sequence(smpc_oracle_opt * o, const float * u)
{
  if (!o || !u) {return SMPC_ERR_INVALID;}
  std::memcpy(o->u.data(), u, o->u.size() * sizeof(float));
  return SMPC_OK;
}

int smpc_oracle_opt_get_constraints(smpc_oracle_opt * o, float * c4, int32_t * shift)
{
  if (!o || !c4) {return SMPC_ERR_INVALID;}
  std::memcpy(c4, o->cur, sizeof(o->cur));
  if (shift) {*shift = o->shift_control_sequence ? 1 : 0;}
  return SMPC_OK;
}

// Optimizer::getOptimizedTrajectory + integrateStateVelocities(trajectory, sequence)
// (optimizer.cpp:345-360, 275-311)
int smpc_oracle_opt_get_optimized_trajectory(smpc_oracle_opt * o, float * xyyaw)
{
  if (!o || !xyyaw) {return SMPC_ERR_INVALID;}
  const uint32_t T = o->cfg.time_steps;
  const float dt = o->cfg.model_dt;
  const float * vx = o->u.data(), * vy = o->u.data() + T, * wz = o->u.data() + 2 * T;
  const float initial_yaw = o->pose_yaw;
  std::vector<float> yaws(T), yaw_cos(T), yaw_sin(T), dx(T), dy(T);
  float acc = 0.0f;
  for (uint32_t t = 0; t < T; ++t) {
    const float inc = wz[t] * dt;
    acc = t == 0 ? inc : acc + inc;
    yaws[t] = acc + initial_yaw;
  }
  yaw_cos[0] = cosf(initial_yaw);
  yaw_sin[0] = sinf(initial_yaw);
  for (uint32_t t = 1; t < T; ++t) {
    yaw_cos[t] = cosf(yaws[t - 1]);
    yaw_sin[t] = sinf(yaws[t - 1]);
  }
  for (uint32_t t = 0; t < T; ++t) {
    dx[t] = vx[t] * yaw_cos[t];
    dy[t] = vx[t] * yaw_sin[t];
    if (o->cfg.motion_model == SMPC_MODEL_OMNI) {   // :304-307
      dx[t] = dx[t] - vy[t] * yaw_sin[t];
      dy[t] = dy[t] + vy[t] * yaw_cos[t];
    }
  }
  float ax = 0.0f, ay = 0.0f;
  for (uint32_t t = 0; t < T; ++t) {
    ax = t == 0 ? dx[t] * dt : ax + dx[t] * dt;
    ay = t == 0 ? dy[t] * dt : ay + dy[t] * dt;
    xyyaw[3 * t] = static_cast<float>(o->pose_x + static_cast<double>(ax));
    xyyaw[3 * t + 1] = static_cast<float>(o->pose_y + static_cast<double>(ay));
    xyyaw[3 * t + 2] = yaws[t];
  }
  return SMPC_OK;
}

}  // extern "C"
