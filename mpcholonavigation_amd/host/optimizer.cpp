// optimizer.cpp — see optimizer.hpp.  Host orchestration of one
// computeVelocityCommands() tick over the libsmpc C-ABI.
#include "optimizer.hpp"

#include <cmath>
#include <cstring>
#include <stdexcept>

namespace SORTHAM_HOST_NS
{

namespace
{
void ck(smpc_ctx * ctx, int rc, const char * what)
{
  if (rc != SMPC_OK) {
    throw std::runtime_error(std::string(what) + ": " + smpc_last_error(ctx));
  }
}
}  // namespace

Optimizer::~Optimizer() {shutdown();}

void Optimizer::shutdown()
{
  if (ctx_) {
    smpc_destroy(ctx_);
    ctx_ = nullptr;
  }
}

void Optimizer::setMotionModel(const std::string & model)
{
  // DiffDrive, Omni and Ackermann (optimizer.cpp:412-426).  Omni is the model the north star
  // names; the two non-holonomic ones run the same kernels with vy held at zero (smpc.h).
  if (model == "Omni") {
    motion_model_ = SMPC_MODEL_OMNI;
    return;
  }
  if (model == "DiffDrive") {
    motion_model_ = SMPC_MODEL_DIFF_DRIVE;
    return;
  }
  if (model == "Ackermann") {
    motion_model_ = SMPC_MODEL_ACKERMANN;
    return;
  }
  throw std::runtime_error(
          std::string(
            "Model " + model + " is not valid! Valid options are DiffDrive, Omni, "
            "or Ackermann"));
}

void Optimizer::setOffset(double controller_frequency)
{
  const double controller_period = 1.0 / controller_frequency;
  constexpr double eps = 1e-6;
  if ((controller_period + eps) < settings_.model_dt) {
    // reference warns: "Controller period is less then model dt, consider setting it equal"
  } else if (std::abs(controller_period - settings_.model_dt) < eps) {
    settings_.shift_control_sequence = true;
  } else {
    throw std::runtime_error("Controller period more then model dt, set it equal to model dt");
  }
}

void Optimizer::initialize(
  const models::OptimizerSettings & settings, const std::string & motion_model,
  double controller_frequency, const CriticsConfig & critics, bool regenerate_noises,
  uint64_t noise_seed, int device)
{
  // A call with the configuration the device context was built for (the controller's reset()
  // after every idle period, src/controller.cpp:89-92, re-reads unchanged parameters) keeps the
  // context: decided below, once the new smpc_config is known.
  smpc_ctx * keep = ctx_;
  ctx_ = nullptr;
  struct Guard
  {
    smpc_ctx *& k;
    ~Guard() {if (k) {smpc_destroy(k);}}   // a throwing initialize() leaves no context behind, as before
  } guard{keep};
  settings_ = settings;
  settings_.constraints = settings_.base_constraints;
  setMotionModel(motion_model);
  setOffset(controller_frequency);
  critics_ = critics;
  regenerate_noises_ = regenerate_noises;
  noise_seed_ = noise_seed;
  device_ = device;

  // CriticManager::loadCritics (critic_manager.cpp:42-60): the YAML list decides which
  // critics exist; the ones outside the fused set cannot be scored here
  auto & p = critics_.params;
  p.obstacles.enabled = p.path_align.enabled = p.path_follow.enabled = 0;
  p.goal_angle.enabled = p.prefer_forward.enabled = 0;
  p.cost.enabled = p.goal.enabled = p.constraint.enabled = p.twirling.enabled = 0;
  p.path_angle.enabled = p.velocity_deadband.enabled = p.path_align_legacy.enabled = 0;
  for (const auto & name : critics_.critics) {
    if (name == "ObstaclesCritic") {
      p.obstacles.enabled = 1;
    } else if (name == "PathAlignCritic") {
      p.path_align.enabled = 1;
    } else if (name == "PathFollowCritic") {
      p.path_follow.enabled = 1;
    } else if (name == "GoalAngleCritic") {
      p.goal_angle.enabled = 1;
    } else if (name == "PreferForwardCritic") {
      p.prefer_forward.enabled = 1;
    } else if (name == "CostCritic") {
      p.cost.enabled = 1;
    } else if (name == "GoalCritic") {
      p.goal.enabled = 1;
    } else if (name == "ConstraintCritic") {
      p.constraint.enabled = 1;
    } else if (name == "TwirlingCritic") {
      p.twirling.enabled = 1;
    } else if (name == "PathAngleCritic") {
      p.path_angle.enabled = 1;
    } else if (name == "VelocityDeadbandCritic") {
      p.velocity_deadband.enabled = 1;
    } else if (name == "PathAlignLegacyCritic") {
      p.path_align_legacy.enabled = 1;
    } else {
      // (critics.xml registers twelve classes and all twelve are fused: what is left is a name
      // pluginlib would not find either, critic_manager.cpp:45-57)
      throw std::runtime_error(
              "Critic sortham::critics::" + name + " is not one of the registered critic classes (critics.xml)");
    }
  }

  smpc_config cfg;
  std::memset(&cfg, 0, sizeof(cfg));   // (compared byte for byte below: no stray padding)
  smpc_config_default(&cfg);
  cfg.batch_size = settings_.batch_size;
  cfg.time_steps = settings_.time_steps;
  cfg.iteration_count = settings_.iteration_count;
  cfg.motion_model = motion_model_;
  cfg.ackermann_min_turning_r = ackermann_min_turning_r_;
  cfg.model_dt = settings_.model_dt;
  cfg.temperature = settings_.temperature;
  cfg.gamma = settings_.gamma;
  cfg.vx_max = settings_.base_constraints.vx_max;
  cfg.vx_min = settings_.base_constraints.vx_min;
  cfg.vy_max = settings_.base_constraints.vy;
  cfg.wz_max = settings_.base_constraints.wz;
  cfg.vx_std = settings_.sampling_std.vx;
  cfg.vy_std = settings_.sampling_std.vy;
  cfg.wz_std = settings_.sampling_std.wz;
  cfg.device = device_;
  cfg.flags = visualize_ ? SMPC_FLAG_STORE_TRAJECTORIES : 0u;
  if (keep && have_built_ && std::memcmp(&cfg, &built_cfg_, sizeof(cfg)) == 0 && noise_seed_ == built_seed_) {
    // same shapes, model, sampling and seed: only the critics' parameters can differ.  What the
    // reference's reset() does (src/optimizer.cpp:116-132) is reset() below, noise re-draw included.
    ctx_ = keep;
    keep = nullptr;
    ck(ctx_, smpc_set_critics(ctx_, &critics_.params), "smpc_set_critics");
    // as a fresh context: noise supplied through setNoise() does not survive initialize(), and the
    // draw starts from the seed's first epoch again (same seed, same noise as a new object)
    supplied_noise_ = false;
    ck(ctx_, smpc_seed(ctx_, noise_seed_), "smpc_seed");
    reset();
    return;
  }
  if (keep) {
    smpc_destroy(keep);
    keep = nullptr;
  }
  have_built_ = false;
  supplied_noise_ = false;
  int rc = smpc_create(&cfg, &ctx_);
  if (rc != SMPC_OK) {
    throw std::runtime_error(std::string("smpc_create: ") + smpc_last_error(nullptr));
  }
  built_cfg_ = cfg;
  built_seed_ = noise_seed_;
  have_built_ = true;
  ck(ctx_, smpc_set_critics(ctx_, &critics_.params), "smpc_set_critics");
  // NoiseGenerator::initialize draws once (noise_generator.cpp:26-42) ...
  ck(ctx_, smpc_seed(ctx_, noise_seed_), "smpc_seed");
  // ... and Optimizer::initialize ends in reset(), which draws again (H7)
  reset();
}

void Optimizer::pushConstraints()
{
  const auto & c = settings_.constraints;
  ck(ctx_, smpc_set_constraints(ctx_, c.vx_max, c.vx_min, c.vy, c.wz), "smpc_set_constraints");
}

void Optimizer::reset()
{
  control_sequence_.reset(settings_.time_steps);
  control_history_[0] = {0.0f, 0.0f, 0.0f};
  control_history_[1] = {0.0f, 0.0f, 0.0f};
  control_history_[2] = {0.0f, 0.0f, 0.0f};
  control_history_[3] = {0.0f, 0.0f, 0.0f};
  settings_.constraints = settings_.base_constraints;
  if (ctx_) {
    ck(ctx_, smpc_reset(ctx_), "smpc_reset");   // costs, noise re-draw (noise_generator.cpp:76-95)
  }
}

void Optimizer::setCostmap(const CostmapView & m)
{
  const bool infl = m.has_inflation_layer;
  ck(
    ctx_, smpc_set_costmap(
      ctx_, m.cells, m.size_x, m.size_y, m.origin_x, m.origin_y, m.resolution,
      m.track_unknown ? 1 : 0, m.inscribed_radius, infl ? critics_.cost_scaling_factor : 0.0f,
      infl ? critics_.inflation_radius : 0.0f), "smpc_set_costmap");
  if (!m.footprint_xy.empty()) {
    ck(
      ctx_, smpc_set_footprint(
        ctx_, m.footprint_xy.data(), static_cast<uint32_t>(m.footprint_xy.size() / 2),
        m.circumscribed_radius, infl ? m.layer_cost_scaling_factor : -1.0), "smpc_set_footprint");
  }
}

void Optimizer::setNoise(const float * nvx, const float * nvy, const float * nwz)
{
  ck(ctx_, smpc_set_noise(ctx_, nvx, nvy, nwz), "smpc_set_noise");
  supplied_noise_ = true;
}

void Optimizer::prepare(
  const Pose2D & robot_pose, const Twist2D & robot_speed, const models::Path & plan,
  const Pose2D & goal)
{
  pose_ = robot_pose;
  speed_ = robot_speed;
  path_ = plan;
  goal_ = goal;
  fail_flag_ = false;   // costs_.fill(0) and the CriticData caches are per smpc_optimize call
}

void Optimizer::optimize()
{
  smpc_tick_in in;
  std::memset(&in, 0, sizeof(in));
  in.pose_x = pose_.x;
  in.pose_y = pose_.y;
  in.pose_yaw = static_cast<float>(pose_.yaw);   // float initial_yaw = tf2::getYaw(...)
  in.speed_vx = speed_.vx;
  in.speed_vy = speed_.vy;
  in.speed_wz = speed_.wz;
  in.path_x = path_.x.data();
  in.path_y = path_.y.data();
  in.path_yaw = path_.yaws.data();
  in.path_len = static_cast<uint32_t>(path_.x.size());
  in.goal_x = goal_.x;
  in.goal_y = goal_.y;
  in.path_pts_valid = nullptr;
  in.fail_flag_in = fail_flag_ ? 1 : 0;   // sticky across the retry (critic_manager.cpp:70-73)
  in.goal_checker_xy_tolerance = goal_checker_xy_tolerance_;

  const unsigned int T = settings_.time_steps;
  std::vector<float> u(3 * T);
  std::memcpy(u.data(), control_sequence_.vx.data(), T * sizeof(float));
  std::memcpy(u.data() + T, control_sequence_.vy.data(), T * sizeof(float));
  std::memcpy(u.data() + 2 * T, control_sequence_.wz.data(), T * sizeof(float));
  pushConstraints();
  ck(ctx_, smpc_optimize(ctx_, &in, u.data(), &last_out_), "smpc_optimize");
  std::memcpy(control_sequence_.vx.data(), u.data(), T * sizeof(float));
  std::memcpy(control_sequence_.vy.data(), u.data() + T, T * sizeof(float));
  std::memcpy(control_sequence_.wz.data(), u.data() + 2 * T, T * sizeof(float));
  fail_flag_ = last_out_.fail_flag != 0;
  if (regenerate_noises_ && !supplied_noise_) {
    // NoiseGenerator::generateNextNoises (noise_generator.cpp:54-63): next tick's noise, drawn
    // while this tick's result travels on (the reference's noise thread, :97-105)
    ck(ctx_, smpc_redraw_noise_async(ctx_), "smpc_redraw_noise_async");
  }
}

bool Optimizer::fallback(bool fail)
{
  if (!fail) {
    retry_counter_ = 0;
    return false;
  }
  reset();
  if (++retry_counter_ > settings_.retry_attempt_limit) {
    retry_counter_ = 0;
    throw std::runtime_error("Optimizer fail to compute path");
  }
  return true;
}

Twist2D Optimizer::evalControl(
  const Pose2D & robot_pose, const Twist2D & robot_speed, const models::Path & plan,
  const Pose2D & goal, float goal_checker_xy_tolerance)
{
  goal_checker_xy_tolerance_ = goal_checker_xy_tolerance;
  prepare(robot_pose, robot_speed, plan, goal);
  do {
    optimize();
  } while (fallback(fail_flag_));
  utils::savitskyGolayFilter(control_sequence_, control_history_, settings_);
  auto control = getControlFromSequenceAsTwist();
  if (settings_.shift_control_sequence) {
    shiftControlSequence();
  }
  return control;
}

void Optimizer::shiftControlSequence()
{
  auto roll = [](std::vector<float> & v) {
      if (v.size() < 2) {
        return;
      }
      for (size_t i = 0; i + 1 < v.size(); ++i) {
        v[i] = v[i + 1];
      }
      v[v.size() - 1] = v[v.size() - 2];
    };
  roll(control_sequence_.vx);
  roll(control_sequence_.wz);
  if (isHolonomic()) {
    roll(control_sequence_.vy);
  }
}

Twist2D Optimizer::getControlFromSequenceAsTwist()
{
  const unsigned int offset = settings_.shift_control_sequence ? 1 : 0;
  Twist2D t;
  t.vx = control_sequence_.vx.at(offset);
  t.wz = control_sequence_.wz.at(offset);
  // toTwistStamped(vx, wz, ...) leaves linear.y at 0 for a non-holonomic model (:404-409)
  t.vy = isHolonomic() ? control_sequence_.vy.at(offset) : 0.0f;
  return t;
}

void Optimizer::setSpeedLimit(double speed_limit, bool percentage)
{
  auto & s = settings_;
  constexpr double NO_SPEED_LIMIT = 0.0;   // nav2_costmap_2d filter_values.hpp
  if (speed_limit == NO_SPEED_LIMIT) {
    s.constraints = s.base_constraints;
  } else {
    const double ratio = percentage ? speed_limit / 100.0 : speed_limit / s.base_constraints.vx_max;
    s.constraints.vx_max = s.base_constraints.vx_max * ratio;
    s.constraints.vx_min = s.base_constraints.vx_min * ratio;
    s.constraints.vy = s.base_constraints.vy * ratio;
    s.constraints.wz = s.base_constraints.wz * ratio;
  }
}

std::vector<std::array<float, 3>> Optimizer::getOptimizedTrajectory()
{
  // integrateStateVelocities(trajectory, sequence) — optimizer.cpp:275-311: the control
  // sequence itself is integrated (no measured-speed first column here)
  const unsigned int T = settings_.time_steps;
  const float dt = settings_.model_dt;
  const float initial_yaw = static_cast<float>(pose_.yaw);
  std::vector<std::array<float, 3>> traj(T);
  std::vector<float> yaws(T);
  float acc = 0.0f;
  for (unsigned int t = 0; t < T; ++t) {
    const float inc = control_sequence_.wz[t] * dt;
    acc = t == 0 ? inc : acc + inc;
    yaws[t] = acc + initial_yaw;
  }
  float ax = 0.0f, ay = 0.0f;
  for (unsigned int t = 0; t < T; ++t) {
    const float c = t == 0 ? cosf(initial_yaw) : cosf(yaws[t - 1]);
    const float s = t == 0 ? sinf(initial_yaw) : sinf(yaws[t - 1]);
    float dx = control_sequence_.vx[t] * c;
    float dy = control_sequence_.vx[t] * s;
    if (isHolonomic()) {
      dx = dx - control_sequence_.vy[t] * s;
      dy = dy + control_sequence_.vy[t] * c;
    }
    ax = t == 0 ? dx * dt : ax + dx * dt;
    ay = t == 0 ? dy * dt : ay + dy * dt;
    traj[t] = {static_cast<float>(pose_.x + static_cast<double>(ax)),
      static_cast<float>(pose_.y + static_cast<double>(ay)), yaws[t]};
  }
  return traj;
}

void Optimizer::getGeneratedTrajectories(
  std::vector<float> & x, std::vector<float> & y, std::vector<float> & yaws)
{
  const size_t n = static_cast<size_t>(settings_.batch_size) * settings_.time_steps;
  x.resize(n);
  y.resize(n);
  yaws.resize(n);
  ck(ctx_, smpc_get_trajectories(ctx_, x.data(), y.data(), yaws.data()), "smpc_get_trajectories");
}

namespace utils
{

void savitskyGolayFilter(
  models::ControlSequence & control_sequence, std::array<models::Control, 4> & control_history,
  const models::OptimizerSettings & settings)
{
  // Savitzky-Golay quadratic, 9 points: {-21,14,39,54,59,54,39,14,-21}/231, applied in
  // place front to back; the first four outputs lean on the last four executed controls,
  // the tail repeats the last sample; index T-5 is left as is (the reference's loop ends
  // one short and then steps over it, tools/utils.hpp:518-533)
  const unsigned int n = static_cast<unsigned int>(control_sequence.vx.size());
  if (n == 0 || n - 1 < 20) {
    return;
  }
  const unsigned int last = n - 1;
  float w[9] = {-21.0f, 14.0f, 39.0f, 54.0f, 59.0f, 54.0f, 39.0f, 14.0f, -21.0f};
  for (float & c : w) {
    c /= 231.0f;
  }
  auto run = [&](std::vector<float> & q, float h0, float h1, float h2, float h3) {
      const float hist[4] = {h0, h1, h2, h3};
      // sample at signed position i: history for i < 0, clamped to `last` beyond the end
      auto at = [&](int i) -> float {
          if (i < 0) {
            return hist[4 + i];
          }
          return q[static_cast<unsigned int>(i) > last ? last : static_cast<unsigned int>(i)];
        };
      for (unsigned int idx = 0; idx <= last; ++idx) {
        if (idx == last - 4) {
          continue;
        }
        float acc = 0.0f;
        for (int k = 0; k < 9; ++k) {
          acc += at(static_cast<int>(idx) + k - 4) * w[k];
        }
        q[idx] = acc;
      }
    };
  const auto & h = control_history;
  run(control_sequence.vx, h[0].vx, h[1].vx, h[2].vx, h[3].vx);
  run(control_sequence.vy, h[0].vy, h[1].vy, h[2].vy, h[3].vy);
  run(control_sequence.wz, h[0].wz, h[1].wz, h[2].wz, h[3].wz);
  const unsigned int offset = settings.shift_control_sequence ? 1 : 0;
  control_history[0] = control_history[1];
  control_history[1] = control_history[2];
  control_history[2] = control_history[3];
  control_history[3] = {control_sequence.vx[offset], control_sequence.vy[offset],
    control_sequence.wz[offset]};
}

}  // namespace utils
}  // namespace SORTHAM_HOST_NS
