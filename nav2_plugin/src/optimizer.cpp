// optimizer.cpp (Nav2 build) — ROS-typed adaptor over sortham_host::Optimizer.
// Compiled only inside a ROS 2 Humble + Nav2 workspace; not built by this repository's
// tests (no ROS here) — see INTEGRATION.md for the exact file swap.
#include "nav2_sortham_controller/optimizer.hpp"

#include <stdexcept>

#include "nav2_costmap_2d/inflation_layer.hpp"
#include "nav2_sortham_controller/fused_critic_registry.hpp"
#include "tf2/utils.h"

namespace sortham
{

void Optimizer::initialize(
  rclcpp_lifecycle::LifecycleNode::WeakPtr parent, const std::string & name,
  std::shared_ptr<nav2_costmap_2d::Costmap2DROS> costmap_ros, ParametersHandler * param_handler)
{
  parent_ = parent;
  name_ = name;
  costmap_ros_ = costmap_ros;
  parameters_handler_ = param_handler;
  logger_ = parent_.lock()->get_logger();
  getParams();
  // pluginlib loads "sortham::critics::<Name>" for every entry of the YAML `critics` list
  // (critic_manager.cpp:42-65); the fused critic classes publish their parameters
  critic_manager_.on_configure(parent_, name_, costmap_ros_, parameters_handler_);
  reset();
}

void Optimizer::getParams()
{
  auto & s = settings_;
  auto getParam = parameters_handler_->getParamGetter(name_);
  auto getParentParam = parameters_handler_->getParamGetter("");
  // names and defaults of the reference (src/optimizer.cpp:69-91)
  getParam(s.model_dt, "model_dt", 0.05f);
  getParam(s.time_steps, "time_steps", 56);
  getParam(s.batch_size, "batch_size", 1000);
  getParam(s.iteration_count, "iteration_count", 1);
  getParam(s.temperature, "temperature", 0.3f);
  getParam(s.gamma, "gamma", 0.015f);
  getParam(s.base_constraints.vx_max, "vx_max", 0.5);
  getParam(s.base_constraints.vx_min, "vx_min", -0.35);
  getParam(s.base_constraints.vy, "vy_max", 0.5);
  getParam(s.base_constraints.wz, "wz_max", 1.9);
  getParam(s.sampling_std.vx, "vx_std", 0.2);
  getParam(s.sampling_std.vy, "vy_std", 0.2);
  getParam(s.sampling_std.wz, "wz_std", 0.4);
  getParam(s.retry_attempt_limit, "retry_attempt_limit", 1);
  getParam(motion_model_name_, "motion_model", std::string("DiffDrive"));
  getParam(regenerate_noises_, "regenerate_noises", false);
  getParam(visualize_, "visualize", false);
  parameters_handler_->addPostCallback([this]() {reset();});   // :88
  getParentParam(controller_frequency_, "controller_frequency", 0.0, ParameterType::Static);
}

void Optimizer::reset()
{
  // the reference's reset() re-allocates its tensors (src/optimizer.cpp:116-132); here
  // host_.initialize() keeps the device context when the parameters it was built from are
  // unchanged (the controller resets after every idle period) and rebuilds it otherwise
  std::vector<std::string> names;
  auto getParam = parameters_handler_->getParamGetter(name_);
  getParam(names, "critics", std::vector<std::string>{}, ParameterType::Static);
  auto & reg = critics::FusedCriticRegistry::get().entry(name_);
  sortham_host::CriticsConfig cc;
  cc.critics = names;
  cc.params = reg.params;
  cc.cost_scaling_factor = reg.cost_scaling_factor;
  cc.inflation_radius = reg.inflation_radius;
  host_.setVisualize(visualize_);
  if (motion_model_name_ == "Ackermann") {
    // AckermannMotionModel's constructor (include/.../motion_models.hpp:91-95)
    auto getAcker = parameters_handler_->getParamGetter(name_ + ".AckermannConstraints");
    float min_turning_r = 0.2f;
    getAcker(min_turning_r, "min_turning_r", 0.2);
    host_.setAckermannMinTurningRadius(min_turning_r);
  }
  host_.initialize(settings_, motion_model_name_, controller_frequency_, cc, regenerate_noises_);
  generated_trajectories_.reset(settings_.batch_size, settings_.time_steps);
  RCLCPP_INFO(logger_, "Optimizer reset");
}

void Optimizer::shutdown() {host_.shutdown();}

void Optimizer::uploadCostmap()
{
  auto * cm = costmap_ros_->getCostmap();   // the caller holds its mutex (controller.cpp:99-100)
  auto * layered = costmap_ros_->getLayeredCostmap();
  sortham_host::CostmapView v;
  v.cells = cm->getCharMap();
  v.size_x = cm->getSizeInCellsX();
  v.size_y = cm->getSizeInCellsY();
  v.origin_x = cm->getOriginX();
  v.origin_y = cm->getOriginY();
  v.resolution = cm->getResolution();
  v.track_unknown = layered->isTrackingUnknown();
  v.inscribed_radius = layered->getInscribedRadius();
  for (auto & layer : *layered->getPlugins()) {   // obstacles_critic.cpp:66-74
    if (auto inflation = std::dynamic_pointer_cast<nav2_costmap_2d::InflationLayer>(layer)) {
      v.has_inflation_layer = true;
      v.layer_cost_scaling_factor = inflation->getCostScalingFactor();
    }
  }
  // consider_footprint = true (obstacles_critic.cpp:215-221, cost_critic.cpp:181-186)
  for (const auto & pt : costmap_ros_->getRobotFootprint()) {
    v.footprint_xy.push_back(pt.x);
    v.footprint_xy.push_back(pt.y);
  }
  v.circumscribed_radius = layered->getCircumscribedRadius();
  host_.setCostmap(v);
}

geometry_msgs::msg::TwistStamped Optimizer::evalControl(
  const geometry_msgs::msg::PoseStamped & robot_pose, const geometry_msgs::msg::Twist & robot_speed,
  const nav_msgs::msg::Path & plan, const geometry_msgs::msg::Pose & goal,
  nav2_core::GoalChecker * goal_checker)
{
  uploadCostmap();
  // utils::withinPositionGoalTolerance(goal_checker, ...) (tools/utils.hpp:201-224) needs only
  // the checker's xy tolerance: TwirlingCritic's gate
  float goal_checker_xy_tolerance = -1.0f;
  if (goal_checker) {
    geometry_msgs::msg::Pose pose_tolerance;
    geometry_msgs::msg::Twist velocity_tolerance;
    goal_checker->getTolerances(pose_tolerance, velocity_tolerance);
    goal_checker_xy_tolerance = static_cast<float>(pose_tolerance.position.x);
  }
  sortham_host::Pose2D pose{robot_pose.pose.position.x, robot_pose.pose.position.y,
    tf2::getYaw(robot_pose.pose.orientation)};
  sortham_host::Pose2D g{goal.position.x, goal.position.y, tf2::getYaw(goal.orientation)};
  sortham_host::Twist2D speed{robot_speed.linear.x, robot_speed.linear.y, robot_speed.angular.z};
  sortham_host::models::Path path;   // utils::toTensor (tools/utils.hpp:180-192)
  const size_t n = plan.poses.size();
  path.x.resize(n);
  path.y.resize(n);
  path.yaws.resize(n);
  for (size_t i = 0; i < n; ++i) {
    path.x[i] = plan.poses[i].pose.position.x;
    path.y[i] = plan.poses[i].pose.position.y;
    path.yaws[i] = tf2::getYaw(plan.poses[i].pose.orientation);
  }
  const sortham_host::Twist2D t =
    host_.evalControl(pose, speed, path, g, goal_checker_xy_tolerance);   // may throw
  geometry_msgs::msg::TwistStamped twist;   // utils::toTwistStamped (tools/utils.hpp:145-173)
  twist.header.frame_id = costmap_ros_->getBaseFrameID();
  twist.header.stamp = plan.header.stamp;
  twist.twist.linear.x = t.vx;
  twist.twist.linear.y = t.vy;
  twist.twist.angular.z = t.wz;
  return twist;
}

models::Trajectories & Optimizer::getGeneratedTrajectories()
{
  std::vector<float> x, y, yaws;
  host_.getGeneratedTrajectories(x, y, yaws);
  const size_t B = settings_.batch_size, T = settings_.time_steps;
  for (size_t b = 0; b < B; ++b) {
    for (size_t t = 0; t < T; ++t) {
      generated_trajectories_.x(b, t) = x[b * T + t];
      generated_trajectories_.y(b, t) = y[b * T + t];
      generated_trajectories_.yaws(b, t) = yaws[b * T + t];
    }
  }
  return generated_trajectories_;
}

xt::xtensor<float, 2> Optimizer::getOptimizedTrajectory()
{
  const auto tr = host_.getOptimizedTrajectory();
  auto out = xt::xtensor<float, 2>::from_shape({tr.size(), 3});
  for (size_t t = 0; t < tr.size(); ++t) {
    out(t, 0) = tr[t][0];
    out(t, 1) = tr[t][1];
    out(t, 2) = tr[t][2];
  }
  return out;
}

void Optimizer::setSpeedLimit(double speed_limit, bool percentage)
{
  host_.setSpeedLimit(speed_limit, percentage);
}

}  // namespace sortham
