// optimizer.hpp (Nav2 build) — drop-in replacement for the reference header of the same
// name (include/nav2_sortham_controller/optimizer.hpp:51-263): the same public surface in
// ROS types, so the reference's controller.cpp / trajectory_visualizer.cpp compile
// unchanged; the work goes to sortham_host::Optimizer -> libsmpc (MI355X).
// Compiled only inside a ROS 2 Humble + Nav2 workspace — see INTEGRATION.md.
#ifndef NAV2_SORTHAM_CONTROLLER__OPTIMIZER_HPP_
#define NAV2_SORTHAM_CONTROLLER__OPTIMIZER_HPP_

#include <memory>
#include <string>
#include <vector>

#include <xtensor/xtensor.hpp>

#include "rclcpp_lifecycle/lifecycle_node.hpp"
#include "nav2_costmap_2d/costmap_2d_ros.hpp"
#include "nav2_core/goal_checker.hpp"
#include "geometry_msgs/msg/twist.hpp"
#include "geometry_msgs/msg/pose_stamped.hpp"
#include "geometry_msgs/msg/twist_stamped.hpp"
#include "nav_msgs/msg/path.hpp"

#include "nav2_sortham_controller/models/trajectories.hpp"       // reference, unchanged
#include "nav2_sortham_controller/critic_manager.hpp"            // reference, unchanged
#include "nav2_sortham_controller/tools/parameters_handler.hpp"  // reference, unchanged

#define SORTHAM_HOST_NS sortham_host
#include "host/optimizer.hpp"   // mpcholonavigation_amd/host/optimizer.hpp

namespace sortham
{

class Optimizer
{
public:
  Optimizer() = default;
  ~Optimizer() {shutdown();}

  void initialize(
    rclcpp_lifecycle::LifecycleNode::WeakPtr parent, const std::string & name,
    std::shared_ptr<nav2_costmap_2d::Costmap2DROS> costmap_ros,
    ParametersHandler * dynamic_parameters_handler);
  void shutdown();

  geometry_msgs::msg::TwistStamped evalControl(
    const geometry_msgs::msg::PoseStamped & robot_pose,
    const geometry_msgs::msg::Twist & robot_speed, const nav_msgs::msg::Path & plan,
    const geometry_msgs::msg::Pose & goal, nav2_core::GoalChecker * goal_checker);

  models::Trajectories & getGeneratedTrajectories();
  xt::xtensor<float, 2> getOptimizedTrajectory();
  void setSpeedLimit(double speed_limit, bool percentage);
  void reset();

protected:
  void getParams();
  void uploadCostmap();

  rclcpp_lifecycle::LifecycleNode::WeakPtr parent_;
  std::shared_ptr<nav2_costmap_2d::Costmap2DROS> costmap_ros_;
  std::string name_;
  ParametersHandler * parameters_handler_{nullptr};
  CriticManager critic_manager_;          // loads the critic plugins: YAML order + parameters
  sortham_host::Optimizer host_;
  sortham_host::models::OptimizerSettings settings_;
  std::string motion_model_name_;
  double controller_frequency_{0.0};
  bool regenerate_noises_{false};
  bool visualize_{false};
  models::Trajectories generated_trajectories_;
  rclcpp::Logger logger_{rclcpp::get_logger("SORTHAMController")};
};

}  // namespace sortham
#endif
