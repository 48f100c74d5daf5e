// fused_critics.cpp — the twelve sortham::critics::* classes critics.xml registers.
// Compiled only inside a ROS 2 Humble + Nav2 workspace (needs the reference package's
// critic_function.hpp, pluginlib, nav2_costmap_2d); NOT built by this repository's
// tests — see INTEGRATION.md.
//
// The twelve critics read exactly the parameters their reference initialize() reads (same
// names, same defaults) and publish them to FusedCriticRegistry; score() does nothing because
// libsmpc scores inside the fused kernels.
#include <stdexcept>
#include <string>
#include <vector>

#include "nav2_sortham_controller/critic_function.hpp"   // from the reference package
#include "nav2_sortham_controller/fused_critic_registry.hpp"

namespace sortham::critics
{

namespace
{
// name_ is "<controller>.<CriticName>" (critic_manager.cpp:55-57)
std::string controller_of(const std::string & full)
{
  const auto dot = full.rfind('.');
  return dot == std::string::npos ? full : full.substr(0, dot);
}
}  // namespace

#define FUSED_CRITIC_BEGIN(Class)                                   \
  class Class : public CriticFunction                               \
  {                                                                 \
public:                                                             \
    void score(CriticData &) override {}                            \
    void initialize() override                                      \
    {                                                               \
      auto getParam = parameters_handler_->getParamGetter(name_);   \
      auto & e = FusedCriticRegistry::get().entry(controller_of(name_));

#define FUSED_CRITIC_END }                                          \
  };

FUSED_CRITIC_BEGIN(ObstaclesCritic)   // ref src/critics/obstacles_critic.cpp:21-31,76-80
auto & p = e.params.obstacles;
bool consider_footprint = false;
p.enabled = enabled_;
getParam(consider_footprint, "consider_footprint", false);
getParam(p.cost_power, "cost_power", 1);
getParam(p.repulsion_weight, "repulsion_weight", 1.5);
getParam(p.critical_weight, "critical_weight", 20.0);
getParam(p.collision_cost, "collision_cost", 10000.0);
getParam(p.collision_margin_distance, "collision_margin_distance", 0.10);
getParam(p.near_goal_distance, "near_goal_distance", 0.5);
getParam(e.cost_scaling_factor, "cost_scaling_factor", 10.0);
getParam(e.inflation_radius, "inflation_radius", 0.55);
p.consider_footprint = consider_footprint;   // footprint uploaded with the costmap
FUSED_CRITIC_END

FUSED_CRITIC_BEGIN(PathAlignCritic)   // ref src/critics/path_align_critic.cpp:26-38
auto & p = e.params.path_align;
bool use_path_orientations = false;
p.enabled = enabled_;
getParam(p.cost_power, "cost_power", 1);
getParam(p.cost_weight, "cost_weight", 10.0);
getParam(p.max_path_occupancy_ratio, "max_path_occupancy_ratio", 0.07);
getParam(p.offset_from_furthest, "offset_from_furthest", 20);
getParam(p.trajectory_point_step, "trajectory_point_step", 4);
getParam(p.threshold_to_consider, "threshold_to_consider", 0.5);
getParam(use_path_orientations, "use_path_orientations", false);
p.use_path_orientations = use_path_orientations;
FUSED_CRITIC_END

FUSED_CRITIC_BEGIN(PathFollowCritic)  // ref src/critics/path_follow_critic.cpp:23-33
auto & p = e.params.path_follow;
p.enabled = enabled_;
getParam(p.threshold_to_consider, "threshold_to_consider", 1.4);
getParam(p.offset_from_furthest, "offset_from_furthest", 6);
getParam(p.cost_power, "cost_power", 1);
getParam(p.cost_weight, "cost_weight", 5.0);
FUSED_CRITIC_END

FUSED_CRITIC_BEGIN(GoalAngleCritic)   // ref src/critics/goal_angle_critic.cpp:20-27
auto & p = e.params.goal_angle;
p.enabled = enabled_;
getParam(p.cost_power, "cost_power", 1);
getParam(p.cost_weight, "cost_weight", 3.0);
getParam(p.threshold_to_consider, "threshold_to_consider", 0.5);
FUSED_CRITIC_END

FUSED_CRITIC_BEGIN(PreferForwardCritic)  // ref src/critics/prefer_forward_critic.cpp:20-27
auto & p = e.params.prefer_forward;
p.enabled = enabled_;
getParam(p.cost_power, "cost_power", 1);
getParam(p.cost_weight, "cost_weight", 5.0);
getParam(p.threshold_to_consider, "threshold_to_consider", 0.5);
FUSED_CRITIC_END

FUSED_CRITIC_BEGIN(CostCritic)   // ref src/critics/cost_critic.cpp:25-34
auto & p = e.params.cost;
bool consider_footprint = false;
p.enabled = enabled_;
getParam(consider_footprint, "consider_footprint", false);
getParam(p.cost_power, "cost_power", 1);
getParam(p.cost_weight, "cost_weight", 3.81);   // libsmpc divides by 254 like :34
getParam(p.critical_cost, "critical_cost", 300.0);
getParam(p.collision_cost, "collision_cost", 1000000.0);
getParam(p.near_goal_distance, "near_goal_distance", 0.5);
p.consider_footprint = consider_footprint;   // footprint uploaded with the costmap
FUSED_CRITIC_END

FUSED_CRITIC_BEGIN(GoalCritic)   // ref src/critics/goal_critic.cpp:26-28
auto & p = e.params.goal;
p.enabled = enabled_;
getParam(p.cost_power, "cost_power", 1);
getParam(p.cost_weight, "cost_weight", 5.0);
getParam(p.threshold_to_consider, "threshold_to_consider", 1.4);
FUSED_CRITIC_END

FUSED_CRITIC_BEGIN(ConstraintCritic)   // ref src/critics/constraint_critic.cpp:27-38
auto & p = e.params.constraint;
auto getParentParam = parameters_handler_->getParamGetter(parent_name_);
p.enabled = enabled_;
getParam(p.cost_power, "cost_power", 1);
getParam(p.cost_weight, "cost_weight", 4.0);
getParentParam(p.vx_max, "vx_max", 0.5);
getParentParam(p.vy_max, "vy_max", 0.0);
getParentParam(p.vx_min, "vx_min", -0.35);
FUSED_CRITIC_END

FUSED_CRITIC_BEGIN(TwirlingCritic)   // ref src/critics/twirling_critic.cpp:24-25
auto & p = e.params.twirling;
p.enabled = enabled_;
getParam(p.cost_power, "cost_power", 1);
getParam(p.cost_weight, "cost_weight", 10.0);
FUSED_CRITIC_END

FUSED_CRITIC_BEGIN(PathAngleCritic)   // ref src/critics/path_angle_critic.cpp:24-45
auto & p = e.params.path_angle;
auto getParentParam = parameters_handler_->getParamGetter(parent_name_);
bool forward_preference = true;
p.enabled = enabled_;
getParentParam(p.vx_min, "vx_min", -0.35);
getParam(p.offset_from_furthest, "offset_from_furthest", 4);
getParam(p.cost_power, "cost_power", 1);
getParam(p.cost_weight, "cost_weight", 2.0);
getParam(p.threshold_to_consider, "threshold_to_consider", 0.5);
getParam(p.max_angle_to_furthest, "max_angle_to_furthest", 1.2);
getParam(forward_preference, "forward_preference", true);
p.forward_preference = forward_preference;
FUSED_CRITIC_END

FUSED_CRITIC_BEGIN(VelocityDeadbandCritic)   // ref src/critics/velocity_deadband_critic.cpp:24-33
auto & p = e.params.velocity_deadband;
std::vector<double> deadband{0.0, 0.0, 0.0};
p.enabled = enabled_;
getParam(p.cost_power, "cost_power", 1);
getParam(p.cost_weight, "cost_weight", 35.0);
getParam(deadband, "deadband_velocities", std::vector<double>{0.0, 0.0, 0.0});
for (size_t k = 0; k < 3 && k < deadband.size(); ++k) {
  p.deadband_velocities[k] = static_cast<float>(deadband[k]);
}
FUSED_CRITIC_END

FUSED_CRITIC_BEGIN(PathAlignLegacyCritic)   // ref src/critics/path_align_legacy_critic.cpp:26-37
auto & p = e.params.path_align_legacy;
bool use_path_orientations = false;
p.enabled = enabled_;
getParam(p.cost_power, "cost_power", 1);
getParam(p.cost_weight, "cost_weight", 10.0);
getParam(p.max_path_occupancy_ratio, "max_path_occupancy_ratio", 0.07);
getParam(p.offset_from_furthest, "offset_from_furthest", 20);
getParam(p.trajectory_point_step, "trajectory_point_step", 4);
getParam(p.threshold_to_consider, "threshold_to_consider", 0.5);
getParam(use_path_orientations, "use_path_orientations", false);
p.use_path_orientations = use_path_orientations;
FUSED_CRITIC_END

}  // namespace sortham::critics

#include <pluginlib/class_list_macros.hpp>

#define EXPORT(Class) PLUGINLIB_EXPORT_CLASS(sortham::critics::Class, sortham::critics::CriticFunction)
EXPORT(ObstaclesCritic)
EXPORT(CostCritic)
EXPORT(GoalCritic)
EXPORT(GoalAngleCritic)
EXPORT(PathAlignCritic)
EXPORT(PathAlignLegacyCritic)
EXPORT(PathAngleCritic)
EXPORT(PathFollowCritic)
EXPORT(PreferForwardCritic)
EXPORT(TwirlingCritic)
EXPORT(ConstraintCritic)
EXPORT(VelocityDeadbandCritic)
