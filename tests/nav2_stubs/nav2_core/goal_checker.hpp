// stand-in (declarations only): nav2_core::GoalChecker::getTolerances
#pragma once
#include "geometry_msgs/msg/twist.hpp"
namespace nav2_core {
class GoalChecker {
public:
  virtual ~GoalChecker() = default;
  virtual bool getTolerances(geometry_msgs::msg::Pose & pose_tolerance, geometry_msgs::msg::Twist & vel_tolerance) = 0;
};
}  // namespace nav2_core
