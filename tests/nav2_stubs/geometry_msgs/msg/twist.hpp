// stand-in (declarations only)
#pragma once
#include <string>
#include "rclcpp/rclcpp.hpp"
namespace std_msgs::msg {struct Header {std::string frame_id; rclcpp::Time stamp;};}
namespace geometry_msgs::msg {
struct Vector3 {double x, y, z;};
struct Point {double x, y, z;};
struct Quaternion {double x, y, z, w;};
struct Twist {Vector3 linear, angular;};
struct Pose {Point position; Quaternion orientation;};
struct Point32 {float x, y, z;};
}  // namespace geometry_msgs::msg
