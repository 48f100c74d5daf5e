// stand-in (declarations only)
#pragma once
#include "geometry_msgs/msg/twist.hpp"
namespace geometry_msgs::msg {struct PoseStamped {std_msgs::msg::Header header; Pose pose;};}
