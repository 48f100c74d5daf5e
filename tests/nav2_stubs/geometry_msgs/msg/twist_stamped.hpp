// stand-in (declarations only)
#pragma once
#include "geometry_msgs/msg/twist.hpp"
namespace geometry_msgs::msg {struct TwistStamped {std_msgs::msg::Header header; Twist twist;};}
