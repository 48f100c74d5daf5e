// stand-in (declarations only): tf2::getYaw (call sites of the reference: src/optimizer.cpp:279,317)
#pragma once
#include "geometry_msgs/msg/twist.hpp"
namespace tf2 {double getYaw(const geometry_msgs::msg::Quaternion & q);}
