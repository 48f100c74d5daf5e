// stand-in (declarations only)
#pragma once
#include <memory>
#include "rclcpp/rclcpp.hpp"
namespace rclcpp_lifecycle {
class LifecycleNode {
public:
  using WeakPtr = std::weak_ptr<LifecycleNode>;
  rclcpp::Logger get_logger() const;
};
}  // namespace rclcpp_lifecycle
