// stand-in (declarations only) for include/nav2_sortham_controller/critic_manager.hpp:35-110
#pragma once
#include <memory>
#include <string>
#include "nav2_sortham_controller/critic_function.hpp"
namespace sortham {
class CriticManager {
public:
  CriticManager() = default;
  virtual ~CriticManager() = default;
  void on_configure(rclcpp_lifecycle::LifecycleNode::WeakPtr parent, const std::string & name,
                    std::shared_ptr<nav2_costmap_2d::Costmap2DROS>, ParametersHandler *);
  void evalTrajectoriesScores(CriticData & data) const;
};
}  // namespace sortham
