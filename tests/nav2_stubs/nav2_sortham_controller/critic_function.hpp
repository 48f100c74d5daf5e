// stand-in (declarations only) for include/nav2_sortham_controller/critic_function.hpp:35-115
#pragma once
#include <memory>
#include <string>
#include "nav2_costmap_2d/costmap_2d_ros.hpp"
#include "nav2_sortham_controller/tools/parameters_handler.hpp"
namespace sortham {struct CriticData;}
namespace sortham::critics {
class CriticFunction {
public:
  CriticFunction() = default;
  virtual ~CriticFunction() = default;
  void on_configure(rclcpp_lifecycle::LifecycleNode::WeakPtr parent, const std::string & parent_name,
                    const std::string & name, std::shared_ptr<nav2_costmap_2d::Costmap2DROS> costmap_ros,
                    ParametersHandler * param_handler);
  virtual void score(CriticData & data) = 0;
  virtual void initialize() = 0;
  std::string getName();
protected:
  bool enabled_;
  std::string name_, parent_name_;
  rclcpp_lifecycle::LifecycleNode::WeakPtr parent_;
  std::shared_ptr<nav2_costmap_2d::Costmap2DROS> costmap_ros_;
  nav2_costmap_2d::Costmap2D * costmap_{nullptr};
  ParametersHandler * parameters_handler_;
  rclcpp::Logger logger_{rclcpp::get_logger("SORTHAMController")};
};
}  // namespace sortham::critics
