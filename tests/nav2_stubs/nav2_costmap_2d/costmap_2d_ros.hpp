// stand-in (declarations only): the nav2_costmap_2d names the adaptor uses
// (call sites of the reference: obstacles_critic.cpp:32,59,66-77; controller.cpp:99-100)
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "geometry_msgs/msg/twist.hpp"
namespace nav2_costmap_2d {
class Costmap2D {
public:
  unsigned char * getCharMap() const;
  unsigned int getSizeInCellsX() const;
  unsigned int getSizeInCellsY() const;
  double getOriginX() const;
  double getOriginY() const;
  double getResolution() const;
};
class Layer {public: virtual ~Layer() = default;};
class LayeredCostmap {
public:
  bool isTrackingUnknown();
  double getInscribedRadius();
  double getCircumscribedRadius();
  std::vector<std::shared_ptr<Layer>> * getPlugins();
};
class Costmap2DROS {
public:
  Costmap2D * getCostmap();
  LayeredCostmap * getLayeredCostmap();
  std::vector<geometry_msgs::msg::Point> getRobotFootprint();
  std::string getBaseFrameID();
};
}  // namespace nav2_costmap_2d
