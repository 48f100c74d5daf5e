// stand-in (declarations only)
#pragma once
#include "nav2_costmap_2d/costmap_2d_ros.hpp"
namespace nav2_costmap_2d {class InflationLayer : public Layer {public: double getCostScalingFactor();};}
