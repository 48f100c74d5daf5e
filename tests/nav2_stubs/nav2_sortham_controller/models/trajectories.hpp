// stand-in (declarations only) for include/nav2_sortham_controller/models/trajectories.hpp:20-50
#pragma once
#include <xtensor/xtensor.hpp>
namespace sortham::models {
struct Trajectories {
  xt::xtensor<float, 2> x, y, yaws;
  void reset(unsigned int batch_size, unsigned int time_steps);
};
}  // namespace sortham::models
