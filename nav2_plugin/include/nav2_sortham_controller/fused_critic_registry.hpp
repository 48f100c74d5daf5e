// fused_critic_registry.hpp — where the pluginlib-loaded critic objects leave their
// parameters for the optimizer.  Compiled only inside a ROS 2 Humble + Nav2 workspace.
//
// In the reference each critic object scores the [batch, time] tensors itself
// (critic_function.hpp:44-114, critic_manager.cpp:67-76).  Here scoring is fused in
// libsmpc; the critic classes stay (critics.xml, the YAML `critics` list and every
// parameter name are unchanged) but only carry their parameters.
#ifndef NAV2_SORTHAM_CONTROLLER__FUSED_CRITIC_REGISTRY_HPP_
#define NAV2_SORTHAM_CONTROLLER__FUSED_CRITIC_REGISTRY_HPP_

#include <map>
#include <mutex>
#include <string>

#include "smpc.h"

namespace sortham::critics
{

struct FusedCriticRegistry
{
  // keyed by the controller's plugin name (several controllers may be loaded)
  struct Entry
  {
    smpc_critic_params params{};
    float cost_scaling_factor{10.0f};
    float inflation_radius{0.55f};
  };
  static FusedCriticRegistry & get()
  {
    static FusedCriticRegistry r;
    return r;
  }
  Entry & entry(const std::string & controller_name)
  {
    std::lock_guard<std::mutex> g(m_);
    return entries_[controller_name];
  }

private:
  std::mutex m_;
  std::map<std::string, Entry> entries_;
};

}  // namespace sortham::critics
#endif
