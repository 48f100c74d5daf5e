// stand-in (declarations only) for the reference's own header
// include/nav2_sortham_controller/tools/parameters_handler.hpp:30-130
#pragma once
#include <mutex>
#include <string>
#include "rclcpp_lifecycle/lifecycle_node.hpp"
namespace sortham {
enum class ParameterType { Dynamic, Static };
class ParametersHandler {
public:
  inline auto getParamGetter(const std::string & ns)
  {
    return [this, ns](auto & setting, const std::string & name, auto default_value,
                      ParameterType param_type = ParameterType::Dynamic) {
             getParam(setting, ns + "." + name, default_value, param_type);
           };
  }
  template <typename T> void addPostCallback(T && callback) {(void)callback;}   // (a template over a local type needs a body)
  std::mutex * getLock();
protected:
  template <typename SettingT, typename ParamT>
  void getParam(SettingT & setting, const std::string & name, ParamT default_value, ParameterType param_type)
  {
    // the reference assigns the declared parameter's value (node->get_parameter(name, setting)):
    // the setting must be assignable from the default's type
    setting = static_cast<SettingT>(default_value);
    (void)name;
    (void)param_type;
  }
};
}  // namespace sortham
