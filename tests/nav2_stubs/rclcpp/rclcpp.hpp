// stand-in (declarations only): rclcpp names the adaptor uses
#pragma once
#include <string>
namespace rclcpp {
class Logger {};
Logger get_logger(const std::string & name);
class Time {};
}  // namespace rclcpp
#define RCLCPP_INFO(logger, ...) do {(void)(logger);} while (0)
