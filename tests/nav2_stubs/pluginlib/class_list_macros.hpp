// stand-in (declarations only): the export macro must name a concrete class derived from the base
#pragma once
#include <type_traits>
#define PLUGINLIB_EXPORT_CLASS(Class, Base) \
  static_assert(std::is_base_of<Base, Class>::value && !std::is_abstract<Class>::value, "plugin class");
