# Changes to nav2_sortham_controller/CMakeLists.txt for the MI355X build
# (the rest of the reference CMakeLists stays).
#
#   set(SMPC_ROOT <path to this repository>)
#
# 1. controller library: swap the optimizer / noise generator / critic manager sources
add_library(sortham_controller SHARED
  src/controller.cpp                         # reference, unchanged
  ${SMPC_ROOT}/nav2_plugin/src/optimizer.cpp # replaces src/optimizer.cpp + src/noise_generator.cpp
  src/critic_manager.cpp                     # reference, unchanged (pluginlib loading of critics)
  src/trajectory_visualizer.cpp              # reference, unchanged
  src/path_handler.cpp                       # reference, unchanged
  src/parameters_handler.cpp                 # reference, unchanged
  ${SMPC_ROOT}/mpcholonavigation_amd/host/optimizer.cpp
)
target_compile_definitions(sortham_controller PRIVATE SORTHAM_HOST_NS=sortham_host)
target_include_directories(sortham_controller BEFORE PRIVATE
  ${SMPC_ROOT}/nav2_plugin/include           # shadows include/nav2_sortham_controller/optimizer.hpp
  ${SMPC_ROOT}/mpcholonavigation_amd
  ${SMPC_ROOT}/include)
target_link_libraries(sortham_controller ${SMPC_ROOT}/mpcholonavigation_amd/libsmpc.so)

# 2. critics library: twelve registered classes, five of them fused parameter carriers
add_library(sortham_critics SHARED ${SMPC_ROOT}/nav2_plugin/src/fused_critics.cpp)
target_include_directories(sortham_critics PRIVATE ${SMPC_ROOT}/nav2_plugin/include ${SMPC_ROOT}/include)

# 3. plugin descriptions: same library names, class names and base classes as the reference
pluginlib_export_plugin_description_file(nav2_core ${SMPC_ROOT}/nav2_plugin/sorthamc.xml)
pluginlib_export_plugin_description_file(nav2_sortham_controller ${SMPC_ROOT}/nav2_plugin/critics.xml)
