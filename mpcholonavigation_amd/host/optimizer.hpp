// optimizer.hpp — host side of the sampling-MPC controller above the C-ABI.
//
// Mirrors sortham::Optimizer of the reference (nav2_sortham_controller
// include/nav2_sortham_controller/optimizer.hpp:51-263, src/optimizer.cpp) with
// the same method names, argument meaning and error behaviour, minus the ROS
// types: poses arrive as (x, y, yaw) — tf2::getYaw is applied by the caller —
// and the costmap as a plain view.  Everything over [batch, time] happens in
// libsmpc.so (include/smpc.h); this class keeps what the reference keeps on the
// host: prepare / fallback / Savitzky-Golay filter / Twist extraction /
// sequence shift / speed limit / reset.
#ifndef SORTHAM_HOST_OPTIMIZER_HPP_
#define SORTHAM_HOST_OPTIMIZER_HPP_

#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/smpc.h"

// The class is sortham::Optimizer, like the reference's.  Inside the Nav2 package the
// ROS-typed adaptor of the same name wraps it (nav2_plugin/), which then builds this file
// with -DSORTHAM_HOST_NS=sortham_host.
#ifndef SORTHAM_HOST_NS
#define SORTHAM_HOST_NS sortham
#endif

namespace SORTHAM_HOST_NS
{

namespace models
{
// ref models/control_sequence.hpp:27-49
struct Control
{
  float vx, vy, wz;
};

struct ControlSequence
{
  std::vector<float> vx, vy, wz;
  void reset(unsigned int time_steps)
  {
    vx.assign(time_steps, 0.0f);
    vy.assign(time_steps, 0.0f);
    wz.assign(time_steps, 0.0f);
  }
};

// ref models/constraints.hpp:25-42
struct ControlConstraints
{
  float vx_max, vx_min, vy, wz;
};
struct SamplingStd
{
  float vx, vy, wz;
};

// ref models/optimizer_settings.hpp:28-41
struct OptimizerSettings
{
  ControlConstraints base_constraints{0, 0, 0, 0};
  ControlConstraints constraints{0, 0, 0, 0};
  SamplingStd sampling_std{0, 0, 0};
  float model_dt{0};
  float temperature{0};
  float gamma{0};
  unsigned int batch_size{0};
  unsigned int time_steps{0};
  unsigned int iteration_count{0};
  bool shift_control_sequence{false};
  size_t retry_attempt_limit{0};
};

// ref models/path.hpp:27-44 (utils::toTensor output)
struct Path
{
  std::vector<float> x, y, yaws;
};
}  // namespace models

struct Pose2D
{
  double x{0}, y{0};
  double yaw{0};  // tf2::getYaw(orientation)
};
struct Twist2D
{
  double vx{0}, vy{0}, wz{0};
};

// what the critics read from nav2_costmap_2d::Costmap2DROS
struct CostmapView
{
  const uint8_t * cells{nullptr};
  unsigned int size_x{0}, size_y{0};
  double origin_x{0}, origin_y{0}, resolution{0};
  bool track_unknown{false};
  float inscribed_radius{0};
  bool has_inflation_layer{false};
  // consider_footprint = true: costmap_ros->getRobotFootprint() as (x, y) pairs, the layered
  // costmap's circumscribed radius, the inflation layer's own cost_scaling_factor
  std::vector<double> footprint_xy;
  double circumscribed_radius{0};
  double layer_cost_scaling_factor{-1.0};
};

struct CriticsConfig
{
  std::vector<std::string> critics;  // YAML `critics` list, in order (critic_manager.cpp:36-41)
  smpc_critic_params params{};       // per-critic parameters (enabled is derived from `critics`)
  float cost_scaling_factor{10.0f};  // ObstaclesCritic params read with an InflationLayer
  float inflation_radius{0.55f};     // (obstacles_critic.cpp:76-80)
};

namespace utils
{
// ref tools/utils.hpp:442-605
void savitskyGolayFilter(
  models::ControlSequence & control_sequence, std::array<models::Control, 4> & control_history,
  const models::OptimizerSettings & settings);
}  // namespace utils

class Optimizer
{
public:
  Optimizer() = default;
  ~Optimizer();
  Optimizer(const Optimizer &) = delete;
  Optimizer & operator=(const Optimizer &) = delete;

  // ref optimizer.cpp:35-55 + getParams :62-93 (parameters arrive resolved)
  void initialize(
    const models::OptimizerSettings & settings, const std::string & motion_model,
    double controller_frequency, const CriticsConfig & critics, bool regenerate_noises = false,
    uint64_t noise_seed = 0, int device = -1);
  void shutdown();  // ref :57-60

  // the caller holds the costmap mutex for the tick (controller.cpp:99-100)
  void setCostmap(const CostmapView & costmap);
  // parity runs: supply NoiseGenerator's tensors instead of drawing them
  void setNoise(const float * nvx, const float * nvy, const float * nwz);

  // ref :134-155; throws std::runtime_error exactly where the reference does
  // goal_checker_xy_tolerance: GoalChecker::getTolerances()'s pose_tolerance.position.x, < 0
  // when the controller server passes no goal checker (TwirlingCritic's gate)
  Twist2D evalControl(
    const Pose2D & robot_pose, const Twist2D & robot_speed, const models::Path & plan,
    const Pose2D & goal, float goal_checker_xy_tolerance = -1.0f);

  void setSpeedLimit(double speed_limit, bool percentage);  // ref :428-453
  void reset();                                             // ref :116-132

  // ref :345-360: [T][3] x, y, yaw of the optimal sequence
  std::vector<std::array<float, 3>> getOptimizedTrajectory();
  // ref :455-458 (needs visualize = true at initialize)
  void getGeneratedTrajectories(std::vector<float> & x, std::vector<float> & y, std::vector<float> & yaws);

  models::ControlSequence & controlSequence() {return control_sequence_;}
  const models::OptimizerSettings & settings() const {return settings_;}
  const smpc_tick_out & lastTick() const {return last_out_;}
  void setVisualize(bool v) {visualize_ = v;}
  bool isHolonomic() const {return motion_model_ == SMPC_MODEL_OMNI;}   // ref :235
  // AckermannConstraints.min_turning_r (motion_models.hpp:91-95); call before initialize()
  void setAckermannMinTurningRadius(float r) {ackermann_min_turning_r_ = r;}

protected:
  void optimize();                         // ref :157-164 -> smpc_optimize
  bool fallback(bool fail);                // ref :166-183 (retry counter per object)
  void prepare(const Pose2D &, const Twist2D &, const models::Path &, const Pose2D &);  // ref :185-204
  void shiftControlSequence();             // ref :206-225
  Twist2D getControlFromSequenceAsTwist(); // ref :396-410
  void setOffset(double controller_frequency);  // ref :95-114
  void setMotionModel(const std::string & model);  // ref :412-426
  void pushConstraints();

  models::OptimizerSettings settings_{};
  models::ControlSequence control_sequence_{};
  std::array<models::Control, 4> control_history_{};
  CriticsConfig critics_{};
  bool regenerate_noises_{false};
  uint32_t motion_model_{SMPC_MODEL_OMNI};
  float ackermann_min_turning_r_{0.2f};
  bool visualize_{false};
  uint64_t noise_seed_{0};
  bool supplied_noise_{false};
  int device_{-1};
  smpc_ctx * ctx_{nullptr};
  smpc_config built_cfg_{};     // what ctx_ was created with (initialize() keeps it for an identical one)
  uint64_t built_seed_{0};
  bool have_built_{false};

  // per-tick (CriticData)
  Pose2D pose_{}, goal_{};
  Twist2D speed_{};
  models::Path path_{};
  bool fail_flag_{false};
  float goal_checker_xy_tolerance_{-1.0f};
  size_t retry_counter_{0};
  smpc_tick_out last_out_{};
};

}  // namespace SORTHAM_HOST_NS

#endif
