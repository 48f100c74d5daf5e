// path_handler.hpp — sortham::PathHandler and sortham::TrajectoryVisualizer for plain types
// (SURVEY 8(f) rank 4).  The reference's classes (include/nav2_sortham_controller/tools/
// path_handler.hpp:46-165, trajectory_visualizer.hpp) hold ROS messages, a tf2 buffer and a
// Costmap2DROS; their ARITHMETIC is restated here on poses {x, y, yaw}: closest-point search
// bounded by max_robot_pose_search_dist, pruning, the prune_distance window, the walk to the
// costmap's edge, inversion enforcement, the marker lists.  What ROS supplied comes in as
// arguments: the robot pose already in the plan's frame (tf2 in the reference,
// path_handler.cpp:105-121) and the rigid transform from the plan's frame to the costmap's
// (identity when they coincide, as in every reference test).  nav2_util::geometry_utils'
// first_after_integrated_distance / min_by / euclidean_distance are not under /root/reference:
// restated from their published Humble form.
#ifndef SORTHAM_PATH_HANDLER_HPP_
#define SORTHAM_PATH_HANDLER_HPP_

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#ifndef SORTHAM_HOST_NS
#define SORTHAM_HOST_NS sortham
#endif

namespace SORTHAM_HOST_NS
{

struct Pose2 {
  double x = 0, y = 0, yaw = 0;
};

// pose in frame B = R(yaw) * pose in frame A + (tx, ty); yaw adds
struct Transform2 {
  double tx = 0, ty = 0, yaw = 0;
  Pose2 apply(const Pose2 & p) const;
};

struct CostmapGeometry {   // what Costmap2D::worldToMap / getSizeInCells / getResolution need
  uint32_t size_x = 0, size_y = 0;
  double resolution = 0.05, origin_x = 0, origin_y = 0;
  bool worldToMap(double wx, double wy, unsigned & mx, unsigned & my) const;
};

struct PathHandlerParams {   // path_handler.cpp:36-46
  double max_robot_pose_search_dist = -1.0;   // < 0: getMaxCostmapDist()
  double prune_distance = 1.5;
  bool enforce_path_inversion = false;
  float inversion_xy_tolerance = 0.2f;
  float inversion_yaw_tolerance = 0.4f;
};

// utils::findFirstPathInversion / removePosesAfterFirstInversion (tools/utils.hpp:612-658)
unsigned int findFirstPathInversion(const std::vector<Pose2> & path);
unsigned int removePosesAfterFirstInversion(std::vector<Pose2> & path);

class PathHandler
{
public:
  void initialize(const CostmapGeometry & costmap, const PathHandlerParams & params);
  void setPath(const std::vector<Pose2> & plan);                    // path_handler.cpp:173-180
  std::vector<Pose2> & getPath() {return global_plan_;}             // :182
  // transformPath (:123-145): global_pose = the robot pose in the plan's frame; returns the
  // pruned plan in the costmap's frame; throws std::runtime_error as the reference does
  std::vector<Pose2> transformPath(const Pose2 & global_pose, const Transform2 & plan_to_costmap);
  Pose2 getTransformedGoal(const Transform2 & plan_to_costmap) const;   // :189-203

  // protected in the reference (its tests reach them through a wrapper subclass)
  double getMaxCostmapDist() const;                                 // :166-171
  // :48-103; closest = index into the plan up to the inversion of the pose closest to the robot
  std::vector<Pose2> getGlobalPlanConsideringBoundsInCostmapFrame(
    const Pose2 & global_pose, const Transform2 & plan_to_costmap, size_t & closest);
  static void prunePlan(std::vector<Pose2> & plan, size_t end);     // :184-187
  bool isWithinInversionTolerances(const Pose2 & robot_pose) const; // :205-220
  std::vector<Pose2> & planUpToInversion() {return global_plan_up_to_inversion_;}
  unsigned int inversionLocale() const {return inversion_locale_;}

private:
  CostmapGeometry costmap_;
  PathHandlerParams p_;
  std::vector<Pose2> global_plan_, global_plan_up_to_inversion_;
  unsigned int inversion_locale_ = 0u;
};

// ---- TrajectoryVisualizer (src/trajectory_visualizer.cpp:59-128) ---------------------------
struct Marker {   // the fields utils::createMarker fills (tools/utils.hpp:119-136), SPHERE / ADD
  int id = 0;
  double x = 0, y = 0, z = 0;
  double scale_x = 0, scale_y = 0, scale_z = 0;
  float r = 0, g = 0, b = 0, a = 0;
  std::string frame_id, ns;
};

class TrajectoryVisualizer
{
public:
  void on_configure(const std::string & frame_id, int trajectory_step = 5, int time_step = 3);
  // the optimal trajectory, [n][2+] row-major with `stride` floats per row (:59-83)
  void add(const float * trajectory, size_t n, size_t stride, const std::string & marker_namespace);
  // candidate trajectories x, y [B][T] (:85-107)
  void add(const float * x, const float * y, size_t B, size_t T, const std::string & marker_namespace);
  void reset();                                                     // :109-113
  // visualize() (:115-126): hands the accumulated markers over (what would be published) and resets
  std::vector<Marker> visualize();
  const std::vector<Marker> & markers() const {return points_;}

private:
  std::string frame_id_;
  int trajectory_step_ = 5, time_step_ = 3, marker_id_ = 0;
  std::vector<Marker> points_;
};

}  // namespace SORTHAM_HOST_NS

#endif  // SORTHAM_PATH_HANDLER_HPP_
