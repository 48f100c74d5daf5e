// path_handler.cpp — see path_handler.hpp.  Line references: /root/reference/
// nav2_sortham_controller/src/path_handler.cpp, src/trajectory_visualizer.cpp.
#include "path_handler.hpp"

#include <algorithm>
#include <cmath>
#include <stdexcept>

namespace SORTHAM_HOST_NS
{

Pose2 Transform2::apply(const Pose2 & p) const
{
  const double c = std::cos(yaw), s = std::sin(yaw);
  return Pose2{c * p.x - s * p.y + tx, s * p.x + c * p.y + ty, p.yaw + yaw};
}

// Costmap2D::worldToMap (nav2_costmap_2d, Humble)
bool CostmapGeometry::worldToMap(double wx, double wy, unsigned & mx, unsigned & my) const
{
  if (wx < origin_x || wy < origin_y) {
    return false;
  }
  mx = static_cast<unsigned>((wx - origin_x) / resolution);
  my = static_cast<unsigned>((wy - origin_y) / resolution);
  return mx < size_x && my < size_y;
}

namespace
{
double euclidean_distance(const Pose2 & a, const Pose2 & b)
{
  return std::hypot(a.x - b.x, a.y - b.y);
}

// nav2_util::geometry_utils::first_after_integrated_distance: the first element whose distance
// along the path from `begin` exceeds `getSearchDist`, else `end`
size_t first_after_integrated_distance(const std::vector<Pose2> & v, size_t begin, size_t end, double search_dist)
{
  if (begin == end) {
    return end;
  }
  double dist = 0.0;
  for (size_t it = begin; it + 1 != end; ++it) {
    dist += euclidean_distance(v[it], v[it + 1]);
    if (dist > search_dist) {
      return it + 1;
    }
  }
  return end;
}
}  // namespace

unsigned int findFirstPathInversion(const std::vector<Pose2> & path)
{
  // (two segments are the least a direction change needs: shorter paths have none, :617-620)
  if (path.size() < 3) {
    return static_cast<unsigned int>(path.size());
  }
  for (unsigned int idx = 1; idx < path.size() - 1; ++idx) {
    // the reference narrows the differences to float (tools/utils.hpp:622-629)
    const float oa_x = static_cast<float>(path[idx].x - path[idx - 1].x);
    const float oa_y = static_cast<float>(path[idx].y - path[idx - 1].y);
    const float ab_x = static_cast<float>(path[idx + 1].x - path[idx].x);
    const float ab_y = static_cast<float>(path[idx + 1].y - path[idx].y);
    const float dot_product = (oa_x * ab_x) + (oa_y * ab_y);
    if (dot_product < 0.0) {
      return idx + 1;
    }
  }
  return static_cast<unsigned int>(path.size());
}

unsigned int removePosesAfterFirstInversion(std::vector<Pose2> & path)
{
  const unsigned int first_after_inversion = findFirstPathInversion(path);
  if (first_after_inversion == path.size()) {
    return 0u;
  }
  path.erase(path.begin() + first_after_inversion, path.end());
  return first_after_inversion;
}

void PathHandler::initialize(const CostmapGeometry & costmap, const PathHandlerParams & params)
{
  costmap_ = costmap;
  p_ = params;
  if (p_.max_robot_pose_search_dist < 0.0) {
    p_.max_robot_pose_search_dist = getMaxCostmapDist();   // the parameter's default (:39)
  }
  inversion_locale_ = 0u;
}

double PathHandler::getMaxCostmapDist() const
{
  return static_cast<double>(std::max(costmap_.size_x, costmap_.size_y)) * costmap_.resolution * 0.50;
}

void PathHandler::setPath(const std::vector<Pose2> & plan)
{
  global_plan_ = plan;
  global_plan_up_to_inversion_ = global_plan_;
  if (p_.enforce_path_inversion) {
    inversion_locale_ = removePosesAfterFirstInversion(global_plan_up_to_inversion_);
  }
}

void PathHandler::prunePlan(std::vector<Pose2> & plan, size_t end)
{
  plan.erase(plan.begin(), plan.begin() + static_cast<std::ptrdiff_t>(end));
}

std::vector<Pose2> PathHandler::getGlobalPlanConsideringBoundsInCostmapFrame(
  const Pose2 & global_pose, const Transform2 & plan_to_costmap, size_t & closest)
{
  const std::vector<Pose2> & plan = global_plan_up_to_inversion_;
  // the nearest pose is looked for only within max_robot_pose_search_dist of path length from the start (:56-61)
  const size_t upper = first_after_integrated_distance(plan, 0, plan.size(), p_.max_robot_pose_search_dist);
  // nav2_util::geometry_utils::min_by: the first minimum
  closest = 0;
  if (upper > 0) {
    double best = euclidean_distance(global_pose, plan[0]);
    for (size_t i = 1; i < upper; ++i) {
      const double d = euclidean_distance(global_pose, plan[i]);
      if (d < best) {
        best = d;
        closest = i;
      }
    }
  }
  std::vector<Pose2> transformed_plan;
  const size_t pruned_end = first_after_integrated_distance(plan, closest, plan.size(), p_.prune_distance);
  unsigned int mx, my;
  // the furthest relevant pose on the path within the costmap's bounds, transformed on the way
  for (size_t i = closest; i < pruned_end; ++i) {
    const Pose2 costmap_plan_pose = plan_to_costmap.apply(plan[i]);
    if (!costmap_.worldToMap(costmap_plan_pose.x, costmap_plan_pose.y, mx, my)) {
      return transformed_plan;
    }
    transformed_plan.push_back(costmap_plan_pose);
  }
  return transformed_plan;
}

std::vector<Pose2> PathHandler::transformPath(const Pose2 & global_pose, const Transform2 & plan_to_costmap)
{
  if (global_plan_up_to_inversion_.empty()) {
    throw std::runtime_error("Received plan with zero length");   // transformToGlobalPlanFrame (:108-110)
  }
  size_t lower_bound = 0;
  std::vector<Pose2> transformed_plan =
    getGlobalPlanConsideringBoundsInCostmapFrame(global_pose, plan_to_costmap, lower_bound);
  prunePlan(global_plan_up_to_inversion_, lower_bound);
  if (p_.enforce_path_inversion && inversion_locale_ != 0u) {
    if (isWithinInversionTolerances(global_pose)) {
      prunePlan(global_plan_, inversion_locale_);
      global_plan_up_to_inversion_ = global_plan_;
      inversion_locale_ = removePosesAfterFirstInversion(global_plan_up_to_inversion_);
    }
  }
  if (transformed_plan.empty()) {
    throw std::runtime_error("Resulting plan has 0 poses in it.");
  }
  return transformed_plan;
}

Pose2 PathHandler::getTransformedGoal(const Transform2 & plan_to_costmap) const
{
  if (global_plan_.empty()) {
    throw std::runtime_error("Received plan with zero length");
  }
  return plan_to_costmap.apply(global_plan_.back());
}

bool PathHandler::isWithinInversionTolerances(const Pose2 & robot_pose) const
{
  // close enough to the inversion pose, in position and heading: the rest of the plan comes back (:209-216)
  const Pose2 & last_pose = global_plan_up_to_inversion_.back();
  const float distance = hypotf(
    static_cast<float>(robot_pose.x - last_pose.x), static_cast<float>(robot_pose.y - last_pose.y));
  // angles::shortest_angular_distance(from, to) = normalize_angle(to - from)
  const double a = std::fmod(std::fmod(last_pose.yaw - robot_pose.yaw, 2.0 * M_PI) + 2.0 * M_PI, 2.0 * M_PI);
  const float angle_distance = static_cast<float>(a > M_PI ? a - 2.0 * M_PI : a);
  return distance <= p_.inversion_xy_tolerance && std::fabs(angle_distance) <= p_.inversion_yaw_tolerance;
}

// ---- TrajectoryVisualizer --------------------------------------------------------------------
void TrajectoryVisualizer::on_configure(const std::string & frame_id, int trajectory_step, int time_step)
{
  frame_id_ = frame_id;
  trajectory_step_ = trajectory_step;
  time_step_ = time_step;
  reset();
}

void TrajectoryVisualizer::add(const float * trajectory, size_t size, size_t stride, const std::string & ns)
{
  if (!size) {
    return;
  }
  for (size_t i = 0; i < size; i++) {
    const float component = static_cast<float>(i) / static_cast<float>(size);
    Marker m;
    m.id = marker_id_++;
    m.x = trajectory[i * stride + 0];
    m.y = trajectory[i * stride + 1];
    m.z = 0.06;
    if (i != size - 1) {
      m.scale_x = 0.03; m.scale_y = 0.03; m.scale_z = 0.07;
    } else {
      m.scale_x = 0.07; m.scale_y = 0.07; m.scale_z = 0.09;
    }
    m.r = 0; m.g = component; m.b = component; m.a = 1;
    m.frame_id = frame_id_;
    m.ns = ns;
    points_.push_back(m);
  }
}

void TrajectoryVisualizer::add(const float * x, const float * y, size_t B, size_t T, const std::string & ns)
{
  const float shape_1 = static_cast<float>(T);
  for (size_t i = 0; i < B; i += static_cast<size_t>(trajectory_step_)) {
    for (size_t j = 0; j < T; j += static_cast<size_t>(time_step_)) {
      const float j_flt = static_cast<float>(j);
      Marker m;
      m.id = marker_id_++;
      m.x = x[i * T + j];
      m.y = y[i * T + j];
      m.z = 0.03;
      m.scale_x = m.scale_y = m.scale_z = 0.03;
      m.r = 0; m.g = j_flt / shape_1; m.b = 1.0f - j_flt / shape_1; m.a = 1;
      m.frame_id = frame_id_;
      m.ns = ns;
      points_.push_back(m);
    }
  }
}

void TrajectoryVisualizer::reset()
{
  marker_id_ = 0;
  points_.clear();
}

std::vector<Marker> TrajectoryVisualizer::visualize()
{
  std::vector<Marker> out;
  out.swap(points_);
  reset();
  return out;
}

}  // namespace SORTHAM_HOST_NS
