// path_handler_c.cpp — the C face of path_handler.hpp (include/smpc_host.h).
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../include/smpc_host.h"
#include "path_handler.hpp"

namespace sortham_ns = SORTHAM_HOST_NS;

struct sortham_path_handler {
  sortham_ns::PathHandler h;
  std::string err;
};

struct sortham_trajectory_visualizer {
  sortham_ns::TrajectoryVisualizer v;
  std::string frame;
};

namespace
{
sortham_ns::Pose2 pose_of(const double * p) {return sortham_ns::Pose2{p[0], p[1], p[2]};}
sortham_ns::Transform2 tf_of(const double * t)
{
  return t ? sortham_ns::Transform2{t[0], t[1], t[2]} : sortham_ns::Transform2{};
}
uint32_t copy_out(const std::vector<sortham_ns::Pose2> & v, double * out, uint32_t cap)
{
  const uint32_t n = static_cast<uint32_t>(v.size());
  for (uint32_t i = 0; i < n && i < cap && out; ++i) {
    out[3 * i] = v[i].x;
    out[3 * i + 1] = v[i].y;
    out[3 * i + 2] = v[i].yaw;
  }
  return n;
}
std::vector<sortham_ns::Pose2> poses_in(const double * poses, uint32_t n)
{
  std::vector<sortham_ns::Pose2> v(n);
  for (uint32_t i = 0; i < n; ++i) {
    v[i] = pose_of(poses + 3 * i);
  }
  return v;
}
}  // namespace

extern "C" {

void sortham_path_handler_config_default(sortham_path_handler_config * c)
{
  std::memset(c, 0, sizeof(*c));
  c->costmap_resolution = 0.05;
  c->max_robot_pose_search_dist = -1.0;
  c->prune_distance = 1.5;
  c->inversion_xy_tolerance = 0.2f;
  c->inversion_yaw_tolerance = 0.4f;
}

int sortham_path_handler_create(const sortham_path_handler_config * cfg, sortham_path_handler ** out)
{
  if (!cfg || !out) {
    return SMPC_ERR_INVALID;
  }
  auto * h = new sortham_path_handler();
  sortham_ns::CostmapGeometry g;
  g.size_x = cfg->costmap_size_x;
  g.size_y = cfg->costmap_size_y;
  g.resolution = cfg->costmap_resolution;
  g.origin_x = cfg->costmap_origin_x;
  g.origin_y = cfg->costmap_origin_y;
  sortham_ns::PathHandlerParams p;
  p.max_robot_pose_search_dist = cfg->max_robot_pose_search_dist;
  p.prune_distance = cfg->prune_distance;
  p.enforce_path_inversion = cfg->enforce_path_inversion != 0;
  p.inversion_xy_tolerance = cfg->inversion_xy_tolerance;
  p.inversion_yaw_tolerance = cfg->inversion_yaw_tolerance;
  h->h.initialize(g, p);
  *out = h;
  return SMPC_OK;
}

void sortham_path_handler_destroy(sortham_path_handler * h) {delete h;}
const char * sortham_path_handler_last_error(const sortham_path_handler * h) {return h ? h->err.c_str() : "";}

int sortham_path_handler_set_path(sortham_path_handler * h, const double * poses, uint32_t n)
{
  if (!h || (n && !poses)) {
    return SMPC_ERR_INVALID;
  }
  h->h.setPath(poses_in(poses, n));
  return SMPC_OK;
}

uint32_t sortham_path_handler_get_path(const sortham_path_handler * h, int up_to_inversion, double * poses, uint32_t cap)
{
  if (!h) {
    return 0;
  }
  auto & hh = const_cast<sortham_path_handler *>(h)->h;
  return copy_out(up_to_inversion ? hh.planUpToInversion() : hh.getPath(), poses, cap);
}

int sortham_path_handler_transform_path(
  sortham_path_handler * h, const double * robot, const double * tf, double * poses_out, uint32_t cap,
  uint32_t * n_out)
{
  if (!h || !robot) {
    return SMPC_ERR_INVALID;
  }
  try {
    const auto v = h->h.transformPath(pose_of(robot), tf_of(tf));
    const uint32_t n = copy_out(v, poses_out, cap);
    if (n_out) {
      *n_out = n;
    }
    return SMPC_OK;
  } catch (const std::runtime_error & e) {
    h->err = e.what();
    return SORTHAM_ERR_THROWN;
  }
}

int sortham_path_handler_plan_in_bounds(
  sortham_path_handler * h, const double * robot, const double * tf, double * poses_out, uint32_t cap,
  uint32_t * n_out, uint32_t * closest)
{
  if (!h || !robot) {
    return SMPC_ERR_INVALID;
  }
  size_t c = 0;
  const auto v = h->h.getGlobalPlanConsideringBoundsInCostmapFrame(pose_of(robot), tf_of(tf), c);
  const uint32_t n = copy_out(v, poses_out, cap);
  if (n_out) {
    *n_out = n;
  }
  if (closest) {
    *closest = static_cast<uint32_t>(c);
  }
  return SMPC_OK;
}

int sortham_path_handler_prune(sortham_path_handler * h, int up_to_inversion, uint32_t end)
{
  if (!h) {
    return SMPC_ERR_INVALID;
  }
  auto & plan = up_to_inversion ? h->h.planUpToInversion() : h->h.getPath();
  if (end > plan.size()) {
    return SMPC_ERR_INVALID;
  }
  sortham_ns::PathHandler::prunePlan(plan, end);
  return SMPC_OK;
}

int sortham_path_handler_transformed_goal(sortham_path_handler * h, const double * tf, double * pose_out)
{
  if (!h || !pose_out) {
    return SMPC_ERR_INVALID;
  }
  try {
    const auto g = h->h.getTransformedGoal(tf_of(tf));
    pose_out[0] = g.x;
    pose_out[1] = g.y;
    pose_out[2] = g.yaw;
    return SMPC_OK;
  } catch (const std::runtime_error & e) {
    h->err = e.what();
    return SORTHAM_ERR_THROWN;
  }
}

int sortham_path_handler_within_inversion_tolerances(const sortham_path_handler * h, const double * robot_pose)
{
  if (!h || !robot_pose) {
    return 0;
  }
  auto & hh = const_cast<sortham_path_handler *>(h)->h;
  if (hh.planUpToInversion().empty()) {
    return 0;
  }
  return hh.isWithinInversionTolerances(pose_of(robot_pose)) ? 1 : 0;
}

double sortham_path_handler_max_costmap_dist(const sortham_path_handler * h)
{
  return h ? h->h.getMaxCostmapDist() : 0.0;
}

uint32_t sortham_utils_find_first_path_inversion(const double * poses, uint32_t n)
{
  return sortham_ns::findFirstPathInversion(poses_in(poses, n));
}

uint32_t sortham_utils_remove_poses_after_first_inversion(double * poses, uint32_t * n)
{
  auto v = poses_in(poses, *n);
  const uint32_t r = sortham_ns::removePosesAfterFirstInversion(v);
  *n = copy_out(v, poses, *n);
  return r;
}

int sortham_visualizer_create(const char * frame_id, int trajectory_step, int time_step,
  sortham_trajectory_visualizer ** out)
{
  if (!out || trajectory_step < 1 || time_step < 1) {
    return SMPC_ERR_INVALID;
  }
  auto * v = new sortham_trajectory_visualizer();
  v->frame = frame_id ? frame_id : "";
  v->v.on_configure(v->frame, trajectory_step, time_step);
  *out = v;
  return SMPC_OK;
}

void sortham_visualizer_destroy(sortham_trajectory_visualizer * v) {delete v;}

int sortham_visualizer_add_trajectory(sortham_trajectory_visualizer * v, const float * xy, uint32_t n, uint32_t stride)
{
  if (!v || (n && (!xy || stride < 2))) {
    return SMPC_ERR_INVALID;
  }
  v->v.add(xy, n, stride, "Optimal Trajectory");
  return SMPC_OK;
}

int sortham_visualizer_add_candidates(
  sortham_trajectory_visualizer * v, const float * x, const float * y, uint32_t B, uint32_t T)
{
  if (!v || ((B && T) && (!x || !y))) {
    return SMPC_ERR_INVALID;
  }
  v->v.add(x, y, B, T, "Candidate Trajectories");
  return SMPC_OK;
}

uint32_t sortham_visualizer_visualize(sortham_trajectory_visualizer * v, double * markers, uint32_t cap)
{
  if (!v) {
    return 0;
  }
  const auto ms = v->v.visualize();
  const uint32_t n = static_cast<uint32_t>(ms.size());
  for (uint32_t i = 0; i < n && i < cap && markers; ++i) {
    const auto & m = ms[i];
    double * o = markers + 10 * i;
    o[0] = m.id; o[1] = m.x; o[2] = m.y; o[3] = m.z;
    o[4] = m.scale_x; o[5] = m.scale_y; o[6] = m.scale_z;
    o[7] = m.g; o[8] = m.b; o[9] = m.a;
  }
  return n;
}

const char * sortham_visualizer_frame(const sortham_trajectory_visualizer * v) {return v ? v->frame.c_str() : "";}

}  // extern "C"
