// smpc_oracle.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Single-threaded CPU restatement ("oracle") of the hot path of
// soham2560/MPCHoloNavigation's nav2_sortham_controller: one
// Optimizer::optimize() call (noise add -> holonomic rollout -> critics ->
// softmax-weighted control update) plus the small host helpers around it.
// It follows the reference's pass structure, op order and float/double mix;
// every function cites the reference file:line it restates (paths relative to
// /root/reference/nav2_sortham_controller/).
//
// Who may use it: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg — as the checker / reported CPU baseline only.  Nothing under
// mpcholonavigation_amd/ links, loads or calls it.
//
// Pinning status: the reference itself cannot be built here (needs ROS 2
// Humble, nav2, xtensor, xsimd), so this restatement is
// pinned by the reference's own known-answer tests, re-encoded in
// tests/test_oracle_reference_kats.py (rollout integration, velocity
// propagation, clip, shift, speed limit, setOffset, GoalAngle 9.42,
// PreferForward 15.0, PathFollow 750, PathAlign 6600 / blocked 0, furthest and
// initial path point = 5, path validity, tolerance gate, angle range).
// UNPINNED (no value assertion exists in the reference): ObstaclesCritic
// arithmetic, the softmax update, Savitzky-Golay edge handling, the noise
// stream.  Those follow the reference source literally.
//
// Third-party arithmetic that is not under /root/reference is restated from
// its published behaviour: nav2_costmap_2d (ROS 2 Humble, Nav2 1.1.x)
// Costmap2D::worldToMap / getCost and the cost constants; tf2::getYaw is the
// caller's job (the ABI takes yaw); xtensor reductions are taken as
// sequential float accumulation along the reduced axis.
//
// Defined behaviour given to reference UB (SURVEY.md §8(a) H1,H2,H9):
//  - findClosestPathPt dereferences end() when dist is beyond the last
//    integrated distance: here it returns size-1.
//  - (unsigned)(double) casts beyond 2^32 in worldToMap: treated as off-map.

#include "smpc_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

namespace {

// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3",
// SC'11).  The reference draws from std::mt19937 + std::normal_distribution
// inside xt::random::randn (src/noise_generator.cpp:107-122), a serial stream
// that no test pins; the build defines its own counter-based stream instead and
// this is its CPU twin.
// ---------------------------------------------------------------------------
inline void philox_round(uint32_t c[4], const uint32_t k[2])
{
  const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c[0];
  const uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c[2];
  const uint32_t hi0 = static_cast<uint32_t>(p0 >> 32), lo0 = static_cast<uint32_t>(p0);
  const uint32_t hi1 = static_cast<uint32_t>(p1 >> 32), lo1 = static_cast<uint32_t>(p1);
  const uint32_t n0 = hi1 ^ c[1] ^ k[0];
  const uint32_t n2 = hi0 ^ c[3] ^ k[1];
  c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}

inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
  uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
  uint32_t k[2] = {key[0], key[1]};
  for (int r = 0; r < 10; ++r) {
    if (r > 0) { k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u; }
    philox_round(c, k);
  }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

// One N(0,1) sample for flat element e of stream s at draw epoch `epoch`.
// Elements 4q .. 4q + 3 share one Philox block, all four words used: words (0, 1) give the
// Box–Muller cos / sin pair of elements 4q, 4q + 1, words (2, 3) that of 4q + 2, 4q + 3.
inline float normal_sample(uint64_t seed, uint32_t stream, uint32_t epoch, uint64_t e)
{
  const uint64_t q = e >> 2;
  const uint32_t ctr[4] = {static_cast<uint32_t>(q), static_cast<uint32_t>(q >> 32), stream,
                           epoch};
  const uint32_t key[2] = {static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32)};
  uint32_t r[4];
  philox4x32_10(ctr, key, r);
  const uint32_t w = (e & 2) ? 2u : 0u;
  const float u1 = (static_cast<float>(r[w] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = (static_cast<float>(r[w + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float radius = sqrtf(-2.0f * logf(u1));
  const float ang = 6.2831853071795864769f * u2;
  return (e & 1) ? radius * sinf(ang) : radius * cosf(ang);
}

// ---------------------------------------------------------------------------
// nav2_costmap_2d restatement (Costmap2D::worldToMap / getCost).
// Call sites in the reference: src/critics/obstacles_critic.cpp:203-212,
// include/.../tools/utils.hpp:365-372.
// ---------------------------------------------------------------------------
struct Costmap {
  std::vector<uint8_t> cells;
  uint32_t W = 0, H = 0;
  double ox = 0, oy = 0, res = 1;
  bool track_unknown = false;
  float inscribed_radius = 0;
  float cost_scaling_factor = 0;  // ObstaclesCritic::inflation_scale_factor_
  float inflation_radius = 0;     // ObstaclesCritic::inflation_radius_
  bool set = false;
  // consider_footprint: robot footprint polygon (robot frame), the layered
  // costmap's circumscribed radius and the inflation layer's own cost_scaling_factor (< 0: none)
  std::vector<double> fp_x, fp_y;
  double circumscribed_radius = 0.0;
  double layer_cost_scaling_factor = -1.0;
};

inline bool world_to_map(const Costmap & c, double wx, double wy, unsigned & mx, unsigned & my)
{
  if (wx < c.ox || wy < c.oy) {
    return false;
  }
  const double qx = (wx - c.ox) / c.res;
  const double qy = (wy - c.oy) / c.res;
  if (!(qx < 4294967296.0) || !(qy < 4294967296.0)) {
    return false;  // cast would be UB in the reference; defined as off-map
  }
  mx = static_cast<unsigned>(qx);
  my = static_cast<unsigned>(qy);
  return mx < c.W && my < c.H;
}

inline uint8_t get_cost(const Costmap & c, unsigned mx, unsigned my)
{
  return c.cells[static_cast<size_t>(my) * c.W + mx];
}

// ---------------------------------------------------------------------------
// include/.../tools/utils.hpp helpers
// ---------------------------------------------------------------------------

// utils.hpp:233-249 withinPositionGoalTolerance(float, Pose, Pose)
inline bool within_position_goal_tolerance(float pose_tolerance, double rx, double ry, double gx,
                                           double gy)
{
  const double dist_sq = std::pow(gx - rx, 2) + std::pow(gy - ry, 2);
  const float pose_tolerance_sq = pose_tolerance * pose_tolerance;
  return dist_sq < pose_tolerance_sq;
}

// utils.hpp:258-263 normalize_angles: fmod(a + pi, 2pi); <= 0 ? +pi : -pi, in double
inline double normalize_angle(double a)
{
  const double theta = std::fmod(a + M_PI, 2.0 * M_PI);
  return theta <= 0.0 ? theta + M_PI : theta - M_PI;
}

// utils.hpp:665-675 findClosestPathPt (H1: end() defined as size-1)
inline size_t find_closest_path_pt(const float * vec, size_t n, float dist, size_t init)
{
  const float * first = vec + init;
  const float * last = vec + n;
  const float * iter = std::lower_bound(first, last, dist);
  if (iter == first) {
    return 0;
  }
  if (iter == last) {
    return n - 1;  // reference reads *end(): UB, defined here
  }
  if (dist - *(iter - 1) < *iter - dist) {
    return static_cast<size_t>(iter - 1 - vec);
  }
  return static_cast<size_t>(iter - vec);
}

// utils.hpp:292-319 findPathFurthestReachedPoint
inline size_t find_path_furthest_reached_point(const float * tx, const float * ty, size_t B,
                                               size_t T, const float * px, const float * py,
                                               size_t P)
{
  size_t max_id_by_trajectories = 0;
  if (T == 0) {
    return 0;
  }
  for (size_t i = 0; i < B; i++) {
    const float ex = tx[i * T + (T - 1)];
    const float ey = ty[i * T + (T - 1)];
    size_t min_id_by_path = 0;
    float min_distance_by_path = std::numeric_limits<float>::max();
    for (size_t j = 0; j < P; j++) {
      const float dx = px[j] - ex;
      const float dy = py[j] - ey;
      const float cur_dist = dx * dx + dy * dy;
      if (cur_dist < min_distance_by_path) {
        min_distance_by_path = cur_dist;
        min_id_by_path = j;
      }
    }
    max_id_by_trajectories = std::max(max_id_by_trajectories, min_id_by_path);
  }
  return max_id_by_trajectories;
}

// utils.hpp:327-344 findPathTrajectoryInitialPoint
inline size_t find_path_trajectory_initial_point(float x00, float y00, const float * px,
                                                 const float * py, size_t P)
{
  float min_distance_by_path = std::numeric_limits<float>::max();
  size_t min_id = 0;
  for (size_t j = 0; j < P; j++) {
    const float dx = px[j] - x00;
    const float dy = py[j] - y00;
    const float d = dx * dx + dy * dy;
    if (d < min_distance_by_path) {
      min_distance_by_path = d;
      min_id = j;
    }
  }
  return min_id;
}

// utils.hpp:361-395 findPathCosts
inline void find_path_costs(const Costmap & cm, const float * px, const float * py, size_t P,
                            std::vector<uint8_t> & valid)
{
  const size_t n = P > 0 ? P - 1 : 0;
  valid.assign(n, 0);
  for (size_t idx = 0; idx < n; idx++) {
    unsigned mx, my;
    if (!world_to_map(cm, px[idx], py[idx], mx, my)) {
      valid[idx] = 0;
      continue;
    }
    switch (get_cost(cm, mx, my)) {
      case SMPC_COST_LETHAL:
        valid[idx] = 0;
        continue;
      case SMPC_COST_INSCRIBED:
        valid[idx] = 0;
        continue;
      case SMPC_COST_NO_INFORMATION:
        valid[idx] = cm.track_unknown ? 1 : 0;
        continue;
    }
    valid[idx] = 1;
  }
}

// utils.hpp:442-605 savitskyGolayFilter (H6: in place, sequential, T-5 skipped)
inline void savitsky_golay(float * u, uint32_t T, float * hist /*4x3*/, bool shift)
{
  float filter[9] = {-21.0f, 14.0f, 39.0f, 54.0f, 59.0f, 54.0f, 39.0f, 14.0f, -21.0f};
  for (float & f : filter) {
    f /= 231.0f;
  }
  const unsigned int num_sequences = T - 1;
  if (T == 0 || num_sequences < 20) {
    return;
  }
  auto apply = [&](const float (&d)[9]) -> float {
      float s = 0.0f;
      for (int i = 0; i < 9; ++i) {
        s += d[i] * filter[i];
      }
      return s;
    };
  auto over_axis = [&](float * q, float h0, float h1, float h2, float h3) {
      unsigned int idx = 0;
      q[idx] = apply({h0, h1, h2, h3, q[idx], q[idx + 1], q[idx + 2], q[idx + 3], q[idx + 4]});
      idx++;
      q[idx] =
        apply({h1, h2, h3, q[idx - 1], q[idx], q[idx + 1], q[idx + 2], q[idx + 3], q[idx + 4]});
      idx++;
      q[idx] = apply(
        {h2, h3, q[idx - 2], q[idx - 1], q[idx], q[idx + 1], q[idx + 2], q[idx + 3], q[idx + 4]});
      idx++;
      q[idx] = apply(
        {h3, q[idx - 3], q[idx - 2], q[idx - 1], q[idx], q[idx + 1], q[idx + 2], q[idx + 3],
          q[idx + 4]});
      for (idx = 4; idx != num_sequences - 4; idx++) {
        q[idx] = apply(
          {q[idx - 4], q[idx - 3], q[idx - 2], q[idx - 1], q[idx], q[idx + 1], q[idx + 2],
            q[idx + 3], q[idx + 4]});
      }
      idx++;
      q[idx] = apply(
        {q[idx - 4], q[idx - 3], q[idx - 2], q[idx - 1], q[idx], q[idx + 1], q[idx + 2],
          q[idx + 3], q[idx + 3]});
      idx++;
      q[idx] = apply(
        {q[idx - 4], q[idx - 3], q[idx - 2], q[idx - 1], q[idx], q[idx + 1], q[idx + 2],
          q[idx + 2], q[idx + 2]});
      idx++;
      q[idx] = apply(
        {q[idx - 4], q[idx - 3], q[idx - 2], q[idx - 1], q[idx], q[idx + 1], q[idx + 1],
          q[idx + 1], q[idx + 1]});
      idx++;
      q[idx] = apply(
        {q[idx - 4], q[idx - 3], q[idx - 2], q[idx - 1], q[idx], q[idx], q[idx], q[idx], q[idx]});
    };
  float * vx = u, * vy = u + T, * wz = u + 2 * T;
  over_axis(vx, hist[0 * 3 + 0], hist[1 * 3 + 0], hist[2 * 3 + 0], hist[3 * 3 + 0]);
  over_axis(vy, hist[0 * 3 + 1], hist[1 * 3 + 1], hist[2 * 3 + 1], hist[3 * 3 + 1]);
  over_axis(wz, hist[0 * 3 + 2], hist[1 * 3 + 2], hist[2 * 3 + 2], hist[3 * 3 + 2]);
  const unsigned int offset = shift ? 1 : 0;
  for (int c = 0; c < 3; ++c) {
    hist[0 * 3 + c] = hist[1 * 3 + c];
    hist[1 * 3 + c] = hist[2 * 3 + c];
    hist[2 * 3 + c] = hist[3 * 3 + c];
  }
  hist[3 * 3 + 0] = vx[offset];
  hist[3 * 3 + 1] = vy[offset];
  hist[3 * 3 + 2] = wz[offset];
}

// xt::maximum(a, b) is select(a > b, a, b) (xtensor xmath.hpp, math::maximum; third-party,
// absent from the reference tree): a NaN first operand yields b.  Only the Ackermann
// turning-radius term can see a NaN (0/0 for a robot at rest, state.vx[:,0] = wz[:,0] = 0).
inline double xt_maximum(double a, double b) {return a > b ? a : b;}

inline float clipf(float v, float lo, float hi) {return v < lo ? lo : (v > hi ? hi : v);}

// MotionModel::applyConstraints: a no-op (include/.../motion_models.hpp:79) except for
// Ackermann (:110-117): where |vx|/|wz| < min_turning_r, wz = sign(wz) * |vx| / min_turning_r.
// xt::sign(0) is 0; |vx|/0 is +inf (or NaN for 0/0), never below the radius, so wz == 0 stays.
inline void motion_model_apply_constraints(float * u, uint32_t T, float ackermann_min_r)
{
  if (!(ackermann_min_r >= 0.0f)) {
    return;
  }
  for (uint32_t t = 0; t < T; ++t) {
    const float v = u[t], w = u[2 * T + t];
    if (fabsf(v) / fabsf(w) < ackermann_min_r) {
      const float sgn = w > 0.0f ? 1.0f : (w < 0.0f ? -1.0f : 0.0f);
      u[2 * T + t] = sgn * fabsf(v) / ackermann_min_r;
    }
  }
}

// src/optimizer.cpp:237-249 applyControlSequenceConstraints
inline void apply_constraints(float * u, uint32_t T, float vx_max, float vx_min, float vy,
                              float wz, bool holonomic = true, float ackermann_min_r = -1.0f)
{
  if (holonomic) {
    for (uint32_t t = 0; t < T; ++t) {
      u[T + t] = clipf(u[T + t], -vy, vy);
    }
  }
  for (uint32_t t = 0; t < T; ++t) {
    u[t] = clipf(u[t], vx_min, vx_max);
  }
  for (uint32_t t = 0; t < T; ++t) {
    u[2 * T + t] = clipf(u[2 * T + t], -wz, wz);
  }
  motion_model_apply_constraints(u, T, ackermann_min_r);  // :248
}

}  // namespace

// ---------------------------------------------------------------------------
// The oracle object: Optimizer + NoiseGenerator + CriticData state
// (include/.../optimizer.hpp:245-262, tools/noise_generator.hpp:97-99).
// ---------------------------------------------------------------------------
struct smpc_oracle {
  smpc_config cfg{};
  smpc_critic_params critics{};
  float c_vx_max = 0, c_vx_min = 0, c_vy = 0, c_wz = 0;  // settings_.constraints
  Costmap costmap;
  bool have_noise = false;
  bool rng_mode = false;
  uint64_t seed = 0;
  uint32_t epoch = 0;
  bool accumulate_double = false;
  bool log_float = false;   // distanceToObstacle's log() as the float overload (diagnostic)
  std::vector<float> nvx, nvy, nwz;                // NoiseGenerator::noises_*
  std::vector<float> vx, vy, wz, cvx, cvy, cwz;    // models::State
  std::vector<float> tx, ty, tyaw;                 // models::Trajectories
  std::vector<float> costs;                        // Optimizer::costs_
  // CriticData per-tick caches
  bool fail_flag = false;
  bool furthest_set = false;
  size_t furthest = 0;
  bool path_valid_set = false;
  std::vector<uint8_t> path_pts_valid;
  uint32_t non_colliding = 0;
  float last_min = 0, last_sumw = 0;
  std::string err;

  size_t B() const {return cfg.batch_size;}
  size_t T() const {return cfg.time_steps;}
  // Optimizer::isHolonomic (src/optimizer.cpp:235)
  bool holonomic() const {return cfg.motion_model == SMPC_MODEL_OMNI;}
  float ackermann_r() const
  {
    return cfg.motion_model == SMPC_MODEL_ACKERMANN ? cfg.ackermann_min_turning_r : -1.0f;
  }
};

namespace {

int fail(smpc_oracle * o, int code, const char * msg)
{
  if (o) {
    o->err = msg;
  }
  return code;
}

void draw_noise(smpc_oracle * o)
{
  // src/noise_generator.cpp:107-122: vx, then wz, then vy (holonomic)
  const size_t B = o->B(), T = o->T();
  const uint64_t base = o->cfg.shard_offset * T;
  for (size_t i = 0; i < B * T; ++i) {
    o->nvx[i] = normal_sample(o->seed, 0, o->epoch, base + i) * o->cfg.vx_std;
  }
  for (size_t i = 0; i < B * T; ++i) {
    o->nwz[i] = normal_sample(o->seed, 1, o->epoch, base + i) * o->cfg.wz_std;
  }
  if (o->holonomic()) {
    for (size_t i = 0; i < B * T; ++i) {
      o->nvy[i] = normal_sample(o->seed, 2, o->epoch, base + i) * o->cfg.vy_std;
    }
  }  // else noises_vy_ keeps the zeros of reset() (src/noise_generator.cpp:76-91,117)
  o->have_noise = true;
}

struct Tick {
  const smpc_tick_in * in;
  const float * px, * py, * pyaw;
  size_t P;
};

// NoiseGenerator::setNoisedControls (src/noise_generator.cpp:65-74)
void set_noised_controls(smpc_oracle * o, const float * u)
{
  const size_t B = o->B(), T = o->T();
  const float * uvx = u, * uvy = u + T, * uwz = u + 2 * T;
  for (size_t b = 0; b < B; ++b) {
    for (size_t t = 0; t < T; ++t) {
      o->cvx[b * T + t] = uvx[t] + o->nvx[b * T + t];
    }
  }
  for (size_t b = 0; b < B; ++b) {
    for (size_t t = 0; t < T; ++t) {
      o->cvy[b * T + t] = uvy[t] + o->nvy[b * T + t];
    }
  }
  for (size_t b = 0; b < B; ++b) {
    for (size_t t = 0; t < T; ++t) {
      o->cwz[b * T + t] = uwz[t] + o->nwz[b * T + t];
    }
  }
}

// Optimizer::updateStateVelocities (src/optimizer.cpp:251-273) +
// MotionModel::predict (include/.../motion_models.hpp:53-66); a non-holonomic model never
// writes state.vy, which keeps the zeros of State::reset
void update_state_velocities(smpc_oracle * o, const smpc_tick_in * in)
{
  const size_t B = o->B(), T = o->T();
  const float svx = static_cast<float>(in->speed_vx);
  const float swz = static_cast<float>(in->speed_wz);
  const float svy = static_cast<float>(in->speed_vy);
  for (size_t b = 0; b < B; ++b) {
    o->vx[b * T] = svx;
  }
  for (size_t b = 0; b < B; ++b) {
    o->wz[b * T] = swz;
  }
  if (o->holonomic()) {
    for (size_t b = 0; b < B; ++b) {
      o->vy[b * T] = svy;
    }
  }
  for (size_t b = 0; b < B; ++b) {
    for (size_t t = 1; t < T; ++t) {
      o->vx[b * T + t] = o->cvx[b * T + t - 1];
    }
  }
  for (size_t b = 0; b < B; ++b) {
    for (size_t t = 1; t < T; ++t) {
      o->wz[b * T + t] = o->cwz[b * T + t - 1];
    }
  }
  if (o->holonomic()) {
    for (size_t b = 0; b < B; ++b) {
      for (size_t t = 1; t < T; ++t) {
        o->vy[b * T + t] = o->cvy[b * T + t - 1];
      }
    }
  }
}

// Optimizer::integrateStateVelocities(Trajectories&, const State&)
// (src/optimizer.cpp:313-343)
void integrate_state_velocities(smpc_oracle * o, const smpc_tick_in * in)
{
  const size_t B = o->B(), T = o->T();
  const float dt = o->cfg.model_dt;
  const float initial_yaw = in->pose_yaw;
  std::vector<float> yaw_cos(B * T), yaw_sin(B * T), dx(B * T), dy(B * T);

  // :319-320 yaws = cumsum(wz * dt, 1) + initial_yaw
  for (size_t b = 0; b < B; ++b) {
    float acc = 0.0f;
    for (size_t t = 0; t < T; ++t) {
      const float inc = o->wz[b * T + t] * dt;
      acc = (t == 0) ? inc : acc + inc;
      o->tyaw[b * T + t] = acc + initial_yaw;
    }
  }
  // :326-329
  const float c0 = cosf(initial_yaw), s0 = sinf(initial_yaw);
  for (size_t b = 0; b < B; ++b) {
    yaw_cos[b * T] = c0;
    yaw_sin[b * T] = s0;
  }
  for (size_t b = 0; b < B; ++b) {
    for (size_t t = 1; t < T; ++t) {
      yaw_cos[b * T + t] = cosf(o->tyaw[b * T + t - 1]);
    }
  }
  for (size_t b = 0; b < B; ++b) {
    for (size_t t = 1; t < T; ++t) {
      yaw_sin[b * T + t] = sinf(o->tyaw[b * T + t - 1]);
    }
  }
  // :331-337
  for (size_t i = 0; i < B * T; ++i) {
    dx[i] = o->vx[i] * yaw_cos[i];
  }
  for (size_t i = 0; i < B * T; ++i) {
    dy[i] = o->vx[i] * yaw_sin[i];
  }
  if (o->holonomic()) {
    for (size_t i = 0; i < B * T; ++i) {
      dx[i] = dx[i] - o->vy[i] * yaw_sin[i];
    }
    for (size_t i = 0; i < B * T; ++i) {
      dy[i] = dy[i] + o->vy[i] * yaw_cos[i];
    }
  }
  // :339-342 x = position.x (double) + cumsum(dx * dt, 1), stored as float
  for (size_t b = 0; b < B; ++b) {
    float acc = 0.0f;
    for (size_t t = 0; t < T; ++t) {
      const float inc = dx[b * T + t] * dt;
      acc = (t == 0) ? inc : acc + inc;
      o->tx[b * T + t] = static_cast<float>(in->pose_x + static_cast<double>(acc));
    }
  }
  for (size_t b = 0; b < B; ++b) {
    float acc = 0.0f;
    for (size_t t = 0; t < T; ++t) {
      const float inc = dy[b * T + t] * dt;
      acc = (t == 0) ? inc : acc + inc;
      o->ty[b * T + t] = static_cast<float>(in->pose_y + static_cast<double>(acc));
    }
  }
}

// Optimizer::generateNoisedTrajectories (src/optimizer.cpp:227-233)
void generate_noised_trajectories(smpc_oracle * o, const smpc_tick_in * in, const float * u)
{
  set_noised_controls(o, u);
  update_state_velocities(o, in);
  integrate_state_velocities(o, in);
}

// data.costs += xt::pow(v, power): std::pow(float, unsigned) -> double (H5)
inline void add_cost_pow(float & c, double v, unsigned power)
{
  c = static_cast<float>(static_cast<double>(c) + std::pow(v, static_cast<double>(power)));
}

void set_path_furthest_if_not_set(smpc_oracle * o, const Tick & tk)
{
  // utils.hpp:350-355
  if (!o->furthest_set) {
    o->furthest = find_path_furthest_reached_point(
      o->tx.data(), o->ty.data(), o->B(), o->T(), tk.px, tk.py, tk.P);
    o->furthest_set = true;
  }
}

void set_path_costs_if_not_set(smpc_oracle * o, const Tick & tk)
{
  // utils.hpp:401-407
  if (!o->path_valid_set) {
    if (tk.in->path_pts_valid) {
      const size_t n = tk.P > 0 ? tk.P - 1 : 0;
      o->path_pts_valid.assign(tk.in->path_pts_valid, tk.in->path_pts_valid + n);
    } else {
      find_path_costs(o->costmap, tk.px, tk.py, tk.P, o->path_pts_valid);
    }
    o->path_valid_set = true;
  }
}


// ---- consider_footprint = true: nav2_costmap_2d (ROS 2 Humble, third party; restated from its
// published sources: footprint_collision_checker.cpp, nav2_util/line_iterator.hpp,
// inflation_layer.hpp).  The reference has no test on this path: parity unpinned beyond the
// GPU-vs-oracle comparison.

// InflationLayer::computeCost(distance in cells)
inline unsigned char inflation_compute_cost(const Costmap & cm, double distance)
{
  unsigned char cost = 0;
  if (distance == 0) {
    cost = SMPC_COST_LETHAL;
  } else if (distance * cm.res <= static_cast<double>(cm.inscribed_radius)) {
    cost = SMPC_COST_INSCRIBED;
  } else {
    const double factor =
      std::exp(-1.0 * cm.layer_cost_scaling_factor * (distance * cm.res - static_cast<double>(cm.inscribed_radius)));
    cost = static_cast<unsigned char>((SMPC_COST_INSCRIBED - 1) * factor);
  }
  return cost;
}

// {Obstacles,Cost}Critic::findCircumscribedCost (obstacles_critic.cpp:52-97, cost_critic.cpp:62-106)
inline float find_circumscribed_cost(const Costmap & cm)
{
  double result = -1.0;
  if (cm.layer_cost_scaling_factor >= 0.0) {
    result = inflation_compute_cost(cm, cm.circumscribed_radius / cm.res);
  }
  return static_cast<float>(result);
}

// FootprintCollisionChecker::lineCost with nav2_util::LineIterator
inline double line_cost(const Costmap & cm, int x0, int x1, int y0, int y1)
{
  double cost = 0.0;
  const int deltax = std::abs(x1 - x0), deltay = std::abs(y1 - y0);
  int x = x0, y = y0;
  int xinc1 = x1 >= x0 ? 1 : -1, xinc2 = xinc1;
  int yinc1 = y1 >= y0 ? 1 : -1, yinc2 = yinc1;
  int den, num, numadd, numpixels;
  if (deltax >= deltay) {
    xinc1 = 0;
    yinc2 = 0;
    den = deltax;
    num = deltax / 2;
    numadd = deltay;
    numpixels = deltax;
  } else {
    xinc2 = 0;
    yinc1 = 0;
    den = deltay;
    num = deltay / 2;
    numadd = deltax;
    numpixels = deltay;
  }
  for (int curpixel = 0; curpixel <= numpixels; ++curpixel) {
    const double point_cost = static_cast<double>(
      get_cost(cm, static_cast<unsigned>(x), static_cast<unsigned>(y)));
    if (point_cost == static_cast<double>(SMPC_COST_LETHAL)) {
      return point_cost;
    }
    if (cost < point_cost) {
      cost = point_cost;
    }
    num += numadd;
    if (num >= den) {
      num -= den;
      x += xinc1;
      y += yinc1;
    }
    x += xinc2;
    y += yinc2;
  }
  return cost;
}

// FootprintCollisionChecker::footprintCostAtPose + footprintCost
inline double footprint_cost_at_pose(const Costmap & cm, double x, double y, double theta)
{
  const size_t n = cm.fp_x.size();
  if (n == 0) {
    return static_cast<double>(SMPC_COST_LETHAL);
  }
  const double cos_th = std::cos(theta), sin_th = std::sin(theta);
  auto vertex = [&](size_t i, unsigned & mx, unsigned & my) {
      const double wx = x + (cm.fp_x[i] * cos_th - cm.fp_y[i] * sin_th);
      const double wy = y + (cm.fp_x[i] * sin_th + cm.fp_y[i] * cos_th);
      return world_to_map(cm, wx, wy, mx, my);
    };
  unsigned x0, y0, x1 = 0, y1 = 0;
  double footprint_cost = 0.0;
  if (!vertex(0, x0, y0)) {
    return static_cast<double>(SMPC_COST_LETHAL);
  }
  const unsigned xstart = x0, ystart = y0;
  x1 = x0;
  y1 = y0;
  for (size_t i = 0; i + 1 < n; ++i) {
    if (!vertex(i + 1, x1, y1)) {
      return static_cast<double>(SMPC_COST_LETHAL);
    }
    footprint_cost = std::max(
      line_cost(cm, static_cast<int>(x0), static_cast<int>(x1), static_cast<int>(y0),
      static_cast<int>(y1)), footprint_cost);
    x0 = x1;
    y0 = y1;
    if (footprint_cost == static_cast<double>(SMPC_COST_LETHAL)) {
      return footprint_cost;
    }
  }
  return std::max(
    line_cost(cm, static_cast<int>(xstart), static_cast<int>(x1), static_cast<int>(ystart),
    static_cast<int>(y1)), footprint_cost);
}

// ObstaclesCritic (src/critics/obstacles_critic.cpp:99-232)
void score_obstacles(smpc_oracle * o, const Tick & tk)
{
  const auto & p = o->critics.obstacles;
  if (!p.enabled) {
    return;
  }
  const size_t B = o->B(), T = o->T();
  const Costmap & cm = o->costmap;
  // :124-127
  const bool near_goal = within_position_goal_tolerance(
    p.near_goal_distance, tk.in->pose_x, tk.in->pose_y, tk.in->goal_x, tk.in->goal_y);
  // :119-122
  const float possibly_inscribed_cost = p.consider_footprint ? find_circumscribed_cost(cm) : 0.0f;
  std::vector<float> raw_cost(B, 0.0f), repulsive_cost(B, 0.0f);
  const size_t traj_len = T;
  bool all_trajectories_collide = true;
  uint32_t non_colliding = 0;
  for (size_t i = 0; i < B; ++i) {
    bool trajectory_collide = false;
    float traj_cost = 0.0f;
    for (size_t j = 0; j < traj_len; j++) {
      // costAtPose :203-224
      float cost;
      bool using_footprint = false;
      unsigned x_i, y_i;
      if (!world_to_map(cm, o->tx[i * T + j], o->ty[i * T + j], x_i, y_i)) {
        cost = SMPC_COST_NO_INFORMATION;
      } else {
        cost = static_cast<float>(get_cost(cm, x_i, y_i));
        if (p.consider_footprint &&
          (cost >= possibly_inscribed_cost || possibly_inscribed_cost < 1.0f))
        {
          cost = static_cast<float>(footprint_cost_at_pose(
              cm, o->tx[i * T + j], o->ty[i * T + j], o->tyaw[i * T + j]));
          using_footprint = true;
        }
      }
      if (cost < 1.0f) {continue;}
      // inCollision :185-201
      bool collide = false;
      switch (static_cast<unsigned char>(cost)) {
        case SMPC_COST_LETHAL:
          collide = true;
          break;
        case SMPC_COST_INSCRIBED:
          collide = p.consider_footprint ? false : true;
          break;
        case SMPC_COST_NO_INFORMATION:
          collide = cm.track_unknown ? false : true;
          break;
      }
      if (collide) {
        trajectory_collide = true;
        break;
      }
      // :154-157
      if (cm.inflation_radius == 0.0f || cm.cost_scaling_factor == 0.0f) {
        continue;
      }
      // distanceToObstacle :99-112.  The reference writes an unqualified log(float) after
      // #include <cmath>: with glibc's math.h that is ::log(double) — the default here — unless
      // some header of the translation unit pulled in libstdc++'s <math.h> wrapper, whose
      // using-declarations add ::log(float).  log_float restates that second reading (a
      // diagnostic: tests bound what the ambiguity can move).
      const float scale_factor = cm.cost_scaling_factor;
      const float min_radius = cm.inscribed_radius;
      float dist_to_obj;
      if (o->log_float) {
        dist_to_obj = (scale_factor * min_radius - std::log(cost) + std::log(253.0f)) / scale_factor;
      } else {
        dist_to_obj = static_cast<float>(
          (static_cast<double>(scale_factor * min_radius) - std::log(static_cast<double>(cost)) +
          std::log(static_cast<double>(253.0f))) / static_cast<double>(scale_factor));
      }
      if (!using_footprint) {
        dist_to_obj -= min_radius;   // :106-108
      }
      if (dist_to_obj < p.collision_margin_distance) {
        traj_cost += (p.collision_margin_distance - dist_to_obj);
      } else if (!near_goal) {
        repulsive_cost[i] += (cm.inflation_radius - dist_to_obj);
      }
    }
    if (!trajectory_collide) {
      all_trajectories_collide = false;
      non_colliding++;
    }
    raw_cost[i] = trajectory_collide ? p.collision_cost : traj_cost;
  }
  // :173-177
  for (size_t i = 0; i < B; ++i) {
    const float v = (p.critical_weight * raw_cost[i]) +
      (p.repulsion_weight * repulsive_cost[i] / static_cast<float>(traj_len));
    add_cost_pow(o->costs[i], v, p.cost_power);
  }
  o->fail_flag = all_trajectories_collide;
  o->non_colliding = non_colliding;
}

// PathAlignCritic (src/critics/path_align_critic.cpp:46-136)
void score_path_align(smpc_oracle * o, const Tick & tk)
{
  const auto & p = o->critics.path_align;
  if (!p.enabled || within_position_goal_tolerance(
      p.threshold_to_consider, tk.in->pose_x, tk.in->pose_y, tk.in->goal_x, tk.in->goal_y))
  {
    return;
  }
  set_path_furthest_if_not_set(o, tk);
  const size_t path_segments_count = o->furthest;
  if (path_segments_count < p.offset_from_furthest) {
    return;
  }
  set_path_costs_if_not_set(o, tk);
  const size_t B = o->B(), T = o->T();
  const size_t closest_initial_path_point =
    find_path_trajectory_initial_point(o->tx[0], o->ty[0], tk.px, tk.py, tk.P);
  unsigned int invalid_ctr = 0;
  const float range = static_cast<float>(o->furthest - closest_initial_path_point);
  for (size_t i = closest_initial_path_point; i < o->furthest; i++) {
    if (!o->path_pts_valid[i]) {invalid_ctr++;}
    if (static_cast<float>(invalid_ctr) / range > p.max_path_occupancy_ratio && invalid_ctr > 2) {
      return;
    }
  }
  // P_x / P_y / P_yaw = path without its last point
  const float * P_x = tk.px;
  const float * P_y = tk.py;
  const float * P_yaw = tk.pyaw;
  std::vector<float> cost(B, 0.0f);
  std::vector<float> path_integrated_distances(path_segments_count, 0.0f);
  float dx = 0.0f, dy = 0.0f;
  for (unsigned int i = 1; i < path_segments_count; i++) {
    dx = P_x[i] - P_x[i - 1];
    dy = P_y[i] - P_y[i - 1];
    const float curr_dist = sqrtf(dx * dx + dy * dy);
    path_integrated_distances[i] = path_integrated_distances[i - 1] + curr_dist;
  }
  const size_t step = p.trajectory_point_step;
  for (size_t t = 0; t < B; ++t) {
    float traj_integrated_distance = 0.0f;
    float summed_path_dist = 0.0f;
    float num_samples = 0.0f;
    size_t path_pt = 0u;
    const float * T_x = &o->tx[t * T];
    const float * T_y = &o->ty[t * T];
    const float * T_yaw = &o->tyaw[t * T];
    for (size_t q = step; step > 0 && q < T; q += step) {
      const float Tx = T_x[q];
      const float Ty = T_y[q];
      dx = Tx - T_x[q - step];
      dy = Ty - T_y[q - step];
      traj_integrated_distance += sqrtf(dx * dx + dy * dy);
      path_pt = find_closest_path_pt(
        path_integrated_distances.data(), path_integrated_distances.size(),
        traj_integrated_distance, path_pt);
      if (path_pt < o->path_pts_valid.size() && o->path_pts_valid[path_pt]) {
        dx = P_x[path_pt] - Tx;
        dy = P_y[path_pt] - Ty;
        num_samples += 1.0f;
        if (p.use_path_orientations) {
          // angles::shortest_angular_distance(from, to) = normalize_angle(to - from)
          // (ros/angles: normalize_angle_positive = fmod(fmod(a,2pi)+2pi,2pi)); unpinned
          const double d = static_cast<double>(T_yaw[q]) - static_cast<double>(P_yaw[path_pt]);
          double a = std::fmod(std::fmod(d, 2.0 * M_PI) + 2.0 * M_PI, 2.0 * M_PI);
          if (a > M_PI) {a -= 2.0 * M_PI;}
          const float dyaw = static_cast<float>(a);
          summed_path_dist += sqrtf(dx * dx + dy * dy + dyaw * dyaw);
        } else {
          summed_path_dist += sqrtf(dx * dx + dy * dy);
        }
      }
    }
    cost[t] = num_samples > 0 ? summed_path_dist / num_samples : 0.0f;
  }
  for (size_t i = 0; i < B; ++i) {
    add_cost_pow(o->costs[i], cost[i] * p.cost_weight, p.cost_power);
  }
}

// PathAlignLegacyCritic (src/critics/path_align_legacy_critic.cpp:46-129): the pre-October-2023
// formulation — every trajectory sample against its nearest path point, by brute force
void score_path_align_legacy(smpc_oracle * o, const Tick & tk)
{
  const auto & p = o->critics.path_align_legacy;
  // :48-54 not close to the goal
  if (!p.enabled || within_position_goal_tolerance(
      p.threshold_to_consider, tk.in->pose_x, tk.in->pose_y, tk.in->goal_x, tk.in->goal_y))
  {
    return;
  }
  // :56-60 not while first getting bearing w.r.t. the path
  set_path_furthest_if_not_set(o, tk);
  if (o->furthest < p.offset_from_furthest) {
    return;
  }
  // :62-72 not when obstacles block a significant part of the local path
  set_path_costs_if_not_set(o, tk);
  const size_t B = o->B(), T = o->T();
  const size_t closest_initial_path_point =
    find_path_trajectory_initial_point(o->tx[0], o->ty[0], tk.px, tk.py, tk.P);
  unsigned int invalid_ctr = 0;
  const float range = static_cast<float>(o->furthest - closest_initial_path_point);
  for (size_t i = closest_initial_path_point; i < o->furthest; i++) {
    if (!o->path_pts_valid[i]) {invalid_ctr++;}
    if (static_cast<float>(invalid_ctr) / range > p.max_path_occupancy_ratio && invalid_ctr > 2) {
      return;
    }
  }
  // :78-88 P_x / P_y / P_yaw = the path without its last point
  const float * P_x = tk.px;
  const float * P_y = tk.py;
  const float * P_yaw = tk.pyaw;
  const size_t step = p.trajectory_point_step;
  if (step == 0 || tk.P < 1) {return;}     // (floor(T / 0) in the reference: undefined; a path tensor always has a point)
  const size_t traj_pts_eval = T / step;   // floor(time_steps / trajectory_point_step_)
  const size_t path_segments_count = tk.P - 1;
  if (path_segments_count < 1) {
    return;
  }
  std::vector<float> cost(B, 0.0f);
  float dist_sq = 0.0f, dx = 0.0f, dy = 0.0f, dyaw = 0.0f, summed_dist = 0.0f;
  float min_dist_sq = std::numeric_limits<float>::max();
  size_t min_s = 0;
  for (size_t t = 0; t < B; ++t) {
    summed_dist = 0.0f;
    const float * T_x = &o->tx[t * T];
    const float * T_y = &o->ty[t * T];
    const float * T_yaw = &o->tyaw[t * T];
    for (size_t q = step; q < T; q += step) {
      min_dist_sq = std::numeric_limits<float>::max();
      min_s = 0;
      // :100-114 the closest path point (the loop stops one short of the last segment's start)
      for (size_t s = 0; s + 1 < path_segments_count; s++) {
        dx = P_x[s] - T_x[q];
        dy = P_y[s] - T_y[q];
        if (p.use_path_orientations) {
          // angles::shortest_angular_distance(from, to) = normalize_angle(to - from) (ros/angles; unpinned)
          const double d = static_cast<double>(T_yaw[q]) - static_cast<double>(P_yaw[s]);
          double a = std::fmod(std::fmod(d, 2.0 * M_PI) + 2.0 * M_PI, 2.0 * M_PI);
          if (a > M_PI) {a -= 2.0 * M_PI;}
          dyaw = static_cast<float>(a);
          dist_sq = dx * dx + dy * dy + dyaw * dyaw;
        } else {
          dist_sq = dx * dx + dy * dy;
        }
        if (dist_sq < min_dist_sq) {
          min_dist_sq = dist_sq;
          min_s = s;
        }
      }
      // :116-121 the point must not be in collision (and point 0 never counts)
      if (min_s != 0 && o->path_pts_valid[min_s]) {
        summed_dist += sqrtf(min_dist_sq);
      }
    }
    cost[t] = summed_dist / traj_pts_eval;
  }
  for (size_t i = 0; i < B; ++i) {
    add_cost_pow(o->costs[i], cost[i] * p.cost_weight, p.cost_power);
  }
}

// PathFollowCritic (src/critics/path_follow_critic.cpp:35-71)
void score_path_follow(smpc_oracle * o, const Tick & tk)
{
  const auto & p = o->critics.path_follow;
  if (!p.enabled || tk.P < 2 || within_position_goal_tolerance(
      p.threshold_to_consider, tk.in->pose_x, tk.in->pose_y, tk.in->goal_x, tk.in->goal_y))
  {
    return;
  }
  set_path_furthest_if_not_set(o, tk);
  set_path_costs_if_not_set(o, tk);
  const size_t B = o->B(), T = o->T();
  const size_t path_size = tk.P - 1;
  size_t offseted_idx = std::min(o->furthest + p.offset_from_furthest, path_size);
  bool valid = false;
  while (!valid && offseted_idx < path_size - 1) {
    valid = o->path_pts_valid[offseted_idx];
    if (!valid) {
      offseted_idx++;
    }
  }
  const float path_x = tk.px[offseted_idx];
  const float path_y = tk.py[offseted_idx];
  for (size_t i = 0; i < B; ++i) {
    // xt::pow(float, 2) -> double (H5)
    const double ddx = static_cast<double>(o->tx[i * T + T - 1] - path_x);
    const double ddy = static_cast<double>(o->ty[i * T + T - 1] - path_y);
    const double dist = std::sqrt(std::pow(ddx, 2) + std::pow(ddy, 2));
    add_cost_pow(o->costs[i], static_cast<double>(p.cost_weight) * dist, p.cost_power);
  }
}

// GoalAngleCritic (src/critics/goal_angle_critic.cpp:36-50)
void score_goal_angle(smpc_oracle * o, const Tick & tk)
{
  const auto & p = o->critics.goal_angle;
  if (!p.enabled || !within_position_goal_tolerance(
      p.threshold_to_consider, tk.in->pose_x, tk.in->pose_y, tk.in->goal_x, tk.in->goal_y))
  {
    return;
  }
  if (tk.P == 0) {
    return;
  }
  const size_t B = o->B(), T = o->T();
  const float goal_yaw = tk.pyaw[tk.P - 1];
  for (size_t i = 0; i < B; ++i) {
    double s = 0.0;
    for (size_t t = 0; t < T; ++t) {
      // shortest_angular_distance(from = yaws, to = goal_yaw) -> normalize(to - from)
      const float diff = goal_yaw - o->tyaw[i * T + t];
      s += std::fabs(normalize_angle(static_cast<double>(diff)));
    }
    const double mean = s / static_cast<double>(T);
    add_cost_pow(o->costs[i], mean * static_cast<double>(p.cost_weight), p.cost_power);
  }
}

// PreferForwardCritic (src/critics/prefer_forward_critic.cpp:33-47)
void score_prefer_forward(smpc_oracle * o, const Tick & tk)
{
  const auto & p = o->critics.prefer_forward;
  if (!p.enabled || within_position_goal_tolerance(
      p.threshold_to_consider, tk.in->pose_x, tk.in->pose_y, tk.in->goal_x, tk.in->goal_y))
  {
    return;
  }
  const size_t B = o->B(), T = o->T();
  const float dt = o->cfg.model_dt;
  for (size_t i = 0; i < B; ++i) {
    float s = 0.0f;
    for (size_t t = 0; t < T; ++t) {
      const float back = std::max(-o->vx[i * T + t], 0.0f);
      const float term = back * dt;
      s = (t == 0) ? term : s + term;
    }
    add_cost_pow(o->costs[i], s * p.cost_weight, p.cost_power);
  }
}


// ---- the other registered critics -------------------------------

// CostCritic (src/critics/cost_critic.cpp:108-168), consider_footprint = false.
// weight_ is divided by 254 at initialize() (:34).
void score_cost(smpc_oracle * o, const Tick & tk)
{
  const auto & p = o->critics.cost;
  if (!p.enabled) {
    return;
  }
  const size_t B = o->B(), T = o->T();
  const Costmap & cm = o->costmap;
  const float weight = p.cost_weight / 254.0f;
  // :120-124
  const bool near_goal = within_position_goal_tolerance(
    p.near_goal_distance, tk.in->pose_x, tk.in->pose_y, tk.in->goal_x, tk.in->goal_y);
  // :114-117,44 (possibly_inscribed_cost_ is refreshed when consider_footprint)
  const float possibly_inscribed_cost_c = p.consider_footprint ? find_circumscribed_cost(cm) : 0.0f;
  std::vector<float> repulsive_cost(B, 0.0f);
  const size_t traj_len = T;
  bool all_trajectories_collide = true;
  uint32_t non_colliding = 0;
  for (size_t i = 0; i < B; ++i) {
    bool trajectory_collide = false;
    for (size_t j = 0; j < traj_len; j++) {
      // costAtPose :203-212
      float pose_cost;
      unsigned x_i, y_i;
      if (!world_to_map(cm, o->tx[i * T + j], o->ty[i * T + j], x_i, y_i)) {
        pose_cost = SMPC_COST_NO_INFORMATION;
      } else {
        pose_cost = static_cast<float>(get_cost(cm, x_i, y_i));
      }
      if (pose_cost < 1.0f) {continue;}  // in free space
      // inCollision :175-201
      float check_cost = pose_cost;
      if (p.consider_footprint &&
        (check_cost >= possibly_inscribed_cost_c || possibly_inscribed_cost_c < 1.0f))
      {
        check_cost = static_cast<float>(footprint_cost_at_pose(
            cm, o->tx[i * T + j], o->ty[i * T + j], o->tyaw[i * T + j]));
      }
      bool collide = false;
      switch (static_cast<unsigned char>(check_cost)) {
        case SMPC_COST_LETHAL:
          collide = true;
          break;
        case SMPC_COST_INSCRIBED:
          collide = p.consider_footprint ? false : true;
          break;
        case SMPC_COST_NO_INFORMATION:
          collide = cm.track_unknown ? false : true;
          break;
      }
      if (collide) {
        trajectory_collide = true;
        break;
      }
      // :149-155
      if (pose_cost >= static_cast<float>(SMPC_COST_INSCRIBED)) {
        repulsive_cost[i] += p.critical_cost;
      } else if (!near_goal) {
        repulsive_cost[i] += pose_cost;
      }
    }
    if (!trajectory_collide) {
      all_trajectories_collide = false;
      non_colliding++;
    } else {
      repulsive_cost[i] = p.collision_cost;
    }
  }
  // :165-166
  for (size_t i = 0; i < B; ++i) {
    const float v = weight * repulsive_cost[i] / static_cast<float>(traj_len);
    add_cost_pow(o->costs[i], v, p.cost_power);
  }
  o->fail_flag = all_trajectories_collide;
  o->non_colliding = non_colliding;
}

// GoalCritic (src/critics/goal_critic.cpp:36-55): goal.position is double, so the
// distances are formed in double; xt::mean accumulates in double
void score_goal(smpc_oracle * o, const Tick & tk)
{
  const auto & p = o->critics.goal;
  if (!p.enabled || !within_position_goal_tolerance(
      p.threshold_to_consider, tk.in->pose_x, tk.in->pose_y, tk.in->goal_x, tk.in->goal_y))
  {
    return;
  }
  const size_t B = o->B(), T = o->T();
  const double goal_x = tk.in->goal_x, goal_y = tk.in->goal_y;
  for (size_t i = 0; i < B; ++i) {
    double s = 0.0;
    for (size_t t = 0; t < T; ++t) {
      const double dx = static_cast<double>(o->tx[i * T + t]) - goal_x;
      const double dy = static_cast<double>(o->ty[i * T + t]) - goal_y;
      s += std::sqrt(std::pow(dx, 2) + std::pow(dy, 2));
    }
    const double mean = s / static_cast<double>(T);
    add_cost_pow(o->costs[i], mean * static_cast<double>(p.cost_weight), p.cost_power);
  }
}

// ConstraintCritic (src/critics/constraint_critic.cpp:41-75), holonomic model (no
// Ackermann term).  xt::where(vx > 0.0, 1.0, -1.0) is a double tensor: the bound
// violations are formed in double.
void score_constraint(smpc_oracle * o, const Tick &)
{
  const auto & p = o->critics.constraint;
  if (!p.enabled) {
    return;
  }
  const size_t B = o->B(), T = o->T();
  // initialize() :36-38
  const float min_sgn = p.vx_min > 0.0f ? 1.0f : -1.0f;
  const float max_vel = sqrtf(p.vx_max * p.vx_max + p.vy_max * p.vy_max);
  const float min_vel = min_sgn * sqrtf(p.vx_min * p.vx_min + p.vy_max * p.vy_max);
  const float dt = o->cfg.model_dt;
  const float acker_r = o->ackermann_r();
  for (size_t i = 0; i < B; ++i) {
    double s = 0.0;
    for (size_t t = 0; t < T; ++t) {
      const float vx = o->vx[i * T + t], vy = o->vy[i * T + t];
      const double sgn = vx > 0.0f ? 1.0 : -1.0;
      const double vel_total = sgn * static_cast<double>(std::sqrt(vx * vx + vy * vy));
      const double out_max = std::max(vel_total - static_cast<double>(max_vel), 0.0);
      const double out_min = std::max(static_cast<double>(min_vel) - vel_total, 0.0);
      if (acker_r >= 0.0f) {
        // :54-59 xt::maximum(min_turning_r - fabs(vx) / fabs(wz), 0.0): float quotient,
        // double maximum; |vx|/0 = +inf gives 0, 0/0 = NaN propagates like xt's maximum
        const float q = acker_r - std::fabs(vx) / std::fabs(o->wz[i * T + t]);
        const double out_rad = xt_maximum(static_cast<double>(q), 0.0);
        s += (out_max + out_min + out_rad) * static_cast<double>(dt);
        continue;
      }
      s += (out_max + out_min) * static_cast<double>(dt);
    }
    add_cost_pow(o->costs[i], s * static_cast<double>(p.cost_weight), p.cost_power);
  }
}

// TwirlingCritic (src/critics/twirling_critic.cpp:30-42); the gate is
// utils::withinPositionGoalTolerance(goal_checker, ...) (tools/utils.hpp:201-224)
void score_twirling(smpc_oracle * o, const Tick & tk)
{
  const auto & p = o->critics.twirling;
  if (!p.enabled) {
    return;
  }
  if (tk.in->goal_checker_xy_tolerance >= 0.0f) {
    // pose_tolerance.position.x is a double in the message
    const double tol = static_cast<double>(tk.in->goal_checker_xy_tolerance);
    const double dx = tk.in->pose_x - tk.in->goal_x, dy = tk.in->pose_y - tk.in->goal_y;
    if (dx * dx + dy * dy < tol * tol) {
      return;
    }
  }
  const size_t B = o->B(), T = o->T();
  for (size_t i = 0; i < B; ++i) {
    double s = 0.0;
    for (size_t t = 0; t < T; ++t) {
      s += static_cast<double>(std::fabs(o->wz[i * T + t]));
    }
    const double mean = s / static_cast<double>(T);
    add_cost_pow(o->costs[i], mean * static_cast<double>(p.cost_weight), p.cost_power);
  }
}

// utils::posePointAngle (tools/utils.hpp:417-434); angles::shortest_angular_distance and
// angles::normalize_angle are ros-humble-angles (third party): normalize_angle(to - from)
// with normalize_angle(a) = fmod(a + pi, 2 pi) folded to (-pi, pi]
float pose_point_angle(double pose_xd, double pose_yd, float pose_yaw, double point_x,
                       double point_y, bool forward_preference)
{
  const float pose_x = static_cast<float>(pose_xd), pose_y = static_cast<float>(pose_yd);
  const float yaw = atan2f(static_cast<float>(point_y - static_cast<double>(pose_y)),
                           static_cast<float>(point_x - static_cast<double>(pose_x)));
  if (!forward_preference) {
    const double a = std::fabs(normalize_angle(static_cast<double>(pose_yaw) - static_cast<double>(yaw)));
    const double flipped = normalize_angle(static_cast<double>(pose_yaw) + M_PI);
    const double b = std::fabs(normalize_angle(flipped - static_cast<double>(yaw)));
    return static_cast<float>(std::min(a, b));
  }
  return static_cast<float>(
    std::fabs(normalize_angle(static_cast<double>(pose_yaw) - static_cast<double>(yaw))));
}

// PathAngleCritic (src/critics/path_angle_critic.cpp:58-101)
void score_path_angle(smpc_oracle * o, const Tick & tk)
{
  const auto & p = o->critics.path_angle;
  if (!p.enabled || tk.P == 0) {
    return;
  }
  if (within_position_goal_tolerance(
      p.threshold_to_consider, tk.in->pose_x, tk.in->pose_y, tk.in->goal_x, tk.in->goal_y))
  {
    return;
  }
  // initialize() :24-31,52-54
  bool reversing_allowed = true;
  if (std::fabs(p.vx_min) < 1e-6) {
    reversing_allowed = false;
  } else if (p.vx_min < 0.0f) {
    reversing_allowed = true;
  }
  bool forward_preference = p.forward_preference != 0;
  if (!reversing_allowed) {
    forward_preference = true;
  }
  set_path_furthest_if_not_set(o, tk);
  const size_t offseted_idx = std::min(o->furthest + p.offset_from_furthest, tk.P - 1);
  const float goal_x = tk.px[offseted_idx];
  const float goal_y = tk.py[offseted_idx];
  if (pose_point_angle(tk.in->pose_x, tk.in->pose_y, tk.in->pose_yaw, goal_x, goal_y,
      forward_preference) < p.max_angle_to_furthest)
  {
    return;
  }
  const size_t B = o->B(), T = o->T();
  const bool correct = reversing_allowed && !forward_preference;
  for (size_t i = 0; i < B; ++i) {
    double s = 0.0;
    for (size_t t = 0; t < T; ++t) {
      // float tensors: atan2 in float; shortest_angular_distance adds M_PI: double from there
      const float ybp = std::atan2(goal_y - o->ty[i * T + t], goal_x - o->tx[i * T + t]);
      const float yaw = o->tyaw[i * T + t];
      const double d = std::fabs(normalize_angle(static_cast<double>(ybp - yaw)));
      if (correct) {
        const double ybp_c = d < M_PI_2 ? static_cast<double>(ybp) :
          normalize_angle(static_cast<double>(ybp) + M_PI);
        s += std::fabs(normalize_angle(ybp_c - static_cast<double>(yaw)));
      } else {
        s += d;
      }
    }
    const double mean = s / static_cast<double>(T);
    add_cost_pow(o->costs[i], mean * static_cast<double>(p.cost_weight), p.cost_power);
  }
}

// VelocityDeadbandCritic (src/critics/velocity_deadband_critic.cpp:41-98), holonomic
// branch.  The unqualified fabs(float) resolves to ::fabs(double), so
// fabs(deadband) - xt::fabs(v) and everything after it is double arithmetic; the sum is
// added to the float cost in double (xtensor evaluates `costs += expr` element-wise).
void score_velocity_deadband(smpc_oracle * o, const Tick &)
{
  const auto & p = o->critics.velocity_deadband;
  if (!p.enabled) {
    return;
  }
  const size_t B = o->B(), T = o->T();
  const double dt = static_cast<double>(o->cfg.model_dt);
  const double d0 = std::fabs(static_cast<double>(p.deadband_velocities[0]));
  const double d1 = std::fabs(static_cast<double>(p.deadband_velocities[1]));
  const double d2 = std::fabs(static_cast<double>(p.deadband_velocities[2]));
  for (size_t i = 0; i < B; ++i) {
    double s = 0.0;
    for (size_t t = 0; t < T; ++t) {
      if (!o->holonomic()) {  // :78-97, no vy term
        s += (std::max(d0 - static_cast<double>(std::fabs(o->vx[i * T + t])), 0.0) +
          std::max(d2 - static_cast<double>(std::fabs(o->wz[i * T + t])), 0.0)) * dt;
        continue;
      }
      s += (std::max(d0 - static_cast<double>(std::fabs(o->vx[i * T + t])), 0.0) +
        std::max(d1 - static_cast<double>(std::fabs(o->vy[i * T + t])), 0.0) +
        std::max(d2 - static_cast<double>(std::fabs(o->wz[i * T + t])), 0.0)) * dt;
    }
    add_cost_pow(o->costs[i], s * static_cast<double>(p.cost_weight), p.cost_power);
  }
}

// The scoring order of include/smpc.h; `collision_critics_only_up_to_fail`: the re-score
// after the whole batch was found to collide scores the list up to and including the first
// enabled collision critic (critic_manager.cpp:70-73 stops there)
using CriticFn = void (*)(smpc_oracle *, const Tick &);
struct CriticEntry {
  CriticFn fn;
  bool collision;   // sets fail_flag
};
const CriticEntry kCriticOrder[12] = {
  {score_constraint, false}, {score_cost, true}, {score_obstacles, true},
  {score_path_align, false}, {score_path_align_legacy, false}, {score_path_follow, false}, {score_goal_angle, false},
  {score_prefer_forward, false}, {score_goal, false}, {score_path_angle, false},
  {score_twirling, false}, {score_velocity_deadband, false}};
bool collision_critic_enabled(const smpc_oracle * o, CriticFn fn)
{
  return (fn == score_cost && o->critics.cost.enabled) ||
         (fn == score_obstacles && o->critics.obstacles.enabled);
}

// CriticManager::evalTrajectoriesScores (src/critic_manager.cpp:67-76)
void eval_trajectories_scores(smpc_oracle * o, const Tick & tk)
{
  for (const CriticEntry & e : kCriticOrder) {
    if (o->fail_flag) {
      break;
    }
    e.fn(o, tk);
  }
}

// first half of Optimizer::updateControlSequence (src/optimizer.cpp:365-380)
void add_gamma_terms(smpc_oracle * o, const float * u)
{
  const size_t B = o->B(), T = o->T();
  const float * uvx = u, * uvy = u + T, * uwz = u + 2 * T;
  const float gvx = o->cfg.gamma / powf(o->cfg.vx_std, 2);
  const float gwz = o->cfg.gamma / powf(o->cfg.wz_std, 2);
  const float gvy = o->cfg.gamma / powf(o->cfg.vy_std, 2);
  for (size_t b = 0; b < B; ++b) {
    float s = 0.0f;
    for (size_t t = 0; t < T; ++t) {
      const float term = uvx[t] * (o->cvx[b * T + t] - uvx[t]);
      s = (t == 0) ? term : s + term;
    }
    o->costs[b] += gvx * s;
  }
  for (size_t b = 0; b < B; ++b) {
    float s = 0.0f;
    for (size_t t = 0; t < T; ++t) {
      const float term = uwz[t] * (o->cwz[b * T + t] - uwz[t]);
      s = (t == 0) ? term : s + term;
    }
    o->costs[b] += gwz * s;
  }
  if (!o->holonomic()) {
    return;
  }
  for (size_t b = 0; b < B; ++b) {
    float s = 0.0f;
    for (size_t t = 0; t < T; ++t) {
      const float term = uvy[t] * (o->cvy[b * T + t] - uvy[t]);
      s = (t == 0) ? term : s + term;
    }
    o->costs[b] += gvy * s;
  }
}

// Optimizer::updateControlSequence (src/optimizer.cpp:362-394)
void update_control_sequence(smpc_oracle * o, float * u)
{
  const size_t B = o->B(), T = o->T();
  add_gamma_terms(o, u);
  float cmin = std::numeric_limits<float>::max();
  for (size_t b = 0; b < B; ++b) {
    cmin = std::min(cmin, o->costs[b]);
  }
  std::vector<float> w(B);
  const float neg_inv_t = -1 / o->cfg.temperature;
  for (size_t b = 0; b < B; ++b) {
    w[b] = expf(neg_inv_t * (o->costs[b] - cmin));
  }
  float * uvx = u, * uvy = u + T, * uwz = u + 2 * T;
  if (o->accumulate_double) {
    double sum = 0.0;
    for (size_t b = 0; b < B; ++b) {
      sum += w[b];
    }
    std::vector<double> ax(T, 0.0), ay(T, 0.0), az(T, 0.0);
    for (size_t b = 0; b < B; ++b) {
      const double sm = static_cast<double>(w[b]) / sum;
      for (size_t t = 0; t < T; ++t) {
        ax[t] += o->cvx[b * T + t] * sm;
        ay[t] += o->cvy[b * T + t] * sm;
        az[t] += o->cwz[b * T + t] * sm;
      }
    }
    for (size_t t = 0; t < T; ++t) {
      uvx[t] = static_cast<float>(ax[t]);
      if (o->holonomic()) {
        uvy[t] = static_cast<float>(ay[t]);
      }
      uwz[t] = static_cast<float>(az[t]);
    }
    o->last_sumw = static_cast<float>(sum);
  } else {
    float sum = 0.0f;
    for (size_t b = 0; b < B; ++b) {
      sum += w[b];
    }
    for (size_t b = 0; b < B; ++b) {
      w[b] = w[b] / sum;  // softmaxes
    }
    std::vector<float> ax(T, 0.0f), az(T, 0.0f), ay(T, 0.0f);
    for (size_t b = 0; b < B; ++b) {
      for (size_t t = 0; t < T; ++t) {
        ax[t] += o->cvx[b * T + t] * w[b];
      }
    }
    for (size_t b = 0; b < B; ++b) {
      for (size_t t = 0; t < T; ++t) {
        az[t] += o->cwz[b * T + t] * w[b];
      }
    }
    for (size_t b = 0; b < B; ++b) {
      for (size_t t = 0; t < T; ++t) {
        ay[t] += o->cvy[b * T + t] * w[b];
      }
    }
    for (size_t t = 0; t < T; ++t) {
      uvx[t] = ax[t];
      if (o->holonomic()) {
        uvy[t] = ay[t];
      }
      uwz[t] = az[t];
    }
    o->last_sumw = sum;
  }
  o->last_min = cmin;
  apply_constraints(u, static_cast<uint32_t>(T), o->c_vx_max, o->c_vx_min, o->c_vy, o->c_wz,
                    o->holonomic(), o->ackermann_r());
}

int check_ready(smpc_oracle * o, const smpc_tick_in * in)
{
  if (!o || !in) {
    return SMPC_ERR_INVALID;
  }
  if (!o->have_noise) {
    return fail(o, SMPC_ERR_STATE, "no noise: call smpc_oracle_set_noise or smpc_oracle_seed");
  }
  if (in->path_len > 0 && (!in->path_x || !in->path_y || !in->path_yaw)) {
    return fail(o, SMPC_ERR_INVALID, "path arrays missing");
  }
  if (!o->costmap.set &&
    (o->critics.obstacles.enabled || o->critics.cost.enabled || !in->path_pts_valid))
  {
    return fail(o, SMPC_ERR_STATE, "no costmap");
  }
  if (((o->critics.obstacles.enabled && o->critics.obstacles.consider_footprint) ||
    (o->critics.cost.enabled && o->critics.cost.consider_footprint)) && o->costmap.fp_x.empty())
  {
    return fail(o, SMPC_ERR_STATE, "consider_footprint=true needs a footprint (smpc_oracle_set_footprint)");
  }
  return SMPC_OK;
}

void prepare(smpc_oracle * o, const smpc_tick_in * in)
{
  // Optimizer::prepare (src/optimizer.cpp:185-204)
  std::fill(o->costs.begin(), o->costs.end(), 0.0f);
  o->fail_flag = in->fail_flag_in != 0;
  o->furthest_set = false;
  o->path_valid_set = false;
  o->furthest = 0;
  o->non_colliding = 0;
}

}  // namespace

extern "C" {

const char * smpc_oracle_build_info(void)
{
#ifdef __FAST_MATH__
  return "smpc oracle (CPU restatement), -ffast-math build (reference flags; timing baseline)";
#else
  return "smpc oracle (CPU restatement), strict IEEE build (parity checker)";
#endif
}

int smpc_oracle_create(const smpc_config * cfg, smpc_oracle ** out)
{
  if (!cfg || !out || cfg->batch_size == 0 || cfg->time_steps == 0) {
    return SMPC_ERR_INVALID;
  }
  if (cfg->motion_model > SMPC_MODEL_ACKERMANN) {
    return SMPC_ERR_UNSUPPORTED;
  }
  if (cfg->motion_model == SMPC_MODEL_ACKERMANN && !(cfg->ackermann_min_turning_r >= 0.0f)) {
    return SMPC_ERR_INVALID;
  }
  smpc_oracle * o = new (std::nothrow) smpc_oracle();
  if (!o) {
    return SMPC_ERR_NOMEM;
  }
  o->cfg = *cfg;
  smpc_critic_params_default(&o->critics);
  o->c_vx_max = cfg->vx_max;
  o->c_vx_min = cfg->vx_min;
  o->c_vy = cfg->vy_max;
  o->c_wz = cfg->wz_max;
  const size_t n = static_cast<size_t>(cfg->batch_size) * cfg->time_steps;
  for (auto * v : {&o->nvx, &o->nvy, &o->nwz, &o->vx, &o->vy, &o->wz, &o->cvx, &o->cvy, &o->cwz,
      &o->tx, &o->ty, &o->tyaw})
  {
    v->assign(n, 0.0f);
  }
  o->costs.assign(cfg->batch_size, 0.0f);
  *out = o;
  return SMPC_OK;
}

void smpc_oracle_destroy(smpc_oracle * o) {delete o;}

const char * smpc_oracle_last_error(const smpc_oracle * o) {return o ? o->err.c_str() : "";}

int smpc_oracle_reset(smpc_oracle * o)
{
  if (!o) {return SMPC_ERR_INVALID;}
  // Optimizer::reset (src/optimizer.cpp:116-132)
  for (auto * v : {&o->vx, &o->vy, &o->wz, &o->cvx, &o->cvy, &o->cwz, &o->tx, &o->ty, &o->tyaw}) {
    std::fill(v->begin(), v->end(), 0.0f);
  }
  std::fill(o->costs.begin(), o->costs.end(), 0.0f);
  o->c_vx_max = o->cfg.vx_max;
  o->c_vx_min = o->cfg.vx_min;
  o->c_vy = o->cfg.vy_max;
  o->c_wz = o->cfg.wz_max;
  if (o->rng_mode) {
    o->epoch++;
    draw_noise(o);
  }
  return SMPC_OK;
}

int smpc_oracle_set_constraints(smpc_oracle * o, float vx_max, float vx_min, float vy_max,
                                float wz_max)
{
  if (!o) {return SMPC_ERR_INVALID;}
  o->c_vx_max = vx_max;
  o->c_vx_min = vx_min;
  o->c_vy = vy_max;
  o->c_wz = wz_max;
  return SMPC_OK;
}

int smpc_oracle_set_critics(smpc_oracle * o, const smpc_critic_params * p)
{
  if (!o || !p) {return SMPC_ERR_INVALID;}
  o->critics = *p;
  return SMPC_OK;
}

int smpc_oracle_set_costmap(smpc_oracle * o, const uint8_t * cells, uint32_t width,
                            uint32_t height, double origin_x, double origin_y,
                            double resolution, int track_unknown, float inscribed_radius,
                            float cost_scaling_factor, float inflation_radius)
{
  if (!o || !cells || width == 0 || height == 0 || !(resolution > 0.0)) {
    return fail(o, SMPC_ERR_INVALID, "bad costmap");
  }
  Costmap & c = o->costmap;
  c.cells.assign(cells, cells + static_cast<size_t>(width) * height);
  c.W = width;
  c.H = height;
  c.ox = origin_x;
  c.oy = origin_y;
  c.res = resolution;
  c.track_unknown = track_unknown != 0;
  c.inscribed_radius = inscribed_radius;
  c.cost_scaling_factor = cost_scaling_factor;
  c.inflation_radius = inflation_radius;
  c.set = true;
  return SMPC_OK;
}

int smpc_oracle_set_footprint(smpc_oracle * o, const double * xy, uint32_t n_points,
                              double circumscribed_radius, double layer_cost_scaling_factor)
{
  if (!o || (n_points && !xy)) {return SMPC_ERR_INVALID;}
  Costmap & c = o->costmap;
  c.fp_x.clear();
  c.fp_y.clear();
  for (uint32_t i = 0; i < n_points; ++i) {
    c.fp_x.push_back(xy[2 * i]);
    c.fp_y.push_back(xy[2 * i + 1]);
  }
  c.circumscribed_radius = circumscribed_radius;
  c.layer_cost_scaling_factor = layer_cost_scaling_factor;
  return SMPC_OK;
}

int smpc_oracle_set_noise(smpc_oracle * o, const float * nvx, const float * nvy,
                          const float * nwz)
{
  if (!o || !nvx || !nvy || !nwz) {return SMPC_ERR_INVALID;}
  const size_t n = o->B() * o->T();
  o->nvx.assign(nvx, nvx + n);
  o->nvy.assign(nvy, nvy + n);
  o->nwz.assign(nwz, nwz + n);
  o->have_noise = true;
  o->rng_mode = false;
  return SMPC_OK;
}

int smpc_oracle_seed(smpc_oracle * o, uint64_t seed)
{
  if (!o) {return SMPC_ERR_INVALID;}
  o->seed = seed;
  o->epoch = 0;
  o->rng_mode = true;
  draw_noise(o);
  return SMPC_OK;
}

int smpc_oracle_get_noise(smpc_oracle * o, float * nvx, float * nvy, float * nwz)
{
  if (!o) {return SMPC_ERR_INVALID;}
  const size_t n = o->B() * o->T() * sizeof(float);
  if (nvx) {memcpy(nvx, o->nvx.data(), n);}
  if (nvy) {memcpy(nvy, o->nvy.data(), n);}
  if (nwz) {memcpy(nwz, o->nwz.data(), n);}
  return SMPC_OK;
}

int smpc_oracle_set_accumulate_double(smpc_oracle * o, int on)
{
  if (!o) {return SMPC_ERR_INVALID;}
  o->accumulate_double = on != 0;
  return SMPC_OK;
}

int smpc_oracle_set_log_float(smpc_oracle * o, int on)
{
  if (!o) {return SMPC_ERR_INVALID;}
  o->log_float = on != 0;
  return SMPC_OK;
}

// Optimizer::optimize (src/optimizer.cpp:157-164) after prepare (:185-204)
int smpc_oracle_optimize(smpc_oracle * o, const smpc_tick_in * in, float * u_inout,
                         smpc_tick_out * out)
{
  int rc = check_ready(o, in);
  if (rc != SMPC_OK) {return rc;}
  if (!u_inout) {return SMPC_ERR_INVALID;}
  prepare(o, in);
  Tick tk{in, in->path_x, in->path_y, in->path_yaw, in->path_len};
  for (uint32_t i = 0; i < o->cfg.iteration_count; ++i) {
    generate_noised_trajectories(o, in, u_inout);
    eval_trajectories_scores(o, tk);
    update_control_sequence(o, u_inout);
  }
  if (out) {
    memset(out, 0, sizeof(*out));
    out->fail_flag = o->fail_flag ? 1 : 0;
    out->furthest_valid = o->furthest_set ? 1 : 0;
    out->furthest_reached_path_point = static_cast<uint32_t>(o->furthest);
    out->non_colliding = o->non_colliding;
    out->min_cost = o->last_min;
    out->sum_w = o->last_sumw;
    out->passes = o->cfg.iteration_count;
  }
  return SMPC_OK;
}

int smpc_oracle_get_trajectories(smpc_oracle * o, float * x, float * y, float * yaws)
{
  if (!o) {return SMPC_ERR_INVALID;}
  const size_t n = o->B() * o->T() * sizeof(float);
  if (x) {memcpy(x, o->tx.data(), n);}
  if (y) {memcpy(y, o->ty.data(), n);}
  if (yaws) {memcpy(yaws, o->tyaw.data(), n);}
  return SMPC_OK;
}

int smpc_oracle_get_costs(smpc_oracle * o, float * costs)
{
  if (!o || !costs) {return SMPC_ERR_INVALID;}
  memcpy(costs, o->costs.data(), o->B() * sizeof(float));
  return SMPC_OK;
}

// ---- shard phases ---------------------------------------------------------

int smpc_oracle_shard_furthest(smpc_oracle * o, const smpc_tick_in * in, const float * u_in,
                               float * furthest)
{
  int rc = check_ready(o, in);
  if (rc != SMPC_OK) {return rc;}
  if (!u_in || !furthest) {return SMPC_ERR_INVALID;}
  generate_noised_trajectories(o, in, u_in);
  *furthest = static_cast<float>(find_path_furthest_reached_point(
      o->tx.data(), o->ty.data(), o->B(), o->T(), in->path_x, in->path_y, in->path_len));
  return SMPC_OK;
}

// obstacles_only: the re-score after the whole batch was found to collide
// (critic_manager.cpp:70-73: nothing past Obstacles was scored)
static int shard_score_impl(smpc_oracle * o, const smpc_tick_in * in, const float * u_in,
                            uint32_t furthest, bool obstacles_only, float * tuple)
{
  int rc = check_ready(o, in);
  if (rc != SMPC_OK) {return rc;}
  if (!u_in || !tuple) {return SMPC_ERR_INVALID;}
  const size_t B = o->B(), T = o->T();
  prepare(o, in);
  Tick tk{in, in->path_x, in->path_y, in->path_yaw, in->path_len};
  generate_noised_trajectories(o, in, u_in);
  const size_t local_furthest = find_path_furthest_reached_point(
    o->tx.data(), o->ty.data(), B, T, in->path_x, in->path_y, in->path_len);
  o->furthest = furthest;  // batch-wide value agreed by the ranks
  o->furthest_set = true;
  // fail_flag is batch-wide too: a shard whose rollouts all collide must not
  // short-circuit its later critics, so score obstacles and clear the flag.
  const bool fail_in = o->fail_flag;
  if (!fail_in) {
    for (const CriticEntry & e : kCriticOrder) {
      e.fn(o, tk);
      o->fail_flag = false;
      if (obstacles_only && collision_critic_enabled(o, e.fn)) {
        break;   // the critic that stopped the manager: nothing after it was scored
      }
    }
  }
  add_gamma_terms(o, u_in);
  float cmin = std::numeric_limits<float>::max();
  for (size_t b = 0; b < B; ++b) {
    cmin = std::min(cmin, o->costs[b]);
  }
  const float neg_inv_t = -1 / o->cfg.temperature;
  double sum = 0.0;
  std::vector<double> ax(T, 0.0), ay(T, 0.0), az(T, 0.0);
  for (size_t b = 0; b < B; ++b) {
    const double w = expf(neg_inv_t * (o->costs[b] - cmin));
    sum += w;
    for (size_t t = 0; t < T; ++t) {
      ax[t] += o->cvx[b * T + t] * w;
      ay[t] += o->cvy[b * T + t] * w;
      az[t] += o->cwz[b * T + t] * w;
    }
  }
  tuple[0] = cmin;
  tuple[1] = static_cast<float>(sum);
  tuple[2] = static_cast<float>(local_furthest);
  tuple[3] = static_cast<float>(fail_in ? 0u : o->non_colliding);
  for (size_t t = 0; t < T; ++t) {
    tuple[SMPC_TUPLE_HEADER + t] = static_cast<float>(ax[t]);
    tuple[SMPC_TUPLE_HEADER + T + t] = static_cast<float>(ay[t]);
    tuple[SMPC_TUPLE_HEADER + 2 * T + t] = static_cast<float>(az[t]);
  }
  return SMPC_OK;
}

int smpc_oracle_shard_score(smpc_oracle * o, const smpc_tick_in * in, const float * u_in,
                            uint32_t furthest, float * tuple)
{
  return shard_score_impl(o, in, u_in, furthest, false, tuple);
}

int smpc_oracle_shard_rescore_failed(smpc_oracle * o, const smpc_tick_in * in,
                                     const float * u_in, float * tuple)
{
  return shard_score_impl(o, in, u_in, 0, true, tuple);
}

int smpc_oracle_shard_combine(smpc_oracle * o, const float * tuples, uint32_t n_tuples,
                              float * u_out, smpc_tick_out * out)
{
  if (!o || !tuples || n_tuples == 0 || !u_out) {return SMPC_ERR_INVALID;}
  const size_t T = o->T();
  const size_t L = SMPC_TUPLE_HEADER + 3 * T;
  float gmin = std::numeric_limits<float>::max();
  float furthest = 0.0f;
  double non_colliding = 0.0;
  for (uint32_t g = 0; g < n_tuples; ++g) {
    gmin = std::min(gmin, tuples[g * L + 0]);
    furthest = std::max(furthest, tuples[g * L + 2]);
    non_colliding += tuples[g * L + 3];
  }
  const float neg_inv_t = -1 / o->cfg.temperature;
  double sum = 0.0;
  std::vector<double> acc(3 * T, 0.0);
  for (uint32_t g = 0; g < n_tuples; ++g) {
    const double scale = expf(neg_inv_t * (tuples[g * L + 0] - gmin));
    sum += scale * tuples[g * L + 1];
    for (size_t i = 0; i < 3 * T; ++i) {
      acc[i] += scale * tuples[g * L + SMPC_TUPLE_HEADER + i];
    }
  }
  for (size_t i = 0; i < 3 * T; ++i) {
    u_out[i] = static_cast<float>(acc[i] / sum);
  }
  apply_constraints(u_out, static_cast<uint32_t>(T), o->c_vx_max, o->c_vx_min, o->c_vy, o->c_wz,
                    o->holonomic(), o->ackermann_r());
  if (out) {
    memset(out, 0, sizeof(*out));
    out->fail_flag = (non_colliding == 0.0 && o->critics.obstacles.enabled) ? 1 : 0;
    out->furthest_valid = 1;
    out->furthest_reached_path_point = static_cast<uint32_t>(furthest);
    out->non_colliding = static_cast<uint32_t>(non_colliding);
    out->min_cost = gmin;
    out->sum_w = static_cast<float>(sum);
    out->passes = 1;
  }
  return SMPC_OK;
}

// ---- piece-wise entry points ------------------------------------------------

int smpc_oracle_set_state_velocities(smpc_oracle * o, const float * vx, const float * vy,
                                     const float * wz)
{
  if (!o) {return SMPC_ERR_INVALID;}
  const size_t n = o->B() * o->T();
  if (vx) {o->vx.assign(vx, vx + n);}
  if (vy) {o->vy.assign(vy, vy + n);}
  if (wz) {o->wz.assign(wz, wz + n);}
  return SMPC_OK;
}

int smpc_oracle_set_trajectories(smpc_oracle * o, const float * x, const float * y,
                                 const float * yaws)
{
  if (!o) {return SMPC_ERR_INVALID;}
  const size_t n = o->B() * o->T();
  if (x) {o->tx.assign(x, x + n);}
  if (y) {o->ty.assign(y, y + n);}
  if (yaws) {o->tyaw.assign(yaws, yaws + n);}
  return SMPC_OK;
}

int smpc_oracle_update_state_velocities(smpc_oracle * o, const smpc_tick_in * in,
                                        const float * cvx, const float * cvy,
                                        const float * cwz)
{
  if (!o || !in || !cvx || !cvy || !cwz) {return SMPC_ERR_INVALID;}
  const size_t n = o->B() * o->T();
  o->cvx.assign(cvx, cvx + n);
  o->cvy.assign(cvy, cvy + n);
  o->cwz.assign(cwz, cwz + n);
  update_state_velocities(o, in);
  return SMPC_OK;
}

int smpc_oracle_get_state_velocities(smpc_oracle * o, float * vx, float * vy, float * wz)
{
  if (!o) {return SMPC_ERR_INVALID;}
  const size_t n = o->B() * o->T() * sizeof(float);
  if (vx) {memcpy(vx, o->vx.data(), n);}
  if (vy) {memcpy(vy, o->vy.data(), n);}
  if (wz) {memcpy(wz, o->wz.data(), n);}
  return SMPC_OK;
}

int smpc_oracle_integrate(smpc_oracle * o, const smpc_tick_in * in)
{
  if (!o || !in) {return SMPC_ERR_INVALID;}
  integrate_state_velocities(o, in);
  return SMPC_OK;
}

int smpc_oracle_score_critic(smpc_oracle * o, int critic_id, const smpc_tick_in * in,
                             int64_t furthest_preset, float * costs_inout,
                             int32_t * fail_flag_out)
{
  if (!o || !in || !costs_inout) {return SMPC_ERR_INVALID;}
  if ((critic_id == SMPC_ORACLE_CRITIC_OBSTACLES || critic_id == SMPC_ORACLE_CRITIC_COST) &&
    !o->costmap.set)
  {
    return fail(o, SMPC_ERR_STATE, "no costmap");
  }
  if (furthest_preset >= 0 && in->path_len > 0 &&
    static_cast<uint64_t>(furthest_preset) > in->path_len - 1)
  {
    // the reference would index the path out of range here (critics_tests.cpp:516-520 does)
    return fail(o, SMPC_ERR_INVALID, "furthest_preset beyond the path");
  }
  const size_t B = o->B();
  o->costs.assign(costs_inout, costs_inout + B);
  o->fail_flag = false;
  o->furthest_set = furthest_preset >= 0;
  o->furthest = furthest_preset >= 0 ? static_cast<size_t>(furthest_preset) : 0;
  o->path_valid_set = false;
  Tick tk{in, in->path_x, in->path_y, in->path_yaw, in->path_len};
  switch (critic_id) {
    case SMPC_ORACLE_CRITIC_OBSTACLES: score_obstacles(o, tk); break;
    case SMPC_ORACLE_CRITIC_PATH_ALIGN: score_path_align(o, tk); break;
    case SMPC_ORACLE_CRITIC_PATH_FOLLOW: score_path_follow(o, tk); break;
    case SMPC_ORACLE_CRITIC_GOAL_ANGLE: score_goal_angle(o, tk); break;
    case SMPC_ORACLE_CRITIC_PREFER_FORWARD: score_prefer_forward(o, tk); break;
    case SMPC_ORACLE_CRITIC_COST: score_cost(o, tk); break;
    case SMPC_ORACLE_CRITIC_GOAL: score_goal(o, tk); break;
    case SMPC_ORACLE_CRITIC_CONSTRAINT: score_constraint(o, tk); break;
    case SMPC_ORACLE_CRITIC_TWIRLING: score_twirling(o, tk); break;
    case SMPC_ORACLE_CRITIC_PATH_ANGLE: score_path_angle(o, tk); break;
    case SMPC_ORACLE_CRITIC_VELOCITY_DEADBAND: score_velocity_deadband(o, tk); break;
    case SMPC_ORACLE_CRITIC_PATH_ALIGN_LEGACY: score_path_align_legacy(o, tk); break;
    default: return fail(o, SMPC_ERR_INVALID, "unknown critic id");
  }
  memcpy(costs_inout, o->costs.data(), B * sizeof(float));
  if (fail_flag_out) {*fail_flag_out = o->fail_flag ? 1 : 0;}
  return SMPC_OK;
}

int smpc_oracle_within_position_goal_tolerance(float tol, double px, double py, double gx,
                                               double gy)
{
  return within_position_goal_tolerance(tol, px, py, gx, gy) ? 1 : 0;
}

void smpc_oracle_normalize_angles(const float * in, double * out, uint32_t n)
{
  for (uint32_t i = 0; i < n; ++i) {
    out[i] = normalize_angle(static_cast<double>(in[i]));
  }
}

void smpc_oracle_shortest_angular_distance(const float * from, float to, double * out,
                                           uint32_t n)
{
  for (uint32_t i = 0; i < n; ++i) {
    out[i] = normalize_angle(static_cast<double>(to - from[i]));
  }
}

uint32_t smpc_oracle_find_path_furthest_reached_point(const float * traj_x,
                                                      const float * traj_y, uint32_t B,
                                                      uint32_t T, const float * path_x,
                                                      const float * path_y, uint32_t P)
{
  return static_cast<uint32_t>(
    find_path_furthest_reached_point(traj_x, traj_y, B, T, path_x, path_y, P));
}

uint32_t smpc_oracle_find_path_trajectory_initial_point(float x00, float y00,
                                                        const float * path_x,
                                                        const float * path_y, uint32_t P)
{
  return static_cast<uint32_t>(find_path_trajectory_initial_point(x00, y00, path_x, path_y, P));
}

int smpc_oracle_find_path_costs(smpc_oracle * o, const float * path_x, const float * path_y,
                                uint32_t P, uint8_t * valid_out)
{
  if (!o || !o->costmap.set || !valid_out) {return SMPC_ERR_INVALID;}
  std::vector<uint8_t> v;
  find_path_costs(o->costmap, path_x, path_y, P, v);
  if (!v.empty()) {memcpy(valid_out, v.data(), v.size());}
  return SMPC_OK;
}

uint32_t smpc_oracle_find_closest_path_pt(const float * vec, uint32_t n, float dist,
                                          uint32_t init)
{
  return static_cast<uint32_t>(find_closest_path_pt(vec, n, dist, init));
}

void smpc_oracle_apply_constraints(float * u, uint32_t T, float vx_max, float vx_min,
                                   float vy_max, float wz_max)
{
  apply_constraints(u, T, vx_max, vx_min, vy_max, wz_max);
}

void smpc_oracle_motion_model_apply_constraints(float * u, uint32_t T, uint32_t motion_model,
                                                float ackermann_min_turning_r)
{
  motion_model_apply_constraints(
    u, T, motion_model == SMPC_MODEL_ACKERMANN ? ackermann_min_turning_r : -1.0f);
}

// Optimizer::shiftControlSequence (src/optimizer.cpp:206-225), holonomic
void smpc_oracle_shift_control_sequence(float * u, uint32_t T)
{
  if (T < 2) {return;}
  for (int c = 0; c < 3; ++c) {
    float * q = u + static_cast<size_t>(c) * T;
    const float first = q[0];
    for (uint32_t t = 0; t + 1 < T; ++t) {
      q[t] = q[t + 1];
    }
    q[T - 1] = first;      // xt::roll(-1)
    q[T - 1] = q[T - 2];   // view(-1) = view(-2)
  }
}

void smpc_oracle_savitsky_golay(float * u, uint32_t T, float * history, int shift)
{
  savitsky_golay(u, T, history, shift != 0);
}

// Optimizer::setSpeedLimit (src/optimizer.cpp:428-453); NO_SPEED_LIMIT = 0.0
void smpc_oracle_speed_limit(const float * base, double speed_limit, int percentage,
                             float * c)
{
  if (speed_limit == 0.0) {
    c[0] = base[0]; c[1] = base[1]; c[2] = base[2]; c[3] = base[3];
  } else {
    const double ratio = percentage ? speed_limit / 100.0 : speed_limit / base[0];
    c[0] = static_cast<float>(base[0] * ratio);
    c[1] = static_cast<float>(base[1] * ratio);
    c[2] = static_cast<float>(base[2] * ratio);
    c[3] = static_cast<float>(base[3] * ratio);
  }
}

// Optimizer::setOffset (src/optimizer.cpp:95-114)
int smpc_oracle_set_offset(double controller_frequency, float model_dt)
{
  const double controller_period = 1.0 / controller_frequency;
  constexpr double eps = 1e-6;
  if ((controller_period + eps) < model_dt) {
    return 0;
  } else if (std::abs(controller_period - model_dt) < eps) {
    return 1;
  }
  return -1;
}

void smpc_oracle_philox4x32_10(const uint32_t * ctr, const uint32_t * key, uint32_t * out)
{
  philox4x32_10(ctr, key, out);
}

// Defaults: the oracle carries its own copy so it does not link the product.
void smpc_config_default(smpc_config * c)
{
  memset(c, 0, sizeof(*c));
  c->batch_size = 1000;   // src/optimizer.cpp:69-82
  c->time_steps = 56;
  c->iteration_count = 1;
  c->motion_model = SMPC_MODEL_OMNI;
  c->model_dt = 0.05f;
  c->temperature = 0.3f;
  c->gamma = 0.015f;
  c->vx_max = 0.5f;
  c->vx_min = -0.35f;
  c->vy_max = 0.5f;
  c->wz_max = 1.9f;
  c->vx_std = 0.2f;
  c->vy_std = 0.2f;
  c->wz_std = 0.4f;
  c->device = -1;
  c->ackermann_min_turning_r = 0.2f;  // motion_models.hpp:94
}

void smpc_critic_params_default(smpc_critic_params * p)
{
  memset(p, 0, sizeof(*p));
  p->obstacles = {1, 0, 1, 1.5f, 20.0f, 10000.0f, 0.10f, 0.5f};       // obstacles_critic.cpp:21-31
  p->path_align = {1, 0, 1, 10.0f, 0.07f, 20, 4, 0.5f};               // path_align_critic.cpp:26-38
  p->path_align_legacy = {0, 0, 1, 10.0f, 0.07f, 20, 4, 0.5f};        // path_align_legacy_critic.cpp:26-37
  p->path_follow = {1, 1, 5.0f, 1.4f, 6};                             // path_follow_critic.cpp:23-33
  p->goal_angle = {1, 1, 3.0f, 0.5f};                                 // goal_angle_critic.cpp:20-27
  p->prefer_forward = {1, 1, 5.0f, 0.5f};                             // prefer_forward_critic.cpp:20-27
  // not in the north star's critic list: off unless asked for
  p->cost = {0, 0, 1, 3.81f, 300.0f, 1000000.0f, 0.5f};               // cost_critic.cpp:25-31
  p->goal = {0, 1, 5.0f, 1.4f};                                       // goal_critic.cpp:26-28
  p->constraint = {0, 1, 4.0f, 0.5f, 0.5f, -0.35f};                   // constraint_critic.cpp:27-35
  p->twirling = {0, 1, 10.0f};                                        // twirling_critic.cpp:24-25
  p->path_angle = {0, 1, 2.0f, 4, 0.5f, 1.2f, 1, -0.35f};             // path_angle_critic.cpp:24-45
  p->velocity_deadband = {0, 1, 35.0f, {0.0f, 0.0f, 0.0f}};           // velocity_deadband_critic.cpp:24-33
}

}  // extern "C"
