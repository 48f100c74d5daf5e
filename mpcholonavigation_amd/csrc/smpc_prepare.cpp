// smpc_prepare.cpp — the per-tick host work behind the C-ABI (include/smpc.h): what the
// reference's critics do once per tick (goal-distance gates, path validity, cumulative path
// lengths, the per-candidate PathAlign / PathFollow tables), the 256-entry lookup tables of
// the collision critics, the tick block the kernels read, and the LDS carve-up of the passes.
#include "smpc_ctx.h"

namespace smpc_impl {

TickLayout tick_layout(uint32_t T, uint32_t P)
{
  TickLayout l{};
  size_t o = 0;
  // the tick's number sits right in front of u: a pass finds it at u[-4] with no pointer of its
  // own (the lane pass has no scalar register to spare for one)
  l.canary = o; o += 16;
  l.u = o; o += align_up(3 * T * 4, 16);
  l.px = o; o += align_up(P * 4, 16);
  l.py = o; o += align_up(P * 4, 16);
  l.pyaw = o; o += align_up(P * 4, 16);
  l.D = o; o += align_up(P * 4, 16);
  l.pf_idx = o; o += align_up(P * 4, 16);
  l.pvalid = o; o += align_up(P, 16);
  l.pa_active = o; o += align_up(P, 16);
  l.pang_active = o; o += align_up(P, 16);
  l.pal_active = o; o += align_up(P, 16);   // PathAlignLegacy gate per candidate furthest point
  l.lut_cost = o; o += 256 * 4;   // CostCritic repulsive term per 8-bit cost
  l.total = o;
  return l;
}

SmpcLds make_lds(uint32_t window_bytes, uint32_t P, uint32_t T, uint32_t nwave, bool with_map,
                 uint32_t nsamp)
{
  SmpcLds L{};
  uint32_t o = with_map ? align_up(window_bytes, 16) : 0;
  // 256 entries + one all-zero entry (index 256: the lane pass primes its lookup pipeline with it)
  L.off_lut = o; o += with_map ? (256 + 2) * sizeof(SmpcLut) : 0;
  const uint32_t pf = align_up(std::max(P, 1u) * 4, 16);
  L.off_px = o; o += pf;
  L.off_py = o; o += pf;
  L.off_pyaw = o; o += pf;
  L.off_D = o; o += align_up((std::max(P, 1u) + 2) * 4, 16);   // + a sentinel on either side (lane pass)
  L.off_valid = o; o += align_up(std::max(P, 1u), 16);
  L.off_scr = o;
  // lanes per parked rollout: sample slots 0..nsamp fit one segment of 16/32/64 lanes;
  // rollouts per flush: as many segments as a wave has, capped so that the parked
  // controls stay <= 3 KiB per wave
  const uint32_t R = T <= 64 ? 1 : (T <= 128 ? 2 : 4);
  L.seg_shift = nsamp + 1 <= 16 ? 4 : (nsamp + 1 <= 32 ? 5 : 6);
  L.group = std::max(1u, std::min(64u >> L.seg_shift, 4u / R));
  // per wave: [sample points 3x64][endpoint ring 2x64][parked controls group x 3T]; the head is
  // re-used for the block combine [4+3T]
  L.scr_pts = 0;
  L.scr_ring = 3 * 64;
  L.scr_c = L.scr_ring + 2 * 64;
  L.scr_stride = align_up(std::max(L.scr_c + L.group * 3 * T, 4 + 3 * T), 4);
  o += nwave * L.scr_stride * 4;
  L.total = o;
  return L;
}

// LDS layout of the lane-per-rollout pass: window + the NO_INFORMATION byte + one scratch byte
// per lane (cell_byte_exact), LUT, path, per wave the parked wz [64][68] + weights [64]; the
// re-read form parks nothing (per wave only the block combine's tuple)
SmpcLds lane_lds(uint32_t window_bytes, uint32_t P, uint32_t T, bool rr)
{
  const uint32_t lblock = rr ? smpc_lane_block_rr() : smpc_lane_block();
  SmpcLds Lt = make_lds(window_bytes ? window_bytes + 1 + lblock : 0, P, T, lblock / 64,
                        window_bytes != 0, 0);
  Lt.off_pts4 = Lt.off_scr;
  Lt.off_scr += align_up(std::max(P, 1u) * 16, 16) + 16;   // + sum u^2 per control (the gamma terms), 16 bytes
  Lt.scr_stride = rr ? align_up(4u + 3u * T, 4) : align_up(std::max(64u * 68u + 64u, 4u + 3u * T), 4);
  Lt.total = Lt.off_scr + (lblock / 64) * Lt.scr_stride * 4;
  return Lt;
}

// LDS layout of smpc_pass_split: as the lane pass up to the path points, then sum u^2 [4] and the
// control sequence [3][64] in front of the per-wave scratch (parked wz [16][64]; the block
// combine's tuple re-uses it)
SmpcLds split_lds(uint32_t window_bytes, uint32_t P, uint32_t T, uint32_t nseg)
{
  const uint32_t lblock = smpc_split_block();
  SmpcLds Lt = make_lds(window_bytes ? window_bytes + 1 + lblock : 0, P, T, lblock / 64, window_bytes != 0, 0);
  Lt.off_pts4 = Lt.off_scr;
  Lt.off_scr += align_up(std::max(P, 1u) * 16, 16) + 16 + 3 * 64 * 4;
  // parked wz [steps per lane][64]; two segments: vy too (smpc_split.hip PARK_VY)
  Lt.scr_stride = align_up(std::max((nseg == 2 ? 2u : 1u) * (64u / nseg) * 64u, 4u + 3u * T), 4);
  Lt.total = Lt.off_scr + (lblock / 64) * Lt.scr_stride * 4;
  return Lt;
}

// distanceToObstacle (obstacles_critic.cpp:99-112) for an 8-bit cost, point mode
static float distance_to_obstacle(const HostCostmap& m, float cost, bool using_footprint = false)
{
  const float scale_factor = m.cost_scaling_factor;
  const float min_radius = m.inscribed_radius;
  float d = static_cast<float>(
    (static_cast<double>(scale_factor * min_radius) - std::log(static_cast<double>(cost)) +
    std::log(static_cast<double>(253.0f))) / static_cast<double>(scale_factor));
  if (!using_footprint) d -= min_radius;   // obstacles_critic.cpp:106-108
  return d;
}

// consider_fp: the critic's consider_footprint collision rule; using_fp: the cost came from the
// footprint (no inscribed-radius offset, obstacles_critic.cpp:106-108)
// cost_terms (non-null: CostCritic scored WITHOUT ObstaclesCritic): the table's second field is the
// CostCritic's per-cost term (cost_critic.cpp:141-155) instead of ObstaclesCritic's repulsion — the
// first field keeps the shared collision marker and is otherwise zero — so that the lane pass's
// lookup pipeline scores CostCritic as it stands; the wave pass reads only the marker then
static void build_lut(const smpc_ctx* c, bool near_goal, SmpcLut* lut, bool consider_fp = false,
               bool using_fp = false, const float* cost_terms = nullptr)
{
  const auto& m = c->map;
  const auto& p = c->critics.obstacles;
  for (int v = 0; v < 256; ++v) {
    lut[v].crit = 0.f;
    lut[v].rep = 0.f;
    if (cost_terms) {
      if (v == SMPC_COST_LETHAL || (v == SMPC_COST_INSCRIBED && !consider_fp) ||
        (v == SMPC_COST_NO_INFORMATION && !m.track_unknown))
      {
        lut[v].crit = -1.0f;
      } else {
        lut[v].rep = cost_terms[v];
      }
      continue;
    }
    // inCollision (obstacles_critic.cpp:185-201), consider_footprint = false
    if (v == SMPC_COST_LETHAL || (v == SMPC_COST_INSCRIBED && !consider_fp) ||
      (v == SMPC_COST_NO_INFORMATION && !m.track_unknown))
    {
      lut[v].crit = -1.0f;                                   // :152 collision marker
      continue;
    }
    if (v < 1) continue;                                     // :150 free space
    if (m.inflation_radius == 0.0f || m.cost_scaling_factor == 0.0f) continue;  // :155
    const float d = distance_to_obstacle(m, static_cast<float>(v), using_fp);
    if (d < p.collision_margin_distance) {
      lut[v].crit = p.collision_margin_distance - d;         // :165
    } else if (!near_goal) {
      lut[v].rep = m.inflation_radius - d;                   // :167
    }
  }
}

void remember_furthest(smpc_ctx* c, const smpc_tick_in* in, float F)
{
  // what the prediction for THIS tick missed is how far the rollouts' endpoints moved relative
  // to the robot since the last tick (the control sequence is still changing: an accelerating
  // robot reaches further every tick); to first order it does so again
  static const bool trace = getenv("SMPC_TRACE_FURTHEST") != nullptr;
  if (trace)
    fprintf(stderr, "[smpc furthest] true %.3f predicted %.3f (geometry %.3f, valid %d, drift %.3f)\n", F,
            c->hint_Fp + c->hint_drift, c->hint_Fp, (int)c->hint_Fp_valid, c->hint_drift);
  // A tick scored with NEW noise (regenerate_noises = true) moves the extreme rollout by itself:
  // at 2 097 152 rollouts the true value jitters by +-0.5 of an index from epoch to epoch.  What
  // the last prediction missed is then mostly that jitter, not a trend: carrying it forward as
  // drift doubles it (observed: 1.55 scoring passes per tick).  Such ticks update a smoothed
  // estimate instead — the next prediction starts from F' + 0.4 (F - F'), F' the geometric
  // prediction of this tick — and leave the drift alone.
  const bool fresh_noise = c->noise_gen != c->noise_gen_remembered;
  c->noise_gen_remembered = c->noise_gen;
  float F_next = F;
  if (c->hint_Fp_valid) {
    const float d = F - c->hint_Fp;
    if (fresh_noise && std::fabs(d) < 2.f) {
      F_next = c->hint_Fp + c->hint_drift + 0.4f * (F - (c->hint_Fp + c->hint_drift));
      c->hint_drift *= 0.5f;
    } else {
      c->hint_drift = std::fabs(d) < 2.f ? d : 0.f;
    }
  } else {
    c->hint_drift = 0.f;
  }
  c->hint_Fp_valid = false;
  c->hint_F = F_next;
  c->hint = smpc_furthest_index(F_next);
  c->hint_valid = true;
  c->anchor_x = in->pose_x;
  c->anchor_y = in->pose_y;
  c->anchor_ex = c->cur_ex;
  c->anchor_ey = c->cur_ey;
  c->anchor_e_valid = c->cur_e_valid;
  c->cur_e_valid = false;
  c->anchor_px.assign(in->path_x, in->path_x + in->path_len);
  c->anchor_py.assign(in->path_y, in->path_y + in->path_len);
  c->anchor_valid = true;
}

// The index a tick is first scored with.  Between two ticks the robot advances and the plan
// handed over by the controller is pruned to the robot (path_handler.cpp:48-143), so the
// furthest reached path point of the last tick is stale by construction: at 0.3 m/s, 20 Hz
// and 0.05 m between path points the index moves once every three ticks from the motion and
// once every three ticks, the other way, from the pruning.  Both are known on the host: the
// displacement of the pose along the plan at the furthest point, in segment lengths, and the
// number of points the plan lost at its start (the old point the new first point coincides
// with).  F' = F + displacement - shift, index = round(F').  Exactness never depends on this:
// the scoring pass reports the true value and a miss is re-scored.
// Where the rollout WITHOUT noise ends: pose + sum of the rotated state velocities (v[0] the measured
// speed, v[t] = u[t - 1]; optimizer.cpp:258-267, 313-343), in double with the rotation advanced by
// its Taylor terms — this feeds the prediction only, 0.2 us per tick.
static void nominal_endpoint(const smpc_ctx* c, const smpc_tick_in* in, const float* u, double& ex, double& ey)
{
  const uint32_t T = c->cfg.time_steps;
  const double dt = c->cfg.model_dt;
  double co = std::cos(static_cast<double>(in->pose_yaw)), si = std::sin(static_cast<double>(in->pose_yaw));
  double x = 0.0, y = 0.0;
  double vx = in->speed_vx, vy = c->holonomic ? in->speed_vy : 0.0, wz = in->speed_wz;
  for (uint32_t t = 0; t < T; ++t) {
    x += (vx * co - vy * si) * dt;      // cos_[t] = cos(yaw[t - 1])
    y += (vx * si + vy * co) * dt;
    const double a = wz * dt, a2 = a * a;
    const double ca = 1.0 - 0.5 * a2 + a2 * a2 * (1.0 / 24.0), sa = a * (1.0 - a2 * (1.0 / 6.0));
    const double cn = co * ca - si * sa;
    si = si * ca + co * sa;
    co = cn;
    vx = u[t];
    vy = c->holonomic ? u[T + t] : 0.0;
    wz = u[2 * T + t];
  }
  ex = in->pose_x + x;
  ey = in->pose_y + y;
}

void predict_hint(smpc_ctx* c, const smpc_tick_in* in, const float* u_in)
{
  // The rollouts are the nominal one plus the SAME noise tick after tick: their furthest point
  // moves with the nominal endpoint — the robot's motion and the change of the control sequence
  // (a sequence still accelerating reaches further every tick, and stops doing so where it meets
  // the constraints: the kink a tick-to-tick drift estimate overshoots).
  c->cur_e_valid = false;
  if (u_in) {
    nominal_endpoint(c, in, u_in, c->cur_ex, c->cur_ey);
    c->cur_e_valid = true;
  }
  if (!c->hint_valid || !c->anchor_valid) return;
  const uint32_t P0 = static_cast<uint32_t>(c->anchor_px.size()), P = in->path_len;
  c->hint = smpc_furthest_index(c->hint_F);
  if (c->hint_is_this_ticks) {
    // smpc_group_optimize re-runs a member whose prediction missed: hint_F is this very tick's
    // true value.  Score with it, and leave the drift as remember_furthest measured it.
    c->hint_is_this_ticks = false;
    c->hint_Fp = c->hint_F - c->hint_drift;
    c->hint_Fp_valid = true;
    return;
  }
  c->hint_Fp_valid = false;
  if (P0 < 2 || P < 1) return;
  const float* ox = c->anchor_px.data();
  const float* oy = c->anchor_py.data();
  auto d2 = [](float ax, float ay, float bx, float by) {return (ax - bx) * (ax - bx) + (ay - by) * (ay - by);};
  // the old point the new plan starts at
  uint32_t k = 0;
  float best = std::numeric_limits<float>::max();
  for (uint32_t j = 0; j < P0; ++j) {
    const float d = d2(ox[j], oy[j], in->path_x[0], in->path_y[0]);
    if (d < best) {
      best = d;
      k = j;
    }
  }
  const uint32_t kn = k + 1 < P0 ? k + 1 : k - 1;
  const float seg_k = d2(ox[k], oy[k], ox[kn], oy[kn]);
  if (!(best <= 0.0625f * seg_k)) return;   // not a pruned copy of the old plan: nothing to carry over
  uint32_t S0 = c->hint;
  if (S0 >= P0) S0 = P0 - 1;
  const uint32_t Sn = S0 + 1 < P0 ? S0 + 1 : S0 - 1;
  const float sx = (S0 + 1 < P0 ? 1.f : -1.f) * (ox[Sn] - ox[S0]), sy = (S0 + 1 < P0 ? 1.f : -1.f) * (oy[Sn] - oy[S0]);
  const float seg2 = sx * sx + sy * sy;
  if (!(seg2 > 0.f)) return;
  // displacement along the plan at the furthest point, in segment lengths: of the nominal endpoint
  // when both ticks have one, else of the pose alone
  const bool by_endpoint = c->cur_e_valid && c->anchor_e_valid;
  const double dx = by_endpoint ? c->cur_ex - c->anchor_ex : in->pose_x - c->anchor_x;
  const double dy = by_endpoint ? c->cur_ey - c->anchor_ey : in->pose_y - c->anchor_y;
  const float disp = static_cast<float>(dx * sx + dy * sy) / seg2;
  if (!(std::fabs(disp) < (by_endpoint ? 8.f : 4.f))) return;     // a jump, not a controller period's motion
  const float Fp0 = c->hint_F + disp - static_cast<float>(k);   // carried by the geometry alone
  const float Fp = Fp0 + c->hint_drift;                          // ... and by last tick's drift
  c->hint_Fp = Fp0;
  c->hint_Fp_valid = true;
  long h = std::lround(Fp);
  if (h < 0) h = 0;
  if (h > static_cast<long>(P) - 1) h = static_cast<long>(P) - 1;
  c->hint = static_cast<uint32_t>(h);
}

int check_tick(smpc_ctx* c, const smpc_tick_in* in)
{
  if (!c || !in) return SMPC_ERR_INVALID;
  if (!c->have_noise) return fail(c, SMPC_ERR_STATE, "no noise: call smpc_set_noise or smpc_seed");
  if (in->path_len > 0 && (!in->path_x || !in->path_y || !in->path_yaw))
    return fail(c, SMPC_ERR_INVALID, "path arrays missing");
  if (in->path_len > SMPC_MAX_PATH)
    return fail(c, SMPC_ERR_UNSUPPORTED, "path longer than SMPC_MAX_PATH (1024) points");
  if (!c->map.set && (c->critics.obstacles.enabled || c->critics.cost.enabled || !in->path_pts_valid))
    return fail(c, SMPC_ERR_STATE, "no costmap: call smpc_set_costmap");
  if (((c->critics.obstacles.enabled && c->critics.obstacles.consider_footprint) ||
    (c->critics.cost.enabled && c->critics.cost.consider_footprint)) && c->fp_x.empty())
    return fail(c, SMPC_ERR_STATE, "consider_footprint=true needs a footprint: call smpc_set_footprint");
  return SMPC_OK;
}

// Everything the reference's critics decide once per tick on the host, plus the
// upload of the tick block.  Leaves c->dev ready for the launches.
// Which critics score this tick: membership in the critics list, and for the gated ones the
// reference's withinPositionGoalTolerance test at the top of score().  Also the
// number of PathAlign samples per trajectory.
static int tick_gates(smpc_ctx* c, const smpc_tick_in* in, uint32_t& gates_out, uint32_t& nsamp_out)
{
  const uint32_t T = c->cfg.time_steps, P = in->path_len;
  const auto& cr = c->critics;
  const double rx = in->pose_x, ry = in->pose_y, gx = in->goal_x, gy = in->goal_y;
  uint32_t gates = 0;
  if (cr.obstacles.enabled) gates |= SD_OBSTACLES;
  if (cr.path_align.enabled && !within_tol(cr.path_align.threshold_to_consider, rx, ry, gx, gy))
    gates |= SD_PATH_ALIGN;                                   // path_align_critic.cpp:49-54
  if (cr.path_follow.enabled && P >= 2 &&
    !within_tol(cr.path_follow.threshold_to_consider, rx, ry, gx, gy))
    gates |= SD_PATH_FOLLOW;                                  // path_follow_critic.cpp:37-42
  if (cr.goal_angle.enabled && P >= 1 &&
    within_tol(cr.goal_angle.threshold_to_consider, rx, ry, gx, gy))
    gates |= SD_GOAL_ANGLE;                                   // goal_angle_critic.cpp:38-43
  if (cr.prefer_forward.enabled &&
    !within_tol(cr.prefer_forward.threshold_to_consider, rx, ry, gx, gy))
    gates |= SD_PREFER_FORWARD;                               // prefer_forward_critic.cpp:36-41
  // the other registered critics (general pass only)
  if (cr.constraint.enabled) gates |= SD_CONSTRAINT;
  if (cr.cost.enabled) gates |= SD_COST;
  if (cr.obstacles.enabled && cr.obstacles.consider_footprint) gates |= SD_FP_OBSTACLES;
  if (cr.cost.enabled && cr.cost.consider_footprint) gates |= SD_FP_COST;
  if (cr.goal.enabled && within_tol(cr.goal.threshold_to_consider, rx, ry, gx, gy))
    gates |= SD_GOAL;                                         // goal_critic.cpp:38-42
  if (cr.twirling.enabled) {
    // utils::withinPositionGoalTolerance(goal_checker, ...) (tools/utils.hpp:201-224)
    bool within = false;
    if (in->goal_checker_xy_tolerance >= 0.0f) {
      const double tol = static_cast<double>(in->goal_checker_xy_tolerance);
      const double dx = rx - gx, dy = ry - gy;
      within = dx * dx + dy * dy < tol * tol;
    }
    if (!within) gates |= SD_TWIRLING;                        // twirling_critic.cpp:33-37
  }
  if (cr.path_angle.enabled && P >= 1 &&
    !within_tol(cr.path_angle.threshold_to_consider, rx, ry, gx, gy))
    gates |= SD_PATH_ANGLE;                                   // path_angle_critic.cpp:60-69
  if (cr.velocity_deadband.enabled) gates |= SD_DEADBAND;
  // PathAlignLegacyCritic (path_align_legacy_critic.cpp:48-54, :86-88: at least one path segment)
  if (cr.path_align_legacy.enabled && P >= 2 &&
    !within_tol(cr.path_align_legacy.threshold_to_consider, rx, ry, gx, gy))
    gates |= SD_PATH_ALIGN_LEGACY;
  if (P == 0) gates &= ~(SD_PATH_ALIGN | SD_PATH_FOLLOW);
  uint32_t nsamp = 0;
  const uint32_t step = cr.path_align.trajectory_point_step;
  if (gates & SD_PATH_ALIGN) {
    nsamp = step > 0 ? (T - 1) / step : 0;
    if (nsamp > 63)
      return fail(c, SMPC_ERR_UNSUPPORTED,
                  "PathAlign: more than 63 samples per trajectory (time_steps / trajectory_point_step)");
    if (nsamp == 0) gates &= ~SD_PATH_ALIGN;  // no samples: cost 0 for every rollout
    if (cr.path_align.use_path_orientations) gates |= SD_USE_PATH_YAW;
  }
  if (gates & SD_PATH_ALIGN_LEGACY) {
    // its trajectory points are PathAlign's: p = step, 2 step, ... < T (path_align_legacy_critic.cpp:97)
    const uint32_t lstep = cr.path_align_legacy.trajectory_point_step;
    if ((gates & SD_PATH_ALIGN) && lstep != step)
      return fail(c, SMPC_ERR_UNSUPPORTED,
                  "PathAlignCritic and PathAlignLegacyCritic in one list need the same trajectory_point_step");
    const uint32_t ns = lstep > 0 ? (T - 1) / lstep : 0;
    if (ns > 63)
      return fail(c, SMPC_ERR_UNSUPPORTED,
                  "PathAlignLegacy: more than 63 samples per trajectory (time_steps / trajectory_point_step)");
    if (ns == 0) gates &= ~SD_PATH_ALIGN_LEGACY;   // no samples: summed_dist 0, cost 0 for every rollout
    else nsamp = ns;
    if (cr.path_align_legacy.use_path_orientations && (gates & SD_PATH_ALIGN_LEGACY)) gates |= SD_PAL_USE_PATH_YAW;
  }
  if (gates & (SD_PATH_ALIGN | SD_PATH_FOLLOW | SD_PATH_ANGLE | SD_PATH_ALIGN_LEGACY)) gates |= SD_NEED_FURTHEST;
  if (c->map.track_unknown) gates |= SD_TRACK_UNKNOWN;
  if (c->cfg.flags & SMPC_FLAG_STORE_TRAJECTORIES) gates |= SD_STORE_TRAJ;

  gates_out = gates;
  nsamp_out = nsamp;
  return SMPC_OK;
}

// Path validity (utils::findPathCosts) and PathAlign's cumulative path lengths, into the tick
// block: pvalid[P-1], D[P-1].
static void path_tables(const smpc_ctx* c, const smpc_tick_in* in, uint32_t gates, const float* px,
                        const float* py, uint8_t* pvalid, float* D)
{
  const uint32_t P = in->path_len;
  // ---- path validity (utils::findPathCosts, tools/utils.hpp:361-395) ----------
  const uint32_t nseg = P > 0 ? P - 1 : 0;
  if (in->path_pts_valid) {
    memcpy(pvalid, in->path_pts_valid, nseg);
  } else if (gates & (SD_PATH_ALIGN | SD_PATH_FOLLOW | SD_PATH_ALIGN_LEGACY)) {
    for (uint32_t i = 0; i < nseg; ++i) {
      unsigned mx, my;
      uint8_t v = 1;
      if (!world_to_map(c->map, px[i], py[i], mx, my)) {
        v = 0;
      } else {
        const uint8_t cost = c->map.cells[static_cast<size_t>(my) * c->map.W + mx];
        if (cost == SMPC_COST_LETHAL || cost == SMPC_COST_INSCRIBED) v = 0;
        else if (cost == SMPC_COST_NO_INFORMATION) v = c->map.track_unknown ? 1 : 0;
      }
      pvalid[i] = v;
    }
  } else {
    memset(pvalid, 0, std::max(nseg, 1u));
  }

  // ---- PathAlign: cumulative path lengths (path_align_critic.cpp:82-90) --------
  if (nseg) {
    D[0] = 0.0f;
    for (uint32_t i = 1; i < nseg; ++i) {
      const float dx = px[i] - px[i - 1];
      const float dy = py[i] - py[i - 1];
      D[i] = D[i - 1] + sqrtf(dx * dx + dy * dy);
    }
  }

}

// Launch geometry of the tick: the costmap window staged in LDS (centred on the robot), the LDS
// carve-up and persistent grid of the wave-per-rollout pass, which scoring mode the enabled
// critics need, and whether — and how — the lane-per-rollout pass takes the tick.
static int plan_launch(smpc_ctx* c, const smpc_tick_in* in, uint32_t gates, uint32_t nsamp, int& mode_out)
{
  const uint32_t T = c->cfg.time_steps, B = c->cfg.batch_size, P = in->path_len;
  const auto& cr = c->critics;
  const uint32_t step = cr.path_align.trajectory_point_step;
  SmpcDev& d = c->dev;
  uint32_t window_bytes = 0;
  if (c->map.set && (gates & (SD_OBSTACLES | SD_COST))) {
    uint32_t side = 4;
    while ((side + 4) * (side + 4) <= kWindowBytes) side += 4;   // 96 cells
    // Long horizons reach further than that: 128 steps of 0.05 s at 0.5 m/s are 3.2 m, and a
    // rollout outside the window fetches its cells from the global map behind a full memory
    // wait, inside the time loop (the 262 144 x 128 pass: 183 us with the 96-cell window, 150 us
    // with 128..144 cells; beyond that the staging of the window by every block costs more than
    // the stragglers, tools/kbench.py with SMPC_WINDOW_SIDE_MAX).  For T > 64 the window covers
    // the reach at the largest constrained speed plus four standard deviations of the mean
    // noise, up to kWindowSideMax cells; the passes that run T > 64 (the re-read form, the
    // wave-per-rollout pass) have the LDS for it.
    static const bool small_window = getenv("SMPC_SMALL_WINDOW") != nullptr;              // (experiments; read once)
    static const char* const side_max_env = getenv("SMPC_WINDOW_SIDE_MAX");
    if (T > 64 && !small_window) {
      const float sigma = std::max(c->cfg.vx_std, c->holonomic ? c->cfg.vy_std : 0.f);
      const float vmax = std::max(std::max(std::fabs(c->c_vx_max), std::fabs(c->c_vx_min)), std::fabs(c->c_vy)) +
        4.f * sigma / std::sqrt(static_cast<float>(T));
      const double reach_cells = static_cast<double>(T) * c->cfg.model_dt * vmax / c->map.res;
      const uint32_t want = (static_cast<uint32_t>(2.0 * reach_cells) + 8u + 3u) & ~3u;
      uint32_t side_max = kWindowSideMax;
      if (side_max_env) side_max = static_cast<uint32_t>(atoi(side_max_env)) & ~3u;
      side = std::max(side, std::min(want, side_max));
    }
    const uint32_t ww = std::min(c->map.W, side), wh = std::min(c->map.H, side);
    long cx = static_cast<long>((in->pose_x - c->map.ox) / c->map.res);
    long cy = static_cast<long>((in->pose_y - c->map.oy) / c->map.res);
    long wx0 = cx - ww / 2, wy0 = cy - wh / 2;
    wx0 = std::max(0L, std::min(wx0, static_cast<long>(c->map.W) - static_cast<long>(ww)));
    wy0 = std::max(0L, std::min(wy0, static_cast<long>(c->map.H) - static_cast<long>(wh)));
    wx0 &= ~3L;
    d.win_x0 = static_cast<int32_t>(wx0); d.win_y0 = static_cast<int32_t>(wy0);
    d.win_w = static_cast<int32_t>(ww); d.win_h = static_cast<int32_t>(wh);
    window_bytes = ww * wh;
  }
  c->lds = make_lds(window_bytes, P, T, (pass_block(c->R) / 64), window_bytes != 0, nsamp);
  if (c->lds.total > kLdsPerCu) return fail(c, SMPC_ERR_UNSUPPORTED, "LDS budget exceeded");
  // room for the reduction inside the scoring launch (smpc_tail.h works in the launch's LDS) —
  // only for contexts that run that experiment: for 128 < T <= 256 the padding would halve the
  // wave pass's blocks per CU
  if (c->fused_reduce) c->lds.total = std::max(c->lds.total, smpc_tail_lds_bytes(T));

  // persistent grid: as many blocks as stay resident, never more than the work
  const uint32_t waves_per_block = (pass_block(c->R) / 64);
  int mode_now = c->score_mode_for(cr);
  // the lean kernels: MODE 0 scores the north star's five away from the goal, MODE 3 also the
  // additive forms of Cost, Goal, Constraint, Twirling, PathAngle (the deployed list) and of the
  // near-goal GoalAngle term; everything else (a cost_power other than 1, trajectory write-out,
  // path orientations, a footprint, VelocityDeadband) takes the general pass
  const uint32_t lean_extra = SD_CONSTRAINT | SD_COST | SD_GOAL | SD_TWIRLING | SD_PATH_ANGLE | SD_GOAL_ANGLE;
  if (gates & (SD_STORE_TRAJ | SD_USE_PATH_YAW | (SD_EXTRA_CRITICS & ~lean_extra))) {
    mode_now = 2;
  } else if (gates & lean_extra) {
    const bool unit_powers = (!cr.constraint.enabled || cr.constraint.cost_power == 1) &&
      (!cr.cost.enabled || cr.cost.cost_power == 1) && (!cr.goal.enabled || cr.goal.cost_power == 1) &&
      (!cr.twirling.enabled || cr.twirling.cost_power == 1) && (!cr.path_angle.enabled || cr.path_angle.cost_power == 1) &&
      (!(gates & SD_GOAL_ANGLE) || cr.goal_angle.cost_power == 1);
    mode_now = (mode_now == 0 && unit_powers) ? 3 : 2;
  }
  if (c->occ_lds != c->lds.total || c->occ_mode != mode_now) {
    int nb = 0;
    if (smpc_pass_occupancy(c->R, mode_now, T == 64u * static_cast<uint32_t>(c->R), pass_block(c->R), c->lds.total, &nb) != hipSuccess || nb < 1) nb = 1;
    c->occ_blocks = static_cast<uint32_t>(nb);
    c->occ_lds = c->lds.total;
    c->occ_mode = mode_now;
  }
  uint32_t per_cu = std::min(c->occ_blocks, 32u / waves_per_block);
  if (c->knob_max_blocks_per_cu >= 1) per_cu = std::min(per_cu, c->knob_max_blocks_per_cu);   // SMPC_MAX_BLOCKS_PER_CU
  uint32_t grid = std::min((B + waves_per_block - 1) / waves_per_block,
                           static_cast<uint32_t>(c->num_cu) * per_cu);
  c->grid = std::max(1u, std::min(grid, kMaxGrid));
  // the lane-per-rollout pass scores the north star's five, GoalAngle (power 1) included
  // (its GoalAngle instances: the parking form with ObstaclesCritic scored, T <= 64)
  const bool lane_ga = mode_now == 3 && !(gates & (lean_extra & ~SD_GOAL_ANGLE)) && (gates & SD_OBSTACLES) && T <= 64;
  // (its deployed-list instances, DEP: Constraint / Cost / Twirling on a cruise tick — Goal and
  // GoalAngle gated off, PathAngle inside its angle for every candidate furthest point, Cost
  // without Obstacles next to it, no Ackermann term, T = 64 or the default 56)
  const uint32_t dep_set = SD_CONSTRAINT | SD_COST | SD_TWIRLING;
  const bool lane_dep = mode_now == 3 && (gates & dep_set) && !(gates & (SD_GOAL | SD_GOAL_ANGLE)) &&
    !((gates & SD_PATH_ANGLE) && c->pang_any) && !((gates & SD_COST) && (gates & SD_OBSTACLES)) &&
    (gates & (SD_COST | SD_OBSTACLES)) && c->acker_r < 0.f && (T == 64 || T == 56);
  const bool lane_mode = mode_now == 0 || lane_ga || lane_dep;
  c->lane_now = c->use_tpr && lane_mode && T <= kLaneMaxT;
  // the lane pass samples PathAlign's trajectory points at the first step of every quad:
  // trajectory_point_step = 4, the reference's default (path_align_critic.cpp:36)
  if ((gates & SD_PATH_ALIGN) && step != 4) c->lane_now = false;
  // the re-read form (no parked controls): the only one for T > 64; instances exist with
  // ObstaclesCritic scored.  SMPC_LANE_REREAD=1 selects it for T <= 64 too (experiments).
  const bool force_rr = c->knob_lane_reread;
  c->lane_rr = (T > 64 || force_rr) && (gates & SD_OBSTACLES) != 0 && (T == 64 || T == 128);
  if (T > 64 && !c->lane_rr) c->lane_now = false;
  if (c->lane_now) {
    const SmpcLds Lt = lane_lds(window_bytes, P, T, c->lane_rr);
    c->lane_window_bytes = window_bytes;
    c->lds_tpr = Lt;
    if (Lt.total > kLdsPerCu) c->lane_now = false;   // long paths: the parked wz no longer fits
    // (the re-read form: no in-launch reduction, smpc_lane.hip)
    if (!c->lane_rr && c->fused_reduce) c->lds_tpr.total = std::max(Lt.total, smpc_tail_lds_bytes(T));
  }
  if (c->lane_now) {
    const uint32_t lblock = c->lane_rr ? smpc_lane_block_rr() : smpc_lane_block();
    const SmpcLds& Lt = c->lds_tpr;
    const uint32_t occ_key = Lt.total ^ (c->lane_rr ? 0x80000000u : 0u);
    if (c->occ_tpr_lds != occ_key) {
      int nb = 0;
      const hipError_t e = c->lane_rr ? smpc_lane_occupancy_rr(T, Lt.total, &nb) : smpc_lane_occupancy(T == 64, Lt.total, &nb);
      if (e != hipSuccess || nb < 1) nb = 1;
      if (c->lane_rr && nb > 3) nb = 3;   // (four-wave blocks: launch bounds of three waves per SIMD)
      c->occ_tpr_blocks = static_cast<uint32_t>(nb);
      c->occ_tpr_lds = occ_key;
    }
    const uint32_t groups = (B + 63) / 64;
    uint32_t wpb = lblock / 64;
    c->lane_block = lblock;
    // Small batches: when every group can have a SIMD to itself (at most four groups per CU),
    // blocks of four waves, one per CU.  Two waves on a SIMD share its VALU and each runs its
    // group in 22.5 us; a wave alone runs it in 20.5 (tools/lane_timeline.py) — and a batch
    // this small is one group per wave anyway: -2 us per tick at 65 536 x 64.
    if (!c->lane_rr && c->half_blocks && !c->in_group && groups <= static_cast<uint32_t>(c->num_cu) * 4u) {
      wpb = 4;
      c->lane_block = 256;
    }
    uint32_t g = std::min((groups + wpb - 1) / wpb, static_cast<uint32_t>(c->num_cu) * c->occ_tpr_blocks);
    g = std::max(1u, std::min(g, kMaxGrid));
    // Every wave takes whole groups, so a launch lasts ceil(groups / waves) group-times whatever
    // the remainder: launch only the waves that fill every round, and — where several blocks fit
    // a CU — pad the launch's LDS so that the dispatcher cannot stack them unevenly.  (Built for
    // the re-read form's first shape, four-wave blocks at three per CU: 4096 groups on 3072 waves;
    // with the eight-wave blocks it has now the grid is one block per CU and this trims nothing
    // at the benchmarked sizes.  SMPC_NO_BALANCED_GRID=1: off.)
    if (c->lane_rr && c->knob_balanced_grid) {
      const uint32_t waves = g * wpb;
      const uint32_t rounds = (groups + waves - 1) / waves;
      const uint32_t need = (groups + rounds - 1) / rounds;
      const uint32_t g2 = (need + wpb - 1) / wpb;
      const uint32_t per_cu = (g2 + c->num_cu - 1) / static_cast<uint32_t>(c->num_cu);
      if (g2 < g && per_cu < c->occ_tpr_blocks) {
        g = g2;
        const uint32_t pad = align_up(kLdsPerCu / (per_cu + 1) + 1024u, 16);
        if (pad <= kLdsPerCu / per_cu) c->lds_tpr.total = std::max(c->lds_tpr.total, pad);
      }
    }
    c->grid_tpr = g;
    // window-relative float cell index and its guard band.  The pass forms
    //   q~ = fma(ax, (1/res)_f, cxf),   cxf = ((x0 - window corner) / res)_f
    // from the accumulated displacement ax; the reference truncates
    //   Q = ((double)x - origin) / res - win_x0,   x = (float)(x0 + (double)ax).
    // |q~ - Q| is at most: the rounding of x to float, (1/2) ulp(|x|) / res with |x| inside the
    // window (lanes outside it never take the fast path); three float roundings of quantities
    // no larger than the window (the two constants' images and the fma's own); and the rounding
    // of the window corner origin + win0 * res in double.
    const double rinv = 1.0 / c->map.res;
    const double wx = c->map.ox + static_cast<double>(d.win_x0) * c->map.res;
    const double wy = c->map.oy + static_cast<double>(d.win_y0) * c->map.res;
    d.wxf = static_cast<float>(wx);
    d.wyf = static_cast<float>(wy);
    d.cxf = static_cast<float>((in->pose_x - wx) * rinv);
    d.cyf = static_cast<float>((in->pose_y - wy) * rinv);
    const double ext_x = static_cast<double>(d.win_w + 1) * c->map.res, ext_y = static_cast<double>(d.win_h + 1) * c->map.res;
    const double xmax = std::max(std::max(std::fabs(wx - c->map.res), std::fabs(wx + ext_x)),
                                 std::max(std::fabs(wy - c->map.res), std::fabs(wy + ext_y)));
    const double u24 = 5.9604644775390625e-08;   // 2^-24
    const double e_c = 2.3e-16 * (std::fabs(wx) + std::fabs(wy) + 1.0);
    const double qmax = static_cast<double>(std::max(d.win_w, d.win_h)) + 2.0;
    const double eps = 2.0 * (u24 * xmax * rinv + 3.0 * u24 * qmax + e_c * rinv) + 1e-7;
    d.cell_eps_w = static_cast<float>(std::min(eps, 0.5));
  }

  // Small batches at T = 64: while the waves of smpc_pass_split (16 rollouts, four lanes each) have
  // at most two groups each, the horizon split over four lanes runs the tick's one scoring pass in
  // a fraction of a 64-step chain (65 536 x 64: one wave per SIMD on the lane pass).  The plain five
  // critics only; the geometry above (window, cell index constants) is shared.
  c->split_now = false;
  if (c->lane_now && !c->lane_rr && !lane_ga && !lane_dep && mode_now == 0 && T <= 64 && T >= 36 && (T & 3u) == 0 &&
      (gates & SD_OBSTACLES) &&
      !c->in_group && !c->knob_no_split && !c->fused_reduce && (!c->lane_forced || c->knob_force_split)) {
    // (a context that ASKS for the lane pass — SMPC_FLAG_LANE_PER_ROLLOUT, SMPC_PASS=lane — gets it)
    // one block per CU (two waves per SIMD: the kernel's registers), and ONE group per wave: four
    // lanes per rollout while 16-rollout groups fit (32 768 rollouts on 256 CUs), two up to twice that
    const uint32_t waves = static_cast<uint32_t>(c->num_cu) * (smpc_split_block() / 64u);
    // (the two-segment instance — 32 steps per lane, up to 65 536 rollouts at one group per wave —
    // is built and parity-tested but not selected: a half chain at twice the instructions per
    // step, 50.0 us per tick at 65 536 x 64 where the lane pass takes 49.9; SMPC_SPLIT_NSEG=2)
    uint32_t nseg = 0;
    if ((B + 15u) / 16u <= waves) nseg = 4;
    // (horizons below 64 run the masked instance: 16 384 x 56 takes 38.0 us against 37.2 on the wave
    // pass, 20 000 x 60 37.0 against 39.5, 32 768 x 56 39.0 against 43.5)
    if (T != 64 && B < kSplitMinBatchShort && !c->knob_force_split) nseg = 0;
    if (c->knob_split_nseg == 2 || c->knob_split_nseg == 4) nseg = (nseg || c->knob_force_split) ? c->knob_split_nseg : 0;
    if (nseg == 2 && T != 64) nseg = 4;   // (the two-segment instance: T = 64 only)
    if (!nseg && c->knob_force_split) nseg = 4;
    if (nseg) {
      const SmpcLds Ls = split_lds(window_bytes, P, T, nseg);
      if (Ls.total <= kLdsPerCu) {
        const uint32_t key = Ls.total ^ (nseg << 28);
        if (c->occ_split_lds != key) {
          int nb = 0;
          if (smpc_split_occupancy(nseg, Ls.total, &nb) != hipSuccess || nb < 1) nb = 1;
          c->occ_split_blocks = nb;
          c->occ_split_lds = key;
        }
        const uint32_t per_block = smpc_split_rollouts_per_block(nseg);
        const uint32_t blocks = (B + per_block - 1) / per_block;
        const uint32_t resident = static_cast<uint32_t>(c->num_cu) * static_cast<uint32_t>(c->occ_split_blocks);
        c->split_now = true;
        c->split_nseg = nseg;
        c->lds_split = Ls;
        c->grid_split = std::max(1u, std::min(std::min(blocks, resident), kMaxGrid));   // (persistent: forced runs loop over groups)
      }
    }
  }

  // below kLaneMinBatch the lane pass itself loses to the wave pass (one group per CU's worth of
  // waves: crossover measured at ~50 k rollouts); such contexts keep the group-major noise only
  // for the split form
  if (c->lane_now && !c->split_now && B < kLaneMinBatch && !c->lane_forced) c->lane_now = false;

  mode_out = mode_now;
  return SMPC_OK;
}

int prepare_tick(smpc_ctx* c, const smpc_tick_in* in, const float* u_in)
{
  int rc = check_tick(c, in);
  if (rc != SMPC_OK) return rc;
  if (!u_in) return fail(c, SMPC_ERR_INVALID, "control sequence missing");
  rc = absorb_redraw(c);   // noise drawn in the background since the last tick, if any
  if (rc != SMPC_OK) return rc;
  const uint32_t T = c->cfg.time_steps, B = c->cfg.batch_size, P = in->path_len;
  const auto& cr = c->critics;
  HIPCK(c, hipSetDevice(c->device));

  const TickLayout tl = tick_layout(T, std::max(P, 1u));
  if (tl.total > c->tick_cap) return fail(c, SMPC_ERR_INVALID, "tick block overflow");
  uint8_t* h = c->h_tick;
  memcpy(h + tl.u, u_in, 3 * T * sizeof(float));
  if (!c->holonomic) memset(h + tl.u + T * sizeof(float), 0, T * sizeof(float));
  float* px = reinterpret_cast<float*>(h + tl.px);
  float* py = reinterpret_cast<float*>(h + tl.py);
  float* pyaw = reinterpret_cast<float*>(h + tl.pyaw);
  float* D = reinterpret_cast<float*>(h + tl.D);
  uint32_t* pf_idx = reinterpret_cast<uint32_t*>(h + tl.pf_idx);
  uint8_t* pvalid = h + tl.pvalid;
  uint8_t* pa_active = h + tl.pa_active;
  if (P) {
    memcpy(px, in->path_x, P * 4);
    memcpy(py, in->path_y, P * 4);
    memcpy(pyaw, in->path_yaw, P * 4);
  }

  uint32_t gates = 0, nsamp = 0;
  rc = tick_gates(c, in, gates, nsamp);
  if (rc != SMPC_OK) return rc;
  const double rx = in->pose_x, ry = in->pose_y, gx = in->goal_x, gy = in->goal_y;
  // (PathAlignLegacy alone: its own step — tick_gates refuses two different ones)
  const uint32_t step = (gates & SD_PATH_ALIGN) || !(gates & SD_PATH_ALIGN_LEGACY) ? cr.path_align.trajectory_point_step
                                                                                    : cr.path_align_legacy.trajectory_point_step;
  path_tables(c, in, gates, px, py, pvalid, D);

  // ---- per-candidate-furthest-point tables -------------------------------------
  const float yaw0 = in->pose_yaw;
  const float cos0 = cosf(yaw0), sin0 = sinf(yaw0);
  const float svx = static_cast<float>(in->speed_vx);
  // state.vy[:,0] = speed.linear.y only if holonomic (optimizer.cpp:264-266)
  const float svy = c->holonomic ? static_cast<float>(in->speed_vy) : 0.f;
  const float swz = static_cast<float>(in->speed_wz);
  const float dt = c->cfg.model_dt;
  // trajectories(0,0): first rollout point, identical for every rollout because
  // v[:,0] is the measured speed (optimizer.cpp:258-267,331-342)
  const float dx0 = svx * cos0 - svy * sin0;
  const float dy0 = svx * sin0 + svy * cos0;
  const float x00 = static_cast<float>(in->pose_x + static_cast<double>(dx0 * dt));
  const float y00 = static_cast<float>(in->pose_y + static_cast<double>(dy0 * dt));
  uint32_t cost_t0 = SMPC_COST_NO_INFORMATION;   // costAtPose of that point (obstacles_critic.cpp:203-212)
  if (c->map.set) {
    unsigned mx, my;
    if (world_to_map(c->map, x00, y00, mx, my))
      cost_t0 = c->map.cells[static_cast<size_t>(my) * c->map.W + mx];
  }
  // PathAlignCritic's and PathAlignLegacyCritic's gate per candidate furthest point S
  // (path_align_critic.cpp:58-74, path_align_legacy_critic.cpp:56-72: the same code)
  auto occupancy_gate = [&](const smpc_path_align_params& q, uint8_t* active) {
    // utils::findPathTrajectoryInitialPoint (tools/utils.hpp:327-344)
    size_t init = 0;
    float best = std::numeric_limits<float>::max();
    for (uint32_t j = 0; j < P; ++j) {
      const float ddx = px[j] - x00, ddy = py[j] - y00;
      const float d = ddx * ddx + ddy * ddy;
      if (d < best) {
        best = d;
        init = j;
      }
    }
    // occupancy of the path between the initial and the furthest point.  The reference
    // walks i = init..S-1 with a running count of invalid points and stops at the first i where
    // count / range > ratio and count > 2; the count only grows and range is fixed per S, so
    // that happens iff it holds for the final count: prefix sums, O(P).
    std::vector<uint32_t> inval(P + 1, 0);
    for (uint32_t i = 0; i < P; ++i) inval[i + 1] = inval[i] + ((i + 1 < P && !pvalid[i]) ? 1u : 0u);
    for (uint32_t S = 0; S < P; ++S) {
      bool on = S >= q.offset_from_furthest;
      if (on && S > init) {
        const unsigned int invalid_ctr = inval[S] - inval[init];
        const float range = static_cast<float>(static_cast<size_t>(S) - init);
        if (static_cast<float>(invalid_ctr) / range > q.max_path_occupancy_ratio && invalid_ctr > 2) on = false;
      }
      active[S] = on ? 1 : 0;
    }
  };
  if (gates & SD_PATH_ALIGN) occupancy_gate(cr.path_align, pa_active);
  else memset(pa_active, 0, std::max(P, 1u));
  uint8_t* pal_active = h + tl.pal_active;
  if (gates & SD_PATH_ALIGN_LEGACY) occupancy_gate(cr.path_align_legacy, pal_active);
  else memset(pal_active, 0, std::max(P, 1u));
  if (gates & SD_PATH_FOLLOW) {
    // path_follow_critic.cpp:46-57
    const size_t path_size = P - 1;
    for (uint32_t S = 0; S < P; ++S) {
      size_t idx = std::min(static_cast<size_t>(S) + cr.path_follow.offset_from_furthest, path_size);
      bool valid = false;
      while (!valid && idx < path_size - 1) {
        valid = pvalid[idx];
        if (!valid) idx++;
      }
      pf_idx[S] = static_cast<uint32_t>(idx);
    }
  } else {
    memset(pf_idx, 0, std::max(P, 1u) * 4);
  }

  uint8_t* pang_active = h + tl.pang_active;
  bool pang_correct = false;
  c->pang_any = false;     // PathAngleCritic live for some candidate furthest point this tick
  if (gates & SD_PATH_ANGLE) {
    // path_angle_critic.cpp:24-31,52-54: reversing / forward preference
    bool reversing_allowed = true;
    if (std::fabs(cr.path_angle.vx_min) < 1e-6) reversing_allowed = false;
    else if (cr.path_angle.vx_min < 0.0f) reversing_allowed = true;
    bool forward_preference = cr.path_angle.forward_preference != 0;
    if (!reversing_allowed) forward_preference = true;
    pang_correct = reversing_allowed && !forward_preference;
    for (uint32_t S = 0; S < P; ++S) {
      // :73-83 utils::posePointAngle (tools/utils.hpp:417-434) against the offset point
      const size_t idx = std::min(static_cast<size_t>(S) + cr.path_angle.offset_from_furthest,
                                  static_cast<size_t>(P) - 1);
      const float pose_x = static_cast<float>(rx), pose_y = static_cast<float>(ry);
      const double point_x = px[idx], point_y = py[idx];
      const float yaw = atan2f(static_cast<float>(point_y - static_cast<double>(pose_y)),
                               static_cast<float>(point_x - static_cast<double>(pose_x)));
      auto norm = [](double a) {
        const double theta = std::fmod(a + M_PI, 2.0 * M_PI);
        return theta <= 0.0 ? theta + M_PI : theta - M_PI;
      };
      const double pyaw0 = static_cast<double>(in->pose_yaw);
      float ang = static_cast<float>(std::fabs(norm(pyaw0 - static_cast<double>(yaw))));
      if (!forward_preference) {
        const double b = std::fabs(norm(norm(pyaw0 + M_PI) - static_cast<double>(yaw)));
        ang = static_cast<float>(std::min(std::fabs(norm(pyaw0 - static_cast<double>(yaw))), b));
      }
      pang_active[S] = ang < cr.path_angle.max_angle_to_furthest ? 0 : 1;
      if (pang_active[S]) c->pang_any = true;
    }
  } else {
    memset(pang_active, 0, std::max(P, 1u));
  }
  float* lut_cost = reinterpret_cast<float*>(h + tl.lut_cost);
  bool near_goal_cost = false;
  if (gates & SD_COST) {
    // cost_critic.cpp:120-124,141-155 per 8-bit cost (collisions are marked in the shared LUT)
    const bool near_goal_c = within_tol(cr.cost.near_goal_distance, rx, ry, gx, gy);
    near_goal_cost = near_goal_c;
    for (int v = 0; v < 256; ++v) {
      float t = 0.0f;
      if (v >= 1) {
        if (static_cast<float>(v) >= static_cast<float>(SMPC_COST_INSCRIBED)) t = cr.cost.critical_cost;
        else if (!near_goal_c) t = static_cast<float>(v);
      }
      lut_cost[v] = t;
    }
  } else {
    memset(lut_cost, 0, 256 * 4);
  }

  // ---- Obstacles LUT: rebuilt and uploaded only when its inputs changed -----------
  if (gates & (SD_OBSTACLES | SD_COST)) {
    const bool near_goal = within_tol(cr.obstacles.near_goal_distance, rx, ry, gx, gy);  // :124-127
    const bool cost_only = (gates & SD_COST) && !(gates & SD_OBSTACLES) && !(gates & (SD_FP_OBSTACLES | SD_FP_COST));
    const uint64_t key = (c->map_version << 20) ^ (c->critics_version << 4) ^ (near_goal ? 1u : 0u) ^
      ((gates & (SD_FP_OBSTACLES | SD_FP_COST)) ? 2u : 0u) ^ (cost_only ? 4u : 0u) ^ (near_goal_cost ? 8u : 0u);
    if (!c->lut_valid || key != c->lut_key) {
      build_lut(c, near_goal, c->h_lut, false, false, cost_only ? lut_cost : nullptr);
      HIPCK(c, hipMemcpyAsync(c->d_lut, c->h_lut, 256 * sizeof(SmpcLut), hipMemcpyHostToDevice,
                              c->stream));
      if (gates & (SD_FP_OBSTACLES | SD_FP_COST)) {
        build_lut(c, near_goal, c->h_lut_fp, true, false);
        build_lut(c, near_goal, c->h_lut_fp + 256, true, true);
        HIPCK(c, hipMemcpyAsync(c->d_lut_fp, c->h_lut_fp, 512 * sizeof(SmpcLut), hipMemcpyHostToDevice,
                                c->stream));
      }
      c->lut_key = key;
      c->lut_valid = true;
    }
  }

  {
    const int rc_map = wait_map_upload(c);
    if (rc_map != SMPC_OK) return rc_map;
  }
  if (c->cfg.flags & SMPC_FLAG_PROFILE) HIPCK(c, hipEventRecord(c->ev0, c->stream));
  // small ticks travel inside the kernel arguments (SmpcDev::tick_bytes) instead of a copy of
  // their own; the CostCritic table (general pass only) is not part of that
  // (contexts of the wave-per-rollout pass only: see smpc_lane.hip for why not the other)
  const bool inline_tick = !c->defer_upload && !c->use_tpr && T <= 64 && tl.lut_cost - tl.px <= SMPC_INLINE_TICK_CAP &&
    !c->knob_no_inline_tick;
  // SMPC_PINNED_TICK=1 (experiment): no copy at all — the kernels read the tick block where the
  // host assembled it, in pinned host memory.  Measured (tools/tail_ab.py SMPC_PINNED_TICK=1) and
  // not the default: every block of the grid fetches its ~2 KB across PCIe, uncached.
  const bool pinned_tick = c->knob_pinned_tick && !c->defer_upload && !inline_tick;
  const uint8_t* const tb = pinned_tick ? h : c->d_tick;
  // the tick block's number: written last, echoed by block 0 of every pass that reads the block
  // from device memory (SmpcDev::canary)
  if (++c->tick_no == 0) ++c->tick_no;
  *reinterpret_cast<uint32_t*>(h + tl.canary) = c->tick_no;
  c->tick_used = tl.total;
  const bool bar = c->bar_tick && !c->defer_upload && !pinned_tick && !inline_tick;
  // (whoever uploads: this ctx or its group; the in-launch reduction experiment does not carry it)
  c->canary_expect = (!pinned_tick && !inline_tick && !c->fused_reduce) ? c->tick_no : 0u;
  if (bar) {
    // no kernel of an earlier tick may still read the block (every tick ends with fetch_out;
    // a tick abandoned on an error does not)
    if (c->launched) {
      HIPCK(c, hipStreamSynchronize(c->stream));
      c->launched = false;
    }
    // SMPC_DEBUG_STALE_TICK=n (tests of the guard): tick n's block is NOT handed over — the pass
    // reads the previous tick's, echoes its number, and fetch_out has to fail the tick
    if (c->knob_stale_tick && c->tick_no == c->knob_stale_tick) {
    } else {
      bar_copy(c->d_tick, h, tl.total);
      bar_flush(c);
    }
  } else if (!c->defer_upload && !pinned_tick) {
    if (!inline_tick)
      HIPCK(c, hipMemcpyAsync(c->d_tick, h, tl.total, hipMemcpyHostToDevice, c->stream));
    else if (gates & SD_COST)
      HIPCK(c, hipMemcpyAsync(c->d_tick + tl.lut_cost, h + tl.lut_cost, 256 * sizeof(float), hipMemcpyHostToDevice,
                              c->stream));
  }

  // ---- kernel parameter block ------------------------------------------------------
  SmpcDev& d = c->dev;
  memset(&d, 0, sizeof(d));
  d.B = B; d.T = T; d.P = P; d.nsamp = nsamp; d.step = step;
  d.x0 = in->pose_x; d.y0 = in->pose_y;
  d.yaw0 = yaw0; d.cos0 = cos0; d.sin0 = sin0;
  d.svx = svx; d.svy = svy; d.swz = swz; d.dt = dt;
  d.nvx = c->d_nvx; d.nvy = c->d_nvy; d.nwz = c->d_nwz;
  d.tvx = c->d_tvx; d.tvy = c->d_tvy; d.twz = c->d_twz;
  d.u = reinterpret_cast<const float*>(tb + tl.u);
  d.traj_x = c->d_traj[0]; d.traj_y = c->d_traj[1]; d.traj_yaw = c->d_traj[2];
  d.map = c->d_map; d.W = c->map.W; d.H = c->map.H;
  d.ox = c->map.ox; d.oy = c->map.oy; d.res = c->map.res;
  d.cost_t0 = cost_t0;
  d.x00f = x00; d.y00f = y00;
  {
    // fast cell index: float quotient + guard band (see cost_at in smpc_kernels.hip)
    const double rinv = 1.0 / c->map.res;
    d.oxf = static_cast<float>(c->map.ox);
    d.oyf = static_cast<float>(c->map.oy);
    d.rinvf = static_cast<float>(rinv);
    const double e_o = std::max(std::fabs(c->map.ox - static_cast<double>(d.oxf)),
                                std::fabs(c->map.oy - static_cast<double>(d.oyf)));
    const double qmax = static_cast<double>(std::max(c->map.W, c->map.H)) + 2.0;
    const double eps = 2.0 * (e_o * rinv + 3.1 * 5.9604644775390625e-08 * qmax) + 1e-7;
    d.cell_eps = static_cast<float>(std::min(eps, 0.5));
  }
  d.lut = c->d_lut;
  d.px = reinterpret_cast<const float*>(tb + tl.px);
  d.py = reinterpret_cast<const float*>(tb + tl.py);
  d.pyaw = reinterpret_cast<const float*>(tb + tl.pyaw);
  d.D = reinterpret_cast<const float*>(tb + tl.D);
  d.pvalid = tb + tl.pvalid;
  d.pa_active = tb + tl.pa_active;
  d.pf_idx = reinterpret_cast<const uint32_t*>(tb + tl.pf_idx);
  d.obs_critical_w = cr.obstacles.critical_weight;
  d.obs_repulsion_w = cr.obstacles.repulsion_weight;
  d.obs_collision_cost = cr.obstacles.collision_cost;
  d.obs_rep_over_T = cr.obstacles.repulsion_weight / static_cast<float>(T);
  d.obs_power = cr.obstacles.cost_power;
  d.pa_weight = cr.path_align.cost_weight; d.pa_power = cr.path_align.cost_power;
  d.pf_weight = cr.path_follow.cost_weight; d.pf_power = cr.path_follow.cost_power;
  d.ga_weight = cr.goal_angle.cost_weight; d.ga_power = cr.goal_angle.cost_power;
  d.ga_goal_yaw = P ? pyaw[P - 1] : 0.f;
  d.pfw_weight = cr.prefer_forward.cost_weight; d.pfw_power = cr.prefer_forward.cost_power;
  d.con_weight = cr.constraint.cost_weight; d.con_power = cr.constraint.cost_power;
  {
    // ConstraintCritic::initialize (constraint_critic.cpp:36-38)
    const float min_sgn = cr.constraint.vx_min > 0.0f ? 1.0f : -1.0f;
    d.con_max_vel = sqrtf(cr.constraint.vx_max * cr.constraint.vx_max + cr.constraint.vy_max * cr.constraint.vy_max);
    d.con_min_vel = min_sgn * sqrtf(cr.constraint.vx_min * cr.constraint.vx_min + cr.constraint.vy_max * cr.constraint.vy_max);
    d.con_acker_r = c->acker_r;
  }
  d.lut_cost = reinterpret_cast<const float*>(tb + tl.lut_cost);
  d.cost_w254 = cr.cost.cost_weight / 254.0f;   // cost_critic.cpp:34
  d.cost_collision_cost = cr.cost.collision_cost; d.cost_power = cr.cost.cost_power;
  d.cost_critical = cr.cost.critical_cost;
  d.cost_near_goal = near_goal_cost ? 1u : 0u;
  d.goal_x = in->goal_x; d.goal_y = in->goal_y;
  d.goal_weight = cr.goal.cost_weight; d.goal_power = cr.goal.cost_power;
  d.tw_weight = cr.twirling.cost_weight; d.tw_power = cr.twirling.cost_power;
  d.pang_active = tb + tl.pang_active;
  d.pal_active = tb + tl.pal_active;
  d.pal_weight = cr.path_align_legacy.cost_weight;
  d.pal_power = cr.path_align_legacy.cost_power;
  {
    const uint32_t lstep = cr.path_align_legacy.trajectory_point_step;
    d.pal_eval = lstep ? static_cast<float>(T / lstep) : 0.f;   // traj_pts_eval = floor(T / step) (:83)
  }
  d.pang_weight = cr.path_angle.cost_weight; d.pang_power = cr.path_angle.cost_power;
  d.pang_offset = cr.path_angle.offset_from_furthest; d.pang_correct = pang_correct ? 1 : 0;
  d.db_vx = std::fabs(static_cast<double>(cr.velocity_deadband.deadband_velocities[0]));
  // no vy term for a non-holonomic model (velocity_deadband_critic.cpp:78-97); with
  // state.vy = 0 a zero deadband contributes max(0 - 0, 0) = 0
  d.db_vy = c->holonomic ? std::fabs(static_cast<double>(cr.velocity_deadband.deadband_velocities[1])) : 0.0;
  d.db_wz = std::fabs(static_cast<double>(cr.velocity_deadband.deadband_velocities[2]));
  d.db_weight = cr.velocity_deadband.cost_weight; d.db_power = cr.velocity_deadband.cost_power;
  d.lut_fp = c->d_lut_fp;
  d.fp_n = static_cast<uint32_t>(c->fp_x.size());
  for (uint32_t i = 0; i < d.fp_n; ++i) {
    d.fp_x[i] = c->fp_x[i];
    d.fp_y[i] = c->fp_y[i];
  }
  d.fp_pic = 0.0f;
  if (gates & (SD_FP_OBSTACLES | SD_FP_COST)) {
    // {Obstacles,Cost}Critic::findCircumscribedCost with InflationLayer::computeCost
    // (nav2_costmap_2d, Humble): the cost at the circumscribed radius, -1 without a layer
    double result = -1.0;
    if (c->fp_layer_scale >= 0.0) {
      const double distance = c->fp_circumscribed_radius / c->map.res;
      unsigned char cost = 0;
      if (distance == 0) {
        cost = SMPC_COST_LETHAL;
      } else if (distance * c->map.res <= static_cast<double>(c->map.inscribed_radius)) {
        cost = SMPC_COST_INSCRIBED;
      } else {
        const double factor = std::exp(-1.0 * c->fp_layer_scale *
                                       (distance * c->map.res - static_cast<double>(c->map.inscribed_radius)));
        cost = static_cast<unsigned char>((SMPC_COST_INSCRIBED - 1) * factor);
      }
      result = cost;
    }
    d.fp_pic = static_cast<float>(result);
  }
  d.g_vx = c->cfg.gamma / powf(c->cfg.vx_std, 2);
  d.g_vy = c->holonomic ? c->cfg.gamma / powf(c->cfg.vy_std, 2) : 0.f;   // optimizer.cpp:374-380
  d.g_wz = c->cfg.gamma / powf(c->cfg.wz_std, 2);
  d.neg_inv_temp = -1 / c->cfg.temperature;
  d.k2 = d.neg_inv_temp * 1.4426950408889634f;
  d.timeline = c->d_timeline;
  d.canary_echo = c->canary_expect ? 1u : 0u;
  if (inline_tick) memcpy(d.u_arg, h + tl.u, 3 * T * sizeof(float));
  if (inline_tick) {
    d.tick_inline = 1;
    d.u_inline = 1;
    const size_t o = tl.px;   // the path block: everything between u and the CostCritic table
    d.io_px = static_cast<uint16_t>(tl.px - o); d.io_py = static_cast<uint16_t>(tl.py - o);
    d.io_pyaw = static_cast<uint16_t>(tl.pyaw - o); d.io_D = static_cast<uint16_t>(tl.D - o);
    d.io_pf_idx = static_cast<uint16_t>(tl.pf_idx - o); d.io_pvalid = static_cast<uint16_t>(tl.pvalid - o);
    d.io_pa_active = static_cast<uint16_t>(tl.pa_active - o); d.io_pang_active = static_cast<uint16_t>(tl.pang_active - o);
    d.io_pal_active = static_cast<uint16_t>(tl.pal_active - o);
    memcpy(d.tick_bytes, h + o, tl.lut_cost - o);
  }
  d.partials = c->d_partials;
  d.furthest_out = reinterpret_cast<uint32_t*>(c->d_furthest);

  int mode_now = 0;
  rc = plan_launch(c, in, gates, nsamp, mode_now);
  if (rc != SMPC_OK) return rc;

  predict_hint(c, in, u_in);
  c->gate_flags = gates;
  // with a footprint the two collision critics no longer see the same set of colliding rollouts,
  // and a pass reports one non-colliding count (CostCritic's, scored first): smpc_optimize counts
  // ObstaclesCritic's with a pass of its own; the sharded tick and the grouped launch do not
  c->two_coll_fp = (gates & SD_OBSTACLES) && (gates & SD_COST) && (gates & (SD_FP_OBSTACLES | SD_FP_COST));
  c->score_mode = mode_now;
  c->fail_in = in->fail_flag_in != 0;
  c->P = P;
  c->tick_ready = true;
  return SMPC_OK;
}

}  // namespace smpc_impl
