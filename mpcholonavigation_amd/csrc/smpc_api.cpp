// smpc_api.cpp — host side of the C-ABI declared in include/smpc.h.
//
// Owns device memory, the per-tick upload block and the launch sequence; does
// the O(P) host work the reference's critics do once per tick (goal-distance
// gates, path validity, cumulative path lengths, the per-candidate PathAlign /
// PathFollow tables) and the 256-entry Obstacles lookup table.  All [B,T] work
// is in smpc_kernels.hip.  There is no CPU fallback: without a HIP device
// smpc_create() fails.

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only: RCCL itself is resolved at run time (dlopen)
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "../../include/smpc.h"
#include "smpc_dev.h"

hipError_t smpc_launch_pass(int R, int mode, const SmpcDev& p, const SmpcLds& L,
                            uint32_t grid, uint32_t block, hipStream_t st);
hipError_t smpc_pass_occupancy(int R, int mode, bool full, uint32_t block, uint32_t lds_bytes,
                               int* blocks_per_cu);
hipError_t smpc_set_pass_lds_limit(int bytes);
hipError_t smpc_launch_reduce(const float* partials, uint32_t nblk, uint32_t T,
                              float neg_inv_temp, float* tuple, const SmpcFinal& fin,
                              hipStream_t st);
hipError_t smpc_launch_combine(const float* tuples, uint32_t G, uint32_t T, float neg_inv_temp,
                               float vx_max, float vx_min, float vy_max, float wz_max,
                               float* u_out, float* result, const float* furthest_used,
                               float* host_out, uint32_t seq, hipStream_t st);
hipError_t smpc_launch_fill_noise(float* out, uint64_t n, uint64_t base, uint64_t seed,
                                  uint32_t stream, uint32_t epoch, float sigma, hipStream_t st);
hipError_t smpc_launch_ackermann(float* u_dev, float* u_host, uint32_t T, float min_r, uint32_t seq,
                                 hipStream_t st);

hipError_t smpc_launch_transpose(const float* src, float* dst, uint32_t B, uint32_t T, hipStream_t st);
hipError_t smpc_launch_pass_lane(const SmpcDev& p, const SmpcLds& L, uint32_t grid, hipStream_t st);
uint32_t smpc_lane_block();
hipError_t smpc_lane_occupancy(bool full, uint32_t lds_bytes, int* blocks_per_cu);
hipError_t smpc_lane_set_lds_limit(int bytes);
hipError_t smpc_launch_lane_reduce(const float* v, const float* w, float* out, hipStream_t st);
hipError_t smpc_launch_pass_lane_many(const SmpcDev* d_many, uint32_t n, bool full, bool obst,
                                      const SmpcLds& L, uint32_t grid, hipStream_t st);
hipError_t smpc_launch_reduce_many(const SmpcReduceArgs* d_many, uint32_t n, uint32_t T,
                                   float neg_inv_temp, hipStream_t st);
hipError_t smpc_launch_sincos(const float* x, uint32_t n, float* sn, float* cs, hipStream_t st);

namespace {

thread_local std::string g_create_error;

// RCCL entry points, resolved once.  The library is not a link-time dependency: a process
// that already holds RCCL (torch.distributed's "nccl" backend IS RCCL on ROCm) shares that
// copy, a single-GPU user never loads it.
struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
const RcclApi* rccl()
{
  static const RcclApi* api = []() -> const RcclApi* {
    void* h = nullptr;
    for (const char* name : {"librccl.so", "librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);   // the copy the process already has
      if (h) break;
    }
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      if (h) break;
      h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) return nullptr;
    static RcclApi a;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.AllReduce) return nullptr;
    return &a;
  }();
  return api;
}

// threads per block of the streaming pass: 16 waves share one costmap window and produce one
// partial (fewer partials to reduce); T > 128 needs more registers per lane than 16 waves allow
inline uint32_t pass_block(int R) {return R == 4 ? 512u : 1024u;}
constexpr uint32_t kLaneMinBatch = 60u * 1024u;   // lane-per-rollout pass from this batch size up (measured crossover ~50k: 65 536 x 64 takes 35.9 us against 40.4 us)
constexpr uint32_t kLaneMaxT = 64;        // it parks 3 x 64 noised controls per lane in registers
constexpr uint32_t kMaxGrid = 2048;       // smpc_reduce_partials stages this many factors
constexpr uint32_t kWindowBytes = 96 * 96;  // costmap window staged in LDS: 4.8 m x 4.8 m at
                                           // 0.05 m around the robot; the rest is read from HBM/L2
constexpr uint32_t kLdsPerCu = 160 * 1024;

inline uint32_t align_up(uint32_t v, uint32_t a) {return (v + a - 1) / a * a;}

struct HostCostmap {
  uint8_t* cells = nullptr;   // pinned mirror of the device copy: the host-side lookups read it
  size_t cap = 0;             // (path validity, first rollout point) and uploads DMA out of it
  uint32_t W = 0, H = 0;
  double ox = 0, oy = 0, res = 1;
  bool track_unknown = false;
  float inscribed_radius = 0, cost_scaling_factor = 0, inflation_radius = 0;
  bool set = false;
};

// Costmap2D::worldToMap (nav2_costmap_2d, Humble); call sites tools/utils.hpp:365-372
inline bool world_to_map(const HostCostmap& c, double wx, double wy, unsigned& mx, unsigned& my)
{
  if (wx < c.ox || wy < c.oy) return false;
  const double qx = (wx - c.ox) / c.res, qy = (wy - c.oy) / c.res;
  if (!(qx < 4294967296.0) || !(qy < 4294967296.0)) return false;
  mx = static_cast<unsigned>(qx);
  my = static_cast<unsigned>(qy);
  return mx < c.W && my < c.H;
}

// utils::withinPositionGoalTolerance(float, Pose, Pose) (tools/utils.hpp:233-249)
inline bool within_tol(float tol, double rx, double ry, double gx, double gy)
{
  const double dist_sq = std::pow(gx - rx, 2) + std::pow(gy - ry, 2);
  const float tol_sq = tol * tol;
  return dist_sq < tol_sq;
}

}  // namespace

struct smpc_ctx {
  smpc_config cfg{};
  smpc_critic_params critics{};
  float c_vx_max = 0, c_vx_min = 0, c_vy = 0, c_wz = 0;
  int device = 0;
  int num_cu = 256;
  hipStream_t own_stream = nullptr, stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t evp[8] = {};   // SMPC_FLAG_PROFILE: pairs around up to 4 scoring passes
  uint32_t evp_used = 0;
  // tensors
  float* d_nvx = nullptr;
  float* d_nvy = nullptr;
  float* d_nwz = nullptr;
  float* d_tvx = nullptr;       // time-major [T,B] copies for the lane-per-rollout pass:
  float* d_tvy = nullptr;       // one allocation, vy and wz follow vx
  float* d_twz = nullptr;
  bool use_tpr = false;      // time-major noise kept: the lane-per-rollout pass may run
  bool lane_now = false;     // ... and does for this tick (lean scoring mode)
  uint32_t last_pass_kind = 0;
  // member of a smpc_group: the group uploads every member's tick block in one copy
  bool defer_upload = false;
  uint32_t lane_window_bytes = 0;   // first LDS region of the lane pass this tick
  // consider_footprint: robot footprint (smpc_set_footprint) and the LUT pair built for it
  std::vector<double> fp_x, fp_y;
  double fp_circumscribed_radius = 0.0, fp_layer_scale = -1.0;
  SmpcLut* d_lut_fp = nullptr;   // [2][256]
  SmpcLut* h_lut_fp = nullptr;   // pinned
  // native RCCL exchange of the batch-sharded tick (smpc_shard_tick)
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_world = 0;
  float* d_all = nullptr;     // [world][4 + 3T] gathered shard tuples
  SmpcLds lds_tpr{};
  uint32_t grid_tpr = 0;
  uint32_t occ_tpr_blocks = 0, occ_tpr_lds = 0xffffffffu;
  float* d_costs[2] = {nullptr, nullptr};
  float* d_traj[3] = {nullptr, nullptr, nullptr};
  int costs_cur = 0;
  bool have_noise = false, rng_mode = false;
  uint64_t seed = 0;
  uint32_t epoch = 0;
  // costmap
  HostCostmap map;
  uint8_t* d_map = nullptr;
  size_t d_map_bytes = 0;
  hipEvent_t ev_map = nullptr;       // behind the last costmap upload
  bool map_pending = false;          // that upload may still be reading the pinned mirror
  uint64_t map_bytes_last = 0, map_bytes_total = 0;   // uploaded by the last call / so far
  unsigned long long* d_timeline = nullptr;   // SMPC_LANE_TIMELINE=1 (developer aid)
  // per-tick block
  SmpcLut* d_lut = nullptr;
  SmpcLut* h_lut = nullptr;     // pinned
  uint64_t lut_key = 0, map_version = 1, critics_version = 1;
  bool lut_valid = false;
  uint8_t* d_tick = nullptr;
  uint8_t* h_tick = nullptr;  // pinned
  size_t tick_cap = 0;
  // reductions / outputs
  float* d_partials = nullptr;
  float* d_tuple = nullptr;
  float* d_out = nullptr;       // [3T u][8 result]
  float* h_out = nullptr;       // pinned, device-mapped: kernels write the result here
  float* h_out_dev = nullptr;   // its device-side address
  float* d_furthest = nullptr;  // one float (atomicMax on its bits)
  // launch geometry
  int R = 1;
  uint32_t grid = 0;
  SmpcLds lds{};
  // per-tick prepared state
  SmpcDev dev{};
  uint32_t gate_flags = 0;   // critics past their host-side gates this tick
  int score_mode = 0;        // 0: every cost_power == 1 (one fused reduction), 2: general
  uint32_t occ_blocks = 1, occ_lds = 0xffffffffu;
  int occ_mode = -1;
  static int score_mode_for(const smpc_critic_params& cr)
  {
    return (cr.obstacles.cost_power == 1 && cr.path_align.cost_power == 1 &&
           cr.path_follow.cost_power == 1 && cr.goal_angle.cost_power == 1 &&
           cr.prefer_forward.cost_power == 1) ? 0 : 2;
  }
  bool tick_ready = false;
  bool fail_in = false;
  uint32_t P = 0;
  uint32_t passes = 0;
  // speculation on furthest_reached_path_point: last tick's value
  bool hint_valid = false;
  uint32_t hint = 0;
  uint64_t spec_misses = 0;
  // completion polling on the host-mapped result (SMPC_NO_POLL=1 disables)
  bool poll_enabled = true;
  // Optimizer::isHolonomic (optimizer.cpp:235).  A non-holonomic model is the Omni data path
  // with the vy noise, control_sequence.vy and state.vy[:,0] all zero: then state.vy = 0,
  // dx = vx cos - 0 sin, the vy gamma term and the weighted vy update are exactly 0 — what the
  // reference's isHolonomic() branches compute (optimizer.cpp:220-224,241-243,264-266,334-337,
  // 374-389), without a second set of kernels.
  bool holonomic = true;
  float acker_r = -1.f;      // Ackermann min_turning_r, < 0 for the other models
  uint32_t acker_seq = 0;    // completion word the Ackermann launch publishes this tick
  uint32_t seq = 0, poll_seq = 0;
  std::string err;
};

namespace {

int fail(smpc_ctx* c, int code, const std::string& msg)
{
  if (c) c->err = msg; else g_create_error = msg;
  return code;
}

#define HIPCK(ctx, call)                                                               \
  do {                                                                                 \
    hipError_t e__ = (call);                                                           \
    if (e__ != hipSuccess)                                                             \
      return fail(ctx, SMPC_ERR_DEVICE,                                                \
                  std::string(#call) + ": " + hipGetErrorString(e__));                \
  } while (0)

// A costmap upload is asynchronous (pinned mirror -> device on the ctx's stream).  Kernels of
// the same stream are ordered behind it; this host-side wait covers the rest: a tick
// launched on another stream (grouped ticks, smpc_set_stream) and the next overwrite of
// the mirror.
int wait_map_upload(smpc_ctx* c)
{
  if (!c->map_pending) return SMPC_OK;
  HIPCK(c, hipEventSynchronize(c->ev_map));
  c->map_pending = false;
  return SMPC_OK;
}

void free_ctx(smpc_ctx* c)
{
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (float* p : {c->d_tvx, c->d_nvx, c->d_nvy, c->d_nwz, c->d_costs[0], c->d_costs[1], c->d_traj[0],
         c->d_traj[1], c->d_traj[2], c->d_partials, c->d_tuple, c->d_out, c->d_furthest})
    if (p) (void)hipFree(p);
  if (c->d_map) (void)hipFree(c->d_map);
  if (c->d_timeline) (void)hipFree(c->d_timeline);
  if (c->map.cells) (void)hipHostFree(c->map.cells);
  if (c->ev_map) (void)hipEventDestroy(c->ev_map);
  if (c->d_tick && !c->defer_upload) (void)hipFree(c->d_tick);
  if (c->d_lut) (void)hipFree(c->d_lut);
  if (c->d_lut_fp) (void)hipFree(c->d_lut_fp);
  if (c->h_lut_fp) (void)hipHostFree(c->h_lut_fp);
  if (c->h_lut) (void)hipHostFree(c->h_lut);
  if (c->h_tick && !c->defer_upload) (void)hipHostFree(c->h_tick);   // (a group's slot is not ours)
  if (c->h_out) (void)hipHostFree(c->h_out);
  if (c->comm && rccl()) (void)rccl()->CommDestroy(c->comm);
  if (c->d_all) (void)hipFree(c->d_all);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  for (hipEvent_t e : c->evp) if (e) (void)hipEventDestroy(e);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

// tick block layout (offsets in bytes), sized for the ctx's T and SMPC_MAX_PATH
struct TickLayout {
  size_t u, px, py, pyaw, D, pf_idx, pvalid, pa_active, pang_active, lut_cost, total;
};
TickLayout tick_layout(uint32_t T, uint32_t P)
{
  TickLayout l{};
  size_t o = 0;
  l.u = o; o += align_up(3 * T * 4, 16);
  l.px = o; o += align_up(P * 4, 16);
  l.py = o; o += align_up(P * 4, 16);
  l.pyaw = o; o += align_up(P * 4, 16);
  l.D = o; o += align_up(P * 4, 16);
  l.pf_idx = o; o += align_up(P * 4, 16);
  l.pvalid = o; o += align_up(P, 16);
  l.pa_active = o; o += align_up(P, 16);
  l.pang_active = o; o += align_up(P, 16);
  l.lut_cost = o; o += 256 * 4;   // CostCritic repulsive term per 8-bit cost
  l.total = o;
  return l;
}

// LDS carve-up of the streaming pass.  nsamp = PathAlign samples per rollout (0: off).
SmpcLds make_lds(uint32_t window_bytes, uint32_t P, uint32_t T, uint32_t nwave, bool with_map,
                 uint32_t nsamp);
SmpcLds lane_lds(uint32_t window_bytes, uint32_t P, uint32_t T);

SmpcLds make_lds(uint32_t window_bytes, uint32_t P, uint32_t T, uint32_t nwave, bool with_map,
                 uint32_t nsamp = 0)
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
// per lane (cell_byte_exact), LUT, path, per wave the parked wz [64][68] + weights [64]
SmpcLds lane_lds(uint32_t window_bytes, uint32_t P, uint32_t T)
{
  const uint32_t lblock = smpc_lane_block();
  SmpcLds Lt = make_lds(window_bytes ? window_bytes + 1 + lblock : 0, P, T, lblock / 64,
                        window_bytes != 0, 0);
  Lt.off_pts4 = Lt.off_scr;
  Lt.off_scr += align_up(std::max(P, 1u) * 16, 16);
  Lt.scr_stride = align_up(std::max(64u * 68u + 64u, 4u + 3u * T), 4);
  Lt.total = Lt.off_scr + (lblock / 64) * Lt.scr_stride * 4;
  return Lt;
}

// distanceToObstacle (obstacles_critic.cpp:99-112) for an 8-bit cost, point mode
float distance_to_obstacle(const HostCostmap& m, float cost, bool using_footprint = false)
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
void build_lut(const smpc_ctx* c, bool near_goal, SmpcLut* lut, bool consider_fp = false,
               bool using_fp = false)
{
  const auto& m = c->map;
  const auto& p = c->critics.obstacles;
  for (int v = 0; v < 256; ++v) {
    lut[v].crit = 0.f;
    lut[v].rep = 0.f;
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
  // with a footprint the two collision critics no longer see the same set of colliding
  // rollouts, and the tuple carries one non-colliding count
  if (c->critics.obstacles.enabled && c->critics.cost.enabled &&
    (c->critics.obstacles.consider_footprint || c->critics.cost.consider_footprint))
    return fail(c, SMPC_ERR_UNSUPPORTED,
                "consider_footprint=true with both ObstaclesCritic and CostCritic in the list");
  return SMPC_OK;
}

// Everything the reference's critics decide once per tick on the host, plus the
// upload of the tick block.  Leaves c->dev ready for the launches.
int prepare_tick(smpc_ctx* c, const smpc_tick_in* in, const float* u_in)
{
  int rc = check_tick(c, in);
  if (rc != SMPC_OK) return rc;
  if (!u_in) return fail(c, SMPC_ERR_INVALID, "control sequence missing");
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

  // ---- host-side gates: one withinPositionGoalTolerance per critic
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
  if (gates & (SD_PATH_ALIGN | SD_PATH_FOLLOW | SD_PATH_ANGLE)) gates |= SD_NEED_FURTHEST;
  if (c->map.track_unknown) gates |= SD_TRACK_UNKNOWN;
  if (c->cfg.flags & SMPC_FLAG_STORE_TRAJECTORIES) gates |= SD_STORE_TRAJ;

  // ---- path validity (utils::findPathCosts, tools/utils.hpp:361-395) ----------
  const uint32_t nseg = P > 0 ? P - 1 : 0;
  if (in->path_pts_valid) {
    memcpy(pvalid, in->path_pts_valid, nseg);
  } else if (gates & (SD_PATH_ALIGN | SD_PATH_FOLLOW)) {
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
  if (gates & SD_PATH_ALIGN) {
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
    // :64-74 occupancy of the path between the initial and the furthest point.  The
    // reference walks i = init..S-1 with a running count of invalid points and stops at the
    // first i where count / range > ratio and count > 2; the count only grows and range is
    // fixed per S, so that happens iff it holds for the final count: prefix sums, O(P).
    std::vector<uint32_t> inval(P + 1, 0);
    for (uint32_t i = 0; i < P; ++i) inval[i + 1] = inval[i] + ((i + 1 < P && !pvalid[i]) ? 1u : 0u);
    for (uint32_t S = 0; S < P; ++S) {
      bool on = S >= cr.path_align.offset_from_furthest;     // path_align_critic.cpp:58-61
      if (on && S > init) {
        const unsigned int invalid_ctr = inval[S] - inval[init];
        const float range = static_cast<float>(static_cast<size_t>(S) - init);
        if (static_cast<float>(invalid_ctr) / range > cr.path_align.max_path_occupancy_ratio &&
          invalid_ctr > 2)
        {
          on = false;
        }
      }
      pa_active[S] = on ? 1 : 0;
    }
  } else {
    memset(pa_active, 0, std::max(P, 1u));
  }
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
    }
  } else {
    memset(pang_active, 0, std::max(P, 1u));
  }
  float* lut_cost = reinterpret_cast<float*>(h + tl.lut_cost);
  if (gates & SD_COST) {
    // cost_critic.cpp:120-124,141-155 per 8-bit cost (collisions are marked in the shared LUT)
    const bool near_goal_c = within_tol(cr.cost.near_goal_distance, rx, ry, gx, gy);
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
    const uint64_t key = (c->map_version << 20) ^ (c->critics_version << 2) ^ (near_goal ? 1u : 0u) ^
      ((gates & (SD_FP_OBSTACLES | SD_FP_COST)) ? 2u : 0u);
    if (!c->lut_valid || key != c->lut_key) {
      build_lut(c, near_goal, c->h_lut);
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
  if (!c->defer_upload)
    HIPCK(c, hipMemcpyAsync(c->d_tick, h, tl.total, hipMemcpyHostToDevice, c->stream));

  // ---- kernel parameter block ------------------------------------------------------
  SmpcDev& d = c->dev;
  memset(&d, 0, sizeof(d));
  d.B = B; d.T = T; d.P = P; d.nsamp = nsamp; d.step = step;
  d.x0 = in->pose_x; d.y0 = in->pose_y;
  d.yaw0 = yaw0; d.cos0 = cos0; d.sin0 = sin0;
  d.svx = svx; d.svy = svy; d.swz = swz; d.dt = dt;
  d.nvx = c->d_nvx; d.nvy = c->d_nvy; d.nwz = c->d_nwz;
  d.tvx = c->d_tvx; d.tvy = c->d_tvy; d.twz = c->d_twz;
  d.u = reinterpret_cast<const float*>(c->d_tick + tl.u);
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
  d.px = reinterpret_cast<const float*>(c->d_tick + tl.px);
  d.py = reinterpret_cast<const float*>(c->d_tick + tl.py);
  d.pyaw = reinterpret_cast<const float*>(c->d_tick + tl.pyaw);
  d.D = reinterpret_cast<const float*>(c->d_tick + tl.D);
  d.pvalid = c->d_tick + tl.pvalid;
  d.pa_active = c->d_tick + tl.pa_active;
  d.pf_idx = reinterpret_cast<const uint32_t*>(c->d_tick + tl.pf_idx);
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
  d.lut_cost = reinterpret_cast<const float*>(c->d_tick + tl.lut_cost);
  d.cost_w254 = cr.cost.cost_weight / 254.0f;   // cost_critic.cpp:34
  d.cost_collision_cost = cr.cost.collision_cost; d.cost_power = cr.cost.cost_power;
  d.goal_x = in->goal_x; d.goal_y = in->goal_y;
  d.goal_weight = cr.goal.cost_weight; d.goal_power = cr.goal.cost_power;
  d.tw_weight = cr.twirling.cost_weight; d.tw_power = cr.twirling.cost_power;
  d.pang_active = c->d_tick + tl.pang_active;
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
  d.partials = c->d_partials;
  d.furthest_out = reinterpret_cast<uint32_t*>(c->d_furthest);

  // ---- costmap window staged in LDS, centred on the robot ---------------------------
  uint32_t window_bytes = 0;
  if (c->map.set && (gates & (SD_OBSTACLES | SD_COST))) {
    uint32_t side = 4;
    while ((side + 4) * (side + 4) <= kWindowBytes) side += 4;   // 96 cells
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

  // persistent grid: as many blocks as stay resident, never more than the work
  const uint32_t waves_per_block = (pass_block(c->R) / 64);
  int mode_now = c->score_mode_for(cr);
  if (gates & (SD_STORE_TRAJ | SD_USE_PATH_YAW | SD_GOAL_ANGLE | SD_EXTRA_CRITICS)) mode_now = 2;   // lean kernels lack these
  if (c->occ_lds != c->lds.total || c->occ_mode != mode_now) {
    int nb = 0;
    if (smpc_pass_occupancy(c->R, mode_now, T == 64u * static_cast<uint32_t>(c->R), pass_block(c->R), c->lds.total, &nb) != hipSuccess || nb < 1) nb = 1;
    c->occ_blocks = static_cast<uint32_t>(nb);
    c->occ_lds = c->lds.total;
    c->occ_mode = mode_now;
  }
  uint32_t per_cu = std::min(c->occ_blocks, 32u / waves_per_block);
  if (const char* e = getenv("SMPC_MAX_BLOCKS_PER_CU")) {   // tuning knob
    const uint32_t lim = static_cast<uint32_t>(atoi(e));
    if (lim >= 1) per_cu = std::min(per_cu, lim);
  }
  uint32_t grid = std::min((B + waves_per_block - 1) / waves_per_block,
                           static_cast<uint32_t>(c->num_cu) * per_cu);
  c->grid = std::max(1u, std::min(grid, kMaxGrid));
  c->lane_now = c->use_tpr && mode_now == 0 && T <= kLaneMaxT;
  // the lane pass samples PathAlign's trajectory points at the first step of every quad:
  // trajectory_point_step = 4, the reference's default (path_align_critic.cpp:36)
  if ((gates & SD_PATH_ALIGN) && step != 4) c->lane_now = false;
  if (c->lane_now) {
    const SmpcLds Lt = lane_lds(window_bytes, P, T);
    c->lane_window_bytes = window_bytes;
    c->lds_tpr = Lt;
    if (Lt.total > kLdsPerCu) c->lane_now = false;   // long paths: the parked wz no longer fits
  }
  if (c->lane_now) {
    const uint32_t lblock = smpc_lane_block();
    const SmpcLds& Lt = c->lds_tpr;
    if (c->occ_tpr_lds != Lt.total) {
      int nb = 0;
      if (smpc_lane_occupancy(T == 64, Lt.total, &nb) != hipSuccess || nb < 1) nb = 1;
      c->occ_tpr_blocks = static_cast<uint32_t>(nb);
      c->occ_tpr_lds = Lt.total;
    }
    const uint32_t groups = (B + 63) / 64, wpb = lblock / 64;
    uint32_t g = std::min((groups + wpb - 1) / wpb, static_cast<uint32_t>(c->num_cu) * c->occ_tpr_blocks);
    c->grid_tpr = std::max(1u, std::min(g, kMaxGrid));
    // window-relative float cell index and its guard band (cost_at_lane)
    const double rinv = 1.0 / c->map.res;
    const double wx = c->map.ox + static_cast<double>(d.win_x0) * c->map.res;
    const double wy = c->map.oy + static_cast<double>(d.win_y0) * c->map.res;
    d.wxf = static_cast<float>(wx);
    d.wyf = static_cast<float>(wy);
    const double e_o = std::max(std::fabs(wx - static_cast<double>(d.wxf)),
                                std::fabs(wy - static_cast<double>(d.wyf)));
    // the reference divides (x - origin) by the resolution; the window corner is
    // origin + win0 * res in double: one more rounding of that product and sum
    const double e_c = 2.3e-16 * (std::fabs(wx) + std::fabs(wy) + 1.0);
    const double qmax = static_cast<double>(std::max(d.win_w, d.win_h)) + 2.0;
    const double eps = 2.0 * ((e_o + e_c) * rinv + 3.1 * 5.9604644775390625e-08 * qmax) + 1e-7;
    d.cell_eps_w = static_cast<float>(std::min(eps, 0.5));
  }

  c->gate_flags = gates;
  c->score_mode = mode_now;
  c->fail_in = in->fail_flag_in != 0;
  c->P = P;
  c->tick_ready = true;
  return SMPC_OK;
}

int launch_furthest(smpc_ctx* c, float* d_furthest)
{
  HIPCK(c, hipMemsetAsync(d_furthest, 0, sizeof(float), c->stream));
  SmpcDev d = c->dev;
  d.flags = c->gate_flags & SD_NEED_FURTHEST;
  d.furthest_out = reinterpret_cast<uint32_t*>(d_furthest);
  SmpcLds L = make_lds(0, c->P, d.T, (pass_block(c->R) / 64), false);
  HIPCK(c, smpc_launch_pass(c->R, 1, d, L, c->grid, pass_block(c->R), c->stream));
  return SMPC_OK;
}

// one scoring pass + block reduction -> tuple
// finish_furthest: when `finish`, the reduction also produces the new control sequence
// (single-GPU tick) and reports *finish_furthest (or the pass's own value) as the furthest
// point the critics used
// parameter blocks of one scoring pass + reduction of ctx c (no launch)
void fill_score_args(smpc_ctx* c, uint32_t flags, const float* u_dev, const float* d_furthest,
                     uint32_t furthest_hint, bool finish, const float* finish_furthest, SmpcDev& d,
                     SmpcFinal& fin)
{
  d = c->dev;
  d.flags = flags;
  if (u_dev) d.u = u_dev;
  d.d_furthest = d_furthest;
  d.furthest_hint = furthest_hint;
  d.costs = c->d_costs[c->costs_cur];
  d.costs_prev = c->d_costs[c->costs_cur ^ 1];
  fin = SmpcFinal{};
  fin.enabled = finish ? 1 : 0;
  fin.vx_max = c->c_vx_max; fin.vx_min = c->c_vx_min; fin.vy_max = c->c_vy; fin.wz_max = c->c_wz;
  fin.u_dev = c->d_out; fin.u_host = c->h_out_dev; fin.furthest_used = finish_furthest;
  if (finish && c->poll_enabled) {
    fin.done_counter = reinterpret_cast<uint32_t*>(c->d_furthest) + 2;
    fin.seq = ++c->seq;
    if (fin.seq == 0) fin.seq = ++c->seq;
    c->poll_seq = fin.seq;
  }
  if (finish && c->acker_r >= 0.f) {   // the Ackermann launch behind the reduction publishes
    c->acker_seq = fin.seq;
    fin.done_counter = nullptr;
    fin.seq = 0;
  }
}

int launch_score(smpc_ctx* c, uint32_t flags, const float* u_dev, const float* d_furthest,
                 uint32_t furthest_hint, float* d_tuple, bool finish = false,
                 const float* finish_furthest = nullptr)
{
  SmpcDev d;
  SmpcFinal fin;
  fill_score_args(c, flags, u_dev, d_furthest, furthest_hint, finish, finish_furthest, d, fin);
  const bool prof = (c->cfg.flags & SMPC_FLAG_PROFILE) && c->evp_used + 2 <= 8;
  if (prof) HIPCK(c, hipEventRecord(c->evp[c->evp_used], c->stream));
  uint32_t nblk = c->grid;
  // the lane-per-rollout pass scores with the full lean critic stack only
  if (c->lane_now && !(flags & SD_STORE_TRAJ)) {
    nblk = c->grid_tpr;
    HIPCK(c, smpc_launch_pass_lane(d, c->lds_tpr, nblk, c->stream));
    c->last_pass_kind = 1;
  } else {
    c->last_pass_kind = 0;
    HIPCK(c, smpc_launch_pass(c->R, c->score_mode, d, c->lds, c->grid, pass_block(c->R), c->stream));
  }
  if (prof) {
    HIPCK(c, hipEventRecord(c->evp[c->evp_used + 1], c->stream));
    c->evp_used += 2;
  }
  HIPCK(c, smpc_launch_reduce(c->d_partials, nblk, d.T, d.neg_inv_temp, d_tuple, fin, c->stream));
  if (finish && c->acker_r >= 0.f)
    HIPCK(c, smpc_launch_ackermann(c->d_out, c->h_out_dev, d.T, c->acker_r, c->acker_seq, c->stream));
  c->passes++;
  return SMPC_OK;
}

int launch_combine(smpc_ctx* c, const float* d_tuples, uint32_t n, const float* d_furthest_used)
{
  const uint32_t T = c->cfg.time_steps;
  uint32_t seq = 0;
  if (c->poll_enabled) {
    seq = ++c->seq;
    if (seq == 0) seq = ++c->seq;
    c->poll_seq = seq;
  }
  const bool acker = c->acker_r >= 0.f;
  HIPCK(c, smpc_launch_combine(d_tuples, n, T, c->dev.neg_inv_temp, c->c_vx_max, c->c_vx_min,
                               c->c_vy, c->c_wz, c->d_out, c->d_out + 3 * T, d_furthest_used,
                               c->h_out_dev, acker ? 0u : seq, c->stream));
  if (acker) HIPCK(c, smpc_launch_ackermann(c->d_out, c->h_out_dev, T, c->acker_r, seq, c->stream));
  return SMPC_OK;
}

// control_sequence_.vy is never written for a non-holonomic model (optimizer.cpp:387-389):
// the caller's row stays as it was
void store_control_sequence(const smpc_ctx* c, float* u_inout)
{
  const uint32_t T = c->cfg.time_steps;
  if (c->holonomic) {
    memcpy(u_inout, c->h_out, 3 * T * sizeof(float));
    return;
  }
  memcpy(u_inout, c->h_out, T * sizeof(float));
  memcpy(u_inout + 2 * T, c->h_out + 2 * T, T * sizeof(float));
}

int fetch_out(smpc_ctx* c)
{
  // The finishing kernel wrote u and the result into host-mapped memory and then the
  // tick's sequence number: spin on that word (a few microseconds sooner than the
  // runtime's stream wait), and fall back to the stream wait, which also surfaces errors.
  if (c->poll_seq) {
    const volatile uint32_t* flag =
      reinterpret_cast<const volatile uint32_t*>(c->h_out + 3 * c->cfg.time_steps + 7);
    const uint32_t want = c->poll_seq;
    c->poll_seq = 0;
    for (uint32_t spin = 0; spin < 4000000u; ++spin) {
      if (*flag == want) {
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        return SMPC_OK;
      }
      __builtin_ia32_pause();
    }
  }
  HIPCK(c, hipStreamSynchronize(c->stream));
  return SMPC_OK;
}

float profile_pass_ms(smpc_ctx* c)
{
  float sum = 0.f;
  uint32_t n = 0;
  for (uint32_t i = 0; i + 1 < c->evp_used; i += 2) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->evp[i], c->evp[i + 1]) == hipSuccess) {
      sum += ms;
      n++;
    }
  }
  return n ? sum / n : 0.f;
}

// What the reference had scored when a collision critic found every rollout colliding
// (critic_manager.cpp:70-73 stops after it): the list of include/smpc.h up to and including
// the first enabled collision critic — Constraint, Cost | Obstacles.
uint32_t fail_only_flags(const smpc_ctx* c)
{
  const uint32_t coll = (c->gate_flags & SD_COST) ? SD_COST : SD_OBSTACLES;
  return c->gate_flags & (SD_CONSTRAINT | coll | SD_STORE_TRAJ | SD_TRACK_UNKNOWN);
}

uint32_t scoring_flags(const smpc_ctx* c, bool fail_sticky)
{
  // CriticManager::evalTrajectoriesScores breaks on fail_flag (critic_manager.cpp:70-73)
  const uint32_t keep = SD_STORE_TRAJ | SD_TRACK_UNKNOWN;
  return fail_sticky ? (c->gate_flags & keep) : c->gate_flags;
}

// keep the time-major copies in step with the [B,T] tensors
int update_time_major(smpc_ctx* c)
{
  if (!c->use_tpr) return SMPC_OK;
  const uint32_t B = c->cfg.batch_size, T = c->cfg.time_steps;
  HIPCK(c, smpc_launch_transpose(c->d_nvx, c->d_tvx, B, T, c->stream));
  HIPCK(c, smpc_launch_transpose(c->d_nvy, c->d_tvy, B, T, c->stream));
  HIPCK(c, smpc_launch_transpose(c->d_nwz, c->d_twz, B, T, c->stream));
  return SMPC_OK;
}

int draw_noise(smpc_ctx* c)
{
  const uint64_t n = static_cast<uint64_t>(c->cfg.batch_size) * c->cfg.time_steps;
  const uint64_t base = c->cfg.shard_offset * c->cfg.time_steps;
  // draw order vx, wz, vy (noise_generator.cpp:107-122)
  HIPCK(c, smpc_launch_fill_noise(c->d_nvx, n, base, c->seed, 0, c->epoch, c->cfg.vx_std, c->stream));
  HIPCK(c, smpc_launch_fill_noise(c->d_nwz, n, base, c->seed, 1, c->epoch, c->cfg.wz_std, c->stream));
  // noises_vy_ keeps its zeros for a non-holonomic model (noise_generator.cpp:117-121)
  if (c->holonomic)
    HIPCK(c, smpc_launch_fill_noise(c->d_nvy, n, base, c->seed, 2, c->epoch, c->cfg.vy_std, c->stream));
  int rc = update_time_major(c);
  if (rc != SMPC_OK) return rc;
  HIPCK(c, hipStreamSynchronize(c->stream));
  c->have_noise = true;
  return SMPC_OK;
}

}  // namespace

extern "C" {

void smpc_config_default(smpc_config* c)
{
  memset(c, 0, sizeof(*c));
  c->batch_size = 1000;   // ref src/optimizer.cpp:69-82
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
  c->ackermann_min_turning_r = 0.2f;   // ref motion_models.hpp:94
}

void smpc_critic_params_default(smpc_critic_params* p)
{
  memset(p, 0, sizeof(*p));
  p->obstacles = {1, 0, 1, 1.5f, 20.0f, 10000.0f, 0.10f, 0.5f};
  p->path_align = {1, 0, 1, 10.0f, 0.07f, 20, 4, 0.5f};
  p->path_follow = {1, 1, 5.0f, 1.4f, 6};
  p->goal_angle = {1, 1, 3.0f, 0.5f};
  p->prefer_forward = {1, 1, 5.0f, 0.5f};
  // the other registered critics: initialize() defaults, not in the list (enabled 0)
  p->cost = {0, 0, 1, 3.81f, 300.0f, 1000000.0f, 0.5f};
  p->goal = {0, 1, 5.0f, 1.4f};
  p->constraint = {0, 1, 4.0f, 0.5f, 0.5f, -0.35f};
  p->twirling = {0, 1, 10.0f};
  p->path_angle = {0, 1, 2.0f, 4, 0.5f, 1.2f, 1, -0.35f};
  p->velocity_deadband = {0, 1, 35.0f, {0.0f, 0.0f, 0.0f}};
}

int smpc_abi_version(void) {return SMPC_ABI_VERSION;}

const char* smpc_build_info(void)
{
  return "libsmpc (MI355X-native sampling-MPC hot path), HIP gfx950, built " __DATE__ " " __TIME__;
}

const char* smpc_last_error(const smpc_ctx* ctx)
{
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int smpc_create(const smpc_config* cfg, smpc_ctx** out)
{
  if (!cfg || !out) return fail(nullptr, SMPC_ERR_INVALID, "null argument");
  *out = nullptr;
  if (cfg->batch_size == 0 || cfg->time_steps == 0 || cfg->iteration_count == 0)
    return fail(nullptr, SMPC_ERR_INVALID, "batch_size, time_steps, iteration_count must be > 0");
  if (cfg->motion_model > SMPC_MODEL_ACKERMANN)
    return fail(nullptr, SMPC_ERR_UNSUPPORTED, "motion_model must be Omni, DiffDrive or Ackermann");
  if (cfg->motion_model == SMPC_MODEL_ACKERMANN && !(cfg->ackermann_min_turning_r >= 0.f))
    return fail(nullptr, SMPC_ERR_INVALID, "ackermann_min_turning_r must be >= 0");
  if (cfg->time_steps > 64 * SMPC_MAX_R)
    return fail(nullptr, SMPC_ERR_UNSUPPORTED, "time_steps > 256");
  if (!(cfg->temperature > 0.f) || !(cfg->model_dt > 0.f))
    return fail(nullptr, SMPC_ERR_INVALID, "temperature and model_dt must be > 0");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, SMPC_ERR_DEVICE,
                std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count 0") +
                " (libsmpc has no CPU fallback)");
  smpc_ctx* c = new (std::nothrow) smpc_ctx();
  if (!c) return fail(nullptr, SMPC_ERR_NOMEM, "out of host memory");
  c->cfg = *cfg;
  smpc_critic_params_default(&c->critics);
  c->holonomic = cfg->motion_model == SMPC_MODEL_OMNI;
  c->acker_r = cfg->motion_model == SMPC_MODEL_ACKERMANN ? cfg->ackermann_min_turning_r : -1.f;
  c->c_vx_max = cfg->vx_max; c->c_vx_min = cfg->vx_min; c->c_vy = cfg->vy_max; c->c_wz = cfg->wz_max;
  int dev = cfg->device;
  if (dev < 0) {
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  }
  c->device = dev;
#define CK(call)                                                                        \
  do {                                                                                  \
    hipError_t e__ = (call);                                                            \
    if (e__ != hipSuccess) {                                                            \
      g_create_error = std::string(#call) + ": " + hipGetErrorString(e__);              \
      free_ctx(c);                                                                      \
      return SMPC_ERR_DEVICE;                                                           \
    }                                                                                   \
  } while (0)
  CK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  CK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;
  c->poll_enabled = getenv("SMPC_NO_POLL") == nullptr;
  if (getenv("SMPC_LANE_TIMELINE")) {
    CK(hipMalloc(&c->d_timeline, (8192 + kMaxGrid * 8) * sizeof(unsigned long long)));
    CK(hipMemset(c->d_timeline, 0, (8192 + kMaxGrid * 8) * sizeof(unsigned long long)));
  }
  CK(hipEventCreate(&c->ev0));
  CK(hipEventCreateWithFlags(&c->ev_map, hipEventDisableTiming));
  CK(hipEventCreate(&c->ev1));
  for (auto& e : c->evp) CK(hipEventCreate(&e));
  const uint32_t T = cfg->time_steps;
  c->R = T <= 64 ? 1 : (T <= 128 ? 2 : 4);
  const size_t n = static_cast<size_t>(cfg->batch_size) * T * sizeof(float);
  CK(hipMalloc(&c->d_nvx, n));
  CK(hipMalloc(&c->d_nvy, n));
  if (!c->holonomic) CK(hipMemset(c->d_nvy, 0, n));   // noises_vy_ (noise_generator.cpp:76-91)
  CK(hipMalloc(&c->d_nwz, n));
  {
    // which streaming pass: a wave per rollout (latency, small batches) or a lane per
    // rollout (throughput, large batches); SMPC_PASS=wave|lane overrides for experiments
    bool tpr = cfg->batch_size >= kLaneMinBatch && cfg->time_steps <= kLaneMaxT;
    if (cfg->flags & SMPC_FLAG_WAVE_PER_ROLLOUT) tpr = false;
    if (cfg->flags & SMPC_FLAG_LANE_PER_ROLLOUT) tpr = true;
    if (const char* e = getenv("SMPC_PASS")) {
      if (!strcmp(e, "wave")) tpr = false;
      if (!strcmp(e, "lane")) tpr = true;
    }
    if (cfg->flags & SMPC_FLAG_STORE_TRAJECTORIES) tpr = false;   // visualisation path: wave pass
    if (3ull * n >= (1ull << 32)) tpr = false;   // its buffer descriptor spans the three noise tensors
    c->use_tpr = tpr;
    if (tpr) {
      // back to back: the lane pass addresses the three through one buffer descriptor
      CK(hipMalloc(&c->d_tvx, 3 * n));
      c->d_tvy = c->d_tvx + n / sizeof(float);
      c->d_twz = c->d_tvy + n / sizeof(float);
      CK(smpc_lane_set_lds_limit(static_cast<int>(kLdsPerCu)));
    }
  }
  CK(hipMalloc(&c->d_costs[0], cfg->batch_size * sizeof(float)));
  CK(hipMalloc(&c->d_costs[1], cfg->batch_size * sizeof(float)));
  CK(hipMemset(c->d_costs[0], 0, cfg->batch_size * sizeof(float)));
  CK(hipMemset(c->d_costs[1], 0, cfg->batch_size * sizeof(float)));
  if (cfg->flags & SMPC_FLAG_STORE_TRAJECTORIES) {
    for (int i = 0; i < 3; ++i) {
      CK(hipMalloc(&c->d_traj[i], n));
      CK(hipMemset(c->d_traj[i], 0, n));
    }
  }
  c->tick_cap = tick_layout(T, SMPC_MAX_PATH).total;
  CK(hipMalloc(&c->d_tick, c->tick_cap));
  CK(hipHostMalloc(&c->h_tick, c->tick_cap, hipHostMallocDefault));
  CK(hipMalloc(&c->d_lut, 256 * sizeof(SmpcLut)));
  CK(hipMemset(c->d_lut, 0, 256 * sizeof(SmpcLut)));
  CK(hipMalloc(&c->d_lut_fp, 512 * sizeof(SmpcLut)));
  CK(hipMemset(c->d_lut_fp, 0, 512 * sizeof(SmpcLut)));
  CK(hipHostMalloc(&c->h_lut_fp, 512 * sizeof(SmpcLut), hipHostMallocDefault));
  CK(hipHostMalloc(&c->h_lut, 256 * sizeof(SmpcLut), hipHostMallocDefault));
  const size_t TL = 4 + 3 * static_cast<size_t>(T);
  CK(hipMalloc(&c->d_partials, kMaxGrid * TL * sizeof(float)));
  CK(hipMalloc(&c->d_tuple, TL * sizeof(float)));
  CK(hipMalloc(&c->d_out, (3 * T + 8) * sizeof(float)));
  CK(hipHostMalloc(&c->h_out, (3 * T + 8) * sizeof(float), hipHostMallocMapped));
  CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->h_out_dev), c->h_out, 0));
  CK(hipMalloc(&c->d_furthest, 16));
  CK(hipMemset(c->d_furthest, 0, 16));
  CK(smpc_set_pass_lds_limit(static_cast<int>(kLdsPerCu)));
#undef CK
  *out = c;
  return SMPC_OK;
}

void smpc_destroy(smpc_ctx* ctx) {free_ctx(ctx);}

int smpc_reset(smpc_ctx* c)
{
  if (!c) return SMPC_ERR_INVALID;
  HIPCK(c, hipSetDevice(c->device));
  // Optimizer::reset (optimizer.cpp:116-132): constraints back to base, costs zero,
  // NoiseGenerator::reset re-draws (noise_generator.cpp:76-95)
  c->c_vx_max = c->cfg.vx_max; c->c_vx_min = c->cfg.vx_min;
  c->c_vy = c->cfg.vy_max; c->c_wz = c->cfg.wz_max;
  HIPCK(c, hipMemsetAsync(c->d_costs[0], 0, c->cfg.batch_size * sizeof(float), c->stream));
  HIPCK(c, hipMemsetAsync(c->d_costs[1], 0, c->cfg.batch_size * sizeof(float), c->stream));
  c->tick_ready = false;
  c->hint_valid = false;
  if (c->rng_mode) {
    c->epoch++;
    return draw_noise(c);
  }
  HIPCK(c, hipStreamSynchronize(c->stream));
  return SMPC_OK;
}

int smpc_set_constraints(smpc_ctx* c, float vx_max, float vx_min, float vy_max, float wz_max)
{
  if (!c) return SMPC_ERR_INVALID;
  c->c_vx_max = vx_max; c->c_vx_min = vx_min; c->c_vy = vy_max; c->c_wz = wz_max;
  return SMPC_OK;
}

int smpc_set_footprint(smpc_ctx* c, const double* xy, uint32_t n_points, double circumscribed_radius,
                       double layer_cost_scaling_factor)
{
  if (!c || (n_points && !xy)) return fail(c, SMPC_ERR_INVALID, "null argument");
  if (n_points > SMPC_MAX_FOOTPRINT) return fail(c, SMPC_ERR_UNSUPPORTED, "footprint with more than 16 points");
  c->fp_x.clear();
  c->fp_y.clear();
  for (uint32_t i = 0; i < n_points; ++i) {
    c->fp_x.push_back(xy[2 * i]);
    c->fp_y.push_back(xy[2 * i + 1]);
  }
  c->fp_circumscribed_radius = circumscribed_radius;
  c->fp_layer_scale = layer_cost_scaling_factor;
  c->critics_version++;
  return SMPC_OK;
}

int smpc_set_critics(smpc_ctx* c, const smpc_critic_params* p)
{
  if (!c || !p) return SMPC_ERR_INVALID;
  c->critics = *p;
  c->critics_version++;
  return SMPC_OK;
}

int smpc_set_costmap(smpc_ctx* c, const uint8_t* cells, uint32_t width, uint32_t height,
                     double origin_x, double origin_y, double resolution, int track_unknown,
                     float inscribed_radius, float cost_scaling_factor, float inflation_radius)
{
  if (!c) return SMPC_ERR_INVALID;
  if (!cells || width == 0 || height == 0 || !(resolution > 0.0))
    return fail(c, SMPC_ERR_INVALID, "bad costmap");
  HIPCK(c, hipSetDevice(c->device));
  int rc = wait_map_upload(c);   // the previous upload reads the mirror this call overwrites
  if (rc != SMPC_OK) return rc;
  HostCostmap& m = c->map;
  const size_t bytes = static_cast<size_t>(width) * height;
  bool same_size = m.set && m.W == width && m.H == height;
  if (bytes > c->d_map_bytes) {
    // the stream may still read the old device copy (an un-waited tick): drain it first
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (c->d_map) HIPCK(c, hipFree(c->d_map));
    c->d_map = nullptr;
    c->d_map_bytes = 0;
    HIPCK(c, hipMalloc(&c->d_map, (bytes + 255) / 256 * 256));
    c->d_map_bytes = bytes;
    same_size = false;
  }
  if (bytes > m.cap) {
    if (m.cells) HIPCK(c, hipHostFree(m.cells));
    m.cells = nullptr;
    m.cap = 0;
    HIPCK(c, hipHostMalloc(reinterpret_cast<void**>(&m.cells), (bytes + 4095) / 4096 * 4096, hipHostMallocDefault));
    m.cap = bytes;
    same_size = false;
  }
  // The controller hands over the whole costmap every tick (controller.cpp:99-103) while the
  // costmap itself changes at its own, lower update rate: only the band of rows that differ
  // from the mirror is copied and uploaded (nothing at all for an unchanged map).
  uint32_t y0 = height, y1 = 0;
  if (same_size) {
    for (uint32_t y = 0; y < height; ++y) {
      const size_t o = static_cast<size_t>(y) * width;
      if (memcmp(m.cells + o, cells + o, width) != 0) {
        memcpy(m.cells + o, cells + o, width);
        if (y < y0) y0 = y;
        y1 = y;
      }
    }
  } else {
    memcpy(m.cells, cells, bytes);
    y0 = 0;
    y1 = height - 1;
  }
  // the lookup tables depend on these, not on the cells
  const bool same_params = m.set && m.res == resolution && m.track_unknown == (track_unknown != 0) &&
    m.inscribed_radius == inscribed_radius && m.cost_scaling_factor == cost_scaling_factor &&
    m.inflation_radius == inflation_radius;
  m.W = width; m.H = height; m.ox = origin_x; m.oy = origin_y; m.res = resolution;
  m.track_unknown = track_unknown != 0;
  m.inscribed_radius = inscribed_radius;
  m.cost_scaling_factor = cost_scaling_factor;
  m.inflation_radius = inflation_radius;
  m.set = true;
  if (!same_params) c->map_version++;
  c->map_bytes_last = 0;
  if (y0 <= y1) {
    const size_t o = static_cast<size_t>(y0) * width, n = static_cast<size_t>(y1 - y0 + 1) * width;
    HIPCK(c, hipMemcpyAsync(c->d_map + o, m.cells + o, n, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipEventRecord(c->ev_map, c->stream));
    c->map_pending = true;
    c->map_bytes_last = n;
    c->map_bytes_total += n;
  }
  return SMPC_OK;
}

int smpc_update_costmap_region(smpc_ctx* c, const uint8_t* cells, uint32_t row_stride, uint32_t x0,
                               uint32_t y0, uint32_t width, uint32_t height)
{
  if (!c || !cells) return SMPC_ERR_INVALID;
  HostCostmap& m = c->map;
  if (!m.set) return fail(c, SMPC_ERR_STATE, "smpc_set_costmap first");
  if (width == 0 || height == 0 || row_stride < width || x0 >= m.W || y0 >= m.H || width > m.W - x0 ||
      height > m.H - y0)
    return fail(c, SMPC_ERR_INVALID, "region outside the costmap");
  HIPCK(c, hipSetDevice(c->device));
  int rc = wait_map_upload(c);
  if (rc != SMPC_OK) return rc;
  for (uint32_t y = 0; y < height; ++y)
    memcpy(m.cells + static_cast<size_t>(y0 + y) * m.W + x0, cells + static_cast<size_t>(y) * row_stride, width);
  const size_t o = static_cast<size_t>(y0) * m.W + x0;
  HIPCK(c, hipMemcpy2DAsync(c->d_map + o, m.W, m.cells + o, m.W, width, height, hipMemcpyHostToDevice,
                            c->stream));
  HIPCK(c, hipEventRecord(c->ev_map, c->stream));
  c->map_pending = true;
  c->map_bytes_last = static_cast<uint64_t>(width) * height;
  c->map_bytes_total += c->map_bytes_last;
  return SMPC_OK;
}

int smpc_costmap_upload_bytes(const smpc_ctx* c, uint64_t* last_call, uint64_t* total)
{
  if (!c) return SMPC_ERR_INVALID;
  if (last_call) *last_call = c->map_bytes_last;
  if (total) *total = c->map_bytes_total;
  return SMPC_OK;
}

int smpc_set_noise(smpc_ctx* c, const float* nvx, const float* nvy, const float* nwz)
{
  if (!c || !nvx || !nvy || !nwz) return SMPC_ERR_INVALID;
  HIPCK(c, hipSetDevice(c->device));
  const size_t n = static_cast<size_t>(c->cfg.batch_size) * c->cfg.time_steps * sizeof(float);
  HIPCK(c, hipMemcpyAsync(c->d_nvx, nvx, n, hipMemcpyHostToDevice, c->stream));
  if (c->holonomic) HIPCK(c, hipMemcpyAsync(c->d_nvy, nvy, n, hipMemcpyHostToDevice, c->stream));
  HIPCK(c, hipMemcpyAsync(c->d_nwz, nwz, n, hipMemcpyHostToDevice, c->stream));
  {
    int rc = update_time_major(c);
    if (rc != SMPC_OK) return rc;
  }
  HIPCK(c, hipStreamSynchronize(c->stream));
  c->have_noise = true;
  c->rng_mode = false;
  return SMPC_OK;
}

int smpc_seed(smpc_ctx* c, uint64_t seed)
{
  if (!c) return SMPC_ERR_INVALID;
  HIPCK(c, hipSetDevice(c->device));
  c->seed = seed;
  c->epoch = 0;
  c->rng_mode = true;
  return draw_noise(c);
}

int smpc_redraw_noise(smpc_ctx* c)
{
  if (!c) return SMPC_ERR_INVALID;
  if (!c->rng_mode) return fail(c, SMPC_ERR_STATE, "smpc_redraw_noise needs device-RNG mode (smpc_seed)");
  HIPCK(c, hipSetDevice(c->device));
  c->epoch++;
  return draw_noise(c);
}

int smpc_get_noise(smpc_ctx* c, float* nvx, float* nvy, float* nwz)
{
  if (!c) return SMPC_ERR_INVALID;
  if (!c->have_noise) return fail(c, SMPC_ERR_STATE, "no noise yet");
  HIPCK(c, hipSetDevice(c->device));
  const size_t n = static_cast<size_t>(c->cfg.batch_size) * c->cfg.time_steps * sizeof(float);
  HIPCK(c, hipStreamSynchronize(c->stream));
  if (nvx) HIPCK(c, hipMemcpy(nvx, c->d_nvx, n, hipMemcpyDeviceToHost));
  if (nvy) HIPCK(c, hipMemcpy(nvy, c->d_nvy, n, hipMemcpyDeviceToHost));
  if (nwz) HIPCK(c, hipMemcpy(nwz, c->d_nwz, n, hipMemcpyDeviceToHost));
  return SMPC_OK;
}

int smpc_optimize(smpc_ctx* c, const smpc_tick_in* in, float* u_inout, smpc_tick_out* out)
{
  if (!c || !in || !u_inout) return fail(c, SMPC_ERR_INVALID, "null argument");
  HIPCK(c, hipSetDevice(c->device));
  int rc = prepare_tick(c, in, u_inout);
  if (rc != SMPC_OK) return rc;
  const uint32_t T = c->cfg.time_steps;
  const uint32_t iS = 3 * T + 2, iNC = 3 * T + 3, iSused = 3 * T + 4;
  c->passes = 0;
  c->evp_used = 0;
  c->costs_cur = 0;
  bool fail_sticky = c->fail_in;
  bool fail_flag = fail_sticky;
  bool fetched = false;
  // furthest_reached_path_point of this tick: evaluated once, then cached
  // (setPathFurthestPointIfNotSet, utils.hpp:350-355)
  bool S_known = false, S_on_device = false;
  uint32_t S_host = 0;
  const bool speculate = !(c->cfg.flags & SMPC_FLAG_NO_SPECULATION);
  for (uint32_t it = 0; it < c->cfg.iteration_count; ++it) {
    uint32_t flags = scoring_flags(c, fail_sticky);
    if (it > 0) flags |= SD_ACCUMULATE;
    const float* u_dev = it == 0 ? nullptr : c->d_out;
    c->costs_cur = it & 1;
    const float* dF = nullptr;
    uint32_t hintS = 0;
    bool spec_try = false;
    if (flags & SD_NEED_FURTHEST) {
      if (S_known) {
        hintS = S_host;
      } else if (S_on_device) {
        dF = c->d_furthest;
      } else if (speculate && c->hint_valid) {
        // score with the previous tick's furthest point; the pass reports the true
        // one and a miss is re-scored below, so the result is exact either way
        spec_try = true;
        hintS = c->hint;
        flags |= SD_LOCAL_FURTHEST;
      } else {
        rc = launch_furthest(c, c->d_furthest);
        if (rc != SMPC_OK) return rc;
        S_on_device = true;
        dF = c->d_furthest;
      }
    }
    rc = launch_score(c, flags, u_dev, dF, hintS, c->d_tuple, true, dF);
    if (rc != SMPC_OK) return rc;
    if (c->cfg.flags & SMPC_FLAG_PROFILE) HIPCK(c, hipEventRecord(c->ev1, c->stream));
    fetched = false;
    const bool last = it + 1 == c->cfg.iteration_count;
    if ((flags & (SD_OBSTACLES | SD_COST)) || spec_try || last) {
      // one host round trip: it carries fail_flag (obstacles_critic.cpp:177), the true
      // furthest point, and on the last iteration the result itself
      rc = fetch_out(c);
      if (rc != SMPC_OK) return rc;
      fetched = true;
    }
    if (fetched && dF) {
      S_host = static_cast<uint32_t>(c->h_out[iSused]);
      S_known = true;
    }
    if (spec_try) {
      const uint32_t S_true = static_cast<uint32_t>(c->h_out[iS]);
      S_host = S_true;
      S_known = true;
      if (S_true != hintS) {
        c->spec_misses++;
        flags &= ~SD_LOCAL_FURTHEST;
        rc = launch_score(c, flags, u_dev, nullptr, S_host, c->d_tuple, true, nullptr);
        if (rc != SMPC_OK) return rc;
        if (c->cfg.flags & SMPC_FLAG_PROFILE) HIPCK(c, hipEventRecord(c->ev1, c->stream));
        rc = fetch_out(c);
        if (rc != SMPC_OK) return rc;
      }
    }
    if ((flags & (SD_OBSTACLES | SD_COST)) && c->h_out[iNC] == 0.0f) {
      // every rollout collides: the critics after Obstacles were not scored in the
      // reference (critic_manager.cpp:70-73); redo this iteration with Obstacles only
      // so that costs and u match it exactly
      fail_flag = true;
      fail_sticky = true;
      const uint32_t only = fail_only_flags(c) |
        (it > 0 ? SD_ACCUMULATE : 0u);
      rc = launch_score(c, only, u_dev, nullptr, 0, c->d_tuple, true, nullptr);
      if (rc != SMPC_OK) return rc;
      if (c->cfg.flags & SMPC_FLAG_PROFILE) HIPCK(c, hipEventRecord(c->ev1, c->stream));
      fetched = false;
    }
  }
  if (!fetched) {
    rc = fetch_out(c);
    if (rc != SMPC_OK) return rc;
  }
  if (S_known) {
    c->hint = S_host;
    c->hint_valid = true;
  }
  store_control_sequence(c, u_inout);
  if (out) {
    memset(out, 0, sizeof(*out));
    out->fail_flag = fail_flag ? 1 : 0;
    out->furthest_valid = S_known ? 1 : 0;
    out->furthest_reached_path_point = S_known ? S_host : 0;
    out->non_colliding = static_cast<uint32_t>(c->h_out[iNC]);
    out->min_cost = c->h_out[3 * T + 0];
    out->sum_w = c->h_out[3 * T + 1];
    out->passes = c->passes;
    float ms = 0.f;
    if (c->cfg.flags & SMPC_FLAG_PROFILE) {
      // the polled flag can beat the event's completion signal by a few microseconds
      HIPCK(c, hipEventSynchronize(c->ev1));
      if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) out->device_ms = ms;
    }
    out->score_pass_ms = profile_pass_ms(c);
    out->pass_kind = c->last_pass_kind;
  }
  return SMPC_OK;
}

int smpc_get_trajectories(smpc_ctx* c, float* x, float* y, float* yaws)
{
  if (!c) return SMPC_ERR_INVALID;
  if (!(c->cfg.flags & SMPC_FLAG_STORE_TRAJECTORIES))
    return fail(c, SMPC_ERR_STATE, "ctx was created without SMPC_FLAG_STORE_TRAJECTORIES");
  HIPCK(c, hipSetDevice(c->device));
  HIPCK(c, hipStreamSynchronize(c->stream));
  const size_t n = static_cast<size_t>(c->cfg.batch_size) * c->cfg.time_steps * sizeof(float);
  if (x) HIPCK(c, hipMemcpy(x, c->d_traj[0], n, hipMemcpyDeviceToHost));
  if (y) HIPCK(c, hipMemcpy(y, c->d_traj[1], n, hipMemcpyDeviceToHost));
  if (yaws) HIPCK(c, hipMemcpy(yaws, c->d_traj[2], n, hipMemcpyDeviceToHost));
  return SMPC_OK;
}

int smpc_get_costs(smpc_ctx* c, float* costs)
{
  if (!c || !costs) return SMPC_ERR_INVALID;
  HIPCK(c, hipSetDevice(c->device));
  HIPCK(c, hipStreamSynchronize(c->stream));
  HIPCK(c, hipMemcpy(costs, c->d_costs[c->costs_cur], c->cfg.batch_size * sizeof(float),
                     hipMemcpyDeviceToHost));
  return SMPC_OK;
}

int smpc_selftest_sincos(smpc_ctx* c, const float* x, uint32_t n, float* sin_out, float* cos_out)
{
  if (!c || !x || !sin_out || !cos_out || n == 0) return fail(c, SMPC_ERR_INVALID, "null argument");
  HIPCK(c, hipSetDevice(c->device));
  float *dx = nullptr, *ds = nullptr, *dc = nullptr;
  HIPCK(c, hipMalloc(&dx, n * sizeof(float)));
  HIPCK(c, hipMalloc(&ds, n * sizeof(float)));
  HIPCK(c, hipMalloc(&dc, n * sizeof(float)));
  HIPCK(c, hipMemcpyAsync(dx, x, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIPCK(c, smpc_launch_sincos(dx, n, ds, dc, c->stream));
  HIPCK(c, hipMemcpyAsync(sin_out, ds, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCK(c, hipMemcpyAsync(cos_out, dc, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(dx); (void)hipFree(ds); (void)hipFree(dc);
  return SMPC_OK;
}

int smpc_selftest_lane_reduce(smpc_ctx* c, const float* v, const float* w, float* out)
{
  if (!c || !v || !w || !out) return fail(c, SMPC_ERR_INVALID, "null argument");
  HIPCK(c, hipSetDevice(c->device));
  float *dv = nullptr, *dw = nullptr, *dout = nullptr;
  HIPCK(c, hipMalloc(&dv, 64 * 64 * sizeof(float)));
  HIPCK(c, hipMalloc(&dw, 64 * sizeof(float)));
  HIPCK(c, hipMalloc(&dout, 64 * sizeof(float)));
  HIPCK(c, hipMemcpyAsync(dv, v, 64 * 64 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIPCK(c, hipMemcpyAsync(dw, w, 64 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIPCK(c, smpc_launch_lane_reduce(dv, dw, dout, c->stream));
  HIPCK(c, hipMemcpyAsync(out, dout, 64 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(dv); (void)hipFree(dw); (void)hipFree(dout);
  return SMPC_OK;
}

// ---- batch-sharded path -------------------------------------------------------

int smpc_set_stream(smpc_ctx* c, void* hip_stream)
{
  if (!c) return SMPC_ERR_INVALID;
  const int rc = wait_map_upload(c);   // it went out on the stream being left
  if (rc != SMPC_OK) return rc;
  c->stream = hip_stream == SMPC_STREAM_OWN ? c->own_stream : static_cast<hipStream_t>(hip_stream);
  return SMPC_OK;
}

// developer aid: stage durations of the last lane-pass launch, from the stamps of wave 0 of
// every block (shader clocks; stages: entry -> LDS staged -> constants -> group 1 -> group 2
// -> all waves at the final barrier -> partial written); out[7][3] = min, median, max
int smpc_debug_lane_timeline(smpc_ctx* c, double* out, uint32_t* n_blocks)
{
  if (!c || !out || !c->d_timeline) return SMPC_ERR_INVALID;
  HIPCK(c, hipSetDevice(c->device));
  HIPCK(c, hipStreamSynchronize(c->stream));
  const uint32_t nb = c->grid_tpr;
  std::vector<unsigned long long> h(static_cast<size_t>(nb) * 8);
  HIPCK(c, hipMemcpy(h.data(), c->d_timeline, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull;
  for (uint32_t b = 0; b < nb; ++b) t0 = std::min(t0, h[b * 8]);
  for (int k = 0; k < 7; ++k) {
    std::vector<double> v;
    for (uint32_t b = 0; b < nb; ++b)
      if (h[b * 8 + k]) v.push_back(static_cast<double>(h[b * 8 + k] - (k ? h[b * 8 + k - 1] : t0)));
    std::sort(v.begin(), v.end());
    out[3 * k] = v.empty() ? 0 : v.front();
    out[3 * k + 1] = v.empty() ? 0 : v[v.size() / 2];
    out[3 * k + 2] = v.empty() ? 0 : v.back();
  }
  // out[21..28]: per wave of the block, median of (end of its groups - stamp 2 of wave 0)
  std::vector<unsigned long long> hw(static_cast<size_t>(nb) * 8);
  HIPCK(c, hipMemcpy(hw.data(), c->d_timeline + 8192, hw.size() * sizeof(hw[0]), hipMemcpyDeviceToHost));
  for (int w = 0; w < 8; ++w) {
    std::vector<double> v;
    for (uint32_t b = 0; b < nb; ++b)
      if (hw[b * 8 + w] && h[b * 8 + 2]) v.push_back(static_cast<double>(hw[b * 8 + w] - h[b * 8 + 2]));
    std::sort(v.begin(), v.end());
    out[21 + w] = v.empty() ? 0 : v[v.size() / 2];
  }
  if (n_blocks) *n_blocks = nb;
  return SMPC_OK;
}

int smpc_set_profile(smpc_ctx* c, int enable)
{
  if (!c) return SMPC_ERR_INVALID;
  if (enable) c->cfg.flags |= SMPC_FLAG_PROFILE;
  else c->cfg.flags &= ~static_cast<uint32_t>(SMPC_FLAG_PROFILE);
  return SMPC_OK;
}

uint32_t smpc_tuple_len(const smpc_ctx* c)
{
  return c ? SMPC_TUPLE_HEADER + 3 * c->cfg.time_steps : 0;
}

int smpc_shard_begin(smpc_ctx* c, const smpc_tick_in* in, const float* u_in)
{
  if (!c || !in || !u_in) return fail(c, SMPC_ERR_INVALID, "null argument");
  HIPCK(c, hipSetDevice(c->device));
  c->passes = 0;
  c->evp_used = 0;
  c->costs_cur = 0;
  return prepare_tick(c, in, u_in);
}

int smpc_shard_furthest(smpc_ctx* c, float* d_furthest)
{
  if (!c || !d_furthest) return fail(c, SMPC_ERR_INVALID, "null argument");
  if (!c->tick_ready) return fail(c, SMPC_ERR_STATE, "smpc_shard_begin first");
  HIPCK(c, hipSetDevice(c->device));
  return launch_furthest(c, d_furthest);
}

int smpc_shard_score(smpc_ctx* c, const float* d_furthest, uint32_t furthest_hint, float* d_tuple)
{
  if (!c || !d_tuple) return fail(c, SMPC_ERR_INVALID, "null argument");
  if (!c->tick_ready) return fail(c, SMPC_ERR_STATE, "smpc_shard_begin first");
  HIPCK(c, hipSetDevice(c->device));
  // fail_flag is batch-wide: a shard never short-circuits on its own rollouts
  uint32_t flags = scoring_flags(c, c->fail_in);
  if (flags & SD_NEED_FURTHEST) flags |= SD_LOCAL_FURTHEST;  // the tuple carries the true local value
  return launch_score(c, flags, nullptr, d_furthest, furthest_hint, d_tuple);
}

int smpc_shard_rescore_failed(smpc_ctx* c, float* d_tuple)
{
  if (!c || !d_tuple) return fail(c, SMPC_ERR_INVALID, "null argument");
  if (!c->tick_ready) return fail(c, SMPC_ERR_STATE, "smpc_shard_begin first");
  HIPCK(c, hipSetDevice(c->device));
  return launch_score(c, fail_only_flags(c), nullptr, nullptr, 0, d_tuple);
}

int smpc_shard_combine(smpc_ctx* c, const float* d_tuples, uint32_t n_tuples, float* u_out,
                       smpc_tick_out* out)
{
  if (!c || !d_tuples || n_tuples == 0 || !u_out) return fail(c, SMPC_ERR_INVALID, "null argument");
  HIPCK(c, hipSetDevice(c->device));
  int rc = launch_combine(c, d_tuples, n_tuples, nullptr);
  if (rc != SMPC_OK) return rc;
  rc = fetch_out(c);
  if (rc != SMPC_OK) return rc;
  const uint32_t T = c->cfg.time_steps;
  memcpy(u_out, c->h_out, 3 * T * sizeof(float));
  if (out) {
    memset(out, 0, sizeof(*out));
    const bool obstacles_scored = (scoring_flags(c, c->fail_in) & (SD_OBSTACLES | SD_COST)) != 0;
    out->fail_flag = (c->fail_in || (obstacles_scored && c->h_out[3 * T + 3] == 0.0f)) ? 1 : 0;
    out->furthest_valid = (c->gate_flags & SD_NEED_FURTHEST) ? 1 : 0;
    out->furthest_reached_path_point = static_cast<uint32_t>(c->h_out[3 * T + 2]);
    out->non_colliding = static_cast<uint32_t>(c->h_out[3 * T + 3]);
    out->min_cost = c->h_out[3 * T + 0];
    out->sum_w = c->h_out[3 * T + 1];
    out->passes = c->passes;
    out->score_pass_ms = profile_pass_ms(c);
    out->pass_kind = c->last_pass_kind;
  }
  return SMPC_OK;
}

int smpc_shard_comm_id(void* id_out, uint32_t id_bytes)
{
  if (!id_out || id_bytes < sizeof(ncclUniqueId)) return fail(nullptr, SMPC_ERR_INVALID, "id buffer too small");
  const RcclApi* r = rccl();
  if (!r) return fail(nullptr, SMPC_ERR_UNSUPPORTED, "RCCL (librccl.so) could not be loaded");
  ncclUniqueId id;
  const ncclResult_t e = r->GetUniqueId(&id);
  if (e != ncclSuccess) return fail(nullptr, SMPC_ERR_DEVICE, "ncclGetUniqueId failed");
  memcpy(id_out, &id, sizeof(id));
  return SMPC_OK;
}

int smpc_shard_comm_init(smpc_ctx* c, const void* id_in, int rank, int world)
{
  if (!c || !id_in || world < 1 || rank < 0 || rank >= world) return fail(c, SMPC_ERR_INVALID, "bad rank/world");
  const RcclApi* r = rccl();
  if (!r) return fail(c, SMPC_ERR_UNSUPPORTED, "RCCL (librccl.so) could not be loaded");
  HIPCK(c, hipSetDevice(c->device));
  if (c->comm) {
    (void)r->CommDestroy(c->comm);
    c->comm = nullptr;
  }
  ncclUniqueId id;
  memcpy(&id, id_in, sizeof(id));
  const ncclResult_t e = r->CommInitRank(&c->comm, world, id, rank);
  if (e != ncclSuccess) {
    c->comm = nullptr;
    return fail(c, SMPC_ERR_DEVICE, std::string("ncclCommInitRank: ") +
                                   (r->GetErrorString ? r->GetErrorString(e) : "error"));
  }
  c->comm_rank = rank;
  c->comm_world = world;
  if (c->d_all) (void)hipFree(c->d_all);
  c->d_all = nullptr;
  HIPCK(c, hipMalloc(&c->d_all, static_cast<size_t>(world) * (4 + 3 * c->cfg.time_steps) * sizeof(float)));
  return SMPC_OK;
}

// One batch-sharded tick, exchanges included: the protocol of
// mpcholonavigation_amd/sharded.py (ShardedOptimizer.optimize) with ncclAllGather /
// ncclAllReduce enqueued on the ctx's stream between the kernels — one call, no host
// round trip except the final wait (and one more after a speculation miss).
int smpc_shard_tick(smpc_ctx* c, const smpc_tick_in* in, float* u_inout, smpc_tick_out* out,
                    int speculate)
{
  if (!c || !in || !u_inout) return fail(c, SMPC_ERR_INVALID, "null argument");
  if (!c->comm) return fail(c, SMPC_ERR_STATE, "smpc_shard_comm_init first");
  const RcclApi* r = rccl();
  HIPCK(c, hipSetDevice(c->device));
  c->passes = 0;
  c->evp_used = 0;
  c->costs_cur = 0;
  int rc = prepare_tick(c, in, u_inout);
  if (rc != SMPC_OK) return rc;
  const uint32_t T = c->cfg.time_steps, TL = 4 + 3 * T, G = static_cast<uint32_t>(c->comm_world);
  auto nccl_ok = [&](ncclResult_t e, const char* what) {
    if (e == ncclSuccess) return SMPC_OK;
    return fail(c, SMPC_ERR_DEVICE, std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(e) : "error"));
  };
  auto gather_combine_fetch = [&](const float* d_used) -> int {
    int e = nccl_ok(r->AllGather(c->d_tuple, c->d_all, TL, ncclFloat32, c->comm, c->stream), "ncclAllGather");
    if (e != SMPC_OK) return e;
    e = launch_combine(c, c->d_all, G, d_used);
    if (e != SMPC_OK) return e;
    return fetch_out(c);
  };
  // fail_flag is batch-wide: a shard never short-circuits on its own rollouts
  uint32_t flags = scoring_flags(c, c->fail_in);
  const bool need_f = (flags & SD_NEED_FURTHEST) != 0;
  if (need_f) flags |= SD_LOCAL_FURTHEST;   // the tuple carries the true local value
  if (speculate && need_f && c->hint_valid) {
    rc = launch_score(c, flags, nullptr, nullptr, c->hint, c->d_tuple);
    if (rc == SMPC_OK) rc = gather_combine_fetch(nullptr);
    if (rc != SMPC_OK) return rc;
    const uint32_t S_true = static_cast<uint32_t>(c->h_out[3 * T + 2]);
    if (S_true != c->hint) {
      // miss: the gathered tuples carry the true batch-wide furthest point
      c->spec_misses++;
      c->hint = S_true;
      rc = launch_score(c, flags, nullptr, nullptr, S_true, c->d_tuple);
      if (rc == SMPC_OK) rc = gather_combine_fetch(nullptr);
      if (rc != SMPC_OK) return rc;
    }
  } else {
    if (need_f) {
      rc = launch_furthest(c, c->d_furthest);
      if (rc != SMPC_OK) return rc;
      rc = nccl_ok(r->AllReduce(c->d_furthest, c->d_furthest, 1, ncclFloat32, ncclMax, c->comm, c->stream),
                   "ncclAllReduce");
      if (rc != SMPC_OK) return rc;
    }
    rc = launch_score(c, flags, nullptr, need_f ? c->d_furthest : nullptr, 0, c->d_tuple);
    if (rc == SMPC_OK) rc = gather_combine_fetch(need_f ? c->d_furthest : nullptr);
    if (rc != SMPC_OK) return rc;
    if (need_f) {
      c->hint = static_cast<uint32_t>(c->h_out[3 * T + 2]);
      c->hint_valid = true;
    }
  }
  const bool obstacles_scored = (flags & (SD_OBSTACLES | SD_COST)) != 0;
  bool failed = c->fail_in;
  if (!c->fail_in && obstacles_scored && c->h_out[3 * T + 3] == 0.0f) {
    // all rollouts of the WHOLE batch collide: the reference scored nothing past Obstacles
    // (critic_manager.cpp:70-73)
    failed = true;
    rc = launch_score(c, fail_only_flags(c), nullptr, nullptr, 0, c->d_tuple);
    if (rc == SMPC_OK) rc = gather_combine_fetch(nullptr);
    if (rc != SMPC_OK) return rc;
  }
  store_control_sequence(c, u_inout);
  if (out) {
    memset(out, 0, sizeof(*out));
    out->fail_flag = failed ? 1 : 0;
    out->furthest_valid = need_f ? 1 : 0;
    out->furthest_reached_path_point = need_f ? c->hint : 0;
    out->non_colliding = static_cast<uint32_t>(c->h_out[3 * T + 3]);
    out->min_cost = c->h_out[3 * T + 0];
    out->sum_w = c->h_out[3 * T + 1];
    out->passes = c->passes;
    out->score_pass_ms = profile_pass_ms(c);
    out->pass_kind = c->last_pass_kind;
  }
  return SMPC_OK;
}

// ---- several planning instances per launch (BASELINE configs[4]: multi-robot fleets) --------

struct smpc_group {
  std::vector<smpc_ctx*> ctxs;
  std::vector<hipStream_t> saved_stream;
  std::vector<uint8_t*> saved_h_tick, saved_d_tick;
  uint8_t* h_all = nullptr;   // pinned: n tick blocks, then SmpcDev[n], SmpcReduceArgs[n]
  uint8_t* d_all = nullptr;
  size_t slot = 0, off_dev = 0, off_red = 0, total = 0;
  hipStream_t stream = nullptr;
  uint64_t batched_ticks = 0, single_ticks = 0;
};

int smpc_group_create(smpc_ctx* const* ctxs, uint32_t n, smpc_group** out)
{
  if (!ctxs || !n || !out) return fail(nullptr, SMPC_ERR_INVALID, "null argument");
  *out = nullptr;
  for (uint32_t i = 0; i < n; ++i) {
    if (!ctxs[i]) return fail(nullptr, SMPC_ERR_INVALID, "null ctx");
    if (ctxs[i]->device != ctxs[0]->device || ctxs[i]->cfg.time_steps != ctxs[0]->cfg.time_steps)
      return fail(nullptr, SMPC_ERR_INVALID, "a group's contexts share the device and time_steps");
    if (ctxs[i]->defer_upload) return fail(nullptr, SMPC_ERR_STATE, "ctx already in a group");
  }
  smpc_group* g = new (std::nothrow) smpc_group();
  if (!g) return fail(nullptr, SMPC_ERR_NOMEM, "out of memory");
  smpc_ctx* c0 = ctxs[0];
  if (hipSetDevice(c0->device) != hipSuccess) {
    delete g;
    return fail(nullptr, SMPC_ERR_DEVICE, "hipSetDevice");
  }
  g->slot = align_up(static_cast<uint32_t>(c0->tick_cap), 256);
  g->off_dev = g->slot * n;
  g->off_red = g->off_dev + align_up(static_cast<uint32_t>(sizeof(SmpcDev)) * n, 256);
  g->total = g->off_red + align_up(static_cast<uint32_t>(sizeof(SmpcReduceArgs)) * n, 256);
  if (hipHostMalloc(&g->h_all, g->total, hipHostMallocDefault) != hipSuccess ||
    hipMalloc(&g->d_all, g->total) != hipSuccess)
  {
    if (g->h_all) (void)hipHostFree(g->h_all);
    delete g;
    return fail(nullptr, SMPC_ERR_NOMEM, "group buffers");
  }
  g->stream = c0->own_stream;
  for (uint32_t i = 0; i < n; ++i) {
    smpc_ctx* c = ctxs[i];
    (void)hipStreamSynchronize(c->stream);
    g->ctxs.push_back(c);
    g->saved_stream.push_back(c->stream);
    g->saved_h_tick.push_back(c->h_tick);
    g->saved_d_tick.push_back(c->d_tick);
    c->stream = g->stream;
    c->h_tick = g->h_all + g->slot * i;
    c->d_tick = g->d_all + g->slot * i;
    c->defer_upload = true;
    c->lut_valid = false;
  }
  *out = g;
  return SMPC_OK;
}

void smpc_group_destroy(smpc_group* g)
{
  if (!g) return;
  (void)hipStreamSynchronize(g->stream);
  for (size_t i = 0; i < g->ctxs.size(); ++i) {
    smpc_ctx* c = g->ctxs[i];
    c->stream = g->saved_stream[i];
    c->h_tick = g->saved_h_tick[i];
    c->d_tick = g->saved_d_tick[i];
    c->defer_upload = false;
  }
  if (g->h_all) (void)hipHostFree(g->h_all);
  if (g->d_all) (void)hipFree(g->d_all);
  delete g;
}

// One tick of every member.  When every member can take the lane-per-rollout pass with a
// speculated furthest point (the steady state), the group issues ONE upload, ONE scoring launch
// (blockIdx.y = member) and ONE reduction launch; a member that misses its speculation, collides
// everywhere, or is not eligible is ticked on its own with smpc_optimize — results are those of
// smpc_optimize in every case.
int smpc_group_optimize(smpc_group* g, const smpc_tick_in* ins, float* const* u_inout,
                        smpc_tick_out* outs)
{
  if (!g || !ins || !u_inout) return fail(nullptr, SMPC_ERR_INVALID, "null argument");
  const uint32_t n = static_cast<uint32_t>(g->ctxs.size());
  smpc_ctx* c0 = g->ctxs[0];
  HIPCK(c0, hipSetDevice(c0->device));
  auto single = [&](uint32_t i) -> int {
    smpc_ctx* c = g->ctxs[i];
    c->defer_upload = false;   // its own upload, into its slot of the group's buffers
    const int rc = smpc_optimize(c, &ins[i], u_inout[i], outs ? &outs[i] : nullptr);
    c->defer_upload = true;
    g->single_ticks++;
    return rc;
  };
  static const bool timing = getenv("SMPC_GROUP_TIMING") != nullptr;
  auto now = [] {return std::chrono::steady_clock::now();};
  auto us_between = [](auto a, auto b) {return std::chrono::duration<double, std::micro>(b - a).count();};
  const auto t_start = now();
  // ---- prepare every member; decide whether the batched launch applies -----------------
  bool batched = true;
  uint32_t Pmax = 0, window_bytes = 0, gridx = 0;
  bool obst = false;
  std::vector<uint32_t> flags(n);
  for (uint32_t i = 0; i < n; ++i) {
    smpc_ctx* c = g->ctxs[i];
    if (!u_inout[i]) return fail(c, SMPC_ERR_INVALID, "null control sequence");
    c->passes = 0;
    c->evp_used = 0;
    c->costs_cur = 0;
    int rc = prepare_tick(c, &ins[i], u_inout[i]);
    if (rc != SMPC_OK) return rc;
    flags[i] = scoring_flags(c, c->fail_in);
    const bool need_f = (flags[i] & SD_NEED_FURTHEST) != 0;
    if (need_f) flags[i] |= SD_LOCAL_FURTHEST;
    const bool ok = c->lane_now && c->cfg.iteration_count == 1 && !c->fail_in &&
      !(c->cfg.flags & (SMPC_FLAG_NO_SPECULATION | SMPC_FLAG_PROFILE)) && (!need_f || c->hint_valid) &&
      c->poll_enabled && c->acker_r < 0.f;
    if (!ok) batched = false;
    if (i == 0) {
      window_bytes = c->lane_window_bytes;
      obst = (flags[i] & SD_OBSTACLES) != 0;
    } else if (c->lane_window_bytes != window_bytes || ((flags[i] & SD_OBSTACLES) != 0) != obst) {
      batched = false;
    }
    Pmax = std::max(Pmax, c->P);
    gridx = std::max(gridx, c->grid_tpr);
  }
  const uint32_t T = c0->cfg.time_steps;
  const SmpcLds L = lane_lds(window_bytes, Pmax, T);
  if (L.total > kLdsPerCu) batched = false;
  if (!batched) {
    for (uint32_t i = 0; i < n; ++i) {
      const int rc = single(i);
      if (rc != SMPC_OK) return rc;
    }
    return SMPC_OK;
  }
  const auto t_prep = now();
  // ---- one upload, one scoring launch, one reduction launch ---------------------------------
  SmpcDev* hd = reinterpret_cast<SmpcDev*>(g->h_all + g->off_dev);
  SmpcReduceArgs* hr = reinterpret_cast<SmpcReduceArgs*>(g->h_all + g->off_red);
  for (uint32_t i = 0; i < n; ++i) {
    smpc_ctx* c = g->ctxs[i];
    SmpcFinal fin;
    fill_score_args(c, flags[i], nullptr, nullptr, c->hint, true, nullptr, hd[i], fin);
    hr[i].partials = c->d_partials;
    hr[i].tuple = c->d_tuple;
    hr[i].nblk = gridx;
    hr[i].host_out = fin.u_host;
    hr[i].seq = fin.seq;
    fin.u_host = nullptr;          // one publishing block for the whole group instead
    fin.done_counter = nullptr;
    hr[i].fin = fin;
    c->passes++;
    c->last_pass_kind = 1;
  }
  HIPCK(c0, hipMemcpyAsync(g->d_all, g->h_all, g->total, hipMemcpyHostToDevice, g->stream));
  HIPCK(c0, smpc_launch_pass_lane_many(reinterpret_cast<const SmpcDev*>(g->d_all + g->off_dev), n,
                                       T == 64, obst, L, gridx, g->stream));
  HIPCK(c0, smpc_launch_reduce_many(reinterpret_cast<const SmpcReduceArgs*>(g->d_all + g->off_red), n,
                                    T, c0->dev.neg_inv_temp, g->stream));
  g->batched_ticks++;
  const auto t_launch = now();
  // ---- per member: wait, verify the speculation and the collision count --------------------
  for (uint32_t i = 0; i < n; ++i) {
    smpc_ctx* c = g->ctxs[i];
    int rc = fetch_out(c);
    if (rc != SMPC_OK) return rc;
    const bool need_f = (flags[i] & SD_NEED_FURTHEST) != 0;
    const uint32_t S_true = static_cast<uint32_t>(c->h_out[3 * T + 2]);
    const bool miss = need_f && S_true != c->hint;
    const bool all_collide = (flags[i] & (SD_OBSTACLES | SD_COST)) && c->h_out[3 * T + 3] == 0.0f;
    if (miss || all_collide) {
      if (miss) {
        c->spec_misses++;
        c->hint = S_true;   // smpc_optimize speculates with the true value now: one pass
      }
      rc = single(i);
      if (rc != SMPC_OK) return rc;
      continue;
    }
    store_control_sequence(c, u_inout[i]);
    if (outs) {
      smpc_tick_out* o = &outs[i];
      memset(o, 0, sizeof(*o));
      o->furthest_valid = need_f ? 1 : 0;
      o->furthest_reached_path_point = need_f ? S_true : 0;
      o->non_colliding = static_cast<uint32_t>(c->h_out[3 * T + 3]);
      o->min_cost = c->h_out[3 * T + 0];
      o->sum_w = c->h_out[3 * T + 1];
      o->passes = c->passes;
      o->pass_kind = 1;
    }
  }
  if (timing && (g->batched_ticks % 64) == 0)
    fprintf(stderr, "[smpc_group] prepare %.1f us, fill+launch %.1f us, wait+collect %.1f us; batched %llu single %llu\n",
            us_between(t_start, t_prep), us_between(t_prep, t_launch), us_between(t_launch, now()),
            (unsigned long long)g->batched_ticks, (unsigned long long)g->single_ticks);
  return SMPC_OK;
}

}  // extern "C"
