// smpc_ctx.h — what the translation units behind include/smpc.h share: the context object,
// the launchers of the .hip files, and the host-side helpers (smpc_impl).  Internal: not part of
// the C-ABI.
//   smpc_api.cpp      lifecycle, setters, the single-GPU tick (smpc_optimize), launch helpers
//   smpc_prepare.cpp  per-tick host work: gates, path tables, lookup tables, LDS carve-up
//   smpc_shard.cpp    batch-sharded tick, RCCL resolved at run time
//   smpc_group.cpp    several planning instances per launch
#ifndef SMPC_CTX_H_
#define SMPC_CTX_H_

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

// nothing below is part of the C-ABI: keep it out of the dynamic symbol table
#pragma GCC visibility push(hidden)

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
hipError_t smpc_launch_fill_noise_tm(float* dst, uint32_t B, uint32_t T, uint64_t base, uint64_t seed,
                                     uint32_t stream, uint32_t epoch, float sigma, hipStream_t st);
hipError_t smpc_launch_ackermann(float* u_dev, float* u_host, uint32_t T, float min_r, uint32_t seq,
                                 hipStream_t st);
hipError_t smpc_launch_p2p_exchange(const float* my_tuple, const SmpcP2P& x, uint32_t T, int mode,
                                    float* d_furthest, float neg_inv_temp, float vx_max, float vx_min,
                                    float vy_max, float wz_max, float* u_out, float* result,
                                    const float* furthest_used, float* host_out, uint32_t seq,
                                    hipStream_t st);

hipError_t smpc_launch_relayout(const float* src, float* dst, uint32_t B, uint32_t T, bool to_gm, hipStream_t st);
hipError_t smpc_launch_pass_lane(const SmpcDev& p, const SmpcLds& L, uint32_t grid, bool rr, uint32_t block, hipStream_t st);
uint32_t smpc_lane_block();
uint32_t smpc_lane_block_rr();
hipError_t smpc_lane_occupancy(bool full, uint32_t lds_bytes, int* blocks_per_cu);
hipError_t smpc_lane_occupancy_rr(uint32_t T, uint32_t lds_bytes, int* blocks_per_cu);
hipError_t smpc_lane_set_lds_limit(int bytes);
hipError_t smpc_launch_lane_reduce(const float* v, const float* w, float* out, hipStream_t st);
// smpc_split.hip: lane = (rollout, quarter of the horizon), for batches of at most one round of waves
hipError_t smpc_launch_pass_split(const SmpcDev& p, const SmpcLds& L, uint32_t grid, uint32_t nseg, hipStream_t st);
hipError_t smpc_launch_row_reduce(const float* v, float* out, uint32_t n, hipStream_t st);
uint32_t smpc_split_block();
uint32_t smpc_split_rollouts_per_block(uint32_t nseg);
hipError_t smpc_split_occupancy(uint32_t nseg, uint32_t lds_bytes, int* blocks_per_cu);
hipError_t smpc_split_set_lds_limit(int bytes);
hipError_t smpc_launch_pass_lane_many(const SmpcDev* d_many, uint32_t n, bool full, bool obst, bool dep, uint32_t T,
                                      const SmpcLds& L, uint32_t grid, uint32_t block, hipStream_t st);
hipError_t smpc_launch_reduce_many(const SmpcReduceArgs* d_many, uint32_t n, uint32_t T, hipStream_t st);
// developer aid: the scoring-pass instance launched last, as rocprofv3 names it
extern char smpc_last_pass_kernel[96];
hipError_t smpc_launch_sincos(const float* x, uint32_t n, float* sn, float* cs, hipStream_t st);

namespace smpc_impl {

extern thread_local std::string g_create_error;


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
const RcclApi* rccl();   // smpc_shard.cpp; null when RCCL cannot be loaded

// threads per block of the streaming pass: 16 waves share one costmap window and produce one
// partial (fewer partials to reduce); T > 128 needs more registers per lane than 16 waves allow
inline uint32_t pass_block(int R) {return R == 4 ? 512u : 1024u;}
constexpr uint32_t kLaneMinBatch = 60u * 1024u;   // lane-per-rollout pass from this batch size up (measured crossover ~50k: 65 536 x 64 takes 35.9 us against 40.4 us)
constexpr uint32_t kSplitMinBatch = 12u * 1024u;  // smpc_pass_split (T = 64) from this batch size up to one group per wave (measured: 16 384 x 64
                                                  // 38.9 -> 36.2 us per tick, 32 768: 44.7 -> 36.7; 4 096: no gain)
constexpr uint32_t kSplitMinBatchShort = 20000u;  // the same for 36 <= T < 64 (idle step slots are masked, not skipped)
constexpr uint32_t kLaneMaxT = 128;       // T <= 64: 3 x 64 noised controls parked per lane (or re-read); T <= 128: re-read
constexpr uint32_t kPollWords = 32;     // completion words behind h_out[3T + 8] (T = 256: 25 blocks of smpc_reduce_partials)
constexpr uint32_t kMaxGrid = SMPC_MAX_GRID;   // smpc_reduce_partials stages this many factors
constexpr uint32_t kWindowBytes = 96 * 96;  // costmap window staged in LDS: 4.8 m x 4.8 m at
                                           // 0.05 m around the robot; the rest is read from HBM/L2
constexpr uint32_t kWindowSideMax = 144;       // T > 64: up to 144 x 144 cells (20 KB, +-3.6 m at 0.05 m: measured optimum, smpc_prepare.cpp)
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

}  // namespace smpc_impl

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
  float* d_tvx = nullptr;       // group-major copies (smpc_dev.h: SMPC_GM_INDEX) for the lane-per-rollout and split passes:
  float* d_tvy = nullptr;       // one allocation, vy and wz follow vx
  float* d_twz = nullptr;
  // A second set of noise tensors for smpc_redraw_noise_async (regenerate_noises = true): the
  // next epoch is drawn into it on fill_stream while the host and the device go on; the next
  // tick's stream waits for ev_fill and the sets change places (absorb_redraw).
  float* b_nvx = nullptr;
  float* b_nvy = nullptr;
  float* b_nwz = nullptr;
  float* b_tvx = nullptr;       // (one allocation, vy and wz follow vx)
  float* b_tvy = nullptr;
  float* b_twz = nullptr;
  hipStream_t fill_stream = nullptr;
  hipEvent_t ev_fill = nullptr;
  bool redraw_pending = false, redraw_rm_valid = true;
  uint32_t noise_gen = 0;              // counts changes of the noise the ticks score with
  uint32_t noise_gen_remembered = 0;   // ... as of the last remember_furthest
  bool use_tpr = false;      // group-major noise kept: the lane-per-rollout pass may run
  bool rm_valid = true;      // the [B,T] tensors hold the current noise (a device-RNG draw fills the
                             // group-major copy only; ensure_row_major() makes the other on demand)
  bool lane_now = false;     // ... and does for this tick (lean scoring mode)
  bool split_now = false;    // ... as smpc_pass_split (lane = rollout x quarter of the horizon): small batches at T = 64
  SmpcLds lds_split{};
  uint32_t grid_split = 0;
  uint32_t split_nseg = 4;               // lanes per rollout of this tick's split pass: 4 or 2
  int occ_split_blocks = -1;
  uint32_t occ_split_lds = 0xffffffffu;
  uint32_t knob_split_nseg = 0;          // SMPC_SPLIT_NSEG=2|4: experiments
  bool lane_forced = false;              // SMPC_FLAG_LANE_PER_ROLLOUT / SMPC_PASS=lane|split: the lane pass below kLaneMinBatch too
  bool knob_no_split = false;            // SMPC_NO_SPLIT=1
  bool knob_repeat_pass = false;         // SMPC_DEBUG_REPEAT_PASS=1 (tests): every iteration's scoring pass is launched twice
  uint32_t knob_stale_tick = 0;          // SMPC_DEBUG_STALE_TICK=n (tests): the BAR hand-over skips tick n's block
  bool two_coll_fp = false;              // this tick: BOTH collision critics scored and a consider_footprint switch set —
                                         // they then disagree on which rollouts collide (smpc_optimize: a counting pass)
  bool knob_force_split = false;         // SMPC_PASS=split: wherever the instance applies, whatever the batch
  bool lane_rr = false;      // ... in its re-read form (no parked controls; T > 64 or SMPC_LANE_REREAD=1)
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
  // collective-free exchange (smpc_shard_p2p_*): the own mailbox (fine-grained device memory,
  // exported over IPC) and the peers' mailboxes as mapped into this process
  float* p2p_mailbox = nullptr;
  SmpcP2P p2p{};               // world == 0: not set up
  uint32_t p2p_xseq = 0;
  uint32_t p2p_timeout_ms = 10000;   // bound of the in-kernel wait for the peers (smpc_shard_p2p_set_timeout)
  bool p2p_failed = false;           // an exchange timed out: no further mailbox tick until re-init
  SmpcLds lds_tpr{};
  uint32_t grid_tpr = 0;
  uint32_t lane_block = 0;      // threads per block of the lane pass this tick
  bool pang_any = false;        // this tick: PathAngleCritic is live for some candidate furthest point (prepare_tick)
  bool in_group = false;        // member of an smpc_group: full-size blocks always (the group fills the CUs by itself)
  // developer knobs, read from the environment when the context is created (never per tick)
  uint32_t knob_max_blocks_per_cu = 0;   // SMPC_MAX_BLOCKS_PER_CU
  bool knob_lane_reread = false;         // SMPC_LANE_REREAD=1: the re-read form for T = 64 too
  bool knob_no_inline_tick = false;      // SMPC_NO_INLINE_TICK
  bool knob_balanced_grid = true;        // SMPC_NO_BALANCED_GRID=1 turns it off: re-read form, launch only the waves that fill every round
  bool knob_pinned_tick = false;         // SMPC_PINNED_TICK=1: kernels read the tick block from pinned host memory
  bool half_blocks = true;      // (SMPC_NO_HALF_BLOCKS, read when the context is created: experiments)
  uint32_t occ_tpr_blocks = 0, occ_tpr_lds = 0xffffffffu;
  float* d_costs[2] = {nullptr, nullptr};
  float* d_traj[3] = {nullptr, nullptr, nullptr};
  int costs_cur = 0;
  bool have_noise = false, rng_mode = false;
  uint64_t seed = 0;
  uint32_t epoch = 0;
  // costmap
  smpc_impl::HostCostmap map;
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
  // The per-tick upload as CPU stores straight into device memory (large-BAR systems: every
  // MI355X host maps the whole HBM) instead of a copy on the stream: no blit kernel (3.4 us) and
  // no dependent-dispatch gap (4 us) in front of the scoring pass.  SMPC_NO_BAR_TICK=1: the copy.
  bool bar_tick = false;
  volatile uint32_t* hdp_flush = nullptr;   // HDP_MEM_FLUSH_CNTL of this device, or null
  size_t tick_used = 0;         // bytes of the tick block this tick fills (a multiple of 16)
  uint32_t tick_no = 0;         // number of the tick block last handed over (its canary word)
  uint32_t canary_expect = 0;   // != 0: fetch_out checks the pass's echo against it
  bool launched = false;        // kernels of this ctx may still be reading the tick block
  // reductions / outputs
  float* d_partials = nullptr;
  float* d_tuple = nullptr;
  float* d_out = nullptr;       // [3T u][8 result]
  float* d_u_iter = nullptr;    // [3T] the control sequence iteration it > 0 starts from: a copy of d_out, so that a
                                // second pass of the same iteration (a re-score) reads what the first one read
  float* h_out = nullptr;       // pinned, device-mapped: kernels write the result here
  float* h_out_dev = nullptr;   // its device-side address
  bool fused_reduce = false;    // SMPC_FUSED_REDUCE=1: smpc_grid_tail reduces inside the scoring launch (an experiment
                                // that measured no faster than the separate launch: DESIGN.md 4.3)
  float* d_furthest = nullptr;  // word 0: the furthest point (atomicMax on its bits); 2: smpc_reduce_partials' block count;
                                // 3: the mailbox exchange's state; 4: smpc_grid_tail's block count
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
  // speculation on furthest_reached_path_point: the index this tick is scored with first
  // (predict_hint: last tick's value carried forward by the robot's motion and the plan's shift)
  bool hint_valid = false;
  uint32_t hint = 0;
  uint64_t spec_misses = 0;
  // what the prediction starts from: the last known F = index + fraction (smpc_dev.h) and the
  // tick inputs it was measured on
  float hint_F = 0.f;
  float hint_Fp = 0.f;       // the last prediction, unrounded; valid when hint_Fp_valid
  bool hint_Fp_valid = false;
  smpc_tick_in step_in{};       // step-wise sharded API: this tick's inputs (smpc_shard_begin .. smpc_shard_combine),
  std::vector<float> step_px, step_py;   // with the path it points at copied here
  bool step_remembered = false;
  bool hint_is_this_ticks = false;   // set by the group's re-run of a missed member (predict_hint)
  float hint_drift = 0.f;    // last tick's (true - predicted): how fast the endpoints drift relative to the robot
  bool anchor_valid = false;
  double anchor_x = 0, anchor_y = 0;
  // endpoint of the NOMINAL rollout (the control sequence a tick starts from, no noise) of the last
  // remembered tick and of the tick at hand: what carries the furthest point from tick to tick
  double anchor_ex = 0, anchor_ey = 0, cur_ex = 0, cur_ey = 0;
  bool anchor_e_valid = false, cur_e_valid = false;
  std::vector<float> anchor_px, anchor_py;
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
  uint32_t poll_words = 0;   // > 0: the tick's completion arrives as this many words (smpc_tail.h), h_out[3T + 8 ...]
  std::string err;
};

namespace smpc_impl {

int fail(smpc_ctx* c, int code, const std::string& msg);

#define HIPCK(ctx, call)                                                               \
  do {                                                                                 \
    hipError_t e__ = (call);                                                           \
    if (e__ != hipSuccess)                                                             \
      return fail(ctx, SMPC_ERR_DEVICE,                                                \
                  std::string(#call) + ": " + hipGetErrorString(e__));                \
  } while (0)

int wait_map_upload(smpc_ctx* c);
// n bytes (a multiple of 16, both 16-byte aligned) from pinned host memory into BAR-mapped device
// memory: streaming stores, fenced before the caller rings any doorbell
void bar_copy(void* dev_dst, const void* host_src, size_t n);
// ... and the device's host-data-path flush behind them (HSA_AMD_AGENT_INFO_HDP_FLUSH), where the
// runtime exposes one
void bar_flush(const smpc_ctx* c);
void free_ctx(smpc_ctx* c);

// tick block layout (offsets in bytes), sized for the ctx's T and SMPC_MAX_PATH
struct TickLayout {
  size_t u, px, py, pyaw, D, pf_idx, pvalid, pa_active, pang_active, pal_active, lut_cost, canary, total;
};
TickLayout tick_layout(uint32_t T, uint32_t P);

// LDS carve-up of the streaming pass.  nsamp = PathAlign samples per rollout (0: off).
SmpcLds make_lds(uint32_t window_bytes, uint32_t P, uint32_t T, uint32_t nwave, bool with_map,
                 uint32_t nsamp = 0);
SmpcLds lane_lds(uint32_t window_bytes, uint32_t P, uint32_t T, bool rr = false);
SmpcLds split_lds(uint32_t window_bytes, uint32_t P, uint32_t T, uint32_t nseg);

int check_tick(smpc_ctx* c, const smpc_tick_in* in);
// the furthest point F (index + fraction) is now known for the tick inputs `in`: the next
// prediction starts from here
void remember_furthest(smpc_ctx* c, const smpc_tick_in* in, float F);
void predict_hint(smpc_ctx* c, const smpc_tick_in* in, const float* u_in);
int prepare_tick(smpc_ctx* c, const smpc_tick_in* in, const float* u_in);

int launch_furthest(smpc_ctx* c, float* d_furthest);
void fill_score_args(smpc_ctx* c, uint32_t flags, const float* u_dev, const float* d_furthest,
                     uint32_t furthest_hint, bool finish, const float* finish_furthest, SmpcDev& d,
                     SmpcFinal& fin);
int launch_score(smpc_ctx* c, uint32_t flags, const float* u_dev, const float* d_furthest,
                 uint32_t furthest_hint, float* d_tuple, bool finish = false,
                 const float* finish_furthest = nullptr);
int launch_combine(smpc_ctx* c, const float* d_tuples, uint32_t n, const float* d_furthest_used);
void store_control_sequence(const smpc_ctx* c, float* u_inout);
int fetch_out(smpc_ctx* c);
float profile_pass_ms(smpc_ctx* c);
uint32_t fail_only_flags(const smpc_ctx* c);
uint32_t scoring_flags(const smpc_ctx* c, bool fail_sticky);
int update_time_major(smpc_ctx* c);
int ensure_row_major(smpc_ctx* c);
int draw_noise(smpc_ctx* c);
int redraw_async(smpc_ctx* c);
int cancel_redraw(smpc_ctx* c);
int absorb_redraw(smpc_ctx* c);

}  // namespace smpc_impl

#pragma GCC visibility pop

#endif  // SMPC_CTX_H_
