// smpc_dev.h — host<->kernel parameter block and launch geometry (internal).
#ifndef SMPC_DEV_H_
#define SMPC_DEV_H_

#include <stdint.h>

// critic / mode bits of SmpcDev::flags
#define SD_OBSTACLES 0x001u       // ObstaclesCritic scored this pass
#define SD_PATH_ALIGN 0x002u      // PathAlignCritic past its host-side gates (still gated per S)
#define SD_PATH_FOLLOW 0x004u     // PathFollowCritic past its gates
#define SD_GOAL_ANGLE 0x008u      // GoalAngleCritic (robot within threshold of the goal)
#define SD_PREFER_FORWARD 0x010u  // PreferForwardCritic past its gate
#define SD_USE_PATH_YAW 0x020u    // PathAlign use_path_orientations
#define SD_ACCUMULATE 0x040u      // costs start from costs_prev (iteration > 0)
#define SD_STORE_TRAJ 0x080u      // write x,y,yaw [B,T]
#define SD_TRACK_UNKNOWN 0x100u   // costmap tracks unknown space (255 is not a collision)
#define SD_NEED_FURTHEST 0x200u   // some critic consumes furthest_reached_path_point
#define SD_LOCAL_FURTHEST 0x400u  // the scoring pass also reports its own furthest point (speculation)
// the other registered critics (general pass, MODE 2, only)
#define SD_CONSTRAINT 0x800u
#define SD_COST 0x1000u
#define SD_GOAL 0x2000u
#define SD_TWIRLING 0x4000u
#define SD_PATH_ANGLE 0x8000u
#define SD_DEADBAND 0x10000u
#define SD_FP_OBSTACLES 0x20000u   // ObstaclesCritic consider_footprint
#define SD_FP_COST 0x40000u        // CostCritic consider_footprint
#define SD_PATH_ALIGN_LEGACY 0x80000u   // PathAlignLegacyCritic past its host-side gates (still gated per S)
#define SD_PAL_USE_PATH_YAW 0x100000u  // ... with use_path_orientations
#define SD_EXTRA_CRITICS (SD_CONSTRAINT | SD_COST | SD_GOAL | SD_TWIRLING | SD_PATH_ANGLE | SD_DEADBAND | \
                          SD_FP_OBSTACLES | SD_FP_COST | SD_PATH_ALIGN_LEGACY)

// Layout of the noise the lane-per-rollout and split passes read ("group-major"): element (b, t) of
// a [B, T] tensor sits at ((b / 64) T + t) 64 + b % 64 — the 64 rollouts of a wave-group and their T
// steps are one run of 256 T bytes, where the time-major [T][B] layout of rounds 1-2 put a group's
// consecutive steps B x 4 bytes (8 MB at the bench size) apart: one page per step and tensor
// (measured at 2 097 152 x 64: 391 -> 376 us per pass).  B is padded to a multiple of 64.
#define SMPC_GM_ROLLOUTS(B) ((((B) + 63u) / 64u) * 64u)
#define SMPC_GM_INDEX(b, t, T) ((((size_t)(b) >> 6) * (T) + (t)) * 64u + ((b) & 63u))

#define SMPC_MAX_PATH 1024        // path points staged in LDS
#define SMPC_MAX_R 4              // time steps per lane (T <= 64 * SMPC_MAX_R)

struct SmpcLut {  // per 8-bit cost: {margin - d | 0, R_infl - d | 0}  (obstacles_critic.cpp:159-170)
  float crit;        // < 0 marks inCollision(cost) (obstacles_critic.cpp:185-201); a real
  float rep;         // margin - d is always > 0
};

// optional finishing step of smpc_reduce_partials (single tuple -> new control sequence)
struct SmpcFinal {
  int enabled;
  float vx_max, vx_min, vy_max, wz_max;   // current constraints
  float* u_dev;                            // [3T + 8] device copy (next iteration reads u here)
  float* u_host;                           // [3T + 8] host-mapped copy (no D2H memcpy)
  const float* furthest_used;              // device float or null
  uint32_t* done_counter;                  // device word, zero between launches
  uint32_t seq;                            // tick sequence number published at u_host[3T+7]
};

struct SmpcDev {
  // sizes
  uint32_t B, T, P;
  uint32_t nsamp;  // PathAlign samples per trajectory: (T-1)/step
  uint32_t step;   // trajectory_point_step
  uint32_t flags;
  // robot state (Optimizer::prepare)
  double x0, y0;
  float yaw0, cos0, sin0;
  float svx, svy, swz;
  float dt;
  // tensors
  const float* nvx;         // noise [B,T] (reference layout; wave-per-rollout pass)
  const float* nvy;
  const float* nwz;
  const float* tvx;         // the same noise group-major (SMPC_GM_INDEX; lane-per-rollout and split passes)
  const float* tvy;
  const float* twz;
  const float* u;           // [3T] control sequence (device)
  const float* costs_prev;  // [B] (SD_ACCUMULATE)
  float* costs;             // [B]
  float* traj_x;            // [B,T] (SD_STORE_TRAJ)
  float* traj_y;
  float* traj_yaw;
  // costmap
  const uint8_t* map;
  uint32_t W, H;
  double ox, oy, res;
  float oxf, oyf, rinvf;   // float images for the fast cell index
  float cell_eps;          // bound on |float quotient - double quotient| (guard band)
  uint32_t cost_t0;        // costmap cost under trajectory point 0 (same for every rollout)
  int32_t win_x0, win_y0, win_w, win_h;  // window staged in LDS (cells)
  float wxf, wyf;          // float images of the window corner (lane-per-rollout pass)
  float cxf, cyf;          // ((x0, y0) - window corner) / resolution (lane-per-rollout pass)
  float cell_eps_w;        // guard band for the window-relative quotient
  float x00f, y00f;        // trajectory point 0 (identical for every rollout)
  const SmpcLut* lut;                     // [256]
  // path block (device)
  const float* px;
  const float* py;
  const float* pyaw;
  const float* D;            // cumulative path length, P-1 entries (path_align_critic.cpp:83-90)
  const uint8_t* pvalid;     // P-1 entries
  const uint8_t* pa_active;  // [P] PathAlign gate per candidate furthest point
  const uint32_t* pf_idx;    // [P] PathFollow target index per candidate furthest point
  // furthest point: device value (float) if non-null, else the hint
  const float* d_furthest;
  uint32_t furthest_hint;
  // critic constants
  float obs_critical_w, obs_repulsion_w, obs_collision_cost;
  float obs_rep_over_T;    // repulsion_weight / T (MODE 0)
  uint32_t obs_power;
  float pa_weight;
  uint32_t pa_power;
  float pf_weight;
  uint32_t pf_power;
  float ga_weight, ga_goal_yaw;
  uint32_t ga_power;
  float pfw_weight;
  uint32_t pfw_power;
  // the other registered critics (MODE 2)
  float con_weight, con_max_vel, con_min_vel;   // constraint_critic.cpp:36-38
  uint32_t con_power;
  float con_acker_r;   // Ackermann min_turning_r (constraint_critic.cpp:54-59), < 0: other models
  const float* lut_cost;                         // [256] CostCritic repulsive term per 8-bit cost
  float cost_w254, cost_collision_cost;          // cost_weight / 254 (cost_critic.cpp:34)
  float cost_critical;                           // critical_cost (cost_critic.cpp:144-147)
  uint32_t cost_near_goal;                       // robot within near_goal_distance: no repulsion (:120-124,150)
  uint32_t cost_power;
  double goal_x, goal_y;
  float goal_weight;
  uint32_t goal_power;
  float tw_weight;
  uint32_t tw_power;
  const uint8_t* pang_active;                    // [P] PathAngle gate per candidate furthest point
  float pang_weight;
  uint32_t pang_power, pang_offset;
  int32_t pang_correct;                          // reversing allowed and no forward preference
  const uint8_t* pal_active;                     // [P] PathAlignLegacy gate per candidate furthest point
  float pal_weight, pal_eval;                    // cost_weight; floor(T / trajectory_point_step) (path_align_legacy_critic.cpp:83)
  uint32_t pal_power;
  double db_vx, db_vy, db_wz;                    // |deadband_velocities|
  float db_weight;
  uint32_t db_power;
  // consider_footprint: footprint polygon (robot frame), possibly-inscribed cost
  // (findCircumscribedCost), and the LUT pair for the consider_footprint collision rule:
  // [0] cost from the centre point, [1] cost from the footprint (no inscribed-radius offset)
  uint32_t fp_n;
  float fp_pic;
  const SmpcLut* lut_fp;                         // [2][256]
  double fp_x[16], fp_y[16];
  float g_vx, g_vy, g_wz;  // gamma / std^2 (optimizer.cpp:367-379)
  float neg_inv_temp;      // -1 / temperature (optimizer.cpp:383)
  float k2;                // neg_inv_temp * log2(e): weights as 2^(k2 (c - min))
  // outputs
  float* partials;         // [gridDim.x][4 + 3T] per-block softmax partials
  uint32_t* furthest_out;  // atomicMax target of the furthest-only pass (float bits)
  // The reduction of the grid's partials inside the scoring launch (smpc_tail.h): the block that
  // finishes last reduces them — no second launch, no dependency gap in front of it.
  uint32_t tail;            // 1: on (the host sets it for grids of at most SMPC_TAIL_MAX_GRID blocks)
  uint32_t* tail_counter;   // device word, zero between launches
  float* tuple;             // [4 + 3T] the shard tuple {min, sum w, furthest, non-colliding, U[3T]}
  SmpcFinal fin;            // single-GPU tick: finish the control sequence and publish it to the host
  // developer aid (SMPC_LANE_TIMELINE=1): [gridDim.x][8] shader-clock stamps of the lane pass
  unsigned long long* timeline;
  // The tick block reaches device memory by CPU stores through the PCIe BAR (smpc_prepare.cpp).
  // Its first word, four floats in front of u, is the tick's number; block 0 of a scoring pass
  // leaves the value IT read behind the grid's partials (SMPC_CANARY_SLOT), the reduction hands
  // it to the host with the result, and the host refuses a tick whose pass read another tick's
  // block.  (0: u is not in the tick block — kernel arguments, pinned host memory.)
  uint32_t canary_echo;
  // The tick block INSIDE the kernel arguments (small ticks: T <= 64, P <= 64).  The per-tick
  // upload is ~2 KB that the whole grid needs before its first instruction; as a separate
  // host-to-device copy it costs a blit kernel (3.4 us) and the dependency gap behind it
  // (4 us) on every tick.  Kernel arguments travel with the dispatch packet instead: the
  // kernels read u, the path and its tables straight from their own kernarg segment
  // (smpc_tick_ptrs).  tick_inline = 0: the pointers above (device memory) are used.
  uint32_t tick_inline;        // the path and its tables are in tick_bytes
  uint32_t u_inline;           // u is in u_arg (0: p.u, e.g. the previous iteration's result on the device)
  uint16_t io_px, io_py, io_pyaw, io_D, io_pf_idx, io_pvalid, io_pa_active, io_pang_active, io_pal_active, io_pad;
  uint8_t tick_bytes[1536] __attribute__((aligned(16)));
  // The control sequence u[3][T] of a tick with T <= 64 (u_inline).  Only the wave-per-rollout
  // pass, which reads u once, takes its tick block from the kernel arguments; the
  // lane-per-rollout pass re-reads u for every group and keeps it in device memory.
  float u_arg[3 * 64] __attribute__((aligned(16)));
};
#define SMPC_INLINE_TICK_CAP 1536u
#define SMPC_MAX_GRID 2048u                                /* blocks of a pass (partials the reduction stages) */
#define SMPC_CANARY_SLOT(T) (SMPC_MAX_GRID * (4u + 3u * (T)))   /* float index into SmpcDev::partials */
#define SMPC_TAIL_MAX_GRID 512u
#define SMPC_TAIL_STAMPS_AT (8192u + 2048u * 8u)   /* behind the lane pass's stamps in SmpcDev::timeline */
// LDS floats smpc_grid_tail works in (from offset 0 of the launch's dynamic LDS, which the pass
// no longer needs by then): 512 rescale factors, 4 x 16 wave results, [32][64] slice sums per 64 tuple
// columns, a flag
static inline uint32_t smpc_tail_lds_bytes(uint32_t T) {return (512u + 64u + ((4u + 3u * T + 63u) / 64u) * 2048u + 4u) * 4u;}

#if defined(__HIPCC__)
// where a kernel whose FIRST parameter is the SmpcDev finds the tick block (see tick_inline)
struct SmpcTickPtrs {
  const float* u;
  const float* px;
  const float* py;
  const float* pyaw;
  const float* D;
  const uint32_t* pf_idx;
  const uint8_t* pvalid;
  const uint8_t* pa_active;
  const uint8_t* pang_active;
  const uint8_t* pal_active;
};
__device__ __forceinline__ SmpcTickPtrs smpc_tick_ptrs(const SmpcDev& p, bool kernarg_is_dev)
{
  SmpcTickPtrs t{p.u, p.px, p.py, p.pyaw, p.D, p.pf_idx, p.pvalid, p.pa_active, p.pang_active, p.pal_active};
#if defined(__HIP_DEVICE_COMPILE__)
  if (kernarg_is_dev && p.tick_inline) {
    const uint8_t* k = reinterpret_cast<const uint8_t*>(__builtin_amdgcn_kernarg_segment_ptr()) +
                       __builtin_offsetof(SmpcDev, tick_bytes);
    if (p.u_inline) t.u = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(__builtin_amdgcn_kernarg_segment_ptr()) +
                                                         __builtin_offsetof(SmpcDev, u_arg));
    t.px = reinterpret_cast<const float*>(k + p.io_px);
    t.py = reinterpret_cast<const float*>(k + p.io_py);
    t.pyaw = reinterpret_cast<const float*>(k + p.io_pyaw);
    t.D = reinterpret_cast<const float*>(k + p.io_D);
    t.pf_idx = reinterpret_cast<const uint32_t*>(k + p.io_pf_idx);
    t.pvalid = k + p.io_pvalid;
    t.pa_active = k + p.io_pa_active;
    t.pang_active = k + p.io_pang_active;
    t.pal_active = k + p.io_pal_active;
  }
#else
  (void)kernarg_is_dev;
#endif
  return t;
}
#endif

// The furthest reached path point travels as ONE float F = S + f: S the index (max over the
// rollouts of the path point nearest to the rollout's endpoint, tools/utils.hpp:292-319) and
// f in [-0.45, 0.45] how far the extreme rollout's endpoint sits from point S towards S + 1, in
// segment lengths (0.5 would be the Voronoi edge where the nearest point flips).  max, the
// exchange between GPUs and the float slot of the tuple work on F unchanged; every consumer of
// the INDEX rounds.  The fraction only feeds the host's prediction of the next tick's index
// (smpc_prepare.cpp predict_hint): a wrong prediction costs a re-score, never a wrong result.
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline uint32_t smpc_furthest_index(float F) {return (uint32_t)(F + 0.5f);}


// mailboxes of the collective-free shard exchange (smpc_p2p_exchange)
#define SMPC_P2P_MAX_RANKS 16
struct SmpcP2P {
  float* peer[SMPC_P2P_MAX_RANKS];   // every rank's mailbox as mapped here; peer[rank] is the own one
  uint32_t world, rank;
  uint32_t slot_floats;              // floats per slot: the tuple, then the sequence word, padded
  uint32_t xseq;                     // number of this exchange (1, 2, ...; parity picks the half)
  uint32_t* state;                   // device word: 0 healthy, else an exchange of this ctx timed out
  unsigned long long timeout_ticks;  // bound of the wait in s_memrealtime ticks (10 ns each)
};

// one planning instance's arguments of smpc_reduce_partials_many
struct SmpcReduceArgs {
  const float* partials;
  float* tuple;
  uint32_t nblk;
  SmpcFinal fin;       // u_host and done_counter null: smpc_publish_many reports to the host
  float* host_out;     // the instance's host-mapped result mirror [3T + 8]
  uint32_t seq;        // sequence number published at host_out[3T + 7]
  float neg_inv_temp;  // -1 / temperature of THIS instance (members of a group may differ)
};

// LDS carve-up, computed once on the host and passed to the kernel.
struct SmpcLds {
  uint32_t off_lut, off_px, off_py, off_pyaw, off_D, off_valid, off_scr;
  uint32_t off_pts4;    // lane pass: path points as {x, y, segment valid ? 1 : 0, 0} [P]
  uint32_t scr_stride;  // floats per wave of scratch
  uint32_t scr_pts;     // float offsets inside a wave's scratch: sample points [3][64],
  uint32_t scr_ring;    //   parked endpoints [2][64],
  uint32_t scr_c;       //   parked noised controls [group][3][T]
  uint32_t seg_shift;   // log2 of the lanes per parked rollout in the flush (4, 5 or 6)
  uint32_t group;       // rollouts parked per flush
  uint32_t total;       // bytes
};

#endif
