/*
 * smpc.h — C-ABI of the MI355X-native sampling-MPC hot path.
 *
 * This is the drop-in boundary for ONE path of the reference
 * (soham2560/MPCHoloNavigation, package nav2_sortham_controller): the body of
 * sortham::Optimizer::optimize() — noise add, holonomic rollout, critic
 * scoring, softmax-weighted control update — i.e. everything between
 * Optimizer::prepare() and utils::savitskyGolayFilter() in
 * Optimizer::evalControl() (reference src/optimizer.cpp:134-164).
 *
 * Plain C, plain pointers and sizes; no C++/torch/ROS types cross it.
 * Every entry point cites the reference interface it replaces as
 * [ref file:line], paths relative to nav2_sortham_controller/.
 *
 * Conventions
 *   - every function returning int returns SMPC_OK (0) or a negative
 *     SMPC_ERR_* code; smpc_last_error(ctx) gives the text.  No exception
 *     crosses the ABI.  "All trajectories collide" is NOT an error: it is
 *     smpc_tick_out.fail_flag = 1 and the host runs Optimizer::fallback()
 *     [ref src/optimizer.cpp:166-183].
 *   - the caller owns every pointer it passes; the library copies what it
 *     needs before returning (device pointers in the *_device entry points
 *     are the exception and are documented there).
 *   - a ctx is used by one thread at a time [ref src/controller.cpp:94-103
 *     holds the parameter and costmap mutexes around the whole tick];
 *     distinct ctx are independent (the reference's function-static retry
 *     counter, src/optimizer.cpp:168, is per-host-object here).
 *   - tensors are row-major [batch, time] float32 exactly like the
 *     reference's xt::xtensor<float,2> members [ref models/state.hpp:30-57].
 */
#ifndef SMPC_H_
#define SMPC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMPC_ABI_VERSION 2

/* ---- status codes ------------------------------------------------------ */
#define SMPC_OK 0
#define SMPC_ERR_INVALID -1     /* bad argument / inconsistent sizes          */
#define SMPC_ERR_UNSUPPORTED -2 /* feature outside the hot-path scope         */
#define SMPC_ERR_DEVICE -3      /* HIP runtime error (text in last_error)     */
#define SMPC_ERR_STATE -4       /* call order (no costmap / no noise yet)     */
#define SMPC_ERR_NOMEM -5

/* ---- costmap cost constants [nav2_costmap_2d cost_values.hpp; used at
 *      ref src/critics/obstacles_critic.cpp:185-201, tools/utils.hpp:374-386] */
#define SMPC_COST_NO_INFORMATION 255
#define SMPC_COST_LETHAL 254
#define SMPC_COST_INSCRIBED 253
#define SMPC_COST_FREE 0

/* ---- motion models [ref include/.../motion_models.hpp:85-171,
 *      src/optimizer.cpp:414-426 setMotionModel].  Omni is the holonomic
 *      model the north star names.  The two non-holonomic models draw no vy
 *      noise, keep state.vy and control_sequence.vy at zero and return a Twist
 *      without linear.y [ref src/optimizer.cpp:220-224,241-243,264-266,
 *      334-337,374-389,404-410; src/noise_generator.cpp:117-121]; Ackermann
 *      also bounds the turning radius of the updated control sequence
 *      [ref motion_models.hpp:110-117] and adds a ConstraintCritic term
 *      [ref src/critics/constraint_critic.cpp:54-69].                       */
#define SMPC_MODEL_OMNI 0
#define SMPC_MODEL_DIFF_DRIVE 1
#define SMPC_MODEL_ACKERMANN 2

/* ---- smpc_config.flags ------------------------------------------------- */
#define SMPC_FLAG_STORE_TRAJECTORIES 0x1u /* materialise x,y,yaw [B,T] each
                                             pass so smpc_get_trajectories()
                                             works [ref optimizer.cpp:455-458] */
#define SMPC_FLAG_NO_SPECULATION 0x2u     /* always run the furthest-point
                                             pre-pass (exact two-pass mode)  */
#define SMPC_FLAG_WAVE_PER_ROLLOUT 0x8u   /* always the wave-per-rollout pass             */
#define SMPC_FLAG_LANE_PER_ROLLOUT 0x10u  /* the lane-per-rollout pass (csrc/smpc_lane.hip)
                                             on every tick it supports: time_steps <= 64,
                                             every cost_power 1, no GoalAngle term active */
#define SMPC_FLAG_PROFILE 0x4u            /* bracket each scoring pass with HIP
                                             events (smpc_tick_out.score_pass_ms) */

/*
 * Optimizer settings.  Mirrors sortham::models::OptimizerSettings,
 * ControlConstraints and SamplingStd
 * [ref models/optimizer_settings.hpp:28-41, models/constraints.hpp:25-42];
 * defaults are the ones Optimizer::getParams() declares
 * [ref src/optimizer.cpp:69-82]; see smpc_config_default().
 */
typedef struct smpc_config {
  uint32_t batch_size;      /* B, this ctx's rollouts (a shard when sharded) */
  uint32_t time_steps;      /* T                                             */
  uint32_t iteration_count; /* optimize() iterations per tick                */
  uint32_t motion_model;    /* SMPC_MODEL_*                                  */
  float model_dt;
  float temperature;
  float gamma;
  float vx_max, vx_min, vy_max, wz_max; /* base constraints                   */
  float vx_std, vy_std, wz_std;         /* sampling std                       */
  int32_t device;                       /* HIP device ordinal, -1 = current   */
  uint32_t flags;                       /* SMPC_FLAG_*                        */
  /* batch sharding: this ctx holds rows
   * [shard_offset, shard_offset+batch_size) of a global batch of
   * global_batch_size rollouts.  0/0 means "not sharded". */
  uint64_t shard_offset;
  uint64_t global_batch_size;
  /* AckermannConstraints.min_turning_r [ref motion_models.hpp:91-95],
   * default 0.2; read only when motion_model == SMPC_MODEL_ACKERMANN. */
  float ackermann_min_turning_r;
  uint32_t reserved0;
} smpc_config;

/*
 * Critic parameters: the five critics the north star names, each with the
 * parameter names and defaults of its initialize()
 * [ref src/critics/obstacles_critic.cpp:21-51, path_align_critic.cpp:26-44,
 *  path_follow_critic.cpp:23-33, goal_angle_critic.cpp:20-34,
 *  prefer_forward_critic.cpp:20-31].  `enabled` mirrors CriticFunction's
 * enabled_ [ref critic_function.hpp:65-106] AND membership in the YAML
 * `critics` list [ref src/critic_manager.cpp:36-60].
 */
typedef struct smpc_obstacles_params {
  int32_t enabled;
  int32_t consider_footprint; /* 1: SE2 footprint check near obstacles (needs
                                 smpc_set_footprint; general pass only).  With
                                 BOTH ObstaclesCritic and CostCritic in the list
                                 and either one's consider_footprint set, the
                                 two disagree on which rollouts collide and a
                                 shard tuple carries one non-colliding count:
                                 smpc_optimize scores such a tick with an extra
                                 counting pass, the sharded tick
                                 (smpc_shard_*) refuses it (SMPC_ERR_UNSUPPORTED) */
  uint32_t cost_power;
  float repulsion_weight;
  float critical_weight;
  float collision_cost;
  float collision_margin_distance;
  float near_goal_distance;
} smpc_obstacles_params;

typedef struct smpc_path_align_params {
  int32_t enabled;
  int32_t use_path_orientations;
  uint32_t cost_power;
  float cost_weight;
  float max_path_occupancy_ratio;
  uint32_t offset_from_furthest;
  uint32_t trajectory_point_step;
  float threshold_to_consider;
} smpc_path_align_params;

typedef struct smpc_path_follow_params {
  int32_t enabled;
  uint32_t cost_power;
  float cost_weight;
  float threshold_to_consider;
  uint32_t offset_from_furthest;
} smpc_path_follow_params;

typedef struct smpc_goal_angle_params {
  int32_t enabled;
  uint32_t cost_power;
  float cost_weight;
  float threshold_to_consider;
} smpc_goal_angle_params;

typedef struct smpc_prefer_forward_params {
  int32_t enabled;
  uint32_t cost_power;
  float cost_weight;
  float threshold_to_consider;
} smpc_prefer_forward_params;

/*
 * The other registered critics of the holonomic stack, scored by
 * the general streaming pass.  `enabled` defaults to 0 (not in the `critics` list).
 * Parameter names and defaults: cost_critic.cpp:25-34, goal_critic.cpp:26-28,
 * constraint_critic.cpp:27-38, twirling_critic.cpp:24-25, path_angle_critic.cpp:24-50,
 * velocity_deadband_critic.cpp:24-33.
 */
typedef struct smpc_cost_params {
  int32_t enabled;
  int32_t consider_footprint; /* 1: SE2 footprint check near obstacles (needs
                                 smpc_set_footprint; general pass only)            */
  uint32_t cost_power;
  float cost_weight;          /* as in the YAML (3.81); divided by 254 inside    */
  float critical_cost;
  float collision_cost;
  float near_goal_distance;
} smpc_cost_params;

typedef struct smpc_goal_params {
  int32_t enabled;
  uint32_t cost_power;
  float cost_weight;
  float threshold_to_consider;
} smpc_goal_params;

typedef struct smpc_constraint_params {
  int32_t enabled;
  uint32_t cost_power;
  float cost_weight;
  float vx_max, vy_max, vx_min; /* the controller's vx_max / vy_max / vx_min at
                                   initialize() (constraint_critic.cpp:31-38)     */
} smpc_constraint_params;

typedef struct smpc_twirling_params {
  int32_t enabled;
  uint32_t cost_power;
  float cost_weight;
} smpc_twirling_params;

typedef struct smpc_path_angle_params {
  int32_t enabled;
  uint32_t cost_power;
  float cost_weight;
  uint32_t offset_from_furthest;
  float threshold_to_consider;
  float max_angle_to_furthest;
  int32_t forward_preference;
  float vx_min;                 /* the controller's vx_min: reversing allowed iff
                                   it is negative (path_angle_critic.cpp:24-31)   */
} smpc_path_angle_params;

typedef struct smpc_velocity_deadband_params {
  int32_t enabled;
  uint32_t cost_power;
  float cost_weight;
  float deadband_velocities[3]; /* vx, vy, wz                                     */
} smpc_velocity_deadband_params;

/* Scoring order (CriticManager scores in list order and stops at the first critic that
 * sets fail_flag, critic_manager.cpp:67-76): Constraint, Cost, Obstacles, PathAlign, PathAlignLegacy,
 * PathFollow, GoalAngle, PreferForward, Goal, PathAngle, Twirling, VelocityDeadband.  A
 * different YAML order changes float summation order (last bits) and which costs the
 * discarded all-collide tick carries, nothing else. */
typedef struct smpc_critic_params {
  smpc_obstacles_params obstacles;
  smpc_path_align_params path_align;
  smpc_path_follow_params path_follow;
  smpc_goal_angle_params goal_angle;
  smpc_prefer_forward_params prefer_forward;
  smpc_cost_params cost;
  smpc_goal_params goal;
  smpc_constraint_params constraint;
  smpc_twirling_params twirling;
  smpc_path_angle_params path_angle;
  smpc_velocity_deadband_params velocity_deadband;
  /* PathAlignLegacyCritic (path_align_legacy_critic.cpp:26-44: the parameters of PathAlignCritic,
   * the pre-October-2023 formulation: every trajectory sample against its nearest path point by
   * brute force, :84-124), scored right behind PathAlign by the general pass.  ABI version 2. */
  smpc_path_align_params path_align_legacy;
} smpc_critic_params;

/*
 * Per-tick inputs: what Optimizer::prepare() copies into the optimizer
 * [ref src/optimizer.cpp:185-204] plus the sticky fail flag the critic
 * manager consults [ref src/critic_manager.cpp:67-76].
 */
typedef struct smpc_tick_in {
  double pose_x, pose_y; /* state.pose.pose.position (double in the msg)       */
  float pose_yaw;        /* (float)tf2::getYaw(orientation) [optimizer.cpp:317] */
  double speed_vx, speed_vy, speed_wz; /* state.speed (Twist, double)          */
  const float* path_x;   /* utils::toTensor(plan) [ref tools/utils.hpp:180-192] */
  const float* path_y;
  const float* path_yaw;
  uint32_t path_len;     /* P                                                  */
  double goal_x, goal_y; /* goal.position                                      */
  /* Optional P-1 entries of CriticData::path_pts_valid
   * [ref tools/utils.hpp:361-395]; NULL = derive from the ctx's costmap.     */
  const uint8_t* path_pts_valid;
  /* 1 = CriticData::fail_flag is already set (retry after fallback():
   * CriticManager scores nothing) [ref critic_manager.cpp:70-73].            */
  int32_t fail_flag_in;
  /* GoalChecker::getTolerances(): pose_tolerance.position.x; < 0 = no goal checker
   * (TwirlingCritic's gate, twirling_critic.cpp:33-37, tools/utils.hpp:201-224).       */
  float goal_checker_xy_tolerance;
} smpc_tick_in;

typedef struct smpc_tick_out {
  int32_t fail_flag;        /* all trajectories collide [obstacles_critic.cpp:177] */
  int32_t furthest_valid;   /* furthest_reached_path_point was evaluated           */
  uint32_t furthest_reached_path_point; /* [ref tools/utils.hpp:292-319]           */
  uint32_t non_colliding;   /* rollouts that did not collide (last iteration)      */
  float min_cost;           /* xt::amin(costs_) of the last iteration              */
  float sum_w;              /* sum of exp weights of the last iteration            */
  uint32_t passes;          /* scoring passes over the noise (speculation misses
                               show up as passes > iteration_count)               */
  float device_ms;          /* SMPC_FLAG_PROFILE: GPU time of this call, first upload to
                               last kernel (HIP events on the ctx's stream); else 0 */
  float score_pass_ms;      /* SMPC_FLAG_PROFILE: mean GPU time of one scoring-pass
                               kernel of this call, HIP events around it            */
  uint32_t pass_kind;       /* which streaming pass scored the last iteration:
                               0 smpc_pass (wave per rollout), 1 smpc_pass_lane     */
} smpc_tick_out;

typedef struct smpc_ctx smpc_ctx;

/* ---- lifetime ---------------------------------------------------------- */

/* Fill *cfg with Optimizer::getParams() defaults [ref src/optimizer.cpp:69-82]
 * (motion model Omni, device -1, flags 0). */
void smpc_config_default(smpc_config* cfg);

/* Fill *p with every critic's initialize() defaults, all five enabled. */
void smpc_critic_params_default(smpc_critic_params* p);

/* Replaces Optimizer::initialize()'s allocation half + reset()
 * [ref src/optimizer.cpp:35-55,116-132].  Control-sequence state is NOT kept
 * in the ctx: u travels through smpc_optimize(u_inout), as control_sequence_
 * stays a host member in the reference. */
int smpc_create(const smpc_config* cfg, smpc_ctx** out);
void smpc_destroy(smpc_ctx* ctx);

/* Text of the last error on this ctx (never NULL).  ctx may be NULL for the
 * last smpc_create() failure of the calling thread. */
const char* smpc_last_error(const smpc_ctx* ctx);

/* ABI version and build string. */
int smpc_abi_version(void);
const char* smpc_build_info(void);

/* ---- configuration ------------------------------------------------------ */

/* NoiseGenerator::reset() half of Optimizer::reset()
 * [ref src/noise_generator.cpp:76-95]: in device-RNG mode re-draws the noise
 * (next draw epoch); in supplied-noise mode keeps the arrays given to
 * smpc_set_noise().  Also drops cached per-tick state. */
int smpc_reset(smpc_ctx* ctx);

/* Current (speed-limited) constraints used by
 * applyControlSequenceConstraints() [ref src/optimizer.cpp:237-249,428-453]. */
int smpc_set_constraints(smpc_ctx* ctx, float vx_max, float vx_min, float vy_max,
                         float wz_max);

int smpc_set_critics(smpc_ctx* ctx, const smpc_critic_params* p);

/* Replaces the Costmap2D* the critics hold
 * [ref src/critics/obstacles_critic.cpp:32,203-224, tools/utils.hpp:361-395].
 * cells: uint8[height*width], index my*width+mx (Costmap2D::getCost).
 * inscribed_radius: LayeredCostmap::getInscribedRadius().
 * cost_scaling_factor / inflation_radius: ObstaclesCritic's parameters of the
 * same name, read only when the costmap has an InflationLayer
 * [ref obstacles_critic.cpp:70-80]; pass 0,0 when it has none. */
int smpc_set_costmap(smpc_ctx* ctx, const uint8_t* cells, uint32_t width,
                     uint32_t height, double origin_x, double origin_y,
                     double resolution, int track_unknown, float inscribed_radius,
                     float cost_scaling_factor, float inflation_radius);

/* Costmap hand-off at the controller's rate.  The reference holds a
 * Costmap2D* and reads it under the costmap mutex during the tick
 * [ref src/controller.cpp:99-103]; a device needs its own copy.  smpc_set_costmap() may be
 * called every tick: it keeps a pinned host mirror, uploads only the band of rows that
 * changed since the previous call (nothing for an unchanged map of the same size; the
 * lookup tables are rebuilt only when resolution / inflation parameters change) and
 * returns without waiting for the DMA — the next tick is ordered behind it.
 * A caller that knows the updated window (Costmap2D's updated bounds) can hand over just
 * that: `cells` points at the window's first cell inside the caller's map, whose rows are
 * `row_stride` bytes apart; geometry and parameters stay those of the last
 * smpc_set_costmap(). */
int smpc_update_costmap_region(smpc_ctx* ctx, const uint8_t* cells, uint32_t row_stride,
                               uint32_t x0, uint32_t y0, uint32_t width, uint32_t height);
/* Bytes uploaded by the last smpc_set_costmap / smpc_update_costmap_region call, and in
 * total (either pointer may be NULL). */
int smpc_costmap_upload_bytes(const smpc_ctx* ctx, uint64_t* last_call, uint64_t* total);

/* Supplied-noise (parity) mode: the three [B,T] row-major noise tensors
 * NoiseGenerator holds [ref tools/noise_generator.hpp:97-99], already scaled
 * by the sampling std.  Host pointers; copied. */
/* The robot footprint for consider_footprint = true (ObstaclesCritic, CostCritic): n_points
 * (x, y) pairs in the robot frame (costmap_ros->getRobotFootprint()), the layered costmap's
 * circumscribed radius, and the inflation layer's own cost_scaling_factor (< 0: the costmap
 * has no inflation layer) — what {Obstacles,Cost}Critic::findCircumscribedCost and
 * FootprintCollisionChecker::footprintCostAtPose consume [ref obstacles_critic.cpp:52-97,
 * 203-224, cost_critic.cpp:62-106,175-201].  At most SMPC_MAX_FOOTPRINT points. */
#define SMPC_MAX_FOOTPRINT 16
int smpc_set_footprint(smpc_ctx* ctx, const double* xy, uint32_t n_points,
                       double circumscribed_radius, double layer_cost_scaling_factor);

int smpc_set_noise(smpc_ctx* ctx, const float* noise_vx, const float* noise_vy,
                   const float* noise_wz);

/* Device-RNG mode: Philox4x32-10 + Box–Muller, draw order vx, wz, vy
 * [ref src/noise_generator.cpp:107-122].  Draw epoch 0 is drawn here; each
 * smpc_reset() draws the next epoch (the reference's first tick also runs on
 * its second draw, optimizer.cpp:52-54). */
int smpc_seed(smpc_ctx* ctx, uint64_t seed);

/* NoiseGenerator::generateNextNoises() with regenerate_noises = true
 * [ref src/noise_generator.cpp:54-63,97-105]: draw the next epoch's noise now
 * (device-RNG mode only; constraints and costs are left alone). */
int smpc_redraw_noise(smpc_ctx* ctx);
/* The same draw OFF the tick's critical path, as the reference runs it (a thread draws the next
 * tensors while the optimizer goes on, noise_generator.cpp:54-63,97-105): the next epoch is drawn
 * into a second set of tensors on a stream of its own and the call returns at once; the next
 * tick (smpc_optimize, smpc_shard_begin / _tick, smpc_group_optimize) waits for the draw on the
 * device and scores with it.  A second call before a tick has taken the first draw is a no-op;
 * smpc_seed / smpc_reset / smpc_set_noise / smpc_redraw_noise drop a draw not yet taken.  Doubles
 * the noise tensors' memory (allocated at the first call). */
int smpc_redraw_noise_async(smpc_ctx* ctx);

/* Copy the ctx's noise tensors back (tests / RNG parity). Any pointer may be NULL. */
int smpc_get_noise(smpc_ctx* ctx, float* noise_vx, float* noise_vy, float* noise_wz);

/* ---- the hot path -------------------------------------------------------- */

/* Replaces the body of Optimizer::optimize() [ref src/optimizer.cpp:157-164]:
 * iteration_count x { generateNoisedTrajectories (:227-233),
 * CriticManager::evalTrajectoriesScores (critic_manager.cpp:67-76),
 * updateControlSequence (:362-394) }, costs_ zeroed once per call as
 * prepare() does (:197).
 * u_inout: control_sequence_ as 3*T floats {vx[T], vy[T], wz[T]}, host memory. */
int smpc_optimize(smpc_ctx* ctx, const smpc_tick_in* in, float* u_inout,
                  smpc_tick_out* out);

/* Optimizer::getGeneratedTrajectories() [ref src/optimizer.cpp:455-458]:
 * x, y, yaws [B,T] of the last scoring pass.  Needs
 * SMPC_FLAG_STORE_TRAJECTORIES.  Any pointer may be NULL. */
int smpc_get_trajectories(smpc_ctx* ctx, float* x, float* y, float* yaws);

/* costs_ [B] after the last iteration (incl. the gamma terms of
 * updateControlSequence) [ref include/.../optimizer.hpp:255]. */
int smpc_get_costs(smpc_ctx* ctx, float* costs);

/* Diagnostics: the device sin/cos the rollout uses (integrateStateVelocities,
 * ref src/optimizer.cpp:326-329), evaluated on n host values; tests pin its error. */
int smpc_selftest_sincos(smpc_ctx* ctx, const float* x, uint32_t n, float* sin_out,
                         float* cos_out);
/* Diagnostics: the in-register 64 x 64 transpose-reduce of the lane-per-rollout pass
 * (updateControlSequence's weighted sum, ref src/optimizer.cpp:382-393):
 * out[t] = sum_b w[b] * v[b*64 + t], b, t = 0..63. */
int smpc_selftest_lane_reduce(smpc_ctx* ctx, const float* v, const float* w, float* out);

/* ---- batch-sharded path; one ctx per GPU ------------------
 * A tick on G shards is
 *   begin -> [furthest -> MAX over ranks] -> score -> all-gather -> combine
 * where the bracketed exchange is skipped when the caller speculates on the
 * furthest point (score() reports the true local value in its tuple and the
 * caller re-scores on a miss).  The two exchanges are done by the caller
 * (RCCL through torch.distributed in bench.py); the library never blocks on a
 * peer.  All device pointers must be on this ctx's device and all work is
 * enqueued on the stream given to smpc_set_stream(). */

/* hipStream_t to enqueue on.  NULL is HIP's default (null) stream — what
 * torch.cuda.current_stream() is unless the caller changed it;
 * SMPC_STREAM_OWN goes back to the ctx's own non-blocking stream. */
#define SMPC_STREAM_OWN ((void*)(intptr_t)-1)
int smpc_set_stream(smpc_ctx* ctx, void* hip_stream);

/* Turn SMPC_FLAG_PROFILE on/off after creation (event records cost ~15 us of host and
 * queue time per tick, so throughput is measured with it off). */
int smpc_set_profile(smpc_ctx* ctx, int enable);

#define SMPC_TUPLE_HEADER 4 /* floats before U: min, sum_w, furthest, non_colliding */
/* Length in floats of one shard tuple: SMPC_TUPLE_HEADER + 3*T. */
uint32_t smpc_tuple_len(const smpc_ctx* ctx);

/* (The sharded tick — these phases and smpc_shard_tick — runs ONE iteration per tick: a ctx with
 * iteration_count != 1 is refused with SMPC_ERR_UNSUPPORTED, as is consider_footprint with both
 * collision critics in the list.)
 * Upload the tick's inputs and control sequence (host pointers).  Everything the later phases
 * of the tick need (smpc_shard_combine feeds the path and the pose to the furthest-point
 * predictor) is copied before this returns: *in, its arrays and u_in are the caller's again. */
int smpc_shard_begin(smpc_ctx* ctx, const smpc_tick_in* in, const float* u_in);
/* The furthest reached path point this tick is expected to have: the previous tick's value
 * (as smpc_shard_combine saw it) carried forward by the robot's motion, the plan's pruning and
 * the last tick's drift — what smpc_optimize and smpc_shard_tick speculate with.  Returns 1 and
 * writes *hint when there is a prediction (smpc_shard_begin of this tick has run and an
 * earlier tick's smpc_shard_combine reported a furthest point), else 0.  Every rank computes
 * the same value from the same inputs.  Scoring with it is exact either way: the combined
 * tuple reports the true value and a caller that finds a mismatch scores again. */
int smpc_shard_predicted_furthest(smpc_ctx* ctx, uint32_t* hint);
/* Local max over this shard of the nearest-path-point index of each rollout's
 * endpoint [ref tools/utils.hpp:292-319], written as one float to d_furthest. */
int smpc_shard_furthest(smpc_ctx* ctx, float* d_furthest);
/* Score this shard with the batch-wide furthest point read from d_furthest
 * (device, one float) or, if d_furthest is NULL, furthest_hint; writes the
 * shard tuple {min cost, sum w, true local furthest, non-colliding count,
 * sum w*cvx[T], sum w*cvy[T], sum w*cwz[T]} to d_tuple (device). */
int smpc_shard_score(smpc_ctx* ctx, const float* d_furthest, uint32_t furthest_hint,
                     float* d_tuple);
/* After smpc_shard_combine() reported fail_flag = 1 on a tick that did not
 * start with fail_flag_in: the reference scored no critic past Obstacles
 * [ref critic_manager.cpp:70-73].  Re-score this shard that way into d_tuple;
 * the caller gathers and combines again. */
int smpc_shard_rescore_failed(smpc_ctx* ctx, float* d_tuple);
/* Combine n_tuples shard tuples (device, contiguous) into the new control
 * sequence: rescale by exp(-(min_g - min)/temperature), divide, clip
 * [ref src/optimizer.cpp:382-393]; copies u (3*T) to host and fills *out.
 * Synchronises the stream. */
int smpc_shard_combine(smpc_ctx* ctx, const float* d_tuples, uint32_t n_tuples,
                       float* u_out, smpc_tick_out* out);

/* ---- several planning instances per launch (multi-robot fleets; BASELINE "multi-query") ----
 * Independent contexts (own noise, costmap, critics, control sequence) on one GPU whose
 * ticks are issued together: one upload, one scoring launch with the instance as the second
 * grid dimension, one reduction launch.  Results are exactly those of smpc_optimize on each
 * context; a member that cannot take the batched launch this tick (not on the
 * lane-per-rollout pass — create members with SMPC_FLAG_LANE_PER_ROLLOUT —, first tick without a
 * furthest-point guess, speculation miss, every rollout colliding, iteration_count > 1) is
 * ticked on its own inside the call.  Members must share device and time_steps; while grouped,
 * a context must not be used through smpc_optimize directly; destroy the group before its
 * members.  Replaces N calls of Optimizer::optimize() [ref src/optimizer.cpp:157-164] by
 * N controller instances. */
typedef struct smpc_group smpc_group;
int smpc_group_create(smpc_ctx* const* ctxs, uint32_t n, smpc_group** out);
void smpc_group_destroy(smpc_group* group);
/* ins[n], u_inout[n] (pointers to 3*T floats each), outs[n] (may be NULL) */
int smpc_group_optimize(smpc_group* group, const smpc_tick_in* ins, float* const* u_inout,
                        smpc_tick_out* outs);

/* ---- the same tick with the exchanges inside the library (RCCL over xGMI) ----
 * One ncclComm per ctx, one call per tick: upload -> score -> ncclAllGather(tuples) ->
 * combine -> wait, all on the ctx's stream (speculate != 0: the furthest point of the
 * previous tick is used and the tick is re-scored on a miss; else an
 * ncclAllReduce(MAX) of the furthest point precedes the scoring pass).  RCCL is
 * resolved at run time (dlopen of librccl.so, sharing the copy the process already
 * holds, e.g. torch.distributed's): a single-GPU user never loads it.
 *   rank 0: smpc_shard_comm_id(id, SMPC_COMM_ID_BYTES); ship the bytes to every rank;
 *   every rank: smpc_shard_comm_init(ctx, id, rank, world)  (collective);
 *   per tick:   smpc_shard_tick(ctx, &in, u, &out, 1)       (collective).
 * Replaces, for a sharded deployment, the Optimizer::optimize() call of
 * ref src/optimizer.cpp:133-148. */
#define SMPC_COMM_ID_BYTES 128
int smpc_shard_comm_id(void* id_out, uint32_t id_bytes);
int smpc_shard_comm_init(smpc_ctx* ctx, const void* id, int rank, int world);
int smpc_shard_tick(smpc_ctx* ctx, const smpc_tick_in* in, float* u_inout, smpc_tick_out* out,
                    int speculate);

/* The same tick with NO collective (the exchange is a few hundred bytes per rank: a library
 * collective costs more in latency than it moves).  Every rank owns a mailbox in peer-visible
 * device memory; smpc_shard_p2p_handle() creates it and returns its IPC handle
 * (SMPC_P2P_HANDLE_BYTES), the caller hands every rank the handles of all ranks (rank order,
 * world * SMPC_P2P_HANDLE_BYTES bytes) and smpc_shard_p2p_init() maps them; the ranks must
 * synchronise (any barrier) between that call and their first tick.  From then on
 * smpc_shard_tick() exchanges through the mailboxes: the finishing kernel of a rank writes its
 * tuple into its peers' memory over xGMI and waits for theirs.  The wait is bounded in
 * wall-clock time (default 10 000 ms, smpc_shard_p2p_set_timeout): it has to cover the skew
 * between the ranks' calls, not just the exchange.  A peer that does not answer in time makes
 * the tick fail with SMPC_ERR_DEVICE, it does not hang; the failed rank then stops publishing
 * and every later smpc_shard_tick on it returns SMPC_ERR_STATE, so its peers fail at their
 * next exchange at the latest (they may have completed the tick the failed rank lost: treat
 * any failure as "exchange down on all ranks": smpc_shard_p2p_init again on every rank, then
 * smpc_reset).  At most SMPC_P2P_MAX_WORLD ranks, all on one node; the Omni and DiffDrive
 * models (smpc_shard_p2p_init refuses an Ackermann ctx).  This exchange is an opt-in
 * optimisation; the RCCL exchange above is the default a sharded deployment should start with. */
#define SMPC_P2P_HANDLE_BYTES 64
#define SMPC_P2P_MAX_WORLD 16
int smpc_shard_p2p_handle(smpc_ctx* ctx, void* handle_out, uint32_t handle_bytes);
int smpc_shard_p2p_init(smpc_ctx* ctx, const void* handles, int rank, int world);
int smpc_shard_p2p_set_timeout(smpc_ctx* ctx, uint32_t milliseconds);

#ifdef __cplusplus
}
#endif
#endif /* SMPC_H_ */
