// smpc_shard.cpp — the batch-sharded tick of include/smpc.h: phase-by-phase
// entry points for a driver that owns the collectives, and smpc_shard_tick, which runs the
// whole protocol with RCCL called from here (resolved at run time, never a link dependency).
#include "smpc_ctx.h"

namespace smpc_impl {

const RcclApi* rccl()
{
  static const RcclApi* api = []() -> const RcclApi* {
    void* h = nullptr;
    // SMPC_RCCL_LIB: the library to take the six nccl* entry points from instead of librccl.so
    // (tests/fake_rccl: the world > 1 control flow with several processes on one GPU, which RCCL
    // refuses).  Read once per process; no fall-through when it is set and cannot be loaded.
    if (const char* over = getenv("SMPC_RCCL_LIB")) {
      h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
      if (!h) return nullptr;
    }
    for (const char* name : {"librccl.so", "librccl.so.1"}) {
      if (h) break;
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

}  // namespace smpc_impl

using namespace smpc_impl;

extern "C" {

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
  const int rc = prepare_tick(c, in, u_in);
  if (rc != SMPC_OK) return rc;
  if (c->two_coll_fp)
    return fail(c, SMPC_ERR_UNSUPPORTED,
                "sharded tick: consider_footprint=true with both ObstaclesCritic and CostCritic in the list");
  if (c->cfg.iteration_count != 1)   // (one exchange = one iteration; never silently fewer than asked for)
    return fail(c, SMPC_ERR_UNSUPPORTED, "sharded tick: iteration_count must be 1");
  // What smpc_shard_combine hands to the furthest-point predictor, copied: the caller owns *in
  // and its arrays and may free or reuse them as soon as this call returns (include/smpc.h:
  // "the library copies before returning and never retains host pointers").
  c->step_in = *in;
  c->step_px.assign(in->path_x, in->path_x + in->path_len);
  c->step_py.assign(in->path_y, in->path_y + in->path_len);
  c->step_in.path_x = c->step_px.data();
  c->step_in.path_y = c->step_py.data();
  c->step_in.path_yaw = nullptr;          // (not read after this call)
  c->step_in.path_pts_valid = nullptr;
  c->step_remembered = false;
  return SMPC_OK;
}

int smpc_shard_predicted_furthest(smpc_ctx* c, uint32_t* hint)
{
  if (!c || !hint) return 0;
  if (!c->tick_ready || !c->hint_valid || !(c->gate_flags & SD_NEED_FURTHEST)) return 0;
  *hint = c->hint;
  return 1;
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
  // the predictor's state advances once per tick (a re-scored tick combines twice: same value)
  if ((c->gate_flags & SD_NEED_FURTHEST) && !c->step_remembered) {
    remember_furthest(c, &c->step_in, c->h_out[3 * T + 2]);
    c->step_remembered = true;
  }
  if (out) {
    memset(out, 0, sizeof(*out));
    const bool obstacles_scored = (scoring_flags(c, c->fail_in) & (SD_OBSTACLES | SD_COST)) != 0;
    out->fail_flag = (c->fail_in || (obstacles_scored && c->h_out[3 * T + 3] == 0.0f)) ? 1 : 0;
    out->furthest_valid = (c->gate_flags & SD_NEED_FURTHEST) ? 1 : 0;
    out->furthest_reached_path_point = smpc_furthest_index(c->h_out[3 * T + 2]);
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
  // the last exchange set up is the one smpc_shard_tick uses: drop the mailboxes
  for (uint32_t k = 0; k < c->p2p.world; ++k)
    if (k != c->p2p.rank && c->p2p.peer[k]) (void)hipIpcCloseMemHandle(c->p2p.peer[k]);
  c->p2p = SmpcP2P{};
  c->comm_rank = rank;
  c->comm_world = world;
  if (c->d_all) (void)hipFree(c->d_all);
  c->d_all = nullptr;
  HIPCK(c, hipMalloc(&c->d_all, static_cast<size_t>(world) * (4 + 3 * c->cfg.time_steps) * sizeof(float)));
  return SMPC_OK;
}

// ---- exchange without a collective: mailboxes over IPC / xGMI (smpc_p2p_exchange) ----------
static uint32_t p2p_slot_floats(uint32_t T) {return align_up(4 + 3 * T + 1, 16);}

int smpc_shard_p2p_set_timeout(smpc_ctx* c, uint32_t milliseconds)
{
  if (!c || milliseconds == 0) return fail(c, SMPC_ERR_INVALID, "timeout must be > 0 ms");
  c->p2p_timeout_ms = milliseconds;
  c->p2p.timeout_ticks = static_cast<unsigned long long>(milliseconds) * 100000ull;
  return SMPC_OK;
}

int smpc_shard_p2p_handle(smpc_ctx* c, void* handle_out, uint32_t handle_bytes)
{
  if (!c || !handle_out || handle_bytes < sizeof(hipIpcMemHandle_t))
    return fail(c, SMPC_ERR_INVALID, "handle buffer too small");
  HIPCK(c, hipSetDevice(c->device));
  if (!c->p2p_mailbox) {
    const size_t bytes = 2u * SMPC_P2P_MAX_RANKS * p2p_slot_floats(c->cfg.time_steps) * sizeof(float);
    // fine-grained: stores arriving over xGMI must be visible to this GPU's loads without a
    // cache flush in between
    HIPCK(c, hipExtMallocWithFlags(reinterpret_cast<void**>(&c->p2p_mailbox), bytes, hipDeviceMallocFinegrained));
    HIPCK(c, hipMemset(c->p2p_mailbox, 0, bytes));
    HIPCK(c, hipDeviceSynchronize());
  }
  hipIpcMemHandle_t h;
  HIPCK(c, hipIpcGetMemHandle(&h, c->p2p_mailbox));
  memcpy(handle_out, &h, sizeof(h));
  return SMPC_OK;
}

int smpc_shard_p2p_init(smpc_ctx* c, const void* handles, int rank, int world)
{
  if (!c || !handles || world < 1 || world > SMPC_P2P_MAX_RANKS || rank < 0 || rank >= world)
    return fail(c, SMPC_ERR_INVALID, "bad rank/world");
  if (!c->p2p_mailbox) return fail(c, SMPC_ERR_STATE, "smpc_shard_p2p_handle first");
  // AckermannMotionModel::applyConstraints needs a launch of its own behind the combine
  // (smpc_ackermann_constrain); the exchange kernel is the tick's last launch
  if (c->acker_r >= 0.f)
    return fail(c, SMPC_ERR_UNSUPPORTED, "the mailbox exchange does not carry the Ackermann constraint");
  HIPCK(c, hipSetDevice(c->device));
  HIPCK(c, hipStreamSynchronize(c->stream));   // no exchange of an earlier set-up may still run
  for (uint32_t r = 0; r < c->p2p.world; ++r)
    if (r != c->p2p.rank && c->p2p.peer[r]) (void)hipIpcCloseMemHandle(c->p2p.peer[r]);
  c->p2p = SmpcP2P{};
  for (int r = 0; r < world; ++r) {
    if (r == rank) {
      c->p2p.peer[r] = c->p2p_mailbox;
      continue;
    }
    hipIpcMemHandle_t h;
    memcpy(&h, static_cast<const uint8_t*>(handles) + static_cast<size_t>(r) * sizeof(h), sizeof(h));
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      c->p2p.world = static_cast<uint32_t>(r);   // what has been opened so far is closed with the ctx
      c->p2p.rank = static_cast<uint32_t>(rank);
      return fail(c, SMPC_ERR_DEVICE, std::string("hipIpcOpenMemHandle (rank ") + std::to_string(r) + "): " +
                                     hipGetErrorString(e));
    }
    c->p2p.peer[r] = static_cast<float*>(p);
  }
  c->p2p.world = static_cast<uint32_t>(world);
  c->p2p.rank = static_cast<uint32_t>(rank);
  c->p2p.slot_floats = p2p_slot_floats(c->cfg.time_steps);
  // a fresh numbering of the exchanges: no sequence word of an earlier set-up may survive in
  // the own mailbox (the caller synchronises the ranks between this call and the first tick)
  HIPCK(c, hipMemset(c->p2p_mailbox, 0, 2u * SMPC_P2P_MAX_RANKS * c->p2p.slot_floats * sizeof(float)));
  // the sticky "an exchange timed out" word (device) and its host-visible mark: only this
  // collective set-up clears them
  c->p2p.state = reinterpret_cast<uint32_t*>(c->d_furthest) + 3;
  HIPCK(c, hipMemset(c->p2p.state, 0, sizeof(uint32_t)));
  HIPCK(c, hipDeviceSynchronize());   // (the ctx's stream is non-blocking: not ordered behind the memsets)
  c->h_out[3 * c->cfg.time_steps + 6] = 0.0f;
  c->p2p_failed = false;
  c->p2p.timeout_ticks = static_cast<unsigned long long>(c->p2p_timeout_ms) * 100000ull;   // 100 MHz
  c->p2p_xseq = 0;
  c->comm_rank = rank;
  c->comm_world = world;
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
  const bool p2p = c->p2p.world > 0;
  if (!p2p && !c->comm) return fail(c, SMPC_ERR_STATE, "smpc_shard_comm_init or smpc_shard_p2p_init first");
  if (p2p && c->p2p_failed)
    return fail(c, SMPC_ERR_STATE,
                "an earlier mailbox exchange timed out: this rank no longer publishes (its peers fail at their "
                "next exchange); set the exchange up again on every rank (smpc_shard_p2p_init) and smpc_reset");
  const RcclApi* r = p2p ? nullptr : rccl();
  HIPCK(c, hipSetDevice(c->device));
  c->passes = 0;
  c->evp_used = 0;
  c->costs_cur = 0;
  int rc = prepare_tick(c, in, u_inout);
  if (rc != SMPC_OK) return rc;
  if (c->two_coll_fp)
    return fail(c, SMPC_ERR_UNSUPPORTED,
                "sharded tick: consider_footprint=true with both ObstaclesCritic and CostCritic in the list");
  if (c->cfg.iteration_count != 1)   // (one exchange = one iteration; never silently fewer than asked for)
    return fail(c, SMPC_ERR_UNSUPPORTED, "sharded tick: iteration_count must be 1");
  const uint32_t T = c->cfg.time_steps, TL = 4 + 3 * T, G = static_cast<uint32_t>(c->comm_world);
  auto nccl_ok = [&](ncclResult_t e, const char* what) {
    if (e == ncclSuccess) return SMPC_OK;
    return fail(c, SMPC_ERR_DEVICE, std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(e) : "error"));
  };
  // mailbox exchange: publish the tuple to every peer, wait for theirs, combine — one launch
  auto p2p_exchange = [&](int mode, const float* d_used) -> int {
    SmpcP2P x = c->p2p;
    x.xseq = ++c->p2p_xseq;
    uint32_t seq = 0;
    if (mode == 0 && c->poll_enabled) {
      seq = ++c->seq;
      if (seq == 0) seq = ++c->seq;
      c->poll_seq = seq;
    }
    // (h_out[3T + 6], the kernel's "a peer never answered" mark, is cleared by
    // smpc_shard_p2p_init only: the device may write it at any time after a launch)
    HIPCK(c, smpc_launch_p2p_exchange(c->d_tuple, x, T, mode, c->d_furthest, c->dev.neg_inv_temp, c->c_vx_max,
                                      c->c_vx_min, c->c_vy, c->c_wz, c->d_out, c->d_out + 3 * T, d_used,
                                      c->h_out_dev, seq, c->stream));
    return SMPC_OK;
  };
  // a mode-1 exchange (furthest point) that timed out leaves the device-side state set: the
  // mode-0 exchange of the same tick then publishes nothing and reports here
  auto p2p_check = [&]() -> int {
    if (c->h_out[3 * T + 6] == 0.0f) return SMPC_OK;
    c->p2p_failed = true;
    c->hint_valid = false;
    char msg[160];
    snprintf(msg, sizeof(msg), "shard exchange %u: a peer's tuple did not arrive within %u ms", c->p2p_xseq,
             c->p2p_timeout_ms);
    return fail(c, SMPC_ERR_DEVICE, msg);
  };
  auto gather_combine_fetch = [&](const float* d_used) -> int {
    if (p2p) {
      int e = p2p_exchange(0, d_used);
      if (e != SMPC_OK) return e;
      e = fetch_out(c);
      return e != SMPC_OK ? e : p2p_check();
    }
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
    float F_true = c->h_out[3 * T + 2];
    const uint32_t S_true = smpc_furthest_index(F_true);
    if (S_true != c->hint) {
      // miss: the gathered tuples carry the true batch-wide furthest point
      c->spec_misses++;
      c->hint = S_true;
      rc = launch_score(c, flags, nullptr, nullptr, S_true, c->d_tuple);
      if (rc == SMPC_OK) rc = gather_combine_fetch(nullptr);
      if (rc != SMPC_OK) return rc;
      F_true = c->h_out[3 * T + 2];
    }
    remember_furthest(c, in, F_true);
  } else {
    if (need_f) {
      rc = launch_furthest(c, c->d_furthest);
      if (rc != SMPC_OK) return rc;
      if (p2p) {
        // the local furthest point travels in field [2] of an otherwise empty tuple
        HIPCK(c, hipMemcpyAsync(c->d_tuple + 2, c->d_furthest, sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        rc = p2p_exchange(1, nullptr);
      } else {
        rc = nccl_ok(r->AllReduce(c->d_furthest, c->d_furthest, 1, ncclFloat32, ncclMax, c->comm, c->stream),
                     "ncclAllReduce");
      }
      if (rc != SMPC_OK) return rc;
    }
    rc = launch_score(c, flags, nullptr, need_f ? c->d_furthest : nullptr, 0, c->d_tuple);
    if (rc == SMPC_OK) rc = gather_combine_fetch(need_f ? c->d_furthest : nullptr);
    if (rc != SMPC_OK) return rc;
    if (need_f) remember_furthest(c, in, c->h_out[3 * T + 2]);
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
}  // extern "C"
