// smpc_group.cpp — several planning instances per launch (include/smpc.h smpc_group_*;
// BASELINE configs[4]: multi-robot fleets): one upload, one scoring launch, one reduction for a
// set of contexts.
#include "smpc_ctx.h"

using namespace smpc_impl;

extern "C" {

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
    c->in_group = true;
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
    c->in_group = false;
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
  bool obst = false, dep = false;
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
    // (a near-goal member scores GoalAngle: instances of the lane pass the batched launch does not
    // have; members with the deployed critic list — Constraint / Cost / Twirling — have theirs)
    if (flags[i] & (SD_CONSTRAINT | SD_COST | SD_TWIRLING)) dep = true;
    const bool ok = c->lane_now && !c->lane_rr && !(flags[i] & SD_GOAL_ANGLE) &&
      c->cfg.iteration_count == 1 && !c->fail_in &&
      !(c->cfg.flags & (SMPC_FLAG_NO_SPECULATION | SMPC_FLAG_PROFILE)) && (!need_f || c->hint_valid) &&
      c->poll_enabled && c->acker_r < 0.f && !c->two_coll_fp;
    if (!ok) batched = false;
    if (i == 0) {
      window_bytes = c->lane_window_bytes;
      obst = (flags[i] & (SD_OBSTACLES | SD_COST)) != 0;    // (either one: the costmap lookups)
    } else if (c->lane_window_bytes != window_bytes || ((flags[i] & (SD_OBSTACLES | SD_COST)) != 0) != obst ||
               c->lane_block != c0->lane_block) {
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
    hr[i].neg_inv_temp = c->dev.neg_inv_temp;   // (its own: the members share the horizon, nothing else)
    fin.u_host = nullptr;          // one publishing block for the whole group instead
    fin.done_counter = nullptr;
    hr[i].fin = fin;
    c->passes++;
    c->last_pass_kind = 1;
  }
  // one hand-over for every member's tick block and parameter block: CPU stores through the BAR
  // (every member's last tick was fetched: nothing reads the buffer), else a copy on the stream
  if (c0->bar_tick) {
    // (only what this tick wrote: a slot is sized for the longest path, 25 KB, a tick fills ~4)
    for (uint32_t i = 0; i < n; ++i) bar_copy(g->d_all + g->slot * i, g->h_all + g->slot * i, g->ctxs[i]->tick_used);
    bar_copy(g->d_all + g->off_dev, g->h_all + g->off_dev, g->total - g->off_dev);
    bar_flush(c0);
  }
  else HIPCK(c0, hipMemcpyAsync(g->d_all, g->h_all, g->total, hipMemcpyHostToDevice, g->stream));
  HIPCK(c0, smpc_launch_pass_lane_many(reinterpret_cast<const SmpcDev*>(g->d_all + g->off_dev), n,
                                       T == 64, obst, dep, T, L, gridx, c0->lane_block, g->stream));
  HIPCK(c0, smpc_launch_reduce_many(reinterpret_cast<const SmpcReduceArgs*>(g->d_all + g->off_red), n,
                                    T, g->stream));
  g->batched_ticks++;
  const auto t_launch = now();
  // ---- per member: wait, verify the speculation and the collision count --------------------
  for (uint32_t i = 0; i < n; ++i) {
    smpc_ctx* c = g->ctxs[i];
    int rc = fetch_out(c);
    if (rc != SMPC_OK) return rc;
    const bool need_f = (flags[i] & SD_NEED_FURTHEST) != 0;
    const float F_true = c->h_out[3 * T + 2];
    const uint32_t S_true = smpc_furthest_index(F_true);
    const bool miss = need_f && S_true != c->hint;
    const bool all_collide = (flags[i] & (SD_OBSTACLES | SD_COST)) && c->h_out[3 * T + 3] == 0.0f;
    if (miss || all_collide) {
      if (miss) {
        c->spec_misses++;
        remember_furthest(c, &ins[i], F_true);   // smpc_optimize speculates with the true value now: one pass
        c->hint_F = F_true;                      // (not the smoothed estimate a fresh-noise tick leaves)
        c->hint_is_this_ticks = true;
      }
      rc = single(i);
      if (rc != SMPC_OK) return rc;
      if (outs) outs[i].passes++;    // the batched scoring this member took part in counts
      continue;
    }
    if (need_f) remember_furthest(c, &ins[i], F_true);   // as smpc_optimize does: the predictor's state advances every tick
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
