// smpc_api.cpp — host side of the C-ABI declared in include/smpc.h.
//
// Owns device memory, the per-tick upload block and the launch sequence; does
// the O(P) host work the reference's critics do once per tick (goal-distance
// gates, path validity, cumulative path lengths, the per-candidate PathAlign /
// PathFollow tables) and the 256-entry Obstacles lookup table.  All [B,T] work
// is in smpc_kernels.hip.  There is no CPU fallback: without a HIP device
// smpc_create() fails.

#include "smpc_ctx.h"

#include <emmintrin.h>

namespace smpc_impl {

thread_local std::string g_create_error;

int fail(smpc_ctx* c, int code, const std::string& msg)
{
  if (c) c->err = msg; else g_create_error = msg;
  return code;
}

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

void bar_copy(void* dev_dst, const void* host_src, size_t n)
{
  // The BAR mapping is write-combining: 16-byte streaming stores fill whole 64-byte buffers,
  // and the fence drains them before any later store of this thread — the runtime's doorbell
  // write for the launch that reads the block included (PCIe keeps posted writes to one device
  // in order).  The kernel boundary in front of that launch invalidates the GPU's caches.
  const __m128i* src = static_cast<const __m128i*>(host_src);
  __m128i* dst = static_cast<__m128i*>(dev_dst);
  for (size_t i = 0; i < n / 16; ++i) _mm_stream_si128(dst + i, _mm_load_si128(src + i));
  _mm_sfence();
}

void bar_flush(const smpc_ctx* c)
{
  // one posted MMIO write, ordered behind the drained stores and in front of the doorbell
  if (c->hdp_flush) *c->hdp_flush = 1u;
}

// HDP_MEM_FLUSH_CNTL of HIP device `dev` (matched by PCI location), through the HSA runtime the
// HIP runtime already holds; null when it cannot be found
static volatile uint32_t* find_hdp_flush(int dev)
{
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return nullptr;
  void* h = dlopen("libhsa-runtime64.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("libhsa-runtime64.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) return nullptr;
  using agent_t = struct {uint64_t handle;};
  using iterate_fn = int (*)(int (*)(agent_t, void*), void*);
  using info_fn = int (*)(agent_t, int, void*);
  using init_fn = int (*)();
  static init_fn p_init = reinterpret_cast<init_fn>(dlsym(h, "hsa_init"));
  static iterate_fn p_iter = reinterpret_cast<iterate_fn>(dlsym(h, "hsa_iterate_agents"));
  static info_fn p_info = reinterpret_cast<info_fn>(dlsym(h, "hsa_agent_get_info"));
  if (!p_init || !p_iter || !p_info || p_init() != 0) return nullptr;   // (reference-counted; HIP holds it already)
  struct Want {
    uint32_t domain, bdf;
    volatile uint32_t* reg;
  } want{static_cast<uint32_t>(prop.pciDomainID),
         (static_cast<uint32_t>(prop.pciBusID) << 8) | (static_cast<uint32_t>(prop.pciDeviceID) << 3), nullptr};
  auto cb = [](agent_t a, void* data) -> int {
    Want* w = static_cast<Want*>(data);
    int type = 0;   // HSA_AGENT_INFO_DEVICE = 17, HSA_DEVICE_TYPE_GPU = 1
    if (p_info(a, 17, &type) != 0 || type != 1) return 0;
    uint32_t bdf = 0, domain = 0;
    if (p_info(a, 0xA006, &bdf) != 0) return 0;              // HSA_AMD_AGENT_INFO_BDFID
    if (p_info(a, 0xA00F, &domain) != 0) domain = w->domain;   // HSA_AMD_AGENT_INFO_DOMAIN
    if ((bdf & ~7u) != w->bdf || domain != w->domain) return 0;
    struct {uint32_t* mem; uint32_t* reg;} f{nullptr, nullptr};
    if (p_info(a, 0xA00E, &f) == 0) w->reg = f.mem;          // HSA_AMD_AGENT_INFO_HDP_FLUSH
    return 1;   // (HSA_STATUS_INFO_BREAK: stop iterating)
  };
  (void)p_iter(cb, &want);
  return want.reg;
}

void free_ctx(smpc_ctx* c)
{
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (float* p : {c->d_tvx, c->d_nvx, c->d_nvy, c->d_nwz, c->d_costs[0], c->d_costs[1], c->d_traj[0],
         c->d_traj[1], c->d_traj[2], c->d_partials, c->d_tuple, c->d_out, c->d_u_iter, c->d_furthest})
    if (p) (void)hipFree(p);
  if (c->fill_stream) {
    (void)hipStreamSynchronize(c->fill_stream);
    (void)hipStreamDestroy(c->fill_stream);
  }
  if (c->ev_fill) (void)hipEventDestroy(c->ev_fill);
  for (float* p : {c->b_tvx, c->b_nvx, c->b_nvy, c->b_nwz})
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
  for (uint32_t r = 0; r < c->p2p.world; ++r)
    if (r != c->p2p.rank && c->p2p.peer[r]) (void)hipIpcCloseMemHandle(c->p2p.peer[r]);
  if (c->p2p_mailbox) (void)hipFree(c->p2p_mailbox);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  for (hipEvent_t e : c->evp) if (e) (void)hipEventDestroy(e);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int launch_furthest(smpc_ctx* c, float* d_furthest)
{
  {
    const int rc = ensure_row_major(c);   // the wave-per-rollout pass reads the [B,T] tensors
    if (rc != SMPC_OK) return rc;
  }
  HIPCK(c, hipMemsetAsync(d_furthest, 0, sizeof(float), c->stream));
  SmpcDev d = c->dev;
  d.flags = c->gate_flags & SD_NEED_FURTHEST;
  d.furthest_out = reinterpret_cast<uint32_t*>(d_furthest);
  SmpcLds L = make_lds(0, c->P, d.T, (pass_block(c->R) / 64), false);
  c->launched = true;
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
  if (u_dev) {        // a later iteration: the control sequence the previous one left on the device
    d.u = u_dev;
    d.u_inline = 0;
  }
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
                 uint32_t furthest_hint, float* d_tuple, bool finish, const float* finish_furthest)
{
  SmpcDev d;
  SmpcFinal fin;
  fill_score_args(c, flags, u_dev, d_furthest, furthest_hint, finish, finish_furthest, d, fin);
  const bool prof = (c->cfg.flags & SMPC_FLAG_PROFILE) && c->evp_used + 2 <= 8;
  if (prof) HIPCK(c, hipEventRecord(c->evp[c->evp_used], c->stream));
  uint32_t nblk = c->grid;
  // The re-read form has instances with a collision critic scored only.  A tick whose flags were
  // stripped after the launch was planned — fail_flag_in (the retry after fallback(), which scores
  // nothing: critic_manager.cpp:70-73) or the later iterations of an all-collide tick — takes the
  // wave-per-rollout pass instead (its geometry is planned for every tick).
  const bool lane = c->lane_now && !(flags & SD_STORE_TRAJ) &&
    !(c->lane_rr && !(flags & (SD_OBSTACLES | SD_COST)));
  // (the split form scores the plain five with ObstaclesCritic on: a tick whose flags were stripped takes the lane pass)
  const bool split = lane && c->split_now && (flags & SD_OBSTACLES) && !(flags & (SD_GOAL_ANGLE | SD_EXTRA_CRITICS));
  if (lane) nblk = split ? c->grid_split : c->grid_tpr;
  // The block that finishes last reduces the partials inside the scoring launch (smpc_tail.h);
  // larger grids, and launches whose LDS was not sized for it, take the separate reduction.
  const bool tail = !split && c->fused_reduce && nblk <= SMPC_TAIL_MAX_GRID && (!lane || c->lane_block == smpc_lane_block()) &&
    (lane ? c->lds_tpr.total : c->lds.total) >= smpc_tail_lds_bytes(d.T);
  // completion words: one per block of smpc_reduce_partials, or per reducing block of the tail
  c->poll_words = (fin.enabled && fin.done_counter) ? (4u + 3u * d.T + 31u) / 32u : 0u;
  static_assert((4u + 3u * 64u * SMPC_MAX_R + 31u) / 32u <= kPollWords, "one completion word per reducing block");
  if (tail) {
    // (every reducing block publishes its own completion word: smpc_tail.h)
    if (fin.enabled && fin.done_counter) c->poll_words = std::min((4u + 3u * d.T + 63u) / 64u, nblk);
    // the tail's "gave up waiting" mark of an earlier tick must not fail this one (no launch of
    // this ctx is in flight here: every tick ends with fetch_out); the mailbox exchange's mark
    // in the same word is sticky by design and stays
    if (c->h_out[3 * d.T + 6] == 2.0f) c->h_out[3 * d.T + 6] = 0.0f;
    d.tail = 1;
    d.tail_counter = reinterpret_cast<uint32_t*>(c->d_furthest) + 4;
    d.tuple = d_tuple;
    d.fin = fin;
  }
  c->launched = true;
  // the lane-per-rollout pass scores with the full lean critic stack only
  if (split) {
    HIPCK(c, smpc_launch_pass_split(d, c->lds_split, nblk, c->split_nseg, c->stream));
    c->last_pass_kind = 2;
  } else if (lane) {
    HIPCK(c, smpc_launch_pass_lane(d, c->lds_tpr, nblk, c->lane_rr, c->lane_block, c->stream));
    c->last_pass_kind = 1;
  } else {
    c->last_pass_kind = 0;
    const int rc_rm = ensure_row_major(c);
    if (rc_rm != SMPC_OK) return rc_rm;
    HIPCK(c, smpc_launch_pass(c->R, c->score_mode, d, c->lds, c->grid, pass_block(c->R), c->stream));
  }
  if (prof) {
    HIPCK(c, hipEventRecord(c->evp[c->evp_used + 1], c->stream));
    c->evp_used += 2;
  }
  if (!tail) HIPCK(c, smpc_launch_reduce(c->d_partials, nblk, d.T, d.neg_inv_temp, d_tuple, fin, c->stream));
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
    c->poll_words = 0;   // (one word, written by the combining block)
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

// the pass's echo of the tick block's number (SmpcDev::canary): anything but this tick's number
// means the kernels did not read the block the host handed over
static int check_canary(smpc_ctx* c)
{
  c->launched = false;
  if (!c->canary_expect) return SMPC_OK;
  const uint32_t got = reinterpret_cast<const volatile uint32_t*>(c->h_out)[3 * c->cfg.time_steps + 5];
  if (got == c->canary_expect) return SMPC_OK;
  char msg[200];
  snprintf(msg, sizeof(msg), "the scoring pass read tick block %u, not %u: the per-tick upload did not reach the "
           "device in time (CPU stores through the BAR: %s); SMPC_NO_BAR_TICK=1 selects the copy", got, c->canary_expect,
           c->bar_tick ? "on" : "off");
  c->bar_tick = false;   // later ticks of this ctx take the copy
  return fail(c, SMPC_ERR_DEVICE, msg);
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
    const uint32_t nwords = c->poll_words;   // > 0: one word per reducing block, from flag[1] on
    c->poll_words = 0;
    auto arrived = [&]() {
      if (nwords == 0) return *flag == want;
      for (uint32_t r = 0; r < nwords; ++r)
        if (flag[1 + r] != want) return false;
      return true;
    };
    for (uint32_t spin = 0; spin < 4000000u; ++spin) {
      if (arrived()) {
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        // smpc_grid_tail's mark: a reducing block gave up waiting for the grid's other blocks
        if (c->h_out[3 * c->cfg.time_steps + 6] == 2.0f)
          return fail(c, SMPC_ERR_DEVICE, "the scoring launch's reduction did not see every block's partial");
        return check_canary(c);
      }
      __builtin_ia32_pause();
    }
  }
  HIPCK(c, hipStreamSynchronize(c->stream));
  return check_canary(c);
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
  // (with its consider_footprint switch: the re-score has to find the same collisions)
  const uint32_t coll = (c->gate_flags & SD_COST) ? (SD_COST | SD_FP_COST) : (SD_OBSTACLES | SD_FP_OBSTACLES);
  return c->gate_flags & (SD_CONSTRAINT | coll | SD_STORE_TRAJ | SD_TRACK_UNKNOWN);
}

uint32_t scoring_flags(const smpc_ctx* c, bool fail_sticky)
{
  // CriticManager::evalTrajectoriesScores breaks on fail_flag (critic_manager.cpp:70-73)
  const uint32_t keep = SD_STORE_TRAJ | SD_TRACK_UNKNOWN;
  return fail_sticky ? (c->gate_flags & keep) : c->gate_flags;
}

// keep the group-major copies (smpc_dev.h: SMPC_GM_INDEX) in step with the [B,T] tensors
int update_time_major(smpc_ctx* c)
{
  if (!c->use_tpr) return SMPC_OK;
  const uint32_t B = c->cfg.batch_size, T = c->cfg.time_steps;
  HIPCK(c, smpc_launch_relayout(c->d_nvx, c->d_tvx, B, T, true, c->stream));
  HIPCK(c, smpc_launch_relayout(c->d_nvy, c->d_tvy, B, T, true, c->stream));
  HIPCK(c, smpc_launch_relayout(c->d_nwz, c->d_twz, B, T, true, c->stream));
  return SMPC_OK;
}

// the [B,T] tensors (wave-per-rollout pass, smpc_get_noise) from the group-major copy, when a
// device-RNG draw filled only that one
int ensure_row_major(smpc_ctx* c)
{
  if (c->rm_valid) return SMPC_OK;
  const uint32_t B = c->cfg.batch_size, T = c->cfg.time_steps;
  HIPCK(c, smpc_launch_relayout(c->d_tvx, c->d_nvx, B, T, false, c->stream));
  if (c->holonomic) HIPCK(c, smpc_launch_relayout(c->d_tvy, c->d_nvy, B, T, false, c->stream));
  HIPCK(c, smpc_launch_relayout(c->d_twz, c->d_nwz, B, T, false, c->stream));
  c->rm_valid = true;
  return SMPC_OK;
}

// One epoch of the device RNG into the given set of tensors, on stream st (no wait).
// rm_valid_out: whether the [B,T] tensors of the set hold the draw (else only the group-major ones)
static int launch_draw(smpc_ctx* c, float* nvx, float* nvy, float* nwz, float* tvx, float* tvy, float* twz,
                       hipStream_t st, bool* rm_valid_out)
{
  const uint64_t n = static_cast<uint64_t>(c->cfg.batch_size) * c->cfg.time_steps;
  const uint64_t base = c->cfg.shard_offset * c->cfg.time_steps;
  const uint32_t B = c->cfg.batch_size, T = c->cfg.time_steps;
  if (c->use_tpr && (T & 3u) == 0 && !getenv("SMPC_NO_FUSED_FILL")) {
    // lane-per-rollout contexts: draw straight into the group-major layout that pass reads —
    // one write of the noise instead of a write, a read and a second write (fill + transpose);
    // the [B,T] copy is made only if something asks for it (ensure_row_major)
    HIPCK(c, smpc_launch_fill_noise_tm(tvx, B, T, base, c->seed, 0, c->epoch, c->cfg.vx_std, st));
    HIPCK(c, smpc_launch_fill_noise_tm(twz, B, T, base, c->seed, 1, c->epoch, c->cfg.wz_std, st));
    if (c->holonomic)
      HIPCK(c, smpc_launch_fill_noise_tm(tvy, B, T, base, c->seed, 2, c->epoch, c->cfg.vy_std, st));
    else
      HIPCK(c, hipMemsetAsync(tvy, 0, static_cast<size_t>(SMPC_GM_ROLLOUTS(B)) * T * sizeof(float), st));
    *rm_valid_out = false;
    return SMPC_OK;
  }
  // draw order vx, wz, vy (noise_generator.cpp:107-122)
  HIPCK(c, smpc_launch_fill_noise(nvx, n, base, c->seed, 0, c->epoch, c->cfg.vx_std, st));
  HIPCK(c, smpc_launch_fill_noise(nwz, n, base, c->seed, 1, c->epoch, c->cfg.wz_std, st));
  // noises_vy_ keeps its zeros for a non-holonomic model (noise_generator.cpp:117-121)
  if (c->holonomic) HIPCK(c, smpc_launch_fill_noise(nvy, n, base, c->seed, 2, c->epoch, c->cfg.vy_std, st));
  if (c->use_tpr) {
    HIPCK(c, smpc_launch_relayout(nvx, tvx, B, T, true, st));
    HIPCK(c, smpc_launch_relayout(nvy, tvy, B, T, true, st));
    HIPCK(c, smpc_launch_relayout(nwz, twz, B, T, true, st));
  }
  *rm_valid_out = true;
  return SMPC_OK;
}

// a draw requested with smpc_redraw_noise_async that no tick has taken yet is dropped (whoever
// calls this is about to overwrite the noise anyway)
int cancel_redraw(smpc_ctx* c)
{
  if (!c->redraw_pending) return SMPC_OK;
  HIPCK(c, hipEventSynchronize(c->ev_fill));
  c->redraw_pending = false;
  return SMPC_OK;
}

// The tick about to be prepared takes the noise drawn in the background, if there is any: its
// stream waits for the draw ON THE DEVICE and the two sets of tensors change places.
int absorb_redraw(smpc_ctx* c)
{
  if (!c->redraw_pending) return SMPC_OK;
  HIPCK(c, hipStreamWaitEvent(c->stream, c->ev_fill, 0));
  std::swap(c->d_nvx, c->b_nvx);
  std::swap(c->d_nvy, c->b_nvy);
  std::swap(c->d_nwz, c->b_nwz);
  std::swap(c->d_tvx, c->b_tvx);
  std::swap(c->d_tvy, c->b_tvy);
  std::swap(c->d_twz, c->b_twz);
  c->rm_valid = c->redraw_rm_valid;
  c->redraw_pending = false;
  c->noise_gen++;
  return SMPC_OK;
}

int draw_noise(smpc_ctx* c)
{
  int rc = cancel_redraw(c);
  if (rc != SMPC_OK) return rc;
  bool rm = true;
  rc = launch_draw(c, c->d_nvx, c->d_nvy, c->d_nwz, c->d_tvx, c->d_tvy, c->d_twz, c->stream, &rm);
  if (rc != SMPC_OK) return rc;
  HIPCK(c, hipStreamSynchronize(c->stream));
  c->have_noise = true;
  c->rm_valid = rm;
  c->noise_gen++;
  return SMPC_OK;
}

// NoiseGenerator::generateNextNoises as the reference runs it (noise_generator.cpp:54-63,97-105:
// a thread draws the next tensors while the optimizer goes on): the next epoch goes into the
// OTHER set of tensors on a stream of its own, and the call returns at once.
int redraw_async(smpc_ctx* c)
{
  if (c->redraw_pending) return SMPC_OK;   // (the reference's signal is not counted either)
  const size_t n = static_cast<size_t>(c->cfg.batch_size) * c->cfg.time_steps * sizeof(float);
  if (!c->fill_stream) {
    HIPCK(c, hipStreamCreateWithFlags(&c->fill_stream, hipStreamNonBlocking));
    HIPCK(c, hipEventCreateWithFlags(&c->ev_fill, hipEventDisableTiming));
  }
  if (c->use_tpr && !c->b_tvx) {
    const size_t ngm = static_cast<size_t>(SMPC_GM_ROLLOUTS(c->cfg.batch_size)) * c->cfg.time_steps * sizeof(float);
    HIPCK(c, hipMalloc(&c->b_tvx, 3 * ngm));
    c->b_tvy = c->b_tvx + ngm / sizeof(float);
    c->b_twz = c->b_tvy + ngm / sizeof(float);
  }
  if (!c->b_nvx) {
    HIPCK(c, hipMalloc(&c->b_nvx, n));
    HIPCK(c, hipMalloc(&c->b_nvy, n));
    HIPCK(c, hipMalloc(&c->b_nwz, n));
    if (!c->holonomic) HIPCK(c, hipMemset(c->b_nvy, 0, n));
  }
  c->epoch++;
  bool rm = true;
  const int rc = launch_draw(c, c->b_nvx, c->b_nvy, c->b_nwz, c->b_tvx, c->b_tvy, c->b_twz, c->fill_stream, &rm);
  if (rc != SMPC_OK) return rc;
  HIPCK(c, hipEventRecord(c->ev_fill, c->fill_stream));
  c->redraw_rm_valid = rm;
  c->redraw_pending = true;
  return SMPC_OK;
}
}  // namespace smpc_impl

using namespace smpc_impl;

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
  p->path_align_legacy = {0, 0, 1, 10.0f, 0.07f, 20, 4, 0.5f};   // path_align_legacy_critic.cpp:26-37
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
  {
    int large_bar = 0;
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev) != hipSuccess) large_bar = 0;
    c->bar_tick = large_bar != 0 && getenv("SMPC_NO_BAR_TICK") == nullptr;
    if (c->bar_tick && !getenv("SMPC_NO_HDP_FLUSH")) {
      c->hdp_flush = find_hdp_flush(dev);
      // without the device's HDP flush register the stores may sit in the host data path's
      // cache when the pass starts: such a device takes the stream copy (SMPC_NO_HDP_FLUSH=1, an
      // experiment, keeps the stores without the flush; the canary then guards every tick)
      if (!c->hdp_flush) c->bar_tick = false;
    }
  }
  CK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;
  c->poll_enabled = getenv("SMPC_NO_POLL") == nullptr;
  if (getenv("SMPC_LANE_TIMELINE")) {
    static_assert(SMPC_TAIL_STAMPS_AT == 8192 + kMaxGrid * 8, "the tail's stamps sit behind the lane pass's");
    CK(hipMalloc(&c->d_timeline, (8192 + kMaxGrid * 8 + 8 * 16) * sizeof(unsigned long long)));
    CK(hipMemset(c->d_timeline, 0, (8192 + kMaxGrid * 8 + 8 * 16) * sizeof(unsigned long long)));
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
    // (smpc_pass_split, T = 64: the group-major noise from kSplitMinBatch rollouts up; plan_launch keeps
    // the lane pass itself for batches from kLaneMinBatch up, unless it is asked for)
    if (cfg->time_steps <= 64 && cfg->time_steps >= 36 && (cfg->time_steps & 3u) == 0 && cfg->batch_size >= kSplitMinBatch &&
        !getenv("SMPC_NO_SPLIT"))
      tpr = true;
    if (cfg->flags & SMPC_FLAG_WAVE_PER_ROLLOUT) tpr = false;
    if (cfg->flags & SMPC_FLAG_LANE_PER_ROLLOUT) {
      tpr = true;
      c->lane_forced = true;
    }
    if (const char* e = getenv("SMPC_PASS")) {
      if (!strcmp(e, "wave")) tpr = false;
      if (!strcmp(e, "lane")) {
        tpr = true;
        c->lane_forced = true;
      }
      if (!strcmp(e, "split")) {
        tpr = true;
        c->lane_forced = true;
        c->knob_force_split = true;
      }
    }
    c->knob_no_split = getenv("SMPC_NO_SPLIT") != nullptr;
    c->knob_repeat_pass = getenv("SMPC_DEBUG_REPEAT_PASS") != nullptr;
    if (const char* e = getenv("SMPC_DEBUG_STALE_TICK")) c->knob_stale_tick = static_cast<uint32_t>(atoi(e));
    if (const char* e = getenv("SMPC_SPLIT_NSEG")) c->knob_split_nseg = static_cast<uint32_t>(std::max(0, atoi(e)));
    if (cfg->flags & SMPC_FLAG_STORE_TRAJECTORIES) tpr = false;   // visualisation path: wave pass
    // the group-major copies (smpc_dev.h: SMPC_GM_INDEX): the batch padded to whole groups of 64
    const size_t ngm = static_cast<size_t>(SMPC_GM_ROLLOUTS(cfg->batch_size)) * T * sizeof(float);
    if (3ull * ngm >= (1ull << 32)) tpr = false;   // its buffer descriptor spans the three noise tensors
    c->use_tpr = tpr;
    if (tpr) {
      // back to back: the lane pass addresses the three through one buffer descriptor
      CK(hipMalloc(&c->d_tvx, 3 * ngm));
      c->d_tvy = c->d_tvx + ngm / sizeof(float);
      c->d_twz = c->d_tvy + ngm / sizeof(float);
      if (!c->holonomic) CK(hipMemset(c->d_tvy, 0, ngm));
      CK(smpc_lane_set_lds_limit(static_cast<int>(kLdsPerCu)));
      CK(smpc_split_set_lds_limit(static_cast<int>(kLdsPerCu)));
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
  CK(hipMalloc(&c->d_partials, (kMaxGrid * TL + 16) * sizeof(float)));   // + the canary's slot (SMPC_CANARY_SLOT)
  CK(hipMemset(c->d_partials, 0, (kMaxGrid * TL + 16) * sizeof(float)));
  CK(hipMalloc(&c->d_tuple, TL * sizeof(float)));
  CK(hipMalloc(&c->d_out, (3 * T + 8) * sizeof(float)));
  CK(hipMalloc(&c->d_u_iter, 3 * T * sizeof(float)));
  // [3T + 8 ...]: completion words, one per publishing block (at most 13: T = 128)
  CK(hipHostMalloc(&c->h_out, (3 * T + 8 + kPollWords) * sizeof(float), hipHostMallocMapped));
  memset(c->h_out, 0, (3 * T + 8 + kPollWords) * sizeof(float));
  CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->h_out_dev), c->h_out, 0));
  c->half_blocks = getenv("SMPC_NO_HALF_BLOCKS") == nullptr;
  if (const char* e = getenv("SMPC_MAX_BLOCKS_PER_CU")) c->knob_max_blocks_per_cu = static_cast<uint32_t>(std::max(0, atoi(e)));
  if (const char* e = getenv("SMPC_LANE_REREAD")) c->knob_lane_reread = atoi(e) != 0;
  c->knob_no_inline_tick = getenv("SMPC_NO_INLINE_TICK") != nullptr;
  c->knob_pinned_tick = getenv("SMPC_PINNED_TICK") != nullptr;
  c->knob_balanced_grid = getenv("SMPC_NO_BALANCED_GRID") == nullptr;
  c->fused_reduce = getenv("SMPC_FUSED_REDUCE") != nullptr;   // (read per context: tests compare the two)
  CK(hipMalloc(&c->d_furthest, 32));
  CK(hipMemset(c->d_furthest, 0, 32));
  CK(smpc_set_pass_lds_limit(static_cast<int>(kLdsPerCu)));
  // the memsets above went to the default stream; the ctx works on its own non-blocking one
  CK(hipDeviceSynchronize());
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
  c->hint_Fp_valid = false;
  c->hint_is_this_ticks = false;
  c->hint_drift = 0.f;
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
  {
    const int rc = cancel_redraw(c);
    if (rc != SMPC_OK) return rc;
  }
  HIPCK(c, hipMemcpyAsync(c->d_nvx, nvx, n, hipMemcpyHostToDevice, c->stream));
  if (c->holonomic) HIPCK(c, hipMemcpyAsync(c->d_nvy, nvy, n, hipMemcpyHostToDevice, c->stream));
  HIPCK(c, hipMemcpyAsync(c->d_nwz, nwz, n, hipMemcpyHostToDevice, c->stream));
  {
    int rc = update_time_major(c);
    if (rc != SMPC_OK) return rc;
  }
  HIPCK(c, hipStreamSynchronize(c->stream));
  c->have_noise = true;
  c->rm_valid = true;
  c->rng_mode = false;
  c->noise_gen++;
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

int smpc_redraw_noise_async(smpc_ctx* c)
{
  if (!c) return SMPC_ERR_INVALID;
  if (!c->rng_mode) return fail(c, SMPC_ERR_STATE, "smpc_redraw_noise_async needs device-RNG mode (smpc_seed)");
  HIPCK(c, hipSetDevice(c->device));
  return redraw_async(c);
}

int smpc_get_noise(smpc_ctx* c, float* nvx, float* nvy, float* nwz)
{
  if (!c) return SMPC_ERR_INVALID;
  if (!c->have_noise) return fail(c, SMPC_ERR_STATE, "no noise yet");
  HIPCK(c, hipSetDevice(c->device));
  const size_t n = static_cast<size_t>(c->cfg.batch_size) * c->cfg.time_steps * sizeof(float);
  {
    const int rc = ensure_row_major(c);
    if (rc != SMPC_OK) return rc;
  }
  HIPCK(c, hipStreamSynchronize(c->stream));
  if (nvx) HIPCK(c, hipMemcpy(nvx, c->d_nvx, n, hipMemcpyDeviceToHost));
  if (nvy) HIPCK(c, hipMemcpy(nvy, c->d_nvy, n, hipMemcpyDeviceToHost));
  if (nwz) HIPCK(c, hipMemcpy(nwz, c->d_nwz, n, hipMemcpyDeviceToHost));
  return SMPC_OK;
}

int smpc_optimize(smpc_ctx* c, const smpc_tick_in* in, float* u_inout, smpc_tick_out* out)
{
  if (!c || !in || !u_inout) return fail(c, SMPC_ERR_INVALID, "null argument");
  // developer aid (SMPC_TICK_TIMING=1): where the host's share of a tick goes, printed every 1024 ticks
  static const bool timing = getenv("SMPC_TICK_TIMING") != nullptr;
  using clk = std::chrono::steady_clock;
  clk::time_point t_in, t_prep, t_launch, t_fetch;
  if (timing) t_in = clk::now();
  HIPCK(c, hipSetDevice(c->device));
  int rc = prepare_tick(c, in, u_inout);
  if (rc != SMPC_OK) return rc;
  if (timing) t_prep = clk::now();
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
  float F_host = 0.f;
  bool have_nc_obstacles = false;   // (two_coll_fp: ObstaclesCritic's own non-colliding count, the last one assigned)
  float nc_obstacles = 0.f;
  const bool speculate = !(c->cfg.flags & SMPC_FLAG_NO_SPECULATION);
  for (uint32_t it = 0; it < c->cfg.iteration_count; ++it) {
    uint32_t flags = scoring_flags(c, fail_sticky);
    if (it > 0) flags |= SD_ACCUMULATE;
    have_nc_obstacles = false;   // (the count reported is the LAST iteration's, like every other output)
    // a later iteration starts from what the previous one left in d_out — from a COPY of it: the
    // reduction of this iteration's first pass overwrites d_out, and a second pass of the same
    // iteration (speculation miss, all-collide re-score, the counting pass below) has to read
    // the same control sequence as the first
    const float* u_dev = nullptr;
    if (it > 0) {
      HIPCK(c, hipMemcpyAsync(c->d_u_iter, c->d_out, 3 * T * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
      u_dev = c->d_u_iter;
    }
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
    if (timing && it == 0) t_launch = clk::now();
    fetched = false;
    const bool last = it + 1 == c->cfg.iteration_count;
    if ((flags & (SD_OBSTACLES | SD_COST)) || spec_try || last) {
      // one host round trip: it carries fail_flag (obstacles_critic.cpp:177), the true
      // furthest point, and on the last iteration the result itself
      rc = fetch_out(c);
      if (rc != SMPC_OK) return rc;
      fetched = true;
    }
    if (timing && it == 0) t_fetch = clk::now();
    if (fetched && dF) {
      F_host = c->h_out[iSused];
      S_host = smpc_furthest_index(F_host);
      S_known = true;
    }
    if (spec_try) {
      F_host = c->h_out[iS];
      const uint32_t S_true = smpc_furthest_index(F_host);
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
    if (c->knob_repeat_pass) {   // (tests: a second pass of the iteration must see the first one's inputs)
      rc = launch_score(c, flags & ~SD_LOCAL_FURTHEST, u_dev, nullptr, S_known ? S_host : hintS, c->d_tuple, true, nullptr);
      if (rc != SMPC_OK) return rc;
      rc = fetch_out(c);
      if (rc != SMPC_OK) return rc;
      fetched = true;
    }
    bool failed_at_obstacles = false;
    if (c->two_coll_fp && (flags & SD_OBSTACLES) && (flags & SD_COST) && c->h_out[iNC] != 0.0f) {
      // CostCritic (scored first, its count is the one the pass reports) let some rollouts
      // through; ObstaclesCritic is scored next and, with a footprint in play, decides on its OWN
      // collisions (obstacles_critic.cpp:177 assigns fail_flag).  One pass that scores it alone
      // counts them; then either the manager stopped behind it (critic_manager.cpp:70-73:
      // Constraint, Cost, Obstacles were scored) or everything is scored again.
      const uint32_t acc = it > 0 ? SD_ACCUMULATE : 0u;
      const uint32_t keep = SD_STORE_TRAJ | SD_TRACK_UNKNOWN;
      rc = launch_score(c, (c->gate_flags & (SD_OBSTACLES | SD_FP_OBSTACLES | keep)) | acc, u_dev, nullptr, 0,
                        c->d_tuple, true, nullptr);
      if (rc != SMPC_OK) return rc;
      rc = fetch_out(c);
      if (rc != SMPC_OK) return rc;
      nc_obstacles = c->h_out[iNC];
      uint32_t again = flags & ~SD_LOCAL_FURTHEST;
      if (nc_obstacles == 0.0f) {
        failed_at_obstacles = true;
        fail_flag = true;
        fail_sticky = true;
        again = (c->gate_flags & (SD_CONSTRAINT | SD_COST | SD_FP_COST | SD_OBSTACLES | SD_FP_OBSTACLES | keep)) | acc;
      }
      rc = launch_score(c, again, u_dev, nullptr, S_known ? S_host : 0u, c->d_tuple, true, nullptr);
      if (rc != SMPC_OK) return rc;
      if (c->cfg.flags & SMPC_FLAG_PROFILE) HIPCK(c, hipEventRecord(c->ev1, c->stream));
      rc = fetch_out(c);
      if (rc != SMPC_OK) return rc;
      fetched = true;
      have_nc_obstacles = true;
    }
    if (!failed_at_obstacles && (flags & (SD_OBSTACLES | SD_COST)) && c->h_out[iNC] == 0.0f) {
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
  if (S_known) remember_furthest(c, in, F_host);
  store_control_sequence(c, u_inout);
  if (out) {
    memset(out, 0, sizeof(*out));
    out->fail_flag = fail_flag ? 1 : 0;
    out->furthest_valid = S_known ? 1 : 0;
    out->furthest_reached_path_point = S_known ? S_host : 0;
    out->non_colliding = static_cast<uint32_t>(have_nc_obstacles ? nc_obstacles : c->h_out[iNC]);
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
  if (timing) {
    static thread_local double acc[5] = {0, 0, 0, 0, 0};   // (per calling thread: contexts may tick on several)
    static thread_local unsigned n = 0;
    static thread_local clk::time_point t_prev_out;
    const auto t_out = clk::now();
    auto us = [](clk::time_point a, clk::time_point b) {return std::chrono::duration<double, std::micro>(b - a).count();};
    if (n > 0) acc[0] += us(t_prev_out, t_in);   // the caller, between two ticks
    acc[1] += us(t_in, t_prep);
    acc[2] += us(t_prep, t_launch);
    acc[3] += us(t_launch, t_fetch);
    acc[4] += us(t_fetch, t_out);
    t_prev_out = t_out;
    if (++n == 1024) {
      fprintf(stderr, "[smpc tick timing] caller %.2f us, prepare + upload %.2f, launches %.2f, wait %.2f, collect %.2f (per tick, %u ticks)\n",
              acc[0] / (n - 1), acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n, n);
      n = 0;
      for (double& a : acc) a = 0;
    }
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
int smpc_selftest_row_reduce(smpc_ctx* c, const float* v, uint32_t n, float* out)
{
  if (!c || !v || !out || (n != 16 && n != 32)) return fail(c, SMPC_ERR_INVALID, "bad argument");
  HIPCK(c, hipSetDevice(c->device));
  float *dv = nullptr, *dout = nullptr;
  HIPCK(c, hipMalloc(&dv, 64 * n * sizeof(float)));
  HIPCK(c, hipMalloc(&dout, 64 * sizeof(float)));
  HIPCK(c, hipMemcpyAsync(dv, v, 64 * n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIPCK(c, smpc_launch_row_reduce(dv, dout, n, c->stream));
  HIPCK(c, hipMemcpyAsync(out, dout, 64 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(dv); (void)hipFree(dout);
  return SMPC_OK;
}

int smpc_set_stream(smpc_ctx* c, void* hip_stream)
{
  if (!c) return SMPC_ERR_INVALID;
  const int rc = wait_map_upload(c);   // it went out on the stream being left
  if (rc != SMPC_OK) return rc;
  c->stream = hip_stream == SMPC_STREAM_OWN ? c->own_stream : static_cast<hipStream_t>(hip_stream);
  return SMPC_OK;
}

// developer aid: smpc_grid_tail's stamps of the last launch, [8 reducers][16] in s_memrealtime ticks (10 ns)
int smpc_debug_tail_timeline(smpc_ctx* c, unsigned long long* out)
{
  if (!c || !out || !c->d_timeline) return SMPC_ERR_INVALID;
  HIPCK(c, hipSetDevice(c->device));
  HIPCK(c, hipStreamSynchronize(c->stream));
  HIPCK(c, hipMemcpy(out, c->d_timeline + SMPC_TAIL_STAMPS_AT, 8 * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
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

// developer aid (bench.py labels its roofline with it): the scoring-pass instance the calling
// thread launched last, spelled as rocprofv3's kernel trace spells it
const char* smpc_debug_last_pass_kernel(void) {return smpc_last_pass_kernel;}

// developer aid: how this ctx hands its per-tick inputs to the device — 1: CPU stores through the
// PCIe BAR, 0: a copy on the stream (or, for wave-pass contexts, inside the kernel arguments)
int smpc_debug_bar_tick(const smpc_ctx* c) {return c && c->bar_tick ? 1 : 0;}

int smpc_set_profile(smpc_ctx* c, int enable)
{
  if (!c) return SMPC_ERR_INVALID;
  if (enable) c->cfg.flags |= SMPC_FLAG_PROFILE;
  else c->cfg.flags &= ~static_cast<uint32_t>(SMPC_FLAG_PROFILE);
  return SMPC_OK;
}
}  // extern "C"
