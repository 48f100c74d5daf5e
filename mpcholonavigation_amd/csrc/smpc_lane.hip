// smpc_lane.hip — the streaming pass with one LANE per rollout (T <= 64, lean critics).
//
// smpc_kernels.hip gives every rollout a whole wavefront (lane = time step): the lowest
// latency and the right shape for the reference's deployed batch (2 000 rollouts), but it
// pays ~35 cross-lane DPP steps and a lot of per-rollout scalar work executed 64 lanes wide.
// On CDNA4 a wave64 VALU instruction costs 4 cycles and at 10^5..10^6 rollouts that issue
// rate — not HBM — bounds the pass.  Here a wave owns 64 rollouts and
// walks the horizon step by step:
//   * every per-step operation is one instruction for 64 rollouts, no cross-lane traffic;
//     the float cumulative sums run in the reference's own sequential order
//     (optimizer.cpp:313-343);
//   * everything uniform over the batch (u[t], map geometry, weights) sits in SGPRs;
//   * noise is read from a group-major copy [B / 64][T][64] (smpc_dev.h): one coalesced 256-B
//     piece per array and step, a group's steps back to back, prefetched four steps ahead;
//   * the noised controls of the 64 rollouts stay PARKED IN REGISTERS (3 x 64 per lane)
//     until the rollouts' costs, hence softmax weights, are known; then
//     U[t] += sum_b w_b c[b][t] is a 64 x 64 transpose-reduce done in registers:
//     v_permlane32_swap / v_permlane16_swap (gfx950) and bank-masked DPP adds — two
//     instructions per butterfly node, no LDS, no second read of the noise;
//   * the block partial {min, sum w, furthest, non-colliding, U[3T]} has the layout of the
//     wave-per-rollout pass, so reduction, combine and the multi-GPU tuple are shared.
//
// Scope: the lean scoring mode (every cost_power == 1, no GoalAngle term active, no path
// orientations, no trajectory write-out) at T <= 64; smpc_api.cpp routes every other tick
// to smpc_pass.  Compiled with -ffp-contract=off like smpc_kernels.hip.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <type_traits>

#include "smpc_dev.h"
#include "smpc_device_math.h"
#include "smpc_tail.h"
#include "smpc_lane_common.h"

#ifndef WAVE
#define WAVE 64
#endif
#ifndef LANE_X_RR_NOLOAD
#define LANE_X_RR_NOLOAD 0   // timing experiment: the re-read form without its second read (wrong results)
#endif
#define LANE_BLOCK 512   // 8 waves: 2 per SIMD, 256 registers each
#ifndef LANE_BLOCK_RR
// the re-read instances: 8 waves, one block per CU like the parking form.  (First built with 4
// waves and three blocks per CU: 136 registers allow three waves per SIMD.  The pass is
// VALU-bound either way, three blocks stage three copies of the window, and 4096 groups on 3072
// waves leave a ragged second round: 262 144 x 128 the same within noise, 131 072 x 128: 86 -> 78 us,
// 70 000 x 128: 73 -> 67 us.)
#define LANE_BLOCK_RR 512
#endif

// experiment switches (tools/variants.sh builds the library with some of them off)
#ifndef LANE_X_GAMMA_N
#define LANE_X_GAMMA_N 0     // gamma terms from the noise itself instead of c - u (measured: the noise
                             // registers' longer life breaks the four-step prefetch, +35 % at 2 M rollouts)
#endif
#ifndef LANE_X_GAMMA_UC
#define LANE_X_GAMMA_UC 1    // gamma terms as sum u c - sum u^2 (one fma per control and step; the constant from LDS)
#endif
#ifndef LANE_X_FURTHEST_WINDOW
#define LANE_X_FURTHEST_WINDOW 1   // the endpoint's nearest-path-point scan from three blocks below the scored index up, the rest verified for the wave's winner
#endif
#ifndef LANE_X_PFW_MIN
#define LANE_X_PFW_MIN 1     // PreferForward as -dt sum min(vx, 0)
#endif
#ifndef LANE_X_CELL_AX
#define LANE_X_CELL_AX 1     // cell index from the accumulated displacement (no double add per step)
#endif
#ifndef LANE_X_PIN_LDS
#define LANE_X_PIN_LDS 0     // lookup pipeline as inline-asm LDS reads at the end of the step
#endif
#ifndef LANE_X_PARK_STEP
#define LANE_X_PARK_STEP 0   // park the noised controls step by step instead of quad by quad
#endif
#ifndef LANE_X_CONST_VGPR
#define LANE_X_CONST_VGPR 0  // the cell index's loop constants in vector registers
#endif

// ---------------------------------------------------------------------------
// [B][T] row-major <-> group-major (SMPC_GM_INDEX, smpc_dev.h): one-off, after the noise is supplied
// (or drawn row-major), and back when something asks for the [B, T] tensors of a group-major draw
// ---------------------------------------------------------------------------
template <bool TO_GM>
__global__ void __launch_bounds__(256) smpc_relayout(const float* __restrict__ src, float* __restrict__ dst,
                                                    uint32_t B, uint32_t T)
{
  __shared__ float tile[32][33];
  const uint32_t b0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
  const uint32_t lx = threadIdx.x & 31, ly = threadIdx.x >> 5;   // 32 x 8
  if (TO_GM) {
    for (uint32_t k = ly; k < 32; k += 8) {      // rows of the [B][T] tensor: coalesced along t
      const uint32_t b = b0 + k, t = t0 + lx;
      tile[k][lx] = (b < B && t < T) ? src[(size_t)b * T + t] : 0.f;
    }
    __syncthreads();
    for (uint32_t k = ly; k < 32; k += 8) {      // 32 rollouts of one group at one step: 128 bytes in a row
      const uint32_t t = t0 + k, b = b0 + lx;
      if (b < B && t < T) dst[SMPC_GM_INDEX(b, t, T)] = tile[lx][k];
    }
  } else {
    for (uint32_t k = ly; k < 32; k += 8) {
      const uint32_t t = t0 + k, b = b0 + lx;
      tile[lx][k] = (b < B && t < T) ? src[SMPC_GM_INDEX(b, t, T)] : 0.f;
    }
    __syncthreads();
    for (uint32_t k = ly; k < 32; k += 8) {
      const uint32_t b = b0 + k, t = t0 + lx;
      if (b < B && t < T) dst[(size_t)b * T + t] = tile[k][lx];
    }
  }
}

// to_gm: src [B][T] -> dst group-major; else src group-major -> dst [B][T]
hipError_t smpc_launch_relayout(const float* src, float* dst, uint32_t B, uint32_t T, bool to_gm, hipStream_t st)
{
  const dim3 grid((B + 31) / 32, (T + 31) / 32);
  if (to_gm) hipLaunchKernelGGL(smpc_relayout<true>, grid, dim3(256), 0, st, src, dst, B, T);
  else hipLaunchKernelGGL(smpc_relayout<false>, grid, dim3(256), 0, st, src, dst, B, T);
  return hipGetLastError();
}

// self-test of the transpose-reduce (tests/test_gpu_parity.py): v [64 lanes][64], w [64]
__global__ void __launch_bounds__(64) smpc_lane_reduce_kernel(const float* __restrict__ v,
                                                              const float* __restrict__ w,
                                                              float* __restrict__ out)
{
  const int lane = threadIdx.x;
  float V[64];
#pragma unroll
  for (int t = 0; t < 64; ++t) V[t] = v[lane * 64 + t];
  const LaneW lw = lane_weights(w[lane], lane);
  out[lane] = lane_reduce64(V, lw, lane);
}

hipError_t smpc_launch_lane_reduce(const float* v, const float* w, float* out, hipStream_t st)
{
  hipLaunchKernelGGL(smpc_lane_reduce_kernel, dim3(1), dim3(64), 0, st, v, w, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// The pass
// ---------------------------------------------------------------------------
// FULL: T == 64 NCH, every step slot of the unrolled loops is live; OBST: ObstaclesCritic scored
// MANY: several planning instances in one launch (smpc_group_optimize): blockIdx.y picks
// the instance, whose parameter block comes from device memory instead of the kernel arguments
// NCH: horizon in chunks of 64 steps (T <= 64 NCH)
// RR ("re-read"): the noised controls are NOT parked.  Once a group's weights are known the wave
// reads the group's noise a second time — 64 steps of one control at a time, straight into the
// registers the transpose-reduce consumes — and forms c = u + n again (the same single rounding).
// The second read comes seconds of microseconds after the first: from the Infinity Cache, not
// from HBM.  Without the 128 parked registers and the per-wave LDS slot a lane needs ~1/2 of
// the register file's per-wave share: 256-thread blocks, three per CU (three waves per SIMD).
// The only form for T > 64 (3 T parked values per lane do not fit any register budget).
template <bool FULL, bool OBST, bool MANY, int NCH, bool RR, bool GA = false, bool QUADS = FULL, int TC = 0, bool DEP = false>
__global__ void __launch_bounds__(RR ? LANE_BLOCK_RR : LANE_BLOCK, (RR && LANE_BLOCK_RR == 256) ? 3 : 1)
smpc_pass_lane(const SmpcDev p0, const SmpcLds L, const SmpcDev* __restrict__ many)
{
  static_assert(RR || NCH == 1, "parked controls: one chunk of 64 steps");
  // QUADS: T is a multiple of four, so every step of every executed quad is live and the time
  // loop carries no per-step "t < T" branch.  The reference's default horizon is 56: with the
  // branch around every step the scheduler cannot overlap neighbouring steps, and 56 steps took
  // LONGER than 64 (67.7 against 59.1 us at 262 144 rollouts).
  static_assert(!FULL || QUADS, "T == 64 is a multiple of four");
  // TC: a horizon below 64 known at compile time (the reference's default, 56): trip counts,
  // bound checks and the control sequence's offsets fold as they do for T == 64
  static_assert(TC == 0 || (!FULL && QUADS && !RR && TC < 64 && (TC & 3) == 0), "compile-time horizon: whole quads below 64");
  // DEP: the cruise tick of the reference's deployed critic list (robot_bringup/config/
  // nav2_params.yaml:222: Constraint, Cost, Goal, GoalAngle, PathAlign, PathFollow, PathAngle,
  // PreferForward, Twirling).  Away from the goal and on the path, Goal, GoalAngle and PathAngle
  // are gated off (the host checks), Cost takes ObstaclesCritic's place in the lookup pipeline —
  // same costAtPose, same collision rule, its per-cost term in the table's second field — and
  // Constraint and Twirling are two more additive per-step terms (power 1, in float like the
  // other sums of this pass).  Instances of their own: the cruise instances of the five pay nothing.
  static_assert(!DEP || (OBST && !RR && !GA), "deployed-list instances: parking form");
  const SmpcDev& p = MANY ? many[blockIdx.y] : p0;
  // (this pass reads its tick block from device memory only: it fetches u with scalar loads quad
  // by quad, group after group, and reads of the kernarg segment are not cached the way plain
  // device memory is — u inside the kernel arguments cost the 2 097 152-rollout pass 7 %)
  const SmpcTickPtrs tk{p.u, p.px, p.py, p.pyaw, p.D, p.pf_idx, p.pvalid, p.pa_active, p.pang_active, p.pal_active};
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint8_t* s_map = smem;
  const SmpcLut* s_lut = reinterpret_cast<const SmpcLut*>(smem + L.off_lut);
  float* s_px = reinterpret_cast<float*>(smem + L.off_px);
  float* s_py = reinterpret_cast<float*>(smem + L.off_py);
  // cumulative path distances D[0..P-1) at s_D[0..], with a sentinel on either side of the
  // part PathAlign searches: s_D[-1] = -3e38 and s_D[S] = +3e38 (S = furthest point)
  float* s_D = reinterpret_cast<float*>(smem + L.off_D) + 1;
  // PathAlign's view of the path: {x, y, segment valid ? 1 : 0, 0} per point, one 16-byte read
  f32x4* s_pts4 = reinterpret_cast<f32x4*>(smem + L.off_pts4);
  // sum_t u[ctrl][t]^2 of this launch's control sequence, ctrl = vx, vy, wz (the gamma terms):
  // the 16 bytes in front of the per-wave scratch (smpc_prepare.cpp lane_lds)
  float* s_su2 = reinterpret_cast<float*>(smem + L.off_scr) - 4;

  constexpr int BLK = RR ? LANE_BLOCK_RR : LANE_BLOCK;   // the largest block; small batches launch half of it
  const int blk = blockDim.x;
  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwave = blockDim.x >> 6;
  // developer aid: shader-clock stamps of wave 0 (null pointer: one uniform branch each)
  auto stamp = [&](int k) {
    if (__builtin_expect(p.timeline != nullptr, 0) && tid == 0)
      p.timeline[blockIdx.x * 8 + k] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  // per wave: [64][65] parked wz, [64] softmax weights; the head is re-used by the block combine
  float* park = reinterpret_cast<float*>(smem + L.off_scr) + (size_t)wave * L.scr_stride;
  float* s_w = park + 64 * LANE_PARK_STRIDE;

  // ---- stage costmap window, LUT and path into LDS -------------------------
  // Every global load of the staging goes out before the first LDS store (one memory round trip
  // for window, table and path together); what exceeds one pass of the block is looped over
  // afterwards.
  {
    const int ww = OBST ? p.win_w : 0, wh = OBST ? p.win_h : 0;
    const bool vec = OBST && ((ww & 3) == 0) && ((p.W & 3u) == 0) && ((p.win_x0 & 3) == 0);
    const int w4 = OBST ? (ww >> 2) : 1, n4 = vec ? w4 * wh : 0;   // (1: the map-less variants never divide)
    auto word = [&](int i) -> uint32_t {
      const int ry = i / w4, rx = i - ry * w4;
      return reinterpret_cast<const uint32_t*>(p.map + (size_t)(p.win_y0 + ry) * p.W + p.win_x0)[rx];
    };
    constexpr int kAhead = 96 * 96 / 4 / BLK + 1;   // a 96 x 96 window in one sweep of the block
    uint32_t tmp[kAhead];
#pragma unroll
    for (int k = 0; k < kAhead; ++k) {
      const int i = tid + k * blk;
      tmp[k] = i < n4 ? word(i) : 0u;
    }
    const SmpcLut lut_e = (OBST && tid < 256) ? p.lut[tid] : SmpcLut{0.f, 0.f};
    const bool pt_on = (uint32_t)tid < p.P, seg_on = (uint32_t)tid + 1 < p.P;
    const float g_px = pt_on ? tk.px[tid] : 0.f, g_py = pt_on ? tk.py[tid] : 0.f;
    const float g_D = seg_on ? tk.D[tid] : 0.f;
    const bool g_valid = seg_on && tk.pvalid[tid] != 0;

    if (OBST) {
#pragma unroll
      for (int k = 0; k < kAhead; ++k) {
        const int i = tid + k * blk;
        if (i < n4) reinterpret_cast<uint32_t*>(s_map)[i] = tmp[k];
      }
      for (int i = tid + kAhead * blk; i < n4; i += blk)
        reinterpret_cast<uint32_t*>(s_map)[i] = word(i);
      if (!vec) {
        for (int i = tid; i < ww * wh; i += blockDim.x) {
          const int ry = i / ww, rx = i - ry * ww;
          s_map[i] = p.map[(size_t)(p.win_y0 + ry) * p.W + p.win_x0 + rx];
        }
      }
      if (tid < 256) const_cast<SmpcLut*>(s_lut)[tid] = lut_e;
      if (tid == 0) const_cast<SmpcLut*>(s_lut)[256] = SmpcLut{0.f, 0.f};   // the all-zero entry
      // one byte behind the window answers "off the map" (NO_INFORMATION,
      // obstacles_critic.cpp:209-212); behind it one byte per lane of every wave for costs
      // fetched from the global map (cells outside the window)
      if (tid == 0) s_map[ww * wh] = 255;
    }
    for (uint32_t i = p.P + tid; i < ((p.P + 3u) & ~3u); i += blockDim.x) s_px[i] = s_py[i] = 1.0e18f;
    if (pt_on) {
      s_px[tid] = g_px;
      s_py[tid] = g_py;
      if (seg_on) s_D[tid] = g_D;
      s_pts4[tid] = f32x4{g_px, g_py, g_valid ? 1.0f : 0.f, 0.f};
    }
#if LANE_X_GAMMA_UC
    if (tid < WAVE) {   // (one wave: lane t squares u[.][t], then a butterfly)
      float a = 0.f, b = 0.f, c = 0.f;
      const uint32_t Tn = FULL ? 64u * NCH : (TC ? (uint32_t)TC : p.T);
      for (uint32_t t = (uint32_t)tid; t < Tn; t += WAVE) {
        const float ux = tk.u[t], uy = tk.u[Tn + t], uz = tk.u[2 * Tn + t];
        a = fmaf(ux, ux, a);
        b = fmaf(uy, uy, b);
        c = fmaf(uz, uz, c);
      }
      for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, WAVE);
        b += __shfl_xor(b, o, WAVE);
        c += __shfl_xor(c, o, WAVE);
      }
      if (tid == 0) {
        s_su2[0] = a;
        s_su2[1] = b;
        s_su2[2] = c;
      }
    }
#endif
    for (uint32_t i = tid + blk; i < p.P; i += blk) {   // paths beyond one point per thread
      const float qx = tk.px[i], qy = tk.py[i];
      const bool seg = i + 1 < p.P;
      s_px[i] = qx;
      s_py[i] = qy;
      if (seg) s_D[i] = tk.D[i];
      s_pts4[i] = f32x4{qx, qy, (seg && tk.pvalid[i]) ? 1.0f : 0.f, 0.f};
    }
  }
  __syncthreads();
  stamp(1);

  // ---- constants (wave-uniform: scalar registers) -------------------------------
  // u and the path are inputs of the launch: read them through the constant address space,
  // so that uniform loads stay scalar loads although the kernel also stores to global memory
  const cfloat_p cu = (cfloat_p)(uintptr_t)p.u;
  const uint32_t T = FULL ? 64u * NCH : (TC ? (uint32_t)TC : p.T), B = p.B;
  // group-major noise through buffer loads: per step one scalar offset (t * 256) serves the
  // three tensors, the lane's own offset (its group's start + lane * 4) is the vector offset
  // (the host lays the three tensors out back to back: ONE descriptor, four scalar
  // registers instead of twelve — the loop is short of them — and the tensor is part of the
  // scalar offset)
  const uint32_t noise_bytes = T * SMPC_GM_ROLLOUTS(B) * 4u;   // one tensor, group-major (smpc_dev.h)
  const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.tvx), 0, 3u * noise_bytes, 0x00020000);
  const float dt = p.dt, yaw0 = p.yaw0;
  const double x0 = p.x0, y0 = p.y0;
  uint32_t S = 0;
  if (p.flags & SD_NEED_FURTHEST) {
    S = p.d_furthest ? smpc_furthest_index(*p.d_furthest) : p.furthest_hint;
    if (S >= p.P) S = p.P ? p.P - 1 : 0;
  }
  // the sentinels of s_D (every wave writes the same two values before its first read: no
  // second barrier, and S need not be known while the block stages)
  if (lane == 0) {
    s_D[-1] = -3.0e38f;
    s_D[S] = 3.0e38f;
  }
  const bool pa_on = (p.flags & SD_PATH_ALIGN) && p.P > 0 && tk.pa_active[S] && S > 0;
  float pf_x = 0.f, pf_y = 0.f;
  if ((p.flags & SD_PATH_FOLLOW) && p.P > 0) {
    const uint32_t idx = tk.pf_idx[S];
    pf_x = tk.px[idx];
    pf_y = tk.py[idx];
  }
  const uint32_t bs_iters = S > 1 ? 32u - (uint32_t)__builtin_clz(S - 1) : 0u;
  float pa_inv_spacing = 0.f;
  if (pa_on && S > 1 && tk.D[S - 1] > 0.f) pa_inv_spacing = (float)(S - 1) / tk.D[S - 1];
  const bool want_local_furthest = (p.flags & SD_NEED_FURTHEST) && (p.flags & SD_LOCAL_FURTHEST);
  const bool con_on = DEP && (p.flags & SD_CONSTRAINT) != 0, tw_on = DEP && (p.flags & SD_TWIRLING) != 0;
  const float k_con = p.dt * p.con_weight, k_tw = p.tw_weight / (float)(FULL ? 64 * NCH : (TC ? TC : (int)p.T));
  // CostCritic in ObstaclesCritic's place (DEP): the table's second field holds its per-cost term
  const bool cost_mode = DEP && (p.flags & SD_COST) != 0;
  // GoalAngleCritic is a near-goal term: it is compiled into instances of their own (GA), which the
  // launcher picks when the tick's flags carry it — as a run-time branch of the cruise
  // instances it cost them 1.3-3 % (413 against 401 us on the 2 097 152-rollout pass)
  constexpr bool ga_on = GA;
  const uint32_t nquad = (T + 3u) >> 2;

  // ---- per-wave running softmax state; U[ctrl][t] lives in lane t ----------------
  float m_run = 3.0e38f, s_run = 0.f;
  float Ux[NCH], Uy[NCH], Uz[NCH];   // chunk h: steps [64 h, 64 h + 64)
#pragma unroll
  for (int h = 0; h < NCH; ++h) Ux[h] = Uy[h] = Uz[h] = 0.f;
  float F_local = 0.f;   // furthest point of this wave's rollouts, index + fraction (smpc_dev.h)
  uint32_t n_noncoll = 0;

  const uint32_t ngroups = (B + WAVE - 1) / WAVE;
  const uint32_t gw = blockIdx.x * nwave + wave;
  const uint32_t nW = gridDim.x * nwave;

  // One group of 64 rollouts.  SAFE = false is the fast instance: sin/cos without the
  // huge-argument branch (a divergent branch in the middle of the step would keep the
  // scheduler from overlapping the costmap lookups with it); it only notes whether some
  // |yaw| left the range of the fast reduction, commits nothing in that case and returns
  // true, and the group is redone by the SAFE instance.
  auto group_body = [&](auto safe_c, const uint32_t grp) -> bool {
    constexpr bool SAFE = decltype(safe_c)::value;
    const uint32_t b = grp * WAVE + lane;
    const bool live = b < B;
    const uint32_t bl = live ? b : B - 1;        // tail lanes shadow the last rollout
    // parked noised controls of this group, c[ctrl][t] = P<ctrl><t / 32>[t % 32]: register
    // tuples written through the scalar GPR index (s_set_gpr_idx) inside the rolled time loop
    // and read with static indices by the transpose-reduce, which works in place
    // (vx and vy: 128 registers).  wz is parked in this wave's LDS slot instead, [t][65]:
    // the write is lane-contiguous, the transposed read (lane t, rollout b) conflict-free.
    // Element 8 i + q' of PX<h> holds c_vx[t = 32 h + 4 q' + i]: the eight parks of a quad
    // index with the same scalar q' (the constant 8 i folds into the base register) and can
    // share one s_set_gpr_idx_on/off pair.
    f32x32 PX0, PX1, PY0, PY1;
    if (!FULL && !RR) PX0 = PX1 = PY0 = PY1 = (f32x32)(0.f);

    // ================= rollout + per-step critics, lane = rollout =====================
    float cpx = p.svx, cpy = p.svy, cpz = p.swz;   // v[:,0] = measured speed, v[:,t] = c[:,t-1]
    float acc_yaw = 0.f, ax = 0.f, ay = 0.f;
    float cs_prev = p.cos0, sn_prev = p.sin0;
    float x = 0.f, y = 0.f;
    float crit = 0.f, rep = 0.f;
    float yaw_max = 0.f;   // largest |yaw| seen (fast instance: range check of the sin/cos reduction)
    float alive = 1.0f;   // 1 until the rollout's first collision, then 0 (a float mask: fma(1, a, c) == c + a)
    // The costmap lookup is two dependent LDS reads (the cell's byte, then the byte's table
    // entry) feeding an in-order accumulation.  It runs as a pipeline two steps deep whose three
    // stages all sit at the END of a step, behind one wait: accumulate the entry of step t - 2,
    // issue the table read for the byte of step t - 1, issue the byte read of step t's own cell —
    // every read has had a whole step to land, so the wait is free.  The reads are inline
    // assembly (the wait too: the compiler's own waitcnt pass does not see them), which pins
    // their place in the step; left to the scheduler the byte read sinks to just in front of
    // its use and every step waits out an LDS round trip.  Primed with the all-zero table
    // entry 256; drained after the loop.
    uint32_t cell_q = 256u;
    f32x2 e_q = {0.f, 0.f};   // {crit, rep} of SmpcLut
    [[maybe_unused]] const uint32_t lds_lut = (uint32_t)(uintptr_t)s_lut, lds_map = (uint32_t)(uintptr_t)s_map;   // (LANE_X_PIN_LDS)
    auto lookup_wait = [&]() {   // both reads of the previous step have landed
#if LANE_X_PIN_LDS
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cell_q), "+v"(e_q));
#endif
    };
    auto lookup_accumulate = [&]() {
      // steps after the first collision are never visited in the reference (masked)
      alive = e_q.x < 0.f ? 0.f : alive;   // inCollision
      crit = fmaf(alive, e_q.x, crit);
      rep = fmaf(alive, e_q.y, rep);
    };
#if LANE_X_PIN_LDS
    auto lookup_issue_entry = [&]() {          // e_q <- s_lut[cell_q]
      const uint32_t a = lds_lut + (cell_q << 3);
      asm volatile("ds_read_b64 %0, %1" : "=v"(e_q) : "v"(a));
    };
    auto lookup_issue_byte = [&](uint32_t idx) {   // cell_q <- s_map[idx]
      const uint32_t a = lds_map + idx;
      asm volatile("ds_read_u8 %0, %1" : "=v"(cell_q) : "v"(a));
    };
#else
    auto lookup_issue_entry = [&]() {
      const SmpcLut e = s_lut[cell_q];
      e_q = f32x2{e.crit, e.rep};
    };
    auto lookup_issue_byte = [&](uint32_t idx) {cell_q = s_map[idx];};
#endif
    float pfw = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
    float ga_sum = 0.f;   // GoalAngleCritic: sum over the steps of |shortest angular distance to the goal's yaw|
    float ext = 0.f;      // DEP: ConstraintCritic + TwirlingCritic, weights and 1/T folded in
    // PathAlign running state (path_align_critic.cpp:92-133)
    // (trajectory point 0 is the same for every rollout: host-computed, same arithmetic)
    float traj_dist = 0.f, pa_sum = 0.f, pa_num = 0.f, sx_prev = p.x00f, sy_prev = p.y00f;
    uint32_t path_pt = 0;

    // loop constants of the cell index, in vector registers: a scalar operand halves the issue
    // rate of the instruction that reads it (tools/ubench)
#if LANE_X_CONST_VGPR
    const float k_rinv = in_vgpr(p.rinvf), k_cx = in_vgpr(p.cxf), k_cy = in_vgpr(p.cyf);
#else
    const float k_rinv = p.rinvf, k_cx = p.cxf, k_cy = p.cyf;
#endif
    const float k_edge = 0.5f - p.cell_eps_w;
    // one time step for the 64 rollouts of this wave; t, ux, uy, uz are wave-uniform
    // sample_slot: this step is a multiple of four (known at compile time in the unrolled quad)
    auto do_step = [&](const uint32_t t, const bool sample_slot, const float ux, const float uy, const float uz,
                       const float n0, const float n1, const float n2, float& cvx, float& cvy,
                       float& cwz) {
      // NoiseGenerator::setNoisedControls (noise_generator.cpp:65-74)
      cvx = ux + n0;
      cvy = uy + n1;
      cwz = uz + n2;
      const float vx = cpx, vy = cpy, wz = cpz;
      cpx = cvx;
      cpy = cvy;
      cpz = cwz;
      // integrateStateVelocities (optimizer.cpp:313-343): sequential float cumsums
      acc_yaw = acc_yaw + wz * dt;
      const float yaw = acc_yaw + yaw0;
      const float dxr = vx * cs_prev - vy * sn_prev;
      const float dyr = vx * sn_prev + vy * cs_prev;
      ax = ax + dxr * dt;
      ay = ay + dyr * dt;
      if constexpr (DEP) {
        // ConstraintCritic (constraint_critic.cpp:41-75; holonomic and differential models: the
        // host keeps Ackermann off these instances): how far the signed speed leaves
        // [min_vel, max_vel], times dt and the weight
        if (con_on) {
          const float sp = fast_sqrt(vx * vx + vy * vy);
          const float vt = vx > 0.f ? sp : -sp;
          const float e = fmaxf(vt - p.con_max_vel, 0.f) + fmaxf(p.con_min_vel - vt, 0.f);
          ext = fmaf(e, k_con, ext);
        }
        // TwirlingCritic (twirling_critic.cpp:30-42): mean |wz| times the weight
        if (tw_on) ext = fmaf(fabsf(wz), k_tw, ext);
      }
      if (ga_on) {
        // GoalAngleCritic (goal_angle_critic.cpp:36-50), near-goal ticks only (a uniform branch):
        // |normalize_angles(goal yaw - yaw)|.  The reference normalises in double; here the
        // float difference (the reference's own) is reduced by 2 pi in two fused multiply-adds
        // — the remainder is within 2e-7 of the double one, the mean of 64 of them moves the
        // rollout's cost by ~1e-7 relative.
        const float a = p.ga_goal_yaw - yaw;
        const float kf = fmaf(a, 0.15915494309189535f, 12582912.0f);
        const float k = kf - 12582912.0f;
        float r = fmaf(-k, 6.2831854820251465f, a);
        r = fmaf(-k, -1.7484555e-07f, r);
        ga_sum += fabsf(r);
      }
      // The trajectory point itself, x = (float)(x0 + (double)ax) as the reference narrows it
      // (optimizer.cpp:331-342), is formed only where its VALUE is consumed: at PathAlign's
      // sample steps, at the endpoint and on the exact path of the cell index below.  The
      // three double-precision instructions per axis run at half the rate of the plain float
      // ones (tools/ubench: 4 against 2 SIMD cycles per wave64 instruction).

      // ObstaclesCritic lookup (obstacles_critic.cpp:139-171).  Fast cell index first: the
      // window-relative quotient from the accumulated displacement in ONE fused multiply-add,
      // q = ax / res + (x0 - window corner) / res.  Its distance to the quotient the reference
      // truncates — ((double)x - origin) / res with x ROUNDED to float first — is bounded on the
      // host (cell_eps_w: that rounding of x, the float images of the two constants, the fma's
      // own rounding).  Lanes within that bound of a cell edge, outside the window or off the
      // map get their LDS byte index from the exact path (the reference's own double arithmetic
      // on the rounded x); then ONE pair of dependent LDS reads serves every lane and overlaps
      // the sin/cos below.
      uint32_t idx = 0;
      if (OBST) {
#if LANE_X_CELL_AX
        const float qx = fmaf(ax, k_rinv, k_cx), qy = fmaf(ay, k_rinv, k_cy);
#else
        x = (float)(x0 + (double)ax);
        y = (float)(y0 + (double)ay);
        const float qx = (x - p.wxf) * p.rinvf, qy = (y - p.wyf) * p.rinvf;
#endif
        const float rx = __builtin_amdgcn_fractf(qx), ry = __builtin_amdgcn_fractf(qy);
        const int lx = cvt_floor_i32(qx), ly = cvt_floor_i32(qy);
        // guard band as ONE compare: both fractions at least eps away from a cell edge <=>
        // max(|rx - 1/2|, |ry - 1/2|) <= 1/2 - eps (a NaN fails it; the two extra float
        // roundings, < 1e-7, sit inside the factor 2 the host puts on eps).
        const float edge = fmaxf(fabsf(rx - 0.5f), fabsf(ry - 0.5f));
        const bool fast = (edge <= k_edge) &
                          ((uint32_t)lx < (uint32_t)p.win_w) & ((uint32_t)ly < (uint32_t)p.win_h);
        // window cells fit 24 bits: v_mad_u32_u24 instead of a 64-bit multiply-add
        idx = __umul24((uint32_t)ly, (uint32_t)p.win_w) + (uint32_t)lx;   // (meaningless if !fast)
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!fast) != 0, 0)) {
          if (!fast)
            idx = cell_byte_exact(p, s_map, (float)(x0 + (double)ax), (float)(y0 + (double)ay),
                                  (uint32_t)(wave * WAVE + lane));
        }
      }

      // cos_[t+1] = cos(yaw[t]); the last step's is never used
      if (SAFE) {
        smpc_sincos(yaw, sn_prev, cs_prev);
      } else {
        yaw_max = fmaxf(yaw_max, fabsf(yaw));
        smpc_sincos_fast(yaw, sn_prev, cs_prev);
      }
      // PreferForwardCritic (prefer_forward_critic.cpp:42-46): sum_t max(-vx, 0) dt, here as
      // -dt sum_t min(vx, 0) — the factor once per rollout instead of once per step
#if LANE_X_PFW_MIN
      pfw = pfw + fminf(vx, 0.f);
#else
      pfw = fmaf(fmaxf(-vx, 0.f), dt, pfw);
#endif
      // updateControlSequence gamma terms (optimizer.cpp:365-380): sum_t u (c - u).  c - u is
      // the noise up to the rounding of c = u + n (|c - u - n| <= ulp(c) / 2: a few 1e-8 on
      // terms that gamma / sigma^2 scales to ~1e-7 of a cost): the noise itself is used
#if LANE_X_GAMMA_UC
      // ... as sum_t u c - sum_t u^2: one fused multiply-add per control here, the constant
      // (s_su2, formed once per launch) subtracted once per rollout.  Half the instructions of
      // u (c - u); the running sums reach T |u| |c| instead of staying near zero, which is ~1e-6
      // absolute on a cost after the gamma / sigma^2 scaling (costs are compared at 2e-4).
      gx = fmaf(ux, cvx, gx);
      gz = fmaf(uz, cwz, gz);
      gy = fmaf(uy, cvy, gy);
#elif LANE_X_GAMMA_N
      gx = fmaf(ux, n0, gx);
      gz = fmaf(uz, n2, gz);
      gy = fmaf(uy, n1, gy);
#else
      gx = fmaf(ux, cvx - ux, gx);
      gz = fmaf(uz, cwz - uz, gz);
      gy = fmaf(uy, cvy - uy, gy);
#endif

      // PathAlignCritic sample (uniform in t): trajectory points step, 2 step, ...
      // (trajectory_point_step is 4 here, the reference's default — the host sends any other
      // value to the wave-per-rollout pass — so the sample steps are the first of every quad
      // but the very first: no per-step bookkeeping, no branch in the other three steps)
      if (sample_slot && pa_on && t != 0) {
        x = (float)(x0 + (double)ax);
        y = (float)(y0 + (double)ay);
        const float ddx = x - sx_prev, ddy = y - sy_prev;
        traj_dist += fast_sqrt(ddx * ddx + ddy * ddy);
        sx_prev = x;
        sy_prev = y;
        // utils::findClosestPathPt(D, traj_dist, path_pt) (tools/utils.hpp:665-675):
        // std::lower_bound over D[0..S) guessed from the mean spacing, confirmed against
        // D[g-1], D[g], D[g+1]; binary search only if some lane is unconfirmed
        const float dist = traj_dist;
        uint32_t gi = (uint32_t)(dist * pa_inv_spacing);
        gi = gi < S ? gi : S - 1;
        const float da = s_D[(int)gi - 1];     // the sentinels stand in at either end
        const float db = s_D[gi];
        const float dc = s_D[gi + 1];
        // D is non-decreasing, so (da < dist) >= (db < dist) >= (dc < dist): the lower bound
        // is g or g + 1 exactly when the first holds and the last does not
        const bool c0 = da < dist, c1 = db < dist, c2 = dc < dist;
        uint32_t lo = gi + (c1 ? 1u : 0u);
        float dl = c1 ? db : da, dh = c1 ? dc : db;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!c0 || c2) != 0, 0)) {
          uint32_t base = 0, nn = S;
          for (uint32_t it = 0; it < bs_iters; ++it) {
            const uint32_t half = nn >> 1;
            base = (s_D[base + half - 1 + (half == 0)] < dist && half) ? base + half : base;
            nn -= half;
          }
          const float d_base = s_D[base];
          lo = base + (d_base < dist ? 1u : 0u);
          dl = lo > 0 ? s_D[lo - 1] : 0.f;
          dh = lo < S ? s_D[lo] : 0.f;
        }
        // lower_bound restricted to [path_pt, S) is the global one, since path_pt <= lo
        uint32_t pt;
        if (lo == path_pt) pt = 0;                 // iter == begin + init
        else if (lo >= S) pt = S - 1;              // end(): defined as size-1
        else pt = (dist - dl < dh - dist) ? lo - 1 : lo;
        path_pt = pt;
        const f32x4 q = s_pts4[pt];
        const float ex = q[0] - x, ey = q[1] - y;
        const float d = fast_sqrt(ex * ex + ey * ey);
        pa_num += q[2];                  // segment valid ? 1 : 0 (path_align_critic.cpp:119-127)
        pa_sum = fmaf(q[2], d, pa_sum);
      }
      if (OBST) {   // the lookup pipeline's three stages (see above)
        lookup_wait();
        lookup_accumulate();       // entry of step t - 2
        lookup_issue_entry();      // byte of step t - 1
        lookup_issue_byte(idx);    // this step's cell
      }
    };

    // noise: step t is a uniform base + this lane's offset; four steps in flight.
    // The control sequence of the next four steps is fetched (scalar loads) a quad ahead too.
    // noise, group-major: step t of this wave's 64 rollouts is 256 bytes behind step t - 1
    const uint32_t loff = ((bl >> 6) * T * 64u + (bl & 63u)) * 4u;   // SMPC_GM_INDEX(bl, 0, T) in 32 bits: the descriptor spans < 4 GB
    constexpr uint32_t step_bytes = 256u;
    auto ld = [&](uint32_t tensor, uint32_t t) -> float {
      // (the whole-quads instances of T < 64 prefetch unconditionally — see run_quad — so their
      // last quad's prefetch is clamped to the last row; T = 64: never out of range)
      const uint32_t tc = (FULL || t < T) ? t : T - 1;
      return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rn, loff, tensor * noise_bytes + tc * step_bytes, 0));
    };
    auto ldu = [&](uint32_t ctrl, uint32_t t) -> float {return cu[ctrl * T + ((FULL || t < T) ? t : T - 1)];};
    float nq[12], uq[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      nq[3 * i + 0] = ld(0, i);
      nq[3 * i + 1] = ld(1, i);
      nq[3 * i + 2] = ld(2, i);
#pragma unroll
      for (int k = 0; k < 3; ++k) uq[3 * i + k] = ldu(k, i);
    }
    uint64_t clk = __builtin_amdgcn_s_memtime();
    // four steps; each parks its noised controls at once (LANE_X_PARK_STEP) or the quad returns
    // them in cq[3 i + ctrl] for the caller to park
    auto run_quad = [&](auto hi_c, const uint32_t q, float (&cq)[12]) {
      [[maybe_unused]] constexpr bool HI = decltype(hi_c)::value;
      // The two waves of a SIMD do not share it evenly by themselves: the older one wins every
      // tie and finishes its groups ~25 % sooner (41 us against 51 us for two groups), then the
      // younger one runs alone.  Swapping their priorities every 2^15 shader clocks — by the clock,
      // read a quad earlier: the same for both whatever their progress, in anti-phase between
      // waves w and w + 4 — lets both finish together at 47 us (measured: tools/lane_timeline.py;
      // shorter periods share less evenly, 2^12: 45 / 48 us).
      if (((uint32_t)(clk >> 15) + (uint32_t)(wave >> 2)) & 1u) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
      clk = __builtin_amdgcn_s_memtime();
      float uc[12];
#pragma unroll
      for (int j = 0; j < 12; ++j) uc[j] = uq[j];
      // The next quad's controls and noise are fetched a quad ahead.  Where the trip count is a
      // run-time value (T < 64) the fetch is unconditional — the last quad re-reads the last
      // row: behind a run-time "is there a next quad" the waitcnt pass gives up the prefetch
      // depth (every wait became vmcnt(0), and 56 steps took longer than 64).
      constexpr bool kAlwaysAhead = !FULL && TC == 0;
      if (kAlwaysAhead || q + 1 < nquad) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int k = 0; k < 3; ++k) uq[3 * i + k] = ldu(k, 4 * (q + 1) + i);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t t = 4 * q + i;
        // this step's noise; its registers are refilled at once with step t + 4
        const float n0 = nq[3 * i], n1 = nq[3 * i + 1], n2 = nq[3 * i + 2];
        if (kAlwaysAhead || q + 1 < nquad) {
          nq[3 * i + 0] = ld(0, t + 4);
          nq[3 * i + 1] = ld(1, t + 4);
          nq[3 * i + 2] = ld(2, t + 4);
        }
        cq[3 * i] = cq[3 * i + 1] = cq[3 * i + 2] = 0.f;
        if (QUADS || t < T)
          do_step(t, i == 0, uc[3 * i], uc[3 * i + 1], uc[3 * i + 2], n0, n1, n2, cq[3 * i], cq[3 * i + 1],
                  cq[3 * i + 2]);
#if LANE_X_PARK_STEP
        // park this step's controls now: three values live per step instead of twelve per quad
        if constexpr (RR) {
        } else if constexpr (HI) {
          PX1[8 * i + (q - 8)] = cq[3 * i];
          PY1[8 * i + (q - 8)] = cq[3 * i + 1];
        } else {
          PX0[8 * i + q] = cq[3 * i];
          PY0[8 * i + q] = cq[3 * i + 1];
        }
        if constexpr (!RR) park[(4 * q + i) * LANE_PARK_STRIDE + lane] = cq[3 * i + 2];
#endif
      }
    };
    if constexpr (RR) {
      // nothing is parked: the controls are formed again from the noise once the weights are known
#pragma unroll 2
      for (uint32_t q = 0; q < nquad; ++q) {
        float cq[12];
        run_quad(std::false_type{}, q, cq);
      }
    } else {
      // steps [0, 32) park into the <..>0 tuples, [32, 64) into <..>1, element t % 32
      const uint32_t qh = nquad < 8u ? nquad : 8u;
      if constexpr (QUADS && !FULL && TC == 0) {
        auto quad_lo = [&](const uint32_t q) {
          float cq[12];
          run_quad(std::false_type{}, q, cq);
#if !LANE_X_PARK_STEP
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            PX0[8 * i + q] = cq[3 * i];
            PY0[8 * i + q] = cq[3 * i + 1];
            park[(4 * q + i) * LANE_PARK_STRIDE + lane] = cq[3 * i + 2];
          }
#endif
        };
        auto quad_hi = [&](const uint32_t q) {
          float cq[12];
          run_quad(std::true_type{}, q, cq);
#if !LANE_X_PARK_STEP
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            PX1[8 * i + (q - 8)] = cq[3 * i];
            PY1[8 * i + (q - 8)] = cq[3 * i + 1];
            park[(4 * q + i) * LANE_PARK_STRIDE + lane] = cq[3 * i + 2];
          }
#endif
        };
        // two quads per iteration by hand: with a run-time trip count "#pragma unroll 2" is not
        // honoured here, and one quad per iteration leaves the scheduler nothing to overlap
        // the next quad's loads and lookups with (699 VALU per 4 steps in a loop of its own)
        uint32_t q = 0;
        for (; q + 1 < qh; q += 2) {
          quad_lo(q);
          quad_lo(q + 1);
        }
        if (q < qh) quad_lo(q);
        q = 8;
        for (; q + 1 < nquad; q += 2) {
          quad_hi(q);
          quad_hi(q + 1);
        }
        if (q < nquad) quad_hi(q);
      } else {
        // (the T = 64 instances: these two loops verbatim — moving their bodies into lambdas cost
        // the main instance 896 bytes of scratch)
#pragma unroll 2
        for (uint32_t q = 0; q < qh; ++q) {
          float cq[12];
          run_quad(std::false_type{}, q, cq);
#if !LANE_X_PARK_STEP
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            PX0[8 * i + q] = cq[3 * i];
            PY0[8 * i + q] = cq[3 * i + 1];
            park[(4 * q + i) * LANE_PARK_STRIDE + lane] = cq[3 * i + 2];
          }
#endif
        }
#pragma unroll 2
        for (uint32_t q = 8; q < nquad; ++q) {
          float cq[12];
          run_quad(std::true_type{}, q, cq);
#if !LANE_X_PARK_STEP
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            PX1[8 * i + (q - 8)] = cq[3 * i];
            PY1[8 * i + (q - 8)] = cq[3 * i + 1];
            park[(4 * q + i) * LANE_PARK_STRIDE + lane] = cq[3 * i + 2];
          }
#endif
        }
      }
    }
    if (OBST) {   // drain the lookup pipeline: the entries of the last two steps
      lookup_wait();
      lookup_accumulate();
      lookup_issue_entry();
      lookup_wait();
      lookup_accumulate();
    }
    // (a NaN yaw is sticky in the cumulative sum: the last one shows it)
    if (!SAFE && __builtin_expect(__any(!(yaw_max < 65536.0f) || !(fabsf(acc_yaw) < 65536.0f)), 0)) return true;

    // the endpoint (trajectory point T - 1), as the reference narrows it
    x = (float)(x0 + (double)ax);
    y = (float)(y0 + (double)ay);

    // ================= per-rollout epilogue, lane = rollout ==============================
    // nearest path point of the endpoint (utils.hpp:292-319): first minimum wins
    if (want_local_furthest) {
      // Four path points per pair of LDS broadcast reads (the arrays are padded to a multiple
      // of four with far-away points that never win).  The strict "<" scan runs over the
      // MINIMUM of each block of four — the first block that holds the overall minimum wins —
      // and the first point of that block that attains it is found afterwards, from the same
      // arithmetic: the reference's first minimum at a third of the compare/select work.
      auto block_d2 = [&](const float* bx, const float* by, float (&dd)[4]) {
        const f32x4 qx = *reinterpret_cast<const f32x4*>(bx);
        const f32x4 qy = *reinterpret_cast<const f32x4*>(by);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ex = qx[e] - x, ey = qy[e] - y;
          dd[e] = ex * ex + ey * ey;
        }
      };
      float best = 3.4028234663852886e38f;
      uint32_t bj = 0;
      const uint32_t P4 = (p.P + 3u) & ~3u;
      // The scan is 15 blocks for a 60-point path and only its batch-wide MAXIMUM is consumed.  So
      // it starts three blocks below the block of the index this tick is scored with (where the
      // maximum has been every tick so far) and runs to the path's end; what that leaves out is
      // checked for the ONE lane that ends up holding the wave's maximum (below).  (The plain
      // instances only: in the GoalAngle, deployed-list and grouped ones the extra code costs
      // 16-48 bytes of scratch.)
      constexpr bool kWindow = LANE_X_FURTHEST_WINDOW && !GA && !DEP && !MANY;
      const uint32_t j_lo = (kWindow && (S >> 2) > 3u) ? ((S >> 2) - 3u) << 2 : 0u;
      for (uint32_t j = j_lo; j < P4; j += 4) {
        float dd[4];
        block_d2(s_px + j, s_py + j, dd);
        const float mn = fminf(fminf(dd[0], dd[1]), fminf(dd[2], dd[3]));
        if (mn < best) {      // a NaN or infinite distance never wins, as in the plain scan
          best = mn;
          bj = j;
        }
      }
      // index + how far the endpoint sits towards the next point, in segment lengths (what the
      // host predicts the next tick's index from; smpc_dev.h)
      auto point_F = [&]() -> float {
        float dd[4];
        block_d2(s_px + bj, s_py + bj, dd);
        const uint32_t bi = bj + (dd[0] == best ? 0u : dd[1] == best ? 1u : dd[2] == best ? 2u : dd[3] == best ? 3u : 0u);
        float F = (float)bi;
        if (bi + 1 < p.P) {
          const float nx = s_px[bi + 1], ny = s_py[bi + 1];
          const float sgx = nx - s_px[bi], sgy = ny - s_py[bi];
          const float d_next = (nx - x) * (nx - x) + (ny - y) * (ny - y);
          const float seg2 = sgx * sgx + sgy * sgy;
          const float tt = seg2 > 0.f ? 0.5f + 0.5f * (best - d_next) * fast_rcp(seg2) : 0.f;
          F = fmaxf(F + fminf(fmaxf(tt, -0.45f), 0.45f), 0.f);
        }
        return F;
      };
      float F = point_F();
      float m = live ? F : 0.f;
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, WAVE));
      if (kWindow && j_lo) {
        // A lane whose true nearest point lies BELOW the window holds a value that is too high,
        // never too low: max over the lanes of the windowed values >= the true maximum, with
        // equality as soon as ONE lane that attains it is exact.  So: take a lane holding the
        // maximum and test every point below the window against its endpoint — one point per
        // lane of the wave, "<=" because an equal distance at a lower index wins the reference's
        // strict scan.  If one of them beats it (a path that doubles back under the endpoint),
        // every lane scans the lower blocks after all and the maximum is formed again.
        const unsigned long long holders = __ballot(live && F == m);
        bool below = false;
        if (holders) {
          const int wl = __builtin_ctzll(holders);
          const float wx = __shfl(x, wl, WAVE), wy = __shfl(y, wl, WAVE), wbest = __shfl(best, wl, WAVE);
          for (uint32_t k = (uint32_t)lane; k < j_lo; k += WAVE) {
            const float ex = s_px[k] - wx, ey = s_py[k] - wy;
            below = below || (ex * ex + ey * ey <= wbest);
          }
        }
        if (__builtin_expect(__any(below), 0)) {
          for (uint32_t jj = j_lo; jj > 0; jj -= 4) {      // downwards: ties go to the lower block
            const uint32_t j = jj - 4;
            float dd[4];
            block_d2(s_px + j, s_py + j, dd);
            const float mn = fminf(fminf(dd[0], dd[1]), fminf(dd[2], dd[3]));
            if (mn <= best) {
              best = mn;
              bj = j;
            }
          }
          F = point_F();
          m = live ? F : 0.f;
          for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, WAVE));
        }
      }
      F_local = fmaxf(F_local, m);
    }
    // costs (every cost_power == 1): the lean association of smpc_pass MODE 0
    float cost = (p.flags & SD_ACCUMULATE) ? p.costs_prev[bl] : 0.f;
    float lin = 0.f, uni = 0.f;
    if (OBST) {
      const bool collided = alive == 0.f;
      if (cost_mode) {
        // cost_critic.cpp:157-166: collision_cost for a colliding rollout, else the sum of the
        // per-cost terms, times weight / 254 / T
        const float k = p.cost_w254 / (float)T;
        lin = collided ? 0.f : k * rep;
        uni = collided ? k * p.cost_collision_cost : 0.f;
      } else {
        lin = (collided ? 0.f : p.obs_critical_w * crit) + p.obs_rep_over_T * rep;
        uni = collided ? p.obs_critical_w * p.obs_collision_cost : 0.f;
      }
      n_noncoll += (uint32_t)__popcll(__ballot(live && !collided));
    }
    if constexpr (DEP) lin += ext;
    if (p.flags & SD_PATH_FOLLOW) {
      const float fdx = x - pf_x, fdy = y - pf_y;
      uni += p.pf_weight * fast_sqrt(fdx * fdx + fdy * fdy);
    }
#if LANE_X_PFW_MIN
    if (p.flags & SD_PREFER_FORWARD) lin += (pfw * -dt) * p.pfw_weight;
#else
    if (p.flags & SD_PREFER_FORWARD) lin += pfw * p.pfw_weight;
#endif
    if (ga_on) uni += (ga_sum / (float)T) * p.ga_weight;
#if LANE_X_GAMMA_UC
    lin += p.g_vx * (gx - s_su2[0]);
    lin += p.g_wz * (gz - s_su2[2]);
    lin += p.g_vy * (gy - s_su2[1]);
#else
    lin += p.g_vx * gx;
    lin += p.g_wz * gz;
    lin += p.g_vy * gy;
#endif
    cost += uni + lin;
    if (pa_on) {
      const float c_pa = pa_num > 0.f ? pa_sum * fast_rcp(pa_num) : 0.f;
      cost += c_pa * p.pa_weight;
    }
    if (live) p.costs[b] = cost;

    // ---- softmax of the group (optimizer.cpp:382-391 as an online sum) --------------
    float cmin = live ? cost : 3.0e38f;
    for (int o = 32; o > 0; o >>= 1) cmin = fminf(cmin, __shfl_xor(cmin, o, WAVE));
    const float m_new = fminf(m_run, cmin);
    const float f = __builtin_amdgcn_exp2f(p.k2 * (m_run - m_new));
    const float w = live ? __builtin_amdgcn_exp2f(p.k2 * (cost - m_new)) : 0.f;
    float wsum = w;
    for (int o = 32; o > 0; o >>= 1) wsum += __shfl_xor(wsum, o, WAVE);
    s_run = fmaf(s_run, f, wsum);
    m_run = m_new;

    // ================= U[t] += sum_b w_b c[b][t]: transpose-reduce in registers ==========
    const LaneW lw = lane_weights(w, lane);
    if constexpr (RR) {
      // the group's noise again (it was read within the last tens of microseconds: served on
      // die), 64 steps of one control at a time, and c = u + n with the rounding of the step
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
#pragma unroll
        for (int ctrl = 0; ctrl < 3; ++ctrl) {
          // all 64 loads go out before the first use (left alone the scheduler pairs every load
          // with its add and waits for each in turn: 192 memory round trips per group)
          float V[64];
#pragma unroll
          for (int t = 0; t < 64; ++t) V[t] = LANE_X_RR_NOLOAD ? (float)t : ld(ctrl, 64u * h + t);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < 64; ++t) {
            const uint32_t tt = 64u * h + t;
            const float c = ldu(ctrl, tt) + V[t];
            V[t] = (FULL || tt < T) ? c : 0.f;
          }
          const float r = lane_reduce64(V, lw, lane);
          if (ctrl == 0) Ux[h] = fmaf(Ux[h], f, r);
          else if (ctrl == 1) Uy[h] = fmaf(Uy[h], f, r);
          else Uz[h] = fmaf(Uz[h], f, r);
        }
      }
    } else {
      {
        float V[64];
#pragma unroll
        for (int t = 0; t < 32; ++t) {
          V[t] = PX0[8 * (t & 3) + (t >> 2)];
          V[32 + t] = PX1[8 * (t & 3) + (t >> 2)];
        }
        Ux[0] = fmaf(Ux[0], f, lane_reduce64(V, lw, lane));
#pragma unroll
        for (int t = 0; t < 32; ++t) {
          V[t] = PY0[8 * (t & 3) + (t >> 2)];
          V[32 + t] = PY1[8 * (t & 3) + (t >> 2)];
        }
        Uy[0] = fmaf(Uy[0], f, lane_reduce64(V, lw, lane));
      }
      {
        // wz from the LDS slot: lane t walks its row of 64 rollouts
        s_w[lane] = w;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const f32x4* row = reinterpret_cast<const f32x4*>(park + lane * LANE_PARK_STRIDE);
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int bq = 0; bq < 16; bq += 2) {
          const f32x4 w0 = reinterpret_cast<const f32x4*>(s_w)[bq], c0 = row[bq];
          const f32x4 w1 = reinterpret_cast<const f32x4*>(s_w)[bq + 1], c1 = row[bq + 1];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc0 = fmaf(w0[e], c0[e], acc0);
            acc1 = fmaf(w1[e], c1[e], acc1);
          }
        }
        Uz[0] = fmaf(Uz[0], f, acc0 + acc1);
        __builtin_amdgcn_wave_barrier();
      }
    }
    return false;
  };

  stamp(2);
  int stamp_k = 3;
  for (uint32_t grp = gw; grp < ngroups; grp += nW) {
    if (__builtin_expect(group_body(std::false_type{}, grp), 0)) group_body(std::true_type{}, grp);
    if (stamp_k < 5) stamp(stamp_k++);
  }

  if (__builtin_expect(p.timeline != nullptr, 0) && lane == 0)
    p.timeline[8192 + blockIdx.x * 8 + wave] = __builtin_amdgcn_s_memtime();
  // ---- block combine -> one partial per block (same tuple as the wave-per-rollout pass)
  __syncthreads();
  stamp(5);
  const uint32_t TL = 4 + 3 * T;
  float* myp = reinterpret_cast<float*>(smem + L.off_scr) + (size_t)wave * L.scr_stride;
  if (lane == 0) {
    myp[0] = m_run;
    myp[1] = s_run;
    myp[2] = F_local;
    myp[3] = (float)n_noncoll;
  }
#pragma unroll
  for (int h = 0; h < NCH; ++h) {
    const uint32_t tt = 64u * h + (uint32_t)lane;
    if (tt < T) {
      myp[4 + tt] = Ux[h];
      myp[4 + T + tt] = Uy[h];
      myp[4 + 2 * T + tt] = Uz[h];
    }
  }
  __syncthreads();
  const float* allp = reinterpret_cast<const float*>(smem + L.off_scr);
  float bm = 3.0e38f;
  for (int w = 0; w < nwave; ++w) bm = fminf(bm, allp[(size_t)w * L.scr_stride]);
  float* outp = p.partials + (size_t)blockIdx.x * TL;
  for (uint32_t i = tid; i < TL; i += blockDim.x) {
    float acc = 0.f;
    if (i == 0) {
      acc = bm;
    } else if (i == 2) {
      for (int w = 0; w < nwave; ++w) acc = fmaxf(acc, allp[(size_t)w * L.scr_stride + 2]);
    } else if (i == 3) {
      for (int w = 0; w < nwave; ++w) acc += allp[(size_t)w * L.scr_stride + 3];
    } else {
      for (int w = 0; w < nwave; ++w) {
        const float mw = allp[(size_t)w * L.scr_stride];
        const float sc = __builtin_amdgcn_exp2f(p.k2 * (mw - bm));   // as the per-wave rescale above
        acc += sc * allp[(size_t)w * L.scr_stride + i];
      }
    }
    smpc_store_partial(outp + i, acc);
  }
  stamp(6);
  // Which tick block this launch read (SmpcDev::canary_echo): its number sits four floats in front
  // of u; block 0 leaves it behind the grid's partials for the reduction to hand to the host.
  // Unconditional wherever u is the tick block's own (the host knows when the word means nothing)
  // and through the two pointers the kernel holds anyway: a pointer or a flag of its own, live
  // across the time loop, cost the T = 64 instance 20 bytes of scratch.
  if (!(p.flags & SD_ACCUMULATE) && blockIdx.x == 0 && tid == 0) outp[SMPC_CANARY_SLOT(T)] = cu[-4];
  if constexpr (!MANY && !RR) {   // (the re-read form's grid is three blocks per CU: over SMPC_TAIL_MAX_GRID)
    if (p.tail) smpc_grid_tail<(RR ? LANE_BLOCK_RR : LANE_BLOCK) / 64>(p, smem);   // (the host: full-size blocks only)
  }
}

extern char smpc_last_pass_kernel[96];   // smpc_kernels.hip
// the instance's name with every template argument written out, as rocprofv3 prints it
static void lane_name(bool full, bool obst, bool many, int nch, bool rr, bool ga, bool quads, int tc, bool dep)
{
  auto b = [](bool v) {return v ? "true" : "false";};
  snprintf(smpc_last_pass_kernel, sizeof(smpc_last_pass_kernel), "smpc_pass_lane<%s, %s, %s, %d, %s, %s, %s, %d, %s>", b(full),
           b(obst), b(many), nch, b(rr), b(ga), b(quads), tc, b(dep));
}

// rr: the re-read instances (no parked controls; required for T > 64; ObstaclesCritic scored)
// block: threads per block of the parking form — LANE_BLOCK, or LANE_BLOCK / 2 for batches of at
// most one group per SIMD (a wave alone on its SIMD runs a group in 2/3 of the time)
hipError_t smpc_launch_pass_lane(const SmpcDev& p, const SmpcLds& L, uint32_t grid, bool rr, uint32_t block, hipStream_t st)
{
  if (!rr && block != LANE_BLOCK && block != LANE_BLOCK / 2) return hipErrorInvalidValue;
  const bool obst = (p.flags & (SD_OBSTACLES | SD_COST)) != 0;   // (Cost: the deployed-list instances, same lookup)
  if (rr) {
    // whole chunks only (T = 64 or 128): the ragged instances spill registers, and a spill in
    // the time loop costs the noise prefetch its depth (every scratch access waits vmcnt(0))
    if (!obst || (p.T != 64u && p.T != 128u)) return hipErrorInvalidValue;
#define SMPC_LANE_LAUNCH_RR(N) \
  hipLaunchKernelGGL((smpc_pass_lane<true, true, false, N, true>), dim3(grid), dim3(LANE_BLOCK_RR), L.total, st, p, L, \
                     static_cast<const SmpcDev*>(nullptr))
    if (p.T > 64u) SMPC_LANE_LAUNCH_RR(2);
    else SMPC_LANE_LAUNCH_RR(1);
#undef SMPC_LANE_LAUNCH_RR
    lane_name(true, true, false, p.T > 64u ? 2 : 1, true, false, true, 0, false);
    return hipGetLastError();
  }
  const bool full = p.T == 64u;
  if (p.flags & (SD_CONSTRAINT | SD_COST | SD_TWIRLING)) {   // cruise tick of the deployed critic list
    if (!obst || (p.flags & (SD_GOAL_ANGLE | SD_GOAL)) || ((p.flags & SD_COST) && (p.flags & SD_OBSTACLES)) ||
        p.con_acker_r >= 0.f || (!full && p.T != 56u))
      return hipErrorInvalidValue;
    if (full)
      hipLaunchKernelGGL((smpc_pass_lane<true, true, false, 1, false, false, true, 0, true>), dim3(grid), dim3(block), L.total,
                         st, p, L, static_cast<const SmpcDev*>(nullptr));
    else
      hipLaunchKernelGGL((smpc_pass_lane<false, true, false, 1, false, false, true, 56, true>), dim3(grid), dim3(block), L.total,
                         st, p, L, static_cast<const SmpcDev*>(nullptr));
    lane_name(full, true, false, 1, false, false, true, full ? 0 : 56, true);
    return hipGetLastError();
  }
  if (p.flags & SD_GOAL_ANGLE) {   // near-goal tick: the instances with the GoalAngle term
    if (!obst) return hipErrorInvalidValue;
#define SMPC_LANE_LAUNCH_GA(F) \
  hipLaunchKernelGGL((smpc_pass_lane<F, true, false, 1, false, true>), dim3(grid), dim3(block), L.total, st, p, L, \
                     static_cast<const SmpcDev*>(nullptr))
    if (full) SMPC_LANE_LAUNCH_GA(true);
    else SMPC_LANE_LAUNCH_GA(false);
#undef SMPC_LANE_LAUNCH_GA
    lane_name(full, true, false, 1, false, true, full, 0, false);
    return hipGetLastError();
  }
#define SMPC_LANE_LAUNCH(F, O) \
  hipLaunchKernelGGL((smpc_pass_lane<F, O, false, 1, false>), dim3(grid), dim3(block), L.total, st, p, L, \
                     static_cast<const SmpcDev*>(nullptr))
  if (full) lane_name(true, obst, false, 1, false, false, true, 0, false);
  else if (obst && p.T == 56u) lane_name(false, true, false, 1, false, false, true, 56, false);
  else if (obst && (p.T & 3u) == 0u) lane_name(false, true, false, 1, false, false, true, 0, false);
  else lane_name(false, obst, false, 1, false, false, false, 0, false);
  if (full && obst) SMPC_LANE_LAUNCH(true, true);
  else if (full) SMPC_LANE_LAUNCH(true, false);
  else if (obst && p.T == 56u)         // the reference's default horizon, at compile time
    hipLaunchKernelGGL((smpc_pass_lane<false, true, false, 1, false, false, true, 56>), dim3(grid), dim3(block), L.total, st, p, L,
                       static_cast<const SmpcDev*>(nullptr));
  else if (obst && (p.T & 3u) == 0u)   // whole quads
    hipLaunchKernelGGL((smpc_pass_lane<false, true, false, 1, false, false, true>), dim3(grid), dim3(block), L.total, st, p, L,
                       static_cast<const SmpcDev*>(nullptr));
  else if (obst) SMPC_LANE_LAUNCH(false, true);
  else SMPC_LANE_LAUNCH(false, false);
#undef SMPC_LANE_LAUNCH
  return hipGetLastError();
}

// n planning instances (same T, same critic set) in one launch; d_many: their parameter
// blocks in device memory
// dep: some instance scores Constraint / Cost / Twirling (the deployed critic list's cruise tick):
// the DEP instances, T = 64 (full) or 56; an instance without those critics runs them unchanged
hipError_t smpc_launch_pass_lane_many(const SmpcDev* d_many, uint32_t n, bool full, bool obst, bool dep, uint32_t T,
                                      const SmpcLds& L, uint32_t grid, uint32_t block, hipStream_t st)
{
  if (block != LANE_BLOCK && block != LANE_BLOCK / 2) return hipErrorInvalidValue;
  const SmpcDev none{};
  if (dep) lane_name(full, true, true, 1, false, false, true, full ? 0 : 56, true);
  else lane_name(full, obst, true, 1, false, false, full, 0, false);
  if (dep) {
    if (!obst || (!full && T != 56u)) return hipErrorInvalidValue;
    if (full)
      hipLaunchKernelGGL((smpc_pass_lane<true, true, true, 1, false, false, true, 0, true>), dim3(grid, n), dim3(block), L.total,
                         st, none, L, d_many);
    else
      hipLaunchKernelGGL((smpc_pass_lane<false, true, true, 1, false, false, true, 56, true>), dim3(grid, n), dim3(block), L.total,
                         st, none, L, d_many);
    return hipGetLastError();
  }
#define SMPC_LANE_LAUNCH(F, O) \
  hipLaunchKernelGGL((smpc_pass_lane<F, O, true, 1, false>), dim3(grid, n), dim3(block), L.total, st, none, L, \
                     d_many)
  if (full && obst) SMPC_LANE_LAUNCH(true, true);
  else if (full) SMPC_LANE_LAUNCH(true, false);
  else if (obst) SMPC_LANE_LAUNCH(false, true);
  else SMPC_LANE_LAUNCH(false, false);
#undef SMPC_LANE_LAUNCH
  return hipGetLastError();
}

uint32_t smpc_lane_block() {return LANE_BLOCK;}
uint32_t smpc_lane_block_rr() {return LANE_BLOCK_RR;}

static const void* lane_kernel(int k)   // bit 0 FULL, bit 1 OBST, bit 2 MANY; 8, 9: re-read with one, two chunks; 10, 11: GoalAngle
{
  switch (k) {
    case 0: return reinterpret_cast<const void*>(&smpc_pass_lane<false, false, false, 1, false>);
    case 1: return reinterpret_cast<const void*>(&smpc_pass_lane<true, false, false, 1, false>);
    case 2: return reinterpret_cast<const void*>(&smpc_pass_lane<false, true, false, 1, false>);
    case 3: return reinterpret_cast<const void*>(&smpc_pass_lane<true, true, false, 1, false>);
    case 4: return reinterpret_cast<const void*>(&smpc_pass_lane<false, false, true, 1, false>);
    case 5: return reinterpret_cast<const void*>(&smpc_pass_lane<true, false, true, 1, false>);
    case 6: return reinterpret_cast<const void*>(&smpc_pass_lane<false, true, true, 1, false>);
    case 7: return reinterpret_cast<const void*>(&smpc_pass_lane<true, true, true, 1, false>);
    case 8: return reinterpret_cast<const void*>(&smpc_pass_lane<true, true, false, 1, true>);
    case 9: return reinterpret_cast<const void*>(&smpc_pass_lane<true, true, false, 2, true>);
    case 10: return reinterpret_cast<const void*>(&smpc_pass_lane<false, true, false, 1, false, true>);   // near-goal
    case 11: return reinterpret_cast<const void*>(&smpc_pass_lane<true, true, false, 1, false, true>);
    case 12: return reinterpret_cast<const void*>(&smpc_pass_lane<false, true, false, 1, false, false, true>);   // whole quads
    case 13: return reinterpret_cast<const void*>(&smpc_pass_lane<false, true, false, 1, false, false, true, 56>);   // T = 56
    case 14: return reinterpret_cast<const void*>(&smpc_pass_lane<true, true, false, 1, false, false, true, 0, true>);   // deployed list
    case 15: return reinterpret_cast<const void*>(&smpc_pass_lane<false, true, false, 1, false, false, true, 56, true>);
    case 16: return reinterpret_cast<const void*>(&smpc_pass_lane<true, true, true, 1, false, false, true, 0, true>);    // ... grouped
    default: return reinterpret_cast<const void*>(&smpc_pass_lane<false, true, true, 1, false, false, true, 56, true>);
  }
}

hipError_t smpc_lane_occupancy(bool full, uint32_t lds_bytes, int* blocks_per_cu)
{
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, lane_kernel((full ? 1 : 0) | 2),
                                                      LANE_BLOCK, lds_bytes);
}

hipError_t smpc_lane_occupancy_rr(uint32_t T, uint32_t lds_bytes, int* blocks_per_cu)
{
  const int k = T > 64u ? 9 : 8;
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, lane_kernel(k), LANE_BLOCK_RR, lds_bytes);
}

hipError_t smpc_lane_set_lds_limit(int bytes)
{
  hipError_t e = hipSuccess;
  for (int k = 0; k < 18 && e == hipSuccess; ++k)
    e = hipFuncSetAttribute(lane_kernel(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  return e;
}
