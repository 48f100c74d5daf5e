// smpc_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the sampling-MPC hot path.
//
// One wavefront (64 lanes) owns one rollout at a time; lane l holds time steps
// [l*R, l*R+R) of the horizon in registers (R = ceil(T/64)).  The reference's
// [B,T] tensor passes (noise add, predict, integrate, critics, softmax update;
// reference src/optimizer.cpp:157-164,227-233,313-343,362-394) collapse into
// one streaming pass over the three noise tensors:
//
//   noise row (coalesced 256 B/array) -> c = u + n -> v = shift(c) ->
//   yaw = scan(wz dt) -> sincos -> x,y = scan(...) -> critics (costmap window
//   and path in LDS) -> cost -> online softmax accumulation of w*c per wave ->
//   per-block partial {min, sum w, sum w*c[3T]}.
//
// Cross-lane work uses DPP row shifts / broadcasts (no LDS traffic); there is
// no dense contraction anywhere, hence no MFMA: the bound is HBM (12*T bytes of
// noise per rollout) against FP32 VALU for the two transcendentals per step.
//
// Compiled with -ffp-contract=off so that a*b+c keeps the two roundings of the
// reference's separate xtensor passes.

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <stdint.h>

#include "smpc_dev.h"
#include "smpc_device_math.h"
#include "smpc_tail.h"

#define WAVE 64

// ---------------------------------------------------------------------------
// cross-lane helpers (DPP; gfx9 row_shr / row_bcast / wave_shr)
// ---------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float old, float v)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL,
                                                    ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_u(uint32_t old, uint32_t v)
{
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xF, false);
}

// Inclusive + scan over the 64 lanes (all lanes must be active): Kogge–Stone inside the
// 16-lane DPP rows, then row_bcast:15 / :31 carry the row totals.  Written as one asm
// block so that every step is a single v_add_f32_dpp (hipcc leaves the two broadcast steps
// as mov + add + re-zeroing) and the VALU-write -> DPP-read wait states are explicit.
#define SCAN_STEP1(op, ctl) op " %0, %0, %0 " ctl "\n\t"
__device__ __forceinline__ float wave_scan_add(float v)
{
  asm volatile(
    "s_nop 1\n\t"
    "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    "s_nop 1\n\t"
    "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    "s_nop 1\n\t"
    "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    "s_nop 1\n\t"
    "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    "s_nop 1\n\t"
    "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
    "s_nop 1\n\t"
    "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
    "s_nop 1\n\t"
    : "+v"(v));
  return v;
}
// two independent scans interleaved: one wait state between dependent steps instead of two
__device__ __forceinline__ void wave_scan_add2(float& a, float& b)
{
  asm volatile(
    "s_nop 1\n\t"
    "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    "s_nop 0\n\t"
    "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    "s_nop 0\n\t"
    "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    "s_nop 0\n\t"
    "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    "s_nop 0\n\t"
    "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
    "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
    "s_nop 0\n\t"
    "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
    "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
    "s_nop 1\n\t"
    : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float wave_sum(float v)
{
  v = wave_scan_add(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}
__device__ __forceinline__ uint32_t wave_min_u(uint32_t v)
{
  v = min(v, dpp_u<0x111>(0xffffffffu, v));
  v = min(v, dpp_u<0x112>(0xffffffffu, v));
  v = min(v, dpp_u<0x114>(0xffffffffu, v));
  v = min(v, dpp_u<0x118>(0xffffffffu, v));
  v = min(v, dpp_u<0x142, 0xA>(0xffffffffu, v));
  v = min(v, dpp_u<0x143, 0xC>(0xffffffffu, v));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// min of non-negative floats: unsigned min of the bit patterns (one v_min_u32_dpp per step)
__device__ __forceinline__ float wave_min_nonneg(float v)
{
  return __uint_as_float(wave_min_u(__float_as_uint(v)));
}
// shr1(src) + addend in ONE instruction: lane l gets src[l-1] + addend[l], lane 0 gets
// 0 + addend[0] (bound_ctrl).  With addend = {first, 0, 0, ...} it is the reference's
// "column 0 is the initial value, column t is the previous column" shift.
__device__ __forceinline__ float dpp_shr1_add(float src, float addend)
{
  float d;
  asm volatile("s_nop 1\n\t"
               "v_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
               : "=v"(d) : "v"(src), "v"(addend));
  return d;
}
// value of lane-1 (lane 0 gets `first`)
__device__ __forceinline__ float wave_shr1(float v, float first)
{
  return dpp_f<0x138>(first, v);  // wave_shr:1
}
__device__ __forceinline__ float lane_bcast(float v, int lane)
{
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// max over the parked endpoints of argmin_j |path_j - endpoint|^2 (first minimum wins), as the
// BITS of the float F = index + fraction (smpc_dev.h): F >= 0, so its bits order like the value
__device__ __forceinline__ float furthest_fraction(float d_j, float d_next, float seg2)
{
  // the endpoint's coordinate along the segment from point j to j + 1, in segment lengths:
  // (L^2 + d_j^2 - d_{j+1}^2) / (2 L^2); clamped so that the sum still rounds to j
  const float t = seg2 > 0.f ? 0.5f + 0.5f * (d_j - d_next) / seg2 : 0.f;
  return fminf(fmaxf(t, -0.45f), 0.45f);
}

__device__ __forceinline__ uint32_t flush_endpoint_ring(const float* ring_x, const float* ring_y,
                                                        uint32_t n, const float* s_px,
                                                        const float* s_py, uint32_t P, int lane)
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const bool on = (uint32_t)lane < n;
  const float ex = on ? ring_x[lane] : 0.f, ey = on ? ring_y[lane] : 0.f;
  float best = 3.4028234663852886e38f;  // numeric_limits<float>::max()
  uint32_t bi = 0;
  for (uint32_t j = 0; j < P; ++j) {
    const float ddx = s_px[j] - ex, ddy = s_py[j] - ey;
    const float d = ddx * ddx + ddy * ddy;
    if (d < best) {
      best = d;
      bi = j;
    }
  }
  float F = (float)bi;
  if (bi + 1 < P) {
    const float nx = s_px[bi + 1], ny = s_py[bi + 1];
    const float sx = nx - s_px[bi], sy = ny - s_py[bi];
    const float d_next = (nx - ex) * (nx - ex) + (ny - ey) * (ny - ey);
    F = fmaxf(F + furthest_fraction(best, d_next, sx * sx + sy * sy), 0.f);
  }
  uint32_t m = on ? __float_as_uint(F) : 0u;
  m = max(m, dpp_u<0x111>(0u, m));
  m = max(m, dpp_u<0x112>(0u, m));
  m = max(m, dpp_u<0x114>(0u, m));
  m = max(m, dpp_u<0x118>(0u, m));
  m = max(m, dpp_u<0x142, 0xA>(0u, m));
  m = max(m, dpp_u<0x143, 0xC>(0u, m));
  __builtin_amdgcn_wave_barrier();
  return (uint32_t)__builtin_amdgcn_readlane((int)m, 63);
}

// float min over the wave (any sign): one v_min_f32_dpp per step, no canonicalisation
__device__ __forceinline__ float wave_min_f(float v)
{
  asm volatile(
    "s_nop 1\n\t"
    "v_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
    "s_nop 1\n\t"
    "v_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
    "s_nop 1\n\t"
    "v_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
    "s_nop 1\n\t"
    "v_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
    "s_nop 1\n\t"
    "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
    "s_nop 1\n\t"
    "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
    "s_nop 1\n\t"
    : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// inclusive + scan inside segments of 16 << (seg_shift - 4) lanes (16, 32 or 64)
__device__ __forceinline__ float seg_scan_add(float v, uint32_t seg_shift)
{
  asm volatile(
    "s_nop 1\n\t"
    "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    "s_nop 1\n\t"
    "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    "s_nop 1\n\t"
    "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    "s_nop 1\n\t"
    "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    "s_nop 1\n\t"
    : "+v"(v));
  if (seg_shift >= 5) {
    asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 : "+v"(v));
  }
  if (seg_shift >= 6) {
    asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1\n\t"
                 : "+v"(v));
  }
  return v;
}

// ---------------------------------------------------------------------------
// The streaming pass.
//   MODE 0: score, every cost_power == 1 (the defaults): all per-step terms of
//           the additive critics go through ONE wave reduction
//   MODE 1: rollout + endpoint argmin only (utils::findPathFurthestReachedPoint,
//           tools/utils.hpp:292-319)
//   MODE 2: score, general cost_power (pow in double per critic)
//   MODE 3: MODE 0 plus the additive forms of CostCritic, GoalCritic, ConstraintCritic,
//           TwirlingCritic and PathAngleCritic (every cost_power == 1): the deployed critic
//           list (robot_bringup/config/nav2_params.yaml:222) in ONE wave reduction instead of
//           one double-precision reduction and one pow per critic
//   FULL:   T == 64*R, every lane owns R valid steps (no tail predicates)
//
// A wave rolls out one rollout at a time (lane = time step) and PARKS it: its noised
// controls go to an LDS ring, its PathAlign sample points to a slot table, its cost so
// far to a lane slot.  Every L.group rollouts the wave FLUSHES: PathAlign for the whole
// group with lane = (rollout, sample) so all 64 lanes work, one shared softmax rescale,
// then the weighted controls of the group are accumulated.
// ---------------------------------------------------------------------------
template <int R, int MODE, bool FULL>
__global__ void __launch_bounds__((R == 4 ? 512 : 1024), (R == 4 ? 2 : 4)) smpc_pass(const SmpcDev p, const SmpcLds L)
{
  constexpr bool FURTHEST_ONLY = MODE == 1;
  constexpr bool GENERIC = MODE == 2;
  // MODE 0 is the lean kernel: the rarely used features (trajectory write-out, path
  // orientations) live only in MODE 2, so their pointers and parameters do not occupy scalar
  // registers in the hot loop; the near-goal GoalAngle term is scored by MODE 3 (power 1) or 2
  constexpr bool RARE = MODE != 0 && MODE != 3;
  constexpr bool EXTRA = MODE == 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint8_t* s_map = smem;
  const SmpcLut* s_lut = reinterpret_cast<const SmpcLut*>(smem + L.off_lut);
  float* s_px = reinterpret_cast<float*>(smem + L.off_px);
  float* s_py = reinterpret_cast<float*>(smem + L.off_py);
  float* s_pyaw = reinterpret_cast<float*>(smem + L.off_pyaw);
  float* s_D = reinterpret_cast<float*>(smem + L.off_D);
  uint8_t* s_valid = smem + L.off_valid;

  const SmpcTickPtrs tk = smpc_tick_ptrs(p, true);
  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  // wave-uniform by construction: tell the compiler, so that the rollout index, the row
  // addresses and the loop control live on the scalar unit
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwave = blockDim.x >> 6;
  float* scr = reinterpret_cast<float*>(smem + L.off_scr) + (size_t)wave * L.scr_stride;
  float* pts_x = scr + L.scr_pts;        // [64] PathAlign sample points, slot = g*SEG + s
  float* pts_y = pts_x + WAVE;
  float* pts_yaw = pts_y + WAVE;
  float* ring_x = scr + L.scr_ring;      // [64] parked rollout endpoints
  float* ring_y = ring_x + WAVE;
  float* cring = scr + L.scr_c;          // [group][3][T] parked noised controls
  // which tick block this launch reads (SmpcDev::canary_echo): its number sits four floats in
  // front of u; left behind the partials for the reduction to hand to the host
  if (!FURTHEST_ONLY && p.canary_echo && !(p.flags & SD_ACCUMULATE) && blockIdx.x == 0 && tid == 0)
    p.partials[SMPC_CANARY_SLOT(p.T)] = p.u[-4];

  // ---- stage costmap window, LUT and path into LDS -------------------------
  if (!FURTHEST_ONLY && (p.flags & (SD_OBSTACLES | SD_COST))) {
    const int ww = p.win_w, wh = p.win_h;
    const bool vec = ((ww & 3) == 0) && ((p.W & 3u) == 0) && ((p.win_x0 & 3) == 0);
    if (vec) {
      const int w4 = ww >> 2;
      for (int i = tid; i < w4 * wh; i += blockDim.x) {
        const int ry = i / w4, rx = i - ry * w4;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(
          p.map + (size_t)(p.win_y0 + ry) * p.W + p.win_x0);
        reinterpret_cast<uint32_t*>(s_map)[ry * w4 + rx] = src[rx];
      }
    } else {
      for (int i = tid; i < ww * wh; i += blockDim.x) {
        const int ry = i / ww, rx = i - ry * ww;
        s_map[i] = p.map[(size_t)(p.win_y0 + ry) * p.W + p.win_x0 + rx];
      }
    }
    for (int i = tid; i < 256; i += blockDim.x)
      const_cast<SmpcLut*>(s_lut)[i] = p.lut[i];
  }
  for (uint32_t i = tid; i < p.P; i += blockDim.x) {
    s_px[i] = tk.px[i];
    s_py[i] = tk.py[i];
    if (!FURTHEST_ONLY) {
      s_pyaw[i] = tk.pyaw[i];
      if (i + 1 < p.P) {
        s_D[i] = tk.D[i];
        s_valid[i] = tk.pvalid[i];
      }
    }
  }
  __syncthreads();

  // ---- per-lane constants --------------------------------------------------
  const uint32_t T = p.T;
  const int t0 = lane * R;
#define STEP_OK(r) (FULL || (uint32_t)(t0 + (r)) < T)
  float uvx[R], uvy[R], uwz[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool a = STEP_OK(r);
    uvx[r] = a ? tk.u[t0 + r] : 0.f;
    uvy[r] = a ? tk.u[T + t0 + r] : 0.f;
    uwz[r] = a ? tk.u[2 * T + t0 + r] : 0.f;
  }
  const float dt = in_vgpr(p.dt);
  const float yaw0_v = in_vgpr(p.yaw0);
  const double x0_v = in_vgpr(p.x0), y0_v = in_vgpr(p.y0);
  const CellConsts cellk = {in_vgpr(p.oxf), in_vgpr(p.oyf), in_vgpr(p.rinvf), in_vgpr(p.cell_eps),
                            in_vgpr(1.0f - p.cell_eps)};
  const float obs_cw = in_vgpr(p.obs_critical_w), obs_rt = in_vgpr(p.obs_rep_over_T);
  const float pfw_w = in_vgpr(p.pfw_weight), pf_w = in_vgpr(p.pf_weight), pa_w = in_vgpr(p.pa_weight);
  const float k2_v = in_vgpr(p.k2);
  // {initial value in lane 0, 0 elsewhere}: addends of the fused shift (dpp_shr1_add)
  const float first_vx = lane == 0 ? p.svx : 0.f, first_vy = lane == 0 ? p.svy : 0.f;
  const float first_wz = lane == 0 ? p.swz : 0.f;
  const float first_cos = lane == 0 ? p.cos0 : 0.f, first_sin = lane == 0 ? p.sin0 : 0.f;
  float gux[R], guy[R], guz[R];   // gamma/std^2 * u (optimizer.cpp:367-379), MODE 0
#pragma unroll
  for (int r = 0; r < R; ++r) {
    gux[r] = p.g_vx * uvx[r];
    guy[r] = p.g_vy * uvy[r];
    guz[r] = p.g_wz * uwz[r];
  }

  uint32_t S = 0;       // batch-wide furthest point used for scoring
  bool pa_on = false;
  bool pal_on = false;  // PathAlignLegacyCritic (general pass only)
  float pf_x = 0.f, pf_y = 0.f;
  uint32_t bs_iters = 0;
  float pa_inv_spacing = 0.f;   // (S-1) / D[S-1]: mean inverse spacing of the plan
  if (!FURTHEST_ONLY) {
    if (p.flags & SD_NEED_FURTHEST) {
      S = p.d_furthest ? smpc_furthest_index(*p.d_furthest) : p.furthest_hint;
      if (S >= p.P) S = p.P ? p.P - 1 : 0;
    }
    pa_on = (p.flags & SD_PATH_ALIGN) && p.P > 0 && tk.pa_active[S] && S > 0;
    if (GENERIC) pal_on = (p.flags & SD_PATH_ALIGN_LEGACY) && p.P > 1 && tk.pal_active[S];
    if ((p.flags & SD_PATH_FOLLOW) && p.P > 0) {
      const uint32_t idx = tk.pf_idx[S];
      pf_x = s_px[idx];
      pf_y = s_py[idx];
    }
    bs_iters = S > 1 ? 32u - (uint32_t)__builtin_clz(S - 1) : 0u;  // ceil(log2 S)
    if (pa_on && S > 1 && s_D[S - 1] > 0.f) pa_inv_spacing = (float)(S - 1) / s_D[S - 1];
  }
  pf_x = in_vgpr(pf_x);
  pf_y = in_vgpr(pf_y);
  pa_inv_spacing = in_vgpr(pa_inv_spacing);
  const bool want_local_furthest =
    FURTHEST_ONLY || ((p.flags & SD_NEED_FURTHEST) && (p.flags & SD_LOCAL_FURTHEST));
  // group geometry: GROUP rollouts parked per flush, SEG lanes (sample slots) per rollout
  const uint32_t seg_shift = L.seg_shift, SEG = 1u << seg_shift, GROUP = L.group;
  const uint32_t K = p.nsamp;   // PathAlign samples: trajectory points step, 2 step, ..., K step
  // sample slot of each step this lane owns: t = s*step, s = 0..K  (-1: not a sample)
  int slot[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t t = t0 + r;
    const uint32_t q = p.step ? t / p.step : 0u;
    slot[r] = ((pa_on || pal_on) && p.step && STEP_OK(r) && q * p.step == t && q <= K) ? (int)q : -1;
  }
  const uint32_t end_lane = (T - 1) / R, end_r = (T - 1) % R;

  // running softmax state of this wave (optimizer.cpp:382-391 as an online sum)
  float m_run = 3.0e38f, s_run = 0.f;
  float Ux[R], Uy[R], Uz[R];
#pragma unroll
  for (int r = 0; r < R; ++r) Ux[r] = Uy[r] = Uz[r] = 0.f;
  uint32_t S_local = 0, n_noncoll = 0;
  uint32_t n_ring = 0, n_pend = 0;
  float pend_cost = 0.f;        // lanes of segment g: cost so far of parked rollout g

  const uint32_t gw = blockIdx.x * nwave + wave;
  const uint32_t nW = gridDim.x * nwave;

  // flush of the parked group: PathAlign, costs, softmax, weighted controls
  auto flush = [&](uint32_t n, uint32_t b_last) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t g = (uint32_t)lane >> seg_shift, s = (uint32_t)lane & (SEG - 1);
    const bool rowon = g < n;
    float cost = pend_cost;
    // ---- PathAlignCritic (path_align_critic.cpp:92-135), lane = (rollout g, sample s) ----
    if (pa_on) {
      const bool smp = rowon && s >= 1 && s <= K;
      const float Tx = pts_x[lane], Ty = pts_y[lane];
      const float Qx = wave_shr1(Tx, 0.f), Qy = wave_shr1(Ty, 0.f);   // previous sample point
      float chord = 0.f;
      if (smp) {
        const float ddx = Tx - Qx, ddy = Ty - Qy;
        chord = GENERIC ? sqrtf(ddx * ddx + ddy * ddy) : fast_sqrt(ddx * ddx + ddy * ddy);
      }
      const float dist = seg_scan_add(chord, seg_shift);
      // std::lower_bound over D[0..S).  Plans are close to uniformly spaced, so the
      // index is guessed from the mean spacing and confirmed against D[g-1], D[g],
      // D[g+1]; a wave with any unconfirmed lane runs the branch-free binary search.
      uint32_t gi = (uint32_t)(dist * pa_inv_spacing);
      gi = gi < S ? gi : S - 1;
      const float da = gi > 0 ? s_D[gi - 1] : -3.0e38f;
      const float db = s_D[gi];
      const float dc = gi + 1 < S ? s_D[gi + 1] : 3.0e38f;
      uint32_t lo;
      float dl, dh;   // D[lo-1], D[lo]
      const bool at_g = da < dist && !(db < dist);
      const bool at_g1 = db < dist && !(dc < dist);
      if (at_g) {
        lo = gi; dl = da; dh = db;
      } else {
        lo = gi + 1; dl = db; dh = dc;
      }
      if (__builtin_expect(__any(!(at_g || at_g1)), 0)) {
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
      uint32_t cand;
      if (lo >= S) {
        cand = S - 1;  // reference dereferences end(): defined as size-1
      } else if (lo == 0) {
        cand = 0;
      } else {
        cand = (dist - dl < dh - dist) ? lo - 1 : lo;
      }
      // findClosestPathPt's `iter == begin + init -> return 0` chains through path_pt:
      // path_pt_k = 0 where lo_k == path_pt_{k-1}.  With E_k = (lo_k == cand_{k-1} != 0 ...)
      // the chain F_k = E_k & ~F_{k-1} is "every other bit of each run of ones"; slot 0 of
      // a segment is never a sample, so runs do not cross rollouts.
      uint32_t cand_v = cand;
      asm volatile("" : "+v"(cand_v));   // DPP sources must sit in a VGPR
      const uint32_t cand_prev = dpp_u<0x138>(0u, cand_v);   // wave_shr:1
      unsigned long long E = __ballot(smp && lo == cand_prev && cand != 0u);
      uint32_t pt = cand;
      if (E) {
        unsigned long long F = E & ~(E << 1);
        unsigned long long M = E & (E << 1);
        F |= (F << 2) & M;
        M &= M << 2;
        F |= (F << 4) & M;
        M &= M << 4;
        F |= (F << 8) & M;
        M &= M << 8;
        F |= (F << 16) & M;
        M &= M << 16;
        F |= (F << 32) & M;
        if ((F >> lane) & 1ull) pt = 0;
      }
      const bool ok = smp && s_valid[pt];
      float d = 0.f;
      if (ok) {
        const float ddx = s_px[pt] - Tx, ddy = s_py[pt] - Ty;
        if (RARE && (p.flags & SD_USE_PATH_YAW)) {
          const double dd = (double)pts_yaw[lane] - (double)s_pyaw[pt];
          double a = fmod(fmod(dd, 2.0 * M_PI) + 2.0 * M_PI, 2.0 * M_PI);
          if (a > M_PI) a -= 2.0 * M_PI;
          const float dyaw = (float)a;
          d = sqrtf(ddx * ddx + ddy * ddy + dyaw * dyaw);
        } else {
          d = GENERIC ? sqrtf(ddx * ddx + ddy * ddy) : fast_sqrt(ddx * ddx + ddy * ddy);
        }
      }
      // per-rollout sum and count: last lane of the segment after a segmented scan
      const float dsum = seg_scan_add(d, seg_shift);
      const float summed = __shfl(dsum, lane | (int)(SEG - 1), WAVE);
      const unsigned long long okm = __ballot(ok) >> (g << seg_shift);
      const unsigned long long segm = SEG == 64 ? ~0ull : ((1ull << SEG) - 1ull);
      const float num = (float)__popcll(okm & segm);
      const float c_pa = num > 0.f ? (GENERIC ? summed / num : summed * fast_rcp(num)) : 0.f;
      if (GENERIC) cost = add_cost_pow(cost, (double)(c_pa * p.pa_weight), p.pa_power);
      else cost += c_pa * pa_w;
    }
    // ---- PathAlignLegacyCritic (path_align_legacy_critic.cpp:74-129), lane = (rollout g, sample s):
    // every trajectory point step, 2 step, ... against its nearest path point by brute force over
    // s' = 0 .. P - 3 (the loop runs to path_segments_count - 1 EXCLUSIVE, :101), first minimum
    // wins; the point counts unless it is point 0 or invalid (:119-121); the sum is divided by
    // floor(T / step), not by the count (:124)
    if (GENERIC && pal_on) {
      const bool smp = rowon && s >= 1 && s <= K;
      const float Tx = pts_x[lane], Ty = pts_y[lane];
      float min_dist_sq = 3.4028234663852886e38f;
      uint32_t min_s = 0;
      const uint32_t n_pts = p.P - 2u;      // (P >= 2: the host's gate)
      for (uint32_t q = 0; q < n_pts; ++q) {
        const float dx = s_px[q] - Tx, dy = s_py[q] - Ty;
        float dist_sq = dx * dx + dy * dy;
        if (p.flags & SD_PAL_USE_PATH_YAW) {
          // angles::shortest_angular_distance(P_yaw(s), T_yaw(t, p)) = normalize_angle(to - from)
          const double dd = (double)pts_yaw[lane] - (double)s_pyaw[q];
          double a = fmod(fmod(dd, 2.0 * M_PI) + 2.0 * M_PI, 2.0 * M_PI);
          if (a > M_PI) a -= 2.0 * M_PI;
          const float dyaw = (float)a;
          dist_sq = dx * dx + dy * dy + dyaw * dyaw;
        }
        if (dist_sq < min_dist_sq) {
          min_dist_sq = dist_sq;
          min_s = q;
        }
      }
      const bool ok = smp && min_s != 0u && s_valid[min_s];
      const float d = ok ? sqrtf(min_dist_sq) : 0.f;
      const float dsum = seg_scan_add(d, seg_shift);
      const float summed = __shfl(dsum, lane | (int)(SEG - 1), WAVE);
      const float c_pal = summed / p.pal_eval;
      cost = add_cost_pow(cost, (double)(c_pal * p.pal_weight), p.pal_power);
    }
    // costs_ [B]: one lane per parked rollout
    if (rowon && s == 0) p.costs[b_last - (n - 1 - g) * nW] = cost;

    // ---- softmax over the group (optimizer.cpp:382-391 as an online sum) ----------
    const float cmin = wave_min_f(rowon ? cost : 3.0e38f);
    const float m_new = fminf(m_run, cmin);
    // exp(-(c - m)/temperature) as one v_exp_f32: 2^(k2 (c - m)), k2 = -log2(e)/temperature
    const float f = __builtin_amdgcn_exp2f(k2_v * (m_run - m_new));   // rescale old sums (<= 1)
    const float w = rowon ? __builtin_amdgcn_exp2f(k2_v * (cost - m_new)) : 0.f;
    s_run = fmaf(s_run, f, wave_sum(s == 0 ? w : 0.f));
#pragma unroll
    for (int r = 0; r < R; ++r) {
      Ux[r] *= f;
      Uy[r] *= f;
      Uz[r] *= f;
    }
    for (uint32_t gg = 0; gg < n; ++gg) {
      const float wg = lane_bcast(w, (int)(gg << seg_shift));
      const float* cg = cring + (size_t)gg * 3 * T;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (STEP_OK(r)) {
          Ux[r] = fmaf(wg, cg[t0 + r], Ux[r]);
          Uy[r] = fmaf(wg, cg[T + t0 + r], Uy[r]);
          Uz[r] = fmaf(wg, cg[2 * T + t0 + r], Uz[r]);
        }
      }
    }
    m_run = m_new;
    __builtin_amdgcn_wave_barrier();
  };

  // noise rows are prefetched one rollout ahead
  float n0[R], n1[R], n2[R];
  if (gw < p.B) {
    const size_t row = (size_t)gw * T;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool a = STEP_OK(r);
      n0[r] = a ? p.nvx[row + t0 + r] : 0.f;
      n1[r] = a ? p.nvy[row + t0 + r] : 0.f;
      n2[r] = a ? p.nwz[row + t0 + r] : 0.f;
    }
  }

  uint32_t b_prev = gw;
  for (uint32_t b = gw; b < p.B; b += nW) {
    b_prev = b;
    // ---- NoiseGenerator::setNoisedControls (noise_generator.cpp:65-74) -----
    const size_t row = (size_t)b * T;
    float cvx[R], cvy[R], cwz[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      cvx[r] = uvx[r] + n0[r];
      cvy[r] = uvy[r] + n1[r];
      cwz[r] = uwz[r] + n2[r];
    }
    if (b + nW < p.B) {
      const size_t nrow = (size_t)(b + nW) * T;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const bool a = STEP_OK(r);
        n0[r] = a ? p.nvx[nrow + t0 + r] : 0.f;
        n1[r] = a ? p.nvy[nrow + t0 + r] : 0.f;
        n2[r] = a ? p.nwz[nrow + t0 + r] : 0.f;
      }
    }
    // ---- updateStateVelocities + predict: v[:,0]=speed, v[:,1:]=c[:,:-1] ---
    float vx[R], vy[R], wz[R];
    vx[0] = dpp_shr1_add(cvx[R - 1], first_vx);
    vy[0] = dpp_shr1_add(cvy[R - 1], first_vy);
    wz[0] = dpp_shr1_add(cwz[R - 1], first_wz);
#pragma unroll
    for (int r = 1; r < R; ++r) {
      vx[r] = cvx[r - 1];
      vy[r] = cvy[r - 1];
      wz[r] = cwz[r - 1];
    }
    if (!FULL) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (!STEP_OK(r)) vx[r] = vy[r] = wz[r] = 0.f;
      }
    }
    // ---- integrateStateVelocities (optimizer.cpp:313-343) -------------------
    float yaw[R];
    {
      float acc = wz[0] * dt;
      yaw[0] = acc;
#pragma unroll
      for (int r = 1; r < R; ++r) {
        acc += wz[r] * dt;
        yaw[r] = acc;
      }
      const float incl = wave_scan_add(acc);
      if (R == 1) {
        yaw[0] = incl + yaw0_v;   // one step per lane: the inclusive scan is the cumsum
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) yaw[r] = dpp_shr1_add(incl, yaw[r]) + yaw0_v;
      }
    }
    float x[R], y[R];
    {
      float sn[R], cs[R];
#pragma unroll
      for (int r = 0; r < R; ++r) smpc_sincos(yaw[r], sn[r], cs[r]);
      // cos_[t] = cos(yaw[t-1]), cos_[0] = cosf(initial_yaw)
      float c_prev = dpp_shr1_add(cs[R - 1], first_cos);
      float s_prev = dpp_shr1_add(sn[R - 1], first_sin);
      float ax = 0.f, ay = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float dxr = vx[r] * c_prev - vy[r] * s_prev;
        const float dyr = vx[r] * s_prev + vy[r] * c_prev;
        ax = (r == 0) ? dxr * dt : ax + dxr * dt;
        ay = (r == 0) ? dyr * dt : ay + dyr * dt;
        x[r] = ax;
        y[r] = ay;
        c_prev = cs[r];
        s_prev = sn[r];
      }
      float sx = ax, sy = ay;
      wave_scan_add2(sx, sy);
      if (R == 1) {
        x[0] = (float)(x0_v + (double)sx);
        y[0] = (float)(y0_v + (double)sy);
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          x[r] = (float)(x0_v + (double)dpp_shr1_add(sx, x[r]));
          y[r] = (float)(y0_v + (double)dpp_shr1_add(sy, y[r]));
        }
      }
    }
    if (!FURTHEST_ONLY && (RARE && (p.flags & SD_STORE_TRAJ))) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (STEP_OK(r)) {
          p.traj_x[row + t0 + r] = x[r];
          p.traj_y[row + t0 + r] = y[r];
          p.traj_yaw[row + t0 + r] = yaw[r];
        }
      }
    }
    // endpoint of the rollout (uniform): lane end_lane, step end_r
    float ex, ey;
    {
      float xe = x[0], ye = y[0];
#pragma unroll
      for (int r = 1; r < R; ++r) {
        if ((uint32_t)r == end_r) {
          xe = x[r];
          ye = y[r];
        }
      }
      ex = lane_bcast(xe, (int)end_lane);
      ey = lane_bcast(ye, (int)end_lane);
    }

    // ---- nearest path point of the endpoint (utils.hpp:292-319) -----------
    // Only max_b argmin_j matters and nothing in this rollout's cost depends on it, so
    // the endpoints are parked in a 64-slot LDS ring and resolved 64 rollouts at a time
    // with lane = rollout (sequential j loop: the reference's first-minimum order).
    if (want_local_furthest) {
      if (lane == 0) {
        ring_x[n_ring] = ex;
        ring_y[n_ring] = ey;
      }
      if (++n_ring == WAVE) {
        S_local = max(S_local, flush_endpoint_ring(ring_x, ring_y, WAVE, s_px, s_py, p.P, lane));
        n_ring = 0;
      }
    }
    if (FURTHEST_ONLY) continue;

    float cost = (p.flags & SD_ACCUMULATE) ? p.costs_prev[b] : 0.f;
    float lin = 0.f;    // MODE 0: per-lane sum of every additive per-step term
    float uni = 0.f;    // MODE 0: wave-uniform terms
    double lin_d = 0.0; // MODE 3: per-lane sum of the terms the reference forms in double

    // ---- ConstraintCritic (constraint_critic.cpp:41-75); MODE 2 only.  With the Ackermann
    //      model (con_acker_r >= 0) each step also pays min_turning_r - |vx|/|wz| (:54-69;
    //      float quotient, xt::maximum = select(a > b, a, b), so the 0/0 of a robot at rest
    //      adds nothing) ----
    if (RARE && (p.flags & SD_CONSTRAINT)) {
      double sa = 0.0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (STEP_OK(r)) {
          // xt::where(vx > 0.0, 1.0, -1.0) is a double tensor: double from there on
          const double sgn = vx[r] > 0.0f ? 1.0 : -1.0;
          const double vel_total = sgn * (double)sqrtf(vx[r] * vx[r] + vy[r] * vy[r]);
          const double out_max = fmax(vel_total - (double)p.con_max_vel, 0.0);
          const double out_min = fmax((double)p.con_min_vel - vel_total, 0.0);
          if (p.con_acker_r >= 0.f) {
            const double q = (double)(p.con_acker_r - fabsf(vx[r]) / fabsf(wz[r]));
            sa += (out_max + out_min + (q > 0.0 ? q : 0.0)) * (double)p.dt;
          } else {
            sa += (out_max + out_min) * (double)p.dt;
          }
        }
      }
      cost = add_cost_pow(cost, wave_sum_d(sa) * (double)p.con_weight, p.con_power);
    }

    if (EXTRA && (p.flags & SD_CONSTRAINT)) {
      // the same per-step double arithmetic; with cost_power == 1 the critic's total is additive
      const double kw = (double)p.dt * (double)p.con_weight;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (STEP_OK(r)) {
          const double sgn = vx[r] > 0.0f ? 1.0 : -1.0;
          const double vel_total = sgn * (double)sqrtf(vx[r] * vx[r] + vy[r] * vy[r]);
          double e = fmax(vel_total - (double)p.con_max_vel, 0.0) + fmax((double)p.con_min_vel - vel_total, 0.0);
          if (p.con_acker_r >= 0.f) {
            const double q = (double)(p.con_acker_r - fabsf(vx[r]) / fabsf(wz[r]));
            e += q > 0.0 ? q : 0.0;
          }
          lin_d += e * kw;
        }
      }
    }

    // ---- consider_footprint = true for either collision critic: MODE 2 only ----------
    // (obstacles_critic.cpp:139-171,203-224; cost_critic.cpp:128-166,175-201.)  The centre
    // cost is looked up as always; a step whose centre cost reaches the possibly-inscribed
    // cost gets the SE2 footprint cost.  Obstacles scores the footprint cost, Cost scores the
    // centre cost and only collision-checks with the footprint; each critic keeps its own
    // first collision.
    if (RARE && (p.flags & (SD_FP_OBSTACLES | SD_FP_COST))) {
      float crit = 0.f, rep = 0.f, crep = 0.f;
      int first_o = R, first_c = R;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (!STEP_OK(r)) continue;
        const uint32_t c = (t0 + r == 0) ? p.cost_t0 : cost_at(p, cellk, s_map, x[r], y[r]);
        // off the map: NO_INFORMATION without a footprint check (costAtPose returns early)
        uint32_t mxe = 0, mye = 0;
        bool on = cell_index_exact((double)x[r], p.ox, p.res, p.W, mxe);
        on = cell_index_exact((double)y[r], p.oy, p.res, p.H, mye) && on;
        // Obstacles checks the footprint inside costAtPose (on-map centres only); Cost checks it
        // inside inCollision, also for an off-map centre (cost NO_INFORMATION)
        const bool cond = (float)c >= p.fp_pic || p.fp_pic < 1.0f;
        const bool want_o = (p.flags & SD_OBSTACLES) && (p.flags & SD_FP_OBSTACLES) && first_o == R &&
                            cond && on;
        const bool want_c = (p.flags & SD_COST) && (p.flags & SD_FP_COST) && first_c == R && cond &&
                            c >= 1u;
        float cf = (float)c;
        if (want_o || want_c) cf = footprint_cost_at_pose(p, s_map, x[r], y[r], yaw[r]);
        if ((p.flags & SD_OBSTACLES) && first_o == R) {
          const bool fpo = (p.flags & SD_FP_OBSTACLES) != 0;
          const uint32_t co = want_o ? (uint32_t)cf : c;
          const SmpcLut e = fpo ? p.lut_fp[(want_o ? 256u : 0u) + co] : s_lut[co];
          if (e.crit < 0.f) {
            first_o = r;
          } else {
            crit += e.crit;
            rep += e.rep;
          }
        }
        if ((p.flags & SD_COST) && first_c == R && c >= 1u) {
          const bool fpc = (p.flags & SD_FP_COST) != 0;
          const uint32_t cc = want_c ? (uint32_t)cf : c;
          const float marker = fpc ? p.lut_fp[cc].crit : s_lut[cc].crit;
          if (marker < 0.f) first_c = r;
          else crep += p.lut_cost[c];
        }
      }
      bool coll_o = false, coll_c = false;
      if (p.flags & SD_OBSTACLES) {
        const unsigned long long cm = __ballot(first_o < R);
        coll_o = cm != 0ull;
        if (coll_o) {
          const int lc = __ffsll((long long)cm) - 1;
          if (lane > lc) rep = 0.f;
        }
      }
      if (p.flags & SD_COST) {
        const unsigned long long cm = __ballot(first_c < R);
        coll_c = cm != 0ull;
        if (coll_c) {
          const int lc = __ffsll((long long)cm) - 1;
          if (lane > lc) crep = 0.f;
        }
      }
      // the critic that would stop the manager decides the batch-wide fail flag: Cost first
      if (!((p.flags & SD_COST) ? coll_c : coll_o)) n_noncoll++;
      if (p.flags & SD_COST) {
        const float repulsive = coll_c ? p.cost_collision_cost : wave_sum(crep);
        const float v = p.cost_w254 * repulsive / (float)T;
        cost = add_cost_pow(cost, (double)v, p.cost_power);
      }
      if (p.flags & SD_OBSTACLES) {
        const float rep_sum = wave_sum(rep);
        const float raw = coll_o ? p.obs_collision_cost : wave_sum(crit);
        const float v = (p.obs_critical_w * raw) + (p.obs_repulsion_w * rep_sum / (float)T);
        cost = add_cost_pow(cost, (double)v, p.obs_power);
      }
    } else
    // ---- costmap lookups shared by CostCritic and ObstaclesCritic ----------------
    // (cost_critic.cpp:128-166, obstacles_critic.cpp:114-178: the same costAtPose and the same
    // inCollision switch in point mode, so one lookup and one first-collision search)
    if (p.flags & (SD_OBSTACLES | SD_COST)) {
      float crit = 0.f, rep = 0.f, crep = 0.f;
      int first_r = R;  // first colliding step inside this lane
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (STEP_OK(r) && first_r == R) {
          // point 0 is the same for every rollout (v[:,0] is the measured speed): its
          // cell was looked up once on the host with the same arithmetic
          const uint32_t c = (t0 + r == 0) ? p.cost_t0 : cost_at(p, cellk, s_map, x[r], y[r]);
          const SmpcLut e = s_lut[c];
          if (e.crit < 0.f) {   // inCollision
            first_r = r;
          } else {
            crit += e.crit;
            rep += e.rep;
            if (RARE && (p.flags & SD_COST)) crep += p.lut_cost[c];
            // cost_critic.cpp:141-155 per 8-bit cost, as arithmetic (the table sits in global memory)
            if (EXTRA && (p.flags & SD_COST) && c >= 1u)
              crep += c >= 253u ? p.cost_critical : (p.cost_near_goal ? 0.f : (float)c);
          }
        }
      }
      const unsigned long long cm = __ballot(first_r < R);
      const bool collided = cm != 0ull;
      if (collided) {
        const int lc = __ffsll((long long)cm) - 1;
        if (lane > lc) rep = 0.f;  // steps after the first collision are never visited
      } else {
        n_noncoll++;
      }
      if (RARE && (p.flags & SD_COST)) {
        // repulsive_cost is a sum of 8-bit costs / critical_cost: exact in float in any order
        const float repulsive = collided ? p.cost_collision_cost : wave_sum(crep);
        const float v = p.cost_w254 * repulsive / (float)T;
        cost = add_cost_pow(cost, (double)v, p.cost_power);
      }
      if (EXTRA && (p.flags & SD_COST)) {
        const float k = p.cost_w254 / (float)T;
        if (collided) uni += k * p.cost_collision_cost;
        else lin += k * crep;
      }
      if (p.flags & SD_OBSTACLES) {
        if (GENERIC) {
          const float rep_sum = wave_sum(rep);
          const float raw = collided ? p.obs_collision_cost : wave_sum(crit);
          const float v = (p.obs_critical_w * raw) + (p.obs_repulsion_w * rep_sum / (float)T);
          cost = add_cost_pow(cost, (double)v, p.obs_power);
        } else {
          // (added, not assigned: with CostCritic in the list too its terms are already in here)
          lin += (collided ? 0.f : obs_cw * crit) + obs_rt * rep;
          uni += collided ? p.obs_critical_w * p.obs_collision_cost : 0.f;
        }
      }
    }

    // ---- PathFollowCritic (path_follow_critic.cpp:56-70) ---------------------
    if (p.flags & SD_PATH_FOLLOW) {
      const float fdx = ex - pf_x, fdy = ey - pf_y;
      if (GENERIC) {
        const double ddx = (double)fdx, ddy = (double)fdy;
        const double dist = sqrt(ddx * ddx + ddy * ddy);
        cost = add_cost_pow(cost, (double)p.pf_weight * dist, p.pf_power);
      } else {
        uni += pf_w * fast_sqrt(fdx * fdx + fdy * fdy);
      }
    }

    // ---- GoalAngleCritic (goal_angle_critic.cpp:36-50) -----------------------
    if (RARE && (p.flags & SD_GOAL_ANGLE)) {
      double sa = 0.0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (STEP_OK(r)) sa += fabs(normalize_angle((double)(p.ga_goal_yaw - yaw[r])));
      }
      const double mean = wave_sum_d(sa) / (double)T;
      if (GENERIC) cost = add_cost_pow(cost, mean * (double)p.ga_weight, p.ga_power);
      else uni += (float)(mean * (double)p.ga_weight);
    }

    // ---- PreferForwardCritic (prefer_forward_critic.cpp:33-47) ---------------
    if (p.flags & SD_PREFER_FORWARD) {
      float sb = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) sb += fmaxf(-vx[r], 0.f) * dt;
      if (GENERIC) cost = add_cost_pow(cost, (double)(wave_sum(sb) * p.pfw_weight), p.pfw_power);
      else lin += sb * pfw_w;
    }

    // ---- GoalCritic, PathAngleCritic, TwirlingCritic, VelocityDeadbandCritic: MODE 2 only ----
    if (RARE && (p.flags & SD_GOAL)) {
      // goal.position is double: distances in double, xt::mean in double (goal_critic.cpp:44-54)
      double sa = 0.0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (STEP_OK(r)) {
          const double ddx = (double)x[r] - p.goal_x, ddy = (double)y[r] - p.goal_y;
          sa += sqrt(ddx * ddx + ddy * ddy);
        }
      }
      cost = add_cost_pow(cost, wave_sum_d(sa) / (double)T * (double)p.goal_weight, p.goal_power);
    }
    if (RARE && (p.flags & SD_PATH_ANGLE) && tk.pang_active[S]) {
      // path_angle_critic.cpp:72-100: float atan2, then the double angle arithmetic of
      // utils::shortest_angular_distance / normalize_angles (tools/utils.hpp:258-284)
      const uint32_t idx = min(S + p.pang_offset, p.P - 1);
      const float tgx = s_px[idx], tgy = s_py[idx];
      double sa = 0.0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (STEP_OK(r)) {
          const float ybp = atan2f(tgy - y[r], tgx - x[r]);
          const double d = fabs(normalize_angle((double)(ybp - yaw[r])));
          if (p.pang_correct) {
            const double ybp_c = d < M_PI_2 ? (double)ybp : normalize_angle((double)ybp + M_PI);
            sa += fabs(normalize_angle(ybp_c - (double)yaw[r]));
          } else {
            sa += d;
          }
        }
      }
      cost = add_cost_pow(cost, wave_sum_d(sa) / (double)T * (double)p.pang_weight, p.pang_power);
    }
    if (RARE && (p.flags & SD_TWIRLING)) {
      double sa = 0.0;   // twirling_critic.cpp:40-41
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (STEP_OK(r)) sa += (double)fabsf(wz[r]);
      }
      cost = add_cost_pow(cost, wave_sum_d(sa) / (double)T * (double)p.tw_weight, p.tw_power);
    }
    if (RARE && (p.flags & SD_DEADBAND)) {
      double sa = 0.0;   // velocity_deadband_critic.cpp:52-76 (::fabs(double): double arithmetic)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (STEP_OK(r)) {
          sa += (fmax(p.db_vx - (double)fabsf(vx[r]), 0.0) + fmax(p.db_vy - (double)fabsf(vy[r]), 0.0) +
                 fmax(p.db_wz - (double)fabsf(wz[r]), 0.0)) * (double)p.dt;
        }
      }
      cost = add_cost_pow(cost, wave_sum_d(sa) * (double)p.db_weight, p.db_power);
    }

    if (EXTRA) {
      if (p.flags & SD_GOAL_ANGLE) {
        // GoalAngleCritic with cost_power 1 (goal_angle_critic.cpp:36-50): the mean over the steps
        // of |shortest_angular_distance(yaw, goal yaw)|, as a sum of per-step shares like the
        // other additive critics of this pass — a tick inside the goal thresholds stays on the
        // lean kernel instead of falling to the general one (3.7 ms against 0.43 ms for
        // 2 097 152 x 64 when it did, tools/near_goal_tick.py)
        const double kw = (double)p.ga_weight / (double)T;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (STEP_OK(r)) lin_d += fabs(normalize_angle((double)(p.ga_goal_yaw - yaw[r]))) * kw;
        }
      }
      if (p.flags & SD_GOAL) {
        const double kw = (double)p.goal_weight / (double)T;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (STEP_OK(r)) {
            const double ddx = (double)x[r] - p.goal_x, ddy = (double)y[r] - p.goal_y;
            lin_d += sqrt(ddx * ddx + ddy * ddy) * kw;
          }
        }
      }
      if ((p.flags & SD_PATH_ANGLE) && tk.pang_active[S]) {
        const uint32_t idx = min(S + p.pang_offset, p.P - 1);
        const float tgx = s_px[idx], tgy = s_py[idx];
        const double kw = (double)p.pang_weight / (double)T;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (STEP_OK(r)) {
            const float ybp = atan2f(tgy - y[r], tgx - x[r]);
            const double d = fabs(normalize_angle((double)(ybp - yaw[r])));
            if (p.pang_correct) {
              const double ybp_c = d < M_PI_2 ? (double)ybp : normalize_angle((double)ybp + M_PI);
              lin_d += fabs(normalize_angle(ybp_c - (double)yaw[r])) * kw;
            } else {
              lin_d += d * kw;
            }
          }
        }
      }
      if (p.flags & SD_TWIRLING) {
        const double kw = (double)p.tw_weight / (double)T;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (STEP_OK(r)) lin_d += (double)fabsf(wz[r]) * kw;
        }
      }
    }

    // ---- updateControlSequence gamma terms (optimizer.cpp:365-380) -----------
    if (GENERIC) {
      float gx = 0.f, gz = 0.f, gy = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        gx += uvx[r] * (cvx[r] - uvx[r]);
        gz += uwz[r] * (cwz[r] - uwz[r]);
        gy += uvy[r] * (cvy[r] - uvy[r]);
      }
      cost += p.g_vx * wave_sum(gx);
      cost += p.g_wz * wave_sum(gz);
      cost += p.g_vy * wave_sum(gy);
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        lin = fmaf(gux[r], cvx[r] - uvx[r], lin);
        lin = fmaf(guz[r], cwz[r] - uwz[r], lin);
        lin = fmaf(guy[r], cvy[r] - uvy[r], lin);
      }
      if (EXTRA) cost += uni + (float)wave_sum_d(lin_d + (double)lin);
      else cost += uni + wave_sum(lin);
    }

    // ---- park the rollout -----------------------------------------------------
    {
      float* cg = cring + (size_t)n_pend * 3 * T;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (STEP_OK(r)) {
          cg[t0 + r] = cvx[r];
          cg[T + t0 + r] = cvy[r];
          cg[2 * T + t0 + r] = cwz[r];
        }
        if (slot[r] >= 0) {
          const uint32_t q = (n_pend << seg_shift) + (uint32_t)slot[r];
          pts_x[q] = x[r];
          pts_y[q] = y[r];
          if (RARE && (p.flags & (SD_USE_PATH_YAW | SD_PAL_USE_PATH_YAW))) pts_yaw[q] = yaw[r];
        }
      }
      if (((uint32_t)lane >> seg_shift) == n_pend) pend_cost = cost;
      if (++n_pend == GROUP) {
        flush(n_pend, b);
        n_pend = 0;
      }
    }
  }
  if (!FURTHEST_ONLY && n_pend) flush(n_pend, b_prev);
  if (want_local_furthest && n_ring)
    S_local = max(S_local, flush_endpoint_ring(ring_x, ring_y, n_ring, s_px, s_py, p.P, lane));

  // ---- block combine -> one partial per block --------------------------------
  __syncthreads();
  if (FURTHEST_ONLY) {
    // S_local holds the bits of F >= 0: order preserving
    uint32_t* s_red = reinterpret_cast<uint32_t*>(smem + L.off_scr);
    if (lane == 0) s_red[wave] = S_local;
    __syncthreads();
    if (tid == 0) {
      uint32_t m = 0;
      for (int w = 0; w < nwave; ++w) m = max(m, s_red[w]);
      atomicMax(p.furthest_out, m);
    }
    return;
  }
  const uint32_t TL = 4 + 3 * T;  // scr_stride >= TL is guaranteed by the host
  float* myp = reinterpret_cast<float*>(smem + L.off_scr) + (size_t)wave * L.scr_stride;
  if (lane == 0) {
    myp[0] = m_run;
    myp[1] = s_run;
    myp[2] = __uint_as_float(S_local);   // F = index + fraction (smpc_dev.h)
    myp[3] = (float)n_noncoll;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (STEP_OK(r)) {
      myp[4 + t0 + r] = Ux[r];
      myp[4 + T + t0 + r] = Uy[r];
      myp[4 + 2 * T + t0 + r] = Uz[r];
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
        const float sc = expf(p.neg_inv_temp * (mw - bm));
        acc += sc * allp[(size_t)w * L.scr_stride + i];
      }
    }
    smpc_store_partial(outp + i, acc);
  }
#undef STEP_OK
  if (p.tail) smpc_grid_tail<(R == 4 ? 8 : 16)>(p, smem);
}

// ---------------------------------------------------------------------------
// Reduce the per-block partials of one pass into one shard tuple
// {min, sum w, furthest, non-colliding, U[3T]}.  Block j owns 32 tuple columns;
// its 1024 threads are 32 columns x 32 row slices (coalesced 128-B row pieces).
// ---------------------------------------------------------------------------
// With fin.enabled (single-GPU tick, one tuple) every block also finishes its columns:
// divide by sum w, clip, write the new control sequence to device AND host-mapped memory
// (optimizer.cpp:382-393,237-249) — the separate combine launch and the D2H copy go away.
__device__ __forceinline__ void reduce_partials_body(const float* __restrict__ partials,
                                                     uint32_t nblk, uint32_t T, float neg_inv_temp,
                                                     float* __restrict__ tuple, const SmpcFinal& fin)
{
  __shared__ float s_red[16], s_red2[16], s_red3[16];
  __shared__ float s_sc[2048];
  __shared__ float s_acc[32][33];
  const uint32_t TL = 4 + 3 * T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
  const uint32_t col = blockIdx.x * 32 + (tid & 31);
  const uint32_t sl = tid >> 5;  // 32 slices
  const bool colon = col < TL && col != 0 && col != 2 && col != 3;
  // Every global load this thread needs is issued before the first wait: the tuple headers
  // {min, sum w, furthest, non-colliding} of (up to two) partials, and the first 16 rows of
  // its column — one memory round trip instead of five dependent ones.
  float hm[2], hw[2], hf[2], hn[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const uint32_t g = tid + k * 1024;
    const bool on = g < nblk;
    const float* h = partials + (size_t)(on ? g : 0) * TL;
    hm[k] = on ? h[0] : 3.0e38f;
    hw[k] = on ? h[1] : 0.f;
    hf[k] = on ? h[2] : 0.f;
    hn[k] = on ? h[3] : 0.f;
  }
  float v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const uint32_t g = sl + 32 * k;
    v[k] = (colon && g < nblk) ? partials[(size_t)g * TL + col] : 0.f;
  }
  float m = fminf(hm[0], hm[1]), fu = fmaxf(hf[0], hf[1]), nc = hn[0] + hn[1];
  for (int o = 32; o > 0; o >>= 1) {
    m = fminf(m, __shfl_xor(m, o, WAVE));
    fu = fmaxf(fu, __shfl_xor(fu, o, WAVE));
    nc += __shfl_xor(nc, o, WAVE);
  }
  if (lane == 0) {
    s_red[wave] = m;
    s_red2[wave] = fu;
    s_red3[wave] = nc;
  }
  __syncthreads();
  m = 3.0e38f;
  fu = 0.f;
  nc = 0.f;
  for (int w = 0; w < nwave; ++w) {
    m = fminf(m, s_red[w]);
    fu = fmaxf(fu, s_red2[w]);
    nc += s_red3[w];
  }
  // per-block rescale factors exp(-(m_g - m)/temperature), once; sum of weights from the headers
  float a = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const uint32_t g = tid + k * 1024;
    if (g < nblk) {
      const float sc = expf(neg_inv_temp * (hm[k] - m));
      s_sc[g] = sc;
      a += sc * hw[k];
    }
  }
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, WAVE);
  __syncthreads();   // s_sc complete, s_red* free again
  if (lane == 0) s_red[wave] = a;
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const uint32_t g = sl + 32 * k;
    if (g < nblk) acc += s_sc[g] * v[k];
  }
  if (colon) {   // grids beyond 512 blocks: the rest of the column
    for (uint32_t g = sl + 512; g < nblk; g += 32) acc += s_sc[g] * partials[(size_t)g * TL + col];
  }
  s_acc[sl][tid & 31] = acc;
  __syncthreads();
  float sw = 0.f;
  for (int w = 0; w < nwave; ++w) sw += s_red[w];
  if (sl == 0 && col < TL) {
    float r = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) r += s_acc[s][tid & 31];
    if (col == 0) r = m;
    if (col == 1) r = sw;
    if (col == 2) r = fu;
    if (col == 3) r = nc;
    tuple[col] = r;
    if (fin.enabled) {
      if (col >= 4) {
        const uint32_t i = col - 4;
        float v2 = r / sw;
        // applyControlSequenceConstraints (optimizer.cpp:237-249)
        if (i < T) v2 = fminf(fmaxf(v2, fin.vx_min), fin.vx_max);
        else if (i < 2 * T) v2 = fminf(fmaxf(v2, -fin.vy_max), fin.vy_max);
        else v2 = fminf(fmaxf(v2, -fin.wz_max), fin.wz_max);
        fin.u_dev[i] = v2;
        if (fin.u_host) fin.u_host[i] = v2;
      } else if (col == 0) {
        const float used = fin.furthest_used ? *fin.furthest_used : fu;
        const float res[5] = {m, sw, fu, nc, used};
        for (int k = 0; k < 5; ++k) {
          fin.u_dev[3 * T + k] = res[k];
          if (fin.u_host) fin.u_host[3 * T + k] = res[k];
        }
      }
    }
    // the tick block's number as the pass read it (SMPC_CANARY_SLOT): to the device copy of the
    // result (the combine kernels forward it from there) and, when finishing, to the host
    if (col == 0 && fin.u_dev) {
      const float cn = partials[SMPC_CANARY_SLOT(T)];
      fin.u_dev[3 * T + 5] = cn;
      if (fin.enabled && fin.u_host) fin.u_host[3 * T + 5] = cn;
    }
  }
  // completion for the polling host: every block publishes the tick's sequence number in its
  // own word (u_host[3T + 8 + block]) behind a system-scope fence over its host stores, and the
  // host waits for all of them — no counter and no second fence between the last store and the
  // host (the tail timeline of tools/tail_timeline.py prices them at ~2 us)
  if (fin.enabled && fin.done_counter) {
    __threadfence_system();
    __syncthreads();
    if (tid == 0)
      __hip_atomic_store(reinterpret_cast<uint32_t*>(fin.u_host + 3 * T + 8 + blockIdx.x), fin.seq,
                         __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}


__global__ void __launch_bounds__(1024) smpc_reduce_partials(const float* __restrict__ partials,
                                                            uint32_t nblk, uint32_t T,
                                                            float neg_inv_temp,
                                                            float* __restrict__ tuple,
                                                            const SmpcFinal fin)
{
  reduce_partials_body(partials, nblk, T, neg_inv_temp, tuple, fin);
}

// several planning instances in one launch (smpc_group_optimize): blockIdx.y picks the instance
__global__ void __launch_bounds__(1024) smpc_reduce_partials_many(const SmpcReduceArgs* __restrict__ many,
                                                                 uint32_t T)
{
  const SmpcReduceArgs& a = many[blockIdx.y];
  reduce_partials_body(a.partials, a.nblk, T, a.neg_inv_temp, a.tuple, a.fin);
}

// ---------------------------------------------------------------------------
// Combine G shard tuples into the new control sequence (optimizer.cpp:382-393):
// rescale by exp(-(min_g - min)/temperature), divide by sum w, clip.
// result: {min, sum_w, furthest, non_colliding}.  One block.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void combine_tuples_body(const float* __restrict__ tuples, uint32_t stride,
                                                    uint32_t G, uint32_t T, float neg_inv_temp,
                                                    float vx_max, float vx_min, float vy_max,
                                                    float wz_max, float* __restrict__ u_out,
                                                    float* __restrict__ result,
                                                    const float* __restrict__ furthest_used,
                                                    float* __restrict__ host_out, uint32_t seq)
{
  const uint32_t TL = stride;
  float m = 3.0e38f, fu = 0.f, nc = 0.f;
  for (uint32_t g = 0; g < G; ++g) {
    m = fminf(m, tuples[(size_t)g * TL]);
    fu = fmaxf(fu, tuples[(size_t)g * TL + 2]);
    nc += tuples[(size_t)g * TL + 3];
  }
  float sw = 0.f;
  for (uint32_t g = 0; g < G; ++g)
    sw += expf(neg_inv_temp * (tuples[(size_t)g * TL] - m)) * tuples[(size_t)g * TL + 1];
  for (uint32_t i = threadIdx.x; i < 3 * T; i += blockDim.x) {
    float acc = 0.f;
    for (uint32_t g = 0; g < G; ++g)
      acc += expf(neg_inv_temp * (tuples[(size_t)g * TL] - m)) * tuples[(size_t)g * TL + 4 + i];
    float v = acc / sw;
    // applyControlSequenceConstraints (optimizer.cpp:237-249)
    if (i < T) v = fminf(fmaxf(v, vx_min), vx_max);
    else if (i < 2 * T) v = fminf(fmaxf(v, -vy_max), vy_max);
    else v = fminf(fmaxf(v, -wz_max), wz_max);
    u_out[i] = v;
    if (host_out) host_out[i] = v;
  }
  if (threadIdx.x == 0) {
    // [4]: the furthest point the critics consumed (cached across iterations)
    const float res[5] = {m, sw, fu, nc, furthest_used ? *furthest_used : fu};
    for (int k = 0; k < 5; ++k) {
      result[k] = res[k];
      if (host_out) host_out[3 * T + k] = res[k];
    }
    if (host_out) host_out[3 * T + 5] = result[5];   // the pass's echo of the tick block's number (smpc_reduce_partials)
  }
  // completion word for the polling host, behind every host store of this (single) block
  if (host_out && seq) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
      __hip_atomic_store(reinterpret_cast<uint32_t*>(host_out + 3 * T + 7), seq, __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ void __launch_bounds__(256) smpc_combine_tuples(const float* __restrict__ tuples,
                                                          uint32_t G, uint32_t T,
                                                          float neg_inv_temp, float vx_max,
                                                          float vx_min, float vy_max,
                                                          float wz_max, float* __restrict__ u_out,
                                                          float* __restrict__ result,
                                                          const float* __restrict__ furthest_used,
                                                          float* __restrict__ host_out, uint32_t seq)
{
  combine_tuples_body(tuples, 4 + 3 * T, G, T, neg_inv_temp, vx_max, vx_min, vy_max, wz_max, u_out, result,
                      furthest_used, host_out, seq);
}

// ---------------------------------------------------------------------------
// Exchange of the shard tuples WITHOUT a collective (include/smpc.h smpc_shard_p2p_*): every
// rank owns a mailbox in fine-grained device memory that its peers have mapped (IPC, xGMI):
// [2 parities][world slots][slot_floats], the last word of a slot being its sequence number.
// One block: copy this rank's tuple into slot `rank` of every mailbox, fence at system scope,
// publish the sequence number; wait until every slot of the own mailbox carries it; then
// combine as smpc_combine_tuples does (mode 0) or just take the maximum of the furthest points
// (mode 1: the exchange of the non-speculative first tick).  Two parities: a rank can run at
// most one exchange ahead of its slowest peer (it needs that peer's tuple to finish its own).
//
// The wait is bounded in WALL-CLOCK time (s_memrealtime, a constant 100 MHz counter:
// x.timeout_ticks of 10 ns each).  On expiry the block sets *x.state (sticky, device memory) and
// host_out[3T + 6] = 1, publishes the completion word so that the host returns at once, and
// publishes nothing else.  Every later exchange of this ctx sees *x.state != 0 on entry and
// does the same WITHOUT writing to any peer: a rank that failed once stays silent, so its peers
// fail at their next exchange at the latest instead of consuming tuples of a rank whose own
// tick was lost.  Only smpc_shard_p2p_init (collective) clears the state.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) smpc_p2p_exchange(const float* __restrict__ my_tuple,
                                                        const SmpcP2P x, uint32_t T, int mode,
                                                        float* __restrict__ d_furthest,
                                                        float neg_inv_temp, float vx_max, float vx_min,
                                                        float vy_max, float wz_max,
                                                        float* __restrict__ u_out,
                                                        float* __restrict__ result,
                                                        const float* __restrict__ furthest_used,
                                                        float* __restrict__ host_out, uint32_t seq)
{
  __shared__ int s_late;
  const uint32_t TL = 4 + 3 * T, tid = threadIdx.x;
  const size_t slot0 = (size_t)((x.xseq & 1u) * x.world) * x.slot_floats;
  if (tid == 0) s_late = (__hip_atomic_load(x.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 2 : 0;
  __syncthreads();
  if (s_late == 0) {
    for (uint32_t r = 0; r < x.world; ++r) {
      float* dst = x.peer[r] + slot0 + (size_t)x.rank * x.slot_floats;
      for (uint32_t i = tid; i < TL; i += blockDim.x) dst[i] = my_tuple[i];
    }
    __threadfence_system();
    __syncthreads();
    if (tid < x.world)
      __hip_atomic_store(reinterpret_cast<uint32_t*>(x.peer[tid] + slot0 + (size_t)x.rank * x.slot_floats + TL),
                         x.xseq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const float* mine = x.peer[x.rank] + slot0;
  if (s_late == 0 && tid < x.world) {
    const uint32_t* flag = reinterpret_cast<const uint32_t*>(mine + (size_t)tid * x.slot_floats + TL);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != x.xseq) {
      __builtin_amdgcn_s_sleep(8);
      if (__builtin_amdgcn_s_memrealtime() - t0 > x.timeout_ticks) {
        s_late = 1;
        break;
      }
    }
  }
  __syncthreads();
  if (s_late) {
    if (tid == 0) {
      __hip_atomic_store(x.state, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (host_out) {
        host_out[3 * T + 6] = 1.0f;
        __threadfence_system();
        if (seq)
          __hip_atomic_store(reinterpret_cast<uint32_t*>(host_out + 3 * T + 7), seq, __ATOMIC_RELEASE,
                             __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    return;
  }
  if (mode == 1) {
    if (tid == 0) {
      float fu = 0.f;
      for (uint32_t r = 0; r < x.world; ++r) fu = fmaxf(fu, mine[(size_t)r * x.slot_floats + 2]);
      *d_furthest = fu;
    }
    return;
  }
  combine_tuples_body(mine, x.slot_floats, x.world, T, neg_inv_temp, vx_max, vx_min, vy_max, wz_max, u_out,
                      result, furthest_used, host_out, seq);
}

// ---------------------------------------------------------------------------
// AckermannMotionModel::applyConstraints (motion_models.hpp:110-117), the last step of
// applyControlSequenceConstraints (optimizer.cpp:248): where |vx|/|wz| < min_turning_r,
// wz = sign(wz) |vx| / min_turning_r.  It needs the finished vx AND wz of a step, which the
// column-parallel finishing kernels hold in different blocks, so it is its own T-thread
// launch behind them and publishes the tick's completion word in their place.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) smpc_ackermann_constrain(float* __restrict__ u_dev,
                                                               float* __restrict__ u_host,
                                                               uint32_t T, float min_r, uint32_t seq)
{
  for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) {
    const float v = u_dev[t], w = u_dev[2 * T + t];
    if (fabsf(v) / fabsf(w) < min_r) {
      const float sgn = w > 0.f ? 1.f : (w < 0.f ? -1.f : 0.f);
      const float nw = sgn * fabsf(v) / min_r;
      u_dev[2 * T + t] = nw;
      if (u_host) u_host[2 * T + t] = nw;
    }
  }
  if (u_host && seq) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
      __hip_atomic_store(reinterpret_cast<uint32_t*>(u_host + 3 * T + 7), seq, __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---------------------------------------------------------------------------
// Device RNG: Philox4x32-10 + Box–Muller, one block of two samples per thread
// (stands in for xt::random::randn, noise_generator.cpp:107-122).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&o)[4])
{
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    if (r > 0) {
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    // (one 32 x 32 -> 64 multiply each, v_mad_u64_u32: half the quarter-rate instructions of a
    // v_mul_hi_u32 / v_mul_lo_u32 pair)
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0;
    c1 = lo1;
    c2 = n2;
    c3 = lo0;
  }
  o[0] = c0;
  o[1] = c1;
  o[2] = c2;
  o[3] = c3;
}

// Box–Muller on one pair of Philox words: {radius cos, radius sin} of N(0, 1).  The hardware's
// log2 and square root (1 ulp) and the rollout's own sin/cos (absolute error 1.3e-7): the normals
// stay within 1e-6 of the CPU twin's libm evaluation (tests: test_device_rng_matches_cpu_twin),
// at a third of the instructions of logf / sincosf — with regenerate_noises the draw is 3 x B x T
// normals per tick.
__device__ __forceinline__ void box_muller(uint32_t r0, uint32_t r1, float& z0, float& z1)
{
  const float u1 = ((float)(r0 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(r1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  // -2 ln u1 = -2 ln 2 log2 u1
  const float radius = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
  float sn, cs;
  smpc_sincos_fast(6.2831853071795864769f * u2, sn, cs);
  z0 = radius * cs;
  z1 = radius * sn;
}

// Flat elements 4q .. 4q + 3 of a tensor share ONE Philox block (all four words are used: two
// Box–Muller pairs): thread q of the grid covering [base, base + n).
__global__ void __launch_bounds__(256) smpc_fill_noise(float* __restrict__ out, uint64_t n,
                                                      uint64_t base, uint64_t seed,
                                                      uint32_t stream, uint32_t epoch,
                                                      float sigma)
{
  const uint64_t q_first = base >> 2;
  const uint64_t q_last = (base + n + 3) >> 2;  // exclusive
  for (uint64_t q = q_first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < q_last;
       q += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t r[4];
    philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), stream, epoch, (uint32_t)seed,
                  (uint32_t)(seed >> 32), r);
    float z[4];
    box_muller(r[0], r[1], z[0], z[1]);
    box_muller(r[2], r[3], z[2], z[3]);
    const uint64_t e0 = q << 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint64_t e = e0 + (uint64_t)k;
      if (e >= base && e < base + n) out[e - base] = z[k] * sigma;
    }
  }
}

// The same stream written GROUP-MAJOR (SMPC_GM_INDEX, smpc_dev.h: the layout the lane-per-rollout
// and split passes read): thread (b, group of four steps) with b fastest, so a wave writes four
// coalesced 256-byte pieces.  Element e = base + b * T + t of the global [B_global, T] tensor is word
// e % 4 of Philox block e / 4 exactly as in smpc_fill_noise; T must be a multiple of four, so
// that a block never straddles two rollouts (base is a multiple of T).
__global__ void __launch_bounds__(256) smpc_fill_noise_tm(float* __restrict__ dst, uint32_t B, uint32_t T,
                                                         uint64_t base, uint64_t seed, uint32_t stream,
                                                         uint32_t epoch, float sigma)
{
  // blockIdx.y: the group of four steps; x: rollouts (no 64-bit division of a flat index)
  const uint32_t tq = blockIdx.y;
  for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
    const uint64_t q = (base + (uint64_t)b * T + 4u * tq) >> 2;
    uint32_t r[4];
    philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), stream, epoch, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    float z[4];
    box_muller(r[0], r[1], z[0], z[1]);
    box_muller(r[2], r[3], z[2], z[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[SMPC_GM_INDEX(b, 4u * tq + (uint32_t)k, T)] = z[k] * sigma;
  }
}

// ---------------------------------------------------------------------------
// Self-test hook: the device sin/cos used by the rollout, evaluated on caller data.
// ---------------------------------------------------------------------------
__global__ void smpc_sincos_kernel(const float* __restrict__ x, uint32_t n,
                                   float* __restrict__ sn, float* __restrict__ cs)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float s, c;
    smpc_sincos(x[i], s, c);
    sn[i] = s;
    cs[i] = c;
  }
}

hipError_t smpc_launch_sincos(const float* x, uint32_t n, float* sn, float* cs, hipStream_t st)
{
  hipLaunchKernelGGL(smpc_sincos_kernel, dim3((n + 255) / 256), dim3(256), 0, st, x, n, sn, cs);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// launch wrappers (called from smpc_api.cpp through plain C++ linkage)
// ---------------------------------------------------------------------------
char smpc_last_pass_kernel[96] = "";   // (a developer aid: not per thread)

template <int MODE, bool FULL>
static hipError_t launch_pass_r(int R, const SmpcDev& p, const SmpcLds& L, uint32_t grid,
                                uint32_t block, hipStream_t st)
{
  snprintf(smpc_last_pass_kernel, sizeof(smpc_last_pass_kernel), "smpc_pass<%d, %d, %s>", R, MODE, FULL ? "true" : "false");
  switch (R) {
    case 1: hipLaunchKernelGGL((smpc_pass<1, MODE, FULL>), dim3(grid), dim3(block), L.total, st, p, L); break;
    case 2: hipLaunchKernelGGL((smpc_pass<2, MODE, FULL>), dim3(grid), dim3(block), L.total, st, p, L); break;
    case 4: hipLaunchKernelGGL((smpc_pass<4, MODE, FULL>), dim3(grid), dim3(block), L.total, st, p, L); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// mode: 0 score (all cost_power == 1), 1 furthest only, 2 score (general cost_power),
// 3 score (all cost_power == 1, with the additive forms of Cost, Goal, Constraint, Twirling, PathAngle)
hipError_t smpc_launch_pass(int R, int mode, const SmpcDev& p, const SmpcLds& L,
                            uint32_t grid, uint32_t block, hipStream_t st)
{
  const bool full = p.T == 64u * (uint32_t)R;
  switch (mode) {
    case 0: return full ? launch_pass_r<0, true>(R, p, L, grid, block, st)
                        : launch_pass_r<0, false>(R, p, L, grid, block, st);
    case 1: return full ? launch_pass_r<1, true>(R, p, L, grid, block, st)
                        : launch_pass_r<1, false>(R, p, L, grid, block, st);
    case 3: return full ? launch_pass_r<3, true>(R, p, L, grid, block, st)
                        : launch_pass_r<3, false>(R, p, L, grid, block, st);
    default: return full ? launch_pass_r<2, true>(R, p, L, grid, block, st)
                         : launch_pass_r<2, false>(R, p, L, grid, block, st);
  }
}

template <typename F>
static void for_each_pass_kernel(F&& f)
{
#define EACH(RR, MM) f(reinterpret_cast<const void*>(&smpc_pass<RR, MM, true>), RR, MM, true); \
                     f(reinterpret_cast<const void*>(&smpc_pass<RR, MM, false>), RR, MM, false);
  EACH(1, 0) EACH(2, 0) EACH(4, 0) EACH(1, 1) EACH(2, 1) EACH(4, 1) EACH(1, 2) EACH(2, 2) EACH(4, 2)
  EACH(1, 3) EACH(2, 3) EACH(4, 3)
#undef EACH
}

hipError_t smpc_pass_occupancy(int R, int mode, bool full, uint32_t block, uint32_t lds_bytes,
                               int* blocks_per_cu)
{
  const void* fn = nullptr;
  for_each_pass_kernel([&](const void* f, int r, int m, bool fl) {
    if (r == R && m == mode && fl == full) fn = f;
  });
  if (!fn) return hipErrorInvalidValue;
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, fn, (int)block, lds_bytes);
}

hipError_t smpc_set_pass_lds_limit(int bytes)
{
  hipError_t e = hipSuccess;
  for_each_pass_kernel([&](const void* f, int, int, bool) {
    if (e == hipSuccess)
      e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  });
  return e;
}

hipError_t smpc_launch_reduce(const float* partials, uint32_t nblk, uint32_t T,
                              float neg_inv_temp, float* tuple, const SmpcFinal& fin,
                              hipStream_t st)
{
  const uint32_t TL = 4 + 3 * T;
  hipLaunchKernelGGL(smpc_reduce_partials, dim3((TL + 31) / 32), dim3(1024), 0, st, partials, nblk,
                     T, neg_inv_temp, tuple, fin);
  return hipGetLastError();
}

// After smpc_reduce_partials_many: ONE block copies every instance's result (3T + 8 floats) to its
// host-mapped mirror and publishes the sequence words behind a single system-scope fence
// (hundreds of blocks fencing PCIe writes one by one cost far more than the reduction itself).
__global__ void __launch_bounds__(1024) smpc_publish_many(const SmpcReduceArgs* __restrict__ many,
                                                         uint32_t n, uint32_t T)
{
  const uint32_t len = 3 * T + 6;   // u, the five results, the tick block's number
  for (uint32_t i = threadIdx.x; i < n * len; i += blockDim.x) {
    const uint32_t q = i / len, k = i - q * len;
    many[q].host_out[k] = many[q].fin.u_dev[k];
  }
  __threadfence_system();
  __syncthreads();
  for (uint32_t q = threadIdx.x; q < n; q += blockDim.x)
    __hip_atomic_store(reinterpret_cast<uint32_t*>(many[q].host_out + 3 * T + 7), many[q].seq,
                       __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t smpc_launch_reduce_many(const SmpcReduceArgs* d_many, uint32_t n, uint32_t T, hipStream_t st)
{
  const uint32_t TL = 4 + 3 * T;
  hipLaunchKernelGGL(smpc_reduce_partials_many, dim3((TL + 31) / 32, n), dim3(1024), 0, st, d_many, T);
  hipLaunchKernelGGL(smpc_publish_many, dim3(1), dim3(1024), 0, st, d_many, n, T);
  return hipGetLastError();
}

hipError_t smpc_launch_combine(const float* tuples, uint32_t G, uint32_t T, float neg_inv_temp,
                               float vx_max, float vx_min, float vy_max, float wz_max,
                               float* u_out, float* result, const float* furthest_used,
                               float* host_out, uint32_t seq, hipStream_t st)
{
  hipLaunchKernelGGL(smpc_combine_tuples, dim3(1), dim3(256), 0, st, tuples, G, T, neg_inv_temp,
                     vx_max, vx_min, vy_max, wz_max, u_out, result, furthest_used, host_out, seq);
  return hipGetLastError();
}

hipError_t smpc_launch_p2p_exchange(const float* my_tuple, const SmpcP2P& x, uint32_t T, int mode,
                                    float* d_furthest, float neg_inv_temp, float vx_max, float vx_min,
                                    float vy_max, float wz_max, float* u_out, float* result,
                                    const float* furthest_used, float* host_out, uint32_t seq,
                                    hipStream_t st)
{
  hipLaunchKernelGGL(smpc_p2p_exchange, dim3(1), dim3(256), 0, st, my_tuple, x, T, mode, d_furthest,
                     neg_inv_temp, vx_max, vx_min, vy_max, wz_max, u_out, result, furthest_used, host_out,
                     seq);
  return hipGetLastError();
}

hipError_t smpc_launch_ackermann(float* u_dev, float* u_host, uint32_t T, float min_r, uint32_t seq,
                                 hipStream_t st)
{
  hipLaunchKernelGGL(smpc_ackermann_constrain, dim3(1), dim3(256), 0, st, u_dev, u_host, T, min_r,
                     seq);
  return hipGetLastError();
}

hipError_t smpc_launch_fill_noise_tm(float* dst, uint32_t B, uint32_t T, uint64_t base, uint64_t seed,
                                     uint32_t stream, uint32_t epoch, float sigma, hipStream_t st)
{
  if (T & 3u) return hipErrorInvalidValue;   // a Philox block of four must not straddle two rollouts
  uint32_t grid = std::min<uint32_t>((B + 255u) / 256u, 2048u);
  if (grid == 0) grid = 1;
  hipLaunchKernelGGL(smpc_fill_noise_tm, dim3(grid, T >> 2), dim3(256), 0, st, dst, B, T, base, seed, stream, epoch, sigma);
  return hipGetLastError();
}

hipError_t smpc_launch_fill_noise(float* out, uint64_t n, uint64_t base, uint64_t seed,
                                  uint32_t stream, uint32_t epoch, float sigma, hipStream_t st)
{
  const uint64_t blocks4 = (n + 6) / 4;   // (covers a base that is not a multiple of four)
  uint32_t grid = (uint32_t)((blocks4 + 255) / 256);
  if (grid > 4096) grid = 4096;
  if (grid == 0) grid = 1;
  hipLaunchKernelGGL(smpc_fill_noise, dim3(grid), dim3(256), 0, st, out, n, base, seed, stream,
                     epoch, sigma);
  return hipGetLastError();
}
