// smpc_split.hip — the streaming pass for batches too small to fill the chip with one lane per
// rollout: lane = (rollout, QUARTER OF THE HORIZON).
//
// smpc_lane.hip walks the horizon step by step with lane = rollout: 64 dependent steps of ~105
// instructions per wave.  At 65 536 rollouts (BASELINE configs[1]) that is ONE wave per SIMD,
// and a wave alone issues one instruction per ~4.5 cycles whatever the other 63 % of the SIMD's
// cycles could do: 20.5 us per group against a VALU budget of 7 (VERDICT r02
// item 2).  Here a wave owns 16 rollouts and every rollout is spread over four lanes, one per
// segment of 16 steps, lane = 16 segment + rollout — four times the waves, a quarter of the
// chain each:
//   A  yaw: every lane sums its segment's wz dt (optimizer.cpp:313-343: yaw is a plain cumulative
//      sum, so a segment's total is all its successors need), exclusive prefix over the four
//      segments (three lane gathers), in the order of the sequential sum;
//   B  displacement: with the segment's starting yaw known, sin/cos and the rotated velocities of
//      its 16 steps, accumulated from zero; exclusive prefix of the segment displacements;
//   C  critics at the absolute position: costmap lookups (first collision masks the rest of the
//      SEGMENT; the segments before it mask it afterwards), PreferForward, gamma terms,
//      PathAlign's sample points (kept: the closest-path-point rule needs the arc length from the
//      rollout's start, another prefix over the segments, and its one sequential dependency —
//      "the lower bound equals the previous point" — is resolved segment after segment);
//   D  per rollout: sums over its four lanes, nearest path point of the endpoint (each lane scans
//      a quarter of the path), cost, softmax weight;
//   E  U[t] += sum_b w_b c[b][t]: lane 16 s + i holds step 16 s + i of ITS rollout's parked
//      controls in register i, so a 16 x 16 transpose-reduce inside each DPP row leaves
//      sum_b w_b c[b][t] in lane t — the layout of smpc_lane.hip's update and block partial.
// It spends VALU (the displacement prefix costs a second position accumulation, the per-wave
// epilogue is amortised over 16 steps instead of 64: ~125 VALU per 64 rollout-steps against 104)
// to buy latency exactly where the VALU idles.  Cumulative sums are associated segment-wise
// (last-ulp differences to the sequential order, like smpc_pass's scans: the parity tests'
// counted cell flips cover both).
//
// Scope: T == 64, ObstaclesCritic scored, the north star's five critics with cost_power 1
// (the lean mode without GoalAngle), trajectory_point_step 4, batches whose waves are all
// resident at once (the host decides: smpc_prepare.cpp plan_launch).  Everything else keeps
// smpc_pass_lane / smpc_pass.  Compiled with -ffp-contract=off like the other passes.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "smpc_dev.h"
#include "smpc_device_math.h"
#include "smpc_lane_common.h"

#define SPLIT_BLOCK 512   // 8 waves, one block per CU: two waves per SIMD (the parked controls, the segment's displacements and
                          // PathAlign's samples take ~215 registers; at 128 for four waves per SIMD 89 of them spill)
// NSEG (template): lanes per rollout = segments of the horizon, 4 or 2; a lane walks 64 / NSEG
// steps and a wave holds 64 / NSEG rollouts.  What a lone wave costs is its INSTRUCTION COUNT (one
// issue per ~4.5 cycles whatever the SIMD's idle share): four segments quarter the chain but
// carry ~80 % more instructions per rollout-step than smpc_pass_lane (two position passes, the
// control sequence from LDS instead of scalar registers, a per-wave epilogue amortised over 16
// steps), two segments halve it at ~35 % more.  The host picks NSEG so that every wave has ONE
// group: 4 up to 32 768 rollouts on 256 CUs, 2 up to 65 536.

typedef const float __attribute__((address_space(4))) * cfloat_ps;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
template <int N> struct vecN;
template <> struct vecN<32> {typedef f32x32 type;};
template <> struct vecN<16> {typedef f32x16 type;};
template <> struct vecN<8> {typedef f32x8 type;};
template <> struct vecN<4> {typedef f32x4 type;};

// 16 x 16 transpose-reduce inside every 16-lane DPP row: in: V[i] of lane r; out: lane i of the
// row = sum_r V_r[i].  The four within-row levels of smpc_lane_common.h's 64-lane butterfly.
template <int N, int K, int Rr>   // N registers: 16 (levels 3..6) or 32 (levels 2..6)
__device__ __forceinline__ float row_reduce_node(const float (&V)[N], int lane)
{
  if constexpr ((64 >> K) == N) {
    return V[Rr];
  } else {
    const float a = row_reduce_node<N, K - 1, Rr>(V, lane);
    const float b = row_reduce_node<N, K - 1, Rr + (64 >> K)>(V, lane);
    if constexpr (K == 2) {
      // odd 16-lane rows of a <-> even rows of b: the sum over lanes l, l ^ 16 of the kept half
      const u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
      return __uint_as_float(sw.x) + __uint_as_float(sw.y);
    } else if constexpr (K == 3) {
      return node_bit3(a, b);
    } else if constexpr (K == 4) {
      return node_bit2(a, b);
    } else if constexpr (K == 5) {
      const float ta = a + dpp_mov<0x4E>(a);   // quad_perm [2,3,0,1]: partner l ^ 2
      const float tb = b + dpp_mov<0x4E>(b);
      return (lane & 2) ? tb : ta;
    } else {
      const float ta = a + dpp_mov<0xB1>(a);   // quad_perm [1,0,3,2]: partner l ^ 1
      const float tb = b + dpp_mov<0xB1>(b);
      return (lane & 1) ? tb : ta;
    }
  }
}
// in: V[i] of every lane of a group of N lanes (a 16-lane row, or a 32-lane half); out: lane i of the group = sum over the group of V[i]
template <int N>
__device__ __forceinline__ float row_reduce(const float (&V)[N], int lane) {return row_reduce_node<N, 6, 0>(V, lane);}

// self-test of the row transpose-reduce (tests/test_gpu_parity.py): v [64 lanes][N]
template <int N>
__global__ void __launch_bounds__(64) smpc_row_reduce_kernel(const float* __restrict__ v, float* __restrict__ out)
{
  const int lane = threadIdx.x;
  float V[N];
#pragma unroll
  for (int i = 0; i < N; ++i) V[i] = v[lane * N + i];
  out[lane] = row_reduce<N>(V, lane);
}

hipError_t smpc_launch_row_reduce(const float* v, float* out, uint32_t n, hipStream_t st)
{
  if (n == 16) hipLaunchKernelGGL(smpc_row_reduce_kernel<16>, dim3(1), dim3(64), 0, st, v, out);
  else if (n == 32) hipLaunchKernelGGL(smpc_row_reduce_kernel<32>, dim3(1), dim3(64), 0, st, v, out);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// value of lane `src` (a ds_bpermute: any lane to any lane)
__device__ __forceinline__ float lane_get(float v, int src) {return __shfl(v, src, WAVE);}

// exclusive prefix over the segments of a rollout, associated like the sequential sum:
// 0, v0, v0 + v1, (v0 + v1) + v2
template <int NSEG>
__device__ __forceinline__ float seg_exclusive(float v, int r, int sg)
{
  constexpr int R = 64 / NSEG;
  const float v0 = lane_get(v, r);
  if constexpr (NSEG == 2) {
    return sg == 0 ? 0.f : v0;
  } else {
    const float v1 = lane_get(v, r + R), v2 = lane_get(v, r + 2 * R);
    const float s01 = v0 + v1;
    return sg == 0 ? 0.f : (sg == 1 ? v0 : (sg == 2 ? s01 : s01 + v2));
  }
}
// sum over the segments, bit-identical in the lanes of a rollout
template <int NSEG>
__device__ __forceinline__ float seg_total(float v)
{
  if constexpr (NSEG == 4) v = v + __shfl_xor(v, 16, WAVE);
  return v + __shfl_xor(v, 32, WAVE);
}

// FULL: T == 64.  Otherwise T is a multiple of four below 64 (the reference's default horizon is
// 56): the segments keep their 64 / NSEG step slots, the slots from T on are idle — their noised
// controls are parked as zeros, their critic terms, sample points and lookups masked (the
// horizon's end falls into one segment; the segments behind it are idle altogether).
template <int NSEG, bool FULL>
__global__ void __launch_bounds__(SPLIT_BLOCK, 2) smpc_pass_split(const SmpcDev p, const SmpcLds L)
{
  static_assert(NSEG == 2 || NSEG == 4, "two or four segments");
  const uint32_t T = FULL ? 64u : p.T;
  constexpr int SPLIT_SEG = 64 / NSEG;    // steps per lane
  constexpr int SPLIT_ROLL = 64 / NSEG;   // rollouts per wave
  constexpr int NSAMP = SPLIT_SEG / 4;    // PathAlign samples per lane
  // two segments: 32 steps per lane — the noised vy is parked in LDS next to wz.  (With its 32
  // critic steps unrolled this instance spilled 340 bytes per lane at 256 registers, and not
  // because of what is live: without the exact cell path's BRANCH in each step it needs 242
  // registers and no scratch, whatever that path computes — a division-free form, a uniform
  // branch only, no global fetch: all tried, 332-384 bytes; the allocator loses the plot across
  // 32 conditional blocks.  Its critic loop is therefore ROLLED over the quads, the per-step values
  // in register vectors read through one scalar index: 251 registers, no scratch.  Parity-green —
  // and no faster than the lane pass (65 536 x 64: 50.0 against 49.9 us per tick): a half chain at
  // twice the instructions per step.  Not selected by the host: plan_launch.)
  constexpr bool PARK_VY = NSEG == 2;
  const SmpcTickPtrs tk{p.u, p.px, p.py, p.pyaw, p.D, p.pf_idx, p.pvalid, p.pa_active, p.pang_active, p.pal_active};
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint8_t* s_map = smem;
  const SmpcLut* s_lut = reinterpret_cast<const SmpcLut*>(smem + L.off_lut);
  float* s_px = reinterpret_cast<float*>(smem + L.off_px);
  float* s_py = reinterpret_cast<float*>(smem + L.off_py);
  float* s_D = reinterpret_cast<float*>(smem + L.off_D) + 1;           // sentinels at [-1] and [S]
  f32x4* s_pts4 = reinterpret_cast<f32x4*>(smem + L.off_pts4);
  // in front of the per-wave scratch: sum u^2 per control [4], then u [3][64]
  float* s_u = reinterpret_cast<float*>(smem + L.off_scr) - 3 * 64;
  float* s_su2 = s_u - 4;

  const int blk = blockDim.x;
  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwave = blockDim.x >> 6;
  const int r = lane & (SPLIT_ROLL - 1), sg = lane / SPLIT_ROLL;
  // per wave: parked wz [SPLIT_SEG][64], weights are not needed in LDS here; the head is re-used by the block combine
  float* park = reinterpret_cast<float*>(smem + L.off_scr) + (size_t)wave * L.scr_stride;
  float* park_y = park + SPLIT_SEG * WAVE;   // (PARK_VY)

  // ---- stage costmap window, LUT, path and u into LDS (one memory round trip) ----------------
  {
    const int ww = p.win_w, wh = p.win_h;
    const bool vec = ((ww & 3) == 0) && ((p.W & 3u) == 0) && ((p.win_x0 & 3) == 0);
    const int w4 = ww >> 2, n4 = vec ? w4 * wh : 0;
    auto word = [&](int i) -> uint32_t {
      const int ry = i / w4, rx = i - ry * w4;
      return reinterpret_cast<const uint32_t*>(p.map + (size_t)(p.win_y0 + ry) * p.W + p.win_x0)[rx];
    };
    constexpr int kAhead = 96 * 96 / 4 / SPLIT_BLOCK + 1;
    uint32_t tmp[kAhead];
#pragma unroll
    for (int k = 0; k < kAhead; ++k) {
      const int i = tid + k * blk;
      tmp[k] = i < n4 ? word(i) : 0u;
    }
    const SmpcLut lut_e = tid < 256 ? p.lut[tid] : SmpcLut{0.f, 0.f};
    const bool pt_on = (uint32_t)tid < p.P, seg_on = (uint32_t)tid + 1 < p.P;
    const float g_px = pt_on ? tk.px[tid] : 0.f, g_py = pt_on ? tk.py[tid] : 0.f;
    const float g_D = seg_on ? tk.D[tid] : 0.f;
    const bool g_valid = seg_on && tk.pvalid[tid] != 0;
    // u [3][T] -> LDS [3][64], zeros behind the horizon
    const uint32_t u_c = (uint32_t)tid >> 6, u_t = (uint32_t)tid & 63u;
    const float g_u = (tid < 3 * 64 && u_t < T) ? tk.u[u_c * T + u_t] : 0.f;
#pragma unroll
    for (int k = 0; k < kAhead; ++k) {
      const int i = tid + k * blk;
      if (i < n4) reinterpret_cast<uint32_t*>(s_map)[i] = tmp[k];
    }
    for (int i = tid + kAhead * blk; i < n4; i += blk) reinterpret_cast<uint32_t*>(s_map)[i] = word(i);
    if (!vec) {
      for (int i = tid; i < ww * wh; i += blk) {
        const int ry = i / ww, rx = i - ry * ww;
        s_map[i] = p.map[(size_t)(p.win_y0 + ry) * p.W + p.win_x0 + rx];
      }
    }
    if (tid < 256) const_cast<SmpcLut*>(s_lut)[tid] = lut_e;
    if (tid == 0) s_map[ww * wh] = 255;   // "off the map": NO_INFORMATION (obstacles_critic.cpp:209-212)
    for (uint32_t i = p.P + tid; i < ((p.P + 3u) & ~3u); i += blk) s_px[i] = s_py[i] = 1.0e18f;
    if (pt_on) {
      s_px[tid] = g_px;
      s_py[tid] = g_py;
      if (seg_on) s_D[tid] = g_D;
      s_pts4[tid] = f32x4{g_px, g_py, g_valid ? 1.0f : 0.f, 0.f};
    }
    for (uint32_t i = tid + blk; i < p.P; i += blk) {
      const float qx = tk.px[i], qy = tk.py[i];
      const bool seg = i + 1 < p.P;
      s_px[i] = qx;
      s_py[i] = qy;
      if (seg) s_D[i] = tk.D[i];
      s_pts4[i] = f32x4{qx, qy, (seg && tk.pvalid[i]) ? 1.0f : 0.f, 0.f};
    }
    if (tid < 3 * 64) s_u[tid] = g_u;
    if (tid < WAVE) {   // sum u^2 per control (the gamma terms as sum u c - sum u^2)
      const bool in = (uint32_t)tid < T;
      float a = in ? tk.u[tid] : 0.f, b = in ? tk.u[T + tid] : 0.f, c = in ? tk.u[2 * T + tid] : 0.f;
      a *= a;
      b *= b;
      c *= c;
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
  }
  __syncthreads();

  // ---- constants ------------------------------------------------------------------------------
  const uint32_t B = p.B;
  const uint32_t noise_bytes = T * SMPC_GM_ROLLOUTS(B) * 4u;   // one tensor, group-major (smpc_dev.h)
  const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.tvx), 0, 3u * noise_bytes, 0x00020000);
  constexpr uint32_t row_bytes = 256u;                         // step t + 1 of a 64-rollout group: 256 bytes on
  const float dt = p.dt, yaw0 = p.yaw0;
  const double x0 = p.x0, y0 = p.y0;
  uint32_t S = 0;
  if (p.flags & SD_NEED_FURTHEST) {
    S = p.d_furthest ? smpc_furthest_index(*p.d_furthest) : p.furthest_hint;
    if (S >= p.P) S = p.P ? p.P - 1 : 0;
  }
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
  const float k_rinv = p.rinvf, k_cx = p.cxf, k_cy = p.cyf;
  const float k_edge = 0.5f - p.cell_eps_w;

  // ---- per-wave running softmax state; U[ctrl][t] lives in lane t -----------------------------
  float m_run = 3.0e38f, s_run = 0.f;
  float Ux = 0.f, Uy = 0.f, Uz = 0.f;
  float F_local = 0.f;
  uint32_t n_noncoll = 0;

  const uint32_t ngroups = (B + SPLIT_ROLL - 1) / SPLIT_ROLL;
  const uint32_t gw = blockIdx.x * nwave + wave;
  const uint32_t nW = gridDim.x * nwave;
  const uint32_t t0 = (uint32_t)sg * SPLIT_SEG;

  for (uint32_t grp = gw; grp < ngroups; grp += nW) {
    const uint32_t b = grp * SPLIT_ROLL + (uint32_t)r;
    const bool live = b < B;
    const uint32_t bl = live ? b : B - 1;
    // noise row t0 + i of this lane: the lane's own part (rollout, first row of its segment) is the
    // vector offset, tensor and i the scalar one; the row in front of the segment separately
    const uint32_t voff = ((bl >> 6) * T * 64u + (bl & 63u)) * 4u + t0 * row_bytes;   // SMPC_GM_INDEX(bl, t0, T) in 32 bits
    const uint32_t vprev = sg ? voff - row_bytes : voff;   // (segment 0: loaded, not used)
    auto ld = [&](uint32_t tensor, uint32_t i) -> float {
      return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rn, voff, tensor * noise_bytes + i * row_bytes, 0));
    };
    auto ld_prev = [&](uint32_t tensor) -> float {
      return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rn, vprev, tensor * noise_bytes, 0));
    };
    const float* ux_s = s_u + t0;
    const float* uy_s = s_u + 64 + t0;
    const float* uz_s = s_u + 128 + t0;
    auto act = [&](int i) -> bool {return FULL || t0 + (uint32_t)i < T;};   // (the same for the 16 lanes of a row)

    // ================= A: yaw ====================================================================
    // v[:, t] = c[:, t - 1], v[:, 0] = the measured speed (optimizer.cpp:258-267)
    float ya[SPLIT_SEG];     // acc_yaw - (its value at the segment's start), inclusive
    float wprev;             // c_wz of the step in front of the segment
    {
      float nz[SPLIT_SEG];
      const float nzp = ld_prev(2);
#pragma unroll
      for (int i = 0; i < SPLIT_SEG; ++i) nz[i] = ld(2, i);
      const float uzp = s_u[128 + (sg ? t0 - 1 : 0)];
      wprev = sg ? uzp + nzp : p.swz;
      float acc = 0.f, w = wprev;
#pragma unroll
      for (int i = 0; i < SPLIT_SEG; ++i) {
        acc = acc + w * dt;
        ya[i] = acc;
        w = act(i) ? uz_s[i] + nz[i] : 0.f;       // NoiseGenerator::setNoisedControls (noise_generator.cpp:65-74)
        park[i * WAVE + lane] = w;                // parked for the update
      }
    }
    const float yaw_off = seg_exclusive<NSEG>(ya[SPLIT_SEG - 1], r, sg);

    // ================= B: displacement of the segment ============================================
    // Per-step values a lane keeps across the phases live in register VECTORS, element Q j + k for
    // step 4 k + j (Q = steps per lane / 4): the rolled critic loop of the two-segment instance
    // reads the four steps of quad k with one scalar index (s_set_gpr_idx, as smpc_lane.hip parks)
    constexpr int Q = SPLIT_SEG / 4;
#define SLOT(i) (Q * ((i) & 3) + ((i) >> 2))
    typename vecN<SPLIT_SEG>::type cvx;                   // parked noised vx of the lane's own steps
    float cvy[PARK_VY ? 1 : SPLIT_SEG];                   // ... vy (or in LDS: PARK_VY)
    typename vecN<SPLIT_SEG>::type AX, AY;                // displacement from the segment's start, inclusive
    float vxp, vyp;                         // state velocities of the segment's first step
    {
      const float nxp = ld_prev(0), nyp = ld_prev(1);
      float ny[SPLIT_SEG];
#pragma unroll
      for (int i = 0; i < SPLIT_SEG; ++i) {
        cvx[SLOT(i)] = ld(0, i);
        ny[i] = ld(1, i);
      }
      vxp = sg ? s_u[t0 - 1] + nxp : p.svx;
      vyp = sg ? s_u[64 + t0 - 1] + nyp : p.svy;
#pragma unroll
      for (int i = 0; i < SPLIT_SEG; ++i) {
        cvx[SLOT(i)] = act(i) ? ux_s[i] + cvx[SLOT(i)] : 0.f;
        ny[i] = act(i) ? uy_s[i] + ny[i] : 0.f;
        if constexpr (PARK_VY) park_y[i * WAVE + lane] = ny[i];
        else cvy[i] = ny[i];
      }
      // cos_[t] = cos(yaw[t - 1]) (optimizer.cpp:319-326): the segment's first step uses the yaw at
      // the end of the segment before it, which IS the prefix
      float cs_prev = p.cos0, sn_prev = p.sin0;
      if (sg) smpc_sincos(yaw_off + yaw0, sn_prev, cs_prev);
      float ax = 0.f, ay = 0.f;
#pragma unroll
      for (int i = 0; i < SPLIT_SEG; ++i) {
        const float vx = i ? cvx[SLOT(i - 1)] : vxp, vy = i ? ny[i - 1] : vyp;
        const float dxr = vx * cs_prev - vy * sn_prev;
        const float dyr = vx * sn_prev + vy * cs_prev;
        ax = ax + dxr * dt;
        ay = ay + dyr * dt;
        AX[SLOT(i)] = ax;
        AY[SLOT(i)] = ay;
        if (i + 1 < SPLIT_SEG) smpc_sincos((yaw_off + ya[i]) + yaw0, sn_prev, cs_prev);
      }
    }
    const float x_off = seg_exclusive<NSEG>(AX[SLOT(SPLIT_SEG - 1)], r, sg);
    const float y_off = seg_exclusive<NSEG>(AY[SLOT(SPLIT_SEG - 1)], r, sg);

    // ================= C: critics at the absolute position =======================================
    float crit = 0.f, rep = 0.f, alive = 1.0f;   // ObstaclesCritic, masked inside the segment
    float pfw = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
    typename vecN<NSAMP>::type sxv, syv;         // PathAlign's sample points of this lane: steps t0 + 0, 4, 8, ...
    float x_end = 0.f, y_end = 0.f;
    float vx_state = vxp;                        // state vx of the step at hand: the step before's noised control
    // one step; i = 4 k + j with j known at compile time, k either so (unrolled) or a scalar (rolled)
    auto step_c = [&](const int k, const int j) {
      const int i = 4 * k + j;
      const float axa = x_off + AX[Q * j + k], aya = y_off + AY[Q * j + k];
      // ObstaclesCritic lookup (obstacles_critic.cpp:139-171): window-relative float cell index
      // with its guard band, the exact double path for the lanes near a cell edge / outside the
      // window / off the map (smpc_lane.hip has the argument)
      const float qx = fmaf(axa, k_rinv, k_cx), qy = fmaf(aya, k_rinv, k_cy);
      const float rx = __builtin_amdgcn_fractf(qx), ry = __builtin_amdgcn_fractf(qy);
      const int lx = cvt_floor_i32(qx), ly = cvt_floor_i32(qy);
      const float edge = fmaxf(fabsf(rx - 0.5f), fabsf(ry - 0.5f));
      const bool fast = (edge <= k_edge) & ((uint32_t)lx < (uint32_t)p.win_w) & ((uint32_t)ly < (uint32_t)p.win_h);
      uint32_t idx = __umul24((uint32_t)ly, (uint32_t)p.win_w) + (uint32_t)lx;
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(!fast) != 0, 0)) {
        if (!fast)
          idx = cell_byte_exact(p, s_map, (float)(x0 + (double)axa), (float)(y0 + (double)aya), (uint32_t)(wave * WAVE + lane));
      }
      SmpcLut e = s_lut[s_map[idx]];
      const bool on_i = act(i);
      if (!FULL) {
        e.crit = on_i ? e.crit : 0.f;
        e.rep = on_i ? e.rep : 0.f;
      }
      alive = e.crit < 0.f ? 0.f : alive;    // inCollision: the rest of the segment is not visited
      crit = fmaf(alive, e.crit, crit);
      rep = fmaf(alive, e.rep, rep);
      // PreferForwardCritic (prefer_forward_critic.cpp:42-46) as -dt sum min(vx, 0)
      pfw = pfw + ((FULL || on_i) ? fminf(vx_state, 0.f) : 0.f);
      // gamma terms (optimizer.cpp:365-380) as sum u c - sum u^2
      const float cx_i = cvx[Q * j + k];
      gx = fmaf(ux_s[i], cx_i, gx);
      if constexpr (PARK_VY) gy = fmaf(uy_s[i], park_y[i * WAVE + lane], gy);
      gz = fmaf(uz_s[i], park[i * WAVE + lane], gz);
      vx_state = cx_i;
      if (j == 0) {   // PathAlign's trajectory points: every fourth step (trajectory_point_step 4)
        sxv[k] = (float)(x0 + (double)axa);
        syv[k] = (float)(y0 + (double)aya);
      }
    };
    if constexpr (NSEG == 2) {
      // ROLLED over the quads: 32 unrolled steps, each with the exact cell path's branch, are
      // what made this instance spill (340 bytes at 256 registers; without the branches 242 and none)
#pragma unroll 1
      for (int k = 0; k < Q; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) step_c(k, j);
      }
    } else {
#pragma unroll
      for (int k = 0; k < Q; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          step_c(k, j);
          if constexpr (!PARK_VY) gy = fmaf(uy_s[4 * k + j], cvy[PARK_VY ? 0 : 4 * k + j], gy);
        }
      }
    }
    // the endpoint, trajectory point T - 1: slot (T - 1) % (steps per lane) of segment (T - 1) / (steps per lane)
    const uint32_t e_seg = FULL ? (uint32_t)(NSEG - 1) : (T - 1u) / (uint32_t)SPLIT_SEG;
    if constexpr (FULL) {
      const float axa = x_off + AX[SLOT(SPLIT_SEG - 1)], aya = y_off + AY[SLOT(SPLIT_SEG - 1)];
      x_end = (float)(x0 + (double)axa);
      y_end = (float)(y0 + (double)aya);
    } else {
      const uint32_t e_loc = (T - 1u) % (uint32_t)SPLIT_SEG;
#pragma unroll
      for (int i = 3; i < SPLIT_SEG; i += 4) {   // (T a multiple of four: the last step sits in slot 3 mod 4)
        if ((uint32_t)i == e_loc) {
          x_end = (float)(x0 + (double)(x_off + AX[SLOT(i)]));
          y_end = (float)(y0 + (double)(y_off + AY[SLOT(i)]));
        }
      }
    }
    float sx[NSAMP], sy[NSAMP];
#pragma unroll
    for (int j = 0; j < NSAMP; ++j) {
      sx[j] = sxv[j];
      sy[j] = syv[j];
    }

    // ---- ObstaclesCritic over the segments: a collision masks every later segment --------------
    {
      float before = 1.0f;
      const float a0 = lane_get(alive, r);
      if constexpr (NSEG == 2) {
        before = sg == 0 ? 1.0f : a0;
      } else {
        const float a1 = lane_get(alive, r + SPLIT_ROLL), a2 = lane_get(alive, r + 2 * SPLIT_ROLL);
        before = sg == 0 ? 1.0f : (sg == 1 ? a0 : (sg == 2 ? a0 * a1 : a0 * a1 * a2));
      }
      crit = seg_total<NSEG>(before * crit);
      rep = seg_total<NSEG>(before * rep);
      alive = before * alive;                 // (the last segment's lane: the whole rollout; idle segments leave it alone)
      alive = lane_get(alive, r + (NSEG - 1) * SPLIT_ROLL);
    }
    pfw = seg_total<NSEG>(pfw);
    gx = seg_total<NSEG>(gx);
    gy = seg_total<NSEG>(gy);
    gz = seg_total<NSEG>(gz);
    // the endpoint (trajectory point T - 1) of the rollout, in its four lanes
    x_end = lane_get(x_end, r + (int)e_seg * SPLIT_ROLL);
    y_end = lane_get(y_end, r + (int)e_seg * SPLIT_ROLL);

    // ================= PathAlignCritic (path_align_critic.cpp:92-133) ===========================
    float pa_sum = 0.f, pa_num = 0.f;
    if (pa_on) {
      // chord lengths between consecutive sample points; the sample in front of a segment's first is
      // the last one of the segment before it, trajectory point 0 for the rollout's first (the
      // same for every rollout: host-computed, same arithmetic)
      const float px3 = __shfl_up(sx[NSAMP - 1], SPLIT_ROLL, WAVE), py3 = __shfl_up(sy[NSAMP - 1], SPLIT_ROLL, WAVE);
      float cl[NSAMP];
      float run = 0.f;
#pragma unroll
      for (int j = 0; j < NSAMP; ++j) {
        float qx, qy;
        if (j == 0) {
          qx = px3;
          qy = py3;
        } else if (j == 1) {
          qx = sg ? sx[0] : p.x00f;
          qy = sg ? sy[0] : p.y00f;
        } else {
          qx = sx[j - 1];
          qy = sy[j - 1];
        }
        const float ddx = sx[j] - qx, ddy = sy[j] - qy;
        const float ch = fast_sqrt(ddx * ddx + ddy * ddy);
        if (!(j == 0 && sg == 0) && act(4 * j)) run = run + ch;   // (step 0 is not a sample; nor are the slots behind the horizon)
        cl[j] = run;
      }
      const float d_off = seg_exclusive<NSEG>(run, r, sg);
      // per sample: std::lower_bound over D[0..S) and the candidate point the rule would pick if
      // the lower bound were not the previous point (utils.hpp:665-675, smpc_lane.hip)
      uint32_t lo_j[NSAMP], cand[NSAMP];
#pragma unroll
      for (int j = 0; j < NSAMP; ++j) {
        const float dist = d_off + cl[j];
        uint32_t gi = (uint32_t)(dist * pa_inv_spacing);
        gi = gi < S ? gi : S - 1;
        const float da = s_D[(int)gi - 1];
        const float db = s_D[gi];
        const float dc = s_D[gi + 1];
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
        lo_j[j] = lo;
        cand[j] = lo >= S ? S - 1 : ((dist - dl < dh - dist) ? lo - 1 : lo);   // end(): size - 1
      }
      // the one sequential dependency: "iter == begin + init" returns 0 — the previous sample's
      // point travels through the samples of a segment and from segment to segment
      uint32_t pt_j[NSAMP];
#pragma unroll
      for (int j = 0; j < NSAMP; ++j) pt_j[j] = 0u;
      uint32_t state = 0;
#pragma unroll
      for (int s = 0; s < NSEG; ++s) {
        uint32_t st = state;
#pragma unroll
        for (int j = 0; j < NSAMP; ++j) {
          if (s == 0 && j == 0) continue;   // step 0 is not a sample
          if (!FULL && !((uint32_t)(SPLIT_SEG * s + 4 * j) < T)) continue;   // (uniform: behind the horizon)
          const uint32_t ptv = lo_j[j] == st ? 0u : cand[j];
          if (sg == s) pt_j[j] = ptv;
          st = ptv;
        }
        state = (uint32_t)__shfl((int)st, r + SPLIT_ROLL * s, WAVE);   // what segment s left behind
      }
#pragma unroll
      for (int j = 0; j < NSAMP; ++j) {
        const f32x4 q = s_pts4[pt_j[j]];
        const float ex = q[0] - sx[j], ey = q[1] - sy[j];
        const float d = fast_sqrt(ex * ex + ey * ey);
        const float on = ((j == 0 && sg == 0) || !act(4 * j)) ? 0.f : q[2];   // segment valid ? 1 : 0 (path_align_critic.cpp:119-127)
        pa_num += on;
        pa_sum += on != 0.f ? d : 0.f;      // (a select: an idle slot's distance may be anything)
      }
      pa_num = seg_total<NSEG>(pa_num);
      pa_sum = seg_total<NSEG>(pa_sum);
    }

    // ================= per rollout (identical in its four lanes) =================================
    if (want_local_furthest) {
      // nearest path point of the endpoint (utils.hpp:292-319), first minimum wins: every lane
      // scans a quarter of the path (blocks of four points), the four are merged with the lower
      // index winning a tie
      const uint32_t P4 = (p.P + 3u) & ~3u;
      const uint32_t nb = P4 >> 2, per = (nb + (uint32_t)NSEG - 1u) / (uint32_t)NSEG;
      const uint32_t j0 = (uint32_t)sg * per * 4u, j1 = min(j0 + per * 4u, P4);
      float best = 3.4028234663852886e38f;
      uint32_t bi = 0;
      for (uint32_t j = j0; j < j1; j += 4) {
        const f32x4 qx = *reinterpret_cast<const f32x4*>(s_px + j);
        const f32x4 qy = *reinterpret_cast<const f32x4*>(s_py + j);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ex = qx[e] - x_end, ey = qy[e] - y_end;
          const float dd = ex * ex + ey * ey;
          if (dd < best) {
            best = dd;
            bi = j + (uint32_t)e;
          }
        }
      }
#pragma unroll
      for (int o = SPLIT_ROLL; o <= 32; o <<= 1) {
        const float ob = __shfl_xor(best, o, WAVE);
        const uint32_t oi = (uint32_t)__shfl_xor((int)bi, o, WAVE);
        const bool take = ob < best || (ob == best && oi < bi);
        best = take ? ob : best;
        bi = take ? oi : bi;
      }
      float F = (float)bi;
      if (bi + 1 < p.P) {
        const float nx = s_px[bi + 1], ny = s_py[bi + 1];
        const float sgx = nx - s_px[bi], sgy = ny - s_py[bi];
        const float d_next = (nx - x_end) * (nx - x_end) + (ny - y_end) * (ny - y_end);
        const float seg2 = sgx * sgx + sgy * sgy;
        const float tt = seg2 > 0.f ? 0.5f + 0.5f * (best - d_next) * fast_rcp(seg2) : 0.f;
        F = fmaxf(F + fminf(fmaxf(tt, -0.45f), 0.45f), 0.f);
      }
      float m = live ? F : 0.f;
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, WAVE));
      F_local = fmaxf(F_local, m);
    }
    // costs (every cost_power == 1): the association of smpc_pass_lane
    float cost = (p.flags & SD_ACCUMULATE) ? p.costs_prev[bl] : 0.f;
    const bool collided = alive == 0.f;
    float lin = (collided ? 0.f : p.obs_critical_w * crit) + p.obs_rep_over_T * rep;
    float uni = collided ? p.obs_critical_w * p.obs_collision_cost : 0.f;
    n_noncoll += (uint32_t)__popcll(__ballot(live && !collided && sg == 0));
    if (p.flags & SD_PATH_FOLLOW) {
      const float fdx = x_end - pf_x, fdy = y_end - pf_y;
      uni += p.pf_weight * fast_sqrt(fdx * fdx + fdy * fdy);
    }
    if (p.flags & SD_PREFER_FORWARD) lin += (pfw * -dt) * p.pfw_weight;
    lin += p.g_vx * (gx - s_su2[0]);
    lin += p.g_wz * (gz - s_su2[2]);
    lin += p.g_vy * (gy - s_su2[1]);
    cost += uni + lin;
    if (pa_on) {
      const float c_pa = pa_num > 0.f ? pa_sum * fast_rcp(pa_num) : 0.f;
      cost += c_pa * p.pa_weight;
    }
    if (live && sg == 0) p.costs[b] = cost;

    // ---- softmax of the group (optimizer.cpp:382-391 as an online sum) --------------------------
    float cmin = live ? cost : 3.0e38f;
    for (int o = 32; o > 0; o >>= 1) cmin = fminf(cmin, __shfl_xor(cmin, o, WAVE));
    const float m_new = fminf(m_run, cmin);
    const float f = __builtin_amdgcn_exp2f(p.k2 * (m_run - m_new));
    const float w = live ? __builtin_amdgcn_exp2f(p.k2 * (cost - m_new)) : 0.f;   // the same in the rollout's four lanes
    float wsum = sg == 0 ? w : 0.f;
    for (int o = SPLIT_ROLL / 2; o > 0; o >>= 1) wsum += __shfl_xor(wsum, o, WAVE);   // over the rollouts: the lanes of segment 0
    wsum = lane_get(wsum, 0);
    s_run = fmaf(s_run, f, wsum);
    m_run = m_new;

    // ================= U[t] += sum_b w_b c[b][t]: 16 x 16 transpose-reduce per row ===============
    {
      float V[SPLIT_SEG];
#pragma unroll
      for (int i = 0; i < SPLIT_SEG; ++i) V[i] = w * cvx[SLOT(i)];
      Ux = fmaf(Ux, f, row_reduce<SPLIT_SEG>(V, lane));
#pragma unroll
      for (int i = 0; i < SPLIT_SEG; ++i) V[i] = w * (PARK_VY ? park_y[i * WAVE + lane] : cvy[PARK_VY ? 0 : i]);
      Uy = fmaf(Uy, f, row_reduce<SPLIT_SEG>(V, lane));
#pragma unroll
      for (int i = 0; i < SPLIT_SEG; ++i) V[i] = w * park[i * WAVE + lane];
      Uz = fmaf(Uz, f, row_reduce<SPLIT_SEG>(V, lane));
    }
  }

  // ---- block combine -> one partial per block (the tuple of the other passes) --------------------
  __syncthreads();
  const uint32_t TL = 4 + 3 * T;
  float* myp = reinterpret_cast<float*>(smem + L.off_scr) + (size_t)wave * L.scr_stride;
  if (lane == 0) {
    myp[0] = m_run;
    myp[1] = s_run;
    myp[2] = F_local;
    myp[3] = (float)n_noncoll;
  }
  if (FULL || (uint32_t)lane < T) {
    myp[4 + lane] = Ux;
    myp[4 + T + lane] = Uy;
    myp[4 + 2 * T + lane] = Uz;
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
        const float sc = __builtin_amdgcn_exp2f(p.k2 * (mw - bm));
        acc += sc * allp[(size_t)w * L.scr_stride + i];
      }
    }
    outp[i] = acc;
  }
  // which tick block this launch read (SmpcDev::canary_echo)
  if (p.canary_echo && !(p.flags & SD_ACCUMULATE) && blockIdx.x == 0 && tid == 0)
    p.partials[SMPC_CANARY_SLOT(T)] = tk.u[-4];
#undef SLOT
}

extern char smpc_last_pass_kernel[96];   // smpc_kernels.hip

hipError_t smpc_launch_pass_split(const SmpcDev& p, const SmpcLds& L, uint32_t grid, uint32_t nseg, hipStream_t st)
{
  if (nseg != 2u && nseg != 4u) return hipErrorInvalidValue;
  const uint32_t need = SD_OBSTACLES, never = SD_GOAL_ANGLE | SD_EXTRA_CRITICS | SD_STORE_TRAJ | SD_USE_PATH_YAW;
  if (p.T > 64u || p.T < 20u || (p.T & 3u) || (p.flags & need) != need || (p.flags & never) ||
      ((p.flags & SD_PATH_ALIGN) && p.step != 4u))
    return hipErrorInvalidValue;
  const bool full = p.T == 64u;
  snprintf(smpc_last_pass_kernel, sizeof(smpc_last_pass_kernel), "smpc_pass_split<%u, %s>", nseg, full ? "true" : "false");
  if (nseg == 4u && full) hipLaunchKernelGGL((smpc_pass_split<4, true>), dim3(grid), dim3(SPLIT_BLOCK), L.total, st, p, L);
  else if (nseg == 4u) hipLaunchKernelGGL((smpc_pass_split<4, false>), dim3(grid), dim3(SPLIT_BLOCK), L.total, st, p, L);
  else if (full) hipLaunchKernelGGL((smpc_pass_split<2, true>), dim3(grid), dim3(SPLIT_BLOCK), L.total, st, p, L);
  else return hipErrorInvalidValue;   // (the two-segment instance: experiments at T = 64 only)
  return hipGetLastError();
}

uint32_t smpc_split_block() {return SPLIT_BLOCK;}
uint32_t smpc_split_rollouts_per_block(uint32_t nseg) {return SPLIT_BLOCK / WAVE * (64u / nseg);}

hipError_t smpc_split_occupancy(uint32_t nseg, uint32_t lds_bytes, int* blocks_per_cu)
{
  const void* k = nseg == 4u ? reinterpret_cast<const void*>(&smpc_pass_split<4, true>)
                             : reinterpret_cast<const void*>(&smpc_pass_split<2, true>);
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, k, SPLIT_BLOCK, lds_bytes);
}

hipError_t smpc_split_set_lds_limit(int bytes)
{
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&smpc_pass_split<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&smpc_pass_split<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&smpc_pass_split<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  return e;
}
