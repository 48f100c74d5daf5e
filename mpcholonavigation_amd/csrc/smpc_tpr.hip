// smpc_tpr.hip — an alternative streaming pass: one lane per rollout.
//
// STATUS (round 1): parity-green on every test shape, NOT the default.  On the bench
// workload (262 144 x 64) it issues ~28 % fewer VALU instructions than the wave-per-rollout
// pass but ends at 121-128 us vs 118 us: the matrix-core phase (re-read of the noise through
// Infinity Cache) costs ~35 us that nothing overlaps when every wave owns exactly one group,
// and the register allocation under the 128-VGPR cap spills.  Enable with
// SMPC_FLAG_LANE_PER_ROLLOUT (or SMPC_PASS=lane).  See DESIGN.md §4.4.
//
// smpc_kernels.hip gives every rollout a whole wavefront (lane = time step): the
// lowest latency and the right shape for the reference's deployed batch (2 000
// rollouts).  It pays for that with ~35 cross-lane DPP steps per rollout and with
// per-rollout scalar work executed 64 lanes wide; on CDNA4 a wave64 VALU op costs
// 4 cycles, and at 10^5..10^6 rollouts that issue rate, not HBM, is the bound.
//
// Here a wave owns 64 rollouts (lane = rollout) and walks the horizon step by
// step, so every per-step operation is one instruction for 64 rollouts with NO
// cross-lane traffic, the float cumulative sums run in the reference's own
// sequential order, and per-rollout work (PathAlign samples, nearest path point,
// cost assembly with the reference's float/double mix) is amortised 64-fold.
//   * noise is read from a time-major copy [T][B]: one coalesced 256-B row piece
//     per array and step;
//   * the weighted-control update  U[t] = sum_b w_b c[b,t]  — the one dense
//     contraction of the path, a (1 x 64)·(64 x 3T) product per wave — goes to the
//     matrix cores (v_mfma_f32_16x16x4_f32, exact f32 fma chain) with the noise
//     re-read through L2 / Infinity Cache, so it costs the VALU ~3 ops per rollout;
//   * everything batch-wide (block partials, reduction, combine) is shared with
//     the wave-per-rollout path.
//
// Compiled with -ffp-contract=off like smpc_kernels.hip.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smpc_dev.h"
#include "smpc_device_math.h"

#define WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// [B][T] -> [T][B] (one-off, after the noise is drawn or supplied)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) smpc_transpose_bt(const float* __restrict__ src,
                                                        float* __restrict__ dst, uint32_t B,
                                                        uint32_t T)
{
  __shared__ float tile[32][33];
  const uint32_t bx = blockIdx.x * 32, ty0 = blockIdx.y * 32;
  const uint32_t lx = threadIdx.x & 31, ly = threadIdx.x >> 5;   // 32 x 8
  for (uint32_t k = ly; k < 32; k += 8) {
    const uint32_t b = bx + k, t = ty0 + lx;
    tile[k][lx] = (b < B && t < T) ? src[(size_t)b * T + t] : 0.f;
  }
  __syncthreads();
  for (uint32_t k = ly; k < 32; k += 8) {
    const uint32_t t = ty0 + k, b = bx + lx;
    if (b < B && t < T) dst[(size_t)t * B + b] = tile[lx][k];
  }
}

hipError_t smpc_launch_transpose(const float* src, float* dst, uint32_t B, uint32_t T,
                                 hipStream_t st)
{
  hipLaunchKernelGGL(smpc_transpose_bt, dim3((B + 31) / 32, (T + 31) / 32), dim3(256), 0, st, src,
                     dst, B, T);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// The pass.  One block = TPR_WAVES waves; LDS: costmap window, LUT, path, and per
// wave {weights[64], U tile staging[3*Tpad]}.
// ---------------------------------------------------------------------------
template <int R>   // R = ceil(T / 64): registers holding U[ctrl][t] per lane
__global__ void __launch_bounds__(256, 4) smpc_pass_tpr(const SmpcDev p, const SmpcLds L)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint8_t* s_map = smem;
  const SmpcLut* s_lut = reinterpret_cast<const SmpcLut*>(smem + L.off_lut);
  float* s_px = reinterpret_cast<float*>(smem + L.off_px);
  float* s_py = reinterpret_cast<float*>(smem + L.off_py);
  float* s_pyaw = reinterpret_cast<float*>(smem + L.off_pyaw);
  float* s_D = reinterpret_cast<float*>(smem + L.off_D);
  uint8_t* s_valid = smem + L.off_valid;

  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwave = blockDim.x >> 6;
  float* scr = reinterpret_cast<float*>(smem + L.off_scr) + (size_t)wave * L.scr_stride;
  const float* s_uu = reinterpret_cast<const float*>(smem + L.off_scr) +
                      (size_t)nwave * L.scr_stride;   // [3T] control sequence (block-shared)
  float* s_w = scr;              // [64] softmax weights of the current group
  float* s_u = scr + WAVE;       // [3 * Tpad] weighted-control sums of the current group

  // ---- stage costmap window, LUT and path into LDS -------------------------
  if (p.flags & SD_OBSTACLES) {
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
    s_px[i] = p.px[i];
    s_py[i] = p.py[i];
    s_pyaw[i] = p.pyaw[i];
    if (i + 1 < p.P) {
      s_D[i] = p.D[i];
      s_valid[i] = p.pvalid[i];
    }
  }
  for (uint32_t i = tid; i < 3 * p.T; i += blockDim.x) const_cast<float*>(s_uu)[i] = p.u[i];
  __syncthreads();

  // ---- constants ------------------------------------------------------------
  const uint32_t T = p.T, B = p.B;
  const uint32_t Tpad = (T + 15u) & ~15u;
  const float dt = in_vgpr(p.dt), yaw0_v = in_vgpr(p.yaw0);
  const double x0_v = in_vgpr(p.x0), y0_v = in_vgpr(p.y0);
  const CellConsts cellk = {in_vgpr(p.oxf), in_vgpr(p.oyf), in_vgpr(p.rinvf), in_vgpr(p.cell_eps),
                            in_vgpr(1.0f - p.cell_eps)};
  const float svx_v = in_vgpr(p.svx), svy_v = in_vgpr(p.svy), swz_v = in_vgpr(p.swz);
  uint32_t S = 0;
  bool pa_on = false;
  float pf_x = 0.f, pf_y = 0.f;
  uint32_t bs_iters = 0;
  float pa_inv_spacing = 0.f;
  if (p.flags & SD_NEED_FURTHEST) {
    S = p.d_furthest ? (uint32_t)(*p.d_furthest) : p.furthest_hint;
    if (S >= p.P) S = p.P ? p.P - 1 : 0;
  }
  pa_on = (p.flags & SD_PATH_ALIGN) && p.P > 0 && p.pa_active[S] && S > 0;
  if ((p.flags & SD_PATH_FOLLOW) && p.P > 0) {
    const uint32_t idx = p.pf_idx[S];
    pf_x = s_px[idx];
    pf_y = s_py[idx];
  }
  bs_iters = S > 1 ? 32u - (uint32_t)__builtin_clz(S - 1) : 0u;
  if (pa_on && S > 1 && s_D[S - 1] > 0.f) pa_inv_spacing = (float)(S - 1) / s_D[S - 1];
  const bool want_local_furthest = (p.flags & SD_NEED_FURTHEST) && (p.flags & SD_LOCAL_FURTHEST);
  const uint32_t step = p.step;

  // ---- per-wave running softmax state; U[ctrl][t] lives in lane t % 64, register t / 64
  float m_run = 3.0e38f, s_run = 0.f;
  float Ux[R], Uy[R], Uz[R];
#pragma unroll
  for (int r = 0; r < R; ++r) Ux[r] = Uy[r] = Uz[r] = 0.f;
  uint32_t S_local = 0, n_noncoll = 0;

  const uint32_t ngroups = (B + WAVE - 1) / WAVE;
  const uint32_t gw = blockIdx.x * nwave + wave;
  const uint32_t nW = gridDim.x * nwave;

  for (uint32_t grp = gw; grp < ngroups; grp += nW) {
    const uint32_t b0 = grp * WAVE;
    const uint32_t b = b0 + lane;
    const bool live = b < B;
    const uint32_t bl = live ? b : B - 1;        // tail lanes shadow the last rollout
    // row t of the time-major noise: uniform base + this lane's 32-bit byte offset
    const uint32_t loff = bl * 4u;
    auto ld = [&](const float* base, uint32_t t) -> float {
      const char* row = reinterpret_cast<const char*>(base) + (size_t)t * B * 4u;   // uniform
      return *reinterpret_cast<const float*>(row + loff);
    };

    // ================= phase 1: rollout + per-step critics, lane = rollout ==========
    float cpx = 0.f, cpy = 0.f, cpz = 0.f;   // previous noised controls (state v[t] = c[t-1])
    float acc_yaw = 0.f, ax = 0.f, ay = 0.f;
    float cs_prev = p.cos0, sn_prev = p.sin0;
    float x = 0.f, y = 0.f, yaw = 0.f;
    float crit = 0.f, rep = 0.f;
    bool collided = false;
    float pfw = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
    double ga = 0.0;
    // PathAlign running state (path_align_critic.cpp:92-133)
    float traj_dist = 0.f, pa_sum = 0.f, pa_num = 0.f, sx_prev = 0.f, sy_prev = 0.f;
    uint32_t path_pt = 0;

    // one time step for the 64 rollouts of this wave; t is wave-uniform
    auto do_step = [&](const uint32_t t, const float n0, const float n1, const float n2) {
      const float ux = s_uu[t], uy = s_uu[T + t], uz = s_uu[2 * T + t];   // LDS broadcast reads
      // NoiseGenerator::setNoisedControls (noise_generator.cpp:65-74)
      const float cvx = ux + n0, cvy = uy + n1, cwz = uz + n2;
      // updateStateVelocities + predict: v[:,0] = speed, v[:,t] = c[:,t-1]
      const float vx = t == 0 ? svx_v : cpx, vy = t == 0 ? svy_v : cpy, wz = t == 0 ? swz_v : cpz;
      cpx = cvx;
      cpy = cvy;
      cpz = cwz;
      // integrateStateVelocities (optimizer.cpp:313-343): sequential float cumsums
      const float inc = wz * dt;
      acc_yaw = t == 0 ? inc : acc_yaw + inc;
      yaw = acc_yaw + yaw0_v;
      const float dxr = vx * cs_prev - vy * sn_prev;
      const float dyr = vx * sn_prev + vy * cs_prev;
      ax = t == 0 ? dxr * dt : ax + dxr * dt;
      ay = t == 0 ? dyr * dt : ay + dyr * dt;
      x = (float)(x0_v + (double)ax);
      y = (float)(y0_v + (double)ay);
      if (t + 1 < T) smpc_sincos(yaw, sn_prev, cs_prev);   // cos_[t+1] = cos(yaw[t])

      // ObstaclesCritic per step (obstacles_critic.cpp:139-171)
      if (p.flags & SD_OBSTACLES) {
        if (!collided) {
          const uint32_t c = t == 0 ? p.cost_t0 : cost_at(p, cellk, s_map, x, y);
          const SmpcLut e = s_lut[c];
          if (e.crit < 0.f) {
            collided = true;
          } else {
            crit += e.crit;
            rep += e.rep;
          }
        }
      }
      // PreferForwardCritic (prefer_forward_critic.cpp:42-46)
      if (p.flags & SD_PREFER_FORWARD) {
        const float term = fmaxf(-vx, 0.f) * dt;
        pfw = t == 0 ? term : pfw + term;
      }
      // GoalAngleCritic (goal_angle_critic.cpp:46-49)
      if (p.flags & SD_GOAL_ANGLE) ga += fabs(normalize_angle((double)(p.ga_goal_yaw - yaw)));
      // updateControlSequence gamma terms (optimizer.cpp:365-380)
      {
        const float tx = ux * (cvx - ux), tz = uz * (cwz - uz), ty = uy * (cvy - uy);
        gx = t == 0 ? tx : gx + tx;
        gz = t == 0 ? tz : gz + tz;
        gy = t == 0 ? ty : gy + ty;
      }
      // PathAlignCritic sample (uniform in t): points step, 2 step, ...
      if (pa_on) {
        if (t == 0) {
          sx_prev = x;
          sy_prev = y;
        } else if (step && t % step == 0) {
          const float ddx = x - sx_prev, ddy = y - sy_prev;
          traj_dist += sqrtf(ddx * ddx + ddy * ddy);
          sx_prev = x;
          sy_prev = y;
          // utils::findClosestPathPt(D, traj_dist, path_pt) (tools/utils.hpp:665-675)
          const float dist = traj_dist;
          uint32_t gi = (uint32_t)(dist * pa_inv_spacing);
          gi = gi < S ? gi : S - 1;
          const float da = gi > 0 ? s_D[gi - 1] : -3.0e38f;
          const float db = s_D[gi];
          const float dc = gi + 1 < S ? s_D[gi + 1] : 3.0e38f;
          uint32_t lo;
          float dl, dh;
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
          // lower_bound restricted to [path_pt, S): the global one, since path_pt <= lo
          uint32_t pt;
          if (lo == path_pt) pt = 0;                 // iter == begin + init
          else if (lo >= S) pt = S - 1;              // end(): defined as size-1
          else pt = (dist - dl < dh - dist) ? lo - 1 : lo;
          path_pt = pt;
          if (s_valid[pt]) {
            const float ex = s_px[pt] - x, ey = s_py[pt] - y;
            pa_num += 1.0f;
            if (p.flags & SD_USE_PATH_YAW) {
              const double dd = (double)yaw - (double)s_pyaw[pt];
              double a = fmod(fmod(dd, 2.0 * M_PI) + 2.0 * M_PI, 2.0 * M_PI);
              if (a > M_PI) a -= 2.0 * M_PI;
              const float dyaw = (float)a;
              pa_sum += sqrtf(ex * ex + ey * ey + dyaw * dyaw);
            } else {
              pa_sum += sqrtf(ex * ex + ey * ey);
            }
          }
        }
      }
    };
    // noise prefetched two steps ahead in two alternating register sets (no moves)
    float a0 = ld(p.tvx, 0), a1 = ld(p.tvy, 0), a2 = ld(p.twz, 0);
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if (T > 1) {
      c0 = ld(p.tvx, 1);
      c1 = ld(p.tvy, 1);
      c2 = ld(p.twz, 1);
    }
    for (uint32_t t = 0; t < T; t += 2) {
      const float u0 = a0, u1 = a1, u2 = a2;
      if (t + 2 < T) {
        a0 = ld(p.tvx, t + 2);
        a1 = ld(p.tvy, t + 2);
        a2 = ld(p.twz, t + 2);
      }
      do_step(t, u0, u1, u2);
      if (t + 1 < T) {
        const float w0 = c0, w1 = c1, w2 = c2;
        if (t + 3 < T) {
          c0 = ld(p.tvx, t + 3);
          c1 = ld(p.tvy, t + 3);
          c2 = ld(p.twz, t + 3);
        }
        do_step(t + 1, w0, w1, w2);
      }
    }

    // ================= phase 2: per-rollout epilogue, lane = rollout ==================
    // nearest path point of the endpoint (utils.hpp:292-319): first minimum wins
    if (want_local_furthest) {
      float best = 3.4028234663852886e38f;
      uint32_t bi = 0;
      for (uint32_t j = 0; j < p.P; ++j) {
        const float ddx = s_px[j] - x, ddy = s_py[j] - y;
        const float d = ddx * ddx + ddy * ddy;
        if (d < best) {
          best = d;
          bi = j;
        }
      }
      uint32_t m = live ? bi : 0u;
      for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, WAVE));
      S_local = max(S_local, m);
    }
    // costs in the reference's critic order and float/double mix
    float cost = (p.flags & SD_ACCUMULATE) ? p.costs_prev[bl] : 0.f;
    if (p.flags & SD_OBSTACLES) {
      const float raw = collided ? p.obs_collision_cost : crit;
      const float v = (p.obs_critical_w * raw) + (p.obs_repulsion_w * rep / (float)T);
      cost = add_cost_pow(cost, (double)v, p.obs_power);
      n_noncoll += (uint32_t)__popcll(__ballot(live && !collided));
    }
    if (pa_on) {
      const float c_pa = pa_num > 0.f ? pa_sum / pa_num : 0.f;
      cost = add_cost_pow(cost, (double)(c_pa * p.pa_weight), p.pa_power);
    }
    if (p.flags & SD_PATH_FOLLOW) {
      const double ddx = (double)(x - pf_x), ddy = (double)(y - pf_y);
      cost = add_cost_pow(cost, (double)p.pf_weight * sqrt(ddx * ddx + ddy * ddy), p.pf_power);
    }
    if (p.flags & SD_GOAL_ANGLE)
      cost = add_cost_pow(cost, (ga / (double)T) * (double)p.ga_weight, p.ga_power);
    if (p.flags & SD_PREFER_FORWARD)
      cost = add_cost_pow(cost, (double)(pfw * p.pfw_weight), p.pfw_power);
    cost += p.g_vx * gx;
    cost += p.g_wz * gz;
    cost += p.g_vy * gy;
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

    // ================= phase 3: U[t] += sum_b w_b c[b,t] on the matrix cores ============
    // D[16x16] += A[16x4] B[4x16]:  A[i][k] = w of rollout 4 kk + k (every row i alike),
    // B[k][j] = c[rollout 4 kk + k][t = 16 jt + j];  lane l holds A[l&15][l>>4], B[l>>4][l&15]
    // and D[row 4 (l>>4) + reg][col l&15].  Every row of D is the wanted sum.
    s_w[lane] = w;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t kq = (uint32_t)lane >> 4, jc = (uint32_t)lane & 15u;
    // k-step (m, e), m, e = 0..3, contracts rollouts 16 m + 4 kq + e (kq = 0..3): lane (kq, jc)
    // fetches its four e-values with ONE 16-byte load of row t = 16 jt + jc of the time-major
    // noise (the four kq lanes of a row read one contiguous 64-B sector).
    const uint32_t ntile = Tpad / 16, nitem = 3 * ntile;
    const bool full_group = b0 + WAVE <= B && (B & 3u) == 0;   // aligned 16-B loads in range
    auto item_loads = [&](uint32_t tile, f32x4 (&bv)[4], float& ut) {
      const uint32_t ctrl = tile / ntile, jt = tile - ctrl * ntile;
      const float* tn = ctrl == 0 ? p.tvx : (ctrl == 1 ? p.tvy : p.twz);
      const uint32_t t = jt * 16 + jc;
      const bool tin = t < T;
      ut = tin ? s_uu[ctrl * T + t] : 0.f;
      const float* col = tn + (size_t)(tin ? t : 0) * B;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const uint32_t rb = b0 + 16 * m + 4 * kq;
        if (full_group) {
          bv[m] = *reinterpret_cast<const f32x4*>(col + rb);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[m][e] = col[rb + e < B ? rb + e : B - 1];
        }
      }
    };
    // The A operand (weights of rollouts 16 m + 4 kq + e) is read from LDS where it is used:
    // one ds_read_b128 per m.  Two tiles are in flight (8 x 16-B loads) while one is multiplied.
    const f32x4* s_w4 = reinterpret_cast<const f32x4*>(s_w);
    f32x4 bva[4], bvb[4];
    float uta = 0.f, utb = 0.f;
    const uint32_t nit = nitem;
    auto item_mfma = [&](uint32_t it, const f32x4 (&bv)[4], const float ut) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const f32x4 wv = s_w4[4 * m + kq];   // w[16 m + 4 kq + 0..3]
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[e], ut + bv[m][e], acc, 0, 0, 0);
      }
      const uint32_t ctrl = it / ntile, jt = it - ctrl * ntile;
      if (lane < 16) s_u[ctrl * Tpad + jt * 16 + jc] = acc[0];   // row 0 of D
    };
    if (nit) item_loads(0, bva, uta);
    for (uint32_t it = 0; it < nit; it += 2) {
      if (it + 1 < nit) item_loads(it + 1, bvb, utb);
      item_mfma(it, bva, uta);
      if (it + 2 < nit) item_loads(it + 2, bva, uta);
      if (it + 1 < nit) item_mfma(it + 1, bvb, utb);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t t = (uint32_t)r * WAVE + lane;
      if (t < T) {
        Ux[r] = fmaf(Ux[r], f, s_u[t]);
        Uy[r] = fmaf(Uy[r], f, s_u[Tpad + t]);
        Uz[r] = fmaf(Uz[r], f, s_u[2 * Tpad + t]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }

  // ---- block combine -> one partial per block (same tuple as the wave-per-rollout pass)
  __syncthreads();
  const uint32_t TL = 4 + 3 * T;
  float* myp = reinterpret_cast<float*>(smem + L.off_scr) + (size_t)wave * L.scr_stride;
  if (lane == 0) {
    myp[0] = m_run;
    myp[1] = s_run;
    myp[2] = (float)S_local;
    myp[3] = (float)n_noncoll;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t t = (uint32_t)r * WAVE + lane;
    if (t < T) {
      myp[4 + t] = Ux[r];
      myp[4 + T + t] = Uy[r];
      myp[4 + 2 * T + t] = Uz[r];
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
    outp[i] = acc;
  }
}

static const void* tpr_kernel(int R)
{
  switch (R) {
    case 1: return reinterpret_cast<const void*>(&smpc_pass_tpr<1>);
    case 2: return reinterpret_cast<const void*>(&smpc_pass_tpr<2>);
    default: return reinterpret_cast<const void*>(&smpc_pass_tpr<4>);
  }
}

hipError_t smpc_launch_pass_tpr(const SmpcDev& p, const SmpcLds& L, uint32_t grid, uint32_t block,
                                hipStream_t st)
{
  const int R = p.T <= 64 ? 1 : (p.T <= 128 ? 2 : 4);
  switch (R) {
    case 1: hipLaunchKernelGGL(smpc_pass_tpr<1>, dim3(grid), dim3(block), L.total, st, p, L); break;
    case 2: hipLaunchKernelGGL(smpc_pass_tpr<2>, dim3(grid), dim3(block), L.total, st, p, L); break;
    default: hipLaunchKernelGGL(smpc_pass_tpr<4>, dim3(grid), dim3(block), L.total, st, p, L); break;
  }
  return hipGetLastError();
}

hipError_t smpc_tpr_occupancy(int R, uint32_t block, uint32_t lds_bytes, int* blocks_per_cu)
{
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, tpr_kernel(R), (int)block,
                                                      lds_bytes);
}

hipError_t smpc_tpr_set_lds_limit(int bytes)
{
  hipError_t e = hipSuccess;
  for (int R : {1, 2, 4})
    if (e == hipSuccess)
      e = hipFuncSetAttribute(tpr_kernel(R), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  return e;
}
