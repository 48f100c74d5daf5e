// smpc_lane_common.h — device helpers shared by the passes that walk the horizon step by step with
// lane = rollout (smpc_lane.hip) or lane = (rollout, segment of the horizon) (smpc_split.hip):
// vector typedefs, the in-register transpose-reduce over the lanes of a wave, the window-relative
// costmap cell index.  Internal; included by those two files only.
#ifndef SMPC_LANE_COMMON_H_
#define SMPC_LANE_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smpc_dev.h"
#include "smpc_device_math.h"

#ifndef WAVE
#define WAVE 64
#endif

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef const float __attribute__((address_space(4))) * cfloat_p;
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define LANE_PARK_STRIDE 68   // floats per time step of the LDS-parked control: 16-B aligned rows
                              // for ds_read_b128, 4-bank skew per lane (8 lanes cover the 32 banks)

// ---------------------------------------------------------------------------
// 64 x 64 transpose-reduce in registers.
//   in : V[t] in lane b = c[b][t]      (64 registers, lane = rollout)
//   out: lane t = sum_b w[b] * c[b][t]
// Butterfly over the lane bits 5..0; at the node for bit k a lane keeps the register half
// its own bit selects and receives the partner lane's copy of it, so after six levels the
// one remaining register of lane l belongs to t = l.  Depth-first, so only ~one register
// per level is live besides the inputs.
// ---------------------------------------------------------------------------
struct LaneW {      // first-level weights: {own, partner} ordered by the lane's bit 5
  float wa, wb;
};

// a + dpp(a) everywhere, then b + dpp(b) in the banks whose lanes keep the b half
#define SMPC_DPP_NODE(NAME, CTRL, BANKS)                                                    \
  __device__ __forceinline__ float NAME(float a, float b)                                   \
  {                                                                                         \
    float t;                                                                                \
    asm("s_nop 1\n\t"                                                              \
                 "v_add_f32_dpp %0, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"         \
                 "v_add_f32_dpp %0, %2, %2 " CTRL " row_mask:0xf bank_mask:" BANKS          \
                 : "=&v"(t)                                                                 \
                 : "v"(a), "v"(b));                                                         \
    return t;                                                                               \
  }
SMPC_DPP_NODE(node_bit3, "row_ror:8", "0xc")          // partner l ^ 8, b half in lanes 8..15
SMPC_DPP_NODE(node_bit2, "row_half_mirror", "0xa")    // partner l ^ 7, b half where bit 2 is set

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
  return __uint_as_float(
    (uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, false));
}

template <int K, int Rr>
__device__ __forceinline__ float lane_reduce_node(const float (&V)[64], const LaneW& w, int lane)
{
  if constexpr (K == 0) {
    return V[Rr];
  } else {
    const float a = lane_reduce_node<K - 1, Rr>(V, w, lane);
    const float b = lane_reduce_node<K - 1, Rr + (64 >> K)>(V, w, lane);
    if constexpr (K == 1) {
      // lanes 32..63 of a <-> lanes 0..31 of b; then a = {a.lo, b.lo}, b = {a.hi, b.hi}
      const u32x2 s = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b),
                                                       false, false);
      return fmaf(w.wa, __uint_as_float(s.x), w.wb * __uint_as_float(s.y));
    } else if constexpr (K == 2) {
      // odd 16-lane rows of a <-> even rows of b
      const u32x2 s = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b),
                                                       false, false);
      return __uint_as_float(s.x) + __uint_as_float(s.y);
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

__device__ __forceinline__ LaneW lane_weights(float w, int lane)
{
  const float wo = __shfl_xor(w, 32, WAVE);
  LaneW r;
  r.wa = lane < 32 ? w : wo;
  r.wb = lane < 32 ? wo : w;
  return r;
}

__device__ __forceinline__ float lane_reduce64(const float (&V)[64], const LaneW& w, int lane)
{
  return lane_reduce_node<6, 0>(V, w, lane);
}

// ---------------------------------------------------------------------------
// Costmap2D::worldToMap + getCost, window-relative (see cost_at in smpc_device_math.h
// for the guard-band argument; here the float origin is the LDS window's corner, so
// the truncated quotient is the window cell itself).
// ---------------------------------------------------------------------------
__device__ __forceinline__ int cvt_floor_i32(float q)
{
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(q));
  return r;
}

// The reference's own double arithmetic (nav2_costmap_2d worldToMap) for the lanes whose
// float quotient is near a cell edge, outside the LDS window or off the map.  Returns the
// index of the LDS byte that holds the cell's cost: a window cell, the NO_INFORMATION byte
// behind the window (off the map, obstacles_critic.cpp:209-212), or this lane's own
// scratch byte after fetching the cost from the global map.
__device__ __forceinline__ uint32_t cell_byte_exact(const SmpcDev& p, uint8_t* s_map, float x,
                                                    float y, uint32_t slot)
{
  const uint32_t nwin = (uint32_t)(p.win_w * p.win_h);
  uint32_t mx = 0, my = 0;
  bool on = cell_index_exact((double)x, p.ox, p.res, p.W, mx);
  on = cell_index_exact((double)y, p.oy, p.res, p.H, my) && on;
  if (!on) return nwin;
  const uint32_t wx = mx - (uint32_t)p.win_x0, wy = my - (uint32_t)p.win_y0;
  if (wx < (uint32_t)p.win_w && wy < (uint32_t)p.win_h) return wy * p.win_w + wx;
  s_map[nwin + 1 + slot] = p.map[(size_t)my * p.W + mx];
  return nwin + 1 + slot;
}

#endif  // SMPC_LANE_COMMON_H_
