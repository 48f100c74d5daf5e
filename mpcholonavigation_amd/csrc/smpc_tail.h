// smpc_grid_tail: the reduction of a scoring launch's per-block partials inside that launch, by
// the blocks that finish last, instead of by a second launch (smpc_reduce_partials).
//
// AN EXPERIMENT, OFF BY DEFAULT (SMPC_FUSED_REDUCE=1 when the context is created).  The second
// launch costs ~9 us per tick and the idea was to save most of that.  Measured (MI355X, same
// process, interleaved: tools/tail_ab.py; stage by stage: tools/tail_timeline.py): the tail
// takes 10-11 us behind the grid's last block, so a tick is 2 us (65 536 x 64) to 9 us
// (2 000 x 56) SLOWER with it.  Where it goes: the partials were written by other XCDs, so
// they have to be read past this XCD's L2 — 6.5-7.5 us for 200 KB even split over four blocks,
// against ~2 us for ordinary loads behind a kernel boundary — and the agent-scope ticket,
// the wait for the last block and the host fence add ~1 us each.  A grid-wide step inside a
// kernel pays the cross-XCD coherence price that the kernel boundary pays once for everything.
// Kept, with its test, so that the measurement can be repeated.
//
// The arithmetic — which partial is added to which accumulator in which order — is that of
// reduce_partials_body (smpc_kernels.hip) exactly, so a tick gives the same bits whichever of
// the two reduces it (tests/test_gpu_properties.py::test_fused_reduction_matches_the_launch):
//   weights:  a_g = sc_g * w_g, butterfly sum over each run of 64 blocks, runs added in order
//   columns:  slice s (0..31) adds rows s, s + 32, ...; the 32 slices are added in order
// Softmax over all rollouts, control update and clipping: optimizer.cpp:382-393, 237-249.
#ifndef SMPC_TAIL_H_
#define SMPC_TAIL_H_

#include <type_traits>

#include "smpc_dev.h"

// how a scoring pass writes its block's partial (see the protocol of smpc_grid_tail)
__device__ __forceinline__ void smpc_store_partial(float* a, float v)
{
  __hip_atomic_store(a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// NW: waves per block (4, 8 or 16).  smem: the launch's dynamic LDS from offset 0, at least
// smpc_tail_lds_bytes(T); every thread of the block calls this after its part of the block's
// partial is written (smpc_store_partial).
//
// Protocol.  No cache-wide fence: an agent-scope release / acquire pair writes back and
// invalidates the XCD's whole L2 — in every block, under the other blocks' noise stream
// (measured: +38 us on the 2 097 152-rollout pass).  Instead the partials are written with
// agent-scope atomic stores (write-through to where the other XCDs see them), each thread waits
// for its own stores to complete, the block takes a ticket with an agent-scope atomic, and the
// reducers read with agent-scope atomic loads (they do not hit their L2's stale lines).
// One CU cannot pull 200 KB of uncached lines quickly (measured: +17 us with a single
// reducing block), so the LAST `nred` blocks to finish share the tuple's column groups: ticket
// nblk - nred + r waits until the counter reads nblk, then reduces groups r, r + nred, ...
// A waiting reducer never blocks the blocks it waits for: by then nblk - nred blocks have
// left the machine.  The last reducer to finish resets both counters and, on a single-GPU
// tick, publishes the completion word.
template <int NW>
__device__ __forceinline__ void smpc_grid_tail(const SmpcDev& p, unsigned char* smem)
{
  static_assert(NW == 4 || NW == 8 || NW == 16, "block of 256, 512 or 1024 threads");
  constexpr int SPT = 32 / NW;                 // column slices per thread
  constexpr int VPT = NW >= 8 ? 1 : 8 / NW;    // header rows per thread
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t nblk = gridDim.x, T = p.T, TL = 4 + 3 * T;
  float* s_sc = reinterpret_cast<float*>(smem);   // [512]
  float* s_red = s_sc + 512;                      // [4][16]: min, furthest, non-colliding, weights
  float* s_acc = s_red + 64;                      // [column group][32][64]
  const uint32_t ncg = (TL + 63) / 64;
  uint32_t* s_flag = reinterpret_cast<uint32_t*>(s_acc + (size_t)ncg * 2048);
  const uint32_t nred = ncg < nblk ? ncg : nblk;
  uint32_t* tickets = p.tail_counter;
  uint32_t* finished = p.tail_counter + 1;
  // developer aid (SMPC_LANE_TIMELINE=1, tools/tail_timeline.py): s_memrealtime stamps (10 ns) of
  // thread 0 of every reducing block, 16 per reducer behind the lane pass's own stamps
  unsigned long long st[9];
  const bool stamps = p.timeline != nullptr;
#define SMPC_TAIL_STAMP(k) \
  do { \
    if (__builtin_expect(stamps, 0)) st[k] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
  SMPC_TAIL_STAMP(0);

  // ---- take a ticket; the last nred blocks go on ---------------------------------------------
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  SMPC_TAIL_STAMP(1);
  if (tid == 0) {
    const uint32_t ticket = __hip_atomic_fetch_add(tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    SMPC_TAIL_STAMP(2);
    uint32_t late = 0u;
    if (ticket + nred >= nblk) {
      // Read the counter with a read-modify-write: it is performed where the other XCDs'
      // increments are, while an atomic LOAD may be served from this XCD's L2 over and over
      // (the first version of this loop never came back).  Bounded all the same: two seconds
      // of s_memrealtime (100 MHz), then the reducer goes on and marks the tick as failed.
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (atomicCAS(tickets, 0xffffffffu, 0u) < nblk) {   // (never matches: a read at the atomic unit)
        __builtin_amdgcn_s_sleep(2);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
          late = 0x80000000u;
          break;
        }
      }
    }
    SMPC_TAIL_STAMP(3);
    *s_flag = ticket | late;
  }
  __syncthreads();
  const bool late = (*s_flag & 0x80000000u) != 0u;
  const uint32_t ticket = *s_flag & 0x7fffffffu;
  if (ticket + nred < nblk) return;
  const uint32_t role = ticket + nred - nblk;

  float* partials = p.partials;
  auto ld = [](float* a) {return __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);};
  // Every load is issued before the first wait — the headers {min, sum w, furthest,
  // non-colliding} of every partial and the rows of this reducer's (first) column group: these
  // are uncached reads, ~2 us a round trip, and the chain behind the grid's last block is
  // nothing but round trips.
  constexpr int K = 16;
  float h0[VPT], h1[VPT], h2[VPT], h3[VPT];
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const uint32_t g = tid + j * NW * 64;
    const bool on = g < nblk && g < 512u;
    float* h = partials + (size_t)(on ? g : 0) * TL;
    h0[j] = ld(h);
    h1[j] = ld(h + 1);
    h2[j] = ld(h + 2);
    h3[j] = ld(h + 3);
  }
  float v[SPT][K];
  auto load_rows = [&](uint32_t cg) {
    const uint32_t col = cg * 64 + lane;
    const bool colon = col < TL && col != 0 && col != 2 && col != 3;
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const uint32_t g = (wave + j * NW) + 32 * k;
        v[j][k] = 0.f;
        if (k < 8 || nblk > 256u) {
          if (colon && g < nblk) v[j][k] = ld(partials + (size_t)g * TL + col);
        }
      }
    }
  };
  load_rows(role);
  float hm[VPT], hw[VPT], m_ = 3.0e38f;
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const uint32_t g = tid + j * NW * 64;
    const bool on = g < nblk && g < 512u;
    hm[j] = on ? h0[j] : 3.0e38f;
    hw[j] = on ? h1[j] : 0.f;
    float m = hm[j], fu = on ? h2[j] : 0.f, nc = on ? h3[j] : 0.f;
    for (int o = 32; o > 0; o >>= 1) {
      m = fminf(m, __shfl_xor(m, o, 64));
      fu = fmaxf(fu, __shfl_xor(fu, o, 64));
      nc += __shfl_xor(nc, o, 64);
    }
    const int vw = wave + j * NW;
    if (lane == 0 && vw < 8) {
      s_red[vw] = m;
      s_red[16 + vw] = fu;
      s_red[32 + vw] = nc;
    }
  }
  __syncthreads();
  SMPC_TAIL_STAMP(4);
  float fu_ = 0.f, nc_ = 0.f;
  for (int w = 0; w < 8; ++w) {
    m_ = fminf(m_, s_red[w]);
    fu_ = fmaxf(fu_, s_red[16 + w]);
    nc_ += s_red[32 + w];
  }
  // per-block rescale factors exp(-(m_g - m)/temperature); sum of weights
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const uint32_t g = tid + j * NW * 64;
    float a = 0.f;
    if (g < nblk && g < 512u) {
      const float sc = expf(p.neg_inv_temp * (hm[j] - m_));
      s_sc[g] = sc;
      a = sc * hw[j];
    }
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    const int vw = wave + j * NW;
    if (lane == 0 && vw < 8) s_red[48 + vw] = a;
  }
  __syncthreads();
  float sw = 0.f;
  for (int w = 0; w < 8; ++w) sw += s_red[48 + w];

  // ---- this reducer's column groups: thread = (column lane, SPT slices)
  for (uint32_t cg = role; cg < ncg; cg += nred) {
    if (cg != role) load_rows(cg);   // (grids of fewer blocks than column groups)
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const uint32_t g = (wave + j * NW) + 32 * k;
        if (g < nblk) a += s_sc[g] * v[j][k];
      }
      s_acc[(size_t)cg * 2048 + (wave + j * NW) * 64 + lane] = a;
    }
  }
  __syncthreads();
  SMPC_TAIL_STAMP(5);

  // ---- finish: wave w adds the 32 slices of the w-th, (w + NW)-th, ... of its groups ----------
  const SmpcFinal& fin = p.fin;
  for (uint32_t cg = role + wave * nred; cg < ncg; cg += NW * nred) {
    const uint32_t col = cg * 64 + lane;
    if (col >= TL) continue;
    float r = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) r += s_acc[(size_t)cg * 2048 + s * 64 + lane];
    if (col == 0) r = m_;
    if (col == 1) r = sw;
    if (col == 2) r = fu_;
    if (col == 3) r = nc_;
    p.tuple[col] = r;
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
        const float used = fin.furthest_used ? *fin.furthest_used : fu_;
        float* outs[2] = {fin.u_dev + 3 * T, fin.u_host ? fin.u_host + 3 * T : nullptr};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          float* o = outs[k];
          if (!o) continue;
          o[0] = m_;
          o[1] = sw;
          o[2] = fu_;
          o[3] = nc_;
          o[4] = used;
        }
      }
    }
  }
  // ---- every reducer publishes its own completion word (fin.u_host[3T + 8 + role]) behind a
  // system-scope fence over its host stores: the host waits for all nred of them, and no
  // counter sits between the last store and the host.  The reducer that finishes last resets
  // the counters for the next launch, off the critical path.
  const bool publish = fin.enabled && fin.done_counter;
  if (late && tid == 0 && fin.enabled && fin.u_host) fin.u_host[3 * T + 6] = 2.0f;   // fetch_out fails the tick
  SMPC_TAIL_STAMP(6);
  if (publish) __threadfence_system();
  __syncthreads();
  SMPC_TAIL_STAMP(7);
  if (tid == 0) {
    if (publish)
      __hip_atomic_store(reinterpret_cast<uint32_t*>(fin.u_host + 3 * T + 8 + role), fin.seq, __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    SMPC_TAIL_STAMP(8);
    if (stamps) {
      for (int k = 0; k < 9; ++k) p.timeline[SMPC_TAIL_STAMPS_AT + role * 16 + k] = st[k];
    }
    const uint32_t prev = __hip_atomic_fetch_add(finished, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1u == nred) {
      __hip_atomic_store(finished, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

#undef SMPC_TAIL_STAMP

#endif
