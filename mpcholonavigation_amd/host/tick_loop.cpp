// tick_loop.cpp — n consecutive closed-loop ticks issued by a compiled caller (include/smpc_host.h).
//
// The reference's caller is C++: controller.cpp:80-116 calls Optimizer::evalControl, which runs
// optimize() and then shifts the control sequence (optimizer.cpp:134-164, 206-225), tick after
// tick, with no interpreter between two ticks.  This is that loop over the C-ABI, for callers
// (bench.py, soak tools) that would otherwise put a Python frame, a ctypes call and two numpy
// allocations (~4 us) between the end of one tick and the start of the next.
#include <cstring>

#include "../../include/smpc_host.h"

extern "C" int sortham_run_ticks(
  smpc_ctx * ctx, const smpc_tick_in * in, float * u, uint32_t time_steps, uint32_t n,
  uint32_t flags, smpc_tick_out * outs, uint32_t * done)
{
  if (done) {
    *done = 0;
  }
  if (!ctx || !in || !u || !outs || time_steps == 0) {
    return SMPC_ERR_INVALID;
  }
  const bool sharded = (flags & (SORTHAM_TICKS_SHARD | SORTHAM_TICKS_SHARD_SPECULATE)) != 0;
  const int speculate = (flags & SORTHAM_TICKS_SHARD_SPECULATE) ? 1 : 0;
  for (uint32_t k = 0; k < n; ++k) {
    const int rc = sharded ? smpc_shard_tick(ctx, in, u, &outs[k], speculate)
                           : smpc_optimize(ctx, in, u, &outs[k]);
    if (rc != SMPC_OK) {
      return rc;
    }
    if (flags & SORTHAM_TICKS_REDRAW_ASYNC) {
      // regenerate_noises: the next epoch is drawn behind this tick (noise_generator.cpp:54-63)
      const int rr = smpc_redraw_noise_async(ctx);
      if (rr != SMPC_OK) {
        return rr;
      }
    }
    if (flags & SORTHAM_TICKS_SHIFT) {
      // Optimizer::shiftControlSequence (optimizer.cpp:206-225): one step to the left, the last
      // control kept
      for (int r = 0; r < 3; ++r) {
        float * row = u + static_cast<size_t>(r) * time_steps;
        std::memmove(row, row + 1, (time_steps - 1) * sizeof(float));
      }
    }
    if (done) {
      *done = k + 1;
    }
  }
  return SMPC_OK;
}
