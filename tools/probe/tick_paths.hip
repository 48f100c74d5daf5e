// tick_paths.hip — what a tick's FIXED cost is made of, measured with trivial kernels
// (developer aid; build: hipcc -O2 --offload-arch=gfx950 -o tick_paths tick_paths.hip).
//
// A tick is: host hands over ~4 KB, a scoring kernel reads it, a reduction kernel publishes the
// result to host-mapped memory, the host polls.  Variants of the hand-over and of the submission:
//   copy     hipMemcpyAsync(pinned -> device) + 2 launches              (round 2's tick)
//   bar      CPU stores straight into device memory (large BAR) + 2 launches
//   graph    hipGraphLaunch of {memcpy, k1, k2}
//   gbar     CPU stores + hipGraphLaunch of {k1, k2}
//   one      CPU stores + ONE launch that also publishes (the floor)
// Every variant checks that the kernel saw THIS tick's bytes (a per-tick pattern), so a stale
// read through the BAR path would show as a mismatch count.
#include <hip/hip_runtime.h>
#include <emmintrin.h>
#include <setjmp.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

constexpr int kWords = 1024;   // 4 KB tick block

// "scoring": every block sums the tick block, block 0 leaves the sum in device memory
__global__ void k_score(const uint32_t* __restrict__ tick, uint32_t* __restrict__ partial, int spin)
{
  uint32_t s = 0;
  for (int i = threadIdx.x; i < kWords; i += blockDim.x) s += tick[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __shared__ uint32_t sh[16];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (unsigned w = 0; w < blockDim.x / 64; ++w) t += sh[w];
    partial[blockIdx.x] = t;
  }
  // optional busy time (shader clocks) to stand in for a real pass
  if (spin > 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}
  }
}

// "reduction": one block publishes partial[0] and the sequence number to host-mapped memory
__global__ void k_publish(const uint32_t* __restrict__ partial, uint32_t* __restrict__ host_out, uint32_t seq)
{
  if (threadIdx.x == 0) {
    host_out[0] = partial[0];
    __threadfence_system();
    __hip_atomic_store(host_out + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// the same result as {value, sequence number} in ONE 8-byte store: no fence, no separate flag
__global__ void k_publish_pair(const uint32_t* __restrict__ partial, unsigned long long* __restrict__ host_out, uint32_t seq)
{
  if (threadIdx.x == 0) host_out[1] = ((unsigned long long)seq << 32) | partial[0];
}

// both in one launch: block 0 publishes its own sum (the floor of a single submission)
__global__ void k_one(const uint32_t* __restrict__ tick, uint32_t* __restrict__ host_out, uint32_t seq)
{
  uint32_t s = 0;
  for (int i = threadIdx.x; i < kWords; i += blockDim.x) s += tick[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __shared__ uint32_t sh[16];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    uint32_t t = 0;
    for (unsigned w = 0; w < blockDim.x / 64; ++w) t += sh[w];
    host_out[0] = t;
    __threadfence_system();
    __hip_atomic_store(host_out + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static sigjmp_buf g_jmp;
static void on_segv(int) {siglongjmp(g_jmp, 1);}

// can the CPU store to this pointer?
static bool cpu_can_write(volatile uint32_t* p)
{
  struct sigaction sa{}, old_segv{}, old_bus{};
  sa.sa_handler = on_segv;
  sigemptyset(&sa.sa_mask);
  sigaction(SIGSEGV, &sa, &old_segv);
  sigaction(SIGBUS, &sa, &old_bus);
  bool ok = false;
  if (sigsetjmp(g_jmp, 1) == 0) {
    p[0] = 0x12345678u;
    _mm_sfence();
    ok = true;
  }
  sigaction(SIGSEGV, &old_segv, nullptr);
  sigaction(SIGBUS, &old_bus, nullptr);
  return ok;
}

static double now_us()
{
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static bool wait_seq(volatile uint32_t* host_out, uint32_t seq)
{
  const double t0 = now_us();
  while (host_out[1] != seq) {
    __builtin_ia32_pause();
    if (now_us() - t0 > 2.0e6) return false;
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return true;
}

struct Stat {double med, p10, p90; int bad;};
static Stat summarize(std::vector<double>& v, int bad)
{
  std::sort(v.begin(), v.end());
  return {v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10], bad};
}

int main(int argc, char** argv)
{
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  const int grid = argc > 2 ? atoi(argv[2]) : 256;
  const int spin = argc > 3 ? atoi(argv[3]) : 0;
  CK(hipSetDevice(0));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  uint32_t *h_tick, *d_tick, *d_tick_fg = nullptr, *d_tick_uc = nullptr, *d_partial, *h_out, *h_out_dev;
  CK(hipHostMalloc(&h_tick, kWords * 4, hipHostMallocDefault));
  CK(hipMalloc(&d_tick, kWords * 4));
  if (hipExtMallocWithFlags(reinterpret_cast<void**>(&d_tick_fg), kWords * 4, hipDeviceMallocFinegrained) != hipSuccess) d_tick_fg = nullptr;
  if (hipExtMallocWithFlags(reinterpret_cast<void**>(&d_tick_uc), kWords * 4, hipDeviceMallocUncached) != hipSuccess) d_tick_uc = nullptr;
  CK(hipMalloc(&d_partial, 4096 * 4));
  CK(hipHostMalloc(&h_out, 64, hipHostMallocMapped));
  memset(h_out, 0, 64);
  CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&h_out_dev), h_out, 0));
  CK(hipMemset(d_tick, 0, kWords * 4));
  CK(hipDeviceSynchronize());

  const bool bar_plain = cpu_can_write(d_tick);
  const bool bar_fg = d_tick_fg && cpu_can_write(d_tick_fg);
  const bool bar_uc = d_tick_uc && cpu_can_write(d_tick_uc);
  printf("CPU stores into device memory: hipMalloc %s, fine-grained %s, uncached %s\n", bar_plain ? "yes" : "NO",
         d_tick_fg ? (bar_fg ? "yes" : "NO") : "n/a", d_tick_uc ? (bar_uc ? "yes" : "NO") : "n/a");
  int v = 0;
  (void)hipDeviceGetAttribute(&v, hipDeviceAttributeHostNativeAtomicSupported, 0);
  printf("hipDeviceAttributeHostNativeAtomicSupported %d\n", v);

  uint32_t seq = 0;
  auto fill = [&](uint32_t* dst, uint32_t k) -> uint32_t {
    uint32_t sum = 0;
    for (int i = 0; i < kWords; ++i) {
      const uint32_t w = k * 2654435761u + (uint32_t)i * 40503u;
      dst[i] = w;
      sum += w;
    }
    return sum;
  };
  // CPU stores into device memory, non-temporal 16-byte stores + sfence
  auto fill_bar = [&](uint32_t* dst, uint32_t k) -> uint32_t {
    const uint32_t sum = fill(h_tick, k);
    const __m128i* src = reinterpret_cast<const __m128i*>(h_tick);
    __m128i* d = reinterpret_cast<__m128i*>(dst);
    for (int i = 0; i < kWords / 4; ++i) _mm_stream_si128(d + i, _mm_load_si128(src + i));
    _mm_sfence();
    return sum;
  };

  // warm the clocks a little
  for (int i = 0; i < 200; ++i) {
    hipLaunchKernelGGL(k_score, dim3(grid), dim3(256), 0, st, d_tick, d_partial, 20000);
  }
  CK(hipStreamSynchronize(st));

  auto run = [&](const char* name, auto&& tick_fn) {
    std::vector<double> t;
    int bad = 0;
    for (int k = 0; k < iters + 50; ++k) {
      const double t0 = now_us();
      uint32_t want = 0;
      const bool ok = tick_fn((uint32_t)k + 1u, want);
      const double t1 = now_us();
      if (!ok) {
        printf("%-28s TIMEOUT at tick %d\n", name, k);
        return;
      }
      if (h_out[0] != want) bad++;
      if (k >= 50) t.push_back(t1 - t0);
    }
    Stat s = summarize(t, bad);
    printf("%-28s median %7.2f us  p10 %7.2f  p90 %7.2f   stale/mismatched ticks %d of %d\n", name, s.med, s.p10, s.p90,
           s.bad, iters + 50);
    fflush(stdout);
  };

  run("copy + 2 launches", [&](uint32_t k, uint32_t& want) {
    want = fill(h_tick, k);
    CK(hipMemcpyAsync(d_tick, h_tick, kWords * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_score, dim3(grid), dim3(256), 0, st, d_tick, d_partial, spin);
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, d_partial, h_out_dev, ++seq);
    return wait_seq(h_out, seq);
  });
  run("copy + 1 launch", [&](uint32_t k, uint32_t& want) {
    want = fill(h_tick, k);
    CK(hipMemcpyAsync(d_tick, h_tick, kWords * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_one, dim3(grid), dim3(256), 0, st, d_tick, h_out_dev, ++seq);
    return wait_seq(h_out, seq);
  });
  run("pinned host read + 2 launches", [&](uint32_t k, uint32_t& want) {
    want = fill(h_tick, k);
    hipLaunchKernelGGL(k_score, dim3(grid), dim3(256), 0, st, h_tick, d_partial, spin);
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, d_partial, h_out_dev, ++seq);
    return wait_seq(h_out, seq);
  });
  struct {const char* n; uint32_t* p; bool ok;} bars[3] = {{"hipMalloc", d_tick, bar_plain}, {"fine-grained", d_tick_fg, bar_fg},
                                                         {"uncached", d_tick_uc, bar_uc}};
  for (auto& b : bars) {
    if (!b.ok) continue;
    char nm[64];
    snprintf(nm, sizeof(nm), "BAR(%s) + 2 launches", b.n);
    run(nm, [&](uint32_t k, uint32_t& want) {
      want = fill_bar(b.p, k);
      hipLaunchKernelGGL(k_score, dim3(grid), dim3(256), 0, st, b.p, d_partial, spin);
      hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, d_partial, h_out_dev, ++seq);
      return wait_seq(h_out, seq);
    });
    snprintf(nm, sizeof(nm), "BAR(%s) + 1 launch", b.n);
    run(nm, [&](uint32_t k, uint32_t& want) {
      want = fill_bar(b.p, k);
      hipLaunchKernelGGL(k_one, dim3(grid), dim3(256), 0, st, b.p, h_out_dev, ++seq);
      return wait_seq(h_out, seq);
    });
  }
  if (bar_plain) {
    run("BAR + 2 launches, pair store", [&](uint32_t k, uint32_t& want) {
      want = fill_bar(d_tick, k);
      hipLaunchKernelGGL(k_score, dim3(grid), dim3(256), 0, st, d_tick, d_partial, spin);
      hipLaunchKernelGGL(k_publish_pair, dim3(1), dim3(64), 0, st, d_partial, reinterpret_cast<unsigned long long*>(h_out_dev), ++seq);
      volatile unsigned long long* hp = reinterpret_cast<volatile unsigned long long*>(h_out);
      const double t0 = now_us();
      while ((uint32_t)(hp[1] >> 32) != seq) {
        __builtin_ia32_pause();
        if (now_us() - t0 > 2.0e6) return false;
      }
      h_out[0] = (uint32_t)hp[1];   // (where the checker looks)
      return true;
    });
  }
  // how long the CPU stores themselves take
  for (auto& b : bars) {
    if (!b.ok) continue;
    std::vector<double> t;
    for (int k = 0; k < 2000; ++k) {
      fill(h_tick, k);
      const double t0 = now_us();
      const __m128i* src = reinterpret_cast<const __m128i*>(h_tick);
      __m128i* d = reinterpret_cast<__m128i*>(b.p);
      for (int i = 0; i < kWords / 4; ++i) _mm_stream_si128(d + i, _mm_load_si128(src + i));
      _mm_sfence();
      t.push_back(now_us() - t0);
    }
    Stat s = summarize(t, 0);
    printf("4 KB of CPU stores into %-14s median %6.2f us  p90 %6.2f\n", b.n, s.med, s.p90);
  }

  // ---- graphs: the tick's sequence number travels in the tick block's last word ----------
  // (kernel arguments of a graph are frozen at instantiation; k_publish_g reads seq from device memory)
  {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    CK(hipMemcpyAsync(d_tick, h_tick, kWords * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_score, dim3(grid), dim3(256), 0, st, d_tick, d_partial, spin);
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, d_partial, h_out_dev, 0xfffffff0u);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    // the frozen seq means the completion word does not change between ticks: clear it on the host
    run("graph{copy, k1, k2}", [&](uint32_t k, uint32_t& want) {
      want = fill(h_tick, k);
      reinterpret_cast<volatile uint32_t*>(h_out)[1] = 0;
      CK(hipGraphLaunch(ge, st));
      return wait_seq(h_out, 0xfffffff0u);
    });
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  for (auto& b : bars) {
    if (!b.ok) continue;
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k_score, dim3(grid), dim3(256), 0, st, b.p, d_partial, spin);
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, d_partial, h_out_dev, 0xfffffff1u);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    char nm[64];
    snprintf(nm, sizeof(nm), "BAR(%s) + graph{k1, k2}", b.n);
    run(nm, [&](uint32_t k, uint32_t& want) {
      want = fill_bar(b.p, k);
      reinterpret_cast<volatile uint32_t*>(h_out)[1] = 0;
      CK(hipGraphLaunch(ge, st));
      return wait_seq(h_out, 0xfffffff1u);
    });
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    break;   // one is enough
  }
  CK(hipStreamSynchronize(st));
  printf("done\n");
  return 0;
}
