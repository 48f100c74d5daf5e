// fake_rccl.cpp — TEST INFRASTRUCTURE ONLY.  A stand-in for the six RCCL entry points libsmpc
// resolves at run time (mpcholonavigation_amd/csrc/smpc_shard.cpp: ncclGetUniqueId,
// ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclAllReduce, ncclGetErrorString), so that
// smpc_shard_tick's world > 1 control flow runs with several PROCESSES ON ONE GPU — RCCL itself
// refuses two ranks on one device.  Loaded only when a test points SMPC_RCCL_LIB at it.
//
// What it is: host-staged collectives over a POSIX shared-memory segment named by the unique id.
// A collective drains the caller's stream, copies its contribution to the segment, publishes a
// per-rank sequence number, waits (bounded) for every rank's, and copies the result back to the
// device.  What it proves: the protocol of the sharded tick (which exchanges happen, in which
// order, on which data, and that every rank takes the same branches).  What it cannot: RCCL
// itself, stream-ordered collectives, xGMI, timing.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <thread>

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kSlotBytes = 64 * 1024;   // per rank and parity: far above a shard tuple
constexpr double kTimeoutS = 30.0;

struct Segment {
  std::atomic<uint32_t> arrived[kMaxRanks];   // collectives this rank has contributed to
  std::atomic<uint32_t> left[kMaxRanks];      // collectives this rank has finished reading
  unsigned char slot[2][kMaxRanks][kSlotBytes];
};

struct Comm {
  Segment* seg = nullptr;
  int rank = 0, world = 0;
  uint32_t n = 0;            // collectives so far
  char name[64] = {0};
  unsigned char* stage = nullptr;   // pinned
};

struct UniqueId {char internal[128];};

bool wait_all(const std::atomic<uint32_t>* words, int world, uint32_t want)
{
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    bool ok = true;
    for (int r = 0; r < world; ++r)
      if (words[r].load(std::memory_order_acquire) < want) ok = false;
    if (ok) return true;
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutS) return false;
    std::this_thread::yield();
  }
}

// contribute `bytes` from device memory, return when every rank's contribution of this collective
// is in the segment; *parity tells which half holds them
int exchange(Comm* c, const void* send, size_t bytes, hipStream_t st, int* parity)
{
  if (bytes > kSlotBytes) return 4;   // ncclInvalidArgument
  if (hipStreamSynchronize(st) != hipSuccess) return 1;
  const uint32_t n = ++c->n;
  const int par = static_cast<int>(n & 1u);
  // a rank may overwrite a half only after every rank has read the collective two back
  if (n > 2 && !wait_all(c->seg->left, c->world, n - 2)) return 2;
  if (hipMemcpy(c->seg->slot[par][c->rank], send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  c->seg->arrived[c->rank].store(n, std::memory_order_release);
  if (!wait_all(c->seg->arrived, c->world, n)) return 2;   // ncclSystemError: a peer never came
  *parity = par;
  return 0;
}

void done_reading(Comm* c) {c->seg->left[c->rank].store(c->n, std::memory_order_release);}

}  // namespace

extern "C" {

int ncclGetUniqueId(UniqueId* id)
{
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/smpc_fake_rccl_%d_%lld", static_cast<int>(getpid()),
           static_cast<long long>(std::chrono::steady_clock::now().time_since_epoch().count()));
  return 0;
}

int ncclCommInitRank(Comm** out, int world, UniqueId id, int rank)
{
  if (!out || world < 1 || world > kMaxRanks || rank < 0 || rank >= world) return 4;
  Comm* c = new (std::nothrow) Comm();
  if (!c) return 2;
  c->rank = rank;
  c->world = world;
  snprintf(c->name, sizeof(c->name), "%s", id.internal);
  // (O_CREAT by whoever comes first; ftruncate to the same size by everyone is harmless; a fresh
  // segment is zero-filled: no rank has arrived at any collective)
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(Segment)) != 0) {
    delete c;
    return 2;
  }
  void* p = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    delete c;
    return 2;
  }
  c->seg = static_cast<Segment*>(p);
  *out = c;
  return 0;
}

int ncclCommDestroy(Comm* c)
{
  if (!c) return 0;
  if (c->seg) munmap(c->seg, sizeof(Segment));
  if (c->rank == 0) shm_unlink(c->name);
  delete c;
  return 0;
}

// datatype 7 = ncclFloat32 (the only one libsmpc uses)
int ncclAllGather(const void* send, void* recv, size_t count, int datatype, Comm* c, hipStream_t st)
{
  if (!c || datatype != 7) return 4;
  const size_t bytes = count * 4;
  int par = 0;
  const int e = exchange(c, send, bytes, st, &par);
  if (e) return e;
  for (int r = 0; r < c->world; ++r)
    if (hipMemcpy(static_cast<unsigned char*>(recv) + r * bytes, c->seg->slot[par][r], bytes, hipMemcpyHostToDevice) !=
        hipSuccess)
      return 1;
  done_reading(c);
  return 0;
}

// op: 0 sum, 2 max (ncclRedOp_t); float32 only
int ncclAllReduce(const void* send, void* recv, size_t count, int datatype, int op, Comm* c, hipStream_t st)
{
  if (!c || datatype != 7 || (op != 0 && op != 2) || count * 4 > 4096) return 4;
  int par = 0;
  const int e = exchange(c, send, count * 4, st, &par);
  if (e) return e;
  float acc[1024];
  for (size_t i = 0; i < count; ++i) {
    float v = reinterpret_cast<const float*>(c->seg->slot[par][0])[i];
    for (int r = 1; r < c->world; ++r) {
      const float w = reinterpret_cast<const float*>(c->seg->slot[par][r])[i];
      v = op == 0 ? v + w : (w > v ? w : v);
    }
    acc[i] = v;
  }
  if (hipMemcpy(recv, acc, count * 4, hipMemcpyHostToDevice) != hipSuccess) return 1;
  done_reading(c);
  return 0;
}

const char* ncclGetErrorString(int e)
{
  switch (e) {
    case 0: return "no error";
    case 1: return "fake RCCL: HIP call failed";
    case 2: return "fake RCCL: a peer did not arrive within 30 s (or shared memory failed)";
    default: return "fake RCCL: invalid argument";
  }
}

}  // extern "C"
