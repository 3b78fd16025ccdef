// TEST INFRASTRUCTURE, not product code: a second stand-in for librccl.so, DEVICE-SIDE, for what the host-staged one
// (shm_collective.hip) cannot exercise: collectives INSIDE a captured hipGraph (rgcn_capture_begin on a sharded
// context, BASELINE.json configs[4]).  Several ranks run as separate processes on ONE GPU; every rank owns a mailbox and
// a pair of counters in device memory, shared with the peers through hipIpc handles.  A collective is a chain of
// kernels on the caller's stream -- no host synchronisation, so a stream capture records it like RCCL's own kernels:
//     wait until every peer has read my mailbox of the previous collective   (counter `consumed`)
//     copy my contribution into my mailbox, publish it                        (counter `posted`)
//     wait until every peer has published, combine the mailboxes in rank order (the same bits on every rank)
//     tell the peers I have read them
// Kernel boundaries carry the data (a finished kernel's stores are in memory before the flag kernel runs); the counters
// are system-scope atomics.  Waits are BOUNDED (3 s of the constant-rate wall clock): a peer that never arrives makes
// the collective fail (the next call returns an error, results are garbage) instead of hanging the box.
// Like RCCL, collectives of one communicator execute in the order they were called in, whatever streams they were
// given: each is chained behind the previous one through an event.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kMailFloats = (size_t)10 << 20;       // 40 MB per rank: [V,d] fp32 at FB15k size is 30 MB
constexpr unsigned long long kWaitTicks = 300000000ull;  // 3 s at 100 MHz

struct Counters {
  uint32_t posted, consumed, seq, pad;
};

struct Shared {                      // the rendezvous file
  std::atomic<int32_t> attached, published, opened;
  hipIpcMemHandle_t mail[kMaxRanks];
  hipIpcMemHandle_t counters[kMaxRanks];
};

struct Dev {                         // what the kernels need, by value
  float* mail[kMaxRanks];
  Counters* counters[kMaxRanks];
  int* failed;                       // host-mapped: set when a wait ran out
  int rank, nranks;
};

struct Comm {
  Dev d;
  char name[64] = {0};
  Shared* shared = nullptr;
  hipEvent_t last = nullptr;         // end of the previous collective
  unsigned long long last_capture = 0;
  bool have_last = false;
  int* failed_host = nullptr;
};

__device__ bool wait_at_least(const uint32_t* p, uint32_t want, int* failed) {
  const unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
    if (wall_clock64() - t0 > kWaitTicks) { *failed = 1; return false; }
    __builtin_amdgcn_s_sleep(32);
  }
  return true;
}

__global__ void k_wait_consumed(Dev d) {
  const uint32_t seq = d.counters[d.rank]->seq;
  if ((int)threadIdx.x < d.nranks) wait_at_least(&d.counters[threadIdx.x]->consumed, seq, d.failed);
}
__global__ void k_copy(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void k_post_and_wait(Dev d) {
  const uint32_t seq = d.counters[d.rank]->seq;
  if (threadIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store(&d.counters[d.rank]->posted, seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if ((int)threadIdx.x < d.nranks) wait_at_least(&d.counters[threadIdx.x]->posted, seq + 1, d.failed);
}
// out[i] = sum over ranks (rank order) of mail[r][off + i]
__global__ void k_reduce(Dev d, float* __restrict__ out, size_t off, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float s = d.mail[0][off + i];
    for (int r = 1; r < d.nranks; ++r) s += d.mail[r][off + i];
    out[i] = s;
  }
}
// out[r * n + i] = mail[r][i]
__global__ void k_gather(Dev d, float* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = n * (size_t)d.nranks;
  for (; i < total; i += (size_t)gridDim.x * blockDim.x) out[i] = d.mail[i / n][i % n];
}
__global__ void k_done(Dev d) {
  if (threadIdx.x == 0) {
    Counters* me = d.counters[d.rank];
    const uint32_t seq = me->seq;
    __threadfence_system();
    __hip_atomic_store(&me->consumed, seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    me->seq = seq + 1;
  }
}

bool wait_host(std::atomic<int32_t>& a, int want) {
  const time_t t0 = time(nullptr);
  while (a.load() < want) {
    if (time(nullptr) - t0 > 120) return false;
    usleep(100);
  }
  return true;
}

unsigned grid_for(size_t n) {
  const size_t b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

unsigned long long capture_id(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  if (hipStreamGetCaptureInfo(s, &st, &id) != hipSuccess || st != hipStreamCaptureStatusActive) return 0;
  return id;
}

// chain this collective behind the previous one of the communicator (same capture, or both outside any)
bool begin(Comm* c, hipStream_t s) {
  if (*c->failed_host) return false;
  const unsigned long long id = capture_id(s);
  if (c->have_last && id == c->last_capture && hipStreamWaitEvent(s, c->last, 0) != hipSuccess) return false;
  return true;
}
bool end(Comm* c, hipStream_t s) {
  hipLaunchKernelGGL(k_done, dim3(1), dim3(64), 0, s, c->d);
  if (hipGetLastError() != hipSuccess) return false;
  if (hipEventRecord(c->last, s) != hipSuccess) return false;
  c->last_capture = capture_id(s);
  c->have_last = true;
  return true;
}

}  // namespace

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId* id) {
  memset(id->internal, 0, 128);
  snprintf(id->internal, 64, "/tmp/rgcn_ipc_%d_%ld_%d", (int)getpid(), (long)time(nullptr), rand() & 0xffff);
  return 0;
}

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks > kMaxRanks) return 4;
  Comm* c = new Comm();
  c->d.rank = rank;
  c->d.nranks = nranks;
  strncpy(c->name, id.internal, 63);
  int fd = open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) { delete c; return 2; }
  if (ftruncate(fd, (off_t)sizeof(Shared)) != 0) { close(fd); delete c; return 2; }
  void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { delete c; return 2; }
  c->shared = static_cast<Shared*>(p);
  float* mail = nullptr;
  Counters* ctr = nullptr;
  if (hipMalloc((void**)&mail, kMailFloats * sizeof(float)) != hipSuccess) return 1;
  if (hipMalloc((void**)&ctr, sizeof(Counters)) != hipSuccess) return 1;
  if (hipMemset(ctr, 0, sizeof(Counters)) != hipSuccess) return 1;
  if (hipHostMalloc((void**)&c->failed_host, sizeof(int), hipHostMallocMapped) != hipSuccess) return 1;
  *c->failed_host = 0;
  c->d.failed = c->failed_host;
  if (hipIpcGetMemHandle(&c->shared->mail[rank], mail) != hipSuccess) return 1;
  if (hipIpcGetMemHandle(&c->shared->counters[rank], ctr) != hipSuccess) return 1;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  c->shared->published.fetch_add(1);
  if (!wait_host(c->shared->published, nranks)) return 3;
  for (int r = 0; r < nranks; ++r) {
    if (r == rank) { c->d.mail[r] = mail; c->d.counters[r] = ctr; continue; }
    if (hipIpcOpenMemHandle((void**)&c->d.mail[r], c->shared->mail[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) return 1;
    if (hipIpcOpenMemHandle((void**)&c->d.counters[r], c->shared->counters[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) return 1;
  }
  c->shared->opened.fetch_add(1);
  if (!wait_host(c->shared->opened, nranks)) return 3;
  if (hipEventCreateWithFlags(&c->last, hipEventDisableTiming) != hipSuccess) return 1;
  *comm = c;
  return 0;
}

int ncclCommDestroy(void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return 0;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < c->d.nranks; ++r) {
    if (r == c->d.rank) continue;
    if (c->d.mail[r]) (void)hipIpcCloseMemHandle(c->d.mail[r]);
    if (c->d.counters[r]) (void)hipIpcCloseMemHandle(c->d.counters[r]);
  }
  // the peers may still have this rank's buffers open: the owner's allocations are left to process exit
  if (c->last) (void)hipEventDestroy(c->last);
  if (c->shared) munmap(c->shared, sizeof(Shared));
  unlink(c->name);
  delete c;
  return 0;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(comm);
  if (dtype != 7 || op != 0 || count > kMailFloats) return 4;        // float32 sum only
  if (!begin(c, stream)) return 3;
  hipLaunchKernelGGL(k_wait_consumed, dim3(1), dim3(64), 0, stream, c->d);
  hipLaunchKernelGGL(k_copy, dim3(grid_for(count)), dim3(256), 0, stream, static_cast<const float*>(send),
                     c->d.mail[c->d.rank], count);
  hipLaunchKernelGGL(k_post_and_wait, dim3(1), dim3(64), 0, stream, c->d);
  hipLaunchKernelGGL(k_reduce, dim3(grid_for(count)), dim3(256), 0, stream, c->d, static_cast<float*>(recv), (size_t)0, count);
  return end(c, stream) ? 0 : 1;
}

// recv (count floats) = sum over ranks of their send[rank * count, +count)
int ncclReduceScatter(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(comm);
  const size_t total = count * (size_t)c->d.nranks;
  if (dtype != 7 || op != 0 || total > kMailFloats) return 4;
  if (!begin(c, stream)) return 3;
  hipLaunchKernelGGL(k_wait_consumed, dim3(1), dim3(64), 0, stream, c->d);
  hipLaunchKernelGGL(k_copy, dim3(grid_for(total)), dim3(256), 0, stream, static_cast<const float*>(send),
                     c->d.mail[c->d.rank], total);
  hipLaunchKernelGGL(k_post_and_wait, dim3(1), dim3(64), 0, stream, c->d);
  hipLaunchKernelGGL(k_reduce, dim3(grid_for(count)), dim3(256), 0, stream, c->d, static_cast<float*>(recv),
                     (size_t)c->d.rank * count, count);
  return end(c, stream) ? 0 : 1;
}

// recv (nranks * count floats) = the ranks' send buffers (count floats each) in rank order
int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(comm);
  if (dtype != 7 || count > kMailFloats) return 4;
  if (!begin(c, stream)) return 3;
  hipLaunchKernelGGL(k_wait_consumed, dim3(1), dim3(64), 0, stream, c->d);
  hipLaunchKernelGGL(k_copy, dim3(grid_for(count)), dim3(256), 0, stream, static_cast<const float*>(send),
                     c->d.mail[c->d.rank], count);
  hipLaunchKernelGGL(k_post_and_wait, dim3(1), dim3(64), 0, stream, c->d);
  hipLaunchKernelGGL(k_gather, dim3(grid_for(count * (size_t)c->d.nranks)), dim3(256), 0, stream, c->d,
                     static_cast<float*>(recv), count);
  return end(c, stream) ? 0 : 1;
}

const char* ncclGetErrorString(int code) {
  switch (code) {
    case 0: return "success";
    case 1: return "hip error (device-side test collective)";
    case 2: return "shared memory error (device-side test collective)";
    case 3: return "a peer never arrived (device-side test collective)";
    case 4: return "unsupported argument (device-side test collective: float32 sum, <= 40 MB, <= 8 ranks)";
    default: return "unknown";
  }
}

}  // extern "C"
