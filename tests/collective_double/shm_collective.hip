// TEST INFRASTRUCTURE, not product code: a stand-in for librccl.so that lets several ranks run as separate
// PROCESSES ON ONE GPU (RCCL itself refuses two ranks on one device).  It implements the nccl* entry points
// csrc/comm.hip binds, with the all-reduce staged through a file-backed shared mapping: every rank copies its
// buffer to its slot, all ranks add the slots in rank order (so every rank computes the same bits), and copy the
// sum back.  Selected with RGCN_RCCL_LIBRARY=<this .so>; used by tests/test_gpu_multiprocess.py to drive the
// real multi-process code paths (bench.py --gpus N, rgcn_step_device / rgcn_train_step_device on world > 1
// contexts with a communicator) on the single-GPU test box.
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

constexpr size_t kSlotBytes = (size_t)96 << 20;      // largest exchange: [V,d] fp32 at WN18 size = 82 MB

struct Header {
  std::atomic<int32_t> arrived;
  std::atomic<int32_t> generation;
  std::atomic<int32_t> attached;
};

struct Comm {
  int rank = 0, nranks = 1;
  char name[64] = {0};
  uint8_t* base = nullptr;
  size_t bytes = 0;
  Header* hdr() const { return reinterpret_cast<Header*>(base); }
  float* slot(int r) const { return reinterpret_cast<float*>(base + 4096 + (size_t)r * kSlotBytes); }
};

bool barrier(Comm* c) {
  Header* h = c->hdr();
  const int gen = h->generation.load();
  if (h->arrived.fetch_add(1) + 1 == c->nranks) {
    h->arrived.store(0);
    h->generation.fetch_add(1);
    return true;
  }
  const time_t t0 = time(nullptr);
  while (h->generation.load() == gen) {
    if (time(nullptr) - t0 > 120) return false;      // a peer died: fail instead of hanging the box
    usleep(20);
  }
  return true;
}

}  // namespace

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId* id) {
  memset(id->internal, 0, 128);
  snprintf(id->internal, 64, "/tmp/rgcn_shm_%d_%ld", (int)getpid(), (long)time(nullptr));
  return 0;
}

int ncclCommCount(void* comm, int* count) { *count = static_cast<Comm*>(comm)->nranks; return 0; }
int ncclCommUserRank(void* comm, int* rank) { *rank = static_cast<Comm*>(comm)->rank; return 0; }

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  Comm* c = new Comm();
  c->rank = rank;
  c->nranks = nranks;
  strncpy(c->name, id.internal, 63);
  c->bytes = 4096 + (size_t)nranks * kSlotBytes;
  // a MAP_SHARED file under /tmp rather than shm_open: containers often cap /dev/shm at 64 MB
  int fd = open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) { delete c; return 2; }
  if (ftruncate(fd, (off_t)c->bytes) != 0) { close(fd); delete c; return 2; }
  void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { delete c; return 2; }
  c->base = static_cast<uint8_t*>(p);
  c->hdr()->attached.fetch_add(1);
  const time_t t0 = time(nullptr);
  while (c->hdr()->attached.load() < nranks) {       // everyone mapped the segment before anyone uses it
    if (time(nullptr) - t0 > 120) return 2;
    usleep(100);
  }
  *comm = c;
  return 0;
}

int ncclCommDestroy(void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return 0;
  if (c->base) munmap(c->base, c->bytes);
  unlink(c->name);                               // the last unlink wins; earlier ones are harmless
  delete c;
  return 0;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(comm);
  if (dtype != 7 || op != 0) return 4;               // float32 sum only
  if (count * 4 > kSlotBytes) return 4;
  if (hipMemcpyAsync(c->slot(c->rank), send, count * 4, hipMemcpyDeviceToHost, stream) != hipSuccess) return 1;
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  if (!barrier(c)) return 3;
  float* out = static_cast<float*>(malloc(count * 4));
  if (!out) return 2;
  memcpy(out, c->slot(0), count * 4);
  for (int r = 1; r < c->nranks; ++r) {
    const float* s = c->slot(r);
    for (size_t i = 0; i < count; ++i) out[i] += s[i];
  }
  if (!barrier(c)) { free(out); return 3; }          // nobody overwrites a slot before everyone has read it
  hipError_t e = hipMemcpyAsync(recv, out, count * 4, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  free(out);
  return e == hipSuccess ? 0 : 1;
}

// recv (count floats) = sum over ranks of their send[rank * count, +count)
int ncclReduceScatter(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(comm);
  if (dtype != 7 || op != 0) return 4;
  const size_t total = count * (size_t)c->nranks;
  if (total * 4 > kSlotBytes) return 4;
  if (hipMemcpyAsync(c->slot(c->rank), send, total * 4, hipMemcpyDeviceToHost, stream) != hipSuccess) return 1;
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  if (!barrier(c)) return 3;
  float* out = static_cast<float*>(malloc(count * 4));
  if (!out) return 2;
  const size_t off = (size_t)c->rank * count;
  memcpy(out, c->slot(0) + off, count * 4);
  for (int r = 1; r < c->nranks; ++r) {
    const float* s = c->slot(r) + off;
    for (size_t i = 0; i < count; ++i) out[i] += s[i];
  }
  if (!barrier(c)) { free(out); return 3; }
  hipError_t e = hipMemcpyAsync(recv, out, count * 4, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  free(out);
  return e == hipSuccess ? 0 : 1;
}

// recv (nranks * count floats) = the ranks' send buffers (count floats each) in rank order
int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(comm);
  if (dtype != 7) return 4;
  if (count * 4 > kSlotBytes) return 4;
  if (hipMemcpyAsync(c->slot(c->rank), send, count * 4, hipMemcpyDeviceToHost, stream) != hipSuccess) return 1;
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  if (!barrier(c)) return 3;
  hipError_t e = hipSuccess;
  for (int r = 0; r < c->nranks && e == hipSuccess; ++r)
    e = hipMemcpyAsync(static_cast<float*>(recv) + (size_t)r * count, c->slot(r), count * 4, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (!barrier(c)) return 3;
  return e == hipSuccess ? 0 : 1;
}

const char* ncclGetErrorString(int code) {
  switch (code) {
    case 0: return "success";
    case 1: return "hip error (test collective)";
    case 2: return "shared memory error (test collective)";
    case 3: return "a peer never arrived (test collective)";
    case 4: return "unsupported argument (test collective: float32 sum, <= 96 MB)";
    default: return "unknown";
  }
}

}  // extern "C"
