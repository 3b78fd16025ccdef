// What a FORK costs the stream that is forked from, by mechanism (gfx950, ROCm 7.2).  Main stream: A (writes 32 MB) ->
// [fork] -> B (writes 32 MB); side stream: a tiny kernel S ordered behind A.  300 iterations, main-stream time per
// iteration by timing events, against the same loop without a fork:
//   1  hipEventRecord(default event) on main + hipStreamWaitEvent on the side stream
//   2  the same with hipEventDisableTiming
//   3  the same with hipEventDisableTiming | hipEventDisableSystemFence
//   4  A launched by hipExtLaunchKernel with stopEvent = a default event (no record call) + hipStreamWaitEvent
//   5  the same, stopEvent with hipEventDisableTiming
//   6  the same, stopEvent with hipEventDisableTiming | hipEventDisableSystemFence
// S checks that it sees A's writes of ITS iteration (a counter in the buffer's last word); mismatches are counted.
//   hipcc --offload-arch=gfx950 -O2 tools/fork_probe.hip -o /tmp/fork_probe && /tmp/fork_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>

__global__ void fill(float* p, size_t n, float v, int* stamp, int it) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
  if (i == n - 1) *stamp = it;
}
__global__ void check(const int* stamp, int it, int* bad) {
  if (threadIdx.x == 0 && *stamp != it) atomicAdd(bad, 1);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const size_t n = 8u << 20;
  float *a, *b; int *stamp, *stamp2, *bad;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&stamp, 4)); CK(hipMalloc(&stamp2, 4)); CK(hipMalloc(&bad, 4));
  hipStream_t s, s2;
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo)); CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi));
  hipEvent_t t0, t1, ev[7];
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  const unsigned flags[7] = {0, hipEventDefault, hipEventDisableTiming, hipEventDisableTiming | hipEventDisableSystemFence,
                             hipEventDefault, hipEventDisableTiming, hipEventDisableTiming | hipEventDisableSystemFence};
  for (int m = 1; m < 7; ++m) CK(hipEventCreateWithFlags(&ev[m], flags[m]));
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 7; ++mode) {
      CK(hipMemset(bad, 0, 4));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(t0, s));
      for (int it = 0; it < 300; ++it) {
        if (mode >= 4) {
          float v = 1.0f;
          void* args[] = {(void*)&a, (void*)&n, (void*)&v, (void*)&stamp, (void*)&it};
          CK(hipExtLaunchKernel((const void*)fill, grid, block, args, 0, s, nullptr, ev[mode], 0));
        } else {
          hipLaunchKernelGGL(fill, grid, block, 0, s, a, n, 1.0f, stamp, it);
          if (mode >= 1) CK(hipEventRecord(ev[mode], s));
        }
        if (mode >= 1) {
          CK(hipStreamWaitEvent(s2, ev[mode], 0));
          hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, s2, stamp, it, bad);
        }
        hipLaunchKernelGGL(fill, grid, block, 0, s, b, n, 2.0f, stamp2, it);
        // (the next iteration's A overwrites the stamp: S must have read it by then -> the side stream is joined
        // every iteration through a wait that is NOT on the timed path's critical resources: a host sync every 50)
        if (it % 50 == 49) { CK(hipStreamSynchronize(s2)); }
      }
      CK(hipEventRecord(t1, s));
      CK(hipDeviceSynchronize());
      float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
      int hbad = 0; CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
      printf("rep %d mode %d: %.2f us per iteration   (stale reads: %d)\n", rep, mode, ms * 1e3 / 300, hbad);
    }
  return 0;
}
