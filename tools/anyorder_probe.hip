// Does hipExtAnyOrderLaunch (AQL packet without the barrier bit) let a kernel start beside its predecessor in the SAME
// stream on gfx950?  hip_ext.h says the flag is "not supported on AMD GFX9xx boards"; this measures what happens.
// Kernel A spins ~60 us on one workgroup and stamps start / end (s_memrealtime, 100 MHz); kernel B (any-order flag)
// stamps its own start.  Overlap <=> B.start < A.end.  Also times a dependent third kernel (normal launch) to see what
// the in-queue join costs, against the cross-stream event join the library uses today.
//   hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o /tmp/anyorder_probe && /tmp/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>

__global__ void spin(uint64_t* stamps, int slot, uint64_t ticks) {
  const uint64_t t0 = wall_clock64();
  if (threadIdx.x == 0) stamps[2 * slot] = t0;
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0) stamps[2 * slot + 1] = wall_clock64();
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  uint64_t* d; CK(hipMalloc(&d, 64 * sizeof(uint64_t)));
  uint64_t h[64];
  hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t ev, ev2; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
  int rate = 0; CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));   // kHz
  const double us_per_tick = 1e3 / rate;
  const uint64_t ticks60 = (uint64_t)(60.0 / us_per_tick), ticks20 = (uint64_t)(20.0 / us_per_tick);
  printf("wall clock %d kHz\n", rate);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemsetAsync(d, 0, 64 * sizeof(uint64_t), s));
      CK(hipStreamSynchronize(s));
      if (mode == 0) {          // plain in-order: A, B, C
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 0, ticks60);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 1, ticks20);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 2, ticks20);
      } else if (mode == 1) {   // B any-order in the same stream, C normal
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 0, ticks60);
        hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d, 1, ticks20);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 2, ticks20);
      } else {                  // B on a second stream (fork / join with events), C on the first
        CK(hipEventRecord(ev, s));
        CK(hipStreamWaitEvent(s2, ev, 0));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 0, ticks60);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s2, d, 1, ticks20);
        CK(hipEventRecord(ev2, s2));
        CK(hipStreamWaitEvent(s, ev2, 0));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 2, ticks20);
      }
      CK(hipGetLastError());
      CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
      CK(hipMemcpy(h, d, 64 * sizeof(uint64_t), hipMemcpyDeviceToHost));
      const double a0 = 0, a1 = (h[1] - h[0]) * us_per_tick, b0 = ((double)h[2] - (double)h[0]) * us_per_tick,
                   b1 = ((double)h[3] - (double)h[0]) * us_per_tick, c0 = ((double)h[4] - (double)h[0]) * us_per_tick;
      printf("mode %d (%s) rep %d: A [%.1f, %.1f]  B [%.1f, %.1f]  C starts %.1f  -> B %s A; join gap %.1f us\n", mode,
             mode == 0 ? "in-order" : mode == 1 ? "any-order flag" : "second stream", rep, a0, a1, b0, b1, c0,
             b0 < a1 ? "OVERLAPS" : "follows", c0 - (a1 > b1 ? a1 : b1));
    }
  }
  return 0;
}
