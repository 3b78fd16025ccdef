// Standalone microbenchmark: fp32-input MFMA (v_mfma_f32_32x32x2_f32) issue ceiling on this GPU.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int wg_per_cu, int iters) {
  float* out;
  int grid = 256 * wg_per_cu;
  hipMalloc(&out, grid * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = 5.0 * grid * 4.0 * iters * NACC * 4096.0;
  printf("acc=%d wg/cu=%d iters=%d : %.3f ms  %.1f TFLOP/s\n", NACC, wg_per_cu, iters, ms / 5, flops / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  run<4>(1, 20000); run<4>(2, 20000); run<2>(2, 20000); run<1>(2, 20000); run<4>(1, 400); run<4>(2, 400);
  return 0;
}
