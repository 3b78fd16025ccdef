// Standalone microbenchmark: what v_mfma_f32_32x32x16_bf16 sustains on this GPU -- in a long run and in launches as short
// as the encoder's GEMMs (7.27 GFLOP x 6 partial products = 43.6 GFLOP of bf16 MFMA work, nominally 17.4 us at 2.5 PF).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_peak.hip -o /tmp/mfma_bf16_peak && /tmp/mfma_bf16_peak
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(float* out, int iters, unsigned a0, unsigned b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const bf16x8 a = __builtin_bit_cast(bf16x8, u32x4{a0, a0 + threadIdx.x, a0, a0});
  const bf16x8 b = __builtin_bit_cast(bf16x8, u32x4{b0, b0, b0 + threadIdx.x, b0});
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int grid, int iters, int reps) {
  float* out;
  hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 0x3f803f80u, 0x3f003f00u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 0x3f803f80u, 0x3f003f00u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)reps * grid * 4.0 * iters * NACC * 32768.0;
  printf("acc=%d grid=%4d (%.2f wg/cu) mfma/wave=%6d : %8.2f us/launch  %7.1f TFLOP/s\n", NACC, grid, grid / 256.0, iters * NACC,
         1e3 * ms / reps, flops / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  // long runs: the sustained ceiling
  run<4>(256, 20000, 5); run<4>(512, 20000, 5); run<8>(256, 10000, 5);
  // the GEMM's own amount of work per launch: 456 workgroups x 4 waves x 768 MFMAs, and 228 x 4 x 1536
  run<4>(456, 192, 50); run<8>(228, 192, 50); run<4>(512, 192, 50); run<8>(256, 192, 50);
  return 0;
}
