// How many independent instructions hide behind one v_mfma_f32_32x32x16_bf16 on gfx950?
// One workgroup of 256 threads per CU (one wave per SIMD) or 512 (two per SIMD); each wave issues a chain of
// MFMAs over 4 accumulators with F filler instructions after each one.  Prints ns per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

template <int F, int KIND>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
  __shared__ uint32_t lds[4096];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  u32x4 av = {threadIdx.x, 2u, 3u, 4u}, bv = {5u, 6u, threadIdx.x, 8u};
  const bf16x8 a = __builtin_bit_cast(bf16x8, av), b = __builtin_bit_cast(bf16x8, bv);
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
  uint32_t* dst = lds + (threadIdx.x & 255) * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < F; ++f) {
        float& v = x[(m * F + f) & 7];
        if constexpr (KIND == 0) v = v * 1.0001f + 0.5f;                                   // v_fma
        if constexpr (KIND == 1) v = v - __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) & 0xffff0000u);  // and+sub (2 ops)
        if constexpr (KIND == 2) {                                                         // v_dot2c
          const bf16x2 s = {(__bf16)-1.0f, (__bf16)-0.0f};
          v = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, __builtin_bit_cast(uint32_t, v)), s, v, false);
        }
        if constexpr (KIND == 3) dst[f & 3] = __builtin_bit_cast(uint32_t, v);             // ds_write_b32
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  for (int i = 0; i < 8; ++i) s += x[i];
  if (s == 123.456f) out[0] = s + lds[threadIdx.x];
}

template <int F, int KIND>
void run(const char* name, int threads, float* out) {
  const int iters = 2000, blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<F, KIND>), dim3(blocks), dim3(threads), 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<F, KIND>), dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * 8 * (threads / 256);
  printf("%-10s F=%d  %d waves/SIMD: %.1f ns per MFMA per SIMD\n", name, F, threads / 256, ms * 1e6 / mfma_per_simd);
}

int main() {
  float* out; hipMalloc(&out, 4096);
  for (int threads : {256, 512}) {
    run<0, 0>("none", threads, out);
    run<2, 0>("fma", threads, out); run<4, 0>("fma", threads, out); run<6, 0>("fma", threads, out); run<8, 0>("fma", threads, out);
    run<2, 1>("and+sub", threads, out); run<3, 1>("and+sub", threads, out);
    run<2, 2>("dot2c", threads, out); run<4, 2>("dot2c", threads, out);
    run<1, 3>("ds_write", threads, out); run<2, 3>("ds_write", threads, out);
  }
  return 0;
}
