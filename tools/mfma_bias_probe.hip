// Mean signed error of ONE v_mfma_f32_32x32x16_bf16 against the exactly rounded result, as a function of how small
// the products are beside the accumulator.  (tools/gemm_bias_probe.py: the split-bf16 GEMM's error has a mean of
// -0.08 rms -- toward -inf whatever the sign of the result -- the fp32 MFMA's has none.  Which MFMAs of the six per
// k-slice carry it?)
//   D = A.B + C with A [32x16], B [16x32] random bf16 scaled by 2^-shift, C [32x32] random fp32 of unit scale.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_bias_probe.hip -o /tmp/mfma_bias_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

// A: [T][32][16] floats (bf16-exact), B: [T][16][32], C/D: [T][32][32]
__global__ void k_mfma(const float* A, const float* B, const float* C, float* D, int T) {
  const int lane = threadIdx.x, li = lane & 31, h = lane >> 5;
  for (int t = blockIdx.x; t < T; t += gridDim.x) {
    bf16x8 a, b;
    for (int q = 0; q < 8; ++q) {
      a[q] = (__bf16)A[(size_t)t * 512 + li * 16 + 8 * h + q];
      b[q] = (__bf16)B[(size_t)t * 512 + (8 * h + q) * 32 + li];
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = C[(size_t)t * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + li];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[(size_t)t * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + li] = acc[r];
  }
}

static float to_bf16(float x) {   // round to nearest even
  uint32_t u; std::memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  float y; std::memcpy(&y, &u, 4);
  return y;
}

int main() {
  const int T = 256;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> A(T * 512), B(T * 512), C(T * 1024), D(T * 1024);
  float *dA, *dB, *dC, *dD;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
  printf("%-44s %12s %12s %12s\n", "scenario", "rms err/ulp", "mean err/ulp", "mean/rms");
  for (int cmode = 0; cmode < 3; ++cmode)       // 0: C random sign, 1: C > 0, 2: C = 0
    for (int shift = 0; shift <= 36; shift += 3) {
      for (auto& x : A) x = to_bf16(nd(rng));
      for (auto& x : B) x = to_bf16(std::ldexp(nd(rng), -shift));
      for (auto& x : C) x = cmode == 2 ? 0.f : (cmode == 1 ? std::fabs(nd(rng)) + 4.f : 4.f * nd(rng));
      hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
      hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
      hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k_mfma, dim3(64), dim3(64), 0, 0, dA, dB, dC, dD, T);
      if (hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
      double s1 = 0, s2 = 0; size_t n = 0;
      for (int t = 0; t < T; ++t)
        for (int i = 0; i < 32; ++i)
          for (int j = 0; j < 32; ++j) {
            long double ex = C[(size_t)t * 1024 + i * 32 + j];
            for (int k = 0; k < 16; ++k) ex += (long double)A[(size_t)t * 512 + i * 16 + k] * (long double)B[(size_t)t * 512 + k * 32 + j];
            const float got = D[(size_t)t * 1024 + i * 32 + j];
            int e; std::frexp((double)(got != 0.f ? got : (float)ex), &e);
            const double ulp = std::ldexp(1.0, e - 24);
            const double err = (double)((long double)got - ex) / ulp;
            s1 += err; s2 += err * err; ++n;
          }
      char name[96];
      std::snprintf(name, sizeof name, "C %s, products ~2^-%d of C", cmode == 0 ? "~4 N(0,1)" : cmode == 1 ? "in [4,8)" : "= 0 (vs own scale)", shift);
      printf("%-44s %12.4f %+12.4f %+12.4f\n", name, std::sqrt(s2 / n), s1 / n, (s1 / n) / std::sqrt(s2 / n));
    }
  return 0;
}
