// Does a kernel that issues v_mfma_f32_32x32x16_bf16 disturb a kernel running beside it on the same CUs?
// A = matrix-core burner, B = a register-holding streaming kernel of the shape of k_block_msg_fwd (25
// coefficients loaded once, then y = W x over many rows).  B's output is checked against the host.
// Usage: mfma_corun <burner: 0 none, 1 bf16 32x32x16, 2 fp32 32x32x2> <flags>
//   flags: 1 keeper on a high-priority stream, 2 burner allocates 62 KB of LDS, 4 burner has a barrier per
//          iteration, 8 few burner workgroups (12) instead of 512, 16 keeper uses 1000-thread workgroups
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

template <int KIND>
__global__ void __launch_bounds__(256, 2) burner(float* out, int iters, int barrier) {
  extern __shared__ uint32_t lds[];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  u32x4 av = {0u, 0u, 0u, 0u}, bv = {0u, 0u, 0u, 0u};
  asm volatile("" : "+v"(av), "+v"(bv));
  const bf16x8 a = __builtin_bit_cast(bf16x8, av), b = __builtin_bit_cast(bf16x8, bv);
  const float fa = threadIdx.x * 1e-3f, fb = 1.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 24; ++m) {
      if constexpr (KIND == 1) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
      else acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[m & 3], 0, 0, 0);
    }
    if (barrier) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s + lds[threadIdx.x];
}

__global__ void keeper(const float* __restrict__ W, const float* __restrict__ X, float* __restrict__ Y,
                       int nb, int d, int rows_per_block, int G, int live_blocks) {
  if ((int)blockIdx.x >= live_blocks) return;
  const int g = threadIdx.x / nb;
  if (g >= G) return;
  const int b = threadIdx.x - g * nb;
  const int rel = blockIdx.x % 16;
  float w[25];
#pragma unroll
  for (int k = 0; k < 25; ++k) w[k] = W[((size_t)rel * 25 + k) * nb + b];
  for (int r = g; r < rows_per_block; r += G) {
    const size_t row = (size_t)blockIdx.x * rows_per_block + r;
    const float* xp = X + row * d + 5 * b;
    float x[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) x[q] = xp[q];
    float* yp = Y + row * d + 5 * b;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      float y = 0.f;
#pragma unroll
      for (int q = 0; q < 5; ++q) y = fmaf(w[i * 5 + q], x[q], y);
      yp[i] = y;
    }
  }
}

int main(int argc, char** argv) {
  const int kind = argc > 1 ? atoi(argv[1]) : 1, flags = argc > 2 ? atoi(argv[2]) : 0;
  const int nb = 100, d = 500, blocks = 600, rpb = 48, rows = blocks * rpb;
  std::vector<float> hW(16 * 25 * nb), hX((size_t)rows * d), hY((size_t)rows * d), ref((size_t)rows * d);
  srand(1);
  for (auto& v : hW) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hX) v = (rand() % 2001 - 1000) * 1e-3f;
  for (int blk = 0; blk < blocks; ++blk)
    for (int r = 0; r < rpb; ++r)
      for (int b = 0; b < nb; ++b)
        for (int i = 0; i < 5; ++i) {
          float y = 0.f;
          const size_t row = (size_t)blk * rpb + r;
          for (int q = 0; q < 5; ++q) y = fmaf(hW[((size_t)(blk % 16) * 25 + i * 5 + q) * nb + b], hX[row * d + 5 * b + q], y);
          ref[row * d + 5 * b + i] = y;
        }
  float *W, *X, *Y, *out;
  hipMalloc(&W, hW.size() * 4); hipMalloc(&X, hX.size() * 4); hipMalloc(&Y, hY.size() * 4); hipMalloc(&out, 4096);
  hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice);
  hipStream_t s1, s2;
  int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, lo);
  hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, (flags & 1) ? hi : lo);
  const size_t ldsb = (flags & 2) ? 62976 : 0;
  hipFuncSetAttribute(reinterpret_cast<const void*>(burner<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(burner<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
  const int bgrid = (flags & 8) ? 12 : 512, G = (flags & 16) ? 10 : 1, kthreads = (flags & 16) ? 1000 : 128;
  int bad_runs = 0; long bad_total = 0;
  for (int run = 0; run < 40; ++run) {
    hipMemsetAsync(Y, 0, hY.size() * 4, s2);
    hipStreamSynchronize(s2);
    if (kind == 1) hipLaunchKernelGGL((burner<1>), dim3(bgrid), dim3(256), ldsb, s1, out, 400, (flags & 4) ? 1 : 0);
    if (kind == 2) hipLaunchKernelGGL((burner<2>), dim3(bgrid), dim3(256), ldsb, s1, out, 200, (flags & 4) ? 1 : 0);
    hipLaunchKernelGGL(keeper, dim3(blocks + 400), dim3(kthreads), 0, s2, W, X, Y, nb, d, rpb, G, blocks);
    hipStreamSynchronize(s1); hipStreamSynchronize(s2);
    hipMemcpy(hY.data(), Y, hY.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (size_t i = 0; i < hY.size(); ++i)
      if (hY[i] != ref[i]) {
        if (bad < 6 && run < 2 && getenv("CORUN_SHOW"))
          printf("   run %d row %zu col %zu (block %zu, i=%zu): got %.7g expected %.7g\n", run, i / d, i % d, (i % d) / 5, i % 5, hY[i], ref[i]);
        ++bad;
      }
    bad_total += bad; bad_runs += bad > 0;
  }
  printf("burner %d flags %2d: %d of 40 runs with wrong keeper output (%ld wrong elements in total)\n",
         kind, flags, bad_runs, bad_total);
  return 0;
}
