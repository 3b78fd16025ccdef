// Microbenchmark: the GEMM's inner loop in isolation (LDS-resident operands, no global traffic):
// how close do different LDS->MFMA schedules get to the 155 TF fp32 MFMA ceiling?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_lds_bench.hip -o /tmp/mlb && /tmp/mlb
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int BM = 128, BN = 128, BK = 16, LDK = BK + 4;

template <int MODE, int TM, int TN, int NT>
__global__ void __launch_bounds__(NT) k_loop(float* out, int iters, int barrier) {
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM * LDK + BK * BN)];
  for (int i = threadIdx.x; i < 2 * (BM * LDK + BK * BN); i += NT) lds[i] = (float)((i * 7) % 13) * 0.01f;
  __syncthreads();
  constexpr int WGN = BN / (TN * 32);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave / WGN) * (TM * 32), wn = (wave % WGN) * (TN * 32);
  const int li = lane & 31, h = lane >> 5;
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    const float* a_lds = lds + (it & 1) * (BM * LDK + BK * BN);
    const float* b_lds = a_lds + BM * LDK;
    if (MODE == 0) {
#pragma unroll
      for (int kk = 0; kk < BK / 8; ++kk) {
        float fa[TM][4], fb[TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          float4 v = *reinterpret_cast<const float4*>(a_lds + (wm + 32 * i + li) * LDK + 8 * kk + 4 * h);
          fa[i][0] = v.x; fa[i][1] = v.y; fa[i][2] = v.z; fa[i][3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float* p = b_lds + (8 * kk + 4 * h) * BN + wn + 32 * j + li;
          fb[j][0] = p[0]; fb[j][1] = p[BN]; fb[j][2] = p[2 * BN]; fb[j][3] = p[3 * BN];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][t], fb[j][t], acc[i][j], 0, 0, 0);
      }
    } else {
      // MODE 1: every fragment of the k-tile first, one scheduling fence, then the MFMAs back-to-back
      float fa[BK / 8][TM][4], fb[BK / 8][TN][4];
#pragma unroll
      for (int kk = 0; kk < BK / 8; ++kk) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          float4 v = *reinterpret_cast<const float4*>(a_lds + (wm + 32 * i + li) * LDK + 8 * kk + 4 * h);
          fa[kk][i][0] = v.x; fa[kk][i][1] = v.y; fa[kk][i][2] = v.z; fa[kk][i][3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float* p = b_lds + (8 * kk + 4 * h) * BN + wn + 32 * j + li;
          fb[kk][j][0] = p[0]; fb[kk][j][1] = p[BN]; fb[kk][j][2] = p[2 * BN]; fb[kk][j][3] = p[3 * BN];
        }
      }
      if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < BK / 8; ++kk)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][i][t], fb[kk][j][t], acc[i][j], 0, 0, 0);
    }
    if (barrier) __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * NT + threadIdx.x] = s;
}

template <int MODE, int TM, int TN, int NT>
void run(const char* name, int wgs, int iters, int barrier) {
  float* out; hipMalloc(&out, (size_t)wgs * NT * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_loop<MODE, TM, TN, NT>), dim3(wgs), dim3(NT), 0, 0, out, iters, barrier);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k_loop<MODE, TM, TN, NT>), dim3(wgs), dim3(NT), 0, 0, out, iters, barrier);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = 5.0 * wgs * iters * 2.0 * BM * BN * BK;
  printf("%-28s wgs=%4d iters=%5d barrier=%d : %8.3f ms %7.1f TF\n", name, wgs, iters, barrier, ms / 5, flops / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  for (int barrier = 0; barrier < 2; ++barrier) {
    for (int wgs : {256, 456, 512, 1024}) {
      run<0, 1, 2, 512>("8w(1x2) per-kk", wgs, 2000, barrier);
      run<1, 1, 2, 512>("8w(1x2) all-frags-first", wgs, 2000, barrier);
      run<2, 1, 2, 512>("8w(1x2) all-frags+fence", wgs, 2000, barrier);
      run<0, 2, 2, 256>("4w(2x2) per-kk", wgs, 2000, barrier);
      run<2, 2, 2, 256>("4w(2x2) all-frags+fence", wgs, 2000, barrier);
    }
  }
  run<0, 1, 2, 512>("8w(1x2) per-kk short", 456, 32, 1);
  run<2, 1, 2, 512>("8w(1x2) fence short", 456, 32, 1);
  return 0;
}
