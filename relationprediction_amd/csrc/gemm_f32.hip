// fp32 MFMA GEMM for the dense contractions of the encoder: the self-loop product H.W_self
// (dot_or_lookup matmul branch, code/common/shared_functions.py:5-9 via
// gcn_basis_concat.py:65-66 / gcn_basis.py:70-71), its two gradients, and the basis
// contraction [V, 2B.d] x [2B.d, d] (gcn_basis.py:54-68, aggregate-first form).
//
// gfx950 has no TF32/xf32; `v_mfma_f32_32x32x2_f32` is exact fp32 (bitwise an fmaf chain) at the
// fp32 vector rate (157 TF peak), which is what the 1e-4 parity bar needs.
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 2x2 MFMA
// 32x32 tiles = 64 accumulator registers), BK = 16, register-staged double-buffered LDS.
// Operands come in two storage flavours, handled without any transposition in flight:
//   k-contiguous ("KC",  X[row][k]):  LDS tile [128][16] with row stride 20 dwords (bank-conflict
//        free for ds_read_b128 in its 16-lane groups); a lane fetches 4 consecutive k with one
//        ds_read_b128.
//   row-contiguous ("RC", X[k][row]): LDS tile [16][128]; a lane fetches its 4 k values with four
//        conflict-free ds_read_b32.
// Within each group of 8 k, MFMA #t consumes k = {t, 4+t} (lane half h supplies k = 4h+t) for BOTH
// operands, so the permuted k order is consistent and only changes the fp32 summation order.
//   NN (forward):   A = H   [M,K] KC,  B = W   [K,N] RC
//   NT (dH):        A = dS  [M,K] KC,  B = W   [N,K] KC
//   TN (dW):        A = H   [K,M] RC,  B = dS  [K,N] RC, split over K into slabs + ordered reduce
#include "rgcn_internal.h"

namespace rgcn {

namespace {

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDK = 20;    // row stride (dwords) of a k-contiguous tile
constexpr int LDR = 128;   // row stride (dwords) of a row-contiguous tile
constexpr int TILE_KC = BM * LDK;   // 2560 floats
constexpr int TILE_RC = BK * LDR;   // 2048 floats

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;          // output (or slab base when split_k > 1)
  int M, N, K;
  int lda, ldb, ldc;
  int k_per_split;   // multiple of BK
  int vecA, vecB;    // 16-byte vector loads legal
};

// ---- global -> registers (one k-tile of one operand: 512 float4, 2 per thread) -----------------
template <bool KC>
__device__ __forceinline__ void load_tile(const float* __restrict__ X, int ld, int rows, int kdim,
                                          int row0, int k0, int kend, int vec, float4 (&r)[2]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int f = t + 256 * p;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (KC) {
      const int row = row0 + (f >> 2);
      const int k = k0 + ((f & 3) << 2);
      if (row < rows) {
        const float* src = X + (size_t)row * ld + k;
        if (vec && k + 3 < kend) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          if (k + 0 < kend) v.x = src[0];
          if (k + 1 < kend) v.y = src[1];
          if (k + 2 < kend) v.z = src[2];
          if (k + 3 < kend) v.w = src[3];
        }
      }
    } else {
      const int k = k0 + (f >> 5);
      const int row = row0 + ((f & 31) << 2);
      if (k < kend) {
        const float* src = X + (size_t)k * ld + row;
        if (vec && row + 3 < rows) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          if (row + 0 < rows) v.x = src[0];
          if (row + 1 < rows) v.y = src[1];
          if (row + 2 < rows) v.z = src[2];
          if (row + 3 < rows) v.w = src[3];
        }
      }
    }
    r[p] = v;
  }
  (void)kdim;
}

// ---- registers -> LDS -------------------------------------------------------------------------
template <bool KC>
__device__ __forceinline__ void store_tile(float* __restrict__ lds, const float4 (&r)[2]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int f = t + 256 * p;
    int off;
    if constexpr (KC) off = (f >> 2) * LDK + ((f & 3) << 2);
    else off = (f >> 5) * LDR + ((f & 31) << 2);
    *reinterpret_cast<float4*>(lds + off) = r[p];
  }
}

// ---- LDS -> MFMA fragments: 4 k values (k = 8*kk + 4*h + t, t = 0..3) of row `row` ------------
template <bool KC>
__device__ __forceinline__ void load_frag(const float* __restrict__ lds, int row, int kk, int h,
                                          float (&f)[4]) {
  if constexpr (KC) {
    const float4 v = *reinterpret_cast<const float4*>(lds + row * LDK + 8 * kk + 4 * h);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  } else {
    const float* p = lds + (8 * kk + 4 * h) * LDR + row;
    f[0] = p[0]; f[1] = p[LDR]; f[2] = p[2 * LDR]; f[3] = p[3 * LDR];
  }
}

template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(256) k_gemm_f32(GemmArgs g) {
  constexpr int TA = A_KC ? TILE_KC : TILE_RC;
  constexpr int TB = B_KC ? TILE_KC : TILE_RC;
  __shared__ __attribute__((aligned(16))) float lds[2 * (TA + TB)];
  // buffer b of A lives at lds + b*(TA+TB), buffer b of B right behind it

  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int ks = blockIdx.z * g.k_per_split;
  const int ke = min(g.K, ks + g.k_per_split);
  const int nkt = (ke - ks + BK - 1) / BK;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int li = lane & 31, h = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  float4 ra[2], rb[2];
  if (nkt > 0) {
    load_tile<A_KC>(g.A, g.lda, g.M, g.K, m0, ks, ke, g.vecA, ra);
    load_tile<B_KC>(g.B, g.ldb, g.N, g.K, n0, ks, ke, g.vecB, rb);
    store_tile<A_KC>(lds, ra);
    store_tile<B_KC>(lds + TA, rb);
  }
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nkt;
    if (more) {   // next tile's global loads fly under this tile's MFMAs
      load_tile<A_KC>(g.A, g.lda, g.M, g.K, m0, ks + (kt + 1) * BK, ke, g.vecA, ra);
      load_tile<B_KC>(g.B, g.ldb, g.N, g.K, n0, ks + (kt + 1) * BK, ke, g.vecB, rb);
    }
    const float* a_lds = lds + cur * (TA + TB);
    const float* b_lds = a_lds + TA;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      float fa[2][4], fb[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i) load_frag<A_KC>(a_lds, wm + 32 * i + li, kk, h, fa[i]);
#pragma unroll
      for (int j = 0; j < 2; ++j) load_frag<B_KC>(b_lds, wn + 32 * j + li, kk, h, fb[j]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][t], fb[j][t], acc[i][j], 0, 0, 0);
    }
    if (more) {
      store_tile<A_KC>(lds + (cur ^ 1) * (TA + TB), ra);
      store_tile<B_KC>(lds + (cur ^ 1) * (TA + TB) + TA, rb);
    }
    __syncthreads();
  }

  // epilogue: acc register r of lane l holds C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
  float* C = g.C + (size_t)blockIdx.z * g.M * g.ldc;   // slab z (ldc == N for slabs)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn + 32 * j + li;
      if (col < g.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < g.M) C[(size_t)row * g.ldc + col] = acc[i][j][r];
        }
      }
    }
}

__global__ void k_splitk_reduce(const float* __restrict__ slab, float* __restrict__ C, int M, int N,
                                int ldc, int splits) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t mn = (int64_t)M * N;
  if (i >= mn) return;
  float acc = 0.0f;
  for (int s = 0; s < splits; ++s) acc += slab[(size_t)s * mn + i];
  const int row = (int)(i / N), col = (int)(i - (int64_t)row * N);
  C[(size_t)row * ldc + col] = acc;
}

bool vec_ok(const float* p, int ld) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0 && (ld % 4) == 0; }

}  // namespace

rgcn_status gemm_f32(rgcn_ctx* c, const char* tag, bool a_kc, bool b_kc, int M, int N, int K,
                     const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                     int split_k) {
  if (M <= 0 || N <= 0) return RGCN_OK;
  if (a_kc == false && b_kc == true) RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "gemm TT form not instantiated");
  GemmArgs g;
  g.A = A; g.B = B; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb;
  g.vecA = vec_ok(A, lda) ? 1 : 0;
  g.vecB = vec_ok(B, ldb) ? 1 : 0;
  if (split_k < 1) split_k = 1;
  int kps = (K + split_k - 1) / split_k;
  kps = ((kps + BK - 1) / BK) * BK;
  if (kps < BK) kps = BK;
  split_k = K > 0 ? (K + kps - 1) / kps : 1;
  g.k_per_split = kps;
  const bool slabs = split_k > 1;
  if (slabs) {
    if ((size_t)split_k * M * N > c->slab_floats) RGCN_FAIL(c, RGCN_ERR_STATE, "internal: split-K slab too small");
    g.C = c->slab; g.ldc = N;
  } else {
    g.C = C; g.ldc = ldc;
  }
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, split_k), block(256);
  {
    ProfScope ps(c, tag, 4.0 * ((double)M * K + (double)K * N + (double)M * N), 2.0 * M * N * K);
    if (a_kc && !b_kc) hipLaunchKernelGGL((k_gemm_f32<true, false>), grid, block, 0, c->stream, g);
    else if (a_kc && b_kc) hipLaunchKernelGGL((k_gemm_f32<true, true>), grid, block, 0, c->stream, g);
    else hipLaunchKernelGGL((k_gemm_f32<false, false>), grid, block, 0, c->stream, g);
    RGCN_HIP(c, hipGetLastError());
  }
  if (slabs) {
    const int64_t mn = (int64_t)M * N;
    ProfScope ps(c, "splitk_reduce", 4.0 * mn * (split_k + 1), 0);
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, c->stream,
                       c->slab, C, M, N, ldc, split_k);
    RGCN_HIP(c, hipGetLastError());
  }
  return RGCN_OK;
}

}  // namespace rgcn
