// fp32 MFMA GEMM for the dense contractions of the encoder: the self-loop product H.W_self
// (dot_or_lookup matmul branch, code/common/shared_functions.py:5-9 via
// gcn_basis_concat.py:65-66 / gcn_basis.py:70-71), its two gradients, and the basis
// contraction [V, 2B.d] x [2B.d, d] (gcn_basis.py:54-68, aggregate-first form).
//
// gfx950 has no TF32/xf32; `v_mfma_f32_32x32x2_f32` is exact fp32 (bitwise an fmaf chain) at the
// fp32 vector rate (157 TF peak), which is what the 1e-4 parity bar needs.
//
// Tiling: BM x BN output tile per workgroup of WGM x WGN waves, each wave owning a
// (BM/WGM) x (BN/WGN) sub-tile as TM x TN MFMA 32x32 tiles, BK-deep register-staged
// double-buffered LDS.  Operands come in two storage flavours, handled without any transposition:
//   k-contiguous ("KC",  X[row][k]):  LDS tile [rows][BK] with row stride BK+4 dwords (bank-conflict
//        free for ds_read_b128 in its 16-lane groups); a lane fetches 4 consecutive k with one
//        ds_read_b128.
//   row-contiguous ("RC", X[k][row]): LDS tile [BK][rows]; a lane fetches its 4 k values with four
//        conflict-free ds_read_b32.
// Within each group of 8 k, MFMA #t consumes k = {t, 4+t} (lane half h supplies k = 4h+t) for BOTH
// operands, so the permuted k order is consistent and only changes the fp32 summation order.
//   NN (forward):   A = H   [M,K] KC,  B = W   [K,N] RC
//   NT (dH):        A = dS  [M,K] KC,  B = W   [N,K] KC
//   TN (dW):        A = H   [K,M] RC,  B = dS  [K,N] RC, split over K into slabs + ordered reduce
// Workgroup ids are remapped so that the tiles sharing an A row-panel (and, for split-K, the tiles
// of one K slab) run on the same XCD and hit its L2 instead of re-fetching the panel over the fabric.
#include <cstdlib>

#include "rgcn_internal.h"

namespace rgcn {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;          // output (or slab base when split_k > 1)
  const float* zeros;  // >= 16 zero bytes, 16-byte aligned
  int M, N, K;
  int lda, ldb, ldc;
  int k_per_split;   // multiple of BK
  int tiles_m, tiles_n, splits;
  int swizzle;       // XCD-aware workgroup remap
  int vecC;          // 16-byte row-contiguous stores legal (aligned C, ldc % 4 == 0, N % 4 == 0)
  GemmBatch batch;   // groups (blockIdx.y) and their device-side extents
};

template <int BM_, int BN_, int BK_, int WGM_, int WGN_>
struct Cfg {
  static constexpr int BM = BM_, BN = BN_, BK = BK_, WGM = WGM_, WGN = WGN_;
  static constexpr int NT = 64 * WGM * WGN;
  static constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  static constexpr int LDK = BK + 4;
  static_assert(BM % (32 * WGM) == 0 && BN % (32 * WGN) == 0, "wave tiles are multiples of 32");
  static_assert(BK % 8 == 0, "BK multiple of 8");
  static_assert((BM * BK / 4) % NT == 0 && (BN * BK / 4) % NT == 0, "whole float4 passes");
};

// ---- global -> registers: one k-tile of one operand, ROWS x BK floats, P float4 per thread ------
// Branch-free and select-free on the DATA side: an out-of-range lane redirects its ADDRESS to a
// small zero-filled buffer, so the loads issue back-to-back at the top of the k-loop and nothing
// touches their results until the LDS store at the bottom (a guarded load, or a select on the loaded
// value, makes hipcc wait vmcnt(0) right behind the load and serialises HBM latency with the MFMAs).
// VEC: 16-byte loads are legal and validity is all-or-nothing per float4 (K resp. rows % 4 == 0).
template <bool KC, int ROWS, int BK, int NT, int P, bool VEC>
__device__ __forceinline__ void load_tile(const float* __restrict__ X, const float* __restrict__ zeros,
                                          int ld, int rows, int row0, int k0, int kend,
                                          float4 (&r)[P]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int f = t + NT * p;
    int row, k;
    size_t off;
    if constexpr (KC) {
      constexpr int Q = BK / 4;
      row = row0 + f / Q;
      k = k0 + (f % Q) * 4;
      off = (size_t)row * ld + k;
    } else {
      constexpr int Q = ROWS / 4;
      k = k0 + f / Q;
      row = row0 + (f % Q) * 4;
      off = (size_t)k * ld + row;
    }
    if constexpr (VEC) {
      const bool ok = row < rows && k < kend;
      const float* src = ok ? X + off : zeros;
      r[p] = *reinterpret_cast<const float4*>(src);
    } else {
      // element e advances along the contiguous dimension: k for KC, row for RC
      float e[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool ok = KC ? (row < rows && k + q < kend) : (k < kend && row + q < rows);
        const float* src = ok ? X + off + q : zeros;
        e[q] = *src;
      }
      r[p] = make_float4(e[0], e[1], e[2], e[3]);
    }
  }
}

// ---- registers -> LDS -------------------------------------------------------------------------
template <bool KC, int ROWS, int BK, int NT, int P>
__device__ __forceinline__ void store_tile(float* __restrict__ lds, const float4 (&r)[P]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int f = t + NT * p;
    int off;
    if constexpr (KC) off = (f / (BK / 4)) * (BK + 4) + (f % (BK / 4)) * 4;
    else off = (f / (ROWS / 4)) * ROWS + (f % (ROWS / 4)) * 4;
    *reinterpret_cast<float4*>(lds + off) = r[p];
  }
}

// ---- LDS -> MFMA fragments: 4 k values (k = 8*kk + 4*h + t, t = 0..3) of row `row` ------------
template <bool KC, int ROWS, int BK>
__device__ __forceinline__ void load_frag(const float* __restrict__ lds, int row, int kk, int h,
                                          float (&f)[4]) {
  if constexpr (KC) {
    const float4 v = *reinterpret_cast<const float4*>(lds + row * (BK + 4) + 8 * kk + 4 * h);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  } else {
    const float* p = lds + (8 * kk + 4 * h) * ROWS + row;
    f[0] = p[0]; f[1] = p[ROWS]; f[2] = p[2 * ROWS]; f[3] = p[3 * ROWS];
  }
}

template <bool A_KC, bool B_KC, bool VEC, class CF>
__global__ void __launch_bounds__(CF::NT) k_gemm_f32(GemmArgs g) {
  constexpr int BM = CF::BM, BN = CF::BN, BK = CF::BK, NT = CF::NT, TM = CF::TM, TN = CF::TN;
  constexpr int TA = A_KC ? BM * CF::LDK : BK * BM;
  constexpr int TB = B_KC ? BN * CF::LDK : BK * BN;
  constexpr int PA = BM * BK / 4 / NT, PB = BN * BK / 4 / NT;
  constexpr int EPI_LD = BN + 4;                       // row stride of the epilogue staging tile
  constexpr int EPI = 64 * EPI_LD;                     // 64 output rows at a time
  constexpr int LDS_FLOATS = 2 * (TA + TB) > EPI ? 2 * (TA + TB) : EPI;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

  // ---- workgroup -> (k-slab, row tile, column tile); XCD-aware so that neighbours share an L2
  int wg = blockIdx.x;
  const int total = g.tiles_m * g.tiles_n * g.splits;
  if (g.swizzle == 1) {
    const int xcd = wg & 7, idx = wg >> 3;
    const int q = total >> 3, r = total & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int per_split = g.tiles_m * g.tiles_n;
  int z = wg / per_split;
  const int rem = wg - z * per_split;
  int m0 = (rem / g.tiles_n) * BM, n0 = (rem % g.tiles_n) * BN;
  if (g.swizzle == 2) {
    // groups whose row extent is read on the device (GemmBatch::limit on M): only the leading row panels exist, so they
    // go round-robin over the XCDs (workgroup i runs on XCD i % 8) -- panel 8 j + x with all its column tiles on XCD x
    // -- instead of a contiguous range per XCD, which would put all the live tiles on two or three XCDs
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int rp = (idx / g.tiles_n) * 8 + xcd;
    if (rp >= g.tiles_m) return;
    z = 0;
    m0 = rp * BM;
    n0 = (idx % g.tiles_n) * BN;
  }
  // group of this workgroup and its extent (GemmBatch): rows beyond Mlim do not exist, the contraction ends at Klim
  const int grp = blockIdx.y;
  int Mlim = g.M, Klim = g.K, kps = g.k_per_split;
  if (g.batch.limit != nullptr) {
    const int n = g.batch.limit[grp * g.batch.limit_stride];
    if (g.batch.limit_on_k) {
      Klim = min(Klim, max(n, 0));
      const int per = (Klim + g.splits - 1) / g.splits;
      kps = max(BK, ((per + BK - 1) / BK) * BK);
    } else {
      Mlim = min(Mlim, n);
      if (m0 >= Mlim) return;
    }
  }
  const float* const gA = g.A + (size_t)grp * g.batch.strideA;
  const float* const gB = g.B + (size_t)grp * g.batch.strideB;
  const int ks = z * kps;
  const int ke = max(ks, min(Klim, ks + kps));
  const int nkt = (ke - ks + BK - 1) / BK;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave / CF::WGN) * (TM * 32), wn = (wave % CF::WGN) * (TN * 32);
  const int li = lane & 31, h = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  float4 ra[PA], rb[PB];
  if (nkt > 0) {
    load_tile<A_KC, BM, BK, NT, PA, VEC>(gA, g.zeros, g.lda, Mlim, m0, ks, ke, ra);
    load_tile<B_KC, BN, BK, NT, PB, VEC>(gB, g.zeros, g.ldb, g.N, n0, ks, ke, rb);
    store_tile<A_KC, BM, BK, NT, PA>(lds, ra);
    store_tile<B_KC, BN, BK, NT, PB>(lds + TA, rb);
  }
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nkt;
    if (more) {   // next tile's global loads fly under this tile's MFMAs
      load_tile<A_KC, BM, BK, NT, PA, VEC>(gA, g.zeros, g.lda, Mlim, m0, ks + (kt + 1) * BK, ke, ra);
      load_tile<B_KC, BN, BK, NT, PB, VEC>(gB, g.zeros, g.ldb, g.N, n0, ks + (kt + 1) * BK, ke, rb);
    }
    const float* a_lds = lds + cur * (TA + TB);
    const float* b_lds = a_lds + TA;
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      float fa[TM][4], fb[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) load_frag<A_KC, BM, BK>(a_lds, wm + 32 * i + li, kk, h, fa[i]);
#pragma unroll
      for (int j = 0; j < TN; ++j) load_frag<B_KC, BN, BK>(b_lds, wn + 32 * j + li, kk, h, fb[j]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][t], fb[j][t], acc[i][j], 0, 0, 0);
    }
    if (more) {
      store_tile<A_KC, BM, BK, NT, PA>(lds + (cur ^ 1) * (TA + TB), ra);
      store_tile<B_KC, BN, BK, NT, PB>(lds + (cur ^ 1) * (TA + TB) + TA, rb);
    }
    __syncthreads();
  }
  // epilogue: acc register r of lane l holds C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
  float* C = g.C + (size_t)grp * g.batch.strideC + (size_t)z * g.M * g.ldc;   // slab z of this group (ldc == N for slabs)
  if (g.vecC) {
    // stage 64 output rows at a time through LDS so that every global store is a full 16-byte,
    // row-contiguous access (8x fewer store instructions than the per-register dword stores)
#pragma unroll
    for (int pass = 0; pass < BM / 64; ++pass) {
      if (wm / 64 == pass) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = (wm % 64) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
              lds[row * EPI_LD + wn + 32 * j + li] = acc[i][j][r];
            }
      }
      __syncthreads();
      constexpr int C4 = BN / 4;
#pragma unroll
      for (int q = 0; q < 64 * C4 / NT; ++q) {
        const int f = threadIdx.x + NT * q;
        const int row = f / C4, c4 = f % C4;
        const int grow = m0 + 64 * pass + row, gcol = n0 + 4 * c4;
        if (grow < Mlim && gcol < g.N) {
          const float4 v = *reinterpret_cast<const float4*>(lds + row * EPI_LD + 4 * c4);
          *reinterpret_cast<float4*>(C + (size_t)grow * g.ldc + gcol) = v;
        }
      }
      __syncthreads();
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn + 32 * j + li;
      if (col < g.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < Mlim) C[(size_t)row * g.ldc + col] = acc[i][j][r];
        }
      }
    }
}

// slabs are added in slab order (fixed summation order); VEC = 4: 16-byte accesses, 8 slab loads in flight
template <int VEC>
__global__ void k_splitk_reduce(const float* __restrict__ slab, float* __restrict__ C, int M, int N,
                                int ldc, int splits, size_t group_stride_c) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  const int64_t mn = (int64_t)M * N;
  if (i >= mn) return;
  slab += (size_t)blockIdx.y * splits * mn;       // group blockIdx.y: its own slabs, its own C
  C += (size_t)blockIdx.y * group_stride_c;
  float acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
  int s = 0;
  for (; s + 8 <= splits; s += 8) {
    float v[8][VEC];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* p = slab + (size_t)(s + u) * mn + i;
      if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
      } else {
        v[u][0] = *p;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] += v[u][k];
  }
  for (; s < splits; ++s)
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] += slab[(size_t)s * mn + i + k];
  const int row = (int)(i / N), col = (int)(i - (int64_t)row * N);     // N % VEC == 0: the VEC elements share a row
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(C + (size_t)row * ldc + col) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  } else {
    C[(size_t)row * ldc + col] = acc[0];
  }
}

// 16-byte loads legal and float4 validity all-or-nothing: aligned base, ld % 4 == 0, and the
// contiguous extent (K for a k-contiguous operand, the row count for a row-contiguous one) % 4 == 0.
bool vec_ok(const float* p, int ld, int contiguous_extent) {
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0 && (ld % 4) == 0 && (contiguous_extent % 4) == 0 &&
         contiguous_extent >= 4;
}

template <class CF, bool VEC>
void launch_v(rgcn_ctx* c, bool a_kc, bool b_kc, GemmArgs& g) {
  g.tiles_m = (g.M + CF::BM - 1) / CF::BM;
  g.tiles_n = (g.N + CF::BN - 1) / CF::BN;
  const int gx = g.swizzle == 2 ? ((g.tiles_m + 7) / 8) * 8 * g.tiles_n : g.tiles_m * g.tiles_n * g.splits;
  dim3 grid((unsigned)gx, (unsigned)g.batch.groups), block(CF::NT);
  static_assert((64 * CF::BN / 4) % CF::NT == 0 && CF::TM * 32 <= 64, "epilogue staging geometry");
  if (a_kc && !b_kc) hipLaunchKernelGGL((k_gemm_f32<true, false, VEC, CF>), grid, block, 0, c->stream, g);
  else if (a_kc && b_kc) hipLaunchKernelGGL((k_gemm_f32<true, true, VEC, CF>), grid, block, 0, c->stream, g);
  else hipLaunchKernelGGL((k_gemm_f32<false, false, VEC, CF>), grid, block, 0, c->stream, g);
}
template <class CF>
void launch(rgcn_ctx* c, bool a_kc, bool b_kc, GemmArgs& g, bool vec) {
  if (vec) launch_v<CF, true>(c, a_kc, b_kc, g); else launch_v<CF, false>(c, a_kc, b_kc, g);
}

}  // namespace

rgcn_status gemm_f32(rgcn_ctx* c, const char* tag, bool a_kc, bool b_kc, int M, int N, int K,
                     const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                     int split_k, const GemmBatch* batch, double prof_scale) {
  if (M <= 0 || N <= 0) return RGCN_OK;
  const int groups = batch ? batch->groups : 1;
  if (a_kc == false && b_kc == true) RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "gemm TT form not instantiated");
  const int bk = 16;
  GemmArgs g;
  g.A = A; g.B = B; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb;
  const bool vec = vec_ok(A, lda, a_kc ? K : M) && vec_ok(B, ldb, b_kc ? K : N) &&
                   (!batch || (batch->strideA % 4 == 0 && batch->strideB % 4 == 0));
  g.swizzle = 1;
  g.zeros = c->zeros;
  if (batch) g.batch = *batch;
  if (batch && batch->limit != nullptr && !batch->limit_on_k && split_k <= 1) g.swizzle = 2;
  if (split_k < 1) split_k = 1;
  int kps = (K + split_k - 1) / split_k;
  kps = ((kps + bk - 1) / bk) * bk;
  if (kps < bk) kps = bk;
  split_k = K > 0 ? (K + kps - 1) / kps : 1;
  g.k_per_split = kps;
  g.splits = split_k;
  const bool slabs = split_k > 1;
  if (slabs) {
    if ((size_t)groups * split_k * M * N > c->slab_floats) RGCN_FAIL(c, RGCN_ERR_STATE, "internal: split-K slab too small");
    g.C = c->slab; g.ldc = N;
    g.batch.strideC = (size_t)split_k * M * N;
  } else {
    g.C = C; g.ldc = ldc;
  }
  g.vecC = ((reinterpret_cast<uintptr_t>(g.C) & 15u) == 0 && g.ldc % 4 == 0 && N % 4 == 0) ? 1 : 0;
  {
    // (prof_scale: the share of the launch's M x K extent that exists on the device side -- compacted groups)
    ProfScope ps(c, tag, prof_scale * groups * 4.0 * ((double)M * K + (double)K * N + (double)M * N),
                 prof_scale * groups * 2.0 * M * N * K);
    if (c->gemm_mode != 0) {
      RGCN_HIP(c, gemm_bf16x3_launch(c, c->gemm_mode, a_kc, b_kc, vec, M, N, K, A, lda, B, ldb, g.C, g.ldc,
                                     g.k_per_split, g.splits, g.swizzle, g.vecC, &g.batch));
    } else {
      launch<Cfg<128, 128, 16, 4, 2>>(c, a_kc, b_kc, g, vec);
    }
    RGCN_HIP(c, hipGetLastError());
  }
  if (slabs) {
    const int64_t mn = (int64_t)M * N;
    ProfScope ps(c, "splitk_reduce", 4.0 * groups * mn * (split_k + 1), 0);
    const size_t gsc = batch ? batch->strideC : 0;
    if (N % 4 == 0 && ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(C) & 15u) == 0 && gsc % 4 == 0)
      hipLaunchKernelGGL((k_splitk_reduce<4>), dim3((unsigned)((mn / 4 + 255) / 256), (unsigned)groups), dim3(256), 0,
                         c->stream, c->slab, C, M, N, ldc, split_k, gsc);
    else
      hipLaunchKernelGGL((k_splitk_reduce<1>), dim3((unsigned)((mn + 255) / 256), (unsigned)groups), dim3(256), 0,
                         c->stream, c->slab, C, M, N, ldc, split_k, gsc);
    RGCN_HIP(c, hipGetLastError());
  }
  return RGCN_OK;
}

}  // namespace rgcn
