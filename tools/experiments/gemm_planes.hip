// fp32 GEMM on the bf16 matrix cores from PRE-SPLIT operands ("planes").
//
// Same contractions and the same arithmetic as gemm_bf16x3.hip -- every fp32 operand x is the exact sum of three
// bf16 numbers hi + mid + lo, and a.b is accumulated in fp32 from 6 (or 9) bf16 x bf16 partial products on
// v_mfma_f32_32x32x16_bf16 -- but the split is no longer done inside the GEMM.  In gemm_bf16x3.hip half of the
// kernel's issue slots went to the conversion (5.5 VALU + 1 LDS store per MFMA, profiles/r01_gemm_pmc.log): the
// matrix pipe sat at 42 %.  Here the kernels that PRODUCE an operand (input layer, combine, top-gradient dropout:
// elementwise.hip) emit its three bf16 planes next to the fp32 copy while the values are in registers, the tiny
// W_self is split once per step, and this kernel only moves bytes: global -> LDS by LDS-direct loads
// (global_load_lds_dwordx4, no VGPRs, no VALU, no ds_write), LDS -> MFMA fragments by ds_read_b128 or the
// hardware transpose read.  Reference contractions: self-loop H.W_self (gcn_basis_concat.py:65-66,
// gcn_basis.py:70-71) and its two gradients (tf.gradients, optimization/abstract.py:117-118).
//
// PLANE LAYOUT (one buffer per logical matrix X[rows][K], rows = the long / vertex dimension, K = features):
//     bf16 P[kb = K/16][plane 3][rows_p][16],  rows_p = rows rounded up to 128, zero padded (rows and K),
//     element (r, k) of plane p at ((kb*3 + p)*rows_p + r)*16 + ((k & 15) ^ (8 * ((r >> 4) & 1)))
// i.e. the two 8-element halves of a row's 32 bytes are swapped for rows with bit 4 set.  The SAME buffer feeds
//   FORM 0 (k = features):  a [128 rows] x [16 k] x 3 planes operand tile is three contiguous 4 KB chunks; its LDS
//           image [plane][row][32 B] is read with ds_read_b128 (lane: row l&31, half l>>5); the half swap makes the
//           16 lanes of every ds_read_b128 lane group fall on 16 different 16-byte slots (conflict free);
//   FORM 1 (k = vertices, the dW = H^T.dS contraction): a [16 vertices] x [128 features] x 3 planes operand tile is
//           24 contiguous 512-byte sub-blocks [16 v][16 f]; ds_read_b64_tr_b16 hands every lane the 4 vertices of its
//           feature column (a 16-lane group reads 4 rows x 32 B = 128 contiguous bytes: conflict free).
// Producers write 8-byte pieces (4 features of one row and plane); 8 consecutive rows of one workgroup fill whole
// 128-byte lines.
//
// Kernel: 128 x 128 output tile, 4 waves (2 x 2) of 64 x 64 = four 32x32 accumulators, BK = 16, THREE LDS stages of
// 24 KB (72 KB -> two workgroups per CU).  Step t: issue the 6 LDS-direct loads of tile t+2, read the fragments of
// tile t, 4*TERMS MFMAs, s_waitcnt vmcnt(6) (tile t+1 has landed, tile t+2 stays in flight across the barrier),
// s_barrier.  One barrier per step; no __syncthreads() (it would drain vmcnt).
#include <cstdlib>
#include <type_traits>

#include "rgcn_internal.h"

namespace rgcn {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using lds_s16x4_ptr = __attribute__((address_space(3))) s16x4*;

constexpr int BM = 128, BN = 128, NTH = 256;
constexpr int OPER_BYTES = 3 * 128 * 32;      // one operand tile: 3 planes x 128 rows (features) x 16 bf16
constexpr int STAGE_BYTES = 2 * OPER_BYTES;   // A + B
constexpr int NSTAGE = 3;
constexpr int EPI_LD = BN + 4;
constexpr size_t kLdsBytes = (size_t)NSTAGE * STAGE_BYTES;
static_assert(kLdsBytes >= (size_t)64 * EPI_LD * 4, "the epilogue stages 64 rows of C through the same LDS");

struct PArgs {
  const uint16_t* A;
  const uint16_t* B;
  float* C;            // output, or the slab base when splits > 1 ([splits][M][N], ldc = N)
  int rowsA_p, rowsB_p;
  int M, N;
  int ktiles;          // k-tiles of 16 in total
  int kt_per_split, splits;
  int tiles_m, tiles_n;
  int ldc;
  int swizzle;
  int vecC;
};

__device__ __forceinline__ void glds16(const uint16_t* src, unsigned char* lds_dst) {
  // 16 bytes per lane, LDS destination = wave-uniform base + 16 * lane
  __builtin_amdgcn_global_load_lds(src, lds_dst, 16, 0, 0);
}

template <int FORM, int TERMS>
__global__ void __launch_bounds__(NTH, 2) k_gemm_planes(PArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  int wg = blockIdx.x;
  const int total = g.tiles_m * g.tiles_n * g.splits;
  if (g.swizzle) {       // workgroup b runs on XCD b % 8: give every XCD a contiguous range of tiles
    const int xcd = wg & 7, idx = wg >> 3;
    const int q = total >> 3, r = total & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int per_split = g.tiles_m * g.tiles_n;
  const int z = wg / per_split;
  const int rem = wg - z * per_split;
  const int m0 = (rem / g.tiles_n) * BM, n0 = (rem % g.tiles_n) * BN;
  const int kt0 = z * g.kt_per_split;
  const int nkt = min(g.ktiles - kt0, g.kt_per_split);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int li = lane & 31, h = lane >> 5, s = (lane >> 4) & 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // ---- this wave's three LDS-direct loads per operand and tile: instruction u fills bytes [(wave + 4u) KiB, +1 KiB)
  const uint16_t* srcA[3];
  const uint16_t* srcB[3];
  size_t strideA, strideB;      // elements per k-tile
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int j = wave + 4 * u;
    if constexpr (FORM == 0) {
      const int p = j >> 2, q = j & 3;
      srcA[u] = g.A + ((size_t)(kt0 * 3 + p) * g.rowsA_p + m0) * 16 + q * 512 + lane * 8;
      srcB[u] = g.B + ((size_t)(kt0 * 3 + p) * g.rowsB_p + n0) * 16 + q * 512 + lane * 8;
    } else {
      const int sub = 2 * j + (lane >> 5);          // 0..23 = (feature block kbi, plane p)
      const int kbi = sub / 3, p = sub - 3 * kbi;
      srcA[u] = g.A + ((size_t)((m0 / 16 + kbi) * 3 + p) * g.rowsA_p + (size_t)kt0 * 16) * 16 + (lane & 31) * 8;
      srcB[u] = g.B + ((size_t)((n0 / 16 + kbi) * 3 + p) * g.rowsB_p + (size_t)kt0 * 16) * 16 + (lane & 31) * 8;
    }
  }
  if constexpr (FORM == 0) {
    strideA = (size_t)3 * g.rowsA_p * 16;
    strideB = (size_t)3 * g.rowsB_p * 16;
  } else {
    strideA = strideB = 256;
  }
  int issued = 0;      // tiles issued so far; past the last tile the last one is loaded again (into a stage
                       // nobody reads any more), which keeps the step body branch-free and the vmcnt arithmetic fixed
  auto issue = [&](int stage) {
    unsigned char* base = lds + stage * STAGE_BYTES + wave * 1024;
#pragma unroll
    for (int u = 0; u < 3; ++u) glds16(srcA[u], base + u * 4096);
#pragma unroll
    for (int u = 0; u < 3; ++u) glds16(srcB[u], base + OPER_BYTES + u * 4096);
    ++issued;
    const size_t sa = issued < nkt ? strideA : 0, sb = issued < nkt ? strideB : 0;
#pragma unroll
    for (int u = 0; u < 3; ++u) { srcA[u] += sa; srcB[u] += sb; }
  };

  // ---- fragment offsets (bytes inside an operand tile)
  int offA, offB;
  if constexpr (FORM == 0) {
    offA = (wm + li) * 32 + ((h ^ s) * 16);
    offB = (wn + li) * 32 + ((h ^ s) * 16);
  } else {
    const int c16 = lane & 15;
    const int rq = (8 * h + (c16 >> 2)) * 32 + (c16 & 3) * 8;
    offA = (wm / 16 + s) * 1536 + rq;
    offB = (wn / 16 + s) * 1536 + rq;
  }

  constexpr int NP = 9;
  constexpr int pa_[NP] = {2, 2, 1, 2, 0, 1, 1, 0, 0};     // partial products from the smallest to the largest;
  constexpr int pb_[NP] = {2, 1, 2, 0, 2, 1, 0, 1, 0};     // plane 0 = hi, 1 = mid, 2 = lo

  // LDS -> registers: the 2 x 3 A and 2 x 3 B fragments of the tile in `stage` (k-tile kt_abs)
  auto load_frags = [&](int stage, int kt_abs, bf16x8 (&fa)[2][3], bf16x8 (&fb)[2][3]) {
    const unsigned char* a_lds = lds + stage * STAGE_BYTES;
    const unsigned char* b_lds = a_lds + OPER_BYTES;
    if constexpr (FORM == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          fa[i][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(a_lds + offA + p * 4096 + i * 1024));
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          fb[j][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(b_lds + offB + p * 4096 + j * 1024));
    } else {
      // vertex tile kt_abs starts at row 16 * kt_abs: its rows have bit 4 set iff kt_abs is odd -> halves swapped.
      // The transpose reads are issued from inline asm: through the builtin (__builtin_amdgcn_ds_read_tr16_b64_*)
      // hipcc (ROCm 7.2) puts "s_waitcnt vmcnt(0)" in front of the first one -- it cannot tell that they do not
      // touch the stage the LDS-direct loads just issued are filling -- which drains the load pipeline every step.
      // The compiler does not count asm-issued DS operations: frags_wait() waits for them explicitly and fences the
      // scheduler so that no MFMA is hoisted above the wait (cdna_hip_programming.md, methodology rule 18).
      const int sw = (kt_abs & 1) << 4;
      const uint32_t pa = (uint32_t)(uintptr_t)(a_lds) + (uint32_t)(offA ^ sw);
      const uint32_t pb = (uint32_t)(uintptr_t)(b_lds) + (uint32_t)(offB ^ sw);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          u32x2 a0, a1, b0, b1;
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a0) : "v"(pa), "n"((6 * i + p) * 512));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a1) : "v"(pa), "n"((6 * i + p) * 512 + 128));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(b0) : "v"(pb), "n"((6 * i + p) * 512));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(b1) : "v"(pb), "n"((6 * i + p) * 512 + 128));
          fa[i][p] = __builtin_bit_cast(bf16x8, u32x4{a0[0], a0[1], a1[0], a1[1]});
          fb[i][p] = __builtin_bit_cast(bf16x8, u32x4{b0[0], b0[1], b1[0], b1[1]});
        }
    }
  };
  auto frags_wait = [&]() {
    if constexpr (FORM == 1) {
      __builtin_amdgcn_sched_barrier(0);       // the step's MFMAs stay above the wait ...
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);       // ... and the next step's below it
    }
  };
  // partial products T0 .. T1-1 of the term list (smallest first)
  auto mma = [&](auto t0, auto t1, bf16x8 (&fa)[2][3], bf16x8 (&fb)[2][3]) {
#pragma unroll
    for (int t = NP - TERMS + decltype(t0)::value; t < NP - TERMS + decltype(t1)::value; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa_[t]], fb[j][pb_[t]], acc[i][j], 0, 0, 0);
  };
  using T0 = std::integral_constant<int, 0>;
  using T1 = std::integral_constant<int, 1>;
  using TN = std::integral_constant<int, TERMS>;

  // Software pipeline over k-tiles.  Tile t is multiplied from REGISTERS (fragments read one step ahead, so no MFMA
  // ever waits for LDS); tiles t+1 .. t+3 sit in the three LDS stages, t+1 landed, t+2 and t+3 in flight.
  //   step t:  wait for tile t+1 (vmcnt) + barrier   -> its stage is complete, and the stage of tile t is free
  //            issue the LDS-direct loads of tile t+3 into the stage tile t used
  //            issue the fragment reads of tile t+1 (register set B)
  //            24 MFMAs on tile t (register set A)
  // Two steps per loop iteration so that the register sets alternate statically.
  if (nkt > 0) {
    bf16x8 fa0[2][3], fb0[2][3], fa1[2][3], fb1[2][3];
    issue(0);
    issue(1);
    issue(2);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");     // tile 0 has landed (this wave's part)
    __builtin_amdgcn_s_barrier();                         // ... and everybody else's
    load_frags(0, kt0, fa0, fb0);
    frags_wait();
    int st = 0;                                           // stage of tile t
    auto step = [&](int t, bf16x8 (&fa)[2][3], bf16x8 (&fb)[2][3], bf16x8 (&na)[2][3], bf16x8 (&nb)[2][3]) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // tile t + 1 has landed; tile t + 2 stays in flight
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      issue(st);                                          // tile t + 3 into the stage tile t was read from
      int nst = st + 1;
      if (nst == NSTAGE) nst = 0;
      // The first 4 MFMAs go ahead of the fragment reads: hipcc waits lgkmcnt(0) in front of the first MFMA of a
      // step (loop-carried fragment registers), which must find nothing but long-finished reads outstanding.
      mma(T0{}, T1{}, fa, fb);
      __builtin_amdgcn_sched_barrier(0);
      load_frags(nst, kt0 + t + 1, na, nb);               // (past the last tile: a re-loaded copy, never used)
      __builtin_amdgcn_sched_barrier(0);                  // reads here: hipcc would sink them behind the MFMAs
      mma(T1{}, TN{}, fa, fb);
      frags_wait();
      st = nst;
    };
    int t = 0;
    for (; t + 1 < nkt; t += 2) {
      step(t, fa0, fb0, fa1, fb1);
      step(t + 1, fa1, fb1, fa0, fb0);
    }
    if (t < nkt) step(t, fa0, fb0, fa1, fb1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the re-loaded last tile must not land in the epilogue's LDS
  __builtin_amdgcn_s_barrier();

  // epilogue: acc register r of lane l holds C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
  float* C = g.C + (size_t)z * g.M * g.ldc;
  if (g.vecC) {
    float* stage = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      if (wm == 64 * pass) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
              stage[row * EPI_LD + wn + 32 * j + li] = acc[i][j][r];
            }
      }
      __syncthreads();
      constexpr int C4 = BN / 4;
#pragma unroll
      for (int q = 0; q < 64 * C4 / NTH; ++q) {
        const int f = threadIdx.x + NTH * q;
        const int row = f / C4, c4 = f % C4;
        const int grow = m0 + 64 * pass + row, gcol = n0 + 4 * c4;
        if (grow < g.M && gcol < g.N) {
          const float4 v = *reinterpret_cast<const float4*>(stage + row * EPI_LD + 4 * c4);
          *reinterpret_cast<float4*>(C + (size_t)grow * g.ldc + gcol) = v;
        }
      }
      __syncthreads();
    }
    return;
  }
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

// ---- fp32 -> planes for operands nobody produces in plane form (W_self; the debug / test entry) --------------
// TRANS = 0: X[r][k] (k contiguous, ld), TRANS = 1: X[k][r] (r contiguous, ld).  The grid covers the PADDED extent
// (rows_p x Kp): padding is written as zeros, so a plane buffer needs no other initialisation.
template <int TRANS>
__global__ void __launch_bounds__(256) k_planes_from_f32(const float* __restrict__ X, int ld, int rows, int K,
                                                         uint16_t* __restrict__ P, int rows_p, int kb_n) {
  const int64_t n = (int64_t)rows_p * kb_n * 4;           // one thread per (row, group of 4 k)
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int r, k;
  if constexpr (TRANS == 0) {
    r = (int)(i / (kb_n * 4));
    k = (int)(i % (kb_n * 4)) * 4;
  } else {
    r = (int)(i % rows_p);
    k = (int)(i / rows_p) * 4;
  }
  float v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool ok = r < rows && k + q < K;
    const size_t off = TRANS == 0 ? (size_t)r * ld + k + q : (size_t)(k + q) * ld + r;
    v[q] = ok ? X[off] : 0.0f;
  }
  emit_planes4(P, rows_p, r, k, v[0], v[1], v[2], v[3]);
}

template <int FORM, int TERMS>
hipError_t launch_one(rgcn_ctx* c, const PArgs& g) {
  auto kern = k_gemm_planes<FORM, TERMS>;
  static bool configured = false;     // per instantiation; contexts are single-threaded per process
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e != hipSuccess) return e;
    configured = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(g.tiles_m * g.tiles_n * g.splits)), dim3(NTH), kLdsBytes, c->stream, g);
  return hipGetLastError();
}

template <int FORM>
hipError_t launch_terms(rgcn_ctx* c, int terms, const PArgs& g) {
  if (terms == 9) return launch_one<FORM, 9>(c, g);
  if (terms == 3) return launch_one<FORM, 3>(c, g);
  return launch_one<FORM, 6>(c, g);
}

}  // namespace

size_t planes_elems(int rows, int K) {
  return (size_t)planes_kb(K) * 3 * planes_rows_p(rows) * 16;
}

rgcn_status planes_from_f32(rgcn_ctx* c, const float* X, int ld, int rows, int K, bool transposed, uint16_t* P) {
  const int rows_p = planes_rows_p(rows), kb_n = planes_kb(K);
  const int64_t n = (int64_t)rows_p * kb_n * 4;
  ProfScope ps(c, "split_to_planes", 4.0 * rows * K + 6.0 * rows * K, 0);
  if (transposed)
    hipLaunchKernelGGL((k_planes_from_f32<1>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, X, ld,
                       rows, K, P, rows_p, kb_n);
  else
    hipLaunchKernelGGL((k_planes_from_f32<0>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, X, ld,
                       rows, K, P, rows_p, kb_n);
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

// C[M,N] = A . B from plane buffers.
//   form 0: A = planes of [M rows][K], B = planes of [N rows][K]      (NN with B = W^T, NT with B = W)
//   form 1: A = planes of [K rows][M], B = planes of [K rows][N]      (TN: both operands vertex-major, k = the rows)
// split_k > 1 (form 1) writes partial slabs to c->slab and reduces them in slab order.
rgcn_status gemm_planes(rgcn_ctx* c, const char* tag, int form, int M, int N, int K, const uint16_t* A,
                        const uint16_t* B, float* C, int ldc, int split_k) {
  if (M <= 0 || N <= 0) return RGCN_OK;
  const int terms = c->gemm_mode;
  if (terms != 3 && terms != 6 && terms != 9) RGCN_FAIL(c, RGCN_ERR_STATE, "internal: plane GEMM in a non-split gemm mode");
  static const int swizzle = getenv("RGCN_GEMM_SWIZZLE") ? atoi(getenv("RGCN_GEMM_SWIZZLE")) : 1;
  PArgs g;
  g.A = A; g.B = B; g.M = M; g.N = N;
  g.rowsA_p = planes_rows_p(form == 0 ? M : K);
  g.rowsB_p = planes_rows_p(form == 0 ? N : K);
  g.ktiles = (K + 15) / 16;
  g.tiles_m = (M + BM - 1) / BM;
  g.tiles_n = (N + BN - 1) / BN;
  g.swizzle = swizzle;
  if (split_k < 1) split_k = 1;
  int ktps = (g.ktiles + split_k - 1) / split_k;
  if (ktps < 1) ktps = 1;
  split_k = g.ktiles > 0 ? (g.ktiles + ktps - 1) / ktps : 1;
  g.kt_per_split = ktps;
  g.splits = split_k;
  const bool slabs = split_k > 1;
  if (slabs) {
    if ((size_t)split_k * M * N > c->slab_floats) RGCN_FAIL(c, RGCN_ERR_STATE, "internal: split-K slab too small");
    g.C = c->slab; g.ldc = N;
  } else {
    g.C = C; g.ldc = ldc;
  }
  g.vecC = ((reinterpret_cast<uintptr_t>(g.C) & 15u) == 0 && g.ldc % 4 == 0 && N % 4 == 0) ? 1 : 0;
  {
    ProfScope ps(c, tag, 4.0 * ((double)M * K + (double)K * N + (double)M * N), 2.0 * M * N * K);
    RGCN_HIP(c, form == 0 ? launch_terms<0>(c, terms, g) : launch_terms<1>(c, terms, g));
  }
  if (slabs) RGCN_TRY(splitk_reduce(c, c->slab, C, M, N, ldc, split_k));
  return RGCN_OK;
}

}  // namespace rgcn
