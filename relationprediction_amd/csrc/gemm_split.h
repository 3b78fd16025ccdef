// Shared pieces of the split-arithmetic GEMM kernels (gemm_bf16x3.hip: 128x128 tiles, operands staged through registers;
// gemm_bf16x3_w8.hip: 128x256 tiles, 8 wavefronts, pre-split B by LDS-DMA): vector types, the launch arguments and the
// exact bf16 split of an fp32 pair.  Device code only.
#ifndef RGCN_GEMM_SPLIT_H_
#define RGCN_GEMM_SPLIT_H_

#include <type_traits>
#include <utility>

#include "rgcn_internal.h"

namespace rgcn {
namespace gx {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct XArgs {
  const float* A;
  const float* B;
  float* C;
  const float* zeros;
  int M, N, K;
  int lda, ldb, ldc;
  int k_per_split;
  int tiles_m, tiles_n, splits;
  int swizzle;
  int vecC;
  GemmBatch batch;    // groups (blockIdx.y) and their device-side extents
  const u32x4* bfrag; // B_PRE instantiations: B pre-split into MFMA fragments (k_presplit_b), [ktiles][nt32][3][2][32]
  int nt32;           // 32-column tiles of the fragment table (bfrag_nt32(N))
  uint64_t* tl;       // devtools builds: per-wavefront timeline of k_gemm_w8<.., DBG_TIMELINE>, else nullptr
};

constexpr int BM = 128, BN = 128, BK = 16, NTH = 256;
constexpr int KC_LD = 12;                 // dwords per row of a k-contiguous plane (8 + 4 pad)
constexpr int RC_LD = 136;                // dwords per k-pair row of a row-contiguous plane (128 + 8 pad)
constexpr int KC_PLANE = 128 * KC_LD;     // 1536 dwords
constexpr int RC_PLANE = 8 * RC_LD;       // 1088 dwords
constexpr int EPI_LD = BN + 4;

using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;

// One splitting level for the pair (v0[I0], v1[I1]): returns the bf16 pair nearest to it
// (v_cvt_pk_bf16_f32, first element in the low half) and replaces the two floats by their residuals
// x - bf16(x) (shift / mask / subtract).  bf16(x) agrees with x in its leading 8 significand bits, so the
// difference is exactly representable: no rounding happens at any level and x = hi + mid + lo + (a
// remainder below 2^-26 |x| that is zero unless all three roundings went the same way).
// (v_dot2c_f32_bf16 against the pairs (-1, -0), (-0, -1) computes the same residuals in one instruction,
// but it holds up the matrix pipe for ~10 cycles where a plain VALU op costs 2: tools/mfma_fill.hip.)
// negmask (0 or 0x80000000, first level of the A operand only) splits -x instead of x: see the accumulator sign groups
// in the kernel.
template <int I0, int I1>
__device__ __forceinline__ uint32_t split_level(f32x4& v0, f32x4& v1, uint32_t negmask = 0u) {
  const float x0 = v0[I0], x1 = v1[I1];      // (copies: __builtin_bit_cast of a vector-element lvalue reads element 0)
  const f32x2 f = {__builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x0) ^ negmask),
                   __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x1) ^ negmask)};
  const uint32_t u = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2));
  v0[I0] = f[0] - __builtin_bit_cast(float, u << 16);
  v1[I1] = f[1] - __builtin_bit_cast(float, u & 0xffff0000u);
  return u;
}
__device__ __forceinline__ uint32_t split_last(float x0, float x1) {
  const f32x2 f = {x0, x1};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2));
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait
// for the global loads of the tiles after next that were issued at the top of the step.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// 32-column tiles of a fragment table: the table is padded to whole 256-column tiles so that both kernels can use it
__host__ __device__ constexpr int bfrag_nt32(int N) { return 8 * ((N + 255) / 256); }

}  // namespace gx
}  // namespace rgcn
#endif  // RGCN_GEMM_SPLIT_H_
