// fp32 GEMM evaluated on the bf16 matrix cores by exact operand splitting.
//
// Same contractions as gemm_f32.hip (self-loop H.W_self, gcn_basis_concat.py:65-66 / gcn_basis.py:70-71,
// its two gradients, and the basis contraction), same operand flavours, same fp32 C.
// `v_mfma_f32_32x32x2_f32` runs at 1/16 of the bf16 MFMA rate on gfx950 and there is no TF32/xf32, so the
// fp32 tensor path tops out at 157 TF.  Here every fp32 operand x is split into three bf16 numbers
//     x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)
// (round to nearest; both residuals are exactly representable, so nothing is rounded away: |mid| <=
// 2^-9 |x|, |lo| <= 2^-18 |x|, what is left is below 2^-26 |x| and usually zero), and a.b is the sum of the
// bf16 x bf16 partial products, each exact in fp32 (16-bit significand), accumulated in fp32 by
// `v_mfma_f32_32x32x16_bf16`:
//     TERMS = 9: every partial product        - the products of exact fp32 arithmetic
//     TERMS = 6: without mid.lo, lo.mid, lo.lo - defect <= 2^-26 |a||b| per product, a quarter of one fp32
//                                               rounding of that product; measured error against float64
//                                               equals the fp32 MFMA's (test_gemm_modes_are_fp32_accurate)
// K = 16 of one 32x32 tile costs 6 (9) x 32 cycles against 8 x 64 on the fp32 MFMA.
//
// What the measurements on MI355X say about this kernel's shape (tools/mfma_fill.hip, profiles/):
//   * VALU work is NOT hidden behind the matrix pipe: every plain VALU instruction issued on a SIMD adds
//     ~2 cycles to that SIMD's MFMA stream, whichever wave issues it (v_dot2c_f32_bf16: ~10).  So the
//     split is done ONCE per element, when the tile goes to LDS (11 plain VALU per element pair), not per
//     consuming wave, and its chunks are placed one behind each MFMA so that they at least never stall
//     the wave at a barrier.
//   * __syncthreads() drains vmcnt; the k-loop uses an LDS-only barrier so that the global loads of the
//     tile after next stay in flight across it.
//   * Loads of full tiles go through per-thread walking pointers (one 64-bit add per load and step).
//   * PACKED-FP32 ERRATUM: a v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 issued by ANY wave on a CU where
//     other waves issue v_mfma_f32_32x32x16_bf16 returns wrong low halves (tools/mfma_corun.hip: a plain
//     streaming kernel beside a bf16 MFMA loop, 40 of 40 runs corrupted; none with -packed-fp32-ops or
//     beside fp32 MFMAs).  The encoder runs its HBM-bound kernels on side streams beside this GEMM, so
//     the whole library is built without packed-FP32 instructions (relationprediction_amd/build.py checks
//     the linked code objects).
//
// Tiling: 128x128 output tile, 4 waves (2x2) of 64x64 = four 32x32 accumulators each, BK = 16,
// register-staged double-buffered LDS holding three bf16 planes per operand (<= 72 KB -> two workgroups
// per CU).  LDS layouts, per plane:
//   k-contiguous operand  X[row][k]:  [128 rows][8 dwords + 4 pad]; a lane's 8 consecutive k are one
//        ds_read_b128
//   row-contiguous operand X[k][row]: [8 k-pairs][128 rows + 8 pad] dwords, (k, k+1) packed per dword;
//        the producer writes ds_write_b128 along rows, a lane gathers its 4 k-pairs with four
//        conflict-free ds_read_b32 - no transposition anywhere.
// Lane half h supplies k = 8h .. 8h+7 for BOTH operands, so the k order inside the MFMA is consistent.
#include <cstdlib>

#include "gemm_split.h"

namespace rgcn {

using namespace gx;

namespace {

// ---- global -> registers -------------------------------------------------------------------------
// k-contiguous operand: float4 p of thread t = 4 consecutive k of row (t + 256p)/4.
template <bool VEC>
__device__ __forceinline__ void load_kc(const float* __restrict__ X, const float* __restrict__ zeros,
                                        int ld, int rows, int row0, int k0, int kend, f32x4 (&r)[2]) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int f = threadIdx.x + NTH * p;
    const int row = row0 + (f >> 2), k = k0 + (f & 3) * 4;
    const size_t off = (size_t)row * ld + k;
    if constexpr (VEC) {
      const float* src = (row < rows && k < kend) ? X + off : zeros;
      r[p] = *reinterpret_cast<const f32x4*>(src);
    } else {
      float e[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* src = (row < rows && k + q < kend) ? X + off + q : zeros;
        e[q] = *src;
      }
      r[p] = f32x4{e[0], e[1], e[2], e[3]};
    }
  }
}
// row-contiguous operand: thread t owns k-pair t/32 and rows 4(t%32)..+3; float4 q = k 2kp+q.
template <bool VEC>
__device__ __forceinline__ void load_rc(const float* __restrict__ X, const float* __restrict__ zeros,
                                        int ld, int rows, int row0, int k0, int kend, f32x4 (&r)[2]) {
  const int kp = threadIdx.x >> 5, rg = threadIdx.x & 31;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int k = k0 + 2 * kp + q, row = row0 + 4 * rg;
    const size_t off = (size_t)k * ld + row;
    if constexpr (VEC) {
      const float* src = (k < kend && row < rows) ? X + off : zeros;
      r[q] = *reinterpret_cast<const f32x4*>(src);
    } else {
      float e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* src = (k < kend && row + i < rows) ? X + off + i : zeros;
        e[i] = *src;
      }
      r[q] = f32x4{e[0], e[1], e[2], e[3]};
    }
  }
}

// ---- registers -> split -> LDS planes (hi, mid, lo), in 12 small chunks per operand ----------------
// The conversion of one operand's staged tile (two float4 per thread) is cut into chunks of <= 3 VALU
// (+ one LDS store) so that the k-loop can place one chunk behind each MFMA.  Chunk C of:
//   k-contiguous operand, float4 q = C / 6, c = C % 6:
//     c0 hi(x,y)  c1 hi(z,w) + store plane 0  c2 mid(x,y)  c3 mid(z,w) + store plane 1
//     c4 lo(x,y), lo(z,w) + store plane 2     c5 -
//   row-contiguous operand, (even-k, odd-k) float4 pair, level = C / 4 for C < 8:
//     C0..3 hi of row 0..3 (+ store plane 0 at C3)   C4..7 mid (+ store plane 1 at C7)
//     C8 lo rows 0,1   C9 lo rows 2,3 + store plane 2   C10, C11 -
struct SplitRegs {
  u32x4 w;       // packed words waiting for their store
};
template <bool KC, int PLANE, int C>
__device__ __forceinline__ void split_chunk(f32x4 (&r)[2], SplitRegs& st, uint32_t* __restrict__ lds,
                                            uint32_t negmask) {
  if constexpr (KC) {
    constexpr int q = C / 6, c = C % 6;
    const int f = threadIdx.x + NTH * q;
    uint32_t* dst = lds + (f >> 2) * KC_LD + (f & 3) * 2;
    f32x4& v = r[q];
    if constexpr (c == 0 || c == 2) st.w[0] = split_level<0, 1>(v, v, c == 0 ? negmask : 0u);
    if constexpr (c == 1 || c == 3) {
      st.w[1] = split_level<2, 3>(v, v, c == 1 ? negmask : 0u);
      *reinterpret_cast<uint2*>(dst + (c / 2) * PLANE) = make_uint2(st.w[0], st.w[1]);
    }
    if constexpr (c == 4)
      *reinterpret_cast<uint2*>(dst + 2 * PLANE) = make_uint2(split_last(v[0], v[1]), split_last(v[2], v[3]));
  } else {
    const int kp = threadIdx.x >> 5, rg = threadIdx.x & 31;
    uint32_t* dst = lds + kp * RC_LD + 4 * rg;
    f32x4& e = r[0];
    f32x4& o = r[1];
    if constexpr (C < 8) {
      constexpr int i = C % 4;
      st.w[i] = split_level<i, i>(e, o, C < 4 ? negmask : 0u);
      if constexpr (i == 3) *reinterpret_cast<u32x4*>(dst + (C / 4) * PLANE) = st.w;
    }
    if constexpr (C == 8) { st.w[0] = split_last(e[0], o[0]); st.w[1] = split_last(e[1], o[1]); }
    if constexpr (C == 9) {
      st.w[2] = split_last(e[2], o[2]); st.w[3] = split_last(e[3], o[3]);
      *reinterpret_cast<u32x4*>(dst + 2 * PLANE) = st.w;
    }
  }
}
template <bool KC, int PLANE, int C0, int C1>
__device__ __forceinline__ void split_chunks(f32x4 (&r)[2], SplitRegs& st, uint32_t* __restrict__ lds,
                                             uint32_t negmask = 0u) {
  if constexpr (C0 < C1) {
    split_chunk<KC, PLANE, C0>(r, st, lds, negmask);
    split_chunks<KC, PLANE, C0 + 1, C1>(r, st, lds, negmask);
  }
}

// ---- LDS -> MFMA fragment: the 8 k values 8h..8h+7 of row `row`, one plane ------------------------
template <bool KC>
__device__ __forceinline__ bf16x8 load_frag(const uint32_t* __restrict__ plane, int row, int h) {
  u32x4 v;
  if constexpr (KC) {
    v = *reinterpret_cast<const u32x4*>(plane + row * KC_LD + 4 * h);
  } else {
    const uint32_t* p = plane + (4 * h) * RC_LD + row;
    v[0] = p[0]; v[1] = p[RC_LD]; v[2] = p[2 * RC_LD]; v[3] = p[3 * RC_LD];
  }
  return __builtin_bit_cast(bf16x8, v);
}

// B_PRE: the B operand is a WEIGHT that k_presplit_b has already split into its three bf16 planes, stored in the order
// the MFMA wants its fragments (GemmArgs::bfrag).  Each wave then fetches the 2 x 3 fragments of a k-tile straight from
// L2 into registers with six coalesced 16-byte loads, one k-tile ahead: no global -> register -> split -> LDS -> register
// round trip for B, i.e. half the LDS reads and writes of a step, half its split VALU work and half the LDS footprint;
// A (the activations) keeps the staged path.  Same products in the same order: bitwise the plain kernel's result.
// (Measured, profiles/r05_gemm_presplit.md: 5-12 % on the NN form, 0-6 % on NT; letting the two waves that own the same 64
// columns fetch half of the fragments each and swap them through LDS -- half the L1 requests -- changed nothing.)
template <bool A_KC, bool B_KC, bool VEC, int TERMS, bool B_PRE = false>
__global__ void __launch_bounds__(NTH, 2) k_gemm_bf16x3(XArgs g) {
  constexpr int PA = A_KC ? KC_PLANE : RC_PLANE, PB = B_KC ? KC_PLANE : RC_PLANE;
  constexpr int TA = 3 * PA, TB = B_PRE ? 0 : 3 * PB;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];

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
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int li = lane & 31, h = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // Software pipeline, one k-tile per step: the global loads of tile t+2 are issued at the top, tile t
  // (in LDS buffer t&1) feeds the MFMAs, and the split + LDS store of tile t+1 (in registers since the
  // previous step) is placed chunk by chunk behind those MFMAs (order pinned by scheduling fences).
  // Tiles past the end of K load zeros from the zero buffer; the step body is branch-free.
  f32x4 ra[2][2], rb[2][2];
  auto gload = [&](int k0, f32x4 (&xa)[2], f32x4 (&xb)[2]) {
    if constexpr (A_KC) load_kc<VEC>(gA, g.zeros, g.lda, Mlim, m0, k0, ke, xa);
    else load_rc<VEC>(gA, g.zeros, g.lda, Mlim, m0, k0, ke, xa);
    if constexpr (!B_PRE) {
      if constexpr (B_KC) load_kc<VEC>(gB, g.zeros, g.ldb, g.N, n0, k0, ke, xb);
      else load_rc<VEC>(gB, g.zeros, g.ldb, g.N, n0, k0, ke, xb);
    }
  };
  // Full tiles of a 16-byte-loadable operand are fetched through per-thread walking pointers (one
  // 64-bit add per load and step instead of the whole address + validity computation); a thread whose
  // rows lie outside the matrix walks on the zero buffer with stride 0.
  const float* wa[2];
  const float* wb[2];
  int ia[2] = {0, 0}, ib[2] = {0, 0};          // pointer advance per k-tile, in floats
  if constexpr (VEC) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if constexpr (A_KC) {
        const int f = threadIdx.x + NTH * p, row = m0 + (f >> 2);
        wa[p] = row < Mlim ? gA + (size_t)row * g.lda + ks + (f & 3) * 4 : g.zeros;
        ia[p] = row < Mlim ? BK : 0;
      } else {
        const int row = m0 + 4 * (threadIdx.x & 31), k = ks + 2 * (threadIdx.x >> 5) + p;
        wa[p] = row < Mlim ? gA + (size_t)k * g.lda + row : g.zeros;
        ia[p] = row < Mlim ? BK * g.lda : 0;
      }
      if constexpr (B_PRE) {
        wb[p] = g.zeros;
      } else if constexpr (B_KC) {
        const int f = threadIdx.x + NTH * p, row = n0 + (f >> 2);
        wb[p] = row < g.N ? gB + (size_t)row * g.ldb + ks + (f & 3) * 4 : g.zeros;
        ib[p] = row < g.N ? BK : 0;
      } else {
        const int row = n0 + 4 * (threadIdx.x & 31), k = ks + 2 * (threadIdx.x >> 5) + p;
        wb[p] = row < g.N ? gB + (size_t)k * g.ldb + row : g.zeros;
        ib[p] = row < g.N ? BK * g.ldb : 0;
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) { wa[p] += 2 * (size_t)ia[p]; wb[p] += 2 * (size_t)ib[p]; }   // first walked tile = 2
  }
  auto gwalk = [&](f32x4 (&xa)[2], f32x4 (&xb)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      xa[p] = *reinterpret_cast<const f32x4*>(wa[p]);
      wa[p] += ia[p];
      if constexpr (!B_PRE) {
        xb[p] = *reinterpret_cast<const f32x4*>(wb[p]);
        wb[p] += ib[p];
      }
    }
  };
  auto sstore = [&](uint32_t* buf, f32x4 (&xa)[2], f32x4 (&xb)[2]) {
    SplitRegs st;
    split_chunks<A_KC, PA, 0, 12>(xa, st, buf);
    if constexpr (!B_PRE) split_chunks<B_KC, PB, 0, 12>(xb, st, buf + TA);
  };
  // B_PRE: this lane's fragments of k-tile kt_abs (two 32-column tiles x three planes), one k-tile ahead of their use
  const int nkt_all = (g.K + BK - 1) / BK;
  const u32x4* const bq = B_PRE ? g.bfrag + (size_t)grp * g.batch.strideBfrag +
                                      ((size_t)((n0 + wn) >> 5) * 6 + h) * 32 + li
                                : nullptr;
  bf16x8 fbp[2][2][3];
  auto bfetch = [&](int kt_abs, bf16x8 (&f)[2][3]) {
    const u32x4* q = bq + (size_t)min(kt_abs, nkt_all - 1) * g.nt32 * 192;      // (a tile past the end meets a zero A tile)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) f[j][pl] = __builtin_bit_cast(bf16x8, q[(j * 3 + pl) * 64]);
  };
  // Accumulator sign groups.  v_mfma_f32_32x32x16_bf16 adds products that are small beside the accumulator with a
  // bias toward -inf: about -0.004 ulp of the accumulator per MFMA whose products are 2^-9 .. 2^-24 of it, none for
  // comparable or negligible ones (tools/mfma_bias_probe.hip).  Five of the six MFMAs of a k-slice are such, so the
  // error of this GEMM, although as small as the fp32 MFMA's element by element, had a MEAN of -0.08 of its rms, and a
  // sum over ~10^6 of its outputs with non-negative weights (the basis-coefficient gradient) sat 10x further from
  // float64 than the fp32 MFMA's (tools/gemm_bias_probe.py, test_float64_tie_break).  The bias does not depend on the
  // sign of what is accumulated, so the k-tiles are cut into four groups and groups 1 and 3 accumulate the NEGATED
  // product: A goes to LDS negated (one xor per element at the split) and the accumulators change sign at the three
  // group boundaries and at the end -- 256 VALU per wave and GEMM tile.  The boundaries balance the two signs for
  // accumulators that stay level, grow like sqrt(k) (random walk) or linearly (a mean): with f = (0.173, 0.5, 0.849),
  // 2 (f1^p - f2^p + f3^p) - 1 = 0.04, 0.00, 0.00 for p = 1, 1.5, 2.
  int flip[3] = {1 << 30, 1 << 30, 1 << 30};
  if (nkt >= 16) {
    flip[0] = max(2, (int)(0.173f * nkt + 0.5f) & ~1);
    flip[1] = (nkt / 2 + 1) & ~1;
    flip[2] = (int)(0.849f * nkt + 0.5f) & ~1;
  }
  auto sign_group = [&](int t) { return (t >= flip[0] ? 1 : 0) + (t >= flip[1] ? 1 : 0) + (t >= flip[2] ? 1 : 0); };
  auto negate_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f - acc[i][j][r];      // (0 - x: zeros stay +0)
  };
  auto step = [&](int kt, auto parity, auto walk) {
    constexpr int P = decltype(parity)::value;
    constexpr bool WALK = decltype(walk)::value;
    if (kt == flip[0] || kt == flip[1] || kt == flip[2]) negate_acc();      // tile kt opens a group of the other sign
    const uint32_t negmask = (sign_group(kt + 1) & 1) ? 0x80000000u : 0u;   // sign of the tile this step stages
    if constexpr (WALK) gwalk(ra[P], rb[P]); else gload(ks + (kt + 2) * BK, ra[P], rb[P]);
    if constexpr (B_PRE) bfetch(kt + 1, fbp[1 - P]);
    const uint32_t* a_lds = lds + P * (TA + TB);
    const uint32_t* b_lds = a_lds + TA;
    uint32_t* nxt = lds + (1 - P) * (TA + TB);
    bf16x8 fa[2][3], fb[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) fa[i][pl] = load_frag<A_KC>(a_lds + pl * PA, wm + 32 * i + li, h);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        if constexpr (B_PRE) fb[j][pl] = fbp[P][j][pl];
        else fb[j][pl] = load_frag<B_KC>(b_lds + pl * PB, wn + 32 * j + li, h);
      }
    __builtin_amdgcn_sched_barrier(0);
    // partial products from the smallest to the largest; plane 0 = hi, 1 = mid, 2 = lo.  Behind MFMA m
    // goes conversion chunk m of the NEXT tile (24 chunks; order pinned by the scheduling fences).
    constexpr int NP = 9;
    constexpr int pa_[NP] = {2, 2, 1, 2, 0, 1, 1, 0, 0};
    constexpr int pb_[NP] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
    constexpr int NM = 4 * TERMS;                       // MFMAs per step
    constexpr int NCH = B_PRE ? 12 : 24;                // conversion chunks of a step (A alone when B comes pre-split)
    constexpr int PER = (NCH + NM - 1) / NM;            // chunks behind each MFMA
    SplitRegs st;
    auto weave = [&](auto mi) {
      constexpr int m = decltype(mi)::value;
      constexpr int t = NP - TERMS + m / 4, i = (m / 2) % 2, j = m % 2;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa_[t]], fb[j][pb_[t]], acc[i][j], 0, 0, 0);
      // (B_PRE: the twelve chunks of A behind every other MFMA, as far apart as before)
      constexpr int c0 = B_PRE ? (m % 2 == 0 ? m / 2 : 12) : (m * PER < 24 ? m * PER : 24);
      constexpr int c1 = B_PRE ? (m % 2 == 0 ? (m / 2 + 1 < 12 ? m / 2 + 1 : 12) : 12) : ((m + 1) * PER < 24 ? (m + 1) * PER : 24);
      split_chunks<A_KC, PA, (c0 < 12 ? c0 : 12), (c1 < 12 ? c1 : 12)>(ra[1 - P], st, nxt, negmask);
      if constexpr (!B_PRE)
        split_chunks<B_KC, PB, (c0 > 12 ? c0 - 12 : 0), (c1 > 12 ? c1 - 12 : 0)>(rb[1 - P], st, nxt + TA);
      __builtin_amdgcn_sched_barrier(0);
    };
    static_for<0, NM>(weave);
    lds_barrier();
  };

  if (nkt > 0) {
    gload(ks, ra[0], rb[0]);
    if constexpr (B_PRE) bfetch(0, fbp[0]);
    sstore(lds, ra[0], rb[0]);
    gload(ks + BK, ra[1], rb[1]);
  }
  lds_barrier();
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  int kt = 0;
  if constexpr (VEC) {
    // steps whose prefetch (tile kt + 2) is a full tile: walking pointers
    const int nwalk = ((ke - ks) / BK - 2) & ~1;      // even count, so the parity pattern continues below
    for (; kt < nwalk; kt += 2) {
      step(kt, I0{}, std::true_type{});
      step(kt + 1, I1{}, std::true_type{});
    }
  }
  for (; kt < nkt; kt += 2) {
    step(kt, I0{}, std::false_type{});
    if (kt + 1 < nkt) step(kt + 1, I1{}, std::false_type{});
  }
  if (nkt > 0 && (sign_group(nkt - 1) & 1)) negate_acc();

  // epilogue: acc register r of lane l holds C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
  float* C = g.C + (size_t)grp * g.batch.strideC + (size_t)z * g.M * g.ldc;
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
        if (grow < Mlim && gcol < g.N) {
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
          if (row < Mlim) C[(size_t)row * g.ldc + col] = acc[i][j][r];
        }
      }
    }
}

constexpr size_t lds_bytes(bool a_kc, bool b_kc, bool b_pre = false) {
  const size_t tiles = 2 * 3 * ((a_kc ? KC_PLANE : RC_PLANE) + (b_pre ? 0 : (b_kc ? KC_PLANE : RC_PLANE))) * 4;
  const size_t epi = 64 * EPI_LD * 4;       // 64 staged rows of the product
  return tiles > epi ? tiles : epi;
}

template <bool A_KC, bool B_KC, bool VEC, int TERMS, bool B_PRE = false>
hipError_t launch_one(rgcn_ctx* c, const XArgs& g) {
  constexpr size_t bytes = lds_bytes(A_KC, B_KC, B_PRE);
  auto kern = k_gemm_bf16x3<A_KC, B_KC, VEC, TERMS, B_PRE>;
  static bool configured = false;     // per instantiation; contexts are single-threaded per process
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    configured = true;
  }
  const int gx = g.swizzle == 2 ? ((g.tiles_m + 7) / 8) * 8 * g.tiles_n : g.tiles_m * g.tiles_n * g.splits;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)g.batch.groups), dim3(NTH), bytes, c->stream, g);
  return hipGetLastError();
}

template <bool VEC, int TERMS>
hipError_t launch_form(rgcn_ctx* c, bool a_kc, bool b_kc, const XArgs& g) {
  if constexpr (VEC) {      // pre-split B: k-contiguous A with 16-byte rows, no split over K (the caller checks)
    if (g.bfrag != nullptr && a_kc) return launch_one<true, true, true, TERMS, true>(c, g);
  }
  if (a_kc && !b_kc) return launch_one<true, false, VEC, TERMS>(c, g);
  if (a_kc && b_kc) return launch_one<true, true, VEC, TERMS>(c, g);
  return launch_one<false, false, VEC, TERMS>(c, g);
}

// B (k, n) of a contraction -- stored [n][k] (b_kc) or [k][n] -- split ONCE into the three bf16 planes, laid out so that
// lane (li, h) of the wave that owns 32-column tile nt finds the eight k = 16 kt + 8 h .. + 7 of column 32 nt + li of
// plane p in ONE aligned 16-byte word: F[kt][nt][p][h][li].  Zero beyond K and N.  The arithmetic is split_level's, the
// word order load_frag's: the B_PRE kernel feeds the MFMAs bit for bit what the staged path does.  Several tables per
// launch (blockIdx.y): a train step rebuilds every weight's tables -- W_self of every layer in both orientations, the
// basis tensors -- with one launch when the optimizer has moved the weights.
struct PresplitJobs {
  PresplitJob j[8];
};
__global__ void __launch_bounds__(256) k_presplit_b(PresplitJobs jobs) {
  const PresplitJob job = jobs.j[blockIdx.y];
  const int ktiles = (job.K + BK - 1) / BK, nt32 = bfrag_nt32(job.N);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ktiles * nt32 * 64) return;
  const float* __restrict__ B = job.B;
  const int li = i & 31, h = (i >> 5) & 1, nt = (i >> 6) % nt32, kt = (i >> 6) / nt32;
  const int n = 32 * nt + li, k0 = 16 * kt + 8 * h;
  f32x4 lo4 = {0.f, 0.f, 0.f, 0.f}, hi4 = {0.f, 0.f, 0.f, 0.f};      // k0 .. k0 + 3, k0 + 4 .. k0 + 7
  if (n < job.N) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = k0 + q;
      const float x = k < job.K ? (job.b_kc ? B[(size_t)n * job.ldb + k] : B[(size_t)k * job.ldb + n]) : 0.0f;
      if (q < 4) lo4[q] = x; else hi4[q - 4] = x;
    }
  }
  u32x4* dst = reinterpret_cast<u32x4*>(job.F) + ((size_t)(kt * nt32 + nt) * 6 + h) * 32 + li;
#pragma unroll
  for (int level = 0; level < 3; ++level) {
    u32x4 w;
    if (level < 2) {
      w[0] = split_level<0, 1>(lo4, lo4); w[1] = split_level<2, 3>(lo4, lo4);
      w[2] = split_level<0, 1>(hi4, hi4); w[3] = split_level<2, 3>(hi4, hi4);
    } else {
      w[0] = split_last(lo4[0], lo4[1]); w[1] = split_last(lo4[2], lo4[3]);
      w[2] = split_last(hi4[0], hi4[1]); w[3] = split_last(hi4[2], hi4[3]);
    }
    dst[level * 64] = w;
  }
}

}  // namespace

size_t gemm_bfrag_words(int K, int N) {      // 16-byte words of one operand's fragment table
  return (size_t)((K + BK - 1) / BK) * bfrag_nt32(N) * 192;
}

// the fragment tables of n operands (PresplitJob: B (k, n), its storage form and leading dimension, the table), eight per launch
rgcn_status gemm_presplit_b(rgcn_ctx* c, const PresplitJob* jobs, int n) {
  for (int j0 = 0; j0 < n; j0 += 8) {
    PresplitJobs pj;
    int nj = 0, max_threads = 0;
    double bytes = 0;
    for (; nj < 8 && j0 + nj < n; ++nj) {
      pj.j[nj] = jobs[j0 + nj];
      const PresplitJob& q = pj.j[nj];
      const int threads = ((q.K + BK - 1) / BK) * bfrag_nt32(q.N) * 64;
      if (threads > max_threads) max_threads = threads;
      bytes += 4.0 * q.K * q.N + 16.0 * 3 * threads;
    }
    for (int k = nj; k < 8; ++k) pj.j[k] = pj.j[0];
    ProfScope ps(c, "gemm_presplit_b", bytes, 0);
    hipLaunchKernelGGL(k_presplit_b, dim3((unsigned)((max_threads + 255) / 256), (unsigned)nj), dim3(256), 0, c->stream, pj);
    RGCN_HIP(c, hipGetLastError());
  }
  return RGCN_OK;
}

// Called by gemm_f32() when the context's gemm mode asks for the split evaluation; same contract.
hipError_t gemm_bf16x3_launch(rgcn_ctx* c, int terms, bool a_kc, bool b_kc, bool vec, int M, int N, int K,
                              const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                              int k_per_split, int splits, int swizzle, int vecC, const GemmBatch* batch) {
  XArgs g;
  if (batch) g.batch = *batch;
  g.bfrag = (batch && batch->bfrag && splits == 1 && a_kc && vec) ? reinterpret_cast<const u32x4*>(batch->bfrag) : nullptr;
  g.A = A; g.B = B; g.C = C; g.zeros = c->zeros;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.k_per_split = k_per_split; g.splits = splits; g.swizzle = swizzle; g.vecC = vecC;
  g.tiles_m = (M + BM - 1) / BM;
  g.tiles_n = (N + BN - 1) / BN;
  g.nt32 = bfrag_nt32(N);
  // a weight on the B side: the eight-wavefront kernel (gemm_bf16x3_w8.hip), bitwise the same product
  // (devtools knob RGCN_GEMM_W8: 0 never, 1 where the call site asks for it, 2 / 3 everywhere, >= 1000 the lab's variants)
  const int w8 = knob("RGCN_GEMM_W8", 1);
  bool wide = g.batch.wide != 0;
  if (wide && g.batch.limit == nullptr) {
    // one workgroup per CU and nothing to hide a tile's fill and its stores behind: the wide kernel wins when the launch is ONE
    // round of tiles that fills most of the chip (FB15k-237: 228 tiles, 45.8 against 48.8 us), and loses to the two-per-CU
    // kernel over several rounds (WN18, 640 tiles: 125 against 114 us) -- profiles/r06_gemm_w8.md
    const long t = (long)((M + 127) / 128) * ((N + 255) / 256) * g.batch.groups;
    wide = t <= 256 && t >= 160;
  }
  if (g.bfrag != nullptr && (terms == 6 || terms == 9) && (w8 >= 2 || (w8 == 1 && wide)))
    return gemm_bf16x3_w8_launch(c, terms, M, N, K, A, lda, C, ldc, swizzle, vecC, g.batch);
  if (terms == 9) return vec ? launch_form<true, 9>(c, a_kc, b_kc, g) : launch_form<false, 9>(c, a_kc, b_kc, g);
  if (terms == 3) return vec ? launch_form<true, 3>(c, a_kc, b_kc, g) : launch_form<false, 3>(c, a_kc, b_kc, g);
  return vec ? launch_form<true, 6>(c, a_kc, b_kc, g) : launch_form<false, 6>(c, a_kc, b_kc, g);
}

}  // namespace rgcn
