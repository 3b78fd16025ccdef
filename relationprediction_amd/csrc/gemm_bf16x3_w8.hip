// The split-arithmetic GEMM for a WEIGHT on the B side, eight wavefronts per workgroup (round 6).
//
// Same contraction, same arithmetic and the same order of partial products per output element as k_gemm_bf16x3<.., B_PRE>
// (gemm_bf16x3.hip: x = hi + mid + lo in bf16, six or nine bf16 MFMAs per 32x32x16 block, accumulator sign groups) --
// bitwise the same C -- for the products whose B operand is a weight that k_presplit_b has laid out in fragment order:
// S = H.W_self and G = dS.W_self^T (gcn_basis_concat.py:65-66, gcn_basis.py:70-71 and their gradients), Zc.W' and
// Dc.W'^T of the basis kind (gcn_basis.py:39-46).
//
// What round 5's ablation said about the 128x128 / 4-wave kernel (profiles/r05_gemm_presplit.md): not bound by the
// matrix pipe (22.5 us of 47), bound by getting a k-tile from L2 into fragment form -- 467 MB of L2 -> L1 traffic for a
// 60 MB problem (every wave pulled its own B fragments, every column tile re-read and re-split A four times), B one
// k-tile ahead only because the fragments in flight were registers.  This kernel changes the structure instead of tuning it:
//   * tile 128 x 256, 512 threads = 8 wavefronts (2 x 4) of 64 x 64, ONE workgroup per CU, two wavefronts per SIMD:
//     A is read and split twice instead of four times (N = 500: two column tiles), B's fragments are fetched once per
//     workgroup instead of once per wavefront  =>  231 MB of L2 -> L1 traffic instead of 467;
//   * B goes L2 -> LDS by LDS-DMA (global_load_lds_dwordx4): the table is already in fragment order, so the 24 KB of a
//     k-tile are 24 lane-linear 1 KB pieces (three per wavefront), no registers, no VALU, no ds_write;
//   * a ring of FOUR k-tile stages in LDS (4 x 36.5 KB): the loads of stage s are issued four steps before its MFMAs,
//     A (one float4 per thread and stage) is split into its bf16 planes two steps ahead, the fragments of stage s are
//     read from LDS into a second register set during the MFMAs of stage s - 1 -- a step opens with MFMAs, never with
//     a round of ds_reads behind the barrier;
//   * every VMEM operation of the loop is inline asm and counted by hand: ONE `s_waitcnt vmcnt(8)` per step (the two
//     youngest steps' loads stay in flight across the barriers), one LDS-only barrier per step.
// Pipeline of stage s:   step s-4: DMA(B) + load(A)   step s-2: vmcnt, split A -> LDS, barrier
//                        step s-1: fragments LDS -> registers, barrier      step s: 24 (36) MFMAs
#include "gemm_split.h"

namespace rgcn {

using namespace gx;

namespace {

constexpr int WBM = 128, WBN = 256, WNTH = 512;
constexpr int A_HS = 2048 + 64;           // bytes between the k-half blocks (h = 0, 1) of a plane: [h][128 rows] x 16 B, the
                                          // 64 bytes keep the two ds_write_b64 of a row's halves on different banks
constexpr int A_PL = 2 * A_HS;            // one plane of A: 4224 B
constexpr int A_SZ = 12800;               // three planes (12672 B), rounded up
constexpr int B_SZ = 24 * 1024;           // 8 column tiles x 3 planes x 1 KB, the order of the fragment table
constexpr int ST_SZ = A_SZ + B_SZ;        // one stage: 37376 B
constexpr int W_LDS_BYTES = 4 * ST_SZ;    // 149504 B
constexpr int WEPI_LD = WBN + 4;          // floats per staged row of the product

// loads the compiler must not count (it would drain them at the next ordinary use): destination registers are named by
// the wait statement that retires them
__device__ __forceinline__ void aload(f32x4& dst, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
// the same from a wave-uniform base + a per-lane byte offset: the base walks on the scalar unit, no VALU per stage
__device__ __forceinline__ void aload_s(f32x4& dst, uint32_t voff, const float* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
// DBG (devtools builds only; 0 in the product): ablation switches and the per-wavefront timeline of tools/gemm_w8_lab.py
enum : int { DBG_NO_MFMA = 1, DBG_NO_SPLIT = 2, DBG_NO_FRAG = 4, DBG_NO_DMA = 8, DBG_NO_ALOAD = 16, DBG_NO_STORE = 32,
              DBG_TIMELINE = 64, DBG_FINE = 128, DBG_NT_STORE = 256 };
constexpr int TL_SLOTS = 24;      // 8-byte stamps per wavefront

// one 1 KB piece (PIECE = 0, 1, 2: the immediate offset moves the source and the destination alike)
template <int PIECE>
__device__ __forceinline__ void dma1(uint32_t voff, const u32x4* sbase, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:%4\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_addr), "i"(PIECE * 1024)
      : "memory");
}

// (192 registers, not the 193 the allocator would take: two wavefronts of this kernel then leave a SIMD 128 registers, one
// wavefront of the 126-128-register layer kernels)
template <int TERMS, int DBG = 0>
__global__ void __launch_bounds__(WNTH, 2) __attribute__((amdgpu_num_vgpr(192))) k_gemm_w8(XArgs g) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];

  // ---- the tile of this workgroup (tiles_n counts 256-column tiles here)
  int wg = blockIdx.x;
  const int total = g.tiles_m * g.tiles_n;
  if (g.swizzle == 1) {
    const int xcd = wg & 7, idx = wg >> 3;
    const int q = total >> 3, r = total & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int m0 = (wg / g.tiles_n) * WBM, n0 = (wg % g.tiles_n) * WBN;
  if (g.swizzle == 2) {      // device-side row extent: the leading row panels go round-robin over the XCDs
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int rp = (idx / g.tiles_n) * 8 + xcd;
    if (rp >= g.tiles_m) return;
    m0 = rp * WBM;
    n0 = (idx % g.tiles_n) * WBN;
  }
  const int grp = blockIdx.y;
  int Mlim = g.M;
  if (g.batch.limit != nullptr) {
    Mlim = min(Mlim, g.batch.limit[grp * g.batch.limit_stride]);
    if (m0 >= Mlim) return;
  }
  const float* const gA = g.A + (size_t)grp * g.batch.strideA;
  const int nkt = (g.K + BK - 1) / BK;       // k-tiles
  const int nfull = g.K / BK;                // of which full

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = (wave >> 2) * 64, wn = (wave & 3) * 64;
  const int li = lane & 31, h = lane >> 5;
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
  uint64_t tl[TL_SLOTS];
  auto stamp = [&](auto slot) {
    if constexpr ((DBG & (DBG_TIMELINE | DBG_FINE)) != 0) tl[decltype(slot)::value] = __builtin_amdgcn_s_memtime();
  };
  if constexpr ((DBG & (DBG_TIMELINE | DBG_FINE)) != 0) {
#pragma unroll
    for (int i = 0; i < TL_SLOTS; ++i) tl[i] = 0;
  }
  stamp(std::integral_constant<int, 0>{});

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // ---- A: thread f owns the four k = 4 (f & 3) .. + 3 of row f >> 2 of every stage
  const int arow = threadIdx.x >> 2, aq = threadIdx.x & 3;
  // (rows past the end of the matrix re-read its last row: finite values whose products are never stored)
  const float* sa = gA + (size_t)m0 * g.lda;            // wave-uniform, + BK floats per stage
  const uint32_t a_off = ((uint32_t)min(arow, Mlim - 1 - m0) * (uint32_t)g.lda + 4u * aq) * 4u;
  const bool atail_ok = 4 * aq < (g.K & (BK - 1));      // this thread's float4 exists in a partial last tile
  f32x4 ra[4];
  // its place in a stage: plane p at p * A_PL
  const uint32_t a_wr = (aq >> 1) * A_HS + arow * 16 + (aq & 1) * 8;
  // ---- B: wavefront w moves the pieces 3 w .. 3 w + 2 (column tile w of the eight, its three planes)
  const u32x4* sb = g.bfrag + (size_t)grp * g.batch.strideBfrag + ((size_t)(n0 >> 5) + wave) * 192;
  const size_t sb_step = (size_t)g.nt32 * 192;
  const u32x4* const szero = reinterpret_cast<const u32x4*>(g.zeros);
  const uint32_t voff = lane * 16;
  const uint32_t b_wr = A_SZ + wave * 3072;
  // ---- fragments: lane (li, h) reads 16 bytes per plane
  const uint32_t a_rd = h * A_HS + (wm + li) * 16;                         // + 512 for i = 1, + p * A_PL
  const uint32_t b_rd = A_SZ + ((wave & 3) * 2) * 3072 + lane * 16;       // + 3072 for j = 1, + p * 1024
  bf16x8 F[2][12];                                                         // [set][A: i * 3 + p | B: 6 + j * 3 + p]

  // stage s -> its loads.  Stages past the end load zeros (the count of operations in flight stays what the waits assume).
  // In a step the four operations go out one by one BEHIND MFMAs (issue_part 0..2: B's pieces, 3: A): eight wavefronts
  // issuing 32 of them at the top of a step kept the CU's address unit busy for ~300 cycles with no MFMA in flight
  // (profiles/r06_gemm_w8.md section 2).
  auto issue_part = [&](int s, auto slot, auto part) {
    constexpr int P = decltype(slot)::value;
    constexpr int Q = decltype(part)::value;
    const bool live = s < nkt;
    if constexpr (Q < 3) {
      if constexpr (!(DBG & DBG_NO_DMA)) dma1<Q>(voff, live ? sb : szero, lds0 + P * ST_SZ + b_wr);
      if constexpr (Q == 2) sb += sb_step;
    } else {
      if constexpr (!(DBG & DBG_NO_ALOAD)) {
        if (s >= nfull) {      // the partial last tile and the stages past the end: zeros where there is no A
          const float* p = (live && atail_ok) ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(sa) + a_off) : g.zeros;
          aload(ra[P], p);
        } else {
          aload_s(ra[P], a_off, sa);
        }
      } else {
        ra[P] = f32x4{1.0f, 2.0f, 3.0f, 4.0f};
      }
      sa += BK;
    }
  };
  auto issue = [&](int s, auto slot) {
    issue_part(s, slot, std::integral_constant<int, 0>{});
    issue_part(s, slot, std::integral_constant<int, 1>{});
    issue_part(s, slot, std::integral_constant<int, 2>{});
    issue_part(s, slot, std::integral_constant<int, 3>{});
  };
  // accumulator sign groups: as k_gemm_bf16x3 (the bf16 MFMA's accumulation bias cancels between the groups)
  int flip[3] = {1 << 30, 1 << 30, 1 << 30};
  if (nkt >= 16) {
    flip[0] = __builtin_amdgcn_readfirstlane(max(2, (int)(0.173f * nkt + 0.5f) & ~1));      // (scalar registers: the
    flip[1] = (nkt / 2 + 1) & ~1;                                                             //  float conversion is VALU)
    flip[2] = __builtin_amdgcn_readfirstlane((int)(0.849f * nkt + 0.5f) & ~1);
  }
  auto sign_group = [&](int t) { return (t >= flip[0] ? 1 : 0) + (t >= flip[1] ? 1 : 0) + (t >= flip[2] ? 1 : 0); };
  auto negate_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f - acc[i][j][r];
  };
  // split of A's float4 of one stage in five pieces (<= 5 VALU + one ds_write_b64 each)
  uint32_t w0 = 0;
  auto split_piece = [&](auto piece, f32x4& v, uint32_t* dst, uint32_t negmask) {
    constexpr int c = decltype(piece)::value;
    if constexpr (c == 0 || c == 2) w0 = split_level<0, 1>(v, v, c == 0 ? negmask : 0u);
    if constexpr (c == 1 || c == 3) {
      const uint32_t w1 = split_level<2, 3>(v, v, c == 1 ? negmask : 0u);
      *reinterpret_cast<uint2*>(dst + (c / 2) * (A_PL / 4)) = make_uint2(w0, w1);
    }
    if constexpr (c == 4)
      *reinterpret_cast<uint2*>(dst + 2 * (A_PL / 4)) = make_uint2(split_last(v[0], v[1]), split_last(v[2], v[3]));
  };
  auto read_frag = [&](auto idx, const uint32_t* st, bf16x8 (&f)[12]) {
    constexpr int n = decltype(idx)::value;
    if constexpr (n < 6)
      f[n] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(st + (a_rd + (n / 3) * 512 + (n % 3) * A_PL) / 4));
    else
      f[n] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(st + (b_rd + ((n - 6) / 3) * 3072 + ((n - 6) % 3) * 1024) / 4));
  };

  constexpr int NP = 9;
  constexpr int pa_[NP] = {2, 2, 1, 2, 0, 1, 1, 0, 0};
  constexpr int pb_[NP] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
  constexpr int NM = 4 * TERMS;
  constexpr int M_WAIT = NM / 2 - 2;                 // the MFMA behind which the step waits for stage t + 2's loads
  // One step: MFMAs of stage t (fragment set t & 1); behind them the fragment reads of stage t + 1, the wait for the loads of
  // stage t + 2 and the split of its A.  MF = false: the same without MFMAs (pipeline fill).
  auto step = [&](int t, auto slot, auto mf) {
    constexpr int P = decltype(slot)::value;
    constexpr bool MF = decltype(mf)::value && !(DBG & DBG_NO_MFMA);
    const bool fine = (DBG & DBG_FINE) != 0 && t == 12;      // one step of the middle of the loop in detail
    if (fine) stamp(std::integral_constant<int, 12>{});
    if (MF && (t == flip[0] || t == flip[1] || t == flip[2])) negate_acc();
    const uint32_t negmask = __builtin_amdgcn_readfirstlane((sign_group(t + 2) & 1) ? 0x80000000u : 0u);
    const uint32_t* st_rd = lds + ((P + 1) & 3) * (ST_SZ / 4);
    uint32_t* st_wr = lds + ((P + 2) & 3) * (ST_SZ / 4) + a_wr / 4;
    bf16x8(&fc)[12] = F[P & 1];
    bf16x8(&fn)[12] = F[(P + 1) & 1];
    f32x4& rs = ra[(P + 2) & 3];
    __builtin_amdgcn_sched_barrier(0);
    auto weave = [&](auto mi) {
      constexpr int m = decltype(mi)::value;
      if constexpr (MF) {
        constexpr int tt = NP - TERMS + m / 4, i = (m / 2) % 2, j = m % 2;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fc[i * 3 + pa_[tt]], fc[6 + j * 3 + pb_[tt]], acc[i][j], 0, 0, 0);
      }
      if constexpr (m < 8 && m % 2 == 1) issue_part(t + 4, slot, std::integral_constant<int, m / 2>{});
      if constexpr (m == 7) {
        if (fine) stamp(std::integral_constant<int, 13>{});
      }
      if constexpr (m < 12 && !(DBG & DBG_NO_FRAG)) read_frag(mi, st_rd, fn);
      if constexpr (m == M_WAIT) {
        if (fine) stamp(std::integral_constant<int, 14>{});
        // two steps' loads stay in flight: 2 x (3 DMA + 1 A); fewer in the ablations that drop a kind of load
        constexpr int VMW = 2 * (((DBG & DBG_NO_DMA) ? 0 : 3) + ((DBG & DBG_NO_ALOAD) ? 0 : 1));
        if constexpr (VMW == 8) asm volatile("s_waitcnt vmcnt(8)" : "+v"(rs) : : "memory");
        else if constexpr (VMW == 6) asm volatile("s_waitcnt vmcnt(6)" : "+v"(rs) : : "memory");
        else if constexpr (VMW == 2) asm volatile("s_waitcnt vmcnt(2)" : "+v"(rs) : : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(rs) : : "memory");
        if (fine) stamp(std::integral_constant<int, 15>{});
      }
      if constexpr (m > M_WAIT && (m - M_WAIT) % 2 == 1 && (m - M_WAIT) / 2 < 5 && !(DBG & DBG_NO_SPLIT))
        split_piece(std::integral_constant<int, (m - M_WAIT) / 2>{}, rs, st_wr, negmask);
      if constexpr (m == 5 || m == NM - 1) {
        if (fine) stamp(std::integral_constant<int, m == 5 ? 16 : 17>{});
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    static_for<0, NM>(weave);
    if constexpr ((DBG & DBG_NO_MFMA) != 0 && !(DBG & DBG_NO_FRAG)) {      // keep the fragment reads alive
#pragma unroll
      for (int n = 0; n < 12; ++n) asm volatile("" : : "v"(fn[n]));
    }
    if constexpr ((DBG & DBG_NO_SPLIT) != 0) asm volatile("" : : "v"(rs));
    if (fine) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stamp(std::integral_constant<int, 18>{});
    }
    lds_barrier();
    if (fine) stamp(std::integral_constant<int, 19>{});
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  // ---- fill: stages 0 .. 3 on their way, A of stages 0 and 1 split, fragments of stage 0 in registers
  issue(0, I0{});
  issue(1, I1{});
  step(-2, I2{}, std::false_type{});      // (loads stage 2; "reads" stage -1: nothing yet, harmless; splits stage 0)
  step(-1, I3{}, std::false_type{});      // loads stage 3, reads stage 0, splits stage 1
  stamp(std::integral_constant<int, 1>{});
  // ---- the contraction
  int t = 0;
  for (; t + 4 <= nkt; t += 4) {
    step(t, I0{}, std::true_type{});
    step(t + 1, I1{}, std::true_type{});
    step(t + 2, I2{}, std::true_type{});
    step(t + 3, I3{}, std::true_type{});
    if constexpr ((DBG & DBG_TIMELINE) != 0) {      // after steps 4, 8, 16, 24, 32
      if (t == 0) stamp(std::integral_constant<int, 2>{});
      if (t == 4) stamp(std::integral_constant<int, 3>{});
      if (t == 12) stamp(std::integral_constant<int, 4>{});
      if (t == 20) stamp(std::integral_constant<int, 5>{});
      if (t == 28) stamp(std::integral_constant<int, 6>{});
    }
  }
  if (t < nkt) step(t, I0{}, std::true_type{});
  if (t + 1 < nkt) step(t + 1, I1{}, std::true_type{});
  if (t + 2 < nkt) step(t + 2, I2{}, std::true_type{});
  if (nkt > 0 && (sign_group(nkt - 1) & 1)) negate_acc();
  // the loads of the four stages past the end are still landing in the ring: the product is staged in the same memory
  // (and every asm load's destination stays reserved until here: a register the compiler believed free would be overwritten
  // when a load of a stage past the end lands)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]) : : "memory");
  stamp(std::integral_constant<int, 7>{});
  __syncthreads();
  stamp(std::integral_constant<int, 8>{});

  // ---- epilogue: acc register r of lane l holds C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]; two passes of 64 rows
  // (pass = the i of every wavefront) through LDS, written out as float4 rows
  float* C = g.C + (size_t)grp * g.batch.strideC;
  float* stage = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm >> 1) + (r & 3) + 8 * (r >> 2) + 4 * h;
        stage[row * WEPI_LD + wn + 32 * j + li] = acc[pass][j][r];
      }
    __syncthreads();
    constexpr int C4 = WBN / 4;
#pragma unroll
    for (int q = 0; q < 64 * C4 / WNTH; ++q) {
      const int f = threadIdx.x + WNTH * q;
      const int row = f / C4, c4 = f % C4;
      // staged row -> row of the tile: rows 0..31 belong to the wavefronts with wm = 0, 32..63 to wm = 64
      const int grow = m0 + (row >> 5) * 64 + 32 * pass + (row & 31), gcol = n0 + 4 * c4;
      if (grow < Mlim && gcol < g.N && !((DBG & DBG_NO_STORE) != 0 && g.M > 0)) {
        const float4 v = *reinterpret_cast<const float4*>(stage + row * WEPI_LD + 4 * c4);
        if (g.vecC) {
          if constexpr ((DBG & DBG_NT_STORE) != 0)
            __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(C + (size_t)grow * g.ldc + gcol));
          else
            *reinterpret_cast<float4*>(C + (size_t)grow * g.ldc + gcol) = v;
        } else {
          float* o = C + (size_t)grow * g.ldc + gcol;
          o[0] = v.x;
          if (gcol + 1 < g.N) o[1] = v.y;
          if (gcol + 2 < g.N) o[2] = v.z;
          if (gcol + 3 < g.N) o[3] = v.w;
        }
      }
    }
    __syncthreads();
  }
  if constexpr ((DBG & (DBG_TIMELINE | DBG_FINE)) != 0) {
    stamp(std::integral_constant<int, 9>{});
    if (g.tl != nullptr && lane == 0) {
      uint32_t xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      tl[20] = ((uint64_t)(xcc & 0xf) << 32) | (uint32_t)__builtin_amdgcn_s_getreg(/*HW_ID*/ 4 | (0 << 6) | (31 << 11));
      uint64_t* o = g.tl + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * TL_SLOTS;
#pragma unroll
      for (int i = 0; i < TL_SLOTS; ++i) o[i] = tl[i];
    }
  }
}

template <int TERMS, int DBG = 0>
hipError_t launch_w8(rgcn_ctx* c, const XArgs& g) {
  auto kern = k_gemm_w8<TERMS, DBG>;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       W_LDS_BYTES);
    if (e != hipSuccess) return e;
    configured = true;
  }
  const int gx = g.swizzle == 2 ? ((g.tiles_m + 7) / 8) * 8 * g.tiles_n : g.tiles_m * g.tiles_n;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)g.batch.groups), dim3(WNTH), W_LDS_BYTES, c->stream, g);
  return hipGetLastError();
}

}  // namespace

// A k-contiguous with 16-byte rows, B pre-split (batch->bfrag), no split over K: the caller (gemm_bf16x3_launch) checks.
hipError_t gemm_bf16x3_w8_launch(rgcn_ctx* c, int terms, int M, int N, int K, const float* A, int lda, float* C, int ldc,
                                 int swizzle, int vecC, const GemmBatch& batch) {
  XArgs g;
  g.batch = batch;
  g.bfrag = reinterpret_cast<const u32x4*>(batch.bfrag);
  g.A = A; g.B = nullptr; g.C = C; g.zeros = c->zeros;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = 0; g.ldc = ldc;
  g.k_per_split = K; g.splits = 1; g.swizzle = swizzle; g.vecC = vecC;
  g.tiles_m = (M + WBM - 1) / WBM;
  g.tiles_n = (N + WBN - 1) / WBN;
  g.nt32 = bfrag_nt32(N);
  g.tl = nullptr;
#ifdef RGCN_DEVTOOLS
  // RGCN_GEMM_W8 = 1000 + DBG: the ablations and timelines of tools/gemm_w8_lab.py (mode 6 only)
  const int v = knob("RGCN_GEMM_W8", 1);
  if (v == 4 && terms == 6) return launch_w8<6, DBG_NT_STORE>(c, g);
  if (v >= 1000 && terms == 6) {
    g.tl = reinterpret_cast<uint64_t*>(c->debug_buf);
    switch (v - 1000) {
#define W8_CASE(D) case D: return launch_w8<6, D>(c, g);
      W8_CASE(1) W8_CASE(2) W8_CASE(4) W8_CASE(8) W8_CASE(16) W8_CASE(32) W8_CASE(3) W8_CASE(7) W8_CASE(15) W8_CASE(31)
      W8_CASE(63) W8_CASE(64) W8_CASE(128) W8_CASE(24) W8_CASE(26) W8_CASE(256) W8_CASE(320)
#undef W8_CASE
      default: break;
    }
  }
#endif
  if (terms == 9) return launch_w8<9>(c, g);
  return launch_w8<6>(c, g);
}

size_t gemm_w8_timeline_bytes(int M, int N, int groups) {
  return (size_t)((M + WBM - 1) / WBM + 8) * ((N + WBN - 1) / WBN) * groups * 8 * TL_SLOTS * 8;
}

}  // namespace rgcn
