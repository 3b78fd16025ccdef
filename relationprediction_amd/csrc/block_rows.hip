// Destination-major, single-pass block-diagonal relational layer (ConcatGcn.compute_messages + combine_messages,
// code/encoders/message_gcns/gcn_basis_concat.py:35-52,69-83, inside MessageGcn.compute_vertex_embeddings,
// message_gcn.py:49-79) and its gradient w.r.t. the layer input:
//     forward   H'[v] = relu?( dropout(S[v]) + sum over the messages into v of  n_m * W[r_m] . H[src_m] )
//     backward  D'[u] = ( G[u] + sum over the messages out of u of  W[r_m]^T . (n_m * D[dst_m]) ) * relu'(H[u]),
//               dS'[u] = D'[u] * dropout'
// straight from the incidence CSR: no [2E, d] message buffer is written and read back (the two-kernel form,
// block_msgs.hip + k_combine, moves 2 x 60 MB per layer pass that way at FB15k-237 minibatch size, 2 x 1.1 GB at the
// 272,115-edge training graph).
//
// Decomposition:
//   * the row is cut into EIGHT column bands of nb/8 blocks, band x handled by the workgroups with blockIdx % 8 == x,
//     i.e. by the workgroups the dispatcher places on XCD x: that XCD touches 1/8 of every relation's weights
//     (0.6 MB at FB15k-237: L2-resident) and 1/8 of every gathered row (250 bytes; the band of the whole [V,d] operand
//     is 3.6 MB, about one XCD's L2), so the per-message weight re-reads the destination-major form needs are L2 hits
//     and what crosses the fabric is close to the compulsory traffic (operand + self-loop term in, result out);
//   * a GROUP of GW lanes (GW = 16 at nb = 100) owns (row, band): lane i holds block b0 + i -- its sd x sd coefficients
//     come from the weights' own [rel][sd*sd][nb] layout (one dword per coefficient, contiguous across the group), its
//     sd inputs are 4 sd bytes of a 250-byte contiguous piece of the partner row -- no LDS table, so occupancy is
//     bounded by registers only;
//   * short rows (<= kLongRow slots): one group per row, 64 / GW rows per wavefront, the row's slots one after the
//     other (slot indices fetched lane-parallel, one coalesced load for GW slots, and handed round by ds_bpermute);
//   * long rows: one WORKGROUP per row, tile by tile of 64 slots -- every group computes the messages of its slots of
//     the tile into LDS (20 KB), then eight interleaved lanes add them up (lane q: the slots beg + q, beg + q + 8, ...
//     in increasing order) and the eight partial sums are combined in lane order: the segmented reduction runs over
//     LDS, the loads it depends on run in parallel; long-row workgroups lead the grid;
//   * giant rows (full-graph scale, more than kGiantRow slots): kGiantRow-slot pieces, one workgroup each, into the piece
//     slab; k_combine's finishing pass (elementwise.hip) adds a row's pieces in piece order.
// Every sum is formed in EXACTLY k_combine's order and every message with k_block_msg_fwd / _bwd's arithmetic, so this
// kernel and the two-kernel form agree BITWISE (tests/test_gpu_parity.py::test_single_pass_layer_equals_the_two_kernel_form).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "rgcn_internal.h"


namespace rgcn {

namespace {

template <int N>
struct __attribute__((packed, aligned(4))) FloatN {
  float v[N];
};

struct RowsArgs {
  const float* X;            // gathered operand [V,d]: H_in (forward), D (backward)
  const float* W;            // band-tiled weights [2R][8][NT][GW][4] (k_wtile_build)
  const int32_t* row_ptr;    // [V+1] incidence CSR
  const int32_t* row_order;  // [V] rows by descending number of slots (long rows last)
  const int32_t* slot_v;     // per slot: the partner vertex (d_src forward, s_dst backward)
  const int32_t* slot_rel;   // per slot: directed relation in [0, 2R)
  const float* slot_norm;    // per slot: neighbour normalisation of that message
  const int32_t* long_rows;  // rows with more than kLongRow slots (and at most the giant threshold)
  const int32_t* nlong;
  const int32_t* piece_row;  // giant-row pieces (null when the cut is off)
  const int32_t* piece_k;
  const int32_t* ngiant;
  float* giant_slab;         // [pieces][d]
  const float* base;         // [V,d] self-loop term (S forward, G backward); dropout `drop` applies; rows [row_lo,row_hi)
  const float* gate;         // optional: result *= (gate > 0)
  float* out;
  float* out2;               // optional: out * dropout(drop2)
  DropSpec drop, drop2;
  int32_t V, d, nb, relu, row_lo, row_hi;
  int32_t n_long_wg;         // workgroups per band that walk the long-row / piece lists (they lead the grid)
  int32_t rows_per_wg;       // rows of one short-row workgroup
  int32_t giant_len;         // rows with more slots are giant rows (summed piece by piece); INT_MAX when the cut is off
  float* colpart;            // optional [gridDim / 8][d]: the column sums of `out` over the rows of workgroup j (every band's
                             // workgroup j writes its columns of row j) -- db_emb without a pass over dL/dH0
};

constexpr int kRowsThreads = 256;

// lanes per (row, band) group: the smallest of 8 / 16 / 32 / 64 that holds a band's ceil(nb / 8) blocks
int rows_group_width(const rgcn_ctx* c) {
  const int band = (c->nb + 7) / 8;
  return band <= 8 ? 8 : (band <= 16 ? 16 : (band <= 32 ? 32 : 64));
}

// One message of lane (group, block b): the sd inputs of the partner row and the relation's sd x sd block.
template <int SD>
struct SlotRegs {
  float w[SD * SD];
  float nr;
};

constexpr int nt_of(int sd) { return (sd * sd + 3) / 4; }       // float4s per (relation, block)

// Band-tiled weights: Wt[rel][band x][t][lane li][c] = W[rel][4 t + c][b0(x) + li]  (zero where 4 t + c >= sd*sd or
// li is beyond the band).  The GW lanes of a group read their t-th float4 from ONE aligned run of 16 GW bytes: every
// 128-byte line the vector cache fetches is used whole (the weights' own [rel][sd*sd][nb] layout gives a group 52
// useful bytes per line touched, and the kernel ran at the vector cache's line rate: 72 us for the short rows).
struct WtileJobs {
  const float* W[8];
  float* Wt[8];
};
// (blockIdx.y = layer: a train step rebuilds the copies of every layer with one launch)
__global__ void k_wtile_build(WtileJobs jobs, int R2, int nb, int sd2, int nt, int gw) {
  const float* __restrict__ W = jobs.W[blockIdx.y];
  float* __restrict__ Wt = jobs.Wt[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one float4 of Wt
  if (i >= (int64_t)R2 * 8 * nt * gw) return;
  const int li = (int)(i % gw);
  const int t = (int)((i / gw) % nt);
  const int x = (int)((i / ((int64_t)gw * nt)) % 8);
  const int rel = (int)(i / ((int64_t)gw * nt * 8));
  const int b0 = (x * nb) >> 3, b1 = ((x + 1) * nb) >> 3;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (b0 + li < b1) {
    const float* wp = W + ((size_t)rel * sd2 + 4 * t) * nb + b0 + li;
    if (4 * t + 0 < sd2) o.x = wp[0];
    if (4 * t + 1 < sd2) o.y = wp[(size_t)nb];
    if (4 * t + 2 < sd2) o.z = wp[2 * (size_t)nb];
    if (4 * t + 3 < sd2) o.w = wp[3 * (size_t)nb];
  }
  reinterpret_cast<float4*>(Wt)[i] = o;
}

// wq: this lane's float4 column of the band's tile, Wt + ((x * NT) * GW + li) * 4; a relation is 8 NT GW float4s on
// The coefficients of relation rl for this lane's block -- fetched only when the group's registers hold ANOTHER relation's
// (`held`): the graph preparation orders a row's slots by directed relation, and an entity's edges use few relations, so
// consecutive slots of a group mostly repeat the relation (weight fetches per slot: 0.60 on the FB15k-237 minibatch, 0.07 /
// 0.06 / 0.02 on the 272,115 / 141,442 / 483,142-edge training graphs; before round 5 every slot fetched its 28 registers
// of coefficients through L1: 420 MB per launch at the minibatch, 7.8 GB at 272 k edges).  The norm changes per slot.
template <int SD, int GW>
__device__ __forceinline__ void load_w(const float4* __restrict__ wq, int rl, float nr, SlotRegs<SD>& r, int& held) {
  constexpr int NT = nt_of(SD);
  if (rl != held) {
    const float4* wp = wq + (size_t)rl * (8 * NT * GW);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 q = wp[t * GW];
      r.w[4 * t] = q.x;
      if (4 * t + 1 < SD * SD) r.w[4 * t + 1] = q.y;
      if (4 * t + 2 < SD * SD) r.w[4 * t + 2] = q.z;
      if (4 * t + 3 < SD * SD) r.w[4 * t + 3] = q.w;
    }
    held = rl;
  }
  r.nr = nr;
}
// the lane's sd inputs of partner row pv (4 sd bytes of the row's 16 GW-lane band piece)
template <int SD>
__device__ __forceinline__ FloatN<SD> load_x(const RowsArgs& a, int pv, int col) {
  return *reinterpret_cast<const FloatN<SD>*>(a.X + (size_t)pv * a.d + col);
}

// the message's contribution to this lane's sd outputs: the arithmetic of k_block_msg_fwd / k_block_msg_bwd
template <int SD, bool BWD>
__device__ __forceinline__ void slot_value(const SlotRegs<SD>& r, const FloatN<SD>& x, float (&y)[SD]) {
#pragma clang fp contract(off)
  if constexpr (!BWD) {
#pragma unroll
    for (int i = 0; i < SD; ++i) {
      float t = 0.0f;
#pragma unroll
      for (int q = 0; q < SD; ++q) t = fmaf(r.w[i * SD + q], x.v[q], t);   // out_i = sum_j T[i][j] x_j
      y[i] = t * r.nr;
    }
  } else {
    float gr[SD];
#pragma unroll
    for (int q = 0; q < SD; ++q) gr[q] = x.v[q] * r.nr;
#pragma unroll
    for (int q = 0; q < SD; ++q) {
      float z = 0.0f;
#pragma unroll
      for (int i = 0; i < SD; ++i) z = fmaf(r.w[i * SD + q], gr[i], z);      // (T^T (n g))_j
      y[q] = z;
    }
  }
}

// Row state of lane (group, block b): the self-loop term and the gate of its sd outputs, fetched when the row is taken up
// (the loads then fly under the row's messages).
template <int SD>
struct RowRegs {
  FloatN<SD> base, gate;
  bool has_base;
};

template <int SD>
__device__ __forceinline__ void row_open(const RowsArgs& a, int v, size_t off, RowRegs<SD>& r) {
  r.has_base = a.base != nullptr && v >= a.row_lo && v < a.row_hi;
  if (a.base != nullptr) r.base = *reinterpret_cast<const FloatN<SD>*>(a.base + off);      // (uniform test)
  if (a.gate != nullptr) r.gate = *reinterpret_cast<const FloatN<SD>*>(a.gate + off);
}

// the lane's sd dropout factors: ONE (uniform) dispatch on the spec's mode for all of them
template <int SD>
__device__ __forceinline__ void drop_factors(const DropSpec& ds, const DropKey& k, size_t off, float (&f)[SD]) {
  if (ds.mode == DROP_RNG) {
#pragma unroll
    for (int i = 0; i < SD; ++i) f[i] = drop_bits24(k, off + i) < ds.thresh ? ds.inv_keep : 0.0f;
  } else if (ds.mode == DROP_MASK) {
#pragma unroll
    for (int i = 0; i < SD; ++i) f[i] = ds.mask[off + i] ? ds.inv_keep : 0.0f;
  } else {
#pragma unroll
    for (int i = 0; i < SD; ++i) f[i] = 1.0f;
  }
}

// tot = dropout(base)  (k_combine's prologue: scale, no fused multiply-add with what follows)
template <int SD>
__device__ __forceinline__ void row_prologue(const RowsArgs& a, const DropKey& k1, const RowRegs<SD>& r, size_t off,
                                             float (&tot)[SD]) {
#pragma clang fp contract(off)
  float f[SD];
  drop_factors<SD>(a.drop, k1, off, f);
#pragma unroll
  for (int i = 0; i < SD; ++i) {
    float t = r.base.v[i];
    t *= f[i];
    tot[i] = r.has_base ? t : 0.0f;
  }
}

// relu' gate, relu, dropout-scaled second copy (k_combine's epilogue): the values, and their store
template <int SD>
struct RowOut {
  FloatN<SD> o, o2;
  size_t off;
  bool store;
};
template <int SD>
__device__ __forceinline__ void row_epilogue(const RowsArgs& a, const DropKey& k2, const RowRegs<SD>& r, size_t off,
                                             const float (&tot)[SD], bool store, RowOut<SD>& out) {
#pragma clang fp contract(off)
  float f[SD];
  if (a.out2 != nullptr) drop_factors<SD>(a.drop2, k2, off, f);
#pragma unroll
  for (int i = 0; i < SD; ++i) {
    float val = tot[i];
    if (a.gate != nullptr) val = r.gate.v[i] > 0.0f ? val : 0.0f;
    if (a.relu) val = fmaxf(val, 0.0f);
    out.o.v[i] = val;
    out.o2.v[i] = a.out2 != nullptr ? val * f[i] : 0.0f;
  }
  out.off = off;
  out.store = store;
}
template <int SD>
__device__ __forceinline__ void row_store(const RowsArgs& a, const RowOut<SD>& out) {
  if (out.store) {
    *reinterpret_cast<FloatN<SD>*>(a.out + out.off) = out.o;
    if (a.out2 != nullptr) *reinterpret_cast<FloatN<SD>*>(a.out2 + out.off) = out.o2;
  }
}

// Lanes of slot tiles: phase 1 of a long row computes the messages of TS consecutive slots in parallel (one group per
// slot and turn) into LDS, phase 2 adds them up in k_combine's long-row order.  TS x GW x SD floats <= 20 KB, so that
// the LDS never bounds the occupancy the registers allow.
template <int SD, int GW>
struct LongTile {
  static constexpr int TS = GW * SD <= 80 ? 64 : (GW * SD <= 160 ? 32 : (GW * SD <= 320 ? 16 : 8));
  static constexpr int NGW = (kRowsThreads / 64) * (64 / GW);      // groups of a workgroup
  static constexpr int SPG = TS / NGW;                              // slots per group and tile
  static constexpr int PT = 8 * GW < kRowsThreads ? 8 * GW : kRowsThreads;   // phase-2 threads
  static constexpr int NQ = 8 * GW / PT;                            // interleaved lanes per phase-2 thread
  static_assert(TS % 8 == 0 && SPG >= 2 && SPG % 2 == 0, "tile geometry");
};

// Sum of the slots [beg, end) by the whole workgroup in k_combine's long-row order: eight interleaved lanes (lane q
// adds the slots beg + q, beg + q + 8, ... in increasing order), combined ((0 + 1) + 2) ... + 7.  The MESSAGES are
// computed tile by tile with every group of the workgroup busy (a hub row is a chain of hundreds of dependent adds but
// its messages are independent: one wavefront walking the chain with its loads in line took ~80 us for a 431-slot hub);
// the chain itself then runs over LDS.  The result is valid in the threads tid < GW (= group 0 of wave 0).
template <int SD, bool BWD, int GW>
__device__ __forceinline__ void wg_long_sum(const RowsArgs& a, int beg, int end, float* __restrict__ ybuf,
                                            const float4* __restrict__ wq, int col, float (&sum)[SD]) {
#pragma clang fp contract(off)
  using T = LongTile<SD, GW>;
  const int tid = threadIdx.x, lane = tid & 63, li = lane % GW;
  const int G = (tid >> 6) * (64 / GW) + lane / GW;                 // group of the workgroup
  float part[T::NQ][SD];
#pragma unroll
  for (int jj = 0; jj < T::NQ; ++jj)
#pragma unroll
    for (int i = 0; i < SD; ++i) part[jj][i] = 0.0f;
  // The gathered inputs (x) of a slot miss the L2 (they come from the Infinity Cache / HBM, ~0.6 us), its weights hit it:
  // x is requested ONE SLOT AHEAD of the weights it meets (across tile boundaries too), the tile's slot indices one
  // TILE ahead, so that a step waits for an L2 hit only.
  auto tile_idx = [&](int c0, int& pv, int& rl, float& nr) {
    const int sl = c0 + lane;
    const bool ok = lane < T::TS && sl < end;     // every wavefront fetches the tile's indices itself (one coalesced
    pv = ok ? a.slot_v[sl] : 0;                   // load per array) and hands them round by ds_bpermute
    rl = ok ? a.slot_rel[sl] : 0;
    nr = ok ? a.slot_norm[sl] : 0.0f;
  };
  int my_pv, my_rl, nx_pv = 0, nx_rl = 0;
  float my_nr, nx_nr = 0.0f;
  tile_idx(beg, my_pv, my_rl, my_nr);
  // group G takes the SPG CONSECUTIVE slots G SPG .. of every tile (a run of one relation stays inside a group; the sums
  // below go by slot index, whichever group computed the message)
  FloatN<SD> xcur = load_x<SD>(a, __shfl(my_pv, G * T::SPG, 64), col);
  SlotRegs<SD> r0;
  int held = -1;                                  // relation whose coefficients r0 holds
  for (int c0 = beg; c0 < end; c0 += T::TS) {
    // ---- phase 1: the tile's messages
    const bool more = c0 + T::TS < end;
#pragma unroll
    for (int u = 0; u < T::SPG; ++u) {
      const int s0 = G * T::SPG + u;
      const int rl0 = __shfl(my_rl, s0, 64);
      const float nr0 = __shfl(my_nr, s0, 64);
      // (the memory counter retires loads in issue order: what a step waits for -- its weights -- is requested FIRST,
      // what may stay in flight -- the next slot's inputs, the next tile's indices -- behind it)
      const bool live = c0 + s0 < end;           // (group-uniform; a dead slot has index 0: valid addresses)
      if (live) load_w<SD, GW>(wq, rl0, nr0, r0, held);
      if (u == 0 && more) tile_idx(c0 + T::TS, nx_pv, nx_rl, nx_nr);
      FloatN<SD> xnext = xcur;
      if (u + 1 < T::SPG) xnext = load_x<SD>(a, __shfl(my_pv, G * T::SPG + (u + 1 < T::SPG ? u + 1 : u), 64), col);
      else if (more) xnext = load_x<SD>(a, __shfl(nx_pv, G * T::SPG, 64), col);
      if (live) {
        float y0[SD];
        slot_value<SD, BWD>(r0, xcur, y0);
#pragma unroll
        for (int i = 0; i < SD; ++i) ybuf[(s0 * GW + li) * SD + i] = y0[i];
      }
      xcur = xnext;
    }
    my_pv = nx_pv; my_rl = nx_rl; my_nr = nx_nr;
    __syncthreads();
    // ---- phase 2: thread (q, li) adds lane q's slots of the tile, in increasing order
    if (tid < T::PT) {
#pragma unroll
      for (int jj = 0; jj < T::NQ; ++jj) {
        const int q = tid / GW + jj * (T::PT / GW), pl = tid % GW;
        const int nlive = min(T::TS, end - c0);
#pragma unroll
        for (int t = 0; t < T::TS / 8; ++t) {
          const int s = q + 8 * t;
          if (s < nlive) {
#pragma unroll
            for (int i = 0; i < SD; ++i) part[jj][i] = part[jj][i] + ybuf[(s * GW + pl) * SD + i];
          }
        }
      }
    }
    __syncthreads();
  }
  // ---- the eight lanes in lane order
  if (tid < T::PT) {
#pragma unroll
    for (int jj = 0; jj < T::NQ; ++jj) {
      const int q = tid / GW + jj * (T::PT / GW), pl = tid % GW;
#pragma unroll
      for (int i = 0; i < SD; ++i) ybuf[(q * GW + pl) * SD + i] = part[jj][i];
    }
  }
  __syncthreads();
  if (tid < GW) {
#pragma unroll
    for (int i = 0; i < SD; ++i) {
      float t = ybuf[(0 * GW + tid) * SD + i];
#pragma unroll
      for (int q = 1; q < 8; ++q) t = t + ybuf[(q * GW + tid) * SD + i];
      sum[i] = t;
    }
  }
  __syncthreads();
}

template <int SD, bool BWD, int GW, bool CS>
__global__ void __launch_bounds__(kRowsThreads) k_block_rows(RowsArgs a) {
#pragma clang fp contract(off)
  constexpr int NG = 64 / GW;
  __shared__ float ybuf[LongTile<SD, GW>::TS * GW * SD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / GW, li = lane % GW;
  // band of this workgroup's XCD
  const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int b0 = (x * a.nb) >> 3, b1 = ((x + 1) * a.nb) >> 3;
  const int nbx = b1 - b0;
  if (nbx <= 0) return;
  const bool lane_ok = li < nbx;
  // CS: this lane's columns of `out`, summed over the rows it stores -- in LDS, a private slot per thread (five more
  // registers would cost the kernel its fourth wavefront per SIMD)
  __shared__ float cs[CS ? kRowsThreads * SD : 1];
  if constexpr (CS) {
#pragma unroll
    for (int i = 0; i < SD; ++i) cs[threadIdx.x * SD + i] = 0.0f;
  }
  const int b = b0 + (lane_ok ? li : nbx - 1);       // idle lanes shadow the band's last block (loads only)
  const int col = b * SD;
  const float4* wq = reinterpret_cast<const float4*>(a.W) + ((size_t)x * nt_of(SD)) * GW + li;
  DropKey key1 = drop_key(a.drop), key2 = drop_key(a.drop2);      // (wave-uniform: keep them in scalar registers)
  key1.k0 = __builtin_amdgcn_readfirstlane(key1.k0); key1.k1 = __builtin_amdgcn_readfirstlane(key1.k1);
  key2.k0 = __builtin_amdgcn_readfirstlane(key2.k0); key2.k1 = __builtin_amdgcn_readfirstlane(key2.k1);
  if (j < a.n_long_wg) {
    // ---- long rows and giant-row pieces: one workgroup each
    const bool writer = threadIdx.x < GW && lane_ok;
    // the long rows in the order of their vertex ids: the tail of row_order (stable sort, the long rows' key is the
    // largest).  GraphBufs::long_rows lists the same rows in the order their threads registered them, which differs
    // from run to run -- harmless for the rows themselves, not for the column sums a workgroup forms over ITS rows.
    const int n = *a.nlong + (a.ngiant != nullptr ? a.ngiant[0] : 0);
    for (int idx = j; idx < n; idx += a.n_long_wg) {
      const int v = a.row_order[a.V - n + idx];
      const int beg = a.row_ptr[v], end = a.row_ptr[v + 1];
      if (end - beg > a.giant_len) continue;          // a giant row: its pieces, below
      const size_t off = (size_t)v * a.d + col;
      float tot[SD], sum[SD];
      RowRegs<SD> rr;
      if (writer) row_open<SD>(a, v, off, rr);
      wg_long_sum<SD, BWD, GW>(a, beg, end, ybuf, wq, col, sum);
      if (writer) {
        row_prologue<SD>(a, key1, rr, off, tot);
#pragma unroll
        for (int i = 0; i < SD; ++i) tot[i] = tot[i] + sum[i];
        RowOut<SD> ro;
        row_epilogue<SD>(a, key2, rr, off, tot, true, ro);
        row_store<SD>(a, ro);
        if constexpr (CS) {
#pragma unroll
          for (int i = 0; i < SD; ++i) cs[threadIdx.x * SD + i] += ro.o.v[i];
        }
      }
    }
    if (a.ngiant != nullptr) {
      const int np = a.ngiant[1];
      for (int p = j; p < np; p += a.n_long_wg) {
        const int v = a.piece_row[p];
        const int beg = a.row_ptr[v] + a.piece_k[p] * kGiantRow;
        const int end = min(a.row_ptr[v + 1], beg + kGiantRow);
        float sum[SD];
        wg_long_sum<SD, BWD, GW>(a, beg, end, ybuf, wq, col, sum);
        if (writer) {
          FloatN<SD> o;
#pragma unroll
          for (int i = 0; i < SD; ++i) o.v[i] = sum[i];
          *reinterpret_cast<FloatN<SD>*>(a.giant_slab + (size_t)p * a.d + col) = o;
        }
      }
    }
    if constexpr (CS) {
      if (writer) {
        FloatN<SD> o;
#pragma unroll
        for (int i = 0; i < SD; ++i) o.v[i] = cs[threadIdx.x * SD + i];
        *reinterpret_cast<FloatN<SD>*>(a.colpart + (size_t)j * a.d + col) = o;
      }
    }
    return;
  }

  // ---- short rows: one group per row, NG rows per wavefront and turn.  Rows come in the order of graph_prep's
  // row_order (descending number of slots): the NG rows of a turn have the same length, so the wavefront does not
  // idle on its longest row (in vertex order 59 % of the lane-slots were idle), and the heavy workgroups start first.
  // A wavefront owns RW consecutive positions: row ids and row pointers of all its turns come with two coalesced loads,
  // the slot indices of turn t + 1 are fetched under the messages of turn t.
  // Turn t of wavefront W (of NW) takes the NG consecutive positions (t NW + W) NG ..: the rows of a turn are
  // neighbours in the sorted order (equal lengths), the turns of a wavefront are dealt from the whole order like
  // cards -- every wavefront gets its share of heavy and of light rows (contiguous chunks of the sorted order left the
  // first workgroups with all the 32-slot rows: 75 us at 64 rows per workgroup against 47 at 32).
  const int RW = a.rows_per_wg / (kRowsThreads / 64);                 // rows of a wavefront, <= 64
  const int nturn = RW / NG;
  const int W = (j - a.n_long_wg) * (kRowsThreads / 64) + wave;
  const int NW = (int)(gridDim.x >> 3) - a.n_long_wg;
  const int NWtot = NW * (kRowsThreads / 64);
  int my_v = -1, my_beg = 0, my_n = 0;
  {
    const int t = lane / NG, gg = lane % NG;
    const int p = (t * NWtot + W) * NG + gg;
    if (t < nturn && p < a.V) {
      my_v = a.row_order[p];
      my_beg = a.row_ptr[my_v];
      my_n = a.row_ptr[my_v + 1] - my_beg;
      if (my_n > kLongRow) { my_v = -1; my_n = 0; }      // a long-row workgroup (or the giant-row pass) owns this row
    }
  }
  // Pipeline of a wavefront's turns (the gathered inputs x miss the L2, ~0.6 us; the weights hit it): slot indices
  // TWO turns ahead, the row's self-loop / gate pieces and the first x of a turn ONE turn ahead, inside a row x one
  // slot ahead of the weights it meets -- a step of the slot loop waits for an L2 hit only.
  struct Idx { int pv, rl; float nr; };
  auto fetch_idx = [&](int beg, int n, int c0, Idx& r) {
    const bool ok = c0 + li < n;                        // this group's next GW slot indices, one coalesced load per array
    r.pv = ok ? a.slot_v[beg + c0 + li] : 0;
    r.rl = ok ? a.slot_rel[beg + c0 + li] : 0;
    r.nr = ok ? a.slot_norm[beg + c0 + li] : 0.0f;
  };
  auto turn_row = [&](int t, int& v, int& beg, int& n) {
    const int src = min(t * NG + g, 63);
    const bool ok = t < nturn;
    v = __shfl(my_v, src, 64); beg = __shfl(my_beg, src, 64); n = __shfl(my_n, src, 64);
    if (!ok) { v = -1; beg = 0; n = 0; }
  };
  int vA, begA, nA, vB, begB, nB, vC, begC, nC;
  Idx iA, iB, iC;
  turn_row(0, vA, begA, nA);
  turn_row(1, vB, begB, nB);
  fetch_idx(begA, nA, 0, iA);
  fetch_idx(begB, nB, 0, iB);
  RowRegs<SD> rrA, rrB;
  size_t offA = (size_t)max(vA, 0) * a.d + col;
  row_open<SD>(a, max(vA, 0), offA, rrA);
  rrA.has_base = rrA.has_base && vA >= 0;
  FloatN<SD> xcur = load_x<SD>(a, __shfl(iA.pv, g * GW, 64), col);
  SlotRegs<SD> r0;
  int held = -1;                                  // relation whose coefficients r0 holds (kept across slots, rows, turns)
  for (int turn = 0; turn < nturn; ++turn) {
    turn_row(turn + 2, vC, begC, nC);
    const size_t offB = (size_t)max(vB, 0) * a.d + col;
    // rows come by descending length: group 0 holds the turn's longest row, so the loop bounds are SCALAR (lane 0's
    // row) and the other groups are predicated -- no exec-mask bookkeeping in the slot loop; a group whose row is
    // shorter (a class boundary of the order, rare) re-reads a slot and discards the value
    const int n_u = __builtin_amdgcn_readfirstlane(nA);
    const int nB_u = __builtin_amdgcn_readfirstlane(nB);
    if (n_u == 0) {        // empty rows (the tail of the order): nothing to wait for but the row's own pieces
      fetch_idx(begC, nC, 0, iC);
      row_open<SD>(a, max(vB, 0), offB, rrB);
    }
    if (__builtin_amdgcn_readfirstlane(vA) >= 0) {
      float tot[SD];
      for (int c0 = 0; c0 < n_u; c0 += GW) {
        if (c0 > 0) {      // a row with more than GW slots: next chunk of indices, its first x (both in line: rare)
          fetch_idx(begA, nA, c0, iA);
          xcur = load_x<SD>(a, __shfl(iA.pv, g * GW, 64), col);
        }
        const int m_u = min(GW, n_u - c0);
        for (int k = 0; k < m_u; ++k) {
          const int s0 = g * GW + k;
          // (the memory counter retires loads in issue order: the weights this step waits for are requested FIRST, what
          // may stay in flight -- the next x, the next turn's row pieces, the indices two turns on -- behind them)
          const bool live0 = c0 + k < nA;
          {
            const int rl0 = __shfl(iA.rl, s0, 64);
            const float nr0 = __shfl(iA.nr, s0, 64);
            if (live0) load_w<SD, GW>(wq, rl0, nr0, r0, held);      // (a dead slot's value is discarded below)
          }
          FloatN<SD> xnext = xcur;
          if (k + 1 < m_u) xnext = load_x<SD>(a, __shfl(iA.pv, s0 + 1, 64), col);
          else if (c0 + GW >= n_u && nB_u > 0) xnext = load_x<SD>(a, __shfl(iB.pv, g * GW, 64), col);
          if (c0 + k == 0) {
            fetch_idx(begC, nC, 0, iC);
            row_open<SD>(a, max(vB, 0), offB, rrB);
            row_prologue<SD>(a, key1, rrA, offA, tot);
          }
          float y0[SD];
          slot_value<SD, BWD>(r0, xcur, y0);
#pragma unroll
          for (int i = 0; i < SD; ++i) {
            const float t0 = tot[i] + y0[i];
            tot[i] = live0 ? t0 : tot[i];
          }
          xcur = xnext;
        }
      }
      if (n_u == 0) row_prologue<SD>(a, key1, rrA, offA, tot);
      RowOut<SD> ro;      // (stored at once: holding it back behind the next turn's first weight request -- the memory
      row_epilogue<SD>(a, key2, rrA, offA, tot, vA >= 0 && lane_ok, ro);      // counter retires in issue order -- cost 14
      row_store<SD>(a, ro);                                                    // registers, a wavefront per SIMD, +1 us)
      if constexpr (CS) {
        if (ro.store) {
#pragma unroll
          for (int i = 0; i < SD; ++i) cs[threadIdx.x * SD + i] += ro.o.v[i];
        }
      }
    }
    vA = vB; begA = begB; nA = nB; iA = iB; offA = offB;
    rrA = rrB;
    rrA.has_base = rrA.has_base && vA >= 0;
    vB = vC; begB = begC; nB = nC; iB = iC;
  }
  if constexpr (CS) {
    // the workgroup's (wavefront, group) slots of a column in a fixed order
    __syncthreads();
    if (threadIdx.x < GW && lane_ok) {
      FloatN<SD> o;
#pragma unroll
      for (int i = 0; i < SD; ++i) {
        float t = 0.0f;
        for (int q = 0; q < (kRowsThreads / 64) * NG; ++q) t += cs[(q * GW + li) * SD + i];
        o.v[i] = t;
      }
      *reinterpret_cast<FloatN<SD>*>(a.colpart + (size_t)j * a.d + col) = o;
    }
  }
}

template <int SD, int GW>
hipError_t launch_rows(rgcn_ctx* c, const RowsArgs& a, bool backward, int grid) {
  if (backward && a.colpart != nullptr)
    hipLaunchKernelGGL((k_block_rows<SD, true, GW, true>), dim3((unsigned)grid), dim3(kRowsThreads), 0, c->stream, a);
  else if (backward) hipLaunchKernelGGL((k_block_rows<SD, true, GW, false>), dim3((unsigned)grid), dim3(kRowsThreads), 0, c->stream, a);
  else hipLaunchKernelGGL((k_block_rows<SD, false, GW, false>), dim3((unsigned)grid), dim3(kRowsThreads), 0, c->stream, a);
  return hipGetLastError();
}

template <int SD>
hipError_t launch_rows_gw(rgcn_ctx* c, const RowsArgs& a, bool backward, int grid, int gw) {
  switch (gw) {
    case 8: return launch_rows<SD, 8>(c, a, backward, grid);
    case 16: return launch_rows<SD, 16>(c, a, backward, grid);
    case 32: return launch_rows<SD, 32>(c, a, backward, grid);
    default: return launch_rows<SD, 64>(c, a, backward, grid);
  }
}

}  // namespace


// floats of one layer's band-tiled weight copy (allocated at create: a capture may be the first call that needs it)
size_t block_rows_weight_floats(const rgcn_ctx* c) {
  return (size_t)2 * c->R * 8 * nt_of(c->sd) * rows_group_width(c) * 4;
}

bool block_rows_available(const rgcn_ctx* c) {
  return c->kind == RGCN_KIND_BLOCK && c->nb <= 512 && c->g.d_src != nullptr && c->g.row_order != nullptr;
}

// band-tiled copies of the layers' relation weights, rebuilt -- every layer's, one launch -- when the weights changed
// (set_param, Adam) and once inside every captured step, whose replays follow weights the host does not see
static rgcn_status block_rows_refresh_weights(rgcn_ctx* c, int layer) {
  const int gw = rows_group_width(c), nt = nt_of(c->sd);
  const int64_t n4 = (int64_t)2 * c->R * 8 * nt * gw;
  LayerBufs& mine = c->layers[layer];
  if (!mine.wtile) {
    if (c->capturing) RGCN_FAIL(c, RGCN_ERR_STATE, "internal: the band-tiled weight copy must exist before a capture");
    RGCN_HIP(c, hipMalloc((void**)&mine.wtile, sizeof(float) * 4 * (size_t)n4));
  }
  if (c->capturing ? c->wtile_fresh : mine.wtile_version == c->weights_version) return RGCN_OK;
  c->wtile_fresh = true;
  for (int l0 = 1; l0 <= c->L; l0 += 8) {
    WtileJobs jobs;
    int nj = 0;
    for (int l = l0; l <= c->L && nj < 8; ++l) {
      LayerBufs& lb = c->layers[l];
      if (!lb.wtile) continue;
      jobs.W[nj] = lb.wrel; jobs.Wt[nj] = lb.wtile;
      lb.wtile_version = c->capturing ? ~0ull : c->weights_version;
      ++nj;
    }
    if (nj == 0) continue;
    for (int k = nj; k < 8; ++k) { jobs.W[k] = jobs.W[0]; jobs.Wt[k] = jobs.Wt[0]; }
    ProfScope ps(c, "block_wtile_build", nj * (4.0 * (2.0 * c->R * c->sd * c->sd * c->nb) + 16.0 * n4), 0);
    hipLaunchKernelGGL(k_wtile_build, dim3((unsigned)((n4 + 255) / 256), (unsigned)nj), dim3(256), 0, c->stream, jobs,
                       2 * c->R, c->nb, c->sd * c->sd, nt, gw);
    RGCN_HIP(c, hipGetLastError());
  }
  return RGCN_OK;
}

rgcn_status block_rows(rgcn_ctx* c, const char* tag, int layer, bool backward, const float* X, const CombineArgs& ca) {
  RGCN_TRY(block_rows_refresh_weights(c, layer));
  RowsArgs a;
  a.X = X;
  a.W = c->layers[layer].wtile;
  a.row_ptr = c->g.row_ptr;
  a.row_order = c->g.row_order;
  a.slot_v = backward ? c->g.s_dst : c->g.d_src;
  a.slot_rel = backward ? c->g.s_rel : c->g.d_rel;
  a.slot_norm = backward ? c->g.s_norm : c->g.d_norm;
  a.long_rows = c->g.long_rows;
  a.nlong = c->g.nlong;
  const bool giant = c->g.giant_on;
  if (giant && !c->giant_slab)
    RGCN_HIP(c, hipMalloc((void**)&c->giant_slab, sizeof(float) * (size_t)c->g.piece_cap * c->d));
  a.piece_row = giant ? c->g.piece_row : nullptr;
  a.piece_k = giant ? c->g.piece_k : nullptr;
  a.ngiant = giant ? c->g.ngiant : nullptr;
  a.giant_slab = giant ? c->giant_slab : nullptr;
  a.giant_len = giant ? kGiantRow : 0x7fffffff;
  a.base = ca.base; a.gate = ca.gate; a.out = ca.out; a.out2 = ca.out2; a.drop = ca.drop; a.drop2 = ca.drop2;
  a.V = c->V; a.d = c->d; a.nb = c->nb; a.relu = ca.relu; a.row_lo = ca.row_lo; a.row_hi = ca.row_hi;
  // rows of a short-row workgroup: a multiple of 4 NG (whole turns for its four wavefronts), at most 256
  {
    const int quantum = 4 * (64 / rows_group_width(c));
    const int rpw = std::max(quantum, std::min(256, 64 / quantum * quantum));
    a.rows_per_wg = rpw;
  }
  // one workgroup per long row and turn: about one per 256 slots of the graph, 32 .. 1024 workgroups per band
  const int64_t want = (2 * c->g.E) / 256;
  a.n_long_wg = c->g.E > 0 ? (int)std::max<int64_t>(32, std::min<int64_t>(1024, want)) : 0;
  const int n_row_wg = (c->V + a.rows_per_wg - 1) / a.rows_per_wg;
  const int grid = 8 * (a.n_long_wg + n_row_wg);
  // column sums of the output on the way (CombineArgs::colsum: the bias gradient behind the bottom layer's row
  // gradients): part row j = workgroup j of every band; giant rows, finished by another kernel, get rows of their own
  a.colpart = nullptr;
  c->colsum_parts = 0;
  const int giant_parts = giant ? c->g.giant_cap : 0;
  if (backward && ca.colsum && c->colsum_part != nullptr &&
      (size_t)(grid / 8 + giant_parts) * c->d <= c->colsum_part_floats) {
    a.colpart = c->colsum_part;
    c->colsum_parts = grid / 8 + giant_parts;
  }
  const double M = 2.0 * c->g.E, P = 4.0 * c->V * c->d;
  const double Wb = 8.0 * c->R * c->nb * c->sd * c->sd;
  const double streams = (ca.base ? 1.0 : 0.0) + (ca.gate ? 1.0 : 0.0) + 1.0 + (ca.out2 ? 1.0 : 0.0);
  const double rows = M < c->V ? M : (double)c->V;
  // design: every slot gathers one operand row and one relation's weights (through L2), the slot arrays once per band;
  // compulsory: the distinct gathered rows once, the slot arrays and the weights once, the row streams once
  ProfScope ps(c, tag, 4.0 * c->d * M + streams * P + 12.0 * M * 8.0 + 4.0 * c->sd * c->sd * c->nb * M + 4.0 * c->V,
               M * 2.0 * c->d * c->sd, 4.0 * c->d * rows + streams * P + 12.0 * M + Wb + 4.0 * c->V);
  hipError_t e = hipSuccess;
  const int gw = rows_group_width(c);
#define RGCN_ROWS_CASE(SDV) \
  case SDV: e = launch_rows_gw<SDV>(c, a, backward, grid, gw); break;
  switch (c->sd) {
    RGCN_ROWS_CASE(1) RGCN_ROWS_CASE(2) RGCN_ROWS_CASE(3) RGCN_ROWS_CASE(4) RGCN_ROWS_CASE(5) RGCN_ROWS_CASE(8)
    default: RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "block size d/nb must be one of 1,2,3,4,5,8");
  }
#undef RGCN_ROWS_CASE
  RGCN_HIP(c, e);
  if (giant) {
    CombineArgs f = ca;
    f.msg = nullptr; f.add = nullptr;
    RGCN_TRY(combine_giant_finish(c, f));
    if (a.colpart != nullptr)
      RGCN_TRY(column_sum_giant_rows(c, ca.out, a.colpart + (size_t)(grid / 8) * c->d));
  }
  return RGCN_OK;
}

}  // namespace rgcn
