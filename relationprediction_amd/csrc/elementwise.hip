// HBM-bound row kernels of the encoder:
//   * input layer  H0 = relu(W_emb + b_emb)            (code/encoders/affine_transform.py:63-83)
//   * combine      out = act( dropout(base) + sum_{slots of row} msg )   -- the segmented reduction
//     over incoming messages fused with the self-loop add, the self-loop dropout
//     (message_gcn.py:60-64) and the relu (gcn_basis_concat.py:73-81 / gcn_basis.py:78-86);
//     in the backward pass the same kernel sums the per-source message gradients, adds the
//     self-loop gradient, applies relu' of the layer below and emits the dropout-scaled copy the
//     self-loop GEMMs consume.
// One group of 64/128/256 lanes walks one row with float4 accesses (d % 4 == 0) so every
// global access is a full coalesced 16 B/lane stream; rows are independent => no atomics.
#include "rgcn_internal.h"

namespace rgcn {

namespace {

__device__ __forceinline__ float drop_scale(const DropSpec& ds, size_t idx) { return drop_factor(ds, idx); }

template <int VEC>
struct VecT;
template <>
struct VecT<4> {
  using type = float4;
};
template <>
struct VecT<1> {
  using type = float;
};

template <int VEC>
__device__ __forceinline__ void vload(const float* p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    v[0] = *p;
  }
}
template <int VEC>
__device__ __forceinline__ void vstore(float* p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    *p = v[0];
  }
}

// Epilogue shared by the row kernels: gate (relu'), relu, store, dropout-scaled second copy.
template <int VEC>
__device__ __forceinline__ void combine_epilogue(const CombineArgs& a, size_t off, float (&acc)[VEC]) {
  if (a.gate != nullptr) {
    float gt[VEC];
    vload<VEC>(a.gate + off, gt);
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = gt[k] > 0.0f ? acc[k] : 0.0f;
  }
  if (a.relu) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = fmaxf(acc[k], 0.0f);
  }
  vstore<VEC>(a.out + off, acc);
  if (a.out2 != nullptr) {
    float o2[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) o2[k] = acc[k] * drop_scale(a.drop2, off + k);
    vstore<VEC>(a.out2 + off, o2);
  }
}

// dropout(base) + add for one vector of one row
template <int VEC>
__device__ __forceinline__ void combine_prologue(const CombineArgs& a, int v, size_t off, float (&acc)[VEC]) {
  // no fused multiply-add across "dropout scale, then add": the single-pass layer kernel (block_rows.hip) computes the
  // same expression and both must round alike (the two forms are held bitwise equal by tests/test_gpu_parity.py)
#pragma clang fp contract(off)
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
  if (a.base != nullptr && v >= a.row_lo && v < a.row_hi) {
    vload<VEC>(a.base + off, acc);
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] *= drop_scale(a.drop, off + k);
  }
  if (a.add != nullptr) {
    float ad[VEC];
    vload<VEC>(a.add + off, ad);
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] += ad[k];
  }
  if (a.add_units != nullptr) {      // forward direction first, then backward: a fixed order
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
      const int32_t* up = a.unit_ptr + (size_t)dir * (a.V + 1) + v;
      const int u = up[0];
      if (up[1] > u) {
        float ad[VEC];
        vload<VEC>(a.add_units + ((size_t)dir * a.V + u) * a.d + (off - (size_t)v * a.d), ad);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] += ad[k];
      }
    }
  }
}

constexpr int kCombineThreads = 256;
constexpr int kLongBlocksMin = 64;  // extra workgroups of the combine grid that walk the long-row list

// Sum of the slots [beg, end) of one row at column vector cidx, by the whole workgroup: EIGHT interleaved slot lanes
// (lane q adds the slots beg + q, beg + q + 8, ... in increasing order), THREADS / 128 of them physical
// (threadIdx.x >> 7), each carrying VL = 8 / (THREADS / 128) virtual ones side by side (16 loads in flight per thread
// either way); the eight partial sums are then added in lane order ((0 + 1) + 2) ... -- the same arithmetic whatever
// THREADS is, and the one the single-pass layer kernel uses.  The result is valid in the threads with sl == 0 (and
// cidx < nvec).  red: [(THREADS / 128 - 1) * VL][128 * VEC] floats of LDS.
template <int VEC, int THREADS>
__device__ __forceinline__ void sum_long_row(const float* __restrict__ msg, int d, int nvec, int cidx, int beg, int end,
                                             float* __restrict__ red, float (&tot)[VEC]) {
  constexpr int NSL = THREADS / 128, VL = 8 / NSL, UNR = 16 / VL;
  const int cl = threadIdx.x & 127, sl = threadIdx.x >> 7;
  float part[VL][VEC];
#pragma unroll
  for (int j = 0; j < VL; ++j)
#pragma unroll
    for (int k = 0; k < VEC; ++k) part[j][k] = 0.0f;
  if (cidx < nvec) {
    const float* mp = msg + (size_t)cidx * VEC;
    for (int s0 = beg + sl * VL; s0 < end; s0 += 8 * UNR) {      // slot of virtual lane 0, round by round
      float m[UNR][VL][VEC];
#pragma unroll
      for (int u = 0; u < UNR; ++u)
#pragma unroll
        for (int j = 0; j < VL; ++j) {
          const int sidx = s0 + j + 8 * u;
          if (sidx < end) vload<VEC>(mp + (size_t)sidx * d, m[u][j]);
          else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) m[u][j][k] = 0.0f;
          }
        }
#pragma unroll
      for (int u = 0; u < UNR; ++u)
#pragma unroll
        for (int j = 0; j < VL; ++j)
          if (s0 + j + 8 * u < end) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) part[j][k] += m[u][j][k];
          }
    }
  }
  if (sl > 0) {
#pragma unroll
    for (int j = 0; j < VL; ++j)
#pragma unroll
      for (int k = 0; k < VEC; ++k) red[((size_t)((sl - 1) * VL + j) * 128 + cl) * VEC + k] = part[j][k];
  }
  __syncthreads();
  if (sl == 0) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      float t = part[0][k];
#pragma unroll
      for (int j = 1; j < VL; ++j) t += part[j][k];
#pragma unroll
      for (int q = 0; q < (NSL - 1) * VL; ++q) t += red[((size_t)q * 128 + cl) * VEC + k];
      tot[k] = t;
    }
  }
  __syncthreads();
}

// Workgroups [0, n_long_blocks): one LONG row (more than kLongRow slots) at a time, summed by the whole workgroup
// (sum_long_row); they come FIRST in the grid so that they start at t = 0 and finish under the cover of the ordinary
// rows -- a hub never serialises one wave group, and no extra launch.  Workgroups behind them: TPR lanes per row,
// THREADS / TPR rows per workgroup, long rows skipped.
// 256-thread workgroups: measured (forward / backward launch, us) 128: 33.9 / 38.0, 256: 27.4 / 33.0, 512: 29.7 / 35.5,
// 1024: 31 / 36-38 (finer scheduling grain;
// and a 256-thread workgroup with 8 KB of LDS finds room on a CU that two GEMM workgroups occupy, where a 1024-thread
// one needs the CU to itself).  Round 2 also had the column sums of the last backward combine (db_emb) gathered in
// this kernel, one partial row per 1024-thread workgroup: that launch took ~42 us against 31 + 7 for the small
// workgroups and a separate column-sum pass, and is gone.
template <int VEC, int TPR>
__global__ void __launch_bounds__(kCombineThreads) k_combine(CombineArgs a, int n_long_blocks) {
  constexpr int THREADS = kCombineThreads;
  const int nvec = a.d / VEC;
  constexpr int NSL = THREADS / 128, VL = 8 / NSL;
  __shared__ float shm[(NSL - 1) * VL * 128 * VEC];       // partial sums of the long rows' slot lanes
  if ((int)blockIdx.x < n_long_blocks) {
    const int cl = threadIdx.x & 127, sl = threadIdx.x >> 7;
    if (a.msg == nullptr) return;
    const int n = *a.nlong;
    for (int b = blockIdx.x; b < n; b += n_long_blocks) {
      const int v = a.long_rows[b];
      const int beg = a.row_ptr[v], end = a.row_ptr[v + 1];
      for (int c0 = 0; c0 < nvec; c0 += 128) {
        const int cidx = c0 + cl;
        float sum[VEC];
        sum_long_row<VEC, THREADS>(a.msg, a.d, nvec, cidx, beg, end, shm, sum);
        if (sl == 0 && cidx < nvec) {
          const size_t off = (size_t)v * a.d + (size_t)cidx * VEC;
          float tot[VEC];
          combine_prologue<VEC>(a, v, off, tot);
#pragma unroll
          for (int k = 0; k < VEC; ++k) tot[k] += sum[k];
          combine_epilogue<VEC>(a, off, tot);
        }
      }
    }
    // giant rows (full-graph scale): one kGiantRow-slot PIECE per turn, partial sum to the piece slab; the
    // finishing kernel adds the pieces of a row in order and applies prologue / epilogue
    if (a.ngiant != nullptr) {
      const int np = a.ngiant[1];
      for (int b = blockIdx.x; b < np; b += n_long_blocks) {
        const int v = a.piece_row[b];
        const int beg = a.row_ptr[v] + a.piece_k[b] * kGiantRow;
        const int end = min(a.row_ptr[v + 1], beg + kGiantRow);
        for (int c0 = 0; c0 < nvec; c0 += 128) {
          const int cidx = c0 + cl;
          float tot[VEC];
          sum_long_row<VEC, THREADS>(a.msg, a.d, nvec, cidx, beg, end, shm, tot);
          if (sl == 0 && cidx < nvec) vstore<VEC>(a.giant_slab + (size_t)b * a.d + (size_t)cidx * VEC, tot);
        }
      }
    }
    return;
  }
  const int rows_per_block = THREADS / TPR;
  const int local = ((int)blockIdx.x - n_long_blocks) * rows_per_block + threadIdx.x / TPR;
  const int v = a.v_begin + local;
  const int lane = threadIdx.x % TPR;
  bool active = local < (a.v_count < 0 ? a.V : a.v_count) && v < a.V;
  int beg = 0, end = 0;
  if (active && a.msg != nullptr) {
    beg = a.row_ptr[v];
    end = a.row_ptr[v + 1];
    if (end - beg > kLongRow) active = false;       // handled by a long-row workgroup of this same launch
  }
  if (!active) return;
  for (int c0 = 0; c0 < nvec; c0 += TPR) {
    const int cidx = c0 + lane;
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
    if (active && cidx < nvec) {
      const size_t off = (size_t)v * a.d + (size_t)cidx * VEC;
      combine_prologue<VEC>(a, v, off, acc);
      const float* mp = a.msg + (size_t)cidx * VEC;
      int s = beg;
      for (; s + 4 <= end; s += 4) {   // 4 independent 16-B loads in flight per lane
        float m0[VEC], m1[VEC], m2[VEC], m3[VEC];
        vload<VEC>(mp + (size_t)(s + 0) * a.d, m0);
        vload<VEC>(mp + (size_t)(s + 1) * a.d, m1);
        vload<VEC>(mp + (size_t)(s + 2) * a.d, m2);
        vload<VEC>(mp + (size_t)(s + 3) * a.d, m3);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = (((acc[k] + m0[k]) + m1[k]) + m2[k]) + m3[k];
      }
      for (; s < end; ++s) {
        float m0[VEC];
        vload<VEC>(mp + (size_t)s * a.d, m0);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] += m0[k];
      }
      combine_epilogue<VEC>(a, off, acc);             // acc = what went to `out`
    }
  }
}

// giant rows: prologue + pieces in piece order + epilogue
template <int VEC>
__global__ void __launch_bounds__(256) k_combine_giant_finish(CombineArgs a) {
  const int nvec = a.d / VEC;
  const int n = a.ngiant[0];
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const int v = a.giant_rows[i], first = a.giant_first[i], cnt = a.giant_cnt[i];
    for (int cidx = threadIdx.x; cidx < nvec; cidx += 256) {
      const size_t off = (size_t)v * a.d + (size_t)cidx * VEC;
      float tot[VEC], sum[VEC], part[VEC];
      combine_prologue<VEC>(a, v, off, tot);
#pragma unroll
      for (int k = 0; k < VEC; ++k) sum[k] = 0.0f;
      for (int p = 0; p < cnt; ++p) {
        vload<VEC>(a.giant_slab + (size_t)(first + p) * a.d + (size_t)cidx * VEC, part);
#pragma unroll
        for (int k = 0; k < VEC; ++k) sum[k] += part[k];
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) tot[k] += sum[k];
      combine_epilogue<VEC>(a, off, tot);
    }
  }
}

template <int VEC>
__global__ void k_input_fwd(const float* __restrict__ W, const float* __restrict__ b,
                            float* __restrict__ H, int64_t nvec_total, int d) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nvec_total; i += stride) {
    const int64_t off = i * VEC;
    const int col = (int)(off % d);
    float w[VEC], bb[VEC], h[VEC];
    vload<VEC>(W + off, w);
    vload<VEC>(b + col, bb);
#pragma unroll
    for (int k = 0; k < VEC; ++k) h[k] = fmaxf(w[k] + bb[k], 0.0f);
    vstore<VEC>(H + off, h);
  }
}

template <int VEC>
__global__ void k_scale_dropout(const float* __restrict__ in, float* __restrict__ out, int64_t nvec,
                                DropSpec ds) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const DropKey key = drop_key(ds);
  for (; i < nvec; i += stride) {
    const size_t off = (size_t)i * VEC;
    float v[VEC];
    vload<VEC>(in + off, v);
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] *= drop_factor(ds, key, off + k);
    vstore<VEC>(out + off, v);
  }
}

__global__ void k_relu_copy(const float* __restrict__ in, float* __restrict__ out, int64_t n, int relu) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float x = in[i];
    out[i] = relu ? fmaxf(x, 0.0f) : x;
  }
}

__global__ void k_materialize_mask(uint8_t* out, int64_t n, DropSpec ds) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const DropKey key = drop_key(ds);
  for (; i < n; i += stride) out[i] = drop_factor(ds, key, (size_t)i) != 0.0f ? 1 : 0;
}

// column sums of a [rows, cols] matrix, deterministic two-stage reduction.
// stage 1: block (bx, by): 64 column lanes x 4 row lanes; rows [by*32, by*32+32): every thread keeps
// 8 independent row loads in flight, the 4 row lanes combine through LDS -> part[by][col].
// stage 2: 64 column lanes x 16 part lanes per block sum the partials in a fixed order.
constexpr int kColRowsPerBlock = 32;
// stage 1: block by: rows [by * 32, by * 32 + 32); 128 column lanes (VEC columns each) x 2 row lanes; every thread adds its
// 16 rows 8 at a time (8 independent 16-byte loads in flight), the two row lanes meet in LDS -> part[by][cols].
template <int VEC>
__global__ void __launch_bounds__(256) k_colsum_part(const float* __restrict__ in, float* __restrict__ part,
                                                    int rows, int cols) {
  __shared__ float red[128 * VEC];
  const int cl = threadIdx.x & 127, rl = threadIdx.x >> 7;
  const int col = (blockIdx.x * 128 + cl) * VEC;
  const int r0 = blockIdx.y * kColRowsPerBlock + rl;          // rows r0, r0 + 2, ...
  float acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
  if (col < cols) {
    for (int r = r0; r < min(rows, (int)(blockIdx.y + 1) * kColRowsPerBlock); r += 16) {
      float m[8][VEC];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (r + 2 * u < rows && r + 2 * u < (int)(blockIdx.y + 1) * kColRowsPerBlock)
          vload<VEC>(in + (size_t)(r + 2 * u) * cols + col, m[u]);
        else {
#pragma unroll
          for (int k = 0; k < VEC; ++k) m[u][k] = 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] += m[u][k];
    }
  }
  if (rl == 1) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) red[cl * VEC + k] = acc[k];
  }
  __syncthreads();
  if (rl == 0 && col < cols) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] += red[cl * VEC + k];
    vstore<VEC>(part + (size_t)blockIdx.y * cols + col, acc);
  }
}
__global__ void __launch_bounds__(1024) k_colsum_final(const float* __restrict__ part, float* __restrict__ out,
                                                      int nparts, int cols) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < cols) {
    int p = pl;
    for (; p + 48 < nparts; p += 64) {
      const float* q = part + (size_t)p * cols + col;
      a0 += q[0]; a1 += q[(size_t)16 * cols]; a2 += q[(size_t)32 * cols]; a3 += q[(size_t)48 * cols];
    }
    for (; p < nparts; p += 16) a0 += part[(size_t)p * cols + col];
  }
  red[pl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (pl == 0 && col < cols) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][cl];
    out[col] = t;
  }
}

int grid_for(int64_t n, int threads) {
  int64_t b = (n + threads - 1) / threads;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

DropSpec make_drop(const rgcn_ctx* c, int layer, bool active) {
  DropSpec ds;
  ds.mode = DROP_NONE;
  ds.layer = (uint32_t)layer;
  ds.thresh = (uint32_t)((double)c->cfg.keep_prob * 16777216.0);
  ds.inv_keep = 1.0f / c->cfg.keep_prob;
  ds.seed = c->seed;
  ds.mask = nullptr;
  ds.seed_offset = c->capturing ? c->replay_counter : nullptr;
  if (active && c->fwd_train && layer >= 1 && layer <= c->L) {
    if (c->explicit_masks) {
      ds.mode = DROP_MASK;
      ds.mask = c->masks + (size_t)(layer - 1) * c->V * c->d;
    } else {
      ds.mode = DROP_RNG;
    }
  }
  return ds;
}

rgcn_status combine(rgcn_ctx* c, const char* tag, const CombineArgs& a_in, double alg_bytes) {
  if (a_in.V <= 0) return RGCN_OK;
  CombineArgs a = a_in;
  const bool giant = a.msg != nullptr && a.row_ptr == c->g.row_ptr && c->g.giant_on;
  if (giant) {
    if (!c->giant_slab) RGCN_HIP(c, hipMalloc((void**)&c->giant_slab, sizeof(float) * (size_t)c->g.piece_cap * c->d));
    a.giant_rows = c->g.giant_rows; a.giant_first = c->g.giant_first; a.giant_cnt = c->g.giant_cnt;
    a.piece_row = c->g.piece_row; a.piece_k = c->g.piece_k; a.ngiant = c->g.ngiant; a.giant_slab = c->giant_slab;
  }
  const bool vec4 = (a.d % 4 == 0) && aligned16(a.out) && aligned16(a.base) && aligned16(a.msg) && aligned16(a.add) &&
                    aligned16(a.gate) && aligned16(a.out2);
  const int nvec = vec4 ? a.d / 4 : a.d;
  const int tpr = nvec <= 64 ? 64 : (nvec <= 128 ? 128 : 256);
  const int nrows = a.v_count < 0 ? a.V : a.v_count;
  // enough long-row workgroups for the graph at hand: one per ~1024 slots, at least 64, at most 1024
  int n_long_blocks = 0;
  if (a.msg != nullptr) {
    int64_t want = 2 * c->g.E / 1024;
    n_long_blocks = (int)(want < kLongBlocksMin ? kLongBlocksMin : (want > 1024 ? 1024 : want));
  }
  n_long_blocks *= 4;       // (sized for 1024-thread workgroups: the same number of threads on the long rows)
  const int rows_per_block = kCombineThreads / tpr;
  const int nb_rows = (nrows + rows_per_block - 1) / rows_per_block;
  dim3 grid(nb_rows + n_long_blocks), block(kCombineThreads);
  ProfScope ps(c, tag, alg_bytes, 0);
#define RGCN_LAUNCH_COMBINE(VEC, TPR) \
  hipLaunchKernelGGL((k_combine<VEC, TPR>), grid, block, 0, c->stream, a, n_long_blocks)
  if (vec4) {
    if (tpr == 64) RGCN_LAUNCH_COMBINE(4, 64);
    else if (tpr == 128) RGCN_LAUNCH_COMBINE(4, 128);
    else RGCN_LAUNCH_COMBINE(4, 256);
  } else {
    if (tpr == 64) RGCN_LAUNCH_COMBINE(1, 64);
    else if (tpr == 128) RGCN_LAUNCH_COMBINE(1, 128);
    else RGCN_LAUNCH_COMBINE(1, 256);
  }
#undef RGCN_LAUNCH_COMBINE
  if (giant) {
    if (vec4) hipLaunchKernelGGL((k_combine_giant_finish<4>), dim3(64), dim3(256), 0, c->stream, a);
    else hipLaunchKernelGGL((k_combine_giant_finish<1>), dim3(64), dim3(256), 0, c->stream, a);
  }
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

rgcn_status combine_giant_finish(rgcn_ctx* c, const CombineArgs& a_in) {
  if (!c->g.giant_on) return RGCN_OK;
  CombineArgs a = a_in;
  if (!c->giant_slab) RGCN_FAIL(c, RGCN_ERR_STATE, "internal: giant-row pieces were never written");
  a.giant_rows = c->g.giant_rows; a.giant_first = c->g.giant_first; a.giant_cnt = c->g.giant_cnt;
  a.piece_row = c->g.piece_row; a.piece_k = c->g.piece_k; a.ngiant = c->g.ngiant; a.giant_slab = c->giant_slab;
  const bool vec4 = (a.d % 4 == 0) && aligned16(a.out) && aligned16(a.base) && aligned16(a.add) && aligned16(a.gate) &&
                    aligned16(a.out2);
  if (vec4) hipLaunchKernelGGL((k_combine_giant_finish<4>), dim3(64), dim3(256), 0, c->stream, a);
  else hipLaunchKernelGGL((k_combine_giant_finish<1>), dim3(64), dim3(256), 0, c->stream, a);
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

rgcn_status input_forward(rgcn_ctx* c) {
  const int64_t n = (int64_t)c->V * c->d;
  ProfScope ps(c, "input_fwd", 8.0 * n, 0);
  if (c->d % 4 == 0) {
    hipLaunchKernelGGL((k_input_fwd<4>), dim3(grid_for(n / 4, 256)), dim3(256), 0, c->stream, c->w_emb,
                       c->b_emb, c->H[0], n / 4, c->d);
  } else {
    hipLaunchKernelGGL((k_input_fwd<1>), dim3(grid_for(n, 256)), dim3(256), 0, c->stream, c->w_emb,
                       c->b_emb, c->H[0], n, c->d);
  }
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

rgcn_status scale_dropout(rgcn_ctx* c, const float* in, float* out, const DropSpec& ds) {
  const int64_t n = (int64_t)c->V * c->d;
  ProfScope ps(c, "top_grad_dropout", 8.0 * n, 0);
  if (n % 4 == 0 && aligned16(in) && aligned16(out))
    hipLaunchKernelGGL((k_scale_dropout<4>), dim3(grid_for(n / 4, 256)), dim3(256), 0, c->stream, in, out, n / 4, ds);
  else
    hipLaunchKernelGGL((k_scale_dropout<1>), dim3(grid_for(n, 256)), dim3(256), 0, c->stream, in, out, n, ds);
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

rgcn_status relu_copy(rgcn_ctx* c, const float* in, float* out, int64_t n, int relu) {
  ProfScope ps(c, "activation", 8.0 * n, 0);
  hipLaunchKernelGGL(k_relu_copy, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, in, out, n, relu);
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

rgcn_status materialize_mask(rgcn_ctx* c, const DropSpec& ds, uint8_t* out_dev, int64_t n) {
  hipLaunchKernelGGL(k_materialize_mask, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, out_dev, n, ds);
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

// part row of a giant row = its rank by vertex id among the giant rows (the list itself is in registration order, which
// differs from run to run; there are a handful of them at most)
__global__ void k_colsum_giant_rows(const float* __restrict__ out, const int32_t* __restrict__ giant_rows,
                                    const int32_t* __restrict__ ngiant, int cap, int d, float* __restrict__ part) {
  const int i = blockIdx.x;
  if (i >= cap) return;
  const int n = min(ngiant[0], cap);
  if (i >= n) {
    for (int k = threadIdx.x; k < d; k += blockDim.x) part[(size_t)i * d + k] = 0.0f;
    return;
  }
  const int v = giant_rows[i];
  int rank = 0;
  for (int k = 0; k < n; ++k) rank += giant_rows[k] < v ? 1 : 0;
  const float* row = out + (size_t)v * d;
  for (int k = threadIdx.x; k < d; k += blockDim.x) part[(size_t)rank * d + k] = row[k];
}

rgcn_status column_sum_giant_rows(rgcn_ctx* c, const float* out, float* part) {
  if (!c->g.giant_on || c->g.giant_cap <= 0) return RGCN_OK;
  hipLaunchKernelGGL(k_colsum_giant_rows, dim3((unsigned)c->g.giant_cap), dim3(256), 0, c->stream, out, c->g.giant_rows,
                     c->g.ngiant, c->g.giant_cap, c->d, part);
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

rgcn_status column_sum_finish(rgcn_ctx* c, float* out, int nparts, int cols) {
  ProfScope ps(c, "bias_grad_colsum", 4.0 * nparts * cols, 0);
  hipLaunchKernelGGL(k_colsum_final, dim3((cols + 63) / 64), dim3(1024), 0, c->stream, c->colsum_part, out, nparts, cols);
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

rgcn_status column_sum(rgcn_ctx* c, const float* in, float* out, int rows, int cols) {
  float* part = c->colsum_part;
  const int nparts = (rows + kColRowsPerBlock - 1) / kColRowsPerBlock;
  if ((size_t)nparts * cols > c->colsum_part_floats) RGCN_FAIL(c, RGCN_ERR_STATE, "internal: column-sum scratch too small");
  ProfScope ps(c, "bias_grad_colsum", 4.0 * rows * cols, 0);
  if (cols % 4 == 0 && aligned16(in))
    hipLaunchKernelGGL((k_colsum_part<4>), dim3((cols / 4 + 127) / 128, nparts), dim3(256), 0, c->stream, in, part, rows,
                       cols);
  else
    hipLaunchKernelGGL((k_colsum_part<1>), dim3((cols + 127) / 128, nparts), dim3(256), 0, c->stream, in, part, rows,
                       cols);
  hipLaunchKernelGGL(k_colsum_final, dim3((cols + 63) / 64), dim3(1024), 0, c->stream,
                     part, out, nparts, cols);
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

}  // namespace rgcn
