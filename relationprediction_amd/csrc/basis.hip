// Basis-decomposition relational messages (BasisGcn, code/encoders/message_gcns/gcn_basis.py:39-68)
// and their gradients, in the aggregate-first form.
//
// Reference dataflow: T = H[s] . W.reshape(d, B.d)  ([E,d] x [d,B.d], 2 x 37.5 GFLOP at B = 5 on a
// 15,000-edge graph), scale by C[type] ([E,B,d] intermediates, 150 MB), reduce over B, then the
// [V,E] x [E,d] sparse product.  Because the coefficient is a per-edge scalar, aggregation commutes
// with the basis contraction:
//     Z[(v,dir),b,:] = sum_{messages m -> v of direction dir} n_m C[rel_m,b] H[src_m,:]     (HBM-bound gather)
//     pre-activation[v] = dropout(H.W_self)[v] + sum_dir Z[(v,dir)] . W'_dir     ([units, B.d] x [B.d, d] on the MFMA GEMM)
// so the dense work is done once per (row, direction) UNIT that receives a message instead of once per edge, and no
// [E,B,d] tensor ever exists.  Like the reference, which works per edge (gcn_basis.py:39-46), nothing is computed for
// a vertex without messages: Z, dZ and the operand of dW' are COMPACTED over the units (GraphBufs::unit_ptr /
// unit_rows, built by the graph preparation; on the FB15k-237 minibatch 5,370 + 7,082 of 2 x 14,541), one group of the
// batched GEMM per direction with its row count read on the device (GemmBatch).  Layout: Zc, dZc [2][V][B.d] (V = the
// capacity of a direction), products and gathered upstream rows [2][V][d].
// Backward:  dZ[(v,dir)] = D[v] . W'_dir^T,  dW'_dir = Zc_dir^T . D[rows of the units]  (GEMMs),
//     dC[rel,b]  = sum_{m: rel_m = rel} n_m <H[src_m], dZ[(dst_m,dir),b,:]>        (per-relation chunks)
//     dH[u]     += sum_{m: src_m = u} n_m sum_b C[rel_m,b] dZ[(dst_m,dir),b,:]      (source-major gather,
//                  fused with the self-loop gradient add, relu' and the dropout-scaled copy).
#include "rgcn_internal.h"

namespace rgcn {

namespace {

constexpr int BT = 8;   // basis functions handled per launch (register budget); B > 8 loops on the host

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
  if constexpr (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  else *p = v[0];
}

__device__ __forceinline__ float drop_scale(const DropSpec& ds, size_t idx) { return drop_factor(ds, idx); }

constexpr int kRowThreads = 1024;
// leading workgroups of a row launch that walk the long-row list: 64 at minibatch scale, 512 at full-graph scale (the
// 272,115-edge training graph has 5,000+ long rows and a 2,397-slot hub)
inline int long_blocks(const rgcn_ctx* c) { return 2 * c->g.E > 65536 ? 512 : 64; }

struct AggArgs {
  const float* Hin;          // [V,d]
  float* Z;                  // compacted [2][V][B*d]
  const int32_t* unit_ptr;   // [2][V+1]
  const int32_t* row_ptr;    // incidence CSR (rows = destinations)
  const int32_t* d_src;      // per slot: source vertex, directed relation, normalisation
  const int32_t* d_rel;
  const float* d_norm;
  const float* coef;         // [2R][B]
  const int32_t* long_rows;
  const int32_t* nlong;
  int32_t V, d, B, R, b0, nbt;
};

// accumulate one slot into the forward- or backward-direction accumulators (wave-uniform branch)
template <int VEC>
__device__ __forceinline__ void agg_entry(const AggArgs& a, int rel, float nrm, const float (&x)[VEC],
                                          float (&accf)[BT][VEC], float (&accb)[BT][VEC]) {
  const float* cf = a.coef + (size_t)rel * a.B + a.b0;
  if (rel < a.R) {
#pragma unroll
    for (int b = 0; b < BT; ++b)
      if (b < a.nbt) {
        const float w = nrm * cf[b];
#pragma unroll
        for (int k = 0; k < VEC; ++k) accf[b][k] = fmaf(w, x[k], accf[b][k]);
      }
  } else {
#pragma unroll
    for (int b = 0; b < BT; ++b)
      if (b < a.nbt) {
        const float w = nrm * cf[b];
#pragma unroll
        for (int k = 0; k < VEC; ++k) accb[b][k] = fmaf(w, x[k], accb[b][k]);
      }
  }
}

// slots [s0, s1) with stride `step` of one row, for one vector column: 4 gathers in flight
template <int VEC>
__device__ __forceinline__ void agg_range(const AggArgs& a, int s0, int s1, int step, int cidx,
                                          float (&accf)[BT][VEC], float (&accb)[BT][VEC]) {
  int s = s0;
  for (; s + 3 * step < s1; s += 4 * step) {
    int src[4], rel[4];
    float nrm[4], x[4][VEC];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      src[u] = a.d_src[s + u * step]; rel[u] = a.d_rel[s + u * step]; nrm[u] = a.d_norm[s + u * step];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) vload<VEC>(a.Hin + (size_t)src[u] * a.d + (size_t)cidx * VEC, x[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) agg_entry<VEC>(a, rel[u], nrm[u], x[u], accf, accb);
  }
  for (; s < s1; s += step) {
    float x[VEC];
    vload<VEC>(a.Hin + (size_t)a.d_src[s] * a.d + (size_t)cidx * VEC, x);
    agg_entry<VEC>(a, a.d_rel[s], a.d_norm[s], x, accf, accb);
  }
}

// Workgroups [0, n_long_blocks): one LONG row at a time, 8 slot-lanes x 128 column lanes, partial sums
// combined through LDS in a fixed order.  The others: TPR lanes per destination row, 1024/TPR rows per
// workgroup.  Forward-direction messages (rel < R) feed the first B column blocks, backward the last B.
template <int VEC, int TPR>
__global__ void __launch_bounds__(kRowThreads) k_basis_agg(AggArgs a, int n_long_blocks) {
  const int nvec = a.d / VEC;
  const size_t zstride = (size_t)a.B * a.d;
  if ((int)blockIdx.x < n_long_blocks) {
    __shared__ float red[8][128 * VEC];
    const int cl = threadIdx.x & 127, sl = threadIdx.x >> 7;
    const int n = *a.nlong;
    for (int lb = blockIdx.x; lb < n; lb += n_long_blocks) {
      const int v = a.long_rows[lb];
      const int beg = a.row_ptr[v], end = a.row_ptr[v + 1];
      for (int c0 = 0; c0 < nvec; c0 += 128) {
        const int cidx = c0 + cl;
        float accf[BT][VEC], accb[BT][VEC];
#pragma unroll
        for (int b = 0; b < BT; ++b)
#pragma unroll
          for (int k = 0; k < VEC; ++k) { accf[b][k] = 0.f; accb[b][k] = 0.f; }
        if (cidx < nvec) agg_range<VEC>(a, beg + sl, end, 8, cidx, accf, accb);
#pragma unroll
        for (int q = 0; q < 2 * BT; ++q) {
          const int b = q % BT;
          if (b < a.nbt) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) red[sl][cl * VEC + k] = q < BT ? accf[b][k] : accb[b][k];
            __syncthreads();
            if (sl == 0 && cidx < nvec) {
              float t[VEC];
#pragma unroll
              for (int k = 0; k < VEC; ++k) {
                float u = red[0][cl * VEC + k];
#pragma unroll
                for (int w = 1; w < 8; ++w) u += red[w][cl * VEC + k];
                t[k] = u;
              }
              const int dir = q < BT ? 0 : 1;
              const int32_t* up = a.unit_ptr + (size_t)dir * (a.V + 1) + v;
              const int u = up[0];
              if (up[1] > u)
                vstore<VEC>(a.Z + ((size_t)dir * a.V + u) * zstride + (size_t)(a.b0 + b) * a.d + (size_t)cidx * VEC, t);
            }
            __syncthreads();
          }
        }
      }
    }
    return;
  }
  const int v = ((int)blockIdx.x - n_long_blocks) * (kRowThreads / TPR) + threadIdx.x / TPR;
  if (v >= a.V) return;
  const int lane = threadIdx.x % TPR;
  const int beg = a.row_ptr[v], end = a.row_ptr[v + 1];
  if (end == beg || end - beg > kLongRow) return;      // no unit at all / a long-row workgroup of this launch owns it
  const int uf = a.unit_ptr[v], ub = a.unit_ptr[(size_t)a.V + 1 + v];
  const bool has_f = a.unit_ptr[v + 1] > uf, has_b = a.unit_ptr[(size_t)a.V + 2 + v] > ub;
  float* zf = a.Z + (size_t)uf * zstride + (size_t)a.b0 * a.d;
  float* zb = a.Z + ((size_t)a.V + ub) * zstride + (size_t)a.b0 * a.d;
  for (int cidx = lane; cidx < nvec; cidx += TPR) {
    float accf[BT][VEC], accb[BT][VEC];
#pragma unroll
    for (int b = 0; b < BT; ++b)
#pragma unroll
      for (int k = 0; k < VEC; ++k) { accf[b][k] = 0.f; accb[b][k] = 0.f; }
    agg_range<VEC>(a, beg, end, 1, cidx, accf, accb);
#pragma unroll
    for (int b = 0; b < BT; ++b)
      if (b < a.nbt) {
        if (has_f) vstore<VEC>(zf + (size_t)b * a.d + (size_t)cidx * VEC, accf[b]);
        if (has_b) vstore<VEC>(zb + (size_t)b * a.d + (size_t)cidx * VEC, accb[b]);
      }
  }
}

// Dc[dir][i][:] = D[unit_rows[dir][i]][:] for the units of both directions: the row operand of dZ = D.W'^T and the
// K operand of dW' = Zc^T.D, compacted like Zc.  One 128-lane group per unit; groups beyond a direction's count leave.
template <int VEC>
__global__ void __launch_bounds__(256) k_gather_units(const float* __restrict__ D, const int32_t* __restrict__ unit_ptr,
                                                      const int32_t* __restrict__ unit_rows, float* __restrict__ Dc,
                                                      int V, int d) {
  const int i = blockIdx.x * 2 + (threadIdx.x >> 7), dir = blockIdx.y;
  if (i >= unit_ptr[(size_t)dir * (V + 1) + V]) return;
  const int v = unit_rows[(size_t)dir * V + i];
  const float* src = D + (size_t)v * d;
  float* dst = Dc + ((size_t)dir * V + i) * d;
  for (int c = (threadIdx.x & 127) * VEC; c < d; c += 128 * VEC) {
    float t[VEC];
    vload<VEC>(src + c, t);
    vstore<VEC>(dst + c, t);
  }
}

struct BwdGatherArgs {
  const float* dZ;           // compacted [2][V][B*d]
  const int32_t* unit_ptr;   // [2][V+1]
  int32_t V;
  const int32_t* row_ptr;    // incidence CSR (rows = sources); nullptr: no relational part
  const int32_t* s_dst;      // per source-order slot: destination vertex, directed relation, normalisation
  const int32_t* s_rel;
  const float* s_norm;
  const float* coef;         // [2R][B]
  const int32_t* long_rows;
  const int32_t* nlong;
  int32_t B, R;
  CombineArgs c;             // epilogue: out = (base + gathered) * gate ; out2 = out * dropout
};

// acc += sum over slots [s0, s1) step `step` of  n * sum_b C[rel,b] * dZ[dst, dir, b, :]
template <int VEC>
__device__ __forceinline__ void gather_range(const BwdGatherArgs& a, int s0, int s1, int step, int d,
                                             int cidx, float (&acc)[VEC]) {
  for (int s = s0; s < s1; s += 2 * step) {
    const bool two = s + step < s1;
    const int dst0 = a.s_dst[s], rel0 = a.s_rel[s];
    const float n0 = a.s_norm[s];
    const int dst1 = two ? a.s_dst[s + step] : dst0, rel1 = two ? a.s_rel[s + step] : rel0;
    const float n1 = two ? a.s_norm[s + step] : 0.f;
    // the unit of (destination, direction): it exists, this very message lands there
    const int dir0 = rel0 < a.R ? 0 : 1, dir1 = rel1 < a.R ? 0 : 1;
    const int u0 = a.unit_ptr[(size_t)dir0 * (a.V + 1) + dst0], u1 = a.unit_ptr[(size_t)dir1 * (a.V + 1) + dst1];
    const float* z0 = a.dZ + ((size_t)dir0 * a.V + u0) * a.B * d + (size_t)cidx * VEC;
    const float* z1 = a.dZ + ((size_t)dir1 * a.V + u1) * a.B * d + (size_t)cidx * VEC;
    const float* c0 = a.coef + (size_t)rel0 * a.B;
    const float* c1 = a.coef + (size_t)rel1 * a.B;
    for (int b = 0; b < a.B; ++b) {
      float v0[VEC], v1[VEC];
      vload<VEC>(z0 + (size_t)b * d, v0);
      vload<VEC>(z1 + (size_t)b * d, v1);
      const float w0 = n0 * c0[b], w1 = n1 * c1[b];
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] = fmaf(w1, v1[k], fmaf(w0, v0[k], acc[k]));
    }
  }
}

template <int VEC>
__device__ __forceinline__ void gather_epilogue(const BwdGatherArgs& a, int v, size_t off, float (&acc)[VEC]) {
  if (a.c.base != nullptr && v >= a.c.row_lo && v < a.c.row_hi) {
    float bs[VEC];
    vload<VEC>(a.c.base + off, bs);
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] += bs[k];
  }
  if (a.c.gate != nullptr) {
    float gt[VEC];
    vload<VEC>(a.c.gate + off, gt);
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = gt[k] > 0.f ? acc[k] : 0.f;
  }
  vstore<VEC>(a.c.out + off, acc);
  if (a.c.out2 != nullptr) {
    float o2[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) o2[k] = acc[k] * drop_scale(a.c.drop2, off + k);
    vstore<VEC>(a.c.out2 + off, o2);
  }
}

template <int VEC, int TPR>
__global__ void __launch_bounds__(kRowThreads) k_basis_bwd_gather(BwdGatherArgs a, int n_long_blocks) {
  const int d = a.c.d;
  const int nvec = d / VEC;
  if ((int)blockIdx.x < n_long_blocks) {
    __shared__ float red[8][128 * VEC];
    const int cl = threadIdx.x & 127, sl = threadIdx.x >> 7;
    const int n = *a.nlong;
    for (int lb = blockIdx.x; lb < n; lb += n_long_blocks) {
      const int v = a.long_rows[lb];
      const int beg = a.row_ptr[v], end = a.row_ptr[v + 1];
      for (int c0 = 0; c0 < nvec; c0 += 128) {
        const int cidx = c0 + cl;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        if (cidx < nvec) gather_range<VEC>(a, beg + sl, end, 8, d, cidx, acc);
#pragma unroll
        for (int k = 0; k < VEC; ++k) red[sl][cl * VEC + k] = acc[k];
        __syncthreads();
        if (sl == 0 && cidx < nvec) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            float u = red[0][cl * VEC + k];
#pragma unroll
            for (int w = 1; w < 8; ++w) u += red[w][cl * VEC + k];
            acc[k] = u;
          }
          gather_epilogue<VEC>(a, v, (size_t)v * d + (size_t)cidx * VEC, acc);
        }
        __syncthreads();
      }
    }
    return;
  }
  const int v = ((int)blockIdx.x - n_long_blocks) * (kRowThreads / TPR) + threadIdx.x / TPR;
  if (v >= a.c.V) return;
  const int lane = threadIdx.x % TPR;
  int beg = 0, end = 0;
  if (a.row_ptr != nullptr) {
    beg = a.row_ptr[v];
    end = a.row_ptr[v + 1];
    if (end - beg > kLongRow) return;
  }
  for (int cidx = lane; cidx < nvec; cidx += TPR) {
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    gather_range<VEC>(a, beg, end, 1, d, cidx, acc);
    gather_epilogue<VEC>(a, v, (size_t)v * d + (size_t)cidx * VEC, acc);
  }
}

struct DcoefArgs {
  const float* Hin;
  const float* dZ;           // compacted [2][V][B*d]
  const int32_t* unit_ptr;   // [2][V+1]
  int32_t V;
  const int32_t* m_src;
  const int32_t* m_dst;
  const float* m_norm;
  const int32_t* rel_ptr;
  const int32_t* chunk_ptr;
  float* slab;               // [chunks][B]
  int32_t R, B, d, chunk;
};

__device__ __forceinline__ int find_segment(const int32_t* __restrict__ ptr, int n_seg, int x) {
  int lo = 0, hi = n_seg;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (ptr[mid] <= x) lo = mid; else hi = mid;
  }
  return lo;
}

// One workgroup (4 waves) per relation chunk; wave w takes messages beg+w, beg+w+4, ...; the wave's
// lanes split the d features, reduce the B dot products by shuffles, partial sums meet in LDS.
template <int VEC>
__global__ void __launch_bounds__(256) k_basis_dcoef(DcoefArgs a) {
  __shared__ float red[4][64];
  const int bid = blockIdx.x;
  const int R2 = 2 * a.R;
  if (bid >= a.chunk_ptr[R2]) return;
  const int rel = find_segment(a.chunk_ptr, R2, bid);
  const int beg = a.rel_ptr[rel] + (bid - a.chunk_ptr[rel]) * a.chunk;
  const int end = min(beg + a.chunk, a.rel_ptr[rel + 1]);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dir = rel < a.R ? 0 : 1;
  const int32_t* up = a.unit_ptr + (size_t)dir * (a.V + 1);
  const int nvec = a.d / VEC;
  for (int b0 = 0; b0 < a.B; b0 += BT) {
    float part[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b) part[b] = 0.f;
    for (int j = beg + wave; j < end; j += 4) {
      const float nrm = a.m_norm[j];
      const float* xp = a.Hin + (size_t)a.m_src[j] * a.d;
      const float* zp = a.dZ + (((size_t)dir * a.V + up[a.m_dst[j]]) * a.B + b0) * a.d;
      for (int cidx = lane; cidx < nvec; cidx += 64) {
        float x[VEC];
        vload<VEC>(xp + (size_t)cidx * VEC, x);
#pragma unroll
        for (int b = 0; b < BT; ++b)
          if (b0 + b < a.B) {
            float z[VEC];
            vload<VEC>(zp + (size_t)b * a.d + (size_t)cidx * VEC, z);
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < VEC; ++k) t = fmaf(x[k], z[k], t);
            part[b] = fmaf(nrm, t, part[b]);
          }
      }
    }
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      float t = part[b];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
      if (lane == 0) red[wave][b] = t;
    }
    __syncthreads();
    if (threadIdx.x < BT && b0 + (int)threadIdx.x < a.B)
      a.slab[(size_t)bid * a.B + b0 + threadIdx.x] =
          ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    __syncthreads();
  }
}

__global__ void k_basis_dcoef_reduce(const float* __restrict__ slab, const int32_t* __restrict__ chunk_ptr,
                                     float* __restrict__ gcoef, int R2, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R2 * B) return;
  const int rel = i / B, b = i - rel * B;
  // a popular relation is cut into hundreds of chunks whose partial sums cancel: compensated (Kahan) summation in
  // chunk order keeps the coefficient gradient at the fp32 oracle's distance from float64 (test_float64_tie_break)
  float acc = 0.f, comp = 0.f;
  for (int c = chunk_ptr[rel]; c < chunk_ptr[rel + 1]; ++c) {
    const float y = slab[(size_t)c * B + b] - comp;
    const float t = acc + y;
    comp = (t - acc) - y;
    acc = t;
  }
  gcoef[i] = acc;
}

// host [d][B][d] (in, basis, out)  <->  device [B][d][d]
__global__ void k_basis_transpose(const float* __restrict__ in, float* __restrict__ out, int d, int B,
                                  int to_device) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = (int64_t)B * d * d;
  if (i >= n) return;
  const int k = (int)(i % d);
  const int j = (int)((i / d) % d);
  const int b = (int)(i / ((int64_t)d * d));
  const int64_t h = ((int64_t)j * B + b) * d + k;
  if (to_device) out[i] = in[h]; else out[h] = in[i];
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

// Number of (row, direction) units of the current graph, for the profile accounting only (the kernels read the counts
// on the device).  Exact while the per-kernel profile is on (one readback per graph), the bound min(messages, 2 V)
// otherwise.
double basis_units(rgcn_ctx* c) {
  const double cap = 2.0 * c->V;
  if (c->prof_on && !c->capturing) {
    if (c->g.units_host < 0) {
      int32_t n[2] = {0, 0};
      // (the unit lists come from the graph preparation: on the prefetch stream, or in line on the MAIN stream -- which is
      // not c->stream when this accounting runs inside a side-stream scope.  One readback per graph, profile runs only: the
      // step that contains it is perturbed, which is why bench.py profiles in a pass of its own.)
      if (c->pf_stream) (void)hipStreamSynchronize(c->pf_stream);
      (void)hipStreamSynchronize(c->main_stream);
      if (c->stream != c->main_stream) (void)hipStreamSynchronize(c->stream);
      if (hipMemcpy(&n[0], c->g.unit_ptr + c->V, sizeof(int32_t), hipMemcpyDeviceToHost) == hipSuccess &&
          hipMemcpy(&n[1], c->g.unit_ptr + 2 * c->V + 1, sizeof(int32_t), hipMemcpyDeviceToHost) == hipSuccess)
        c->g.units_host = (int64_t)n[0] + n[1];
    }
    if (c->g.units_host >= 0) return (double)c->g.units_host;
  }
  const double M = 2.0 * c->g.E / c->world;
  return M < cap ? M : cap;
}

rgcn_status basis_aggregate_forward(rgcn_ctx* c, int layer, const float* Hin, float* Z) {
  AggArgs a;
  a.Hin = Hin; a.Z = Z; a.row_ptr = c->g.row_ptr; a.d_src = c->g.d_src; a.d_rel = c->g.d_rel;
  a.d_norm = c->g.d_norm; a.coef = c->layers[layer].coef; a.long_rows = c->g.long_rows; a.nlong = c->g.nlong;
  a.unit_ptr = c->g.unit_ptr;
  a.V = c->V; a.d = c->d; a.B = c->B; a.R = c->R;
  const bool vec4 = (c->d % 4 == 0) && aligned16(Hin) && aligned16(Z);
  const int nvec = vec4 ? c->d / 4 : c->d;
  const int tpr = nvec <= 64 ? 64 : (nvec <= 128 ? 128 : 256);
  const int rpb = kRowThreads / tpr;
  const int nlb = long_blocks(c);
  dim3 grid(nlb + (c->V + rpb - 1) / rpb), block(kRowThreads);
  const double M = 2.0 * c->g.E / c->world;
  for (int b0 = 0; b0 < c->B; b0 += BT) {
    a.b0 = b0;
    a.nbt = c->B - b0 < BT ? c->B - b0 : BT;
    const double rows = M < c->V ? M : (double)c->V;     // compulsory: each gathered row of H once
    const double units = basis_units(c);
    ProfScope ps(c, "basis_aggregate", 4.0 * c->d * (M + a.nbt * units) + 20.0 * M, 4.0 * M * a.nbt * c->d,
                 4.0 * c->d * (rows + a.nbt * units) + 20.0 * M);
#define RGCN_LAUNCH_AGG(VEC, TPR) \
  hipLaunchKernelGGL((k_basis_agg<VEC, TPR>), grid, block, 0, c->stream, a, nlb)
    if (vec4) {
      if (tpr == 64) RGCN_LAUNCH_AGG(4, 64); else if (tpr == 128) RGCN_LAUNCH_AGG(4, 128); else RGCN_LAUNCH_AGG(4, 256);
    } else {
      if (tpr == 64) RGCN_LAUNCH_AGG(1, 64); else if (tpr == 128) RGCN_LAUNCH_AGG(1, 128); else RGCN_LAUNCH_AGG(1, 256);
    }
#undef RGCN_LAUNCH_AGG
    RGCN_HIP(c, hipGetLastError());
  }
  return RGCN_OK;
}

rgcn_status basis_backward_gather(rgcn_ctx* c, int layer, const float* dZ, const CombineArgs& ca,
                                  bool with_messages) {
  BwdGatherArgs a;
  a.dZ = dZ;
  a.row_ptr = with_messages ? c->g.row_ptr : nullptr;
  a.s_dst = c->g.s_dst; a.s_rel = c->g.s_rel; a.s_norm = c->g.s_norm;
  a.coef = c->layers[layer].coef; a.long_rows = c->g.long_rows; a.nlong = c->g.nlong;
  a.unit_ptr = c->g.unit_ptr; a.V = c->V;
  a.B = c->B; a.R = c->R; a.c = ca;
  const bool vec4 = (c->d % 4 == 0) && aligned16(dZ) && aligned16(ca.out) && aligned16(ca.base) &&
                    aligned16(ca.gate) && aligned16(ca.out2);
  const int nvec = vec4 ? c->d / 4 : c->d;
  const int tpr = nvec <= 64 ? 64 : (nvec <= 128 ? 128 : 256);
  const int rpb = kRowThreads / tpr;
  const int nlb = with_messages ? long_blocks(c) : 0;
  dim3 grid(nlb + (c->V + rpb - 1) / rpb), block(kRowThreads);
  const double M = 2.0 * c->g.E / c->world;
  const double units = basis_units(c);
  // compulsory: dZ of every unit once, base / gate / out / out2 once each
  ProfScope ps(c, "basis_bwd_gather", 4.0 * c->d * (M * c->B + 4.0 * c->V) + 20.0 * M, 2.0 * M * c->B * c->d,
               4.0 * c->d * (c->B * units + 4.0 * c->V) + 20.0 * M);
#define RGCN_LAUNCH_BG(VEC, TPR) \
  hipLaunchKernelGGL((k_basis_bwd_gather<VEC, TPR>), grid, block, 0, c->stream, a, nlb)
  if (vec4) {
    if (tpr == 64) RGCN_LAUNCH_BG(4, 64); else if (tpr == 128) RGCN_LAUNCH_BG(4, 128); else RGCN_LAUNCH_BG(4, 256);
  } else {
    if (tpr == 64) RGCN_LAUNCH_BG(1, 64); else if (tpr == 128) RGCN_LAUNCH_BG(1, 128); else RGCN_LAUNCH_BG(1, 256);
  }
#undef RGCN_LAUNCH_BG
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

rgcn_status basis_dcoef(rgcn_ctx* c, int layer, const float* Hin, const float* dZ) {
  const int R2 = 2 * c->R;
  if (c->g.E > 0) {
    const int nchunks = (int)((2 * c->g.E + c->g.chunk - 1) / c->g.chunk) + R2;
    if ((size_t)nchunks * c->B > c->slab_dw_floats) RGCN_FAIL(c, RGCN_ERR_STATE, "internal: dC slab too small");
    DcoefArgs a;
    a.Hin = Hin; a.dZ = dZ; a.m_src = c->g.m_src; a.m_dst = c->g.m_dst; a.m_norm = c->g.m_norm;
    a.unit_ptr = c->g.unit_ptr; a.V = c->V;
    a.rel_ptr = c->g.rel_ptr; a.chunk_ptr = c->g.chunk_ptr; a.slab = c->slab_dw;
    a.R = c->R; a.B = c->B; a.d = c->d; a.chunk = c->g.chunk;
    const double M = 2.0 * c->g.E / c->world;
    const double rows = M < c->V ? M : (double)c->V;
    const double units = basis_units(c);
    ProfScope ps(c, "basis_dcoef", 4.0 * c->d * M * (1.0 + c->B), 2.0 * M * c->B * c->d,
                 4.0 * c->d * (rows + c->B * units) + 16.0 * M);
    if (c->d % 4 == 0 && aligned16(Hin) && aligned16(dZ))
      hipLaunchKernelGGL((k_basis_dcoef<4>), dim3(nchunks), dim3(256), 0, c->stream, a);
    else
      hipLaunchKernelGGL((k_basis_dcoef<1>), dim3(nchunks), dim3(256), 0, c->stream, a);
    RGCN_HIP(c, hipGetLastError());
  }
  {
    const int n = R2 * c->B;
    ProfScope ps(c, "basis_dcoef_reduce", 8.0 * n, 0);
    hipLaunchKernelGGL(k_basis_dcoef_reduce, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->slab_dw,
                       c->g.chunk_ptr, c->layers[layer].gcoef, R2, c->B);
    RGCN_HIP(c, hipGetLastError());
  }
  return RGCN_OK;
}

rgcn_status basis_gather_units(rgcn_ctx* c, const float* D, float* Dc) {
  const double units = basis_units(c);
  ProfScope ps(c, "basis_gather_units", 8.0 * c->d * units + 8.0 * units, 0);
  dim3 grid((unsigned)((c->V + 1) / 2), 2), block(256);
  if (c->d % 4 == 0 && aligned16(D) && aligned16(Dc))
    hipLaunchKernelGGL((k_gather_units<4>), grid, block, 0, c->stream, D, c->g.unit_ptr, c->g.unit_rows, Dc, c->V, c->d);
  else
    hipLaunchKernelGGL((k_gather_units<1>), grid, block, 0, c->stream, D, c->g.unit_ptr, c->g.unit_rows, Dc, c->V, c->d);
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

rgcn_status basis_to_device_layout(rgcn_ctx* c, const float* host_layout_dev, float* dst) {
  const int64_t n = (int64_t)c->B * c->d * c->d;
  hipLaunchKernelGGL(k_basis_transpose, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                     host_layout_dev, dst, c->d, c->B, 1);
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

rgcn_status basis_from_device_layout(rgcn_ctx* c, const float* src, float* host_layout_dev) {
  const int64_t n = (int64_t)c->B * c->d * c->d;
  hipLaunchKernelGGL(k_basis_transpose, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, src,
                     host_layout_dev, c->d, c->B, 0);
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

}  // namespace rgcn
