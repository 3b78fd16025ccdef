// Block-diagonal relational messages (ConcatGcn.compute_messages,
// code/encoders/message_gcns/gcn_basis_concat.py:35-52) and their gradients.
//
// Reference dataflow: gather W[type] -> [E,nb,sd,sd] (150 MB per direction at FB15k-237),
// 2 x 1.5 M batched 5x5 matvecs, then a [V,E] x [E,d] sparse product; the gradient of the
// gather comes back as IndexedSlices of the same size.  Here:
//
//   * messages are grouped per directed relation ("per-relation CSR", graph_prep.hip) and cut
//     into chunks; one workgroup owns one chunk, so its lanes load the relation's nb x sd x sd
//     weights ONCE into registers (device layout [rel][sd*sd][nb]: lane b reads a coalesced
//     dword per coefficient) and stream the chunk's edges through them;
//   * lane (slot g, block b) handles block b of every G-th message of the chunk: reads the sd
//     source features, does the sd x sd matvec, scales by the neighbour normalisation and writes
//     the message straight into its row of the destination-sorted message buffer -- the
//     [E,nb,sd,sd] gathers are never materialised;
//   * backward: the same pass computes W^T g for the per-source gradient rows AND accumulates the
//     outer products g (x) x in registers; slots are combined through LDS and each chunk writes one
//     [sd*sd][nb] slab; a second tiny kernel sums a relation's slabs in chunk order
//     (deterministic; no float atomics anywhere).
#include <type_traits>

#include "rgcn_internal.h"

namespace rgcn {

namespace {

__device__ __forceinline__ int find_segment(const int32_t* __restrict__ ptr, int n_seg, int x) {
  // largest s in [0, n_seg) with ptr[s] <= x   (ptr has n_seg + 1 entries, ptr[n_seg] > x)
  int lo = 0, hi = n_seg;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (ptr[mid] <= x) lo = mid; else hi = mid;
  }
  return lo;
}

struct MsgArgs {
  const float* Hin;       // [V,d] layer input
  const float* D;         // [V,d] upstream gradient (backward only)
  const float* W;         // [2R][SD*SD][nb]
  float* out;             // Y (forward) or Z (backward): [slots, d]
  float* slab;            // backward: [chunks][SD*SD][nb]
  const int32_t* m_src;
  const int32_t* m_dst;
  const int32_t* m_slot;  // m_dslot (forward) or m_sslot (backward)
  const float* m_norm;
  const int32_t* rel_ptr;
  const int32_t* chunk_ptr;
  int32_t R2, nb, d, chunk, G;
};

template <int SD>
__global__ void k_block_msg_fwd(MsgArgs a) {
  const int bid = blockIdx.x;
  if (bid >= a.chunk_ptr[a.R2]) return;
  const int g = threadIdx.x / a.nb;
  if (g >= a.G) return;
  const int b = threadIdx.x - g * a.nb;
  const int rel = find_segment(a.chunk_ptr, a.R2, bid);
  const int beg = a.rel_ptr[rel] + (bid - a.chunk_ptr[rel]) * a.chunk;
  const int end = min(beg + a.chunk, a.rel_ptr[rel + 1]);
  float w[SD * SD];
#pragma unroll
  for (int k = 0; k < SD * SD; ++k) w[k] = a.W[((size_t)rel * SD * SD + k) * a.nb + b];
  const int col = b * SD;
  for (int j = beg + g; j < end; j += a.G) {
    const int src = a.m_src[j];
    const int slot = a.m_slot[j];
    const float nrm = a.m_norm[j];
    const float* xp = a.Hin + (size_t)src * a.d + col;
    float x[SD];
#pragma unroll
    for (int q = 0; q < SD; ++q) x[q] = xp[q];
    float* yp = a.out + (size_t)slot * a.d + col;
#pragma unroll
    for (int i = 0; i < SD; ++i) {
      float y = 0.0f;
#pragma unroll
      for (int q = 0; q < SD; ++q) y = fmaf(w[i * SD + q], x[q], y);   // out_i = sum_j T[i][j] x_j
      yp[i] = y * nrm;
    }
  }
}

// ZOUT = false: only the weight-gradient slabs (the row gradients come from the single-pass layer kernel)
template <int SD, bool ZOUT>
__global__ void k_block_msg_bwd(MsgArgs a) {
  extern __shared__ float red[];   // [(G-1)][SD*SD][nb]
  const int bid = blockIdx.x;
  if (bid >= a.chunk_ptr[a.R2]) return;   // uniform per workgroup
  const int g = threadIdx.x / a.nb;
  const bool active = g < a.G;
  const int b = threadIdx.x - g * a.nb;
  const int rel = find_segment(a.chunk_ptr, a.R2, bid);
  const int beg = a.rel_ptr[rel] + (bid - a.chunk_ptr[rel]) * a.chunk;
  const int end = min(beg + a.chunk, a.rel_ptr[rel + 1]);
  float w[SD * SD], dw[SD * SD];
#pragma unroll
  for (int k = 0; k < SD * SD; ++k) dw[k] = 0.0f;
  if (active) {
    if constexpr (ZOUT) {
#pragma unroll
      for (int k = 0; k < SD * SD; ++k) w[k] = a.W[((size_t)rel * SD * SD + k) * a.nb + b];
    }
    const int col = b * SD;
    for (int j = beg + g; j < end; j += a.G) {
      const int src = a.m_src[j];
      const int dst = a.m_dst[j];
      const int slot = ZOUT ? a.m_slot[j] : 0;
      const float nrm = a.m_norm[j];
      const float* xp = a.Hin + (size_t)src * a.d + col;
      const float* gp = a.D + (size_t)dst * a.d + col;
      float x[SD], gr[SD];
#pragma unroll
      for (int q = 0; q < SD; ++q) { x[q] = xp[q]; gr[q] = gp[q] * nrm; }
      if constexpr (ZOUT) {
        float* zp = a.out + (size_t)slot * a.d + col;
#pragma unroll
        for (int q = 0; q < SD; ++q) {
          float z = 0.0f;
#pragma unroll
          for (int i = 0; i < SD; ++i) z = fmaf(w[i * SD + q], gr[i], z);   // (T^T g)_j
          zp[q] = z;
        }
      }
#pragma unroll
      for (int i = 0; i < SD; ++i)
#pragma unroll
        for (int q = 0; q < SD; ++q) dw[i * SD + q] = fmaf(gr[i], x[q], dw[i * SD + q]);
    }
  }
  if (a.G > 1) {
    if (active && g > 0) {
#pragma unroll
      for (int k = 0; k < SD * SD; ++k) red[((size_t)(g - 1) * SD * SD + k) * a.nb + b] = dw[k];
    }
    __syncthreads();
    if (active && g == 0) {
      for (int gg = 1; gg < a.G; ++gg)
#pragma unroll
        for (int k = 0; k < SD * SD; ++k) dw[k] += red[((size_t)(gg - 1) * SD * SD + k) * a.nb + b];
    }
  }
  if (active && g == 0) {
#pragma unroll
    for (int k = 0; k < SD * SD; ++k) a.slab[((size_t)bid * SD * SD + k) * a.nb + b] = dw[k];
  }
}

// dW[rel] = sum over the relation's chunk slabs (chunk order => deterministic)
__global__ void k_block_dw_reduce(const float* __restrict__ slab, const int32_t* __restrict__ chunk_ptr,
                                  float* __restrict__ gW, int R2, int per_rel) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)R2 * per_rel) return;
  const int rel = (int)(i / per_rel);
  const int k = (int)(i - (int64_t)rel * per_rel);
  const int c0 = chunk_ptr[rel], c1 = chunk_ptr[rel + 1];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
  int c = c0;
  for (; c + 8 <= c1; c += 8) {      // 8 slab rows in flight (a popular relation has hundreds of chunks)
    const float* p = slab + (size_t)c * per_rel + k;
    a0 += p[0]; a1 += p[(size_t)per_rel]; a2 += p[2 * (size_t)per_rel]; a3 += p[3 * (size_t)per_rel];
    a4 += p[4 * (size_t)per_rel]; a5 += p[5 * (size_t)per_rel]; a6 += p[6 * (size_t)per_rel];
    a7 += p[7 * (size_t)per_rel];
  }
  for (; c < c1; ++c) a0 += slab[(size_t)c * per_rel + k];
  gW[i] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}

// host [R][nb][sd2]  <->  device [R][sd2][nb]
__global__ void k_block_transpose(const float* __restrict__ in, float* __restrict__ out, int R, int nb,
                                  int sd2, int to_device) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = (int64_t)R * nb * sd2;
  if (i >= n) return;
  // i indexes the device layout
  const int b = (int)(i % nb);
  const int k = (int)((i / nb) % sd2);
  const int r = (int)(i / ((int64_t)nb * sd2));
  const int64_t h = ((int64_t)r * nb + b) * sd2 + k;
  if (to_device) out[i] = in[h]; else out[h] = in[i];
}

int max_chunks(const rgcn_ctx* c) {
  const int64_t M = 2 * c->g.E;
  return (int)((M + c->g.chunk - 1) / c->g.chunk) + 2 * c->R;
}

template <typename F>
rgcn_status dispatch_sd(rgcn_ctx* c, F&& f) {
  switch (c->sd) {
    case 1: f(std::integral_constant<int, 1>()); break;
    case 2: f(std::integral_constant<int, 2>()); break;
    case 3: f(std::integral_constant<int, 3>()); break;
    case 4: f(std::integral_constant<int, 4>()); break;
    case 5: f(std::integral_constant<int, 5>()); break;
    case 8: f(std::integral_constant<int, 8>()); break;
    default:
      RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "block size d/nb must be one of 1,2,3,4,5,8");
  }
  return RGCN_OK;
}

}  // namespace

// Launch geometry: G message slots x nb lanes per workgroup, padded to whole wavefronts.
rgcn_status block_geometry(rgcn_ctx* c) {
  if (c->nb > 512) RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "NumberOfBasisFunctions (block count) > 512");
  if (!(c->sd == 1 || c->sd == 2 || c->sd == 3 || c->sd == 4 || c->sd == 5 || c->sd == 8))
    RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "block size d/nb must be one of 1,2,3,4,5,8");
  int best_block = 0, best_g = 0;
  double best_util = 0.0;
  for (int blk = 64; blk <= 512; blk += 64) {
    int g = blk / c->nb;
    if (g < 1) continue;
    if (g > 8) g = 8;
    // LDS for the slot reduction of the backward kernel must stay below 64 KiB
    while (g > 1 && (size_t)(g - 1) * c->sd * c->sd * c->nb * sizeof(float) > 60 * 1024) --g;
    double util = (double)(g * c->nb) / blk;
    if (util > best_util + 1e-9) { best_util = util; best_block = blk; best_g = g; }
  }
  c->msg_block = best_block;
  c->msg_slots = best_g;
  return RGCN_OK;
}

rgcn_status block_msg_forward(rgcn_ctx* c, int layer, const float* Hin, float* Ybuf) {
  if (c->g.E == 0) return RGCN_OK;
  MsgArgs a;
  a.Hin = Hin; a.D = nullptr; a.W = c->layers[layer].wrel; a.out = Ybuf; a.slab = nullptr;
  a.m_src = c->g.m_src; a.m_dst = c->g.m_dst; a.m_slot = c->g.m_dslot; a.m_norm = c->g.m_norm;
  a.rel_ptr = c->g.rel_ptr; a.chunk_ptr = c->g.chunk_ptr;
  a.R2 = 2 * c->R; a.nb = c->nb; a.d = c->d; a.chunk = c->g.chunk; a.G = c->msg_slots;
  const double M = 2.0 * c->g.E / c->world;
  // compulsory: a gathered row of H counts once however many messages read it (at most min(M, V) distinct rows)
  const double rows = M < c->V ? M : (double)c->V;
  const double Wb = 8.0 * c->R * c->nb * c->sd * c->sd;
  ProfScope ps(c, "block_msg_fwd", M * (8.0 * c->d + 16.0) + Wb, M * 2.0 * c->d * c->sd,
               4.0 * c->d * (rows + M) + 16.0 * M + Wb);
  RGCN_TRY(dispatch_sd(c, [&](auto sdc) {
    constexpr int SD = decltype(sdc)::value;
    hipLaunchKernelGGL((k_block_msg_fwd<SD>), dim3(max_chunks(c)), dim3(c->msg_block), 0, c->stream, a);
  }));
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

rgcn_status block_msg_backward(rgcn_ctx* c, int layer, const float* Hin, const float* D, float* Zbuf) {
  const int per_rel = c->sd * c->sd * c->nb;
  const int R2 = 2 * c->R;
  if (c->g.E > 0) {
    const int nchunks = max_chunks(c);
    if ((size_t)nchunks * per_rel > c->slab_dw_floats)
      RGCN_FAIL(c, RGCN_ERR_STATE, "internal: dW slab buffer too small");
    MsgArgs a;
    a.Hin = Hin; a.D = D; a.W = c->layers[layer].wrel; a.out = Zbuf; a.slab = c->slab_dw;
    a.m_src = c->g.m_src; a.m_dst = c->g.m_dst; a.m_slot = c->g.m_sslot; a.m_norm = c->g.m_norm;
    a.rel_ptr = c->g.rel_ptr; a.chunk_ptr = c->g.chunk_ptr;
    a.R2 = R2; a.nb = c->nb; a.d = c->d; a.chunk = c->g.chunk; a.G = c->msg_slots;
    const size_t lds = (size_t)(c->msg_slots - 1) * per_rel * sizeof(float);
    const double M = 2.0 * c->g.E / c->world;
    const double rows = M < c->V ? M : (double)c->V;     // distinct rows of H and of D a launch can touch
    if (Zbuf != nullptr) {
      ProfScope ps(c, "block_msg_bwd", M * (12.0 * c->d + 20.0) + 16.0 * c->R * per_rel,
                   M * 4.0 * c->d * c->sd, 4.0 * c->d * (2.0 * rows + M) + 20.0 * M + 16.0 * c->R * per_rel);
      RGCN_TRY(dispatch_sd(c, [&](auto sdc) {
        constexpr int SD = decltype(sdc)::value;
        hipLaunchKernelGGL((k_block_msg_bwd<SD, true>), dim3(nchunks), dim3(c->msg_block), lds, c->stream, a);
      }));
    } else {
      // weight gradients only: two row gathers per message, one slab per chunk
      ProfScope ps(c, "block_dw_msgs", M * (8.0 * c->d + 16.0) + 8.0 * c->R * per_rel, M * 2.0 * c->d * c->sd,
                   4.0 * c->d * 2.0 * rows + 16.0 * M + 8.0 * c->R * per_rel);
      RGCN_TRY(dispatch_sd(c, [&](auto sdc) {
        constexpr int SD = decltype(sdc)::value;
        hipLaunchKernelGGL((k_block_msg_bwd<SD, false>), dim3(nchunks), dim3(c->msg_block), lds, c->stream, a);
      }));
    }
    RGCN_HIP(c, hipGetLastError());
  }
  return RGCN_OK;
}

// dW[rel] = ordered sum of the relation's chunk slabs written by block_msg_backward
rgcn_status block_dw_reduce(rgcn_ctx* c, int layer) {
  const int per_rel = c->sd * c->sd * c->nb;
  const int R2 = 2 * c->R;
  {
    // also correct for E == 0: chunk_ptr is all zeros => every relation gets a zero gradient
    const int64_t n = (int64_t)R2 * per_rel;
    ProfScope ps(c, "block_dw_reduce", 8.0 * n, 0);
    hipLaunchKernelGGL(k_block_dw_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                       c->slab_dw, c->g.chunk_ptr, c->layers[layer].grel, R2, per_rel);
    RGCN_HIP(c, hipGetLastError());
  }
  return RGCN_OK;
}

rgcn_status block_to_device_layout(rgcn_ctx* c, const float* host_layout_dev, float* dst, int R) {
  const int64_t n = (int64_t)R * c->nb * c->sd * c->sd;
  hipLaunchKernelGGL(k_block_transpose, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                     host_layout_dev, dst, R, c->nb, c->sd * c->sd, 1);
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

rgcn_status block_from_device_layout(rgcn_ctx* c, const float* src, float* host_layout_dev, int R) {
  const int64_t n = (int64_t)R * c->nb * c->sd * c->sd;
  hipLaunchKernelGGL(k_block_transpose, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                     src, host_layout_dev, R, c->nb, c->sd * c->sd, 0);
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

}  // namespace rgcn
