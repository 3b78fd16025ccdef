// Single-pass block-diagonal relational layer (ConcatGcn.compute_messages + combine_messages,
// code/encoders/message_gcns/gcn_basis_concat.py:35-52,69-83, inside MessageGcn.compute_vertex_embeddings,
// message_gcn.py:49-79) and its gradient w.r.t. the layer input: one kernel per layer and direction computes
//     forward   H'[v] = relu?( dropout(S[v]) + sum over the messages into v of  n_m * W[r_m] . H[src_m] )
//     backward  D'[u] = ( G[u] + sum over the messages out of u of  W[r_m]^T . (n_m * D[dst_m]) ) * relu'(H[u]),
//               dS'[u] = D'[u] * dropout'
// straight from the incidence CSR -- no [2E, d] message buffer travels to HBM and back (the two-kernel form,
// block_msgs.hip + k_combine, stages 2 x 60 MB per layer pass there at FB15k-237 size).
//
// Why this shape (the destination-major SpMM with per-message weight re-reads measured 51 us):
// the block-diagonal product is INDEPENDENT per block b -- out[v, b*sd : (b+1)*sd] needs only column band b of the
// gathered rows and the nb-th part of every relation's weights, W[:, b] = [2R][sd*sd] = 47 KB at FB15k-237 size.
// So a workgroup owns (block b, a chunk of rows): it stages W[:, b] in LDS ONCE (a block-major copy of the weights,
// k_wbm_build), then
//   phase 1  lane <-> incidence slot (CSR order, perfectly balanced whatever the degrees): gather the sd floats of
//            the partner row (one 4-byte-aligned dwordx4 + dword for sd = 5), read the relation's sd x sd block from
//            LDS, y = n * W x  ->  LDS tile;
//   phase 2  lane <-> row: add the row's slots of the tile in slot order (rows with more than kLongRow slots: eight
//            interleaved lanes, then lane order; giant rows: piece by piece) -- the wavefront segmented reduction,
//            in EXACTLY the order k_combine uses, so this kernel and the two-kernel form agree bitwise
//            (tests/test_gpu_parity.py::test_fused_layer_kernel_equals_the_two_kernel_form);
//   pro / epilogue (self-loop term with its dropout, relu', relu, the dropout-scaled copy for the next GEMMs) in the
//            row lanes: each reads / writes its row's sd floats of the band (20 bytes, dwordx4 + dword).
// XCD-aware grid: workgroup id = 8 j + x lands on XCD x (round-robin dispatch); XCD x owns the CONTIGUOUS band of
// blocks [x nb / 8, (x+1) nb / 8) -- 250 bytes of every row at nb = 100, sd = 5 -- so the lines its gathers touch
// (V x 2-3 lines = 4-5 MB) stay in that XCD's 4 MB L2 / the Infinity Cache instead of being fetched by all eight, and
// within an XCD consecutive workgroups take the band's blocks of ONE row chunk, so the 20-byte pieces of an output
// line are written close together in time.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

#include "rgcn_internal.h"

namespace rgcn {

namespace {

template <int N>
struct __attribute__((packed, aligned(4))) FloatN {
  float v[N];
};

constexpr int sdp_of(int sd) { return ((sd * sd + 3) / 4) * 4; }      // floats per relation row of the LDS weight table

struct SpmmArgs {
  const float* X;            // gathered operand [V,d]: H_in (forward), D (backward)
  const float* Wbm;          // [nb][R2][SDP] block-major weights
  const int32_t* row_ptr;    // [V+1] incidence CSR
  const int32_t* slot_v;     // per slot: the partner vertex (d_src forward, s_dst backward)
  const int32_t* slot_rel;   // per slot: directed relation in [0, 2R)
  const float* slot_norm;    // per slot: neighbour normalisation of that message
  const float* base;         // [V,d] self-loop term (S forward, G backward); dropout `drop` applies
  const float* gate;         // optional: result *= (gate > 0)
  float* out;
  float* out2;               // optional: out * dropout(drop2)
  DropSpec drop, drop2;
  int32_t V, d, R2, nb, relu, rch, nchunk, giant, order;
};

// W_bm[b][rel][k] = W[rel][k][b]  (k < sd*sd; the padding floats stay zero)
__global__ void k_wbm_build(const float* __restrict__ W, float* __restrict__ Wbm, int R2, int nb, int sd2, int sdp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // index into W: coalesced reads
  if (i >= (int64_t)R2 * sd2 * nb) return;
  const int b = (int)(i % nb);
  const int k = (int)((i / nb) % sd2);
  const int rel = (int)(i / ((int64_t)nb * sd2));
  Wbm[((size_t)b * R2 + rel) * sdp + k] = W[i];
}

// Registers of one row window (lane <-> row v = first row of the window + lane) and of one slot tile
// (lane <-> slots tile start + lane + u * THREADS).
template <int SD>
struct RowRegs {
  int beg, end, wend;          // the row's slots [beg, end); wend = first slot behind the window's last row (uniform)
  FloatN<SD> base, gate;
};
template <int UNR>
struct IdxRegs {
  int sv[UNR], rl[UNR];
  float nr[UNR];
};

// Both axes are cut on FIXED grids -- slot tiles of TILE = THREADS * UNR slots, row windows of THREADS rows -- and the
// kernel walks the merge of the two sequences: with (tile t, window w) in hand every row lane adds its row's slots that
// lie in the tile; if the window's last row ends inside the tile the window is finished (epilogue, next window, same
// tile), otherwise the tile is exhausted (barrier, phase 1 of the next tile).  Fixed grids mean every address is known
// ahead of time: the slot indices are fetched two tiles ahead, the gathered rows one tile ahead, the next window's
// row pointers and self-loop / gate rows one window ahead -- no load sits on the critical path of a step, which with
// two workgroups per CU (the weight table takes a third of the LDS) is what keeps the CU busy.
template <int SD, bool BWD, int THREADS, int UNR>
__global__ void __launch_bounds__(THREADS) k_block_spmm(SpmmArgs a) {
#pragma clang fp contract(off)
  constexpr int SDP = sdp_of(SD), TILE = THREADS * UNR;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Wl = lds;                          // [R2][SDP]
  float* ybuf = lds + (size_t)a.R2 * SDP;   // [TILE][SD] messages of the slot tile in hand
  const int tid = threadIdx.x;
  // (XCD x, j-th workgroup of that XCD) -> (row chunk, block of the XCD's band)
  const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int b0 = (x * a.nb) >> 3, b1 = ((x + 1) * a.nb) >> 3;
  const int nbx = b1 - b0;
  if (nbx <= 0 || j >= nbx * a.nchunk) return;
  // chunk-major inside the XCD (consecutive workgroups = the band's blocks of one row chunk: the 20-byte pieces of an
  // output line are written close together in time); block-major (RGCN_SPMM_ORDER=1: the ~64 workgroups an XCD runs
  // at a time sit on few neighbouring blocks, a smaller gather working set) measured 5-10 % slower
  int chunk, b;
  if (a.order == 0) { chunk = j / nbx; b = b0 + (j - chunk * nbx); }
  else { const int bi = j / a.nchunk; chunk = j - bi * a.nchunk; b = b0 + bi; }
  const int col = b * SD;
  const int r_begin = chunk * a.rch, r_end = min(a.V, r_begin + a.rch);
  const int T0 = a.row_ptr[r_begin], SE = a.row_ptr[r_end];      // the chunk's slots
  const int nwin = (r_end - r_begin + THREADS - 1) / THREADS;

  auto load_idx = [&](int t, IdxRegs<UNR>& r) {
    const int s0 = T0 + t * TILE;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int s = s0 + tid + u * THREADS;
      const bool ok = s < SE;
      r.sv[u] = ok ? a.slot_v[s] : 0;
      r.rl[u] = ok ? a.slot_rel[s] : 0;
      r.nr[u] = ok ? a.slot_norm[s] : 0.0f;
    }
  };
  auto load_x = [&](int t, const IdxRegs<UNR>& r, FloatN<SD> (&xg)[UNR]) {
    if (T0 + t * TILE >= SE) return;          // uniform: nothing behind the chunk's last slot
#pragma unroll
    for (int u = 0; u < UNR; ++u) xg[u] = *reinterpret_cast<const FloatN<SD>*>(a.X + (size_t)r.sv[u] * a.d + col);
  };
  auto load_row = [&](int w, RowRegs<SD>& r) {
    const int wb = r_begin + w * THREADS, we = min(r_end, wb + THREADS);
    const int v = wb + tid;
    r.wend = a.row_ptr[we];
    r.beg = r.end = SE;
    if (v < we) {
      r.beg = a.row_ptr[v];
      r.end = a.row_ptr[v + 1];
      const size_t off = (size_t)v * a.d + col;
      if (a.base != nullptr) r.base = *reinterpret_cast<const FloatN<SD>*>(a.base + off);
      if (a.gate != nullptr) r.gate = *reinterpret_cast<const FloatN<SD>*>(a.gate + off);
    }
  };
  // phase 1, lane <-> slot: y = n W x (forward) / W^T (n g) (backward) of the tile's slots -> ybuf
  auto phase1 = [&](int t, const IdxRegs<UNR>& r, const FloatN<SD> (&xg)[UNR]) {
    const int s0 = T0 + t * TILE;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (s0 + tid + u * THREADS >= SE) continue;
      float w[SDP];
      const float4* wp = reinterpret_cast<const float4*>(Wl + (size_t)r.rl[u] * SDP);
#pragma unroll
      for (int q = 0; q < SDP / 4; ++q) {
        const float4 tq = wp[q];
        w[4 * q] = tq.x; w[4 * q + 1] = tq.y; w[4 * q + 2] = tq.z; w[4 * q + 3] = tq.w;
      }
      float* yp = ybuf + (size_t)(tid + u * THREADS) * SD;
      if constexpr (!BWD) {
        // out_i = n * sum_q T[i][q] x_q   (the arithmetic of k_block_msg_fwd)
#pragma unroll
        for (int i = 0; i < SD; ++i) {
          float y = 0.0f;
#pragma unroll
          for (int q = 0; q < SD; ++q) y = fmaf(w[i * SD + q], xg[u].v[q], y);
          yp[i] = y * r.nr[u];
        }
      } else {
        // (T^T (n g))_q   (the arithmetic of k_block_msg_bwd)
        float gr[SD];
#pragma unroll
        for (int q = 0; q < SD; ++q) gr[q] = xg[u].v[q] * r.nr[u];
#pragma unroll
        for (int q = 0; q < SD; ++q) {
          float z = 0.0f;
#pragma unroll
          for (int i = 0; i < SD; ++i) z = fmaf(w[i * SD + q], gr[i], z);
          yp[q] = z;
        }
      }
    }
  };

  IdxRegs<UNR> idx_cur, idx_nxt;
  FloatN<SD> xg[UNR];
  RowRegs<SD> row, row_nxt;
  load_idx(0, idx_cur);
  load_idx(1, idx_nxt);
  load_row(0, row);
  {
    // weight table -> LDS, eight 16-byte loads in flight per lane (one load-then-store per trip waits a full memory
    // round trip thirteen times over: measured, that alone was 40 us of the kernel)
    const float4* src = reinterpret_cast<const float4*>(a.Wbm + (size_t)b * a.R2 * SDP);
    float4* dst = reinterpret_cast<float4*>(Wl);
    const int n4 = a.R2 * (SDP / 4);
    for (int i0 = tid; i0 < n4; i0 += 8 * THREADS) {
      float4 w8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * THREADS;
        w8[u] = src[i < n4 ? i : i0];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * THREADS;
        if (i < n4) dst[i] = w8[u];
      }
    }
  }
  load_x(0, idx_cur, xg);
  if (nwin > 1) load_row(1, row_nxt);
  __syncthreads();                     // weight table in place
  phase1(0, idx_cur, xg);
  idx_cur = idx_nxt;
  load_x(1, idx_cur, xg);
  load_idx(2, idx_nxt);
  __syncthreads();

  // per-row state of the window in hand
  float tot[SD], lp[8][SD], gsum[SD];
  int cls = 0;      // 0: short row, slots added one after the other; 1: long row, eight interleaved lanes; 2: giant row, pieces
  auto open_window = [&](int w) {
    const int v = r_begin + w * THREADS + tid;
    const size_t off = (size_t)v * a.d + col;
#pragma unroll
    for (int i = 0; i < SD; ++i) {
      float t0 = 0.0f;
      if (a.base != nullptr && v < r_end) {
        t0 = row.base.v[i];
        t0 *= drop_factor(a.drop, off + i);
      }
      tot[i] = t0;
      gsum[i] = 0.0f;
#pragma unroll
      for (int q = 0; q < 8; ++q) lp[q][i] = 0.0f;
    }
    cls = (a.giant && row.end - row.beg > kGiantRow) ? 2 : (row.end - row.beg > kLongRow ? 1 : 0);
  };
  int t = 0, w = 0;
  open_window(0);
  while (true) {
    const int tlo = T0 + t * TILE, thi = tlo + TILE;
    // ---- phase 2, lane <-> row: this row's slots inside the tile, in k_combine's order.  The loops are cut into groups
    // of 4 (short rows) / 8 (long rows) slots whose LDS reads are all issued before the first add: the adds keep their
    // order, the reads no longer wait for each other (one read-then-add per trip ran at LDS latency, ~130 cycles a
    // slot, and the long rows of a minibatch held their workgroups for tens of microseconds).
    {
      int lo = max(row.beg, tlo);
      const int hi = min(row.end, thi);
      if (lo < hi) {
        if (cls == 0) {
          for (int s = lo; s < hi; s += 4) {
            float y[4][SD];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const float* yp = ybuf + (size_t)(min(s + u, hi - 1) - tlo) * SD;
#pragma unroll
              for (int i = 0; i < SD; ++i) y[u][i] = yp[i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (s + u < hi) {
#pragma unroll
                for (int i = 0; i < SD; ++i) tot[i] = tot[i] + y[u][i];
              }
          }
        } else {
          while (lo < hi) {
            int pend = hi;
            if (cls == 2) pend = min(hi, row.beg + ((lo - row.beg) / kGiantRow + 1) * kGiantRow);
            // groups of eight consecutive slots aligned with the row's start: slot beg + 8k + q belongs to lane q
            for (int s8 = lo - ((lo - row.beg) & 7); s8 < pend; s8 += 8) {
              float y[8][SD];
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                const int sc = min(max(s8 + q, lo), pend - 1);
                const float* yp = ybuf + (size_t)(sc - tlo) * SD;
#pragma unroll
                for (int i = 0; i < SD; ++i) y[q][i] = yp[i];
              }
#pragma unroll
              for (int q = 0; q < 8; ++q)
                if (s8 + q >= lo && s8 + q < pend) {
#pragma unroll
                  for (int i = 0; i < SD; ++i) lp[q][i] = lp[q][i] + y[q][i];
                }
            }
            lo = pend;
            if (cls == 2 && ((lo - row.beg) % kGiantRow == 0 || lo == row.end)) {      // a piece is complete
#pragma unroll
              for (int i = 0; i < SD; ++i) {
                float ts = lp[0][i];
#pragma unroll
                for (int q = 1; q < 8; ++q) ts = ts + lp[q][i];
                gsum[i] = gsum[i] + ts;
              }
#pragma unroll
              for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int i = 0; i < SD; ++i) lp[q][i] = 0.0f;
            }
          }
        }
      }
    }
    if (row.wend <= thi || thi >= SE) {
      // ---- every row of the window ends inside this tile: epilogue (relu', relu, store, dropout-scaled copy)
      const int v = r_begin + w * THREADS + tid;
      if (v < r_end) {
        if (cls == 1) {
#pragma unroll
          for (int i = 0; i < SD; ++i) {
            float ts = lp[0][i];
#pragma unroll
            for (int q = 1; q < 8; ++q) ts = ts + lp[q][i];
            tot[i] = tot[i] + ts;
          }
        } else if (cls == 2) {
#pragma unroll
          for (int i = 0; i < SD; ++i) tot[i] = tot[i] + gsum[i];
        }
        const size_t off = (size_t)v * a.d + col;
        FloatN<SD> o, o2;
#pragma unroll
        for (int i = 0; i < SD; ++i) {
          float val = tot[i];
          if (a.gate != nullptr) val = row.gate.v[i] > 0.0f ? val : 0.0f;
          if (a.relu) val = fmaxf(val, 0.0f);
          o.v[i] = val;
          if (a.out2 != nullptr) o2.v[i] = val * drop_factor(a.drop2, off + i);
        }
        *reinterpret_cast<FloatN<SD>*>(a.out + off) = o;
        if (a.out2 != nullptr) *reinterpret_cast<FloatN<SD>*>(a.out2 + off) = o2;
      }
      if (++w == nwin) break;
      row = row_nxt;
      if (w + 1 < nwin) load_row(w + 1, row_nxt);
      open_window(w);
    } else {
      // ---- the tile is exhausted: next tile (its indices and gathered rows are already in registers)
      __syncthreads();                   // every row lane is done reading ybuf
      ++t;
      phase1(t, idx_cur, xg);
      idx_cur = idx_nxt;
      load_x(t + 1, idx_cur, xg);
      load_idx(t + 2, idx_nxt);
      __syncthreads();
    }
  }
}

constexpr int kSpmmThreads = 256, kSpmmUnroll = 4;
constexpr size_t kSpmmLdsBudget = 80 * 1024;      // two workgroups per CU

size_t spmm_lds_bytes(const rgcn_ctx* c) {
  return ((size_t)2 * c->R * sdp_of(c->sd) + (size_t)kSpmmThreads * kSpmmUnroll * c->sd) * sizeof(float);
}

template <int SD, bool BWD>
hipError_t launch_spmm(rgcn_ctx* c, const SpmmArgs& a, int grid, size_t lds) {
  auto kern = k_block_spmm<SD, BWD, kSpmmThreads, kSpmmUnroll>;
  static std::mutex mu;
  static std::set<int> configured;            // devices this instantiation's dynamic-LDS limit is set on
  {
    std::lock_guard<std::mutex> lk(mu);
    if (!configured.count(c->cfg.device)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)kSpmmLdsBudget);
      if (e != hipSuccess) return e;
      configured.insert(c->cfg.device);
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kSpmmThreads), lds, c->stream, a);
  return hipGetLastError();
}

}  // namespace

// The single-pass form needs the block's whole weight table in LDS beside the slot tile: 2R x sd^2 floats (47 KB at
// FB15k-237's 237 relations; FB15k's 1,345 do not fit and keep the two-kernel form), one GPU.
bool block_spmm_available(const rgcn_ctx* c) {
  return c->kind == RGCN_KIND_BLOCK && c->world == 1 && spmm_lds_bytes(c) <= kSpmmLdsBudget;
}

size_t block_spmm_weight_floats(const rgcn_ctx* c) { return (size_t)c->nb * 2 * c->R * sdp_of(c->sd); }

// block-major copy of a layer's relation weights, rebuilt when the weights changed (set_param, Adam) -- and inside
// every captured step, whose replays follow weights the host does not see
rgcn_status block_spmm_refresh_weights(rgcn_ctx* c, int layer) {
  LayerBufs& lb = c->layers[layer];
  if (!lb.wbm) RGCN_FAIL(c, RGCN_ERR_STATE, "internal: no block-major weight buffer");
  if (!c->capturing && lb.wbm_version == c->weights_version) return RGCN_OK;
  const int64_t n = (int64_t)2 * c->R * c->sd * c->sd * c->nb;
  ProfScope ps(c, "block_wbm_build", 8.0 * n, 0);
  hipLaunchKernelGGL(k_wbm_build, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, lb.wrel, lb.wbm, 2 * c->R,
                     c->nb, c->sd * c->sd, sdp_of(c->sd));
  RGCN_HIP(c, hipGetLastError());
  lb.wbm_version = c->capturing ? ~0ull : c->weights_version;
  return RGCN_OK;
}

rgcn_status block_spmm(rgcn_ctx* c, const char* tag, int layer, bool backward, const float* X, const CombineArgs& ca) {
  RGCN_TRY(block_spmm_refresh_weights(c, layer));
  SpmmArgs a;
  a.X = X;
  a.Wbm = c->layers[layer].wbm;
  a.row_ptr = c->g.row_ptr;
  a.slot_v = backward ? c->g.s_dst : c->g.d_src;
  a.slot_rel = backward ? c->g.s_rel : c->g.d_rel;
  a.slot_norm = backward ? c->g.s_norm : c->g.d_norm;
  a.base = ca.base; a.gate = ca.gate; a.out = ca.out; a.out2 = ca.out2; a.drop = ca.drop; a.drop2 = ca.drop2;
  a.V = c->V; a.d = c->d; a.R2 = 2 * c->R; a.nb = c->nb; a.relu = ca.relu;
  a.giant = c->g.giant_on ? 1 : 0;
  // row chunks: about two workgroups per CU slot and XCD (32 CUs x 2 resident workgroups), never finer than one
  // sub-chunk of kSpmmThreads rows
  const int maxband = (c->nb + 7) / 8;
  static const int chunks_env = getenv("RGCN_SPMM_CHUNKS") ? atoi(getenv("RGCN_SPMM_CHUNKS")) : 0;
  static const int order_env = getenv("RGCN_SPMM_ORDER") ? atoi(getenv("RGCN_SPMM_ORDER")) : 0;
  a.order = order_env;
  int nchunk = chunks_env > 0 ? chunks_env : 9;
  const int max_by_rows = (c->V + kSpmmThreads - 1) / kSpmmThreads;
  nchunk = std::max(1, std::min(nchunk, max_by_rows));
  int rch = (c->V + nchunk - 1) / nchunk;
  rch = ((rch + kSpmmThreads - 1) / kSpmmThreads) * kSpmmThreads;
  nchunk = (c->V + rch - 1) / rch;
  a.rch = rch; a.nchunk = nchunk;
  const int grid = 8 * maxband * nchunk;
  const double M = 2.0 * c->g.E, P = 4.0 * c->V * c->d;
  const double Wb = 8.0 * c->R * c->nb * c->sd * c->sd;
  const double streams = (ca.base ? 1.0 : 0.0) + (ca.gate ? 1.0 : 0.0) + 1.0 + (ca.out2 ? 1.0 : 0.0);
  const double rows = M < c->V ? M : (double)c->V;
  // design: every slot gathers sd floats per block (M rows of d floats in all), the slot arrays once per block,
  // the weight table once per workgroup; compulsory: the distinct gathered rows once, the slot arrays and weights once
  ProfScope ps(c, tag, 4.0 * c->d * M + streams * P + 12.0 * M * c->nb + Wb * nchunk + 4.0 * c->V,
               M * 2.0 * c->d * c->sd, 4.0 * c->d * rows + streams * P + 12.0 * M + Wb + 4.0 * c->V);
  hipError_t e = hipSuccess;
  const size_t lds = spmm_lds_bytes(c);
#define RGCN_SPMM_CASE(SDV)                                                                   \
  case SDV:                                                                                   \
    e = backward ? launch_spmm<SDV, true>(c, a, grid, lds) : launch_spmm<SDV, false>(c, a, grid, lds); \
    break;
  switch (c->sd) {
    RGCN_SPMM_CASE(1) RGCN_SPMM_CASE(2) RGCN_SPMM_CASE(3) RGCN_SPMM_CASE(4) RGCN_SPMM_CASE(5) RGCN_SPMM_CASE(8)
    default: RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "block size d/nb must be one of 1,2,3,4,5,8");
  }
#undef RGCN_SPMM_CASE
  RGCN_HIP(c, e);
  return RGCN_OK;
}

}  // namespace rgcn
