// DistMult decoder on the device: energies, loss, regulariser and their gradients
// (reference: code/decoders/bilinear_diag.py; "next" row f1 of SURVEY.md section 8f).
//
//   x_n   = sum_k e1_k r_k e2_k            e1 = codes[X_n0], r = W_relation[X_n1], e2 = codes[X_n2]   (:18-21,30)
//   loss  = mean_n[(1-y_n) x_n + log1p(exp(-|x_n|)) + max(-x_n,0)]                                     (:32-34, pos_weight = 1)
//         + lambda (mean(e1^2) + mean(r^2) + mean(e2^2))                                               (:63-69)
//
// The reference materialises three [N,d] gathers (660 MB each at N = 330,000) and gets their gradients
// back as IndexedSlices.  Here nothing [N,d] is ever written:
//   K1  one wavefront per triple: the three rows are read once, x_n, dx_n = (sigmoid(x_n) - y_n)/N and the
//       loss terms fall out of one shuffle reduction;
//   K2  entity gradient = segmented reduction over a by-entity CSR of the 2N (triple, side) incidences:
//       dcodes[v] = sum_{incidences at v} dx_n (r (.) other) + (2 lambda / (N d)) cnt_v codes[v]
//       -- the same formula for the subject and the object side, rows are independent, no atomics;
//   K3  relation gradient = per-relation chunks of 128 triples -> slab, ordered reduce per relation.
// The two CSRs depend only on X, not on the encoder: they are built on a side stream while the encoder's
// forward pass runs (rgcn_train_step_device).
#include <algorithm>

#include "rgcn_internal.h"

namespace rgcn {

namespace {

constexpr int kDecLongRow = 256;    // entity rows with more incidences are cut into pieces, one workgroup each
constexpr int kDecPiece = 512;      // incidences per piece: a hub entity (tens of thousands of incidences in a
                                    // neighbourhood-sampled batch with its tiled negatives) spreads over many CUs
constexpr int kDecChunk = 128;      // triples per relation chunk
constexpr int kRowThreads = 1024;

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

__device__ __forceinline__ int lower_bound_u32(const uint32_t* a, int n, uint32_t x) {
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// ---- CSR construction -----------------------------------------------------------------------
__global__ void k_dec_keys(const int32_t* __restrict__ X, int N, int V, int R, int n_rel, uint32_t* keyv,
                           uint32_t* keyr, int32_t* errflag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * N) return;
  const bool subj = i < N;
  const int n = subj ? i : i - N;
  const int s = X[3 * n], r = X[3 * n + 1], o = X[3 * n + 2];
  const bool ok = (unsigned)s < (unsigned)V && (unsigned)o < (unsigned)V && (unsigned)r < (unsigned)R;
  if (subj && !ok) atomicOr(errflag, 2);
  keyv[i] = ok ? (uint32_t)(subj ? s : o) : (uint32_t)V;
  // n_rel < N: a tiled batch -- only the first copy of every triple is sorted by relation, k_dec_expand puts the others
  // behind it
  if (subj && n < n_rel) keyr[n] = ok ? (uint32_t)r : (uint32_t)R;
}

// A batch the negative sampler tiled (rows p, p + period, p + 2 period, ... are copies of row p with one entity
// replaced): the relation order of all N triples follows from the order of the first `period` -- the copies of a triple
// directly behind it, which is also what lets a relation chunk fetch their shared rows once.  A copy that is not what
// the sampler wrote (another relation, an id out of range: someone rewrote the buffer) raises flag 16 and is replaced by
// its first copy, so that the kernels behind this one stay inside their arrays.
__global__ void k_dec_expand(const int32_t* __restrict__ X, int N, int V, int R, int period, int L,
                             const int32_t* __restrict__ perm_pos, int32_t* __restrict__ permr, int32_t* errflag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int g = i / L, j = i - g * L;
  const int p = perm_pos[g];
  int t = p + j * period;
  if (j > 0) {
    const int s = X[3 * t], r = X[3 * t + 1], o = X[3 * t + 2];
    if (r != X[3 * p + 1] || (unsigned)s >= (unsigned)V || (unsigned)o >= (unsigned)V) {
      atomicOr(errflag, 16);
      t = p;
    }
  }
  permr[i] = t;
}

// block 0: relation offsets + exclusive scan of the chunk counts; blocks 1..: entity row offsets + long rows
__global__ void __launch_bounds__(1024) k_dec_ptrs(const uint32_t* __restrict__ keyv_s,
                                                   const uint32_t* __restrict__ keyr_s, int N, int V, int R,
                                                   int n_rel, int rel_scale, int32_t* row_ptr, int32_t* long_rows, int32_t* nlong, int cap,
                                                   int32_t* long_first, int32_t* long_cnt, int32_t* piece_row,
                                                   int32_t* piece_k, int piece_cap, int32_t* rel_ptr,
                                                   int32_t* chunk_ptr, uint32_t* row_key) {
  if (blockIdx.x == 0) {
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base <= R; base += 1024) {
      const int r = base + tid;
      int lo = 0, nchunks = 0;
      if (r <= R) {
        lo = lower_bound_u32(keyr_s, n_rel, (uint32_t)r) * rel_scale;
        rel_ptr[r] = lo;
        if (r < R) nchunks = (lower_bound_u32(keyr_s, n_rel, (uint32_t)(r + 1)) * rel_scale - lo + kDecChunk - 1) / kDecChunk;
      }
      int incl = nchunks;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
      }
      if (lane == 63) wsum[wid] = incl;
      __syncthreads();
      int wbase = 0, total = 0;
      for (int w = 0; w < 16; ++w) {
        if (w < wid) wbase += wsum[w];
        total += wsum[w];
      }
      const int carry = carry_s;
      if (r <= R) chunk_ptr[r] = carry + wbase + incl - nchunks;
      __syncthreads();
      if (tid == 0) carry_s = carry + total;
      __syncthreads();
    }
    return;
  }
  const int v = (blockIdx.x - 1) * blockDim.x + threadIdx.x;
  if (v > V) return;
  const int beg = lower_bound_u32(keyv_s, 2 * N, (uint32_t)v);
  row_ptr[v] = beg;
  if (v == V) return;
  const int end = lower_bound_u32(keyv_s, 2 * N, (uint32_t)(v + 1));
  // sort key of the line form's row order: descending length (long rows, which it skips, at the end)
  row_key[v] = end - beg > kDecLongRow ? 255u : (uint32_t)(255 - min(end - beg, 255));
  if (end - beg > kDecLongRow) {
    // pieces of one row get consecutive ids, so the finishing pass adds them in a fixed order whatever
    // order the rows were registered in
    const int np = (end - beg + kDecPiece - 1) / kDecPiece;
    const int i = atomicAdd(nlong, 1);
    const int base = atomicAdd(nlong + 1, np);
    if (i < cap) { long_rows[i] = v; long_first[i] = base; long_cnt[i] = np; }
    for (int k = 0; k < np; ++k)
      if (base + k < piece_cap) { piece_row[base + k] = v; piece_k[base + k] = k; }
  }
}

// slot-ordered incidence arrays: the OTHER entity of the triple, its relation, the triple id
__global__ void k_dec_slots(const int32_t* __restrict__ X, int N, const int32_t* __restrict__ permv,
                            const int32_t* __restrict__ row_ptr, int V, int32_t* e_other, int32_t* e_rel,
                            int32_t* e_trip) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= row_ptr[V]) return;            // invalid triples sort behind the last row
  const int i = permv[s];
  const int n = i < N ? i : i - N;
  e_other[s] = i < N ? X[3 * n + 2] : X[3 * n];
  e_rel[s] = X[3 * n + 1];
  e_trip[s] = n;
}

// ---- negative sampling on the device ------------------------------------------------
// NegativeSampler.transform (code/common/auxilliaries.py:13-33): the batch tiled (rate + 1) times, labels 1 for the
// first copy and 0 after it; every further row has its object (fair coin) or else its subject replaced by a uniformly
// drawn entity.  Same layout and distribution, counter-based generator instead of numpy's streams.
__global__ void k_negative_sample(const int32_t* __restrict__ batch, int n, int rate, int num_entities, uint64_t seed,
                                  const uint64_t* __restrict__ seed_offset, int32_t* __restrict__ X,
                                  float* __restrict__ Y) {
  if (seed_offset) seed += *seed_offset;      // replayed hipGraphs: replay k draws the corruptions of (captured seed + k)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)n * (rate + 1);
  if (i >= total) return;
  const int b = (int)(i % n);
  int s = batch[3 * b], r = batch[3 * b + 1], o = batch[3 * b + 2];
  if (i >= n) {
    const uint64_t j = (uint64_t)(i - n);
    const uint32_t coin = drop_bits(seed, 0x6e65u, 2 * j);            // 24 random bits
    const uint32_t a = drop_bits(seed, 0x6e66u, 2 * j), c = drop_bits(seed, 0x6e67u, 2 * j + 1);
    const uint64_t u = ((uint64_t)a << 24) | c;                        // 48 random bits
    const int32_t ent = (int32_t)((u * (uint64_t)num_entities) >> 48); // uniform in [0, num_entities)
    if (coin & 1u) o = ent; else s = ent;
  }
  X[3 * i] = s; X[3 * i + 1] = r; X[3 * i + 2] = o;
  Y[i] = i < n ? 1.0f : 0.0f;
}

// ---- K1: energies, dx, loss terms ---------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(256) k_dec_energy(const float* __restrict__ codes, const float* __restrict__ Wr,
                                                    const int32_t* __restrict__ X, const float* __restrict__ Y,
                                                    int N, int Nt, int V, int R, int d, float* __restrict__ dx,
                                                    float* __restrict__ part /* [gridDim.x][2] */) {
  // N = triples of this launch; Nt = triples of the whole batch (the mean's denominator: a relation-sharded run cuts
  // the batch into per-rank slices and adds the partial results up)
  __shared__ float red[4][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = d / VEC;
  float xent = 0.f, sq = 0.f;            // accumulated by lane 0 of each wave
  for (int n = blockIdx.x * 4 + wave; n < N; n += gridDim.x * 4) {
    const int s = X[3 * n], r = X[3 * n + 1], o = X[3 * n + 2];
    const bool ok = (unsigned)s < (unsigned)V && (unsigned)o < (unsigned)V && (unsigned)r < (unsigned)R;
    float x = 0.f, q = 0.f;
    if (ok) {
      const float* p1 = codes + (size_t)s * d;
      const float* pr = Wr + (size_t)r * d;
      const float* p2 = codes + (size_t)o * d;
      for (int c = lane; c < nvec; c += 64) {
        float a[VEC], b[VEC], e[VEC];
        vload<VEC>(p1 + (size_t)c * VEC, a);
        vload<VEC>(pr + (size_t)c * VEC, b);
        vload<VEC>(p2 + (size_t)c * VEC, e);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          x = fmaf(a[k] * b[k], e[k], x);
          q += a[k] * a[k] + b[k] * b[k] + e[k] * e[k];
        }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      x += __shfl_down(x, off, 64);
      q += __shfl_down(q, off, 64);
    }
    if (lane == 0) {
      const float y = Y[n];
      const float ax = fabsf(x);
      const float ex = __expf(-ax);
      const float sig = x >= 0.f ? 1.0f / (1.0f + ex) : ex / (1.0f + ex);
      dx[n] = ok ? (sig - y) / (float)Nt : 0.f;
      if (ok) {
        xent += (1.0f - y) * x + log1pf(ex) + fmaxf(-x, 0.f);
        sq += q;
      }
    }
  }
  if (lane == 0) { red[wave][0] = xent; red[wave][1] = sq; }
  __syncthreads();
  if (threadIdx.x < 2)
    part[(size_t)blockIdx.x * 2 + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// loss = sum(xent)/N + lambda * sum(sq)/(N d)   (double accumulation, one workgroup, fixed order)
__global__ void __launch_bounds__(256) k_dec_loss(const float* __restrict__ part, int nparts, int N, int d,
                                                  float lambda, double* __restrict__ loss) {
  __shared__ double red[256][2];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) { a += part[2 * i]; b += part[2 * i + 1]; }
  red[threadIdx.x][0] = a; red[threadIdx.x][1] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[threadIdx.x][0] += red[threadIdx.x + s][0];
      red[threadIdx.x][1] += red[threadIdx.x + s][1];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = red[0][0] / (double)N + (double)lambda * red[0][1] / ((double)N * (double)d);
}

// ---- K2: entity gradients ------------------------------------------------------------------------
struct EntArgs {
  const float* codes;
  const float* Wr;
  const float* dx;
  const int32_t* row_ptr;
  const int32_t* e_other;
  const int32_t* e_rel;
  const int32_t* e_trip;
  const int32_t* long_rows;
  const int32_t* nlong;
  const int32_t* long_first;
  const int32_t* long_cnt;
  const int32_t* piece_row;
  const int32_t* piece_k;
  float* piece_slab;
  float* dcodes;
  float* dcodes_drop;       // optional: dcodes * dropout of the encoder's top layer (what its self-loop GEMMs consume):
  DropSpec drop;            // saves the encoder's own scale-and-copy pass over [V,d] (top_grad_dropout) in a train step
  int32_t V, d;
  float k;                  // 2 lambda / (N d)
};

template <int VEC>
__device__ __forceinline__ void store_dcodes(const EntArgs& a, size_t off, const float (&acc)[VEC]) {
#pragma clang fp contract(off)
  vstore<VEC>(a.dcodes + off, acc);
  if (a.dcodes_drop != nullptr) {
    float o2[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) o2[k] = acc[k] * drop_factor(a.drop, off + k);
    vstore<VEC>(a.dcodes_drop + off, o2);
  }
}

template <int VEC>
__device__ __forceinline__ void ent_range(const EntArgs& a, int s0, int s1, int step, int cidx, float (&acc)[VEC]) {
  int s = s0;
  for (; s + 3 * step < s1; s += 4 * step) {
    float g[4], u[4][VEC], w[4][VEC];
    int oth[4], rel[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      oth[t] = a.e_other[s + t * step]; rel[t] = a.e_rel[s + t * step]; g[t] = a.dx[a.e_trip[s + t * step]];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      vload<VEC>(a.codes + (size_t)oth[t] * a.d + (size_t)cidx * VEC, u[t]);
      vload<VEC>(a.Wr + (size_t)rel[t] * a.d + (size_t)cidx * VEC, w[t]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] = fmaf(g[t] * u[t][k], w[t][k], acc[k]);
  }
  for (; s < s1; s += step) {
    float u[VEC], w[VEC];
    const float g = a.dx[a.e_trip[s]];
    vload<VEC>(a.codes + (size_t)a.e_other[s] * a.d + (size_t)cidx * VEC, u);
    vload<VEC>(a.Wr + (size_t)a.e_rel[s] * a.d + (size_t)cidx * VEC, w);
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = fmaf(g * u[k], w[k], acc[k]);
  }
}

// kEntThreads = 256 (was 1024): the workgroups are scheduled in finer grain (k_combine gained 10-13 % that way,
// elementwise.hip).  A piece of a long row is still summed over EIGHT interleaved slot lanes whose partial sums are
// added in lane order; 256 threads carry them as 2 physical x 4 virtual lanes -- the same arithmetic as before.
constexpr int kEntThreads = 256;
template <int VEC, int TPR>
__global__ void __launch_bounds__(kEntThreads) k_dec_entity_grad(EntArgs a, int n_long_blocks) {
  const int nvec = a.d / VEC;
  constexpr int NSL = kEntThreads / 128, VL = 8 / NSL;
  if ((int)blockIdx.x < n_long_blocks) {
    __shared__ float red[(NSL - 1) * VL][128 * VEC];
    const int cl = threadIdx.x & 127, sl = threadIdx.x >> 7;
    // one PIECE of a long row per turn, partial sum to the piece slab
    const int n = a.nlong[1];
    for (int lb = blockIdx.x; lb < n; lb += n_long_blocks) {
      const int v = a.piece_row[lb];
      const int beg = a.row_ptr[v] + a.piece_k[lb] * kDecPiece;
      const int end = min(a.row_ptr[v + 1], beg + kDecPiece);
      for (int c0 = 0; c0 < nvec; c0 += 128) {
        const int cidx = c0 + cl;
        float part[VL][VEC];
#pragma unroll
        for (int j = 0; j < VL; ++j) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) part[j][k] = 0.f;
          if (cidx < nvec) ent_range<VEC>(a, beg + sl * VL + j, end, 8, cidx, part[j]);
        }
        if (sl > 0) {
#pragma unroll
          for (int j = 0; j < VL; ++j)
#pragma unroll
            for (int k = 0; k < VEC; ++k) red[(sl - 1) * VL + j][cl * VEC + k] = part[j][k];
        }
        __syncthreads();
        if (sl == 0 && cidx < nvec) {
          float acc[VEC];
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            float t = part[0][k];
#pragma unroll
            for (int j = 1; j < VL; ++j) t += part[j][k];
#pragma unroll
            for (int w = 0; w < (NSL - 1) * VL; ++w) t += red[w][cl * VEC + k];
            acc[k] = t;
          }
          vstore<VEC>(a.piece_slab + (size_t)lb * a.d + (size_t)cidx * VEC, acc);
        }
        __syncthreads();
      }
    }
    return;
  }
  const int v = ((int)blockIdx.x - n_long_blocks) * (kEntThreads / TPR) + threadIdx.x / TPR;
  if (v >= a.V) return;
  const int lane = threadIdx.x % TPR;
  const int beg = a.row_ptr[v], end = a.row_ptr[v + 1];
  if (end - beg > kDecLongRow) return;
  for (int cidx = lane; cidx < nvec; cidx += TPR) {
    const size_t off = (size_t)v * a.d + (size_t)cidx * VEC;
    float acc[VEC], self[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    ent_range<VEC>(a, beg, end, 1, cidx, acc);
    vload<VEC>(a.codes + off, self);
    const float kc = a.k * (float)(end - beg);
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = fmaf(kc, self[k], acc[k]);
    store_dcodes<VEC>(a, off, acc);
  }
}

// long rows: add the pieces of every row in piece order, plus the regulariser's share of the row itself
template <int VEC>
__global__ void __launch_bounds__(256) k_dec_long_finish(EntArgs a) {
  const int nvec = a.d / VEC;
  const int n = a.nlong[0];
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const int v = a.long_rows[i], first = a.long_first[i], cnt = a.long_cnt[i];
    const float kc = a.k * (float)(a.row_ptr[v + 1] - a.row_ptr[v]);
    for (int cidx = threadIdx.x; cidx < nvec; cidx += 256) {
      const size_t off = (size_t)v * a.d + (size_t)cidx * VEC;
      float acc[VEC], self[VEC], part[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
      for (int p = 0; p < cnt; ++p) {
        vload<VEC>(a.piece_slab + (size_t)(first + p) * a.d + (size_t)cidx * VEC, part);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] += part[k];
      }
      vload<VEC>(a.codes + off, self);
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] = fmaf(kc, self[k], acc[k]);
      store_dcodes<VEC>(a, off, acc);
    }
  }
}

// ---- K2, line form (round 4) -----------------------------------------------------------------------
// The kernel above gathers one 2 KB partner row per incidence: 660,000 rows at N = 330,000, 1.28 GB through the
// L2 <-> fabric interface, because the 29 MB code table fits no L2 (8 XCDs x 4 MB).  Every element of dL/dcodes is its
// own sum over the row's incidences, so the columns can be taken band by band: here a band is 32 floats = ONE 128-byte
// line of a band-major, line-aligned COPY of the table (`cb [bands][V][32]`, built per call by k_dec_band_tables, 12 us)
// -- 1.86 MB per band at FB15k-237 size.  The workgroups with blockIdx % 8 == x (one XCD) take band x of every row,
// then band x + 8: the band lives in that XCD's L2 while its 660,000 line gathers go by.  (Round 4's first attempt banded the
// ROW-MAJOR table: 250-byte pieces = three lines each, a 5.6 MB footprint per XCD, and lost to the full-row kernel.)
// Eight lanes (float4 each) own a (row, band); a wavefront takes eight rows per turn, each group walking its row's
// incidences one after the other in k_dec_entity_grad's own order and arithmetic -- every sum is that kernel's, bit for
// bit; pieces of long rows: eight interleaved slot groups, their partial sums added in group order, as there.  The band
// of W_relation ([R][32], 30 KB at R = 237) sits in LDS when it fits.
constexpr int kLine = 32;                 // floats per (row, band)
constexpr int kLineWaves = 4;             // wavefronts per workgroup

struct LineArgs {
  const float* cb;          // [bands][V][32]
  const float* rb;          // [bands][R][32]
  const float* e_g;         // loss gradient of every incidence slot
  const int32_t* row_order; // rows by descending number of incidences
  int32_t R, nbands;
};

__global__ void __launch_bounds__(256) k_dec_band_tables(const float* __restrict__ codes, const float* __restrict__ Wr,
                                                         int V, int R, int d, int nbands, float* __restrict__ cb,
                                                         float* __restrict__ rb) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per = nbands * (kLine / 4);
  const int64_t row = i / per;
  const int q = (int)(i - row * per);
  if (row >= (int64_t)V + R) return;
  const bool ent = row < V;
  const int64_t r = ent ? row : row - V;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (4 * q < d) v = *reinterpret_cast<const float4*>((ent ? codes : Wr) + r * d + 4 * q);
  float* dst = ent ? cb + ((int64_t)(q >> 3) * V + r) * kLine : rb + ((int64_t)(q >> 3) * R + r) * kLine;
  *reinterpret_cast<float4*>(dst + (q & 7) * 4) = v;
}

// One group's sum over the incidences s0, s0 + step, s0 + 2 step, ... < s1, in that order (ent_range's terms and
// order).  The group's eight lanes fetch the (partner, relation, loss gradient) of eight incidences with one load each
// -- lane cl the (base + cl)-th of the sequence -- and hand them round by ds_bpermute; the next eight are requested
// before this block's lines are waited for, so the chain per block is ONE memory level (the lines), not three.
// Lanes past the end of the sequence hold partner 0 / relation 0 / gradient 0: a valid line and a term g u w = +-0,
// which leaves the accumulator as it is, bit for bit (it never holds -0: it starts at +0) -- no branch per incidence.
// Byte offsets are 32-bit (a band of the table is V x 128 bytes) on a uniform base: one VALU add per address.
template <bool RLDS>
__device__ __forceinline__ void line_range(const EntArgs& a, const float* __restrict__ e_g, const char* __restrict__ cbb,
                                           const char* __restrict__ rbb, const char* rl, int s0, int s1, int step,
                                           int g, int cl, float (&acc)[4]) {
  const int n = s1 > s0 ? (s1 - s0 + step - 1) / step : 0;
  uint32_t oth = 0, rel = 0;
  float gg = 0.f;
  if (cl < n) {
    const int s = s0 + cl * step;
    oth = (uint32_t)a.e_other[s]; rel = (uint32_t)a.e_rel[s]; gg = e_g[s];
  }
  const uint32_t cb16 = (uint32_t)cl * 16;
  for (int base = 0; base < n; base += 8) {
    const uint32_t co = oth, cr = rel;
    const float cg = gg;
    float4 u[8], w[8];
    float gt[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const uint32_t ot = (uint32_t)__shfl((int)co, g * 8 + t, 64) * (kLine * 4);
      const uint32_t rt = (uint32_t)__shfl((int)cr, g * 8 + t, 64) * (kLine * 4);
      gt[t] = __shfl(cg, g * 8 + t, 64);
      u[t] = *reinterpret_cast<const float4*>(cbb + (ot + cb16));
      if constexpr (RLDS) w[t] = *reinterpret_cast<const float4*>(rl + (rt + cb16));
      else w[t] = *reinterpret_cast<const float4*>(rbb + (rt + cb16));
    }
    oth = 0; rel = 0; gg = 0.f;
    if (base + 8 + cl < n) {
      const int s = s0 + (base + 8 + cl) * step;
      // (raw indices: nothing may consume the loaded values before the next block's shuffles, or the wait for them --
      // and, the memory counter retiring in order, for this block's lines -- lands here)
      oth = (uint32_t)a.e_other[s]; rel = (uint32_t)a.e_rel[s]; gg = e_g[s];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      acc[0] = fmaf(gt[t] * u[t].x, w[t].x, acc[0]);
      acc[1] = fmaf(gt[t] * u[t].y, w[t].y, acc[1]);
      acc[2] = fmaf(gt[t] * u[t].z, w[t].z, acc[2]);
      acc[3] = fmaf(gt[t] * u[t].w, w[t].w, acc[3]);
    }
  }
}

// e_g[slot] = dx[triple of the slot]: the loss gradients in the order of the by-entity incidence lists
__global__ void k_dec_slot_grad(const int32_t* __restrict__ e_trip, const float* __restrict__ dx,
                                const int32_t* __restrict__ row_ptr, int V, float* __restrict__ e_g) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < row_ptr[V]) e_g[s] = dx[e_trip[s]];
}

// Persistent workgroups: gridDim = 8 x (workgroups per XCD); workgroup (x, q) takes band x of every row, then band
// x + 8, ...: one fill of the relation band per workgroup and pass, and every wavefront of an XCD moves to the next band
// at about the same time.  Inside a pass the band's W wavefronts are dealt the work units round-robin: first the pieces
// of long rows (from the LAST wavefront down: the first ones get the longest short rows), then the turns of eight short
// rows in the order of descending length (b.row_order) -- turn w, w + W, w + 2 W, ... for wavefront w, so every
// wavefront gets long turns and short ones; the next turn's rows and row pointers are requested a turn ahead.
template <bool RLDS>
__global__ void __launch_bounds__(64 * kLineWaves) k_dec_entity_lines(EntArgs a, LineArgs b) {
  extern __shared__ float rl[];                       // RLDS: this band of W_relation, [R][32]
  const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 3, cl = lane & 7;
  const int W = (gridDim.x >> 3) * kLineWaves, w = q * kLineWaves + wave;
  const int npieces = a.nlong[1], nturns = (a.V + 7) >> 3;
  for (int band = x, pass = 0; band < b.nbands; band += 8, ++pass) {
    const char* __restrict__ cbb = reinterpret_cast<const char*>(b.cb + (size_t)band * a.V * kLine);
    const char* __restrict__ rbb = reinterpret_cast<const char*>(b.rb + (size_t)band * b.R * kLine);
    // the first turn's rows: requested before the relation band is copied
    int slot = w * 8 + g, v_n = -1, beg_n = 0, end_n = 0;
    if (slot < a.V) { v_n = b.row_order[slot]; beg_n = a.row_ptr[v_n]; end_n = a.row_ptr[v_n + 1]; }
    if constexpr (RLDS) {
      if (pass > 0) __syncthreads();                  // every wavefront is done with the previous band
      for (int i = threadIdx.x; i < b.R * (kLine / 4); i += 64 * kLineWaves)
        reinterpret_cast<float4*>(rl)[i] = reinterpret_cast<const float4*>(rbb)[i];
      __syncthreads();
    }
    const int col = band * kLine + cl * 4;
    const bool colok = col < a.d;                     // d % 4 == 0: the whole float4 is inside the row, or none of it
    for (int lb = W - 1 - w; lb < npieces; lb += W) {
      const int v = a.piece_row[lb];
      const int beg = a.row_ptr[v] + a.piece_k[lb] * kDecPiece;
      const int end = min(a.row_ptr[v + 1], beg + kDecPiece);
      float part[4] = {0.f, 0.f, 0.f, 0.f};
      line_range<RLDS>(a, b.e_g, cbb, rbb, reinterpret_cast<const char*>(rl), beg + g, end, 8, g, cl, part);
      float t[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        t[k] = __shfl(part[k], cl, 64);
#pragma unroll
        for (int p = 1; p < 8; ++p) t[k] += __shfl(part[k], p * 8 + cl, 64);
      }
      if (g == 0 && colok) vstore<4>(a.piece_slab + (size_t)lb * a.d + col, t);
    }
    for (int turn = w; turn < nturns; turn += W) {
      const int v = v_n, beg = beg_n, end = end_n;
      slot += W * 8;
      v_n = -1; beg_n = end_n = 0;
      if (turn + W < nturns && slot < a.V) { v_n = b.row_order[slot]; beg_n = a.row_ptr[v_n]; end_n = a.row_ptr[v_n + 1]; }
      if (v < 0 || end - beg > kDecLongRow) continue;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      line_range<RLDS>(a, b.e_g, cbb, rbb, reinterpret_cast<const char*>(rl), beg, end, 1, g, cl, acc);
      const float4 self = *reinterpret_cast<const float4*>(cbb + ((size_t)v * kLine + cl * 4) * sizeof(float));
      const float kc = a.k * (float)(end - beg);
      acc[0] = fmaf(kc, self.x, acc[0]);
      acc[1] = fmaf(kc, self.y, acc[1]);
      acc[2] = fmaf(kc, self.z, acc[2]);
      acc[3] = fmaf(kc, self.w, acc[3]);
      if (colok) store_dcodes<4>(a, (size_t)v * a.d + col, acc);
    }
  }
}

// ---- K3: relation gradients ------------------------------------------------------------------------
__device__ __forceinline__ int find_segment(const int32_t* __restrict__ ptr, int n_seg, int x) {
  int lo = 0, hi = n_seg;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (ptr[mid] <= x) lo = mid; else hi = mid;
  }
  return lo;
}

template <int VEC>
__global__ void __launch_bounds__(kRowThreads) k_dec_rel_partial(const float* __restrict__ codes,
                                                                 const float* __restrict__ dx,
                                                                 const int32_t* __restrict__ X,
                                                                 const int32_t* __restrict__ permr,
                                                                 const int32_t* __restrict__ rel_ptr,
                                                                 const int32_t* __restrict__ chunk_ptr, int R,
                                                                 int d, float* __restrict__ slab) {
  __shared__ float red[8][128 * VEC];
  const int bid = blockIdx.x;
  if (bid >= chunk_ptr[R]) return;
  const int rel = find_segment(chunk_ptr, R, bid);
  const int beg = rel_ptr[rel] + (bid - chunk_ptr[rel]) * kDecChunk;
  const int end = min(beg + kDecChunk, rel_ptr[rel + 1]);
  const int cl = threadIdx.x & 127, sl = threadIdx.x >> 7;
  const int nvec = d / VEC;
  for (int c0 = 0; c0 < nvec; c0 += 128) {
    const int cidx = c0 + cl;
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    if (cidx < nvec) {
      int j = beg + sl;
      for (; j + 8 < end; j += 16) {
        const int n0 = permr[j], n1 = permr[j + 8];
        float a0[VEC], b0[VEC], a1[VEC], b1[VEC];
        vload<VEC>(codes + (size_t)X[3 * n0] * d + (size_t)cidx * VEC, a0);
        vload<VEC>(codes + (size_t)X[3 * n0 + 2] * d + (size_t)cidx * VEC, b0);
        vload<VEC>(codes + (size_t)X[3 * n1] * d + (size_t)cidx * VEC, a1);
        vload<VEC>(codes + (size_t)X[3 * n1 + 2] * d + (size_t)cidx * VEC, b1);
        const float g0 = dx[n0], g1 = dx[n1];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = fmaf(g1 * a1[k], b1[k], fmaf(g0 * a0[k], b0[k], acc[k]));
      }
      for (; j < end; j += 8) {
        const int n0 = permr[j];
        float a0[VEC], b0[VEC];
        vload<VEC>(codes + (size_t)X[3 * n0] * d + (size_t)cidx * VEC, a0);
        vload<VEC>(codes + (size_t)X[3 * n0 + 2] * d + (size_t)cidx * VEC, b0);
        const float g0 = dx[n0];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = fmaf(g0 * a0[k], b0[k], acc[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) red[sl][cl * VEC + k] = acc[k];
    __syncthreads();
    if (sl == 0 && cidx < nvec) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        float t = red[0][cl * VEC + k];
#pragma unroll
        for (int w = 1; w < 8; ++w) t += red[w][cl * VEC + k];
        acc[k] = t;
      }
      vstore<VEC>(slab + (size_t)bid * d + (size_t)cidx * VEC, acc);
    }
    __syncthreads();
  }
}

// gWr[r] = sum of r's chunk slabs (chunk order) + k cnt_r Wr[r];   rows >= R of W_relation are never used
// ---- K1 fused with the relation gradient's partials ------------------------------------------------------
// One workgroup per relation chunk (<= kDecChunk triples of ONE relation, in the order of the relation sort): each
// wave takes every fourth triple, keeps the relation's row and the two entity rows in registers (column c of a row
// lives in lane c mod 64, as in k_dec_energy, so the energy is the same number bit for bit), reduces the energy over
// the wave, and -- the point of the fusion -- adds g * e1 * e2 to its register accumulators while the rows are still
// there, instead of a second kernel gathering both rows again (0.2 ms at N = 330,000, and it used to run beside the
// encoder's backward pass and slow its GEMMs).  The four waves' accumulators are summed in a fixed order into the
// chunk's slab row; k_dec_rel_reduce finishes as before.  T = ceil(d / VEC / 64) <= 4; wider rows use the two
// separate kernels.
template <int VEC, int T>
__global__ void __launch_bounds__(256) k_dec_energy_rel(const float* __restrict__ codes, const float* __restrict__ Wr,
                                                        const int32_t* __restrict__ X, const float* __restrict__ Y,
                                                        const int32_t* __restrict__ permr,
                                                        const int32_t* __restrict__ rel_ptr,
                                                        const int32_t* __restrict__ chunk_ptr, int N, int R, int d,
                                                        float* __restrict__ dx, float* __restrict__ part,
                                                        float* __restrict__ slab) {
  extern __shared__ float sm[];                      // [4][d] wave accumulators, then [4][2] loss terms
  const int bid = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (bid >= chunk_ptr[R]) {
    if (threadIdx.x < 2) part[(size_t)bid * 2 + threadIdx.x] = 0.f;
    return;
  }
  const int rel = find_segment(chunk_ptr, R, bid);
  const int beg = rel_ptr[rel] + (bid - chunk_ptr[rel]) * kDecChunk;
  const int end = min(beg + kDecChunk, rel_ptr[rel + 1]);
  const int nvec = d / VEC;
  float rr[T][VEC], acc[T][VEC];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int c = lane + 64 * t;
#pragma unroll
    for (int k = 0; k < VEC; ++k) { rr[t][k] = 0.f; acc[t][k] = 0.f; }
    if (c < nvec) vload<VEC>(Wr + (size_t)rel * d + (size_t)c * VEC, rr[t]);
  }
  float xent = 0.f, sq = 0.f;                        // accumulated by lane 0 of each wave
  // Two triples per turn (this wave's j and j + 4: the order it always took them in), their four entity rows in
  // flight together, and the NEXT turn's indices fetched before this turn's rows are waited for: the kernel is bound
  // by the dependent chain permr -> X -> rows -> shuffle reduction of one triple after another, not by bytes (the
  // tiled order below halves its HBM traffic and took 10 % off its time).
  int j = beg + wave;
  int n0 = j < end ? permr[j] : -1, n1 = j + 4 < end ? permr[j + 4] : -1;
  int s0 = 0, o0 = 0, s1 = 0, o1 = 0;
  if (n0 >= 0) { s0 = X[3 * n0]; o0 = X[3 * n0 + 2]; }          // in range: invalid triples sort behind rel_ptr[R]
  if (n1 >= 0) { s1 = X[3 * n1]; o1 = X[3 * n1 + 2]; }
  while (n0 >= 0) {
    const bool two = n1 >= 0;
    float a[2][T][VEC], e[2][T][VEC];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float* p1 = codes + (size_t)(u ? s1 : s0) * d;
      const float* p2 = codes + (size_t)(u ? o1 : o0) * d;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int c = lane + 64 * t;
#pragma unroll
        for (int k = 0; k < VEC; ++k) { a[u][t][k] = 0.f; e[u][t][k] = 0.f; }
        if (c < nvec && (u == 0 || two)) {
          vload<VEC>(p1 + (size_t)c * VEC, a[u][t]);
          vload<VEC>(p2 + (size_t)c * VEC, e[u][t]);
        }
      }
    }
    const int m0 = n0, m1 = n1;
    const float y0 = Y[m0], y1 = two ? Y[m1] : 0.f;
    j += 8;
    n0 = j < end ? permr[j] : -1;
    n1 = j + 4 < end ? permr[j + 4] : -1;
    if (n0 >= 0) { s0 = X[3 * n0]; o0 = X[3 * n0 + 2]; }
    if (n1 >= 0) { s1 = X[3 * n1]; o1 = X[3 * n1 + 2]; }
    float x[2] = {0.f, 0.f}, q[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int c = lane + 64 * t;
        if (c < nvec) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            x[u] = fmaf(a[u][t][k] * rr[t][k], e[u][t][k], x[u]);
            q[u] += a[u][t][k] * a[u][t][k] + rr[t][k] * rr[t][k] + e[u][t][k] * e[u][t][k];
          }
        }
      }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        x[u] += __shfl_down(x[u], off, 64);
        q[u] += __shfl_down(q[u], off, 64);
      }
    }
    float g[2] = {0.f, 0.f};
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 1 && !two) break;
        const float y = u ? y1 : y0;
        const float ax = fabsf(x[u]);
        const float ex = __expf(-ax);
        const float sig = x[u] >= 0.f ? 1.0f / (1.0f + ex) : ex / (1.0f + ex);
        g[u] = (sig - y) / (float)N;
        dx[u ? m1 : m0] = g[u];
        xent += (1.0f - y) * x[u] + log1pf(ex) + fmaxf(-x[u], 0.f);
        sq += q[u];
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float gu = __shfl(g[u], 0, 64);
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[t][k] = fmaf(gu * a[u][t][k], e[u][t][k], acc[t][k]);
    }
  }
  float* red = sm;
  float* lossred = sm + (size_t)4 * d;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int c = lane + 64 * t;
    if (c < nvec) vstore<VEC>(red + (size_t)wave * d + (size_t)c * VEC, acc[t]);
  }
  if (lane == 0) { lossred[wave * 2] = xent; lossred[wave * 2 + 1] = sq; }
  __syncthreads();
  for (int i = threadIdx.x; i < nvec * VEC; i += 256)
    slab[(size_t)bid * d + i] = ((red[i] + red[d + i]) + red[2 * (size_t)d + i]) + red[3 * (size_t)d + i];
  if (threadIdx.x < 2)
    part[(size_t)bid * 2 + threadIdx.x] = ((lossred[threadIdx.x] + lossred[2 + threadIdx.x]) + lossred[4 + threadIdx.x]) +
                                          lossred[6 + threadIdx.x];
}

__global__ void k_dec_rel_reduce(const float* __restrict__ slab, const int32_t* __restrict__ chunk_ptr,
                                 const int32_t* __restrict__ rel_ptr, const float* __restrict__ Wr,
                                 float* __restrict__ gWr, int R, int d, float k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * d) return;
  const int rel = i / d, col = i - rel * d;
  const int c0 = chunk_ptr[rel], c1 = chunk_ptr[rel + 1];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int c = c0;
  for (; c + 4 <= c1; c += 4) {
    const float* p = slab + (size_t)c * d + col;
    a0 += p[0]; a1 += p[d]; a2 += p[2 * (size_t)d]; a3 += p[3 * (size_t)d];
  }
  for (; c < c1; ++c) a0 += slab[(size_t)c * d + col];
  const float cnt = (float)(rel_ptr[rel + 1] - rel_ptr[rel]);
  gWr[i] = fmaf(k * cnt, Wr[i], (a0 + a1) + (a2 + a3));
}

__global__ void k_loss_narrow(const double* __restrict__ loss, float* __restrict__ f) { f[0] = (float)loss[0]; }
__global__ void k_loss_widen(const float* __restrict__ f, double* __restrict__ loss) { loss[0] = (double)f[0]; }

int bits_for(uint32_t max_value) {
  int b = 1;
  while (b < 32 && (1ull << b) <= max_value) ++b;
  return b;
}

template <class T>
rgcn_status dalloc(rgcn_ctx* c, T** p, size_t n) {
  RGCN_HIP(c, hipMalloc((void**)p, (n ? n : 1) * sizeof(T)));
  return RGCN_OK;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

rgcn_status negative_sample(rgcn_ctx* c, const int32_t* batch_dev, int64_t n, int rate, uint64_t seed, int32_t* X,
                            float* Y) {
  const int64_t total = n * (rate + 1);
  if (total <= 0) return RGCN_OK;
  ProfScope ps(c, "negative_sample", 12.0 * n + 16.0 * total, 0);
  hipLaunchKernelGGL(k_negative_sample, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, batch_dev,
                     (int)n, rate, c->V, seed, c->capturing ? c->replay_counter : nullptr, X, Y);
  RGCN_HIP(c, hipGetLastError());
  c->dec.tiled_X = X;
  c->dec.tiled_period = n;
  c->dec.tiled_N = total;
  return RGCN_OK;
}

rgcn_status decoder_reserve(rgcn_ctx* c, int64_t maxN) {
  DecoderBufs& q = c->dec;
  if (q.maxN >= maxN && q.maxN > 0) return RGCN_OK;
  decoder_free(c);
  const size_t N = (size_t)maxN, V = c->V, R = c->R, d = c->d;
  RGCN_TRY(dalloc(c, &q.keyv, 2 * N)); RGCN_TRY(dalloc(c, &q.keyv_s, 2 * N));
  RGCN_TRY(dalloc(c, &q.valv, 2 * N)); RGCN_TRY(dalloc(c, &q.permv, 2 * N));
  RGCN_TRY(dalloc(c, &q.keyr, N)); RGCN_TRY(dalloc(c, &q.keyr_s, N));
  RGCN_TRY(dalloc(c, &q.valr, N)); RGCN_TRY(dalloc(c, &q.permr, N)); RGCN_TRY(dalloc(c, &q.perm_pos, N));
  RGCN_TRY(dalloc(c, &q.row_ptr, V + 1)); RGCN_TRY(dalloc(c, &q.rel_ptr, R + 1)); RGCN_TRY(dalloc(c, &q.chunk_ptr, R + 1));
  RGCN_TRY(dalloc(c, &q.e_other, 2 * N)); RGCN_TRY(dalloc(c, &q.e_rel, 2 * N)); RGCN_TRY(dalloc(c, &q.e_trip, 2 * N));
  q.long_cap = (int32_t)(2 * N / kDecLongRow + 1);
  RGCN_TRY(dalloc(c, &q.long_rows, (size_t)q.long_cap));
  RGCN_TRY(dalloc(c, &q.long_first, (size_t)q.long_cap));
  RGCN_TRY(dalloc(c, &q.long_cnt, (size_t)q.long_cap));
  q.piece_cap = (int32_t)(2 * N / kDecPiece + q.long_cap + 1);     // sum of ceil(len / piece) over the long rows
  RGCN_TRY(dalloc(c, &q.piece_row, (size_t)q.piece_cap));
  RGCN_TRY(dalloc(c, &q.piece_k, (size_t)q.piece_cap));
  RGCN_TRY(dalloc(c, &q.piece_slab, (size_t)q.piece_cap * d));
  RGCN_TRY(dalloc(c, &q.nlong, 2));
  RGCN_TRY(dalloc(c, &q.dx, N));
  q.nbands = (int32_t)((d + kLine - 1) / kLine);
  {
    int dev = 0, cus = 0;
    RGCN_HIP(c, hipGetDevice(&dev));
    RGCN_HIP(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    q.cus = cus > 0 ? cus : 256;
  }
  RGCN_TRY(dalloc(c, &q.cb, (size_t)q.nbands * V * kLine));
  RGCN_TRY(dalloc(c, &q.rb, (size_t)q.nbands * R * kLine));
  RGCN_TRY(dalloc(c, &q.e_g, 2 * N));
  RGCN_TRY(dalloc(c, &q.row_key, V)); RGCN_TRY(dalloc(c, &q.row_key_s, V)); RGCN_TRY(dalloc(c, &q.row_order, V));
  RGCN_TRY(dalloc(c, &q.row_tab, sort_table_elems(V)));
  q.energy_blocks = 2048;
  q.max_chunks = (int32_t)(N / kDecChunk + R + 1);
  RGCN_TRY(dalloc(c, &q.loss_part, 2 * (size_t)(q.energy_blocks > q.max_chunks ? q.energy_blocks : q.max_chunks)));
  RGCN_TRY(dalloc(c, &q.loss, 1));
  RGCN_TRY(dalloc(c, &q.slab, (size_t)q.max_chunks * d));
  RGCN_TRY(dalloc(c, &q.keyv_t, 2 * N)); RGCN_TRY(dalloc(c, &q.keyr_t, N));
  RGCN_TRY(dalloc(c, &q.tablev, sort_table_elems(2 * N))); RGCN_TRY(dalloc(c, &q.tabler, sort_table_elems(N)));
  RGCN_HIP(c, hipEventCreateWithFlags(&q.ev_ready, order_event_flags(c)));
  if (!c->dcodes_own) RGCN_HIP(c, hipMalloc((void**)&c->dcodes_own, sizeof(float) * V * d));
  q.maxN = maxN;
  return RGCN_OK;
}

void decoder_free(rgcn_ctx* c) {
  DecoderBufs& q = c->dec;
  void* ptrs[] = {q.keyv, q.keyv_s, q.valv, q.permv, q.keyr, q.keyr_s, q.valr, q.permr, q.row_ptr, q.rel_ptr,
                  q.chunk_ptr, q.e_other, q.e_rel, q.e_trip, q.long_rows, q.nlong, q.dx, q.loss_part, q.loss,
                  q.slab, q.keyv_t, q.keyr_t, q.tablev, q.tabler, q.long_first, q.long_cnt, q.piece_row, q.piece_k,
                  q.piece_slab, q.perm_pos, q.cb, q.rb, q.e_g, q.row_key, q.row_key_s, q.row_order, q.row_tab};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (q.ev_ready) (void)hipEventDestroy(q.ev_ready);
  q = DecoderBufs();
}

// CSR of the decoder batch (depends on X only).  Runs on whatever stream is current.
rgcn_status decoder_prepare(rgcn_ctx* c, const int32_t* X, int64_t N64, int64_t N_total) {
  DecoderBufs& q = c->dec;
  const int N = (int)N64, V = c->V, R = c->R;
  if (N64 > q.maxN) RGCN_FAIL(c, RGCN_ERR_INVALID, "decoder batch larger than rgcn_decoder_reserve'd");
  q.N = N;
  q.N_total = N_total > 0 ? N_total : N64;
  q.X = X;
  // a batch the negative sampler tiled (DecoderBufs::tiled_X): sort its first `period` triples by relation and expand
  int period = 0, copies = 1;
  if (X == q.tiled_X && N64 == q.tiled_N && (N_total <= 0 || N_total == N64) && q.tiled_period > 1 &&
      N64 % q.tiled_period == 0) {
    period = (int)q.tiled_period;
    copies = (int)(N64 / q.tiled_period);
  }
  const int n_rel = period > 0 ? period : N;
  RGCN_HIP(c, hipMemsetAsync(q.nlong, 0, 2 * sizeof(int32_t), c->stream));
  const int T = 256;
  if (N > 0) {
    {
      ProfScope ps(c, "dec_keys", 36.0 * N, 0);
      hipLaunchKernelGGL(k_dec_keys, dim3((2 * N + T - 1) / T), dim3(T), 0, c->stream, X, N, V, R, n_rel, q.keyv, q.keyr,
                         c->g.errflag);
    }
    // entity incidences (2N, by entity) and triples (N, or their first copies, by relation): one call, shared launches
    SortSpec sp[2];
    sp[0] = SortSpec{q.keyv, q.keyv_s, q.permv, q.keyv_t, q.valv, nullptr, q.tablev, (int64_t)2 * N, (uint32_t)V};
    sp[1] = SortSpec{q.keyr, q.keyr_s, period > 0 ? q.perm_pos : q.permr, q.keyr_t, q.valr, nullptr, q.tabler,
                     (int64_t)n_rel, (uint32_t)R};
    RGCN_TRY(sort_pairs(c, "dec_sort", 2, sp));
    if (period > 0)
      hipLaunchKernelGGL(k_dec_expand, dim3((N + T - 1) / T), dim3(T), 0, c->stream, X, N, V, R, period, copies, q.perm_pos,
                         q.permr, c->g.errflag);
  }
  {
    ProfScope ps(c, "dec_ptrs", 8.0 * (V + R), 0);
    hipLaunchKernelGGL(k_dec_ptrs, dim3(1 + (V + 1 + 1023) / 1024), dim3(1024), 0, c->stream, q.keyv_s, q.keyr_s, N,
                       V, R, n_rel, copies, q.row_ptr, q.long_rows, q.nlong, q.long_cap, q.long_first, q.long_cnt, q.piece_row,
                       q.piece_k, q.piece_cap, q.rel_ptr, q.chunk_ptr, q.row_key);
  }
  {
    // (one 8-bit pass of the library's stable radix sort over V keys)
    SortSpec sp{q.row_key, q.row_key_s, q.row_order, q.row_key_s, q.row_order, nullptr, q.row_tab, (int64_t)V, 255u};
    RGCN_TRY(sort_pairs(c, "dec_row_order", 1, &sp));
  }
  if (N > 0) {
    ProfScope ps(c, "dec_slots", 40.0 * N, 0);
    hipLaunchKernelGGL(k_dec_slots, dim3((2 * N + T - 1) / T), dim3(T), 0, c->stream, X, N, q.permv, q.row_ptr, V,
                       q.e_other, q.e_rel, q.e_trip);
  }
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

// loss + gradients w.r.t. the codes (-> c->dcodes_own) and W_relation (-> its grad buffer)
rgcn_status decoder_compute(rgcn_ctx* c, const float* codes, const float* Y, float reg_param, float* dcodes_drop,
                            const DropSpec* drop) {
  DecoderBufs& q = c->dec;
  const int N = q.N, V = c->V, R = c->R, d = c->d;
  const int Nt = q.N_total > 0 ? (int)q.N_total : N;      // denominator of the batch mean
  const float* Wr = c->w_rel;
  float* gWr = c->g_rel;
  if (N < 0 || (N == 0 && q.N_total <= 0)) RGCN_FAIL(c, RGCN_ERR_INVALID, "empty decoder batch");
  const bool vec4 = (d % 4 == 0) && aligned16(codes) && aligned16(Wr) && aligned16(c->dcodes_own) && aligned16(q.slab) &&
                    aligned16(dcodes_drop);
  const double Nd = (double)Nt * d;
  const float k = (float)(2.0 * reg_param / Nd);
  // energies + loss terms (+ the relation gradient's chunk partials when the rows fit a wave's registers)
  const int nvec_e = vec4 ? d / 4 : d;
  const bool fused = nvec_e <= 256;
  if (fused) {
    const int T = nvec_e <= 64 ? 1 : (nvec_e <= 128 ? 2 : 4);
    const size_t lds = ((size_t)4 * d + 8) * sizeof(float);
    // design: three row gathers per triple (what the kernel asks of L2); compulsory: the code table, the relation
    // rows and the batch once, the per-triple gradients and chunk partials once
    ProfScope ps(c, "dec_energy_rel", 12.0 * N * d + 20.0 * N + 4.0 * q.max_chunks * d, 9.0 * N * d,
                 4.0 * d * ((double)V + R) + 20.0 * N + 4.0 * q.max_chunks * d);
#define RGCN_LAUNCH_ER(VEC, TT)                                                                                    \
  hipLaunchKernelGGL((k_dec_energy_rel<VEC, TT>), dim3(q.max_chunks), dim3(256), lds, c->stream, codes, Wr, q.X, Y, \
                     q.permr, q.rel_ptr, q.chunk_ptr, Nt, R, d, q.dx, q.loss_part, q.slab)
    if (vec4) {
      if (T == 1) RGCN_LAUNCH_ER(4, 1); else if (T == 2) RGCN_LAUNCH_ER(4, 2); else RGCN_LAUNCH_ER(4, 4);
    } else {
      if (T == 1) RGCN_LAUNCH_ER(1, 1); else if (T == 2) RGCN_LAUNCH_ER(1, 2); else RGCN_LAUNCH_ER(1, 4);
    }
#undef RGCN_LAUNCH_ER
  } else {
    ProfScope ps(c, "dec_energy", 12.0 * N * d + 20.0 * N, 6.0 * N * d, 4.0 * d * ((double)V + R) + 20.0 * N);
    if (vec4)
      hipLaunchKernelGGL((k_dec_energy<4>), dim3(q.energy_blocks), dim3(256), 0, c->stream, codes, Wr, q.X, Y, N, Nt, V,
                         R, d, q.dx, q.loss_part);
    else
      hipLaunchKernelGGL((k_dec_energy<1>), dim3(q.energy_blocks), dim3(256), 0, c->stream, codes, Wr, q.X, Y, N, Nt, V,
                         R, d, q.dx, q.loss_part);
  }
  {
    // dL/dW_relation depends on dx only and is not needed before the optimizer: side stream 2, so that it runs
    // beside the entity gradient (forked right behind the energy kernel) and the encoder's backward pass (the caller
    // joins it: stream_join(c, 2))
    StreamScope side(c, 2);
    // the loss itself (one workgroup summing the kernel's partials) is nobody's input on the device: it goes there too
    hipLaunchKernelGGL(k_dec_loss, dim3(1), dim3(256), 0, c->stream, q.loss_part, fused ? q.max_chunks : q.energy_blocks,
                       Nt, d, reg_param, q.loss);
    ProfScope ps(c, "dec_relation_grad", fused ? 4.0 * q.max_chunks * d + 8.0 * R * d : 8.0 * N * d + 8.0 * R * d,
                 fused ? 0.0 : 3.0 * N * d);
    if (fused) {
      // the chunk partials are already in the slab (k_dec_energy_rel)
    } else if (vec4)
      hipLaunchKernelGGL((k_dec_rel_partial<4>), dim3(q.max_chunks), dim3(kRowThreads), 0, c->stream, codes, q.dx,
                         q.X, q.permr, q.rel_ptr, q.chunk_ptr, R, d, q.slab);
    else
      hipLaunchKernelGGL((k_dec_rel_partial<1>), dim3(q.max_chunks), dim3(kRowThreads), 0, c->stream, codes, q.dx,
                         q.X, q.permr, q.rel_ptr, q.chunk_ptr, R, d, q.slab);
    hipLaunchKernelGGL(k_dec_rel_reduce, dim3((R * d + 255) / 256), dim3(256), 0, c->stream, q.slab, q.chunk_ptr,
                       q.rel_ptr, Wr, gWr, R, d, k);
  }
  {
    EntArgs a;
    a.codes = codes; a.Wr = Wr; a.dx = q.dx; a.row_ptr = q.row_ptr; a.e_other = q.e_other; a.e_rel = q.e_rel;
    a.e_trip = q.e_trip; a.long_rows = q.long_rows; a.nlong = q.nlong; a.dcodes = c->dcodes_own; a.V = V; a.d = d;
    a.long_first = q.long_first; a.long_cnt = q.long_cnt; a.piece_row = q.piece_row; a.piece_k = q.piece_k;
    a.piece_slab = q.piece_slab;
    a.k = k;
    a.dcodes_drop = (dcodes_drop != nullptr && drop != nullptr && drop->mode != DROP_NONE) ? dcodes_drop : nullptr;
    if (drop != nullptr) a.drop = *drop; else a.drop = DropSpec{DROP_NONE, 0, 0, 1.0f, 0, nullptr, nullptr};
    // line form (whenever the rows are float4-addressable); the devtools build's RGCN_DEC_LINES=0 selects the full-row
    // kernel, whose sums it reproduces bit for bit (tests/test_gpu_train_step.py::test_entity_gradient_forms_are_bitwise_equal)
    const bool lines = vec4 && knob("RGCN_DEC_LINES", 1) != 0 && (int64_t)V * kLine * 4 < (1ll << 32);
    if (lines) {
      {
        ProfScope ps(c, "dec_band_tables", (4.0 * d + 4.0 * kLine * q.nbands) * ((double)V + R), 0);
        const int64_t threads = ((int64_t)V + R) * q.nbands * (kLine / 4);
        hipLaunchKernelGGL(k_dec_band_tables, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c->stream, codes, Wr, V,
                           R, d, q.nbands, q.cb, q.rb);
      }
      if (N > 0)
        hipLaunchKernelGGL(k_dec_slot_grad, dim3((unsigned)((2 * N + 255) / 256)), dim3(256), 0, c->stream, q.e_trip, q.dx,
                           q.row_ptr, V, q.e_g);
      LineArgs b;
      b.cb = q.cb; b.rb = q.rb; b.R = R; b.nbands = q.nbands; b.e_g = q.e_g; b.row_order = q.row_order;
      const size_t lds = (size_t)R * kLine * sizeof(float);
      const bool rlds = lds <= 64 * 1024;
      // persistent workgroups: three per CU (measured: 768 workgroups 123 us, 1,024: 127, 512: 132), a multiple of 8
      const int wgs = 12 / kLineWaves * q.cus;
      dim3 grid((unsigned)((wgs + 7) / 8 * 8)), block(64 * kLineWaves);
      // design: one line of the partner row and of the relation row per incidence and band (what the kernel asks of the
      // L2s) + the index lists once per band + the rows written; compulsory as for the full-row kernel
      ProfScope ps(c, "dec_entity_grad", (8.0 * kLine + 12.0) * q.nbands * 2.0 * N + 8.0 * V * d, 6.0 * N * d,
                   4.0 * d * (2.0 * V + R) + 32.0 * N);
      if (rlds) hipLaunchKernelGGL((k_dec_entity_lines<true>), grid, block, lds, c->stream, a, b);
      else hipLaunchKernelGGL((k_dec_entity_lines<false>), grid, block, 0, c->stream, a, b);
    } else {
      const int nvec = vec4 ? d / 4 : d;
      const int tpr = nvec <= 64 ? 64 : (nvec <= 128 ? 128 : 256);
      const int rpb = kEntThreads / tpr;
      int64_t want = 2 * (int64_t)N / 2048;
      // (sized for 1024-thread workgroups; four times as many of the 256-thread ones walk the long-row pieces)
      const int nlb = 4 * (int)(want < 64 ? 64 : (want > 1024 ? 1024 : want));
      dim3 grid(nlb + (V + rpb - 1) / rpb), block(kEntThreads);
      // design: two row gathers per incidence (2N incidences) + the row written; compulsory: codes and relation rows
      // once, the incidence lists and per-triple gradients once, dL/dcodes written once
      ProfScope ps(c, "dec_entity_grad", 16.0 * N * d + 8.0 * V * d, 6.0 * N * d,
                   4.0 * d * (2.0 * V + R) + 32.0 * N);
#define RGCN_LAUNCH_EG(VEC, TPR) hipLaunchKernelGGL((k_dec_entity_grad<VEC, TPR>), grid, block, 0, c->stream, a, nlb)
      if (vec4) {
        if (tpr == 64) RGCN_LAUNCH_EG(4, 64); else if (tpr == 128) RGCN_LAUNCH_EG(4, 128); else RGCN_LAUNCH_EG(4, 256);
      } else {
        if (tpr == 64) RGCN_LAUNCH_EG(1, 64); else if (tpr == 128) RGCN_LAUNCH_EG(1, 128); else RGCN_LAUNCH_EG(1, 256);
      }
#undef RGCN_LAUNCH_EG
    }
    if (vec4) hipLaunchKernelGGL((k_dec_long_finish<4>), dim3(128), dim3(256), 0, c->stream, a);
    else hipLaunchKernelGGL((k_dec_long_finish<1>), dim3(128), dim3(256), 0, c->stream, a);
  }
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

// Relation-sharded train step: every rank ran the decoder on ITS slice of the batch's triples (means normalised by the
// whole batch's size), so dL/dcodes, dL/dW_relation and the loss are partial sums: three all-reduces ([V,d], [R,d], one
// float).  The reference has no counterpart (single device); DistMult's loss and both gradients are sums over triples
// (bilinear_diag.py:27-34,63-69), which is what makes the split exact up to summation order.
rgcn_status decoder_allreduce(rgcn_ctx* c) {
  DecoderBufs& q = c->dec;
  RGCN_TRY(comm_allreduce(c, c->dcodes_own, (int64_t)c->V * c->d));
  RGCN_TRY(comm_allreduce(c, c->g_rel, (int64_t)c->R * c->d));
  hipLaunchKernelGGL(k_loss_narrow, dim3(1), dim3(1), 0, c->stream, q.loss, q.loss_part);
  RGCN_TRY(comm_allreduce(c, q.loss_part, 1));
  hipLaunchKernelGGL(k_loss_widen, dim3(1), dim3(1), 0, c->stream, q.loss_part, q.loss);
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

}  // namespace rgcn
