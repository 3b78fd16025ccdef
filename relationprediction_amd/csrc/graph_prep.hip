// Device-side graph preparation: everything the reference derives from the `graph_edges`
// placeholder inside session.run (code/extras/graph_representations.py:21-27 edge split,
// :82-93 / :122-133 'global' normalisation = per-row softmax of ones) plus the layouts the HIP
// kernels want:
//
//   * indeg / outdeg over ALL fed edges (normalisation is global, also under relation sharding);
//   * the MESSAGE list: 2E messages (m <  E: forward  message of edge m, dst = object,  src = subject,
//                                     rel2 = r,     norm = 1/indeg[object];
//                                    m >= E: backward message of edge m-E, dst = subject, src = object,
//                                     rel2 = R + r, norm = 1/outdeg[subject]),
//     stably sorted by rel2 ("per-relation CSR": rel_ptr) and cut into chunks of <= `chunk`
//     messages of one relation (chunk_ptr) -- the unit of work of the block-diagonal kernels;
//   * the INCIDENCE CSR: 2E incidences (i < E: edge i seen from its object, i >= E: edge i-E seen
//     from its subject) stably sorted by vertex (row_ptr).  Message m is written to slot pos[m] in
//     the forward pass (rows = destinations) and to slot pos[(m+E) mod 2E] in the backward pass
//     (rows = sources): one sort serves both directions.
//
// All of it is stream-ordered with no host synchronisation, so a step can be graph-captured.
// The two sorts are ONE call of the library's own stable LSD radix sort (csr_sort.hip: both sorts share its
// launches; deterministic order => deterministic fp32 sums), whose last pass also writes the slot of every incidence.
#include "rgcn_internal.h"

namespace rgcn {

namespace {

constexpr int kScanThreads = 1024;
constexpr int kScanItems = 16;

struct ScanJob {
  const int32_t* in;
  int32_t* out;     // n + 1 entries
  int32_t n;
  int32_t div;      // 0: identity, else value -> ceil(value / div)
  int32_t* where;   // optional: where[out[i]] = i for every i with in[i] != 0 (the compacted index list of a 0/1 input)
};
struct ScanJobs {
  ScanJob j[5];
};

// One workgroup per job; exclusive scan with carry over tiles of 1024 x 16 elements.
__global__ void __launch_bounds__(kScanThreads) k_exscan(ScanJobs jobs) {
  const ScanJob job = jobs.j[blockIdx.x];
  __shared__ int32_t wsum[kScanThreads / 64];
  __shared__ int32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < job.n; base += kScanThreads * kScanItems) {
    int32_t v[kScanItems];
    int32_t tsum = 0;
    const int start = base + tid * kScanItems;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      int idx = start + k;
      int32_t x = idx < job.n ? job.in[idx] : 0;
      if (job.div) x = (x + job.div - 1) / job.div;
      v[k] = tsum;   // exclusive within the thread
      tsum += x;
    }
    // inclusive wave scan of tsum
    int32_t incl = tsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int32_t t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    int32_t wbase = 0;
    for (int w = 0; w < wid; ++w) wbase += wsum[w];
    int32_t total = 0;
    for (int w = 0; w < kScanThreads / 64; ++w) total += wsum[w];
    const int32_t carry = carry_s;
    const int32_t tbase = carry + wbase + incl - tsum;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      int idx = start + k;
      if (idx < job.n) {
        job.out[idx] = tbase + v[k];
        if (job.where != nullptr && job.in[idx] != 0) job.where[tbase + v[k]] = idx;
      }
    }
    __syncthreads();
    if (tid == 0) carry_s = carry + total;
    __syncthreads();
  }
  if (tid == 0) job.out[job.n] = carry_s;
}

// One thread per incidence: sort keys for the two sorts; the first E threads also count the GLOBAL
// degrees (all fed edges, owned or not) and validate the ids.
__global__ void k_keys(const int32_t* __restrict__ tri, int E, int V, int R,
                       const int32_t* __restrict__ owner, int rank, uint32_t* keyv,
                       uint32_t* keyr, int count_degrees, int32_t* indeg, int32_t* outdeg,
                       int32_t* errflag, int32_t* small_counters, uint32_t vmul) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  // one GPU: the degrees are overwritten row by row (k_ptrs reads them off the sorted rows), only the long / giant row
  // counters need a zero before k_ptrs counts into them -- no memset launch
  if (small_counters != nullptr && i < 3) small_counters[i] = 0;
  if (i >= 2 * E) return;
  const bool fwd = i < E;
  const int e = fwd ? i : i - E;
  int s = tri[3 * e], r = tri[3 * e + 1], o = tri[3 * e + 2];
  bool ok = (unsigned)s < (unsigned)V && (unsigned)o < (unsigned)V && (unsigned)r < (unsigned)R;
  if (fwd) {
    if (!ok) {
      atomicOr(errflag, 1);
    } else if (count_degrees) {
      // sharded run: the CSR below holds only this rank's relations, so the GLOBAL degrees are counted
      // here (a hub serialises these atomics; the single-GPU path reads them off the sorted rows instead)
      atomicAdd(&indeg[o], 1);
      atomicAdd(&outdeg[s], 1);
    }
  }
  bool owned = ok && owner[r] == rank;
  // incidence i sits at this vertex.  vmul = 2R: the key also carries the directed relation, so that a row's slots come
  // out ordered by relation (runs of equal relations let the destination-major layer kernel keep a relation's
  // coefficients in registers); vmul = 1 (key range too wide for that): by vertex alone, ties in incidence order
  const uint32_t rel2 = (uint32_t)(fwd ? r : R + r);
  keyv[i] = owned ? (uint32_t)(fwd ? o : s) * vmul + (vmul > 1 ? rel2 : 0u) : (uint32_t)V * vmul;
  keyr[i] = owned ? rel2 : (uint32_t)(2 * R);
}

__device__ __forceinline__ int lower_bound_u32(const uint32_t* a, int n, uint32_t x) {
  int lo = 0, hi = n;   // first index with a[i] >= x
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// CSR offsets straight from the sorted keys (no counting atomics, no scan):
//   blocks 1..: row_ptr[v] = lower_bound(vertex keys, v); row_end / long-row list alongside;
//   block 0  : rel_ptr[r] = lower_bound(relation keys, r) and the exclusive scan of the per-relation
//              chunk counts (2R <= a few thousand entries: one workgroup).
__global__ void __launch_bounds__(1024) k_ptrs(const uint32_t* __restrict__ keyv_s,
                                               const uint32_t* __restrict__ keyr_s,
                                               const int32_t* __restrict__ permv, int E, int degrees_from_rows,
                                               int32_t* indeg, int32_t* outdeg, int M, int V, int R2,
                                               int chunk, int32_t* row_ptr,
                                               int32_t* long_rows, int32_t* nlong, int cap, int giant_threshold,
                                               int32_t* giant_rows, int32_t* giant_first, int32_t* giant_cnt,
                                               int32_t* piece_row, int32_t* piece_k, int32_t* ngiant, int giant_cap,
                                               int piece_cap, int32_t* rel_ptr, int32_t* chunk_ptr,
                                               uint32_t* row_key, int32_t* has_dir, uint32_t vmul) {
  if (blockIdx.x == 0) {
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base <= R2; base += 1024) {
      const int r = base + tid;
      int lo = 0, nchunks = 0;
      if (r <= R2) {
        lo = lower_bound_u32(keyr_s, M, (uint32_t)r);
        rel_ptr[r] = lo;
        if (r < R2) {
          const int hi = lower_bound_u32(keyr_s, M, (uint32_t)(r + 1));
          nchunks = (hi - lo + chunk - 1) / chunk;
        }
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
      if (r <= R2) chunk_ptr[r] = carry + wbase + incl - nchunks;
      __syncthreads();
      if (tid == 0) carry_s = carry + total;
      __syncthreads();
    }
    return;
  }
  const int v = (blockIdx.x - 1) * blockDim.x + threadIdx.x;
  if (v > V) return;
  const int beg = lower_bound_u32(keyv_s, M, (uint32_t)v * vmul);
  row_ptr[v] = beg;
  if (v == V) return;
  const int end = lower_bound_u32(keyv_s, M, (uint32_t)(v + 1) * vmul);
  if (row_key != nullptr) row_key[v] = end - beg > kLongRow ? (uint32_t)(kLongRow + 1) : (uint32_t)(kLongRow - (end - beg));
  if (degrees_from_rows || has_dir != nullptr) {
    // row v holds every LOCAL incidence of v, the incidences with index < E (edges arriving at v: forward-direction
    // messages, directed relation < R) first -- by the stable sort's tie order, or by the relation part of the key -> their
    // count = position of the first index >= E.  All relations local: that is the in-degree
    int lo = beg, hi = end;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (permv[mid] < E) lo = mid + 1; else hi = mid;
    }
    if (degrees_from_rows) {
      indeg[v] = lo - beg;
      outdeg[v] = end - lo;
    }
    if (has_dir != nullptr) {      // basis kind: which (row, direction) units exist
      has_dir[v] = lo > beg ? 1 : 0;
      has_dir[V + v] = end > lo ? 1 : 0;
    }
  }
  if (end - beg > giant_threshold) {
    // pieces of one row get consecutive ids: the finishing pass adds them in a fixed order
    const int np = (end - beg + kGiantRow - 1) / kGiantRow;
    const int i = atomicAdd(ngiant, 1);
    const int base = atomicAdd(ngiant + 1, np);
    if (i < giant_cap) { giant_rows[i] = v; giant_first[i] = base; giant_cnt[i] = np; }
    for (int k = 0; k < np; ++k)
      if (base + k < piece_cap) { piece_row[base + k] = v; piece_k[base + k] = k; }
  } else if (end - beg > kLongRow) {
    const int i = atomicAdd(nlong, 1);
    if (i < cap) long_rows[i] = v;
  }
}

// (The long-row list comes out of k_ptrs in the order its atomics happened to run.  Nothing depends on that order:
// every long row is summed by exactly one workgroup, in slot order, whichever workgroup draws it.)

__device__ __forceinline__ int upper_bound_dev(const int32_t* a, int n, int x) {
  // first index i in [0,n) with a[i] > x
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (a[mid] <= x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ void k_build_msgs(const int32_t* __restrict__ tri, int E, int V, int R, int norm_mode,
                             const int32_t* __restrict__ permr, const int32_t* __restrict__ rel_ptr,
                             const int32_t* __restrict__ pos, const int32_t* __restrict__ indeg,
                             const int32_t* __restrict__ outdeg, const int32_t* __restrict__ cum_in,
                             const int32_t* __restrict__ cum_out, int32_t* m_src, int32_t* m_dst,
                             int32_t* m_dslot, int32_t* m_sslot, float* m_norm,
                             const uint32_t* __restrict__ keyr_s, int slot_arrays, int32_t* d_src,
                             int32_t* d_rel, float* d_norm, int32_t* s_dst, int32_t* s_rel, float* s_norm) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= 2 * E) return;
  if (j >= rel_ptr[2 * R]) return;    // beyond the owned messages
  const int m = permr[j];
  const bool fwd = m < E;
  const int e = fwd ? m : m - E;
  const int s = tri[3 * e], o = tri[3 * e + 2];
  const int src = fwd ? s : o, dst = fwd ? o : s;
  float norm;
  if (norm_mode == RGCN_NORM_NONE) {
    norm = 1.0f;
  } else if (norm_mode == RGCN_NORM_INTENDED) {
    norm = 1.0f / (float)(fwd ? indeg[o] : outdeg[s]);
  } else {
    // tf_as_executed: value k of the sorted-row softmax is attached to edge k:
    // the k-th smallest row index is the vertex v with cum[v] <= k < cum[v+1].
    const int32_t* cum = fwd ? cum_in : cum_out;
    const int v = upper_bound_dev(cum, V + 1, e) - 1;
    norm = 1.0f / (float)(fwd ? indeg[v] : outdeg[v]);
  }
  m_src[j] = src;
  m_dst[j] = dst;
  m_norm[j] = norm;
  const int ds = pos[m], ss = pos[fwd ? m + E : m - E];
  m_dslot[j] = ds;
  m_sslot[j] = ss;
  if (slot_arrays) {
    const int rel2 = (int)keyr_s[j];
    d_src[ds] = src; d_rel[ds] = rel2; d_norm[ds] = norm;
    s_dst[ss] = dst; s_rel[ss] = rel2; s_norm[ss] = norm;
  }
}

// ---- edge dropout on the device (reference: code/train.py:233-238) ----------------------------------------------
// The reference draws the message-passing graph of a step as np.random.choice(batch, size=k, replace=False): a
// uniformly random k-subset of the graph batch (exact k, no replacement), and only THOSE edges are fed to
// `graph_edges`, so degrees / normalisation see the kept edges alone while the decoder keeps every batch
// edge as a positive (H9).  Here: every batch edge e gets the 64-bit key (40 random bits of a counter-based generator
// keyed by (seed, e)) << 24 | e -- all keys distinct -- and the k smallest keys are kept, which is a uniform k-subset.
// The k-th smallest key is found by a radix select (one LDS histogram per byte, most significant first, keys recomputed
// on the fly, nothing stored; as soon as the wanted bin holds a single key it is fetched directly), then the kept edges
// are compacted in batch order with one block-wide scan.  One workgroup, one launch, deterministic.
// keep_mask != nullptr: the caller's 0/1 choice instead of the draw (parity tests inject the reference's set).
__device__ __forceinline__ uint64_t dropout_edge_key(uint64_t seed, uint32_t e) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)e + 0x51ED27ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= (z >> 31);
  return (z & 0xFFFFFFFFFF000000ull) | (uint64_t)e;
}

// CACHED: n <= 32 x 1024 -- every thread computes the keys of its <= 32 edges ONCE and keeps them in registers (the
// 64-bit generator is most of the kernel's work; the sweeps below then only compare); otherwise they are recomputed.
template <bool CACHED>
__global__ void __launch_bounds__(1024) k_edge_dropout(const int32_t* __restrict__ batch, int n, int keep, uint64_t seed,
                                                       const uint64_t* __restrict__ seed_offset,
                                                       const uint8_t* __restrict__ keep_mask,
                                                       int32_t* __restrict__ out, int32_t* errflag) {
  if (seed_offset) seed += *seed_offset;      // replayed hipGraphs: replay k draws the subset of (captured seed + k)
  __shared__ uint32_t hist[256];
  __shared__ uint32_t wtot[16];
  __shared__ uint64_t prefix_s, thresh_s;
  __shared__ uint32_t want_s, unique_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // thread t owns the contiguous edges [e_lo, e_hi): the compaction below is then ONE block-wide scan
  const int per = (n + 1023) / 1024;
  const int e_lo = min(n, tid * per), e_hi = min(n, e_lo + per);
  uint64_t thresh = 0;      // keep iff key <= thresh (only used without a mask)
  constexpr int KP = CACHED ? 32 : 1;
  uint64_t kc[KP];
  const bool draw = keep_mask == nullptr && keep > 0 && keep < n;
  if constexpr (CACHED) {
#pragma unroll
    for (int i = 0; i < KP; ++i) kc[i] = (draw && e_lo + i < e_hi) ? dropout_edge_key(seed, (uint32_t)(e_lo + i)) : 0ull;
  }
  // f(key, e) for every edge of this thread
  auto for_each_key = [&](auto&& f) {
    if constexpr (CACHED) {
#pragma unroll
      for (int i = 0; i < KP; ++i)
        if (e_lo + i < e_hi) f(kc[i], e_lo + i);
    } else {
      for (int e = e_lo; e < e_hi; ++e) f(dropout_edge_key(seed, (uint32_t)e), e);
    }
  };
  if (draw) {
    if (tid == 0) { prefix_s = 0; want_s = (uint32_t)keep; unique_s = 0; thresh_s = 0; }
    // radix select of the keep-th smallest key, most significant byte first.  The keys are 40 random bits | the edge
    // index, so after two or three bytes the wanted bin holds ONE key: it is then fetched directly instead of being
    // pinned down byte by byte (8 sweeps -> typically 3).
    for (int pass = 0; pass < 8; ++pass) {
      const int shift = 8 * (7 - pass);
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      // both read HERE, a barrier interval before the thread that owns the selected bin rewrites them (reading want_s
      // beside that write let a later wave's smaller want' select a second bin: round-2 advisor finding)
      const uint64_t prefix = prefix_s;
      const uint32_t want = want_s;
      for_each_key([&](uint64_t key, int) {
        if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(uint32_t)(key >> shift) & 255u], 1u);
      });
      __syncthreads();
      // the bin in which the running count reaches what is still wanted: inclusive scan over the 256 bins by four waves
      uint32_t h = 0, incl = 0;
      if (tid < 256) {
        h = hist[tid];
        incl = h;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const uint32_t t = __shfl_up(incl, off, 64);
          if (lane >= off) incl += t;
        }
        if (lane == 63) wtot[wid] = incl;
      }
      __syncthreads();
      if (tid < 256) {
        for (int w = 0; w < wid; ++w) incl += wtot[w];
        const uint32_t excl = incl - h;
        if ((excl < want && incl >= want) || (tid == 255 && incl < want)) {
          prefix_s = (prefix << 8) | (uint64_t)tid;
          want_s = want - excl;
          unique_s = (h == 1 && pass < 7) ? 1u : 0u;
        }
      }
      __syncthreads();
      if (unique_s) {           // uniform: one key carries the new prefix -- fetch it
        const uint64_t p2 = prefix_s;
        for_each_key([&](uint64_t key, int) {
          if ((key >> shift) == p2) thresh_s = key;
        });
        __syncthreads();
        break;
      }
      if (pass == 7 && tid == 0) thresh_s = prefix_s;
    }
    __syncthreads();
    thresh = thresh_s;
  }
  // stable compaction: per-thread counts over its own edges, one exclusive scan over the 1024 threads, then the writes
  auto kept_at = [&](uint64_t key, int e) -> bool {
    if (keep_mask != nullptr) return keep_mask[e] != 0;
    return keep >= n || (keep > 0 && key <= thresh);
  };
  uint32_t cnt = 0;
  for_each_key([&](uint64_t key, int e) { cnt += kept_at(key, e) ? 1u : 0u; });
  uint32_t incl = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  __syncthreads();             // (wtot is reused)
  if (lane == 63) wtot[wid] = incl;
  __syncthreads();
  uint32_t before = 0, total = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wid) before += wtot[w];
    total += wtot[w];
  }
  int slot = (int)(before + incl - cnt);
  if constexpr (CACHED) {
    // the kept edge ids go to LDS in output order (e < 32,768: 16 bits), then the rows are copied slot by slot:
    // consecutive lanes write consecutive rows and read rows a few apart (a thread copying its own run of rows touched
    // 64 cache lines per wave instruction and was 2/3 of the kernel)
    extern __shared__ uint16_t sel[];
    for_each_key([&](uint64_t key, int e) {
      if (!kept_at(key, e)) return;
      if (slot < keep) sel[slot] = (uint16_t)e;
      ++slot;
    });
    __syncthreads();
    const int ncopy = min((int)total, keep);
    for (int i = tid; i < 3 * ncopy; i += 1024) {
      const int sl = i / 3, f = i - 3 * sl;
      out[i] = batch[3 * (int)sel[sl] + f];
    }
  } else {
    for_each_key([&](uint64_t key, int e) {
      if (!kept_at(key, e)) return;
      if (slot < keep) {
        out[3 * slot] = batch[3 * e];
        out[3 * slot + 1] = batch[3 * e + 1];
        out[3 * slot + 2] = batch[3 * e + 2];
      }
      ++slot;
    });
  }
  if (tid == 0 && (int)total != keep) atomicOr(errflag, 4);     // a mask that does not hold exactly `keep` ones
}

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

}  // namespace

// `share`: a previously allocated set whose read-only / OR-only members (relation owner, error flag)
// this set aliases.
rgcn_status graph_alloc(rgcn_ctx* c, const GraphBufs* share) {
  GraphBufs& g = c->g;
  const size_t V = c->V, R2 = 2 * (size_t)c->R, M = 2 * (size_t)c->cfg.max_edges;
  RGCN_TRY(dalloc(c, &g.triples, 3 * (size_t)c->cfg.max_edges));
  g.counters_bytes = (2 * V + 3) * sizeof(int32_t);
  RGCN_TRY(dalloc(c, &g.counters, 2 * V + 3));
  g.indeg = g.counters;
  g.outdeg = g.counters + V;
  g.nlong = g.counters + 2 * V;
  g.ngiant = g.counters + 2 * V + 1;
  g.giant_cap = (int32_t)(M / kGiantRow + 1);
  g.piece_cap = (int32_t)(2 * (M / kGiantRow) + 2);
  RGCN_TRY(dalloc(c, &g.giant_rows, (size_t)g.giant_cap));
  RGCN_TRY(dalloc(c, &g.giant_first, (size_t)g.giant_cap));
  RGCN_TRY(dalloc(c, &g.giant_cnt, (size_t)g.giant_cap));
  RGCN_TRY(dalloc(c, &g.piece_row, (size_t)g.piece_cap));
  RGCN_TRY(dalloc(c, &g.piece_k, (size_t)g.piece_cap));
  g.long_cap = (int32_t)(M / kLongRow + 1);
  RGCN_TRY(dalloc(c, &g.long_rows, (size_t)g.long_cap));
  RGCN_TRY(dalloc(c, &g.row_ptr, V + 1));
  RGCN_TRY(dalloc(c, &g.rel_ptr, R2 + 1));
  RGCN_TRY(dalloc(c, &g.chunk_ptr, R2 + 1));
  RGCN_TRY(dalloc(c, &g.cum_in, V + 1));
  RGCN_TRY(dalloc(c, &g.cum_out, V + 1));
  RGCN_TRY(dalloc(c, &g.keyv, M));
  RGCN_TRY(dalloc(c, &g.keyv_s, M));
  RGCN_TRY(dalloc(c, &g.keyr, M));
  RGCN_TRY(dalloc(c, &g.keyr_s, M));
  RGCN_TRY(dalloc(c, &g.valv, M));
  RGCN_TRY(dalloc(c, &g.permv, M));
  RGCN_TRY(dalloc(c, &g.valr, M));
  RGCN_TRY(dalloc(c, &g.permr, M));
  RGCN_TRY(dalloc(c, &g.pos, M));
  RGCN_TRY(dalloc(c, &g.m_src, M));
  RGCN_TRY(dalloc(c, &g.m_dst, M));
  RGCN_TRY(dalloc(c, &g.m_dslot, M));
  RGCN_TRY(dalloc(c, &g.m_sslot, M));
  RGCN_TRY(dalloc(c, &g.m_norm, M));
  {      // slot-ordered message lists (row-major gathers): the basis kind and the destination-major block layer
    RGCN_TRY(dalloc(c, &g.d_src, M));
    RGCN_TRY(dalloc(c, &g.d_rel, M));
    RGCN_TRY(dalloc(c, &g.d_norm, M));
    RGCN_TRY(dalloc(c, &g.s_dst, M));
    RGCN_TRY(dalloc(c, &g.s_rel, M));
    RGCN_TRY(dalloc(c, &g.s_norm, M));
  }
  if (c->kind == RGCN_KIND_BASIS) {
    RGCN_TRY(dalloc(c, &g.has_dir, 2 * V));
    RGCN_TRY(dalloc(c, &g.unit_ptr, 2 * (V + 1)));
    RGCN_TRY(dalloc(c, &g.unit_rows, 2 * V));
    RGCN_HIP(c, hipMemsetAsync(g.unit_ptr, 0, sizeof(int32_t) * 2 * (V + 1), c->stream));
  }
  if (c->kind == RGCN_KIND_BLOCK) {
    RGCN_TRY(dalloc(c, &g.row_key, V));
    RGCN_TRY(dalloc(c, &g.row_key_s, V));
    RGCN_TRY(dalloc(c, &g.row_order, V));
    RGCN_TRY(dalloc(c, &g.row_tab, sort_table_elems(V)));
  }
  if (share) {
    g.owner = share->owner;
    g.errflag = share->errflag;
  } else {
    RGCN_TRY(dalloc(c, &g.owner, (size_t)c->R));
    RGCN_TRY(dalloc(c, &g.errflag, 1));
    RGCN_HIP(c, hipMemsetAsync(g.owner, 0, sizeof(int32_t) * (size_t)(c->R ? c->R : 1), c->stream));
    RGCN_HIP(c, hipMemsetAsync(g.errflag, 0, sizeof(int32_t), c->stream));
  }
  RGCN_HIP(c, hipEventCreateWithFlags(&g.ev_ready, order_event_flags(c)));
  RGCN_HIP(c, hipEventCreateWithFlags(&g.ev_free, order_event_flags(c)));
  RGCN_HIP(c, hipMemsetAsync(g.row_ptr, 0, sizeof(int32_t) * (V + 1), c->stream));
  RGCN_HIP(c, hipMemsetAsync(g.rel_ptr, 0, sizeof(int32_t) * (R2 + 1), c->stream));
  RGCN_HIP(c, hipMemsetAsync(g.chunk_ptr, 0, sizeof(int32_t) * (R2 + 1), c->stream));
  RGCN_TRY(dalloc(c, &g.keyv_t, M));
  RGCN_TRY(dalloc(c, &g.keyr_t, M));
  RGCN_TRY(dalloc(c, &g.tablev, sort_table_elems(M)));
  RGCN_TRY(dalloc(c, &g.tabler, sort_table_elems(M)));
  return RGCN_OK;
}

static void graph_free_one(GraphBufs& g, bool owns_shared) {
  if (!owns_shared) { g.owner = nullptr; g.errflag = nullptr; }
  if (g.ev_ready) (void)hipEventDestroy(g.ev_ready);
  if (g.ev_free) (void)hipEventDestroy(g.ev_free);
  void* ptrs[] = {g.giant_rows, g.giant_first, g.giant_cnt, g.piece_row, g.piece_k, g.long_rows, g.triples, g.counters, g.row_ptr, g.rel_ptr, g.chunk_ptr, g.cum_in, g.cum_out,
                  g.keyv, g.keyv_s, g.keyr, g.keyr_s, g.valv, g.permv, g.valr, g.permr, g.pos,
                  g.m_src, g.m_dst, g.m_dslot, g.m_sslot, g.m_norm, g.d_src, g.d_rel, g.d_norm, g.s_dst, g.s_rel,
                  g.s_norm, g.owner, g.errflag, g.keyv_t, g.keyr_t, g.tablev, g.tabler, g.row_key, g.row_key_s, g.row_order,
                  g.row_tab, g.has_dir, g.unit_ptr, g.unit_rows};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  g = GraphBufs();
}

void graph_free(rgcn_ctx* c) {
  graph_free_one(c->g_alt, false);
  graph_free_one(c->g, true);
}

rgcn_status graph_build(rgcn_ctx* c, const int32_t* tri, int64_t E64) {
  GraphBufs& g = c->g;
  const int E = (int)E64, V = c->V, R = c->R;
  const int M = 2 * E;
  g.cur = tri;
  g.E = E;
  // messages per relation chunk: 48 at minibatch scale, growing with THIS graph's size so that a full graph
  // does not cut a popular relation into thousands of chunks (a context serves 15,000-edge training steps and
  // the 272,115-edge evaluation pass alike); the capacity-scaled c->chunk is the upper bound the slabs are sized for
  g.chunk = std::min(c->chunk, 48 * (int)std::max<int64_t>(1, ((int64_t)M + 65535) / 65536));
  // giant-row cut: full-graph scale, block kind, one GPU (the basis gathers and the sharded finish walk the long-row
  // list themselves)
  g.giant_on = c->kind == RGCN_KIND_BLOCK && c->world == 1 && M > 65536;
  g.ready = false;
  g.pf_valid = false;
  g.units_host = -1;
  c->fwd_done = false;
  // the vertex key carries the directed relation when (V + 1) 2R fits the sort's key range (every dataset of the reference;
  // otherwise rows keep incidence order: the layer kernels find fewer runs, nothing else changes)
  const uint32_t vmul = (uint64_t)(V + 1) * (2 * (uint64_t)R) < (1ull << 31) ? (uint32_t)(2 * R) : 1u;
  const bool zero_in_keys = c->world == 1 && E > 0;      // (sharded: the degree counters are atomically added to)
  if (!zero_in_keys) RGCN_HIP(c, hipMemsetAsync(g.counters, 0, g.counters_bytes, c->stream));
  const int T = 256;
  if (E > 0) {
    {
      ProfScope ps(c, "prep_keys", 12.0 * E + 16.0 * M, 0);
      hipLaunchKernelGGL(k_keys, dim3((M + T - 1) / T), dim3(T), 0, c->stream, tri, E, V, R, g.owner,
                         c->rank, g.keyv, g.keyr, c->world > 1 ? 1 : 0, g.indeg, g.outdeg, g.errflag,
                         zero_in_keys ? g.nlong : (int32_t*)nullptr, vmul);
    }
    // incidences by vertex (-> permv, pos) and messages by directed relation (-> permr), in the same launches
    SortSpec sp[2];
    sp[0] = SortSpec{g.keyv, g.keyv_s, g.permv, g.keyv_t, g.valv, g.pos, g.tablev, (int64_t)M, (uint32_t)V * vmul};
    sp[1] = SortSpec{g.keyr, g.keyr_s, g.permr, g.keyr_t, g.valr, nullptr, g.tabler, (int64_t)M, (uint32_t)(2 * R)};
    RGCN_TRY(sort_pairs(c, "prep_sort", 2, sp));
  }
  {
    ProfScope ps(c, "prep_ptrs", 8.0 * (V + 2 * R) + 4.0 * M, 0);
    hipLaunchKernelGGL(k_ptrs, dim3(1 + (V + 1 + 1023) / 1024), dim3(1024), 0, c->stream, g.keyv_s, g.keyr_s,
                       g.permv, E, c->world > 1 ? 0 : 1, g.indeg, g.outdeg, M, V, 2 * R, g.chunk, g.row_ptr, g.long_rows,
                       g.nlong, g.long_cap,
                       g.giant_on ? kGiantRow : 0x7fffffff, g.giant_rows, g.giant_first, g.giant_cnt, g.piece_row,
                       g.piece_k, g.ngiant, g.giant_cap, g.piece_cap, g.rel_ptr, g.chunk_ptr, g.row_key, g.has_dir, vmul);
  }
  if (g.row_key != nullptr) {
    // rows by descending length (stable, one 8-bit pass of the library's radix sort over V keys <= 33)
    SortSpec sp{g.row_key, g.row_key_s, g.row_order, g.row_key_s, g.row_order, nullptr, g.row_tab, (int64_t)V,
                (uint32_t)(kLongRow + 1)};
    RGCN_TRY(sort_pairs(c, "prep_row_order", 1, &sp));
  }
  {
    // one launch, one workgroup per scan: the cumulative degrees of the as-executed normalisation, and the
    // basis kind's (row, direction) units -- index of every unit and the ascending list of their rows
    ScanJobs jobs;
    int nj = 0;
    if (c->cfg.norm_mode == RGCN_NORM_TF_AS_EXECUTED) {
      jobs.j[nj++] = {g.indeg, g.cum_in, V, 0, nullptr};
      jobs.j[nj++] = {g.outdeg, g.cum_out, V, 0, nullptr};
    }
    if (g.has_dir != nullptr) {
      jobs.j[nj++] = {g.has_dir, g.unit_ptr, V, 0, g.unit_rows};
      jobs.j[nj++] = {g.has_dir + V, g.unit_ptr + (V + 1), V, 0, g.unit_rows + V};
    }
    for (int k = nj; k < 5; ++k) jobs.j[k] = jobs.j[0];
    if (nj > 0) {
      ProfScope ps(c, "prep_scan", 8.0 * nj * V, 0);
      hipLaunchKernelGGL(k_exscan, dim3(nj), dim3(kScanThreads), 0, c->stream, jobs);
    }
  }
  if (E > 0) {
    ProfScope ps(c, "prep_build_msgs", 12.0 * E + 36.0 * M, 0);
    hipLaunchKernelGGL(k_build_msgs, dim3((M + T - 1) / T), dim3(T), 0, c->stream, tri, E, V, R,
                       c->cfg.norm_mode, g.permr, g.rel_ptr, g.pos, g.indeg, g.outdeg, g.cum_in,
                       g.cum_out, g.m_src, g.m_dst, g.m_dslot, g.m_sslot, g.m_norm, g.keyr_s,
                       g.d_src != nullptr ? 1 : 0, g.d_src, g.d_rel, g.d_norm, g.s_dst, g.s_rel, g.s_norm);
  }
  RGCN_HIP(c, hipGetLastError());
  g.ready = true;
  return RGCN_OK;
}

// graph = the kept edges of `batch` under edge dropout (k_edge_dropout), compacted into the set's own triple buffer
rgcn_status graph_build_dropout(rgcn_ctx* c, const int32_t* batch, int64_t n, int64_t keep, uint64_t seed,
                                const uint8_t* keep_mask) {
  GraphBufs& g = c->g;
  if (n > 0) {
    ProfScope ps(c, "prep_edge_dropout", 12.0 * n + 12.0 * keep, 0);
    const uint64_t* seed_off = c->capturing ? c->replay_counter : nullptr;
    if (n <= 32 * 1024) {
      if (!c->dropout_lds_configured) {    // up to 64 KB of kept-edge ids beside the static 1.2 KB; per device, so per context
        RGCN_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_edge_dropout<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32 * 1024));
        c->dropout_lds_configured = true;
      }
      hipLaunchKernelGGL(k_edge_dropout<true>, dim3(1), dim3(1024), 2 * (size_t)std::max<int64_t>(keep, 1), c->stream,
                         batch, (int)n, (int)keep, seed, seed_off, keep_mask, g.triples, g.errflag);
    }
    else
      hipLaunchKernelGGL(k_edge_dropout<false>, dim3(1), dim3(1024), 0, c->stream, batch, (int)n, (int)keep, seed,
                         seed_off, keep_mask, g.triples, g.errflag);
    RGCN_HIP(c, hipGetLastError());
  }
  return graph_build(c, g.triples, keep);
}

}  // namespace rgcn
