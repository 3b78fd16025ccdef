// The reference's neighbourhood edge sampler (sample_edge_neighborhood, code/train.py:133-139,161-198) ON THE DEVICE,
// as a parallel algorithm with the same distribution.
//
// The reference grows the graph batch one edge at a time: a vertex is drawn with probability proportional to
// (free incident edge ends) x (already touched), then one of its free edge ends uniformly; when no touched vertex has a
// free end (always at the first draw) a vertex with free ends is drawn uniformly.  Every pick depends on all earlier
// ones -- 30,000 dependent weighted draws per batch, which one GPU workgroup walks in ~60-90 ms (4-5 dependent memory
// round trips per pick) against 3.6 ms on a CPU core.  But the chain is the JUMP CHAIN of a continuous-time process:
// "vertex ~ free ends x touched, then a free end uniformly" = every free edge END at a touched vertex is equally likely
// = every such end carries an independent Exp(1) clock that starts when its vertex is first touched, and the next pick
// is the end whose clock rings first (memorylessness).  That is first-passage percolation: with i.i.d. X[v,e] ~ Exp(1)
// per edge end,
//     touched-time d(v)   = shortest-path distance from the start vertex, crossing edge e out of v costs X[v,e],
//     pick-time   t(e)    = min( d(s_e) + X[s_e,e], d(o_e) + X[o_e,e] )           (a self loop has two ends at one vertex),
// and the reference's pick order is the order of t(e).  The restart rule -- a uniformly drawn vertex among those that
// still have free ends, whenever the touched part is exhausted -- visits connected components in the order of a uniform
// random permutation of the vertices: give every vertex with edges an independent uniform priority U(v); a component's
// start is its vertex of least priority, components are visited in the order of that least priority.  So the batch is
//     every edge of the components visited in full, plus the r edges of least pick time of the component the budget
//     runs out in (for a training graph with one giant component: the sample_size edges of least pick time).
// Division of labour: components are a property of the graph, found once on the host when the graph is handed over
// (rgcn_neighborhood_reserve: union-find, adjacency CSR); per draw the host evaluates the V vertex priorities (a
// counter-based hash: ~40 us), which settles the component order, the boundary component, its start vertex and r; the
// device does the rest -- pull-style Bellman-Ford sweeps over the adjacency (one wavefront per 256-entry segment of a
// vertex's list: d(v) = min over incident edges of d(other) + X[other,e]), pick times, a radix select of the r-th
// smallest (12-bit digits, LDS histograms), a stable compaction -- a function of the seed alone, one replayed hipGraph,
// a few hundred microseconds on the prefetch stream beside the running train step, no batch built on the host, nothing
// uploaded but the 48-byte parameter block (and a byte per component when more than one component takes part).
// tests/test_gpu_sampler.py holds the distribution of the drawn sets to oracle.sample_edge_neighborhood (the
// reference's loop, step for step) on graphs with several components, self loops and parallel edges.
//
// The sweeps stop by themselves: launch i returns at once when launch i-1 moved nothing; the budget (sized at
// rgcn_neighborhood_reserve from the graph's diameter: 48 launches = 104 sweeps for a small world -- the 272,115-edge
// training graph settles in 23-26 --, up to 1,024 launches = 4,008 sweeps for a graph thousands of hops across) covers the hops
// of the deepest shortest path several times over, and a graph that still exhausts it raises the context's error flag
// instead of returning a wrong batch (train.py --host-sampler is the way out for such a graph).
#include <algorithm>
#include <cstdlib>
#include <numeric>

#include "rgcn_internal.h"

namespace rgcn {

namespace {

constexpr int kSweepLaunches = 48, kMaxSweepLaunches = 1024, kSegment = 256;   // launches: floor and ceiling of the budget
// Sweeps per launch.  What a launch's wavefronts store reaches the others' plain loads at the NEXT launch, so a second
// sweep inside a launch mostly repeats the first (measured: two sweeps per launch settled the training graph in 19-20
// launches, ONE sweep per launch in 23-26 -- the same hops, half the work): one sweep per launch while distances still
// move everywhere, more per launch (coherent loads from the second on) only in the tail that deep graphs need.
__host__ __device__ constexpr int sweeps_of_launch(int it) { return it < 24 ? 1 : (it < 32 ? 2 : 4); }
constexpr int64_t sweeps_of_budget(int launches) {
  return (launches < 24 ? launches : 24) + 2 * (launches < 24 ? 0 : (launches < 32 ? launches - 24 : 8)) +
         4 * (int64_t)(launches < 32 ? 0 : launches - 32);
}
// "something moved" flags: kFlagSlots per launch, each in its own 128-byte line -- sixteen thousand wavefronts storing
// to ONE address serialise in the L2 (it was 30 of the 43 us of an early sweep)
constexpr int kFlagSlots = 16, kFlagStride = 32;
constexpr uint32_t kFar = 0x7f7f7f7fu;          // 3.39e38: "not touched" (the byte pattern of a memset)
constexpr int kDigitBits = 12, kBins = 1 << kDigitBits, kPasses = 6;      // 64-bit key: 5 x 12 + 4 bits

__host__ __device__ inline uint64_t mix64(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 0x632BE5ABull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// priority of vertex v: 40 random bits | the vertex id (all distinct); evaluated on the host
inline uint64_t vertex_priority(uint64_t seed, int v) {
  return (mix64(seed ^ 0x5bf03635aa2e1d47ull, (uint64_t)v) & 0xFFFFFFFFFF000000ull) | (uint64_t)(uint32_t)v;
}
// Exp(1) clock of edge end `end` (2 e: the subject's end of edge e, 2 e + 1: the object's)
__device__ __forceinline__ float end_clock(uint64_t seed, uint64_t end) {
  const uint32_t r = (uint32_t)(mix64(seed, end) >> 40);                 // 24 random bits
  return -logf(((float)r + 0.5f) * (1.0f / 16777216.0f));
}

// What one draw is a function of: written by the host into a 64-byte block of device memory, read by every kernel --
// the kernels' arguments never change, so the ~75 launches of a draw are ONE replayed hipGraph (launching them one by
// one costs the host 0.3 ms a draw, more than the device spends).
struct DrawParams {
  uint64_t seed;
  unsigned long long want;       // edges to take from the boundary component
  int32_t* out;                  // [k,3] the batch
  int32_t start, boundary;       // start vertex, boundary component
  int32_t k, use_state;          // sample size; 1: components taken in full are marked in comp_state
};

// dist <- far (start vertex: 0), flags <- 0
__global__ void __launch_bounds__(256) k_nbr_init(const DrawParams* __restrict__ p, uint32_t* __restrict__ dist, int V,
                                                  int32_t* __restrict__ changed, int nflags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < V) dist[i] = i == p->start ? 0u : kFar;
  if (i < nflags) changed[i] = 0;
}

// d(v) <- min over v's incident edges of d(other) + X[other end].  One wavefront per SEGMENT of at most kSegment
// adjacency entries of one vertex (a hub's ten thousand entries walked by one wavefront took 70 us a sweep), its minimum
// written with a plain store when the segment is the vertex's whole list (one writer), merged with atomicMin on the
// non-negative float's bits when a hub's list has several (a minimum does not depend on the order it is taken in).
// Device-scope atomics on a 58 KB array written from all eight XCDs cost ~1 us each (the line changes L2): an atomicMin
// per moved vertex made the early sweeps, where most vertices move, 40 us each.  Plain stores are seen by the next
// launch at the latest, which is all the stop test needs: a launch that moved nothing has read, at its first sweep,
// values no one was writing -- the fixed point.
// A lane keeps its (at most four) entries and their clocks in registers for the launch's sweeps; relaxations are
// monotone and idempotent, so sweeps need no barrier between them, only the stop test does.  The first sweep of a
// launch reads d with plain loads (whatever earlier launches wrote is visible), later ones with coherent loads (four
// independent ones per lane: a coherent load that misses the local L2 costs microseconds while other XCDs write d).
__global__ void __launch_bounds__(256) k_nbr_sweep(const DrawParams* __restrict__ p, const int32_t* __restrict__ seg_v,
                                                   const int32_t* __restrict__ seg_beg, const int32_t* __restrict__ seg_end,
                                                   int nseg, const int32_t* __restrict__ adj_other,
                                                   const int32_t* __restrict__ adj_end, uint32_t* dist, int32_t* changed, int it,
                                                   int nsweeps) {
  static_assert(kSegment == 256, "four entries per lane");
  const int lane = threadIdx.x & 63;
  if (it > 0) {
    const int32_t f = lane < kFlagSlots ? changed[((it - 1) * kFlagSlots + lane) * kFlagStride] : 0;
    if (__ballot(f != 0) == 0ull) return;
  }
  const int g = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (g >= nseg) return;
  const int sv = seg_v[g], beg = seg_beg[g], end = seg_end[g];
  const bool shared = sv < 0;                    // one of several segments of a hub's list (~v): merge with atomicMin
  const int v = shared ? ~sv : sv;
  const uint64_t seed = p->seed;
  int other[4];
  float clock[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = beg + lane + 64 * i;
    other[i] = j < end ? adj_other[j] : -1;
    clock[i] = j < end ? end_clock(seed, (uint64_t)(uint32_t)adj_end[j]) : 0.f;
  }
  uint32_t cur = dist[v];                        // a vertex with ONE segment has one writer: this wavefront
  bool moved = false;
  for (int sweep = 0; sweep < nsweeps; ++sweep) {
    uint32_t du[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      du[i] = other[i] < 0 ? kFar : sweep == 0 ? dist[other[i]] : __atomic_load_n(&dist[other[i]], __ATOMIC_RELAXED);
    uint32_t best = kFar;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (du[i] != kFar) best = min(best, __float_as_uint(__uint_as_float(du[i]) + clock[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, off, 64));
    if (best < cur) {
      if (!shared) {
        if (lane == 0) dist[v] = best;
        moved = true;
      } else if (lane == 0 && atomicMin(&dist[v], best) > best) {
        moved = true;
      }
      cur = best;
    }
  }
  if (lane == 0 && moved) {
    int32_t* flag = changed + (it * kFlagSlots + (g & (kFlagSlots - 1))) * kFlagStride;
    if (__atomic_load_n(flag, __ATOMIC_RELAXED) == 0) __atomic_store_n(flag, 1, __ATOMIC_RELAXED);
  }
}

// the sweeps ran out of launches while still moving: refuse (flag 8) rather than return a wrong batch
__global__ void k_nbr_check(const int32_t* __restrict__ changed, int last, int32_t* errflag) {
  if (blockIdx.x == 0 && threadIdx.x < kFlagSlots && changed[(last * kFlagSlots + threadIdx.x) * kFlagStride] != 0)
    atomicOr(errflag, 8);
}

// per edge of the boundary component: (pick time | edge id); every other edge: the largest key
__global__ void k_nbr_keys(const DrawParams* __restrict__ p, const int32_t* __restrict__ tri, int n,
                           const int32_t* __restrict__ comp, const uint32_t* __restrict__ dist,
                           unsigned long long* __restrict__ tkey) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int s = tri[3 * e], o = tri[3 * e + 2];
  const uint64_t seed = p->seed;
  unsigned long long key = ~0ull;
  if (comp[s] == p->boundary) {
    const float ts = __uint_as_float(dist[s]) + end_clock(seed, 2ull * (uint64_t)e);
    const float to = __uint_as_float(dist[o]) + end_clock(seed, 2ull * (uint64_t)e + 1ull);
    key = ((unsigned long long)__float_as_uint(fminf(ts, to)) << 32) | (unsigned long long)(uint32_t)e;
  }
  tkey[e] = key;
}

// radix select of the r-th smallest key, 12-bit digits from the top; state: prefix found so far, rank still wanted
struct SelState {
  unsigned long long prefix, want;
};
__device__ __forceinline__ int digit_shift(int pass) { return pass < 5 ? 52 - kDigitBits * pass : 0; }
__global__ void k_nbr_set_want(const DrawParams* __restrict__ p, SelState* __restrict__ s) {
  s->prefix = 0ull;
  s->want = p->want;
}
__global__ void __launch_bounds__(1024) k_nbr_hist(int pass, const unsigned long long* __restrict__ tkey, int n,
                                                   const SelState* __restrict__ state, uint32_t* __restrict__ hist) {
  __shared__ uint32_t lh[kBins];
  for (int i = threadIdx.x; i < kBins; i += 1024) lh[i] = 0;
  __syncthreads();
  const int e = blockIdx.x * 1024 + threadIdx.x;
  if (e < n) {
    const unsigned long long k = tkey[e], pre = state->prefix;
    const int sh = digit_shift(pass);
    const int width = pass < 5 ? kDigitBits : 4;
    const bool match = pass == 0 || (k >> (sh + width)) == (pre >> (sh + width));
    if (match) atomicAdd(&lh[(uint32_t)(k >> sh) & (uint32_t)((1 << width) - 1)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kBins; i += 1024)
    if (lh[i] != 0) atomicAdd(&hist[i], lh[i]);
}
// one workgroup: the digit in which the running count reaches `want`; clears the histogram for the next pass
__global__ void __launch_bounds__(1024) k_nbr_scan(int pass, uint32_t* __restrict__ hist, SelState* __restrict__ state) {
  __shared__ unsigned long long wsum[16];
  __shared__ uint32_t found_digit;
  __shared__ unsigned long long found_before;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint4 h4 = reinterpret_cast<const uint4*>(hist)[tid];
  const uint32_t h[4] = {h4.x, h4.y, h4.z, h4.w};
  const unsigned long long mine = (unsigned long long)h[0] + h[1] + h[2] + h[3];
  unsigned long long incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  unsigned long long before = incl - mine;
  for (int w = 0; w < wid; ++w) before += wsum[w];
  const unsigned long long want = state->want;
  if (before < want && before + mine >= want) {
    unsigned long long run = before;
    for (int i = 0; i < 4; ++i) {
      if (run + h[i] >= want) { found_digit = (uint32_t)(tid * 4 + i); found_before = run; break; }
      run += h[i];
    }
  }
  __syncthreads();
  if (tid == 0) {
    state->prefix |= (unsigned long long)found_digit << digit_shift(pass);
    state->want = want - found_before;
  }
  reinterpret_cast<uint4*>(hist)[tid] = make_uint4(0u, 0u, 0u, 0u);
}

// in the batch: the boundary component's edges up to the threshold key, and every edge of a component taken in full
__device__ __forceinline__ bool nbr_included(int e, const int32_t* __restrict__ tri, const unsigned long long* __restrict__ tkey,
                                             unsigned long long thresh, const int32_t* __restrict__ comp,
                                             const uint8_t* __restrict__ comp_state, bool use_state) {
  if (tkey[e] <= thresh) return true;
  return use_state && comp_state[comp[tri[3 * e]]] != 0;
}
constexpr int kCompactBlock = 1024;
__global__ void __launch_bounds__(kCompactBlock) k_nbr_count(const DrawParams* __restrict__ p, const int32_t* __restrict__ tri,
                                                            const unsigned long long* __restrict__ tkey, int n,
                                                            const SelState* __restrict__ state,
                                                            const int32_t* __restrict__ comp,
                                                            const uint8_t* __restrict__ comp_state,
                                                            uint32_t* __restrict__ bcnt) {
  __shared__ uint32_t wcnt[kCompactBlock / 64];
  const int e = blockIdx.x * kCompactBlock + threadIdx.x;
  const bool inc = e < n && nbr_included(e, tri, tkey, state->prefix, comp, comp_state, p->use_state != 0);
  const unsigned long long vote = __ballot(inc);
  if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = (uint32_t)__popcll(vote);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < kCompactBlock / 64; ++w) t += wcnt[w];
    bcnt[blockIdx.x] = t;
  }
}
// exclusive scan of the block counts (one workgroup, any number of blocks); the total must be the sample size
__global__ void __launch_bounds__(1024) k_nbr_offsets(const DrawParams* __restrict__ p, uint32_t* __restrict__ bcnt, int nblocks,
                                                      int32_t* errflag) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {
    const int b = base + tid;
    const uint32_t v = b < nblocks ? bcnt[b] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
      if (w < wid) wbase += wsum[w];
      total += wsum[w];
    }
    const uint32_t carry = carry_s;
    if (b < nblocks) bcnt[b] = carry + wbase + incl - v;
    __syncthreads();
    if (tid == 0) carry_s = carry + total;
    __syncthreads();
  }
  if (tid == 0 && (int)carry_s != p->k) atomicOr(errflag, 32);   // (not a diameter problem: its own flag and message)
}
// the chosen edges' rows, in edge order, to the caller's batch buffer
__global__ void __launch_bounds__(kCompactBlock) k_nbr_write(const DrawParams* __restrict__ p, const int32_t* __restrict__ tri,
                                                            const unsigned long long* __restrict__ tkey, int n,
                                                            const SelState* __restrict__ state,
                                                            const int32_t* __restrict__ comp,
                                                            const uint8_t* __restrict__ comp_state,
                                                            const uint32_t* __restrict__ boff) {
  __shared__ uint32_t wcnt[kCompactBlock / 64];
  const int e = blockIdx.x * kCompactBlock + threadIdx.x;
  const bool inc = e < n && nbr_included(e, tri, tkey, state->prefix, comp, comp_state, p->use_state != 0);
  const unsigned long long vote = __ballot(inc);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) wcnt[wid] = (uint32_t)__popcll(vote);
  __syncthreads();
  if (!inc) return;
  uint32_t slot = boff[blockIdx.x] + (uint32_t)__popcll(vote & ((1ull << lane) - 1ull));
  for (int w = 0; w < wid; ++w) slot += wcnt[w];
  if ((int)slot >= p->k) return;
  int32_t* __restrict__ out = p->out;
  out[3 * slot] = tri[3 * e];
  out[3 * slot + 1] = tri[3 * e + 1];
  out[3 * slot + 2] = tri[3 * e + 2];
}

template <class T>
rgcn_status dalloc(rgcn_ctx* c, T** p, size_t n) {
  RGCN_HIP(c, hipMalloc((void**)p, (n ? n : 1) * sizeof(T)));
  return RGCN_OK;
}

int find_root(std::vector<int32_t>& parent, int x) {
  while (parent[x] != x) {
    parent[x] = parent[parent[x]];
    x = parent[x];
  }
  return x;
}

}  // namespace

void neighborhood_free(rgcn_ctx* c) {
  NeighborhoodBufs& q = c->nbr;
  void* ptrs[] = {q.triples, q.seg_v, q.seg_beg, q.seg_end, q.adj_other, q.adj_end, q.comp, q.comp_state, q.dist, q.tkey, q.hist, q.state,
                  q.changed, q.bcnt, q.params};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (q.draw_exec) (void)hipGraphExecDestroy(q.draw_exec);
  if (q.draw_graph) (void)hipGraphDestroy(q.draw_graph);
  if (q.capture_stream) (void)hipStreamDestroy(q.capture_stream);
  if (q.ev_draw) (void)hipEventDestroy(q.ev_draw);
  q = NeighborhoodBufs();
}

// The training graph moves to the device once (the reference builds its adjacency lists once, train.py:133-139):
// triples, adjacency CSR, connected components (ids validated by the caller).
rgcn_status neighborhood_reserve(rgcn_ctx* c, const int32_t* tri, int64_t n64) {
  neighborhood_free(c);
  NeighborhoodBufs& q = c->nbr;
  const int n = (int)n64, V = c->V;
  q.n = n64;
  std::vector<int32_t> ptr((size_t)V + 1, 0), parent((size_t)V);
  std::iota(parent.begin(), parent.end(), 0);
  std::vector<uint8_t> has((size_t)V, 0);
  for (int e = 0; e < n; ++e) {
    const int s = tri[3 * e], o = tri[3 * e + 2];
    has[s] = has[o] = 1;
    if (s == o) continue;
    ptr[s + 1]++; ptr[o + 1]++;
    const int a = find_root(parent, s), b = find_root(parent, o);
    if (a != b) parent[std::max(a, b)] = std::min(a, b);
  }
  for (int v = 0; v < V; ++v) ptr[v + 1] += ptr[v];
  std::vector<int32_t> other((size_t)ptr[V]), endid((size_t)ptr[V]), fill(ptr.begin(), ptr.end() - 1);
  for (int e = 0; e < n; ++e) {
    const int s = tri[3 * e], o = tri[3 * e + 2];
    if (s == o) continue;
    other[fill[s]] = o; endid[fill[s]++] = 2 * e + 1;      // relaxing s adds the clock of the end at o
    other[fill[o]] = s; endid[fill[o]++] = 2 * e;
  }
  std::vector<int32_t> seg_v, seg_beg, seg_end;
  for (int v = 0; v < V; ++v)
    for (int b0 = ptr[v]; b0 < ptr[v + 1]; b0 += kSegment) {
      seg_v.push_back(ptr[v + 1] - ptr[v] > kSegment ? ~v : v);
      seg_beg.push_back(b0);
      seg_end.push_back(std::min(ptr[v + 1], b0 + kSegment));
    }
  q.nseg = (int32_t)seg_v.size();
  q.comp_h.assign((size_t)V, -1);
  std::vector<int32_t> root_comp((size_t)V, -1);
  q.ncomp = 0;
  for (int v = 0; v < V; ++v) {
    if (!has[v]) continue;
    const int r = find_root(parent, v);
    if (root_comp[r] < 0) root_comp[r] = q.ncomp++;
    q.comp_h[v] = root_comp[r];
  }
  q.comp_edges_h.assign((size_t)q.ncomp, 0);
  for (int e = 0; e < n; ++e) q.comp_edges_h[q.comp_h[tri[3 * e]]]++;
  q.comp_state_h.assign((size_t)q.ncomp, 0);
  // The sweep budget follows the graph: a shortest path of the percolation has a few times as many hops as the graph
  // distance it spans (3.5 times on the training graph), so the budget is 12 x (largest component diameter, by a double
  // breadth-first sweep per component) + 32 sweeps, at least kSweepLaunches launches, at most kMaxSweepLaunches.
  {
    std::vector<int32_t> depth((size_t)V, -1), queue;
    queue.reserve((size_t)V);
    auto bfs = [&](int src, int& far) {
      queue.clear();
      queue.push_back(src);
      depth[src] = 0;
      far = src;
      for (size_t h = 0; h < queue.size(); ++h) {
        const int v = queue[h];
        for (int j = ptr[v]; j < ptr[v + 1]; ++j) {
          const int u = other[j];
          if (depth[u] < 0) { depth[u] = depth[v] + 1; queue.push_back(u); far = u; }
        }
      }
      const int ecc = depth[far];
      for (int v : queue) depth[v] = -1;
      return ecc;
    };
    std::vector<uint8_t> done((size_t)q.ncomp, 0);
    int diameter = 0;
    for (int v = 0; v < V; ++v) {
      const int cid = q.comp_h[v];
      if (cid < 0 || done[cid]) continue;
      done[cid] = 1;
      int far = v, far2 = v;
      (void)bfs(v, far);
      diameter = std::max(diameter, bfs(far, far2));
    }
    const int64_t sweeps = 12 * (int64_t)diameter + 32;
    int launches = kSweepLaunches;
    while (launches < kMaxSweepLaunches && sweeps_of_budget(launches) < sweeps) ++launches;
    q.launches = launches;
  }
  const size_t nb = (size_t)((n + kCompactBlock - 1) / kCompactBlock) + 1;
  RGCN_TRY(dalloc(c, &q.triples, 3 * (size_t)n));
  RGCN_TRY(dalloc(c, &q.seg_v, seg_v.size()));
  RGCN_TRY(dalloc(c, &q.seg_beg, seg_v.size()));
  RGCN_TRY(dalloc(c, &q.seg_end, seg_v.size()));
  RGCN_TRY(dalloc(c, &q.adj_other, other.size()));
  RGCN_TRY(dalloc(c, &q.adj_end, endid.size()));
  RGCN_TRY(dalloc(c, &q.comp, (size_t)V));
  RGCN_TRY(dalloc(c, &q.comp_state, (size_t)q.ncomp));
  RGCN_TRY(dalloc(c, &q.dist, (size_t)V));
  RGCN_TRY(dalloc(c, &q.tkey, (size_t)n));
  RGCN_TRY(dalloc(c, &q.hist, (size_t)kBins));
  RGCN_TRY(dalloc(c, &q.state, (size_t)2));
  RGCN_TRY(dalloc(c, &q.changed, (size_t)q.launches * kFlagSlots * kFlagStride));
  RGCN_TRY(dalloc(c, &q.bcnt, nb));
  RGCN_HIP(c, hipMalloc(&q.params, sizeof(DrawParams)));
  hipStream_t st = c->stream;
  RGCN_HIP(c, hipMemcpyAsync(q.triples, tri, sizeof(int32_t) * 3 * (size_t)n, hipMemcpyHostToDevice, st));
  if (!other.empty()) {
    RGCN_HIP(c, hipMemcpyAsync(q.seg_v, seg_v.data(), sizeof(int32_t) * seg_v.size(), hipMemcpyHostToDevice, st));
    RGCN_HIP(c, hipMemcpyAsync(q.seg_beg, seg_beg.data(), sizeof(int32_t) * seg_v.size(), hipMemcpyHostToDevice, st));
    RGCN_HIP(c, hipMemcpyAsync(q.seg_end, seg_end.data(), sizeof(int32_t) * seg_v.size(), hipMemcpyHostToDevice, st));
    RGCN_HIP(c, hipMemcpyAsync(q.adj_other, other.data(), sizeof(int32_t) * other.size(), hipMemcpyHostToDevice, st));
    RGCN_HIP(c, hipMemcpyAsync(q.adj_end, endid.data(), sizeof(int32_t) * endid.size(), hipMemcpyHostToDevice, st));
  }
  RGCN_HIP(c, hipMemcpyAsync(q.comp, q.comp_h.data(), sizeof(int32_t) * (size_t)V, hipMemcpyHostToDevice, st));
  RGCN_HIP(c, hipMemsetAsync(q.hist, 0, sizeof(uint32_t) * kBins, st));
  RGCN_HIP(c, hipMemsetAsync(q.comp_state, 0, (size_t)(q.ncomp ? q.ncomp : 1), st));
  RGCN_HIP(c, hipStreamSynchronize(st));        // the host arrays are borrowed for the call only
  return RGCN_OK;
}

// the kernels of one draw, recorded into a hipGraph on a stream of its own (every per-draw quantity is read from q.params)
static rgcn_status record_draw(rgcn_ctx* c) {
  NeighborhoodBufs& q = c->nbr;
  const int n = (int)q.n, V = c->V, T = 256;
  const int launches = q.launches;
  if (!q.capture_stream) RGCN_HIP(c, hipStreamCreateWithFlags(&q.capture_stream, hipStreamNonBlocking));
  hipStream_t st = q.capture_stream;
  const DrawParams* p = reinterpret_cast<const DrawParams*>(q.params);
  const int nflags = launches * kFlagSlots * kFlagStride;
  const dim3 ge((unsigned)((n + T - 1) / T)), gw((unsigned)(((size_t)std::max(q.nseg, 1) * 64 + T - 1) / T)), bt(T);
  const dim3 gi((unsigned)((std::max(V, nflags) + T - 1) / T));
  RGCN_HIP(c, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(k_nbr_init, gi, bt, 0, st, p, q.dist, V, q.changed, nflags);
  for (int it = 0; it < launches; ++it)
    hipLaunchKernelGGL(k_nbr_sweep, gw, bt, 0, st, p, q.seg_v, q.seg_beg, q.seg_end, q.nseg, q.adj_other, q.adj_end, q.dist,
                       q.changed, it, sweeps_of_launch(it));
  hipLaunchKernelGGL(k_nbr_check, dim3(1), dim3(64), 0, st, q.changed, launches - 1, c->g.errflag);
  hipLaunchKernelGGL(k_nbr_keys, ge, bt, 0, st, p, q.triples, n, q.comp, q.dist, q.tkey);
  SelState* state = reinterpret_cast<SelState*>(q.state);
  hipLaunchKernelGGL(k_nbr_set_want, dim3(1), dim3(1), 0, st, p, state);
  const int nblocks = (n + kCompactBlock - 1) / kCompactBlock;
  for (int pass = 0; pass < kPasses; ++pass) {
    hipLaunchKernelGGL(k_nbr_hist, dim3(nblocks), dim3(1024), 0, st, pass, q.tkey, n, state, q.hist);
    hipLaunchKernelGGL(k_nbr_scan, dim3(1), dim3(1024), 0, st, pass, q.hist, state);
  }
  hipLaunchKernelGGL(k_nbr_count, dim3(nblocks), dim3(kCompactBlock), 0, st, p, q.triples, q.tkey, n, state, q.comp,
                     q.comp_state, q.bcnt);
  hipLaunchKernelGGL(k_nbr_offsets, dim3(1), dim3(1024), 0, st, p, q.bcnt, nblocks, c->g.errflag);
  hipLaunchKernelGGL(k_nbr_write, dim3(nblocks), dim3(kCompactBlock), 0, st, p, q.triples, q.tkey, n, state, q.comp,
                     q.comp_state, q.bcnt);
  const hipError_t e1 = hipGetLastError();
  const hipError_t e2 = hipStreamEndCapture(st, &q.draw_graph);
  if (e1 != hipSuccess || e2 != hipSuccess || !q.draw_graph)
    RGCN_FAIL(c, RGCN_ERR_HIP, std::string("device neighbourhood sampler: recording the draw failed: ") +
                                   hipGetErrorString(e1 != hipSuccess ? e1 : e2));
  RGCN_HIP(c, hipGraphInstantiate(&q.draw_exec, q.draw_graph, nullptr, nullptr, 0));
  return RGCN_OK;
}

// sample_size rows of the training graph -> batch_out [sample_size, 3], in edge order; runs on the current stream
rgcn_status neighborhood_sample(rgcn_ctx* c, int64_t k64, uint64_t seed, int32_t* batch_out, bool on_prefetch_stream) {
  NeighborhoodBufs& q = c->nbr;
  const int n = (int)q.n, k = (int)k64, V = c->V;
  if (k == 0) return RGCN_OK;
  if (c->capturing) RGCN_FAIL(c, RGCN_ERR_STATE, "the device neighbourhood sampler uploads the draw's parameters: not while a hipGraph is being captured");
  // ---- host: the vertex priorities settle which components are taken in full, where the budget runs out, and the
  // start vertex there
  std::vector<uint64_t> cmin((size_t)q.ncomp, ~0ull);
  std::vector<int32_t> cstart((size_t)q.ncomp, -1);
  for (int v = 0; v < V; ++v) {
    const int cid = q.comp_h[v];
    if (cid < 0) continue;
    const uint64_t p = vertex_priority(seed, v);
    if (p < cmin[cid]) { cmin[cid] = p; cstart[cid] = v; }
  }
  const int first = (int)(std::min_element(cmin.begin(), cmin.end()) - cmin.begin());
  int boundary = first;
  int64_t want = k;
  bool any_full = false;
  if (q.comp_edges_h[first] < k) {             // rare: the first component is exhausted, visit the others in order
    std::vector<int32_t> order((size_t)q.ncomp);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return cmin[a] < cmin[b]; });
    std::fill(q.comp_state_h.begin(), q.comp_state_h.end(), 0);
    for (int cid : order) {
      if (q.comp_edges_h[cid] < want) {
        q.comp_state_h[cid] = 1;
        any_full = true;
        want -= q.comp_edges_h[cid];
      } else {
        boundary = cid;
        break;
      }
    }
  }
  hipStream_t st = c->stream;
  // The draw's state is shared by every draw of the context.  Same stream: stream order.  Other stream (the driver
  // draws batch 1 on the main stream and, at once, batch 2 on the prefetch stream): wait for the previous draw's end
  // BEFORE this draw's parameters overwrite the ones its kernels still read.
  if (!q.ev_draw) RGCN_HIP(c, hipEventCreateWithFlags(&q.ev_draw, order_event_flags(c)));
  if (q.last_draw_stream != nullptr && q.last_draw_stream != st) RGCN_HIP(c, hipStreamWaitEvent(st, q.ev_draw, 0));
  if (any_full)
    RGCN_TRY(rgcn_copy_to_device_async(c, q.comp_state, q.comp_state_h.data(), (int64_t)q.ncomp, on_prefetch_stream ? 1 : 0));
  DrawParams dp;
  dp.seed = seed;
  dp.want = (unsigned long long)want;
  dp.out = batch_out;
  dp.start = cstart[boundary];
  dp.boundary = boundary;
  dp.k = k;
  dp.use_state = any_full ? 1 : 0;
  RGCN_TRY(rgcn_copy_to_device_async(c, q.params, &dp, (int64_t)sizeof(dp), on_prefetch_stream ? 1 : 0));
  // ---- device: one graph, recorded at the first draw
  if (!q.draw_exec) RGCN_TRY(record_draw(c));
  ProfScope ps(c, "nbr_sample", 8.0 * 2.0 * n * 12 + 32.0 * n, 0, 12.0 * n + 12.0 * k);
  RGCN_HIP(c, hipGraphLaunch(q.draw_exec, st));
  RGCN_HIP(c, hipEventRecord(q.ev_draw, st));
  q.last_draw_stream = st;
  return RGCN_OK;
}

}  // namespace rgcn
