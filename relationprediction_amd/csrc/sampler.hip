// Host-side minibatch construction: the neighbourhood edge sampler of the reference's train loop.
//
// Reference: sample_edge_neighborhood (code/train.py:161-198).  It grows a connected patch of the training
// graph edge by edge: a vertex is drawn with probability proportional to (its number of not-yet-picked
// incident edge ends) x (has it been touched yet), then one of its not-yet-picked incident edges uniformly
// (the reference draws among ALL its adjacency entries and retries while the edge is already picked — the
// same distribution); when nothing touched has free edges left (always at the first draw), any vertex that
// still has free edges is drawn uniformly.  The numpy original renormalises a length-V probability vector
// and calls np.random.choice per edge — O(V) per pick, about 6 s for the 30,000-edge batch of
// settings/gcn_block.exp.  Same process here with a Fenwick tree over the vertex weights and swap-remove
// adjacency lists: O(log V) per pick, a few milliseconds per batch, so it hides behind the GPU step.
// Pure host code (no HIP calls); lives in librgcn.so so that the driver has one native dependency.
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/rgcn.h"

struct rgcn_sampler {
  int64_t n_edges = 0;
  int64_t n_with_edges = 0;                     // vertices with at least one incident edge
  int32_t V = 0;
  std::vector<int32_t> src, dst;
  std::vector<int64_t> adj_ptr;                 // [V+1]
  std::vector<int32_t> adj_edge0;               // pristine adjacency: edge id of every entry
  std::vector<uint8_t> adj_side0;               // 0: this vertex is the edge's subject, 1: its object
  // per-sample state
  std::vector<int32_t> adj_edge;
  std::vector<uint8_t> adj_side;
  std::vector<int32_t> alive;                   // [V] live entries of each vertex (= sample_counts)
  std::vector<int64_t> pos[2];                  // [E] position of the edge's subject- / object-side entry
  std::vector<uint8_t> seen, picked;
  std::vector<int64_t> fen;                     // Fenwick tree over seen[v] ? alive[v] : 0
  int64_t fen_sum = 0;                          // its total, kept beside it
  // The per-sample state is restored, not rebuilt: a sample touches O(sample_size) adjacency entries and vertices of a
  // graph that may hold ten times as many edges, so every change is logged and undone afterwards.
  bool pristine = false;                        // adj_edge / adj_side / pos / alive / seen / picked / fen are in their initial state
  struct Undo { int64_t p; int32_t edge; uint8_t side; };
  std::vector<Undo> undo;
  std::vector<int32_t> touched;                 // vertices whose alive / seen changed
};

namespace {

struct Rng {                                    // splitmix64-seeded xoshiro256**
  uint64_t s[4];
  explicit Rng(uint64_t seed) {
    for (int i = 0; i < 4; ++i) {
      seed += 0x9e3779b97f4a7c15ull;
      uint64_t z = seed;
      z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
      z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
      s[i] = z ^ (z >> 31);
    }
  }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() {
    const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
  }
  uint64_t below(uint64_t n) {                  // unbiased integer in [0, n)
    const uint64_t lim = UINT64_MAX - UINT64_MAX % n;
    uint64_t x;
    do { x = next(); } while (x >= lim);
    return x % n;
  }
};

void fen_add(rgcn_sampler* s, int32_t i, int64_t delta) {
  int64_t* f = s->fen.data();
  const int32_t V = s->V;
  for (int32_t k = i + 1; k <= V; k += k & -k) f[k] += delta;
  s->fen_sum += delta;
}
// smallest index whose prefix sum exceeds r (0 <= r < total)
int32_t fen_find(const std::vector<int64_t>& f, int32_t V, int64_t r) {
  int32_t idx = 0, step = 1;
  while ((step << 1) <= V) step <<= 1;
  for (; step > 0; step >>= 1) {
    const int32_t nxt = idx + step;
    if (nxt <= V && f[nxt] <= r) { idx = nxt; r -= f[nxt]; }
  }
  return idx;
}

void remove_entry(rgcn_sampler* s, int32_t v, int64_t p) {
  const int64_t base = s->adj_ptr[v], last = base + s->alive[v] - 1;
  if (p != last) {                              // move the last live entry into the hole
    s->undo.push_back({p, s->adj_edge[p], s->adj_side[p]});
    s->undo.push_back({last, s->adj_edge[last], s->adj_side[last]});
    s->adj_edge[p] = s->adj_edge[last];
    s->adj_side[p] = s->adj_side[last];
    s->pos[s->adj_side[p]][s->adj_edge[p]] = p;
  }
  s->alive[v] -= 1;
  if (s->seen[v]) fen_add(s, v, -1);
}

void touch(rgcn_sampler* s, int32_t v) {
  if (!s->seen[v]) {
    s->seen[v] = 1;
    s->touched.push_back(v);
    if (s->alive[v]) fen_add(s, v, s->alive[v]);
  }
}

// back to the initial state: undo the adjacency moves in reverse, then the per-vertex / per-edge marks
void restore(rgcn_sampler* s, const int32_t* picked_ids, int64_t n_picked) {
  for (size_t k = s->undo.size(); k-- > 0;) {
    const rgcn_sampler::Undo& u = s->undo[k];
    s->adj_edge[u.p] = u.edge;
    s->adj_side[u.p] = u.side;
    s->pos[u.side][u.edge] = u.p;
  }
  s->undo.clear();
  for (int32_t v : s->touched) {
    s->alive[v] = (int32_t)(s->adj_ptr[v + 1] - s->adj_ptr[v]);
    s->seen[v] = 0;
  }
  s->touched.clear();
  for (int64_t i = 0; i < n_picked; ++i) s->picked[picked_ids[i]] = 0;
  std::memset(s->fen.data(), 0, s->fen.size() * sizeof(int64_t));
  s->fen_sum = 0;
}

}  // namespace

extern "C" {

rgcn_status rgcn_sampler_create(const int32_t* triples, int64_t num_triples, int32_t num_entities,
                                rgcn_sampler** out) {
  if (!out) return RGCN_ERR_INVALID;
  *out = nullptr;
  if (!triples || num_triples <= 0 || num_entities <= 0 || num_triples > INT32_MAX) return RGCN_ERR_INVALID;
  for (int64_t e = 0; e < num_triples; ++e)
    if (triples[3 * e] < 0 || triples[3 * e] >= num_entities || triples[3 * e + 2] < 0 ||
        triples[3 * e + 2] >= num_entities)
      return RGCN_ERR_INVALID;
  rgcn_sampler* s = new (std::nothrow) rgcn_sampler;
  if (!s) return RGCN_ERR_NOMEM;
  try {
    s->n_edges = num_triples;
    s->V = num_entities;
    s->src.resize(num_triples);
    s->dst.resize(num_triples);
    s->adj_ptr.assign((size_t)num_entities + 1, 0);
    for (int64_t e = 0; e < num_triples; ++e) {
      s->src[e] = triples[3 * e];
      s->dst[e] = triples[3 * e + 2];
      s->adj_ptr[s->src[e] + 1] += 1;
      s->adj_ptr[s->dst[e] + 1] += 1;
    }
    for (int32_t v = 0; v < num_entities; ++v) s->adj_ptr[v + 1] += s->adj_ptr[v];
    s->adj_edge0.resize(2 * (size_t)num_triples);
    s->adj_side0.resize(2 * (size_t)num_triples);
    std::vector<int64_t> fill(s->adj_ptr.begin(), s->adj_ptr.end() - 1);
    for (int64_t e = 0; e < num_triples; ++e) {       // same entry order as the reference's adj_list
      int64_t p = fill[s->src[e]]++;
      s->adj_edge0[p] = (int32_t)e; s->adj_side0[p] = 0;
      p = fill[s->dst[e]]++;
      s->adj_edge0[p] = (int32_t)e; s->adj_side0[p] = 1;
    }
    s->pos[0].resize(num_triples);
    s->pos[1].resize(num_triples);
  } catch (const std::bad_alloc&) {
    delete s;
    return RGCN_ERR_NOMEM;
  }
  *out = s;
  return RGCN_OK;
}

void rgcn_sampler_destroy(rgcn_sampler* s) { delete s; }

rgcn_status rgcn_sampler_edge_neighborhood(rgcn_sampler* s, int64_t sample_size, uint64_t seed, int32_t* out_ids) {
  if (!s || !out_ids || sample_size < 0) return RGCN_ERR_INVALID;
  // the reference dies with "probabilities contain NaN" once every edge is picked
  if (sample_size > s->n_edges) return RGCN_ERR_INVALID;
  const int32_t V = s->V;
  try {
    if (!s->pristine) {                         // first sample: build the working state once
      s->adj_edge = s->adj_edge0;
      s->adj_side = s->adj_side0;
      s->alive.resize(V);
      s->n_with_edges = 0;
      for (int32_t v = 0; v < V; ++v) {
        s->alive[v] = (int32_t)(s->adj_ptr[v + 1] - s->adj_ptr[v]);
        s->n_with_edges += s->alive[v] > 0;
      }
      for (size_t p = 0; p < s->adj_edge.size(); ++p) s->pos[s->adj_side[p]][s->adj_edge[p]] = (int64_t)p;
      s->seen.assign(V, 0);
      s->picked.assign((size_t)s->n_edges, 0);
      s->fen.assign((size_t)V + 1, 0);
      s->fen_sum = 0;
      s->pristine = true;
    }
    s->undo.reserve(4 * (size_t)sample_size);
    s->touched.reserve(2 * (size_t)sample_size + 16);
  } catch (const std::bad_alloc&) {
    return RGCN_ERR_NOMEM;
  }
  Rng rng(seed);
  int64_t with_free = s->n_with_edges;          // vertices that still have free edge ends
  for (int64_t i = 0; i < sample_size; ++i) {
    int32_t v;
    const int64_t total = s->fen_sum;
    if (total > 0) {
      v = fen_find(s->fen, V, (int64_t)rng.below((uint64_t)total));
    } else {                                    // nothing touched has free edges: uniform over vertices that do
      int64_t k = (int64_t)rng.below((uint64_t)with_free);
      v = 0;
      for (int32_t u = 0; u < V; ++u)
        if (s->alive[u] > 0 && k-- == 0) { v = u; break; }
    }
    touch(s, v);
    const int64_t p = s->adj_ptr[v] + (int64_t)rng.below((uint64_t)s->alive[v]);
    const int32_t e = s->adj_edge[p];
    const int32_t a = s->src[e], b = s->dst[e];
    out_ids[i] = e;
    s->picked[e] = 1;
    // both ends lose one free edge end (a self loop loses two at the same vertex)
    const bool a_had = s->alive[a] > 0, b_had = s->alive[b] > 0;
    remove_entry(s, a, s->pos[0][e]);
    remove_entry(s, b, s->pos[1][e]);
    if (a == b) { with_free -= (a_had && s->alive[a] == 0); }
    else { with_free -= (a_had && s->alive[a] == 0); with_free -= (b_had && s->alive[b] == 0); }
    touch(s, a);
    touch(s, b);
  }
  restore(s, out_ids, sample_size);
  return RGCN_OK;
}

}  // extern "C"
