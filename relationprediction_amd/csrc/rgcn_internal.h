// Internal declarations shared by the translation units of librgcn.so.
// Nothing here is part of the C ABI (include/rgcn.h is).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rgcn.h"

namespace rgcn {

// Experiment knobs exist in the devtools build only (librgcn_devtools.so, -DRGCN_DEVTOOLS: tools/ and a few tests); the
// product library reads no tuning variable from the environment and runs ONE configuration, the default.
#ifdef RGCN_DEVTOOLS
inline int knob(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}
#else
constexpr int knob(const char*, int dflt) { return dflt; }
#endif

// ---------------------------------------------------------------- error plumbing
void set_global_error(const std::string& s);

#define RGCN_HIP(ctx, expr)                                                                      \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ ":" +       \
                   std::to_string(__LINE__) + ")";                                               \
      return RGCN_ERR_HIP;                                                                       \
    }                                                                                            \
  } while (0)

#define RGCN_TRY(expr)                       \
  do {                                       \
    rgcn_status _s = (expr);                 \
    if (_s != RGCN_OK) return _s;            \
  } while (0)

#define RGCN_FAIL(ctx, code, msg)            \
  do {                                       \
    (ctx)->err = (msg);                      \
    return (code);                           \
  } while (0)

// ---------------------------------------------------------------- dropout generator
// Counter-based Bernoulli(keep) draw: nothing is stored, forward and backward re-derive the same
// bit from (seed, layer, flat element index).  Statistically equivalent to tf.nn.dropout's floor(keep + U[0,1))
// (message_gcn.py:64); the exact TF Philox stream is not reproducible (parity tests inject masks).
// drop_bits (splitmix64 per draw, 24 bits out) serves the negative sampler; the dropout masks use the two-stage form
// below since round 4: a 64-bit key per (seed, layer) from splitmix64 -- uniform over a launch -- and a 32-bit
// two-multiply hash per element keyed at both rounds (9 VALU operations instead of ~40: the per-element 64-bit
// multiplies were a third of the VALU work of the single-pass layer kernels); keep iff the top 24 bits < keep * 2^24.
__host__ __device__ inline uint64_t splitmix64_mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ inline uint32_t drop_bits(uint64_t seed, uint32_t layer, uint64_t idx) {
  return (uint32_t)(splitmix64_mix(seed + 0x9E3779B97F4A7C15ull * (idx + ((uint64_t)layer << 44) + 1ull)) >> 40);
}
struct DropKey {
  uint32_t k0, k1;
};
__host__ __device__ inline DropKey drop_key_of(uint64_t seed, uint32_t layer) {
  const uint64_t z = splitmix64_mix(seed + 0x9E3779B97F4A7C15ull * (((uint64_t)layer << 44) + 1ull));
  return DropKey{(uint32_t)z, (uint32_t)(z >> 32)};
}
__host__ __device__ inline uint32_t drop_bits24(const DropKey& k, uint64_t idx) {
  const uint32_t hi = (uint32_t)(idx >> 32);
  uint32_t x = (uint32_t)idx ^ k.k0 ^ ((hi << 16) | (hi >> 16));
  x ^= x >> 16;
  x *= 0x7feb352du;
  x += k.k1;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x >> 8;
}

enum DropMode { DROP_NONE = 0, DROP_RNG = 1, DROP_MASK = 2 };

// Rows of the incidence CSR with more slots than this are pre-reduced by a whole workgroup
// (k_long_row_reduce) so that the one-wave-group-per-row combine never walks a hub serially.
constexpr int kLongRow = 32;
// Full-graph scale only (more than 65536 messages, block kind): a row with more than kGiantRow slots is cut into
// kGiantRow-slot pieces that separate workgroups sum; a finishing pass adds a row's pieces in piece order.
constexpr int kGiantRow = 2048;
constexpr int kAuxStreams = 3;

struct DropSpec {
  int32_t mode;         // DropMode
  uint32_t layer;       // 1..L
  uint32_t thresh;      // keep iff drop_bits < thresh
  float inv_keep;       // 1 / keep_prob
  uint64_t seed;
  const uint8_t* mask;  // [V,d] 0/1 for this layer (DROP_MASK)
  const uint64_t* seed_offset;  // device counter added to `seed` (replayed hipGraphs draw fresh masks); may be null
};

#if defined(__HIPCC__)
// the (seed, layer) key of a launch's dropout spec; derive it ONCE per thread, outside the element loops
__device__ __forceinline__ DropKey drop_key(const DropSpec& ds) {
  if (ds.mode != DROP_RNG) return DropKey{0u, 0u};
  return drop_key_of(ds.seed + (ds.seed_offset ? *ds.seed_offset : 0ull), ds.layer);
}
// dropout factor of flat element idx: 1/keep or 0 (1 when the spec is inactive)
__device__ __forceinline__ float drop_factor(const DropSpec& ds, const DropKey& key, size_t idx) {
  if (ds.mode == DROP_NONE) return 1.0f;
  if (ds.mode == DROP_RNG) return drop_bits24(key, idx) < ds.thresh ? ds.inv_keep : 0.0f;
  return ds.mask[idx] ? ds.inv_keep : 0.0f;
}
__device__ __forceinline__ float drop_factor(const DropSpec& ds, size_t idx) { return drop_factor(ds, drop_key(ds), idx); }
#endif

// ---------------------------------------------------------------- profiling records
struct ProfRec {
  const char* name;
  hipEvent_t e0, e1;
  double bytes, flops, cbytes;
};
struct ProfAgg {
  std::string name;
  int64_t calls;
  double ms, bytes, flops, cbytes;
};

// ---------------------------------------------------------------- parameters
enum ParamLayout {
  LAYOUT_PLAIN = 0,        // device layout == host layout
  LAYOUT_BLOCK_T = 1,      // host [R][nb][sd*sd]  <-> device [R][sd*sd][nb]
  LAYOUT_BASIS_T = 2       // host [d][B][d]       <-> device [B][d][d]
};

struct Param {
  std::string name;
  int32_t ndim;
  int64_t shape[4];
  int64_t count;
  float* val;    // device (view into a layer allocation for W_f/W_b)
  float* grad;   // device
  int32_t layout;
  bool no_grad = false;   // created but never used by the forward pass (the per-layer bias)
};

struct LayerBufs {
  // BLOCK: wrel = [2R][sd*sd][nb] (first R = W_forward, last R = W_backward), same for grel.
  // BASIS: wrel = [2][B][d][d]  (GEMM operand [2B*d, d]); coef = [2R][B] (first R = C_forward).
  float* wrel = nullptr;
  float* grel = nullptr;
  float* coef = nullptr;
  float* gcoef = nullptr;
  float* wself = nullptr;   // [d,d]
  float* gwself = nullptr;
  float* bias = nullptr;    // [d], unused by the math
  float* gbias = nullptr;
  // MFMA-fragment tables of the weights that are the B operand of a contraction (gemm_presplit_b: the three bf16 planes of
  // the exact split, in fragment order), rebuilt when the weights change: W_self for H.W_self (nn) and dS.W_self^T (nt);
  // basis kind: W'_dir for Zc.W' (nn) and Dc.W'^T (nt), two groups each.  Split arithmetic only (rgcn_set_gemm_mode 6 / 9).
  void *wself_nn = nullptr, *wself_nt = nullptr, *wrel_nn = nullptr, *wrel_nt = nullptr;
  // BLOCK, destination-major banded layer kernel (block_rows.hip): band-tiled copy of wrel,
  // [2R][8 bands][ceil(sd*sd/4)][GW lanes][4], allocated at first use
  float* wtile = nullptr;
  uint64_t wtile_version = ~0ull;
};

struct GraphBufs {
  int32_t* triples = nullptr;   // [maxE,3] (only when the host variant of set_graph is used)
  const int32_t* cur = nullptr; // triples of the current graph (ours or the caller's)
  int64_t E = 0;
  int32_t* indeg = nullptr;     // [V]
  int32_t* outdeg = nullptr;    // [V]
  int32_t* counters = nullptr;  // base of the zero-initialised block (indeg, outdeg, nlong)
  size_t counters_bytes = 0;
  int32_t* row_ptr = nullptr;   // [V+1]
  int32_t* long_rows = nullptr; // compacted list of rows with more than kLongRow slots (and at most the giant threshold)
  // giant rows: per row its first piece id and piece count; per piece its row and its index inside the row;
  // ngiant[0] = rows, ngiant[1] = pieces (device counters in the zero-initialised block)
  int32_t *giant_rows = nullptr, *giant_first = nullptr, *giant_cnt = nullptr, *piece_row = nullptr, *piece_k = nullptr;
  int32_t* ngiant = nullptr;
  int32_t giant_cap = 0, piece_cap = 0;
  bool giant_on = false;        // this graph was prepared with the giant-row cut enabled
  // rows ordered by DESCENDING number of slots, long / giant rows last (block kind, one GPU: the destination-major
  // layer kernel hands the four lane groups of a wavefront rows of equal length); row_key[v] = 32 - slots (33: long)
  uint32_t *row_key = nullptr, *row_key_s = nullptr;
  int32_t* row_order = nullptr;
  uint16_t* row_tab = nullptr;
  int32_t* nlong = nullptr;     // device counter (lives in the zero-initialised counter block)
  int32_t long_cap = 0;
  int32_t* rel_ptr = nullptr;   // [2R+1]
  int chunk = 48;               // messages per relation chunk of THIS graph (scales with its size)
  int32_t* chunk_ptr = nullptr; // [2R+1]
  int32_t* cum_in = nullptr;    // [V+1]  (tf_as_executed)
  int32_t* cum_out = nullptr;   // [V+1]
  uint32_t *keyv = nullptr, *keyv_s = nullptr, *keyr = nullptr, *keyr_s = nullptr;  // [2maxE]
  int32_t *valv = nullptr, *permv = nullptr, *valr = nullptr, *permr = nullptr;      // [2maxE]
  int32_t* pos = nullptr;       // [2maxE] incidence -> CSR slot
  // relation-sorted message list (SoA, [2maxE] each)
  int32_t *m_src = nullptr, *m_dst = nullptr, *m_dslot = nullptr, *m_sslot = nullptr;
  // slot-ordered copies for the row-major gathers of the basis path (destination order: d_*, source
  // order: s_*): partner vertex, directed relation, normalisation of the message in that slot
  int32_t *d_src = nullptr, *d_rel = nullptr, *s_dst = nullptr, *s_rel = nullptr;
  float *d_norm = nullptr, *s_norm = nullptr;
  float* m_norm = nullptr;
  // basis kind: the (row, direction) UNITS of the graph -- row v has a unit of direction dir (0: messages of W_forward,
  // arriving along an edge; 1: W_backward) iff one of this rank's messages of that direction lands on it.  The dense
  // contractions of the basis layer run over units, not over all V rows (basis.hip).  has_dir [2][V] 0/1;
  // unit_ptr [2][V+1] = its exclusive scan (unit index of (v, dir); entry V = number of units); unit_rows [2][V] = the
  // row of every unit, ascending
  int32_t *has_dir = nullptr, *unit_ptr = nullptr, *unit_rows = nullptr;
  int64_t units_host = -1;      // number of units, once somebody has read it back (profile accounting); -1: unknown
  uint32_t *keyv_t = nullptr, *keyr_t = nullptr;   // sort scratch (csr_sort.hip)
  uint16_t *tablev = nullptr, *tabler = nullptr;
  // prefetch bookkeeping (rgcn_prefetch_graph_device): which graph this set was prepared for
  const int32_t* pf_tri = nullptr;
  int64_t pf_E = -1;
  int64_t pf_keep = -1;         // >= 0: prepared under edge dropout (keep of pf_E batch edges, seed pf_eseed)
  uint64_t pf_eseed = 0;
  bool pf_valid = false;
  hipEvent_t ev_ready = nullptr;   // recorded on the prefetch stream when the set is complete
  hipEvent_t ev_free = nullptr;    // recorded on the main stream when the last step using the set ended
  bool ready_in_capture = false, free_in_capture = false;   // those records belong to the running capture
  int32_t* owner = nullptr;     // [R]
  int32_t* errflag = nullptr;   // device int: nonzero = bad id seen
  bool ready = false;
};

// decoder batch structures (csrc/decoder.hip)
struct DecoderBufs {
  int64_t maxN = 0;
  int32_t N = 0;
  int64_t N_total = 0;         // triples of the whole batch when N is one rank's slice of it (denominator of the means)
  // the last batch rgcn_negative_sample_device (or the fused minibatch step) wrote: `tiled_period` rows tiled rate + 1
  // times into tiled_X -- rows p, p + period, p + 2 period, ... differ from row p in ONE entity.  decoder_prepare sorts
  // only the first copies by relation and puts the others directly behind them (k_dec_expand): a third of the sort work,
  // and the copies of a triple sit next to each other in a relation chunk, their shared rows fetched from HBM once
  // instead of rate + 1 times.  The expansion checks that every copy still has its first copy's relation (flag 16).
  const int32_t* tiled_X = nullptr;
  int64_t tiled_period = 0, tiled_N = 0;
  int32_t* perm_pos = nullptr; // first copies in relation order
  const int32_t* X = nullptr;
  uint32_t *keyv = nullptr, *keyv_s = nullptr, *keyr = nullptr, *keyr_s = nullptr;
  int32_t *valv = nullptr, *permv = nullptr, *valr = nullptr, *permr = nullptr;
  int32_t *row_ptr = nullptr, *rel_ptr = nullptr, *chunk_ptr = nullptr;
  int32_t *e_other = nullptr, *e_rel = nullptr, *e_trip = nullptr;
  int32_t *long_rows = nullptr, *nlong = nullptr;     // nlong[0] = long rows, nlong[1] = their pieces
  int32_t *long_first = nullptr, *long_cnt = nullptr;  // per long row: first piece id, number of pieces
  int32_t *piece_row = nullptr, *piece_k = nullptr;    // per piece: its row, its index inside the row
  float* piece_slab = nullptr;                         // [piece_cap][d] partial sums of the pieces
  int32_t long_cap = 0, piece_cap = 0, max_chunks = 0, energy_blocks = 0;
  float *dx = nullptr, *loss_part = nullptr, *slab = nullptr;
  // band-major, line-aligned copies of the codes and of W_relation for the line form of the entity gradient
  // (decoder.hip, k_dec_entity_lines): [bands][V][32] and [bands][R][32], bands = ceil(d / 32); rebuilt every call
  float *cb = nullptr, *rb = nullptr, *e_g = nullptr;   // e_g: the loss gradient of every incidence slot
  uint32_t *row_key = nullptr, *row_key_s = nullptr;    // rows by descending number of incidences (one radix pass)
  int32_t* row_order = nullptr;
  uint16_t* row_tab = nullptr;
  int32_t nbands = 0, cus = 0;     // cus: compute units of the device (persistent workgroups of the line kernel)
  double* loss = nullptr;
  uint32_t *keyv_t = nullptr, *keyr_t = nullptr;   // sort scratch (csr_sort.hip)
  uint16_t *tablev = nullptr, *tabler = nullptr;
  hipEvent_t ev_ready = nullptr;
  bool loss_valid = false;
};

// device neighbourhood sampler (csrc/neighborhood.hip): the training graph and the scratch of one draw
struct NeighborhoodBufs {
  int64_t n = 0;                         // training triples
  int32_t* triples = nullptr;            // [n,3]
  // adjacency CSR by vertex, self loops left out: the other endpoint, and the id 2 e + side of the edge END AT THAT
  // OTHER endpoint (whose clock a relaxation adds)
  int32_t *adj_other = nullptr, *adj_end = nullptr;
  int32_t *seg_v = nullptr, *seg_beg = nullptr, *seg_end = nullptr;     // segments of <= 256 entries of one vertex's list
  int32_t nseg = 0;
  int32_t launches = 64;                 // sweep launches of one draw (from the graph's diameter, at reserve)
  int32_t* comp = nullptr;               // [V] connected component of the vertex (-1: no edges)
  uint8_t* comp_state = nullptr;         // [ncomp] 1: every edge of the component is in the batch (device copy)
  uint32_t* dist = nullptr;              // [V] touched-time (bits of a non-negative float)
  unsigned long long* tkey = nullptr;    // [n] (pick time | edge id) of the boundary component's edges, ~0 elsewhere
  uint32_t* hist = nullptr;              // [4096] digit histogram of the radix select
  unsigned long long* state = nullptr;   // select state (prefix, wanted rank)
  int32_t* changed = nullptr;            // per-launch "something moved" flags of the relaxation
  uint32_t* bcnt = nullptr;              // per-block counts / offsets of the compaction
  void* params = nullptr;                // the draw's parameters (seed, start vertex, ...: neighborhood.hip DrawParams)
  hipGraph_t draw_graph = nullptr;       // the kernels of one draw, recorded once, replayed per draw
  hipGraphExec_t draw_exec = nullptr;
  hipStream_t capture_stream = nullptr;
  // one set of per-draw state (params, distances, keys, select state, the graph) serves draws on the main AND the
  // prefetch stream: a draw on the other stream waits for the previous draw's end
  hipEvent_t ev_draw = nullptr;
  hipStream_t last_draw_stream = nullptr;
  // host side (components are a property of the graph: found once, at reserve)
  std::vector<int32_t> comp_h;           // [V]
  std::vector<int64_t> comp_edges_h;     // [ncomp] edges per component
  std::vector<uint8_t> comp_state_h;     // [ncomp] scratch of one draw
  int32_t ncomp = 0;
};

struct OptimizerState {
  bool configured = false;
  float lr = 0.01f, beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f, max_norm = 0.f;
  int64_t t = 0;
  std::vector<float*> m, v;
  float* part = nullptr;
  size_t part_cap = 0;
  float* state = nullptr;    // [clip scale, global norm, step count t, bias-corrected rate]
  float* shard_sq = nullptr; // world > 1: squared norm of this rank's relation-sharded gradients (all-reduced)
  bool norm_pending = false; // optimizer_norm_partial ran, optimizer_apply has not
};

}  // namespace rgcn

struct rgcn_ctx {
  rgcn_config cfg;
  int V = 0, R = 0, d = 0, L = 0, nb = 0, sd = 0, kind = 0, B = 0;
  int rank = 0, world = 1;
  int row_lo = 0, row_hi = 0;   // row shard of this rank: [rank * shard_rows, +shard_rows) cut at V
  int shard_rows = 0;           // ceil(V / world): equal chunks for the reduce-scatter / all-gather
  int V_pad = 0;                // world * shard_rows: rows every exchanged [V,d] buffer is allocated with
  float* repl_grads = nullptr;  // world > 1: W_self (and basis W') gradients of all layers, contiguous -> ONE all-reduce
  size_t repl_grads_floats = 0;
  hipEvent_t ev_gather = nullptr;   // the all-gather of the rows finished last (side stream 1) is complete
  bool gather_pending = false;      // ... and somebody still has to wait for it
  hipStream_t stream = nullptr;           // stream launches go to (main stream unless a StreamScope is active)
  hipStream_t main_stream = nullptr;
  hipStream_t aux[rgcn::kAuxStreams] = {nullptr, nullptr, nullptr};  // side streams: 0, 1 for independent kernels of one layer,
                                                               // 2 for the decoder's relation gradient (runs beside the backward pass)
  hipEvent_t ev_fork = nullptr, ev_join[rgcn::kAuxStreams] = {nullptr, nullptr, nullptr};
  bool use_aux = true;
  bool aux_dirty[rgcn::kAuxStreams] = {false, false, false};   // something was forked onto side stream k since its last join
  // block kind: 0 = message kernel + k_combine (two kernels, [2E,d] message buffer), 1 = the destination-major banded
  // single-pass layer (block_rows.hip; no message buffer, weights through L2)
  int fuse = 1;
  uint64_t weights_version = 1;           // bumped whenever a parameter value changes (set_param, Adam)
  bool dw_pending = false;                // a dW-only message-gradient kernel of the previous backward layer is still on side stream 0
  std::string err;

  std::vector<rgcn::Param> params;
  std::vector<rgcn::LayerBufs> layers;   // index 1..L (0 unused)
  float *w_emb = nullptr, *g_emb = nullptr, *b_emb = nullptr, *gb_emb = nullptr;
  float *w_rel = nullptr, *g_rel = nullptr;   // W_relation [EntityCount, d] (decoder weight)
  rgcn::DecoderBufs dec;
  rgcn::NeighborhoodBufs nbr;
  rgcn::OptimizerState opt;

  std::vector<float*> H;                 // H[0..L], [V,d] each
  float* self_buf = nullptr;             // S / G : [V,d]
  float* exch = nullptr;                 // exchange buffer [V,d] (world > 1)
  float* dbuf[2] = {nullptr, nullptr};   // D_l ping-pong
  float* dsbuf[2] = {nullptr, nullptr};  // dS_l ping-pong
  float* msgbuf = nullptr;               // Y / Z : [2*maxE, d]  (BLOCK)   or Z [V, 2B*d] (BASIS)
  float* msgbuf2 = nullptr;              // BASIS: dZ [V, 2B*d]
  std::vector<float*> zsave;             // BASIS: Z of every layer, kept for dW' = Z^T.D
  float* aggbuf = nullptr;               // BASIS: Z.W' product [V,d]
  float* slab = nullptr;                 // split-K partial slabs of the GEMM
  size_t slab_floats = 0;
  float* slab_dw = nullptr;              // per-chunk dW slabs of the block-diagonal backward
  size_t slab_dw_floats = 0;
  float* stage = nullptr;                // host<->device staging for layout conversion
  size_t stage_floats = 0;
  uint8_t* masks = nullptr;              // [L,V,d] explicit dropout masks
  float* colsum_part = nullptr;
  size_t colsum_part_floats = 0;
  int32_t colsum_parts = 0;    // > 0: the last block_rows backward launch left that many [d] partial rows in colsum_part
  void* debug_buf = nullptr;             // devtools builds: where k_gemm_w8<.., DBG_TIMELINE> leaves its stamps
  float* zeros = nullptr;                // 1024 zero floats (masked-lane load target of the GEMMs; 3 KB for a masked LDS-DMA)

  rgcn::GraphBufs g;                     // ACTIVE graph structures
  rgcn::GraphBufs g_alt;                 // second set: next graph is prepared here beside the running step
  hipStream_t pf_stream = nullptr;       // stream of rgcn_prefetch_graph_device
  int aux_priority = 0, pf_priority = 0;  // what the side / prefetch streams were created with (stream pool key)
  // pinned staging ring of rgcn_copy_to_device_async: the caller's memory is copied here during the call, the
  // transfer itself is stream-ordered and the host does not wait for it
  static constexpr int kStageSlots = 8;
  static constexpr size_t kStageBytes = (size_t)1 << 20;
  uint8_t* stage_host = nullptr;         // kStageSlots * kStageBytes, hipHostMalloc
  uint8_t* readback_host = nullptr;      // 32 pinned bytes: rgcn_get_loss fetches the loss and the error flag with one wait
  hipEvent_t stage_done[kStageSlots] = {};
  int stage_next = 0;
  int chunk = 48;                        // messages per relation chunk (capacity-scaled upper bound; GraphBufs::chunk is per graph)
  int gemm_mode = 0;                     // 0: fp32 MFMA; 3/6/9: bf16 split with that many partial products
  int msg_block = 0, msg_slots = 0;      // k_msg launch geometry

  // forward/backward state
  uint64_t frag_version = ~0ull;         // weights_version the weight fragment tables (LayerBufs::wself_nn ...) were built from
  bool frag_fresh = false;               // ... built since the last rgcn_forward_begin (captured steps rebuild once per step)
  bool wtile_fresh = false;              // the same for the band-tiled block weights (LayerBufs::wtile)
  bool fwd_done = false;
  int fwd_train = 0;
  uint64_t seed = 0;
  bool explicit_masks = false;
  const float* bwd_D = nullptr;          // D_l of the backward layer in flight
  const float* bwd_dS = nullptr;         // dS_l = D_l * dropout_l
  int bwd_layer = 0;                     // next layer the backward pass will process (L..1, 0 = done)
  // hipGraph capture (rgcn_capture_begin / _end / rgcn_graph_launch)
  bool capturing = false;
  bool use_aux_before_capture = true;    // side-stream setting to restore when the capture ends
  bool cap_pf_forked = false;            // the prefetch stream has joined the capture and must be joined back
  uint64_t* replay_counter = nullptr;    // device counter bumped at the start of every captured graph
  hipEvent_t ev_step_begin = nullptr;    // recorded on the main stream where a step starts (capture: fork point of the prefetch)
  bool step_begin_in_capture = false;
  std::vector<hipGraphExec_t> graphs;
  std::vector<hipGraph_t> graph_defs;
  float* giant_slab = nullptr;           // [piece_cap][d] partial sums of giant-row pieces (scratch of one combine launch)
  float *rank_q = nullptr, *rank_s = nullptr;   // ranking: query rows [max,d], energies [max,V]
  int32_t* rank_bad = nullptr;
  float* rank_thr = nullptr;             // per query: the smallest energy whose fp32 sigmoid reaches the gold entity's
  int64_t rank_max = 0;
  float* dcodes_own = nullptr;           // [V,d] staging for the host variant of backward

  // comm
  void* comm = nullptr;                  // ncclComm_t
  // profiling
  bool prof_on = false;
  bool dropout_lds_configured = false;   // k_edge_dropout's dynamic-LDS attribute set on this context's device
  std::vector<rgcn::ProfRec> prof;
  std::vector<hipEvent_t> event_pool;
  std::vector<rgcn::ProfAgg> prof_agg;
  hipEvent_t t0 = nullptr, t1 = nullptr;
};

namespace rgcn {

// Flags of the events that only ORDER this context's own streams on its own device (fork / join of the side streams, the
// prefetched graph structures, the decoder batch, the sampler's state).  A default hipEventRecord is a barrier packet with a
// system-scope release and acquire; what these events order is kernels of one device, each of which already carries its
// agent-scope fences, so on one GPU the event's own fence is dropped: 0.528-0.534 ms per headline step against 0.535-0.543,
// three interleaved runs each on one box (release-to-device scope: 0.538-0.540; tools/gpu_r5_evflag.sh).  A sharded
// context, whose buffers other ranks' kernels read and write, keeps the default.
// kernel_only: the event orders kernels against kernels (fork, join).  Events that can also order copy-engine work -- a
// graph set's ready / free (host staging copies and memsets on the prefetch stream), the sampler's draw, the decoder batch's
// ready event, the step-begin marker -- keep the default fence: HIP documents no agent-scope release for SDMA operations.
inline unsigned order_event_flags(const rgcn_ctx* c, bool kernel_only = false) {
  return hipEventDisableTiming | (kernel_only && c->world == 1 ? hipEventDisableSystemFence : 0u);
}

// Runs the launches inside its scope on side stream k, ordered after everything already queued on
// the main stream (fork); join() makes the main stream wait for that side stream again.
struct StreamScope {
  rgcn_ctx* c;
  hipStream_t saved;
  bool active;
  // in_capture: the fork is also taken while a step is being captured (a captured step is otherwise one chain: the
  // side streams are switched off for the capture, rgcn_capture_begin)
  StreamScope(rgcn_ctx* ctx, int k, bool in_capture = false);
  ~StreamScope();
};
rgcn_status stream_join(rgcn_ctx* c, int k);

// Brackets a launch with events when profiling is on.
struct ProfScope {
  rgcn_ctx* c;
  int idx;
  // bytes = DESIGN bytes (what the kernel requests from the memory system by construction: gathers counted per use,
  // staging slabs included); compulsory = every distinct input byte once + every output byte once, < 0:
  // the same figure.  Roofline fractions are computed on the compulsory bytes.
  ProfScope(rgcn_ctx* ctx, const char* name, double bytes, double flops, double compulsory = -1.0);
  ~ProfScope();
};

// ---- graph_prep.hip
rgcn_status graph_alloc(rgcn_ctx* c, const GraphBufs* share);
void graph_free(rgcn_ctx* c);
rgcn_status graph_build(rgcn_ctx* c, const int32_t* triples_dev, int64_t E);
rgcn_status graph_build_dropout(rgcn_ctx* c, const int32_t* batch_dev, int64_t n, int64_t keep, uint64_t seed,
                                const uint8_t* keep_mask_dev);

// ---- csr_sort.hip: stable sort of (key, position) pairs by a small integer key, up to two sorts per call
struct SortSpec {
  const uint32_t* key_in;   // n keys, each <= max_key < 2^24
  uint32_t* key_out;        // sorted keys
  int32_t* val_out;         // original positions in key order (ties in position order)
  uint32_t* key_tmp;        // scratch, n elements each
  int32_t* val_tmp;
  int32_t* pos_out;         // optional: pos_out[position] = slot (inverse of val_out)
  uint16_t* table;          // scratch, sort_table_elems(n) elements
  int64_t n;
  uint32_t max_key;
};
size_t sort_table_elems(size_t n);
rgcn_status sort_pairs(rgcn_ctx* c, const char* tag, int njobs, const SortSpec* specs);

// ---- gemm_f32.hip
// Several contractions of one shape in ONE launch (blockIdx.y = group), each with its own operands and -- read on the
// device, so that nothing about the graph has to come back to the host -- its own extent along M (rows of A and C that
// exist; workgroups of tiles beyond it leave at once and write nothing) or along K (the depth of the contraction; the
// split-K slices divide the ACTUAL depth evenly).  The row-compacted basis contraction (basis.hip) is two groups, one
// per message direction, whose row counts the graph preparation leaves in GraphBufs::unit_ptr.
struct GemmBatch {
  int groups = 1;
  size_t strideA = 0, strideB = 0, strideC = 0;   // floats between consecutive groups' operands
  const int32_t* limit = nullptr;                 // device, limit[g * limit_stride]: extent of group g (<= M resp. K)
  int limit_stride = 1;
  int limit_on_k = 0;                             // 0: rows of A / C; 1: depth K
  // optional: B already split into bf16 planes in MFMA fragment order (gemm_presplit_b; a weight, split once per weight
  // update instead of once per tile and step).  Used by the split-arithmetic kernel when A is k-contiguous and there is no
  // split over K; ignored otherwise (B itself must still be passed).  strideBfrag: 16-byte words between groups.
  const void* bfrag = nullptr;
  size_t strideBfrag = 0;
  // with bfrag: take the 128 x 256 / eight-wavefront kernel (gemm_bf16x3_w8.hip; one workgroup holds a whole CU) instead of
  // the 128 x 128 / four-wavefront one (two per CU, room for another kernel's workgroups beside them).  Same result bit for
  // bit; which is faster in the step depends on what runs beside the product: the forward products
  // run alone on the main stream (wide), the backward ones beside dW_self and the relation-weight kernels (not wide).
  int wide = 0;
};
size_t gemm_bfrag_words(int K, int N);
struct PresplitJob {
  const float* B;      // the operand B (k, n): stored [n][k] (b_kc) or [k][n], leading dimension ldb
  void* F;             // its fragment table, gemm_bfrag_words(K, N) 16-byte words
  int ldb, K, N, b_kc;
};
rgcn_status gemm_presplit_b(rgcn_ctx* c, const PresplitJob* jobs, int n);

// C[M,N] (ldc) = A(m,k) . B(k,n).  a_kc: A stored [m][k] (k contiguous, lda) else [k][m];
// b_kc: B stored [n][k] (k contiguous, ldb) else [k][n].  split_k > 1 writes partial slabs to
// `slab` ([group][split_k][M][N]) and reduces them into C deterministically.
rgcn_status gemm_f32(rgcn_ctx* c, const char* tag, bool a_kc, bool b_kc, int M, int N, int K,
                     const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                     int split_k, const GemmBatch* batch = nullptr, double prof_scale = 1.0);

// gemm_bf16x3.hip: the same contraction on the bf16 matrix cores (exact 3-way operand split)
hipError_t gemm_bf16x3_launch(rgcn_ctx* c, int terms, bool a_kc, bool b_kc, bool vec, int M, int N, int K,
                              const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                              int k_per_split, int splits, int swizzle, int vecC, const GemmBatch* batch = nullptr);
// gemm_bf16x3_w8.hip: the form for a pre-split weight on the B side (A k-contiguous with 16-byte rows, no split over K)
hipError_t gemm_bf16x3_w8_launch(rgcn_ctx* c, int terms, int M, int N, int K, const float* A, int lda, float* C, int ldc,
                                 int swizzle, int vecC, const GemmBatch& batch);
size_t gemm_w8_timeline_bytes(int M, int N, int groups);

// ---- block_msgs.hip
rgcn_status block_geometry(rgcn_ctx* c);
rgcn_status block_msg_forward(rgcn_ctx* c, int layer, const float* Hin, float* Ybuf);
// Zbuf == nullptr: the per-relation weight gradients only (the single-pass layer kernel computes the row gradients)
rgcn_status block_msg_backward(rgcn_ctx* c, int layer, const float* Hin, const float* D, float* Zbuf);
rgcn_status block_dw_reduce(rgcn_ctx* c, int layer);
rgcn_status block_to_device_layout(rgcn_ctx* c, const float* host_layout_dev, float* dst, int R);
rgcn_status block_from_device_layout(rgcn_ctx* c, const float* src, float* host_layout_dev, int R);

struct CombineArgs;
// ---- block_rows.hip: the block layer destination-major in ONE pass per direction, one column band per XCD, weights
// through L2 (no LDS table, no message buffer); bitwise equal to the two-kernel form
bool block_rows_available(const rgcn_ctx* c);
size_t block_rows_weight_floats(const rgcn_ctx* c);
rgcn_status block_rows(rgcn_ctx* c, const char* tag, int layer, bool backward, const float* X, const CombineArgs& ca);

// ---- basis.hip
rgcn_status basis_aggregate_forward(rgcn_ctx* c, int layer, const float* Hin, float* Z);
struct CombineArgs;
rgcn_status basis_backward_gather(rgcn_ctx* c, int layer, const float* dZ, const CombineArgs& ca,
                                  bool with_messages);
rgcn_status basis_dcoef(rgcn_ctx* c, int layer, const float* Hin, const float* dZ);
double basis_units(rgcn_ctx* c);      // (row, direction) units of the current graph (profile accounting)
// Dc[dir][i][:] = D[row of unit i of direction dir][:]  (the compacted upstream rows, [2][V][d])
rgcn_status basis_gather_units(rgcn_ctx* c, const float* D, float* Dc);
rgcn_status basis_to_device_layout(rgcn_ctx* c, const float* host_layout_dev, float* dst);
rgcn_status basis_from_device_layout(rgcn_ctx* c, const float* src, float* host_layout_dev);

// ---- elementwise.hip
struct CombineArgs {
  float* out;            // primary output [V,d]
  float* out2;           // optional: out * dropout(drop2)
  const float* base;     // optional [V,d] (valid for rows in [row_lo,row_hi)); dropout `drop` applies
  const float* add;      // optional [V,d] added as is
  // optional (basis kind): the compacted products [2][V][d] of the units of each direction; row v receives the rows
  // unit_ptr[dir][v] of both directions it has a unit in (GraphBufs::unit_ptr, [2][V+1])
  const float* add_units = nullptr;
  const int32_t* unit_ptr = nullptr;
  const float* msg;      // optional message rows [slots,d]; summed per CSR row
  const int32_t* row_ptr;
  const int32_t* long_rows; // rows with more than kLongRow slots (GraphBufs::long_rows / nlong)
  const int32_t* nlong;
  // giant rows (GraphBufs; all null when the cut is off)
  const int32_t* giant_rows = nullptr;
  const int32_t* giant_first = nullptr;
  const int32_t* giant_cnt = nullptr;
  const int32_t* piece_row = nullptr;
  const int32_t* piece_k = nullptr;
  const int32_t* ngiant = nullptr;
  float* giant_slab = nullptr;  // [pieces][d]
  const float* gate;     // optional: result *= (gate > 0)
  int32_t V, d;
  int32_t relu;
  int32_t row_lo, row_hi;
  int32_t v_begin = 0, v_count = -1;   // rows the launch walks: [v_begin, v_begin + v_count) (v_count < 0: all V)
  int32_t colsum = 0;    // block_rows backward: also leave the column sums of `out` as partials (rgcn_ctx::colsum_parts)
  DropSpec drop;         // applied to base
  DropSpec drop2;        // applied to out2
};
rgcn_status combine(rgcn_ctx* c, const char* tag, const CombineArgs& a, double alg_bytes);
// giant rows only: prologue + the pieces in GraphBufs' piece slab (written by the caller's own kernel) + epilogue
rgcn_status combine_giant_finish(rgcn_ctx* c, const CombineArgs& a);
// part[i][:] = out[giant_rows[i]][:] for the graph's giant rows, zeros up to giant_cap rows (column sums: block_rows.hip)
rgcn_status column_sum_giant_rows(rgcn_ctx* c, const float* out, float* part);
// out[col] = the sum of the nparts partial rows in rgcn_ctx::colsum_part (k_colsum_final's fixed order)
rgcn_status column_sum_finish(rgcn_ctx* c, float* out, int nparts, int cols);
rgcn_status input_forward(rgcn_ctx* c);                      // H0 = relu(W_emb + b_emb)
rgcn_status scale_dropout(rgcn_ctx* c, const float* in, float* out, const DropSpec& ds);
rgcn_status column_sum(rgcn_ctx* c, const float* in, float* out, int rows, int cols);
rgcn_status relu_copy(rgcn_ctx* c, const float* in, float* out, int64_t n, int relu);
rgcn_status materialize_mask(rgcn_ctx* c, const DropSpec& ds, uint8_t* out_dev, int64_t n);

DropSpec make_drop(const rgcn_ctx* c, int layer, bool active);

// ---- decoder.hip / optimizer.hip
rgcn_status negative_sample(rgcn_ctx* c, const int32_t* batch_dev, int64_t n, int rate, uint64_t seed, int32_t* X,
                            float* Y);
rgcn_status decoder_reserve(rgcn_ctx* c, int64_t max_triples);
void decoder_free(rgcn_ctx* c);
// N_total > N: X_dev is one rank's slice of a batch of N_total triples (relation-sharded train step); 0: N
rgcn_status decoder_prepare(rgcn_ctx* c, const int32_t* X_dev, int64_t N, int64_t N_total = 0);
// dcodes_drop / drop (optional): also write dL/dcodes * drop, the dropout-scaled copy the encoder's top layer consumes
rgcn_status decoder_compute(rgcn_ctx* c, const float* codes, const float* Y_dev, float reg_param,
                            float* dcodes_drop = nullptr, const DropSpec* drop = nullptr);
rgcn_status decoder_allreduce(rgcn_ctx* c);      // sharded run: sum the per-rank partial decoder results
rgcn_status optimizer_step(rgcn_ctx* c);
rgcn_status optimizer_norm_partial(rgcn_ctx* c);
rgcn_status optimizer_apply(rgcn_ctx* c);
// ---- neighborhood.hip: sample_edge_neighborhood on the device (parallel first-passage percolation)
rgcn_status neighborhood_reserve(rgcn_ctx* c, const int32_t* triples_host, int64_t n);
rgcn_status neighborhood_sample(rgcn_ctx* c, int64_t sample_size, uint64_t seed, int32_t* batch_out_dev,
                                bool on_prefetch_stream);
void neighborhood_free(rgcn_ctx* c);
// ---- ranking.hip
rgcn_status rank_reserve(rgcn_ctx* c, int64_t max_queries);
void rank_free(rgcn_ctx* c);
rgcn_status rank_compute(rgcn_ctx* c, const int32_t* X_dev, int64_t N, int predict_object, const int64_t* filt_ptr,
                         const int32_t* filt_idx, int32_t* raw_out, int32_t* filt_out);
void optimizer_free(rgcn_ctx* c);

// ---- comm.cpp (RCCL via dlopen)
rgcn_status comm_unique_id(uint8_t id[128]);
rgcn_status comm_init(rgcn_ctx* c, const uint8_t id[128]);
rgcn_status comm_allreduce(rgcn_ctx* c, float* buf, int64_t count);
rgcn_status comm_reduce_scatter(rgcn_ctx* c, float* buf, int64_t count_per_rank);
rgcn_status comm_all_gather(rgcn_ctx* c, float* buf, int64_t count_per_rank);
rgcn_status comm_info(rgcn_ctx* c, int32_t* ranks, int32_t* rank, int32_t* device);
void comm_destroy(rgcn_ctx* c);

}  // namespace rgcn
