// C ABI of librgcn.so (include/rgcn.h): contexts, parameters, graphs, steps, capture, profile.  The per-layer orchestration
// of the encoder (which kernel on which stream) lives in rgcn_schedule.hip, the stand-alone entry points of the devtools
// build in rgcn_devtools.hip; rgcn_api_internal.h is what the three share.
#include <cstdlib>
#include <mutex>
#include <utility>
#include <new>

#include "rgcn_api_internal.h"

namespace rgcn {

static std::mutex g_err_mu;
static std::string g_err;
void set_global_error(const std::string& s) {
  std::lock_guard<std::mutex> lk(g_err_mu);
  g_err = s;
}

// ---------------------------------------------------------------- profiling
ProfScope::ProfScope(rgcn_ctx* ctx, const char* name, double bytes, double flops, double compulsory) : c(ctx), idx(-1) {
  if (!c->prof_on) return;
  ProfRec r;
  r.name = name;
  r.bytes = bytes;
  r.flops = flops;
  r.cbytes = compulsory < 0 ? bytes : compulsory;
  for (int k = 0; k < 2; ++k) {
    hipEvent_t e = nullptr;
    if (!c->event_pool.empty()) {
      e = c->event_pool.back();
      c->event_pool.pop_back();
    } else if (hipEventCreate(&e) != hipSuccess) {
      return;
    }
    (k == 0 ? r.e0 : r.e1) = e;
  }
  (void)hipEventRecord(r.e0, c->stream);
  c->prof.push_back(r);
  idx = (int)c->prof.size() - 1;
}
ProfScope::~ProfScope() {
  if (idx >= 0) (void)hipEventRecord(c->prof[idx].e1, c->stream);
}

StreamScope::StreamScope(rgcn_ctx* ctx, int k, bool in_capture) : c(ctx), saved(ctx->stream), active(false) {
  const bool on = c->use_aux || (in_capture && c->capturing && c->use_aux_before_capture && c->world == 1);
  if (k < 0 || !on || c->stream != c->main_stream) return;   // nested or disabled: stay on the current stream
  if (hipEventRecord(c->ev_fork, c->main_stream) != hipSuccess) return;
  if (hipStreamWaitEvent(c->aux[k], c->ev_fork, 0) != hipSuccess) return;
  c->stream = c->aux[k];
  c->aux_dirty[k] = true;
  active = true;
}
StreamScope::~StreamScope() { c->stream = saved; }

rgcn_status stream_join(rgcn_ctx* c, int k) {
  if (c->stream != c->main_stream) return RGCN_OK;
  // a join costs the main stream ~3.6 us even when the side stream is idle (tools/anyorder_probe.hip): skip it when
  // nothing went to that stream since its last join (side streams switched off, or no fork taken)
  if (!c->aux_dirty[k]) return RGCN_OK;
  c->aux_dirty[k] = false;
  RGCN_HIP(c, hipEventRecord(c->ev_join[k], c->aux[k]));
  RGCN_HIP(c, hipStreamWaitEvent(c->main_stream, c->ev_join[k], 0));
  return RGCN_OK;
}

// Side streams 0 and 1 behind ONE wait of the main stream (side 0 waits for side 1 first): where both were forked in the
// same layer, the second wait is a packet the main stream can do without.
rgcn_status stream_join_both(rgcn_ctx* c) {
  if (c->stream != c->main_stream) return RGCN_OK;
  if (c->aux_dirty[0] && c->aux_dirty[1]) {
    RGCN_HIP(c, hipEventRecord(c->ev_join[1], c->aux[1]));
    RGCN_HIP(c, hipStreamWaitEvent(c->aux[0], c->ev_join[1], 0));
    c->aux_dirty[1] = false;
  }
  RGCN_TRY(stream_join(c, 0));
  return stream_join(c, 1);
}

// A backward pass driven layer by layer (phase API) and abandoned between layers 2 and 1 leaves layer 2's side kernels
// unjoined (bwd_layer_partial defers their joins to layer 1).  They read D / dS / the activations and the graph's message
// lists: whoever is about to rewrite any of those -- the next forward pass, the next backward pass, a graph build on the
// main stream -- joins them first.  Nothing is queued (and nothing is paid) in the usual case, where every pass ended joined.
rgcn_status join_abandoned_side_work(rgcn_ctx* c) {
  if (c->stream != c->main_stream) return RGCN_OK;
  RGCN_TRY(stream_join(c, 0));
  return stream_join(c, 1);
}

static rgcn_status sync_all(rgcn_ctx* c) {
  if (c->capturing) RGCN_FAIL(c, RGCN_ERR_STATE, "this call synchronises with the device: not allowed between rgcn_capture_begin and rgcn_capture_end");
  if (c->pf_stream) RGCN_HIP(c, hipStreamSynchronize(c->pf_stream));
  for (int k = 0; k < kAuxStreams; ++k)
    if (c->aux[k]) RGCN_HIP(c, hipStreamSynchronize(c->aux[k]));
  RGCN_HIP(c, hipStreamSynchronize(c->main_stream));
  for (int k = 0; k < kAuxStreams; ++k) c->aux_dirty[k] = false;     // nothing left to join
  return RGCN_OK;
}

static rgcn_status profile_collect(rgcn_ctx* c) {
  RGCN_TRY(sync_all(c));
  for (ProfRec& r : c->prof) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.e0, r.e1);
    ProfAgg* a = nullptr;
    for (ProfAgg& x : c->prof_agg)
      if (x.name == r.name) { a = &x; break; }
    if (!a) {
      c->prof_agg.push_back({r.name, 0, 0.0, 0.0, 0.0, 0.0});
      a = &c->prof_agg.back();
    }
    a->calls += 1;
    a->ms += ms;
    a->bytes += r.bytes;
    a->flops += r.flops;
    a->cbytes += r.cbytes;
    c->event_pool.push_back(r.e0);
    c->event_pool.push_back(r.e1);
  }
  c->prof.clear();
  return RGCN_OK;
}

// ---------------------------------------------------------------- helpers

static void add_param(rgcn_ctx* c, const std::string& name, std::initializer_list<int64_t> shape,
                      float* val, float* grad, int layout) {
  Param p;
  p.name = name;
  p.ndim = (int)shape.size();
  p.count = 1;
  int i = 0;
  for (int64_t s : shape) { p.shape[i++] = s; p.count *= s; }
  for (; i < 4; ++i) p.shape[i] = 1;
  p.val = val;
  p.grad = grad;
  p.layout = layout;
  c->params.push_back(p);
}

// target: workgroups of a split-K launch.  Since the dW_self GEMM runs beside the dH GEMM (the backward layer's pairing)
// FEWER workgroups pay in the step -- headline 0.555-0.562 ms per step at 256, 0.558-0.568 at 384, 0.565-0.568 at 448,
// 0.567-0.575 at 512 -- but cost the GEMM itself (alone on the chip 54.6 us at 256, 51 at 448, 48.7 at 512: one workgroup
// per CU).  The headline shape keeps 512 (the step's gain is ~1 %, inside the box-to-box spread);
// `narrow` -- 256 -- where the step gains 3-5 %: K >= 32,768 rows (WN18 sizes: 1.128 ms against 1.166) and the
// basis kind (B = 2: 1.64 against 1.72).  profiles/r04_*: splitk A/B.  The rule reads the model's
// dimensions only, one figure for every form of the layer: the split decides the summation order of dW_self, and the forms
// are held bitwise equal to each other.
int auto_split_k(int M, int N, int K, bool narrow) {
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  if (tiles >= 192) return 1;
  const int target = narrow ? 256 : 512;
  int s = (target + tiles - 1) / tiles;
  const int max_by_k = (K + 127) / 128;   // at least 128 of K per slab
  if (s > max_by_k) s = max_by_k;
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  return s;
}

// main_only: wait for the main stream alone (everything a finished step depends on was joined into it); the
// graph preparation of the NEXT minibatch, running on the prefetch stream, is left alone and its id check
// surfaces at the next synchronising call.
static rgcn_status check_dev_flag(rgcn_ctx* c, bool main_only = false) {
  if (c->capturing) RGCN_FAIL(c, RGCN_ERR_STATE, "this call synchronises with the device: not allowed between rgcn_capture_begin and rgcn_capture_end");
  int32_t flag = 0;
  if (!main_only) {
    if (c->pf_stream) RGCN_HIP(c, hipStreamSynchronize(c->pf_stream));
    for (int k = 0; k < kAuxStreams; ++k)
      if (c->aux[k]) RGCN_HIP(c, hipStreamSynchronize(c->aux[k]));
  }
  RGCN_HIP(c, hipMemcpyAsync(&flag, c->g.errflag, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  RGCN_HIP(c, hipStreamSynchronize(c->stream));
  if (flag) {
    RGCN_HIP(c, hipMemsetAsync(c->g.errflag, 0, sizeof(int32_t), c->stream));
    RGCN_FAIL(c, RGCN_ERR_INVALID, std::string(flag & 4 ? "edge dropout: the keep mask does not hold exactly `keep` ones; " : "") +
                                   (flag & 8 ? "device neighbourhood sampler: the relaxations did not settle within their iteration budget "
                                               "(a graph of very large diameter: use the host sampler); " : "") +
                                   (flag & 32 ? "device neighbourhood sampler: the compacted batch does not hold the requested number "
                                                "of edges (internal error: two draws sharing the sampler's state?); " : "") +
                                   (flag & 16 ? "the decoder batch is no longer the tiled batch rgcn_negative_sample_device wrote into that "
                                                "buffer (rewritten by the caller's own kernels?): pass it in another buffer; " : "") +
                                   (flag & 3 ? "graph_edges / the decoder batch contains a vertex id outside [0,EntityCount) or a "
                                               "relation id outside [0,RelationCount)" : ""));
  }
  return RGCN_OK;
}

// A caller that rewrites (or frees) the triple buffer a prefetched graph was prepared from makes that
// preparation stale: drop it, the next step rebuilds in line.
static void invalidate_prefetch_of(rgcn_ctx* c, const void* dev, size_t bytes) {
  // a caller that rewrites the buffer the negative sampler tiled makes the decoder's knowledge of its layout stale
  if (c->dec.tiled_X) {
    const char* t = reinterpret_cast<const char*>(c->dec.tiled_X);
    const char* lo = reinterpret_cast<const char*>(dev);
    if (t + sizeof(int32_t) * 3 * (size_t)c->dec.tiled_N > lo && t < lo + (bytes ? bytes : 1)) c->dec.tiled_X = nullptr;
  }
  for (GraphBufs* g : {&c->g, &c->g_alt}) {
    if (!g->pf_valid || !g->pf_tri) continue;
    const char* t = reinterpret_cast<const char*>(g->pf_tri);
    const char* lo = reinterpret_cast<const char*>(dev);
    if (t + sizeof(int32_t) * 3 * (size_t)g->pf_E > lo && t < lo + (bytes ? bytes : 1)) g->pf_valid = false;
  }
}

rgcn_status to_host(rgcn_ctx* c, void* host, const void* dev, size_t bytes) {
  if (c->capturing) RGCN_FAIL(c, RGCN_ERR_STATE, "host transfers are not allowed while a hipGraph is being captured");
  RGCN_HIP(c, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
  RGCN_HIP(c, hipStreamSynchronize(c->stream));
  return RGCN_OK;
}
rgcn_status to_dev(rgcn_ctx* c, void* dev, const void* host, size_t bytes) {
  if (c->capturing) RGCN_FAIL(c, RGCN_ERR_STATE, "host transfers are not allowed while a hipGraph is being captured");
  RGCN_HIP(c, hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->stream));
  RGCN_HIP(c, hipStreamSynchronize(c->stream));   // host memory is borrowed for the call only
  return RGCN_OK;
}

// Process-wide stream pool.  ROCm hands a new stream its hardware queue round-robin, and objects that outlive a
// context (an instantiated hipGraph with a forked branch keeps internal streams) shift that assignment: a context
// created LATER in the same process could find two of its four streams on one hardware queue, and its stream-launched
// step ran at half speed with the GPU idle between kernels (seen in bench.py: the second train-step workload at 2.7 ms
// against 1.3 ms in a fresh process, graph replay unaffected).  Streams are therefore kept when a context is destroyed
// and handed to the next one on that device: every context of a process runs on the streams -- and the queue
// assignment -- of the first.  Key: (device, priority).  Never destroyed (the runtime goes first at exit).
namespace {
struct PooledStream { int device; int priority; hipStream_t s; };
struct StreamPool {
  std::mutex mu;
  std::vector<PooledStream> free_streams;
};
StreamPool& stream_pool() {          // allocated once, never destroyed: usable from whatever runs at process exit
  static StreamPool* pool = new StreamPool();
  return *pool;
}
}  // namespace
static hipError_t acquire_stream(int device, int priority, hipStream_t* out) {
  {
    StreamPool& sp = stream_pool();
    std::lock_guard<std::mutex> lk(sp.mu);
    for (size_t i = 0; i < sp.free_streams.size(); ++i)
      if (sp.free_streams[i].device == device && sp.free_streams[i].priority == priority) {
        *out = sp.free_streams[i].s;
        sp.free_streams.erase(sp.free_streams.begin() + (long)i);
        return hipSuccess;
      }
  }
  return hipStreamCreateWithPriority(out, hipStreamNonBlocking, priority);
}
static void release_stream(int device, int priority, hipStream_t s) {
  if (!s) return;
  StreamPool& sp = stream_pool();
  std::lock_guard<std::mutex> lk(sp.mu);
  sp.free_streams.push_back(PooledStream{device, priority, s});
}

static void free_all(rgcn_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->cfg.device);
  if (c->capturing && c->main_stream) {      // destroyed in mid-capture: the streams go back to the pool in a usable state
    hipGraph_t dangling = nullptr;
    (void)hipStreamEndCapture(c->main_stream, &dangling);
    if (dangling) (void)hipGraphDestroy(dangling);
    (void)hipGetLastError();
    c->capturing = false;
  }
  if (c->pf_stream) (void)hipStreamSynchronize(c->pf_stream);
  for (int k = 0; k < kAuxStreams; ++k)
    if (c->aux[k]) (void)hipStreamSynchronize(c->aux[k]);
  if (c->main_stream) (void)hipStreamSynchronize(c->main_stream);
  comm_destroy(c);
  graph_free(c);
  auto F = [](void* p) { if (p) (void)hipFree(p); };
  decoder_free(c);
  neighborhood_free(c);
  optimizer_free(c);
  rank_free(c);
  if (c->giant_slab) (void)hipFree(c->giant_slab);
  F(c->w_emb); F(c->g_emb); F(c->b_emb); F(c->gb_emb); F(c->w_rel); F(c->g_rel);
  for (LayerBufs& lb : c->layers) {
    if (c->repl_grads) {      // views into repl_grads
      lb.gwself = nullptr;
      if (c->kind == RGCN_KIND_BASIS) lb.grel = nullptr;
    }
    F(lb.wrel); F(lb.grel); F(lb.coef); F(lb.gcoef); F(lb.wself); F(lb.gwself); F(lb.bias); F(lb.gbias); F(lb.wtile);
    F(lb.wself_nn); F(lb.wself_nt); F(lb.wrel_nn); F(lb.wrel_nt);
  }
  for (float* h : c->H) F(h);
  F(c->self_buf); F(c->exch); F(c->dbuf[0]); F(c->dbuf[1]); F(c->dsbuf[0]); F(c->dsbuf[1]);
  F(c->msgbuf); F(c->msgbuf2); F(c->slab); F(c->slab_dw); F(c->aggbuf);
  for (float* z : c->zsave) F(z); F(c->stage); F(c->masks); F(c->colsum_part); F(c->dcodes_own); F(c->zeros);
  for (ProfRec& r : c->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
  if (c->t0) (void)hipEventDestroy(c->t0);
  if (c->t1) (void)hipEventDestroy(c->t1);
  for (int k = 0; k < kAuxStreams; ++k) {
    if (k < 2) release_stream(c->cfg.device, c->aux_priority, c->aux[k]);     // aux[2] is the prefetch stream, below
    if (c->ev_join[k]) (void)hipEventDestroy(c->ev_join[k]);
  }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_gather) (void)hipEventDestroy(c->ev_gather);
  F(c->repl_grads);
  if (c->ev_step_begin) (void)hipEventDestroy(c->ev_step_begin);
  for (hipGraphExec_t g : c->graphs) if (g) (void)hipGraphExecDestroy(g);
  for (hipGraph_t g : c->graph_defs) if (g) (void)hipGraphDestroy(g);
  if (c->replay_counter) (void)hipFree(c->replay_counter);
  if (c->readback_host) (void)hipHostFree(c->readback_host);
  if (c->stage_host) {
    (void)hipHostFree(c->stage_host);
    for (hipEvent_t e : c->stage_done) if (e) (void)hipEventDestroy(e);
  }
  release_stream(c->cfg.device, c->pf_priority, c->pf_stream);
  release_stream(c->cfg.device, 0, c->main_stream);
  delete c;
}

static rgcn_status create_impl(rgcn_ctx* c) {
  const rgcn_config& f = c->cfg;
  if (f.abi_version != RGCN_ABI_VERSION) RGCN_FAIL(c, RGCN_ERR_INVALID, "abi_version mismatch");
  if (f.num_entities <= 0 || f.num_relations <= 0 || f.dim <= 0 || f.num_layers <= 0 || f.num_bases <= 0)
    RGCN_FAIL(c, RGCN_ERR_INVALID, "EntityCount, RelationCount, dimension, NumberOfLayers and "
                                   "NumberOfBasisFunctions must be positive");
  if (!(f.keep_prob > 0.0f && f.keep_prob <= 1.0f)) RGCN_FAIL(c, RGCN_ERR_INVALID, "DropoutKeepProbability must be in (0,1]");
  if (f.kind != RGCN_KIND_BLOCK && f.kind != RGCN_KIND_BASIS) RGCN_FAIL(c, RGCN_ERR_INVALID, "unknown kind");
  if (f.norm_mode < 0 || f.norm_mode > 2) RGCN_FAIL(c, RGCN_ERR_INVALID, "unknown norm_mode");
  if (f.world < 1 || f.rank < 0 || f.rank >= f.world) RGCN_FAIL(c, RGCN_ERR_INVALID, "need 0 <= rank < world");
  if (f.max_edges < 0 || f.max_edges > (int64_t)500 * 1000 * 1000) RGCN_FAIL(c, RGCN_ERR_INVALID, "max_edges out of range");
  if (f.reserved != 0) RGCN_FAIL(c, RGCN_ERR_INVALID, "reserved must be 0");
  c->V = f.num_entities; c->R = f.num_relations; c->d = f.dim; c->L = f.num_layers; c->kind = f.kind;
  c->rank = f.rank; c->world = f.world;
  // the library's radix sort takes keys below 2^24 (csr_sort.hip): vertex ids and directed-relation ids
  if (c->V >= (1 << 24) || 2 * (int64_t)c->R >= (1 << 24))
    RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "EntityCount and 2 x RelationCount must be below 2^24 (sort key range of this build)");
  if ((int64_t)c->V * c->d > (int64_t)1 << 31 || 2 * f.max_edges * (int64_t)c->d > ((int64_t)1 << 40))
    RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "problem too large for this build");
  // messages per relation chunk: 48 at minibatch scale; grows with the capacity so that a full-graph
  // context does not cut a popular relation into thousands of chunks
  if (2 * f.max_edges > 65536) c->chunk = 48 * (int)((2 * f.max_edges + 65535) / 65536);
  if (c->kind == RGCN_KIND_BLOCK) {
    c->nb = f.num_bases;
    if (c->d % c->nb != 0)
      RGCN_FAIL(c, RGCN_ERR_INVALID, "InternalEncoderDimension must be divisible by NumberOfBasisFunctions "
                                     "(the reference silently mis-groups otherwise: gcn_basis_concat.py:15,42)");
    c->sd = c->d / c->nb;
    RGCN_TRY(block_geometry(c));
  } else {
    c->B = f.num_bases;
    if (c->B > 64) RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "NumberOfBasisFunctions > 64 (basis)");
  }
  // equal row chunks (the reduce-scatter / all-gather want them equal): rank g finishes rows [g * shard_rows, +shard_rows)
  c->shard_rows = (c->V + c->world - 1) / c->world;
  c->V_pad = c->shard_rows * c->world;
  c->row_lo = std::min(c->V, c->rank * c->shard_rows);
  c->row_hi = std::min(c->V, c->row_lo + c->shard_rows);

  RGCN_HIP(c, hipSetDevice(f.device));
  RGCN_HIP(c, acquire_stream(f.device, 0, &c->stream));
  c->main_stream = c->stream;
  {
    // side streams get the highest priority: their short HBM-bound kernels should claim wave slots
    // ahead of the long MFMA-bound grid they run beside
    int prio_lo = 0, prio_hi = 0;
    RGCN_HIP(c, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    // Two high-priority side streams; "side stream 2" is the prefetch stream (created here, normal or low priority):
    // both of its jobs -- the next minibatch's graph prep and the decoder's relation gradient -- are fillers.  A
    // fifth HIP stream would share a hardware queue with one of the other four (ROCm maps streams onto 4 queues by
    // default) and serialise against it: measured, the pipelined encoder step went from 0.60 to 0.97 ms.
    c->aux_priority = prio_hi;
    for (int k = 0; k < 2; ++k) RGCN_HIP(c, acquire_stream(f.device, c->aux_priority, &c->aux[k]));
    c->pf_priority = prio_lo;
    RGCN_HIP(c, acquire_stream(f.device, c->pf_priority, &c->pf_stream));
    c->aux[2] = c->pf_stream;
    for (int k = 0; k < kAuxStreams; ++k)
      RGCN_HIP(c, hipEventCreateWithFlags(&c->ev_join[k], order_event_flags(c, true)));
  }
  RGCN_HIP(c, hipEventCreateWithFlags(&c->ev_fork, order_event_flags(c, true)));
  RGCN_HIP(c, hipEventCreateWithFlags(&c->ev_step_begin, order_event_flags(c)));
  // Block kind: the destination-major banded single pass (block_rows.hip; rgcn_set_fusion 1, the default) -- no message
  // buffer, weights through L2: 39-41 us forward / 46-48 backward per layer against 57 / 75 for the two-kernel form
  // (rgcn_set_fusion 0: k_block_msg_* + k_combine, the form every other is held bitwise equal to) at FB15k-237 minibatch
  // size, 0.39 ms against 0.77 per layer pass at the 272,115-edge training graph (profiles/r04_rowmajor_spmm_ab.md).
  // Two more forms were built, measured slower and removed in round 5: the combine as the self-loop GEMM's epilogue
  // (profiles/r02_fused_layer_ab.log) and per-block workgroups with an LDS weight table (profiles/r03_block_spmm_ab.md).
  c->use_aux = true;
  c->fuse = 1;
  c->gemm_mode = 6;
  RGCN_HIP(c, hipEventCreate(&c->t0));
  RGCN_HIP(c, hipEventCreate(&c->t1));

  // every [V,d] buffer carries V_pad rows (== V on one GPU): the collectives work on world equal chunks, the
  // padding rows stay zero
  const size_t V = c->V, d = c->d, R = c->R, Vd = (size_t)c->V_pad * d;
  RGCN_HIP(c, hipEventCreateWithFlags(&c->ev_gather, hipEventDisableTiming));
  RGCN_TRY(dmalloc(c, &c->w_emb, Vd));
  RGCN_TRY(dmalloc(c, &c->g_emb, Vd));
  RGCN_TRY(dmalloc(c, &c->b_emb, d));
  RGCN_TRY(dmalloc(c, &c->gb_emb, d));
  add_param(c, "W_emb", {(int64_t)V, (int64_t)d}, c->w_emb, c->g_emb, LAYOUT_PLAIN);
  add_param(c, "b_emb", {(int64_t)d}, c->b_emb, c->gb_emb, LAYOUT_PLAIN);
  c->layers.resize(c->L + 1);
  // world > 1: the gradients every rank holds a partial sum of -- W_self, and the basis tensors -- sit in ONE
  // allocation, layer after layer, so that the backward pass ends with one all-reduce instead of 2-3 per layer
  const size_t repl_per_layer = d * d + (c->kind == RGCN_KIND_BASIS ? 2 * (size_t)f.num_bases * d * d : 0);
  if (c->world > 1) {
    c->repl_grads_floats = repl_per_layer * c->L;
    RGCN_TRY(dmalloc(c, &c->repl_grads, c->repl_grads_floats));
  }
  for (int l = 1; l <= c->L; ++l) {
    LayerBufs& lb = c->layers[l];
    const std::string sl = std::to_string(l);
    float* repl = c->repl_grads ? c->repl_grads + repl_per_layer * (l - 1) : nullptr;
    if (c->kind == RGCN_KIND_BLOCK) {
      const size_t per_dir = R * c->nb * c->sd * c->sd;
      RGCN_TRY(dmalloc(c, &lb.wrel, 2 * per_dir));
      RGCN_TRY(dmalloc(c, &lb.grel, 2 * per_dir));
      add_param(c, "W_f" + sl, {(int64_t)R, c->nb, c->sd, c->sd}, lb.wrel, lb.grel, LAYOUT_BLOCK_T);
      add_param(c, "W_b" + sl, {(int64_t)R, c->nb, c->sd, c->sd}, lb.wrel + per_dir, lb.grel + per_dir, LAYOUT_BLOCK_T);
      if (c->nb <= 512) RGCN_TRY(dmalloc(c, &lb.wtile, block_rows_weight_floats(c)));
    } else {
      const size_t per_dir = (size_t)c->B * d * d;
      RGCN_TRY(dmalloc(c, &lb.wrel, 2 * per_dir));
      if (repl) lb.grel = repl + d * d;
      else RGCN_TRY(dmalloc(c, &lb.grel, 2 * per_dir));
      RGCN_TRY(dmalloc(c, &lb.coef, 2 * R * c->B));
      RGCN_TRY(dmalloc(c, &lb.gcoef, 2 * R * c->B));
      add_param(c, "W_f" + sl, {(int64_t)d, c->B, (int64_t)d}, lb.wrel, lb.grel, LAYOUT_BASIS_T);
      add_param(c, "W_b" + sl, {(int64_t)d, c->B, (int64_t)d}, lb.wrel + per_dir, lb.grel + per_dir, LAYOUT_BASIS_T);
      add_param(c, "C_f" + sl, {(int64_t)R, c->B}, lb.coef, lb.gcoef, LAYOUT_PLAIN);
      add_param(c, "C_b" + sl, {(int64_t)R, c->B}, lb.coef + R * c->B, lb.gcoef + R * c->B, LAYOUT_PLAIN);
    }
    RGCN_TRY(dmalloc(c, &lb.wself, d * d));
    {      // fragment tables (allocated here: a capture may be the first call that needs them)
      const size_t ws = 16 * gemm_bfrag_words((int)d, (int)d);
      RGCN_HIP(c, hipMalloc(&lb.wself_nn, ws));
      RGCN_HIP(c, hipMalloc(&lb.wself_nt, ws));
      if (c->kind == RGCN_KIND_BASIS) {
        const int Bd = c->B * (int)d;
        RGCN_HIP(c, hipMalloc(&lb.wrel_nn, 2 * 16 * gemm_bfrag_words(Bd, (int)d)));
        RGCN_HIP(c, hipMalloc(&lb.wrel_nt, 2 * 16 * gemm_bfrag_words((int)d, Bd)));
      }
    }
    if (repl) lb.gwself = repl;
    else RGCN_TRY(dmalloc(c, &lb.gwself, d * d));
    RGCN_TRY(dmalloc(c, &lb.bias, d));
    RGCN_TRY(dmalloc(c, &lb.gbias, d));
    add_param(c, "W_self" + sl, {(int64_t)d, (int64_t)d}, lb.wself, lb.gwself, LAYOUT_PLAIN);
    add_param(c, "b" + sl, {(int64_t)d}, lb.bias, lb.gbias, LAYOUT_PLAIN);
    c->params.back().no_grad = true;
  }
  RGCN_TRY(dmalloc(c, &c->w_rel, Vd));
  RGCN_TRY(dmalloc(c, &c->g_rel, Vd));
  add_param(c, "W_relation", {(int64_t)V, (int64_t)d}, c->w_rel, c->g_rel, LAYOUT_PLAIN);
  c->H.assign(c->L + 1, nullptr);
  for (int l = 0; l <= c->L; ++l) RGCN_TRY(dmalloc(c, &c->H[l], Vd));
  RGCN_TRY(dmalloc(c, &c->self_buf, Vd));
  if (c->world > 1) RGCN_TRY(dmalloc(c, &c->exch, Vd));
  for (int k = 0; k < 2; ++k) {
    RGCN_TRY(dmalloc(c, &c->dbuf[k], Vd));
    RGCN_TRY(dmalloc(c, &c->dsbuf[k], Vd));
  }
  const size_t M = 2 * (size_t)f.max_edges;
  size_t slab = 64 * d * d;   // split-K slabs of the dW_self GEMM
  // most relation chunks any graph that fits the context can have: the chunk size follows the graph's own size
  // (graph_build), so a graph of 65536 messages cut at 48 can have more chunks than the largest graph cut coarser
  const size_t max_rel_chunks = std::max((M + c->chunk - 1) / c->chunk, (std::min<size_t>(M, 65536) + 47) / 48) + 2 * R;
  if (c->kind == RGCN_KIND_BLOCK) {
    RGCN_TRY(dmalloc(c, &c->msgbuf, (M ? M : 1) * d, false));
    const size_t per_rel = (size_t)c->sd * c->sd * c->nb;
    c->slab_dw_floats = max_rel_chunks * per_rel;
    RGCN_TRY(dmalloc(c, &c->slab_dw, c->slab_dw_floats, false));
  } else {
    const size_t zc = 2 * (size_t)c->B * d;
    RGCN_TRY(dmalloc(c, &c->msgbuf2, V * zc));
    RGCN_TRY(dmalloc(c, &c->aggbuf, 2 * V * d));     // [2][V][d]: unit products (forward), gathered upstream rows (backward)
    c->zsave.assign(c->L + 1, nullptr);
    for (int l = 1; l <= c->L; ++l) RGCN_TRY(dmalloc(c, &c->zsave[l], V * zc));
    const size_t s2 = 16 * zc * d;
    if (s2 > slab) slab = s2;
    c->slab_dw_floats = max_rel_chunks * (size_t)c->B;
    RGCN_TRY(dmalloc(c, &c->slab_dw, c->slab_dw_floats, false));
  }
  c->slab_floats = slab;
  RGCN_TRY(dmalloc(c, &c->slab, slab));
  size_t stage = Vd;
  for (const Param& p : c->params)
    if ((size_t)p.count > stage) stage = (size_t)p.count;
  c->stage_floats = stage;
  RGCN_TRY(dmalloc(c, &c->stage, stage, false));
  // one partial row per combine workgroup (4 rows each at worst) + the second-level partials of their sum
  c->colsum_part_floats = ((V + 3) / 4 + 1024 + 2 + ((V + 3) / 4 + 1024) / 32 + 2) * d;
  RGCN_TRY(dmalloc(c, &c->colsum_part, c->colsum_part_floats));
  RGCN_TRY(dmalloc(c, &c->zeros, 1024));
  RGCN_TRY(graph_alloc(c, nullptr));
  std::swap(c->g, c->g_alt);
  RGCN_TRY(graph_alloc(c, &c->g_alt));
  std::swap(c->g, c->g_alt);
  {
    std::vector<int32_t> owner(c->R);
    for (int r = 0; r < c->R; ++r) owner[r] = r % c->world;
    RGCN_TRY(to_dev(c, c->g.owner, owner.data(), sizeof(int32_t) * owner.size()));
  }
  RGCN_HIP(c, hipStreamSynchronize(c->stream));
  return RGCN_OK;
}

// ---------------------------------------------------------------- layout conversion
static rgcn_status param_upload(rgcn_ctx* c, const Param& p, float* dst, const float* host) {
  c->weights_version += 1;       // derived copies (the block-major weight tables) are stale now
  if (p.layout == LAYOUT_PLAIN) return to_dev(c, dst, host, sizeof(float) * p.count);
  RGCN_TRY(to_dev(c, c->stage, host, sizeof(float) * p.count));
  if (p.layout == LAYOUT_BLOCK_T) RGCN_TRY(block_to_device_layout(c, c->stage, dst, c->R));
  else RGCN_TRY(basis_to_device_layout(c, c->stage, dst));
  RGCN_HIP(c, hipStreamSynchronize(c->stream));
  return RGCN_OK;
}
static rgcn_status param_download(rgcn_ctx* c, const Param& p, const float* src, float* host) {
  if (p.layout == LAYOUT_PLAIN) return to_host(c, host, src, sizeof(float) * p.count);
  if (p.layout == LAYOUT_BLOCK_T) RGCN_TRY(block_from_device_layout(c, src, c->stage, c->R));
  else RGCN_TRY(basis_from_device_layout(c, src, c->stage));
  return to_host(c, host, c->stage, sizeof(float) * p.count);
}

}  // namespace rgcn

using namespace rgcn;

// ================================================================= C ABI

extern "C" {

int32_t rgcn_abi_version(void) { return RGCN_ABI_VERSION; }

rgcn_status rgcn_create(const rgcn_config* cfg, rgcn_ctx** out) {
  if (!cfg || !out) { set_global_error("rgcn_create: NULL argument"); return RGCN_ERR_INVALID; }
  *out = nullptr;
  rgcn_ctx* c = new (std::nothrow) rgcn_ctx();
  if (!c) { set_global_error("out of host memory"); return RGCN_ERR_NOMEM; }
  c->cfg = *cfg;
  rgcn_status s = create_impl(c);
  if (s != RGCN_OK) {
    set_global_error(c->err);
    free_all(c);
    return s;
  }
  *out = c;
  return RGCN_OK;
}

rgcn_status rgcn_destroy(rgcn_ctx* ctx) {
  if (!ctx) return RGCN_OK;
  free_all(ctx);
  return RGCN_OK;
}

const char* rgcn_last_error(const rgcn_ctx* ctx) {
  if (ctx) return ctx->err.c_str();
  static thread_local std::string copy;
  std::lock_guard<std::mutex> lk(g_err_mu);
  copy = g_err;
  return copy.c_str();
}

rgcn_status rgcn_sync(rgcn_ctx* c) {
  RGCN_NEED(c);
  return check_dev_flag(c);
}

int32_t rgcn_param_count(const rgcn_ctx* c) { return c ? (int32_t)c->params.size() : 0; }

rgcn_status rgcn_param_info(const rgcn_ctx* c, int32_t index, char* name, int32_t name_cap,
                            int64_t shape[4], int32_t* ndim) {
  if (!c || index < 0 || index >= (int32_t)c->params.size()) return RGCN_ERR_INVALID;
  const Param& p = c->params[index];
  if (name && name_cap > 0) {
    strncpy(name, p.name.c_str(), (size_t)name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = p.shape[i];
  if (ndim) *ndim = p.ndim;
  return RGCN_OK;
}

static rgcn_status param_check(rgcn_ctx* c, int32_t index, const void* host, int64_t count) {
  if (index < 0 || index >= (int32_t)c->params.size()) RGCN_FAIL(c, RGCN_ERR_INVALID, "parameter index out of range");
  if (!host) RGCN_FAIL(c, RGCN_ERR_INVALID, "NULL host pointer");
  if (count != c->params[index].count)
    RGCN_FAIL(c, RGCN_ERR_INVALID, "element count does not match parameter " + c->params[index].name);
  return RGCN_OK;
}

rgcn_status rgcn_set_param(rgcn_ctx* c, int32_t index, const float* host, int64_t count) {
  RGCN_NEED(c);
  RGCN_TRY(param_check(c, index, host, count));
  return param_upload(c, c->params[index], c->params[index].val, host);
}
rgcn_status rgcn_get_param(rgcn_ctx* c, int32_t index, float* host, int64_t count) {
  RGCN_NEED(c);
  RGCN_TRY(param_check(c, index, host, count));
  return param_download(c, c->params[index], c->params[index].val, host);
}
rgcn_status rgcn_get_grad(rgcn_ctx* c, int32_t index, float* host, int64_t count) {
  RGCN_NEED(c);
  RGCN_TRY(param_check(c, index, host, count));
  RGCN_TRY(check_dev_flag(c));
  return param_download(c, c->params[index], c->params[index].grad, host);
}

rgcn_status rgcn_set_graph(rgcn_ctx* c, const int32_t* tri, int64_t E) {
  RGCN_NEED(c);
  if (E < 0 || E > c->cfg.max_edges) RGCN_FAIL(c, RGCN_ERR_INVALID, "num_edges outside [0, max_edges]");
  if (E > 0 && !tri) RGCN_FAIL(c, RGCN_ERR_INVALID, "NULL triples");
  for (int64_t e = 0; e < E; ++e) {
    const int32_t s = tri[3 * e], r = tri[3 * e + 1], o = tri[3 * e + 2];
    if (s < 0 || s >= c->V || o < 0 || o >= c->V || r < 0 || r >= c->R)
      RGCN_FAIL(c, RGCN_ERR_INVALID, "graph_edges row " + std::to_string(e) + " has an id out of range");
  }
  RGCN_TRY(join_abandoned_side_work(c));      // (they read the message lists the build rewrites)
  if (E > 0) RGCN_TRY(to_dev(c, c->g.triples, tri, sizeof(int32_t) * 3 * (size_t)E));
  return graph_build(c, c->g.triples, E);
}

rgcn_status rgcn_set_graph_device(rgcn_ctx* c, const int32_t* tri_dev, int64_t E) {
  RGCN_NEED(c);
  if (E < 0 || E > c->cfg.max_edges) RGCN_FAIL(c, RGCN_ERR_INVALID, "num_edges outside [0, max_edges]");
  if (E > 0 && !tri_dev) RGCN_FAIL(c, RGCN_ERR_INVALID, "NULL triples");
  RGCN_TRY(join_abandoned_side_work(c));
  return graph_build(c, tri_dev, E);
}

rgcn_status rgcn_forward(rgcn_ctx* c, int32_t train, uint64_t seed, const uint8_t* masks) {
  RGCN_NEED(c);
  return forward_all(c, train, seed, masks);
}

rgcn_status rgcn_get_activation(rgcn_ctx* c, int32_t layer, float* host, int64_t count) {
  RGCN_NEED(c);
  if (layer < 0 || layer > c->L) RGCN_FAIL(c, RGCN_ERR_INVALID, "layer out of range");
  if (!host || count != (int64_t)c->V * c->d) RGCN_FAIL(c, RGCN_ERR_INVALID, "need a [V,d] host buffer");
  if (!c->fwd_done) RGCN_FAIL(c, RGCN_ERR_STATE, "no completed forward pass");
  RGCN_TRY(check_dev_flag(c));
  return to_host(c, host, c->H[layer], sizeof(float) * (size_t)count);
}
rgcn_status rgcn_get_codes(rgcn_ctx* c, float* host, int64_t count) {
  if (!c) return RGCN_ERR_INVALID;
  return rgcn_get_activation(c, c->L, host, count);
}
const float* rgcn_codes_device(rgcn_ctx* c) { return c ? c->H[c->L] : nullptr; }

rgcn_status rgcn_get_dropout_mask(rgcn_ctx* c, int32_t layer, uint8_t* host, int64_t count) {
  RGCN_NEED(c);
  if (layer < 1 || layer > c->L) RGCN_FAIL(c, RGCN_ERR_INVALID, "layer out of range");
  if (!host || count != (int64_t)c->V * c->d) RGCN_FAIL(c, RGCN_ERR_INVALID, "need a [V,d] host buffer");
  DropSpec ds = make_drop(c, layer, true);
  uint8_t* tmp = reinterpret_cast<uint8_t*>(c->stage);
  RGCN_TRY(materialize_mask(c, ds, tmp, count));
  return to_host(c, host, tmp, (size_t)count);
}

rgcn_status rgcn_backward_device(rgcn_ctx* c, const float* dcodes_dev) {
  RGCN_NEED(c);
  return backward_all(c, dcodes_dev);
}
rgcn_status rgcn_backward(rgcn_ctx* c, const float* dcodes_host, int64_t count) {
  RGCN_NEED(c);
  if (!dcodes_host || count != (int64_t)c->V * c->d) RGCN_FAIL(c, RGCN_ERR_INVALID, "dcodes must be [V,d]");
  if (!c->dcodes_own) RGCN_TRY(dmalloc(c, &c->dcodes_own, (size_t)count, false));
  RGCN_TRY(to_dev(c, c->dcodes_own, dcodes_host, sizeof(float) * (size_t)count));
  return backward_all(c, c->dcodes_own);
}

// Start of a step: mark the point a prefetch may fork from, then adopt the prefetched structures of this
// graph or build them in line.  While a hipGraph is being captured, events recorded BEFORE the capture began
// must not be waited on (the replayed graph is ordered behind everything earlier on the stream anyway).
// keep >= 0: the graph is the edge-dropout subset (keep of the E batch edges, generator key eseed) of tri_dev.
// fork_point: something of this step forks from its start (the decoder's batch preparation; inside a capture, the prefetch):
// an encoder step outside a capture has no such fork and saves the main stream the packet.
static rgcn_status step_begin(rgcn_ctx* c, const int32_t* tri_dev, int64_t E, bool fork_point, int64_t keep = -1,
                              uint64_t eseed = 0) {
  RGCN_TRY(join_abandoned_side_work(c));
  if (fork_point || c->capturing) RGCN_HIP(c, hipEventRecord(c->ev_step_begin, c->main_stream));
  c->step_begin_in_capture = c->capturing;
  if (c->g_alt.pf_valid && c->g_alt.pf_tri == tri_dev && c->g_alt.pf_E == E && c->g_alt.pf_keep == keep &&
      (keep < 0 || c->g_alt.pf_eseed == eseed)) {
    // the structures for this graph were prepared beside the previous step: swap them in
    std::swap(c->g, c->g_alt);
    c->g.pf_valid = false;
    c->fwd_done = false;
    if (!c->capturing || c->g.ready_in_capture) RGCN_HIP(c, hipStreamWaitEvent(c->main_stream, c->g.ev_ready, 0));
    return RGCN_OK;
  }
  return keep < 0 ? graph_build(c, tri_dev, E) : graph_build_dropout(c, tri_dev, E, keep, eseed, nullptr);
}
static rgcn_status step_end(rgcn_ctx* c) {
  RGCN_HIP(c, hipEventRecord(c->g.ev_free, c->main_stream));
  c->g.free_in_capture = c->capturing;
  return RGCN_OK;
}

rgcn_status rgcn_step_device(rgcn_ctx* c, const int32_t* tri_dev, int64_t E, int32_t train,
                             uint64_t seed, const float* dcodes_dev) {
  RGCN_NEED(c);
  if (E < 0 || E > c->cfg.max_edges) RGCN_FAIL(c, RGCN_ERR_INVALID, "num_edges outside [0, max_edges]");
  RGCN_TRY(step_begin(c, tri_dev, E, false));
  // (Round 5 tried forming the top layer's dS = dL/dcodes * dropout_L -- which depends on the caller's gradient and the seed
  // only -- on a side stream beside the forward pass: 0.567 ms per step against 0.559 on the same box.  A fork + join costs
  // the main stream more than the 13 us pass it hides.)
  RGCN_TRY(forward_all(c, train, seed, nullptr));
  RGCN_TRY(backward_all(c, dcodes_dev));
  return step_end(c);
}

// ---- decoder / optimizer / whole train step ("next" rows f1, f2) ---------------------------------
rgcn_status rgcn_decoder_reserve(rgcn_ctx* c, int64_t max_triples) {
  RGCN_NEED(c);
  if (max_triples <= 0 || max_triples > ((int64_t)1 << 30)) RGCN_FAIL(c, RGCN_ERR_INVALID, "max_triples out of range");
  RGCN_TRY(sync_all(c));
  return decoder_reserve(c, max_triples);
}

rgcn_status rgcn_decoder_loss_backward_device(rgcn_ctx* c, const int32_t* X_dev, const float* Y_dev, int64_t N,
                                              float reg_param) {
  RGCN_NEED(c);
  if (!c->fwd_done) RGCN_FAIL(c, RGCN_ERR_STATE, "the decoder needs a completed rgcn_forward");
  if (!X_dev || !Y_dev || N <= 0) RGCN_FAIL(c, RGCN_ERR_INVALID, "bad decoder batch");
  // world > 1: the codes are replicated after the last layer's exchange, so every rank runs the same decoder
  // pass on the same batch and holds identical dL/dcodes and dL/dW_relation (the kernels are deterministic)
  if (c->dec.maxN < N) RGCN_FAIL(c, RGCN_ERR_STATE, "call rgcn_decoder_reserve(ctx, max_triples) first");
  RGCN_TRY(decoder_prepare(c, X_dev, N));
  RGCN_TRY(decoder_compute(c, c->H[c->L], Y_dev, reg_param));
  RGCN_TRY(stream_join(c, 2));      // the relation gradient ran on its own side stream
  c->dec.loss_valid = true;
  return RGCN_OK;
}

const float* rgcn_dcodes_device(rgcn_ctx* c) { return c ? c->dcodes_own : nullptr; }

rgcn_status rgcn_get_loss(rgcn_ctx* c, double* loss) {
  RGCN_NEED(c);
  if (!loss) RGCN_FAIL(c, RGCN_ERR_INVALID, "NULL output");
  if (!c->dec.loss_valid) RGCN_FAIL(c, RGCN_ERR_STATE, "no decoder pass has run");
  if (c->capturing) RGCN_FAIL(c, RGCN_ERR_STATE, "host transfers are not allowed while a hipGraph is being captured");
  // the training loop reads the loss every iteration (optimize.py:81-88 returns it from session.run): the loss and the
  // device-side error flag come back through pinned memory with ONE wait for the main stream
  if (!c->readback_host) RGCN_HIP(c, hipHostMalloc((void**)&c->readback_host, 32, hipHostMallocDefault));
  double* h_loss = reinterpret_cast<double*>(c->readback_host);
  int32_t* h_flag = reinterpret_cast<int32_t*>(c->readback_host + 16);
  RGCN_HIP(c, hipMemcpyAsync(h_flag, c->g.errflag, sizeof(int32_t), hipMemcpyDeviceToHost, c->main_stream));
  RGCN_HIP(c, hipMemcpyAsync(h_loss, c->dec.loss, sizeof(double), hipMemcpyDeviceToHost, c->main_stream));
  RGCN_HIP(c, hipStreamSynchronize(c->main_stream));
  if (*h_flag) RGCN_TRY(check_dev_flag(c, /*main_only=*/true));      // reads it again, clears it, names the failure
  *loss = *h_loss;
  return RGCN_OK;
}

rgcn_status rgcn_negative_sample_device(rgcn_ctx* c, const int32_t* batch_dev, int64_t n, int32_t rate, uint64_t seed,
                                        int32_t* x_out_dev, float* y_out_dev) {
  RGCN_NEED(c);
  if (n < 0 || rate < 0 || rate > 1024 || (n > 0 && (!batch_dev || !x_out_dev || !y_out_dev)))
    RGCN_FAIL(c, RGCN_ERR_INVALID, "bad arguments");
  if (n * (int64_t)(rate + 1) > ((int64_t)1 << 30)) RGCN_FAIL(c, RGCN_ERR_INVALID, "batch too large");
  return negative_sample(c, batch_dev, n, rate, seed, x_out_dev, y_out_dev);
}

rgcn_status rgcn_rank_reserve(rgcn_ctx* c, int64_t max_queries) {
  RGCN_NEED(c);
  if (max_queries <= 0) RGCN_FAIL(c, RGCN_ERR_INVALID, "max_queries must be positive");
  RGCN_TRY(sync_all(c));
  return rank_reserve(c, max_queries);
}

rgcn_status rgcn_rank_device(rgcn_ctx* c, const int32_t* x_dev, int64_t n, int32_t predict_object,
                             const int64_t* filter_ptr_dev, const int32_t* filter_idx_dev, int32_t* raw_rank_dev,
                             int32_t* filtered_rank_dev) {
  RGCN_NEED(c);
  if (n < 0 || (n > 0 && (!x_dev || !filter_ptr_dev || !raw_rank_dev || !filtered_rank_dev)))
    RGCN_FAIL(c, RGCN_ERR_INVALID, "bad arguments");
  if (!c->fwd_done) RGCN_FAIL(c, RGCN_ERR_STATE, "rgcn_rank_device needs a completed rgcn_forward (test mode on the full graph)");
  // world > 1: the codes are replicated after the last exchange, so ranking needs no collective -- each rank
  // passes its own slice of the queries and the caller concatenates
  if (c->rank_max <= 0) RGCN_FAIL(c, RGCN_ERR_STATE, "call rgcn_rank_reserve first");
  if (n == 0) return RGCN_OK;
  return rank_compute(c, x_dev, n, predict_object ? 1 : 0, filter_ptr_dev, filter_idx_dev, raw_rank_dev, filtered_rank_dev);
}

rgcn_status rgcn_optimizer_config(rgcn_ctx* c, float lr, float beta1, float beta2, float eps, float max_grad_norm) {
  RGCN_NEED(c);
  if (!(lr > 0.f) || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps > 0.f) ||
      max_grad_norm < 0.f)
    RGCN_FAIL(c, RGCN_ERR_INVALID, "bad optimizer hyper-parameters");
  c->opt.lr = lr; c->opt.beta1 = beta1; c->opt.beta2 = beta2; c->opt.eps = eps; c->opt.max_norm = max_grad_norm;
  c->opt.configured = true;
  return RGCN_OK;
}

rgcn_status rgcn_optimizer_step(rgcn_ctx* c) {
  RGCN_NEED(c);
  if (c->world > 1 && !c->comm)
    RGCN_FAIL(c, RGCN_ERR_STATE, "world > 1: call rgcn_comm_init first (or drive rgcn_optimizer_norm_partial / _apply yourself)");
  return optimizer_step(c);
}
rgcn_status rgcn_optimizer_norm_partial(rgcn_ctx* c) { RGCN_NEED(c); return optimizer_norm_partial(c); }
rgcn_status rgcn_optimizer_apply(rgcn_ctx* c) { RGCN_NEED(c); return optimizer_apply(c); }

// everything of a train step behind the graph preparation: encoder forward, decoder loss + gradients, encoder
// backward, clip + Adam
static rgcn_status train_step_tail(rgcn_ctx* c, const int32_t* X_dev, const float* Y_dev, int64_t N, uint64_t seed,
                                   float reg_param) {
  RGCN_TRY(forward_all(c, 1, seed, nullptr));
  // Relation-sharded run: the decoder is divided by TRIPLES -- rank g takes the slice [g ceil(N / world), ...) of the
  // batch (the codes are replicated after the last forward exchange), its loss terms and gradients are normalised by
  // the whole batch's N, and the partial dL/dcodes [V,d], dL/dW_relation [R,d] and loss are summed over the ranks
  // (decoder_allreduce) before the backward pass starts.
  const int64_t per = (N + c->world - 1) / c->world;
  const int64_t lo = c->world > 1 ? std::min<int64_t>(N, (int64_t)c->rank * per) : 0;
  const int64_t n_loc = c->world > 1 ? std::min<int64_t>(N, lo + per) - lo : N;
  const int32_t* X_loc = X_dev + 3 * lo;
  const float* Y_loc = Y_dev + lo;
  // The decoder batch's CSRs depend on X only.  They are built on side stream 1, forked at the START of the step,
  // but enqueued AFTER the encoder's prep and forward: their two sorts are ~45 launches of a few microseconds, and
  // queued first they kept the host from feeding the main stream for the first 0.3 ms of every step.
  // A captured step is a single chain (rgcn_capture_begin) with ONE exception, this fork: the preparation is ~0.15 ms
  // of small launches that nothing in the encoder's forward pass depends on, worth far more than the two cross-stream
  // edges it puts into the graph.
  const bool fork_in_capture = c->capturing && c->use_aux_before_capture && c->step_begin_in_capture && c->world == 1;
  if (c->use_aux || fork_in_capture) {
    RGCN_HIP(c, hipStreamWaitEvent(c->aux[1], c->ev_step_begin, 0));
    c->stream = c->aux[1];
    c->aux_dirty[1] = true;
    const rgcn_status ps = decoder_prepare(c, X_loc, n_loc, N);
    c->stream = c->main_stream;
    RGCN_TRY(ps);
  } else {
    RGCN_TRY(decoder_prepare(c, X_loc, n_loc, N));
  }
  RGCN_TRY(stream_join(c, 1));
  // one GPU: the decoder's entity-gradient kernel also writes dL/dcodes * dropout_L, the operand of the top layer's
  // self-loop GEMMs (a sharded run scales after the all-reduce of the partial dL/dcodes, in bwd_begin)
  const DropSpec top_drop = make_drop(c, c->L, true);
  float* ds_ready = (c->world == 1 && top_drop.mode != DROP_NONE) ? c->dsbuf[c->L & 1] : nullptr;
  RGCN_TRY(decoder_compute(c, c->H[c->L], Y_loc, reg_param, ds_ready, &top_drop));
  if (c->world > 1) {
    RGCN_TRY(stream_join(c, 2));    // the relation gradient's reduce ran on its own side stream: it is all-reduced too
    RGCN_TRY(decoder_allreduce(c));
  }
  c->dec.loss_valid = true;
  RGCN_TRY(backward_all(c, c->dcodes_own, ds_ready));
  RGCN_TRY(stream_join(c, 2));      // dL/dW_relation, computed beside the backward pass
  RGCN_TRY(step_end(c));
  if (c->opt.configured) RGCN_TRY(optimizer_step(c));
  return RGCN_OK;
}

static rgcn_status prefetch_impl(rgcn_ctx* c, const int32_t* tri_dev, int64_t E, int64_t keep, uint64_t eseed) {
  if (E < 0 || E > c->cfg.max_edges) RGCN_FAIL(c, RGCN_ERR_INVALID, "num_edges outside [0, max_edges]");
  if (keep > E) RGCN_FAIL(c, RGCN_ERR_INVALID, "edge dropout: keep outside [0, num_edges]");
  if (keep >= 0 && E >= ((int64_t)1 << 24)) RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "edge dropout: more than 2^24 batch edges");
  if (E > 0 && !tri_dev) RGCN_FAIL(c, RGCN_ERR_INVALID, "NULL triples");
  // build into the INACTIVE set on the prefetch stream; it only has to wait for the last step that
  // used that set (not for the step currently queued on the main stream)
  std::swap(c->g, c->g_alt);
  rgcn_status s = RGCN_OK;
  do {
    if (c->capturing) {
      // join the capture at the start of the step queued last (so that the preparation overlaps it), and
      // behind the last captured step that used this buffer set
      if (!c->step_begin_in_capture) { s = RGCN_ERR_STATE; c->err = "capture: queue a step before its prefetch"; break; }
      if (hipStreamWaitEvent(c->pf_stream, c->ev_step_begin, 0) != hipSuccess) { s = RGCN_ERR_HIP; c->err = "hipStreamWaitEvent"; break; }
      if (c->g.free_in_capture && hipStreamWaitEvent(c->pf_stream, c->g.ev_free, 0) != hipSuccess) { s = RGCN_ERR_HIP; c->err = "hipStreamWaitEvent"; break; }
      c->cap_pf_forked = true;
    } else if (hipStreamWaitEvent(c->pf_stream, c->g.ev_free, 0) != hipSuccess) { s = RGCN_ERR_HIP; c->err = "hipStreamWaitEvent"; break; }
    c->stream = c->pf_stream;
    const bool was_done = c->fwd_done;
    s = keep < 0 ? graph_build(c, tri_dev, E) : graph_build_dropout(c, tri_dev, E, keep, eseed, nullptr);
    c->fwd_done = was_done;          // the ACTIVE set's forward state is untouched
    c->stream = c->main_stream;
    if (s != RGCN_OK) break;
    if (hipEventRecord(c->g.ev_ready, c->pf_stream) != hipSuccess) { s = RGCN_ERR_HIP; c->err = "hipEventRecord"; break; }
    c->g.ready_in_capture = c->capturing;
    c->g.pf_tri = tri_dev;
    c->g.pf_E = E;
    c->g.pf_keep = keep;
    c->g.pf_eseed = eseed;
    c->g.pf_valid = true;
  } while (0);
  c->stream = c->main_stream;
  std::swap(c->g, c->g_alt);
  return s;
}

rgcn_status rgcn_train_step_device(rgcn_ctx* c, const int32_t* tri_dev, int64_t E, const int32_t* X_dev,
                                   const float* Y_dev, int64_t N, uint64_t seed, float reg_param) {
  RGCN_NEED(c);
  if (E < 0 || E > c->cfg.max_edges) RGCN_FAIL(c, RGCN_ERR_INVALID, "num_edges outside [0, max_edges]");
  if (!X_dev || !Y_dev || N <= 0) RGCN_FAIL(c, RGCN_ERR_INVALID, "bad decoder batch");
  if (c->world > 1 && !c->comm)
    RGCN_FAIL(c, RGCN_ERR_STATE, "world > 1: call rgcn_comm_init first (or drive the phase API yourself)");
  if (c->dec.maxN < N) RGCN_FAIL(c, RGCN_ERR_STATE, "call rgcn_decoder_reserve(ctx, max_triples) first");
  RGCN_TRY(step_begin(c, tri_dev, E, true));
  return train_step_tail(c, X_dev, Y_dev, N, seed, reg_param);
}

rgcn_status rgcn_prefetch_graph_device(rgcn_ctx* c, const int32_t* tri_dev, int64_t E) {
  RGCN_NEED(c);
  return prefetch_impl(c, tri_dev, E, -1, 0);
}
rgcn_status rgcn_prefetch_graph_dropout_device(rgcn_ctx* c, const int32_t* batch_dev, int64_t n, int64_t keep,
                                               uint64_t seed) {
  RGCN_NEED(c);
  if (keep < 0) RGCN_FAIL(c, RGCN_ERR_INVALID, "edge dropout: keep outside [0, num_edges]");
  return prefetch_impl(c, batch_dev, n, keep, seed);
}

rgcn_status rgcn_set_graph_dropout_device(rgcn_ctx* c, const int32_t* batch_dev, int64_t n, int64_t keep, uint64_t seed,
                                          const uint8_t* keep_mask_dev) {
  RGCN_NEED(c);
  if (n < 0 || n > c->cfg.max_edges) RGCN_FAIL(c, RGCN_ERR_INVALID, "num_edges outside [0, max_edges]");
  if (keep < 0 || keep > n) RGCN_FAIL(c, RGCN_ERR_INVALID, "edge dropout: keep outside [0, num_edges]");
  if (n >= ((int64_t)1 << 24)) RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "edge dropout: more than 2^24 batch edges");
  if (n > 0 && !batch_dev) RGCN_FAIL(c, RGCN_ERR_INVALID, "NULL triples");
  return graph_build_dropout(c, batch_dev, n, keep, seed, keep_mask_dev);
}

rgcn_status rgcn_get_graph_edges(rgcn_ctx* c, int32_t* host, int64_t count) {
  RGCN_NEED(c);
  if (!c->g.ready) RGCN_FAIL(c, RGCN_ERR_STATE, "no graph has been set");
  if (count != c->g.E || (count > 0 && !host)) RGCN_FAIL(c, RGCN_ERR_INVALID, "need an [E,3] host buffer of the current graph's size");
  if (count == 0) return RGCN_OK;
  RGCN_TRY(check_dev_flag(c));
  return to_host(c, host, c->g.cur, sizeof(int32_t) * 3 * (size_t)count);
}

rgcn_status rgcn_train_step_minibatch_device(rgcn_ctx* c, const int32_t* batch_dev, int64_t n, int64_t keep,
                                             uint64_t edge_seed, int32_t negative_rate, uint64_t negative_seed,
                                             int32_t* x_scratch_dev, float* y_scratch_dev, uint64_t dropout_seed,
                                             float reg_param) {
  RGCN_NEED(c);
  if (n <= 0 || n > c->cfg.max_edges) RGCN_FAIL(c, RGCN_ERR_INVALID, "num_edges outside [1, max_edges]");
  if (keep < 0 || keep > n) RGCN_FAIL(c, RGCN_ERR_INVALID, "edge dropout: keep outside [0, num_edges]");
  if (n >= ((int64_t)1 << 24)) RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "edge dropout: more than 2^24 batch edges");
  if (negative_rate < 0 || negative_rate > 1024 || !batch_dev || !x_scratch_dev || !y_scratch_dev)
    RGCN_FAIL(c, RGCN_ERR_INVALID, "bad arguments");
  const int64_t N = n * (int64_t)(negative_rate + 1);
  if (c->world > 1 && !c->comm)
    RGCN_FAIL(c, RGCN_ERR_STATE, "world > 1: call rgcn_comm_init first (or drive the phase API yourself)");
  if (c->dec.maxN < N) RGCN_FAIL(c, RGCN_ERR_STATE, "call rgcn_decoder_reserve(ctx, max_triples) first");
  // the message graph: adopted from the prefetch of this very (batch, keep, seed), or drawn and prepared in line
  RGCN_TRY(step_begin(c, batch_dev, n, false, keep, edge_seed));   // (the fork point is recorded below)
  // the decoder batch: all n batch edges as positives (the dropped ones included) + their corruptions
  RGCN_TRY(negative_sample(c, batch_dev, n, negative_rate, negative_seed, x_scratch_dev, y_scratch_dev));
  // the decoder's batch preparation forks from ev_step_begin onto a side stream: it must see the sampled batch
  RGCN_HIP(c, hipEventRecord(c->ev_step_begin, c->main_stream));
  return train_step_tail(c, x_scratch_dev, y_scratch_dev, N, dropout_seed, reg_param);
}

// ---- neighbourhood edge sampler on the device ------------------------------------------
rgcn_status rgcn_neighborhood_reserve(rgcn_ctx* c, const int32_t* triples_host, int64_t n) {
  RGCN_NEED(c);
  if (n <= 0 || n >= ((int64_t)1 << 31) / 3 || !triples_host) RGCN_FAIL(c, RGCN_ERR_INVALID, "need the training triples");
  if (c->capturing) RGCN_FAIL(c, RGCN_ERR_STATE, "not while a hipGraph is being captured");
  for (int64_t e = 0; e < n; ++e) {
    const int32_t s = triples_host[3 * e], r = triples_host[3 * e + 1], o = triples_host[3 * e + 2];
    if (s < 0 || s >= c->V || o < 0 || o >= c->V || r < 0 || r >= c->R)
      RGCN_FAIL(c, RGCN_ERR_INVALID, "training triple " + std::to_string(e) + " has an id out of range");
  }
  RGCN_TRY(sync_all(c));
  return neighborhood_reserve(c, triples_host, n);
}

rgcn_status rgcn_sample_neighborhood_device(rgcn_ctx* c, int64_t sample_size, uint64_t seed, int32_t* batch_out_dev,
                                            int32_t on_prefetch_stream) {
  RGCN_NEED(c);
  if (c->nbr.n <= 0) RGCN_FAIL(c, RGCN_ERR_STATE, "call rgcn_neighborhood_reserve first");
  // the reference's loop dies on NaN probabilities when asked for more edges than the graph has
  if (sample_size < 0 || sample_size > c->nbr.n) RGCN_FAIL(c, RGCN_ERR_INVALID, "sample_size outside [0, number of training triples]");
  if (sample_size > 0 && !batch_out_dev) RGCN_FAIL(c, RGCN_ERR_INVALID, "NULL output");
  if (c->capturing) RGCN_FAIL(c, RGCN_ERR_STATE, "not while a hipGraph is being captured");
  invalidate_prefetch_of(c, batch_out_dev, sizeof(int32_t) * 3 * (size_t)sample_size);
  if (on_prefetch_stream) c->stream = c->pf_stream;
  const rgcn_status s = neighborhood_sample(c, sample_size, seed, batch_out_dev, on_prefetch_stream != 0);
  c->stream = c->main_stream;
  return s;
}

// ---- hipGraph capture of whole steps --------------------------------------------------------------
namespace {
__global__ void k_bump_counter(uint64_t* counter) { counter[0] += 1; }
}

rgcn_status rgcn_capture_begin(rgcn_ctx* c) {
  RGCN_NEED(c);
  if (c->capturing) RGCN_FAIL(c, RGCN_ERR_STATE, "a capture is already running");
  if (c->prof_on) RGCN_FAIL(c, RGCN_ERR_STATE, "switch the per-kernel profile off before capturing");
  // Sharded contexts: the step contains RCCL collectives, which a stream capture records like any other launch.  That
  // path is exercised with device-side stand-in collectives of several ranks on ONE GPU only
  // (tests/test_gpu_multiprocess.py::test_captured_sharded_train_step); real librccl kernels inside a capture have never
  // run here (no multi-GPU box), so it stays EXPERIMENTAL and opt-in: RGCN_CAPTURE_SHARDED=1.
  if (c->world > 1 && knob("RGCN_CAPTURE_SHARDED", 0) != 1)
    RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "capture on a sharded context is experimental (never run against real RCCL on "
                                       "several GPUs): the devtools build enables it with RGCN_CAPTURE_SHARDED=1");
  RGCN_TRY(sync_all(c));
  if (!c->replay_counter) {
    RGCN_HIP(c, hipMalloc((void**)&c->replay_counter, sizeof(uint64_t)));
    RGCN_HIP(c, hipMemset(c->replay_counter, 0, sizeof(uint64_t)));
  }
  c->g.ready_in_capture = c->g.free_in_capture = false;
  c->g_alt.ready_in_capture = c->g_alt.free_in_capture = false;
  c->step_begin_in_capture = false;
  c->cap_pf_forked = false;
  RGCN_HIP(c, hipStreamBeginCapture(c->main_stream, hipStreamCaptureModeRelaxed));
  c->capturing = true;
  // whatever the capture does first, the weights' derived tables (MFMA fragments, band tiles) are rebuilt INSIDE it: a
  // replay follows weights the host does not see, tables built before the capture would be baked in stale
  c->frag_fresh = false;
  c->wtile_fresh = false;
  // A captured step is recorded as ONE chain on the main stream (only the prefetch of the next graph forks off):
  // the replayed graph pays more than streams do for every cross-stream edge (measured, profiles/r02_hipgraph_ab.log:
  // 0.687 ms per step with the side streams captured, 0.634-0.640 as a chain, 0.614-0.631 stream-launched), and
  // the side streams buy the stream-launched step under 3 %.
  // (devtools knob RGCN_CAPTURE_STREAMS=1 records the stream-launched step's own fork / join DAG instead: the A/B of
  // every round, tools/gpu_r6_capture_ab.sh)
  c->use_aux_before_capture = c->use_aux;
  if (knob("RGCN_CAPTURE_STREAMS", 0) != 1) c->use_aux = false;
  hipLaunchKernelGGL(k_bump_counter, dim3(1), dim3(1), 0, c->main_stream, c->replay_counter);
  return RGCN_OK;
}

rgcn_status rgcn_capture_end(rgcn_ctx* c, int32_t* graph_id) {
  RGCN_NEED(c);
  if (!c->capturing) RGCN_FAIL(c, RGCN_ERR_STATE, "no capture is running");
  if (!graph_id) RGCN_FAIL(c, RGCN_ERR_INVALID, "NULL graph_id");
  hipError_t e = hipSuccess;
  for (int k = 0; k < kAuxStreams; ++k)        // (side streams forked inside the capture rejoin the origin stream)
    if (c->use_aux && c->aux[k] && c->aux_dirty[k]) {
      c->aux_dirty[k] = false;
      if (e == hipSuccess) e = hipEventRecord(c->ev_join[k], c->aux[k]);
      if (e == hipSuccess) e = hipStreamWaitEvent(c->main_stream, c->ev_join[k], 0);
    }
  if (c->cap_pf_forked) {                      // the prefetch stream must rejoin the origin stream
    e = hipEventRecord(c->ev_join[0], c->pf_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->main_stream, c->ev_join[0], 0);
  }
  hipGraph_t graph = nullptr;
  const hipError_t e2 = hipStreamEndCapture(c->main_stream, &graph);
  c->capturing = false;
  c->frag_fresh = false;
  c->wtile_fresh = false;
  c->use_aux = c->use_aux_before_capture;
  // events last recorded inside the capture are unusable outside it: give them a fresh, ordinary record
  (void)hipEventRecord(c->ev_fork, c->main_stream);
  (void)hipEventRecord(c->ev_step_begin, c->main_stream);
  for (int k = 0; k < kAuxStreams; ++k) if (c->aux[k]) (void)hipEventRecord(c->ev_join[k], c->aux[k]);
  for (rgcn::GraphBufs* g : {&c->g, &c->g_alt}) {
    if (g->ev_ready) (void)hipEventRecord(g->ev_ready, c->pf_stream);
    if (g->ev_free) (void)hipEventRecord(g->ev_free, c->main_stream);
    g->ready_in_capture = g->free_in_capture = false;
  }
  if (c->dec.ev_ready) (void)hipEventRecord(c->dec.ev_ready, c->main_stream);
  c->step_begin_in_capture = false;
  if (e != hipSuccess || e2 != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    c->err = std::string("stream capture failed: ") + hipGetErrorString(e2 != hipSuccess ? e2 : e);
    return RGCN_ERR_HIP;
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(graph);
    c->err = std::string("hipGraphInstantiate: ") + hipGetErrorString(e);
    return RGCN_ERR_HIP;
  }
  c->graph_defs.push_back(graph);
  c->graphs.push_back(exec);
  *graph_id = (int32_t)c->graphs.size() - 1;
  return RGCN_OK;
}

rgcn_status rgcn_graph_launch(rgcn_ctx* c, int32_t graph_id) {
  RGCN_NEED(c);
  if (c->capturing) RGCN_FAIL(c, RGCN_ERR_STATE, "a capture is running");
  if (graph_id < 0 || graph_id >= (int32_t)c->graphs.size() || !c->graphs[graph_id])
    RGCN_FAIL(c, RGCN_ERR_INVALID, "unknown graph id");
  RGCN_HIP(c, hipGraphLaunch(c->graphs[graph_id], c->main_stream));
  c->fwd_done = true;       // the replayed steps leave activations / gradients of their last step behind
  c->weights_version += 1;  // a replayed train step moved the weights behind the host's back: derived weight tables
                            // (block_rows) built before the replay are stale for uncaptured passes
  return RGCN_OK;
}

rgcn_status rgcn_graph_destroy(rgcn_ctx* c, int32_t graph_id) {
  RGCN_NEED(c);
  if (graph_id < 0 || graph_id >= (int32_t)c->graphs.size() || !c->graphs[graph_id])
    RGCN_FAIL(c, RGCN_ERR_INVALID, "unknown graph id");
  RGCN_TRY(sync_all(c));
  (void)hipGraphExecDestroy(c->graphs[graph_id]);
  (void)hipGraphDestroy(c->graph_defs[graph_id]);
  c->graphs[graph_id] = nullptr;
  c->graph_defs[graph_id] = nullptr;
  return RGCN_OK;
}

rgcn_status rgcn_set_relation_owner(rgcn_ctx* c, const int32_t* owner, int32_t count) {
  RGCN_NEED(c);
  if (!owner || count != c->R) RGCN_FAIL(c, RGCN_ERR_INVALID, "owner must have RelationCount entries");
  for (int r = 0; r < count; ++r)
    if (owner[r] < 0 || owner[r] >= c->world) RGCN_FAIL(c, RGCN_ERR_INVALID, "owner[r] outside [0, world)");
  c->g.ready = false;
  c->g.pf_valid = false;
  c->g_alt.pf_valid = false;
  c->fwd_done = false;
  RGCN_TRY(sync_all(c));
  return to_dev(c, c->g.owner, owner, sizeof(int32_t) * (size_t)count);
}

rgcn_status rgcn_comm_unique_id(uint8_t id[128]) {
  if (!id) return RGCN_ERR_INVALID;
  return comm_unique_id(id);
}
rgcn_status rgcn_comm_init(rgcn_ctx* c, const uint8_t id[128]) {
  RGCN_NEED(c);
  if (!id) RGCN_FAIL(c, RGCN_ERR_INVALID, "NULL id");
  return comm_init(c, id);
}
rgcn_status rgcn_comm_info(rgcn_ctx* c, int32_t* comm_ranks, int32_t* comm_rank, int32_t* comm_device) {
  RGCN_NEED(c);
  return comm_info(c, comm_ranks, comm_rank, comm_device);
}

rgcn_status rgcn_device_info(int32_t device, int32_t* count, int64_t* pci_address) {
  if (!count || !pci_address) return RGCN_ERR_INVALID;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
  *count = n;
  *pci_address = -1;
  if (device < 0 || device >= n) return RGCN_OK;
  int dom = 0, bus = 0, dev = 0;
  if (hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, device) != hipSuccess ||
      hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, device) != hipSuccess ||
      hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, device) != hipSuccess) {
    (void)hipGetLastError();
    return RGCN_ERR_HIP;
  }
  *pci_address = ((int64_t)dom << 16) | ((int64_t)(bus & 0xff) << 8) | (int64_t)(dev & 0xff);
  return RGCN_OK;
}

rgcn_status rgcn_comm_allreduce_sum(rgcn_ctx* c, float* dev, int64_t count) {
  RGCN_NEED(c);
  return comm_allreduce(c, dev, count);
}

rgcn_status rgcn_forward_begin(rgcn_ctx* c, int32_t train, uint64_t seed, const uint8_t* masks) {
  RGCN_NEED(c);
  return fwd_begin(c, train, seed, masks);
}
rgcn_status rgcn_forward_layer_partial(rgcn_ctx* c, int32_t l) { RGCN_NEED(c); return fwd_layer_partial(c, l); }
rgcn_status rgcn_forward_layer_finish(rgcn_ctx* c, int32_t l) { RGCN_NEED(c); return fwd_layer_finish(c, l); }
rgcn_status rgcn_backward_begin(rgcn_ctx* c, const float* dcodes_dev) { RGCN_NEED(c); return bwd_begin(c, dcodes_dev); }
rgcn_status rgcn_backward_layer_partial(rgcn_ctx* c, int32_t l) { RGCN_NEED(c); return bwd_layer_partial(c, l); }
rgcn_status rgcn_backward_layer_finish(rgcn_ctx* c, int32_t l) { RGCN_NEED(c); return bwd_layer_finish(c, l); }
rgcn_status rgcn_backward_end(rgcn_ctx* c) { RGCN_NEED(c); return bwd_end(c); }

static rgcn_status buffer_of(rgcn_ctx* c, int32_t which, void** p, int64_t* bytes) {
  const int64_t Vd = (int64_t)c->V * c->d * 4;
  switch (which) {
    case RGCN_BUF_EXCHANGE:
      if (!c->exch) RGCN_FAIL(c, RGCN_ERR_STATE, "no exchange buffer on a world == 1 context");
      *p = c->exch; *bytes = Vd; return RGCN_OK;
    case RGCN_BUF_SELF: *p = c->self_buf; *bytes = Vd; return RGCN_OK;
    case RGCN_BUF_DSELF_EXCHANGE: {
      int l = c->bwd_layer;
      if (l < 1 || l > c->L) RGCN_FAIL(c, RGCN_ERR_STATE, "no backward layer in flight");
      *p = c->layers[l].gwself; *bytes = (int64_t)c->d * c->d * 4; return RGCN_OK;
    }
    case RGCN_BUF_DBASIS_EXCHANGE: {
      int l = c->bwd_layer;
      if (c->kind != RGCN_KIND_BASIS) RGCN_FAIL(c, RGCN_ERR_STATE, "basis kind only");
      if (l < 1 || l > c->L) RGCN_FAIL(c, RGCN_ERR_STATE, "no backward layer in flight");
      *p = c->layers[l].grel; *bytes = (int64_t)2 * c->B * c->d * c->d * 4; return RGCN_OK;
    }
    case RGCN_BUF_NORM_EXCHANGE:
      if (!c->opt.shard_sq) RGCN_FAIL(c, RGCN_ERR_STATE, "no sharded-norm buffer (world == 1, or no rgcn_optimizer_norm_partial yet)");
      *p = c->opt.shard_sq; *bytes = 4; return RGCN_OK;
    case RGCN_BUF_INDEG: *p = c->g.indeg; *bytes = (int64_t)c->V * 4; return RGCN_OK;
    case RGCN_BUF_OUTDEG: *p = c->g.outdeg; *bytes = (int64_t)c->V * 4; return RGCN_OK;
    case RGCN_BUF_ROWPTR: *p = c->g.row_ptr; *bytes = (int64_t)(c->V + 1) * 4; return RGCN_OK;
    case RGCN_BUF_PERM_VERTEX: *p = c->g.permv; *bytes = (int64_t)2 * c->g.E * 4; return RGCN_OK;
    case RGCN_BUF_PERM_RELATION: *p = c->g.permr; *bytes = (int64_t)2 * c->g.E * 4; return RGCN_OK;
    case RGCN_BUF_RANK_ENERGIES:
      if (!c->rank_s) RGCN_FAIL(c, RGCN_ERR_STATE, "no score buffer (rgcn_rank_reserve first)");
      *p = c->rank_s; *bytes = (int64_t)c->rank_max * c->V * 4; return RGCN_OK;
    default: RGCN_FAIL(c, RGCN_ERR_INVALID, "unknown buffer id");
  }
}
rgcn_status rgcn_read_buffer(rgcn_ctx* c, int32_t which, void* host, int64_t bytes) {
  RGCN_NEED(c);
  void* p; int64_t n;
  RGCN_TRY(buffer_of(c, which, &p, &n));
  if (!host || bytes != n) RGCN_FAIL(c, RGCN_ERR_INVALID, "buffer size mismatch");
  return to_host(c, host, p, (size_t)n);
}
rgcn_status rgcn_write_buffer(rgcn_ctx* c, int32_t which, const void* host, int64_t bytes) {
  RGCN_NEED(c);
  void* p; int64_t n;
  RGCN_TRY(buffer_of(c, which, &p, &n));
  if (!host || bytes != n) RGCN_FAIL(c, RGCN_ERR_INVALID, "buffer size mismatch");
  if (which != RGCN_BUF_EXCHANGE && which != RGCN_BUF_DSELF_EXCHANGE && which != RGCN_BUF_NORM_EXCHANGE &&
      which != RGCN_BUF_DBASIS_EXCHANGE)
    RGCN_FAIL(c, RGCN_ERR_INVALID, "only the exchange buffers are writable");
  return to_dev(c, p, host, (size_t)n);
}

rgcn_status rgcn_device_alloc(rgcn_ctx* c, int64_t bytes, void** dev) {
  RGCN_NEED(c);
  if (!dev || bytes < 0) RGCN_FAIL(c, RGCN_ERR_INVALID, "bad arguments");
  RGCN_HIP(c, hipMalloc(dev, (size_t)(bytes ? bytes : 1)));
  return RGCN_OK;
}
rgcn_status rgcn_device_free(rgcn_ctx* c, void* dev) {
  RGCN_NEED(c);
  if (c->capturing) RGCN_FAIL(c, RGCN_ERR_STATE, "this call synchronises with the device: not allowed between rgcn_capture_begin and rgcn_capture_end");
  RGCN_HIP(c, hipStreamSynchronize(c->stream));
  if (dev) {
    invalidate_prefetch_of(c, dev, 1);
    if (c->pf_stream) RGCN_HIP(c, hipStreamSynchronize(c->pf_stream));   // a prefetch may still be reading it
    RGCN_HIP(c, hipFree(dev));
  }
  return RGCN_OK;
}
rgcn_status rgcn_copy_to_device(rgcn_ctx* c, void* dev, const void* host, int64_t bytes) {
  RGCN_NEED(c);
  if (!dev || !host || bytes < 0) RGCN_FAIL(c, RGCN_ERR_INVALID, "bad arguments");
  invalidate_prefetch_of(c, dev, (size_t)bytes);
  return to_dev(c, dev, host, (size_t)bytes);
}
rgcn_status rgcn_copy_to_device_async(rgcn_ctx* c, void* dev, const void* host, int64_t bytes,
                                      int32_t on_prefetch_stream) {
  RGCN_NEED(c);
  if (!dev || !host || bytes < 0) RGCN_FAIL(c, RGCN_ERR_INVALID, "bad arguments");
  if (c->capturing) RGCN_FAIL(c, RGCN_ERR_STATE, "host transfers are not allowed while a hipGraph is being captured");
  if (bytes == 0) return RGCN_OK;
  invalidate_prefetch_of(c, dev, (size_t)bytes);
  hipStream_t st = on_prefetch_stream ? c->pf_stream : c->main_stream;
  if ((size_t)bytes > rgcn_ctx::kStageBytes) {      // too large for a staging slot: ordered on `st`, host waits
    RGCN_HIP(c, hipMemcpyAsync(dev, host, (size_t)bytes, hipMemcpyHostToDevice, st));
    RGCN_HIP(c, hipStreamSynchronize(st));
    return RGCN_OK;
  }
  if (!c->stage_host) {
    RGCN_HIP(c, hipHostMalloc((void**)&c->stage_host, rgcn_ctx::kStageSlots * rgcn_ctx::kStageBytes, hipHostMallocDefault));
    for (int k = 0; k < rgcn_ctx::kStageSlots; ++k)
      RGCN_HIP(c, hipEventCreateWithFlags(&c->stage_done[k], hipEventDisableTiming));
  }
  const int slot = c->stage_next;
  c->stage_next = (slot + 1) % rgcn_ctx::kStageSlots;
  RGCN_HIP(c, hipEventSynchronize(c->stage_done[slot]));      // the transfer that last used this slot has run
  uint8_t* stage = c->stage_host + (size_t)slot * rgcn_ctx::kStageBytes;
  memcpy(stage, host, (size_t)bytes);
  RGCN_HIP(c, hipMemcpyAsync(dev, stage, (size_t)bytes, hipMemcpyHostToDevice, st));
  RGCN_HIP(c, hipEventRecord(c->stage_done[slot], st));
  return RGCN_OK;
}
rgcn_status rgcn_copy_to_host(rgcn_ctx* c, void* host, const void* dev, int64_t bytes) {
  RGCN_NEED(c);
  if (!dev || !host || bytes < 0) RGCN_FAIL(c, RGCN_ERR_INVALID, "bad arguments");
  return to_host(c, host, dev, (size_t)bytes);
}

rgcn_status rgcn_timer_start(rgcn_ctx* c) {
  RGCN_NEED(c);
  if (c->capturing) RGCN_FAIL(c, RGCN_ERR_STATE, "not allowed between rgcn_capture_begin and rgcn_capture_end");
  RGCN_HIP(c, hipEventRecord(c->t0, c->stream));
  return RGCN_OK;
}
rgcn_status rgcn_timer_stop(rgcn_ctx* c, float* ms) {
  RGCN_NEED(c);
  if (c->capturing) RGCN_FAIL(c, RGCN_ERR_STATE, "not allowed between rgcn_capture_begin and rgcn_capture_end");
  if (!ms) RGCN_FAIL(c, RGCN_ERR_INVALID, "NULL output");
  RGCN_HIP(c, hipEventRecord(c->t1, c->stream));
  RGCN_HIP(c, hipEventSynchronize(c->t1));
  RGCN_HIP(c, hipEventElapsedTime(ms, c->t0, c->t1));
  return RGCN_OK;
}

rgcn_status rgcn_set_overlap(rgcn_ctx* c, int32_t on) {
  RGCN_NEED(c);
  RGCN_TRY(sync_all(c));
  c->use_aux = on != 0;
  return RGCN_OK;
}

rgcn_status rgcn_set_fusion(rgcn_ctx* c, int32_t mode) {
  RGCN_NEED(c);
  if (mode != 0 && mode != 1) RGCN_FAIL(c, RGCN_ERR_INVALID, "fusion mode must be 0 (two kernels) or 1 (single pass)");
  c->fuse = mode;
  return RGCN_OK;
}

rgcn_status rgcn_set_gemm_mode(rgcn_ctx* c, int32_t mode) {
  RGCN_NEED(c);
  if (mode != 0 && mode != 3 && mode != 6 && mode != 9) RGCN_FAIL(c, RGCN_ERR_INVALID, "gemm mode must be 0, 3, 6 or 9");
  c->gemm_mode = mode;
  return RGCN_OK;
}

rgcn_status rgcn_profile_enable(rgcn_ctx* c, int32_t on) {
  RGCN_NEED(c);
  if (!on && c->prof_on) RGCN_TRY(profile_collect(c));
  c->prof_on = on != 0;
  return RGCN_OK;
}
rgcn_status rgcn_profile_reset(rgcn_ctx* c) {
  RGCN_NEED(c);
  RGCN_TRY(profile_collect(c));
  c->prof_agg.clear();
  return RGCN_OK;
}
int32_t rgcn_profile_count(rgcn_ctx* c) {
  if (!c) return 0;
  (void)hipSetDevice(c->cfg.device);
  if (profile_collect(c) != RGCN_OK) return 0;
  return (int32_t)c->prof_agg.size();
}
rgcn_status rgcn_profile_get(rgcn_ctx* c, int32_t i, char* name, int32_t name_cap, int64_t* calls,
                             double* total_ms, double* alg_bytes, double* alg_flops) {
  RGCN_NEED(c);
  if (i < 0 || i >= (int32_t)c->prof_agg.size()) RGCN_FAIL(c, RGCN_ERR_INVALID, "profile index out of range");
  const ProfAgg& a = c->prof_agg[i];
  if (name && name_cap > 0) {
    strncpy(name, a.name.c_str(), (size_t)name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (calls) *calls = a.calls;
  if (total_ms) *total_ms = a.ms;
  if (alg_bytes) *alg_bytes = a.bytes;
  if (alg_flops) *alg_flops = a.flops;
  return RGCN_OK;
}
rgcn_status rgcn_profile_get_compulsory(rgcn_ctx* c, int32_t i, double* compulsory_bytes) {
  RGCN_NEED(c);
  if (i < 0 || i >= (int32_t)c->prof_agg.size()) RGCN_FAIL(c, RGCN_ERR_INVALID, "profile index out of range");
  if (compulsory_bytes) *compulsory_bytes = c->prof_agg[i].cbytes;
  return RGCN_OK;
}


}  // extern "C"
