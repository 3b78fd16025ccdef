// The per-layer orchestration of the encoder: which kernel goes to which stream, in which order.
//   forward  (MessageGcn.compute_vertex_embeddings, code/encoders/message_gcns/message_gcn.py:49-79)
//   backward (tf.gradients(loss, weights), code/optimization/abstract.py:117-118; formulas SURVEY 8a a15)
// One schedule per situation: minibatch scale on one GPU with side streams, the chain a captured
// step records, full-graph scale / a relation-sharded run (the exchange points of DESIGN.md section 7).  The C ABI that
// drives these functions is rgcn_api.hip.
#include "rgcn_api_internal.h"

namespace rgcn {

// ---------------------------------------------------------------- forward
rgcn_status fwd_begin(rgcn_ctx* c, int train, uint64_t seed, const uint8_t* masks_host) {
  if (!c->g.ready) RGCN_FAIL(c, RGCN_ERR_STATE, "rgcn_forward before rgcn_set_graph");
  // (a backward pass that was driven layer by layer and abandoned between layers 2 and 1 left its side kernels unjoined:
  // they read the activations this pass overwrites.  Nothing is queued in the usual case, where every pass ended joined.)
  RGCN_TRY(join_abandoned_side_work(c));
  c->fwd_done = false;
  c->frag_fresh = false;
  c->wtile_fresh = false;
  c->fwd_train = train ? 1 : 0;
  c->seed = seed;
  c->explicit_masks = false;
  if (train && masks_host) {
    const size_t n = (size_t)c->L * c->V * c->d;
    if (!c->masks) RGCN_TRY(dmalloc(c, &c->masks, n, false));
    RGCN_TRY(to_dev(c, c->masks, masks_host, n));
    c->explicit_masks = true;
  }
  return input_forward(c);
}

// The all-gather of the rows finished last runs on side stream 1; whoever needs ALL rows (message kernels, the
// decoder, the column sums) makes its stream wait here, whoever needs the rank's own rows only (self-loop GEMMs) does not.
static rgcn_status wait_gather(rgcn_ctx* c) {
  if (c->gather_pending) {
    RGCN_HIP(c, hipStreamWaitEvent(c->stream, c->ev_gather, 0));
    if (c->stream == c->main_stream) c->gather_pending = false;
  }
  return RGCN_OK;
}
// all-gather the [V_pad,d] buffer whose own rows this rank just finished, beside whatever the main stream does next
static rgcn_status gather_rows(rgcn_ctx* c, float* buf) {
  StreamScope side(c, 1);
  RGCN_TRY(comm_all_gather(c, buf, (int64_t)c->shard_rows * c->d));
  if (side.active) {
    RGCN_HIP(c, hipEventRecord(c->ev_gather, c->aux[1]));
    c->gather_pending = true;
  }
  return RGCN_OK;
}

// The block layer destination-major in ONE pass over the incidence CSR, one column band per XCD, weights through L2
// (block_rows.hip): block kind, any world.  Otherwise (rgcn_set_fusion 0, or more blocks than the kernel's lane groups
// cover) the two-kernel form: relation-major message kernel + k_combine.
static bool rows_layer(const rgcn_ctx* c) { return c->fuse == 1 && block_rows_available(c); }

// Fragment tables of the weights that are the B operand of a contraction (W_self of every layer in both orientations, the
// basis tensors), rebuilt -- all of them, one launch -- when the weights changed (set_param, Adam) and once inside every
// captured step, whose replays follow weights the host does not see.  Nothing to do when the dense contractions run on
// the fp32 MFMA (no split, nothing to pre-split).
static rgcn_status refresh_weight_fragments(rgcn_ctx* c) {
  if (c->gemm_mode == 0) return RGCN_OK;
  if (c->capturing ? c->frag_fresh : c->frag_version == c->weights_version) return RGCN_OK;
  std::vector<PresplitJob> jobs;
  const int d = c->d, Bd = c->B * c->d;
  for (int l = 1; l <= c->L; ++l) {
    LayerBufs& lb = c->layers[l];
    if (!lb.wself_nn || !lb.wself_nt) continue;
    jobs.push_back(PresplitJob{lb.wself, lb.wself_nn, d, d, d, 0});      // H . W_self:      B (k, n) = W[k][n]
    jobs.push_back(PresplitJob{lb.wself, lb.wself_nt, d, d, d, 1});      // dS . W_self^T:   B (k, n) = W[n][k]
    if (c->kind == RGCN_KIND_BASIS && lb.wrel_nn && lb.wrel_nt)
      for (int g = 0; g < 2; ++g) {
        const float* W = lb.wrel + (size_t)g * Bd * d;                   // W'_dir [B.d, d]
        jobs.push_back(PresplitJob{W, static_cast<char*>(lb.wrel_nn) + 16 * g * gemm_bfrag_words(Bd, d), d, Bd, d, 0});
        jobs.push_back(PresplitJob{W, static_cast<char*>(lb.wrel_nt) + 16 * g * gemm_bfrag_words(d, Bd), d, d, Bd, 1});
      }
  }
  if (!jobs.empty()) RGCN_TRY(gemm_presplit_b(c, jobs.data(), (int)jobs.size()));
  c->frag_fresh = true;
  c->frag_version = c->capturing ? ~0ull : c->weights_version;
  return RGCN_OK;
}
// the self-loop products: one group, W_self as the pre-split B operand (forward: [k][n]; dH: used transposed, [n][k])
static rgcn_status self_loop_batch(rgcn_ctx* c, int l, bool transposed, GemmBatch* b) {
  *b = GemmBatch();
  RGCN_TRY(refresh_weight_fragments(c));
  if (c->gemm_mode != 0) b->bfrag = transposed ? c->layers[l].wself_nt : c->layers[l].wself_nn;
  b->wide = transposed ? 0 : 1;
  return RGCN_OK;
}

// Basis kind: the two direction groups of a batched GEMM over the (row, direction) units of the current graph; the
// group's extent (rows of A / C, or the depth of dW') is the direction's unit count, read on the device.
static GemmBatch basis_batch(const rgcn_ctx* c, size_t strideA, size_t strideB, size_t strideC, bool limit_on_k) {
  GemmBatch b;
  b.groups = 2;
  b.strideA = strideA; b.strideB = strideB; b.strideC = strideC;
  b.limit = c->g.unit_ptr + c->V;
  b.limit_stride = c->V + 1;
  b.limit_on_k = limit_on_k ? 1 : 0;
  return b;
}
static double basis_unit_share(rgcn_ctx* c) { return basis_units(c) / (2.0 * c->V); }

rgcn_status fwd_layer_partial(rgcn_ctx* c, int l) {
  if (l < 1 || l > c->L) RGCN_FAIL(c, RGCN_ERR_INVALID, "layer out of range");
  const float* Hin = c->H[l - 1];
  const int d = c->d, V = c->V;
  const int lo = c->world > 1 ? c->row_lo : 0, hi = c->world > 1 ? c->row_hi : V;
  float* dst = c->world > 1 ? c->exch : c->H[l];
  const double Mmsg = 2.0 * c->g.E / c->world;
  GemmBatch sb;
  RGCN_TRY(self_loop_batch(c, l, false, &sb));
  if (c->kind == RGCN_KIND_BLOCK && rows_layer(c)) {
    // S = H . W_self, then ONE kernel: H' = relu(dropout(S) + sum over the row's messages of n W_r H[src]) straight from
    // the incidence CSR (no message buffer)
    // (sharded run: the self-loop GEMM covers this rank's row shard, the kernel walks the rank's own messages and writes
    // the PARTIAL pre-activations -- the self-loop term inside the shard only, no relu -- for the reduce-scatter that
    // follows)
    RGCN_TRY(gemm_f32(c, "gemm_self_fwd", true, false, hi - lo, d, d, Hin + (size_t)lo * d, d, c->layers[l].wself, d,
                      c->self_buf + (size_t)lo * d, d, 1, &sb));
    RGCN_TRY(wait_gather(c));
    CombineArgs a;
    a.add = nullptr; a.msg = nullptr; a.row_ptr = nullptr; a.long_rows = nullptr; a.nlong = nullptr;
    a.out = dst; a.out2 = nullptr; a.base = c->self_buf; a.gate = nullptr; a.V = V; a.d = d;
    a.relu = (c->world == 1 && l < c->L) ? 1 : 0;
    a.row_lo = lo; a.row_hi = hi;
    a.drop = make_drop(c, l, true);
    a.drop2 = make_drop(c, l, false);
    RGCN_TRY(block_rows(c, "block_rows_fwd", l, false, Hin, a));
  } else if (c->kind == RGCN_KIND_BLOCK) {
    // Two-kernel form.  The relational messages (HBM-bound) run beside the self-loop GEMM.  A stream that blocks on
    // another stream's event resumes ~10 us after the event fires, so the chain that continues (the combine) stays on
    // the stream of the kernel that finishes LAST: the messages on the main stream, the (shorter) GEMM forked.
    {   // self-loop: S = H . W_self  (rows of this rank's shard)
      StreamScope side(c, 0);
      RGCN_TRY(gemm_f32(c, "gemm_self_fwd", true, false, hi - lo, d, d, Hin + (size_t)lo * d, d,
                        c->layers[l].wself, d, c->self_buf + (size_t)lo * d, d, 1, &sb));
    }
    RGCN_TRY(wait_gather(c));
    RGCN_TRY(block_msg_forward(c, l, Hin, c->msgbuf));
    RGCN_TRY(stream_join(c, 0));
    CombineArgs a;
    a.add = nullptr;
    a.out = dst; a.out2 = nullptr; a.base = c->self_buf; a.msg = c->g.E > 0 ? c->msgbuf : nullptr;
    a.row_ptr = c->g.row_ptr; a.long_rows = c->g.long_rows; a.nlong = c->g.nlong; a.gate = nullptr; a.V = V; a.d = d;
    a.relu = (c->world == 1 && l < c->L) ? 1 : 0;
    a.row_lo = lo; a.row_hi = hi;
    a.drop = make_drop(c, l, true);
    a.drop2 = make_drop(c, l, false);
    RGCN_TRY(combine(c, "combine_fwd", a, 4.0 * d * (2.0 * V + Mmsg) + 4.0 * V));
  } else {
    // aggregate first, per (row, direction) unit: Zc[(v,dir),b,:] = sum n C[rel,b] H[src];
    // pre[v] = dropout(H.W_self)[v] + sum_dir Zc[(v,dir)] . W'_dir  -- two groups of one batched GEMM over the units
    const int Bd = c->B * d;
    // the self-loop GEMM needs the layer input only: it runs on side stream 1 beside the aggregation (HBM-bound) and
    // then beside the basis GEMM
    {
      StreamScope side(c, 1);
      RGCN_TRY(gemm_f32(c, "gemm_self_fwd", true, false, hi - lo, d, d, Hin + (size_t)lo * d, d,
                        c->layers[l].wself, d, c->self_buf + (size_t)lo * d, d, 1, &sb));
    }
    RGCN_TRY(wait_gather(c));
    RGCN_TRY(basis_aggregate_forward(c, l, Hin, c->zsave[l]));
    GemmBatch gb = basis_batch(c, (size_t)V * Bd, (size_t)Bd * d, (size_t)V * d, false);
    RGCN_TRY(refresh_weight_fragments(c));
    if (c->gemm_mode != 0) gb.bfrag = c->layers[l].wrel_nn;
    gb.strideBfrag = gemm_bfrag_words(Bd, d);
    gb.wide = 1;
    RGCN_TRY(gemm_f32(c, "gemm_basis_fwd", true, false, V, d, Bd, c->zsave[l], Bd, c->layers[l].wrel, d,
                      c->aggbuf, d, 1, &gb, basis_unit_share(c)));
    RGCN_TRY(stream_join(c, 1));
    CombineArgs a;
    a.add = nullptr;
    a.add_units = c->aggbuf; a.unit_ptr = c->g.unit_ptr;
    a.out = dst; a.out2 = nullptr; a.base = c->self_buf; a.msg = nullptr; a.row_ptr = nullptr;
    a.long_rows = nullptr; a.nlong = nullptr; a.gate = nullptr; a.V = V; a.d = d;
    a.relu = (c->world == 1 && l < c->L) ? 1 : 0;
    a.row_lo = lo; a.row_hi = hi;
    a.drop = make_drop(c, l, true);
    a.drop2 = make_drop(c, l, false);
    RGCN_TRY(combine(c, "combine_fwd", a, 4.0 * d * (2.0 * V + 2.0 * V * basis_unit_share(c)) + 8.0 * V));
  }
  return RGCN_OK;
}

rgcn_status fwd_layer_finish(rgcn_ctx* c, int l) {
  if (l < 1 || l > c->L) RGCN_FAIL(c, RGCN_ERR_INVALID, "layer out of range");
  if (c->world > 1)
    RGCN_TRY(relu_copy(c, c->exch, c->H[l], (int64_t)c->V * c->d, l < c->L ? 1 : 0));
  if (l == c->L) c->fwd_done = true;
  return RGCN_OK;
}

// ---------------------------------------------------------------- backward
// ds_ready: dcodes * dropout of the top layer, already written by the producer of dcodes (the device decoder does,
// inside a train step): the scale-and-copy pass over [V,d] is skipped
rgcn_status bwd_begin(rgcn_ctx* c, const float* dcodes_dev, const float* ds_ready) {
  if (!c->fwd_done) RGCN_FAIL(c, RGCN_ERR_STATE, "rgcn_backward needs a completed rgcn_forward on the current graph");
  if (!dcodes_dev) RGCN_FAIL(c, RGCN_ERR_INVALID, "dcodes is NULL");
  RGCN_TRY(join_abandoned_side_work(c));      // (an abandoned pass's side kernels read the dS / D buffers this one rewrites)
  c->bwd_layer = c->L;
  c->bwd_D = dcodes_dev;
  DropSpec ds = make_drop(c, c->L, true);
  if (ds.mode != DROP_NONE && ds_ready != nullptr) {
    c->bwd_dS = ds_ready;
  } else if (ds.mode != DROP_NONE) {
    RGCN_TRY(scale_dropout(c, dcodes_dev, c->dsbuf[c->L & 1], ds));
    c->bwd_dS = c->dsbuf[c->L & 1];
  } else {
    c->bwd_dS = dcodes_dev;
  }
  return RGCN_OK;
}

rgcn_status bwd_layer_partial(rgcn_ctx* c, int l) {
  if (l != c->bwd_layer || l < 1) RGCN_FAIL(c, RGCN_ERR_STATE, "backward layers must run L..1 in order");
  const float* Hin = c->H[l - 1];
  const int d = c->d, V = c->V;
  const int lo = c->world > 1 ? c->row_lo : 0, hi = c->world > 1 ? c->row_hi : V;
  const int rows = hi - lo;
  LayerBufs& lb = c->layers[l];
  const double Mmsg = 2.0 * c->g.E / c->world;
  const bool narrow_dw = c->kind == RGCN_KIND_BASIS || rows >= 32768;       // (see auto_split_k)

  // epilogue shared by both kinds: (self-loop gradient + relational gradient) -> relu' -> next D / dS
  CombineArgs a;
  a.add = nullptr;
  a.base = c->self_buf; a.msg = nullptr; a.row_ptr = nullptr; a.long_rows = nullptr; a.nlong = nullptr;
  a.V = V; a.d = d; a.relu = 0; a.row_lo = lo; a.row_hi = hi;
  a.drop = make_drop(c, l, false);
  if (c->world == 1) {
    a.out = (l - 1 == 0) ? c->g_emb : c->dbuf[(l - 1) & 1];
    a.colsum = (l - 1 == 0) ? 1 : 0;       // db_emb = the column sums of dL/dH0 * relu'(H0): partials from the kernel that writes it
    a.drop2 = make_drop(c, l - 1, l - 1 >= 1);
    a.out2 = a.drop2.mode != DROP_NONE ? c->dsbuf[(l - 1) & 1] : nullptr;
    a.gate = Hin;
  } else {
    a.out = c->exch; a.out2 = nullptr; a.gate = nullptr; a.drop2 = make_drop(c, l, false);
  }

  if (c->dw_pending) {       // the previous layer's weight-gradient kernel reads the D buffer this layer may overwrite
    RGCN_TRY(stream_join(c, 0));
    c->dw_pending = false;
  }
  // the three dense / relation-weight pieces every block schedule below is made of
  auto self_dw = [&]() {      // dW_self = H_in^T . dS   (split-K)
    return gemm_f32(c, "gemm_self_dw", false, false, d, d, rows, Hin + (size_t)lo * d, d, c->bwd_dS + (size_t)lo * d, d,
                    lb.gwself, d, auto_split_k(d, d, rows, narrow_dw));
  };
  GemmBatch sbt;
  RGCN_TRY(self_loop_batch(c, l, true, &sbt));
  auto self_dh = [&]() {      // G = dS . W_self^T
    return gemm_f32(c, "gemm_self_dh", true, true, rows, d, d, c->bwd_dS + (size_t)lo * d, d, lb.wself, d,
                    c->self_buf + (size_t)lo * d, d, 1, &sbt);
  };
  bool defer_joins = false;
  if (c->kind == RGCN_KIND_BLOCK && rows_layer(c)) {
    // Row gradients: the single-pass kernel behind G = dS . W_self^T.  Relation-weight gradients (dW_r = sum n g (x) x,
    // relation-major, two row gathers per message): k_block_msg_bwd<dW only> + its slab reduce.  ONE schedule per
    // situation, each the measured best of round 4's A/Bs (profiles/r04_rowmajor_spmm_ab.md, r04_block_forms_ab.txt):
    //   minibatch scale, one GPU, side streams on:  dW_self forked BEFORE dH is launched (MFMA beside MFMA: the second
    //       GEMM fills the slots the first leaves idle, 456 workgroups on 512, and its tail), the relation-weight kernels
    //       forked BEHIND dH, beside the row-gradient kernel (two gather kernels share the chip better than either does
    //       with a GEMM), joined at the end of the layer: 0.553-0.556 ms per step against 0.597-0.599 as a chain (round 5,
    //       same box: dW_self forked BEHIND dH instead, beside the row-gradient kernel: 0.567-0.578 against 0.541-0.547);
    //   the same inside a capture (a captured step is a chain, rgcn_capture_begin) or with the side streams off: the
    //       chain, with the one fork a replayed graph gains from -- the slab reduce + dW_self beside dH;
    //   full-graph scale or a sharded run: the relation-weight kernels on side stream 0 from the start of the layer
    //       (at 272,115 edges they are five GEMMs long; a sharded run's all-gather rides on side stream 1), joined
    //       before the next layer overwrites D.
    const bool minibatch = c->world == 1 && c->g.E <= 65536;
    if (minibatch && c->use_aux) {
      {
        StreamScope side(c, 1);
        RGCN_TRY(self_dw());
      }
      RGCN_TRY(self_dh());
      {
        StreamScope side(c, 0);
        RGCN_TRY(block_msg_backward(c, l, Hin, c->bwd_D, nullptr));
        RGCN_TRY(block_dw_reduce(c, l));
      }
      RGCN_TRY(block_rows(c, "block_rows_bwd", l, true, c->bwd_D, a));
      // Layer 2's side kernels read H_1, D_2 and dS_2 and write their own slabs and gradients.  Layer 1, next, overwrites
      // none of those (its rows go to g_emb) and queues its side kernels behind them in stream order: its joins cover
      // both layers, and the main stream saves two waits between the layers.
      defer_joins = l == 2;
      if (!defer_joins) RGCN_TRY(stream_join_both(c));
    } else if (minibatch) {
      RGCN_TRY(block_msg_backward(c, l, Hin, c->bwd_D, nullptr));
      {
        StreamScope side(c, 1, /*in_capture=*/true);
        RGCN_TRY(block_dw_reduce(c, l));
        RGCN_TRY(self_dw());
      }
      RGCN_TRY(self_dh());
      RGCN_TRY(block_rows(c, "block_rows_bwd", l, true, c->bwd_D, a));
    } else {
      {
        StreamScope side(c, 0);
        RGCN_TRY(wait_gather(c));          // D_l of every row (sharded run: gathered beside the self-loop GEMMs)
        RGCN_TRY(block_msg_backward(c, l, Hin, c->bwd_D, nullptr));
        RGCN_TRY(block_dw_reduce(c, l));
        c->dw_pending = side.active;
      }
      {
        StreamScope side(c, 1);
        RGCN_TRY(self_dw());
      }
      RGCN_TRY(self_dh());
      RGCN_TRY(wait_gather(c));
      RGCN_TRY(block_rows(c, "block_rows_bwd", l, true, c->bwd_D, a));
    }
  } else if (c->kind == RGCN_KIND_BLOCK) {
    {   // two-kernel form: the relational gradient kernels (HBM-bound) on a side stream beside the self-loop GEMMs
      StreamScope side(c, 0);
      RGCN_TRY(wait_gather(c));          // D_l of every row (sharded run: gathered beside the self-loop GEMMs)
      RGCN_TRY(block_msg_backward(c, l, Hin, c->bwd_D, c->msgbuf));
      // the combine below needs only the message rows: mark the join point here, then let the
      // per-relation dW reduction trail behind on the side stream
      if (side.active) RGCN_HIP(c, hipEventRecord(c->ev_join[0], c->aux[0]));
      RGCN_TRY(block_dw_reduce(c, l));
    }
    RGCN_TRY(self_dh());
    {   // dW_self on side stream 1, queued behind the dH GEMM
      StreamScope side(c, 1);
      RGCN_TRY(self_dw());
    }
    if (c->use_aux) RGCN_HIP(c, hipStreamWaitEvent(c->main_stream, c->ev_join[0], 0));
    a.msg = c->g.E > 0 ? c->msgbuf : nullptr;
    a.row_ptr = c->g.row_ptr;
    a.long_rows = c->g.long_rows;
    a.nlong = c->g.nlong;
    RGCN_TRY(combine(c, "combine_bwd", a, 4.0 * d * ((a.out2 ? 4.0 : 3.0) * V + Mmsg) + 4.0 * V));
  } else {
    const int Bd = c->B * d;
    // The upstream rows of the units, compacted like Zc (the row operand of dZ, the depth operand of dW')
    RGCN_TRY(wait_gather(c));            // D_l of every row
    RGCN_TRY(basis_gather_units(c, c->bwd_D, c->aggbuf));
    // The four dense contractions of the layer depend on D_l / dS_l only.  Two of them -- the weight gradients dW_self =
    // H^T.dS and dW'_dir = Zc_dir^T.D[units], needed at the end of the pass -- go to side stream 1, the two whose
    // products the gather kernels below consume (dH's self-loop part, dZ) stay on the main stream: the pairs fill each
    // other's idle CU slots and tails.
    {
      StreamScope side(c, 1);
      RGCN_TRY(self_dw());
      // dW'_dir = Zc_dir^T . Dc_dir   ([B.d, units] x [units, d], split over the units; two groups)
      const GemmBatch gk = basis_batch(c, (size_t)V * Bd, (size_t)V * d, (size_t)Bd * d, true);
      RGCN_TRY(gemm_f32(c, "gemm_basis_dw", false, false, Bd, d, V, c->zsave[l], Bd, c->aggbuf, d, lb.grel, d,
                        auto_split_k(2 * Bd, d, V), &gk, basis_unit_share(c)));
    }
    RGCN_TRY(self_dh());
    // dZc_dir = Dc_dir . W'_dir^T   ([units, d] x [d, B.d], two groups)
    GemmBatch gm = basis_batch(c, (size_t)V * d, (size_t)Bd * d, (size_t)V * Bd, false);
    RGCN_TRY(refresh_weight_fragments(c));
    if (c->gemm_mode != 0) gm.bfrag = lb.wrel_nt;
    gm.strideBfrag = gemm_bfrag_words(d, Bd);
    RGCN_TRY(gemm_f32(c, "gemm_basis_dz", true, true, V, Bd, d, c->aggbuf, d, lb.wrel, d, c->msgbuf2, Bd, 1, &gm,
                      basis_unit_share(c)));
    RGCN_TRY(basis_dcoef(c, l, Hin, c->msgbuf2));
    RGCN_TRY(basis_backward_gather(c, l, c->msgbuf2, a, true));
  }
  // the dW_self GEMM must be done before the next layer overwrites its dS operand / the caller
  // all-reduces gwself
  if (!defer_joins) RGCN_TRY(stream_join(c, 1));
  return RGCN_OK;
}

rgcn_status bwd_layer_finish(rgcn_ctx* c, int l) {
  if (l != c->bwd_layer || l < 1) RGCN_FAIL(c, RGCN_ERR_STATE, "backward layers must run L..1 in order");
  float* out = (l - 1 == 0) ? c->g_emb : c->dbuf[(l - 1) & 1];
  DropSpec d2 = make_drop(c, l - 1, l - 1 >= 1);
  float* out2 = d2.mode != DROP_NONE ? c->dsbuf[(l - 1) & 1] : nullptr;
  if (c->world > 1) {
    CombineArgs a;
    a.add = nullptr;
    a.out = out; a.out2 = out2; a.base = c->exch; a.msg = nullptr; a.row_ptr = nullptr; a.long_rows = nullptr; a.nlong = nullptr;
    a.gate = c->H[l - 1]; a.V = c->V; a.d = c->d; a.relu = 0; a.row_lo = 0; a.row_hi = c->V;
    a.drop = make_drop(c, l, false);
    a.drop2 = d2;
    RGCN_TRY(combine(c, "combine_bwd_finish", a, 4.0 * c->d * (out2 ? 4.0 : 3.0) * c->V));
  }
  c->bwd_D = out;
  c->bwd_dS = out2 ? out2 : out;
  c->bwd_layer = l - 1;
  return RGCN_OK;
}

rgcn_status bwd_end(rgcn_ctx* c) {
  if (c->bwd_layer != 0) RGCN_FAIL(c, RGCN_ERR_STATE, "rgcn_backward_end before all layers ran");
  // AffineTransform: dW_emb = dH0 * (H0 > 0) is already in g_emb; db_emb = column sums
  // (single-pass block layer on one GPU: the bottom layer's row-gradient kernel left the column sums of its rows as partials)
  if (c->colsum_parts > 0) RGCN_TRY(column_sum_finish(c, c->gb_emb, c->colsum_parts, c->d));
  else RGCN_TRY(column_sum(c, c->g_emb, c->gb_emb, c->V, c->d));
  c->colsum_parts = 0;
  c->dw_pending = false;
  return stream_join(c, 0);   // trailing per-relation dW reductions
}

// Sharded run with a communicator, forward exchange of layer l: the partial pre-activations are reduce-scattered,
// this rank applies the relu to ITS rows only, and the finished rows are all-gathered on side stream 1 while the main
// stream goes on (the next self-loop GEMM needs the rank's own rows, nothing else).
static rgcn_status fwd_exchange(rgcn_ctx* c, int l) {
  const int64_t chunk = (int64_t)c->shard_rows * c->d;
  RGCN_TRY(comm_reduce_scatter(c, c->exch, chunk));
  const size_t off = (size_t)c->rank * chunk;
  RGCN_TRY(relu_copy(c, c->exch + off, c->H[l] + off, chunk, l < c->L ? 1 : 0));
  RGCN_TRY(gather_rows(c, c->H[l]));
  if (l == c->L) {
    RGCN_TRY(wait_gather(c));          // the codes: every row, on the main stream
    c->fwd_done = true;
  }
  return RGCN_OK;
}

// Backward exchange of layer l: partial dH reduce-scattered, (relu', dropout) on the rank's rows, D_{l-1} gathered.
static rgcn_status bwd_exchange(rgcn_ctx* c, int l) {
  const int64_t chunk = (int64_t)c->shard_rows * c->d;
  RGCN_TRY(comm_reduce_scatter(c, c->exch, chunk));
  float* out = (l - 1 == 0) ? c->g_emb : c->dbuf[(l - 1) & 1];
  DropSpec d2 = make_drop(c, l - 1, l - 1 >= 1);
  float* out2 = d2.mode != DROP_NONE ? c->dsbuf[(l - 1) & 1] : nullptr;
  CombineArgs a;
  a.add = nullptr;
  a.out = out; a.out2 = out2; a.base = c->exch; a.msg = nullptr; a.row_ptr = nullptr; a.long_rows = nullptr; a.nlong = nullptr;
  a.gate = c->H[l - 1]; a.V = c->V; a.d = c->d; a.relu = 0; a.row_lo = c->row_lo; a.row_hi = c->row_hi;
  a.v_begin = c->row_lo; a.v_count = c->row_hi - c->row_lo;
  a.drop = make_drop(c, l, false);
  a.drop2 = d2;
  if (a.v_count > 0) RGCN_TRY(combine(c, "combine_bwd_finish", a, 4.0 * c->d * (out2 ? 4.0 : 3.0) * a.v_count));
  RGCN_TRY(gather_rows(c, out));
  c->bwd_D = out;
  c->bwd_dS = out2 ? out2 : out;      // own rows only: all the row-sharded self-loop GEMMs read
  c->bwd_layer = l - 1;
  return RGCN_OK;
}

rgcn_status forward_all(rgcn_ctx* c, int train, uint64_t seed, const uint8_t* masks) {
  if (c->world > 1 && !c->comm)
    RGCN_FAIL(c, RGCN_ERR_STATE, "world > 1: call rgcn_comm_init first (or drive the phase API yourself)");
  RGCN_TRY(fwd_begin(c, train, seed, masks));
  for (int l = 1; l <= c->L; ++l) {
    RGCN_TRY(fwd_layer_partial(c, l));
    if (c->world > 1) RGCN_TRY(fwd_exchange(c, l));
    else RGCN_TRY(fwd_layer_finish(c, l));
  }
  return RGCN_OK;
}

rgcn_status backward_all(rgcn_ctx* c, const float* dcodes_dev, const float* ds_ready) {
  if (c->world > 1 && !c->comm)
    RGCN_FAIL(c, RGCN_ERR_STATE, "world > 1: call rgcn_comm_init first (or drive the phase API yourself)");
  RGCN_TRY(bwd_begin(c, dcodes_dev, ds_ready));
  for (int l = c->L; l >= 1; --l) {
    RGCN_TRY(bwd_layer_partial(c, l));
    if (c->world > 1) RGCN_TRY(bwd_exchange(c, l));
    else RGCN_TRY(bwd_layer_finish(c, l));
  }
  if (c->world > 1) {
    RGCN_TRY(wait_gather(c));          // dW_emb of every row (optimizer, column sums)
    // W_self (and basis W') gradients of all layers: partial sums over the row / relation shards, one collective
    RGCN_TRY(comm_allreduce(c, c->repl_grads, (int64_t)c->repl_grads_floats));
  }
  return bwd_end(c);
}

}  // namespace rgcn
