/*
 * rgcn.h -- C ABI of librgcn.so: the MI355X-native R-GCN encoder hot path
 * (forward + backward of MichSchli/RelationPrediction's `gcn_basis` encoder
 * chain: AffineTransform -> L x {ConcatGcn | BasisGcn}).
 *
 * This is the drop-in boundary.  In the reference the
 * Python plugin chain (code/model.py, code/encoders, code/common/model_builder.py)
 * builds a TF-1.4 graph and crosses into native code at `session.run`
 * (code/optimization/optimize.py:81-88 for training, code/model.py:56,69,81 for
 * scoring).  Here the same plugin chain (package `relationprediction_amd`)
 * crosses into this library through ctypes.  Each entry point names the
 * reference interface it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain C: `extern "C"`, no C++/torch types, no exceptions cross the boundary;
 *   - every call returns an rgcn_status (0 = ok); rgcn_last_error() has the text;
 *   - host pointers are BORROWED for the duration of the call only;
 *   - device memory is owned by the context; `*_device` variants take device
 *     pointers (HIP, same device) that the caller owns and keeps alive until
 *     the next rgcn_sync();
 *   - one context per (process, GPU); a context is not thread-safe; calls are
 *     stream-ordered on the context's own HIP stream and asynchronous unless
 *     they return host data;
 *   - all floating point is float32, all indices int32 (reference:
 *     code/extras/graph_representations.py:174, code/common/shared_functions.py:17).
 */
#ifndef RGCN_H_
#define RGCN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: these declarations are its whole dynamic symbol table */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define RGCN_ABI_VERSION 1

typedef struct rgcn_ctx rgcn_ctx;
typedef int32_t rgcn_status;

enum {
  RGCN_OK = 0,
  RGCN_ERR_INVALID = 1,     /* bad argument / out-of-range index / wrong size */
  RGCN_ERR_HIP = 2,         /* a HIP runtime call failed */
  RGCN_ERR_RCCL = 3,        /* an RCCL call failed / librccl not loadable */
  RGCN_ERR_STATE = 4,       /* call order violated (e.g. backward before forward) */
  RGCN_ERR_UNSUPPORTED = 5, /* configuration outside what the kernels implement */
  RGCN_ERR_NOMEM = 6
};

/* Encoder.Concatenation=Yes -> ConcatGcn (block-diagonal), else BasisGcn:
 * code/common/model_builder.py:291-294. */
enum { RGCN_KIND_BLOCK = 0, RGCN_KIND_BASIS = 1 };

/* Neighbour normalisation of the incidence matrices, normalization=('global', ...)
 * (code/extras/graph_representations.py:82-93,122-133).  INTENDED = 1/deg(row of this edge);
 * TF_AS_EXECUTED = SURVEY.md section 9 H1 (tf.sparse_softmax on non-canonical indices returns the
 * values in sorted-row order); NONE = the 'none' branch (:70-81). */
enum { RGCN_NORM_INTENDED = 0, RGCN_NORM_TF_AS_EXECUTED = 1, RGCN_NORM_NONE = 2 };

/* Buffers readable through rgcn_read_buffer (tests / the sharding exchange). */
enum {
  RGCN_BUF_EXCHANGE = 0,   /* [V,d] buffer a multi-GPU run all-reduces (partial pre-activation / partial dH) */
  RGCN_BUF_SELF = 1,       /* [V,d] self-loop product H.W_self of the last layer run */
  RGCN_BUF_DSELF_EXCHANGE = 2, /* [d,d] partial dW_self a multi-GPU run all-reduces */
  RGCN_BUF_INDEG = 3,      /* int32 [V] in-degree inside the fed graph  */
  RGCN_BUF_OUTDEG = 4,     /* int32 [V] out-degree inside the fed graph */
  RGCN_BUF_ROWPTR = 5,     /* int32 [V+1] incidence CSR offsets (owned relations only) */
  RGCN_BUF_NORM_EXCHANGE = 6,   /* float [1] squared norm of this rank's relation-sharded gradients (optimizer phases) */
  RGCN_BUF_DBASIS_EXCHANGE = 7, /* basis kind: [2,B,d,d] partial gradient of the replicated basis tensors of the
                                   backward layer in flight, which a multi-GPU run all-reduces */
  /* the two orderings graph preparation produces (library's own stable sort, csr_sort.hip), for inspection:
   * incidence i < E = edge i seen from its object, i >= E = edge i-E seen from its subject; message m likewise with
   * directed relation r (m < E) or R + r */
  RGCN_BUF_PERM_VERTEX = 8,     /* int32 [2E] incidence ids in incidence-CSR order (by vertex, ties by id) */
  RGCN_BUF_PERM_RELATION = 9,   /* int32 [2E] message ids in message-list order (by directed relation, ties by id) */
  RGCN_BUF_RANK_ENERGIES = 10   /* float [reserved queries, V] energies of the last chunk rgcn_rank_device scored (the
                                   float half of the ranking; the counts are integer work on exactly these values) */
};

/*
 * Replaces the settings the reference components parse in their constructors:
 * Model.__init__ (code/model.py:17-25: EntityCount, RelationCount), ConcatGcn/BasisGcn.parse_settings
 * (gcn_basis_concat.py:10-15, gcn_basis.py:10-13: DropoutKeepProbability, NumberOfBasisFunctions),
 * build_encoder (model_builder.py:121-184: InternalEncoderDimension, NumberOfLayers).
 */
typedef struct rgcn_config {
  int32_t abi_version;    /* must be RGCN_ABI_VERSION */
  int32_t device;         /* HIP device ordinal */
  int32_t num_entities;   /* V = EntityCount */
  int32_t num_relations;  /* R = RelationCount */
  int32_t dim;            /* d = InternalEncoderDimension (= CodeDimension, UseOutputTransform=No) */
  int32_t num_layers;     /* L = NumberOfLayers */
  int32_t kind;           /* RGCN_KIND_* */
  int32_t num_bases;      /* NumberOfBasisFunctions: block count nb (BLOCK, d % nb == 0) or B (BASIS) */
  float   keep_prob;      /* DropoutKeepProbability (self-loop dropout, train mode only) */
  int32_t norm_mode;      /* RGCN_NORM_* */
  int64_t max_edges;      /* capacity: largest E ever passed to rgcn_set_graph* */
  int32_t rank;           /* relation-sharding rank in [0, world) */
  int32_t world;          /* number of relation shards (1 = single GPU) */
  int32_t reserved;       /* must be 0 */
} rgcn_config;

/* ---- lifecycle ------------------------------------------------------------------------------ */

int32_t rgcn_abi_version(void);

/* Replaces tf.Session() + Model.initialize_train() variable allocation
 * (code/train.py:258,278; code/model.py:93-94).  Weights start as zeros: the host plugin chain
 * draws the reference's numpy initialisers (shared_functions.py:16-29) and pushes them. */
rgcn_status rgcn_create(const rgcn_config* cfg, rgcn_ctx** out);
rgcn_status rgcn_destroy(rgcn_ctx* ctx);

/* Text of the last error on this context (ctx == NULL: last error of a failed rgcn_create). */
const char* rgcn_last_error(const rgcn_ctx* ctx);

/* Wait for everything queued on the context's stream; also surfaces asynchronous device-side
 * validation failures (out-of-range vertex / relation ids in a device-resident graph). */
rgcn_status rgcn_sync(rgcn_ctx* ctx);

/* ---- parameters (the encoder part of Model.get_weights(), code/model.py:96-97) ----------------
 * Index order = reference order, innermost component first:
 *   W_emb [V,d], b_emb [d]                                  (affine_transform.py:30-31)
 *   per layer l=1..L:  BLOCK: W_f [R,nb,sd,sd], W_b [R,nb,sd,sd], W_self [d,d], b [d]
 *                                                           (gcn_basis_concat.py:30-33)
 *                      BASIS: W_f [d,B,d], W_b [d,B,d], C_f [R,B], C_b [R,B], W_self [d,d], b [d]
 *                                                           (gcn_basis.py:33-37)
 * Host layouts are the reference's (row-major, shapes above); the device layout is private.
 * `b` is created but never used by the reference layers: it is stored, never read,
 * and its gradient is all zeros.  The LAST parameter is the decoder's W_relation [EntityCount, d]
 * (relation_embedding.py:15-18; only rows < RelationCount are ever used): the encoder path
 * does not touch it, the device decoder below does. */
int32_t     rgcn_param_count(const rgcn_ctx* ctx);
rgcn_status rgcn_param_info(const rgcn_ctx* ctx, int32_t index, char* name, int32_t name_cap,
                            int64_t shape[4], int32_t* ndim);
/* replaces tf.Variable(initializer) / Saver.restore: shared_functions.py:16-22, model.py:38 */
rgcn_status rgcn_set_param(rgcn_ctx* ctx, int32_t index, const float* host, int64_t count);
/* replaces Saver.save / session.run(variable): model.py:30-37 */
rgcn_status rgcn_get_param(rgcn_ctx* ctx, int32_t index, float* host, int64_t count);
/* replaces the tensors tf.gradients(loss, weights) returns: optimization/abstract.py:117-118 */
rgcn_status rgcn_get_grad(rgcn_ctx* ctx, int32_t index, float* host, int64_t count);

/* ---- graph (the `graph_edges` placeholder, graph_representations.py:173-177) -------------------
 * triples = int32 [E,3] rows (subject, relation, object) exactly as fed to the placeholder
 * (MessageGraph.process, :21-27).  The call builds, on the device, everything the reference derives
 * inside session.run: degrees, the 'global' normalisation values (:82-93,122-133), the
 * relation-sorted message list and the incidence CSR.  Edge dropout (GraphSplitSize, train.py:235-238)
 * is applied by the caller before the call: degrees are counted over the edges actually fed.
 * 0 <= E <= max_edges.  The host variant validates ids and returns RGCN_ERR_INVALID on a bad one. */
rgcn_status rgcn_set_graph(rgcn_ctx* ctx, const int32_t* triples_host, int64_t num_edges);
rgcn_status rgcn_set_graph_device(rgcn_ctx* ctx, const int32_t* triples_dev, int64_t num_edges);

/* Edge dropout of the reference's minibatch construction, on the device (code/train.py:233-238):
 *     graph_split_ids = np.random.choice(graph_batch_ids, size=int(GraphSplitSize * n), replace=False)
 * the message-passing graph is a uniformly random subset of EXACTLY `keep` of the n batch edges; only those are fed
 * to `graph_edges`, so degrees and normalisation are counted over the kept edges, while the decoder keeps
 * every batch edge as a positive (H9: pass the whole batch to the decoder / rgcn_train_step_minibatch_device).
 * batch_dev = int32 [n,3]; the draw is a function of (seed, n, keep) alone (counter-based keys, the `keep` smallest
 * win); keep_mask_dev (nullable) = uint8 [n] with exactly `keep` ones: the caller's set instead of the draw (how a
 * test injects the reference's choice).  The kept edges are compacted in batch order (the reference's come in the
 * random order of np.random.choice: same set distribution, other row order) and prepared like rgcn_set_graph_device;
 * rgcn_get_graph_edges returns the rows of the graph currently set ([keep,3]; synchronises). */
rgcn_status rgcn_set_graph_dropout_device(rgcn_ctx* ctx, const int32_t* batch_dev, int64_t num_edges, int64_t keep,
                                          uint64_t seed, const uint8_t* keep_mask_dev);
rgcn_status rgcn_get_graph_edges(rgcn_ctx* ctx, int32_t* host, int64_t count);

/* ---- forward: get_all_codes(mode) (message_gcn.py:44-79, affine_transform.py:63-83) ------------
 * train != 0 applies self-loop dropout (message_gcn.py:60-64).  The Bernoulli(keep_prob) draw is
 *   - dropout_masks_host != NULL: the caller's uint8 [L,V,d] 0/1 masks (parity testing), else
 *   - a counter-based generator keyed by (dropout_seed, layer, element) -- nothing is stored;
 *     rgcn_get_dropout_mask() re-materialises what the last forward used.
 * The result H_L [V,d] is the subject AND object code matrix (relation_embedding.py:23-25). */
rgcn_status rgcn_forward(rgcn_ctx* ctx, int32_t train, uint64_t dropout_seed,
                         const uint8_t* dropout_masks_host);
rgcn_status rgcn_get_codes(rgcn_ctx* ctx, float* host, int64_t count);              /* H_L  */
rgcn_status rgcn_get_activation(rgcn_ctx* ctx, int32_t layer, float* host, int64_t count); /* H_0..H_L */
rgcn_status rgcn_get_dropout_mask(rgcn_ctx* ctx, int32_t layer /*1..L*/, uint8_t* host, int64_t count);
const float* rgcn_codes_device(rgcn_ctx* ctx);                                      /* device H_L */

/* ---- backward: tf.gradients(loss, weights) restricted to the encoder (abstract.py:117-118) -----
 * dcodes = dL/dH_L [V,d] (the decoder's gradient w.r.t. subject+object codes, already summed).
 * Needs a preceding rgcn_forward on the same graph.  Gradients are then readable by rgcn_get_grad. */
rgcn_status rgcn_backward(rgcn_ctx* ctx, const float* dcodes_host, int64_t count);
rgcn_status rgcn_backward_device(rgcn_ctx* ctx, const float* dcodes_dev);

/* One whole 'step' of the BASELINE metric, fully asynchronous: graph prep + forward(train) +
 * backward from device-resident inputs.  Replaces one session.run of the encoder part of
 * TensorflowOptimizer.update_from_batch (optimize.py:81-88). */
rgcn_status rgcn_step_device(rgcn_ctx* ctx, const int32_t* triples_dev, int64_t num_edges,
                             int32_t train, uint64_t dropout_seed, const float* dcodes_dev);

/* ---- "next" rows: DistMult decoder, clip + Adam, whole train step on the device ----
 * X = int32 [N,3] (subject, relation, object) positives + sampled negatives, Y = float32 [N] labels, as
 * NegativeSampler.transform emits them (code/common/auxilliaries.py:13-33).  Replaces the decoder part of
 * the train graph: BilinearDiag.get_loss + local_get_regularization (code/decoders/bilinear_diag.py:27-34,
 * 63-69) and their tf.gradients.  Needs rgcn_decoder_reserve(max N) once and a completed rgcn_forward.
 * Leaves dL/dcodes in rgcn_dcodes_device() (feed it to rgcn_backward_device) and dL/dW_relation in the
 * gradient of the last parameter; rgcn_get_loss returns loss + regulariser (synchronises). */
rgcn_status rgcn_decoder_reserve(rgcn_ctx* ctx, int64_t max_triples);
rgcn_status rgcn_decoder_loss_backward_device(rgcn_ctx* ctx, const int32_t* x_dev, const float* y_dev,
                                              int64_t num_triples, float regularization_parameter);
const float* rgcn_dcodes_device(rgcn_ctx* ctx);
rgcn_status rgcn_get_loss(rgcn_ctx* ctx, double* loss);
/* ---- minibatch construction on the host ---------------------------------------------
 * sample_edge_neighborhood of the reference's train loop (code/train.py:133-139 adjacency, :161-198 sampler):
 * the same random process (vertex ~ free edge ends x touched, then a free incident edge uniformly; uniform
 * restart over vertices with free edges when nothing touched has any) in O(log V) per pick instead of the
 * reference's O(V).  Host memory in, host memory out, no context, no GPU.  triples = int32 [n,3] rows
 * (subject, relation, object) of the training graph; out_edge_ids = int32 [sample_size] distinct row ids.
 * sample_size > n is an error (the reference crashes there). */
typedef struct rgcn_sampler rgcn_sampler;
rgcn_status rgcn_sampler_create(const int32_t* triples, int64_t num_triples, int32_t num_entities,
                                rgcn_sampler** out);
void rgcn_sampler_destroy(rgcn_sampler* sampler);
rgcn_status rgcn_sampler_edge_neighborhood(rgcn_sampler* sampler, int64_t sample_size, uint64_t seed,
                                           int32_t* out_edge_ids);

/* The same sampler ON THE DEVICE (csrc/neighborhood.hip): sample_edge_neighborhood's process has the distribution of
 * first-passage percolation with an Exp(1) clock on every edge end (a touched vertex's free edge ends are equally
 * likely to be picked next = they race with memoryless clocks), its restart rule that of a uniform random vertex order
 * over the connected components -- so the batch is every edge of the components visited in full plus the edges of
 * least pick time of the component the budget runs out in: parallel Bellman-Ford sweeps over the adjacency, a radix
 * select and a stable compaction on the device, the component order from V hashes on the host inside the call.  Same
 * distribution as code/train.py:161-198 (held to it by tests/test_gpu_sampler.py), another random stream, a function of
 * `seed` alone, no batch built on the host and no upload.  rgcn_neighborhood_reserve hands the training graph over once
 * (code/train.py:133-139 builds the adjacency lists once): ids are validated, components and adjacency are built on
 * the host and copied; rgcn_sample_neighborhood_device writes the drawn rows int32 [sample_size,3] (in edge order) to
 * batch_out_dev, asynchronously on the main stream or -- on_prefetch_stream != 0 -- on the stream
 * rgcn_prefetch_graph*_device works on, so that "draw the next batch, drop edges, prepare its graph" runs beside the
 * current step.  sample_size above the number of training triples is an error; a graph whose sweeps do not
 * settle within the budget (sized at rgcn_neighborhood_reserve from the graph's diameter; graphs several thousand hops
 * across exceed its ceiling) is refused at the next synchronising call, never answered wrongly. */
rgcn_status rgcn_neighborhood_reserve(rgcn_ctx* ctx, const int32_t* triples_host, int64_t num_triples);
rgcn_status rgcn_sample_neighborhood_device(rgcn_ctx* ctx, int64_t sample_size, uint64_t seed, int32_t* batch_out_dev,
                                            int32_t on_prefetch_stream);

/* NegativeSampler.transform (code/common/auxilliaries.py:13-33) on the device: x_out = int32 [n*(rate+1), 3], the batch
 * tiled rate+1 times with, in every row after the first n, the object (fair coin) or else the subject replaced by a
 * uniform entity id; y_out = float32 [n*(rate+1)], 1 for the first n rows, 0 after.  Same layout and distribution as
 * the reference's numpy code, other random stream (counter-based, a function of `seed`).  Ids of `batch` are NOT
 * validated here; the decoder rejects out-of-range ids when it consumes x_out.  Asynchronous, capturable. */
rgcn_status rgcn_negative_sample_device(rgcn_ctx* ctx, const int32_t* batch_dev, int64_t n, int32_t rate, uint64_t seed,
                                        int32_t* x_out_dev, float* y_out_dev);

/* ---- evaluation: raw and filtered link-prediction ranks -------------------------------
 * Replaces Model.score_all_subjects / score_all_objects + Scorer.evaluate_mrr + MrrScore.append_line
 * (code/model.py:59-81, code/decoders/bilinear_diag.py:51-61, code/common/evaluation.py:148-153,349-389).
 * For every query triple x[i] = (s, r, o): score every entity e as sigmoid(sum_k codes[e,k] (W_relation[r] *
 * codes[o])[k]) (predict_object = 0: rank the subject) or sigmoid(sum_k (codes[s] * W_relation[r])[k] codes[e,k])
 * (predict_object = 1) on the codes of the last rgcn_forward, then
 *   raw_rank[i]      = #{e : score[e] >= score[gold]}
 *   filtered_rank[i] = raw_rank[i] - #{e in filter_idx[filter_ptr[i] : filter_ptr[i+1]] : score[e] >= score[gold]} + 1
 * (the filter list holds the known completions of the pair, the gold entity among them).  Comparisons are made
 * on fp32 sigmoid values as in the reference (saturated scores tie).  Everything is a device pointer;
 * rgcn_rank_reserve(max_queries) sizes the [max_queries, V] score buffer, longer inputs are chunked.
 * Relation-sharded contexts: the codes are replicated after the forward pass, so each rank ranks its own slice
 * of the queries (no collective) and the caller concatenates. */
rgcn_status rgcn_rank_reserve(rgcn_ctx* ctx, int64_t max_queries);
rgcn_status rgcn_rank_device(rgcn_ctx* ctx, const int32_t* x_dev, int64_t num_queries, int32_t predict_object,
                             const int64_t* filter_ptr_dev, const int32_t* filter_idx_dev, int32_t* raw_rank_dev,
                             int32_t* filtered_rank_dev);

/* GradientClipping(max_norm) + Adam(lr) of the Converge chain (optimization/tensorflow_backend/
 * algorithms.py:27-42,58-68).  max_grad_norm = 0 disables clipping. */
rgcn_status rgcn_optimizer_config(rgcn_ctx* ctx, float learning_rate, float beta1, float beta2, float epsilon,
                                  float max_grad_norm);
rgcn_status rgcn_optimizer_step(rgcn_ctx* ctx);
/* The same step in two phases around its one exchange point on a relation-sharded context: the
 * squared norm of the relation-sharded gradients (block W_forward / W_backward, basis C_forward / C_backward:
 * owner-only, zero elsewhere) is a sum over ranks.  _norm_partial leaves this rank's share in
 * RGCN_BUF_NORM_EXCHANGE; the caller sum-all-reduces it (rgcn_optimizer_step does, with RCCL); _apply clips by
 * the global norm and runs Adam.  Replicated tensors receive identical updates on every rank; a relation's
 * weights are current on its owner only, so the owner map must stay fixed over a sharded training run.  The
 * stand-alone decoder pass (rgcn_decoder_loss_backward_device) is replicated: same batch, same result on every rank;
 * inside rgcn_train_step_device / rgcn_train_step_minibatch_device on a context with a communicator the decoder is
 * DIVIDED by triples: rank g scores the slice [g ceil(N / world), ...) of the batch, and the partial loss, dL/dcodes
 * and dL/dW_relation are summed over the ranks (three all-reduces: [V,d], [R,d], one float). */
rgcn_status rgcn_optimizer_norm_partial(rgcn_ctx* ctx);
rgcn_status rgcn_optimizer_apply(rgcn_ctx* ctx);
/* One TensorflowOptimizer.update_from_batch (optimize.py:81-88) entirely on the device, asynchronous:
 * graph prep (or adoption of a prefetched one), encoder forward (train), decoder loss + gradients,
 * encoder backward, and -- if rgcn_optimizer_config was called -- clip + Adam.  On a sharded context (after
 * rgcn_comm_init) the same sequence with its exchanges on RCCL; every rank passes the same graph and batch. */
rgcn_status rgcn_train_step_device(rgcn_ctx* ctx, const int32_t* triples_dev, int64_t num_edges,
                                   const int32_t* x_dev, const float* y_dev, int64_t num_triples,
                                   uint64_t dropout_seed, float regularization_parameter);

/* The reference's t_func after the neighbourhood sampling + update_from_batch (code/train.py:227-245,
 * optimize.py:81-88) in ONE asynchronous call on a graph batch that is resident in HBM: edge dropout (exact `keep`
 * of the n batch edges -> message graph, as rgcn_set_graph_dropout_device with edge_seed), negative sampling
 * (rgcn_negative_sample_device on ALL n batch edges -> x_scratch_dev int32 [n*(rate+1),3], y_scratch_dev float
 * [n*(rate+1)]), then the train step of rgcn_train_step_device.  Only the n batch triples cross PCIe per step. */
rgcn_status rgcn_train_step_minibatch_device(rgcn_ctx* ctx, const int32_t* batch_dev, int64_t num_edges, int64_t keep,
                                             uint64_t edge_seed, int32_t negative_rate, uint64_t negative_seed,
                                             int32_t* x_scratch_dev, float* y_scratch_dev, uint64_t dropout_seed,
                                             float regularization_parameter);

/* Software pipelining across steps: prepare the graph structures of the NEXT minibatch (same work as
 * rgcn_set_graph_device) on a side stream into a second buffer set while the step already queued
 * keeps running.  A later rgcn_step_device with the same (pointer, num_edges) adopts them instead of
 * rebuilding.  Call it AFTER queueing the current step.  The reference has no counterpart: its graph
 * arrives through feed_dict at session.run (optimize.py:81-88).
 * Contract: the triple buffer must not change between this call and the step that adopts the preparation.
 * Writing it through rgcn_copy_to_device* or freeing it through rgcn_device_free drops the preparation (the
 * step then rebuilds in line); writes the library cannot see (the caller's own kernels / copies) are the
 * caller's responsibility -- call rgcn_prefetch_graph_device again after them. */
rgcn_status rgcn_prefetch_graph_device(rgcn_ctx* ctx, const int32_t* triples_dev_next, int64_t num_edges);
/* the same for a step that draws its graph by edge dropout: adopted by rgcn_train_step_minibatch_device called with
 * the same (batch pointer, num_edges, keep, edge seed) */
rgcn_status rgcn_prefetch_graph_dropout_device(rgcn_ctx* ctx, const int32_t* batch_dev_next, int64_t num_edges,
                                               int64_t keep, uint64_t edge_seed);

/* ---- hipGraph capture of whole steps (BASELINE config 5: "hipGraph-captured train step") -------------
 * Between rgcn_capture_begin and rgcn_capture_end every asynchronous device call on the context
 * (rgcn_step_device, rgcn_train_step_device, rgcn_prefetch_graph_device, rgcn_set_graph_device + rgcn_forward +
 * rgcn_backward_device ...) is recorded into a hipGraph instead of being executed, side streams included;
 * rgcn_graph_launch replays it on the context's stream with one launch.  Shapes and device pointers are the
 * captured ones (static shapes: refresh the CONTENTS of the triple / batch buffers between launches); every random
 * draw differs from replay to replay -- a device counter offsets the captured seeds: replay k of a graph draws the
 * dropout masks, the edge-dropout subset and the negative samples of (captured seed + k) -- and Adam's step count
 * advances on the device.  Calls that synchronise or touch host memory return RGCN_ERR_STATE during a capture;
 * run one ordinary step first so that every lazily allocated buffer exists.  To keep the graph preparation of
 * the next minibatch overlapped inside a graph, capture an even number of steps, each followed by the prefetch
 * of the other triple buffer, with the first buffer prefetched (and finished) before the capture begins. */
rgcn_status rgcn_capture_begin(rgcn_ctx* ctx);
rgcn_status rgcn_capture_end(rgcn_ctx* ctx, int32_t* graph_id);
rgcn_status rgcn_graph_launch(rgcn_ctx* ctx, int32_t graph_id);
rgcn_status rgcn_graph_destroy(rgcn_ctx* ctx, int32_t graph_id);

/* ---- relation sharding across GPUs (new: the reference is single-device) ------------
 * owner[r] in [0, world) assigns relation r's edges and W_f[r]/W_b[r] (BLOCK) or C_f[r]/C_b[r]
 * (BASIS) to one rank.  Degrees stay global.  Must be identical on all ranks. */
rgcn_status rgcn_set_relation_owner(rgcn_ctx* ctx, const int32_t* owner, int32_t count);
/* RCCL bootstrap: rank 0 makes the id, the launcher broadcasts the 128 bytes, every rank inits.
 * librccl.so.1 is dlopen'ed on first use; a world==1 context never touches it. */
rgcn_status rgcn_comm_unique_id(uint8_t id[128]);
rgcn_status rgcn_comm_init(rgcn_ctx* ctx, const uint8_t id[128]);
/* sum-all-reduce of `count` floats at a device pointer on the context's stream (used by bench.py
 * for its barrier / max-over-ranks reduction so that no second RCCL client is needed). */
rgcn_status rgcn_comm_allreduce_sum(rgcn_ctx* ctx, float* dev, int64_t count);
/* What the collective library itself reports about the context's communicator (ncclCommCount / ncclCommUserRank /
 * ncclCommCuDevice): how many ranks it SEES, this rank's index, the device it is bound to -- the answer to "did RCCL
 * come up with N ranks" from inside a run (bench.py prints it as "rccl_ranks").  -1 where there is no communicator
 * (world == 1 / before rgcn_comm_init) or the bound library lacks the entry point. */
rgcn_status rgcn_comm_info(rgcn_ctx* ctx, int32_t* comm_ranks, int32_t* comm_rank, int32_t* comm_device);
/* Context-free: how many HIP devices this process sees (0 without a GPU) and, for 0 <= device < count, the PCI address
 * of that device as domain << 16 | bus << 8 | device (-1 otherwise).  A launcher that masks the visible devices per
 * rank (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES) makes every rank's GPU "device 0": bench.py --gpus N picks its
 * device by this count and compares PCI addresses, not indices, when it checks that no two ranks share a GPU. */
rgcn_status rgcn_device_info(int32_t device, int32_t* count, int64_t* pci_address);

/* Phase API: the same forward/backward cut at the points where a sharded run exchanges data.
 * rgcn_forward == begin; for l: partial(l) [all-reduce EXCHANGE] finish(l).
 * rgcn_backward == begin; for l=L..1: partial(l) [all-reduce EXCHANGE, DSELF_EXCHANGE] finish(l); end.
 * With world > 1 and no communicator the caller performs the exchange itself through
 * rgcn_read_buffer / rgcn_write_buffer (single-GPU test double for the collective). */
rgcn_status rgcn_forward_begin(rgcn_ctx* ctx, int32_t train, uint64_t dropout_seed,
                               const uint8_t* dropout_masks_host);
rgcn_status rgcn_forward_layer_partial(rgcn_ctx* ctx, int32_t layer);
rgcn_status rgcn_forward_layer_finish(rgcn_ctx* ctx, int32_t layer);
rgcn_status rgcn_backward_begin(rgcn_ctx* ctx, const float* dcodes_dev);
rgcn_status rgcn_backward_layer_partial(rgcn_ctx* ctx, int32_t layer);
rgcn_status rgcn_backward_layer_finish(rgcn_ctx* ctx, int32_t layer);
rgcn_status rgcn_backward_end(rgcn_ctx* ctx);
rgcn_status rgcn_read_buffer(rgcn_ctx* ctx, int32_t which, void* host, int64_t bytes);
rgcn_status rgcn_write_buffer(rgcn_ctx* ctx, int32_t which, const void* host, int64_t bytes);

/* ---- device memory + timing helpers (so callers need no other GPU runtime) -------------------- */
rgcn_status rgcn_device_alloc(rgcn_ctx* ctx, int64_t bytes, void** dev);
rgcn_status rgcn_device_free(rgcn_ctx* ctx, void* dev);
rgcn_status rgcn_copy_to_device(rgcn_ctx* ctx, void* dev, const void* host, int64_t bytes);
/* As rgcn_copy_to_device, but the host does not wait for the transfer: the data is copied to a pinned staging
 * slot during the call (the caller's memory is free again on return) and the transfer is ordered on the
 * context's main stream, or -- on_prefetch_stream != 0 -- on the stream rgcn_prefetch_graph_device works on, so
 * that "upload the next minibatch's triples, then prepare its graph" runs beside the current step (the training
 * driver's double-buffered feed; the reference re-feeds numpy arrays through feed_dict on every session.run,
 * optimize.py:81-88).  Transfers above 1 MB fall back to waiting. */
rgcn_status rgcn_copy_to_device_async(rgcn_ctx* ctx, void* dev, const void* host, int64_t bytes,
                                      int32_t on_prefetch_stream);
rgcn_status rgcn_copy_to_host(rgcn_ctx* ctx, void* host, const void* dev, int64_t bytes);
/* HIP-event stopwatch on the context's stream (torch.cuda.Event cannot see this stream). */
rgcn_status rgcn_timer_start(rgcn_ctx* ctx);
rgcn_status rgcn_timer_stop(rgcn_ctx* ctx, float* elapsed_ms); /* synchronises */

/* Side-stream overlap on/off (default on).  With overlap off every kernel runs alone on the main stream: per-kernel
 * durations are exclusive.  (The product library reads NO tuning variable from the environment: its configuration is
 * what these setters say; experiment knobs exist in librgcn_devtools.so only, include/rgcn_devtools.h.) */
rgcn_status rgcn_set_overlap(rgcn_ctx* ctx, int32_t on);

/* Form of the block-diagonal layer (ConcatGcn.compute_messages + combine_messages, gcn_basis_concat.py:35-83, and
 * their gradient).  Both give bitwise the same activations and gradients -- with one exception, the embedding bias
 * gradient db_emb, which form 1 sums from the per-workgroup column partials of its row-gradient kernel (within 2e-6 of
 * scale of form 0's separate column-sum pass, bit-identical from engine to engine) --
 * tests/test_gpu_parity.py::test_single_pass_layer_equals_the_two_kernel_form.
 *   1 : (default) destination-major banded single pass (csrc/block_rows.hip): ONE kernel per layer and direction walks
 *       the incidence CSR -- a 16-lane group per (row, column band), one band per XCD, the relation's sd x sd blocks
 *       read through L2 from a band-tiled copy of the weights, rows taken by descending length, long rows by whole
 *       workgroups through LDS tiles -- and applies self-loop term, dropout and relu / relu'; no message buffer.
 *       39 / 46 us per layer forward / backward against 57 / 74 for form 0 at FB15k-237 minibatch size, half the time
 *       at the 272,115-edge training graph (profiles/r04_rowmajor_spmm_ab.md).  On relation-sharded contexts
 *       (world > 1) it walks the rank's own messages and writes the partial pre-activations for the exchange.
 *   0 : relation-major message kernel -> [2E,d] message buffer -> row-major reduce (k_combine): the reference form the
 *       other is held equal to, and what runs when the block count exceeds form 1's lane groups.
 * (Rounds 2-3 built two more forms -- the reduce as the self-loop GEMM's epilogue, per-block workgroups with an LDS
 * weight table --, measured them slower and removed them in round 5: profiles/r02_fused_layer_ab.log,
 * profiles/r03_block_spmm_ab.md.)  The basis kind runs its own kernels whatever the setting. */
rgcn_status rgcn_set_fusion(rgcn_ctx* ctx, int32_t mode);

/* Arithmetic of the dense contractions (self-loop and basis GEMMs); all of them take and return fp32.
 *   6 : (default) every fp32 operand is split exactly into three bf16 numbers hi + mid + lo (round to
 *       nearest + exact residual, twice) and the product is accumulated in fp32 from 6 of the 9 partial
 *       products on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16); the three dropped ones (mid*lo,
 *       lo*mid, lo*lo) are below 2^-26 of |a||b| per product, a quarter of one fp32 rounding.  Measured
 *       error against float64 equals the fp32-MFMA path's (tests/test_gpu_parity.py::test_gemm_modes).
 *   9 : all 9 partial products - the products of exact fp32 arithmetic in another summation order
 *   0 : fp32 MFMA (v_mfma_f32_32x32x2_f32), 1/16 of the bf16 rate on gfx950
 *   3 : hi*hi, hi*mid, mid*hi only (about 2^-17; NOT fp32 - experiments only)
 * RGCN_GEMM_MODE in the environment overrides the default at create. */
rgcn_status rgcn_set_gemm_mode(rgcn_ctx* ctx, int32_t mode);

/* Per-kernel profile: when enabled every launch is bracketed by HIP events on the context's stream.
 * Records aggregate by kernel name, summed over calls.  alg_bytes = the DESIGN bytes of the launches (what the
 * kernel asks of the memory system by construction: a gathered row counts once per use, staging slabs count);
 * alg_flops = the algorithmic flops; rgcn_profile_get_compulsory = the COMPULSORY bytes (every distinct input byte
 * once + every output byte once) -- the figure roofline fractions are computed on.
 * The reference has no counterpart (TF's timeline is not used by code/train.py). */
rgcn_status rgcn_profile_enable(rgcn_ctx* ctx, int32_t on);
rgcn_status rgcn_profile_reset(rgcn_ctx* ctx);
int32_t     rgcn_profile_count(rgcn_ctx* ctx); /* synchronises, aggregates; number of kernel names */
rgcn_status rgcn_profile_get(rgcn_ctx* ctx, int32_t i, char* name, int32_t name_cap, int64_t* calls,
                             double* total_ms, double* alg_bytes, double* alg_flops);
rgcn_status rgcn_profile_get_compulsory(rgcn_ctx* ctx, int32_t i, double* compulsory_bytes);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* RGCN_H_ */
