"""CPU oracle: op-for-op numpy/scipy restatement of the reference R-GCN encoder.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED at the TF-kernel
boundary (the reference has no golden vectors and TF 1.4 cannot run here); pinned above
it against the reference's own model code run over a numpy stand-in for the TF primitives
(tests/golden/make_reference_model_fixtures.py, tests/test_reference_model.py).

Every function cites the reference file:line it restates (paths relative to
the reference checkout, ``code/...``).  The dataflow is deliberately the
reference's ("TF-shaped"): materialised ``H[s]``/``H[o]``/``W[type]`` gathers,
batched small matmuls, ``[V,E]`` sparse incidence matrices multiplied into
``[E,d]`` message matrices.  That makes it (a) an honest restatement to check
the HIP path against, and (b) the CPU baseline timed by bench.py
(``cpu_baseline.kind == "port"``).

All floating point is float32 (the reference builds float32 variables,
code/common/shared_functions.py:17,26); indices are int32
(code/extras/graph_representations.py:174).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

__all__ = [
    "KIND_BLOCK", "KIND_BASIS", "NORM_INTENDED", "NORM_TF_AS_EXECUTED", "NORM_NONE", "distmult_ranks", "ranks_from_energies", "sample_edge_neighborhood",
    "glorot_variance", "init_params", "weight_names", "split_graph",
    "incidence_values", "incidence_matrix", "affine_onehot_forward",
    "concat_messages", "basis_messages", "self_loop", "dropout", "combine_messages",
    "encoder_forward", "encoder_backward", "encoder_step",
    "dense_closed_form_forward", "distmult_loss_and_grads", "sample_minibatch_graph",
    "synthetic_graph",
]

KIND_BLOCK = "block"   # ConcatGcn, code/encoders/message_gcns/gcn_basis_concat.py
KIND_BASIS = "basis"   # BasisGcn,  code/encoders/message_gcns/gcn_basis.py

NORM_INTENDED = "intended"              # 1/deg(row of THIS edge)
NORM_TF_AS_EXECUTED = "tf_as_executed"  # values come back in sorted-row order
NORM_NONE = "none"

F32 = np.float32


# --------------------------------------------------------------------------- init

def glorot_variance(shape):
    """code/common/shared_functions.py:12-13 (used as the *scale* of np.random.normal, :17)."""
    return 3.0 / np.sqrt(shape[0] + shape[1])


def weight_names(kind, num_layers):
    """Order of ``Model.get_weights()`` (innermost component first): code/model.py:96-97,169-182;
    per-layer lists: gcn_basis_concat.py:30-33, gcn_basis.py:33-37; affine_transform.py:30-31;
    relation_embedding.py:20-21."""
    names = ["W_emb", "b_emb"]
    for l in range(1, num_layers + 1):
        names += [f"W_f{l}", f"W_b{l}"]
        if kind == KIND_BASIS:
            names += [f"C_f{l}", f"C_b{l}"]
        names += [f"W_self{l}", f"b{l}"]
    names.append("W_relation")
    return names


def init_params(V, R, d, num_layers, kind, nb, rng=None):
    """Initial weights with the reference's distributions and RNG consumption order.

    Variable creation order = outermost component first (code/model.py:156-164):
    RelationEmbedding (relation_embedding.py:15-18, randn [EntityCount, d]: SURVEY H3) ->
    top GCN layer ... bottom GCN layer (gcn_basis_concat.py:17-27 / gcn_basis.py:15-30) ->
    AffineTransform (affine_transform.py:24-28).
    ``rng`` needs ``normal(mean, scale, size=)`` and ``randn(*shape)`` (np.random or a RandomState).
    """
    rng = np.random if rng is None else rng
    p = {}
    p["W_relation"] = rng.randn(V, d).astype(F32)
    for l in range(num_layers, 0, -1):
        if kind == KIND_BLOCK:
            if d % nb != 0:
                raise ValueError("InternalEncoderDimension must be divisible by NumberOfBasisFunctions "
                                 "(gcn_basis_concat.py:15 truncates; the reshape at :42 then mis-groups)")
            sd = d // nb
            shape = (R, nb, sd, sd)                       # gcn_basis_concat.py:18-19
            var = glorot_variance([shape[0], shape[2]])   # :22  (-> 3/sqrt(R+sd))
            p[f"W_f{l}"] = rng.normal(0, var, size=shape).astype(F32)
            p[f"W_b{l}"] = rng.normal(0, var, size=shape).astype(F32)
            p[f"W_self{l}"] = rng.normal(0, var, size=(d, d)).astype(F32)
        else:
            shape = (d, nb, d)                            # gcn_basis.py:18  (in, basis, out)
            var = glorot_variance([shape[0], shape[2]])   # :21  (-> 3/sqrt(2d))
            p[f"W_f{l}"] = rng.normal(0, var, size=shape).astype(F32)
            p[f"W_b{l}"] = rng.normal(0, var, size=shape).astype(F32)
            p[f"W_self{l}"] = rng.normal(0, var, size=(d, d)).astype(F32)
            p[f"C_f{l}"] = rng.normal(0, 1, size=(R, nb)).astype(F32)   # :26-28
            p[f"C_b{l}"] = rng.normal(0, 1, size=(R, nb)).astype(F32)
        p[f"b{l}"] = np.zeros(d, dtype=F32)               # created, never used: SURVEY H2
    p["W_emb"] = rng.normal(0, glorot_variance([V, d]), size=(V, d)).astype(F32)  # affine_transform.py:25-27
    p["b_emb"] = np.zeros(d, dtype=F32)
    return p


# --------------------------------------------------------------------------- graph

def split_graph(triples):
    """MessageGraph.process, code/extras/graph_representations.py:21-27:
    column 0 = sender (subject), column 1 = type (relation), column 2 = receiver (object)."""
    t = np.asarray(triples)
    if t.ndim != 2 or t.shape[1] != 3:
        raise ValueError("graph_edges must be [E,3]")
    t = t.astype(np.int32)
    return t[:, 0].copy(), t[:, 1].copy(), t[:, 2].copy()


def incidence_values(row_index, V, mode=NORM_INTENDED):
    """Values of the [V,E] incidence matrix with 'global' normalisation:
    tf.sparse_softmax over rows of ones (graph_representations.py:82-93, 122-133) = 1/deg(row),
    with deg counted inside the fed graph.

    ``tf_as_executed``: the SparseSoftmax kernel reorders a deep copy into canonical
    row-major order, writes group results sequentially, and the Python wrapper re-attaches the
    ORIGINAL (edge-order) indices -> edge k gets 1/deg(sort(row_index)[k]).
    """
    row_index = np.asarray(row_index)
    E = row_index.shape[0]
    if mode == NORM_NONE:
        return np.ones(E, dtype=F32)
    deg = np.bincount(row_index, minlength=V).astype(np.int64)
    if mode == NORM_INTENDED:
        rows = row_index
    elif mode == NORM_TF_AS_EXECUTED:
        rows = np.sort(row_index, kind="stable")
    else:
        raise ValueError(mode)
    return (F32(1.0) / deg[rows].astype(F32)).astype(F32)


def incidence_matrix(row_index, values, V):
    """[V,E] sparse matrix with entry (row_index[e], e) = values[e]
    (graph_representations.py:86-91 forward: rows = receivers; :126-131 backward: rows = senders)."""
    E = row_index.shape[0]
    return sp.csr_matrix((values.astype(F32), (row_index, np.arange(E))), shape=(V, E), dtype=F32)


# --------------------------------------------------------------------------- forward pieces

def affine_onehot_forward(W_emb, b_emb):
    """AffineTransform.get_all_codes, onehot branch + bias + relu (affine_transform.py:63-83;
    constructed with onehot_input=True, use_bias=True, use_nonlinearity=True: model_builder.py:141-146)."""
    return np.maximum(W_emb + b_emb, F32(0)).astype(F32)


def concat_messages(H, s, r, o, W_f, W_b):
    """ConcatGcn.compute_messages, gcn_basis_concat.py:35-52 (TF-shaped: gathers materialised)."""
    E = s.shape[0]
    R, nb, sd, _ = W_f.shape
    sender = H[s]                                   # message_gcn.py:39-40 (embedding_lookup)
    receiver = H[o]
    fwd_T = W_f[r]                                  # :38  [E,nb,sd,sd]
    bwd_T = W_b[r]                                  # :39
    xs = sender.reshape(E, nb, sd)                  # :42
    xr = receiver.reshape(E, nb, sd)                # :43
    # :46-47  matmul(T, expand_dims(x,-1)) squeezed:  out[e,b,i] = sum_j T[e,b,i,j] x[e,b,j]
    F = np.einsum("ebij,ebj->ebi", fwd_T, xs, optimize=False).astype(F32)
    K = np.einsum("ebij,ebj->ebi", bwd_T, xr, optimize=False).astype(F32)
    return F.reshape(E, nb * sd), K.reshape(E, nb * sd)   # :50-51


def basis_messages(H, s, r, o, W_f, W_b, C_f, C_b):
    """BasisGcn.compute_messages / compute_coefficients / compute_basis_functions / dot_or_tensor_mul,
    gcn_basis.py:39-68."""
    E = s.shape[0]
    d_in, B, d_out = W_f.shape
    sender = H[s]
    receiver = H[o]
    f_scale = C_f[r]                                # :43  [E,B]
    b_scale = C_b[r]                                # :44
    s_terms = (sender @ W_f.reshape(d_in, B * d_out)).reshape(E, B, d_out)     # :54-55,60-68
    r_terms = (receiver @ W_b.reshape(d_in, B * d_out)).reshape(E, B, d_out)   # :56
    F = np.sum(s_terms * f_scale[:, :, None], axis=1).astype(F32)              # :43
    K = np.sum(r_terms * b_scale[:, :, None], axis=1).astype(F32)              # :44
    return F, K


def self_loop(H, W_self):
    """compute_self_loop_messages -> dot_or_lookup(matmul branch): gcn_basis_concat.py:65-66,
    gcn_basis.py:70-71, shared_functions.py:5-9."""
    return (H @ W_self).astype(F32)


def dropout(x, keep_prob, mask):
    """tf.nn.dropout(x, keep): x / keep * floor(keep + U) (message_gcn.py:64).  The Bernoulli draw
    is an explicit 0/1 ``mask`` here (TF's Philox stream is not reproducible)."""
    return (x * (mask.astype(F32) / F32(keep_prob))).astype(F32)


def combine_messages(F, K, S, s, o, V, use_nonlinearity, norm_mode=NORM_INTENDED):
    """combine_messages, gcn_basis_concat.py:69-83 / gcn_basis.py:74-88 (bias b NOT added: SURVEY H2)."""
    mtr_f = incidence_matrix(o, incidence_values(o, V, norm_mode), V)   # receivers, :70
    mtr_b = incidence_matrix(s, incidence_values(s, V, norm_mode), V)   # senders,   :71
    collected = (mtr_f @ F + mtr_b @ K).astype(F32)                      # :73-76
    pre = (collected + S).astype(F32)
    return np.maximum(pre, F32(0)) if use_nonlinearity else pre          # :78-81


# --------------------------------------------------------------------------- encoder forward

def _layer_weights(params, l, kind):
    if kind == KIND_BLOCK:
        return params[f"W_f{l}"], params[f"W_b{l}"], None, None, params[f"W_self{l}"]
    return (params[f"W_f{l}"], params[f"W_b{l}"], params[f"C_f{l}"], params[f"C_b{l}"],
            params[f"W_self{l}"])


def encoder_forward(params, triples, V, num_layers, kind, mode="train", keep_prob=0.8,
                    dropout_masks=None, norm_mode=NORM_INTENDED):
    """Whole encoder: AffineTransform -> L x (ConcatGcn | BasisGcn); relu on all but the last layer
    (model_builder.py:273-309, :275).  Returns [H0, H1, ..., HL]; HL is both the subject and the
    object code matrix (relation_embedding.py:23-25, message_gcn.py:44-47).

    ``mode == 'train'`` applies dropout to the self-loop term only (message_gcn.py:60-64) using
    ``dropout_masks[l-1]`` ([V,d] 0/1)."""
    s, r, o = split_graph(triples)
    H = affine_onehot_forward(params["W_emb"], params["b_emb"])
    acts = [H]
    for l in range(1, num_layers + 1):
        W_f, W_b, C_f, C_b, W_self = _layer_weights(params, l, kind)
        if kind == KIND_BLOCK:
            F, K = concat_messages(H, s, r, o, W_f, W_b)
        else:
            F, K = basis_messages(H, s, r, o, W_f, W_b, C_f, C_b)
        S = self_loop(H, W_self)                                          # message_gcn.py:58
        if mode == "train":
            if dropout_masks is None:
                raise ValueError("train mode needs explicit dropout masks")
            S = dropout(S, keep_prob, dropout_masks[l - 1])               # message_gcn.py:64
        H = combine_messages(F, K, S, s, o, V, use_nonlinearity=(l < num_layers), norm_mode=norm_mode)
        acts.append(H)
    return acts


# --------------------------------------------------------------------------- encoder backward

def _segment_sum_rows(index, n_rows, M):
    """sum rows of M [E,k] into n_rows buckets (what tf.gradients does for embedding_lookup:
    IndexedSlices -> unsorted_segment_sum)."""
    E = index.shape[0]
    P = sp.csr_matrix((np.ones(E, dtype=F32), (index, np.arange(E))), shape=(n_rows, E), dtype=F32)
    return np.asarray(P @ M, dtype=F32)


def encoder_backward(params, triples, V, num_layers, kind, acts, dcodes, mode="train", keep_prob=0.8,
                     dropout_masks=None, norm_mode=NORM_INTENDED):
    """Explicit reverse-mode of ``encoder_forward`` (= tf.gradients(loss, weights),
    code/optimization/abstract.py:117-118; formulas: SURVEY.md section 8a row a15).

    ``dcodes`` = dL/dH_L [V,d].  Returns a dict name -> gradient (b{l} gets zeros: it is unused in
    the forward, so TF returns None for it; W_relation is not an encoder-path weight)."""
    s, r, o = split_graph(triples)
    E = s.shape[0]
    n_f = incidence_values(o, V, norm_mode)
    n_b = incidence_values(s, V, norm_mode)
    grads = {}
    dH = np.asarray(dcodes, dtype=F32)
    for l in range(num_layers, 0, -1):
        W_f, W_b, C_f, C_b, W_self = _layer_weights(params, l, kind)
        Hin, Hout = acts[l - 1], acts[l]
        D = dH * (Hout > 0) if l < num_layers else dH            # relu' (relu only below the top layer)
        D = D.astype(F32)
        # self-loop branch
        dS = dropout(D, keep_prob, dropout_masks[l - 1]) if mode == "train" else D
        grads[f"W_self{l}"] = (Hin.T @ dS).astype(F32)
        dHin = (dS @ W_self.T).astype(F32)
        # incidence-matrix transposes:  dF[e] = n_f[e] D[o_e],  dK[e] = n_b[e] D[s_e]
        dF = (D[o] * n_f[:, None]).astype(F32)
        dK = (D[s] * n_b[:, None]).astype(F32)
        xs, xr = Hin[s], Hin[o]
        if kind == KIND_BLOCK:
            R, nb, sd, _ = W_f.shape
            dFb, dKb = dF.reshape(E, nb, sd), dK.reshape(E, nb, sd)
            xsb, xrb = xs.reshape(E, nb, sd), xr.reshape(E, nb, sd)
            # d/dT[e,b,i,j] = dF[e,b,i] x[e,b,j]; summed per relation (IndexedSlices of W[type])
            gT_f = np.einsum("ebi,ebj->ebij", dFb, xsb).reshape(E, nb * sd * sd)
            gT_b = np.einsum("ebi,ebj->ebij", dKb, xrb).reshape(E, nb * sd * sd)
            grads[f"W_f{l}"] = _segment_sum_rows(r, R, gT_f).reshape(W_f.shape)
            grads[f"W_b{l}"] = _segment_sum_rows(r, R, gT_b).reshape(W_b.shape)
            dxs = np.einsum("ebij,ebi->ebj", W_f[r], dFb).reshape(E, nb * sd).astype(F32)
            dxr = np.einsum("ebij,ebi->ebj", W_b[r], dKb).reshape(E, nb * sd).astype(F32)
        else:
            d_in, B, d_out = W_f.shape
            R = C_f.shape[0]
            Wf2, Wb2 = W_f.reshape(d_in, B * d_out), W_b.reshape(d_in, B * d_out)
            s_terms = (xs @ Wf2).reshape(E, B, d_out)
            r_terms = (xr @ Wb2).reshape(E, B, d_out)
            # F = sum_b scale[e,b] terms[e,b,:]
            g_fscale = np.einsum("ebk,ek->eb", s_terms, dF).astype(F32)
            g_bscale = np.einsum("ebk,ek->eb", r_terms, dK).astype(F32)
            grads[f"C_f{l}"] = _segment_sum_rows(r, R, g_fscale)
            grads[f"C_b{l}"] = _segment_sum_rows(r, R, g_bscale)
            g_sterms = (C_f[r][:, :, None] * dF[:, None, :]).reshape(E, B * d_out).astype(F32)
            g_rterms = (C_b[r][:, :, None] * dK[:, None, :]).reshape(E, B * d_out).astype(F32)
            grads[f"W_f{l}"] = (xs.T @ g_sterms).reshape(W_f.shape).astype(F32)
            grads[f"W_b{l}"] = (xr.T @ g_rterms).reshape(W_b.shape).astype(F32)
            dxs = (g_sterms @ Wf2.T).astype(F32)
            dxr = (g_rterms @ Wb2.T).astype(F32)
        dHin = dHin + _segment_sum_rows(s, V, dxs) + _segment_sum_rows(o, V, dxr)
        grads[f"b{l}"] = np.zeros(Hin.shape[1], dtype=F32)
        dH = dHin.astype(F32)
    # AffineTransform: H0 = relu(W + b)
    g0 = (dH * (acts[0] > 0)).astype(F32)
    grads["W_emb"] = g0
    grads["b_emb"] = g0.sum(axis=0, dtype=F32).astype(F32)
    return grads


def encoder_step(params, triples, V, num_layers, kind, dcodes, keep_prob=0.8, dropout_masks=None,
                 norm_mode=NORM_INTENDED, mode="train"):
    """One 'step' of the BASELINE metric: encoder forward + backward.  Returns (acts, grads)."""
    acts = encoder_forward(params, triples, V, num_layers, kind, mode=mode, keep_prob=keep_prob,
                           dropout_masks=dropout_masks, norm_mode=norm_mode)
    grads = encoder_backward(params, triples, V, num_layers, kind, acts, dcodes, mode=mode,
                             keep_prob=keep_prob, dropout_masks=dropout_masks, norm_mode=norm_mode)
    return acts, grads


# --------------------------------------------------------------------------- independent restatement

def dense_closed_form_forward(params, triples, V, num_layers, kind, mode="train", keep_prob=0.8,
                              dropout_masks=None):
    """Second, structurally different restatement (small graphs only), float64, NORM_INTENDED:
    H' = act( sum_r  A^f_r H Wf_r^T + A^b_r H Wb_r^T  + dropout(H W_self) ) with dense per-relation
    adjacency matrices and dense per-relation [d,d] weights."""
    s, r, o = split_graph(triples)
    d = params["W_emb"].shape[1]
    indeg = np.bincount(o, minlength=V).astype(np.float64)
    outdeg = np.bincount(s, minlength=V).astype(np.float64)
    H = np.maximum(params["W_emb"].astype(np.float64) + params["b_emb"].astype(np.float64), 0.0)
    acts = [H]
    R = (params["W_f1"].shape[0] if kind == KIND_BLOCK else params["C_f1"].shape[0])
    for l in range(1, num_layers + 1):
        W_f, W_b, C_f, C_b, W_self = _layer_weights(params, l, kind)
        pre = np.zeros((V, d))
        for rel in range(R):
            sel = np.nonzero(r == rel)[0]
            if sel.size == 0:
                continue
            if kind == KIND_BLOCK:
                nb, sd = W_f.shape[1], W_f.shape[2]
                Mf = np.zeros((d, d)); Mb = np.zeros((d, d))
                for b in range(nb):   # message = T x  ->  row-vector form x @ T^T
                    Mf[b * sd:(b + 1) * sd, b * sd:(b + 1) * sd] = W_f[rel, b].T
                    Mb[b * sd:(b + 1) * sd, b * sd:(b + 1) * sd] = W_b[rel, b].T
            else:
                Mf = np.einsum("b,jbk->jk", C_f[rel].astype(np.float64), W_f.astype(np.float64))
                Mb = np.einsum("b,jbk->jk", C_b[rel].astype(np.float64), W_b.astype(np.float64))
            Af = np.zeros((V, V)); Ab = np.zeros((V, V))
            for e in sel:
                Af[o[e], s[e]] += 1.0 / indeg[o[e]]
                Ab[s[e], o[e]] += 1.0 / outdeg[s[e]]
            pre += Af @ H @ Mf + Ab @ H @ Mb
        S = H @ W_self.astype(np.float64)
        if mode == "train":
            S = S * dropout_masks[l - 1].astype(np.float64) / float(keep_prob)
        pre += S
        H = np.maximum(pre, 0.0) if l < num_layers else pre
        acts.append(H)
    return acts


# --------------------------------------------------------------------------- decoder (boundary consumer)

def distmult_loss_and_grads(codes, W_relation, X, Y, reg_param=0.01):
    """BilinearDiag: code/decoders/bilinear_diag.py:14-34 (loss; pos_weight forced to 1 at :32-33)
    and :63-69 (regulariser).  Returns (loss, dL/dcodes, dL/dW_relation).  Used to form a real
    scalar loss for gradient checks; the decoder itself is a 'next' row."""
    X = np.asarray(X)
    e1, rr, e2 = codes[X[:, 0]], W_relation[X[:, 1]], codes[X[:, 2]]
    x = np.sum(e1 * rr * e2, axis=1).astype(F32)
    z = np.asarray(Y, dtype=F32)
    # tf.nn.weighted_cross_entropy_with_logits(targets=z, logits=x, pos_weight=1)
    per = (1 - z) * x + (np.log1p(np.exp(-np.abs(x))) + np.maximum(-x, 0))
    N = x.shape[0]
    loss = per.mean(dtype=np.float64)
    reg = reg_param * (np.mean(e1.astype(np.float64) ** 2) + np.mean(rr.astype(np.float64) ** 2)
                       + np.mean(e2.astype(np.float64) ** 2))
    ex = np.exp(-np.abs(x))
    sig = np.where(x >= 0, 1.0 / (1.0 + ex), ex / (1.0 + ex))
    dx = (sig - z) / N                                         # d mean(xent)/dx = sigmoid(x) - z
    dcol = X.shape[0] * codes.shape[1]
    g_e1 = dx[:, None] * (rr * e2) + reg_param * 2.0 * e1 / dcol
    g_r = dx[:, None] * (e1 * e2) + reg_param * 2.0 * rr / dcol
    g_e2 = dx[:, None] * (e1 * rr) + reg_param * 2.0 * e2 / dcol
    dcodes = _segment_sum_rows(X[:, 0], codes.shape[0], g_e1.astype(F32)) \
        + _segment_sum_rows(X[:, 2], codes.shape[0], g_e2.astype(F32))
    dW_rel = _segment_sum_rows(X[:, 1], W_relation.shape[0], g_r.astype(F32))
    return float(loss + reg), dcodes.astype(F32), dW_rel.astype(F32)


# --------------------------------------------------------------------------- workload builders

def ranks_from_energies(energies, gold, known_idx):
    """MrrScore.append_line (evaluation.py:148-153) on one row of energies: the integer half of the ranking.
    raw = #{score >= score[gold]}, filtered = raw - #{known with score >= score[gold]} + 1, on fp32 sigmoid values
    (bilinear_diag.py:51-61; sigmoid in float64, rounded once to float32: saturated scores tie, as in TF)."""
    with np.errstate(over="ignore"):
        scores = (1.0 / (1.0 + np.exp(-np.asarray(energies, dtype=np.float64)))).astype(np.float32)
    g = scores[gold]
    n_raw = int(np.sum(scores >= g))
    idx = np.asarray(known_idx, dtype=np.int64)
    return n_raw, n_raw - int(np.sum(scores[idx] >= g)) + 1


def distmult_ranks(codes, W_relation, triples, predict_object, known):
    """Raw and filtered ranks as the reference computes them: scores against every entity
    (bilinear_diag.py:51-61: sigmoid of the energies), then ranks_from_energies.
    ``known`` maps (entity, relation) -> list of completing entities (Scorer.known_object_triples for
    predict_object, known_subject_triples otherwise; evaluation.py:232-270)."""
    codes = np.asarray(codes, dtype=F32)
    rel = np.asarray(W_relation, dtype=F32)
    raw, filt = [], []
    for s, r, o in np.asarray(triples):
        if predict_object:
            q, gold, key = codes[s] * rel[r], o, (s, r)
        else:
            q, gold, key = rel[r] * codes[o], s, (o, r)
        a, b = ranks_from_energies((codes @ q).astype(F32), gold, known[key])
        raw.append(a)
        filt.append(b)
    return np.asarray(raw, dtype=np.int32), np.asarray(filt, dtype=np.int32)


def sample_edge_neighborhood(triples, num_entities, sample_size, rng):
    """The reference's neighbourhood edge sampler, step for step (code/train.py:133-139, 161-198): adjacency
    entries [edge, other vertex] per vertex, `sample_counts` = free edge ends, `seen` = touched; a vertex is
    drawn with p ~ sample_counts * seen (uniform over vertices with free ends when that is all zero), then
    adjacency entries of it uniformly until one is not picked yet.  O(V) per pick — small graphs only."""
    triples = np.asarray(triples)
    adj = [[] for _ in range(num_entities)]
    for i, (s, _, o) in enumerate(triples):
        adj[s].append((i, o))
        adj[o].append((i, s))
    sample_counts = np.array([len(a) for a in adj], dtype=np.int64)
    picked = np.zeros(len(triples), dtype=bool)
    seen = np.zeros(num_entities, dtype=bool)
    edges = np.zeros(sample_size, dtype=np.int32)
    for i in range(sample_size):
        weights = sample_counts * seen
        if weights.sum() == 0:
            weights = np.ones_like(weights)
            weights[sample_counts == 0] = 0
        v = rng.choice(num_entities, p=weights / weights.sum())
        seen[v] = True
        e, other = adj[v][rng.randint(len(adj[v]))]
        while picked[e]:
            e, other = adj[v][rng.randint(len(adj[v]))]
        edges[i] = e
        picked[e] = True
        sample_counts[v] -= 1
        sample_counts[other] -= 1
        seen[other] = True
    return edges


def sample_minibatch_graph(triples, graph_batch_size, graph_split_size, rng):
    """Shape of what t_func feeds the encoder (code/train.py:227-238): pick ``graph_batch_size``
    triples, then keep ``int(split * n)`` of them chosen without replacement (exact-k edge dropout,
    SURVEY H8).  The reference's neighbourhood sampler (train.py:161-198) is a host-side Python
    loop outside the measured path; the bench uses a uniform sample."""
    triples = np.asarray(triples)
    n = triples.shape[0]
    ids = rng.choice(n, size=min(graph_batch_size, n), replace=False)
    split = int(graph_split_size * ids.shape[0])
    keep = rng.choice(ids, size=split, replace=False)
    return triples[keep].astype(np.int32)


def synthetic_graph(V, R, E, rng, rel_alpha=1.1, ent_alpha=0.9):
    """Synthetic triples with a Zipf-like relation histogram and skewed endpoint popularity
    (training splits are missing from the reference mount: .MISSING_LARGE_BLOBS).  Unique triples."""
    rel_p = 1.0 / np.arange(1, R + 1) ** rel_alpha
    rel_p /= rel_p.sum()
    ent_p = 1.0 / np.arange(1, V + 1) ** ent_alpha
    ent_p /= ent_p.sum()
    perm_s, perm_o = rng.permutation(V), rng.permutation(V)
    out = np.empty((0, 3), dtype=np.int64)
    while out.shape[0] < E:
        n = int((E - out.shape[0]) * 1.3) + 16
        t = np.stack([perm_s[rng.choice(V, size=n, p=ent_p)],
                      rng.choice(R, size=n, p=rel_p),
                      perm_o[rng.choice(V, size=n, p=ent_p)]], axis=1)
        out = np.unique(np.concatenate([out, t], axis=0), axis=0)
    out = out[rng.permutation(out.shape[0])[:E]]
    return out.astype(np.int32)
