"""Shared test helpers: seeded workloads in the reference's shapes."""
import numpy as np

import oracle


def make_case(V, R, d, L, kind, nb, E, seed=0, train=True, keep=0.8, self_edges=True):
    rng = np.random.RandomState(seed)
    params = oracle.init_params(V, R, d, L, kind, nb, rng=rng)
    # non-zero biases so the test exercises b_emb
    params["b_emb"] = (rng.randn(d) * 0.05).astype(np.float32)
    triples = np.stack([rng.randint(0, V, size=E), rng.randint(0, R, size=E),
                        rng.randint(0, V, size=E)], axis=1).astype(np.int32)
    if not self_edges and E:
        triples[:, 2] = np.where(triples[:, 2] == triples[:, 0], (triples[:, 2] + 1) % V, triples[:, 2])
    masks = [(rng.rand(V, d) < keep).astype(np.uint8) for _ in range(L)] if train else None
    dcodes = (rng.randn(V, d) * 1e-1).astype(np.float32)
    return params, triples, masks, dcodes


# ----------------------------------------------------------------------------- golden cases
import os

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> (graph fixture, V, R, d, L, kind, nb, seed).  Shapes follow BASELINE.json's configs:
# config 1 (Toy, gcn_basis.exp keys, 1 layer), config 2 (FB15k-237 gcn_block), config 3 (gcn_basis B=2).
GOLDEN_CASES = {
    "toy_basis_L1": ("toy_train", 16, 9, 500, 1, "basis", 5, 11),
    "toy_block_L2": ("toy_train", 16, 9, 500, 2, "block", 100, 12),
    "toy_block_small": ("toy_train", 16, 9, 10, 2, "block", 2, 13),
    "fb237_block_L2": ("fb237_minibatch", 14541, 237, 500, 2, "block", 100, 21),
    "fb237_basis_B2_L2": ("fb237_minibatch", 14541, 237, 500, 2, "basis", 2, 22),
    # settings/gcn_basis.exp:5's own basis count (no stored probes: compared with the oracle directly)
    "fb237_basis_B5_L2": ("fb237_minibatch", 14541, 237, 500, 2, "basis", 5, 23),
}


def load_graph(name):
    with np.load(os.path.join(GOLDEN_DIR, "graphs.npz")) as z:
        return z[name].astype(np.int32)


def golden_inputs(name):
    """Deterministic inputs of a golden case (weights are regenerated from the seed, not stored)."""
    graph, V, R, d, L, kind, nb, seed = GOLDEN_CASES[name]
    rng = np.random.RandomState(seed)
    params = oracle.init_params(V, R, d, L, kind, nb, rng=rng)
    params["b_emb"] = (rng.randn(d) * 0.01).astype(np.float32)
    masks = [(rng.rand(V, d) < 0.8).astype(np.uint8) for _ in range(L)]
    dcodes = (rng.randn(V, d) * 1e-2).astype(np.float32)
    return dict(V=V, R=R, d=d, L=L, kind=kind, nb=nb, params=params, masks=masks, dcodes=dcodes,
                triples=load_graph(graph))


def probe(arr, n=256, seed=0):
    """Compact fingerprint of a tensor: l2 norm, sum and n sampled entries (fixed positions)."""
    a = np.asarray(arr, dtype=np.float32).ravel()
    idx = np.random.RandomState(seed).randint(0, a.size, size=min(n, a.size))
    return dict(l2=np.float64(np.sqrt(np.sum(a.astype(np.float64) ** 2))),
                sum=np.float64(a.astype(np.float64).sum()), idx=idx.astype(np.int64), val=a[idx].copy())


# Gradient comparisons tolerate ISOLATED relu-gate flips: a pre-activation within fp32 rounding of 0
# can land on different sides in two correct fp32 implementations (different summation order), which
# moves a handful of gradient entries by O(1e-3 * scale).  The oracle itself shows this against its own
# float64 rerun (W_self1 of fb237_basis_B2_L2: max 1.3e-3 relative, while the l2 error stays ~1e-6).
# So: the bulk must agree to `rel` (l2 / 98 % of sampled entries), no entry may be off by more than
# `spike` x scale.
def check_probe(arr, pr, rel=2e-4, spike=5e-3, name=""):
    a = np.asarray(arr, dtype=np.float32).ravel()
    scale = max(float(np.abs(pr["val"]).max()), 1e-12)
    diff = np.abs(a[pr["idx"]] - pr["val"])
    err = float(diff.max())
    assert err <= spike * scale + 1e-7, "%s: sampled entry off by %.3e (scale %.3e)" % (name, err, scale)
    frac_bad = float((diff > rel * scale + 1e-7).mean())
    assert frac_bad <= 0.02, "%s: %.1f%% of sampled entries differ by more than %.0e x scale" % (
        name, 100 * frac_bad, rel)
    l2 = float(np.sqrt(np.sum(a.astype(np.float64) ** 2)))
    assert abs(l2 - float(pr["l2"])) <= rel * max(float(pr["l2"]), 1e-12) + 1e-7, "%s: l2 norm differs" % name


def assert_close(got, ref, rel=2e-4, abs_tol=1e-7, spike=5e-3, name=""):
    got = np.asarray(got)
    ref = np.asarray(ref)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    if ref.size == 0:
        return
    assert np.isfinite(got).all(), "%s: non-finite values" % name
    g64, r64 = got.astype(np.float64), ref.astype(np.float64)
    scale = float(np.abs(r64).max())
    diff = np.abs(g64 - r64)
    err = float(diff.max())
    assert err <= spike * scale + abs_tol, "%s: max err %.3e vs scale %.3e" % (name, err, scale)
    l2 = float(np.sqrt((diff ** 2).sum()))
    l2ref = float(np.sqrt((r64 ** 2).sum()))
    assert l2 <= rel * l2ref + abs_tol * np.sqrt(ref.size), "%s: l2 err %.3e vs %.3e (rel %.1e)" % (
        name, l2, l2ref, rel)
    frac_bad = float((diff > rel * scale + abs_tol).mean())
    assert frac_bad <= 0.01, "%s: %.2f%% of entries differ by more than %.0e x scale" % (name, 100 * frac_bad, rel)


def gate_consistent_oracle_grads(c, acts, norm="intended", keep=0.8):
    """Oracle backward evaluated AT THE ENGINE'S OWN forward activations.

    A relu pre-activation within fp32 rounding of zero can land on either side in two correct fp32
    implementations; one such gate flip toggles a whole upstream-gradient entry on or off and moves rows
    of dW_emb / dW_self by percents of their scale, although both backward passes are exact for the
    forward they belong to.  When the direct comparison trips over that, the gradient check is repeated
    against the oracle's reverse mode of the forward the engine actually computed (whose activations
    are separately required to match the oracle's within 1e-4).  Returns (grads, n_flips)."""
    norm_mode = {"intended": oracle.NORM_INTENDED, "tf_as_executed": oracle.NORM_TF_AS_EXECUTED,
                 "none": oracle.NORM_NONE}[norm] if isinstance(norm, str) else norm
    oacts = oracle.encoder_forward(c["params"], c["triples"], c["V"], c["L"], c["kind"], mode="train",
                                   keep_prob=keep, dropout_masks=c["masks"], norm_mode=norm_mode)
    flips = 0
    for l in range(0, c["L"]):           # relu layers: the input layer and every GCN layer but the top one
        a, b = np.asarray(acts[l]), np.asarray(oacts[l])
        differ = (a > 0) != (b > 0)
        flips += int(differ.sum())
        if differ.any():                 # a flipped gate must sit on a pre-activation that is ~0 on both sides
            assert float(np.maximum(np.abs(a[differ]), np.abs(b[differ])).max()) <= 1e-4, "gate differs on a non-zero activation"
    grads = oracle.encoder_backward(c["params"], c["triples"], c["V"], c["L"], c["kind"], [np.asarray(a) for a in acts],
                                    c["dcodes"], mode="train", keep_prob=keep, dropout_masks=c["masks"],
                                    norm_mode=norm_mode)
    return grads, flips


# ----------------------------------------------------------------------------- float64 tie-break
import contextlib


@contextlib.contextmanager
def oracle_float64():
    """The oracle's arithmetic type switched to float64 (every cast in oracle/rgcn_oracle.py goes through its
    module-level F32 name), restored on exit."""
    from oracle import rgcn_oracle
    old = rgcn_oracle.F32
    rgcn_oracle.F32 = np.float64
    try:
        yield
    finally:
        rgcn_oracle.F32 = old


def float64_grads_at(c, acts, norm="intended", keep=0.8):
    """float64 reverse mode of the forward pass the ENGINE computed: its own activations (exact fp32 values, hence
    its own relu gates) go in as float64, so the result is the exact derivative the engine's backward approximates and
    no gate can sit on different sides.  What is left between the two is fp32 rounding of the backward pass alone,
    which lets the tolerance drop two orders below the fp32-vs-fp32 comparison's."""
    norm_mode = {"intended": oracle.NORM_INTENDED, "tf_as_executed": oracle.NORM_TF_AS_EXECUTED,
                 "none": oracle.NORM_NONE}[norm] if isinstance(norm, str) else norm
    with oracle_float64():
        p64 = {k: np.asarray(v, dtype=np.float64) for k, v in c["params"].items()}
        return oracle.encoder_backward(p64, c["triples"], c["V"], c["L"], c["kind"],
                                       [np.asarray(a, dtype=np.float64) for a in acts],
                                       np.asarray(c["dcodes"], dtype=np.float64), mode="train", keep_prob=keep,
                                       dropout_masks=c["masks"], norm_mode=norm_mode)


def float64_forward(c, norm="intended", keep=0.8):
    norm_mode = {"intended": oracle.NORM_INTENDED, "tf_as_executed": oracle.NORM_TF_AS_EXECUTED,
                 "none": oracle.NORM_NONE}[norm] if isinstance(norm, str) else norm
    with oracle_float64():
        p64 = {k: np.asarray(v, dtype=np.float64) for k, v in c["params"].items()}
        return oracle.encoder_forward(p64, c["triples"], c["V"], c["L"], c["kind"], mode="train", keep_prob=keep,
                                      dropout_masks=c["masks"], norm_mode=norm_mode)


def error_against(ref64, got):
    """(max |got - ref| / max |ref|, l2 error / l2 of ref) in float64."""
    r = np.asarray(ref64, dtype=np.float64)
    g = np.asarray(got, dtype=np.float64)
    scale = float(np.abs(r).max()) if r.size else 0.0
    if scale == 0.0:
        return (float(np.abs(g).max()) if g.size else 0.0), 0.0
    return float(np.abs(g - r).max()) / scale, float(np.sqrt(((g - r) ** 2).sum() / (r ** 2).sum()))


def chunked_distmult_float64(codes, w_rel, X, Y, reg, chunk=20000):
    """oracle.distmult_loss_and_grads (bilinear_diag.py:14-34,63-69) in float64, chunk by chunk and with sparse products
    for the two scatter-adds, so that N = 330,000 x d = 500 fits: the same formulas, nothing [N, d] held at once."""
    import scipy.sparse as sp
    V, d = codes.shape
    N = len(X)
    c64, w64 = codes.astype(np.float64), w_rel.astype(np.float64)
    xent = sq = 0.0
    dcodes, drel = np.zeros((V, d)), np.zeros(w_rel.shape)
    for lo in range(0, N, chunk):
        x = X[lo:lo + chunk]
        y = Y[lo:lo + chunk].astype(np.float64)
        e1, rr, e2 = c64[x[:, 0]], w64[x[:, 1]], c64[x[:, 2]]
        en = (e1 * rr * e2).sum(1)
        xent += float(((1 - y) * en + np.log1p(np.exp(-np.abs(en))) + np.maximum(-en, 0)).sum())
        sq += float((e1 ** 2).sum() + (rr ** 2).sum() + (e2 ** 2).sum())
        g = ((1.0 / (1.0 + np.exp(-en)) - y) / N)[:, None]
        k = 2.0 * reg / (N * d)
        n = len(x)
        rows = np.arange(n)
        for idx, contrib, out in ((x[:, 0], g * rr * e2 + k * e1, dcodes), (x[:, 2], g * e1 * rr + k * e2, dcodes),
                                  (x[:, 1], g * e1 * e2 + k * rr, drel)):
            out += sp.coo_matrix((np.ones(n), (idx, rows)), shape=(out.shape[0], n)).tocsr() @ contrib
    return xent / N + reg * sq / (N * d), dcodes, drel


# ----------------------------------------------------------------------------- chunked float64 encoder (block kind)
_CHUNK_THREADS = min(8, os.cpu_count() or 1)


def _chunked_block_messages(H, W, rows_in, rel, rows_out, nrm, V, chunk, absolute=False):
    """sum over the messages e of  nrm[e] * (W[rel[e]] . H[rows_in[e]])  scattered to rows_out[e]  ([V, d], float64),
    edge chunk by edge chunk: gather W[type] ([chunk, nb, sd, sd]: gcn_basis_concat.py:38-39), the batched sd x sd
    products (:46-47), one sparse [V, chunk] x [chunk, d] product per chunk (:73-76).  absolute = True sums |terms|
    instead (the per-element error scale of the same sum)."""
    import scipy.sparse as sp
    from concurrent.futures import ThreadPoolExecutor
    R, nb, sd, _ = W.shape
    d = nb * sd
    E = len(rel)

    def one(lo):
        sl = slice(lo, lo + chunk)
        n = len(rel[sl])
        x = H[rows_in[sl]].reshape(n, nb, sd)
        T = W[rel[sl]]
        if absolute:
            m = np.einsum("ebij,ebj->ebi", np.abs(T), np.abs(x)).reshape(n, d)
        else:
            m = np.einsum("ebij,ebj->ebi", T, x).reshape(n, d)
        A = sp.coo_matrix((nrm[sl], (rows_out[sl], np.arange(n))), shape=(V, n)).tocsr()
        return A @ m

    out = np.zeros((V, d))
    with ThreadPoolExecutor(max_workers=_CHUNK_THREADS) as ex:       # numpy drops the GIL inside these calls
        for part in ex.map(one, range(0, E, chunk)):                 # added in chunk order whatever finishes first
            out += part
    return out


def chunked_block_layer_float64(params, l, L, H, triples, V, mode="train", keep=0.8, mask=None, norm_mode=None,
                                chunk=16384, with_scale=False, return_pre=False):
    """ONE block layer (message_gcn.py:49-79 around gcn_basis_concat.py:35-83) in float64 from the given layer input:
    relu?( dropout(H . W_self) + A_f F + A_b K ).  with_scale: also the sum of the ABSOLUTE values of the terms of every
    output element -- what the fp32 rounding error of that element is proportional to."""
    norm_mode = oracle.NORM_INTENDED if norm_mode is None else norm_mode
    s, r, o = oracle.split_graph(triples)
    n_f = oracle.incidence_values(o, V, norm_mode).astype(np.float64)
    n_b = oracle.incidence_values(s, V, norm_mode).astype(np.float64)
    H = np.asarray(H, dtype=np.float64)
    W_f, W_b = params["W_f%d" % l].astype(np.float64), params["W_b%d" % l].astype(np.float64)
    W_self = params["W_self%d" % l].astype(np.float64)
    S = H @ W_self
    drop = (np.asarray(mask, dtype=np.float64) / keep) if mode == "train" else None
    if drop is not None:
        S = S * drop
    # forward messages: sender s -> receiver o with W_f; backward messages: o -> s with W_b
    pre = (_chunked_block_messages(H, W_f, s, r, o, n_f, V, chunk) +
           _chunked_block_messages(H, W_b, o, r, s, n_b, V, chunk)) + S
    out = pre if return_pre else (np.maximum(pre, 0.0) if l < L else pre)
    if not with_scale:
        return out
    sc = (_chunked_block_messages(H, W_f, s, r, o, n_f, V, chunk, absolute=True) +
          _chunked_block_messages(H, W_b, o, r, s, n_b, V, chunk, absolute=True))
    Sabs = np.abs(H) @ np.abs(W_self)
    if drop is not None:
        Sabs = Sabs * drop
    return out, sc + Sabs


def chunked_block_encoder_forward_float64(params, triples, V, L, mode="train", keep=0.8, masks=None,
                                          norm_mode=None, chunk=16384, with_scale=False):
    """oracle.encoder_forward for the block kind (affine_transform.py:63-83, message_gcn.py:49-79,
    gcn_basis_concat.py:35-83) in float64, never holding an [E, d] or [E, nb, sd, sd] array whole -- what the
    272,115-edge GPU test compares with.  Returns [H0..HL] (and, with_scale, the per-element scales of every layer)."""
    H = np.maximum(params["W_emb"].astype(np.float64) + params["b_emb"].astype(np.float64), 0.0)
    acts, scales = [H], [np.abs(H)]
    for l in range(1, L + 1):
        res = chunked_block_layer_float64(params, l, L, H, triples, V, mode=mode, keep=keep,
                                          mask=masks[l - 1] if mode == "train" else None, norm_mode=norm_mode,
                                          chunk=chunk, with_scale=with_scale)
        H = res[0] if with_scale else res
        if with_scale:
            scales.append(res[1])
        acts.append(H)
    return (acts, scales) if with_scale else acts


def chunked_block_encoder_backward_float64(params, triples, V, L, acts, dcodes, mode="train", keep=0.8, masks=None,
                                           norm_mode=None, chunk=16384):
    """oracle.encoder_backward for the block kind (SURVEY 8a row a15 = tf.gradients of the forward above) in float64,
    edge chunk by edge chunk, evaluated at the given activations."""
    import scipy.sparse as sp
    from concurrent.futures import ThreadPoolExecutor
    norm_mode = oracle.NORM_INTENDED if norm_mode is None else norm_mode
    s, r, o = oracle.split_graph(triples)
    E = len(s)
    n_f = oracle.incidence_values(o, V, norm_mode).astype(np.float64)
    n_b = oracle.incidence_values(s, V, norm_mode).astype(np.float64)
    acts = [np.asarray(a, dtype=np.float64) for a in acts]
    grads = {}
    dH = np.asarray(dcodes, dtype=np.float64)
    for l in range(L, 0, -1):
        W_f, W_b = params["W_f%d" % l].astype(np.float64), params["W_b%d" % l].astype(np.float64)
        W_self = params["W_self%d" % l].astype(np.float64)
        R, nb, sd, _ = W_f.shape
        Hin, Hout = acts[l - 1], acts[l]
        D = dH * (Hout > 0) if l < L else dH
        dS = D * (np.asarray(masks[l - 1], dtype=np.float64) / keep) if mode == "train" else D
        grads["W_self%d" % l] = Hin.T @ dS
        dHin = dS @ W_self.T
        gW = {"f": np.zeros((R, nb * sd * sd)), "b": np.zeros((R, nb * sd * sd))}
        for tag, W, rows_in, rows_out, nrm in (("f", W_f, s, o, n_f), ("b", W_b, o, s, n_b)):
            def one(lo, W=W, rows_in=rows_in, rows_out=rows_out, nrm=nrm):
                sl = slice(lo, lo + chunk)
                n = len(r[sl])
                g = (D[rows_out[sl]] * nrm[sl, None]).reshape(n, nb, sd)        # dF[e] = n[e] D[receiver]
                x = Hin[rows_in[sl]].reshape(n, nb, sd)
                gT = np.einsum("ebi,ebj->ebij", g, x).reshape(n, nb * sd * sd)
                dw = sp.coo_matrix((np.ones(n), (r[sl], np.arange(n))), shape=(R, n)).tocsr() @ gT
                dx = np.einsum("ebij,ebi->ebj", W[r[sl]], g).reshape(n, nb * sd)
                return dw, sp.coo_matrix((np.ones(n), (rows_in[sl], np.arange(n))), shape=(V, n)).tocsr() @ dx
            with ThreadPoolExecutor(max_workers=_CHUNK_THREADS) as ex:
                for dw, dh in ex.map(one, range(0, E, chunk)):
                    gW[tag] += dw
                    dHin += dh
        grads["W_f%d" % l] = gW["f"].reshape(W_f.shape)
        grads["W_b%d" % l] = gW["b"].reshape(W_b.shape)
        grads["b%d" % l] = np.zeros(Hin.shape[1])
        dH = dHin
    g0 = dH * (acts[0] > 0)
    grads["W_emb"] = g0
    grads["b_emb"] = g0.sum(axis=0)
    return grads


# ----------------------------------------------------------------------------- chunked float64 encoder (basis kind)
def _chunked_basis_messages(H, W, C, rows_in, rel, rows_out, nrm, V, chunk, absolute=False):
    """sum over the messages e of  nrm[e] * sum_b C[rel[e], b] * (H[rows_in[e]] . W[:, b, :])  scattered to rows_out[e]
    ([V, d], float64), edge chunk by edge chunk in the REFERENCE's dataflow (gcn_basis.py:39-68: transform every edge's
    endpoint row with the [d, B.d] basis tensor, scale the [chunk, B, d] terms by the gathered coefficients, sum over
    the bases, then the sparse [V, chunk] x [chunk, d] product of gcn_basis.py:74-88) -- deliberately NOT the engine's
    aggregate-first form.  absolute = True sums |terms| instead (the per-element error scale of the same sum)."""
    import scipy.sparse as sp
    from concurrent.futures import ThreadPoolExecutor
    d_in, B, d_out = W.shape
    W2 = (np.abs(W) if absolute else W).reshape(d_in, B * d_out)
    Cx = np.abs(C) if absolute else C
    Hx = np.abs(H) if absolute else H
    E = len(rel)

    def one(lo):
        sl = slice(lo, lo + chunk)
        n = len(rel[sl])
        terms = (Hx[rows_in[sl]] @ W2).reshape(n, B, d_out)                     # gcn_basis.py:54-56,60-68
        m = np.einsum("ebk,eb->ek", terms, Cx[rel[sl]])                         # :43-44
        A = sp.coo_matrix((nrm[sl], (rows_out[sl], np.arange(n))), shape=(V, n)).tocsr()
        return A @ m

    out = np.zeros((V, d_out))
    with ThreadPoolExecutor(max_workers=_CHUNK_THREADS) as ex:
        for part in ex.map(one, range(0, E, chunk)):
            out += part
    return out


def chunked_basis_layer_float64(params, l, L, H, triples, V, mode="train", keep=0.8, mask=None, norm_mode=None,
                                chunk=16384, with_scale=False, return_pre=False):
    """ONE basis layer (message_gcn.py:49-79 around gcn_basis.py:39-88) in float64 from the given layer input.
    with_scale: also the sum of the absolute values of the terms of every output element."""
    norm_mode = oracle.NORM_INTENDED if norm_mode is None else norm_mode
    s, r, o = oracle.split_graph(triples)
    n_f = oracle.incidence_values(o, V, norm_mode).astype(np.float64)
    n_b = oracle.incidence_values(s, V, norm_mode).astype(np.float64)
    H = np.asarray(H, dtype=np.float64)
    W_f, W_b = params["W_f%d" % l].astype(np.float64), params["W_b%d" % l].astype(np.float64)
    C_f, C_b = params["C_f%d" % l].astype(np.float64), params["C_b%d" % l].astype(np.float64)
    W_self = params["W_self%d" % l].astype(np.float64)
    S = H @ W_self
    drop = (np.asarray(mask, dtype=np.float64) / keep) if mode == "train" else None
    if drop is not None:
        S = S * drop
    pre = (_chunked_basis_messages(H, W_f, C_f, s, r, o, n_f, V, chunk) +
           _chunked_basis_messages(H, W_b, C_b, o, r, s, n_b, V, chunk)) + S
    out = pre if return_pre else (np.maximum(pre, 0.0) if l < L else pre)
    if not with_scale:
        return out
    sc = (_chunked_basis_messages(H, W_f, C_f, s, r, o, n_f, V, chunk, absolute=True) +
          _chunked_basis_messages(H, W_b, C_b, o, r, s, n_b, V, chunk, absolute=True))
    Sabs = np.abs(H) @ np.abs(W_self)
    if drop is not None:
        Sabs = Sabs * drop
    return out, sc + Sabs


def chunked_basis_encoder_forward_float64(params, triples, V, L, mode="train", keep=0.8, masks=None,
                                          norm_mode=None, chunk=16384, with_scale=False):
    """oracle.encoder_forward for the basis kind in float64, never holding an [E, B, d] array whole."""
    H = np.maximum(params["W_emb"].astype(np.float64) + params["b_emb"].astype(np.float64), 0.0)
    acts, scales = [H], [np.abs(H)]
    for l in range(1, L + 1):
        res = chunked_basis_layer_float64(params, l, L, H, triples, V, mode=mode, keep=keep,
                                          mask=masks[l - 1] if mode == "train" else None, norm_mode=norm_mode,
                                          chunk=chunk, with_scale=with_scale)
        H = res[0] if with_scale else res
        if with_scale:
            scales.append(res[1])
        acts.append(H)
    return (acts, scales) if with_scale else acts


def chunked_basis_encoder_backward_float64(params, triples, V, L, acts, dcodes, mode="train", keep=0.8, masks=None,
                                           norm_mode=None, chunk=16384):
    """oracle.encoder_backward for the basis kind (SURVEY 8a row a15 = tf.gradients of gcn_basis.py:39-88) in float64,
    edge chunk by edge chunk in the reference's per-edge dataflow, evaluated at the given activations."""
    import scipy.sparse as sp
    from concurrent.futures import ThreadPoolExecutor
    norm_mode = oracle.NORM_INTENDED if norm_mode is None else norm_mode
    s, r, o = oracle.split_graph(triples)
    E = len(s)
    n_f = oracle.incidence_values(o, V, norm_mode).astype(np.float64)
    n_b = oracle.incidence_values(s, V, norm_mode).astype(np.float64)
    acts = [np.asarray(a, dtype=np.float64) for a in acts]
    grads = {}
    dH = np.asarray(dcodes, dtype=np.float64)
    for l in range(L, 0, -1):
        W_f, W_b = params["W_f%d" % l].astype(np.float64), params["W_b%d" % l].astype(np.float64)
        C_f, C_b = params["C_f%d" % l].astype(np.float64), params["C_b%d" % l].astype(np.float64)
        W_self = params["W_self%d" % l].astype(np.float64)
        d_in, B, d_out = W_f.shape
        R = C_f.shape[0]
        Hin, Hout = acts[l - 1], acts[l]
        D = dH * (Hout > 0) if l < L else dH
        dS = D * (np.asarray(masks[l - 1], dtype=np.float64) / keep) if mode == "train" else D
        grads["W_self%d" % l] = Hin.T @ dS
        dHin = dS @ W_self.T
        for tag, W, C, rows_in, rows_out, nrm in (("f", W_f, C_f, s, o, n_f), ("b", W_b, C_b, o, s, n_b)):
            W2 = W.reshape(d_in, B * d_out)

            def one(lo, W2=W2, C=C, rows_in=rows_in, rows_out=rows_out, nrm=nrm):
                sl = slice(lo, lo + chunk)
                n = len(r[sl])
                g = D[rows_out[sl]] * nrm[sl, None]                              # dF[e] = n[e] D[receiver]
                x = Hin[rows_in[sl]]
                terms = (x @ W2).reshape(n, B, d_out)
                gscale = np.einsum("ebk,ek->eb", terms, g)                       # d/dC[rel[e], b]
                dc = sp.coo_matrix((np.ones(n), (r[sl], np.arange(n))), shape=(R, n)).tocsr() @ gscale
                gterms = (C[r[sl]][:, :, None] * g[:, None, :]).reshape(n, B * d_out)
                dw = x.T @ gterms
                dx = gterms @ W2.T
                return dc, dw, sp.coo_matrix((np.ones(n), (rows_in[sl], np.arange(n))), shape=(V, n)).tocsr() @ dx
            gC, gW = np.zeros((R, B)), np.zeros((d_in, B * d_out))
            with ThreadPoolExecutor(max_workers=_CHUNK_THREADS) as ex:
                for dc, dw, dh in ex.map(one, range(0, E, chunk)):
                    gC += dc
                    gW += dw
                    dHin += dh
            grads["C_%s%d" % (tag, l)] = gC
            grads["W_%s%d" % (tag, l)] = gW.reshape(W.shape)
        grads["b%d" % l] = np.zeros(Hin.shape[1])
        dH = dHin
    g0 = dH * (acts[0] > 0)
    grads["W_emb"] = g0
    grads["b_emb"] = g0.sum(axis=0)
    return grads
