"""Pin the oracle three independent ways (the reference has no golden vectors -> "parity unpinned"):
  1. dense closed-form restatement (different structure, float64),
  2. torch-CPU autograd of a TF-shaped torch restatement (gradients),
  3. hand-computed tiny cases + normalisation-mode semantics.
"""
import numpy as np
import pytest

import oracle
from helpers import make_case

torch = pytest.importorskip("torch")


def torch_forward(params, triples, V, L, kind, masks, keep, norm_mode, train=True):
    """TF-dataflow-shaped torch restatement (embedding lookups, batched matmul, index_add)."""
    s = torch.as_tensor(triples[:, 0].astype(np.int64))
    r = torch.as_tensor(triples[:, 1].astype(np.int64))
    o = torch.as_tensor(triples[:, 2].astype(np.int64))
    n_f = torch.as_tensor(oracle.incidence_values(triples[:, 2], V, norm_mode))
    n_b = torch.as_tensor(oracle.incidence_values(triples[:, 0], V, norm_mode))
    H = torch.relu(params["W_emb"] + params["b_emb"])
    acts = [H]
    for l in range(1, L + 1):
        W_f, W_b, W_self = params[f"W_f{l}"], params[f"W_b{l}"], params[f"W_self{l}"]
        xs, xr = H[s], H[o]
        E = xs.shape[0]
        if kind == oracle.KIND_BLOCK:
            nb, sd = W_f.shape[1], W_f.shape[2]
            F = torch.matmul(W_f[r], xs.reshape(E, nb, sd, 1)).reshape(E, -1)
            K = torch.matmul(W_b[r], xr.reshape(E, nb, sd, 1)).reshape(E, -1)
        else:
            d_in, B, d_out = W_f.shape
            st = (xs @ W_f.reshape(d_in, B * d_out)).reshape(E, B, d_out)
            rt = (xr @ W_b.reshape(d_in, B * d_out)).reshape(E, B, d_out)
            F = (st * params[f"C_f{l}"][r].unsqueeze(-1)).sum(1)
            K = (rt * params[f"C_b{l}"][r].unsqueeze(-1)).sum(1)
        S = H @ W_self
        if train:
            S = S * torch.as_tensor(masks[l - 1].astype(np.float32)) / keep
        agg = torch.zeros_like(S).index_add(0, o, F * n_f[:, None]).index_add(0, s, K * n_b[:, None])
        pre = agg + S
        H = torch.relu(pre) if l < L else pre
        acts.append(H)
    return acts


CASES = [
    # V, R, d, L, kind, nb, E
    (16, 9, 10, 1, "basis", 2, 43),
    (16, 9, 10, 2, "block", 2, 43),
    (50, 7, 20, 2, "block", 4, 200),
    (50, 7, 12, 2, "basis", 5, 200),
    (30, 3, 8, 3, "block", 8, 60),     # sd = 1
    (30, 3, 8, 2, "block", 1, 60),     # nb = 1 (one dense block)
    (20, 4, 6, 2, "basis", 1, 1),      # E = 1
]


@pytest.mark.parametrize("V,R,d,L,kind,nb,E", CASES)
@pytest.mark.parametrize("norm_mode", [oracle.NORM_INTENDED, oracle.NORM_TF_AS_EXECUTED])
def test_oracle_matches_torch_autograd(V, R, d, L, kind, nb, E, norm_mode):
    params, triples, masks, dcodes = make_case(V, R, d, L, kind, nb, E, seed=3)
    acts, grads = oracle.encoder_step(params, triples, V, L, kind, dcodes, keep_prob=0.8,
                                      dropout_masks=masks, norm_mode=norm_mode)
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in params.items()}
    tacts = torch_forward(tp, triples, V, L, kind, masks, 0.8, norm_mode)
    for a, ta in zip(acts, tacts):
        np.testing.assert_allclose(a, ta.detach().numpy(), rtol=1e-5, atol=1e-5)
    (tacts[-1] * torch.as_tensor(dcodes)).sum().backward()
    for name in oracle.weight_names(kind, L):
        if name == "W_relation":
            continue
        g = tp[name].grad
        if name.startswith("b") and name != "b_emb":
            assert g is None                      # bias created but unused: SURVEY H2
            assert not grads[name].any()
            continue
        np.testing.assert_allclose(grads[name], g.numpy(), rtol=2e-4, atol=2e-5, err_msg=name)


@pytest.mark.parametrize("kind,nb", [("block", 2), ("block", 5), ("basis", 2), ("basis", 3)])
@pytest.mark.parametrize("train", [True, False])
def test_oracle_matches_dense_closed_form(kind, nb, train):
    V, R, d, L, E = 12, 4, 10, 2, 40
    params, triples, masks, _ = make_case(V, R, d, L, kind, nb, E, seed=5, train=train)
    mode = "train" if train else "test"
    acts = oracle.encoder_forward(params, triples, V, L, kind, mode=mode, dropout_masks=masks)
    dense = oracle.dense_closed_form_forward(params, triples, V, L, kind, mode=mode, dropout_masks=masks)
    for a, b in zip(acts, dense):
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-5)


def test_hand_computed_block_layer():
    # V=3, one relation, d=2, nb=1 (one 2x2 block), edges 0->2 and 1->2
    V, d = 3, 2
    params = {
        "W_emb": np.array([[1, 2], [3, -4], [0.5, 0.5]], np.float32), "b_emb": np.zeros(2, np.float32),
        "W_f1": np.array([[[[1, 2], [3, 4]]]], np.float32),     # [R=1, nb=1, 2, 2]
        "W_b1": np.array([[[[0, 1], [1, 0]]]], np.float32),
        "W_self1": np.eye(2, dtype=np.float32), "b1": np.zeros(2, np.float32),
    }
    triples = np.array([[0, 0, 2], [1, 0, 2]], np.int32)
    H0 = np.array([[1, 2], [3, 0], [0.5, 0.5]], np.float32)      # relu
    acts = oracle.encoder_forward(params, triples, V, 1, "block", mode="test")
    np.testing.assert_array_equal(acts[0], H0)
    T = np.array([[1, 2], [3, 4]], np.float32)
    f0, f1 = T @ H0[0], T @ H0[1]                                  # message = T x (matrix . vector)
    Tb = np.array([[0, 1], [1, 0]], np.float32)
    expect = H0.copy()                                             # self loop = identity
    expect[2] += 0.5 * (f0 + f1)                                   # in-degree(2) = 2
    expect[0] += 1.0 * (Tb @ H0[2])                                # out-degree(0) = 1
    expect[1] += 1.0 * (Tb @ H0[2])
    np.testing.assert_allclose(acts[1], expect, rtol=1e-6)        # last layer: no relu


def test_norm_modes():
    o = np.array([2, 0, 2, 1, 2], np.int32)
    np.testing.assert_allclose(oracle.incidence_values(o, 3, "intended"), [1 / 3, 1, 1 / 3, 1, 1 / 3])
    # sorted rows = [0,1,2,2,2] -> values in that order re-attached to edge order
    np.testing.assert_allclose(oracle.incidence_values(o, 3, "tf_as_executed"), [1, 1, 1 / 3, 1 / 3, 1 / 3])
    np.testing.assert_array_equal(oracle.incidence_values(o, 3, "none"), np.ones(5, np.float32))
    srt = np.sort(o)
    np.testing.assert_array_equal(oracle.incidence_values(srt, 3, "intended"),
                                  oracle.incidence_values(srt, 3, "tf_as_executed"))


def test_empty_graph_and_isolated_rows():
    V, R, d, L = 8, 3, 4, 2
    params, _, masks, dcodes = make_case(V, R, d, L, "block", 2, 5, seed=1)
    empty = np.zeros((0, 3), np.int32)
    acts, grads = oracle.encoder_step(params, empty, V, L, "block", dcodes, dropout_masks=masks)
    # no edges: only the self loop contributes and relation weights get zero gradient
    H0 = acts[0]
    S = (H0 @ params["W_self1"]) * masks[0] / np.float32(0.8)
    np.testing.assert_allclose(acts[1], np.maximum(S, 0), rtol=1e-6, atol=1e-7)
    assert not grads["W_f1"].any() and not grads["W_b2"].any()


def test_init_distributions_and_order():
    V, R, d, L = 400, 11, 20, 2
    rs = np.random.RandomState(0)
    p = oracle.init_params(V, R, d, L, "block", 4, rng=rs)
    assert p["W_relation"].shape == (V, d)
    assert p["W_f1"].shape == (R, 4, 5, 5) and p["W_self2"].shape == (d, d)
    assert abs(p["W_f2"].std() - 3 / np.sqrt(R + 5)) < 0.05        # SURVEY H4 (value used as std)
    assert abs(p["W_emb"].std() - 3 / np.sqrt(V + d)) < 0.01
    # consumption order: W_relation is drawn first (outermost component first)
    rs2 = np.random.RandomState(0)
    np.testing.assert_array_equal(p["W_relation"], rs2.randn(V, d).astype(np.float32))
    pb = oracle.init_params(V, R, d, L, "basis", 3, rng=np.random.RandomState(1))
    assert pb["W_f1"].shape == (d, 3, d) and pb["C_b2"].shape == (R, 3)
    assert oracle.weight_names("basis", 1) == ["W_emb", "b_emb", "W_f1", "W_b1", "C_f1", "C_b1",
                                               "W_self1", "b1", "W_relation"]
    with pytest.raises(ValueError):
        oracle.init_params(V, R, 10, 1, "block", 3)


def test_distmult_grads_match_torch():
    rng = np.random.RandomState(0)
    V, d, N = 20, 6, 50
    codes = rng.randn(V, d).astype(np.float32)
    Wr = rng.randn(V, d).astype(np.float32)
    X = np.stack([rng.randint(0, V, N), rng.randint(0, 5, N), rng.randint(0, V, N)], 1).astype(np.int32)
    Y = (rng.rand(N) < 0.3).astype(np.float32)
    loss, dcodes, dWr = oracle.distmult_loss_and_grads(codes, Wr, X, Y, 0.01)
    tc, tw = torch.tensor(codes, requires_grad=True), torch.tensor(Wr, requires_grad=True)
    Xt = torch.as_tensor(X.astype(np.int64))
    e1, rr, e2 = tc[Xt[:, 0]], tw[Xt[:, 1]], tc[Xt[:, 2]]
    x = (e1 * rr * e2).sum(1)
    tl = torch.nn.functional.binary_cross_entropy_with_logits(x, torch.as_tensor(Y)) \
        + 0.01 * ((e1 ** 2).mean() + (rr ** 2).mean() + (e2 ** 2).mean())
    tl.backward()
    assert abs(loss - tl.item()) < 1e-5
    np.testing.assert_allclose(dcodes, tc.grad.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(dWr, tw.grad.numpy(), rtol=1e-4, atol=1e-6)
