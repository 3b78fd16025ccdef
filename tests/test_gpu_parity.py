"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against the CPU oracle on the
same seeded inputs.  Tolerance: forward activations within 1e-4 absolute (BASELINE.json north_star,
fp32) -- enforced as max|err| <= 1e-4 on every element; gradients: l2 error and >= 99 % of the entries
within 2e-4 of the tensor's scale, no entry off by more than 5e-3 x scale (isolated relu-gate flips,
see tests/helpers.py; fp32 summation order differs from the oracle's).  GEMM alone: 1e-5.
"""
import os

import numpy as np
import pytest

import oracle
import helpers
from helpers import assert_close, make_case

pytestmark = pytest.mark.gpu

FWD_ATOL = 1e-4


@pytest.fixture(scope="module")
def native():
    from relationprediction_amd import _native
    _native.load_library()
    return _native


def run_engine(native, V, R, d, L, kind, nb, params, triples, masks, dcodes, norm="intended", train=True,
               keep=0.8, seed=0):
    eng = native.Engine(V, R, d, L, kind, nb, keep_prob=keep, norm_mode=norm, max_edges=max(len(triples), 1))
    try:
        eng.set_params(params)
        eng.set_graph(triples)
        eng.forward(train=train, seed=seed, masks=masks if train else None)
        acts = [eng.activation(l) for l in range(L + 1)]
        eng.backward(dcodes)
        grads = eng.get_grads()
        return acts, grads
    finally:
        eng.close()


def compare(acts, grads, oacts, ograds, tag=""):
    for l, (a, b) in enumerate(zip(acts, oacts)):
        assert a.shape == b.shape
        assert np.isfinite(a).all(), "%s H%d non-finite" % (tag, l)
        err = float(np.abs(a - b).max()) if a.size else 0.0
        assert err <= FWD_ATOL, "%s H%d: max abs err %.3e" % (tag, l, err)
    for name, g in grads.items():
        if name == "W_relation":        # decoder weight: the encoder path never touches its gradient
            continue
        assert_close(g, ograds[name], rel=2e-4, name="%s grad %s" % (tag, name))


def _bench_module():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_graphs", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def gate_aware_gradient_check(c, acts, grads, norm, direct):
    """Second line of the gradient check (helpers.gate_consistent_oracle_grads): only taken when a handful
    of relu gates sit within rounding of zero; then the engine's backward must match the oracle's
    reverse mode of the engine's own forward, with NO spike allowance."""
    ograds, flips = helpers.gate_consistent_oracle_grads(c, acts, norm)
    total = sum(int(np.asarray(a).size) for a in acts[:-1])
    if flips == 0 or flips > 1e-5 * total:
        raise direct
    for name, g in grads.items():
        if name != "W_relation":
            assert_close(g, ograds[name], rel=2e-4, spike=2e-4, name="gate-consistent grad %s (%d flips)" % (name, flips))


# ------------------------------------------------------------------ dense contraction (fp32 MFMA GEMM)
GEMM_SHAPES = [
    (1, 1, 1), (33, 17, 5), (128, 128, 16), (129, 130, 33), (257, 500, 500), (1000, 500, 500),
    (64, 64, 4), (500, 500, 1000), (300, 77, 129),
    # the basis kind at settings/gcn_basis.exp:5's B = 5: a 5,000-wide (2.B.d) K (forward), N (dZ) and M (dW', split-K)
    (300, 500, 5000), (300, 5000, 500), (5000, 500, 300),
]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("form", ["NN", "NT", "TN"])
@pytest.mark.parametrize("mode", [6, 9, 0])
def test_gemm_forms(native, M, N, K, form, mode):
    """Every storage form on every arithmetic (include/rgcn.h rgcn_set_gemm_mode): 6 / 9 = exact bf16
    operand split on the bf16 matrix cores, 0 = fp32 MFMA.  Same tolerance for all three."""
    rng = np.random.RandomState(M * 7 + N * 3 + K)
    A = rng.randn(M, K).astype(np.float32)
    B = rng.randn(K, N).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    eng = native.Engine(16, 2, 8, 1, "block", 2, max_edges=4, devtools=True)   # rgcn_debug_gemm: devtools build only
    try:
        eng.set_gemm_mode(mode)
        if form == "NN":
            got = eng.debug_gemm(A, B)
        elif form == "NT":
            got = eng.debug_gemm(A, np.ascontiguousarray(B.T), trans_b=True)
        else:
            got = eng.debug_gemm(np.ascontiguousarray(A.T), B, trans_a=True)
            got2 = eng.debug_gemm(np.ascontiguousarray(A.T), B, trans_a=True, split_k=3)
            assert_close(got2, ref, rel=1e-5, spike=1e-5, name="TN split 3")
    finally:
        eng.close()
    # asymmetric random operands: a swapped row/col mapping cannot pass
    assert_close(got, ref, rel=1e-5, spike=1e-5, name=form)


@pytest.mark.parametrize("M,N,K", [(1, 4, 4), (129, 132, 36), (257, 500, 500), (1000, 500, 500), (300, 500, 5000),
                                   (300, 5000, 500), (64, 64, 4), (14541, 500, 500)])
@pytest.mark.parametrize("trans_b", [False, True])
@pytest.mark.parametrize("mode", [6, 9])
def test_gemm_with_presplit_weights_is_bitwise_the_staged_kernel(native, M, N, K, trans_b, mode):
    """The weights (W_self, the basis tensors) reach the split-arithmetic GEMM pre-split into bf16 planes in MFMA fragment
    order (gemm_bf16x3.hip B_PRE: straight from L2 into registers, no LDS staging for B): the same products in the same
    order as the staged kernel -- bitwise equal, ragged edges in M, N and K included -- and right against float64."""
    rng = np.random.RandomState(M + 3 * N + 7 * K)
    A = rng.randn(M, K).astype(np.float32)
    B = (rng.randn(K, N) * np.exp(rng.uniform(-6, 6, (K, N)))).astype(np.float32)
    Bop = np.ascontiguousarray(B.T) if trans_b else B
    # two kernels take a pre-split weight (csrc/gemm_bf16x3.hip picks by call site and tile count): the 128 x 128 /
    # four-wavefront one whose fragments go straight to registers (B_PRE), and the 128 x 256 / eight-wavefront one that
    # brings them in by LDS-DMA (gemm_bf16x3_w8.hip, round 6).  The devtools knob forces each in turn.
    prev = os.environ.get("RGCN_GEMM_W8")
    try:
        with native.Engine(16, 2, 8, 1, "block", 2, max_edges=4, devtools=True) as eng:
            eng.set_gemm_mode(mode)
            staged = eng.debug_gemm(A, Bop, trans_b=trans_b, split_k=1)
            os.environ["RGCN_GEMM_W8"] = "0"
            pre_regs = eng.debug_gemm_presplit(A, Bop, trans_b=trans_b)
            os.environ["RGCN_GEMM_W8"] = "3"
            pre = eng.debug_gemm_presplit(A, Bop, trans_b=trans_b)
    finally:
        if prev is None:
            os.environ.pop("RGCN_GEMM_W8", None)
        else:
            os.environ["RGCN_GEMM_W8"] = prev
    np.testing.assert_array_equal(pre_regs, staged)
    np.testing.assert_array_equal(pre, staged)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    mag = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    assert float((np.abs(pre - ref) / mag).max()) < 2e-6


def test_presplit_weight_kernels_on_random_shapes(native):
    """The two kernels that take a pre-split weight, on 24 seeded random shapes (M 1..700, N and K multiples of 4 up to 520 /
    260: one to six row tiles, one to three 256-column tiles, K from a quarter of a k-tile to 16 k-tiles -- fewer stages than
    the eight-wavefront kernel's ring is deep, the partial last tile, the stages past the end): bitwise the staged kernel."""
    rng = np.random.RandomState(20260930)
    prev = os.environ.get("RGCN_GEMM_W8")
    try:
        with native.Engine(16, 2, 8, 1, "block", 2, max_edges=4, devtools=True) as eng:
            eng.set_gemm_mode(6)
            for case in range(24):
                M = int(rng.randint(1, 701))
                K = 4 * int(rng.randint(1, 66))
                trans_b = bool(rng.randint(0, 2))
                # (a k-contiguous B may have any number of columns: C is then stored element by element)
                N = int(rng.randint(1, 521)) if trans_b else 4 * int(rng.randint(1, 131))
                A = rng.randn(M, K).astype(np.float32)
                B = (rng.randn(K, N) * np.exp(rng.uniform(-4, 4, (K, N)))).astype(np.float32)
                Bop = np.ascontiguousarray(B.T) if trans_b else B
                os.environ.pop("RGCN_GEMM_W8", None)
                staged = eng.debug_gemm(A, Bop, trans_b=trans_b, split_k=1)
                for code in ("0", "3"):
                    os.environ["RGCN_GEMM_W8"] = code
                    got = eng.debug_gemm_presplit(A, Bop, trans_b=trans_b)
                    np.testing.assert_array_equal(got, staged, err_msg="case %d: M %d N %d K %d trans_b %s kernel %s"
                                                  % (case, M, N, K, trans_b, code))
    finally:
        if prev is None:
            os.environ.pop("RGCN_GEMM_W8", None)
        else:
            os.environ["RGCN_GEMM_W8"] = prev


def test_gemm_modes_are_fp32_accurate(native):
    """The split evaluation must be as accurate as the fp32 MFMA on the encoder's own shapes: error against
    float64, normalised by sum |a||b| (the scale fp32 rounding errors live on), within 1.5x of mode 0's, and
    operands spanning 12 orders of magnitude must not lose their small entries."""
    V, d = 3000, 500
    rng = np.random.RandomState(5)
    H = np.maximum(rng.randn(V, d), 0).astype(np.float32)
    W = (rng.randn(d, d) * 0.19).astype(np.float32)
    D = (rng.randn(V, d) * 1e-3).astype(np.float32)
    wide = (rng.randn(V, d) * np.exp(rng.uniform(-14, 14, (V, d)))).astype(np.float32)
    cases = {"NN": (H, W, False, False), "NT": (D, W, False, True), "TN": (H, D, True, False),
             "NN wide": (wide, W, False, False)}
    eng = native.Engine(V, 2, d, 1, "block", 100, max_edges=4, devtools=True)
    try:
        for name, (a, b, ta, tb) in cases.items():
            a64 = (a.T if ta else a).astype(np.float64)
            b64 = (b.T if tb else b).astype(np.float64)
            ref, mag = a64 @ b64, np.abs(a64) @ np.abs(b64)
            err = {}
            for mode in (0, 6, 9):
                eng.set_gemm_mode(mode)
                got = eng.debug_gemm(a, b, trans_a=ta, trans_b=tb).astype(np.float64)
                assert np.isfinite(got).all()
                e = np.abs(got - ref) / mag
                err[mode] = (float(e.max()), float(np.sqrt((e ** 2).mean())))
            for mode in (6, 9):
                assert err[mode][0] <= 1.5 * err[0][0] + 1e-9, (name, mode, err)
                assert err[mode][1] <= 1.5 * err[0][1] + 1e-10, (name, mode, err)
            assert err[0][0] < 2e-6, (name, err)       # and all of them are fp32-grade
    finally:
        eng.close()


# ------------------------------------------------------------------ graph preparation
def test_graph_prep_degrees_and_csr(native):
    rng = np.random.RandomState(0)
    V, R, E = 97, 5, 400
    triples = np.stack([rng.randint(0, V, E), rng.randint(0, R, E), rng.randint(0, V, E)], 1).astype(np.int32)
    triples[:10, 2] = triples[:10, 0]          # self edges
    triples[10:20] = triples[0]                # duplicates
    eng = native.Engine(V, R, 8, 1, "block", 2, max_edges=E)
    try:
        eng.set_graph(triples)
        eng.sync()
        indeg = eng.read_buffer(native.BUF_INDEG)
        outdeg = eng.read_buffer(native.BUF_OUTDEG)
        rowptr = eng.read_buffer(native.BUF_ROWPTR)
    finally:
        eng.close()
    np.testing.assert_array_equal(indeg, np.bincount(triples[:, 2], minlength=V))
    np.testing.assert_array_equal(outdeg, np.bincount(triples[:, 0], minlength=V))
    cnt = np.bincount(triples[:, 2], minlength=V) + np.bincount(triples[:, 0], minlength=V)
    np.testing.assert_array_equal(rowptr, np.concatenate([[0], np.cumsum(cnt)]))


@pytest.mark.parametrize("V,R,E", [(97, 5, 400), (14541, 237, 15000), (40943, 18, 10000), (300, 1345, 2049),
                                   (70000, 3, 5000), (14541, 237, 38001), (90000, 40, 700000), (14951, 1345, 60000),
                                   (6000000, 200, 3000)])
def test_graph_prep_orderings_equal_a_stable_sort(native, V, R, E):
    """the library's own sort (csr_sort.hip) against numpy's stable argsort: incidences by (vertex, directed relation)
    -- a row's slots in relation order, so that the layer kernels find runs of one relation -- and messages by directed
    relation, ties in index order -- bit-exact (integer work), incl. more than one 2048-item block, key ranges of three and
    four radix passes (FB15k's 14,951 x 2,690 pairs), hub rows, 1.4 M items (684 blocks: the column-prefix pass of large
    sorts, k_sort_prefix), and an entity count whose pair key does not fit the sort's 31 bits (the key is then the vertex
    alone: incidence order inside a row)"""
    rng = np.random.RandomState(V + E)
    hub = rng.randint(0, V, 8)
    s = np.where(rng.rand(E) < 0.2, hub[rng.randint(0, 8, E)], rng.randint(0, V, E))
    o = np.where(rng.rand(E) < 0.2, hub[rng.randint(0, 8, E)], rng.randint(0, V, E))
    r = np.minimum((rng.pareto(1.0, E) * 3).astype(np.int64), R - 1)
    triples = np.stack([s, r, o], 1).astype(np.int32)
    eng = native.Engine(V, R, 8, 1, "block", 2, max_edges=E)
    try:
        eng.set_graph(triples)
        eng.sync()
        permv = eng.read_buffer(native.BUF_PERM_VERTEX)
        permr = eng.read_buffer(native.BUF_PERM_RELATION)
        rowptr = eng.read_buffer(native.BUF_ROWPTR)
    finally:
        eng.close()
    keyv = np.concatenate([triples[:, 2], triples[:, 0]])
    keyr = np.concatenate([triples[:, 1], R + triples[:, 1]])
    pair_key = (V + 1) * 2 * R < 2 ** 31
    np.testing.assert_array_equal(permv, np.argsort(keyv.astype(np.int64) * (2 * R) + keyr if pair_key else keyv, kind="stable"))
    np.testing.assert_array_equal(permr, np.argsort(keyr, kind="stable"))
    np.testing.assert_array_equal(rowptr, np.concatenate([[0], np.cumsum(np.bincount(keyv, minlength=V))]))


# ------------------------------------------------------------------ encoder forward + backward
BLOCK_CASES = [
    # V, R, d, L, nb, E
    (16, 9, 10, 1, 2, 43),       # Toy-sized, sd = 5
    (16, 9, 10, 2, 2, 43),
    (50, 7, 20, 2, 4, 200),      # sd = 5
    (64, 3, 16, 2, 4, 150),      # sd = 4
    (30, 3, 8, 3, 8, 60),        # sd = 1, three layers
    (30, 3, 8, 2, 1, 60),        # nb = 1: one dense 8x8 block
    (40, 11, 9, 2, 3, 120),      # sd = 3, d % 4 != 0 (scalar paths)
    (20, 4, 6, 2, 3, 1),         # E = 1
    (25, 6, 12, 2, 6, 0),        # empty graph
    (300, 40, 500, 2, 100, 900), # the real block geometry (nb=100, sd=5) at small V
]


@pytest.mark.parametrize("V,R,d,L,nb,E", BLOCK_CASES)
@pytest.mark.parametrize("norm", ["intended", "tf_as_executed"])
def test_block_encoder_matches_oracle(native, V, R, d, L, nb, E, norm):
    params, triples, masks, dcodes = make_case(V, R, d, L, "block", nb, E, seed=V + E)
    oacts, ograds = oracle.encoder_step(params, triples, V, L, "block", dcodes, keep_prob=0.8,
                                        dropout_masks=masks, norm_mode=norm)
    acts, grads = run_engine(native, V, R, d, L, "block", nb, params, triples, masks, dcodes, norm=norm)
    compare(acts, grads, oacts, ograds, tag="block")


BASIS_CASES = [
    # V, R, d, L, B, E
    (16, 9, 10, 1, 5, 43),       # BASELINE config 1 shape family (Toy-sized, 1 layer)
    (16, 9, 12, 2, 2, 43),
    (50, 7, 20, 2, 3, 200),
    (40, 11, 9, 2, 2, 120),      # d % 4 != 0 (scalar paths)
    (20, 4, 8, 2, 1, 1),         # E = 1, B = 1
    (25, 6, 12, 2, 2, 0),        # empty graph
    (30, 5, 16, 3, 9, 90),       # B > 8: two register passes
    (300, 40, 500, 2, 2, 900),   # real width d = 500, B = 2
    (300, 40, 500, 2, 5, 900),   # settings/gcn_basis.exp:5: B = 5 at d = 500 (5,000-wide contraction)
    (600, 9, 64, 2, 3, 200),     # most rows without a message in either direction (the row-compacted contraction)
]


@pytest.mark.parametrize("V,R,d,L,B,E", BASIS_CASES)
@pytest.mark.parametrize("norm", ["intended", "tf_as_executed"])
def test_basis_encoder_matches_oracle(native, V, R, d, L, B, E, norm):
    params, triples, masks, dcodes = make_case(V, R, d, L, "basis", B, E, seed=V + E + 1)
    oacts, ograds = oracle.encoder_step(params, triples, V, L, "basis", dcodes, keep_prob=0.8,
                                        dropout_masks=masks, norm_mode=norm)
    acts, grads = run_engine(native, V, R, d, L, "basis", B, params, triples, masks, dcodes, norm=norm)
    compare(acts, grads, oacts, ograds, tag="basis")


def test_basis_hub_and_test_mode(native):
    V, R, d, L, B = 60, 12, 20, 2, 3
    params, _, masks, dcodes = make_case(V, R, d, L, "basis", B, 10, seed=19)
    rng = np.random.RandomState(4)
    hub = np.stack([rng.randint(0, 30, 250), rng.randint(0, 3, 250), np.full(250, 7)], 1)
    dup = np.tile(np.array([[1, 2, 3]]), (17, 1))
    triples = np.concatenate([hub, dup, np.stack([np.arange(10), np.full(10, 4), np.arange(10)], 1)]).astype(np.int32)
    oacts = oracle.encoder_forward(params, triples, V, L, "basis", mode="test")
    ograds = oracle.encoder_backward(params, triples, V, L, "basis", oacts, dcodes, mode="test")
    acts, grads = run_engine(native, V, R, d, L, "basis", B, params, triples, None, dcodes, train=False)
    compare(acts, grads, oacts, ograds, tag="basis-hub")
    assert not grads["C_f1"][5:].any()


def test_block_edge_structures(native):
    """self edges, duplicate edges, a hub row, relations with zero edges, isolated vertices."""
    V, R, d, L, nb = 60, 12, 20, 2, 4
    params, _, masks, dcodes = make_case(V, R, d, L, "block", nb, 10, seed=9)
    rng = np.random.RandomState(4)
    hub = np.stack([rng.randint(0, 30, 250), rng.randint(0, 3, 250), np.full(250, 7)], 1)   # in-degree 250
    selfe = np.stack([np.arange(10), np.full(10, 4), np.arange(10)], 1)
    dup = np.tile(np.array([[1, 2, 3]]), (17, 1))
    triples = np.concatenate([hub, selfe, dup]).astype(np.int32)    # relations 5..11 unused, vertices 30..59 isolated
    oacts, ograds = oracle.encoder_step(params, triples, V, L, "block", dcodes, dropout_masks=masks)
    acts, grads = run_engine(native, V, R, d, L, "block", nb, params, triples, masks, dcodes)
    compare(acts, grads, oacts, ograds, tag="edge-structures")
    assert not grads["W_f1"][5:].any() and not grads["W_b2"][5:].any()


def test_test_mode_has_no_dropout(native):
    V, R, d, L, nb, E = 40, 5, 20, 2, 4, 100
    params, triples, masks, dcodes = make_case(V, R, d, L, "block", nb, E, seed=2)
    oacts = oracle.encoder_forward(params, triples, V, L, "block", mode="test")
    ograds = oracle.encoder_backward(params, triples, V, L, "block", oacts, dcodes, mode="test")
    acts, grads = run_engine(native, V, R, d, L, "block", nb, params, triples, None, dcodes, train=False)
    compare(acts, grads, oacts, ograds, tag="test-mode")


def test_generated_dropout_is_consistent_and_bernoulli(native):
    V, R, d, L, nb, E = 200, 5, 40, 2, 8, 500
    params, triples, _, dcodes = make_case(V, R, d, L, "block", nb, E, seed=6)
    eng = native.Engine(V, R, d, L, "block", nb, keep_prob=0.8, max_edges=E)
    try:
        eng.set_params(params)
        eng.set_graph(triples)
        eng.forward(train=True, seed=1234)
        masks = [eng.dropout_mask(l) for l in range(1, L + 1)]
        acts = [eng.activation(l) for l in range(L + 1)]
        eng.backward(dcodes)
        grads = eng.get_grads()
        eng.forward(train=True, seed=1235)
        other = eng.dropout_mask(1)
    finally:
        eng.close()
    for m in masks:
        assert set(np.unique(m)) <= {0, 1}
        assert abs(m.mean() - 0.8) < 0.02
    assert (masks[0] != masks[1]).mean() > 0.2 and (masks[0] != other).mean() > 0.2
    oacts, ograds = oracle.encoder_step(params, triples, V, L, "block", dcodes, dropout_masks=masks)
    compare(acts, grads, oacts, ograds, tag="rng-dropout")


def test_bitwise_deterministic(native):
    V, R, d, L, nb, E = 120, 9, 20, 2, 4, 700
    params, triples, masks, dcodes = make_case(V, R, d, L, "block", nb, E, seed=8)
    a1, g1 = run_engine(native, V, R, d, L, "block", nb, params, triples, masks, dcodes)
    a2, g2 = run_engine(native, V, R, d, L, "block", nb, params, triples, masks, dcodes)
    for x, y in zip(a1, a2):
        np.testing.assert_array_equal(x, y)
    for k in g1:
        np.testing.assert_array_equal(g1[k], g2[k])


def test_prefetched_graph_gives_identical_step(native):
    """rgcn_prefetch_graph_device + rgcn_step_device == rgcn_step_device alone (bitwise), alternating graphs."""
    V, R, d, L, nb, E = 150, 9, 20, 2, 4, 800
    params, tri_a, _, dcodes = make_case(V, R, d, L, "block", nb, E, seed=31)
    _, tri_b, _, _ = make_case(V, R, d, L, "block", nb, E, seed=32)
    results = []
    for prefetch in (False, True):
        eng = native.Engine(V, R, d, L, "block", nb, max_edges=E)
        try:
            eng.set_params(params)
            ga, gb, dc = eng.to_device(tri_a), eng.to_device(tri_b), eng.to_device(dcodes)
            graphs = [ga, gb]
            out = []
            for i in range(5):
                eng.step_device(graphs[i % 2], E, dc, train=True, seed=100 + i)
                if prefetch:
                    eng.prefetch_graph_device(graphs[(i + 1) % 2], E)
                out.append((eng.codes(), eng.get_grad("W_f1"), eng.get_grad("W_emb")))
            results.append(out)
            for b in (ga, gb, dc):
                b.free()
        finally:
            eng.close()
    for (c0, w0, e0), (c1, w1, e1) in zip(*results):
        np.testing.assert_array_equal(c0, c1)
        np.testing.assert_array_equal(w0, w1)
        np.testing.assert_array_equal(e0, e1)
    # and the values are right: last step (i = 4 -> graph a, seed 104) against the oracle
    eng = native.Engine(V, R, d, L, "block", nb, max_edges=E)
    try:
        eng.set_params(params)
        eng.set_graph(tri_a)
        eng.forward(train=True, seed=104)
        masks = [eng.dropout_mask(l) for l in range(1, L + 1)]
    finally:
        eng.close()
    oacts, ograds = oracle.encoder_step(params, tri_a, V, L, "block", dcodes, dropout_masks=masks)
    assert float(np.abs(results[1][-1][0] - oacts[-1]).max()) <= FWD_ATOL
    assert_close(results[1][-1][1], ograds["W_f1"], name="W_f1 after prefetch")


def test_param_round_trip_and_layout(native):
    V, R, d, L, nb = 30, 7, 20, 2, 4
    params, _, _, _ = make_case(V, R, d, L, "block", nb, 5, seed=3)
    eng = native.Engine(V, R, d, L, "block", nb, max_edges=8)
    try:
        assert eng.param_names == oracle.weight_names("block", L)     # = Model.get_weights() order
        eng.set_params(params)
        for n in eng.param_names:
            np.testing.assert_array_equal(eng.get_param(n), params[n])
    finally:
        eng.close()


def test_errors_are_loud(native):
    eng = native.Engine(10, 3, 8, 1, "block", 2, max_edges=4)
    try:
        with pytest.raises(native.RgcnError):
            eng.forward()                                   # no graph yet
        with pytest.raises(native.RgcnError):
            eng.set_graph(np.array([[0, 0, 10]], np.int32))  # vertex id out of range
        with pytest.raises(native.RgcnError):
            eng.set_graph(np.array([[0, 3, 1]], np.int32))   # relation id out of range
        with pytest.raises(native.RgcnError):
            eng.set_graph(np.zeros((5, 3), np.int32))        # more than max_edges
        eng.set_graph(np.array([[0, 1, 2]], np.int32))
        with pytest.raises(native.RgcnError):
            eng.backward(np.zeros((10, 8), np.float32))      # backward before forward
        # a bad id in a device-resident graph is flagged at the next sync
        bad = eng.to_device(np.array([[0, 1, 99]], np.int32))
        eng.set_graph_device(bad, 1)
        with pytest.raises(native.RgcnError):
            eng.sync()
        bad.free()
    finally:
        eng.close()
    with pytest.raises(native.RgcnError):
        native.Engine(10, 3, 10, 1, "block", 3, max_edges=4)   # d % nb != 0


# ------------------------------------------------------------------ golden fixtures (reference-free)
@pytest.fixture(scope="module")
def expected():
    with np.load(os.path.join(helpers.GOLDEN_DIR, "expected.npz")) as z:
        return {k: z[k] for k in z.files}


def probes_for(expected, case, norm):
    pre = "%s/%s/" % (case, norm)
    names = sorted({k[len(pre):].rsplit("/", 1)[0] for k in expected if k.startswith(pre)})
    return {n: {f: expected[pre + n + "/" + f] for f in ("l2", "sum", "idx", "val")} for n in names}


@pytest.mark.parametrize("case", ["toy_block_small", "toy_block_L2", "fb237_block_L2", "toy_basis_L1",
                                  "fb237_basis_B2_L2"])
@pytest.mark.parametrize("norm", ["intended", "tf_as_executed"])
def test_golden(native, expected, case, norm):
    c = helpers.golden_inputs(case)
    acts, grads = run_engine(native, c["V"], c["R"], c["d"], c["L"], c["kind"], c["nb"], c["params"],
                             c["triples"], c["masks"], c["dcodes"], norm=norm)
    pr = probes_for(expected, case, norm)
    codes = acts[-1].ravel()
    err = float(np.abs(codes[pr["codes"]["idx"]] - pr["codes"]["val"]).max())
    assert err <= FWD_ATOL, "codes: %.3e" % err
    helpers.check_probe(acts[0], pr["H0"], name="H0")
    try:
        for k, v in grads.items():
            if k != "W_relation":
                helpers.check_probe(v, pr["grad_" + k], name=k)
    except AssertionError as direct:
        gate_aware_gradient_check(c, acts, grads, norm, direct)


def test_fb237_minibatch_full_parity(native):
    """BASELINE config 2 at full size: FB15k-237 gcn_block, real-structure 15,000-edge minibatch."""
    c = helpers.golden_inputs("fb237_block_L2")
    oacts, ograds = oracle.encoder_step(c["params"], c["triples"], c["V"], c["L"], "block", c["dcodes"],
                                        keep_prob=0.8, dropout_masks=c["masks"])
    acts, grads = run_engine(native, c["V"], c["R"], c["d"], c["L"], "block", c["nb"], c["params"],
                             c["triples"], c["masks"], c["dcodes"])
    try:
        compare(acts, grads, oacts, ograds, tag="fb237")
    except AssertionError as direct:
        if "grad" not in str(direct):
            raise
        gate_aware_gradient_check(c, acts, grads, "intended", direct)


def test_fb237_basis_b5_minibatch_full_parity(native):
    """settings/gcn_basis.exp:5's own basis count at full size (BASELINE config 3 with B = 5; gcn_basis.py:60-68): the
    real-structure 15,000-edge FB15k-237 minibatch, d = 500, two layers, train mode with injected masks -- every
    activation within 1e-4, every gradient (W_f / W_b [500,5,500], C_f / C_b, W_self, W_emb, b_emb) against the oracle."""
    c = helpers.golden_inputs("fb237_basis_B5_L2")
    oacts, ograds = oracle.encoder_step(c["params"], c["triples"], c["V"], c["L"], "basis", c["dcodes"],
                                        keep_prob=0.8, dropout_masks=c["masks"])
    acts, grads = run_engine(native, c["V"], c["R"], c["d"], c["L"], "basis", c["nb"], c["params"],
                             c["triples"], c["masks"], c["dcodes"])
    assert set(grads) - {"W_relation"} == set(ograds) - {"W_relation"}
    try:
        compare(acts, grads, oacts, ograds, tag="fb237-basis-b5")
    except AssertionError as direct:
        if "grad" not in str(direct):
            raise
        gate_aware_gradient_check(c, acts, grads, "intended", direct)


def test_full_graph_inference_shape(native):
    """Full-graph scoring shape (code/model.py:59-81): all 38,001 real FB15k-237 valid+test triples, test mode."""
    c = helpers.golden_inputs("fb237_block_L2")
    triples = helpers.load_graph("fb237_valid_test")
    oacts = oracle.encoder_forward(c["params"], triples, c["V"], c["L"], "block", mode="test")
    eng = native.Engine(c["V"], c["R"], c["d"], c["L"], "block", c["nb"], max_edges=len(triples))
    try:
        eng.set_params(c["params"])
        eng.set_graph(triples)
        eng.forward(train=False)
        codes = eng.codes()
    finally:
        eng.close()
    assert float(np.abs(codes - oacts[-1]).max()) <= FWD_ATOL


def _shape_case_full_parity(native, V, R, d, L, nb, graph, seed, tag):
    """Forward (train mode, injected masks) and EVERY gradient of a block-kind case against the oracle; when an
    isolated relu gate within rounding of zero trips the direct gradient comparison, the gate-consistent check."""
    triples = _bench_module().load_graph(graph) if ":" in graph else helpers.load_graph(graph)
    rng = np.random.RandomState(seed)
    params = oracle.init_params(V, R, d, L, "block", nb, rng=rng)
    params["b_emb"] = (rng.randn(d) * 0.01).astype(np.float32)
    masks = [(rng.rand(V, d) < 0.8).astype(np.uint8) for _ in range(L)]
    dcodes = (rng.randn(V, d) * 0.01).astype(np.float32)
    oacts, ograds = oracle.encoder_step(params, triples, V, L, "block", dcodes, keep_prob=0.8, dropout_masks=masks)
    acts, grads = run_engine(native, V, R, d, L, "block", nb, params, triples, masks, dcodes)
    assert set(grads) - {"W_relation"} == set(ograds) - {"W_relation"}
    try:
        compare(acts, grads, oacts, ograds, tag=tag)
    except AssertionError as direct:
        if "grad" not in str(direct):
            raise
        c = dict(V=V, R=R, d=d, L=L, kind="block", nb=nb, params=params, masks=masks, dcodes=dcodes, triples=triples)
        gate_aware_gradient_check(c, acts, grads, "intended", direct)


def test_wn18_shape_full_parity(native):
    """BASELINE config 4's entity / relation space on one GPU: WN18 (V 40,943, R 18; top-3 relations hold 70 %
    of the edges), the 10,000 real valid+test triples as graph, d = 500, block kind, train mode with injected
    masks: every activation and every gradient (W_emb, b_emb, both layers' W_f / W_b / W_self)."""
    _shape_case_full_parity(native, 40943, 18, 500, 2, 100, "wn18_valid_test", 11, "wn18")


def test_wn18_minibatch_full_parity(native):
    """BASELINE config 4 at SURVEY 8d's minibatch size -- bench.py's `wn18_block` workload itself: a uniform 15,000-edge
    minibatch of the 141,442-edge WN18 training-graph shape: every activation and every gradient against the oracle."""
    _shape_case_full_parity(native, 40943, 18, 500, 2, 100, "sample:wn18_valid_test:141442:15000", 13, "wn18-15k")


def test_fb15k_shape_full_parity(native):
    """BASELINE config 5's entity / relation space on one GPU: FB15k (V 14,951, R 1,345 -> 27.9 MB of block
    weights per layer, 694 non-empty relations in the graph, most with a handful of edges: the many-small-relations
    regime of the relation-major kernels and their slab reduce), 15,000 real valid triples as graph, d = 500, block
    kind, train mode with injected masks: every activation and every gradient, layer-1 and W_emb included."""
    _shape_case_full_parity(native, 14951, 1345, 500, 2, 100, "fb15k_minibatch", 12, "fb15k")


# ------------------------------------------------------------------ float64 tie-break
F64_CASES = {
    "block_small": lambda: _made_case(300, 11, 40, 2, "block", 8, 1500, 3),
    "basis_small": lambda: _made_case(300, 11, 40, 2, "basis", 3, 1500, 4),
    "block_d500": lambda: _made_case(257, 6, 500, 2, "block", 100, 4000, 5),
    "fb237_block_L2": lambda: helpers.golden_inputs("fb237_block_L2"),
    "fb237_basis_B2_L2": lambda: helpers.golden_inputs("fb237_basis_B2_L2"),
    "fb237_basis_B5_L2": lambda: helpers.golden_inputs("fb237_basis_B5_L2"),
}


def _made_case(V, R, d, L, kind, nb, E, seed):
    params, triples, masks, dcodes = make_case(V, R, d, L, kind, nb, E, seed=seed)
    return dict(V=V, R=R, d=d, L=L, kind=kind, nb=nb, params=params, masks=masks, dcodes=dcodes, triples=triples)


@pytest.mark.parametrize("name,gemm_mode", [(n, 6) for n in sorted(F64_CASES)] +
                         [(n, 0) for n in sorted(F64_CASES) if n != "fb237_block_L2"])
def test_float64_tie_break(native, name, gemm_mode):
    """The bulk tolerance of the fp32-vs-fp32 comparisons (2e-4 of scale, 1 % of entries excused, spikes to 5e-3) is set
    by relu gates that two correct fp32 forward passes may resolve differently.  float64 settles it:
      forward   |engine - float64 forward| <= 2e-5 of each layer's scale (north_star: 1e-4 absolute),
      backward  against the float64 reverse mode of the engine's OWN forward (helpers.float64_grads_at: same gates by
                construction): every entry of every gradient within 5e-6 of the tensor's scale, l2 error <= 2e-6 --
                no excused fraction, no spike allowance -- and not worse than 4x the fp32 oracle's own distance from
                float64 (the oracle's reverse mode evaluated at the same activations).
    Both GEMM arithmetics: 6 = split-bf16 (default, also at BASELINE's full sizes), 0 = fp32 MFMA.  (This test found
    the bf16 MFMA's accumulation bias: before the accumulator sign groups of gemm_bf16x3.hip the basis-coefficient
    gradients of fb237_basis_B2_L2 sat at 3.7e-6 / 5.8e-6 in mode 6 against 4e-7 in mode 0 and for the fp32 oracle.)"""
    c = F64_CASES[name]()
    eng = native.Engine(c["V"], c["R"], c["d"], c["L"], c["kind"], c["nb"], max_edges=len(c["triples"]))
    try:
        eng.set_gemm_mode(gemm_mode)
        eng.set_params(c["params"])
        eng.set_graph(c["triples"])
        eng.forward(train=True, masks=c["masks"])
        acts = [eng.activation(l) for l in range(c["L"] + 1)]
        eng.backward(c["dcodes"])
        grads = eng.get_grads()
    finally:
        eng.close()
    acts64 = helpers.float64_forward(c)
    for l, (a, b) in enumerate(zip(acts, acts64)):
        worst, l2 = helpers.error_against(b, a)
        assert worst <= 2e-5 and l2 <= 5e-6, "%s H%d against float64: max %.2e l2 %.2e" % (name, l, worst, l2)
    g64 = helpers.float64_grads_at(c, acts)
    g32 = oracle.encoder_backward(c["params"], c["triples"], c["V"], c["L"], c["kind"], acts, c["dcodes"],
                                  mode="train", keep_prob=0.8, dropout_masks=c["masks"])
    report, bad = [], []
    for k in sorted(g64):
        if k == "W_relation" or not np.abs(g64[k]).max() > 0:     # biases of the GCN layers: unconnected (H2)
            continue
        worst, l2 = helpers.error_against(g64[k], grads[k])
        oworst, ol2 = helpers.error_against(g64[k], g32[k])
        report.append("%s %.1e/%.1e (oracle %.1e/%.1e)" % (k, worst, l2, oworst, ol2))
        if not (worst <= 5e-6 and l2 <= 2e-6):
            bad.append("%s grad %s against float64: max %.2e l2 %.2e" % (name, k, worst, l2))
        if not l2 <= 4 * ol2 + 1e-7:
            bad.append("%s grad %s: l2 %.2e, the fp32 oracle's %.2e" % (name, k, l2, ol2))
    print(name, "gemm_mode", gemm_mode, "; ".join(report))
    assert not bad, bad


# ------------------------------------------------------------------ relation sharding on one GPU
@pytest.mark.parametrize("world", [2, 4])
def test_relation_sharding_with_host_exchange(native, world, kind="block", nb=4, fusion=1, partials=None):
    """world ranks as separate contexts on ONE device; the all-reduce is done by the test through
    read_buffer/write_buffer (the collective's test double).  Result must equal the unsharded run.
    partials (a list): every rank's PARTIAL pre-activation / gradient buffer of every layer is appended to it."""
    V, R, d, L, nb, E = 90, 10, 20, 2, 4, 600
    params, triples, masks, dcodes = make_case(V, R, d, L, kind, nb, E, seed=12)
    oacts, ograds = oracle.encoder_step(params, triples, V, L, kind, dcodes, dropout_masks=masks)
    from relationprediction_amd.sharding import lpt_partition
    owner = lpt_partition(np.bincount(triples[:, 1], minlength=R), world)
    engs = [native.Engine(V, R, d, L, kind, nb, max_edges=E, rank=r, world=world) for r in range(world)]
    try:
        for e in engs:
            e.set_fusion(fusion)
            e.set_params(params)
            e.set_relation_owner(owner)
            e.set_graph(triples)
            e.forward_begin(train=True, masks=masks)
        for l in range(1, L + 1):
            for e in engs:
                e.forward_layer_partial(l)
            parts = [e.read_buffer(native.BUF_EXCHANGE) for e in engs]
            if partials is not None:
                partials.extend(parts)
            total = sum(parts)
            for e in engs:
                e.write_buffer(native.BUF_EXCHANGE, total)
                e.forward_layer_finish(l)
        for e in engs:
            assert float(np.abs(e.codes() - oacts[-1]).max()) <= FWD_ATOL
        dbufs = [e.to_device(dcodes) for e in engs]
        for e, b in zip(engs, dbufs):
            e.backward_begin(b)
        for l in range(L, 0, -1):
            for e in engs:
                e.backward_layer_partial(l)
            parts = [e.read_buffer(native.BUF_EXCHANGE) for e in engs]
            if partials is not None:
                partials.extend(parts)
            tot = sum(parts)
            totw = sum(e.read_buffer(native.BUF_DSELF_EXCHANGE) for e in engs)
            for e in engs:
                e.write_buffer(native.BUF_EXCHANGE, tot)
                e.write_buffer(native.BUF_DSELF_EXCHANGE, totw)
                e.backward_layer_finish(l)
        for e in engs:
            e.backward_end()
        for r, e in enumerate(engs):
            g = e.get_grads()
            if partials is not None:
                partials.append(g)
            for name in ["W_emb", "b_emb", "W_self1", "W_self2"]:
                assert_close(g[name], ograds[name], rel=2e-4, name="rank%d %s" % (r, name))
            per_rel = ["W_f1", "W_b1", "W_f2", "W_b2"] if kind == "block" else ["C_f1", "C_b1", "C_f2", "C_b2"]
            for name in per_rel:       # only the owner holds a relation's gradient
                mine = owner == r
                assert_close(g[name][mine], ograds[name][mine], rel=2e-4, name="rank%d %s" % (r, name))
                assert not g[name][~mine].any()
        if kind == "basis":            # replicated basis tensors: the gradient is the sum over shards
            for name in ["W_f1", "W_b1", "W_f2", "W_b2"]:
                tot = sum(e.get_grad(name) for e in engs)
                assert_close(tot, ograds[name], rel=2e-4, name="sum over ranks " + name)
        for b in dbufs:
            b.free()
    finally:
        for e in engs:
            e.close()


def test_backward_driven_by_layers_can_be_read_and_abandoned_between_layers(native):
    """One GPU, the phase API, minibatch schedule: the side-stream kernels of layer 2 (dW_self, the relation-weight
    gradients) are joined into the main stream at the END OF LAYER 1 only (rgcn_schedule.hip, bwd_layer_partial).  A caller that
    reads layer 2's gradients between the layers gets the finished values (the getters wait for every stream), and a
    backward pass abandoned there does not race what comes next -- a forward pass, ANOTHER backward pass (bwd_begin rewrites
    the dS buffer the abandoned side kernels read), a new graph (its build rewrites their message lists): each of those
    joins what was left (join_abandoned_side_work)."""
    V, R, d, L, nb, E = 300, 12, 20, 2, 4, 2000
    params, triples, masks, dcodes = make_case(V, R, d, L, "block", nb, E, seed=21)
    oacts, ograds = oracle.encoder_step(params, triples, V, L, "block", dcodes, dropout_masks=masks)
    e = native.Engine(V, R, d, L, "block", nb, max_edges=E)
    try:
        e.set_params(params)
        e.set_graph(triples)
        buf = e.to_device(dcodes)

        def forward():
            e.forward_begin(train=True, masks=masks)
            for l in range(1, L + 1):
                e.forward_layer_partial(l)
                e.forward_layer_finish(l)
            assert float(np.abs(e.codes() - oacts[-1]).max()) <= FWD_ATOL

        for attempt in range(3):
            forward()
            e.backward_begin(buf)
            e.backward_layer_partial(2)
            e.backward_layer_finish(2)
            for name in ["W_self2", "W_f2", "W_b2"]:
                assert_close(e.get_grad(name), ograds[name], rel=2e-4, name="between the layers: " + name)
            # ... and the pass is abandoned here: the next attempt starts with a forward pass
        # abandoned again, then a second backward pass straight away (no forward pass in between) ...
        forward()
        e.backward_begin(buf)
        e.backward_layer_partial(2)
        e.backward_layer_finish(2)
        e.backward_begin(buf)
        for l in range(L, 0, -1):
            e.backward_layer_partial(l)
            e.backward_layer_finish(l)
        e.backward_end()
        g = e.get_grads()
        for name in ["W_emb", "W_self1", "W_self2", "W_f1", "W_b2"]:
            assert_close(g[name], ograds[name], rel=2e-4, name="second backward pass: " + name)
        # ... and abandoned before a new graph is set: another graph first, then the old one again
        forward()
        e.backward_begin(buf)
        e.backward_layer_partial(2)
        e.backward_layer_finish(2)
        e.set_graph(np.ascontiguousarray(triples[::-1][: E // 2]))
        e.set_graph(triples)
        forward()
        e.backward_begin(buf)
        for l in range(L, 0, -1):
            e.backward_layer_partial(l)
            e.backward_layer_finish(l)
        e.backward_end()
        g = e.get_grads()
        for name in ["W_emb", "b_emb", "W_self1", "W_self2", "W_f1", "W_b1", "W_f2", "W_b2"]:
            assert_close(g[name], ograds[name], rel=2e-4, name=name)
        buf.free()
    finally:
        e.close()


def test_relation_sharding_basis(native):
    test_relation_sharding_with_host_exchange(native, 2, kind="basis", nb=3)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_single_pass_equals_the_sharded_two_kernel_form(native, world):
    """Relation-sharded contexts run the destination-major single pass (rgcn_set_fusion 1) by default: every rank's
    partial pre-activations, partial row gradients and final gradients are BITWISE those of the two-kernel form
    (rgcn_set_fusion 0) on the same shard, layer by layer."""
    a, b = [], []
    test_relation_sharding_with_host_exchange(native, world, fusion=1, partials=a)
    test_relation_sharding_with_host_exchange(native, world, fusion=0, partials=b)
    assert len(a) == len(b) and len(a) == 2 * 2 * world + world
    for x, y in zip(a, b):
        if isinstance(x, dict):
            for k in x:
                np.testing.assert_array_equal(x[k], y[k], err_msg=k)
        else:
            np.testing.assert_array_equal(x, y)


def test_rccl_single_rank_communicator(native):
    """world == 1 communicator: proves librccl loads, the id/init/all-reduce path runs on the
    context's stream and leaves data unchanged (sum over one rank)."""
    eng = native.Engine(32, 3, 8, 1, "block", 2, max_edges=4)
    try:
        uid = native.Engine.comm_unique_id()
        eng.comm_init(uid)
        x = np.arange(1000, dtype=np.float32)
        buf = eng.to_device(x)
        eng.comm_allreduce_sum(buf, x.size)
        np.testing.assert_array_equal(buf.download(np.float32, x.shape), x)
        buf.free()
    finally:
        eng.close()


# ------------------------------------------------------------------ fused layer kernels == the two-kernel form
@pytest.mark.parametrize("V,R,d,nb,E,hubs", [(300, 12, 20, 4, 2500, 3), (97, 5, 8, 2, 400, 1), (64, 3, 16, 4, 0, 0),
                                             (2000, 30, 500, 100, 6000, 4), (1100, 7, 24, 3, 9000, 2), (600, 4, 9, 9, 3000, 1),
                                             (400, 6, 170, 170, 3000, 2), (350, 5, 300, 300, 2500, 1)])
@pytest.mark.parametrize("gen_dropout", [False, True])
def test_single_pass_layer_equals_the_two_kernel_form(native, V, R, d, nb, E, hubs, gen_dropout):
    """rgcn_set_fusion 1 (default), the destination-major banded single pass (block_rows.hip: gather + sd x sd products +
    segmented row sums + self-loop term + dropout + relu in ONE kernel per layer and direction, no message buffer; one
    column band per XCD, a lane group per (row, band), long rows by one wavefront each; group widths 8 / 16 / 32 / 64
    lanes at nb <= 64 / 100 / 170 / 300) against form 0 (message kernel + k_combine) -- same products, same summation
    order: every activation and every gradient BITWISE equal, with hub rows (more than 32 slots: eight interleaved
    lanes), rows spanning several slot tiles, injected and generated dropout, an empty graph, sd in {1, 3, 4, 5, 8}"""
    params, triples, masks, dcodes = make_case(V, R, d, 2, "block", nb, E, seed=V + E)
    if hubs:
        rng = np.random.RandomState(1)
        for h in range(hubs):                       # a few vertices with hundreds (or thousands) of incident edges
            idx = rng.choice(E, size=min(E // 4, 1500 if V == 1100 else 300), replace=False)
            triples[idx, 2 if h % 2 == 0 else 0] = h
    out = []
    for fuse in (1, 0):
        eng = native.Engine(V, R, d, 2, "block", nb, keep_prob=0.8, max_edges=max(E, 1))
        try:
            eng.set_fusion(fuse)
            eng.set_params(params)
            eng.set_graph(triples)
            eng.forward(train=True, seed=11, masks=None if gen_dropout else masks)
            acts = [eng.activation(l) for l in range(3)]
            eng.backward(dcodes)
            out.append((acts, eng.get_grads()))
        finally:
            eng.close()
    (fa, fg), (ua, ug) = out
    for l in range(3):
        np.testing.assert_array_equal(fa[l], ua[l], err_msg="H%d" % l)
    for k in fg:
        if k == "b_emb":
            # the single pass sums the columns of dL/dH0 inside the kernel that writes it (per workgroup, then
            # k_colsum_final's fixed order): the same terms in another, equally fixed, order
            scale = np.abs(ug["W_emb"]).sum(axis=0).max() + 1e-30
            assert np.abs(fg[k] - ug[k]).max() <= 2e-6 * scale, k
            continue
        np.testing.assert_array_equal(fg[k], ug[k], err_msg=k)
    assert np.isfinite(fa[2]).all() and (E == 0 or np.abs(fg["W_f1"]).max() > 0)


def test_workgroups_of_a_band_share_an_xcd(native):
    """k_block_rows (and the decoder's line kernel) hand column band x to the workgroups with blockIdx % 8 == x and count
    on those sharing one XCD's L2 (0.6 MB of weights + 3.6 MB of operand band per XCD).  HIP does not promise the
    round-robin placement -- results never depend on it, speed does -- so the suite re-checks it where it runs: in a
    plain launch every residue class of blockIdx % 8 sits on ONE XCD and the eight classes on eight different ones."""
    with native.Engine(16, 2, 8, 1, "block", 2, max_edges=4, devtools=True) as eng:
        probe = eng.debug_xcd_map(4096)
        if len(set(probe.tolist())) < 8:       # a partitioned device (fewer XCDs per agent): the band affinity does not apply
            pytest.skip("this device exposes %d XCDs to a launch" % len(set(probe.tolist())))
        for n in (8, 64, 2048, 3697):
            x = eng.debug_xcd_map(n)
            assert x.min() >= 0 and x.max() <= 7
            for r in range(min(8, n)):
                assert len(set(x[r::8].tolist())) == 1, (n, r, sorted(set(x[r::8].tolist())))
            if n >= 8:
                assert len(set(x[:8].tolist())) == 8, x[:8]


def test_bias_gradient_from_the_row_kernel_is_the_same_in_every_engine(native):
    """db_emb comes from column partials the bottom layer's row-gradient kernel leaves behind (block_rows.hip): every
    workgroup sums the rows IT writes, so which workgroup takes which row must not depend on anything that differs from
    run to run -- the long rows are taken by vertex id from the tail of row_order, not in the order their threads
    registered them (an atomic counter).  Four engines on a graph with dozens of long rows: bit-identical gradients; and
    db_emb equals the column sums of dW_emb (float64) to 2e-6 of scale."""
    V, R, d, nb, E = 3000, 11, 500, 100, 30000
    params, triples, masks, dcodes = make_case(V, R, d, 2, "block", nb, E, seed=9)
    rng = np.random.RandomState(3)
    for h in range(40):                             # forty rows with 100 .. 400 slots
        idx = rng.choice(E, size=rng.randint(100, 400), replace=False)
        triples[idx, 2 if h % 2 == 0 else 0] = 7 * h
    outs = []
    for rep in range(4):
        with native.Engine(V, R, d, 2, "block", nb, keep_prob=0.8, max_edges=E) as eng:
            eng.set_params(params)
            eng.set_graph(triples)
            eng.forward(train=True, seed=2)
            eng.backward(dcodes)
            outs.append(eng.get_grads())
    for g in outs[1:]:
        for k in outs[0]:
            np.testing.assert_array_equal(g[k], outs[0][k], err_msg=k)
    want = outs[0]["W_emb"].astype(np.float64).sum(axis=0)
    scale = np.abs(outs[0]["W_emb"]).astype(np.float64).sum(axis=0).max()
    assert np.abs(outs[0]["b_emb"] - want).max() <= 2e-6 * scale


def test_single_pass_block_layer_at_full_graph_scale(native):
    """the single pass on a graph with giant rows (more than 65,536 messages, a 6,000-slot hub: the piece-by-piece order
    of the giant-row cut) still equals the two-kernel form bitwise in the forward pass and the gradients."""
    V, R, d, nb, E = 3000, 20, 20, 4, 40000
    params, triples, masks, dcodes = make_case(V, R, d, 2, "block", nb, E, seed=5)
    triples[np.random.RandomState(2).choice(E, 6000, replace=False), 2] = 7
    out = []
    for fuse in (1, 0):
        with native.Engine(V, R, d, 2, "block", nb, keep_prob=0.8, max_edges=E) as eng:
            eng.set_fusion(fuse)
            eng.set_params(params)
            eng.set_graph(triples)
            eng.forward(train=True, seed=3)
            codes = eng.codes()
            eng.backward(dcodes)
            out.append((codes, eng.get_grads()))
    np.testing.assert_array_equal(out[0][0], out[1][0])
    for k in out[0][1]:
        if k == "b_emb":       # column sums from the row-gradient kernel (giant rows included)
            scale = np.abs(out[1][1]["W_emb"]).sum(axis=0).max() + 1e-30
            assert np.abs(out[0][1][k] - out[1][1][k]).max() <= 2e-6 * scale
            got = out[0][1]["W_emb"].astype(np.float64).sum(axis=0)
            assert np.abs(out[0][1][k] - got).max() <= 2e-6 * scale
            continue
        np.testing.assert_array_equal(out[0][1][k], out[1][1][k], err_msg=k)


def test_training_graph_scale_properties(native):
    """BASELINE's largest graph on one GPU -- the 272,115-edge FB15k-237 training-graph shape (hub rows of more than ten
    thousand messages: the giant-row path) at d = 500, 100 blocks -- where the numpy oracle's [E, nb, sd, sd] gathers do
    not fit: size-independent properties instead.
    (1) A checksum of checksums, one layer, test mode: every message lands at exactly one vertex, so the column sums of
        (codes - self-loop product) equal  sum_r T_f[r] (sum_{e in r} n_f[e] H[s_e]) + sum_r T_b[r] (sum_{e in r} n_b[e] H[o_e]),
        computed in float64 with one sparse product per direction.
    (2) The backward pass is linear in the upstream gradient (two layers, train mode, the same dropout seed): every
        gradient of 0.7 dY1 - 1.3 dY2 equals 0.7 grad(dY1) - 1.3 grad(dY2)."""
    import importlib.util
    import os
    import scipy.sparse as sp
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_graph2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    triples = bench.load_graph("synth:fb237_valid_test:272115")
    V, R, d, nb, E = 14541, 237, 500, 100, len(triples)
    sd = d // nb
    s, r, o = oracle.split_graph(triples)
    # ---- (1)
    rng = np.random.RandomState(3)
    params = oracle.init_params(V, R, d, 1, "block", nb, rng=rng)
    params["b_emb"] = (rng.randn(d) * 0.01).astype(np.float32)
    eng = native.Engine(V, R, d, 1, "block", nb, max_edges=E)
    try:
        eng.set_params(params)
        eng.set_graph(triples)
        eng.forward(train=False)
        codes = eng.codes().astype(np.float64)
        selfp = eng.read_buffer(native.BUF_SELF).astype(np.float64)
    finally:
        eng.close()
    H0 = oracle.affine_onehot_forward(params["W_emb"], params["b_emb"]).astype(np.float64)
    n_f = oracle.incidence_values(o, V, "intended").astype(np.float64)
    n_b = oracle.incidence_values(s, V, "intended").astype(np.float64)
    S_f = sp.coo_matrix((n_f, (r, s)), shape=(R, V)).tocsr() @ H0            # [R, d]
    S_b = sp.coo_matrix((n_b, (r, o)), shape=(R, V)).tocsr() @ H0
    want = (np.einsum("rbij,rbj->bi", params["W_f1"].astype(np.float64), S_f.reshape(R, nb, sd)) +
            np.einsum("rbij,rbj->bi", params["W_b1"].astype(np.float64), S_b.reshape(R, nb, sd))).reshape(d)
    msgs = codes - selfp
    got, scale = msgs.sum(0), np.abs(msgs).sum(0)
    assert np.isfinite(codes).all() and float(np.abs(got - want).max() / scale.max()) <= 2e-6, \
        float(np.abs(got - want).max() / scale.max())
    # ---- (2)
    params = oracle.init_params(V, R, d, 2, "block", nb, rng=rng)
    dy1 = (rng.randn(V, d) * 0.01).astype(np.float32)
    dy2 = (rng.randn(V, d) * 0.01).astype(np.float32)
    eng = native.Engine(V, R, d, 2, "block", nb, keep_prob=0.8, max_edges=E)
    try:
        eng.set_params(params)
        tri_dev = eng.to_device(triples)
        grads = []
        for dy in (dy1, dy2, (0.7 * dy1 - 1.3 * dy2).astype(np.float32)):
            dc = eng.to_device(dy)
            eng.step_device(tri_dev, E, dc, train=True, seed=21)
            grads.append({k: v.astype(np.float64) for k, v in eng.get_grads().items() if v is not None})
            dc.free()
        tri_dev.free()
    finally:
        eng.close()
    for k in grads[0]:
        if k == "W_relation" or not grads[0][k].size:
            continue
        lin = 0.7 * grads[0][k] - 1.3 * grads[1][k]
        sc = max(float(np.abs(grads[0][k]).max()), float(np.abs(grads[1][k]).max()), 1e-30)
        assert float(np.abs(grads[2][k] - lin).max()) <= 2e-5 * sc, (k, float(np.abs(grads[2][k] - lin).max()), sc)


def _elementwise_parity_at_scale(native, triples, V, R, kind, nb, seed, want_giant=False, gradients=True,
                                 misplaced_message_demo=False):
    """ELEMENT-WISE parity of the 2-layer d = 500 encoder on a graph too large for the numpy oracle's [E, ...] gathers,
    against the edge-chunked float64 restatements of tests/helpers.py (pinned to oracle.encoder_step on the CPU by
    tests/test_oracle_chunked.py):

      * forward in test mode (= the evaluation encode, model.py:59-81) AND in train mode (the device's own dropout draw,
        read back): every element of every layer within 1e-4 absolute of the chained float64 forward (north_star);
      * layer by layer from the ENGINE's own layer input: every element within 1e-5 of ITS error scale (the sum of the
        absolute values of its terms);
      * every gradient of the train-mode pass against the float64 reverse mode at the engine's own activations
        (no gate can differ): every entry within 5e-6 of its tensor's scale, 2e-6 in l2."""
    from helpers import error_against
    import helpers as hp
    layer64 = hp.chunked_block_layer_float64 if kind == "block" else hp.chunked_basis_layer_float64
    fwd64 = hp.chunked_block_encoder_forward_float64 if kind == "block" else hp.chunked_basis_encoder_forward_float64
    bwd64 = hp.chunked_block_encoder_backward_float64 if kind == "block" else hp.chunked_basis_encoder_backward_float64
    d, L, E = 500, 2, len(triples)
    s, r, o = oracle.split_graph(triples)
    slots = np.bincount(o, minlength=V) + np.bincount(s, minlength=V)
    assert (slots > 32).sum() > 1000                                   # the long-row path is in play
    if want_giant:
        assert slots.max() > 2048                                      # ... and the giant-row cut
    rng = np.random.RandomState(seed)
    params = oracle.init_params(V, R, d, L, kind, nb, rng=rng)
    params["b_emb"] = (rng.randn(d) * 0.01).astype(np.float32)
    dcodes = (rng.randn(V, d) * 0.01).astype(np.float32)
    with native.Engine(V, R, d, L, kind, nb, keep_prob=0.8, max_edges=E) as eng:
        eng.set_params(params)
        eng.set_graph(triples)
        eng.forward(train=False)
        acts_test = [eng.activation(l) for l in range(L + 1)]
        eng.forward(train=True, seed=5)
        acts_train = [eng.activation(l) for l in range(L + 1)]
        masks = [eng.dropout_mask(l) for l in range(1, L + 1)]
        if gradients:
            eng.backward(dcodes)
            grads = {k: v for k, v in eng.get_grads().items() if v is not None}
    assert 0.78 < float(masks[0].mean()) < 0.82

    for mode, acts in (("test", acts_test), ("train", acts_train)):
        chained = fwd64(params, triples, V, L, mode=mode, masks=masks)
        for l in range(L + 1):
            err = float(np.abs(acts[l].astype(np.float64) - chained[l]).max())
            assert err <= 1e-4, (mode, l, err)
        for l in range(1, L + 1):
            ref, sc = layer64(params, l, L, acts[l - 1], triples, V, mode=mode,
                              mask=masks[l - 1] if mode == "train" else None, with_scale=True)
            ratio = np.abs(acts[l].astype(np.float64) - ref) / (sc + 1e-30)
            assert float(ratio.max()) <= 1e-5, (mode, l, float(ratio.max()),
                                                np.unravel_index(int(ratio.argmax()), ratio.shape))

    if misplaced_message_demo:
        # the check above SEES one misplaced message of the row with the most slots: take one forward message into it
        # out of the layer-1 reference and hand it to the next vertex; the engine's (correct) output must then fail
        assert kind == "block"
        vg = int(slots.argmax())
        e = int(np.flatnonzero(o == vg)[0])
        H0 = acts_test[0].astype(np.float64)
        pre, sc = layer64(params, 1, L, H0, triples, V, mode="test", with_scale=True, return_pre=True)
        n_f = float(oracle.incidence_values(o, V, oracle.NORM_INTENDED)[e])
        m = n_f * np.einsum("bij,bj->bi", params["W_f1"][r[e]].astype(np.float64),
                            H0[s[e]].reshape(nb, d // nb)).reshape(d)
        wrong = pre.copy()
        wrong[vg] -= m
        wrong[(vg + 1) % V] += m
        got = acts_test[1].astype(np.float64)
        for row in (vg, (vg + 1) % V):
            ok = np.abs(got[row] - np.maximum(pre[row], 0.0)) / (sc[row] + 1e-30)
            bad = np.abs(got[row] - np.maximum(wrong[row], 0.0)) / (sc[row] + 1e-30)
            assert float(ok.max()) <= 1e-5 < float(bad.max()), (row, float(ok.max()), float(bad.max()))

    if gradients:
        g64 = bwd64(params, triples, V, L, acts_train, dcodes, mode="train", masks=masks)
        checked = 0
        for k, g in grads.items():
            if k not in g64 or k.startswith("b") and k != "b_emb":
                continue
            emax, el2 = error_against(g64[k], g)
            assert emax <= 5e-6 and el2 <= 2e-6, (k, emax, el2)
            checked += 1
        assert checked >= (7 if kind == "block" else 11)


def test_training_graph_elementwise_parity(native):
    """BASELINE's largest single-GPU graph, block kind: the 272,115-edge FB15k-237 training-graph shape (V = 14,541,
    100 blocks; 5,000+ rows on the long-row path, one row beyond the 2,048-slot giant-row cut), with the demonstration
    that ONE message of the giant row handed to the neighbouring vertex fails the check.  Reference:
    gcn_basis_concat.py:35-83, message_gcn.py:49-79, model.py:59-81."""
    triples = _bench_module().load_graph("synth:fb237_valid_test:272115")
    _elementwise_parity_at_scale(native, triples, 14541, 237, "block", 100, 17, want_giant=True,
                                 misplaced_message_demo=True)


def test_training_graph_elementwise_parity_basis(native):
    """The same graph through the BASIS kind (BASELINE config 3's evaluation encode and full-graph train pass, B = 2):
    the aggregate-first kernels' long-row path at 5,000+ rows and a 2,397-slot hub, against the float64 restatement of
    the reference's PER-EDGE dataflow (gcn_basis.py:39-88).  Every gradient incl. C_f / C_b of both layers."""
    triples = _bench_module().load_graph("synth:fb237_valid_test:272115")
    _elementwise_parity_at_scale(native, triples, 14541, 237, "basis", 2, 18, want_giant=True)


def test_wn18_training_graph_elementwise_parity(native):
    """BASELINE config 4 at SURVEY 8d's size on one GPU: WN18 (V 40,943, R 18), the 141,442-edge training-graph shape
    drawn from the real valid+test histograms (three relations hold 73 % of the edges: relation chunks thousands of
    messages long), block kind.  Evaluation encode, train-mode forward and every gradient."""
    triples = _bench_module().load_graph("synth:wn18_valid_test:141442")
    _elementwise_parity_at_scale(native, triples, 40943, 18, "block", 100, 19)


def test_fb15k_training_graph_evaluation_encode(native):
    """BASELINE config 5's evaluation encode on one GPU: FB15k (V 14,951, R 1,345), the 483,142-edge training-graph
    shape drawn from the real valid histograms, block kind: test-mode and train-mode forward, element-wise."""
    triples = _bench_module().load_graph("synth:fb15k_minibatch:483142")
    _elementwise_parity_at_scale(native, triples, 14951, 1345, "block", 100, 20, gradients=False)
