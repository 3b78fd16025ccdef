"""The oracle (CPU) and the HIP path (`-m gpu`) against forward passes computed by THE REFERENCE'S OWN MODEL CODE.

tests/golden/reference_model.npz is made by tests/golden/make_reference_model_fixtures.py: the reference's
model_builder / Representation / AffineTransform / ConcatGcn / BasisGcn / RelationEmbedding / BilinearDiag run as
they stand, with an eager numpy stand-in for the ~25 TensorFlow primitives they call (TF 1.4 cannot be installed).
Pinned here: initial weights bit for bit on numpy's global stream (distributions, shapes, creation order), codes in
test and train mode (the dropout masks the reference drew are injected), the training loss with its regulariser, and
the score-every-subject / score-every-object matrices -- i.e. the composition of the reference's dataflow, which
index of W is the output index, which incidence matrix multiplies which messages, where dropout and relu sit."""
import os

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = np.load(os.path.join(HERE, "golden", "reference_model.npz"))
# BASELINE configs 2 and 3 at full size: stored as fingerprints; weights and masks regenerated from the seeds
FULLS = ["fb237_block_full", "fb237_basis_b2_full"]
CASES = sorted({k.split("/")[0] for k in FIX.files} - set(FULLS))
FWD_ATOL = 1e-4            # north_star: "outputs match the reference forward pass within 1e-4 fp32"


def load(name):
    kind_id, V, R, d, nb, L, E, N, seed, mode = (int(x) for x in FIX[name + "/config"])
    kind = "block" if kind_id == 0 else "basis"
    names = oracle.weight_names(kind, L)
    weights = [FIX["%s/weight%02d" % (name, i)] for i in range(len(names))]
    assert "%s/weight%02d" % (name, len(names)) not in FIX.files              # the reference holds exactly these
    c = dict(kind=kind, V=V, R=R, d=d, nb=nb, L=L, E=E, N=N, seed=seed,
             norm="intended" if mode == 0 else "tf_as_executed",
             params=dict(zip(names, weights)), names=names,
             triples=FIX[name + "/triples"], X=FIX[name + "/X"], Y=FIX[name + "/Y"],
             masks=[FIX["%s/mask%d" % (name, l)] for l in range(1, L + 1)])
    for key in ("loss_train", "codes_train", "codes_test", "subject_scores", "object_scores"):
        c[key] = FIX[name + "/" + key]
    c["grads"] = {n: FIX["%s/grad%02d" % (name, i)] for i, n in enumerate(names)}
    c["connected"] = {n: bool(FIX["%s/grad%02d_connected" % (name, i)]) for i, n in enumerate(names)}
    return c


def assert_grad_close(got, want, what, rel=2e-4):
    scale = max(float(np.abs(want).max()), 1e-6)
    assert float(np.abs(got - want).max()) <= rel * scale + 1e-7, (what, float(np.abs(got - want).max()), scale)


@pytest.mark.parametrize("name", CASES)
def test_initial_weights_are_the_reference_draws(name):
    """same seed, same numpy global stream: the oracle's init_params (and through it the plugin chain, whose
    stream consumption tests/test_plugin_surface.py ties to init_params) reproduces the reference's variables exactly"""
    c = load(name)
    np.random.seed(c["seed"])
    mine = oracle.init_params(c["V"], c["R"], c["d"], c["L"], c["kind"], c["nb"], rng=np.random)
    for n in c["names"]:
        assert mine[n].dtype == c["params"][n].dtype == np.float32, n
        np.testing.assert_array_equal(mine[n], c["params"][n], err_msg=n)


@pytest.mark.parametrize("name", CASES)
def test_oracle_forward_and_loss_equal_the_reference_dataflow(name):
    c = load(name)
    test = oracle.encoder_forward(c["params"], c["triples"], c["V"], c["L"], c["kind"], mode="test",
                                  norm_mode=c["norm"])
    assert float(np.abs(test[-1] - c["codes_test"]).max()) <= 2e-6 * max(1.0, float(np.abs(c["codes_test"]).max()))
    train = oracle.encoder_forward(c["params"], c["triples"], c["V"], c["L"], c["kind"], mode="train", keep_prob=0.8,
                                   dropout_masks=c["masks"], norm_mode=c["norm"])
    assert float(np.abs(train[-1] - c["codes_train"]).max()) <= 2e-6 * max(1.0, float(np.abs(c["codes_train"]).max()))
    loss, _, _ = oracle.distmult_loss_and_grads(train[-1], c["params"]["W_relation"], c["X"], c["Y"], 0.01)
    assert loss == pytest.approx(float(c["loss_train"]), rel=2e-6)
    # score-everything graphs (bilinear_diag.py:46-61): row i = sigmoid energies of every entity in triple i's open slot
    codes, rel = test[-1], c["params"]["W_relation"]
    sig = lambda x: 1.0 / (1.0 + np.exp(-x.astype(np.float64)))  # noqa: E731
    subj = np.stack([sig(codes @ (rel[r] * codes[o])) for s, r, o in c["X"]])
    obj = np.stack([sig(codes @ (codes[s] * rel[r])) for s, r, o in c["X"]])
    assert np.abs(subj - c["subject_scores"]).max() <= 1e-5 and np.abs(obj - c["object_scores"]).max() <= 1e-5


@pytest.mark.parametrize("name", CASES)
def test_oracle_gradients_equal_autograd_of_the_reference_dataflow(name):
    """tf.gradients(loss, weights) (optimization/abstract.py:117-118), obtained by running the reference's model code
    on torch tensors: every weight's gradient, the decoder's included; the variables the reference's graph leaves
    unconnected are exactly the per-layer biases"""
    c = load(name)
    assert sorted(n for n, ok in c["connected"].items() if not ok) == sorted("b%d" % l for l in range(1, c["L"] + 1))
    acts = oracle.encoder_forward(c["params"], c["triples"], c["V"], c["L"], c["kind"], mode="train", keep_prob=0.8,
                                  dropout_masks=c["masks"], norm_mode=c["norm"])
    _, dcodes, d_rel = oracle.distmult_loss_and_grads(acts[-1], c["params"]["W_relation"], c["X"], c["Y"], 0.01)
    grads = oracle.encoder_backward(c["params"], c["triples"], c["V"], c["L"], c["kind"], acts, dcodes, mode="train",
                                    keep_prob=0.8, dropout_masks=c["masks"], norm_mode=c["norm"])
    grads["W_relation"] = d_rel
    for n in c["names"]:
        if c["connected"][n]:
            assert_grad_close(grads[n], c["grads"][n], n, rel=2e-5)
        else:
            assert not np.asarray(grads[n]).any(), n


def test_the_two_h1_readings_differ_and_the_fixture_covers_both():
    """the 'sorted_rows' case is the reference's graph under the suspected behaviour of
    tf.sparse_softmax on non-canonical indices; the oracle's tf_as_executed mode must follow it, the intended mode
    must NOT (otherwise the case would not discriminate)."""
    c = load("block_h1_sorted_rows")
    assert c["norm"] == "tf_as_executed"
    other = oracle.encoder_forward(c["params"], c["triples"], c["V"], c["L"], c["kind"], mode="test",
                                   norm_mode="intended")
    assert float(np.abs(other[-1] - c["codes_test"]).max()) > 1e-3


# ------------------------------------------------------------------ BASELINE configs 2 and 3 at full size
def load_full(FULL):
    import helpers
    kind_id, V, R, d, nb, L, E, N, seed, mode = (int(x) for x in FIX[FULL + "/config"])
    assert mode == 0
    kind = "block" if kind_id == 0 else "basis"
    names = oracle.weight_names(kind, L)
    assert int(FIX[FULL + "/n_weights"]) == len(names)
    np.random.seed(seed)                                         # the reference's draws (bitwise: the small cases)
    params = oracle.init_params(V, R, d, L, kind, nb, rng=np.random)
    drop = np.random.RandomState(seed)                           # the shim's dropout stream, bottom layer first
    masks = [np.floor(0.8 + drop.uniform(size=(V, d))).astype(np.uint8) for _ in range(L)]
    fp = lambda key: {f: FIX["%s/%s/%s" % (FULL, key, f)] for f in ("l2", "sum", "idx", "val")}  # noqa: E731
    return dict(V=V, R=R, d=d, nb=nb, L=L, E=E, kind=kind, names=names, params=params, masks=masks,
                triples=helpers.load_graph("fb237_minibatch"), X=FIX[FULL + "/X"], Y=FIX[FULL + "/Y"],
                loss=float(FIX[FULL + "/loss_train"]), fp=fp,
                connected=[bool(FIX["%s/grad%02d_connected" % (FULL, i)]) for i in range(len(names))])


def check_codes(codes, pr, atol):
    a = np.asarray(codes, dtype=np.float32).ravel()
    assert float(np.abs(a[pr["idx"]] - pr["val"]).max()) <= atol
    l2 = float(np.sqrt(np.sum(a.astype(np.float64) ** 2)))
    assert abs(l2 - float(pr["l2"])) <= 1e-5 * float(pr["l2"])


@pytest.mark.parametrize("full", FULLS)
def test_full_size_oracle_equals_the_reference_dataflow(full):
    """FB15k-237 gcn_block (100 blocks) and gcn_basis (B = 2) at full size (V 14,541, d 500, 2 layers, the real
    15,000-edge minibatch): the reference's own model code computed these fingerprints; the oracle must land on
    them -- weights, codes in both modes, loss, every gradient."""
    import helpers
    c = load_full(full)
    for i, n in enumerate(c["names"]):
        pr = c["fp"]("weight%02d" % i)
        flat = c["params"][n].ravel()
        assert np.array_equal(flat[pr["idx"]], pr["val"]), n                       # bitwise at the sampled positions
    test = oracle.encoder_forward(c["params"], c["triples"], c["V"], c["L"], c["kind"], mode="test")
    check_codes(test[-1], c["fp"]("codes_test"), 2e-5)
    acts = oracle.encoder_forward(c["params"], c["triples"], c["V"], c["L"], c["kind"], mode="train", keep_prob=0.8,
                                  dropout_masks=c["masks"])
    check_codes(acts[-1], c["fp"]("codes_train"), 2e-5)
    loss, dcodes, d_rel = oracle.distmult_loss_and_grads(acts[-1], c["params"]["W_relation"], c["X"], c["Y"], 0.01)
    assert loss == pytest.approx(c["loss"], rel=1e-5)
    grads = oracle.encoder_backward(c["params"], c["triples"], c["V"], c["L"], c["kind"], acts, dcodes, mode="train",
                                    keep_prob=0.8, dropout_masks=c["masks"])
    grads["W_relation"] = d_rel
    for i, n in enumerate(c["names"]):
        if c["connected"][i]:
            helpers.check_probe(grads[n], c["fp"]("grad%02d" % i), rel=1e-4, name=n)


@pytest.mark.gpu
@pytest.mark.parametrize("full", FULLS)
def test_full_size_hip_path_equals_the_reference_dataflow(native, full):
    """the same fingerprints through the C ABI: codes within 1e-4 absolute (north_star), loss, all gradients"""
    import helpers
    c = load_full(full)
    eng = native.Engine(c["V"], c["R"], c["d"], c["L"], c["kind"], c["nb"], keep_prob=0.8, max_edges=c["E"])
    bufs = []
    try:
        eng.set_params(c["params"])
        eng.set_graph(c["triples"])
        eng.forward(train=False)
        check_codes(eng.codes(), c["fp"]("codes_test"), FWD_ATOL)
        eng.forward(train=True, masks=c["masks"])
        check_codes(eng.codes(), c["fp"]("codes_train"), FWD_ATOL)
        eng.decoder_reserve(len(c["X"]))
        xd, yd = eng.to_device(np.ascontiguousarray(c["X"])), eng.to_device(np.ascontiguousarray(c["Y"]))
        bufs += [xd, yd]
        eng.decoder_loss_backward_device(xd, yd, len(c["X"]), 0.01)
        assert eng.loss() == pytest.approx(c["loss"], rel=2e-5)
        eng.backward_from_decoder()
        grads = eng.get_grads()
        for i, n in enumerate(c["names"]):
            if c["connected"][i]:
                helpers.check_probe(grads[n], c["fp"]("grad%02d" % i), name=n)    # isolated relu-gate flips tolerated
    finally:
        for b in bufs:
            b.free()
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("full", FULLS)
def test_full_size_float64_tie_break_on_the_reference_fixtures(native, full):
    """The check above excuses 2 % of the sampled gradient entries (two correct fp32 forward passes may resolve a relu
    gate differently).  Here float64 takes the excuse away from the HIP path: the oracle, switched to float64 and fed the
    ENGINE'S OWN activations (same gates by construction), gives the exact derivative of the forward pass the engine
    computed -- DistMult loss on the fixture's batch included -- and
      (1) EVERY entry of EVERY gradient of the HIP path is within 5e-6 of its tensor's scale of it (l2 error 2e-6): no
          excused fraction, no spike allowance;
      (2) the reference's own fp32 autograd values (the fixture's sampled entries) sit within the fp32 gate noise of
          that same exact derivative -- the 2 % / 5e-3 allowance now stands between the REFERENCE's fp32 arithmetic
          and float64, with the HIP path out of it."""
    import helpers
    c = load_full(full)
    eng = native.Engine(c["V"], c["R"], c["d"], c["L"], c["kind"], c["nb"], keep_prob=0.8, max_edges=c["E"])
    bufs = []
    try:
        eng.set_params(c["params"])
        eng.set_graph(c["triples"])
        eng.forward(train=True, masks=c["masks"])
        acts = [eng.activation(l) for l in range(c["L"] + 1)]
        eng.decoder_reserve(len(c["X"]))
        xd, yd = eng.to_device(np.ascontiguousarray(c["X"])), eng.to_device(np.ascontiguousarray(c["Y"]))
        bufs += [xd, yd]
        eng.decoder_loss_backward_device(xd, yd, len(c["X"]), 0.01)
        eng.backward_from_decoder()
        grads = eng.get_grads()
    finally:
        for b in bufs:
            b.free()
        eng.close()
    with helpers.oracle_float64():
        p64 = {k: np.asarray(v, dtype=np.float64) for k, v in c["params"].items()}
        a64 = [np.asarray(a, dtype=np.float64) for a in acts]
        _, dcodes64, drel64 = oracle.distmult_loss_and_grads(a64[-1], p64["W_relation"], c["X"], c["Y"].astype(np.float64), 0.01)
        g64 = oracle.encoder_backward(p64, c["triples"], c["V"], c["L"], c["kind"], a64, dcodes64, mode="train",
                                      keep_prob=0.8, dropout_masks=c["masks"])
    g64["W_relation"] = drel64
    bad, report = [], []
    for i, n in enumerate(c["names"]):
        if not c["connected"][i]:
            continue
        worst, l2 = helpers.error_against(g64[n], grads[n])
        report.append("%s %.1e/%.1e" % (n, worst, l2))
        if not (worst <= 5e-6 and l2 <= 2e-6):      # measured: <= 1e-6 / 5.2e-7
            bad.append("%s: HIP against the float64 derivative of its own forward: max %.2e l2 %.2e" % (n, worst, l2))
        helpers.check_probe(g64[n].astype(np.float32), c["fp"]("grad%02d" % i), name="reference fp32 autograd vs float64, " + n)
    print(full, "; ".join(report))
    assert not bad, bad


@pytest.fixture(scope="module")
def native():
    from relationprediction_amd import _native
    _native.load_library()
    return _native


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_path_equals_the_reference_dataflow(native, name):
    """through the C ABI: codes in both modes within 1e-4 absolute of what the reference's own model code computed,
    the training loss (decoder + regulariser), and raw ranks from the reference's score matrices"""
    c = load(name)
    eng = native.Engine(c["V"], c["R"], c["d"], c["L"], c["kind"], c["nb"], keep_prob=0.8, norm_mode=c["norm"],
                        max_edges=c["E"])
    bufs = []
    try:
        eng.set_params(c["params"])
        eng.set_graph(c["triples"])
        eng.forward(train=False)
        assert float(np.abs(eng.codes() - c["codes_test"]).max()) <= FWD_ATOL
        # ranks straight from the reference's score matrices: number of entities scoring >= the gold one
        eng.rank_reserve(len(c["X"]))
        ptr = np.arange(len(c["X"]) + 1, dtype=np.int64)
        for object_side, scores, gold_col in ((True, c["object_scores"], 2), (False, c["subject_scores"], 0)):
            gold = c["X"][:, gold_col]
            want = (scores >= scores[np.arange(len(gold)), gold][:, None]).sum(1)
            raw, filt = eng.ranks(c["X"], object_side, ptr, gold.astype(np.int32))      # filter list = the gold entity
            assert np.abs(raw - want).max() <= 1 and np.mean(raw != want) <= 0.05        # fp32 near-ties may move a rank
            assert np.array_equal(raw, filt)
        eng.forward(train=True, masks=c["masks"])
        assert float(np.abs(eng.codes() - c["codes_train"]).max()) <= FWD_ATOL
        eng.decoder_reserve(len(c["X"]))
        xd, yd = eng.to_device(np.ascontiguousarray(c["X"])), eng.to_device(np.ascontiguousarray(c["Y"]))
        bufs += [xd, yd]
        eng.decoder_loss_backward_device(xd, yd, len(c["X"]), 0.01)
        assert eng.loss() == pytest.approx(float(c["loss_train"]), rel=2e-5)
        # and the gradient of that loss w.r.t. every weight: autograd over the reference's own dataflow
        eng.backward_from_decoder()
        grads = eng.get_grads()
        for n in c["names"]:
            if c["connected"][n]:
                assert_grad_close(grads[n], c["grads"][n], n)
            elif n in grads:
                assert not grads[n].any(), n
    finally:
        for b in bufs:
            b.free()
        eng.close()
