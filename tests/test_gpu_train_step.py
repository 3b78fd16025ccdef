"""'Next' rows f1/f2 on the GPU: device DistMult decoder (loss, regulariser, gradients), clip + Adam, and the
whole train step, against the oracle's restatements."""
import numpy as np
import pytest

import oracle
import helpers
from helpers import assert_close, make_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from relationprediction_amd import _native
    _native.load_library()
    return _native


def decoder_batch(rng, triples, V, neg_rate=3, hub=None):
    pos = triples.copy()
    neg = np.tile(pos, (neg_rate, 1))
    side = rng.rand(len(neg)) < 0.5
    rnd = rng.randint(0, V, len(neg))
    neg[side, 2] = rnd[side]
    neg[~side, 0] = rnd[~side]
    X = np.concatenate([pos, neg]).astype(np.int32)
    if hub is not None:            # a hub entity with > 256 incidences exercises the long-row workgroups
        X[: len(X) // 3, 0] = hub
    Y = np.concatenate([np.ones(len(pos)), np.zeros(len(neg))]).astype(np.float32)
    return X, Y


@pytest.mark.parametrize("V,R,d,nb,E,hub", [(60, 7, 20, 4, 200, None), (120, 9, 40, 8, 600, 5),
                                            (50, 5, 9, 3, 100, None),
                                            (40, 5, 258, 86, 100, None),     # rows too wide for the fused energy + relation kernel
                                            (40, 5, 520, 104, 150, 3)])      # three register tiles per lane (T = 4)
def test_device_decoder_matches_oracle(native, V, R, d, nb, E, hub):
    L = 2
    params, triples, masks, _ = make_case(V, R, d, L, "block", nb, E, seed=E)
    rng = np.random.RandomState(1)
    X, Y = decoder_batch(rng, triples, V, hub=hub)
    eng = native.Engine(V, R, d, L, "block", nb, max_edges=E)
    try:
        eng.set_params(params)
        eng.set_graph(triples)
        eng.forward(train=True, masks=masks)
        codes = eng.codes()
        eng.decoder_reserve(len(X))
        xd, yd = eng.to_device(X), eng.to_device(Y)
        eng.decoder_loss_backward_device(xd, yd, len(X), 0.01)
        loss, dcodes, dwrel = eng.loss(), eng.dcodes(), eng.get_grad("W_relation")
        eng.backward_from_decoder()
        grads = eng.get_grads()
        xd.free(); yd.free()
    finally:
        eng.close()
    oloss, odcodes, odwrel = oracle.distmult_loss_and_grads(codes, params["W_relation"], X, Y, 0.01)
    assert abs(loss - oloss) <= 2e-5 * max(1.0, abs(oloss))
    assert_close(dcodes, odcodes, rel=2e-4, name="dcodes")
    assert_close(dwrel, odwrel, rel=2e-4, name="dW_relation")
    assert not dwrel[R:].any()
    # and the encoder backward fed from the device-resident dcodes
    acts = oracle.encoder_forward(params, triples, V, L, "block", mode="train", dropout_masks=masks)
    og = oracle.encoder_backward(params, triples, V, L, "block", acts, odcodes, mode="train", dropout_masks=masks)
    for k in og:
        assert_close(grads[k], og[k], rel=5e-4, name=k)


def _decoder_grads(native, V, R, d, nb, X, Y, lines, top_dropout=0.0, seed=3):
    """loss, dL/dcodes (and its dropped copy's consumer: the encoder's gradients) with the entity-gradient form chosen"""
    import os
    L = 1
    params, triples, _, _ = make_case(V, R, d, L, "block", nb, 4 * V, seed=seed)
    old = os.environ.get("RGCN_DEC_LINES")
    os.environ["RGCN_DEC_LINES"] = "1" if lines else "0"
    eng = native.Engine(V, R, d, L, "block", nb, max_edges=len(triples), devtools=True)   # RGCN_DEC_LINES: devtools build only
    try:
        eng.set_params(params)
        eng.set_graph(triples)
        eng.forward(train=False)
        eng.decoder_reserve(len(X))
        xd, yd = eng.to_device(X), eng.to_device(Y)
        eng.decoder_loss_backward_device(xd, yd, len(X), 0.01)
        out = eng.loss(), eng.dcodes(), eng.get_grad("W_relation")
        xd.free(); yd.free()
    finally:
        eng.close()
        if old is None:
            os.environ.pop("RGCN_DEC_LINES", None)
        else:
            os.environ["RGCN_DEC_LINES"] = old
    return out


@pytest.mark.parametrize("V,R,d,nb,n,hub", [(60, 7, 20, 4, 200, None), (120, 9, 40, 8, 3000, 5), (700, 600, 72, 9, 9000, 7),
                                            (257, 5, 520, 104, 4000, 3), (14541, 237, 500, 100, 60000, 11)])
def test_entity_gradient_forms_are_bitwise_equal(native, V, R, d, nb, n, hub):
    """k_dec_entity_lines (column bands of one cache line over a band-major copy of the codes, one band per XCD and
    pass) forms every sum of dL/dcodes in k_dec_entity_grad's order: bit-equal, rows of every length (empty, short,
    > 256 incidences = pieces), R small (LDS table) and large (global table), d with a partial last band."""
    rng = np.random.RandomState(n)
    pos = np.stack([rng.randint(0, V, n), rng.randint(0, R, n), rng.randint(0, V, n)], 1).astype(np.int32)
    pos[rng.rand(n) < 0.3, 0] = rng.randint(0, max(2, V // 10))      # a skewed subject distribution
    X, Y = decoder_batch(rng, pos, V, neg_rate=2, hub=hub)
    a = _decoder_grads(native, V, R, d, nb, X, Y, lines=False)
    b = _decoder_grads(native, V, R, d, nb, X, Y, lines=True)
    assert a[0] == b[0]
    assert np.array_equal(a[1], b[1]), float(np.abs(a[1] - b[1]).max())
    assert np.array_equal(a[2], b[2])
    assert np.abs(a[1]).max() > 0


@pytest.mark.parametrize("V,R,n,rate", [(300, 7, 900, 10), (1400, 1345, 13000, 1)])
def test_decoder_on_a_tiled_batch(native, V, R, n, rate):
    """A batch the device negative sampler tiled (NegativeSampler.transform's layout) is sorted by (relation, row mod
    batch size) -- only its first copies are sorted, the others are put directly behind them -- so that the copies of a
    triple share their row fetches in a relation chunk: another ORDER of the same sums.  Loss and gradients against the
    oracle, and against the untiled order (the same rows in a buffer the sampler did not write); R = 1,345 with n = 13,000
    is the FB15k-sized case."""
    d, L, nb, E = 20, 1, 4, 200
    params, triples, masks, _ = make_case(V, R, d, L, "block", nb, E, seed=3)
    rng = np.random.RandomState(8)
    batch = np.stack([rng.randint(0, V, n), rng.randint(0, R, n), rng.randint(0, V, n)], 1).astype(np.int32)
    N = n * (rate + 1)
    eng = native.Engine(V, R, d, L, "block", nb, max_edges=E)
    try:
        eng.set_params(params)
        eng.set_graph(triples)
        eng.forward(train=True, masks=masks)
        codes = eng.codes()
        eng.decoder_reserve(N)
        bd, xd, yd = eng.to_device(batch), eng.alloc(12 * N), eng.alloc(4 * N)
        eng.negative_sample_device(bd, n, rate, 17, xd, yd)
        X, Y = xd.download(np.int32, (N, 3)), yd.download(np.float32, (N,))
        eng.decoder_loss_backward_device(xd, yd, N, 0.01)                    # tiled order
        tiled = (eng.loss(), eng.dcodes(), eng.get_grad("W_relation"))
        x2 = eng.to_device(X)                                                # the same rows, not known to be tiled
        eng.decoder_loss_backward_device(x2, yd, N, 0.01)
        plain = (eng.loss(), eng.dcodes(), eng.get_grad("W_relation"))
        for b in (bd, xd, yd, x2):
            b.free()
    finally:
        eng.close()
    assert np.array_equal(X[:n], batch) and (Y[:n] == 1).all() and not Y[n:].any()
    oloss, odcodes, odwrel = oracle.distmult_loss_and_grads(codes, params["W_relation"], X, Y, 0.01)
    for loss, dcodes, dwrel in (tiled, plain):
        assert abs(loss - oloss) <= 2e-5 * max(1.0, abs(oloss))
        assert_close(dcodes, odcodes, rel=2e-4, name="dcodes")
        assert_close(dwrel, odwrel, rel=2e-4, name="dW_relation")
    assert np.array_equal(tiled[1], plain[1])              # the entity gradient does not depend on the relation order
    assert_close(tiled[2], plain[2], rel=2e-6, name="dW_relation tiled vs plain")


def test_tiled_batch_rewritten_behind_the_library(native):
    """The decoder orders a batch the device negative sampler tiled by its first copies alone.  Rewriting that buffer
    THROUGH the library (rgcn_copy_to_device) drops that knowledge -- the batch is sorted like any other and decoded
    correctly; rewriting it behind the library's back (here: another context's copy into the same allocation) with a
    copy whose relation differs from its first copy's is reported, not decoded wrongly."""
    V, R, d, L, nb, E, n, rate = 200, 9, 20, 1, 4, 150, 300, 3
    params, triples, masks, _ = make_case(V, R, d, L, "block", nb, E, seed=4)
    rng = np.random.RandomState(11)
    batch = np.stack([rng.randint(0, V, n), rng.randint(0, R, n), rng.randint(0, V, n)], 1).astype(np.int32)
    N = n * (rate + 1)
    eng = native.Engine(V, R, d, L, "block", nb, max_edges=E)
    other = native.Engine(V, R, d, L, "block", nb, max_edges=E)
    try:
        eng.set_params(params)
        eng.set_graph(triples)
        eng.forward(train=True, masks=masks)
        codes = eng.codes()
        eng.decoder_reserve(N)
        bd, xd, yd = eng.to_device(batch), eng.alloc(12 * N), eng.alloc(4 * N)
        eng.negative_sample_device(bd, n, rate, 3, xd, yd)
        X, Y = xd.download(np.int32, (N, 3)), yd.download(np.float32, (N,))
        X2 = X.copy()
        X2[n + 5, 1] = (X2[n + 5, 1] + 1) % R                    # a copy with another relation than its first copy
        # (a) through the library: an ordinary batch again
        eng.copy_to_device(xd, X2)
        eng.decoder_loss_backward_device(xd, yd, N, 0.01)
        loss, dcodes, dwrel = eng.loss(), eng.dcodes(), eng.get_grad("W_relation")
        oloss, odcodes, odwrel = oracle.distmult_loss_and_grads(codes, params["W_relation"], X2, Y, 0.01)
        assert abs(loss - oloss) <= 2e-5 * max(1.0, abs(oloss))
        assert_close(dcodes, odcodes, rel=2e-4, name="dcodes")
        assert_close(dwrel, odwrel, rel=2e-4, name="dW_relation")
        # (b) behind its back
        eng.negative_sample_device(bd, n, rate, 3, xd, yd)
        other.copy_to_device(xd, X2)
        other.sync()
        eng.decoder_loss_backward_device(xd, yd, N, 0.01)
        with pytest.raises(native.RgcnError) as err:
            eng.loss()
        assert "tiled batch" in str(err.value)
        for b in (bd, xd, yd):
            b.free()
    finally:
        other.close()
        eng.close()


def numpy_clip_adam(params, grads, names, lr, b1, b2, eps, max_norm, steps_state):
    gn = np.sqrt(sum(float(np.sum(grads[n].astype(np.float64) ** 2)) for n in names))
    scale = max_norm / max(gn, max_norm)
    t = steps_state["t"] = steps_state.get("t", 0) + 1
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    out = {}
    for n in names:
        g = grads[n] * np.float32(scale)
        m = steps_state.setdefault("m_" + n, np.zeros_like(g))
        v = steps_state.setdefault("v_" + n, np.zeros_like(g))
        m[...] = b1 * m + (1 - b1) * g
        v[...] = b2 * v + (1 - b2) * g * g
        out[n] = params[n] - np.float32(lr_t) * m / (np.sqrt(v) + np.float32(eps))
    return out, gn


@pytest.mark.parametrize("kind,nb", [("block", 4), ("basis", 2)])
def test_train_step_device_with_clip_and_adam(native, kind, nb):
    V, R, d, L, E = 80, 8, 20, 2, 300
    params, triples, _, _ = make_case(V, R, d, L, kind, nb, E, seed=77)
    rng = np.random.RandomState(2)
    X, Y = decoder_batch(rng, triples, V)
    names = [n for n in oracle.weight_names(kind, L) if not (n.startswith("b") and n != "b_emb")]
    eng = native.Engine(V, R, d, L, kind, nb, max_edges=E)
    try:
        eng.set_params(params)
        eng.decoder_reserve(len(X))
        eng.optimizer_config(lr=0.01, beta1=0.9, beta2=0.999, eps=1e-8, max_grad_norm=1.0)
        td, xd, yd = eng.to_device(triples), eng.to_device(X), eng.to_device(Y)
        state = {}
        cur = {k: v.copy() for k, v in params.items()}
        for step in range(3):
            eng.train_step_device(td, E, xd, yd, len(X), seed=500 + step, reg_param=0.01)
            loss = eng.loss()
            masks = [eng.dropout_mask(l) for l in range(1, L + 1)]
            grads = eng.get_grads()
            new = {n: eng.get_param(n) for n in eng.param_names}
            # replay on the oracle from the SAME pre-step weights
            acts = oracle.encoder_forward(cur, triples, V, L, kind, mode="train", dropout_masks=masks)
            oloss, odc, odw = oracle.distmult_loss_and_grads(acts[-1], cur["W_relation"], X, Y, 0.01)
            og = oracle.encoder_backward(cur, triples, V, L, kind, acts, odc, mode="train", dropout_masks=masks)
            og["W_relation"] = odw
            assert abs(loss - oloss) <= 5e-5 * max(1.0, abs(oloss)), (step, loss, oloss)
            for n in names:
                assert_close(grads[n], og[n], rel=1e-3, name="step %d grad %s" % (step, n))
            # the device update, replayed in numpy from the DEVICE gradients (isolates the optimizer)
            expect, gn = numpy_clip_adam(cur, grads, names, 0.01, 0.9, 0.999, 1e-8, 1.0, state)
            for n in names:
                assert_close(new[n], expect[n], rel=2e-5, spike=2e-4, name="step %d weight %s" % (step, n))
            for n in eng.param_names:
                if n not in names:
                    np.testing.assert_array_equal(new[n], cur[n])          # unused biases never move
            cur = new
        for b in (td, xd, yd):
            b.free()
    finally:
        eng.close()


def test_train_step_edge_shapes(native):
    """The shapes a driver can produce at its corners, one context, back to back: an empty message graph (the
    encoder reduces to its self-loops), a single decoder triple, a batch whose triples all hit one entity, the
    graph at exactly max_edges, and sizes that shrink and grow between steps.  Every step: finite loss, equal to
    the oracle's on the weights the step started from."""
    V, R, d, L, nb, Emax = 70, 6, 20, 2, 4, 400
    params, triples, _, _ = make_case(V, R, d, L, "block", nb, Emax, seed=9)
    rng = np.random.RandomState(4)
    X_all, Y_all = decoder_batch(rng, triples, V)
    one_entity = np.stack([np.full(50, 3), rng.randint(0, R, 50), np.full(50, 3)], 1).astype(np.int32)
    cases = [(0, X_all[:40], Y_all[:40]), (Emax, X_all[:1], Y_all[:1]), (17, one_entity, np.ones(50, np.float32)),
             (Emax, X_all, Y_all), (1, X_all[:7], Y_all[:7]), (Emax // 2, X_all[:900], Y_all[:900])]
    eng = native.Engine(V, R, d, L, "block", nb, max_edges=Emax)
    bufs = []
    try:
        eng.set_params(params)
        eng.decoder_reserve(len(X_all))
        eng.optimizer_config(lr=0.01, max_grad_norm=1.0)
        td = eng.to_device(triples)
        bufs.append(td)
        for step, (E, X, Y) in enumerate(cases):
            cur = eng.get_params()
            xd, yd = eng.to_device(np.ascontiguousarray(X)), eng.to_device(np.ascontiguousarray(Y))
            bufs += [xd, yd]
            eng.train_step_device(td, E, xd, yd, len(X), seed=step, reg_param=0.01)
            loss = eng.loss()
            masks = [eng.dropout_mask(l) for l in range(1, L + 1)]
            acts = oracle.encoder_forward(cur, triples[:E], V, L, "block", mode="train", dropout_masks=masks)
            oloss, _, _ = oracle.distmult_loss_and_grads(acts[-1], cur["W_relation"], X, Y, 0.01)
            assert np.isfinite(loss) and abs(loss - oloss) <= 5e-5 * max(1.0, abs(oloss)), (step, loss, oloss)
            after = eng.get_params()
            assert all(np.isfinite(v).all() for v in after.values()), step
            assert any(not np.array_equal(after[k], cur[k]) for k in after), step       # Adam moved something
        with pytest.raises(native.RgcnError):
            eng.train_step_device(td, Emax + 1, xd, yd, len(X), seed=0, reg_param=0.01)
        with pytest.raises(native.RgcnError):
            eng.train_step_device(td, Emax, xd, yd, len(X_all) + 1, seed=0, reg_param=0.01)
    finally:
        for b in bufs:
            b.free()
        eng.close()


def test_device_negative_sampler_layout_and_distribution(native):
    """rgcn_negative_sample_device against the definition of NegativeSampler.transform
    (code/common/auxilliaries.py:13-33): tiling order, labels, which column may change, uniform replacements."""
    V, R, n, rate = 500, 7, 4000, 10
    rng = np.random.RandomState(0)
    batch = np.stack([rng.randint(0, V, n), rng.randint(0, R, n), rng.randint(0, V, n)], 1).astype(np.int32)
    eng = native.Engine(V, R, 8, 1, "block", 2, max_edges=16)
    try:
        bd = eng.to_device(batch)
        total = n * (rate + 1)
        xd, yd = native.DeviceBuffer(eng, 12 * total), native.DeviceBuffer(eng, 4 * total)
        eng.negative_sample_device(bd, n, rate, 123, xd, yd)
        X, Y = xd.download(np.int32, (total, 3)), yd.download(np.float32, (total,))
        eng.negative_sample_device(bd, n, rate, 123, xd, yd)
        assert np.array_equal(X, xd.download(np.int32, (total, 3)))            # a function of the seed
        eng.negative_sample_device(bd, n, rate, 124, xd, yd)
        assert not np.array_equal(X, xd.download(np.int32, (total, 3)))
        for b in (bd, xd, yd):
            b.free()
    finally:
        eng.close()
    assert (Y[:n] == 1).all() and (Y[n:] == 0).all()
    assert np.array_equal(X[:n], batch)
    tiled = np.tile(batch, (rate, 1))
    neg = X[n:]
    assert np.array_equal(neg[:, 1], tiled[:, 1])                              # relations never change
    subj_changed, obj_changed = neg[:, 0] != tiled[:, 0], neg[:, 2] != tiled[:, 2]
    assert not (subj_changed & obj_changed).any()                              # one side per row
    assert (neg[:, [0, 2]] >= 0).all() and (neg[:, [0, 2]] < V).all()
    # fair coin (a replacement can draw the original id: 1/V of the rows look unchanged)
    frac_obj = obj_changed.sum() / float((subj_changed | obj_changed).sum())
    assert abs(frac_obj - 0.5) < 0.01
    assert abs((subj_changed | obj_changed).mean() - (1 - 1.0 / V)) < 0.005
    # replacements are uniform over the entities: chi-square against the flat histogram
    repl = np.concatenate([neg[subj_changed, 0], neg[obj_changed, 2]])
    counts = np.bincount(repl, minlength=V).astype(np.float64)
    expected = len(repl) / float(V)
    chi2 = ((counts - expected) ** 2 / expected).sum()
    assert abs(chi2 - V) < 5 * np.sqrt(2 * V), chi2


@pytest.mark.parametrize("kind,nb,world", [("block", 4, 2), ("block", 4, 3), ("basis", 2, 2)])
def test_sharded_train_steps_match_the_unsharded_run(native, kind, nb, world):
    """Relation-sharded train step (SURVEY 8e + 8f f1/f2) with `world` contexts on ONE device and the test as
    the collective (read_buffer / write_buffer at every exchange point rgcn_train_step_device puts on RCCL):
    forward and backward exchanges per layer, replicated decoder, sharded squared-norm exchange, Adam.  After
    several steps every rank holds the weights of the unsharded run: replicated tensors everywhere, a relation's
    weights on its owner."""
    from relationprediction_amd.sharding import lpt_partition
    V, R, d, L, E = 90, 10, 20, 2, 600
    params, triples, _, _ = make_case(V, R, d, L, kind, nb, E, seed=21)
    X, Y = decoder_batch(np.random.RandomState(2), triples[:200], V)
    owner = lpt_partition(np.bincount(triples[:, 1], minlength=R), world)
    ref = native.Engine(V, R, d, L, kind, nb, max_edges=E)
    engs = [native.Engine(V, R, d, L, kind, nb, max_edges=E, rank=r, world=world) for r in range(world)]
    held = []

    def exchange(which):
        total = sum(e.read_buffer(which) for e in engs)
        for e in engs:
            e.write_buffer(which, total)

    def same_trajectory(got, want, what):
        # Adam's first steps move a weight by about lr * sign(g): where a gradient entry is pure cancellation
        # noise the two summation orders may disagree on its sign, so a handful of entries may sit up to
        # 2 * lr * steps apart; everything else must agree closely
        diff = np.abs(got - want)
        assert float(diff.max()) <= 2 * 0.01 * 4 + 1e-6, what
        assert float((diff > 2e-4).mean()) <= 0.002, (what, float((diff > 2e-4).mean()))

    try:
        for e in [ref] + engs:
            e.set_params(params)
            e.decoder_reserve(len(X))
            e.optimizer_config(lr=0.01, max_grad_norm=1.0)
            held.append((e.to_device(triples), e.to_device(X), e.to_device(Y)))
        for e in engs:
            e.set_relation_owner(owner)
        for step in range(4):
            T, Xd, Yd = held[0]
            ref.train_step_device(T, E, Xd, Yd, len(X), seed=100 + step, reg_param=0.01)
            for e in engs:
                e.set_graph(triples)
                e.forward_begin(train=True, seed=100 + step)
            for l in range(1, L + 1):
                for e in engs:
                    e.forward_layer_partial(l)
                exchange(native.BUF_EXCHANGE)
                for e in engs:
                    e.forward_layer_finish(l)
            for e, (_, Xd, Yd) in zip(engs, held[1:]):
                e.decoder_loss_backward_device(Xd, Yd, len(X), 0.01)
                e.backward_begin()                      # the decoder's own dL/dcodes
            for l in range(L, 0, -1):
                for e in engs:
                    e.backward_layer_partial(l)
                exchange(native.BUF_EXCHANGE)
                exchange(native.BUF_DSELF_EXCHANGE)
                if kind == "basis":
                    exchange(native.BUF_DBASIS_EXCHANGE)
                for e in engs:
                    e.backward_layer_finish(l)
            for e in engs:
                e.backward_end()
                e.optimizer_norm_partial()
            exchange(native.BUF_NORM_EXCHANGE)
            for e in engs:
                e.optimizer_apply()
            for e in engs:
                assert abs(e.loss() - ref.loss()) <= 1e-5 * max(1.0, abs(ref.loss())), step
        want = ref.get_params()
        sharded = ("W_f", "W_b") if kind == "block" else ("C_f", "C_b")
        for r, e in enumerate(engs):
            got = e.get_params()
            for k, w in want.items():
                if k.startswith(sharded):
                    mine = owner == r
                    assert mine.any()
                    same_trajectory(got[k][mine], w[mine], (r, k))
                    assert np.array_equal(got[k][~mine], params[k][~mine]), (r, k)   # never touched off-owner
                else:
                    same_trajectory(got[k], w, (r, k))
        moved = max(float(np.abs(want[k] - params[k]).max()) for k in want)
        assert moved > 0.02                                       # four Adam steps of 0.01 did happen
    finally:
        for t in held:
            for b in t:
                b.free()
        for e in [ref] + engs:
            e.close()


# ------------------------------------------------------------------ edge dropout on the device
def _unique_batch(rng, V, R, n):
    t = np.unique(np.stack([rng.randint(0, V, 3 * n), rng.randint(0, R, 3 * n), rng.randint(0, V, 3 * n)], 1), axis=0)
    assert len(t) >= n
    return np.ascontiguousarray(t[rng.permutation(len(t))[:n]].astype(np.int32))


def test_device_edge_dropout_semantics(native):
    """reference train.py:233-238: the message graph is a random subset of EXACTLY k batch edges, no repeats; degrees
    are those of the kept edges (H8).  Here the draw runs on the device: exact k, rows of the batch in batch order,
    a function of the seed, every edge equally likely; an injected keep mask is obeyed row for row; a forward pass on
    the drawn graph is bitwise the forward pass on the same rows fed as a plain graph."""
    V, R, d, L, nb, n = 150, 7, 16, 2, 4, 3000
    rng = np.random.RandomState(3)
    batch = _unique_batch(rng, V, R, n)
    index_of = {tuple(r): i for i, r in enumerate(batch)}
    params = oracle.init_params(V, R, d, L, "block", nb, rng=rng)
    eng = native.Engine(V, R, d, L, "block", nb, max_edges=n)
    try:
        eng.set_params(params)
        bd = eng.to_device(batch)
        draws = {}
        for seed, keep in ((1, 1500), (1, 1500), (2, 1500), (7, 1), (7, n - 1), (9, 0), (9, n)):
            eng.set_graph_dropout_device(bd, n, keep, seed=seed)
            edges = eng.graph_edges()
            assert edges.shape == (keep, 3)
            idx = np.array([index_of[tuple(r)] for r in edges], dtype=np.int64)
            assert len(set(idx.tolist())) == keep and (np.diff(idx) > 0).all()        # distinct batch rows, batch order
            eng.sync()
            np.testing.assert_array_equal(eng.read_buffer(native.BUF_INDEG), np.bincount(edges[:, 2], minlength=V))
            np.testing.assert_array_equal(eng.read_buffer(native.BUF_OUTDEG), np.bincount(edges[:, 0], minlength=V))
            draws.setdefault((seed, keep), []).append(idx)
        a, b = draws[(1, 1500)]
        assert np.array_equal(a, b) and not np.array_equal(a, draws[(2, 1500)][0])
        # the draw IS "the k smallest of the keys (40 bits of splitmix64(seed, e)) << 24 | e" (graph_prep.hip), on both
        # code paths of the kernel: keys kept in registers (n <= 32,768) and recomputed per sweep (larger batches)
        def smallest_keys(seed, n_edges, k):
            e = np.arange(n_edges, dtype=np.uint64)
            with np.errstate(over="ignore"):
                z = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (e + np.uint64(0x51ED27))
                z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
                z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
                z = z ^ (z >> np.uint64(31))
            keys = (z & np.uint64(0xFFFFFFFFFF000000)) | e
            return np.sort(np.argsort(keys, kind="stable")[:k])
        assert np.array_equal(a, smallest_keys(1, n, 1500))
        big_n = 40000
        big = _unique_batch(np.random.RandomState(4), V, R, big_n)
        big_eng = native.Engine(V, R, d, L, "block", nb, max_edges=big_n)
        try:
            bb = big_eng.to_device(big)
            for seed, keep in ((5, 20000), (6, 123), (6, big_n - 1)):
                big_eng.set_graph_dropout_device(bb, big_n, keep, seed=seed)
                np.testing.assert_array_equal(big_eng.graph_edges(), big[smallest_keys(seed, big_n, keep)])
            bb.free()
        finally:
            big_eng.close()
        # the drawn graph behaves exactly like the same rows fed directly
        eng.set_graph_dropout_device(bd, n, 1500, seed=1)
        eng.forward(train=False)
        got = eng.codes()
        eng.set_graph(batch[a])
        eng.forward(train=False)
        np.testing.assert_array_equal(got, eng.codes())
        # injected set (how a test replays the reference's np.random.choice): obeyed row for row
        mask = np.zeros(n, np.uint8)
        mask[rng.choice(n, 1234, replace=False)] = 1
        md = eng.to_device(mask)
        eng.set_graph_dropout_device(bd, n, 1234, keep_mask=md)
        np.testing.assert_array_equal(eng.graph_edges(), batch[mask.astype(bool)])
        eng.set_graph_dropout_device(bd, n, 1233, keep_mask=md)                        # a mask with another count is refused
        with pytest.raises(native.RgcnError):
            eng.sync()
        # uniformity: 400 edges, keep 100, 300 seeds -> every draw holds exactly 100, each edge ~ Binomial(300, 1/4)
        small = eng.to_device(batch[:400])
        counts = np.zeros(400, np.int64)
        for seed in range(300):
            eng.set_graph_dropout_device(small, 400, 100, seed=1000 + seed)
            kept = eng.graph_edges()
            counts[[index_of[tuple(r)] for r in kept]] += 1
        assert counts.sum() == 300 * 100
        assert counts.min() >= 40 and counts.max() <= 115, (counts.min(), counts.max())     # mean 75, sd 7.5: +-5 sd
        half = counts[:200].sum() / float(counts.sum())
        assert abs(half - 0.5) < 0.02                                                        # no positional bias
        for b in (bd, md, small):
            b.free()
    finally:
        eng.close()


@pytest.mark.parametrize("prefetch", [False, True])
def test_minibatch_step_equals_the_step_on_its_own_draws(native, prefetch):
    """rgcn_train_step_minibatch_device (edge dropout + negatives + train step, one call) against the same three
    pieces called one by one with the rows the dropout kernel drew: weights bitwise equal after three Adam steps; with
    the graph of each step prepared beside the previous one (rgcn_prefetch_graph_dropout_device) as well"""
    V, R, d, L, nb, n, rate = 120, 6, 20, 2, 4, 800, 3
    rng = np.random.RandomState(5)
    batches = [_unique_batch(rng, V, R, n) for _ in range(2)]
    params = oracle.init_params(V, R, d, L, "block", nb, rng=rng)
    N = n * (rate + 1)
    out = []
    for fused in (True, False):
        eng = native.Engine(V, R, d, L, "block", nb, max_edges=n)
        try:
            eng.set_params(params)
            eng.decoder_reserve(N)
            eng.optimizer_config(lr=0.01, max_grad_norm=1.0)
            bds = [eng.to_device(b) for b in batches]
            xd, yd = eng.alloc(12 * N), eng.alloc(4 * N)
            gd = eng.alloc(12 * n)
            for step in range(3):
                bd, keep, eseed, nseed, seed = bds[step % 2], 300 + 50 * step, 40 + step, 90 + step, 7 + step
                if fused:
                    eng.train_step_minibatch_device(bd, n, keep, eseed, rate, nseed, xd, yd, seed=seed, reg_param=0.01)
                    if prefetch and step < 2:
                        eng.prefetch_graph_dropout_device(bds[(step + 1) % 2], n, 300 + 50 * (step + 1), 40 + step + 1)
                else:
                    eng.set_graph_dropout_device(bd, n, keep, seed=eseed)
                    eng.copy_to_device(gd, eng.graph_edges())
                    eng.negative_sample_device(bd, n, rate, nseed, xd, yd)
                    eng.train_step_device(gd, keep, xd, yd, N, seed=seed, reg_param=0.01)
                assert np.isfinite(eng.loss())
            out.append({k: eng.get_param(k) for k in eng.param_names})
            for b in bds + [xd, yd, gd]:
                b.free()
        finally:
            eng.close()
    for k in out[0]:
        np.testing.assert_array_equal(out[0][k], out[1][k], err_msg=k)


def test_decoder_at_baseline_scale(native):
    """The decoder at BASELINE.json's size -- V = 14,541, d = 500, 30,000 positives x (1 + 10) = 330,000 triples drawn
    by the device negative sampler (the tiled order, the pipelined energy kernel, hub entities cut into pieces) -- against
    the chunked float64 restatement of the oracle (tests/helpers.py, pinned to it by tests/test_oracle_chunked.py): loss to 2e-6, every entry of dL/dcodes and dL/dW_relation within 5e-6 of scale."""
    import helpers
    V, R, d, L, nb, rate = 14541, 237, 500, 1, 100, 10
    graph = helpers.load_graph("fb237_minibatch")
    pool = helpers.load_graph("fb237_valid_test")
    rng = np.random.RandomState(12)
    batch = np.ascontiguousarray(np.concatenate([graph, pool[rng.choice(len(pool), 15000, replace=False)]]).astype(np.int32))
    n = len(batch)
    N = n * (rate + 1)
    params = oracle.init_params(V, R, d, L, "block", nb, rng=rng)
    params["W_relation"] = (rng.randn(V, d) * 0.5).astype(np.float32)
    eng = native.Engine(V, R, d, L, "block", nb, max_edges=len(graph))
    try:
        eng.set_params(params)
        eng.set_graph(graph)
        eng.forward(train=False)
        codes = eng.codes()
        eng.decoder_reserve(N)
        bd, xd, yd = eng.to_device(batch), eng.alloc(12 * N), eng.alloc(4 * N)
        eng.negative_sample_device(bd, n, rate, 23, xd, yd)
        X, Y = xd.download(np.int32, (N, 3)), yd.download(np.float32, (N,))
        eng.decoder_loss_backward_device(xd, yd, N, 0.01)
        loss, dcodes, dwrel = eng.loss(), eng.dcodes(), eng.get_grad("W_relation")
        for b in (bd, xd, yd):
            b.free()
    finally:
        eng.close()
    oloss, odcodes, odwrel = helpers.chunked_distmult_float64(codes, params["W_relation"], X, Y, 0.01)
    assert abs(loss - oloss) <= 2e-6 * abs(oloss), (loss, oloss)
    for got, want, name in ((dcodes, odcodes, "dcodes"), (dwrel, odwrel, "dW_relation")):
        err = float(np.abs(got.astype(np.float64) - want).max())
        assert err <= 5e-6 * float(np.abs(want).max()), (name, err, float(np.abs(want).max()))
    assert not dwrel[R:].any()


def test_train_step_at_baseline_scale_optimizer_replay(native):
    """One whole minibatch train step at BASELINE's sizes (V = 14,541, d = 500, 100 blocks, 30,000-edge batch with
    exact-15,000 edge dropout, 330,000 decoder triples): the loss is finite and of the expected size, every gradient is
    finite and non-trivial, and the device's global-norm clip + Adam update over all 10 million weights equals a numpy
    float64 replay of tensorflow_backend/algorithms.py:27-42,58-68 from the device's own gradients -- two steps, so that
    Adam's moments and step count carry over."""
    import helpers
    V, R, d, L, nb, rate, keep = 14541, 237, 500, 2, 100, 10, 15000
    graph = helpers.load_graph("fb237_minibatch")
    pool = helpers.load_graph("fb237_valid_test")
    rng = np.random.RandomState(31)
    batch = np.ascontiguousarray(np.concatenate([graph, pool[rng.choice(len(pool), 15000, replace=False)]]).astype(np.int32))
    n = len(batch)
    N = n * (rate + 1)
    params = oracle.init_params(V, R, d, L, "block", nb, rng=rng)
    names = [k for k in oracle.weight_names("block", L) if not (k.startswith("b") and k != "b_emb")]
    eng = native.Engine(V, R, d, L, "block", nb, keep_prob=0.8, max_edges=n)
    try:
        eng.set_params(params)
        eng.decoder_reserve(N)
        eng.optimizer_config(lr=0.01, beta1=0.9, beta2=0.999, eps=1e-8, max_grad_norm=1.0)
        bd, xd, yd = eng.to_device(batch), eng.alloc(12 * N), eng.alloc(4 * N)
        cur = {k: eng.get_param(k) for k in eng.param_names}
        m = {k: np.zeros_like(cur[k], dtype=np.float64) for k in names}
        v = {k: np.zeros_like(cur[k], dtype=np.float64) for k in names}
        for step in (1, 2):
            eng.train_step_minibatch_device(bd, n, keep, 70 + step, rate, 80 + step, xd, yd, seed=90 + step, reg_param=0.01)
            loss = eng.loss()
            assert np.isfinite(loss) and 0.05 < loss < 2.0, loss
            grads = {k: eng.get_grad(k).astype(np.float64) for k in names}
            assert all(np.isfinite(g).all() and np.abs(g).max() > 0 for g in grads.values())
            gn = np.sqrt(sum(float((g ** 2).sum()) for g in grads.values()))
            scale = 1.0 / max(gn, 1.0)                                   # clip_by_global_norm(max 1)
            lr_t = 0.01 * np.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
            for k in names:
                g = grads[k] * scale
                m[k] = 0.9 * m[k] + 0.1 * g
                v[k] = 0.999 * v[k] + 0.001 * g * g
                want = cur[k].astype(np.float64) - lr_t * m[k] / (np.sqrt(v[k]) + 1e-8)
                got = eng.get_param(k)
                # an update is at most lr_t in size: 1e-3 of that, entry by entry (fp32 moments against float64)
                assert float(np.abs(got - want).max()) <= 1e-3 * lr_t, (step, k, float(np.abs(got - want).max()))
                cur[k] = got
        for b in (bd, xd, yd):
            b.free()
    finally:
        eng.close()
