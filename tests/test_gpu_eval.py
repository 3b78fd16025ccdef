"""Device ranking (include/rgcn.h rgcn_rank_device) against the oracle's restatement of the reference's
MrrScore.append_line (code/common/evaluation.py:148-153) — `-m gpu`, through the C ABI."""
import numpy as np
import pytest

import oracle
from helpers import make_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from relationprediction_amd import _native
    _native.load_library()
    return _native


def known_lists(triples, object_side):
    known = {}
    for s, r, o in triples:
        key, val = ((s, r), o) if object_side else ((o, r), s)
        lst = known.setdefault(key, [])
        if val not in lst:                      # Scorer.extend_triple_dict keeps distinct values
            lst.append(val)
    return known


def csr_for(queries, known, object_side):
    ptr, idx = [0], []
    for s, r, o in queries:
        idx += known[(s, r) if object_side else (o, r)]
        ptr.append(len(idx))
    return np.asarray(ptr, np.int64), np.asarray(idx, np.int32)


@pytest.mark.parametrize("V,R,d,nb,E,scale", [(120, 7, 20, 4, 500, 1.0), (300, 11, 40, 8, 1500, 1.0),
                                              (90, 5, 20, 4, 400, 40.0)])
def test_ranks_match_reference_definition(native, V, R, d, nb, E, scale):
    """scale = 40 drives most energies into sigmoid saturation: scores tie at exactly 1.0 / 0.0 and the
    `>=` comparison must count those ties as the reference does."""
    params, triples, _, _ = make_case(V, R, d, 2, "block", nb, E, seed=E)
    rng = np.random.RandomState(3)
    params["W_relation"] = (rng.randn(V, d) * scale).astype(np.float32)
    queries = triples[rng.choice(len(triples), 150, replace=False)].copy()
    queries[:10] = queries[0]                                     # repeated queries
    eng = native.Engine(V, R, d, 2, "block", nb, max_edges=E)
    try:
        eng.set_params(params)
        eng.set_graph(triples)
        eng.forward(train=False)
        codes = eng.codes()
        codes[5] = codes[6]                                       # (host copy only; ties come from saturation)
        eng.rank_reserve(64)                                      # forces chunking: 150 queries, 64 per chunk
        for object_side in (True, False):
            known = known_lists(np.concatenate([triples, queries]), object_side)
            ptr, idx = csr_for(queries, known, object_side)
            raw, filt = eng.ranks(queries, object_side, ptr, idx)
            oraw, ofilt = oracle.distmult_ranks(eng.codes(), params["W_relation"], queries, object_side, known)
            assert (raw >= 1).all() and (filt >= 1).all() and (filt <= raw).all()
            # energies come from two fp32 GEMMs with different summation orders: a near-tie may flip one rank
            assert np.mean(raw != oraw) <= 0.02 and np.abs(raw - oraw).max() <= 1, (raw[:10], oraw[:10])
            assert np.mean(filt != ofilt) <= 0.02 and np.abs(filt - ofilt).max() <= 1
            if scale > 1:
                assert (raw > 1).mean() > 0.3                     # saturation really produced ties
    finally:
        eng.close()


def test_rank_counts_are_bit_exact_on_the_devices_own_energies(native):
    """The ranking has a float half (energies: one GEMM, compared with the oracle's within rounding, above) and an
    integer half (sigmoid -> `>=` counts -> raw / filtered rank).  The integer half is held BIT-EXACT: the energies the
    device scored are read back (RGCN_BUF_RANK_ENERGIES) and the reference's rank definition
    (oracle.ranks_from_energies; evaluation.py:148-153) applied to exactly those values must give exactly the
    device's ranks -- saturated ties (scale 40), repeated queries, filtered lists and both sides included."""
    for (V, R, d, nb, E, scale) in [(300, 11, 40, 8, 1500, 1.0), (90, 5, 20, 4, 400, 40.0), (1031, 7, 20, 4, 3000, 6.0)]:
        params, triples, _, _ = make_case(V, R, d, 2, "block", nb, E, seed=E + 1)
        rng = np.random.RandomState(5)
        params["W_relation"] = (rng.randn(V, d) * scale).astype(np.float32)
        queries = triples[rng.choice(len(triples), 120, replace=False)].copy()
        queries[:7] = queries[3]
        eng = native.Engine(V, R, d, 2, "block", nb, max_edges=E)
        try:
            eng.set_params(params)
            eng.set_graph(triples)
            eng.forward(train=False)
            eng.rank_reserve(len(queries))                        # one chunk: the buffer holds every query's row
            for object_side in (True, False):
                known = known_lists(np.concatenate([triples, queries]), object_side)
                ptr, idx = csr_for(queries, known, object_side)
                raw, filt = eng.ranks(queries, object_side, ptr, idx)
                energies = eng.read_buffer(native.BUF_RANK_ENERGIES)
                assert energies.shape == (len(queries), V)
                for i, (s, r, o) in enumerate(queries):
                    gold, key = (o, (s, r)) if object_side else (s, (o, r))
                    want = oracle.ranks_from_energies(energies[i], gold, known[key])
                    assert (int(raw[i]), int(filt[i])) == want, (V, object_side, i, raw[i], filt[i], want)
                if scale > 1:
                    assert (raw > 1).mean() > 0.2
        finally:
            eng.close()


def test_sharded_contexts_rank_their_own_query_slices(native):
    """Evaluation on a relation-sharded encoder (SURVEY 8e + f3): two contexts on one device, the test as the
    collective for the test-mode forward, then each rank ranks ITS half of the queries; the concatenation is what
    the unsharded context returns for all of them."""
    from relationprediction_amd.sharding import lpt_partition
    V, R, d, nb, E, L, world = 150, 8, 20, 4, 700, 2, 2
    params, triples, _, _ = make_case(V, R, d, L, "block", nb, E, seed=31)
    queries = triples[np.random.RandomState(5).choice(E, 120, replace=False)]
    known = known_lists(triples, True)
    owner = lpt_partition(np.bincount(triples[:, 1], minlength=R), world)
    ref = native.Engine(V, R, d, L, "block", nb, max_edges=E)
    engs = [native.Engine(V, R, d, L, "block", nb, max_edges=E, rank=r, world=world) for r in range(world)]
    try:
        ref.set_params(params)
        ref.set_graph(triples)
        ref.forward(train=False)
        ref.rank_reserve(64)
        ptr, idx = csr_for(queries, known, True)
        want_raw, want_filt = ref.ranks(queries, True, ptr, idx)
        for e in engs:
            e.set_params(params)
            e.set_relation_owner(owner)
            e.set_graph(triples)
            e.forward_begin(train=False)
            e.rank_reserve(64)
        for l in range(1, L + 1):
            for e in engs:
                e.forward_layer_partial(l)
            total = sum(e.read_buffer(native.BUF_EXCHANGE) for e in engs)
            for e in engs:
                e.write_buffer(native.BUF_EXCHANGE, total)
                e.forward_layer_finish(l)
        raw, filt = [], []
        for r, e in enumerate(engs):
            mine = queries[r::world]
            ptr, idx = csr_for(mine, known, True)
            a, b = e.ranks(mine, True, ptr, idx)
            raw.append(a)
            filt.append(b)
        got_raw, got_filt = np.empty_like(want_raw), np.empty_like(want_filt)
        for r in range(world):
            got_raw[r::world], got_filt[r::world] = raw[r], filt[r]
        # the exchanged sum and the single-context reduction add in different orders: a near-tie may move a rank by 1
        assert np.mean(got_raw != want_raw) <= 0.02 and np.abs(got_raw - want_raw).max() <= 1
        assert np.mean(got_filt != want_filt) <= 0.02 and np.abs(got_filt - want_filt).max() <= 1
    finally:
        for e in [ref] + engs:
            e.close()


def test_rank_argument_checks(native):
    V, R, d = 50, 3, 8
    params, triples, _, _ = make_case(V, R, d, 1, "block", 2, 60, seed=1)
    eng = native.Engine(V, R, d, 1, "block", 2, max_edges=60)
    try:
        eng.set_params(params)
        q = triples[:4]
        ptr, idx = np.arange(5, dtype=np.int64), q[:, 2].astype(np.int32)
        with pytest.raises(native.RgcnError):                     # no forward yet
            eng.rank_reserve(8)
            eng.ranks(q, True, ptr, idx)
        eng.set_graph(triples)
        eng.forward(train=False)
        bad = q.copy(); bad[1, 2] = V
        with pytest.raises(native.RgcnError):
            eng.ranks(bad, True, ptr, idx)
        with pytest.raises(native.RgcnError):
            eng.ranks(q, True, ptr, np.array([0, 1, V + 3, 2], np.int32))
        raw, filt = eng.ranks(q, True, ptr, idx)                  # still usable afterwards
        assert (filt >= 1).all()
    finally:
        eng.close()


def test_ranks_at_baseline_scale_equal_the_reference_scorer(native):
    """tests/golden/reference_ranks_fullscale.npz: the REFERENCE'S OWN Scorer (common/evaluation.py, run by
    tests/golden/make_reference_rank_fixture.py) ranked 1,200 real FB15k-237 triples on both sides against all 14,541
    entities, raw and filtered, on DistMult scores of a seeded code / relation table.  Here the same tables go into the
    engine (a one-layer chain on an empty graph whose codes ARE the table, exactly), this package's Scorer drives
    rgcn_rank_device with the reference's filter lists, and the 2 x 2,400 ranks must come out the same -- up to the one
    place two fp32 GEMMs may legitimately disagree, a near-tie with the gold entity (|rank difference| <= 2 on under
    1 % of the queries; everything else identical), MRR to 1e-6."""
    import os
    import sys
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, golden)
    try:
        import make_reference_rank_fixture as fx
    finally:
        sys.path.remove(golden)
    from relationprediction_amd.common import evaluation
    with np.load(os.path.join(golden, "reference_ranks_fullscale.npz")) as z:
        want_raw, want_filt = z["raw_ranks"], z["filtered_ranks"]
        assert list(z["config"]) == [fx.V, fx.R, fx.HALF, fx.SEED, fx.NQ]
        want_mrr = (float(z["mrr_raw"]), float(z["mrr_filtered"]))
    codes, rel = fx.tables()
    train, valid, test = fx.splits()
    V, R, d = fx.V, fx.R, 2 * fx.HALF
    c = codes[:, :fx.HALF]
    eye = np.eye(fx.HALF, dtype=np.float32)
    w_self = np.zeros((d, d), dtype=np.float32)
    w_self[:fx.HALF, :fx.HALF] = eye
    w_self[fx.HALF:, :fx.HALF] = -eye
    w_rel = np.zeros((V, d), dtype=np.float32)
    w_rel[:R] = rel
    eng = native.Engine(V, R, d, 1, "block", d // 4, max_edges=1)
    try:
        params = {"W_emb": np.concatenate([np.maximum(c, 0), np.maximum(-c, 0)], axis=1), "b_emb": np.zeros(d, np.float32),
                  "W_f1": np.zeros((R, d // 4, 4, 4), np.float32), "W_b1": np.zeros((R, d // 4, 4, 4), np.float32),
                  "W_self1": w_self, "b1": np.zeros(d, np.float32), "W_relation": w_rel}
        eng.set_params(params)
        eng.set_graph(np.zeros((0, 3), dtype=np.int32))
        eng.forward(train=False)
        assert np.array_equal(eng.codes(), codes)                 # the table, bit for bit

        class EngineModel(object):                                # what Scorer asks of a model on the device path
            test_graph = None

            def device_ranks(self, graph, triples, predict_object, ptr, idx):
                return eng.ranks(np.ascontiguousarray(triples, dtype=np.int32), predict_object, ptr, idx)
        eng.rank_reserve(1000)
        scorer = evaluation.Scorer({"Metric": "MRR"})
        for part in (train, valid, test):
            scorer.register_data(part)
        scorer.register_degrees(train)
        scorer.register_model(EngineModel())
        scorer.finalize_frequency_computation(np.concatenate((train, valid, test), axis=0))
        score = scorer.compute_scores(test, verbose=False)
        raw, filt = np.asarray(score.raw_ranks), np.asarray(score.filtered_ranks)
        assert raw.shape == want_raw.shape == (2 * fx.NQ,)
        for got, want in ((raw, want_raw), (filt, want_filt)):
            assert np.mean(got != want) < 0.01 and np.abs(got - want).max() <= 2, (np.mean(got != want), np.abs(got - want).max())
        summary = score.get_summary()
        assert summary.results['Raw'][summary.mrr_string()] == pytest.approx(want_mrr[0], abs=1e-6)
        assert summary.results['Filtered'][summary.mrr_string()] == pytest.approx(want_mrr[1], abs=1e-6)
    finally:
        eng.close()
