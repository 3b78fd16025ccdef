"""The neighbourhood edge sampler on the device (include/rgcn.h rgcn_sample_neighborhood_device, csrc/neighborhood.hip)
against the reference's process (code/train.py:161-198, restated step for step in oracle.sample_edge_neighborhood and
held draw for draw to the reference's own function by tests/test_reference_fixtures.py).  The device algorithm is a
parallel one (first-passage percolation with Exp(1) clocks on the edge ends + a uniform vertex order over the
components), another random stream with the SAME distribution: the tests compare distributions, and the invariants."""
import collections
import time

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from relationprediction_amd import _native
    _native.load_library()
    return _native


def draw(native, eng, buf, k, seed):
    eng.sample_neighborhood_device(k, seed, buf)
    return buf.download(np.int32, (k, 3))


def outcome(rows):
    """the drawn set as a multiset of rows (parallel edges have equal rows: which of them was drawn cannot be told
    from the batch, nor does it matter to anything downstream)"""
    return tuple(sorted(map(tuple, np.asarray(rows).tolist())))


def rows_to_ids(triples, rows):
    """edge ids of the drawn rows (rows come in edge order; parallel edges are told apart by their position)"""
    ids, j = [], 0
    for r in rows:
        while not np.array_equal(triples[j], r):
            j += 1
        ids.append(j)
        j += 1
    return ids


# two components, a triangle, a self loop, a pair of parallel edges; and a small cyclic graph
TINY = [(np.array([[0, 0, 1], [1, 0, 2], [2, 0, 0], [2, 0, 3], [3, 0, 3], [4, 0, 5], [4, 0, 5], [5, 0, 6]]), 8, (3, 6)),
        (np.array([[0, 0, 1], [1, 0, 2], [2, 0, 3], [3, 0, 4], [0, 0, 2]]), 5, (2, 3))]


@pytest.mark.parametrize("case", [0, 1])
def test_set_distribution_equals_the_reference_process(native, case):
    """every possible outcome SET of a tiny graph: frequencies over 6,000 seeds on the device against 6,000 runs of the
    reference's loop, each within 4.5 standard errors -- restarts into a second component, the double weight of an
    edge whose two ends are both touched, self loops and parallel edges included"""
    triples, V, ks = TINY[case]
    triples = triples.astype(np.int32)
    n = 6000
    with native.Engine(V, 1, 4, 1, "block", 1, max_edges=len(triples)) as eng:
        eng.neighborhood_reserve(triples)
        for k in ks:
            buf = native.DeviceBuffer(eng, 12 * k)
            try:
                dev = collections.Counter()
                for seed in range(n):
                    dev[outcome(draw(native, eng, buf, k, 1000 + seed))] += 1
            finally:
                buf.free()
            ref = collections.Counter()
            rng = np.random.RandomState(case * 10 + k)
            for _ in range(n):
                ref[outcome(triples[oracle.sample_edge_neighborhood(triples, V, k, rng)])] += 1
            assert all(len(s) == k for s in dev)
            for s in set(dev) | set(ref):
                pa, pb = dev[s] / n, ref[s] / n
                p = max((pa + pb) / 2, 1e-3)
                se = np.sqrt(2 * p * (1 - min(p, 0.999)) / n)
                assert abs(pa - pb) <= 4.5 * se, (case, k, s, pa, pb)


def test_inclusion_frequencies_on_a_medium_graph(native):
    """300 vertices, 1,500 edges with hubs and a few small components, 400 picks: per-edge inclusion frequency over 1,200
    seeds against 1,200 runs of the reference's loop (5 standard errors, every edge), and the mean number of touched
    vertices"""
    rng = np.random.RandomState(3)
    V, E, k, n = 300, 1500, 400, 1200
    s = np.where(rng.rand(E) < 0.3, rng.randint(0, 6, E), rng.randint(0, 260, E))
    o = rng.randint(0, 260, E)
    triples = np.stack([s, rng.randint(0, 5, E), o], 1).astype(np.int32)
    triples[-30:, 0] = rng.randint(260, 300, 30)            # small components hanging off vertices 260..299
    triples[-30:, 2] = rng.randint(260, 300, 30)
    with native.Engine(V, 5, 4, 1, "block", 1, max_edges=E) as eng:
        eng.neighborhood_reserve(triples)
        buf = native.DeviceBuffer(eng, 12 * k)
        try:
            inc_dev, verts_dev = np.zeros(E), 0.0
            first = draw(native, eng, buf, k, 5)
            assert np.array_equal(first, draw(native, eng, buf, k, 5))          # a function of the seed
            assert not np.array_equal(first, draw(native, eng, buf, k, 6))
            for seed in range(n):
                ids = rows_to_ids(triples, draw(native, eng, buf, k, 77 + seed))
                assert len(set(ids)) == k and ids == sorted(ids)                # exactly k distinct edges, edge order
                inc_dev[ids] += 1
                verts_dev += len(set(triples[ids][:, [0, 2]].ravel().tolist()))
        finally:
            buf.free()
        with pytest.raises(native.RgcnError):
            big = native.DeviceBuffer(eng, 12 * (E + 1))
            try:
                eng.sample_neighborhood_device(E + 1, 1, big)                   # more than the graph has
            finally:
                big.free()
    inc_ref, verts_ref = np.zeros(E), 0.0
    r2 = np.random.RandomState(4)
    for _ in range(n):
        ids = oracle.sample_edge_neighborhood(triples, V, k, r2)
        inc_ref[ids] += 1
        verts_ref += len(set(triples[ids][:, [0, 2]].ravel().tolist()))
    pa, pb = inc_dev / n, inc_ref / n
    p = np.clip((pa + pb) / 2, 1e-3, 0.999)
    z = np.abs(pa - pb) / np.sqrt(2 * p * (1 - p) / n)
    assert z.max() <= 5.0, (int(z.argmax()), float(z.max()), pa[z.argmax()], pb[z.argmax()])
    assert abs(verts_dev - verts_ref) / n <= 2.0, (verts_dev / n, verts_ref / n)


def test_training_graph_scale(native):
    """the 272,115-edge synthetic FB15k-237 training graph, 30,000 picks (settings/gcn_block.exp's GraphBatchSize): the
    batch is 30,000 distinct rows of the graph forming ONE connected patch (the giant component is never exhausted), and
    the whole draw takes a few hundred microseconds of device time"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_graph", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    triples = bench.load_graph("synth:fb237_valid_test:272115")
    V, k = 14541, 30000
    with native.Engine(V, 237, 4, 1, "block", 1, max_edges=k) as eng:
        eng.neighborhood_reserve(triples)
        buf = native.DeviceBuffer(eng, 12 * k)
        try:
            rows = draw(native, eng, buf, k, 123)
            eng.sync()
            t0 = time.perf_counter()
            for i in range(20):
                eng.sample_neighborhood_device(k, 200 + i, buf)
            eng.sync()
            ms = (time.perf_counter() - t0) * 1e3 / 20
            all_rows = native.DeviceBuffer(eng, 12 * len(triples))
            try:
                full = draw(native, eng, all_rows, len(triples), 9)             # the whole graph: every row once
            finally:
                all_rows.free()
        finally:
            buf.free()
    print("device neighbourhood sampler: %.3f ms per 30,000-edge batch" % ms)
    assert np.array_equal(full, triples)
    key = lambda t: (t[:, 0].astype(np.int64) * 237 + t[:, 1]) * V + t[:, 2]    # noqa: E731  (triples are unique)
    assert len(np.unique(key(rows))) == k and np.isin(key(rows), key(triples)).all()
    # one connected patch: union-find over the drawn edges
    parent = np.arange(V)

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for a, _, b in rows:
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[ra] = rb
    touched = np.unique(rows[:, [0, 2]])
    assert len({find(v) for v in touched}) == 1
    assert ms < 5.0


def test_a_graph_a_thousand_hops_across(native):
    """A chain of 1,200 vertices (diameter 1,199): the sweep budget is sized from the graph's diameter when the graph is
    handed over, so the relaxations settle where a fixed small-world budget would refuse.  On a chain the reference's
    process grows an interval, to the left or to the right with equal probability: every batch is 500 consecutive
    edges, and the interval's left end has the distribution of the reference's loop (mean over 150 draws each)."""
    V, k, n = 1200, 500, 150
    triples = np.stack([np.arange(V - 1), np.zeros(V - 1, dtype=np.int64), np.arange(1, V)], 1).astype(np.int32)
    with native.Engine(V, 1, 4, 1, "block", 1, max_edges=V) as eng:
        eng.neighborhood_reserve(triples)
        buf = native.DeviceBuffer(eng, 12 * k)
        try:
            lefts = []
            for seed in range(n):
                rows = draw(native, eng, buf, k, 31 + seed)
                assert np.array_equal(rows[:, 2], rows[:, 0] + 1)                      # rows of the chain
                assert np.array_equal(rows[:, 0], np.arange(rows[0, 0], rows[0, 0] + k))   # consecutive: one interval
                lefts.append(int(rows[0, 0]))
        finally:
            buf.free()
    rng = np.random.RandomState(2)
    ref = [int(np.min(oracle.sample_edge_neighborhood(triples, V, k, rng))) for _ in range(n)]
    se = np.sqrt((np.var(lefts) + np.var(ref)) / n)
    assert abs(np.mean(lefts) - np.mean(ref)) <= 4.5 * se, (np.mean(lefts), np.mean(ref), se)


def test_thousands_of_components_against_the_host_sampler(native):
    """The 10,000 real WN18 valid+test triples: 3,914 components, the largest 482 edges.  A batch of 5,000 edges there is
    ~950 whole components and a piece of one more -- the restart rule at scale.  Touched vertices, components present
    and components taken in full over 40 draws, against the HOST sampler (the reference's process pick for pick, held
    draw for draw to the reference by tests/test_reference_fixtures.py): means within 4.5 standard errors."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with np.load(os.path.join(root, "tests", "golden", "graphs.npz")) as z:
        triples = np.ascontiguousarray(z["wn18_valid_test"].astype(np.int32))
    V, k, n = 40943, 5000, 40
    parent = np.arange(V)

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for s, _, o in triples:
        a, b = find(s), find(o)
        if a != b:
            parent[max(a, b)] = min(a, b)
    comp = np.array([find(v) for v in range(V)])
    comp_edges = np.bincount(comp[triples[:, 0]], minlength=V)

    def stats(t):
        got = np.bincount(comp[t[:, 0]], minlength=V)
        return len(np.unique(t[:, [0, 2]])), int((got > 0).sum()), int(((got > 0) & (got == comp_edges)).sum())
    with native.Engine(V, 18, 4, 1, "block", 1, max_edges=k) as eng:
        eng.neighborhood_reserve(triples)
        buf = native.DeviceBuffer(eng, 12 * k)
        try:
            dev = [stats(draw(native, eng, buf, k, 900 + i)) for i in range(n)]
        finally:
            buf.free()
    host = native.NeighborhoodSampler(triples, V)
    ref = [stats(triples[host.sample(k, 4000 + i)]) for i in range(n)]
    host.close()
    dev, ref = np.array(dev, dtype=np.float64), np.array(ref, dtype=np.float64)
    assert (ref[:, 2] > 500).all() and (ref[:, 1] - ref[:, 2] <= 1).all()      # whole components + at most one piece
    assert (dev[:, 1] - dev[:, 2] <= 1).all()
    for j in range(3):
        se = np.sqrt((dev[:, j].var() + ref[:, j].var()) / n) + 1e-9
        assert abs(dev[:, j].mean() - ref[:, j].mean()) <= 4.5 * se, (j, dev[:, j].mean(), ref[:, j].mean(), se)


def test_draws_on_the_two_streams_do_not_share_a_draw(native):
    """One set of per-draw state (parameters, distances, keys, the recorded graph) serves the main AND the prefetch
    stream.  The driver's first iteration draws batch 1 on the main stream and at once batch 2 on the prefetch stream
    (optimize.py: update_from_batch, then stage()): the second draw must wait for the first one's end instead of
    overwriting the parameters its kernels still read.  Back-to-back draws on alternating streams give, bit for bit,
    the batches the same seeds give one at a time (and never raise the batch-size flag)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_graph4", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    triples = bench.load_graph("synth:fb237_valid_test:272115")
    V, k, rounds = 14541, 30000, 6
    with native.Engine(V, 237, 4, 1, "block", 1, max_edges=k) as eng:
        eng.neighborhood_reserve(triples)
        bufs = [native.DeviceBuffer(eng, 12 * k) for _ in range(2 * rounds)]
        try:
            alone = []
            for i in range(2 * rounds):
                eng.sample_neighborhood_device(k, 1000 + i, bufs[0])
                eng.sync()
                alone.append(bufs[0].download(np.int32, (k, 3)))
            for i in range(2 * rounds):          # no host wait between the draws; even: main stream, odd: prefetch stream
                eng.sample_neighborhood_device(k, 1000 + i, bufs[i], on_prefetch_stream=bool(i % 2))
            eng.sync()
            for i in range(2 * rounds):
                np.testing.assert_array_equal(bufs[i].download(np.int32, (k, 3)), alone[i], err_msg="draw %d" % i)
        finally:
            for b in bufs:
                b.free()


@pytest.mark.slow
@pytest.mark.parametrize("case", range(len(TINY)))
def test_large_sample_chi_square_against_the_reference_process(native, case):
    """The 6,000-draw gate above sees a 4.5-sigma deviation of a single outcome; this one is the homogeneity test the
    round-3 review asked to have inside the suite: 60,000 draws per (graph, k) of the device sampler and of the
    reference's loop (oracle.sample_edge_neighborhood, pinned draw for draw to train.py:161-198), chi-square over all
    outcome sets with at least 10 observations (p > 1e-4: a false alarm once in ten thousand runs per case) and no single
    outcome beyond 5 standard errors.  (tools/nbr_distribution_check.py is the same check up to 400,000 draws.)"""
    import collections
    from scipy import stats
    n = 60000
    triples, V, ks = TINY[case]
    triples = triples.astype(np.int32)
    with native.Engine(V, 1, 4, 1, "block", 1, max_edges=len(triples)) as eng:
        eng.neighborhood_reserve(triples)
        for k in ks:
            buf = native.DeviceBuffer(eng, 12 * k)
            try:
                dev = collections.Counter()
                for seed in range(n):
                    eng.sample_neighborhood_device(k, 9000000 + seed, buf)
                    dev[outcome(buf.download(np.int32, (k, 3)))] += 1
            finally:
                buf.free()
            ref = collections.Counter()
            rng = np.random.RandomState(4242 + 10 * case + k)
            for _ in range(n):
                ref[outcome(triples[oracle.sample_edge_neighborhood(triples, V, k, rng)])] += 1
            keys = sorted(set(dev) | set(ref))
            a = np.array([dev[s] for s in keys], dtype=np.float64)
            b = np.array([ref[s] for s in keys], dtype=np.float64)
            p = (a + b) / (2 * n)
            z = np.abs(a - b) / n / np.sqrt(np.maximum(2 * p * (1 - p) / n, 1e-30))
            keep = (a + b) >= 10
            chi2, pval, dof, _ = stats.chi2_contingency(np.stack([a[keep], b[keep]]))
            assert pval > 1e-4 and z[keep].max() <= 5.0, (case, k, len(keys), float(z.max()), chi2, dof, pval)
