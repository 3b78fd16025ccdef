"""bench.py's synthetic graphs (training-graph shapes drawn from the empirical histograms of the triples the
reference ships) -- host logic, no GPU: the specs the workloads and the full-graph parity tests name produce unique
triples with ids in range, deterministically, and the minibatch of a 'sample:' spec is a subset of its graph."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_for_graph_specs", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("spec,V,R,n", [("synth:wn18_valid_test:141442", 40943, 18, 141442),
                                        ("synth:fb237_valid_test:272115", 14541, 237, 272115),
                                        ("sample:wn18_valid_test:141442:15000", 40943, 18, 15000)])
def test_synthetic_graph_specs(bench, spec, V, R, n):
    g = bench.load_graph(spec)
    assert g.shape == (n, 3) and g.dtype == np.int32
    assert g[:, [0, 2]].min() >= 0 and g[:, [0, 2]].max() < V and g[:, 1].min() >= 0 and g[:, 1].max() < R
    key = (g[:, 0].astype(np.int64) * R + g[:, 1]) * V + g[:, 2]
    assert len(np.unique(key)) == n                                   # unique triples (SURVEY 8d graph B)
    np.testing.assert_array_equal(g, bench.load_graph(spec))          # a function of the spec alone
    if spec.startswith("sample:"):
        full = bench.load_graph("synth:wn18_valid_test:141442")
        fkey = (full[:, 0].astype(np.int64) * R + full[:, 1]) * V + full[:, 2]
        assert np.isin(key, fkey).all()                               # a minibatch OF that training graph


def test_every_workload_names_a_graph_of_its_size(bench):
    for name, (graph, V, R, d, L, kind, nb, E) in bench.WORKLOADS.items():
        if graph.startswith(("synth:", "sample:")):
            assert int(graph.split(":")[-1]) == E, name
        assert kind in ("block", "basis") and d % 4 == 0
    assert set(bench.EXTRA_WORKLOADS) <= set(bench.WORKLOADS)
    for name, (graph, V, R, kind, nb) in bench.EVALUATION_ENCODES.items():
        assert graph.startswith("synth:") and name in bench.WORKLOADS
