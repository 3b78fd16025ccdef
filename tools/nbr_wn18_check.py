#!/usr/bin/env python
"""The device neighbourhood sampler on a WN18-shaped graph (141,442 edges drawn from the degree histograms of the real
WN18 valid+test triples over 40,943 entities: sparse, thousands of components, long chains) against the HOST sampler
(the reference's process pick for pick): per-draw statistics of the batch -- touched vertices, connected patches,
components taken in full -- over `n` draws each, and the device's time per draw.
Usage: python tools/nbr_wn18_check.py [n] [sample_size] [graph]      (graph "wn18_valid_test": the 10,000 real triples themselves,
thousands of small components)"""
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from relationprediction_amd import _native  # noqa: E402

spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
k = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
V = 40943
triples = bench.load_graph(sys.argv[3] if len(sys.argv) > 3 else "synth:wn18_valid_test:141442")
E = len(triples)

# components of the whole graph (to count the ones a batch holds in full)
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
print("graph: %d edges, %d vertices with edges, %d components, largest %d edges"
      % (E, len(np.unique(triples[:, [0, 2]])), len(np.unique(comp[triples[:, 0]])), comp_edges.max()))


def stats(ids):
    t = triples[ids]
    touched = np.unique(t[:, [0, 2]])
    got = np.bincount(comp[t[:, 0]], minlength=V)
    present = got > 0
    full = present & (got == comp_edges)
    return len(touched), int(present.sum()), int(full.sum())


key = (triples[:, 0].astype(np.int64) * 64 + triples[:, 1]) * V + triples[:, 2]
order = np.argsort(key, kind="stable")
with _native.Engine(V, 18, 4, 1, "block", 1, max_edges=k) as eng:
    eng.neighborhood_reserve(triples)
    buf = _native.DeviceBuffer(eng, 12 * k)
    eng.sample_neighborhood_device(k, 1, buf)
    eng.sync()
    t0 = time.perf_counter()
    for i in range(20):
        eng.sample_neighborhood_device(k, 2 + i, buf)
    eng.sync()
    print("device: %.3f ms per draw of %d edges" % ((time.perf_counter() - t0) * 1e3 / 20, k))
    dev = []
    for i in range(n):
        eng.sample_neighborhood_device(k, 1000 + i, buf)
        rows = buf.download(np.int32, (k, 3))
        rk = (rows[:, 0].astype(np.int64) * 64 + rows[:, 1]) * V + rows[:, 2]
        # rows -> edge ids (rows come in edge order; duplicates of a row are told apart by position)
        ids = []
        pos = np.searchsorted(key[order], rk)
        used = {}
        for kk, p0 in zip(rk.tolist(), pos.tolist()):
            j = used.get(kk, 0)
            ids.append(int(order[p0 + j]))
            used[kk] = j + 1
        assert len(set(ids)) == k and (key[ids] == rk).all()
        dev.append(stats(np.array(ids)))
    buf.free()
host = _native.NeighborhoodSampler(triples, V)
ref = [stats(host.sample(k, 5000 + i)) for i in range(n)]
host.close()
dev, ref = np.array(dev, dtype=np.float64), np.array(ref, dtype=np.float64)
for j, name in enumerate(("touched vertices", "components present", "components taken in full")):
    se = np.sqrt((dev[:, j].var() + ref[:, j].var()) / n) + 1e-9
    print("%-26s device %.1f +- %.1f   host %.1f +- %.1f   difference %.2f standard errors"
          % (name, dev[:, j].mean(), dev[:, j].std(), ref[:, j].mean(), ref[:, j].std(),
             (dev[:, j].mean() - ref[:, j].mean()) / se))
