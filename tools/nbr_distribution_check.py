#!/usr/bin/env python
"""A larger-sample version of tests/test_gpu_sampler.py::test_set_distribution_equals_the_reference_process: the
distribution of the device neighbourhood sampler's outcome SETS on the two tiny graphs (two components, triangle,
self loop, parallel edges; a cyclic graph) against the reference's loop (oracle.sample_edge_neighborhood), n draws each:
largest |z| over all outcomes and a chi-square homogeneity test (scipy).  Usage: python tools/nbr_distribution_check.py [n [graph k [first seed]]]"""
import collections
import os
import sys

import numpy as np
from scipy import stats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from relationprediction_amd import _native  # noqa: E402
from test_gpu_sampler import TINY, outcome  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
only = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else None        # [graph k]: one case only
base = int(sys.argv[4]) if len(sys.argv) > 4 else 7000000                          # first device seed
for case, (triples, V, ks) in enumerate(TINY):
    triples = triples.astype(np.int32)
    with _native.Engine(V, 1, 4, 1, "block", 1, max_edges=len(triples)) as eng:
        eng.neighborhood_reserve(triples)
        for k in ks:
            if only is not None and only != (case, k):
                continue
            buf = _native.DeviceBuffer(eng, 12 * k)
            dev = collections.Counter()
            for seed in range(n):
                eng.sample_neighborhood_device(k, base + seed, buf)
                dev[outcome(buf.download(np.int32, (k, 3)))] += 1
            buf.free()
            ref = collections.Counter()
            rng = np.random.RandomState((100 + 10 * case + k + base) % (2 ** 31))
            for _ in range(n):
                ref[outcome(triples[oracle.sample_edge_neighborhood(triples, V, k, rng)])] += 1
            keys = sorted(set(dev) | set(ref))
            a = np.array([dev[s] for s in keys], dtype=np.float64)
            b = np.array([ref[s] for s in keys], dtype=np.float64)
            p = (a + b) / (2 * n)
            z = np.abs(a - b) / n / np.sqrt(np.maximum(2 * p * (1 - p) / n, 1e-30))
            keep = (a + b) >= 10
            chi2, pval, dof, _ = stats.chi2_contingency(np.stack([a[keep], b[keep]]))
            print("graph %d, k = %d: %d outcome sets, n = %d draws each; max |z| = %.2f; chi-square %.1f on %d dof, p = %.3f"
                  % (case, k, len(keys), n, z.max(), chi2, dof, pval))
