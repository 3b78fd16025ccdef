#!/usr/bin/env python
"""How the headline step's time develops over the first steps of a fresh process: chunks of 5 steps, each bracketed by a
device sync (what `bench.py --steps 20 --warmup 5` averages over is chunks 2-5)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from relationprediction_amd import _native  # noqa: E402
from relationprediction_amd.common.shared_functions import init_encoder_params  # noqa: E402

graph_name, V, R, d, L, kind, nb, E_g = bench.WORKLOADS["fb237_block"]
tri = bench.load_graph(graph_name)
pool = bench.load_graph("fb237_valid_test")
tri_b = np.ascontiguousarray(pool[np.random.RandomState(7).choice(pool.shape[0], size=E_g, replace=False)])
params = init_encoder_params(V, R, d, L, kind, nb, rng=np.random.RandomState(1))
dcodes = (np.random.RandomState(2).randn(V, d) * 1e-3).astype(np.float32)
spin = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
with _native.Engine(V, R, d, L, kind, nb, keep_prob=0.8, norm_mode="intended", max_edges=E_g) as eng:
    eng.set_params(params)
    g = [eng.to_device(tri), eng.to_device(tri_b)]
    dc = eng.to_device(dcodes)
    eng.sync()
    if spin > 0:
        time.sleep(spin)
    if len(sys.argv) > 2:      # GPU work of ANOTHER context first (devtools GEMM, ~40 ms): device warm, this context cold
        with _native.Engine(16, 2, 8, 1, "block", 2, max_edges=4, devtools=True) as e2:
            A = np.random.RandomState(0).randn(14541, 500).astype(np.float32)
            B = np.random.RandomState(1).randn(500, 500).astype(np.float32)
            for _ in range(int(sys.argv[2])):
                e2.debug_gemm_time(A, B, trans_b=False, split_k=1, iters=200)
    out = []
    i = 0
    for chunk in range(24):
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            eng.step_device(g[i % 2], E_g, dc, train=True, seed=1000 + i)
            eng.prefetch_graph_device(g[(i + 1) % 2], E_g)
            i += 1
        eng.sync()
        out.append((time.perf_counter() - t0) * 1e3 / 5)
    print("ms/step per chunk of 5:", " ".join("%.3f" % x for x in out))
