#!/usr/bin/env python
"""What the counter-based dropout draw costs the single-pass layer kernels: the headline step's exclusive kernel durations
with keep_prob 0.8 (a draw per element of S and of dS) against keep_prob 1.0 (no draw)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from relationprediction_amd import _native  # noqa: E402
from relationprediction_amd.common.shared_functions import init_encoder_params  # noqa: E402

graph_name, V, R, d, L, kind, nb, E_g = bench.WORKLOADS["fb237_block"]
tri = bench.load_graph(graph_name)
params = init_encoder_params(V, R, d, L, kind, nb, rng=np.random.RandomState(1))
dcodes = (np.random.RandomState(2).randn(V, d) * 1e-3).astype(np.float32)
for keep in (0.8, 1.0, 0.8, 1.0):
    with _native.Engine(V, R, d, L, kind, nb, keep_prob=keep, norm_mode="intended", max_edges=E_g) as eng:
        eng.set_params(params)
        g = eng.to_device(tri)
        dc = eng.to_device(dcodes)
        eng.set_overlap(False)
        for i in range(10):
            eng.step_device(g, E_g, dc, train=True, seed=i)
        eng.sync()
        eng.profile_reset()
        eng.profile_enable(True)
        for i in range(30):
            eng.step_device(g, E_g, dc, train=True, seed=100 + i)
        eng.sync()
        prof = eng.profile()
        eng.profile_enable(False)
        rows = {p["name"]: p["total_ms"] / p["calls"] * 1e3 for p in prof}
        print("keep %.1f  " % keep + "  ".join("%s %.1f" % (k, rows[k]) for k in ("block_rows_fwd", "block_rows_bwd", "top_grad_dropout", "gemm_self_fwd") if k in rows))
