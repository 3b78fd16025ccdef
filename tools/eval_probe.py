#!/usr/bin/env python
"""rgcn_rank_device alone: per-kernel table + host-side split of one ranks() call (2,000 queries against 14,541 entities)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from relationprediction_amd import _native
from relationprediction_amd.common.shared_functions import init_encoder_params
V, R, d, L, nb, n = 14541, 237, 500, 1, 100, 2000
rng = np.random.RandomState(0)
eng = _native.Engine(V, R, d, L, "block", nb, max_edges=16)
eng.set_params(init_encoder_params(V, R, d, L, "block", nb, rng=np.random.RandomState(1)))
eng.set_graph(np.zeros((0, 3), np.int32)); eng.forward(train=False)
eng.rank_reserve(int(sys.argv[1]) if len(sys.argv) > 1 else 1000)
q = np.stack([rng.randint(0, V, n), rng.randint(0, R, n), rng.randint(0, V, n)], 1).astype(np.int32)
ptr = np.arange(n + 1, dtype=np.int64) * 3
idx = rng.randint(0, V, 3 * n).astype(np.int32)
for rep in range(3):
    eng.sync(); t0 = time.perf_counter()
    eng.ranks(q, True, ptr, idx)
    print("ranks() call %d: %.3f ms" % (rep, (time.perf_counter() - t0) * 1e3))
eng.profile_reset(); eng.profile_enable(True)
for rep in range(5):
    eng.ranks(q, True, ptr, idx)
eng.sync()
for k in eng.profile():
    print("   %-16s x%d  %8.1f us/launch" % (k["name"], k["calls"], 1e3 * k["total_ms"] / max(k["calls"], 1)))
eng.close()
