#!/usr/bin/env python
"""Full-graph shape check (SURVEY 8d config 2 'graph B'): synthetic 272,115-edge FB15k-237-sized graph with
Zipf relation / entity popularity (big hub rows).  Forward (test mode) against the oracle, then timing."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from relationprediction_amd import _native
from relationprediction_amd.common.shared_functions import init_encoder_params

V, R, d, L, nb = 14541, 237, 500, 2, 100
E = int(sys.argv[1]) if len(sys.argv) > 1 else 272115
rng = np.random.RandomState(0)
tri = oracle.synthetic_graph(V, R, E, rng)
cnt = np.bincount(tri[:, 0], minlength=V) + np.bincount(tri[:, 2], minlength=V)
print("E", E, "max row", cnt.max(), "rows>32", (cnt > 32).sum(), "top relation share %.3f" % (np.bincount(tri[:, 1]).max() / E))
params = init_encoder_params(V, R, d, L, "block", nb, rng=np.random.RandomState(1))
eng = _native.Engine(V, R, d, L, "block", nb, max_edges=E)
eng.set_params(params)
eng.set_graph(tri)
eng.forward(train=False)
codes = eng.codes()
assert np.isfinite(codes).all()
if os.environ.get("SKIP_ORACLE") != "1":
    t0 = time.time()
    ref = oracle.encoder_forward(params, tri, V, L, "block", mode="test")[-1]
    print("oracle forward %.1f s; max abs err %.3e (scale %.3e)" % (time.time() - t0, np.abs(codes - ref).max(), np.abs(ref).max()))
dc = eng.to_device((np.random.RandomState(2).randn(V, d) * 1e-3).astype(np.float32))
td = eng.to_device(tri)
for name, fn in (("prep+fwd+bwd", lambda i: eng.step_device(td, E, dc, train=True, seed=i)),):
    for i in range(3): fn(i)
    eng.sync(); eng.timer_start()
    for i in range(10): fn(i)
    ms = eng.timer_stop() / 10
    print("%s: %.3f ms/step  %.1f M edges/s" % (name, ms, E / ms / 1e3))
eng.set_graph_device(td, E)
for i in range(3): eng.forward(train=False)
eng.sync(); eng.timer_start()
for i in range(10): eng.forward(train=False)
ms = eng.timer_stop() / 10
print("forward only (static graph, test mode): %.3f ms  %.1f M edges/s" % (ms, E / ms / 1e3))
eng.profile_reset(); eng.set_overlap(False); eng.profile_enable(True)
for i in range(5): eng.step_device(td, E, dc, train=True, seed=i)
for p in sorted(eng.profile(), key=lambda p: -p["total_ms"])[:12]:
    print("   %-22s %7.1f us x %d" % (p["name"], p["total_ms"] / p["calls"] * 1e3, p["calls"] // 5))
eng.close()
