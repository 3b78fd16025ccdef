#!/usr/bin/env python
"""A few draws of the device neighbourhood sampler at FB15k-237 training-graph scale (for rocprofv3 --kernel-trace)."""
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
triples = bench.load_graph("synth:fb237_valid_test:272115")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
with _native.Engine(14541, 237, 4, 1, "block", 1, max_edges=30000) as eng:
    eng.neighborhood_reserve(triples)
    buf = _native.DeviceBuffer(eng, 12 * 30000)
    eng.sample_neighborhood_device(30000, 1, buf)
    eng.sync()
    t0 = time.perf_counter()
    for i in range(n):
        eng.sample_neighborhood_device(30000, 2 + i, buf)
    t1 = time.perf_counter()
    eng.sync()
    print("%.3f ms per draw (the host's share, enqueueing: %.3f ms)" % ((time.perf_counter() - t0) * 1e3 / n, (t1 - t0) * 1e3 / n))
    buf.free()
