#!/usr/bin/env python
"""RCCL load + world-1 communicator + all-reduce in a process that never imports torch (what bench.py's ranks are).
    python tools/rccl_selfcheck.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from relationprediction_amd import _native  # noqa: E402

assert "torch" not in sys.modules
eng = _native.Engine(64, 3, 8, 1, "block", 2, max_edges=4)
uid = _native.Engine.comm_unique_id()
eng.comm_init(uid)
x = np.arange(1 << 20, dtype=np.float32)
buf = eng.to_device(x)
for _ in range(3):
    eng.comm_allreduce_sum(buf, x.size)
eng.sync()
assert np.array_equal(buf.download(np.float32, x.shape), x)
buf.free()
eng.close()
mapped = [l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l]
print("ok; torch imported:", "torch" in sys.modules, "; librccl mapped from:", sorted(set(mapped)))
