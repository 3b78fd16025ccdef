#!/usr/bin/env python
"""Run the three GEMM shapes once each (for rocprofv3 --pmc)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relationprediction_amd import _native
V, d = 14541, 500
rng = np.random.RandomState(0)
H = np.maximum(rng.randn(V, d), 0).astype(np.float32); W = rng.randn(d, d).astype(np.float32); D = rng.randn(V, d).astype(np.float32)
eng = _native.Engine(V, 4, d, 1, "block", 100, max_edges=16, devtools=True)
print("NN", eng.debug_gemm_time(H, W, iters=5))
print("NT", eng.debug_gemm_time(D, W, trans_b=True, iters=5))
print("TN", eng.debug_gemm_time(H, D, trans_a=True, iters=5))
