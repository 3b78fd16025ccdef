#!/usr/bin/env python
"""For rocprofv3 --pmc: a few launches of the pre-split-weight GEMM kernels on H.W_self (variants: RGCN_GEMM_W8 codes)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relationprediction_amd import _native
M, N, K = 14541, 500, 500
rng = np.random.RandomState(0)
A = rng.randn(M, K).astype(np.float32); B = rng.randn(K, N).astype(np.float32)
with _native.Engine(16, 2, 8, 1, "block", 2, max_edges=4, devtools=True) as eng:
    eng.set_gemm_mode(6)
    for v in [int(x) for x in os.environ.get("VARIANTS", "0,1").split(",")]:
        os.environ["RGCN_GEMM_W8"] = str(v)
        _, ms = eng.debug_gemm_presplit(A, B, iters=10)
        print("variant", v, "%.1f us" % (ms * 1e3))
