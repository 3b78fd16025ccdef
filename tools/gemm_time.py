#!/usr/bin/env python
"""Timing of the self-loop contractions (V x d x d, FB15k-237 size) through rgcn_debug_gemm_time: the three storage
forms in the default arithmetic (bf16 planes, 6 partial products) and on the fp32 MFMA."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relationprediction_amd import _native

V, d = 14541, 500
rng = np.random.RandomState(0)
H = np.maximum(rng.randn(V, d), 0).astype(np.float32)
W = (rng.randn(d, d) * 0.19).astype(np.float32)
D = (rng.randn(V, d) * 1e-3).astype(np.float32)
eng = _native.Engine(V, 2, d, 1, "block", 100, max_edges=4, devtools=True)
flops = 2.0 * V * d * d
for mode in (6, 9, 0):
    eng.set_gemm_mode(mode)
    row = []
    for name, a, b, ta, tb, sk in (("NN", H, W, False, False, 0), ("NT", D, W, False, True, 0),
                                   ("TN", H, D, True, False, 0), ("TN sk16", H, D, True, False, 16),
                                   ("TN sk64", H, D, True, False, 64)):
        ms = eng.debug_gemm_time(a, b, trans_a=ta, trans_b=tb, split_k=sk, iters=30)
        terms = mode if mode else 1
        row.append("%s %.1f us (%.0f TF exec, %.0f TF fp32-eq)" % (name, ms * 1e3, terms * flops / ms / 1e9, flops / ms / 1e9))
    print("mode %d: " % mode + "  ".join(row))
eng.close()
