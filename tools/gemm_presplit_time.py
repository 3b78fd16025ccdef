#!/usr/bin/env python
"""Staged against pre-split-B GEMM (gemm_bf16x3.hip B_PRE) on the encoder's shapes: mean time of one product."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relationprediction_amd import _native  # noqa: E402

SHAPES = [("self_fwd  H.W", 14541, 500, 500, False), ("self_dh   dS.W^T", 14541, 500, 500, True),
          ("basis fwd b2 (5370 rows)", 5370, 500, 1000, False), ("basis dz b2 (7082 rows)", 7082, 1000, 500, True),
          ("wn18 self_fwd", 40943, 500, 500, False)]
rng = np.random.RandomState(0)
with _native.Engine(16, 2, 8, 1, "block", 2, max_edges=4, devtools=True) as eng:
    for mode in (6,):
        eng.set_gemm_mode(mode)
        for name, M, N, K, tb in SHAPES:
            A = rng.randn(M, K).astype(np.float32)
            B = rng.randn(K, N).astype(np.float32)
            Bop = np.ascontiguousarray(B.T) if tb else B
            for rep in range(2):
                t0 = eng.debug_gemm_time(A, Bop, trans_b=tb, split_k=1, iters=50)
                _, t1 = eng.debug_gemm_presplit(A, Bop, trans_b=tb, iters=50)
                print("mode %d %-28s staged %7.1f us   pre-split %7.1f us   (%.3f)" % (mode, name, t0 * 1e3, t1 * 1e3, t1 / t0))
