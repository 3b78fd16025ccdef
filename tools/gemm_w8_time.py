#!/usr/bin/env python
"""The two kernels for a pre-split weight on the B side, interleaved on the encoder's shapes: k_gemm_bf16x3<.., B_PRE>
(128x128, 4 wavefronts; RGCN_GEMM_W8=0, devtools knob) against k_gemm_w8 (128x256, 8 wavefronts, LDS-DMA)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relationprediction_amd import _native  # noqa: E402

SHAPES = [("self_fwd  H.W", 14541, 500, 500, False), ("self_dh   dS.W^T", 14541, 500, 500, True),
          ("basis fwd b2 (5370 rows)", 5370, 500, 1000, False), ("basis dz b2 (7082 rows)", 7082, 1000, 500, True),
          ("wn18 self_fwd", 40943, 500, 500, False)]
ROUNDS = int(os.environ.get("ROUNDS", "3"))
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "0,1").split(",")]
rng = np.random.RandomState(0)
with _native.Engine(16, 2, 8, 1, "block", 2, max_edges=4, devtools=True) as eng:
    eng.set_gemm_mode(6)
    for name, M, N, K, tb in SHAPES:
        A = rng.randn(M, K).astype(np.float32)
        B = rng.randn(K, N).astype(np.float32)
        Bop = np.ascontiguousarray(B.T) if tb else B
        ref = None
        times = {v: [] for v in VARIANTS}
        for rep in range(ROUNDS):
            for v in VARIANTS:
                os.environ["RGCN_GEMM_W8"] = str(v)
                out, t = eng.debug_gemm_presplit(A, Bop, trans_b=tb, iters=50)
                times[v].append(t * 1e3)
                if ref is None:
                    ref = out
                elif not np.array_equal(out, ref):
                    print("   !! variant %d differs from variant %d: max |d| %.3g at %d entries" % (
                        v, VARIANTS[0], float(np.abs(out - ref).max()), int((out != ref).sum())))
        print("%-28s " % name + "   ".join("v%d %s us" % (v, "/".join("%.1f" % t for t in times[v])) for v in VARIANTS))
