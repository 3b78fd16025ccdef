#!/usr/bin/env python
"""k_gemm_w8 (csrc/gemm_bf16x3_w8.hip) under the knife: ablations (RGCN_GEMM_W8 = 1000 + DBG bits, devtools build) and the
per-wavefront s_memtime timeline of one launch, on the self-loop product H.W_self (14,541 x 500 x 500).

DBG bits: 1 no MFMA, 2 no split of A (no VALU, no ds_write), 4 no fragment reads, 8 no LDS-DMA of B, 16 no loads of A,
32 no stores of C, 64 coarse timeline, 128 one step in detail."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relationprediction_amd import _native  # noqa: E402

M, N, K = [int(x) for x in os.environ.get("SHAPE", "14541,500,500").split(",")]
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "0,1,1256,1001,1002,1004,1008,1016,1032,1003,1007,1024,1026,1031,1063").split(",")]
NAMES = {0: "old 128x128 B_PRE", 1: "w8", 1001: "w8 - MFMA", 1002: "w8 - split", 1004: "w8 - frag reads", 1008: "w8 - DMA(B)",
         1016: "w8 - loads(A)", 1032: "w8 - stores(C)", 1003: "w8 - MFMA - split", 1007: "w8 - MFMA - split - frag",
         1024: "w8 - DMA - loads(A)", 1026: "w8 - DMA - loads(A) - split", 1031: "w8: loop + barriers + epilogue only",
         1063: "w8: nothing but the skeleton", 1256: "w8, nt stores", 1015: "w8 - MFMA - split - frag - DMA"}
ROUNDS = int(os.environ.get("ROUNDS", "3"))
TL = 24
rng = np.random.RandomState(0)
A = rng.randn(M, K).astype(np.float32)
B = rng.randn(K, N).astype(np.float32)


def timeline(eng, code, label):
    path = "/tmp/w8_tl_%d.bin" % code
    os.environ["RGCN_GEMM_W8"] = str(code)
    os.environ["RGCN_GEMM_TL_FILE"] = path
    eng.debug_gemm_presplit(A, B, iters=0)
    del os.environ["RGCN_GEMM_TL_FILE"]
    raw = np.fromfile(path, dtype=np.uint64).reshape(-1, 8, TL)
    live = raw[:, :, 0] != 0
    wgs = np.where(live.all(axis=1))[0]
    t = raw[wgs].astype(np.int64)                      # [wg, wave, slot]
    t0 = t[:, :, 0].min()
    print("== %s: %d workgroups; all times in cycles of s_memtime (100 MHz?) relative to the first wavefront's start" % (label, len(wgs)))
    return t, t0


with _native.Engine(16, 2, 8, 1, "block", 2, max_edges=4, devtools=True) as eng:
    eng.set_gemm_mode(6)
    times = {v: [] for v in VARIANTS}
    for rep in range(ROUNDS):
        for v in VARIANTS:
            os.environ["RGCN_GEMM_W8"] = str(v)
            if rep == 0:
                print("running variant", v, flush=True)
            _, ms = eng.debug_gemm_presplit(A, B, iters=40)
            times[v].append(ms * 1e3)
    for v in VARIANTS:
        print("%-44s %s us" % (NAMES.get(v, str(v)), "  ".join("%6.1f" % x for x in times[v])))
    if os.environ.get("TIMELINE", "1") != "0":
        t, t0 = timeline(eng, 1064, "coarse timeline")
        names = ["entry", "fill done", "step 4", "step 8", "step 16", "step 24", "step 32", "drained", "barrier", "end"]
        for i, nm in enumerate(names):
            col = t[:, :, i] - t0
            print("  %-10s min %8d  median %8d  max %8d" % (nm, col.min(), int(np.median(col)), col.max()))
        d = np.diff(t[:, :, :10], axis=2)
        print("  segment medians:", "  ".join("%s %d" % (names[i + 1], int(np.median(d[:, :, i]))) for i in range(9)))
        print("  per-wavefront span (end - entry): median %d, min %d, max %d" % (
            int(np.median(t[:, :, 9] - t[:, :, 0])), (t[:, :, 9] - t[:, :, 0]).min(), (t[:, :, 9] - t[:, :, 0]).max()))
        t, t0 = timeline(eng, 1128, "step 12 in detail")
        names = ["top", "MFMA 7 (loads issued)", "MFMA 10 (before vmcnt)", "after vmcnt(8)", "MFMA 5", "MFMA 23", "lgkmcnt(0)", "after barrier"]
        idx = [12, 13, 14, 15, 16, 17, 18, 19]
        base = t[:, :, 12]
        for nm, i in zip(names, idx):
            col = t[:, :, i] - base
            print("  %-24s median %6d  p10 %6d  p90 %6d" % (nm, int(np.median(col)), int(np.percentile(col, 10)), int(np.percentile(col, 90))))
