#!/usr/bin/env python
"""Time the fp32-MFMA GEMM variants on the encoder's three contraction shapes (one process per variant:
the variant is latched from RGCN_GEMM_VARIANT at first use)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import sys, json, numpy as np
sys.path.insert(0, %r)
from relationprediction_amd import _native
V, d = 14541, 500
rng = np.random.RandomState(0)
H = rng.randn(V, d).astype(np.float32); W = rng.randn(d, d).astype(np.float32); D = rng.randn(V, d).astype(np.float32)
eng = _native.Engine(V, 4, d, 1, "block", 100, max_edges=16, devtools=True)
out = {}
ref = H[:256].astype(np.float64) @ W.astype(np.float64)
got = eng.debug_gemm(H[:256], W)
out["err"] = float(np.abs(got - ref).max() / np.abs(ref).max())
out["NN"] = eng.debug_gemm_time(H, W, iters=30)
out["NT"] = eng.debug_gemm_time(D, W, trans_b=True, iters=30)
for sk in (0, 8, 16, 32):
    out["TN_sk%%d" %% sk] = eng.debug_gemm_time(H, D, trans_a=True, split_k=sk, iters=30)
print(json.dumps(out))
""" % ROOT

if __name__ == "__main__":
    flops = 2.0 * 14541 * 500 * 500
    for swz in (1,):
        for var, abl in ((0, 0), (0, 1), (0, 2), (0, 4), (0, 8), (0, 9), (0, 11), (0, 15)):
            env = dict(os.environ, RGCN_GEMM_VARIANT=str(var), RGCN_GEMM_SWIZZLE=str(swz), RGCN_GEMM_ABLATE=str(abl))
            r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
            if r.returncode != 0:
                print("variant", var, "swizzle", swz, "FAILED", r.stderr[-500:])
                continue
            o = json.loads(r.stdout.strip().splitlines()[-1])
            line = "variant %d ablate %d err %.1e :" % (var, abl, o.pop("err"))
            for k, ms in o.items():
                line += "  %s %.1f us (%.0f TF)" % (k, ms * 1e3, flops / (ms * 1e-3) / 1e12)
            print(line, flush=True)
