"""Is the split-bf16 GEMM's rounding error biased?  Element by element it equals the fp32 MFMA's (tools/gemm_modes.py),
but tests/test_gpu_parity.py::test_float64_tie_break found the basis-coefficient gradient -- a sum of ~10^5..10^6
outputs of the dZ GEMM weighted by non-negative activations -- 10x further from float64 in mode 6 than in mode 0.
This prints, for dZ = D . W'^T at BASELINE config 3's shape, per GEMM mode: the rms and the MEAN of the signed error,
its projection on the exact result (a shrink / stretch factor), and the error of non-negative-weighted sums."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relationprediction_amd import _native as native

V, d, B = 14541, 500, 2
rng = np.random.RandomState(7)
D = (rng.randn(V, d) * 0.01).astype(np.float32)
W = (rng.randn(2 * B * d, d) * 0.06).astype(np.float32)
H = np.maximum(rng.randn(V, 2 * B * d), 0).astype(np.float32)
ref = D.astype(np.float64) @ W.astype(np.float64).T
eng = native.Engine(V, 2, d, 1, "block", 100, max_edges=4, devtools=True)
try:
    for mode in (0, 6, 9):
        eng.set_gemm_mode(mode)
        got = eng.debug_gemm(D, W, trans_b=True).astype(np.float64)
        err = got - ref
        rms, mean = np.sqrt((err ** 2).mean()), err.mean()
        proj = (err * ref).sum() / (ref * ref).sum()
        sgn = (err * np.sign(ref)).mean()
        f_true = (H * ref).sum(axis=0)          # 2000 functionals, one per output column, over V rows
        f_err = (H * err).sum(axis=0)
        expect = rms * np.sqrt((H ** 2).sum(axis=0))       # what independent zero-mean errors of that rms would leave
        print("mode %d: rms err %.3e (rel %.2e)  mean err %+.3e (= %+.4f rms)  mean err.sign(ref) %+.3e (= %+.4f rms)  "
              "projection on ref %+.3e" % (mode, rms, rms / np.sqrt((ref ** 2).mean()), mean, mean / rms, sgn, sgn / rms, proj))
        print("        non-negative-weighted column sums over %d rows: |error| / (independent-error expectation): median %.2f  max %.2f;"
              "  l2 rel error of the sums %.2e" % (V, np.median(np.abs(f_err) / expect), (np.abs(f_err) / expect).max(),
                                                    np.sqrt((f_err ** 2).sum() / (f_true ** 2).sum())))
        # same, weights and errors regrouped the way the coefficient gradient does: dot over the 500 features of a block
        blk = (H * err).reshape(V, 2 * B, d).sum(axis=(0, 2))
        blk_true = (H * ref).reshape(V, 2 * B, d).sum(axis=(0, 2))
        print("        per-basis sums (V x d terms each): rel error", np.abs(blk / blk_true))
finally:
    eng.close()
