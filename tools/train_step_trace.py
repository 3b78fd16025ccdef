#!/usr/bin/env python
"""Device train steps at FB15k-237 gcn_block size on fixed device-resident inputs (no host work between steps),
for a rocprofv3 --kernel-trace timeline of rgcn_train_step_device:
    rocprofv3 --kernel-trace -d out -o trace -- python tools/train_step_trace.py [steps]
and tools/train_step_timeline.py out/...kernel_trace.csv prints one steady-state step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from relationprediction_amd import _native  # noqa: E402
from relationprediction_amd.common.shared_functions import init_encoder_params  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
fb15k = len(sys.argv) > 2 and sys.argv[2] == "fb15k"          # BASELINE configs[4]'s entity / relation space
V, R, d, L, nb, E = (14951, 1345, 500, 2, 100, 15000) if fb15k else (14541, 237, 500, 2, 100, 15000)
rng = np.random.RandomState(0)
with np.load(os.path.join(ROOT, "tests", "golden", "graphs.npz")) as z:
    if fb15k:
        triples = np.ascontiguousarray(z["fb15k_minibatch"].astype(np.int32))
        batch = np.concatenate([triples, np.stack([rng.randint(0, V, 15000), rng.randint(0, R, 15000),
                                                   rng.randint(0, V, 15000)], 1).astype(np.int32)])
    else:
        triples = np.ascontiguousarray(z["fb237_minibatch"].astype(np.int32))
        pool = z["fb237_valid_test"].astype(np.int32)
        batch = np.concatenate([triples, pool[rng.choice(len(pool), 15000, replace=False)]])
batch = np.ascontiguousarray(batch.astype(np.int32))
N = len(batch) * 11                      # NegativeSampleRate 10: drawn on the device, as the driver does
params = init_encoder_params(V, R, d, L, "block", nb, rng=np.random.RandomState(1))
params["W_relation"] = np.random.RandomState(2).randn(V, d).astype(np.float32)
eng = _native.Engine(V, R, d, L, "block", nb, keep_prob=0.8, max_edges=E)
if os.environ.get("RGCN_STREAMS", "1") == "0":      # this script's own switch: every kernel on the main stream
    eng.set_overlap(False)
eng.set_params(params)
eng.decoder_reserve(N)
eng.optimizer_config(lr=0.01, max_grad_norm=1.0)
T, Bd, Xd, Yd = eng.to_device(triples), eng.to_device(batch), eng.alloc(12 * N), eng.alloc(4 * N)
eng.negative_sample_device(Bd, len(batch), 10, 5, Xd, Yd)
for i in range(steps):
    eng.train_step_device(T, E, Xd, Yd, N, seed=i, reg_param=0.01)
eng.sync()
import time  # noqa: E402
t0 = time.perf_counter()
for i in range(3):                        # the host's share: three steps enqueued into an empty queue, no wait
    eng.train_step_device(T, E, Xd, Yd, N, seed=50 + i, reg_param=0.01)
t1 = time.perf_counter()
eng.sync()
print("host enqueue: %.3f ms per train step (3 steps into an idle queue; device time %.3f ms per step incl. the wait)"
      % ((t1 - t0) * 1e3 / 3, (time.perf_counter() - t0) * 1e3 / 3))
eng.timer_start()
for i in range(steps):
    eng.train_step_device(T, E, Xd, Yd, N, seed=100 + i, reg_param=0.01)
ms = eng.timer_stop()
print("train step: %.3f ms (N = %d, E_g = %d), loss %.4f" % (ms / steps, N, E, eng.loss()))
for b in (T, Bd, Xd, Yd):
    b.free()
eng.close()
