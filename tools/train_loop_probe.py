#!/usr/bin/env python
"""bench.py's train_loop measurement alone (host sampler with 0 / 8 builder threads, device sampler), e.g. under
RGCN_PF_PRIO=1 or another RGCN_* knob.  Usage: python tools/train_loop_probe.py [iterations]"""
import argparse
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
args = argparse.Namespace(gemm_mode=int(os.environ.get("RGCN_GEMM_MODE", "6")))
it = int(sys.argv[1]) if len(sys.argv) > 1 else 150
out = bench.measure_train_loop(args, float(os.environ.get("DEVICE_STEP_MS", "1.2")), iterations=it)
print(json.dumps({k: out[k] for k in ("ms_per_iteration_by_builder_threads", "ms_per_iteration_device_sampler",
                                      "host_batch_build_ms")}))
