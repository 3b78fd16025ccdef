#!/usr/bin/env python
"""bench.py's train-step measurement alone (minibatch step + captured step, with the per-kernel table), e.g. under an
RGCN_* knob: RGCN_DEC_TILED=0 switches the tiled-batch relation order of the decoder off.
Usage: python tools/train_step_probe.py [fb237_block_train_step|fb15k_block_train_step] [steps]"""
import argparse
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
name = sys.argv[1] if len(sys.argv) > 1 else "fb237_block_train_step"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
args = argparse.Namespace(gemm_mode=int(os.environ.get("RGCN_GEMM_MODE", "6")), no_kernel_profile=False, workload="fb237_block",
                          steps=steps, warmup=5)
t = bench.measure_train_step(name, args, steps, 5)
mb = t["minibatch_step"]
print(name, {k: v for k, v in mb.items() if k != "kernels"}, t.get("captured_step"))
for k in mb.get("kernels", []):
    print("   %-22s x%.0f %7.1f us %s frac %.3f" % (k["kernel"], k["launches_per_step"], k["avg_us"], k["bound"], k["frac"]))
