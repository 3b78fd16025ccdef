#!/usr/bin/env python
"""config 5's question, asked again every round: the train step replayed from a hipGraph against the same step launched on
the four streams.  RGCN_CAPTURE_STREAMS=1 (devtools library) records the stream path's own fork / join DAG instead of the
chain with two forks that rgcn_capture_begin records by default."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workloads", default="fb237_block_train_step,fb15k_block_train_step")
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--warmup", type=int, default=10)
a = ap.parse_args()
args = argparse.Namespace(gemm_mode=6, no_kernel_profile=True)
for wl in a.workloads.split(","):
    o = bench.measure_train_step(wl, args, a.steps, a.warmup)
    print(json.dumps({"workload": wl, "capture_streams": os.environ.get("RGCN_CAPTURE_STREAMS", "0"),
                      "minibatch_ms": o["minibatch_step"]["ms_per_step"],
                      "hipgraph_ms": o["captured_step"]["ms_per_step_hipgraph_replay"],
                      "stream_ms": o["captured_step"]["ms_per_step_stream_launched"],
                      "loss": o["captured_step"]["loss_after"]}), flush=True)
