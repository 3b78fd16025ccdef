#!/bin/bash
# round 6, late changes in one box visit: rgcn_device_info + the shared-GPU refusal, Metric=Accuracy end to end, the
# optimizer's norm kernels (16-byte loads, batched partial fetches) -- tests, then the default bench for the opt_* rows
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_gpu_multiprocess.py tests/test_gpu_driver.py tests/test_gpu_train_step.py tests/test_gpu_capture.py tests/test_abi.py \
  -q -x --timeout 900 -p no:cacheprovider > gpurun_out/pytest_late.log 2>&1
echo "pytest exit $?"; tail -n 5 gpurun_out/pytest_late.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_late.json 2> gpurun_out/bench_late.err
echo "bench exit $?"; cp bench_details.json gpurun_out/bench_late_details.json
python - <<PY
import json
d = json.load(open("gpurun_out/bench_late_details.json"))
print("headline", d["ms_per_step"], "steady", d["steady_state"])
for t in d["train_steps"]:
    m = t["minibatch_step"]
    print(t["workload"], m["ms_per_step"], [(k["kernel"], k["avg_us"]) for k in m["kernels"] if k["kernel"].startswith("opt")])
PY
