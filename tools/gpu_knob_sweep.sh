#!/bin/bash
# sweep of launch-geometry knobs that only the weight-gradient kernels use since form 3 is the default
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
run() { env "$@" timeout 300 python bench.py --steps 30 --warmup 6 --cpu-steps 0 --no-extra-workloads --no-fp32-reference --no-live-traffic > /dev/null 2>&1; python - "$*" <<'PY'
import json, sys
d = json.load(open("bench_details.json"))
ks = {k["kernel"]: k for k in d["kernels"]}
print("%-44s %.4f ms/step  sum-excl %.4f  dw_msgs %.1f dw_reduce %.1f gemm_dw %.1f splitk %.1f" % (sys.argv[1], d["ms_per_step"], d["step_roofline"]["sum_exclusive_kernel_ms"],
      ks["block_dw_msgs"]["avg_us"], ks["block_dw_reduce"]["avg_us"], ks["gemm_self_dw"]["avg_us"], ks["splitk_reduce"]["avg_us"]))
PY
}
run RGCN_NOP=1
for ch in 32 64 96 128 192; do run RGCN_CHUNK=$ch; done
run RGCN_MSG_BLOCK=256
run RGCN_MSG_BLOCK=256 RGCN_CHUNK=96
for sk in 256 384 768 1024; do run RGCN_SPLITK_TARGET=$sk; done
run RGCN_NOP=2
