#!/bin/bash
# bench.py after the measure() -> EncoderBench refactor: every mode once (default, --hipgraph, --no-kernel-profile,
# another workload), then the tests that drive bench.py (single GPU, torchrun ranks, self-spawned ranks, the shared-GPU refusal)
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for mode in "--hipgraph --no-extra-workloads --cpu-steps 0" "--no-kernel-profile --no-extra-workloads --cpu-steps 0" "--workload fb237_basis_b2 --cpu-steps 0 --no-extra-workloads"; do
  timeout 600 python bench.py --steps 20 --warmup 5 $mode > gpurun_out/bench_mode.json 2> gpurun_out/bench_mode.err
  echo "mode [$mode] exit $?"; tail -n 1 gpurun_out/bench_mode.json | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print(o['config']['workload'], o['ms_per_step'], o['value'], (o.get('roofline') or {}).get('kernel'), o['config']['step'][:60])"
done
timeout 1500 python -m pytest tests/test_gpu_multiprocess.py -q -x --timeout 900 -p no:cacheprovider > gpurun_out/pytest_bench.log 2>&1
echo "pytest exit $?"; tail -n 4 gpurun_out/pytest_bench.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_refactor.json 2> gpurun_out/bench_refactor.err
echo "bench exit $?"; tail -n 1 gpurun_out/bench_refactor.json | python -c "
import json,sys
s=sys.stdin.read(); o=json.loads(s); print(len(s), o['ms_per_step'], o['value'], o['steady_state']['ms_per_step'], o['roofline']['kernel'], o['roofline']['frac'], o['cpu_baseline']['value'], [w['ms_per_step'] for w in o['workloads']])"
