#!/bin/bash
# the evaluation tests + bench.py's evaluation measurement alone
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_eval.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -3
python - <<'PY'
import argparse, importlib.util, os, sys, json
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
args = argparse.Namespace(gemm_mode=int(os.environ.get("RGCN_GEMM_MODE", "6")), no_kernel_profile=False, workload="fb237_block", steps=20, warmup=5)
for _ in range(2):
    print(json.dumps(bench.measure_evaluation(args)))
PY
