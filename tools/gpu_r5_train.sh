#!/bin/bash
# round 5: the derived weight tables of a train step (fragment tables, band-tiled copies) in one launch each -- tests + the train steps
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_capture.py tests/test_gpu_parity.py tests/test_gpu_plugin.py tests/test_gpu_driver.py -m gpu -q --timeout 1200 -p no:cacheprovider -x \
  -k "not training_graph and not fb15k_training and not wn18_training and not graph_prep" > gpurun_out/pytest_r5_train.log 2>&1
echo "pytest exit $?"; tail -n 6 gpurun_out/pytest_r5_train.log
timeout 600 python tools/train_step_probe.py 2>&1 | tail -40
