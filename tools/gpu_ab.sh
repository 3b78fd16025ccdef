#!/bin/bash
# quick A/B of env knobs on the bench: usage gpu_ab.sh "ENV1=.. ENV2=.." "ENVB=.."   (full log: gpurun_out/ab.log)
mkdir -p gpurun_out; : > gpurun_out/ab.log
for cfg in "$@"; do
  echo "== $cfg" | tee -a gpurun_out/ab.log
  env $cfg python bench.py --steps 30 --warmup 5 --cpu-steps 0 --no-extra-workloads 2>&1 | python -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step %.4f  edges/s %.0f' % (o['ms_per_step'], o['value']))
for k in o['kernels'][:30]:
    print('   %-22s %5.1f/step  avg %8.2f us  %8.4f ms/step  %s %.3f' % (k['kernel'], k['launches_per_step'], k['avg_us'], k['ms_per_step'], k['bound'], k['frac']))
" >> gpurun_out/ab.log
done
grep -E "==|ms/step " gpurun_out/ab.log
