#!/bin/bash
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
OUT=$GRAFT_REPO_ROOT/gpurun_out/nbr_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp; ( cd $GRAFT_REPO_ROOT && timeout 120 rocprofv3 --kernel-trace --stats -d $OUT -o t -- python tools/nbr_sampler_trace.py 10 ) > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT; tail -3 $OUT/log.txt
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)
print(f)
for row in list(csv.DictReader(open(f[0])))[:14]:
    print({k: row[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")})
PY
rm -rf $OUT/*/ 
