#!/bin/bash
# rocprofv3 --kernel-trace of a few draws of the device neighbourhood sampler at training-graph scale; per-kernel totals
# from the result database (gpurun_out/nbr_trace/t_results.db)
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
OUT=$GRAFT_REPO_ROOT/gpurun_out/nbr_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp; ( cd $GRAFT_REPO_ROOT && timeout 120 rocprofv3 --kernel-trace --stats -d $OUT -o t -- python tools/nbr_sampler_trace.py 10 ) > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT; grep "ms per draw" $OUT/log.txt
python - <<PY
import sqlite3
db = sqlite3.connect("$OUT/t_results.db")
rows = db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                  "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
for r in rows[:12]:
    print("%-44s calls %5d total %9.1f us avg %7.2f min %6.2f max %7.2f (%.1f%%)" % (r[0][:44], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
print("per draw (11 draws, timestamps inside a replayed graph include the dispatch gaps): %.1f us" % (tot / 11))
PY
rm -rf $OUT/*/
