#!/bin/bash
# cycles (SQ_BUSY_CYCLES / 32 shader engines) and durations of the ablation variants: what is time and what is clock
export TMPDIR=/tmp
TAG=${1:-a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/w8_pmc2_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/p1 -o p1 -- python tools/gemm_w8_pmc.py > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python tools/gemm_w8_pmc.py > $OUT/kt.log 2>&1
python - <<PY | tee gpurun_out/w8_pmc2_$TAG.txt
import re, sqlite3
def q(db, sql):
    return sqlite3.connect(db).execute(sql).fetchall()
c = sqlite3.connect("$OUT/kt/kt_results.db")
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
dur = {r[0]: r[1] for r in c.execute("select s.kernel_name, avg(d.end-d.start) from %s d join %s s on d.kernel_id=s.id group by s.kernel_name" % (kd, ks))}
rows = q("$OUT/p1/p1_results.db", "select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name")
pm = {}
for k, cn, v in rows:
    pm.setdefault(k, {})[cn] = v
def short(k):
    m = re.search(r"k_gemm_w8<6, (\d+)>|k_gemm_w8ILi6ELi(\d+)E", k)
    return ("w8 DBG %s" % (m.group(1) or m.group(2))) if m else ("old" if "k_gemm_bf16x3" in k else None)
dd = {short(k): v for k, v in dur.items() if short(k)}
print("%-12s %8s %10s %7s %9s %9s %9s %9s" % ("kernel", "us", "cycles", "GHz", "mfma_busy", "wait_any", "wait_inst", "active"))
for k, d in sorted(pm.items(), key=lambda kv: kv[0]):
    s = short(k)
    if not s: continue
    cyc = d["SQ_BUSY_CYCLES"] / 32.0
    wc = d["SQ_WAVE_CYCLES"]
    us = dd.get(s, 0) / 1e3
    print("%-12s %8.1f %10.0f %7.2f %9.3f %9.3f %9.3f %9.3f" % (s, us, cyc, cyc / us / 1e3 if us else 0, d["SQ_VALU_MFMA_BUSY_CYCLES"] / 912.0 / cyc,
          d["SQ_WAIT_ANY"] / wc, d["SQ_WAIT_INST_ANY"] / wc, d["SQ_ACTIVE_INST_ANY"] / wc))
PY
