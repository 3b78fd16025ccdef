#!/usr/bin/env python
"""Per-kernel averages of every counter in a rocprofv3 --pmc results db."""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
out = {}
for k, cn, n, v in rows:
    k = re.sub(r"\(.*", "", k.replace("rgcn::(anonymous namespace)::", "").replace("void ", ""))
    k = re.sub(r"rgcn::\(anonymous namespace\)::", "", k)
    out.setdefault(k, {})[cn] = (n, v)
for k, d in out.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k[:110])
    for cn, (n, v) in sorted(d.items()):
        print("     %-28s n=%-4d avg=%.4g" % (cn, n, v))
