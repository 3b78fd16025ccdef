#!/usr/bin/env python
"""Per step of a rocprofv3 kernel_trace.csv of the encoder step loop (marker: k_colsum_final closes a step): the step's
span, the busy time of the main queue, its idle time inside the span, and the mean duration of a few named kernels --
which of them changes between the first steps of a process and its steady state."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in rows)
marks = [i for i, k in enumerate(ks) if "k_colsum_final" in k[3]]
names = ["k_gemm_bf16x3<true", "k_gemm_bf16x3<false", "k_block_rows<5, false", "k_block_rows<5, true", "k_block_msg_bwd",
         "k_input_fwd", "k_sort_scatter"]
print("step   span   mainbusy mainidle | " + " ".join("%-12s" % n[-12:] for n in names))
for n in range(1, len(marks)):
    lo, hi = marks[n - 1] + 1, marks[n] + 1
    step = ks[lo:hi]
    t0 = ks[marks[n - 1]][1]
    mainq = ks[marks[n]][2]
    main = [(s, e) for s, e, q, _ in step if q == mainq]
    busy = sum(e - s for s, e in main)
    span = step[-1][1] - t0
    cols = []
    for nm in names:
        d = [e - s for s, e, q, k in step if nm in k]
        cols.append("%-12s" % ("%.1f" % (sum(d) / len(d) / 1e3) if d else "-"))
    print("%4d %7.1f %8.1f %8.1f | %s" % (n, span / 1e3, busy / 1e3, (span - busy) / 1e3, " ".join(cols)))
