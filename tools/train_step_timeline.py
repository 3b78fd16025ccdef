#!/usr/bin/env python
"""Print the kernel timeline of the last complete train step in a rocprofv3 kernel_trace.csv: start offset,
duration, queue, kernel; then idle gaps of the union of all queues."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", r.get("Stream_Id", "?")),
       r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("rgcn::", "")
       .replace("rocprim::ROCPRIM_400200_NS::detail::", "rocprim ").split("(")[0][:70]) for r in rows]
ks.sort()
marker = sys.argv[2] if len(sys.argv) > 2 else "k_adam"       # last kernel of a step
adam = [i for i, k in enumerate(ks) if marker in k[3]]
if len(sys.argv) > 3:                                          # an earlier step (1 = last, 2 = the one before ...)
    adam = adam[:len(adam) - int(sys.argv[3]) + 1]
lo, hi = adam[-2] + 1, adam[-1] + 1
step = ks[lo:hi]
t0 = ks[adam[-2]][1]
print("step wall: %.1f us, %d kernels" % ((step[-1][1] - t0) / 1e3, len(step)))
busy_end = t0
idle = 0.0
for s, e, q, name in step:
    gap = (s - busy_end) / 1e3
    if gap > 0:
        idle += gap
    print("%8.1f  %7.1f us  q%-3s %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name,
                                           "   <-- idle %.1f us before" % gap if gap > 3 else ""))
    busy_end = max(busy_end, e)
print("idle (no kernel on any queue): %.1f us" % idle)
