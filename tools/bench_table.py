#!/usr/bin/env python
"""Readable table of a bench.py JSON line (headline workload, the workloads array, per-kernel durations)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])


def line(o):
    r = o.get("roofline") or {}
    return "%-24s %8.3f ms/step %10.2f M edges/s   top kernel %-16s %6.1f us frac %.3f" % (
        o["config"]["workload"], o["ms_per_step"], o["value"] / 1e6, r.get("kernel"), r.get("avg_us", 0), r.get("frac", 0))


print(line(d))
for o in d.get("workloads", []):
    print(line(o))
for t in d.get("train_steps", []):
    m, c = t.get("minibatch_step", {}), t.get("captured_step", {})
    print("%-24s minibatch step %7.3f ms (%.2f M edges/s, %.0f M triples/s)   captured step: hipGraph replay %7.3f ms, streams %7.3f ms" % (
        t["workload"], m.get("ms_per_step", 0), m.get("edges_per_s", 0) / 1e6, m.get("triples_per_s", 0) / 1e6,
        c.get("ms_per_step_hipgraph_replay", 0), c.get("ms_per_step_stream_launched", 0)))
if d.get("evaluation"):
    e = d["evaluation"]
    print("evaluation: encode %d-edge graph %.3f ms; ranks of %d triples, both sides: %.1f ms (%.0f triples/s)" % (
        e["graph_edges"], e["encode_full_graph_ms"], e["queries"], e["rank_both_sides_ms"], e["ranked_triples_per_s"]))
print("cpu", d.get("cpu_baseline"))
print("fp32 mfma reference", d.get("fp32_mfma_reference"))
for o in [d] + (d.get("workloads", []) if "-v" in sys.argv else []):
    print("--", o["config"]["workload"])
    for k in o["kernels"]:
        print("   %-22s x%.0f %7.1f us (pipelined %7.1f) %s frac %.3f" % (
            k["kernel"], k["launches_per_step"], k["avg_us"], k["avg_us_in_pipeline"], k["bound"], k["frac"]))
