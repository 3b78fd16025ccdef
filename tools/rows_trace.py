#!/usr/bin/env python
"""Summarise the per-wavefront timeline RGCN_ROWS_TRACE=<file> makes k_block_rows write (diagnostic build)."""
import struct, sys
import numpy as np
data = open(sys.argv[1], "rb").read()
pos, launches = 0, []
while pos < len(data):
    n, grid, nlong, bwd = struct.unpack_from("4q", data, pos); pos += 32
    a = np.frombuffer(data, dtype=np.uint64, count=n, offset=pos).reshape(-1, 6); pos += 8 * n
    launches.append((grid, nlong, bwd, a))
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(launches) - 2
grid, nlong, bwd, a = launches[which]
ok = a[:, 1] > 0
t0 = a[ok, 0].min()
st = (a[:, 0].astype(np.int64) - int(t0)) / 100.0      # wall_clock64: 100 MHz -> us
en = (a[:, 1].astype(np.int64) - int(t0)) / 100.0
role = (a[:, 2] & 0xff).astype(int); xcc = (a[:, 2] >> 8).astype(int)
print("launch %d of %d: grid %d, long wgs/band %d, bwd %d; waves that ran %d of %d; span %.1f us" % (which, len(launches), grid, nlong, bwd, ok.sum(), len(a), en[ok].max()))
wg = np.arange(len(a)) // 4
for r, name in ((1, "long"), (0, "short")):
    m = ok & (role == r)
    if not m.any(): continue
    d = en[m] - st[m]
    print(" %-5s waves %5d: start min/med/max %.1f %.1f %.1f   end med/max %.1f %.1f   duration med/p90/max %.1f %.1f %.1f   sum %.0f us" % (
        name, m.sum(), st[m].min(), np.median(st[m]), st[m].max(), np.median(en[m]), en[m].max(), np.median(d), np.percentile(d, 90), d.max(), d.sum()))
    ini, wt = a[m, 3].astype(np.int64) / 100.0, a[m, 4].astype(np.int64) / 100.0
    print("       init wait med/p90 %.2f %.2f   data wait (sum per wave) med/p90 %.2f %.2f   rest med %.2f" % (
        np.median(ini), np.percentile(ini, 90), np.median(wt), np.percentile(wt, 90), np.median(d - ini - wt)))
# band -> xcc mapping check
band = (wg % 8)
tab = np.zeros((8, 16), dtype=int)
np.add.at(tab, (band[ok], xcc[ok]), 1)
print(" band x xcc table:\n", tab[:, :9])
# occupancy over time
ts = np.linspace(0, en[ok].max(), 21)
print(" waves resident at t:", " ".join("%d" % ((st[ok] <= t) & (en[ok] > t)).sum() for t in ts))
