#!/usr/bin/env python
"""Summarise the per-wavefront timeline of k_dec_entity_lines (persistent form) written by a -DDEC_TRACE build
(RGCN_DEC_TRACE=<file>): per wavefront [start, (after fill, after pieces, end of pass) x 2, end, xcc | band << 16, blocks]."""
import struct, sys
import numpy as np
data = open(sys.argv[1], "rb").read()
pos, launches = 0, []
while pos < len(data):
    n, grid, _, _ = struct.unpack_from("4q", data, pos); pos += 32
    a = np.frombuffer(data, dtype=np.uint64, count=n, offset=pos).reshape(-1, 12).astype(np.int64); pos += 8 * n
    launches.append((grid, a))
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(launches) - 1
grid, a = launches[which]
t0 = a[:, 0].min()
us = lambda c: (a[:, c] - t0) / 100.0
st, en = us(0), us(7)
print("launch %d of %d: grid %d; wavefronts %d; span %.1f us" % (which, len(launches), grid, len(a), en.max()))
print(" start min/med/max %.1f %.1f %.1f   end min/med/max %.1f %.1f %.1f   blocks per wavefront min/med/max %d %d %d" % (
    st.min(), np.median(st), st.max(), en.min(), np.median(en), en.max(), a[:, 9].min(), np.median(a[:, 9]), a[:, 9].max()))
for p in range(2):
    f, pc, e = us(1 + 3 * p), us(2 + 3 * p), us(3 + 3 * p)
    prev = st if p == 0 else us(3)
    print(" pass %d: fill (incl. barrier) med/max %.2f %.2f   pieces med/max %.2f %.2f   turns med/p90/max %.2f %.2f %.2f   pass ends min/med/max %.1f %.1f %.1f" % (
        p, np.median(f - prev), (f - prev).max(), np.median(pc - f), (pc - f).max(), np.median(e - pc), np.percentile(e - pc, 90), (e - pc).max(),
        e.min(), np.median(e), e.max()))
print(" us per block (whole wavefront) med %.3f" % np.median((en - st) / np.maximum(a[:, 9], 1)))
xcc = a[:, 8] & 0xff; band = a[:, 8] >> 16
tab = np.zeros((8, 9), dtype=int)
np.add.at(tab, (band, np.minimum(xcc, 8)), 1)
print(" first band x xcc table:\n", tab)
