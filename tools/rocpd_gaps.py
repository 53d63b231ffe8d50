#!/usr/bin/env python
"""Idle-gap analysis of a rocprofv3 rocpd database: for every kernel launch, the idle time between the previous kernel's end and
this kernel's start (same device, time-ordered).  Prints the busy fraction and, per kernel name, the mean gap that PRECEDES it.
Usage: rocpd_gaps.py results.db [max_gap_us=200]   (gaps above max_gap_us are counted as host stalls, listed separately)"""
import re
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    cap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 200e3
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    busy = sum(e - s for _, s, e in rows)
    span = rows[-1][2] - rows[0][1]
    gaps = defaultdict(lambda: [0, 0.0])
    small = big = 0.0
    nbig = 0
    prev_end = rows[0][2]
    for n, s, e in rows[1:]:
        g = max(0, s - prev_end)
        prev_end = max(prev_end, e)
        if g > cap:
            big += g
            nbig += 1
            continue
        small += g
        k = re.sub(r"\(.*", "", n)[:70]
        gaps[k][0] += 1
        gaps[k][1] += g
    print("kernels %d  span %.1f ms  busy %.1f ms (%.1f%%)  short gaps %.1f ms  long gaps (> %.0f us) %.1f ms in %d" %
          (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, small / 1e6, cap / 1e3, big / 1e6, nbig))
    print("| kernel | launches | mean gap before (us) | total gap ms |\n|---|---|---|---|")
    for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
        print("| %s | %d | %.2f | %.2f |" % (k, c, t / c / 1e3, t / 1e6))


if __name__ == "__main__":
    main()
