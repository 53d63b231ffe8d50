#!/usr/bin/env python
"""Long idle gaps (host stalls) of a rocprofv3 rocpd database: gaps above min_us between consecutive kernels, grouped by the pair
(kernel before, kernel after).  Usage: rocpd_longgaps.py results.db [min_us=100] [max_ms=100]"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(n):
    return re.sub(r"^void\s+", "", re.sub(r"\(.*", "", n))[:44]


def main():
    db = sqlite3.connect(sys.argv[1])
    lo = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 100e3
    hi = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 100e6
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    acc = defaultdict(lambda: [0, 0.0])
    prev_n, prev_end = rows[0][0], rows[0][2]
    for n, s, e in rows[1:]:
        g = s - prev_end
        if lo < g < hi:
            k = (short(prev_n), short(n))
            acc[k][0] += 1
            acc[k][1] += g
        if e > prev_end:
            prev_n, prev_end = n, e
    print("| kernel before | kernel after | gaps | mean us | total ms |\n|---|---|---|---|---|")
    for (a, b), (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:30]:
        print("| %s | %s | %d | %.0f | %.1f |" % (a, b, c, t / c / 1e3, t / 1e6))


if __name__ == "__main__":
    main()
