#!/usr/bin/env python
"""Where do the launches of a kernel come from?  For every launch whose name contains <pattern>: grid size, duration and the names of the
kernels launched just before / after it on the same queue.  Usage: rocpd_context.py results.db <pattern> [max_rows=40]"""
import sqlite3
import sys
from collections import Counter


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2]
    lim = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    q = "select name, start, end, %s from kernels order by start" % (gx or "0")
    rows = cur.execute(q).fetchall()
    seen = Counter()
    shown = 0
    for i, (name, s, e, g) in enumerate(rows):
        if pat not in name:
            continue
        prev = rows[i - 1][0][:60] if i else "-"
        nxt = rows[i + 1][0][:60] if i + 1 < len(rows) else "-"
        key = (g, prev, nxt)
        seen[key] += 1
        if seen[key] == 1 and shown < lim:
            shown += 1
            print("grid %-10s %8.1f us   after [%s]   before [%s]" % (g, (e - s) / 1e3, prev, nxt))
    print("-- %d launches, %d distinct (grid, neighbours) contexts" % (sum(seen.values()), len(seen)))
    for (g, prev, nxt), n in seen.most_common(12):
        print("%5d x  grid %-10s after [%s] before [%s]" % (n, g, prev, nxt))


if __name__ == "__main__":
    main()
