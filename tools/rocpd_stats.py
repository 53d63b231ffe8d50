#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (the default output of `rocprofv3 --kernel-trace --stats`) into the per-kernel
table the judge reads: name, calls, total ms, avg us, % of GPU kernel time.  Usage: rocpd_stats.py results.db [out.md]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s "
                       "order by sum(end-start) desc" % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, c, s, a, mn, mx in rows:
        short = re.sub(r"\(.*", "", n)[:90]
        lines.append("| %s | %d | %.3f | %.2f | %.2f | %.2f | %.2f |" % (short, c, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / total))
    lines.append("| TOTAL | %d | %.3f | | | | 100 |" % (sum(r[1] for r in rows), total / 1e6))
    txt = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
