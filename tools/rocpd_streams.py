#!/usr/bin/env python
"""Per-stream kernel totals of a rocprofv3 rocpd database (which stream each kernel family ran on): separates the main stream from the
weight-gradient side stream.  Usage: rocpd_streams.py results.db [top=14]"""
import re
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    rows = db.execute("select name, stream_id, queue_id, end - start from kernels").fetchall()
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    tot = defaultdict(float)
    for n, st, q, d in rows:
        k = re.sub(r"^void\s+", "", re.sub(r"\(.*", "", n))[:52]
        a = acc[(st, q)][k]
        a[0] += 1
        a[1] += d
        tot[(st, q)] += d
    for key in sorted(tot, key=lambda k: -tot[k]):
        print("== stream %s queue %s: %.1f ms of kernel time" % (key[0], key[1], tot[key] / 1e6))
        for k, (c, t) in sorted(acc[key].items(), key=lambda kv: -kv[1][1])[:top]:
            print("   %-52s %7d calls %9.2f ms  %8.1f us" % (k, c, t / 1e6, t / c / 1e3))


if __name__ == "__main__":
    main()
