"""Raw per-kernel counter sums of a rocprofv3 --pmc rocpd database, normalised per launch.  Usage: pmc_dump.py results.db [name-substring]"""
import re, sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name").fetchall()
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
nc = "name" if "name" in cols else [c for c in cols if "name" in c][0]
launches = dict(db.execute("select %s, count(*) from kernels group by %s" % (nc, nc)).fetchall())
d = defaultdict(dict)
for n, c, k, s in rows:
    d[n][c] = s
for n, cs in d.items():
    if pat in n:
        L = max(launches.get(n, 1), 1)
        print(re.sub(r"\(.*", "", n)[:70], "launches", L, {k: "%.4g" % (v / L) for k, v in sorted(cs.items())})
