import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name").fetchall()
from collections import defaultdict
d = defaultdict(dict)
for n, c, k, s in rows:
    d[re.sub(r"\(.*", "", n)[:60]][c] = s
for n, cs in d.items():
    if "gemm" in n: print(n, {k: "%.3g" % v for k, v in cs.items()})
