#!/usr/bin/env python
"""MFMA-busy summary from a rocprofv3 rocpd database collected with
   --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
Per kernel family (template arguments stripped): launches, mean GPU-active cycles, and
   MfmaUtil % = 100 * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 256 CUs * 4 SIMDs)      (the gfx94x derived-metric formula; ROCm 7.2
   ships no gfx950 section, MI355X_MICROARCH.md PMC notes)
   wait % = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked on s_waitcnt / barriers), issue-stall % = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES.
Usage: pmc_mfma.py results.db [out.md]"""
import re
import sqlite3
import sys
from collections import defaultdict


def family(name):
    n = re.sub(r"^void\s+", "", re.sub(r"\(.*", "", name))
    return re.sub(r"<.*", "", n).strip()


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, counter_name, counter_value from pmc_events").fetchall()
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for n, c, v in rows:
        f = family(n)
        acc[f][c] += v
        cnt[f][c] += 1
    # rocprofv3 stores one row per counter INSTANCE (GRBM_GUI_ACTIVE: one per XCD): normalise with the dispatch count of the kernels table
    kc = defaultdict(int)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    for n, c in db.execute("select %s, count(*) from kernels group by %s" % (name_col, name_col)):
        kc[family(n)] += c
    lines = ["| kernel family | launches | GPU-active cycles / launch (per XCD) | MfmaUtil % | waves parked % | issue stall % |", "|---|---|---|---|---|---|"]
    order = sorted(acc, key=lambda f: -acc[f].get("GRBM_GUI_ACTIVE", 0.0))
    for f in order[:30]:
        a = acc[f]
        gui = a.get("GRBM_GUI_ACTIVE", 0.0)
        if gui <= 0:
            continue
        n = kc.get(f) or cnt[f]["GRBM_GUI_ACTIVE"]
        inst = max(1.0, cnt[f]["GRBM_GUI_ACTIVE"] / float(n))          # GRBM instances per dispatch (8 XCDs)
        gui = gui / inst                                                 # mean active cycles of one XCD, summed over launches
        mf = 100.0 * a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 256 * 4)
        wc = a.get("SQ_WAVE_CYCLES", 0.0)
        wa = 100.0 * a.get("SQ_WAIT_ANY", 0.0) / wc if wc else float("nan")
        wi = 100.0 * a.get("SQ_WAIT_INST_ANY", 0.0) / wc if wc else float("nan")
        lines.append("| %s | %d | %.0f | %.1f | %.1f | %.1f |" % (f[:60], n, gui / n, mf, wa, wi))
    txt = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
