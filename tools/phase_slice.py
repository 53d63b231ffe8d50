"""rocprofv3 --kernel-trace CSV -> the kernels between the LAST launch of <start_marker> before each <end_marker> launch and that <end_marker> launch
(e.g. samp_sum_pick_kernel .. logp_bwd_kernel = the log-prob phase of a micro-step): wall time, idle time, per-kernel totals, per stream.
usage: python tools/phase_slice.py <kernel_trace.csv> <start_marker> <end_marker>"""
import sys
from collections import defaultdict
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from trace_timeline import load

rows = load(sys.argv[1])
a, b = sys.argv[2], sys.argv[3]
last_a, slices = None, []
for i, r in enumerate(rows):
    if a in r[2]:
        last_a = i
    elif b in r[2] and last_a is not None:
        slices.append((last_a, i))
        last_a = None
for n, (i0, i1) in enumerate(slices):
    seg = rows[i0 + 1:i1]
    if not seg:
        continue
    t0, t1 = rows[i0][1], rows[i1][0]
    busy, cur = 0, t0
    for s, e, _, _ in sorted(seg):
        if e > cur:
            busy += e - max(s, cur)
            cur = e
    tot = defaultdict(lambda: [0, 0])
    for s, e, k, st in seg:
        tot[(k, st)][0] += 1
        tot[(k, st)][1] += e - s
    per = defaultdict(int)
    for s_, e_, _, st_ in seg:
        per[st_] += e_ - s_
    print("   per stream/queue busy ms:", {k: round(v / 1e6, 2) for k, v in per.items()})
    gaps, cur, prev = [], t0, rows[i0][2]
    for s_, e_, k_, st_ in sorted(seg):
        if s_ > cur:
            gaps.append((s_ - cur, prev, k_))
        if e_ > cur:
            cur, prev = e_, k_
    print("   largest idle gaps (us, after kernel -> before kernel):", [(round(g / 1e3, 1), a_[:28], b_[:28]) for g, a_, b_ in sorted(gaps, reverse=True)[:14]])
    print("slice %d: wall %.2f ms, some kernel running %.2f ms, idle %.2f ms, %d launches" % (n, (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(seg)))
    for (k, st), (c, d) in sorted(tot.items(), key=lambda x: -x[1][1])[:int(sys.argv[4]) if len(sys.argv) > 4 else 14]:
        print("   %-70s %-6s %5d x %9.1f us = %8.2f ms" % (k, st, c, d / c / 1e3, d / 1e6))
