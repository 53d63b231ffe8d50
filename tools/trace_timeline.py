"""rocprofv3 --kernel-trace CSV -> what the backward (or any phase between two marker kernels) spends its wall time on.
usage: python tools/trace_timeline.py <kernel_trace.csv> [start_marker end_marker]
Per kernel name: launches, summed duration, EXCLUSIVE time (no other kernel running), and per stream busy time; plus the neighbours of torch's
fill / copy kernels (which Python line issues them)."""
import csv
import sys
import re
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"(?:void )?([\w:]+(?:<[^(]{0,60})?)", n)
    s = m.group(1) if m else n[:60]
    if "at::native" in n:
        k = re.search(r"(FillFunctor<[\w:]+>|bfloat16_copy_kernel\w*|bfloat16tofloat32_copy\w*|float32tobfloat16\w*|copy_kernel\w*|index_\w+|CatArrayBatchedCopy\w*|direct_copy\w*|\w+Functor\w*)", n)
        s = "at::" + (k.group(1) if k else s[:40])
    return s[:70]


def load(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Stream_Id"] + "/" + r["Queue_Id"]))
    rows.sort()
    return rows


def windows(rows, a, b):
    out, start = [], None
    for i, r in enumerate(rows):
        if start is None and a in r[2]:
            start = i
        elif start is not None and b in r[2]:
            out.append((start, i))
            start = None
    return out


def analyse(rows, i0, i1):
    seg = rows[i0:i1 + 1]
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    ev = []
    for k, (s, e, n, st) in enumerate(seg):
        ev.append((s, 1, k)); ev.append((e, -1, k))
    ev.sort()
    active, last = set(), t0
    excl, tot, cnt, idle = defaultdict(float), defaultdict(float), defaultdict(int), 0.0
    for t, d, k in ev:
        if t > last:
            if len(active) == 1:
                excl[seg[next(iter(active))][2]] += t - last
            elif not active:
                idle += t - last
            last = t
        if d == 1:
            active.add(k)
        else:
            active.discard(k)
    streams = defaultdict(float)
    for s, e, n, st in seg:
        tot[n] += e - s; cnt[n] += 1; streams[st] += e - s
    return dict(wall=(t1 - t0) / 1e6, idle=idle / 1e6, excl=excl, tot=tot, cnt=cnt, streams=streams)


if __name__ == "__main__":
    rows = load(sys.argv[1])
    a, b = (sys.argv[2], sys.argv[3]) if len(sys.argv) > 3 else ("logp_bwd_kernel", "embed_bwd_kernel")
    ws = windows(rows, a, b)
    print("%d windows %s .. %s" % (len(ws), a, b))
    if ws:
        r = analyse(rows, *ws[-1])
        print("last window: wall %.2f ms, idle %.2f ms, stream busy ms: %s" % (r["wall"], r["idle"], {k: round(v / 1e6, 1) for k, v in r["streams"].items()}))
        print("%-72s %6s %9s %9s" % ("kernel", "n", "total ms", "excl ms"))
        for n in sorted(r["tot"], key=lambda n: -r["tot"][n])[:40]:
            print("%-72s %6d %9.2f %9.2f" % (n, r["cnt"][n], r["tot"][n] / 1e6, r["excl"].get(n, 0.0) / 1e6))
    # neighbours of torch glue kernels
    seen = defaultdict(int)
    for i, r in enumerate(rows):
        if r[2].startswith("at::") and r[1] - r[0] > 50000:
            key = (rows[i - 1][2] if i else "", r[2], rows[i + 1][2] if i + 1 < len(rows) else "")
            seen[key] += 1
    print("\ntorch glue kernels > 50 us: (previous, kernel, next) x count")
    for k, v in sorted(seen.items(), key=lambda kv: -kv[1])[:25]:
        print(v, k)
