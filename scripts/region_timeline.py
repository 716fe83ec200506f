#!/usr/bin/env python3
"""Timeline of the contract's K-step timed regions from a rocprofv3 kernel trace of bench.py (scripts/trace_region.sh): where a region's time goes
beyond K steady-state steps. Regions = maximal runs of kernels without an idle gap of more than 40 us (the barriers); for the regions of the
most frequent length: per 0.2-ms bin the number of decoder launches and front-end kernels in flight, and the per-queue end times."""
import csv, glob, sys, collections

def kname(n):  # "void (anonymous namespace)::tdec_pair_kernel(...)" -> "tdec_pair_kernel"
    n = n.replace("(anonymous namespace)::", "")
    if n.startswith("void "):
        n = n[5:]
    return n.split("(")[0].split("<")[0][:28]


f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kname(r["Kernel_Name"]), r.get("Queue_Id", "?")))
rows.sort()
regions, cur, end = [], [], None
for r in rows:
    if end is not None and r[0] - end > 40000:
        regions.append(cur); cur = []
    cur.append(r); end = max(end or 0, r[1])
regions.append(cur)
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sel = [g for g in regions if sum(1 for x in g if x[2].startswith("tdec_pair")) == K]
print("regions: %d, with %d decoder launches: %d" % (len(regions), K, len(sel)))
if not sel:
    sys.exit(0)
lens = sorted((max(x[1] for x in g) - g[0][0]) / 1e6 for g in sel)
print("region length ms: min %.3f median %.3f max %.3f" % (lens[0], lens[len(lens) // 2], lens[-1]))
g = sel[len(sel) // 2]
t0, t1 = g[0][0], max(x[1] for x in g)
print("one region: %.3f ms; first decoder starts at %.3f ms, last front-end kernel ends at %.3f ms" % (
    (t1 - t0) / 1e6, (min(x[0] for x in g if x[2].startswith("tdec")) - t0) / 1e6, (max(x[1] for x in g if not x[2].startswith("tdec")) - t0) / 1e6))
binw = 200000
nb = (t1 - t0) // binw + 1
dec, fe = [0.0] * nb, [0.0] * nb
for s, e, n, q in g:
    arr = dec if n.startswith("tdec") else fe
    b = (s - t0) // binw
    while b * binw + t0 < e:
        lo, hi = max(s, t0 + b * binw), min(e, t0 + (b + 1) * binw)
        arr[b] += (hi - lo) / binw
        b += 1
print("bin(0.2ms)  decoders-in-flight  front-end-kernels-in-flight")
for b in range(nb):
    print("%5.1f  %5.2f  %5.2f" % (b * 0.2, dec[b], fe[b]))
perq = collections.defaultdict(list)
for s, e, n, q in g:
    if n.startswith("tdec"):
        perq[q].append(((s - t0) / 1e6, (e - t0) / 1e6))
for q in sorted(perq):
    print("queue", q, " ".join("%.2f-%.2f" % se for se in perq[q]))
