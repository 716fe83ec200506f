#!/usr/bin/env python3
"""Aggregate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) of the same bench
command into HBM bytes per launch and kernel: traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 B (FETCH_SIZE doubled: gfx950 correction).

  python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [note]"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?|__amd_rocclr_\w+)", name)
    return m.group(1) if m else name[:60]


def load(path, counter):
    tot, n = defaultdict(float), defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in n.items()}


def main():
    f, nf = load(sys.argv[1], "FETCH_SIZE")
    w, nw = load(sys.argv[2], "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of python3 bench.py --steps 3 --warmup 1 --no-cpu --stream-batch 0 "
                     "--streams 1; traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 B per launch (FETCH_SIZE doubled: gfx950 correction, MI355X_MICROARCH.md "
                     "HBM section)" + (" " + sys.argv[4] if len(sys.argv) > 4 else ""), "batch": 128, "kernels": {}}
    for k in sorted(set(f) | set(w)):
        n = max(nf.get(k, 0), nw.get(k, 0), 1)
        fk, wk = f.get(k, 0.0) / max(nf.get(k, 1), 1), w.get(k, 0.0) / max(nw.get(k, 1), 1)
        out["kernels"][k] = {"launches": n, "FETCH_SIZE_KB": round(fk, 1), "WRITE_SIZE_KB": round(wk, 1), "traffic_bytes": int((2 * fk + wk) * 1024)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out["kernels"].items():
        print("%-40s %4d launches  %10.1f MB" % (k, v["launches"], v["traffic_bytes"] / 1e6))


if __name__ == "__main__":
    main()
