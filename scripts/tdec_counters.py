#!/usr/bin/env python3
"""Turn the rocprofv3 passes of scripts/profile_r02.sh (gpurun_out/r2/prof/) into profiles/r02/tdec_counters.json, the file bench.py prices
the turbo decoder's VALU issue and HBM traffic with. Tagged with the sha of the tdec.hip it was measured on: bench.py drops it when the
source has changed since. Also copies the kernel-stats CSVs of the two --stats passes into profiles/r02/.

  python scripts/tdec_counters.py gpurun_out/r2/prof profiles/r02"""
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

KERNEL = "tdec_win_kernel<16, 0>"


def find(d, name, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "%s_%s.csv" % (name, suffix)), recursive=True)) + sorted(glob.glob(os.path.join(d, "**", "*%s*%s.csv" % (name, suffix)), recursive=True))
    return hits[0] if hits else None


def counters(path):
    """{counter: (sum over dispatches of the decoder kernel, number of its dispatches)}"""
    tot, ids = defaultdict(float), defaultdict(set)
    for r in csv.DictReader(open(path)):
        if KERNEL not in r["Kernel_Name"]:
            continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        ids[r["Counter_Name"]].add(r["Dispatch_Id"])
    return {k: (v, len(ids[k])) for k, v in tot.items()}


def bench_line(d, name):
    with open(os.path.join(d, name + ".bench.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sha = hashlib.sha256(open(os.path.join(root, "srslte-emane_amd", "csrc", "tdec.hip"), "rb").read()).hexdigest()[:16]
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=root, stdout=subprocess.PIPE).stdout.decode().strip()
    out = {"kernel": KERNEL, "tdec_hip_sha": sha, "head": head, "batch": 128,
           "source": "scripts/profile_r02.sh: rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ counters - each on its own) of python3 bench.py --streams 1 --steps 3"}
    f, w = counters(find(src, "fetch", "counter_collection")), counters(find(src, "write", "counter_collection"))
    fk, wk = f["FETCH_SIZE"][0] / f["FETCH_SIZE"][1], w["WRITE_SIZE"][0] / w["WRITE_SIZE"][1]
    out["FETCH_SIZE_KB_per_launch"], out["WRITE_SIZE_KB_per_launch"] = round(fk, 1), round(wk, 1)
    out["traffic_bytes_per_launch"] = int((2 * fk + wk) * 1024)  # FETCH_SIZE doubled: gfx950 correction (MI355X_MICROARCH.md, HBM section)
    sq = {}
    for name in ("sq", "sq_full"):
        c = counters(find(src, name, "counter_collection"))
        b = bench_line(src, name)
        waves = c["SQ_WAVES"][0]
        sq[name] = {"passes": b["config"]["avg_siso_passes_per_cb"], "per_wave": {k: round(v[0] / waves, 1) for k, v in c.items() if k != "SQ_WAVES"},
                    "waves_per_launch": round(waves / c["SQ_WAVES"][1], 1)}
    out["sq"] = sq
    p0, p1 = sq["sq"]["passes"], sq["sq_full"]["passes"]
    v0, v1 = sq["sq"]["per_wave"]["SQ_INSTS_VALU"], sq["sq_full"]["per_wave"]["SQ_INSTS_VALU"]
    per_pass = (v1 - v0) / (p1 - p0)
    out["valu_instr_per_wave_per_pass"], out["valu_instr_per_wave_fixed"] = round(per_pass, 1), round(v1 - per_pass * p1, 1)
    for name in ("default", "streams1"):
        st = find(src, name, "kernel_stats")
        if st:
            shutil.copy(st, os.path.join(dst, "%s_kernel_stats.csv" % name))
            shutil.copy(os.path.join(src, name + ".bench.json"), os.path.join(dst, "%s_bench_under_rocprof.json" % name))
            for r in csv.DictReader(open(st)):
                if KERNEL in r["Name"]:
                    out["rocprof_avg_ns_" + name] = float(r["AverageNs"])
                    out["hip_event_avg_ms_" + name] = bench_line(src, name)["roofline"]["avg_launch_ms"]
    for name in ("fetch", "write", "sq", "sq_full"):
        shutil.copy(find(src, name, "counter_collection"), os.path.join(dst, "pmc_%s_counter_collection.csv" % name))
    json.dump(out, open(os.path.join(dst, "tdec_counters.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
