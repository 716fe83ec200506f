#!/usr/bin/env python3
"""Turn the rocprofv3 passes of scripts/profile_r04.sh (gpurun_out/r4/prof/) into profiles/r04/:
  tdec_counters.json      instruction counts and HBM traffic of the turbo decoder launch (what bench.py prices roofline.* with; tagged with the
                          hash of tdec.hip + tdec_pair.inc: bench.py drops it when the source has changed since)
  kernels_by_grid.json    every kernel of the pipeline per launch SIZE (the batch-128 launches of the timed region and the batch-2048 launches of
                          'kernels_large_batch' are different grid sizes of the same kernels): rocprof average duration, FETCH_SIZE, WRITE_SIZE,
                          algorithmic bytes and the fractions of the 8 TB/s HBM peak they give
  *_kernel_stats.csv, *_bench_under_rocprof.json, pmc_*_counter_collection.csv: the raw summaries
    python scripts/profile_r04_post.py gpurun_out/r4/prof profiles/r04"""
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

KERNEL = "tdec_pair_kernel"
HBM = 8000.0e9


def find(d, name, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "%s_%s.csv" % (name, suffix)), recursive=True))
    return hits[0] if hits else None


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()


def by_grid_counters(path):
    """{(kernel, grid size): {counter: mean per dispatch}}"""
    tot, ids = defaultdict(float), defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = (short(r["Kernel_Name"]), int(r["Grid_Size"]), r["Counter_Name"])
        tot[k] += float(r["Counter_Value"])
        ids[k].add(r["Dispatch_Id"])
    out = defaultdict(dict)
    for (kern, grid, ctr), v in tot.items():
        out[(kern, grid)][ctr] = v / len(ids[(kern, grid, ctr)])
    return out


def by_grid_durations(path):
    tot, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))  # threads, as the counter passes report it
        tot[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        n[k] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


def bench_line(d, name):
    with open(os.path.join(d, name + ".bench.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for fn in ("tdec.hip", "tdec_pair.inc"):
        h.update(open(os.path.join(root, "srslte-emane_amd", "csrc", fn), "rb").read())
    sha = h.hexdigest()[:16]
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=root, stdout=subprocess.PIPE).stdout.decode().strip()
    fetch, write = by_grid_counters(find(src, "fetch", "counter_collection")), by_grid_counters(find(src, "write", "counter_collection"))
    dur_def, dur_s1 = by_grid_durations(find(src, "default", "kernel_trace")), by_grid_durations(find(src, "streams1", "kernel_trace"))
    # ---- the decoder launch
    out = {"kernel": KERNEL, "tdec_src_sha": sha, "head": head, "batch": 128,
           "source": "scripts/profile_r04.sh: rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ counters - each on its own) of python3 bench.py --streams 1 --steps 3"}
    tk = [k for k in fetch if KERNEL in k[0]]
    assert tk, "no decoder dispatch in the FETCH pass"
    fk, wk = fetch[tk[0]]["FETCH_SIZE"], write[tk[0]]["WRITE_SIZE"]
    out["FETCH_SIZE_KB_per_launch"], out["WRITE_SIZE_KB_per_launch"] = round(fk, 1), round(wk, 1)
    out["traffic_bytes_per_launch"] = int((2 * fk + wk) * 1024)  # FETCH_SIZE doubled: gfx950 correction (MI355X_MICROARCH.md, HBM section)
    sq = {}
    for name in ("sq", "sq_full"):
        c = by_grid_counters(find(src, name, "counter_collection"))
        k = [x for x in c if KERNEL in x[0]][0]
        b = bench_line(src, name)
        waves = c[k]["SQ_WAVES"]
        sq[name] = {"passes": b["config"]["avg_siso_passes_per_cb"], "waves_per_launch": round(waves, 1),
                    "per_wave": {n: round(v / waves, 1) for n, v in c[k].items() if n != "SQ_WAVES"}}
    out["sq"] = sq
    out["waves_per_launch"] = sq["sq"]["waves_per_launch"]
    p0, p1 = sq["sq"]["passes"], sq["sq_full"]["passes"]
    v0, v1 = sq["sq"]["per_wave"]["SQ_INSTS_VALU"], sq["sq_full"]["per_wave"]["SQ_INSTS_VALU"]
    per_pass = (v1 - v0) / (p1 - p0)  # per wave (two code blocks) and average pass count per block
    out["valu_instr_per_wave_per_pass"], out["valu_instr_per_wave_fixed"] = round(per_pass, 1), round(v1 - per_pass * p1, 1)
    for name, dur in (("default", dur_def), ("streams1", dur_s1)):
        k = [x for x in dur if KERNEL in x[0]]
        if k:
            out["rocprof_avg_ns_" + name] = round(dur[k[0]][0], 1)
            out["hip_event_avg_ms_" + name] = bench_line(src, name)["roofline"]["avg_launch_ms"]
    json.dump(out, open(os.path.join(dst, "tdec_counters.json"), "w"), indent=1)
    # ---- every kernel per launch size
    b = bench_line(src, "default")
    alg = {}
    for name, row in b.get("kernels", {}).items():
        alg[(name, 128)] = row["algorithmic_MB"] * 1e6
    for name, row in b.get("kernels_large_batch", {}).items():
        if isinstance(row, dict):
            alg[(name, b["kernels_large_batch"]["batch"])] = row["algorithmic_MB"] * 1e6
    rows = []
    for (kern, grid), (ns, n) in sorted(dur_def.items()):
        f, w = fetch.get((kern, grid), {}).get("FETCH_SIZE"), write.get((kern, grid), {}).get("WRITE_SIZE")
        rows.append({"kernel": kern, "grid_size": grid, "launches": n, "rocprof_avg_us_default": round(ns / 1e3, 2),
                     "rocprof_avg_us_streams1": round(dur_s1[(kern, grid)][0] / 1e3, 2) if (kern, grid) in dur_s1 else None,
                     "FETCH_SIZE_KB": round(f, 1) if f is not None else None, "WRITE_SIZE_KB": round(w, 1) if w is not None else None,
                     "traffic_MB": round((2 * f + w) * 1024 / 1e6, 2) if f is not None and w is not None else None})
    json.dump({"head": head, "rows": rows, "bench_kernels": b.get("kernels"), "bench_kernels_large_batch": b.get("kernels_large_batch"),
               "note": "grid_size = threads of the launch; the batch-128 launches come from the timed region, the larger grids of the same kernels from "
                       "'kernels_large_batch' (batch 2048). traffic_MB = (2 FETCH_SIZE + WRITE_SIZE) KB, the gfx950 correction of MI355X_MICROARCH.md"},
              open(os.path.join(dst, "kernels_by_grid.json"), "w"), indent=1)
    for name in ("default", "streams1"):
        st = find(src, name, "kernel_stats")
        if st:
            shutil.copy(st, os.path.join(dst, "%s_kernel_stats.csv" % name))
            shutil.copy(os.path.join(src, name + ".bench.json"), os.path.join(dst, "%s_bench_under_rocprof.json" % name))
    for name in ("fetch", "write", "sq", "sq_full"):
        shutil.copy(find(src, name, "counter_collection"), os.path.join(dst, "pmc_%s_counter_collection.csv" % name))
    print(json.dumps(out, indent=1))
    for r in rows:
        print(r)


if __name__ == "__main__":
    main()
