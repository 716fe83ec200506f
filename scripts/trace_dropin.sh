#!/bin/bash
# HIP API + kernel statistics of the reference's phy_dl_test linked against the library (the drop-in's single-call path)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4/trace_dropin
mkdir -p $OUT
(cd /tmp && rocprofv3 --hip-trace --kernel-trace --stats -d $OUT -o t --output-format csv -- $OLDPWD/oracle/_ref/hip/phy_dl_test -p 100 -t 1 -m 28 > $OUT/out.txt 2> $OUT/err.txt) || tail -5 $OUT/err.txt
tail -8 $OUT/out.txt
python3 - <<PY
import csv, glob
for nm in ("t_hip_api_stats.csv", "t_kernel_stats.csv"):
    for f in glob.glob("$OUT/**/" + nm, recursive=True):
        print("==", nm)
        for i, r in enumerate(csv.DictReader(open(f))):
            if i < 14: print("%-70s calls %7s avg %9.1f us total %9.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
