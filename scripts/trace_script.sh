#!/bin/bash
# rocprofv3 kernel trace + stats of an arbitrary python script of this repo (SCRIPT, ARGS; program directly after --); prints the stats table
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4/trace_${TAG:-s}
mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python3 $OLDPWD/$SCRIPT $ARGS > $OUT/out.txt 2> $OUT/err.txt) || tail -5 $OUT/err.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/t_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.3:
        print("%-70s calls %6s avg %10.1f us  total %8.2f ms  %5s%%" % (r["Name"].replace("(anonymous namespace)::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
tail -c 600 $OUT/out.txt
