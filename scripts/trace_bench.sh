#!/bin/bash
# rocprofv3 kernel trace + stats of the bench line (program directly after --); prints the stats table
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3/trace_${TAG:-x}
mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python3 $OLDPWD/bench.py --no-cpu --stream-batch 0 --no-full --min-timed-s 0.05 $BENCH_ARGS > $OUT/bench.json 2> $OUT/err.txt) || tail -5 $OUT/err.txt
python3 - <<PY
import csv, glob, json
f = glob.glob("$OUT/**/t_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-60s calls %6s avg %10.1f us  total %8.2f ms  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"])
PY
