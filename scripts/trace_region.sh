#!/bin/bash
# rocprofv3 kernel trace of the default bench loop (program directly after --) and the timeline of one K-step region (scripts/region_timeline.py)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4/region_${TAG:-x}
mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python3 $OLDPWD/bench.py --no-cpu --stream-batch 0 --no-full --min-timed-s 0.05 $BENCH_ARGS > $OUT/bench.json 2> $OUT/err.txt) || tail -5 $OUT/err.txt
python3 scripts/region_timeline.py $OUT ${K:-20}
