#!/bin/bash
# the three grants lines of profiles/r04 re-taken (after a change that touches the grants modes only)
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r4/final
mkdir -p $OUT
line() { local name=$1; shift; python "$@" > $OUT/$name.json 2> $OUT/$name.err; python -c "import sys,json; d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', d.get('value'), d.get('ms_per_step'))"; }
line final_bench_grants_mix bench.py --grants-mix
line final_bench_grants bench.py --no-cpu --grants
line final_bench_grants_mix_llr8 bench.py --grants-mix --llr8
