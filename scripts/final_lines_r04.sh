#!/bin/bash
# Everything profiles/r04/ holds about the FINAL decoder source, taken on one box in one call (gpurun -- bash scripts/final_lines_r04.sh):
# the rocprofv3 passes (profile_r04.sh -> profile_r04_post.py) and the bench lines of every variant the READMEs quote. Outputs under
# gpurun_out/r4/final/ (copy the *.json / *.csv into profiles/r04/).
set -e
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r4/final
mkdir -p $OUT
if [ -z "$SKIP_PROFILE" ]; then
  bash scripts/profile_r04.sh > $OUT/profile_log.txt 2>&1
  mkdir -p $OUT/profiles_r04
  python scripts/profile_r04_post.py gpurun_out/r4/prof $OUT/profiles_r04 >> $OUT/profile_log.txt 2>&1
  cp $OUT/profiles_r04/* profiles/r04/ # bench.py quotes the counters of the source it runs (tdec_counters.json carries the source's hash)
  echo "== profiles done"
fi
python bench.py > $OUT/final_bench.json 2> $OUT/final_bench.err
tail -c 300 $OUT/final_bench.json; echo
line() { # name, args...
  local name=$1; shift
  python "$@" > $OUT/$name.json 2> $OUT/$name.err
  python -c "import sys,json; d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', d.get('value'), d.get('ms_per_step'))"
}
line final_bench_streams4 bench.py --no-cpu --streams 4
line final_bench_copy bench.py --no-cpu --no-zero-copy
line final_bench_pool bench.py --no-cpu --pool
line final_bench_llr8 bench.py --no-cpu --llr8
line final_bench_streams1 bench.py --no-cpu --streams 1 --stream-batch 0
line final_bench_grants_mix bench.py --grants-mix  # with its cpu_baseline (tests/test_profiles_consistency.py)
line final_bench_grants bench.py --no-cpu --grants
line final_bench_grants_mix_llr8 bench.py --grants-mix --llr8
line cfg4_one_device scripts/cfg4_one_device.py
