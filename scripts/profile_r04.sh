#!/bin/bash
# Round-4 profiling passes of the bench command on the GPU box (gpurun -- bash scripts/profile_r04.sh), program directly after `--`:
#   1. rocprofv3 --kernel-trace --stats of the default line (4 streams; includes the batch-2048 streaming launches of 'kernels_large_batch')
#      and of --streams 1: kernel durations next to the HIP-event figures bench.py prints;
#   2. --pmc passes, each counter set on its own (MI355X_MICROARCH.md, rocprofv3 PMC): FETCH_SIZE, WRITE_SIZE (with the batch-2048 launches),
#      SQ instruction counters at the headline SNR and at the SNR where every block runs all 6 passes.
# Outputs under gpurun_out/r4/prof/; scripts/profile_r04_post.py turns them into profiles/r04/.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4/prof
mkdir -p $OUT
run() { # name, rocprof args..., bench args in $BARGS
  local name=$1; shift
  echo "== $name"
  (cd /tmp && rocprofv3 "$@" -d $OUT -o $name --output-format csv -- python3 $OLDPWD/bench.py $BARGS > $OUT/$name.bench.json 2> $OUT/$name.err) || { tail -5 $OUT/$name.err; return 1; }
  tail -c 200 $OUT/$name.bench.json; echo
}
BARGS="--no-cpu --no-full --min-timed-s 0.05" run default --kernel-trace --stats
BARGS="--no-cpu --no-full --min-timed-s 0.05 --streams 1 --stream-batch 0" run streams1 --kernel-trace --stats
BARGS="--no-cpu --no-full --min-timed-s 0.02 --streams 1 --steps 3 --warmup 1" run fetch --pmc FETCH_SIZE
BARGS="--no-cpu --no-full --min-timed-s 0.02 --streams 1 --steps 3 --warmup 1" run write --pmc WRITE_SIZE
SQ="SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"
BARGS="--no-cpu --no-full --min-timed-s 0.02 --streams 1 --steps 3 --warmup 1 --stream-batch 0" run sq --pmc $SQ
BARGS="--no-cpu --no-full --min-timed-s 0.02 --streams 1 --steps 3 --warmup 1 --stream-batch 0 --snr 14" run sq_full --pmc $SQ
ls $OUT | head -60
# which counters does this rocprofv3 offer? (one that separates DRAM from Infinity-Cache traffic, if any)
(cd /tmp && rocprofv3 -L > $OUT/counters_list.txt 2>&1) || true
grep -i -E "dram|mall|hbm|infinity|EA0_RDREQ|EA0_WRREQ" $OUT/counters_list.txt | cut -c1-160 | sort -u | head -40
