#!/bin/bash
# Round-2 profiling passes of the bench command on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats on the default line and on --streams 1 (kernel durations next to the HIP-event figures bench.py prints)
#   2. --pmc passes, each on its own (MI355X_MICROARCH.md, rocprofv3 PMC slots): FETCH_SIZE, WRITE_SIZE, SQ instruction counters,
#      at the headline SNR and at the SNR where every block runs all 6 passes (instructions per wave as a function of the pass count)
# Outputs under gpurun_out/r2/prof/; scripts/tdec_counters.py turns them into profiles/r02/tdec_counters.json.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2/prof
mkdir -p $OUT
ARGS="--no-cpu --stream-batch 0 --no-full --min-timed-s 0.02"
run() { # name, rocprof args..., -- bench args
  local name=$1; shift
  echo "== $name"
  (cd /tmp && rocprofv3 "$@" -d $OUT -o $name --output-format csv -- python3 $OLDPWD/bench.py $BARGS > $OUT/$name.bench.json 2> $OUT/$name.err) || { tail -5 $OUT/$name.err; return 1; }
  tail -c 300 $OUT/$name.bench.json | head -c 300; echo
}
BARGS="$ARGS" run default --kernel-trace --stats
BARGS="$ARGS --streams 1" run streams1 --kernel-trace --stats
BARGS="$ARGS --streams 1 --steps 3 --warmup 1" run fetch --pmc FETCH_SIZE
BARGS="$ARGS --streams 1 --steps 3 --warmup 1" run write --pmc WRITE_SIZE
BARGS="$ARGS --streams 1 --steps 3 --warmup 1" run sq --pmc SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU
BARGS="$ARGS --streams 1 --steps 3 --warmup 1 --snr 14" run sq_full --pmc SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU
ls $OUT | head -50
