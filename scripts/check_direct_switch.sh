#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4/chk
mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python3 -m pytest $OLDPWD/tests/test_gpu_grants.py -q -m gpu -k assembled_by_the_decoders -p no:cacheprovider > $OUT/log.txt 2>&1)
tail -2 $OUT/log.txt
grep -E "tb_crc_bytes|tdec_mix|tdec_pair|tdec_win|tdec_gen" $(find $OUT -name "t_kernel_stats.csv") | cut -d, -f1-3 | cut -c1-120
