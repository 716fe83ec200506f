#!/bin/bash
# SQ counter pass over the decoder timing script (random LLRs, six passes): instructions and wait cycles per wave of both 16-window kernels.
#   gpurun -- bash scripts/pmc_tdec.sh [ncb]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3/pmc
mkdir -p $OUT
NCB=${1:-1664}
for fw in 16 3016; do
  (cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $OUT -o fw$fw --output-format csv -- python3 $OLDPWD/scripts/tdec_phase_timing.py $fw $NCB > $OUT/fw$fw.txt 2> $OUT/fw$fw.err) || tail -3 $OUT/fw$fw.err
  (cd /tmp && rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES -d $OUT -o fwb$fw --output-format csv -- python3 $OLDPWD/scripts/tdec_phase_timing.py $fw $NCB > $OUT/fwb$fw.txt 2> $OUT/fwb$fw.err) || tail -3 $OUT/fwb$fw.err
done
python3 - <<PY
import csv, glob, collections
for tag in ("fw16", "fwb16", "fw3016", "fwb3016"):
    for f in glob.glob("$OUT/**/%s_counter_collection.csv" % tag, recursive=True):
        tot = collections.defaultdict(float); n = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            if "tdec_" not in r["Kernel_Name"]: continue
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
        print(tag, {k: round(v / len(n[k])) for k, v in tot.items()})
PY
