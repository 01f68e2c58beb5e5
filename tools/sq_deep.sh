#!/bin/bash
# Where k_detect's cycles go: SQ busy / per-class active cycles / LDS conflicts, per workload.
#   bash tools/sq_deep.sh OUT ["bench args" ...]
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/$1; shift
cd /tmp
mkdir -p $(dirname $OUT)
CFGS=("--log2n 30 --format sc8" "--log2n 30")
if [ $# -gt 0 ]; then CFGS=("$@"); fi
PASSES=("SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS"
        "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_VMEM"
        "SQ_INST_CYCLES_SALU SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_WAVE_CYCLES"
        "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES")
: > $OUT
for cfg in "${CFGS[@]}"; do
  echo "== $cfg" >> $OUT
  for p in "${PASSES[@]}"; do
    rm -rf /tmp/sqd
    rocprofv3 --pmc $p --kernel-trace -f csv -d /tmp/sqd -o p -- python $ROOT/bench.py --no-cpu --no-extra --no-hostfed --steps 3 --warmup 1 --min-time 0 $cfg > /tmp/sqd.log 2>&1
    python $ROOT/tools/pmc_summary.py $(find /tmp/sqd -name '*counter_collection.csv' | head -1) | grep -A4 "k_detect" >> $OUT
  done
done
cat $OUT
