#!/bin/bash
# Stall breakdown of k_detect: waiting vs issuing wave-cycles, instruction fetch, per-class instruction cycles.
#   bash tools/sq_stall.sh OUT ["bench args" ...]
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/$1; shift
cd /tmp
mkdir -p $(dirname $OUT)
CFGS=("--log2n 30 --format sc8" "--log2n 30")
if [ $# -gt 0 ]; then CFGS=("$@"); fi
PASSES=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
        "SQ_BUSY_CYCLES SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM"
        "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM"
        "SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM"
        "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_CYCLES")
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
