#!/bin/bash
# SQ instruction mix of k_detect per workload (VALU / SALU / LDS instructions per launch; 2^28 samples = 262144 tiles):
#   bash tools/sq_mix.sh ["bench args" ...]
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
CFGS=("--log2n 28" "--fs 8e6 --bursts 6000 --log2n 28" "--mixed-df --log2n 28" "--fs 20e6 --log2n 28")
if [ $# -gt 0 ]; then CFGS=("$@"); fi
for cfg in "${CFGS[@]}"; do
  rm -rf /tmp/sqx
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace -f csv -d /tmp/sqx -o p -- python $ROOT/bench.py --no-cpu --no-extra --no-hostfed --steps 4 --warmup 1 --min-time 0 $cfg > /tmp/sqx.log 2>&1
  echo "== $cfg"; python $ROOT/tools/pmc_summary.py $(find /tmp/sqx -name '*counter_collection.csv' | head -1) | grep -A4 "k_detect"
done
