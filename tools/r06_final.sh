#!/bin/bash
# Round 6, the final set on ONE box: -m gpu, fuzz, the default bench line, the profile round (kernel traces blocking /
# pipelined / single-stream, PMC traffic, per workload), SQ counters of the shipped kernels.
export TMPDIR=/tmp
ROOT=$(pwd)
O=$ROOT/gpurun_out/r06
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gputest_final.txt 2>&1; echo "pytest rc $?" >> $O/gputest_final.txt; tail -3 $O/gputest_final.txt
timeout 400 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_final.txt 2>&1; tail -2 $O/smoke_final.txt
timeout 500 python tools/fuzz_gpu.py 240 > $O/fuzz_final.txt 2>&1; tail -2 $O/fuzz_final.txt
timeout 900 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc $?"
bash tools/profile_round.sh r06 > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log
cd /tmp
PASSES=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES")
: > $O/sq_final.txt
for cfg in "--log2n 30 --format sc8" "--log2n 30 --format sc8 --sc8-generic" "--log2n 30 --format cu8"; do
  echo "== $cfg" >> $O/sq_final.txt
  for p in "${PASSES[@]}"; do
    rm -rf /tmp/sqd
    timeout 300 rocprofv3 --pmc $p --kernel-trace -f csv -d /tmp/sqd -o p -- python $ROOT/bench.py --no-cpu --no-extra --no-hostfed --steps 3 --warmup 1 --min-time 0 $cfg > /tmp/sqd.log 2>&1
    python $ROOT/tools/pmc_summary.py $(find /tmp/sqd -name '*counter_collection.csv' | head -1) | grep -A4 "k_detect" >> $O/sq_final.txt
  done
done
cat $O/sq_final.txt
