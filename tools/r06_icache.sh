#!/bin/bash
# Instruction-cache counters of k_detect with and without bursts (is the per-burst code -- 7-10 k instructions per instance,
# about the size of the 64 KB instruction cache a CU pair shares -- evicting the quiet loop of the neighbours?)
export TMPDIR=/tmp
ROOT=$(pwd)
O=$ROOT/gpurun_out/r06; mkdir -p $O
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INST_PREFETCH[A-Z_]*\|SQC_TC_INST[A-Z_]*\|SQ_WAIT_INST_ANY\|SQ_WAIT_IFETCH[A-Z_]*" | sort -u > $O/icache_avail.txt
cat $O/icache_avail.txt
PASSES=("SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
        "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
        "SQC_TC_INST_REQ SQC_TC_REQ SQ_BUSY_CYCLES SQ_INSTS_VALU")
: > $O/icache.txt
for cfg in "--log2n 30 --format mag2" "--log2n 30 --format mag2 --bursts 0" "--log2n 30 --format sc8" "--log2n 30 --format sc8 --bursts 0" "--log2n 30"; do
  echo "== $cfg" >> $O/icache.txt
  for p in "${PASSES[@]}"; do
    rm -rf /tmp/sqd
    timeout 300 rocprofv3 --pmc $p --kernel-trace -f csv -d /tmp/sqd -o p -- python $ROOT/bench.py --no-cpu --no-extra --no-hostfed --steps 3 --warmup 1 --min-time 0 $cfg > /tmp/sqd.log 2>&1
    f=$(find /tmp/sqd -name '*counter_collection.csv' | head -1)
    if [ -n "$f" ]; then python $ROOT/tools/pmc_summary.py $f | grep -A4 "k_detect" >> $O/icache.txt; else echo "  (no output for: $p)" >> $O/icache.txt; tail -3 /tmp/sqd.log >> $O/icache.txt; fi
  done
done
cat $O/icache.txt
