#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box and leave only small text summaries under
# gpurun_out/prof/ (gpurun merges <= 64 MiB back).  Run from the repo root:  bash tools/profile_round.sh [rNN]
# Counters are collected in their own passes with --kernel-trace only (never with sys/hip/hsa traces).
set -u
R=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
export TMPDIR=/tmp
W=/tmp/adsb_prof; rm -rf $W; mkdir -p $W
cd /tmp
B="python $ROOT/bench.py --no-cpu --no-extra --no-hostfed"

# 1. kernel trace of the default bench (tail kernels on their own stream, 3 passes in flight)
rocprofv3 --kernel-trace --stats -d $W/kt -o kt -- $B > $W/kt.log 2>&1
DB=$(find $W/kt -name '*.db' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu --no-extra --no-hostfed   (tail on its own stream, 3 passes in flight)";
  echo "# profiled runs overlap the streams less and clock lower than unprofiled ones; the bench's JSON line of THIS run is at the bottom";
  python $ROOT/tools/prof_summary.py "$DB"; echo; grep '"metric"' $W/kt.log | tail -1; } > $OUT/${R}_kernel_trace_stats_bench_default.txt 2>&1

# 2. the same with every kernel on one stream (ADSB_FLAG_SINGLE_STREAM)
rocprofv3 --kernel-trace --stats -d $W/kt1 -o kt1 -- $B --single-stream > $W/kt1.log 2>&1
DB=$(find $W/kt1 -name '*.db' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu --no-extra --no-hostfed --single-stream";
  python $ROOT/tools/prof_summary.py "$DB"; echo; grep '"metric"' $W/kt1.log | tail -1; } > $OUT/${R}_kernel_trace_stats_bench_single_stream.txt 2>&1

# 3. HBM traffic counters, one pass each (2^30 and 2^28 samples per launch)
for L in 30 28; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace -f csv -d $W/pmc_${C}_$L -o p -- $B --log2n $L --steps 4 --warmup 1 --min-time 0 > $W/pmc_${C}_$L.log 2>&1
  done
  { echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python bench.py --log2n $L --steps 4 --warmup 1 --min-time 0 --no-cpu --no-extra --no-hostfed   (KiB per launch)";
    python $ROOT/tools/pmc_summary.py $(find $W/pmc_FETCH_SIZE_$L -name '*counter_collection.csv' | head -1) \
                                      $(find $W/pmc_WRITE_SIZE_$L -name '*counter_collection.csv' | head -1); } > $OUT/${R}_pmc_hbm_traffic_log2n$L.txt 2>&1
done

# 4. SQ counters (waits / issue / instruction mix) for the same command
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -f csv -d $W/sq1 -o p -- $B --steps 4 --warmup 1 --min-time 0 > $W/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace -f csv -d $W/sq2 -o p -- $B --steps 4 --warmup 1 --min-time 0 > $W/sq2.log 2>&1
python $ROOT/tools/pmc_summary.py $(find $W/sq1 -name '*counter_collection.csv' | head -1) \
                                  $(find $W/sq2 -name '*counter_collection.csv' | head -1) > $OUT/${R}_pmc_sq_counters.txt 2>&1

# 5. unprofiled bench lines on the same box: the full default line, the other signal configs, the integer formats
cd $ROOT
python bench.py > $OUT/${R}_bench_unprofiled.json 2> $W/bench.err
python bench.py --fs 8e6 --bursts 6000 --no-cpu --no-extra --no-hostfed > $OUT/${R}_bench_8msps_dense.json 2>> $W/bench.err
python bench.py --fs 20e6 --no-cpu --no-extra --no-hostfed > $OUT/${R}_bench_20msps.json 2>> $W/bench.err
for f in sc16 sc8 cu8; do python bench.py --format $f --no-cpu > $OUT/${R}_bench_$f.json 2>> $W/bench.err; done
tail -5 $W/bench.err > $OUT/${R}_bench_stderr_tail.txt
ls -la $OUT
