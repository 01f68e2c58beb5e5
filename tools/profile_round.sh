#!/bin/bash
# The round's rocprofv3 evidence, per BASELINE config and per input format, collected on the GPU box; only small text
# summaries are left under gpurun_out/prof/ (gpurun merges <= 64 MiB back).   bash tools/profile_round.sh [rNN] [quick]
# For every workload W:
#   <R>_<W>_kernel_trace.txt   rocprofv3 --kernel-trace --stats (per-kernel calls / avg / min / max), the bench's own JSON line
#                              of that (profiled) run, and the clocks / power rocm-smi saw while it ran
#   <R>_<W>_kernel_trace_single_stream.txt   the same on one stream: every kernel alone (compare with roofline.isolated)
#   <R>_<W>_pmc_hbm.txt        rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE: separate passes, --kernel-trace only
#   <R>_<W>_bench.json         an UNPROFILED bench line of the same workload on the same box + its clocks line
# plus the SQ instruction / wait counters for the headline workload and the full default bench line.
# Counters are never collected together with sys / hip / hsa traces.
set -u
R=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
export TMPDIR=/tmp
W=/tmp/adsb_prof; rm -rf $W; mkdir -p $W
S="python $ROOT/tools/smi_sampler.py"
B="python $ROOT/bench.py --no-cpu --no-extra --no-hostfed"
# name | bench arguments (the headline and the other input formats at the bench's own default size, 2^30 samples; configs 3 / 4 / 5 at 2^28 like extra_configs)
WL=(
 "cfg2_2msps_fc32|"
 "cfg3_8msps_dense_fc32|--fs 8e6 --bursts 6000 --log2n 28"
 "cfg4_20msps_fc32|--fs 20e6 --log2n 28"
 "cfg5_mixed_df_fc32|--mixed-df --log2n 28"
 "fmt_mag2|--format mag2"
 "fmt_sc16|--format sc16"
 "fmt_sc8|--format sc8"
 "fmt_cu8|--format cu8 --cu8-generic"
 "fmt_cu8p|--format cu8"
)
# (fmt_cu8 = the generic uint8 instance, any scale; fmt_cu8p = the dot-product instance, the bench's default uint8 leg since
#  round 6.)  ONLY=<name> in the environment: that workload alone, without the closing SQ / full-bench steps.
cd /tmp
for w in "${WL[@]}"; do
  name=${w%%|*}; args=${w#*|}
  [ -n "${ONLY:-}" ] && [ "$ONLY" != "$name" ] && continue
  # 1. kernel trace (default pipeline: tail kernels on their own stream, three passes in flight)
  $S $W/$name.kt.clk -- rocprofv3 --kernel-trace --stats -d $W/kt_$name -o kt -- $B $args --steps 10 --warmup 3 --min-time 0.2 > $W/kt_$name.log 2>&1
  DB=$(find $W/kt_$name -name '*.db' | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu --no-extra --no-hostfed $args --steps 10 --warmup 3 --min-time 0.2";
    echo "# (in the default pipeline the tail chain of a pass starts when the NEXT pass's k_detect drains: k_scan's duration is that wait)";
    python $ROOT/tools/prof_summary.py "$DB"; echo; echo "# the bench's own JSON line of this profiled run (roofline.kernel_ms = HIP events on the compute stream):";
    grep '"metric"' $W/kt_$name.log | tail -1; echo; echo "# rocm-smi while it ran (samples with the GPU busy):"; cat $W/$name.kt.clk; } > $OUT/${R}_${name}_kernel_trace.txt 2>&1
  # 1b. the same with every kernel on ONE stream (ADSB_FLAG_SINGLE_STREAM): k_detect and the tail kernels each alone --
  #     the figure to hold against `roofline.isolated` of the unprofiled line (the pipelined trace above also times the
  #     tail kernels that run BESIDE the next k_detect, which the profiler stretches)
  rocprofv3 --kernel-trace --stats -d $W/kt1_$name -o kt1 -- $B $args --single-stream --steps 10 --warmup 3 --min-time 0.2 > $W/kt1_$name.log 2>&1
  DB=$(find $W/kt1_$name -name '*.db' | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu --no-extra --no-hostfed $args --single-stream --steps 10 --warmup 3 --min-time 0.2";
    python $ROOT/tools/prof_summary.py "$DB" | grep -i "kernel \|adsb\|copyBuffer"; echo; echo "# the bench's own JSON line of this profiled run:";
    grep '"metric"' $W/kt1_$name.log | tail -1; } > $OUT/${R}_${name}_kernel_trace_single_stream.txt 2>&1
  # 1c. BLOCKING passes (--depth 1): what the profiler can time faithfully.  rocprofv3 turns SDMA off, so every device->host
  #     record copy becomes a blit KERNEL (__amd_rocclr_copyBuffer, 0.28 ms for the headline's 15 MB) that, in the pipelined
  #     runs above, executes beside the next pass's k_detect and slows it (profiles/r04_launch_hist_why_sc8.txt: a plain run
  #     with HSA_ENABLE_SDMA=0 shows the same slow mode without any profiler).  With one pass in flight nothing overlaps:
  #     this k_detect average is the one to hold against `roofline.isolated` of the unprofiled line
  rocprofv3 --kernel-trace --stats -d $W/kt2_$name -o kt2 -- $B $args --depth 1 --steps 10 --warmup 3 --min-time 0.2 > $W/kt2_$name.log 2>&1
  DB=$(find $W/kt2_$name -name '*.db' | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu --no-extra --no-hostfed $args --depth 1 --steps 10 --warmup 3 --min-time 0.2";
    echo "# (one pass in flight: the record copy -- a blit kernel under the profiler, SDMA otherwise -- never runs beside k_detect)";
    python $ROOT/tools/prof_summary.py "$DB" | grep -i "kernel \|adsb\|copyBuffer"; echo; echo "# the bench's own JSON line of this profiled run:";
    grep '"metric"' $W/kt2_$name.log | tail -1; } > $OUT/${R}_${name}_kernel_trace_blocking.txt 2>&1
  # 2. HBM traffic counters, one pass each
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace -f csv -d $W/pmc_${C}_$name -o p -- $B $args --steps 4 --warmup 1 --min-time 0 > $W/pmc_${C}_$name.log 2>&1
  done
  { echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python bench.py --no-cpu --no-extra --no-hostfed $args --steps 4 --warmup 1 --min-time 0   (KiB per launch)";
    python $ROOT/tools/pmc_summary.py $(find $W/pmc_FETCH_SIZE_$name -name '*counter_collection.csv' | head -1) \
                                      $(find $W/pmc_WRITE_SIZE_$name -name '*counter_collection.csv' | head -1);
    echo; grep '"metric"' $W/pmc_FETCH_SIZE_$name.log | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('# algorithmic bytes per launch: %d (samples per launch x bytes per sample)' % d['roofline']['algorithmic_bytes_per_launch'])"; } > $OUT/${R}_${name}_pmc_hbm.txt 2>&1
  # 3. the same workload unprofiled, same box
  (cd $ROOT && $S $W/$name.plain.clk -- $B $args > $OUT/${R}_${name}_bench.json 2>> $W/bench.err; cat $W/$name.plain.clk >> $OUT/${R}_${name}_bench.json)
  [ "${2:-}" = quick ] && break
done
[ -n "${ONLY:-}" ] && { ls -la $OUT; exit 0; }
# 4. SQ counters (waits / issue / instruction mix), headline workload
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -f csv -d $W/sq1 -o p -- $B --steps 4 --warmup 1 --min-time 0 > $W/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace -f csv -d $W/sq2 -o p -- $B --steps 4 --warmup 1 --min-time 0 > $W/sq2.log 2>&1
python $ROOT/tools/pmc_summary.py $(find $W/sq1 -name '*counter_collection.csv' | head -1) \
                                  $(find $W/sq2 -name '*counter_collection.csv' | head -1) > $OUT/${R}_pmc_sq_counters.txt 2>&1
# 4b. where k_detect's cycles go (per-class active cycles, LDS conflicts): int8 and complex64 at 2^30
(cd $ROOT && bash tools/sq_deep.sh gpurun_out/prof/${R}_sq_deep.txt > /dev/null 2>&1)
# 5. the full default bench line (CPU baselines, bit-match, host-fed per format, extra_configs), unprofiled
cd $ROOT
$S $W/full.clk -- python bench.py > $OUT/${R}_bench_unprofiled.json 2>> $W/bench.err
cat $W/full.clk >> $OUT/${R}_bench_unprofiled.json
tail -5 $W/bench.err > $OUT/${R}_bench_stderr_tail.txt
ls -la $OUT
