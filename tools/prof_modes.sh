#!/bin/bash
# Why do rocprofv3's k_detect averages exceed the bench's HIP-event figures?  Per-launch sequences of one workload, plain and
# under the profiler in several modes, launch by launch.   bash tools/prof_modes.sh rNN [formats...]   (GPU box)
#   plain                              HIP events only
#   kt      rocprofv3 --kernel-trace -f csv                    (no --stats, no other domain)
#   kts     rocprofv3 --kernel-trace --stats -f csv
#   pmc     rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -f csv   (cycles per launch: a second clock)
# pipelined (three passes in flight, tail on its own stream) and single-stream.  Text summaries -> gpurun_out/prof/.
set -u
R=${1:-r04}; shift
FMTS=${@:-sc8 fc32}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof; mkdir -p $OUT
export TMPDIR=/tmp
W=/tmp/adsb_modes; rm -rf $W; mkdir -p $W
cd /tmp
for f in $FMTS; do
  for ss in "" "--single-stream"; do
    tag=${f}${ss:+_single}
    python $ROOT/tools/launch_hist.py --format $f $ss --tag plain > $W/$tag.plain.json 2> $W/$tag.plain.err
    python $ROOT/tools/launch_hist.py --format $f $ss --tag plain2 > $W/$tag.plain2.json 2>> $W/$tag.plain.err
    rocprofv3 --kernel-trace -f csv -d $W/kt_$tag -o kt -- python $ROOT/tools/launch_hist.py --format $f $ss --tag kt 2> $W/$tag.kt.err | grep '^{' > $W/$tag.kt.json
    rocprofv3 --kernel-trace --stats -f csv -d $W/kts_$tag -o kts -- python $ROOT/tools/launch_hist.py --format $f $ss --tag kts 2> $W/$tag.kts.err | grep '^{' > $W/$tag.kts.json
    rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -f csv -d $W/pmc_$tag -o pmc -- python $ROOT/tools/launch_hist.py --format $f $ss --tag pmc 2> $W/$tag.pmc.err | grep '^{' > $W/$tag.pmc.json
    {
      echo "## plain run repeated (run-to-run spread without any profiler)"
      python - $W/$tag.plain.json $W/$tag.plain2.json <<'PY'
import json, sys, statistics
for p in sys.argv[1:]:
    d = json.load(open(p)); v = sorted(d["hip_event_ms"])
    print("%-8s wall %.4f ms/step   HIP events: min %.4f median %.4f max %.4f" % (d["tag"], d["wall_ms_per_step"], v[0], statistics.median(v), v[-1]))
PY
      echo; echo "## rocprofv3 --kernel-trace -f csv (nothing else)"
      python $ROOT/tools/launch_hist_report.py $W/$tag.plain.json $W/$tag.kt.json $(find $W/kt_$tag -name '*kernel_trace.csv' | head -1) | head -40
      echo; echo "## rocprofv3 --kernel-trace --stats -f csv"
      python $ROOT/tools/launch_hist_report.py $W/$tag.plain.json $W/$tag.kts.json $(find $W/kts_$tag -name '*kernel_trace.csv' | head -1) | head -16
      echo; echo "## rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -f csv"
      python $ROOT/tools/launch_hist_report.py $W/$tag.plain.json $W/$tag.pmc.json $(find $W/pmc_$tag -name '*kernel_trace.csv' | head -1) $(find $W/pmc_$tag -name '*counter_collection.csv' | head -1) | head -30
    } > $OUT/${R}_launch_hist_$tag.txt 2>&1
  done
done
tail -3 $W/*.err > $OUT/${R}_launch_hist_stderr.txt 2>&1
ls -la $OUT | tail -8
