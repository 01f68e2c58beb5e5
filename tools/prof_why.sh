#!/bin/bash
# Narrowing down rocprofv3's slow mode (see tools/prof_modes.sh): sc8 at 2^30, launch by launch.   bash tools/prof_why.sh rNN
set -u
R=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof; mkdir -p $OUT
export TMPDIR=/tmp
W=/tmp/adsb_why; rm -rf $W; mkdir -p $W
cd /tmp
L="python $ROOT/tools/launch_hist.py --format sc8"
sumline() { python - "$@" <<'PY'
import json, sys, statistics
d = json.load(open(sys.argv[1])); v = d["hip_event_ms"]; s = sorted(v)
print("%-34s wall %.4f ms/step  HIP events: first six %s | min %.4f median %.4f max %.4f | last ten mean %.4f" % (
    sys.argv[2], d["wall_ms_per_step"], " ".join("%.3f" % x for x in v[:6]), s[0], statistics.median(s), s[-1], sum(v[-10:]) / 10))
PY
}
{
  $L --tag plain > $W/a.json 2>/dev/null; sumline $W/a.json "plain, 3 in flight"
  HSA_ENABLE_SDMA=0 $L --tag nosdma > $W/b.json 2>/dev/null; sumline $W/b.json "plain, HSA_ENABLE_SDMA=0"
  $L --depth 1 --tag plain_d1 > $W/c.json 2>/dev/null; sumline $W/c.json "plain, blocking calls"
  rocprofv3 --kernel-trace -f csv -d $W/k1 -o k -- $L --depth 1 --tag kt_d1 2>/dev/null | grep '^{' > $W/d.json; sumline $W/d.json "kernel-trace, blocking calls"
  rocprofv3 --kernel-trace -f csv -d $W/k2 -o k -- $L --steps 240 --tag kt_240 2>/dev/null | grep '^{' > $W/e.json; sumline $W/e.json "kernel-trace, 240 launches"
  HSA_ENABLE_SDMA=0 rocprofv3 --kernel-trace -f csv -d $W/k3 -o k -- $L --tag kt_nosdma 2>/dev/null | grep '^{' > $W/f.json; sumline $W/f.json "kernel-trace, HSA_ENABLE_SDMA=0"
  rocprofv3 --kernel-trace --memory-copy-trace -f csv -d $W/k4 -o k -- $L --tag kt_mc 2>/dev/null | grep '^{' > $W/g.json; sumline $W/g.json "kernel-trace + memory-copy-trace"
  GPU_MAX_HW_QUEUES=1 rocprofv3 --kernel-trace -f csv -d $W/k5 -o k -- $L --tag kt_q1 2>/dev/null | grep '^{' > $W/h.json; sumline $W/h.json "kernel-trace, GPU_MAX_HW_QUEUES=1"
  echo "# kernels in the 240-launch trace that are not ours (name, calls, total ms):"
  python - $(find $W/k2 -name '*kernel_trace.csv' | head -1) <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:70]
    acc[k][0] += 1; acc[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print("   %-70s %6d %10.3f" % (k, n, t))
PY
  echo "# memory copies seen with --memory-copy-trace (direction, count, total ms, bytes):"
  python - $(find $W/k4 -name '*memory_copy_trace.csv' | head -1) <<'PY'
import csv, sys, collections
try:
    acc = collections.defaultdict(lambda: [0, 0.0, 0])
    for r in csv.DictReader(open(sys.argv[1])):
        a = acc[r.get("Direction", "?")]
        a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; a[2] += int(r.get("Size", 0) or 0)
    for k, v in acc.items():
        print("   ", k, v)
except Exception as e:
    print("   (none)", e)
PY
} > $OUT/${R}_launch_hist_why_sc8.txt 2>&1
cat $OUT/${R}_launch_hist_why_sc8.txt
