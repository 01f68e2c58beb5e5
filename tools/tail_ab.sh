#!/bin/bash
# Tail kernels (k_order, k_resolve, k_count, k_compact) of the shipped library and of every side copy under
# gr_adsb_amd/_variants/, each timed ALONE: rocprofv3 --kernel-trace --stats over blocking passes (--depth 1) of the headline.
#   bash tools/tail_ab.sh OUT ["bench args"]
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/$1; shift
ARGS="$1"
cd /tmp
: > $OUT
one() {
  rm -rf /tmp/tab
  rocprofv3 --kernel-trace --stats -d /tmp/tab -o t -- python $ROOT/bench.py --no-cpu --no-extra --no-hostfed $ARGS --depth 1 --steps 10 --warmup 3 --min-time 0.2 > /tmp/tab.log 2>&1
  echo "== $1" >> $OUT
  python $ROOT/tools/prof_summary.py "$(find /tmp/tab -name '*.db' | head -1)" | grep -i "kernel \|adsb" >> $OUT
}
one shipped
for f in $ROOT/gr_adsb_amd/_variants/libadsb_*.so; do
  [ -f "$f" ] || continue
  v=$(basename $f .so); v=${v#libadsb_}
  ADSB_HIP_LIB=$f one $v
done
cat $OUT
