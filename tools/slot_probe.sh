#!/bin/bash
# k_detect's HIP-event duration per pipeline SLOT (launch index mod ADSB_MAX_IN_FLIGHT = 3): the slots differ only in the
# addresses of their output buffers.  Shipped library, then every side copy under gr_adsb_amd/_variants/.
#   bash tools/slot_probe.sh [depth] [format]
D=${1:-1}; F=${2:-fc32}
one() {
python tools/launch_hist.py --format $F --steps 60 --depth $D --tag $1 2>/dev/null | python -c "
import json,sys,statistics as st
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['hip_event_ms']
print('%-10s depth $D $F  mean %.4f  per slot' % ('$1', st.mean(h)), ['%.4f' % st.mean(h[r::3]) for r in range(3)], ' spread %.1f %%' % (100*(max(st.mean(h[r::3]) for r in range(3))/min(st.mean(h[r::3]) for r in range(3))-1)))"
}
one shipped
for f in gr_adsb_amd/_variants/libadsb_*.so; do
  [ -f "$f" ] || continue
  v=$(basename $f .so); v=${v#libadsb_}
  ADSB_HIP_LIB=$PWD/$f one $v
done
