#!/bin/bash
# A/B on one box: the shipped library and every side copy under gr_adsb_amd/_variants/ (tools/kbench.py build ...), each
# through bench.py's headline leg.   bash tools/r3_variants.sh "<bench args>" ["<bench args>" ...]
ROOT=$(pwd)
pick() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-10s %-34s %9.1f Msps  %.4f ms/step  frac %.4f  iso %.4f' % (sys.argv[1], sys.argv[2], d['value'], d['ms_per_step'], r['frac'], r['isolated']['frac']))
" "$1" "$2"; }
A="--no-cpu --no-extra --no-hostfed --steps 20 --warmup 5 --min-time 0.25"
for cfg in "$@"; do
  python bench.py $A $cfg 2>/dev/null | pick shipped "$cfg"
  for f in gr_adsb_amd/_variants/libadsb_*.so; do
    [ -f "$f" ] || continue
    v=$(basename $f .so); v=${v#libadsb_}
    ADSB_HIP_LIB=$ROOT/$f python bench.py $A $cfg 2>/dev/null | pick $v "$cfg"
  done
done
