#!/bin/bash
# A/B on one box: the shipped library and every tuning variant under gr_adsb_amd/_variants/ (tools/kbench.py build ...),
# each through bench.py (pipelined, 2^30 samples/step) -- and the round-1 tree if it was exported there.
# Run from the repo root on the GPU box:  bash tools/r2_variants.sh [extra bench args]
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
L=$OUT/r2_variants.log
: > $L
pick() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-8s value %.0f Msps  ms/step %.4f  kernel_ms %.4f frac %.4f  iso_ms %s iso_frac %s  bursts %s grid %s' % (sys.argv[1], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], (r.get('isolated') or {}).get('kernel_ms'), (r.get('isolated') or {}).get('frac'), d['config'].get('bursts_per_step_rank0'), d['config'].get('detect_grid')))
" "$1"; }
A="--no-cpu --no-extra --no-hostfed --steps 20 --warmup 5 --min-time 0.3"
for rep in 1 2; do
  if [ -d gr_adsb_amd/_variants/r1_tree ]; then (cd gr_adsb_amd/_variants/r1_tree && python bench.py --no-cpu --steps 30 --warmup 5 "$@" 2>/dev/null | pick r1) >> $L; fi
  python bench.py $A "$@" 2>/dev/null | pick shipped >> $L
  for f in gr_adsb_amd/_variants/libadsb_*.so; do
    [ -f "$f" ] || continue
    v=$(basename $f .so); v=${v#libadsb_}
    ADSB_HIP_LIB=$ROOT/$f python bench.py $A "$@" 2>/dev/null | pick $v >> $L
  done
done
cat $L
