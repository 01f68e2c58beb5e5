#!/bin/bash
# Round-2 A/B on one box: the round-1 tree (k_burst re-reading the stream) vs the fused-record build and its
# tuning variants, each through its own bench.py (pipelined, 2^30 samples/step), plus the isolated kernel numbers.
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
        print('%-8s value %.0f Msps  ms/step %.4f  kernel_ms %.4f frac %.4f  iso_ms %s iso_frac %s  bursts %s' % (sys.argv[1], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], (r.get('isolated') or {}).get('kernel_ms'), (r.get('isolated') or {}).get('frac'), d['config'].get('bursts_per_step_rank0')))
" "$1"; }
for rep in 1 2; do
  (cd gr_adsb_amd/_variants/r1_tree && python bench.py --no-cpu --steps 30 --warmup 5 "$@" 2>/dev/null | pick r1) >> $L
  python bench.py --no-cpu --steps 30 --warmup 5 "$@" 2>/dev/null | pick shipped >> $L
  for v in nt ei8 ei6 ei2 ntei8; do
    ADSB_HIP_LIB=$ROOT/gr_adsb_amd/_variants/libadsb_$v.so python bench.py --no-cpu --steps 30 --warmup 5 "$@" 2>/dev/null | pick $v >> $L
  done
done
cat $L
