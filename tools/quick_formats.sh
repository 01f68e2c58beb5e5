#!/bin/bash
# One line per input format / signal config: value, ms/step, live and isolated roofline fraction (bench.py, headline leg only).
#   bash tools/quick_formats.sh [log2n] [extra bench args]
L=${1:-28}; shift
pick() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-28s %9.1f Msps  %.4f ms/step  frac %.4f  iso %.4f  bursts %s' % (sys.argv[1], d['value'], d['ms_per_step'], r['frac'], r['isolated']['frac'], d['config']['bursts_per_step_rank0']))
" "$1"; }
A="--no-cpu --no-extra --no-hostfed --steps 20 --warmup 5 --min-time 0.3 --log2n $L"
python bench.py $A "$@" 2>&1 | pick fc32_2msps
python bench.py $A --bursts 0 "$@" 2>&1 | pick fc32_quiet
python bench.py $A --fs 8e6 --bursts 6000 "$@" 2>&1 | pick fc32_8msps_dense
python bench.py $A --fs 20e6 "$@" 2>&1 | pick fc32_20msps
python bench.py $A --mixed-df "$@" 2>&1 | pick fc32_mixed_df
for f in mag2 sc16 sc8 cu8; do python bench.py $A --format $f "$@" 2>&1 | pick $f; done
