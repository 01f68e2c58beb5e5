#!/bin/bash
export TMPDIR=/tmp
ROOT=$(pwd)
O=$ROOT/gpurun_out/r06; mkdir -p $O
for v in shipped d2hstream shipped d2hstream; do
  LIB=$ROOT/gr_adsb_amd/libadsb_hip.so; [ $v = shipped ] || LIB=$ROOT/gr_adsb_amd/_variants/libadsb_$v.so
  ADSB_HIP_LIB=$LIB timeout 400 python bench.py --no-cpu --no-hostfed 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        c = json.loads(l)['config']
        print('$v', 'head ms %.4f untimed %.4f |' % (json.loads(l)['ms_per_step'], c['ms_per_step_untimed_ctx']), ' | '.join('%s %.4f [%.4f]' % (k, c[k + '_ms'], c[k + '_ms_untimed_ctx']) for k in ('cfg3', 'cfg4', 'cfg5', 'mag2', 'sc16', 'sc8', 'sc8g', 'cu8')))
"
done > $O/ab_copy_stream.txt 2>&1
cat $O/ab_copy_stream.txt
