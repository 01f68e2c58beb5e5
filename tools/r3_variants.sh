#!/bin/bash
# A/B on one box: the shipped library and every side copy under gr_adsb_amd/_variants/ (tools/kbench.py build ...), each
# through bench.py's headline leg, with the shader clock the run saw (boxes differ: an instruction-bound format can run a
# third slower on one box than on the next while complex64 shows nothing).
#   bash tools/r3_variants.sh "<bench args>" ["<bench args>" ...]
ROOT=$(pwd)
A="--no-cpu --no-extra --no-hostfed --steps 20 --warmup 5 --min-time 0.25"
one() {   # name, cfg  (ADSB_HIP_LIB from the caller's environment)
  rm -f /tmp/r3v_clk.json
  python tools/smi_sampler.py /tmp/r3v_clk.json -- python bench.py $A $2 > /tmp/r3v_out.txt 2>/dev/null
  python - "$1" "$2" <<'PY'
import json, sys
try:
    clk = json.loads(open('/tmp/r3v_clk.json').readline())
except Exception:
    clk = {}
for l in open('/tmp/r3v_out.txt'):
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('%-10s %-34s %9.1f Msps  %.4f ms/step  frac %.4f  iso %.4f  untimed-ctx %s ms/step  sclk %s' % (sys.argv[1], sys.argv[2], d['value'], d['ms_per_step'], r['frac'], r['isolated']['frac'], d.get('timing', {}).get('ms_per_step_untimed_ctx'), clk.get('sclk', {}).get('median')))
PY
}
for cfg in "$@"; do
  one shipped "$cfg"
  for f in gr_adsb_amd/_variants/libadsb_*.so; do
    [ -f "$f" ] || continue
    v=$(basename $f .so); v=${v#libadsb_}
    ADSB_HIP_LIB=$ROOT/$f one $v "$cfg"
  done
done
