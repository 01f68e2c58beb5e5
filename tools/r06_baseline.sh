#!/bin/bash
# Round 6, first GPU call: the whole -m gpu suite on the new library, a default bench line, config 3 in five fresh
# processes (live / isolated / untimed side by side), the instruction-class counters of the shipped k_detect per format,
# per-channel TCC requests of one process (slow-slot item).
#   bash tools/r06_baseline.sh
export TMPDIR=/tmp
ROOT=$(pwd)
O=$ROOT/gpurun_out/r06
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputest1.txt 2>&1; echo "pytest rc $?" >> $O/gputest1.txt
tail -3 $O/gputest1.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
for i in 1 2 3 4 5; do
  timeout 300 python bench.py --no-cpu --no-hostfed > $O/cfg3_run$i.json 2>/dev/null
done
python - <<'PY' > $O/cfg3_five_runs.txt
import json, glob
print("# five fresh processes of `python bench.py --no-cpu --no-hostfed` on one box: headline and configs 3/4/5, timed (k_detect in line) vs isolated vs untimed context")
for f in sorted(glob.glob('%s/cfg3_run*.json' % '/root/repo/gpurun_out/r06')):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l); c = d['config']
            print(f.split('/')[-1], 'head frac %.4f iso %.4f ms %.4f untimed %.4f |' % (d['roofline']['frac'], c['roofline_frac_isolated'], d['ms_per_step'], c['ms_per_step_untimed_ctx']),
                  ' | '.join('%s frac %.4f iso %.4f ms %.4f untimed %.4f' % (k, c[k + '_frac'], c[k + '_frac_isolated'], c[k + '_ms'], c[k + '_ms_untimed_ctx']) for k in ('cfg3', 'cfg4', 'cfg5')))
PY
cat $O/cfg3_five_runs.txt
# instruction classes / LDS conflicts of the shipped kernels
cd /tmp
PASSES=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"
        "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES")
: > $O/sq_shipped.txt
for cfg in "--log2n 30 --format sc8" "--log2n 30 --format cu8" "--log2n 30 --format sc16" "--log2n 30 --format mag2" "--log2n 30"; do
  echo "== $cfg" >> $O/sq_shipped.txt
  for p in "${PASSES[@]}"; do
    rm -rf /tmp/sqd
    timeout 300 rocprofv3 --pmc $p --kernel-trace -f csv -d /tmp/sqd -o p -- python $ROOT/bench.py --no-cpu --no-extra --no-hostfed --steps 3 --warmup 1 --min-time 0 $cfg > /tmp/sqd.log 2>&1
    python $ROOT/tools/pmc_summary.py $(find /tmp/sqd -name '*counter_collection.csv' | head -1) | grep -A4 "k_detect" >> $O/sq_shipped.txt
  done
done
cat $O/sq_shipped.txt
# per-channel TCC requests, one process, slots 0/1/2
rm -rf /tmp/chn
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ --kernel-trace -f json -d /tmp/chn -o p -- python $ROOT/tools/slot_probe.py --steps 9 --tag chn > $O/chn.log 2>&1
ls -la $(find /tmp/chn -type f) >> $O/chn.log
J=$(find /tmp/chn -name '*.json' | head -1)
[ -n "$J" ] && gzip -c "$J" > $O/chn_results.json.gz
ls -la $O
