timeout 300 python -m pytest tests -m gpu -q 2>&1 | tail -2
for i in 1 2; do python bench.py --no-cpu | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['isolated']['kernel_ms'])"; done
python bench.py --no-cpu --fs 8e6 --bursts 6000 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8M', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['isolated']['kernel_ms'])"
cd /tmp; export TMPDIR=/tmp
ADSB_TAIL_STREAM=0 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python /root/repo/bench.py --no-cpu > /tmp/kt.log 2>&1
python /root/repo/tools/prof_summary.py $(find /tmp/kt -name "*.db" | head -1) | grep "adsb" | cut -c1-30,60-140
