#!/bin/bash
# Round 6: profile set of the uint8 dot-product instance (the bench's default uint8 leg) + the default bench line after the switch.
export TMPDIR=/tmp
ROOT=$(pwd)
O=$ROOT/gpurun_out/r06; mkdir -p $O
ONLY=fmt_cu8p bash tools/profile_round.sh r06 > $O/profile_cu8p.log 2>&1; tail -3 $O/profile_cu8p.log
cd $ROOT
python bench.py --no-cpu --no-extra --no-hostfed --format cu8 --cu8-generic 2>/dev/null | cut -c1-300
timeout 900 python bench.py > $O/bench_check2.json 2> $O/bench_check2.err; echo "bench rc $?"; head -c 400 $O/bench_check2.json
