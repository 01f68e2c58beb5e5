#!/bin/bash
# Round 6: -m gpu, smoke, bounded fuzz and the default bench line on the tree as it stands (one box).
export TMPDIR=/tmp
ROOT=$(pwd)
O=$ROOT/gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gputest_check.txt 2>&1; echo "pytest rc $?" >> $O/gputest_check.txt; tail -3 $O/gputest_check.txt
timeout 400 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_check.txt 2>&1; tail -2 $O/smoke_check.txt
timeout 500 python tools/fuzz_gpu.py 120 > $O/fuzz_check.txt 2>&1; tail -2 $O/fuzz_check.txt
timeout 900 python bench.py > $O/bench_check.json 2> $O/bench_check.err; echo "bench rc $?"; head -c 600 $O/bench_check.json
