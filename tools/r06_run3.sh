#!/bin/bash
# Round 6, third GPU call: -m gpu on the library with the register-built masks and the exact-hint median, fuzz, the bench
# line, the GNU Radio path per chunk size.
export TMPDIR=/tmp
ROOT=$(pwd)
O=$ROOT/gpurun_out/r06
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputest3.txt 2>&1; echo "pytest rc $?" >> $O/gputest3.txt; tail -3 $O/gputest3.txt
timeout 400 python tools/fuzz_gpu.py 150 > $O/fuzz3.txt 2>&1; tail -3 $O/fuzz3.txt
timeout 600 python bench.py > $O/bench3.json 2> $O/bench3.err; echo "bench rc $?"
timeout 600 python tools/gr_latency.py 2e6 --small > $O/gr_latency3.txt 2>&1; cat $O/gr_latency3.txt
