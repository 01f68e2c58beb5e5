#!/bin/bash
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r05/gputest1.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/gputest1.txt
tail -5 gpurun_out/r05/gputest1.txt
timeout 700 bash tools/slot_pmc.sh gpurun_out/r05/slot_pmc_fc32.txt fc32 18 > /dev/null 2>&1
timeout 300 python tools/hostfed_repeat.py 10 > gpurun_out/r05/hostfed_repeat.txt 2>&1
timeout 600 python bench.py > gpurun_out/r05/bench_start.json 2> gpurun_out/r05/bench_start.err
tail -c 600 gpurun_out/r05/bench_start.json
