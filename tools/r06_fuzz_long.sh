#!/bin/bash
# Round 6: long fuzz on the GPU with fresh seeds -- the general campaign (tools/fuzz_gpu.py) and the 8-bit one aimed at the round's changes.
export TMPDIR=/tmp
ROOT=$(pwd)
O=$ROOT/gpurun_out/r06; mkdir -p $O
timeout 500 python tools/fuzz_sim_8bit.py 300 424242 gpu > $O/fuzz8_gpu.txt 2>&1; tail -2 $O/fuzz8_gpu.txt
timeout 700 python tools/fuzz_gpu.py 420 500000 > $O/fuzz_long.txt 2>&1; tail -2 $O/fuzz_long.txt
