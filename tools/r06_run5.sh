#!/bin/bash
export TMPDIR=/tmp
ROOT=$(pwd)
O=$ROOT/gpurun_out/r06
mkdir -p $O
bash tools/r3_variants.sh "--log2n 30 --format mag2" "--log2n 30 --format sc16" "--log2n 30" "--log2n 30 --format sc8" "--log2n 28 --fs 8e6 --bursts 6000" "--log2n 30 --format mag2" "--log2n 30" > $O/ab_stream_prio.txt 2>&1
cat $O/ab_stream_prio.txt
