#!/bin/bash
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 120 python tools/dbg_small.py > gpurun_out/r05/dbg_small.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r05/gputest2.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/gputest2.txt
tail -8 gpurun_out/r05/gputest2.txt
timeout 1500 bash tools/r3_variants.sh "" "--format mag2" "--format sc16" "--format sc8" "--format cu8" "--fs 8e6 --bursts 6000 --log2n 28" "--fs 20e6 --log2n 28" "--mixed-df --log2n 28" > gpurun_out/r05/ab_record_wavefront.txt 2>&1
cat gpurun_out/r05/ab_record_wavefront.txt
