#!/bin/bash
export TMPDIR=/tmp
ROOT=$(pwd)
O=$ROOT/gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "page_locked_once or dropin or paired or registered or gnuradio" > $O/gputest10.txt 2>&1; echo "pytest rc $?" >> $O/gputest10.txt; tail -4 $O/gputest10.txt
timeout 900 python tools/gr_latency.py 2e6 --breakdown > $O/gr_latency_pin.txt 2>&1; grep -v amdgpu $O/gr_latency_pin.txt
timeout 900 python tools/gr_latency.py 2e6 --no-pin > $O/gr_latency_nopin.txt 2>&1; grep -v amdgpu $O/gr_latency_nopin.txt | grep paired
