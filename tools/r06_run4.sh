#!/bin/bash
export TMPDIR=/tmp
ROOT=$(pwd)
O=$ROOT/gpurun_out/r06
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputest4.txt 2>&1; echo "pytest rc $?" >> $O/gputest4.txt; tail -3 $O/gputest4.txt
ADSB_HIP_LIB=$ROOT/gr_adsb_amd/_variants/libadsb_pkd2.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/gputest4_pkd2.txt 2>&1; echo "pytest rc $?" >> $O/gputest4_pkd2.txt; tail -3 $O/gputest4_pkd2.txt
bash tools/r3_variants.sh "--log2n 30 --format sc8" "--log2n 30 --format sc8 --sc8-generic" "--log2n 30 --format cu8" "--log2n 30 --format sc8" "--log2n 30 --format sc8 --sc8-generic" "--log2n 30 --format cu8" > $O/ab_pk_d2.txt 2>&1
cat $O/ab_pk_d2.txt
timeout 900 python tools/gr_latency.py 2e6 --breakdown > $O/gr_latency4.txt 2>&1; grep -v amdgpu $O/gr_latency4.txt | tail -12
