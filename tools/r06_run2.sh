#!/bin/bash
# Round 6, second GPU call: -m gpu on the shipped library and on the all-experiments side copy, the A/B of the kernel
# experiments for the 8-bit formats, the phase x class instruction budget.
export TMPDIR=/tmp
ROOT=$(pwd)
O=$ROOT/gpurun_out/r06
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gputest2.txt 2>&1; echo "pytest rc $?" >> $O/gputest2.txt; tail -3 $O/gputest2.txt
ADSB_HIP_LIB=$ROOT/gr_adsb_amd/_variants/libadsb_e7.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/gputest2_e7.txt 2>&1; echo "pytest rc $?" >> $O/gputest2_e7.txt; tail -3 $O/gputest2_e7.txt
mkdir -p /tmp/hold && mv gr_adsb_amd/_variants/libadsb_ab*.so /tmp/hold/ 2>/dev/null
bash tools/r3_variants.sh "--log2n 30 --format sc8" "--log2n 30 --format cu8" "--log2n 30 --format sc8" "--log2n 30 --format cu8" > $O/ab_exp_8bit.txt 2>&1
cat $O/ab_exp_8bit.txt
mv /tmp/hold/libadsb_ab*.so gr_adsb_amd/_variants/ 2>/dev/null
bash tools/r06_budget.sh run gpurun_out/r06/budget_counters.txt
tail -40 $O/budget_counters.txt
