#!/bin/bash
export TMPDIR=/tmp
ROOT=$(pwd)
O=$ROOT/gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "int8 or uint8 or 8bit or every or fuzz or format" > $O/gputest9.txt 2>&1; echo "pytest rc $?" >> $O/gputest9.txt; tail -3 $O/gputest9.txt
bash tools/r3_variants.sh "--log2n 30 --format cu8" "--log2n 30 --format cu8 --cu8-pow2" "--log2n 30 --format sc8" "--log2n 30 --format cu8" "--log2n 30 --format cu8 --cu8-pow2" > $O/ab_cu8_pow2.txt 2>&1
cat $O/ab_cu8_pow2.txt
