#!/bin/bash
# The DEVICE code (gr_adsb_amd/csrc/adsb_device.h, every kernel) under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU:
# the SIMT emulator build of tests/sim with -fsanitize=address,undefined, driven by the emulator's own test files.  (GPU
# sanitizers are not available on the pool; this is where out-of-bounds LDS / global accesses and UB in the kernels would show.)
#   bash tools/sim_sanitize.sh            (CPU box; ~12 minutes)
set -e
ROOT=$(pwd)
SAN=/tmp/libadsb_sim_asan.so
g++ -O1 -g -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unknown-pragmas -fsanitize=address,undefined -fno-sanitize-recover=undefined \
    tests/sim/sim_driver.cpp -o $SAN
cp tests/sim/libadsb_sim.so /tmp/libadsb_sim_plain.so 2>/dev/null || true
cp $SAN tests/sim/libadsb_sim.so; touch tests/sim/libadsb_sim.so
trap 'cp /tmp/libadsb_sim_plain.so tests/sim/libadsb_sim.so 2>/dev/null; touch tests/sim/libadsb_sim.so' EXIT
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
  python -m pytest tests/test_sim_kernels.py tests/test_sim_property.py -x -q -p no:cacheprovider
