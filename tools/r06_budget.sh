#!/bin/bash
# Where k_detect's instructions go, per phase and class (round 6, review item 1a).
#   CPU box:  bash tools/r06_budget.sh build      side copies with -DADSB_ABLATE=n (results INVALID: phases cut out)
#   GPU box:  bash tools/r06_budget.sh run OUT    SQ_INSTS_{VALU,SALU,LDS,BRANCH} + wave cycles + kernel time per copy and format
# ab1 = no burst records; ab2 = also no fall / centre / chip test; ab3 = also no rise list; ab4 = every tile takes the quiet
# path; ab11 = records without the median search; ab12 = records without the bit slices.  Differences between neighbours
# are the phases' executed instructions per 1024-sample tile (2^30 samples = 2^20 tiles).
export TMPDIR=/tmp
ROOT=$(pwd)
if [ "$1" = build ]; then
  # the hooks are not in the shipped source: tools/r06_ablate.patch puts them into a scratch copy of csrc/
  S=/tmp/ab_src; rm -rf $S; mkdir -p $S/gr_adsb_amd $S/include gr_adsb_amd/_variants
  cp -r gr_adsb_amd/csrc $S/gr_adsb_amd/ && cp include/adsb_hip.h $S/include/
  (cd $S && patch -p0 < $ROOT/tools/r06_ablate.patch) || exit 1
  for n in 1 2 3 4 11 12; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -pthread -DADSB_ABLATE=$n \
      $S/gr_adsb_amd/csrc/adsb_hip.hip -o gr_adsb_amd/_variants/libadsb_ab$n.so || exit 1
  done
  exit 0
fi
OUT=$ROOT/$2
mkdir -p $(dirname $OUT)
: > $OUT
cd /tmp
for cfg in "--log2n 30 --format sc8" "--log2n 30 --format cu8" "--log2n 30 --format sc16" "--log2n 30 --format mag2"; do
  for v in shipped ab1 ab2 ab3 ab4 ab11 ab12; do
    LIB=$ROOT/gr_adsb_amd/_variants/libadsb_$v.so; [ $v = shipped ] && LIB=$ROOT/gr_adsb_amd/libadsb_hip.so
    [ -f $LIB ] || continue
    for p in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
      rm -rf /tmp/bud
      ADSB_HIP_LIB=$LIB timeout 300 rocprofv3 --pmc $p --kernel-trace -f csv -d /tmp/bud -o p -- python $ROOT/bench.py --no-cpu --no-extra --no-hostfed --steps 3 --warmup 1 --min-time 0 $cfg > /tmp/bud.log 2>&1
      echo "== $v $cfg" >> $OUT
      python $ROOT/tools/pmc_summary.py $(find /tmp/bud -name '*counter_collection.csv' | head -1) | grep -A4 "k_detect" >> $OUT
    done
  done
done
