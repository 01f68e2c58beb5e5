#!/bin/bash
# Memory-system counters of k_detect PER PIPELINE SLOT (round-4 open item: the kernel's duration depends on which slot's
# output buffers it writes).  tools/slot_probe.py runs blocking passes of the headline workload, slot = launch index mod 3;
# every counter group is its own rocprofv3 --pmc pass (--kernel-trace only), tools/slot_pmc_report.py averages each
# counter per slot and prints the kernel duration per slot of the same pass beside it.
#   bash tools/slot_pmc.sh OUT [format] [steps]
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/$1; FMT=${2:-fc32}; STEPS=${3:-18}
mkdir -p $(dirname $OUT)
cd /tmp
GROUPS_=(
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
 "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_IB_STALL_sum"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_BUBBLE_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_LATENCY_FIFO_FULL_sum"
 "TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL"
 "GRBM_GUI_ACTIVE TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"
)
: > $OUT
echo "# plain run (no profiler): k_detect HIP-event duration per slot" >> $OUT
python $ROOT/tools/slot_probe.py --format $FMT --steps $STEPS --tag plain >> $OUT 2>/dev/null
for g in "${GROUPS_[@]}"; do
  rm -rf /tmp/spmc
  rocprofv3 --pmc $g --kernel-trace -f csv -d /tmp/spmc -o p -- python $ROOT/tools/slot_probe.py --format $FMT --steps $STEPS --tag pmc > /tmp/spmc.log 2>&1
  echo "== --pmc $g" >> $OUT
  grep "k_detect (HIP events)" /tmp/spmc.log >> $OUT
  python $ROOT/tools/slot_pmc_report.py "$(find /tmp/spmc -name '*counter_collection.csv' | head -1)" "$(find /tmp/spmc -name '*kernel_trace.csv' | head -1)" >> $OUT 2>&1
done
cat $OUT
