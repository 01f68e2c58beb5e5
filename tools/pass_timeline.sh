#!/bin/bash
# Kernel timeline of submitted mid-size passes (one stream per pipeline slot): rocprofv3 --kernel-trace over a dozen
# device-resident passes of 2^LOG2N samples at 20 Msps, three in flight; the last dispatches with start offset, duration and
# gap to the previous END (negative = the kernel started while the previous one was still running: passes overlap).
#   bash tools/pass_timeline.sh OUT [LOG2N]          (GPU box only)
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/$1; LG=${2:-25}
mkdir -p $(dirname $OUT); rm -rf /tmp/ptl; cd /tmp
cat > /tmp/ptl_run.py <<PY
import sys, torch
sys.path.insert(0, "$ROOT")
from gr_adsb_amd import _native, modulator as M
n = 1 << $LG
iq = M.synth_iq_torch(n, 20e6, 1000, 3, torch.device("cuda:0"))
torch.cuda.synchronize()
ctx = _native.Context(20e6, 0.01)
for _ in range(3):
    ctx.process_format_device(_native.FMT_FC32, iq.data_ptr(), n, 0, fetch=False)
pend = []
for _ in range(14):
    pend.append(ctx.submit_format_device(_native.FMT_FC32, iq.data_ptr(), n, 0))
    if len(pend) == 3:
        ctx.wait(pend.pop(0), fetch=False)
while pend:
    ctx.wait(pend.pop(0), fetch=False)
PY
rocprofv3 --kernel-trace -d /tmp/ptl -o t -- python /tmp/ptl_run.py > /tmp/ptl.log 2>&1
DB=$(find /tmp/ptl -name '*.db' | head -1)
{ echo "# rocprofv3 --kernel-trace: 14 submitted passes of 2^$LG complex64 samples (20 Msps), three in flight, untimed context";
  echo "# start offset, duration, gap to the previous kernel's END (negative: started while it was still running)";
  python $ROOT/tools/prof_timeline.py "$DB" 40; } > $OUT 2>&1
cat $OUT
