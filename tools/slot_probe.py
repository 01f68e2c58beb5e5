"""k_detect per pipeline SLOT: blocking passes of the headline workload, host wall time per call AND the HIP-event duration of
k_detect, averaged per slot (call index mod ADSB_MAX_IN_FLIGHT).  The slots differ only in their output buffers / events.
    python tools/slot_probe.py [--format fc32] [--steps 60] [--log2n 30]          (GPU box only)"""
import argparse
import os
import statistics as st
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gr_adsb_amd import _native  # noqa: E402
from gr_adsb_amd import modulator as M  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--format", default="fc32", choices=["fc32", "sc8"])
ap.add_argument("--log2n", type=int, default=30)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--tag", default="")
ap.add_argument("--inputs", type=int, default=1, help="copies of the input at different addresses, used in turn (does the INPUT's placement matter?)")
a = ap.parse_args()
dev = torch.device("cuda:0")
n = 1 << a.log2n
blk = 1 << 24
iq = M.synth_iq_torch(blk, 2e6, 1000, 1, dev)
if a.format == "fc32":
    fmt, data = _native.FMT_FC32, iq.repeat(n // blk, 1).contiguous()
else:
    fmt, data = _native.FMT_SC8, torch.clamp(torch.round(iq * (128.0 / 4.0)), -127, 127).to(torch.int8).repeat(n // blk, 1).contiguous()
datas = [data] + [data.clone() for _ in range(a.inputs - 1)]
torch.cuda.synchronize()
ctx = _native.Context(2e6, 0.01, flags=_native.FLAG_TIMING)
if a.format == "sc8":
    ctx.set_format_scale(fmt, 4.0 / 128.0)
for _ in range(6):
    ctx.wait(ctx.submit_format_device(fmt, data.data_ptr(), n), fetch=False)
ctx.reset_stats()
wall = []
S = 3
which = []
for k in range(a.steps):
    d = datas[(k // S) % len(datas)]                  # the input changes every S calls: slot and input vary independently
    which.append((k // S) % len(datas))
    t0 = time.perf_counter()
    ctx.wait(ctx.submit_format_device(fmt, d.data_ptr(), n), fetch=False)
    wall.append((time.perf_counter() - t0) * 1e3)
h = [float(v) for v in ctx.detect_history()][-a.steps:]
print("%-16s %s  k_detect (HIP events) per slot %s   host wall per call per slot %s" % (
    a.tag, a.format, ["%.4f" % st.mean(h[r::S]) for r in range(S)], ["%.4f" % st.median(wall[r::S]) for r in range(S)]))
if len(datas) > 1:
    print("%-16s %s  k_detect (HIP events) per INPUT copy %s   addresses %s" % (
        a.tag, a.format, ["%.4f" % st.mean([v for v, w in zip(h, which) if w == i]) for i in range(len(datas))],
        ["%#x" % d.data_ptr() for d in datas]))
