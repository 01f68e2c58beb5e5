"""Per-LAUNCH durations of k_detect for one workload, pipelined like bench.py (three passes in flight): the HIP-event
duration of every launch in launch order (adsb_detect_history).  Run plain, and run under `rocprofv3 --kernel-trace -f csv`
(tools/prof_modes.sh does both and holds the two sequences against each other launch by launch): is the profiler's slow
mode every second launch, the first N launches, periodic, or random?     GPU box only.
    python tools/launch_hist.py --format sc8 [--log2n 30] [--steps 60] [--single-stream] [--tag plain]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gr_adsb_amd import _native  # noqa: E402
from gr_adsb_amd import modulator as M  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--format", default="sc8", choices=["fc32", "mag2", "sc16", "sc8", "cu8"])
ap.add_argument("--log2n", type=int, default=30)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--depth", type=int, default=3)
ap.add_argument("--single-stream", action="store_true")
ap.add_argument("--tag", default="plain")
ap.add_argument("--no-torch-after-setup", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
n = 1 << a.log2n
blk = 1 << 24
iq = M.synth_iq_torch(blk, 2e6, 1000, 1, dev)                      # BASELINE config 2's signal, tiled
fmt = {"fc32": _native.FMT_FC32, "mag2": _native.FMT_MAG2, "sc16": _native.FMT_SC16, "sc8": _native.FMT_SC8, "cu8": _native.FMT_CU8}[a.format]
if a.format == "fc32":
    data = iq.repeat(n // blk, 1).contiguous()
elif a.format == "mag2":
    data = (iq[:, 0] * iq[:, 0] + iq[:, 1] * iq[:, 1]).repeat(n // blk).contiguous()
elif a.format == "sc16":
    data = torch.clamp(torch.round(iq * (32767.0 / 4.0)), -32767, 32767).to(torch.int16).repeat(n // blk, 1).contiguous()
elif a.format == "sc8":
    data = torch.clamp(torch.round(iq * (127.0 / 4.0)), -127, 127).to(torch.int8).repeat(n // blk, 1).contiguous()
else:
    data = (torch.clamp(torch.round(iq * (127.0 / 4.0)), -127, 127) + 128).to(torch.uint8).repeat(n // blk, 1).contiguous()
torch.cuda.synchronize()
flags = _native.FLAG_TIMING | (_native.FLAG_SINGLE_STREAM if a.single_stream else 0)
ctx = _native.Context(2e6, 0.01, flags=flags)
if a.format in ("sc8", "cu8"):
    ctx.set_format_scale(fmt, float(np.float32(4.0 / 127.0)))
if a.format == "sc16":
    ctx.set_format_scale(fmt, float(np.float32(4.0 / 32767.0)))
for _ in range(4):
    ctx.process_format_device(fmt, data.data_ptr(), n, 0, fetch=False)
ctx.reset_stats()
pend = []
t0 = time.perf_counter()
for k in range(a.steps):
    pend.append(ctx.submit_format_device(fmt, data.data_ptr(), n))
    if len(pend) == a.depth:
        ctx.wait(pend.pop(0), fetch=False)
while pend:
    ctx.wait(pend.pop(0), fetch=False)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
h = ctx.detect_history()
print(json.dumps({"tag": a.tag, "format": a.format, "log2n": a.log2n, "steps": a.steps, "single_stream": a.single_stream,
                  "wall_ms_per_step": round(wall / a.steps * 1e3, 4), "hip_event_ms": [round(float(v), 4) for v in h]}))
