"""Wall time per pipelined pass (three in flight, context WITHOUT ADSB_FLAG_TIMING: the product default) for a list of
workloads and pass sizes.  Round 5 chose the stream arrangement and the 8-bit formats' workgroup shape with it: one process
per variant (ADSB_HIP_LIB = a side copy built by tools/kbench.py), results in profiles/r05_ab_queue_arrangements.txt,
r05_ab_8bit_workgroup_shape_and_schedule.txt, r05_ab_workgroup_shape_by_size_fc32.txt.
    python tools/sched_ab.py [tag]            (GPU box only)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gr_adsb_amd import _native  # noqa: E402
from gr_adsb_amd import modulator as M  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else ""
dev = torch.device("cuda:0")
blk = 1 << 24


def stream(fs, bursts, log2n, seed):
    n = 1 << log2n
    if n <= (1 << 28):
        return M.synth_iq_torch(n, fs, bursts, seed, dev)
    return M.synth_iq_torch(blk, fs, bursts, seed, dev).repeat(n // blk, 1).contiguous()


def run(name, fmt, data, n, fs, scale=None, reps=5, steps=20):
    ctx = _native.Context(fs, 0.01)
    if scale is not None:
        ctx.set_format_scale(fmt, scale)
    for _ in range(3):
        ctx.process_format_device(fmt, data.data_ptr(), n, 0, fetch=False)
    pend, tt = [], []
    for r in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            pend.append(ctx.submit_format_device(fmt, data.data_ptr(), n, 0))
            if len(pend) == 3:
                nb = ctx.wait(pend.pop(0), fetch=False)
        while pend:
            nb = ctx.wait(pend.pop(0), fetch=False)
        torch.cuda.synchronize()
        if r:
            tt.append((time.perf_counter() - t0) / steps)
    w = float(np.median(tt))
    print("%-8s %-28s %9.4f ms/pass   %8.1f Gsamples/s   bursts %d   (min %.4f max %.4f)" % (
        tag, name, w * 1e3, n / w / 1e9, nb, min(tt) * 1e3, max(tt) * 1e3), flush=True)
    ctx.close()


iq = stream(2e6, 1000, 30, 1)
run("2 Msps fc32 2^30", _native.FMT_FC32, iq, 1 << 30, 2e6)
mag = (iq * iq).sum(dim=1).contiguous()
run("2 Msps |IQ|^2 2^30", _native.FMT_MAG2, mag, 1 << 30, 2e6)
del mag
q8 = torch.clamp(torch.round(iq * (128.0 / 4.0)), -127, 127).to(torch.int8).contiguous()
for lg in (30, 28, 26, 24, 22):
    run("2 Msps int8 2^%d" % lg, _native.FMT_SC8, q8, 1 << lg, 2e6, scale=4.0 / 128.0, steps=20 if lg >= 26 else 100)
u8 = (q8.to(torch.int16) + 128).to(torch.uint8).contiguous()
run("2 Msps uint8 2^30", _native.FMT_CU8, u8, 1 << 30, 2e6, scale=4.0 / 255.0)
del u8
del q8, iq
torch.cuda.empty_cache()
iq = stream(8e6, 6000, 28, 2)
run("8 Msps dense fc32 2^28", _native.FMT_FC32, iq, 1 << 28, 8e6)
del iq
iq = stream(20e6, 1000, 28, 3)
for lg in (28, 27, 26, 25, 24, 22):
    run("20 Msps fc32 2^%d" % lg, _native.FMT_FC32, iq, 1 << lg, 20e6, steps=20 if lg >= 26 else 100)
