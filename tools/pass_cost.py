"""Fixed cost of a mid-size pass: wall time per submitted pass (three in flight) for device-resident complex64 buffers of
2^20 ... 2^26 samples, split into the host time inside adsb_submit_* and inside adsb_wait, next to k_detect's own duration
(HIP events, from a second context with ADSB_FLAG_TIMING).  Canonical passes and shard passes (head_cands = 64).
    python tools/pass_cost.py [--fs 20e6]            (GPU box only)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gr_adsb_amd import _native  # noqa: E402
from gr_adsb_amd import modulator as M  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--fs", type=float, default=20e6)
ap.add_argument("--reps", type=int, default=300)
a = ap.parse_args()
dev = torch.device("cuda:0")
sps = int(a.fs // 1e6)
iq = M.synth_iq_torch(1 << 26, a.fs, 1000, 3, dev)
torch.cuda.synchronize()
print("fs %g: wall per pass (3 in flight), host time in submit / in wait, k_detect alone; us" % a.fs)
for kind in ("canonical", "shard"):
    for log2n in (20, 22, 23, 24, 25, 26):
        n = 1 << log2n
        ctx = _native.Context(a.fs, 0.01)
        tctx = _native.Context(a.fs, 0.01, flags=_native.FLAG_TIMING)
        for _ in range(4):
            tctx.process_format_device(_native.FMT_FC32, iq.data_ptr(), n, 0, fetch=False)
        tctx.reset_stats()
        for _ in range(8):
            tctx.process_format_device(_native.FMT_FC32, iq.data_ptr(), n, 0, fetch=False)
        st = tctx.stats()
        kern = st["detect_ms"] / st["detect_launches"] * 1e3

        def submit():
            if kind == "canonical":
                return ctx.submit_format_device(_native.FMT_FC32, iq.data_ptr(), n, 0)
            return ctx.submit_shard_device(_native.FMT_FC32, iq.data_ptr(), n, 0, 0, n - 4000, 1 << 40, 64)

        pend = []
        for _ in range(6):
            pend.append(submit())
            if len(pend) == 3:
                ctx.wait(pend.pop(0), fetch=False)
        while pend:
            ctx.wait(pend.pop(0), fetch=False)
        ts = tw = 0.0
        nb = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            t1 = time.perf_counter()
            pend.append(submit())
            t2 = time.perf_counter()
            ts += t2 - t1
            if len(pend) == 3:
                nb = ctx.wait(pend.pop(0), fetch=False)
                tw += time.perf_counter() - t2
        while pend:
            t2 = time.perf_counter()
            nb = ctx.wait(pend.pop(0), fetch=False)
            tw += time.perf_counter() - t2
        wall = time.perf_counter() - t0
        print("%-9s 2^%d  wall %7.1f   submit %6.1f   wait %6.1f   k_detect %7.1f   bursts %6d   -> %7.1f Gsamples/s" % (
            kind, log2n, wall / a.reps * 1e6, ts / a.reps * 1e6, tw / a.reps * 1e6, kern, int(nb), n / (wall / a.reps) / 1e9), flush=True)
        ctx.close()
        tctx.close()

# the same passes through the C driver (adsb_process_sharded_device: plan, submit three deep, wait, head fix-up -- no interpreter
# between two passes): the 2^26-sample stream as 2 .. 64 shards
print("adsb_process_sharded_device over 2^26 samples: wall per shard, us")
ctx = _native.Context(a.fs, 0.01)
out = np.empty(1 << 16, dtype=_native.BURST_DTYPE)
for shards in (2, 4, 8, 16, 32, 64):
    for _ in range(3):
        r = ctx.process_sharded_device(_native.FMT_FC32, iq.data_ptr(), 1 << 26, shards, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 40
    for _ in range(reps):
        r = ctx.process_sharded_device(_native.FMT_FC32, iq.data_ptr(), 1 << 26, shards, out=out)
    wall = (time.perf_counter() - t0) / reps
    print("shards %3d (2^%.1f samples each)  wall per shard %7.1f   bursts %6d   -> %7.1f Gsamples/s" % (
        shards, np.log2((1 << 26) / shards), wall / shards * 1e6, len(r), (1 << 26) / wall / 1e9), flush=True)
ctx.close()

# BASELINE config 4 on one GPU as bench.py runs it: 2^28 samples, 8 shards -- raw context, timed context (two more event
# records per pass), and through FrontEnd (a torch event + two stream waits per call)
from gr_adsb_amd.frontend import FrontEnd  # noqa: E402
big = M.synth_iq_torch(1 << 28, a.fs, 1000, 3, dev)
torch.cuda.synchronize()
for name, mk in (("raw context", lambda: _native.Context(a.fs, 0.01)), ("timed context", lambda: _native.Context(a.fs, 0.01, flags=_native.FLAG_TIMING)),
                 ("FrontEnd, timed", lambda: FrontEnd(a.fs, 0.01, timing=True))):
    o = mk()
    call = (lambda: o.process_sharded_tensor(_native.FMT_FC32, big, 8, out=out)) if isinstance(o, FrontEnd) else \
           (lambda: o.process_sharded_device(_native.FMT_FC32, big.data_ptr(), 1 << 28, 8, out=out))
    for _ in range(3):
        r = call()
    torch.cuda.synchronize()
    tt = []
    for _ in range(10):
        t0 = time.perf_counter()
        for _ in range(4):
            r = call()
        torch.cuda.synchronize()
        tt.append((time.perf_counter() - t0) / 4)
    w = float(np.median(tt))
    print("2^28 samples as 8 shards, %-16s %7.1f us per call = %5.1f per shard   bursts %6d   -> %7.1f Gsamples/s" % (
        name, w * 1e6, w * 1e6 / 8, len(r), (1 << 28) / w / 1e9), flush=True)
    whole_t = []
    c2 = o.ctx if isinstance(o, FrontEnd) else o
    for _ in range(3):
        c2.process_format_device(_native.FMT_FC32, big.data_ptr(), 1 << 28, 0, fetch=False)
    for _ in range(6):
        t0 = time.perf_counter()
        c2.process_format_device(_native.FMT_FC32, big.data_ptr(), 1 << 28, 0, fetch=False)
        whole_t.append(time.perf_counter() - t0)
    print("   the same 2^28 samples as ONE blocking pass: %7.1f us" % (float(np.median(whole_t)) * 1e6), flush=True)
