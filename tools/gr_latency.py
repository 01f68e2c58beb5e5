"""Per-work()-call cost of the drop-in blocks (GNU Radio emulation path) and the sample rate they sustain per chunk size,
from scheduler-sized chunks (2 k samples) to 16 M samples: independent blocks (each uploads and scans its input: the
reference's structure) and the paired form demod(fs, framer=...) (one upload and one device pass per chunk).  GPU box only.
    python tools/gr_latency.py [fs]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gc
import numpy as np
import torch  # noqa: F401
from gr_adsb_amd import blocks, modulator as M

# Python's cyclic collector scans every live container object on a full collection -- with torch imported that is ~10^6
# objects, tens of milliseconds, triggered by the tags / PDUs of a multi-megasample call (measured: 3.7 us per PDU inside
# message_port_pub, 0.3 without).  gc.freeze() after the imports (what a long-running flowgraph script does too) moves them
# out of the collector's sight; "--no-freeze" measures without it.
if "--no-freeze" not in sys.argv:
    gc.freeze()

fs = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 2e6
sps = int(fs // 1e6)
L = 1 << 25
x = M.mag2(M.synth_iq(1 << 22, fs, 1000, 3))
x = np.tile(x, L // len(x))
H = 8 * sps
buf = np.concatenate([np.zeros(H - 1, np.float32), x])
print("# fs %g Msps, 1000 bursts/s, host buffers (pageable numpy arrays, as GNU Radio hands them over); gc.freeze() after imports: %s"
      % (fs / 1e6, "--no-freeze" not in sys.argv))
PIN = "--no-pin" not in sys.argv        # page-lock, once, the array the work() inputs are slices of (blocks.pin_source)
print("# inputs page-locked once through their owning array (blocks.pin_source): %s" % PIN)
for paired in (False, True):
    for N in (2048, 8192, 32768, 262144, 1 << 20, 1 << 22, 1 << 24):
        fr = blocks.framer(fs, 0.01, pin_inputs=PIN)
        dm = blocks.demod(fs, framer=fr if paired else None)
        out = np.empty(N, np.float32)
        tf = td = 0.0
        calls = samples = pdus = 0
        budget = max(2, min(1024, (1 << 26) // N))
        pos = 0
        for k in range(-3, budget):                       # the first calls warm a fresh context up (allocations)
            if pos + N > len(x):
                pos = 0
            fr._nread = fr._nwritten = pos
            t1 = time.perf_counter()
            fr.work([buf[pos:pos + N + H - 1]], [out])
            t2 = time.perf_counter()
            dm.tags_in = [t for t in fr.tags_out if t.offset >= pos]       # the tags of this chunk
            dm._nread = dm._nwritten = pos
            dm.work([x[pos:pos + N]], [out])
            t3 = time.perf_counter()
            pos += N
            if k >= 0:
                tf += t2 - t1
                td += t3 - t2
                calls += 1
                samples += N
                pdus += len(dm.messages)
            fr.tags_out.clear()
            dm.messages.clear()
        print("%s chunk %8d samples: %8.3f ms per framer+demod call pair (framer %.3f, demod %.3f; %d pairs, %d PDUs) -> %8.1f Msamples/s"
              % ("paired     " if paired else "independent", N, (tf + td) / calls * 1e3, tf / calls * 1e3, td / calls * 1e3, calls, pdus,
                 samples / (tf + td) / 1e6), flush=True)

if "--profile" in sys.argv:
    # where a large paired call's time goes (cProfile, 4 M-sample chunks)
    import cProfile
    import pstats
    N = 1 << 22
    fr = blocks.framer(fs, 0.01)
    dm = blocks.demod(fs, framer=fr)
    out = np.empty(N, np.float32)
    pr = cProfile.Profile()
    pos = 0
    for k in range(8):
        fr._nread = fr._nwritten = pos
        if k >= 2:
            pr.enable()
        fr.work([buf[pos:pos + N + H - 1]], [out])
        pr.disable()
        dm.tags_in = [t for t in fr.tags_out if t.offset >= pos]
        dm._nread = dm._nwritten = pos
        if k >= 2:
            pr.enable()
        dm.work([x[pos:pos + N]], [out])
        pr.disable()
        pos += N
        fr.tags_out.clear()
        dm.messages.clear()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)

if "--small" in sys.argv:
    # where a scheduler-sized call's time goes: the C entry point alone (ctypes call included), the block's work() around
    # it, the paired demod with and without a tag in the chunk
    from gr_adsb_amd import _native
    N = 2048
    ctx = _native.Context(fs, 0.01, flags=_native.FLAG_FRAMER_SLICES)
    quiet = np.full(N + H - 1, 1e-4, np.float32)
    busy = buf[:N + H - 1].copy()
    for name, arr in (("quiet chunk", quiet), ("chunk of the test stream", busy)):
        for _ in range(200):
            ctx.framer_work(arr, N, 0)
        t0 = time.perf_counter()
        for _ in range(2000):
            ctx.framer_work(arr, N, 0)
        print("adsb_framer_work through ctypes, %-26s %.2f us per call" % (name + ":", (time.perf_counter() - t0) / 2000 * 1e6))
    fr = blocks.framer(fs, 0.01)
    dm = blocks.demod(fs, framer=fr)
    out = np.empty(N, np.float32)
    for name, arr in (("quiet chunk", quiet), ("chunk of the test stream", busy)):
        tf = td = 0.0
        for k in range(2200):
            fr._nread = fr._nwritten = 0
            t1 = time.perf_counter()
            fr.work([arr], [out])
            t2 = time.perf_counter()
            dm.tags_in = list(fr.tags_out)
            dm._nread = dm._nwritten = 0
            t3 = time.perf_counter()
            dm.work([arr[H - 1:]], [out])
            t4 = time.perf_counter()
            if k >= 200:
                tf += t2 - t1
                td += t4 - t3
            fr.tags_out.clear()
            dm.messages.clear()
        print("blocks, %-26s framer.work %.2f us, paired demod.work %.2f us" % (name + ":", tf / 2000 * 1e6, td / 2000 * 1e6))

if "--breakdown" in sys.argv:
    # where a megasample pair's time goes: the C entry point without / with the fused pass-through, the block around it
    # (tags), the paired demod (PDUs, pass-through)
    from gr_adsb_amd import _native
    for N in (1 << 18, 1 << 20, 1 << 22):
        ctx = _native.Context(fs, 0.01, flags=_native.FLAG_FRAMER_SLICES)
        out = np.empty(N, np.float32)
        arr = buf[:N + H - 1]
        reps = max(4, (1 << 25) // N)
        res = {}
        for name, kw in (("C call, no pass-through", {}), ("C call + fused pass-through", {"out0": out})):
            for _ in range(3):
                ctx.framer_work(arr, N, 0, **kw)
            t0 = time.perf_counter()
            for _ in range(reps):
                b = ctx.framer_work(arr, N, 0, **kw)
            res[name] = (time.perf_counter() - t0) / reps * 1e3
        t0 = time.perf_counter()
        for _ in range(reps):
            out[:] = arr[H - 1:]
        res["numpy out0[:] = in0"] = (time.perf_counter() - t0) / reps * 1e3
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.host_copy(out, x[:N])
        res["adsb_host_copy (copy threads)"] = (time.perf_counter() - t0) / reps * 1e3
        pin = _native.PinnedArray(N + H - 1, np.float32, near=ctx)
        pin.array[:] = arr
        for _ in range(3):
            ctx.framer_work(pin.array, N, 0, out0=out)
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.framer_work(pin.array, N, 0, out0=out)
        res["C call + fused pass-through, page-locked input"] = (time.perf_counter() - t0) / reps * 1e3
        fr = blocks.framer(fs, 0.01)
        dm = blocks.demod(fs, framer=fr)
        tf = td = 0.0
        for k in range(-2, reps):
            fr._nread = fr._nwritten = 0
            t1 = time.perf_counter()
            fr.work([arr], [out])
            t2 = time.perf_counter()
            dm.tags_in = list(fr.tags_out)
            dm._nread = dm._nwritten = 0
            t3 = time.perf_counter()
            dm.work([arr[H - 1:]], [out])
            t4 = time.perf_counter()
            if k >= 0:
                tf += t2 - t1
                td += t4 - t3
            ntag, npdu = len(fr.tags_out), len(dm.messages)
            fr.tags_out.clear()
            dm.messages.clear()
        res["framer.work (block: + %d tags)" % ntag] = tf / reps * 1e3
        res["paired demod.work (block: %d PDUs + pass-through)" % npdu] = td / reps * 1e3
        print("chunk %d samples: " % N + "; ".join("%s %.3f ms" % kv for kv in res.items()), flush=True)
