"""Per-work()-call cost of the drop-in blocks (GNU Radio emulation path): one H2D copy + one device pass per
call.  Prints the sustainable sample rate for typical scheduler chunk sizes.  GPU box only."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from gr_adsb_amd import blocks, modulator as M

fs = 2e6
x = M.mag2(M.synth_iq(1 << 21, fs, 1000, 3))
H = 16
buf = np.concatenate([np.zeros(H - 1, np.float32), x])
for N in (2048, 8192, 32768, 262144):
    fr = blocks.framer(fs, 0.01)
    dm = blocks.demod(fs)
    out = np.empty(N, np.float32)
    tf = td = 0.0
    calls = samples = 0
    for rep in range(-1, max(1, 64 * N // len(x))):           # rep -1: untimed warm-up (allocations of a fresh context)
        pos = 0
        while pos + N <= len(x):
            fr._nread = fr._nwritten = pos
            t1 = time.perf_counter()
            fr.work([buf[pos:pos + N + H - 1]], [out])
            t2 = time.perf_counter()
            dm.tags_in = [t for t in fr.tags_out[-256:] if t.offset >= pos]       # the tags of this chunk
            dm._nread = dm._nwritten = pos
            dm.work([x[pos:pos + N]], [out])
            t3 = time.perf_counter()
            pos += N
            if rep >= 0:
                tf += t2 - t1
                td += t3 - t2
                calls += 1
                samples += N
            elif pos >= 4 * N:
                break
        fr.tags_out.clear()
        dm.messages.clear()
    print("chunk %7d samples: %.3f ms per framer+demod call pair (framer %.3f, demod %.3f; %d pairs) -> %.1f Msamples/s sustained"
          % (N, (tf + td) / calls * 1e3, tf / calls * 1e3, td / calls * 1e3, calls, samples / (tf + td) / 1e6))
