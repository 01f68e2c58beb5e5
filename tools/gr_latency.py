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
    pos, calls = 0, 0
    t0 = time.perf_counter()
    while pos + N <= len(x):
        fr._nread = fr._nwritten = pos
        fr.work([buf[pos:pos + N + H - 1]], [out])
        dm.tags_in = fr.tags_out[-64:]
        dm._nread = dm._nwritten = pos
        dm.work([x[pos:pos + N]], [out])
        pos += N
        calls += 1
    dt = time.perf_counter() - t0
    print("chunk %7d samples: %.3f ms per framer+demod call pair -> %.1f Msamples/s sustained (%d tags, %d PDUs)"
          % (N, dt / calls * 1e3, pos / dt / 1e6, len(fr.tags_out), len(dm.messages)))
