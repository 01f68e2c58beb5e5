"""Tuning probe: isolated k_detect rate as a function of the stream length, i.e. of the per-wavefront chunk size
(tiles per unit) -- looks for address-pattern effects (HBM channel spread) in the one-contiguous-chunk-per-wavefront layout.
    python tools/chunk_sweep.py            (GPU box)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gr_adsb_amd import modulator as M          # noqa: E402
from gr_adsb_amd.frontend import FrontEnd       # noqa: E402

fs = 2e6
nmax = 1 << 30
dev = torch.device("cuda:0")
parts = [M.synth_iq_torch(1 << 22, fs, 1000, 1000003 + b, dev) for b in range(nmax >> 22)]
iq = torch.cat(parts)
del parts
torch.cuda.synchronize()
fe = FrontEnd(fs, 0.01, timing=True)
units = 5116
for tiles_per in [205, 204, 203, 202, 201, 200, 199, 198, 197, 196, 192, 190, 180, 160, 128, 127, 129, 100, 64, 65, 63]:
    n = units * tiles_per * 1024
    if n > nmax:
        continue
    for _ in range(2):
        fe.ctx.process_format_device(0, iq.data_ptr(), n, 0, fetch=False)
    fe.ctx.reset_stats()
    for _ in range(6):
        fe.ctx.process_format_device(0, iq.data_ptr(), n, 0, fetch=False)
    st = fe.stats()
    ms = st["detect_ms"] / st["detect_launches"]
    print("tiles/unit %4d  n %11d  grid %d  kernel_ms %.4f  %.1f GB/s  frac %.4f" % (
        tiles_per, n, st["detect_grid"], ms, 8.0 * n / ms / 1e6, 8.0 * n / ms / 1e6 / 8000), flush=True)
