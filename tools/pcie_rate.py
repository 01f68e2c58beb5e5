"""Host-buffer (PCIe-inclusive) rate of the canonical path: adsb_process_iq on a pageable NumPy buffer
(library copies into pinned staging, H2D, full pipeline, records back).  GPU box only."""
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from gr_adsb_amd import modulator as M
from gr_adsb_amd.frontend import FrontEnd

fs = 2e6
n = 1 << 26
iq = M.synth_iq_torch(n, fs, 1000, 1, torch.device("cuda:0")).cpu().numpy().view(np.complex64).reshape(-1)
fe = FrontEnd(fs, 0.01)
fe.process_iq(iq)
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    r = fe.process_iq(iq)
dt = (time.perf_counter() - t0) / reps
print("host-fed adsb_process_iq: %d samples in %.2f ms = %.1f Msamples/s (%.1f GB/s of complex64), %d bursts"
      % (n, dt * 1e3, n / dt / 1e6, 8 * n / dt / 1e9, len(r)))
# pinned source (adsb_host_alloc): DMA'd straight from the caller's buffer
from gr_adsb_amd import _native
pa = _native.PinnedArray(n, np.complex64)
pa.array[:] = iq
h = pa.array
tp = torch.from_numpy(iq.view(np.float32)).pin_memory()
fe.process_iq(h)
t0 = time.perf_counter()
for _ in range(reps):
    r = fe.process_iq(h)
dt = (time.perf_counter() - t0) / reps
print("host-fed, source in adsb_host_alloc memory: %.1f Msamples/s (%.1f GB/s)" % (n / dt / 1e6, 8 * n / dt / 1e9))
# pure H2D copy rate for reference
d = torch.empty(n * 2, dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    d.copy_(tp, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print("plain pinned H2D copy: %.1f GB/s = %.1f Msamples/s" % (8 * n / dt / 1e9, n / dt / 1e6))
