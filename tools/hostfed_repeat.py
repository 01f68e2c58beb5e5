"""The page-locked complex64 host-fed leg, ten times on one box, with what round 4 never logged: the NUMA placement, the
plain pinned H2D rate measured right beside every repeat, and the wall time of every chunk's submit and wait.
    python tools/hostfed_repeat.py [repeats] [log2 chunk samples]         (GPU box only)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gr_adsb_amd import _native  # noqa: E402
from gr_adsb_amd import modulator as M  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
log2c = int(sys.argv[2]) if len(sys.argv) > 2 else 26
chunk, depth, nbuf, per = 1 << log2c, _native.MAX_IN_FLIGHT, 4, 12
dev = torch.device("cuda:0")
iq = M.synth_iq_torch(chunk, 2e6, 1000, 1, dev)
host = iq.cpu().numpy().view(np.complex64).reshape(-1)
ctx = _native.Context(2e6, 0.01)
info = ctx.numa_info()
print(json.dumps({"numa": info, "cpus_allowed": len(os.sched_getaffinity(0)), "chunk_samples": chunk, "depth": depth}))
for kind in ("torch_pin_memory", "adsb_host_alloc_near"):
    if kind == "torch_pin_memory":
        keep = [torch.empty((chunk, 2), dtype=torch.float32).pin_memory() for _ in range(nbuf)]
        views = [k.numpy().view(np.complex64).reshape(-1) for k in keep]
    else:
        keep = [_native.PinnedArray(chunk, np.complex64, near=ctx) for _ in range(nbuf)]
        views = [k.array for k in keep]
    for v in views:
        v[:] = host
    dst = torch.empty((chunk, 2), dtype=torch.float32, device=dev)
    st = torch.cuda.Stream()
    srcs_t = [torch.from_numpy(v.view(np.float32).reshape(-1, 2)) for v in views]
    for r in range(reps):
        with torch.cuda.stream(st):
            for k in range(3):
                dst.copy_(srcs_t[k % nbuf], non_blocking=True)
            st.synchronize()
            t0 = time.perf_counter()
            for k in range(per):
                dst.copy_(srcs_t[k % nbuf], non_blocking=True)
            st.synchronize()
            h2d = per * chunk * 8 / (time.perf_counter() - t0) / 1e9
        for k in range(2):
            ctx.wait(ctx.submit_format_host(_native.FMT_FC32, views[k]), fetch=False)
        pend, t_sub, t_wait = [], [], []
        t0 = time.perf_counter()
        for k in range(per):
            a = time.perf_counter()
            pend.append(ctx.submit_format_host(_native.FMT_FC32, views[k % nbuf]))
            t_sub.append((time.perf_counter() - a) * 1e3)
            if len(pend) == depth:
                a = time.perf_counter()
                ctx.wait(pend.pop(0), fetch=False)
                t_wait.append((time.perf_counter() - a) * 1e3)
        while pend:
            a = time.perf_counter()
            ctx.wait(pend.pop(0), fetch=False)
            t_wait.append((time.perf_counter() - a) * 1e3)
        dt = time.perf_counter() - t0
        gbs = per * chunk * 8 / dt / 1e9
        print(json.dumps({"source": kind, "repeat": r, "hostfed_gbytes_per_s": round(gbs, 2), "plain_h2d_gbytes_per_s": round(h2d, 2),
                          "ratio": round(gbs / h2d, 3), "submit_ms": [round(v, 2) for v in t_sub], "wait_ms": [round(v, 2) for v in t_wait]}))
    del keep, views, srcs_t
