"""Kernel tuning aid: build side copies of libadsb_hip.so with extra -D flags (phase ablations, tile
variants) and time k_detect for each on the GPU box.

    python tools/kbench.py build  name1="-DADSB_ABLATE=1" name2="..."     (CPU box: cross-compiles)
    python tools/kbench.py run [--fs 2e6 --bursts 1000 --log2n 28]         (GPU box)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, "gr_adsb_amd", "_variants")
sys.path.insert(0, ROOT)

CHILD = r'''
import sys, json, numpy as np, torch
sys.path.insert(0, %(root)r)
from gr_adsb_amd import modulator as M
from gr_adsb_amd.frontend import FrontEnd
fs=%(fs)r; n=1<<%(log2n)d
iq = M.synth_iq_torch(n, fs, %(bursts)r, 1, torch.device("cuda:0"))
fe = FrontEnd(fs, 0.01, timing=True)
for _ in range(2): fe.process_iq_tensor(iq, 0, fetch=False)
fe.ctx.reset_stats()
import time
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(%(steps)d): nb = fe.process_iq_tensor(iq, 0, fetch=False)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/%(steps)d
st = fe.stats()
k = st["detect_ms"]/st["detect_launches"]
print(json.dumps(dict(kernel_ms=round(k,4), step_ms=round(dt*1e3,4), gbs=round(8*n/k/1e6,1), bursts=int(nb), grid=st["detect_grid"], bpc=st["blocks_per_cu"])))
'''


def build(specs):
    from gr_adsb_amd import build as b
    os.makedirs(VDIR, exist_ok=True)
    for spec in specs:
        name, flags = spec.split("=", 1)
        out = os.path.join(VDIR, "libadsb_%s.so" % name)
        cmd = [b.hipcc()] + b.FLAGS + flags.split() + [os.path.join(b.CSRC, "adsb_hip.hip"), "-o", out]
        print(" ".join(cmd))
        subprocess.check_call(cmd)
        with open(out + ".flags", "w") as f:
            f.write(flags)


def run(argv):
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--fs", type=float, default=2e6)
    ap.add_argument("--bursts", type=float, default=1000.0)
    ap.add_argument("--log2n", type=int, default=28)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--only", default="")
    a = ap.parse_args(argv)
    libs = [("shipped", os.path.join(ROOT, "gr_adsb_amd", "libadsb_hip.so"))]
    if os.path.isdir(VDIR):
        for f in sorted(os.listdir(VDIR)):
            if f.endswith(".so"):
                libs.append((f[len("libadsb_"):-3], os.path.join(VDIR, f)))
    for name, path in libs:
        if a.only and name not in a.only.split(","):
            continue
        env = dict(os.environ, ADSB_HIP_LIB=path)
        code = CHILD % dict(root=ROOT, fs=a.fs, log2n=a.log2n, bursts=a.bursts, steps=a.steps)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("ERR " + r.stderr[-300:])
        print("%-14s fs=%g bursts=%g  %s" % (name, a.fs, a.bursts, line), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(sys.argv[2:])
