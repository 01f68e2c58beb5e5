"""CPU fuzz of the EMULATED device kernels (tests/simlib: the adsb_device.h that ships, compiled for the host) on the 8-bit
input formats -- aimed at what round 6 changed: the exact-hint median (noise quantised to a few levels: consecutive bursts
with the very same / a neighbouring median, odd and even windows, windows cut short by the start of the stream), the mask
bytes built from the registers, the dot-product instances for power-of-two scales (int8 and offset-binary uint8).
Checker: the C oracle (pinned to the live reference by tests/test_oracle_vs_reference.py) on the oracle's |IQ|^2.
  python tools/fuzz_sim_8bit.py [seconds] [seed]          CPU, emulated kernels
  python tools/fuzz_sim_8bit.py [seconds] [seed] gpu      the same cases through the C ABI on the GPU (adsb_process_format), with
                                                          longer streams mixed in (several chunks per wavefront, many workgroups)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import simlib  # noqa: E402
from helpers import assert_recs_equal  # noqa: E402
from gr_adsb_amd import modulator as M  # noqa: E402
from oracle import adsb_oracle as O  # noqa: E402
from oracle import c_oracle as C  # noqa: E402


def stream(rng, n, sps, unsigned):
    """interleaved 8-bit IQ: few-level noise, bursts of a few amplitudes, some back to back, some at the very start"""
    nl = int(rng.choice([1, 2, 3, 6, 12]))                       # noise amplitude in LSB
    i = rng.integers(-nl, nl + 1, n)
    q = rng.integers(-nl, nl + 1, n)
    if rng.random() < 0.3:                                       # a floor that changes level halfway
        k = int(rng.integers(0, n))
        i[k:] = rng.integers(-2 * nl, 2 * nl + 1, n - k)
    pos = int(rng.choice([0, 1, 3, 50, 99, 100, 101, 150])) if rng.random() < 0.5 else int(rng.integers(0, max(1, n)))
    nb = 0
    while pos < n and nb < 60:
        env = M.burst_waveform(M.make_frame(int(rng.choice([17, 11, 4, 0, 20])), rng), sps)
        a = int(rng.choice([20, 40, 60, 90, 120]))
        ph = rng.random() * 2 * np.pi
        e = min(n, pos + len(env))
        on = env[:e - pos] > 0
        i[pos:e][on] += int(round(a * np.cos(ph)))
        q[pos:e][on] += int(round(a * np.sin(ph)))
        nb += 1
        pos = e + int(rng.choice([0, 1, sps, 8 * sps, 100, 101, 300, 1024, 3000]))
    iq = np.empty(2 * n, dtype=np.int64)
    iq[0::2], iq[1::2] = i, q
    if unsigned:
        return np.clip(iq + 128, 0, 255).astype(np.uint8)
    return np.clip(iq, -128, 127).astype(np.int8)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    on_gpu = len(sys.argv) > 3 and sys.argv[3] == "gpu"
    res = run(budget, seed, on_gpu)
    print("%s, seed %d: %d cases, %d bursts, all identical to the C oracle; per instance: %s" % (
        "GPU (C ABI)" if on_gpu else "emulator", seed, res["cases"], res["bursts"], sorted(res["per_instance"].items())))


def run(budget, seed, on_gpu=False):
    """Cases until `budget` seconds are used; raises AssertionError on the first difference (a bounded run with a fixed seed is
    part of both test suites: tests/test_sim_property.py on the emulator, tests/test_gpu_fuzz.py through the C ABI)."""
    ctxs = {}
    if on_gpu:
        from gr_adsb_amd import _native
        _native.load()
    rng = np.random.default_rng(seed)
    t0, cases, bursts = time.time(), 0, 0
    per = {}
    while time.time() - t0 < budget:
        sps = int(rng.choice([2, 2, 4, 6, 8, 10, 20]))
        n = int(rng.choice([300, 1023, 1024, 1025, 2048, 4096, 5000, 12288, 30000, 60000] + ([250000, 1 << 20, 3000000] if on_gpu else [])))
        n -= n % 8                                               # whole 16-byte groups of 8-bit IQ
        unsigned = bool(rng.random() < 0.5)
        scale = float(rng.choice([2.0 ** -5, 2.0 ** -6, 2.0 ** -7, 2.0 ** -10, 1 / 100.0, 4 / 255.0, 1 / 127.0, 0.013]))
        iq = stream(rng, n, sps, unsigned)
        x = O.mag2_iq8(iq, scale, offset_binary=unsigned)
        hi = float(np.sort(x)[int(0.9 * (len(x) - 1))])
        thr = float(rng.choice([0.01, 0.02, 0.005, max(hi, 1e-6), float(x[int(rng.integers(0, len(x)))])]))
        want = C.canonical(x, sps, thr)
        if on_gpu:
            fmt = _native.FMT_CU8 if unsigned else _native.FMT_SC8
            ctx = ctxs.get(sps)
            if ctx is None:
                ctx = ctxs[sps] = _native.Context(sps * 1e6, thr)
            ctx.reset()
            ctx.set_threshold(thr)
            ctx.set_format_scale(fmt, scale)
            got = ctx.process_format(fmt, iq)
        else:
            got, so = simlib.sim_canonical(4 if unsigned else 3, iq, sps * 1e6, thr, scale=scale, grid_max=int(rng.integers(1, 7)))
            assert so.overflow == 0
        assert_recs_equal(got, want, "seed %d case %d: n %d sps %d %s scale %r thr %r" % (seed, cases, n, sps, "u8" if unsigned else "i8", scale, thr))
        cases += 1
        bursts += len(want)
        k = ("u8" if unsigned else "i8", "pow2" if np.frexp(scale)[0] == 0.5 else "generic")
        per[k] = per.get(k, 0) + 1
    return {"cases": cases, "bursts": bursts, "per_instance": per}


if __name__ == "__main__":
    main()
