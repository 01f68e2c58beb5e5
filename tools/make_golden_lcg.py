"""Round-4 fixtures from the REAL reference whose INPUT is code, not data (container-only; needs /root/reference).

  tests/golden/R*.npz   the sample rates the reference advertises beyond the four instantiated ones ("2 Msps, 4 Msps, 6 Msps,
                        etc", README.md:17; tap stride sps//2, framer.py:45,137): 6 / 10 / 12 / 16 / 24 / 40 Msps and the
                        library's maximum, 100 Msps -- served by the run-time-stride kernels k_detect<fmt, 0> /
                        k_pass_small<0>.  2^20 - 2^22 samples each, single call + fixed 2048 (deaf-state) + two random
                        chunk schedules.
  tests/golden/B*.npz   BULK vectors: 2^28 samples at 2 / 8 / 20 Msps (~32 k / 44 k / 8 k bursts): the size at which the
                        library cuts a pass into eight rounds of short chunks (adsb_plan.h plan_chunks) and the usual-tile
                        instance carries everything -- the reference's own tags and PDUs for them, single call.

The input of every vector is tests/lcg_stream.py (integer hashing of the sample index; NumPy here, torch on the GPU box:
identical bytes); the .npz holds the generator's parameters and what the unmodified framer.py / demod.py produced for
|IQ|^2 of those bytes (component = f32(int8) * scale, one rounded multiply; re*re + im*im).  No reference source is stored.

  python tools/make_golden_lcg.py [rates] [bulk [B8msps_bulk ...]]
"""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import lcg_stream as L  # noqa: E402
import ref_harness as R  # noqa: E402
from make_golden_large import random_schedule, store  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SCALE = np.float32(1.0 / 128.0)

RATES = [
    # name, sps, n, gap (samples per burst slot), seed, threshold
    ("R6msps", 6, 1 << 20, 1000, 401, 0.005),        # slots shorter than a long reply (720 samples): overlapping bursts
    ("R10msps", 10, 1 << 20, 2500, 402, 0.01),
    ("R12msps", 12, 1 << 21, 3000, 403, 0.005),
    ("R16msps", 16, 1 << 21, 4000, 404, 0.01),
    ("R24msps", 24, 1 << 21, 5000, 405, 0.005),
    ("R40msps", 40, 1 << 21, 9000, 406, 0.01),
    ("R100msps", 100, 1 << 22, 16000, 407, 0.005),   # the library's maximum (adsb_create): preamble span 800 samples
]
BULKS = [
    ("B2msps_bulk", 2, 1 << 28, 8192, 501, 0.005),
    # round 4, second batch: the instances with the preamble taps 4 and 10 samples apart at bulk size -- bursts longer than
    # the LDS window (960 / 2400 samples), whose bits are taken tile by tile from the wavefront's pending list
    ("B8msps_bulk", 8, 1 << 28, 6144, 502, 0.005),
    ("B20msps_bulk", 20, 1 << 28, 32768, 503, 0.01),
]


def mag2_of(iq8):
    v = iq8.astype(np.float32) * SCALE
    return (v[0::2] * v[0::2] + v[1::2] * v[1::2]).astype(np.float32)


def gen_keys(p, thr):
    d = {"gen_" + k: np.int64(p[k]) for k in L.PARAM_KEYS}
    d.update(fs=np.float64(p["sps"] * 1e6), threshold=np.float64(thr), scale=SCALE)
    return d


def rates():
    for name, sps, n, gap, seed, thr in RATES:
        p = L.params(n, sps, seed, gap)
        x = mag2_of(L.stream(p))
        fs = sps * 1e6
        rng = np.random.default_rng(seed)
        data = gen_keys(p, thr)
        scheds = {"single": [n], "fixed2048": [2048] * (n // 2048), "random": random_schedule(n, rng),
                  "randombig": random_schedule(n, rng, 150 * sps, 1500 * sps)}
        for sname, sched in scheds.items():
            r = R.run_reference(x, fs, thr, None if sname == "single" else sched)
            assert r["snr_types"] <= {"float32"}
            store(data, sname, sched, r, conf=(sname == "single"))
            print(name, sname, "tags", len(r["tag_offsets"]), "pdus", len(r["pdu_offsets"]), flush=True)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)


def bulk(only=None):
    for spec in BULKS:
        if only and spec[0] not in only:
            continue
        bulk_one(*spec)


def bulk_one(name, sps, n, gap, seed, thr):
    p = L.params(n, sps, seed, gap)
    t0 = time.time()
    x = mag2_of(L.stream(p))
    print("generated", n, "samples in %.0f s" % (time.time() - t0), flush=True)
    t0 = time.time()
    r = R.run_reference(x, sps * 1e6, thr, None)
    print("reference: %.0f s, tags %d pdus %d" % (time.time() - t0, len(r["tag_offsets"]), len(r["pdu_offsets"])), flush=True)
    data = gen_keys(p, thr)
    store(data, "single", [n], r, conf=False)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)


if __name__ == "__main__":
    warnings.simplefilter("ignore")
    what = sys.argv[1:] or ["rates", "bulk"]
    if "rates" in what:
        rates()
    if "bulk" in what:
        bulk([w for w in what if w.startswith("B")])
