"""Property tests (hypothesis) of the emulated device code against the oracle on adversarial streams:
random plateaus, pulses of every length around the tile/window boundaries, preambles planted at tile edges,
ties (equal samples), thresholds sitting exactly on sample values, random GNU Radio chunk schedules."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import simlib
from helpers import assert_recs_equal, snr_bits
from gr_adsb_amd import modulator as M
from oracle import adsb_oracle as O
from oracle import c_oracle as C

# the kernel's real seams (adsb_device.h): one wavefront walks 1024-sample tiles through a sliding LDS window that
# holds 128 samples behind and 256 samples beyond the tile; a pulse longer than the window goes to k_longrun
TILE, FWD, BACK = simlib.kernel_geometry()
WIN = TILE + FWD
assert (TILE, FWD, BACK) == (1024, 256, 128), "seam-targeted cases below assume this geometry: revisit them"


def adversarial_stream(rng, n, sps):
    """float32 |IQ|^2 with quantised amplitudes (many exact ties) and structures aimed at the kernel's seams."""
    levels = np.array([0.0, 0.002, 0.004, 0.0099, 0.01, 0.0101, 0.02, 0.05, 0.3, 1.0], dtype=np.float32)
    x = levels[rng.integers(0, 4, n)].copy()                    # quiet floor with ties
    half = sps // 2
    env = M.burst_waveform(M.make_frame(17, rng), sps)
    seams = [k * TILE + d for k in range(1, n // TILE + 2)
             for d in (-300, -257, -256, -255, -240, -130, -129, -128, -127, -101, -100, -99, -17, -16, -2, -1, 0, 1, 15, 16)]
    seams += [k * TILE + FWD + d for k in range(0, n // TILE + 1) for d in (-1, 0, 1)]        # the window's far edge
    for _ in range(int(rng.integers(0, 12))):                   # bursts, preferably starting at a seam
        s = int(rng.choice(seams)) if rng.random() < 0.7 else int(rng.integers(0, n))
        s = max(0, min(n - 1, s))
        e = min(n, s + len(env))
        amp = levels[rng.integers(6, 10)]
        x[s:e] = np.maximum(x[s:e], amp * env[:e - s])
    for _ in range(int(rng.integers(0, 10))):                   # plateaus of awkward lengths
        s = int(rng.choice(seams)) if rng.random() < 0.7 else int(rng.integers(0, n))
        s = max(0, min(n - 1, s))
        ln = int(rng.choice([1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 1023, 1024, 1025, 1279, 1280, 1281, 2500]))
        x[s:min(n, s + ln)] = levels[rng.integers(4, 10)]
    for _ in range(int(rng.integers(0, 40))):                   # isolated short pulses
        s = int(rng.integers(0, n))
        x[s:min(n, s + int(rng.integers(1, 2 * sps + 2)))] = levels[rng.integers(3, 10)]
    if rng.random() < 0.2:
        x[rng.integers(0, n, 3)] = np.nan
    return x


@settings(max_examples=150, deadline=None, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2**31 - 1),
       n=st.sampled_from([1, 17, 240, 1023, 1024, 1025, 1279, 1280, 1281, 2047, 2048, 2049, 3333, 4096, 4352, 8193, 12288]),
       sps=st.sampled_from([2, 4, 8, 20, 6, 10, 12, 16, 30, 100]),    # the four instantiated rates + run-time-stride ones
       thr=st.sampled_from([0.01, 0.0099, 0.0101, 0.004, 0.05, 0.0, -1.0]))
def test_canonical_equals_c_oracle_on_adversarial_streams(seed, n, sps, thr):
    rng = np.random.default_rng(seed)
    x = adversarial_stream(rng, n, sps)
    want = C.canonical(x, sps, thr)
    got, so = simlib.sim_canonical(1, x, sps * 1e6, thr, grid_max=int(rng.integers(1, 5)))
    assert so.overflow == 0
    assert_recs_equal(got, want, "seed %d n %d sps %d thr %g" % (seed, n, sps, thr))


@settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2**31 - 1), sps=st.sampled_from([2, 8, 6, 12, 100]))
def test_framer_work_random_schedules_equal_numpy_oracle(seed, sps):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([6000, 9000, 13000]))
    x = adversarial_stream(rng, n, sps)
    fs = sps * 1e6
    H = 8 * sps
    sched, rem = [], n
    while rem > 0:
        c = int(min(rem, rng.choice([1, 7, 64, 300, 1000, 1023, 1024, 1025, 1280, 4097, 5000])))
        sched.append(c)
        rem -= c
    o = O.run_stream(x, fs, 0.01, sched)
    buf = np.concatenate([np.zeros(H - 1, np.float32), x])
    fr = simlib.SimFramer(fs, 0.01, grid_max=2)
    pos, outs = 0, []
    for N in sched:
        r, _ = fr.work(buf[pos:pos + N + H - 1], N, pos)
        outs.append(r)
        pos += N
    recs = np.concatenate(outs)
    assert np.array_equal(recs["offset"], o["tag_offsets"])
    assert np.array_equal(snr_bits(recs["peak"], recs["median"]), o["tag_snr"].view(np.uint32))
    assert fr.prev_eob.value == o["final_prev_eob"]
    assert np.float32(fr.prev_in0.value).view(np.uint32) == np.float32(o["final_prev_in0"]).view(np.uint32)


def test_8bit_streams_aimed_at_the_exact_hint_median_and_the_dot_product_instances():
    """A bounded slice of tools/fuzz_sim_8bit.py on the emulator (the long runs: profiles/r06_fuzz_sim_8bit.txt): few-level noise
    floors, bursts back to back and at the start of the stream, int8 / offset-binary uint8 with power-of-two and other scales."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_sim_8bit
    res = fuzz_sim_8bit.run(8.0, 2718)
    assert res["cases"] >= 50 and res["bursts"] > 0 and len(res["per_instance"]) == 4, res
