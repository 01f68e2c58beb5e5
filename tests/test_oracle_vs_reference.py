"""Pins the oracle against the UNMODIFIED reference (imported with a stubbed GNU Radio runtime).
Runs only where /root/reference exists (the build container); the GPU box relies on tests/golden."""
import os
import sys

import numpy as np
import pytest

REF_OK = os.path.exists("/root/reference/python/adsb/framer.py")
pytestmark = pytest.mark.skipif(not REF_OK, reason="/root/reference not present on this machine")

from gr_adsb_amd import modulator as M  # noqa: E402
from oracle import adsb_oracle as O  # noqa: E402


def _ref():
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import ref_harness
    return ref_harness


def _same(r, o):
    assert np.array_equal(r["tag_offsets"], o["tag_offsets"])
    assert np.array_equal(r["tag_snr"].view(np.uint32), o["tag_snr"].view(np.uint32))
    assert np.array_equal(r["pdu_offsets"], o["pdu_offsets"])
    assert np.array_equal(r["pdu_bits"], o["pdu_bits"])
    assert np.array_equal(r["pdu_conf"].view(np.uint32), o["pdu_conf"].view(np.uint32))
    assert r["final_prev_eob"] == o["final_prev_eob"]
    assert np.float32(r["final_prev_in0"]).view(np.uint32) == np.float32(o["final_prev_in0"]).view(np.uint32)


@pytest.mark.parametrize("fs,bps", [(2e6, 2000), (4e6, 3000), (8e6, 6000), (20e6, 3000),
                                    (6e6, 4000), (10e6, 5000), (12e6, 6000), (16e6, 8000), (50e6, 20000), (100e6, 50000)])
def test_single_and_random_schedules(fs, bps):
    R = _ref()
    rng = np.random.default_rng(int(fs))
    n = 1 << 17
    x = M.mag2(M.synth_iq(n, fs, bps, seed=int(fs / 1e6) + 40))
    _same(R.run_reference(x, fs, 0.01), O.run_stream(x, fs, 0.01))
    for lo, hi in [(1000, 9000), (1, 700), (4096, 4097)]:
        sched, rem = [], n
        while rem > 0:
            c = int(min(rem, rng.integers(lo, hi)))
            sched.append(c)
            rem -= c
        _same(R.run_reference(x, fs, 0.01, sched), O.run_stream(x, fs, 0.01, sched))


def test_deaf_state_reproduced():
    """with constant N the reference can go permanently deaf; the oracle must too."""
    R = _ref()
    fs, n = 2e6, 1 << 18
    x = M.mag2(M.synth_iq(n, fs, 6000, seed=77))
    for N in (2048, 4096, 8192):
        sched = [N] * (n // N)
        _same(R.run_reference(x, fs, 0.01, sched), O.run_stream(x, fs, 0.01, sched))


def test_pathological_inputs():
    R = _ref()
    fs = 2e6
    base = M.mag2(M.synth_iq(1 << 15, fs, 5000, seed=9))
    cases = []
    x = base.copy(); x[3000:9000] = 0.5; cases.append((x, 0.01))
    x = base.copy(); x[:50] = 0.7; x[-40:] = 0.7; cases.append((x, 0.01))
    x = base.copy(); x[[100, 5000, 5001, 20000]] = np.nan; x[[3000, 30000]] = np.inf; cases.append((x, 0.01)); cases.append((x, 0.0))
    cases.append((base, 0.0)); cases.append((base, -1.0)); cases.append((base, 0.001))
    cases.append((np.zeros(3000, np.float32), 0.01)); cases.append((np.full(3000, 0.3, np.float32), 0.01))
    for n in (1, 15, 16, 17, 239, 240, 241):
        cases.append((base[7000:7000 + n].copy(), 0.01))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for x, thr in cases:
            _same(R.run_reference(x, fs, thr), O.run_stream(x, fs, thr))


def test_parity_oracle_matches_live_reference_decoder():
    """oracle.mode_s_parity (the restatement the device's pre-filter flags are checked against) vs the imported
    reference decoder's decode_header + check_parity (decoder.py:550-688), incl. a populated aircraft table."""
    R = _ref()
    dec = R.load_reference_decoder("All Messages", "None", "None")
    rng = np.random.default_rng(5)
    rows = []
    for df in (0, 4, 5, 11, 16, 17, 18, 19, 20, 21, 24, 3, 31):
        for _ in range(12):
            f = M.make_frame(df, rng)
            b = rng.integers(0, 2, 112).astype(np.uint8)
            b[:len(f)] = f
            rows.append(b)
            g = b.copy(); g[rng.integers(0, 112)] ^= 1
            rows.append(g)
    bits = np.array(rows)
    p = O.mode_s_parity(bits)
    known_aa = set()
    for i, b in enumerate(bits):
        dec.reset(); dec.bits = b.astype(int); dec.datetime = ""; dec.snr = 0.0; dec.timestamp = 0.0
        dec.decode_header()
        passed = dec.check_parity()
        assert dec.df == p["df"][i]
        assert dec.payload_length == (p["nbits"][i] if p["nbits"][i] else -1)
        if dec.df in (11, 17, 18, 19):
            assert passed == int(p["parity_ok"][i])
        elif p["nbits"][i]:
            assert dec.aa == p["syndrome"][i]
            # address/parity formats pass iff the announced address is in the table (decoder.py:583,653)
            assert passed == int(dec.aa_str in dec.plane_dict)
            if i % 3 == 0:
                dec.plane_dict[dec.aa_str] = {"callsign": ""}
                known_aa.add(dec.aa)
    assert len(known_aa) > 20


def test_c_oracle_equals_live_reference_on_adversarial_streams():
    """The scalar C port (oracle/adsb_oracle.c) is what the GPU box checks the HIP path against on pathological streams;
    here it meets the UNMODIFIED reference directly on 300 streams from the seam-targeted adversarial generator
    (plateaus over tile seams, exact ties, NaN, thresholds <= 0 and sitting on sample values; the four instantiated rates
    and 6 / 10 / 12 / 16 / 30 / 100 Msps, the run-time-stride rates)."""
    import warnings
    from oracle import c_oracle as C
    from helpers import snr_bits, unpack
    from test_sim_property import adversarial_stream
    R = _ref()
    n_tags = 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for seed in range(300):
            rng = np.random.default_rng(7000 + seed)
            n = int(rng.choice([1, 17, 240, 1023, 1024, 1025, 1279, 1280, 1281, 2049, 3333, 4096, 4352, 8193, 12288, 30000]))
            sps = int(rng.choice([2, 4, 8, 20, 6, 10, 12, 16, 30, 100]))
            thr = float(rng.choice([0.01, 0.0099, 0.0101, 0.004, 0.05, 0.0, -1.0]))
            x = adversarial_stream(rng, n, sps)
            if seed % 5 == 0:
                x[rng.integers(0, n, 2)] = np.inf
            r = R.run_reference(x, sps * 1e6, thr)
            c = C.canonical(x, sps, np.float32(thr))
            what = "seed %d n %d sps %d thr %g" % (seed, n, sps, thr)
            assert np.array_equal(c["offset"], r["tag_offsets"]), what
            assert np.array_equal(snr_bits(c["peak"], c["median"]), r["tag_snr"].view(np.uint32)), what
            dem = (c["flags"] & 1) != 0
            assert np.array_equal(c["offset"][dem], r["pdu_offsets"]), what
            assert np.array_equal(unpack(c["bits"][dem]), r["pdu_bits"]), what
            n_tags += len(c)
    assert n_tags > 150
