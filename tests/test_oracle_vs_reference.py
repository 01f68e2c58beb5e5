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


@pytest.mark.parametrize("fs,bps", [(2e6, 2000), (4e6, 3000), (8e6, 6000), (20e6, 3000)])
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
