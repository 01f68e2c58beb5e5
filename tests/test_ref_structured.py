"""Pins oracle/ref_structured.py -- the reference-STRUCTURED CPU baseline bench.py times (vectorised front end +
per-pulse Python loop + np.median) -- against the goldens from the real reference, against the
vectorised oracle, against its own process-per-shard form, and (container only) against the real reference."""
import os
import sys

import numpy as np
import pytest

from helpers import Golden, golden_names
from gr_adsb_amd import modulator as M
from oracle import adsb_oracle as O
from oracle import ref_structured as RS


@pytest.mark.parametrize("name", golden_names())
def test_matches_reference_goldens(name):
    g = Golden(name)
    r = RS.run_stream(g.x, g.fs, g.thr)
    assert np.array_equal(r["tag_offsets"], g.get("single", "tag_offsets"))
    assert np.array_equal(r["tag_snr"].view(np.uint32), g.get("single", "tag_snr_bits"))
    assert np.array_equal(r["tag_offsets"][r["pdu_tag_index"]], g.get("single", "pdu_offsets"))
    assert np.array_equal(r["pdu_bits"], g.pdu_bits("single"))
    assert np.array_equal(r["pdu_conf"].view(np.uint32), g.get("single", "pdu_conf_bits"))


@pytest.mark.parametrize("fs,bps,shards", [(2e6, 1000, 3), (2e6, 3000, 7), (8e6, 6000, 4), (20e6, 1000, 5)])
def test_process_per_shard_equals_single_call(fs, bps, shards):
    n = 1 << 19
    x = M.mag2(M.synth_iq(n, fs, bps, seed=int(bps + fs / 1e6)))
    want = O.run_stream(x, fs, 0.01)
    stats = {}
    one = RS.run_stream(x, fs, 0.01, stats=stats)
    assert np.array_equal(one["tag_offsets"], want["tag_offsets"]) and np.array_equal(one["pdu_bits"], want["pdu_bits"])
    assert np.array_equal(one["tag_snr"].view(np.uint32), want["tag_snr"].view(np.uint32))
    assert stats["evaluated"] <= stats["pulses"]          # the gate skips pulses inside an accepted burst (framer.py:121)
    sh = RS.run_sharded(x, fs, 0.01, shards)
    assert not sh["fallback"]
    assert np.array_equal(sh["tag_offsets"], want["tag_offsets"])
    assert np.array_equal(sh["tag_snr"].view(np.uint32), want["tag_snr"].view(np.uint32))
    assert np.array_equal(sh["pdu_tag_index"], want["pdu_tag_index"]) and np.array_equal(sh["pdu_bits"], want["pdu_bits"])


def test_shard_that_cannot_synchronise_falls_back():
    # a carrier: never 63*sps quiet samples in any warm-up region -> serial fallback, still exact
    fs, n = 2e6, 1 << 17
    x = M.mag2(M.synth_iq(n, fs, 2000, seed=3)) + np.float32(0.02)
    sh = RS.run_sharded(x, fs, 0.01, 3)
    want = O.run_stream(x, fs, 0.01)
    assert sh["fallback"] and np.array_equal(sh["tag_offsets"], want["tag_offsets"])


@pytest.mark.skipif(not os.path.exists("/root/reference/python/adsb/framer.py"), reason="/root/reference not present")
def test_against_the_real_reference():
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import ref_harness as R
    for fs, bps, seed in [(2e6, 2000, 5), (8e6, 6000, 6)]:
        x = M.mag2(M.synth_iq(1 << 17, fs, bps, seed=seed))
        r = R.run_reference(x, fs, 0.01)
        o = RS.run_stream(x, fs, 0.01)
        assert np.array_equal(r["tag_offsets"], o["tag_offsets"])
        assert np.array_equal(r["tag_snr"].view(np.uint32), o["tag_snr"].view(np.uint32))
        assert np.array_equal(r["pdu_bits"], o["pdu_bits"])
        assert np.array_equal(r["pdu_conf"].view(np.uint32), o["pdu_conf"].view(np.uint32))
