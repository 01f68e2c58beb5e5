"""The oracle (NumPy and C) against the golden vectors generated from the REAL reference
(tools/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from helpers import (GOLDEN_DIR, SCHEDULES, Golden, bulk_golden_names, golden_names, large_golden_names, pathological_names,
                     rate_golden_names, schedules_of, snr_bits, unpack)
from oracle import adsb_oracle as O
from oracle import c_oracle as C


@pytest.mark.parametrize("name", golden_names())
@pytest.mark.parametrize("sched", SCHEDULES)
def test_numpy_oracle_matches_reference_goldens(name, sched):
    g = Golden(name)
    o = O.run_stream(g.x, g.fs, g.thr, None if sched == "single" else g.sched(sched))
    assert np.array_equal(o["tag_offsets"], g.get(sched, "tag_offsets"))
    assert np.array_equal(o["tag_snr"].view(np.uint32), g.get(sched, "tag_snr_bits"))
    assert np.array_equal(o["pdu_offsets"], g.get(sched, "pdu_offsets"))
    assert np.array_equal(o["pdu_bits"], g.pdu_bits(sched))
    assert np.array_equal(o["pdu_snr"].view(np.uint32), g.get(sched, "pdu_snr_bits"))
    assert o["final_prev_eob"] == int(g.get(sched, "final_prev_eob"))
    if sched == "single":
        assert np.array_equal(o["pdu_conf"].view(np.uint32), g.get(sched, "pdu_conf_bits"))


@pytest.mark.parametrize("name", golden_names())
def test_c_oracle_matches_reference_goldens(name):
    g = Golden(name)
    r = C.process_iq(g.iq, g.sps, g.thr)
    assert np.array_equal(r["offset"], g.get("single", "tag_offsets"))
    assert np.array_equal(snr_bits(r["peak"], r["median"]), g.get("single", "tag_snr_bits"))
    dem = (r["flags"] & 1) != 0
    assert np.array_equal(r["offset"][dem], g.get("single", "pdu_offsets"))
    assert np.array_equal(unpack(r["bits"][dem]), g.pdu_bits("single"))


def _numpy_oracle_vs(g, sched):
    with np.errstate(all="ignore"):
        o = O.run_stream(g.x, g.fs, g.thr, None if sched == "single" else g.sched(sched))
    assert np.array_equal(o["tag_offsets"], g.get(sched, "tag_offsets"))
    assert np.array_equal(o["tag_snr"].view(np.uint32), g.get(sched, "tag_snr_bits"))
    assert np.array_equal(o["pdu_offsets"], g.get(sched, "pdu_offsets"))
    assert np.array_equal(o["pdu_bits"], g.pdu_bits(sched))
    assert np.array_equal(o["pdu_snr"].view(np.uint32), g.get(sched, "pdu_snr_bits"))
    assert o["final_prev_eob"] == int(g.get(sched, "final_prev_eob"))
    assert np.float32(o["final_prev_in0"]).view(np.uint32) == g.get(sched, "final_prev_in0_bits")
    if sched == "single":
        assert np.array_equal(o["pdu_conf"].view(np.uint32), g.get(sched, "pdu_conf_bits"))


def _c_oracle_vs(g):
    r = C.canonical(g.x, g.sps, np.float32(g.thr))
    assert np.array_equal(r["offset"], g.get("single", "tag_offsets"))
    assert np.array_equal(snr_bits(r["peak"], r["median"]), g.get("single", "tag_snr_bits"))
    dem = (r["flags"] & 1) != 0
    assert np.array_equal(r["offset"][dem], g.get("single", "pdu_offsets"))
    assert np.array_equal(unpack(r["bits"][dem]), g.pdu_bits("single"))


@pytest.mark.parametrize("name", large_golden_names())
def test_oracles_match_large_reference_goldens(name):
    """tests/golden/L*.npz: >= 500 reference tags per rate (20 Msps included), incl. the fixed-2048 deaf-state schedule."""
    g = Golden(name)
    assert len(g.get("single", "tag_offsets")) >= 490
    for sched in schedules_of(name):
        _numpy_oracle_vs(g, sched)
    _c_oracle_vs(g)
    if g.iq8 is not None:                              # the C port's own int8 conversion on the same bytes
        assert np.array_equal(O.mag2_iq8(g.iq8, float(g.scale), False), g.x)


@pytest.mark.parametrize("name", rate_golden_names())
def test_oracles_match_rate_reference_goldens(name):
    """tests/golden/R*.npz (round 4): 6 / 10 / 12 / 16 / 24 / 40 / 100 Msps -- the rates the reference advertises beyond the
    four instantiated ones (README.md:17, framer.py:45,137) -- single call, deaf-state and two random schedules.  The
    input is regenerated from tests/lcg_stream.py."""
    g = Golden(name)
    assert g.sps not in (2, 4, 8, 20) and len(g.get("single", "tag_offsets")) >= 200
    for sched in schedules_of(name):
        _numpy_oracle_vs(g, sched)
    _c_oracle_vs(g)
    assert np.array_equal(O.mag2_iq8(g.iq8, float(g.scale), False), g.x)


@pytest.mark.parametrize("name", bulk_golden_names())
def test_c_oracle_matches_bulk_reference_golden_on_a_prefix(name):
    """tests/golden/B*.npz: the reference's tags for 2^28 generated samples.  Here (CPU) the first 2^23 samples are
    regenerated and the C oracle must reproduce every reference tag whose burst lies inside them (the gate is causal; only
    the PDU drop rule at the end of a call depends on the call's length, demod.py:82); the GPU test runs all 2^28."""
    g = Golden(name, lazy=True)
    n = 1 << 23
    g.load_generated(0, n)
    r = C.canonical(g.x, g.sps, np.float32(g.thr))
    offs = g.get("single", "tag_offsets")
    k = int(np.searchsorted(offs, n - 8 * g.sps + 1))            # tags the prefix call can see: centre < n - (H - 1)
    assert k > 200 and np.array_equal(r["offset"], offs[:k])
    assert np.array_equal(snr_bits(r["peak"], r["median"]), g.get("single", "tag_snr_bits")[:k])
    dem = (r["flags"] & 1) != 0
    kp = int(dem.sum())
    assert np.array_equal(r["offset"][dem], g.get("single", "pdu_offsets")[:kp])
    assert np.array_equal(unpack(r["bits"][dem]), g.pdu_bits("single")[:kp])


def test_generated_streams_are_identical_under_numpy_and_torch():
    """tests/lcg_stream.py: integer hashing only -- the NumPy path (what the reference was run on) and the torch path
    (what the GPU box generates on the device) must give the same bytes, and any window stands on its own."""
    import lcg_stream as L
    for sps, gap in ((2, 4000), (6, 1000), (100, 16000)):
        p = L.params(1 << 17, sps, 17 + sps, gap)
        a = L.stream(p, block=1 << 15)
        assert np.array_equal(a, L.stream(p, device="cpu", block=50000).numpy())
        assert np.array_equal(L.stream(p, lo=12345, hi=99999), a[2 * 12345:2 * 99999])


@pytest.mark.parametrize("name", pathological_names())
def test_oracles_match_pathological_reference_goldens(name):
    """tests/golden/P*.npz: NaN / inf, thresholds <= 0, plateaus over several tiles, start / end high, ties, tiny inputs
    -- the reference's own outputs, so the GPU box does not rest on the C oracle alone for these."""
    g = Golden(name)
    for sched in schedules_of(name):
        _numpy_oracle_vs(g, sched)
    _c_oracle_vs(g)


def test_c_oracle_equals_numpy_oracle_on_fresh_synthetic():
    from gr_adsb_amd import modulator as M
    for fs, bps, seed in [(2e6, 2000, 1), (6e6, 3000, 2), (20e6, 4000, 3)]:
        iq = M.synth_iq(1 << 17, fs, bps, seed)
        sps = int(fs // 1e6)
        x = O.mag2(iq)
        o = O.run_stream(x, fs, 0.01)
        r, cands = C.canonical(x, sps, 0.01, want_cands=True)
        assert np.array_equal(r["offset"], o["tag_offsets"])
        assert np.array_equal(r["median"].view(np.uint32), o["tag_median"].view(np.uint32))
        assert np.array_equal(cands, o["cand_offsets"])
        assert np.array_equal(O.pack_bits(o["pdu_bits"]), r["bits"][(r["flags"] & 1) != 0])


def test_mag2_is_two_rounded_products():
    rng = np.random.default_rng(0)
    z = (rng.standard_normal(4096) + 1j * rng.standard_normal(4096)).astype(np.complex64)
    re = z.real.astype(np.float64)
    im = z.imag.astype(np.float64)
    want = (np.float32(re * re).astype(np.float64) + np.float32(im * im).astype(np.float64)).astype(np.float32)
    assert np.array_equal(O.mag2(z), want)


def test_parity_oracle_matches_reference_decoder_known_answers():
    """tests/golden/g_parity.npz (tools/make_golden_parity.py): DF, payload length, check_parity() verdict
    and announced address of the reference decoder (decoder.py:550-688) for 1028 PDUs, vs both oracles."""
    from oracle import c_oracle as C
    z = np.load(os.path.join(GOLDEN_DIR, "g_parity.npz"))
    bits = np.unpackbits(z["bits"], axis=1)[:, :112]
    p = O.mode_s_parity(bits)
    assert np.array_equal(p["df"], z["df"])
    assert np.array_equal(np.where(p["nbits"] == 0, -1, p["nbits"]), z["payload_length"])
    assert np.array_equal(p["parity_ok"].astype(np.int32), z["parity_passed"])     # empty aircraft table
    ap = np.isin(z["df"], (0, 4, 5, 16, 20, 21, 24))
    assert ap.sum() > 100 and p["parity_ok"].sum() > 100
    assert np.array_equal(p["syndrome"][ap], z["aa"][ap])
    recs = np.zeros(len(bits), dtype=C.REC)
    recs["bits"] = z["bits"]; recs["flags"] = 1
    cf, csyn = C.parity_flags(recs)
    assert np.array_equal(cf & 0x1FE0, p["flags"]) and np.array_equal(csyn.astype(np.int64), p["syndrome"])
