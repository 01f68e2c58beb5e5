"""Shared test helpers: golden-fixture loading and record comparison."""
import glob
import os

import numpy as np

from gr_adsb_amd import modulator as M

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCHEDULES = ("single", "fixed4096", "fixed8192", "random")


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "g*msps*.npz")))


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.fs = float(z["fs"])
        self.sps = int(self.fs // 1e6)
        self.thr = float(z["threshold"])
        self.iq = M.dequantize_iq16(z["iq16"])
        self.x = M.mag2(self.iq)
        self.z = z

    def sched(self, s):
        return [int(v) for v in self.z[s + "_schedule"]]

    def get(self, s, key):
        return self.z[s + "_" + key]

    def pdu_bits(self, s):
        return np.unpackbits(self.z[s + "_pdu_bits"], axis=1)[:, :112]


def snr_bits(peak, median):
    with np.errstate(all="ignore"):
        p = np.asarray(peak, dtype=np.float32)
        m = np.asarray(median, dtype=np.float32)
        return (np.float32(10.0) * np.log10(p / m) + np.float32(1.6)).astype(np.float32).view(np.uint32)


def unpack(bits14):
    return np.unpackbits(np.asarray(bits14, dtype=np.uint8).reshape(-1, 14), axis=1, bitorder="big")[:, :112]


def assert_recs_match_golden(recs, g, s="single"):
    """recs: structured array (offset, peak, median, bits[14], flags) of kept bursts, canonical mode."""
    assert np.array_equal(recs["offset"], g.get(s, "tag_offsets")), "tag offsets differ"
    assert np.array_equal(snr_bits(recs["peak"], recs["median"]), g.get(s, "tag_snr_bits")), "SNR bits differ"
    dem = (recs["flags"] & 1) != 0
    assert np.array_equal(recs["offset"][dem], g.get(s, "pdu_offsets")), "PDU set differs"
    assert np.array_equal(unpack(recs["bits"][dem]), g.pdu_bits(s)), "PDU bits differ"
    if np.any(recs["flags"] & PARITY_BITS):
        assert_parity_flags(recs, g.name)


PARITY_BITS = 0x1FE0      # ADSB_BURST_PARITY_OK | _LONG | _KNOWN_DF | DF << 8


def assert_parity_flags(recs, what=""):
    """The Mode S pre-filter bits of device records equal the oracle's restatement of decoder.py:550-688
    applied to the records' own bits (and are absent on records without a PDU)."""
    from oracle import adsb_oracle as O
    dem = (recs["flags"] & 1) != 0
    assert not np.any(recs["flags"][~dem] & PARITY_BITS), what + ": parity bits on a record without PDU"
    if dem.any():
        want = O.mode_s_parity(unpack(recs["bits"][dem]))["flags"]
        assert np.array_equal(recs["flags"][dem] & PARITY_BITS, want), what + ": parity pre-filter flags"


def assert_recs_equal(a, b, what=""):
    """Two record arrays (e.g. HIP vs C oracle) must agree bit for bit."""
    assert len(a) == len(b), "%s: %d vs %d records" % (what, len(a), len(b))
    assert np.array_equal(a["offset"], b["offset"]), what + ": offsets"
    assert np.array_equal(a["peak"].view(np.uint32), b["peak"].view(np.uint32)), what + ": peak"
    assert np.array_equal(a["median"].view(np.uint32), b["median"].view(np.uint32)), what + ": median"
    assert np.array_equal(a["flags"] & 1, b["flags"] & 1), what + ": demod flags"
    assert np.array_equal(a["bits"], b["bits"]), what + ": bits"
    for r in (a, b):                       # device-produced side(s): the oracle's framer records carry no parity bits
        if np.any(r["flags"] & PARITY_BITS):
            assert_parity_flags(r, what)
