"""Shared test helpers: golden-fixture loading and record comparison."""
import glob
import os

import numpy as np

from gr_adsb_amd import modulator as M

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCHEDULES = ("single", "fixed4096", "fixed8192", "random")


def _names(pattern):
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, pattern)))


def golden_names():
    """Round-1 vectors: 2^17 samples of int16 IQ per rate, four chunk schedules (tools/make_golden.py)."""
    return _names("g*msps*.npz")


def large_golden_names():
    """Round-3 vectors: 2^20 .. 3*2^20 samples of int8 IQ per rate, >= 500 tags each (tools/make_golden_large.py)."""
    return _names("L*.npz")


def rate_golden_names():
    """Round-4 vectors (tools/make_golden_lcg.py): the rates beyond the four instantiated ones -- 6 / 10 / 12 / 16 / 24 / 40 /
    100 Msps, served by the run-time-stride kernels.  Input = tests/lcg_stream.py (code), outputs = the reference's."""
    return _names("R*.npz")


def bulk_golden_names():
    """Round-4 vector: 2^28 samples at 2 Msps (generated, never stored) + the reference's tags and PDUs for them."""
    return _names("B*.npz")


def pathological_names():
    """Round-3 vectors: float32 |IQ|^2 with NaN / inf, thresholds <= 0, multi-tile plateaus, ties, tiny inputs."""
    return _names("P*.npz")


def schedules_of(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return [k[:-len("_schedule")] for k in z.files if k.endswith("_schedule")]


class Golden:
    def __init__(self, name, lazy=False):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.fs = float(z["fs"])
        self.sps = int(self.fs // 1e6)
        self.thr = float(z["threshold"])
        self.iq = self.iq8 = None
        self.gen = None
        if "gen_n" in z.files:
            # input is code (tests/lcg_stream.py): the int8 IQ bytes the reference saw are regenerated, not stored
            import lcg_stream
            self.gen = {k: int(z["gen_" + k]) for k in lcg_stream.PARAM_KEYS}
            self.scale = np.float32(z["scale"])
            if not lazy:
                self.load_generated()
        elif "iq16" in z.files:
            self.iq = M.dequantize_iq16(z["iq16"])
            self.x = M.mag2(self.iq)
        elif "iq8" in z.files:
            # the cs8 wire format: component = f32(int8) * scale (one rounded multiply), then re*re + im*im
            self.iq8 = z["iq8"]
            self.scale = np.float32(z["scale"])
            v = self.iq8.astype(np.float32) * self.scale
            self.iq = (v[0::2] + 1j * v[1::2]).astype(np.complex64)
            self.x = M.mag2(self.iq)
        else:
            self.x = z["x"]
        self.z = z

    def load_generated(self, lo=0, hi=None):
        """(Re)generate samples [lo, hi) of a code-backed vector: iq8 bytes, complex64 IQ and |IQ|^2 exactly as the
        reference saw them (component = f32(int8) * scale, one rounded multiply)."""
        import lcg_stream
        self.iq8 = lcg_stream.stream(self.gen, lo=lo, hi=hi)
        v = self.iq8.astype(np.float32) * self.scale
        self.iq = (v[0::2] + 1j * v[1::2]).astype(np.complex64)
        self.x = M.mag2(self.iq)

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


# ---- the gnuradio.adsb drop-in (packaging/gnuradio_adsb) evaluated the way GRC does, without GNU Radio -----------
def load_gnuradio_adsb():
    """Import packaging/gnuradio_adsb/adsb/__init__.py as `gnuradio.adsb` under a stub `gnuradio` package (the test
    double of the GNU Radio runtime, gr_adsb_amd/grshim.py) unless a real GNU Radio is importable."""
    import importlib.util
    import sys
    import types
    from gr_adsb_amd import grshim
    if "gnuradio" not in sys.modules:
        try:
            import gnuradio  # noqa: F401
        except ImportError:
            pkg = types.ModuleType("gnuradio")
            pkg.__path__ = []
            pkg.gr = grshim
            sys.modules["gnuradio"] = pkg
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "packaging", "gnuradio_adsb", "adsb", "__init__.py")
    spec = importlib.util.spec_from_file_location("gnuradio.adsb", path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["gnuradio.adsb"] = mod
    spec.loader.exec_module(mod)
    sys.modules["gnuradio"].adsb = mod
    return mod


def grc_instantiate(descriptor, **overrides):
    """What GRC's generated flowgraph script does with a block descriptor (dict from the .block.yml): run
    `templates.imports`, evaluate `templates.make` with the parameter values substituted for ${id}.  Returns
    (block, callbacks) -- callbacks(name=value, ...) evaluates the descriptor's callback templates on the block."""
    import re
    # GRC keeps parameter values as Python expressions (the text `2e6` is a string to a YAML 1.1 parser) and evaluates them
    params = {p["id"]: (eval(p["default"]) if isinstance(p.get("default"), str) else p.get("default")) for p in descriptor["parameters"]}
    params.update(overrides)
    ns = {}
    exec(descriptor["templates"]["imports"], ns)
    sub = lambda t, vals: re.sub(r"\$\{(\w+)\}", lambda m: repr(vals[m.group(1)]), t)     # noqa: E731
    blk = eval(sub(descriptor["templates"]["make"], params), ns)

    def callbacks(**new):
        vals = dict(params, **new)
        for cb in descriptor["templates"].get("callbacks", []):
            eval("blk." + sub(cb, vals), dict(ns, blk=blk))
    return blk, callbacks


def every_byte_pair_stream(scale, slot=256, quiet=0):
    """int8 IQ, 2 Msps: slot k holds 100 samples of byte pair W_k (a constant noise window), one zero, then a preamble
    whose four high chips are byte pair V_k = k (i = low byte, q = high byte) -- every pair once as a record's peak and
    once as its median.  Returns (iq8, threshold): the threshold lies below the smallest non-zero |IQ|^2."""
    k = np.arange(65536, dtype=np.int64)
    pair = np.full((65536 * slot,), quiet, dtype=np.uint16)      # (quiet = 0x8080: the resting level of offset-binary bytes)
    w = ((k * 40503 + 12345) & 0xFFFF).astype(np.uint16)
    base = k * slot
    for j in range(20, 120):
        pair[base + j] = w
    for c in (0, 2, 7, 9):
        pair[base + 121 + c] = k.astype(np.uint16)
    s2 = np.float32(scale) * np.float32(scale)
    return pair.view(np.int8), np.float32(0.75) * s2


def preamble_train_iq(n, spacing=32, sps=2, seed=3):
    """complex64 IQ: a bare preamble every `spacing` symbols (pulses at chips 0, 2, 7, 9, nothing else -- each one a matched
    centre, most of them inside the previous one's 63-symbol gate) over a small deterministic noise floor: far more list
    entries per chunk than any real signal, for the paths that depend on the LENGTH of a unit's list."""
    rng = np.random.default_rng(seed)
    a = (rng.random(n, dtype=np.float32) * np.float32(0.02)).astype(np.float32)          # |IQ|^2 <= 4e-4, threshold 0.01
    half = sps // 2
    for c in (0, 2, 7, 9):
        for j in range(half):
            a[40 * sps + c * half + j::spacing * sps] = np.float32(1.0) + np.float32(0.25) * rng.random(len(a[40 * sps + c * half + j::spacing * sps]), dtype=np.float32)
    iq = np.zeros(n, dtype=np.complex64)
    iq.real = a
    iq.imag = (rng.random(n, dtype=np.float32) * np.float32(0.01)).astype(np.float32)
    return iq


def rise_storm_iq8(n, seed=0, offset_binary=False, hi=100, lo=2, half=1):
    """Interleaved 8-bit IQ (2n values) whose |IQ|^2 crosses a threshold of 0.01 (scale 1/128) up to 512 times per 1024-sample
    tile: blocks of 16 * half samples that are either alternating high / low sample by sample (8 * half rises), a bare
    preamble pattern at `half` samples per chip (chips 0, 2, 7, 9 high: 4 rises, the first one a matched centre at
    2 * half Msps) or quiet -- more rises per tile than the 8-bit formats' rise list holds (256), with matched preambles in
    every part of the tile, so the batches of k_detect's B.1 loop are exercised and their hits must come out in stream order.
    half = 1 (2 Msps) is the generator of round 5 bit for bit."""
    rng = np.random.default_rng(seed)
    B = 16 * half
    nb = n // B
    kind = rng.choice(3, size=nb, p=[0.6, 0.3, 0.1])
    i = np.full(n, lo, dtype=np.int16)
    alt = np.zeros(B, dtype=np.int16)
    alt[0::2] = hi - lo
    pre = np.zeros(B, dtype=np.int16)
    for c in (0, 2, 7, 9):
        pre[c * half:(c + 1) * half] = hi - lo
    blocks = i[:nb * B].reshape(nb, B)
    blocks[kind == 0] += alt
    blocks[kind == 1] += pre
    blocks += rng.integers(0, 3, size=blocks.shape, dtype=np.int16) * (blocks > lo)        # unequal peaks: medians / SNR vary
    q = np.zeros(2 * n, dtype=np.int16)
    q[0::2] = i
    q[1::2] = rng.integers(-1, 2, size=n)
    if offset_binary:
        return ((q + 255) // 2).clip(0, 255).astype(np.uint8)       # (2u - 255) ~ q: odd integers, same pattern
    return q.astype(np.int8)
