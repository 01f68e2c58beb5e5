"""Integer-only synthetic ADS-B IQ streams: CODE instead of data.  Test infrastructure.

A golden vector whose input is megasamples (or 2^28 samples) long cannot be committed as data, and NumPy's Generator
streams are not promised to be stable across versions.  Everything here is 32-bit integer hashing
of the sample / burst INDEX, carried in int64 with every intermediate below 2^59: no floating point, no library PRNG,
no dependence on evaluation order -- NumPy (build container: the input the reference was run on) and torch (GPU box:
the same bytes generated on the device) produce the same int8 IQ bytes, and any window [lo, hi) of a stream can be
generated on its own.

Stream (cs8 wire format, interleaved int8 I, Q):
  noise      I, Q = (sum of the four bytes of hash(2 i + c) >> noise_shift) - (512 >> noise_shift)
  burst b    starts at b * gap + hash(b) % gap (so neighbours may overlap), amplitude amp_lo + hash % amp_span LSB, one of
             16 carrier phases, 112 random data bits (one in four bursts: 56), Mode S pulse-position chips of sps/2
             samples each behind the 8 us preamble 1010000101000000 (the reference's template, framer.py:50)
  sum clipped to [-127, 127]
"""
import numpy as np

M32 = 0xFFFFFFFF
_MUL = 0x45D9F3B
_COS64 = (64, 59, 45, 24, 0, -24, -45, -59, -64, -59, -45, -24, 0, 24, 45, 59)
_PREAMBLE = (1, 0, 1, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0)
N_CHIPS = 16 + 2 * 112


def hash32(i, seed):
    """i: int64 array / tensor with 0 <= i < 2^32, seed: Python int < 2^32.  Works on NumPy arrays and torch tensors."""
    h = (i ^ seed) & M32
    h = ((h ^ (h >> 16)) * _MUL) & M32
    h = ((h ^ (h >> 16)) * _MUL) & M32
    return h ^ (h >> 16)


def _sub_seed(seed, k):
    return (seed * 0x9E3779B1 + (k + 1) * 0x85EBCA6B) & M32


class _Np:
    @staticmethod
    def arange(lo, hi):
        return np.arange(lo, hi, dtype=np.int64)

    @staticmethod
    def full(n, v):
        return np.full(n, v, dtype=np.int64)

    @staticmethod
    def table(vals):
        return np.array(vals, dtype=np.int64)

    @staticmethod
    def scatter_add(dst, idx, val):
        np.add.at(dst, idx, val)

    @staticmethod
    def clip(a, lo, hi):
        return np.clip(a, lo, hi)

    @staticmethod
    def where(c, a, b):
        return np.where(c, a, b)

    @staticmethod
    def interleave_i8(i, q):
        out = np.empty(2 * len(i), dtype=np.int8)
        out[0::2] = i
        out[1::2] = q
        return out

    @staticmethod
    def empty_i8(n):
        return np.empty(n, dtype=np.int8)


class _Torch:
    def __init__(self, device):
        import torch
        self.t = torch
        self.dev = device

    def arange(self, lo, hi):
        return self.t.arange(lo, hi, dtype=self.t.int64, device=self.dev)

    def full(self, n, v):
        return self.t.full((n,), v, dtype=self.t.int64, device=self.dev)

    def table(self, vals):
        return self.t.tensor(vals, dtype=self.t.int64, device=self.dev)

    def scatter_add(self, dst, idx, val):
        dst.index_add_(0, idx, val)

    def clip(self, a, lo, hi):
        return self.t.clamp(a, lo, hi)

    def where(self, c, a, b):
        return self.t.where(c, a, b)

    def interleave_i8(self, i, q):
        return self.t.stack((i, q), dim=1).to(self.t.int8).reshape(-1)

    def empty_i8(self, n):
        return self.t.empty(n, dtype=self.t.int8, device=self.dev)


def burst_table(nb_lo, nb_hi, p, xp=_Np):
    """Per-burst parameters of bursts [nb_lo, nb_hi): start sample, I and Q amplitude (LSB), chips [nb, N_CHIPS] (0/1)."""
    b = xp.arange(nb_lo, nb_hi)
    seed = p["seed"]
    gap = p["gap"]
    start = b * gap + hash32(b, _sub_seed(seed, 2)) % gap
    h1 = hash32(b, _sub_seed(seed, 3))
    amp = p["amp_lo"] + h1 % p["amp_span"]
    ph = (h1 >> 12) & 15
    cos = xp.table(_COS64)
    ci, cq = cos[ph], cos[(ph + 12) & 15]                       # sin(phi) = cos(phi - 90 deg): index + 12 of 16
    zero = b * 0
    ai = xp.where(ci < 0, zero - ((amp * (zero - ci)) >> 6), (amp * ci) >> 6)
    aq = xp.where(cq < 0, zero - ((amp * (zero - cq)) >> 6), (amp * cq) >> 6)
    short = ((h1 >> 20) & 3) == 0
    k = xp.arange(0, 112)
    words = [hash32(4 * b + w, _sub_seed(seed, 4)) for w in range(4)]
    bits = []
    for w in range(4):
        kk = k[(k >> 5) == w] & 31
        bits.append((words[w][:, None] >> kk[None, :]) & 1)
    if xp is _Np:
        bits = np.concatenate(bits, axis=1)
        live = np.where(short[:, None], (k < 56)[None, :], True)
        pre = np.broadcast_to(np.array(_PREAMBLE, dtype=np.int64), (len(b), 16))
        chips = np.concatenate([pre, np.stack([bits * live, (1 - bits) * live], axis=2).reshape(len(b), 224)], axis=1)
    else:
        t = xp.t
        bits = t.cat(bits, dim=1)
        live = t.where(short[:, None], (k < 56)[None, :], t.ones(1, dtype=t.bool, device=xp.dev)).to(t.int64)
        pre = xp.table(_PREAMBLE)[None, :].expand(len(b), 16)
        chips = t.cat([pre, t.stack([bits * live, (1 - bits) * live], dim=2).reshape(len(b), 224)], dim=1)
    return start, ai, aq, chips


def window(p, lo, hi, xp=_Np):
    """int8 interleaved IQ of samples [lo, hi) of the stream with parameters p (dict: n, sps, seed, gap, amp_lo, amp_span,
    noise_shift)."""
    sps, half = p["sps"], p["sps"] // 2
    seed = p["seed"]
    i = xp.arange(lo, hi)
    acc = []
    for c in (0, 1):
        h = hash32(2 * i + c, _sub_seed(seed, c))
        s4 = (h & 255) + ((h >> 8) & 255) + ((h >> 16) & 255) + (h >> 24)
        acc.append((s4 >> p["noise_shift"]) - (512 >> p["noise_shift"]))
    nb = p["n"] // p["gap"]
    span = N_CHIPS * half
    b0 = max(0, (lo - span) // p["gap"] - 1)
    b1 = min(nb, hi // p["gap"] + 1)
    if b1 > b0:
        start, ai, aq, chips = burst_table(b0, b1, p, xp)
        kk = xp.arange(0, N_CHIPS) * half
        for j in range(half):
            idx = start[:, None] + kk[None, :] + j                       # [nb, N_CHIPS]
            on = (chips != 0) & (idx >= lo) & (idx < hi)
            sel = idx[on] - lo
            vi = (ai[:, None] + idx * 0)[on]
            vq = (aq[:, None] + idx * 0)[on]
            xp.scatter_add(acc[0], sel, vi)
            xp.scatter_add(acc[1], sel, vq)
    return xp.interleave_i8(xp.clip(acc[0], -127, 127), xp.clip(acc[1], -127, 127))


def stream(p, device=None, block=None, lo=0, hi=None):
    """The whole stream (or samples [lo, hi)) as int8 interleaved IQ: a NumPy array (device None) or a torch tensor on
    `device`, generated block by block."""
    hi = p["n"] if hi is None else hi
    xp = _Np if device is None else _Torch(device)
    block = block or ((1 << 22) if device is None else (1 << 24))
    out = xp.empty_i8(2 * (hi - lo))
    for a in range(lo, hi, block):
        e = min(hi, a + block)
        out[2 * (a - lo):2 * (e - lo)] = window(p, a, e, xp)
    return out


def params(n, sps, seed, gap, amp_lo=24, amp_span=90, noise_shift=6):
    return dict(n=int(n), sps=int(sps), seed=int(seed), gap=int(gap), amp_lo=int(amp_lo), amp_span=int(amp_span),
                noise_shift=int(noise_shift))


PARAM_KEYS = ("n", "sps", "seed", "gap", "amp_lo", "amp_span", "noise_shift")
