"""Synthetic Mode-S / ADS-B burst modulator (the reference ships none).

Builds complex64 IQ containing 1090ES pulse-position-modulated replies over complex AWGN:
  * preamble = 16 half-microsecond chips [1,0,1,0,0,0,0,1,0,1,0,0,0,0,0,0] -- the template the
    reference framer matches against (reference python/adsb/framer.py:50)
  * data bit b of 1 us = chips (1,0) for b=1, (0,1) for b=0 -- the convention the reference demod
    slices (reference python/adsb/demod.py:87-95: "bit 1 pulse" first, "bit 0 pulse" half a symbol later)
  * 24-bit parity with generator 0x1FFF409 (reference python/adsb/decoder.py:268-269)

This is test/bench input generation, not part of the replaced hot path.  `synth_iq` is NumPy (parity
fixtures, CPU tests); `synth_iq_torch` builds the same kind of stream directly in HBM for bench.py.
"""
import numpy as np

PREAMBLE_CHIPS = np.array([1, 0, 1, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0], dtype=np.uint8)
CRC_POLY = 0x1FFF409  # 25-bit generator incl. leading 1
SHORT_DFS = (0, 4, 5, 11)
LONG_DFS = (16, 17, 18, 20, 21)


def crc24(bits):
    """Remainder of bits(x)*x^24 mod generator; bits is a 0/1 array, MSB first."""
    reg = 0
    for b in bits:
        reg = (reg << 1) | int(b)
        if reg & (1 << 24):
            reg ^= CRC_POLY
    for _ in range(24):
        reg <<= 1
        if reg & (1 << 24):
            reg ^= CRC_POLY
    return reg & 0xFFFFFF


def _int_bits(v, n):
    return np.array([(v >> (n - 1 - i)) & 1 for i in range(n)], dtype=np.uint8)


def make_frame(df, rng, icao=None):
    """Random but parity-consistent Mode-S frame of the given downlink format: 56 or 112 bits."""
    nbits = 56 if df in SHORT_DFS else 112
    if icao is None:
        icao = int(rng.integers(1, 1 << 24))
    bits = np.zeros(nbits, dtype=np.uint8)
    bits[0:5] = _int_bits(df, 5)
    if df in (11, 17, 18):
        bits[5:8] = _int_bits(int(rng.integers(0, 8)), 3)  # CA / CF
        bits[8:32] = _int_bits(icao, 24)
        if nbits == 112:
            bits[32:88] = rng.integers(0, 2, 56, dtype=np.uint8)
        par = crc24(bits[: nbits - 24])
    else:
        bits[5: nbits - 24] = rng.integers(0, 2, nbits - 24 - 5, dtype=np.uint8)
        par = crc24(bits[: nbits - 24]) ^ icao  # address/parity overlay
    bits[nbits - 24:] = _int_bits(par, 24)
    return bits


def frame_chips(bits):
    """Half-microsecond chip sequence (0/1) of preamble + PPM data for a frame."""
    data = np.empty(2 * len(bits), dtype=np.uint8)
    data[0::2] = bits
    data[1::2] = 1 - bits
    return np.concatenate([PREAMBLE_CHIPS, data])


def burst_waveform(bits, sps):
    """Real 0/1 envelope sampled at sps samples per microsecond (sps even)."""
    assert sps % 2 == 0 and sps >= 2
    return np.repeat(frame_chips(bits), sps // 2).astype(np.float32)


def synth_iq(n, fs, bursts_per_s, seed, noise_power=1e-3, amp2_range=(0.05, 1.0),
             df_choices=(17,), df_weights=None, snr_db_range=None, return_truth=False):
    """complex64[n]: AWGN with E|z|^2 = noise_power plus randomly placed bursts.

    amp2_range: burst power (amplitude^2) drawn uniformly; if snr_db_range is given the power is
    noise_power * 10^(U(snr_db_range)/10) instead.  Collisions are allowed (bursts simply add).
    """
    rng = np.random.default_rng(seed)
    sps = int(fs // 1e6)
    z = (rng.standard_normal(n, dtype=np.float32) + 1j * rng.standard_normal(n, dtype=np.float32))
    z = (z * np.float32(np.sqrt(noise_power / 2.0))).astype(np.complex64)
    nb = int(round(bursts_per_s * n / fs))
    starts = np.sort(rng.integers(0, max(1, n - 120 * sps), nb))
    dfs = rng.choice(np.array(df_choices), size=nb, p=df_weights)
    truth = []
    for s, df in zip(starts, dfs):
        bits = make_frame(int(df), rng)
        env = burst_waveform(bits, sps)
        if snr_db_range is not None:
            p = noise_power * 10.0 ** (rng.uniform(*snr_db_range) / 10.0)
        else:
            p = rng.uniform(*amp2_range)
        ph = rng.uniform(0, 2 * np.pi)
        m = min(len(env), n - s)                    # (a stream shorter than one reply: the burst is cut by its end)
        z[s:s + m] += (np.float32(np.sqrt(p)) * env[:m] * np.complex64(np.exp(1j * ph))).astype(np.complex64)
        truth.append((int(s), int(df), bits))
    if return_truth:
        return z, truth
    return z


def mag2(iq):
    """float32 re*re + im*im with separately rounded products."""
    iq = np.asarray(iq, dtype=np.complex64)
    re = iq.real.astype(np.float32)
    im = iq.imag.astype(np.float32)
    return re * re + im * im


def quantize_iq16(iq, full_scale=2.0):
    """int16 interleaved IQ (fixture storage format); inverse is dequantize_iq16."""
    v = np.empty(2 * len(iq), dtype=np.float32)
    v[0::2] = iq.real
    v[1::2] = iq.imag
    return np.clip(np.rint(v * (32767.0 / full_scale)), -32768, 32767).astype(np.int16)


def quantize_iq8(iq, full_scale=2.0, offset_binary=False):
    """8-bit interleaved IQ: int8 (cs8) or, with offset_binary, uint8 around 127.5 (cu8, RTL-SDR)."""
    v = np.empty(2 * len(iq), dtype=np.float32)
    v[0::2] = iq.real
    v[1::2] = iq.imag
    if offset_binary:
        return np.clip(np.floor(v * (127.5 / full_scale) + 128.0), 0, 255).astype(np.uint8)
    return np.clip(np.rint(v * (127.0 / full_scale)), -128, 127).astype(np.int8)


def dequantize_iq16(q, full_scale=2.0):
    v = q.astype(np.float32) * np.float32(full_scale / 32767.0)
    return (v[0::2] + 1j * v[1::2]).astype(np.complex64)


def synth_iq_torch(n, fs, bursts_per_s, seed, device, noise_power=1e-3, amp2_range=(0.05, 1.0),
                   df_choices=None, df_weights=None, snr_db_range=None):
    """Same kind of stream as synth_iq, generated in HBM: by default DF17-length random-payload bursts.

    Returns a float32 tensor of shape [n, 2] (interleaved I,Q == complex64 memory layout).
    Payload bits are random (parity irrelevant to framer/demod); layout/amplitudes match synth_iq.
    df_choices / df_weights: per-burst downlink format (its five bits lead the payload; SHORT_DFS bursts end after
    56 bits -- BASELINE config 5's mix).  snr_db_range: burst power = noise_power * 10^(U(range)/10) instead of
    amp2_range.
    """
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    sps = int(fs // 1e6)
    half = sps // 2
    iq = torch.empty((n, 2), dtype=torch.float32, device=device)
    iq.normal_(0.0, float(np.sqrt(noise_power / 2.0)), generator=g)
    nb = int(round(bursts_per_s * n / fs))
    if nb == 0:
        return iq
    blen = 120 * sps
    starts = torch.randint(0, max(1, n - blen), (nb,), device=device, generator=g)
    bits = torch.randint(0, 2, (nb, 112), device=device, generator=g, dtype=torch.int64)
    pre = torch.tensor(PREAMBLE_CHIPS.astype(np.int64), device=device).expand(nb, 16)
    data = torch.stack([bits, 1 - bits], dim=2).reshape(nb, 224)
    if df_choices is not None:
        w = torch.tensor(np.asarray(df_weights if df_weights is not None else np.ones(len(df_choices)), dtype=np.float64),
                         device=device)
        pick = torch.multinomial(w / w.sum(), nb, replacement=True, generator=g)
        dfs = torch.tensor(np.asarray(df_choices, dtype=np.int64), device=device)[pick]
        for k in range(5):
            bits[:, k] = (dfs >> (4 - k)) & 1
        short = torch.zeros(nb, dtype=torch.bool, device=device)
        for d in SHORT_DFS:
            short |= dfs == d
        data = torch.stack([bits, 1 - bits], dim=2).reshape(nb, 224)
        data[:, 112:] *= (~short).to(torch.int64)[:, None]          # a 56-bit reply is silent after its last bit
    chips = torch.cat([pre, data], dim=1).to(torch.float32)  # [nb, 240]
    env = chips.repeat_interleave(half, dim=1)  # [nb, blen]
    if snr_db_range is not None:
        p = noise_power * torch.pow(10.0, torch.empty(nb, device=device).uniform_(snr_db_range[0], snr_db_range[1], generator=g) / 10.0)
    else:
        p = torch.empty(nb, device=device).uniform_(amp2_range[0], amp2_range[1], generator=g)
    ph = torch.empty(nb, device=device).uniform_(0.0, 2 * np.pi, generator=g)
    a = torch.sqrt(p)
    idx = (starts[:, None] + torch.arange(blen, device=device)[None, :]).reshape(-1)
    # process in slices to bound temporary memory
    step = max(1, (1 << 24) // blen)
    # overlapping bursts hit the same samples: the accumulation must be order-deterministic (atomics are not), or two
    # ranks that generate the same block -- the overlap of neighbouring shards -- would differ in the last bit
    det = torch.are_deterministic_algorithms_enabled()
    torch.use_deterministic_algorithms(True)
    try:
        for b0 in range(0, nb, step):
            b1 = min(nb, b0 + step)
            e = env[b0:b1]
            wi = (e * (a[b0:b1] * torch.cos(ph[b0:b1]))[:, None]).reshape(-1)
            wq = (e * (a[b0:b1] * torch.sin(ph[b0:b1]))[:, None]).reshape(-1)
            ii = idx[b0 * blen:b1 * blen]
            iq[:, 0].index_add_(0, ii, wi)
            iq[:, 1].index_add_(0, ii, wq)
    finally:
        torch.use_deterministic_algorithms(det)
    return iq
