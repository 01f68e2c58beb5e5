"""CPU ORACLE (test infrastructure, NOT product code) for the gr-adsb framer + demod hot path.

A NumPy restatement of the reference algorithm, written from the reference's observed behaviour.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (gr_adsb_amd/) never does, and it has no CPU fallback.

Parity pin: this file is checked against the REAL reference (imported unmodified with a stubbed
GNU Radio runtime, tools/ref_harness.py) by tests/test_oracle_vs_reference.py when /root/reference is
present, and against the golden vectors that tools/make_golden.py generated from the real reference
(tests/golden/*.npz) everywhere else.  The reference's own tests pin nothing on this path
(python/adsb/qa_framer.py:34-37 and qa_demod.py:34-37 are empty), and |IQ|^2 itself lives in GNU Radio
(gr-blocks complex_to_mag_squared, not vendored): for that one row parity is "unpinned" and is
defined here as float32 re*re + im*im with separately rounded products.

All citations are to /root/reference/python/adsb/.
"""
import numpy as np

NUM_PREAMBLE_BITS = 8          # framer.py:28
MIN_NUM_BITS = 56              # framer.py:29
NUM_NOISE_SAMPLES = 100        # framer.py:31
MAX_NUM_BITS = 112             # demod.py:29
# framer.py:50 -- chips that must exceed half the centre sample
_TEMPLATE = np.array([1, 0, 1, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0], dtype=bool)


def sps_of(fs):
    """framer.py:44-45 / demod.py:42-43: integer samples per symbol, asserted."""
    assert fs % 1e6 == 0, "ADS-B blocks need an integer number of samples per symbol"
    return int(fs // 1e6)


def mag2(iq):
    """|IQ|^2 as GNU Radio's complex_to_mag_squared feeds it (examples/adsb_rx.py:180): float32,
    two rounded products and one rounded add, no FMA."""
    iq = np.asarray(iq, dtype=np.complex64)
    re = np.ascontiguousarray(iq.real)
    im = np.ascontiguousarray(iq.imag)
    return re * re + im * im


def mag2_iq16(iq16, scale):
    """int16 interleaved IQ -> |IQ|^2 the way the HIP path's int16 input format defines it (no reference
    counterpart): component -> float32 exactly, one rounded multiply by float32(scale),
    then mag2."""
    v = np.asarray(iq16, dtype=np.int16).astype(np.float32) * np.float32(scale)
    re = np.ascontiguousarray(v[0::2])
    im = np.ascontiguousarray(v[1::2])
    return re * re + im * im


def mag2_iq8(iq8, scale, offset_binary=False):
    """8-bit interleaved IQ -> |IQ|^2 as the HIP path's ADSB_FMT_SC8 / ADSB_FMT_CU8 formats define it (no reference
    counterpart).  int8: component = f32(i8) * f32(scale).  Offset binary (RTL-SDR uint8):
    component = f32(2*u8 - 255) * f32(scale) -- the integer 2*u8-255 is exact, one rounded multiply."""
    if offset_binary:
        c = (2 * np.asarray(iq8, dtype=np.uint8).astype(np.int32) - 255).astype(np.float32)
    else:
        c = np.asarray(iq8, dtype=np.int8).astype(np.float32)
    v = c * np.float32(scale)
    re = np.ascontiguousarray(v[0::2])
    im = np.ascontiguousarray(v[1::2])
    return re * re + im * im


def snr_db(peak, med):
    """framer.py:157/159: 10.0*np.log10(in0[p]/median) + 1.6 evaluated in float32 (NumPy 2 promotion)."""
    with np.errstate(all="ignore"):
        return (np.float32(10.0) * np.log10(np.float32(peak) / np.float32(med)) + np.float32(1.6)).astype(np.float32)


def _median_f32(w):
    """np.median semantics on a float32 window: odd n -> middle, even n -> f32(a+b)/2, NaN if any
    NaN or n == 0 (framer.py:157,159 call np.median directly; this is what the HIP path mirrors)."""
    n = len(w)
    if n == 0:
        # np.median([]) is 0/0: the x86 default NaN, sign bit set (0xFFC00000)
        return np.array([0xFFC00000], dtype=np.uint32).view(np.float32)[0]
    if np.isnan(w).any():
        return np.float32(np.nan)          # a quiet NaN from the data propagates (0x7FC00000)
    s = np.sort(w)
    # a median of zero is +0.0 whatever the signs of the zeros: np.median is np.mean of the middle element(s), and that
    # sum starts from +0.0 (0.0 + -0.0 = +0.0) -- independent of how the partition ordered -0.0 and +0.0
    if n & 1:
        return np.float32(np.float32(s[n // 2]) + np.float32(0.0))
    with np.errstate(all="ignore"):
        return np.float32(np.float32(np.float32(s[n // 2 - 1] + s[n // 2]) / np.float32(2.0)) + np.float32(0.0))


class FramerState:
    """framer.py:54,57 -- the two words of cross-call state."""

    def __init__(self):
        self.prev_in0 = np.float32(0.0)
        self.prev_eob = -1


def pulses_of_call(in0, N, thr32, prev_in0):
    """framer.py:83-113: threshold -> edges -> pairing fix-ups -> integer centres.
    Returns (centres int64[], had_edges bool)."""
    a = np.empty(N + 1, dtype=bool)
    with np.errstate(invalid="ignore"):
        a[0] = np.float32(prev_in0) >= thr32               # framer.py:84 (prev sample prepended)
        a[1:] = in0[:N] >= thr32                           # float32 compare (NEP-50 weak scalar)
    rise = np.flatnonzero(a[1:] & ~a[:-1])                 # framer.py:91-92
    fall = np.flatnonzero(~a[1:] & a[:-1])                 # framer.py:93
    had = len(rise) > 0 and len(fall) > 0                  # framer.py:95
    if not had:
        return np.zeros(0, dtype=np.int64), False
    if fall[0] < rise[0]:                                  # framer.py:98-100
        fall = fall[1:]
    if len(rise) > len(fall):                              # framer.py:102-108 (only ever 1 extra)
        rise = rise[:-1]
    return ((fall + rise) // 2).astype(np.int64), True     # framer.py:113 (int mean truncates)


def match_preamble(in0, p, sps):
    """framer.py:137-147 for an array of centres p: 16 taps at stride sps//2, chip = tap > centre/2,
    hit iff all 16 chips equal the template."""
    half = sps // 2
    idx = p[:, None] + np.arange(16)[None, :] * half
    taps = in0[idx]
    with np.errstate(invalid="ignore"):
        chips = taps > (in0[p] / np.float32(2.0))[:, None]
    return (chips == _TEMPLATE[None, :]).all(axis=1)


def framer_call(in0, N, sps, threshold, state, nitems_written, long_aware=False):
    """One framer.work() call (framer.py:72-182).  in0 has N + 8*sps - 1 items.
    Returns dict(tag_offsets, peak, median, snr, cand_idx) ; mutates state.

    long_aware=True is NOT the reference: it restates the opt-in gate of ADSB_FLAG_LONG_AWARE_GATE,
    which holds 119*sps after a burst whose first data bit (demod.py:87-95, k = 0; samples past the end read as 0) is
    set and 63*sps otherwise, instead of framer.py:165's fixed 63*sps.  Only meaningful for one call over a stream."""
    in0 = np.asarray(in0, dtype=np.float32)
    H = NUM_PREAMBLE_BITS * sps
    assert len(in0) == N + H - 1
    thr32 = np.float32(threshold)
    centres, had = pulses_of_call(in0, N, thr32, state.prev_in0)
    state.prev_in0 = np.float32(in0[N - 1])                # framer.py:87
    acc_p, acc_peak, acc_med = [], [], []
    cand = np.zeros(0, dtype=np.int64)
    if had:
        if len(centres):
            cand = centres[match_preamble(in0, centres, sps)]
        eob = state.prev_eob
        for p in cand:                                     # framer.py:117-174, sparse form (H6)
            p = int(p)
            if p > eob:
                acc_p.append(p)
                acc_peak.append(in0[p])
                acc_med.append(_median_f32(in0[max(0, p - NUM_NOISE_SAMPLES):p]))  # framer.py:156-159
                eob = p + (NUM_PREAMBLE_BITS + MIN_NUM_BITS - 1) * sps             # framer.py:165
                if long_aware:
                    i1, i0 = p + NUM_PREAMBLE_BITS * sps, p + NUM_PREAMBLE_BITS * sps + sps // 2
                    b1 = in0[i1] if i1 < len(in0) else np.float32(0.0)
                    b0 = in0[i0] if i0 < len(in0) else np.float32(0.0)
                    if b1 > b0:
                        eob = p + (NUM_PREAMBLE_BITS + 112 - 1) * sps
        # framer.py:121-123: the first pulse past eob resets it to -1, matched or not
        if len(centres) and int(centres[-1]) > eob:
            eob = -1
        if eob >= N:                                       # framer.py:177-179 (inside the `if` of :95)
            eob -= N
        state.prev_eob = eob
    acc_p = np.array(acc_p, dtype=np.int64)
    peak = np.array(acc_peak, dtype=np.float32)
    med = np.array(acc_med, dtype=np.float32)
    return dict(
        tag_offsets=nitems_written - (H - 1) + acc_p,      # framer.py:170
        peak=peak, median=med, snr=snr_db(peak, med) if len(peak) else np.zeros(0, np.float32),
        cand_idx=cand,
    )


def demod_call(in0, sps, nitems_read, tag_offsets):
    """One demod.work() call (demod.py:57-136) for the tags whose offset lies in this chunk.
    Returns (sel, bits[n,112] u8, ratio[n,112] f32, conf[n,112] f32): sel indexes tag_offsets of the
    bursts that were demodulated (the rest, inside the chunk but straddling its end, are dropped)."""
    in0 = np.asarray(in0, dtype=np.float32)
    n = len(in0)
    tag_offsets = np.asarray(tag_offsets, dtype=np.int64)
    inside = np.flatnonzero((tag_offsets >= nitems_read) & (tag_offsets < nitems_read + n))  # demod.py:67
    sob = tag_offsets[inside] + 8 * sps - nitems_read                                        # demod.py:75,79
    eob = tag_offsets[inside] + (8 + 112 - 1) * sps + sps / 2 - nitems_read                  # demod.py:76,80
    ok = eob < n                                                                             # demod.py:82
    sel = inside[ok]
    sob = sob[ok]
    k = np.arange(MAX_NUM_BITS) * sps
    b1 = in0[sob[:, None] + k[None, :]]                    # demod.py:87-88
    b0 = in0[sob[:, None] + sps // 2 + k[None, :]]         # demod.py:91-92
    with np.errstate(all="ignore"):
        bits = (b1 > b0).astype(np.uint8)                  # demod.py:94-95 (strict, tie -> 0)
        ratio = (b1 / b0).astype(np.float32)
        conf = (np.float32(10.0) * np.log10(ratio)).astype(np.float32)  # demod.py:101
    return sel, bits.reshape(-1, 112), ratio.reshape(-1, 112), conf.reshape(-1, 112)


def run_stream(x, fs, threshold, schedule=None, demod_schedule=None, long_aware=False):
    """Whole-stream driver mirroring tools/ref_harness.run_reference: x is the float32 |IQ|^2 stream,
    schedule the framer chunk lengths (None = canonical single call), all tags delivered to demod."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    sps = sps_of(fs)
    L = len(x)
    H = NUM_PREAMBLE_BITS * sps
    assert not (long_aware and schedule is not None), "the long-aware gate is defined for one call over the stream"
    if schedule is None:
        schedule = [L]
    if demod_schedule is None:
        demod_schedule = schedule
    assert sum(schedule) == L and sum(demod_schedule) == L
    buf = np.concatenate([np.zeros(H - 1, dtype=np.float32), x])
    st = FramerState()
    offs, peak, med, snr, cands = [], [], [], [], []
    pos = 0
    for N in schedule:
        r = framer_call(buf[pos:pos + N + H - 1], N, sps, threshold, st, pos, long_aware=long_aware)
        offs.append(r["tag_offsets"]); peak.append(r["peak"]); med.append(r["median"]); snr.append(r["snr"])
        cands.append(r["cand_idx"] + pos - (H - 1))
        pos += N
    offs = np.concatenate(offs) if offs else np.zeros(0, np.int64)
    out = dict(H=H, tag_offsets=offs.astype(np.int64),
               tag_peak=np.concatenate(peak).astype(np.float32), tag_median=np.concatenate(med).astype(np.float32),
               tag_snr=np.concatenate(snr).astype(np.float32), cand_offsets=np.concatenate(cands).astype(np.int64),
               final_prev_eob=int(st.prev_eob), final_prev_in0=np.float32(st.prev_in0))
    pdu_idx, bits, ratio, conf = [], [], [], []
    pos = 0
    for N in demod_schedule:
        sel, b, r, c = demod_call(x[pos:pos + N], sps, pos, offs)
        pdu_idx.append(sel); bits.append(b); ratio.append(r); conf.append(c)
        pos += N
    pdu_idx = np.concatenate(pdu_idx) if pdu_idx else np.zeros(0, np.int64)
    order = np.argsort(pdu_idx, kind="stable")  # PDUs are published chunk by chunk == offset order
    pdu_idx = pdu_idx[order]
    out.update(pdu_tag_index=pdu_idx, pdu_offsets=offs[pdu_idx],
               pdu_bits=np.concatenate(bits).reshape(-1, 112)[order],
               pdu_ratio=np.concatenate(ratio).reshape(-1, 112)[order],
               pdu_conf=np.concatenate(conf).reshape(-1, 112)[order],
               pdu_snr=out["tag_snr"][pdu_idx])
    return out


def pack_bits(bits112):
    """[n,112] 0/1 -> [n,14] bytes, first transmitted bit = MSB of byte 0 (Mode-S hex order)."""
    return np.packbits(np.asarray(bits112, dtype=np.uint8).reshape(-1, 112), axis=1, bitorder="big")


def resolve_candidates(cand_offsets, sps, prev_eob=-1):
    """The greedy re-trigger gate of framer.py:121-123,165 over a sorted list of matched centres."""
    keep = np.zeros(len(cand_offsets), dtype=bool)
    eob = prev_eob
    for i, p in enumerate(cand_offsets):
        if p > eob:
            keep[i] = True
            eob = int(p) + 63 * sps
    return keep


# ---- Mode S parity pre-filter ------------------------------------------------------
# Restates the part of the reference DECODER that decides whether a PDU survives (decoder.py:550-556
# decode_header's DF, :560-688 check_parity, :693-714 compute_crc), so the flags the device attaches to
# every burst can be checked.  Pinned against the imported reference decoder by
# tests/test_oracle_vs_reference.py and against tests/golden/g_parity.npz.
CRC_POLY = np.array([1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 1], dtype=np.int64)  # decoder.py:269
DF_SHORT = (0, 4, 5, 11)                  # decoder.py:565,604: 56-bit replies
DF_LONG = (16, 17, 18, 19, 20, 21, 24)    # decoder.py:636,669: 112-bit replies
DF_PI = (11, 17, 18, 19)                  # parity/interrogator field: passes iff pi == crc (decoder.py:623,677)
FLAG_PARITY_OK, FLAG_LONG, FLAG_KNOWN_DF, DF_SHIFT = 32, 64, 128, 8


def compute_crc(data_bits):
    """decoder.py:693-714 over rows: long division of data * x^24 by the generator, remainder = 24 bits."""
    d = np.asarray(data_bits, dtype=np.int64)
    d = d.reshape(-1, d.shape[-1])
    nd = d.shape[1]
    w = np.concatenate([d, np.zeros((d.shape[0], 24), np.int64)], axis=1)   # decoder.py:704
    for ii in range(nd):                                                      # decoder.py:706-711
        rows = w[:, ii] == 1
        w[rows, ii:ii + 25] ^= CRC_POLY
    return w[:, nd:nd + 24]


def mode_s_parity(bits112):
    """[n,112] PDU bits -> dict(df, nbits, syndrome, flags): `syndrome` = crc ^ last 24 bits of the reply
    (== pi ^ crc for DF 11/17/18/19, == the announced address `aa` for the address/parity formats,
    decoder.py:577,647); `flags` as the device sets them (ADSB_BURST_PARITY_OK | _LONG | _KNOWN_DF | DF<<8)."""
    b = np.asarray(bits112, dtype=np.int64).reshape(-1, 112)
    n = b.shape[0]
    wts5 = 1 << np.arange(4, -1, -1)
    wts24 = 1 << np.arange(23, -1, -1)
    df = b[:, :5] @ wts5                                                     # decoder.py:551
    is_long = np.isin(df, DF_LONG)
    known = is_long | np.isin(df, DF_SHORT)
    nbits = np.where(is_long, 112, np.where(known, 56, 0))
    syn = np.zeros(n, np.int64)
    for L, rows in ((112, is_long), (56, ~is_long)):                         # unknown DFs: 56-bit reading reported
        if rows.any():
            crc = compute_crc(b[rows, :L - 24]) @ wts24
            tail = b[rows, L - 24:L] @ wts24
            syn[rows] = crc ^ tail
    ok = np.isin(df, DF_PI) & (syn == 0)
    flags = (df << DF_SHIFT) | np.where(is_long, FLAG_LONG, 0) | np.where(known, FLAG_KNOWN_DF, 0) | np.where(ok, FLAG_PARITY_OK, 0)
    return dict(df=df, nbits=nbits, syndrome=syn, parity_ok=ok, flags=flags.astype(np.uint16))
