"""CPU BASELINE / ORACLE (test infrastructure, NOT product code): the framer + demod hot path restated WITH THE
REFERENCE'S OWN COST STRUCTURE -- a vectorised NumPy front end (threshold, edges, pairing, centres) followed by
a pure-Python loop over every pulse (gate, 16-tap gather, template compare, np.median on a hit) and a Python loop
over every tag in the demod -- so that timing it on the GPU box's host cores says what the reference itself would
achieve there (the reference cannot travel).  SURVEY.md §8d-M4(b).

oracle/adsb_oracle.py is the *vectorised* NumPy oracle (all pulses matched at once, Python only over matches) and
oracle/adsb_oracle.c the scalar C port: both are faster than the reference and are reported beside this one.

Pinned: tests/test_ref_structured.py checks this module against the vectorised oracle (which is pinned to the real
reference and its goldens) on the golden streams and on random streams, and -- when /root/reference is present --
against the real reference directly.  Citations: /root/reference/python/adsb/framer.py, demod.py.

Only tests/ and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

_TEMPLATE = np.array([1, 0, 1, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0], dtype=bool)    # framer.py:50
NOISE = 100                                                                            # framer.py:31


class State:
    def __init__(self):
        self.prev_in0 = np.float32(0.0)      # framer.py:54
        self.prev_eob = -1                    # framer.py:57


def framer_work(in0, N, sps, threshold, st, nitems_written, stats=None):
    """One framer.work() call (framer.py:72-182): in0 = N + 8*sps - 1 float32 items, history first.
    Returns (tag_offsets int64[], snr float32[])."""
    H = 8 * sps
    thr = np.float32(threshold)
    half = sps // 2
    # -- vectorised front end (framer.py:83-113)
    with np.errstate(invalid="ignore"):
        above = np.empty(N + 1, dtype=np.int8)
        above[0] = np.float32(st.prev_in0) >= thr
        above[1:] = in0[:N] >= thr
    st.prev_in0 = np.float32(in0[N - 1])
    d = np.diff(above)
    rise = np.nonzero(d == 1)[0]
    fall = np.nonzero(d == -1)[0]
    offs, snrs = [], []
    if len(rise) and len(fall):
        if fall[0] < rise[0]:
            fall = fall[1:]
        if len(rise) > len(fall):
            rise = rise[:-1]
        centres = (rise + fall) // 2
        eob = st.prev_eob
        n_eval = 0
        # -- the per-pulse Python loop (framer.py:117-174): where the reference spends ~95 % of its time
        for p in centres.tolist():
            if p <= eob:                                     # framer.py:121
                continue
            eob = -1                                         # framer.py:123
            n_eval += 1
            taps = in0[p:p + 16 * half:half]                 # framer.py:137
            with np.errstate(invalid="ignore"):
                chips = taps > in0[p] / np.float32(2.0)      # framer.py:140-141
            if np.array_equal(chips, _TEMPLATE):             # framer.py:144-147
                with np.errstate(all="ignore"):
                    import warnings
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        med = np.median(in0[max(0, p - NOISE):p])                    # framer.py:156-159
                    snr = np.float32(10.0) * np.log10(in0[p] / med) + np.float32(1.6)
                eob = p + (8 + 56 - 1) * sps                 # framer.py:165
                offs.append(nitems_written - (H - 1) + p)    # framer.py:170
                snrs.append(snr)
        if eob >= N:                                         # framer.py:177-179
            eob -= N
        st.prev_eob = eob
        if stats is not None:
            stats["pulses"] = stats.get("pulses", 0) + len(centres)
            stats["evaluated"] = stats.get("evaluated", 0) + n_eval
    return np.asarray(offs, dtype=np.int64), np.asarray(snrs, dtype=np.float32)


def demod_work(in0, sps, nitems_read, tag_offsets):
    """One demod.work() call (demod.py:57-136): a Python loop over the chunk's tags.  Returns (indices into
    tag_offsets of the published PDUs, bits uint8[n,112], confidence float32[n,112])."""
    n = len(in0)
    half = sps // 2
    sel, bits, conf = [], [], []
    for i, off in enumerate(np.asarray(tag_offsets).tolist()):
        if not (nitems_read <= off < nitems_read + n):       # demod.py:67
            continue
        sob = off + 8 * sps - nitems_read                    # demod.py:75,79
        eob = off + (8 + 112 - 1) * sps + sps / 2 - nitems_read                      # demod.py:76,80
        if eob < n:                                          # demod.py:82
            b1 = in0[sob:sob + 112 * sps:sps]                # demod.py:87-88
            b0 = in0[sob + half:sob + half + 112 * sps:sps]  # demod.py:91-92
            with np.errstate(all="ignore"):
                bits.append((b1 > b0).astype(np.uint8))      # demod.py:94-95
                conf.append((np.float32(10.0) * np.log10(b1 / b0)).astype(np.float32))   # demod.py:101
            sel.append(i)
    return (np.asarray(sel, dtype=np.int64), np.asarray(bits, dtype=np.uint8).reshape(-1, 112),
            np.asarray(conf, dtype=np.float32).reshape(-1, 112))


def run_stream(x, fs, threshold, abs_offset=0, stats=None):
    """Canonical single call over a fresh stream (SURVEY.md §8a chunk semantics): framer then demod.
    Returns dict(tag_offsets, tag_snr, pdu_tag_index, pdu_bits, pdu_conf)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    sps = int(fs // 1e6)
    H = 8 * sps
    buf = np.concatenate([np.zeros(H - 1, dtype=np.float32), x])
    st = State()
    offs, snr = framer_work(buf, len(x), sps, threshold, st, 0, stats)
    sel, bits, conf = demod_work(x, sps, 0, offs)
    return dict(tag_offsets=offs + abs_offset, tag_snr=snr, pdu_tag_index=sel, pdu_bits=bits, pdu_conf=conf)


# ---- all host cores: one process per overlapped time shard, stitched by the parent ---------------------------
# Shard g owns the tags of [own_lo, own_hi).  It runs the reference-structured framer from WARM samples before own_lo
# with fresh state: the true gate state and the fresh one coincide from the first pulse that follows a pulse-free
# stretch of more than 63*sps samples (no matched centre in front of it can still hold the gate, framer.py:121,165),
# so if the warm-up region contains such a stretch the shard's tags inside its own range are exactly those of one
# call over the whole stream; the worker reports whether it found one.  Forward halo: 120*sps samples (demod.py:76).
WARM = 1 << 15


def shard_ranges(n, shards, sps):
    per = -(-n // shards)
    out = []
    for g in range(shards):
        own_lo, own_hi = min(n, g * per), min(n, (g + 1) * per)
        out.append((own_lo, own_hi, max(0, own_lo - WARM), min(n, own_hi + 121 * sps)))
    return out


def _warm_synced(x_warm, thr, sps):
    """True if the warm-up samples contain a stretch longer than 63*sps entirely below the threshold."""
    if len(x_warm) == 0:
        return True
    with np.errstate(invalid="ignore"):
        hi = np.flatnonzero(x_warm >= np.float32(thr))
    edges = np.concatenate([[-1], hi, [len(x_warm)]])
    return bool(np.max(np.diff(edges)) - 1 > 63 * sps)


def shard_worker(args):
    """(x_shard float32[lo:hi), lo, own_lo, own_hi, n_total, fs, thr) -> (tag offsets, snr, pdu sel, bits, synced)."""
    x, lo, own_lo, own_hi, n_total, fs, thr = args
    sps = int(fs // 1e6)
    r = run_stream(x, fs, thr, abs_offset=lo)
    synced = lo == 0 or _warm_synced(x[:own_lo - lo], thr, sps)
    offs = r["tag_offsets"]
    own = (offs >= own_lo) & (offs < own_hi)
    # PDUs: a burst whose end lies beyond this buffer but inside the stream was cut by the shard, not by the stream
    eob = offs + 119 * sps + sps / 2
    assert not np.any(own & (eob >= lo + len(x)) & (eob < n_total)), "forward halo too short"
    pdu = np.zeros(len(offs), dtype=bool)
    pdu[r["pdu_tag_index"]] = True
    keep_pdu = own[r["pdu_tag_index"]]
    return offs[own], r["tag_snr"][own], pdu[own], r["pdu_bits"][keep_pdu], synced


def run_sharded(x, fs, thr, shards, pool_map=map):
    """Whole stream on `shards` workers (pool_map = a process pool's map for real parallelism).  Returns the same
    dict as run_stream; falls back to the serial call if a shard could not synchronise (dense traffic, tiny shards)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    sps = int(fs // 1e6)
    n = len(x)
    jobs = [(x[lo:hi], lo, own_lo, own_hi, n, fs, thr) for own_lo, own_hi, lo, hi in shard_ranges(n, shards, sps) if own_hi > own_lo]
    res = list(pool_map(shard_worker, jobs))
    if not all(r[4] for r in res):
        out = run_stream(x, fs, thr)
        out["fallback"] = True
        return out
    offs = np.concatenate([r[0] for r in res])
    pdu = np.concatenate([r[2] for r in res])
    return dict(tag_offsets=offs, tag_snr=np.concatenate([r[1] for r in res]), pdu_tag_index=np.flatnonzero(pdu),
                pdu_bits=np.concatenate([r[3] for r in res]).reshape(-1, 112), fallback=False)
