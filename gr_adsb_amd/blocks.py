"""Drop-in `framer` / `demod` blocks: the reference's GNU Radio sync-block surface over the HIP path.

Same constructor arguments, block names, port signatures, history, tag (key "burst", value
("SOB", snr), srcid "framer") and PDU (dict{timestamp, snr} . u8vector[112] on port "demodulated") as
/root/reference/python/adsb/framer.py and demod.py; the arithmetic of work() runs on the GPU through
the C ABI (adsb_framer_work / adsb_demod_work).  With GNU Radio installed these subclass the real
gr.sync_block; without it they subclass grshim.sync_block so tests can drive them call by call.
"""
import collections
import datetime
import threading

import numpy as np

try:  # pragma: no cover - GNU Radio is not in this image
    import pmt
    from gnuradio import gr
    HAVE_GNURADIO = True
except ImportError:
    from . import grshim as gr
    pmt = gr.pmt
    HAVE_GNURADIO = False

from . import _native

SYMBOL_RATE = 1e6
NUM_PREAMBLE_BITS = 8
MAX_NUM_BITS = 112
_FEW = 6                                  # bursts per work() call up to which the per-burst scalar paths are used
_F10, _F16, _INF = np.float32(10.0), np.float32(1.6), np.float32(np.inf)


# Page-locking the buffer a work() call's input is a view OF (round 6): the library DMA's a page-locked source where it lies
# (adsb_process_* / adsb_framer_work: ~0.13 instead of ~0.21 ms per megasample call, small calls are read in place over PCIe)
# instead of copying it through its staging buffers first.  A work() call only sees a slice; what can be page-locked ONCE is the
# array that owns the memory -- found through the slice's .base chain.  That is the case for file replay, tests and any
# Python-side driver that hands slices of one ring; the real GNU Radio gateway wraps the scheduler's buffer in a fresh
# array per call with no owner to find, so nothing is registered there (an application that owns such a ring registers it
# itself: adsb_host_register / _native.RegisteredArray).  Registrations end when the owning array dies.
_PIN_MIN, _PIN_MAX, _PIN_SLOTS = 1 << 20, 1 << 32, 8
_pinned_roots = {}
_pin_lock = threading.Lock()


def pin_source(arr):
    """Page-lock (once) the ndarray that owns arr's memory, if there is one of 1 MiB .. 4 GiB; no-op otherwise."""
    root = arr
    while isinstance(getattr(root, "base", None), np.ndarray):
        root = root.base
    if root is arr and arr.base is not None:
        return False                                              # a view of something that is not an ndarray: no owner to find
    if not isinstance(root, np.ndarray) or not root.flags.owndata or not root.flags.c_contiguous:
        return False
    if not (_PIN_MIN <= root.nbytes <= _PIN_MAX):
        return False
    key = root.ctypes.data
    with _pin_lock:
        if key in _pinned_roots:
            return True
        if len(_pinned_roots) >= _PIN_SLOTS:
            return False
        import ctypes
        lib = _native.load()
        # (no reference to the array is kept: the registration must not keep its owner alive -- it ends WITH it)
        if lib.adsb_host_register(ctypes.c_void_p(key), root.nbytes) != 0:
            return False                                          # (already registered by the application, or not lockable)
        _pinned_roots[key] = root.nbytes
    import weakref
    try:
        weakref.finalize(root, _unpin, key)
    except TypeError:                                             # (an ndarray subclass without weak references: lock it for good)
        pass
    return True


def _unpin(key):
    import ctypes
    with _pin_lock:
        had = _pinned_roots.pop(key, None)
    if had is not None:
        _native.load().adsb_host_unregister(ctypes.c_void_p(key))


def unpin_all():
    """Drop every registration pin_source made (they also end one by one when their owning arrays die)."""
    with _pin_lock:
        keys = list(_pinned_roots)
    for k in keys:
        _unpin(k)


def make_pdu(start_timestamp, fs, offset, snr, bits112):
    """The PDU the reference demod publishes for one burst (demod.py:104-110): a pair
    (dict{"timestamp": start + offset/fs, "snr": snr}, u8vector of 112 0/1 values) -- what decoder.py:330-335
    reads back with pmt.car / pmt.cdr."""
    meta = pmt.to_pmt({
        "timestamp": start_timestamp + offset / fs,
        "snr": snr,
    })
    return pmt.cons(meta, pmt.to_pmt(np.ascontiguousarray(bits112, dtype=np.uint8)))


class _SliceStore:
    """Bits the framer's device pass has already sliced, handed to the paired demod (demod(fs, framer=...)).

    GNU Radio runs every block's work() on its own thread, so the framer's put() and the demod's take() race by design:
    both run under one lock and exchange whole per-call ARRAYS in stream order (a deque of (offsets, bits14, flags), one
    entry per framer call that produced PDUs) -- no per-burst dictionary traffic.  The demod sees tags in stream order, so
    take() consumes from the front and forgets everything up to the last offset it was asked for.  Should the demod stall
    (GNU Radio's bounded buffers normally make that impossible) the oldest calls are forgotten beyond `cap` bursts -- never
    the newest one, however large -- and the demod then slices those bursts on the device itself (demod.work's fall-back)."""

    def __init__(self, cap=1 << 18):
        self.cap = int(cap)
        self.evicted = 0
        self._n = 0
        self._q = collections.deque()
        self._lock = threading.Lock()

    def __len__(self):
        return self._n

    def put(self, offs, bits14, flags, offs_list=None):
        """offs_list: the same offsets as a Python list, when the caller has it anyway (a scheduler-sized call with a
        burst or two): take_few() then needs no NumPy call to find them."""
        with self._lock:
            self._q.append((offs, bits14, flags, offs_list))
            self._n += len(offs)
            while self._n > self.cap and len(self._q) > 1:
                self._n -= len(self._q[0][0])
                self.evicted += len(self._q[0][0])
                self._q.popleft()

    def take(self, offs):
        """offs: int64, increasing.  Returns (found[n] bool, bits14[n, 14], flags[n]); rows of bursts that are not (or no
        longer) stored are zero.  Everything stored at or in front of offs[-1] is forgotten."""
        n = len(offs)
        found = np.zeros(n, dtype=bool)
        bits = np.zeros((n, 14), dtype=np.uint8)
        flags = np.zeros(n, dtype=np.uint16)
        last = int(offs[-1])
        with self._lock:
            while self._q:
                so, sb, sf, _ = self._q[0]
                i = np.searchsorted(so, offs)
                hit = i < len(so)
                hit[hit] = so[i[hit]] == offs[hit]
                if hit.any():
                    found |= hit
                    bits[hit] = sb[i[hit]]
                    flags[hit] = sf[i[hit]]
                if int(so[-1]) <= last:                     # consumed to its end
                    self._n -= len(so)
                    self._q.popleft()
                    continue
                k = int(np.searchsorted(so, last, side="right"))
                if k:                                       # the demod's chunk ended inside this framer call: keep the rest
                    self._q[0] = (so[k:], sb[k:], sf[k:], None)
                    self._n -= k
                break
        return found, bits, flags

    def take_few(self, offs):
        """take() for a handful of tags (a scheduler-sized call: Python ints in, no NumPy arithmetic): offs is an increasing
        list of ints; returns a list with, per tag, None or (bits14 row, flags).  Forgets like take()."""
        res = [None] * len(offs)
        last = offs[-1]
        with self._lock:
            while self._q:
                so, sb, sf, sl = self._q[0]
                if sl is None:
                    sl = so.tolist()
                for j, o in enumerate(offs):
                    if res[j] is None and sl[0] <= o <= sl[-1]:
                        try:
                            i = sl.index(o)
                        except ValueError:
                            continue
                        res[j] = (sb[i], int(sf[i]))
                if sl[-1] <= last:                          # consumed to its end
                    self._n -= len(sl)
                    self._q.popleft()
                    continue
                k = 0
                while sl[k] <= last:                        # (sl[-1] > last: terminates)
                    k += 1
                if k:
                    self._q[0] = (so[k:], sb[k:], sf[k:], sl[k:])
                    self._n -= k
                else:
                    self._q[0] = (so, sb, sf, sl)           # keep the list for the next call
                break
        return res


class framer(gr.sync_block):
    """ADS-B preamble detector / tagger (reference python/adsb/framer.py:33-182)."""

    def __init__(self, fs, threshold, device=0, improved=False, long_aware=False, min_chunk=0, pin_inputs=True):
        """pin_inputs (extension, default on): page-lock, once, the array that owns the memory a work() call's input is a slice
        of (pin_source above); harmless where no such owner exists (the real GNU Radio gateway).
        min_chunk (extension, default 0 = whatever the scheduler hands over, like the reference): when > 0 the block asks
        the scheduler for work() calls of a multiple of that many items (gr.basic_block.set_output_multiple): a call costs
        tens of microseconds whatever its size, so large chunks are what lifts the block from tens of Msamples/s to
        Gsamples/s (tools/gr_latency.py).  Chunking is the scheduler's freedom in the reference too (framer.py:72-77);
        results under any chunking equal the reference's under the same chunking.

        improved (extension; default off = the reference's behaviour bit for bit): the tags no
        longer depend on how the scheduler chunks the stream -- pulses straddling a work() boundary are evaluated
        (framer.py:98-108 drops them), the re-trigger state never goes stale (framer.py:177-179) -- and equal those of
        ONE reference work() call over the whole stream.  Price: the block delays its output by `self.delay` samples
        (256 + 121*sps: the look-ahead a complete pulse + burst needs), so that every tag lands on a sample it has
        not produced yet; tag values become ("SOB", snr, input_offset).  Pulses longer than 256 samples are not
        evaluated.  long_aware (needs improved): the gate additionally holds for the whole length of a 112-bit reply
        (first data bit set) instead of always assuming a short one (framer.py:163-165) -- fewer false tags inside long
        replies; this changes the tag set itself, not just its chunk dependence."""
        gr.sync_block.__init__(self, name="ADS-B Framer", in_sig=[np.float32], out_sig=[np.float32])
        self.improved = bool(improved)
        if long_aware and not improved:
            raise ValueError("long_aware needs improved=True (the reference-exact mode keeps the reference's gate)")
        self.long_aware = bool(long_aware)
        self.fs = fs
        assert self.fs % SYMBOL_RATE == 0, \
            "ADS-B Framer is designed to operate on an integer number of samples per symbol, not %f sps" % (self.fs / SYMBOL_RATE)
        self.sps = int(fs // SYMBOL_RATE)
        if self.sps % 2:
            # the reference constructs but raises inside work() (slice step 0 / 24 taps vs 16)
            raise ValueError("fs must be an even multiple of 1 MHz (the reference's tap stride is sps//2)")
        self.threshold = threshold
        self.N_hist = NUM_PREAMBLE_BITS * self.sps
        self.delay = 0
        if self.improved:
            self._back = 100 + 8 * self.sps + 4          # noise window + preamble span behind the first owned rise
            self.delay = 256 + 121 * self.sps             # longest pulse followed + preamble + 112 bits ahead of the last
            self.N_hist = self._back + self.delay + 1
            self._eob = _native.EOB_NONE                  # end-of-burst state carried between calls (stream offsets)
        self.set_history(self.N_hist)
        self.set_tag_propagation_policy(gr.TPP_ONE_TO_ONE)
        if min_chunk:
            self.set_output_multiple(int(min_chunk))
        # the framer's device pass also slices the bits of every burst that ends inside its chunk: a demod paired with
        # this block (demod(fs, framer=this)) publishes them without a second upload / device pass of the same samples
        self._ctx = _native.Context(fs, threshold, device=device,
                                    flags=(_native.FLAG_LONG_AWARE_GATE if self.long_aware else 0) |
                                          (0 if self.improved else _native.FLAG_FRAMER_SLICES))
        self.pin_inputs = bool(pin_inputs)
        self._pin_seen = None             # address of the last owner looked at (one dictionary lookup per call at most)
        self._slices = _SliceStore()      # bits of the bursts this block's pass already sliced, for a paired demod
        self._paired = False              # set by demod(fs, framer=self): only then are the slices kept
        self._pmt_key, self._pmt_src = pmt.to_pmt("burst"), pmt.to_pmt("framer")

    def set_threshold(self, threshold):
        self.threshold = threshold            # read once per work(), like the reference (framer.py:84)

    # the reference keeps its cross-call state as public attributes of the block (framer.py:54,57); here it lives in the
    # library's context (adsb_framer_work) and is read back on demand.  Not meaningful with improved=True.
    @property
    def prev_eob_idx(self):
        return None if self.improved else self._ctx.framer_state()[1]

    @property
    def prev_in0(self):
        return None if self.improved else self._ctx.framer_state()[0]

    def _work_improved(self, in0, out0):
        """One overlapped shard of the unbounded input stream per call (the multi-GPU stitching machinery used in
        time): in0 = [back halo | N owned samples | look-ahead]; owned = pulse rises in input offsets
        [nread - delay, nread - delay + N); the gate state crosses calls as one number (adsb_shard_fixup)."""
        N = len(out0)
        nread = self.nitems_read(0)
        B, F = self._back, self.delay
        buf = in0[:N + B + F]
        origin = nread - (B + F)                          # input offset of in0[0] (negative: GR's zero history)
        own_lo, own_hi = nread - F, nread - F + N
        kept = None
        for hc in (64, _native.MAX_HEAD):
            recs = self._ctx.shard_host(_native.FMT_MAG2, buf, origin, own_lo, own_hi, _native.STREAM_UNBOUNDED,
                                        head_cands=hc, drop_overlong=True)
            kept = _native.shard_fixup(recs, self.sps, self._eob)
            if kept is not None:
                break
        if kept is None:                                  # an unbroken chain of overlapping bursts longer than the head
            from .replay import greedy_gate
            kept = greedy_gate(self._ctx.shard_host(_native.FMT_MAG2, buf, origin, own_lo, own_hi, _native.STREAM_UNBOUNDED,
                                                    head_cands=0, drop_overlong=True), self.sps, self._eob)
        if len(kept):
            self._eob = int(kept["offset"][-1]) + int(_native.gate_window(kept[-1:], self.sps)[0])
        snr = _native.snr_db(kept["peak"], kept["median"])
        for b, s_ in zip(kept, snr):
            off = int(b["offset"])
            self.add_item_tag(0, off + F, pmt.to_pmt("burst"),
                              pmt.to_pmt(("SOB", float(s_) if HAVE_GNURADIO else s_, off)), pmt.to_pmt("framer"))
        out0[:] = in0[B:B + N]                            # the input delayed by F samples
        return N

    def work(self, input_items, output_items):
        in0 = input_items[0]
        out0 = output_items[0]
        N = len(out0)
        self._ctx.set_threshold_cached(self.threshold)
        if self.improved:
            return self._work_improved(in0, out0)
        if self.pin_inputs and in0.nbytes >= (64 << 10):
            b = in0.base
            if b is not None and id(b) != self._pin_seen:
                self._pin_seen = id(b)
                pin_source(in0)
        # chunks of 256 KiB and more: the pass-through copy (framer.py:181) runs inside the library beside the device pass
        fused_copy = out0.nbytes >= _FUSE_COPY_BYTES and out0.flags.c_contiguous and out0.dtype == np.float32
        bursts = self._ctx.framer_work(in0[:N + self.N_hist - 1], N, self.nitems_written(0), out0=out0 if fused_copy else None)
        nb = len(bursts)
        if nb:                                            # (most scheduler-sized work() calls carry no burst)
            key, src, add, to_pmt = self._pmt_key, self._pmt_src, self.add_item_tag, pmt.to_pmt
            # one tag per burst is the only per-burst work the API forces (framer.py:168-174).  real pmt.to_pmt wants Python
            # floats; under the stand-in runtime the value stays np.float32 like the reference's
            peak, med = bursts["peak"], bursts["median"]
            if nb <= _FEW and bool(((med > 0) & (med < _INF)).all()) and bool((peak > 0).all()):
                # a burst or two (a scheduler-sized call): scalar arithmetic, exactly the reference's own expression on
                # np.float32 scalars (framer.py:157); no array temporaries, no errstate (nothing divides by zero here)
                offs = bursts["offset"].tolist()
                for j in range(nb):
                    s_ = _F10 * np.log10(peak[j] / med[j]) + _F16
                    add(0, offs[j], key, to_pmt(("SOB", float(s_) if HAVE_GNURADIO else s_)), src)
                if self._paired:
                    fl = bursts["flags"]
                    if bool((fl & _native.BURST_DEMOD).all()):
                        self._slices.put(bursts["offset"], bursts["bits"], fl, offs)
                    else:
                        dem = (fl & _native.BURST_DEMOD) != 0
                        if dem.any():
                            self._slices.put(bursts["offset"][dem], bursts["bits"][dem], fl[dem])
            else:
                snr = _native.snr_db(peak, med)
                if self._paired:
                    fl = bursts["flags"]
                    dem = (fl & _native.BURST_DEMOD) != 0
                    if dem.all():                         # the usual case: every burst ends inside the chunk
                        self._slices.put(bursts["offset"], bursts["bits"], fl)
                    elif dem.any():
                        self._slices.put(bursts["offset"][dem], bursts["bits"][dem], fl[dem])
                for off, s_ in zip(bursts["offset"].tolist(), snr.tolist() if HAVE_GNURADIO else list(snr)):
                    add(0, off, key, to_pmt(("SOB", s_)), src)
        if not fused_copy:
            _passthrough(self._ctx, out0, in0[self.N_hist - 1:])
        return N


_FUSE_COPY_BYTES = 256 << 10


def _passthrough(ctx, out0, src):
    """out0[:] = src (framer.py:181, demod.py:135); chunks of a megabyte and more through the library's copy threads."""
    if out0.nbytes >= (1 << 20) and out0.flags.c_contiguous and src.flags.c_contiguous and src.dtype == out0.dtype:
        ctx.host_copy(out0, src)
    else:
        out0[:] = src


_DF_WEIGHTS = np.array([16, 8, 4, 2, 1])
_PI_FORMATS = (11, 17, 18, 19)


def _prefilter_pass(flags, df):
    """A PDU can still pass decoder.check_parity(): known DF, and for the parity/interrogator formats a zero
    syndrome (address/parity formats need the decoder's aircraft table and always go through)."""
    if not flags & _native.BURST_KNOWN_DF:
        return False
    return bool(flags & _native.BURST_PARITY_OK) if df in _PI_FORMATS else True


class demod(gr.sync_block):
    """PPM bit slicer / PDU publisher (reference python/adsb/demod.py:31-136)."""

    def __init__(self, fs, device=0, parity_filter=False, improved=False, framer=None, min_chunk=0):
        """framer (extension, default None = an independent block, like the reference's): the framer block of the same
        flowgraph whose output feeds this block.  That framer's device pass has already sliced the bits of every burst
        that ends inside its chunk; a paired demod publishes those PDUs straight from the framer's records -- the same
        samples are not uploaded and scanned a second time -- and only goes to the device itself for a burst that was
        incomplete in the framer's chunk but is complete in its own.  The drop rule (demod.py:82) is applied to THIS
        block's chunk either way, so the PDU set under any pair of schedules equals the reference's.  In paired mode
        `bit_confidence` (demod.py:101, never published) is only maintained when `want_confidence` is set.
        min_chunk: see framer.

        improved (extension; default off): a burst that straddles the end of a work() chunk is
        completed with the next chunk's samples instead of being dropped for good (demod.py:61-64,130-133 is a stub
        that never completes it); with a 3-element tag value from the improved framer the PDU timestamp uses the
        undelayed input offset.

        parity_filter (extension, default off = the reference's behaviour: every PDU is published): when
        True, PDUs the decoder's check_parity() would reject outright -- unknown DF, or DF 11/17/18/19 with
        a non-zero syndrome (decoder.py:560-688) -- are counted in `self.filtered` and not published.  Only
        for decoders run with error_corr="None": a dropped PDU can no longer be repaired by their FEC."""
        gr.sync_block.__init__(self, name="demod", in_sig=[np.float32], out_sig=[np.float32])
        self.parity_filter = bool(parity_filter)
        self.filtered = 0
        self.improved = bool(improved)
        self._carry = np.zeros(0, dtype=np.float32)      # improved: tail of the previous chunk(s) ...
        self._carry_pos = 0                               # ... and the stream offset of its first sample
        self._pending = []                                # improved: tags whose burst was not complete yet
        self.fs = fs
        assert self.fs % SYMBOL_RATE == 0, \
            "ADS-B Demodulator is designed to operate on an integer number of samples per symbol, not %f sps" % (self.fs / SYMBOL_RATE)
        self.sps = int(fs // SYMBOL_RATE)
        if self.sps % 2:
            raise ValueError("fs must be an even multiple of 1 MHz")
        self.start_timestamp = (datetime.datetime.utcnow() - datetime.datetime(1970, 1, 1)).total_seconds()
        self.bits = []
        self.bit_idx = 0
        self.straddled_packet = 0
        self._framer = framer
        if framer is not None and (framer.improved or self.improved):
            raise ValueError("framer= pairing is for the reference-exact blocks (improved=False on both)")
        if framer is not None:
            framer._paired = True
        self.device_calls = 0             # work() calls that went to the device (paired mode: only the fall-back)
        self.want_confidence = framer is None  # demod.py:101 computes it on every burst
        self.set_tag_propagation_policy(gr.TPP_ONE_TO_ONE)
        if min_chunk:
            self.set_output_multiple(int(min_chunk))
        self._pmt_port, self._pmt_key = pmt.to_pmt("demodulated"), pmt.to_pmt("burst")
        self.message_port_register_out(self._pmt_port)
        self._ctx = _native.Context(fs, 0.0, device=device)

    def work(self, input_items, output_items):
        in0 = input_items[0]
        out0 = output_items[0]
        if self.straddled_packet == 1:
            self.straddled_packet = 0
        nread = self.nitems_read(0)
        tags = self.get_tags_in_range(0, nread, nread + len(in0), self._pmt_key)
        if self.improved:
            self._work_improved(in0, nread, tags)
        elif len(tags):
            bits = None
            if self._framer is not None and not self.want_confidence and len(tags) <= _FEW:
                # a tag or two (a scheduler-sized call): Python ints, no array arithmetic
                offl = [t.offset for t in tags]
                end = self.nitems_written(0) + len(in0) - (119 * self.sps + self.sps // 2)
                got = self._framer._slices.take_few(offl)
                if all((g is not None) or o >= end for g, o in zip(got, offl)):
                    ok = np.array([o < end for o in offl], dtype=bool)
                    bits = np.zeros((len(offl), 112), dtype=np.uint8)
                    pf = np.zeros(len(offl), dtype=np.uint16)
                    for j, g in enumerate(got):
                        if g is not None and ok[j]:
                            bits[j] = np.unpackbits(g[0])[:112]
                            pf[j] = g[1]
                    ratio = None
                offs = None
            else:
                offs = np.fromiter((t.offset for t in tags), dtype=np.int64, count=len(tags))
            if offs is not None and self._framer is not None and not self.want_confidence:
                # bits the paired framer's pass already holds; this block's own drop rule (demod.py:76,82)
                end = self.nitems_written(0) + len(in0)
                ok = offs + (119 * self.sps + self.sps // 2) < end
                found, b14, pf = self._framer._slices.take(offs)
                if bool((found | ~ok).all()):
                    bits = np.unpackbits(b14, axis=1)[:, :112]       # ONE call for the whole chunk
                    ratio = None
            if bits is None:
                if offs is None:
                    offs = np.array(offl, dtype=np.int64)
                # demod.py:79 indexes with nitems_written(0); equal to nitems_read(0) for this sync block
                bits, ok, ratio = self._ctx.demod_work(in0, self.nitems_written(0), offs, want_ratio=self.want_confidence)
                pf = self._ctx.last_demod_flags
                self.device_calls += 1
            # one PDU per burst is the only per-burst work the API forces (demod.py:104-110)
            ts0, fs, port, pub = self.start_timestamp, self.fs, self._pmt_port, self.message_port_pub
            to_pmt, to_python, cons = pmt.to_pmt, pmt.to_python, pmt.cons
            if ratio is None and not self.parity_filter and bool(ok.all()):
                for i, tag in enumerate(tags):
                    pub(port, cons(to_pmt({"timestamp": ts0 + tag.offset / fs, "snr": to_python(tag.value)[1]}), to_pmt(bits[i])))
                self.bits = bits[-1]                  # demod.py:95 keeps the last burst's bits on the block
            else:
                if ratio is not None:
                    with np.errstate(all="ignore"):
                        conf = np.float32(10.0) * np.log10(ratio)      # demod.py:101, the whole chunk at once
                for i, tag in enumerate(tags):
                    if not ok[i]:
                        self.straddled_packet = 1     # demod.py:130-133: dropped
                        continue
                    if self.parity_filter and not _prefilter_pass(int(pf[i]), int(bits[i][:5] @ _DF_WEIGHTS)):
                        self.filtered += 1
                        continue
                    self.bits = bits[i]
                    if ratio is not None:
                        self.bit_confidence = conf[i]                   # as it stands when this PDU is published
                    pub(port, cons(to_pmt({"timestamp": ts0 + tag.offset / fs, "snr": to_python(tag.value)[1]}), to_pmt(bits[i])))
        _passthrough(self._ctx, out0, in0)
        return len(out0)

    def _work_improved(self, in0, nread, tags):
        pend = self._pending + [(int(t.offset), pmt.to_python(t.value)) for t in tags]
        self._pending = []
        span = 120 * self.sps                             # samples a burst needs from its tag offset on (demod.py:76)
        if pend:
            first = min(o for o, _ in pend)
            if len(self._carry) and first < nread:
                buf = np.concatenate([self._carry, in0])
                pos = self._carry_pos
            else:
                buf, pos = in0, nread
            offs = np.array([o for o, _ in pend], dtype=np.int64)
            bits, ok, ratio = self._ctx.demod_work(buf, pos, offs, want_ratio=self.want_confidence)
            pf = self._ctx.last_demod_flags
            for i, (off, value) in enumerate(pend):
                if not ok[i]:
                    self._pending.append((off, value))    # completed by a later call
                    continue
                if self.parity_filter and not _prefilter_pass(int(pf[i]), int(bits[i][:5] @ _DF_WEIGHTS)):
                    self.filtered += 1
                    continue
                self.bits = bits[i].copy()
                if ratio is not None:
                    with np.errstate(all="ignore"):
                        self.bit_confidence = np.float32(10.0) * np.log10(ratio[i])
                t_off = value[2] if len(value) > 2 else off
                self.message_port_pub(pmt.to_pmt("demodulated"),
                                      make_pdu(self.start_timestamp, self.fs, t_off, value[1], self.bits))
        # keep the samples the still incomplete bursts start in (every tag of a chunk is visible in that chunk's call)
        if self._pending:
            whole = np.concatenate([self._carry, in0]) if len(self._carry) else in0
            whole_pos = self._carry_pos if len(self._carry) else nread
            keep_from = max(min(o for o, _ in self._pending), whole_pos)
            self._carry = whole[keep_from - whole_pos:].copy()
            self._carry_pos = keep_from
        else:
            self._carry = np.zeros(0, dtype=np.float32)
