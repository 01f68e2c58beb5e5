"""Drop-in `framer` / `demod` blocks: the reference's GNU Radio sync-block surface over the HIP path.

Same constructor arguments, block names, port signatures, history, tag (key "burst", value
("SOB", snr), srcid "framer") and PDU (dict{timestamp, snr} . u8vector[112] on port "demodulated") as
/root/reference/python/adsb/framer.py and demod.py; the arithmetic of work() runs on the GPU through
the C ABI (adsb_framer_work / adsb_demod_work).  With GNU Radio installed these subclass the real
gr.sync_block; without it they subclass grshim.sync_block so tests can drive them call by call.
"""
import datetime

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


def make_pdu(start_timestamp, fs, offset, snr, bits112):
    """The PDU the reference demod publishes for one burst (demod.py:104-110): a pair
    (dict{"timestamp": start + offset/fs, "snr": snr}, u8vector of 112 0/1 values) -- what decoder.py:330-335
    reads back with pmt.car / pmt.cdr."""
    meta = pmt.to_pmt({
        "timestamp": start_timestamp + offset / fs,
        "snr": snr,
    })
    return pmt.cons(meta, pmt.to_pmt(np.ascontiguousarray(bits112, dtype=np.uint8)))


class framer(gr.sync_block):
    """ADS-B preamble detector / tagger (reference python/adsb/framer.py:33-182)."""

    def __init__(self, fs, threshold, device=0):
        gr.sync_block.__init__(self, name="ADS-B Framer", in_sig=[np.float32], out_sig=[np.float32])
        self.fs = fs
        assert self.fs % SYMBOL_RATE == 0, \
            "ADS-B Framer is designed to operate on an integer number of samples per symbol, not %f sps" % (self.fs / SYMBOL_RATE)
        self.sps = int(fs // SYMBOL_RATE)
        if self.sps % 2:
            # the reference constructs but raises inside work() (slice step 0 / 24 taps vs 16)
            raise ValueError("fs must be an even multiple of 1 MHz (the reference's tap stride is sps//2)")
        self.threshold = threshold
        self.N_hist = NUM_PREAMBLE_BITS * self.sps
        self.set_history(self.N_hist)
        self.set_tag_propagation_policy(gr.TPP_ONE_TO_ONE)
        self._ctx = _native.Context(fs, threshold, device=device)

    def set_threshold(self, threshold):
        self.threshold = threshold            # read once per work(), like the reference (framer.py:84)

    @property
    def prev_eob_idx(self):
        return None

    def work(self, input_items, output_items):
        in0 = input_items[0]
        out0 = output_items[0]
        N = len(out0)
        self._ctx.set_threshold(self.threshold)
        bursts = self._ctx.framer_work(in0[:N + self.N_hist - 1], N, self.nitems_written(0))
        snr = _native.snr_db(bursts["peak"], bursts["median"])
        for b, s in zip(bursts, snr):
            self.add_item_tag(
                0,
                int(b["offset"]),
                pmt.to_pmt("burst"),
                pmt.to_pmt(("SOB", float(s) if HAVE_GNURADIO else s)),
                pmt.to_pmt("framer"),
            )
        out0[:] = in0[self.N_hist - 1:]
        return N


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

    def __init__(self, fs, device=0, parity_filter=False):
        """parity_filter (extension, default off = the reference's behaviour: every PDU is published): when
        True, PDUs the decoder's check_parity() would reject outright -- unknown DF, or DF 11/17/18/19 with
        a non-zero syndrome (decoder.py:560-688) -- are counted in `self.filtered` and not published.  Only
        for decoders run with error_corr="None": a dropped PDU can no longer be repaired by their FEC."""
        gr.sync_block.__init__(self, name="demod", in_sig=[np.float32], out_sig=[np.float32])
        self.parity_filter = bool(parity_filter)
        self.filtered = 0
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
        self.want_confidence = True           # demod.py:101 computes it on every burst
        self.set_tag_propagation_policy(gr.TPP_ONE_TO_ONE)
        self.message_port_register_out(pmt.to_pmt("demodulated"))
        self._ctx = _native.Context(fs, 0.0, device=device)

    def work(self, input_items, output_items):
        in0 = input_items[0]
        out0 = output_items[0]
        if self.straddled_packet == 1:
            self.straddled_packet = 0
        nread = self.nitems_read(0)
        tags = self.get_tags_in_range(0, nread, nread + len(in0), pmt.to_pmt("burst"))
        if len(tags):
            offs = np.array([t.offset for t in tags], dtype=np.int64)
            # demod.py:79 indexes with nitems_written(0); equal to nitems_read(0) for this sync block
            bits, ok, ratio = self._ctx.demod_work(in0, self.nitems_written(0), offs, want_ratio=self.want_confidence)
            pf = self._ctx.last_demod_flags
            for i, tag in enumerate(tags):
                if not ok[i]:
                    self.straddled_packet = 1     # demod.py:130-133: dropped
                    continue
                if self.parity_filter and not _prefilter_pass(int(pf[i]), int(bits[i][:5] @ _DF_WEIGHTS)):
                    self.filtered += 1
                    continue
                value = pmt.to_python(tag.value)
                snr = value[1]
                self.bits = bits[i].copy()
                if ratio is not None:
                    with np.errstate(all="ignore"):
                        self.bit_confidence = np.float32(10.0) * np.log10(ratio[i])
                self.message_port_pub(pmt.to_pmt("demodulated"),
                                      make_pdu(self.start_timestamp, self.fs, tag.offset, snr, self.bits))
        out0[:] = in0
        return len(out0)
