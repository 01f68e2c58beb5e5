"""Drive the UNMODIFIED reference blocks (container-only tool, never shipped as a dependency).

Imports /root/reference/python/adsb/{framer,demod}.py by path with a stubbed GNU Radio
runtime and drives work() with an explicit chunk schedule.  Used to
(1) pin oracle/adsb_oracle.py against the real reference and (2) emit the golden vectors
under tests/golden/ (tools/make_golden.py).  Nothing under tests/, bench.py or the package
imports this module at run time on the GPU box: /root/reference does not exist there.
"""
import importlib.util
import sys
import types

import numpy as np

REF = "/root/reference/python/adsb"


class _Tag:
    __slots__ = ("offset", "key", "value", "srcid")

    def __init__(self, offset, key, value, srcid):
        self.offset, self.key, self.value, self.srcid = offset, key, value, srcid


class _SyncBlock:
    """Minimal gr.sync_block double: records tags/messages, counters set by the driver."""

    def __init__(self, name=None, in_sig=None, out_sig=None):
        self._name = name
        self._hist = 1
        self._nread = 0
        self._nwritten = 0
        self.tags_out = []
        self.tags_in = []
        self.msgs = []

    def set_history(self, n):
        self._hist = n

    def history(self):
        return self._hist

    def set_tag_propagation_policy(self, p):
        self._tpp = p

    def nitems_written(self, port):
        return self._nwritten

    def nitems_read(self, port):
        return self._nread

    def add_item_tag(self, port, offset, key, value, srcid):
        self.tags_out.append(_Tag(offset, key, value, srcid))

    def get_tags_in_range(self, port, start, end, key=None):
        return [t for t in self.tags_in if start <= t.offset < end and (key is None or t.key == key)]

    def message_port_register_out(self, name):
        pass

    def message_port_pub(self, port, msg):
        conf = getattr(self, "bit_confidence", None)
        self.msgs.append((port, msg, None if conf is None else np.array(conf, copy=True)))


def _install_stubs():
    if "pmt" in sys.modules and getattr(sys.modules["pmt"], "_adsb_stub", False):
        return
    pmt = types.ModuleType("pmt")
    pmt._adsb_stub = True
    pmt.to_pmt = lambda x: x
    pmt.to_python = lambda x: x
    pmt.cons = lambda a, b: (a, b)
    pmt.car = lambda p: p[0]
    pmt.cdr = lambda p: p[1]
    gnuradio = types.ModuleType("gnuradio")
    gr = types.ModuleType("gnuradio.gr")
    gr.sync_block = _SyncBlock
    gr.TPP_ONE_TO_ONE = 1
    gnuradio.gr = gr
    sys.modules["pmt"] = pmt
    sys.modules["gnuradio"] = gnuradio
    sys.modules["gnuradio.gr"] = gr


def _load(name):
    _install_stubs()
    spec = importlib.util.spec_from_file_location("_adsb_ref_" + name, "%s/%s.py" % (REF, name))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_cache = {}


def ref_modules():
    if not _cache:
        _cache["framer"] = _load("framer")
        _cache["demod"] = _load("demod")
    return _cache["framer"], _cache["demod"]


def run_reference(x, fs, threshold, schedule=None, demod_schedule=None):
    """x: float32 mag^2 stream.  schedule: list of chunk lengths N_j (sum == len(x)); None ==
    canonical single call.  Returns dict with framer tags and demod PDUs (reference objects'
    outputs reduced to arrays).  Demod uses demod_schedule (default: same as schedule)."""
    framer_mod, demod_mod = ref_modules()
    x = np.ascontiguousarray(x, dtype=np.float32)
    L = len(x)
    if schedule is None:
        schedule = [L]
    assert sum(schedule) == L
    if demod_schedule is None:
        demod_schedule = schedule
    assert sum(demod_schedule) == L

    fr = framer_mod.framer(fs, threshold)
    H = fr.history()
    buf = np.concatenate([np.zeros(H - 1, dtype=np.float32), x])
    pos = 0
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for N in schedule:
                in0 = buf[pos:pos + N + H - 1]
                out0 = np.empty(N, dtype=np.float32)
                fr._nread = fr._nwritten = pos
                ret = fr.work([in0], [out0])
                assert ret == N
                assert np.array_equal(out0, x[pos:pos + N], equal_nan=True)
                pos += N
    tags = fr.tags_out
    offs = np.array([t.offset for t in tags], dtype=np.int64)
    snr = np.array([t.value[1] for t in tags], dtype=np.float32)
    snr_types = set(type(t.value[1]).__name__ for t in tags)

    dm = demod_mod.demod(fs)
    dm.start_timestamp = 0.0
    dm.tags_in = [_Tag(t.offset, t.key, t.value, t.srcid) for t in tags]
    pos = 0
    with np.errstate(all="ignore"):
        for N in demod_schedule:
            in0 = x[pos:pos + N]
            out0 = np.empty(N, dtype=np.float32)
            dm._nread = dm._nwritten = pos
            dm.work([in0], [out0])
            pos += N
    pdu_off, pdu_bits, pdu_conf, pdu_snr = [], [], [], []
    for port, (meta, vec), conf in dm.msgs:
        assert port == "demodulated"
        # timestamp = 0.0 + off/fs ; recover offset exactly via the tag list order instead
        pdu_bits.append(np.array(vec, dtype=np.uint8))
        pdu_conf.append(conf.astype(np.float32))
        pdu_snr.append(np.float32(meta["snr"]))
        pdu_off.append(int(round(meta["timestamp"] * fs)))
    return dict(
        H=H,
        tag_offsets=offs,
        tag_snr=snr,
        snr_types=snr_types,
        pdu_offsets=np.array(pdu_off, dtype=np.int64),
        pdu_bits=np.array(pdu_bits, dtype=np.uint8).reshape(-1, 112),
        pdu_conf=np.array(pdu_conf, dtype=np.float32).reshape(-1, 112),
        pdu_snr=np.array(pdu_snr, dtype=np.float32),
        final_prev_eob=int(fr.prev_eob_idx),
        final_prev_in0=np.float32(fr.prev_in0),
    )


def load_reference_decoder(msg_filter="All Messages", error_corr="None", print_level="None"):
    """The UNMODIFIED reference decoder (python/adsb/decoder.py) under the stub runtime:
    needs a colorama stand-in, np.NaN on NumPy 2, and message-port registration on the block double."""
    _install_stubs()
    if "colorama" not in sys.modules:
        col = types.ModuleType("colorama")

        class _Blank:
            def __getattr__(self, name):
                return ""
        col.Fore = col.Back = col.Style = _Blank()
        sys.modules["colorama"] = col
    if not hasattr(np, "NaN"):
        np.NaN = np.nan
    _SyncBlock.message_port_register_in = lambda self, name: None
    _SyncBlock.set_msg_handler = lambda self, name, fn: setattr(self, "_handler", fn)
    mod = _load("decoder")
    import logging
    logging.disable(logging.CRITICAL)
    return mod.decoder(msg_filter, error_corr, print_level)
