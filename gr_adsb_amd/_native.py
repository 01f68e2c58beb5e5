"""ctypes binding of libadsb_hip.so (C ABI: include/adsb_hip.h).

The product path: there is NO CPU implementation behind this module.  If the shared library is
missing, or no HIP device can be opened, the import / constructor raises -- it never falls back.
"""
import ctypes
import errno
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libadsb_hip.so")

BURST_DTYPE = np.dtype([("offset", "<i8"), ("peak", "<f4"), ("median", "<f4"), ("bits", "u1", (14,)), ("flags", "<u2")])
assert BURST_DTYPE.itemsize == 32

FLAG_TIMING = 1
FLAG_LONG_AWARE_GATE = 2  # opt-in: the gate holds 119*sps after a burst whose first data bit is set
FLAG_CONFIDENCE = 4       # opt-in: keep demod.bit_confidence's ratios (demod.py:97-101) for the whole-buffer entry points
FLAG_SINGLE_STREAM = 8    # profiling aid: the sparse tail of a pass on the compute stream instead of beside the next pass
FLAG_FRAMER_SLICES = 32   # adsb_framer_work also returns the 112 bits of tags whose burst ends inside the call's input
FLAG_NO_NUMA_BINDING = 64 # host side not placed on the GPU's NUMA node (default: page-locked buffers and copy threads are)
FLAG_LOW_LATENCY = 16     # the tail of a pass runs beside the next pass's k_detect: results a pass earlier, 1-2 % less throughput
ABI_VERSION = 5
# input sample formats (include/adsb_hip.h ADSB_FMT_*): numpy dtype of the flat host array, items per sample
FMT_FC32, FMT_MAG2, FMT_SC16, FMT_SC8, FMT_CU8 = 0, 1, 2, 3, 4
FMT_LAYOUT = {FMT_FC32: (np.complex64, 1), FMT_MAG2: (np.float32, 1), FMT_SC16: (np.int16, 2), FMT_SC8: (np.int8, 2),
              FMT_CU8: (np.uint8, 2)}
FMT_BYTES = {FMT_FC32: 8, FMT_MAG2: 4, FMT_SC16: 4, FMT_SC8: 2, FMT_CU8: 2}
BURST_DEMOD = 1
BURST_KEPT = 2
BURST_PARITY_OK = 32     # Mode S parity pre-filter bits (include/adsb_hip.h; decoder.py:550-688)
BURST_LONG = 64
BURST_KNOWN_DF = 128
BURST_DF_SHIFT = 8
BURST_LONG_HINT = 0x2000 # records of a long-aware context: this burst holds the gate for 119*sps
MAX_IN_FLIGHT = 3

EXPORTS = [
    "adsb_abi_version", "adsb_create", "adsb_destroy", "adsb_set_threshold", "adsb_set_stream", "adsb_set_copy_threads", "adsb_host_copy", "adsb_wait_for_event", "adsb_reset", "adsb_framer_state",
    "adsb_process_iq", "adsb_process_mag2", "adsb_process_iq_device", "adsb_process_mag2_device", "adsb_last_result",
    "adsb_submit_iq_device", "adsb_submit_mag2_device", "adsb_submit_iq16_device", "adsb_submit_shard_device", "adsb_wait",
    "adsb_set_iq16_scale", "adsb_process_iq16", "adsb_process_iq16_device",
    "adsb_set_format_scale", "adsb_process_format", "adsb_process_format_device", "adsb_submit_format_device",
    "adsb_submit_format_host", "adsb_last_confidence",
    "adsb_framer_work", "adsb_framer_work_passthrough", "adsb_demod_work", "adsb_shard_bounds", "adsb_process_sharded_device", "adsb_shard_device", "adsb_shard_host", "adsb_shard_fixup", "adsb_stitch", "adsb_snr_db", "adsb_mode_s_syndrome", "adsb_plan_chunks", "adsb_get_stats",
    "adsb_process_sharded_multi", "adsb_device_alloc", "adsb_device_free", "adsb_device_upload", "adsb_clear_pending_events",
    "adsb_reset_stats", "adsb_detect_history", "adsb_numa_info", "adsb_host_alloc_near", "adsb_last_error", "adsb_host_alloc", "adsb_host_free", "adsb_host_register", "adsb_host_unregister",
]


class Stats(ctypes.Structure):
    _fields_ = [("detect_launches", ctypes.c_uint64), ("detect_ms", ctypes.c_double), ("detect_samples", ctypes.c_uint64),
                ("detect_bytes", ctypes.c_uint64), ("calls", ctypes.c_uint64), ("retries", ctypes.c_uint64),
                ("longrun_calls", ctypes.c_uint64), ("detect_grid", ctypes.c_uint64), ("blocks_per_cu", ctypes.c_uint64),
                ("detect_gap_ms", ctypes.c_double), ("detect_gaps", ctypes.c_uint64), ("longrun_pulses", ctypes.c_uint64),
                ("poll_fallbacks", ctypes.c_uint64), ("shard_fallbacks", ctypes.c_uint64)]


MULTI_MAX_CTX = 64


class MultiStats(ctypes.Structure):
    _fields_ = [("contexts", ctypes.c_int32), ("shards", ctypes.c_int32), ("fallbacks", ctypes.c_int32), ("pad_", ctypes.c_int32),
                ("wall_s", ctypes.c_double), ("feeder_s", ctypes.c_double * MULTI_MAX_CTX),
                ("device", ctypes.c_int32 * MULTI_MAX_CTX), ("numa_node", ctypes.c_int32 * MULTI_MAX_CTX)]


class AdsbError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libadsb_hip: error %d (%s)%s" % (code, os.strerror(-code) if code < 0 else "?", ": " + msg if msg else ""))
        self.code = code


_lib = None


def load():
    """dlopen the in-tree library; raises if it has not been built (python -m gr_adsb_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (same soname): import it FIRST so that this library binds to the
    # runtime torch uses; loading ours first and torch afterwards puts two runtimes in one process and
    # device discovery then fails in the second one.
    import sys
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    path = os.environ.get("ADSB_HIP_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise ImportError("libadsb_hip.so not built: run `python -m gr_adsb_amd.build` (hipcc, gfx950). "
                          "There is no CPU fallback for the ADS-B hot path.")
    lib = ctypes.CDLL(path)
    c = ctypes
    vp, i64, i32, f32 = c.c_void_p, c.c_int64, c.c_int32, c.c_float
    lib.adsb_abi_version.restype = c.c_int
    lib.adsb_create.argtypes = [c.c_double, f32, c.c_int, c.c_uint32, c.POINTER(vp)]
    lib.adsb_destroy.argtypes = [vp]
    lib.adsb_destroy.restype = None
    lib.adsb_set_threshold.argtypes = [vp, f32]
    lib.adsb_set_stream.argtypes = [vp, vp]
    lib.adsb_reset.argtypes = [vp]
    lib.adsb_wait_for_event.argtypes = [vp, vp]
    lib.adsb_set_copy_threads.argtypes = [vp, i32]
    lib.adsb_host_copy.argtypes = [vp, vp, vp, c.c_size_t]
    lib.adsb_framer_state.argtypes = [vp, c.POINTER(c.c_float), c.POINTER(c.c_int64)]
    lib.adsb_set_iq16_scale.argtypes = [vp, f32]
    lib.adsb_submit_iq16_device.argtypes = [vp, vp, i64, i64, c.POINTER(i32)]
    for name in ("adsb_process_iq", "adsb_process_mag2", "adsb_process_iq_device", "adsb_process_mag2_device",
                 "adsb_process_iq16", "adsb_process_iq16_device"):
        getattr(lib, name).argtypes = [vp, vp, i64, i64, vp, i32, c.POINTER(i32)]
    lib.adsb_set_format_scale.argtypes = [vp, c.c_int, f32]
    lib.adsb_process_format.argtypes = [vp, c.c_int, vp, i64, i64, vp, i32, c.POINTER(i32)]
    lib.adsb_process_format_device.argtypes = [vp, c.c_int, vp, i64, i64, vp, i32, c.POINTER(i32)]
    lib.adsb_submit_format_device.argtypes = [vp, c.c_int, vp, i64, i64, c.POINTER(i32)]
    lib.adsb_submit_format_host.argtypes = [vp, c.c_int, vp, i64, i64, c.POINTER(i32)]
    lib.adsb_last_confidence.argtypes = [vp, c.POINTER(vp), c.POINTER(i32)]
    lib.adsb_last_result.argtypes = [vp, c.POINTER(vp), c.POINTER(i32)]
    lib.adsb_submit_iq_device.argtypes = [vp, vp, i64, i64, c.POINTER(i32)]
    lib.adsb_submit_mag2_device.argtypes = [vp, vp, i64, i64, c.POINTER(i32)]
    lib.adsb_wait.argtypes = [vp, i32, vp, i32, c.POINTER(i32)]
    lib.adsb_submit_shard_device.argtypes = [vp, c.c_int, vp, i64, i64, i64, i64, i64, i32, c.POINTER(i32)]
    lib.adsb_framer_work.argtypes = [vp, vp, i64, i64, i64, vp, i32, c.POINTER(i32)]
    lib.adsb_framer_work_passthrough.argtypes = [vp, vp, i64, i64, i64, vp, vp, i32, c.POINTER(i32)]
    lib.adsb_demod_work.argtypes = [vp, vp, i64, i64, vp, i32, vp, vp, vp]
    lib.adsb_shard_device.argtypes = [vp, c.c_int, vp, i64, i64, i64, i64, i64, i32, vp, i32, c.POINTER(i32)]
    lib.adsb_shard_host.argtypes = [vp, c.c_int, vp, i64, i64, i64, i64, i64, i32, c.c_uint32, vp, i32, c.POINTER(i32)]
    lib.adsb_shard_fixup.argtypes = [vp, i32, c.c_int, i64, c.POINTER(i32)]
    lib.adsb_shard_bounds.argtypes = [i64, i32, i32, c.c_int, i64, c.POINTER(i64), c.POINTER(i64), c.POINTER(i64), c.POINTER(i64)]
    lib.adsb_shard_bounds.restype = c.c_int32
    lib.adsb_process_sharded_device.argtypes = [vp, c.c_int, vp, i64, i64, i32, vp, i32, c.POINTER(i32)]
    lib.adsb_process_sharded_multi.argtypes = [c.POINTER(vp), i32, c.c_int, vp, i64, i64, i32, vp, i32, c.POINTER(i32), c.POINTER(MultiStats)]
    lib.adsb_device_alloc.argtypes = [vp, c.POINTER(vp), c.c_size_t]
    lib.adsb_device_free.argtypes = [vp, vp]
    lib.adsb_device_upload.argtypes = [vp, vp, vp, c.c_size_t]
    lib.adsb_clear_pending_events.argtypes = [vp]
    lib.adsb_stitch.argtypes = [vp, i32, c.c_int, c.POINTER(i32)]
    lib.adsb_snr_db.argtypes = [f32, f32]
    lib.adsb_snr_db.restype = f32
    lib.adsb_mode_s_syndrome.argtypes = [vp, c.POINTER(i32), c.POINTER(i32)]
    lib.adsb_mode_s_syndrome.restype = c.c_uint32
    lib.adsb_plan_chunks.argtypes = [i64, i64, c.POINTER(i64), c.POINTER(i64)]
    lib.adsb_plan_chunks.restype = c.c_int32
    lib.adsb_get_stats.argtypes = [vp, c.POINTER(Stats)]
    lib.adsb_reset_stats.argtypes = [vp]
    lib.adsb_detect_history.argtypes = [vp, vp, i32, c.POINTER(i32)]
    lib.adsb_numa_info.argtypes = [vp, c.POINTER(i32), c.c_char_p, c.c_size_t, c.c_char_p, c.c_size_t]
    lib.adsb_host_alloc_near.argtypes = [vp, c.POINTER(vp), c.c_size_t]
    lib.adsb_host_alloc.argtypes = [c.POINTER(vp), c.c_size_t]
    lib.adsb_host_free.argtypes = [vp]
    lib.adsb_host_register.argtypes = [vp, c.c_size_t]
    lib.adsb_host_unregister.argtypes = [vp]
    lib.adsb_last_error.argtypes = [vp]
    lib.adsb_last_error.restype = c.c_char_p
    _lib = lib
    return lib


class Context:
    """Owns one adsb_ctx (one HIP device + stream).  Thin: every method is one C-ABI call."""

    def __init__(self, fs, threshold, device=0, flags=0):
        self.lib = load()
        self.fs = float(fs)
        self.sps = int(fs // 1e6)
        self._h = ctypes.c_void_p()
        rc = self.lib.adsb_create(float(fs), float(np.float32(threshold)), int(device), int(flags), ctypes.byref(self._h))
        if rc != 0:
            self._h = ctypes.c_void_p()
            raise AdsbError(rc, "adsb_create(fs=%r, device=%r) failed; a HIP device is required" % (fs, device))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.adsb_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise AdsbError(rc, self.lib.adsb_last_error(self._h).decode("utf-8", "replace"))

    def set_threshold_cached(self, thr):
        """set_threshold, skipped while the value is unchanged (the blocks call this once per work())."""
        if thr != getattr(self, "_thr_cached", None):
            self.set_threshold(thr)

    def set_threshold(self, thr):
        self._chk(self.lib.adsb_set_threshold(self._h, float(np.float32(thr))))
        self._thr_cached = thr

    def set_stream(self, stream_handle):
        self._chk(self.lib.adsb_set_stream(self._h, ctypes.c_void_p(int(stream_handle))))

    def reset(self):
        self._chk(self.lib.adsb_reset(self._h))

    def _run(self, fn, ptr, n, abs_offset):
        n_out = ctypes.c_int32(0)
        self._chk(fn(self._h, ctypes.c_void_p(ptr), int(n), int(abs_offset), None, 0, ctypes.byref(n_out)))
        return self.last_result()

    def last_result(self, copy=True):
        """Bursts of the last finished call.  copy=False returns a writable view of the context's pinned
        buffer, valid until the same pipeline slot is used again (MAX_IN_FLIGHT submissions later)."""
        p = ctypes.c_void_p()
        n = ctypes.c_int32(0)
        self._chk(self.lib.adsb_last_result(self._h, ctypes.byref(p), ctypes.byref(n)))
        if n.value == 0:
            return np.zeros(0, dtype=BURST_DTYPE)
        buf = (ctypes.c_char * (n.value * 32)).from_address(p.value)
        v = np.frombuffer(buf, dtype=BURST_DTYPE)
        return v.copy() if copy else v

    def last_confidence(self, copy=True):
        """FLAG_CONFIDENCE contexts: float32 [n,112] ratios bit1_amp / bit0_amp of the last finished call's bursts
        (demod.py:91-101); confidence_db() turns them into demod.bit_confidence."""
        p = ctypes.c_void_p()
        n = ctypes.c_int32(0)
        self._chk(self.lib.adsb_last_confidence(self._h, ctypes.byref(p), ctypes.byref(n)))
        if n.value == 0:
            return np.zeros((0, 112), dtype=np.float32)
        buf = (ctypes.c_char * (n.value * 112 * 4)).from_address(p.value)
        v = np.frombuffer(buf, dtype=np.float32).reshape(n.value, 112)
        return v.copy() if copy else v

    def submit_format_host(self, fmt, data, abs_offset=0):
        """Host-fed pipelined submission (adsb_submit_format_host): data = host array in the format's layout; a
        page-locked one (PinnedArray, torch pin_memory) is DMA'd where it lies and must stay alive until wait()."""
        dt, per = FMT_LAYOUT[int(fmt)]
        data = np.ascontiguousarray(data, dtype=dt)
        t = ctypes.c_int32(-1)
        self._chk(self.lib.adsb_submit_format_host(self._h, int(fmt), ctypes.c_void_p(data.ctypes.data), len(data) // per,
                                                   int(abs_offset), ctypes.byref(t)))
        self._host_keepalive = getattr(self, "_host_keepalive", {})
        self._host_keepalive[t.value] = data
        return t.value

    def process_iq(self, iq, abs_offset=0):
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        return self._run(self.lib.adsb_process_iq, iq.ctypes.data, len(iq), abs_offset)

    def process_mag2(self, x, abs_offset=0):
        x = np.ascontiguousarray(x, dtype=np.float32)
        return self._run(self.lib.adsb_process_mag2, x.ctypes.data, len(x), abs_offset)

    def set_format_scale(self, fmt, scale):
        self._chk(self.lib.adsb_set_format_scale(self._h, int(fmt), float(np.float32(scale))))

    def process_format(self, fmt, data, abs_offset=0):
        """Host array in any ADSB_FMT_* layout (integer IQ: flat interleaved I,Q array of 2n items)."""
        dt, per = FMT_LAYOUT[int(fmt)]
        data = np.ascontiguousarray(data, dtype=dt)
        n_out = ctypes.c_int32(0)
        self._chk(self.lib.adsb_process_format(self._h, int(fmt), ctypes.c_void_p(data.ctypes.data), len(data) // per,
                                               int(abs_offset), None, 0, ctypes.byref(n_out)))
        return self.last_result()

    def process_format_device(self, fmt, dev_ptr, n, abs_offset=0, fetch=True):
        n_out = ctypes.c_int32(0)
        self._chk(self.lib.adsb_process_format_device(self._h, int(fmt), ctypes.c_void_p(int(dev_ptr)), int(n), int(abs_offset),
                                                      None, 0, ctypes.byref(n_out)))
        return self.last_result() if fetch else n_out.value

    def submit_format_device(self, fmt, dev_ptr, n, abs_offset=0):
        t = ctypes.c_int32(-1)
        self._chk(self.lib.adsb_submit_format_device(self._h, int(fmt), ctypes.c_void_p(int(dev_ptr)), int(n), int(abs_offset),
                                                     ctypes.byref(t)))
        return t.value

    def set_iq16_scale(self, scale):
        self._chk(self.lib.adsb_set_iq16_scale(self._h, float(np.float32(scale))))

    def process_iq16(self, iq16, abs_offset=0):
        """iq16: int16 array of interleaved I,Q (2n shorts)."""
        iq16 = np.ascontiguousarray(iq16, dtype=np.int16)
        return self._run(self.lib.adsb_process_iq16, iq16.ctypes.data, len(iq16) // 2, abs_offset)

    def process_iq16_device(self, dev_ptr, n, abs_offset=0, fetch=True):
        n_out = ctypes.c_int32(0)
        self._chk(self.lib.adsb_process_iq16_device(self._h, ctypes.c_void_p(int(dev_ptr)), int(n), int(abs_offset), None, 0,
                                                    ctypes.byref(n_out)))
        return self.last_result() if fetch else n_out.value

    def submit_iq16_device(self, dev_ptr, n, abs_offset=0):
        t = ctypes.c_int32(-1)
        self._chk(self.lib.adsb_submit_iq16_device(self._h, ctypes.c_void_p(int(dev_ptr)), int(n), int(abs_offset), ctypes.byref(t)))
        return t.value

    def process_iq_device(self, dev_ptr, n, abs_offset=0, fetch=True):
        n_out = ctypes.c_int32(0)
        self._chk(self.lib.adsb_process_iq_device(self._h, ctypes.c_void_p(int(dev_ptr)), int(n), int(abs_offset), None, 0,
                                                  ctypes.byref(n_out)))
        return self.last_result() if fetch else n_out.value

    def process_mag2_device(self, dev_ptr, n, abs_offset=0, fetch=True):
        n_out = ctypes.c_int32(0)
        self._chk(self.lib.adsb_process_mag2_device(self._h, ctypes.c_void_p(int(dev_ptr)), int(n), int(abs_offset), None, 0,
                                                    ctypes.byref(n_out)))
        return self.last_result() if fetch else n_out.value

    def submit_iq_device(self, dev_ptr, n, abs_offset=0):
        t = ctypes.c_int32(-1)
        self._chk(self.lib.adsb_submit_iq_device(self._h, ctypes.c_void_p(int(dev_ptr)), int(n), int(abs_offset), ctypes.byref(t)))
        return t.value

    def submit_mag2_device(self, dev_ptr, n, abs_offset=0):
        t = ctypes.c_int32(-1)
        self._chk(self.lib.adsb_submit_mag2_device(self._h, ctypes.c_void_p(int(dev_ptr)), int(n), int(abs_offset), ctypes.byref(t)))
        return t.value

    def submit_shard_device(self, fmt, dev_ptr, n, origin, own_lo, own_hi, stream_len, head_cands=0):
        t = ctypes.c_int32(-1)
        self._chk(self.lib.adsb_submit_shard_device(self._h, int(fmt), ctypes.c_void_p(int(dev_ptr)), int(n), int(origin),
                                                    int(own_lo), int(own_hi), int(stream_len), int(head_cands), ctypes.byref(t)))
        return t.value

    def wait(self, ticket, fetch=True, copy=True):
        n_out = ctypes.c_int32(0)
        try:
            self._chk(self.lib.adsb_wait(self._h, int(ticket), None, 0, ctypes.byref(n_out)))
        finally:
            # the host buffer of a host-fed submission was only kept alive for the upload: let go of it now, also on error
            getattr(self, "_host_keepalive", {}).pop(int(ticket), None)
        return self.last_result(copy=copy) if fetch else n_out.value

    def set_copy_threads(self, threads):
        """Host threads copying pageable host-fed sources into the pinned ring (before the first such submission)."""
        self._chk(self.lib.adsb_set_copy_threads(self._h, int(threads)))

    def host_copy(self, dst, src):
        """dst[:] = src for two contiguous host arrays of equal size, split over the context's copy threads."""
        assert dst.nbytes == src.nbytes and dst.flags.c_contiguous and src.flags.c_contiguous
        self._chk(self.lib.adsb_host_copy(self._h, ctypes.c_void_p(dst.ctypes.data), ctypes.c_void_p(src.ctypes.data), dst.nbytes))

    def wait_for_event(self, hip_event):
        """Everything submitted next runs after this hipEvent_t (raw handle) has completed: device-side ordering."""
        self._chk(self.lib.adsb_wait_for_event(self._h, ctypes.c_void_p(int(hip_event))))

    def clear_pending_events(self):
        self._chk(self.lib.adsb_clear_pending_events(self._h))

    def device_alloc(self, nbytes):
        """Device memory on this context's device (adsb_device_alloc) -> raw pointer; device_free it."""
        d = ctypes.c_void_p()
        self._chk(self.lib.adsb_device_alloc(self._h, ctypes.byref(d), int(nbytes)))
        return int(d.value)

    def device_free(self, dev_ptr):
        self._chk(self.lib.adsb_device_free(self._h, ctypes.c_void_p(int(dev_ptr))))

    def device_upload(self, dev_ptr, data):
        data = np.ascontiguousarray(data)
        self._chk(self.lib.adsb_device_upload(self._h, ctypes.c_void_p(int(dev_ptr)), ctypes.c_void_p(data.ctypes.data), data.nbytes))

    def framer_state(self):
        """(prev_in0, prev_eob_idx): the reference framer's cross-call attributes (framer.py:54,57)."""
        p, e = ctypes.c_float(0), ctypes.c_int64(0)
        self._chk(self.lib.adsb_framer_state(self._h, ctypes.byref(p), ctypes.byref(e)))
        return np.float32(p.value), int(e.value)

    def framer_work(self, in0, N, nitems_written, out0=None):
        """out0 (optional): the block's pass-through output, float32[N], contiguous -- filled by the library beside the device
        pass (adsb_framer_work_passthrough)."""
        if in0.dtype != np.float32 or not in0.flags.c_contiguous:
            in0 = np.ascontiguousarray(in0, dtype=np.float32)
        buf = getattr(self, "_tag_buf", None)
        if buf is None:
            buf = self._tag_buf = np.zeros(512, dtype=BURST_DTYPE)     # tags of one work() call, filled by the library
            self._tag_n = ctypes.c_int32(0)
            self._tag_ptr = ctypes.c_void_p(buf.ctypes.data)
        if out0 is not None:
            assert out0.dtype == np.float32 and out0.flags.c_contiguous and len(out0) == N
            rc = self.lib.adsb_framer_work_passthrough(self._h, ctypes.c_void_p(in0.ctypes.data), len(in0), int(N), int(nitems_written),
                                                       ctypes.c_void_p(out0.ctypes.data), self._tag_ptr, len(buf), ctypes.byref(self._tag_n))
        else:
            rc = self.lib.adsb_framer_work(self._h, ctypes.c_void_p(in0.ctypes.data), len(in0), int(N), int(nitems_written),
                                           self._tag_ptr, len(buf), ctypes.byref(self._tag_n))
        if rc == -28:                                                  # -ENOSPC: more tags than the buffer holds
            return self.last_result()
        self._chk(rc)
        return buf[:self._tag_n.value].copy()

    def demod_work(self, in0, nitems_read, tag_offsets, want_ratio=False):
        in0 = np.ascontiguousarray(in0, dtype=np.float32)
        tags = np.ascontiguousarray(tag_offsets, dtype=np.int64)
        nt = len(tags)
        bits = np.zeros((nt, 112), dtype=np.uint8)
        ok = np.zeros(nt, dtype=np.uint8)
        ratio = np.zeros((nt, 112), dtype=np.float32) if want_ratio else None
        self._chk(self.lib.adsb_demod_work(self._h, ctypes.c_void_p(in0.ctypes.data), len(in0), int(nitems_read),
                                           ctypes.c_void_p(tags.ctypes.data), nt, ctypes.c_void_p(bits.ctypes.data),
                                           ctypes.c_void_p(ok.ctypes.data),
                                           ctypes.c_void_p(ratio.ctypes.data) if want_ratio else None))
        self.last_demod_flags = ok        # ok[t] = BURST_DEMOD | parity pre-filter bits (0 = dropped)
        return bits, ok.astype(bool), ratio

    def process_sharded_device(self, fmt, dev_ptr, n, shards, abs_offset=0, out=None):
        """The resident stream as `shards` overlapped time shards, pipelined and stitched inside the library (one C call:
        adsb_process_sharded_device); bit-identical to process_format_device over the whole buffer.  out: a BURST_DTYPE array
        to receive the records (reused by callers that repeat the call); grown and retried when too small."""
        if out is None:
            out = np.empty(max(4096, int(n) // 4096), dtype=BURST_DTYPE)
        while True:
            n_out = ctypes.c_int32(0)
            rc = self.lib.adsb_process_sharded_device(self._h, int(fmt), ctypes.c_void_p(int(dev_ptr)), int(n), int(abs_offset),
                                                      int(shards), out.ctypes.data_as(ctypes.c_void_p), len(out), ctypes.byref(n_out))
            if rc == -errno.ENOSPC and n_out.value > len(out):
                out = np.empty(n_out.value + n_out.value // 8 + 16, dtype=BURST_DTYPE)
                continue
            self._chk(rc)
            return out[:n_out.value]

    def shard_device(self, fmt, dev_ptr, n, origin, own_lo, own_hi, stream_len, head_cands=0):
        n_out = ctypes.c_int32(0)
        self._chk(self.lib.adsb_shard_device(self._h, int(fmt), ctypes.c_void_p(int(dev_ptr)), int(n), int(origin), int(own_lo),
                                             int(own_hi), int(stream_len), int(head_cands), None, 0, ctypes.byref(n_out)))
        return self.last_result()

    def shard_host(self, fmt, data, origin, own_lo, own_hi, stream_len, head_cands=0, drop_overlong=False):
        """adsb_shard_host: data = host array in the format's layout (see FMT_LAYOUT)."""
        dt, per = FMT_LAYOUT[int(fmt)]
        data = np.ascontiguousarray(data, dtype=dt)
        n_out = ctypes.c_int32(0)
        self._chk(self.lib.adsb_shard_host(self._h, int(fmt), ctypes.c_void_p(data.ctypes.data), len(data) // per, int(origin),
                                           int(own_lo), int(own_hi), int(stream_len), int(head_cands),
                                           SHARD_DROP_OVERLONG if drop_overlong else 0, None, 0, ctypes.byref(n_out)))
        return self.last_result()

    def stats(self):
        s = Stats()
        self._chk(self.lib.adsb_get_stats(self._h, ctypes.byref(s)))
        return {k: getattr(s, k) for k, _ in Stats._fields_}

    def reset_stats(self):
        self._chk(self.lib.adsb_reset_stats(self._h))

    def numa_info(self):
        """{"node": NUMA node of the GPU's PCI device (-1 unknown / not bound), "cpulist": cpus local to it, "pci": address}."""
        node = ctypes.c_int32(-1)
        cl, bdf = ctypes.create_string_buffer(256), ctypes.create_string_buffer(32)
        self._chk(self.lib.adsb_numa_info(self._h, ctypes.byref(node), cl, len(cl), bdf, len(bdf)))
        return {"node": int(node.value), "cpulist": cl.value.decode(), "pci": bdf.value.decode()}

    def detect_history(self):
        """Per-launch k_detect durations (ms) since the last reset_stats, oldest first (FLAG_TIMING contexts)."""
        buf = np.zeros(4096, dtype=np.float32)
        n = ctypes.c_int32(0)
        self._chk(self.lib.adsb_detect_history(self._h, ctypes.c_void_p(buf.ctypes.data), len(buf), ctypes.byref(n)))
        return buf[:n.value].copy()


class PinnedArray:
    """NumPy view of page-locked host memory from adsb_host_alloc (freed when this object dies); with `near=ctx` on the
    NUMA node of that context's GPU (adsb_host_alloc_near)."""

    def __init__(self, n, dtype, near=None):
        self.lib = load()
        self.dtype = np.dtype(dtype)
        self.nbytes = int(n) * self.dtype.itemsize
        self._p = ctypes.c_void_p()
        if near is not None:
            rc = self.lib.adsb_host_alloc_near(near._h, ctypes.byref(self._p), max(1, self.nbytes))
        else:
            rc = self.lib.adsb_host_alloc(ctypes.byref(self._p), max(1, self.nbytes))
        if rc != 0:
            raise AdsbError(rc, "adsb_host_alloc")
        buf = (ctypes.c_char * max(1, self.nbytes)).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(n))

    def __del__(self):
        try:
            if self._p.value:
                self.lib.adsb_host_free(self._p)
                self._p = ctypes.c_void_p()
        except Exception:
            pass


class RegisteredArray:
    """Page-locks an existing NumPy array in place (adsb_host_register) for as long as this object lives, so that
    host-fed submissions DMA it where it lies.  `with RegisteredArray(a): ...` or keep the object around."""

    def __init__(self, array):
        self.lib = load()
        self.array = np.ascontiguousarray(array)
        assert self.array is array or self.array.base is array or np.shares_memory(self.array, array), "array must be contiguous"
        rc = self.lib.adsb_host_register(ctypes.c_void_p(self.array.ctypes.data), self.array.nbytes)
        if rc != 0:
            raise AdsbError(rc, "adsb_host_register")
        self._live = True

    def close(self):
        if getattr(self, "_live", False):
            self.lib.adsb_host_unregister(ctypes.c_void_p(self.array.ctypes.data))
            self._live = False

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def process_sharded_multi(contexts, fmt, data, shards_per_ctx=1, abs_offset=0, out=None, want_stats=False):
    """ONE host buffer, N contexts (normally one per device), one stitched burst list: adsb_process_sharded_multi.  data =
    host array in the format's layout (FMT_LAYOUT; a PinnedArray / RegisteredArray view is DMA'd where it lies).  Returns the
    records -- bit-identical to Context.process_format over the whole buffer -- and, with want_stats, the per-context
    timings of the call as a dict."""
    lib = load()
    dt, per = FMT_LAYOUT[int(fmt)]
    data = np.ascontiguousarray(data, dtype=dt)
    n = len(data) // per
    arr = (ctypes.c_void_p * len(contexts))(*[cx._h for cx in contexts])
    if out is None:
        out = np.empty(max(4096, n // 4096), dtype=BURST_DTYPE)
    st = MultiStats()
    while True:
        n_out = ctypes.c_int32(0)
        rc = lib.adsb_process_sharded_multi(arr, len(contexts), int(fmt), ctypes.c_void_p(data.ctypes.data), n, int(abs_offset),
                                            int(shards_per_ctx), out.ctypes.data_as(ctypes.c_void_p), len(out),
                                            ctypes.byref(n_out), ctypes.byref(st))
        if rc == -errno.ENOSPC and n_out.value > len(out):
            out = np.empty(n_out.value + n_out.value // 8 + 16, dtype=BURST_DTYPE)
            continue
        if rc != 0:
            raise AdsbError(rc, lib.adsb_last_error(contexts[0]._h).decode("utf-8", "replace") if contexts else "")
        recs = out[:n_out.value]
        if not want_stats:
            return recs
        k = st.contexts
        return recs, {"contexts": k, "shards": st.shards, "fallbacks": st.fallbacks, "wall_s": st.wall_s,
                      "feeder_s": list(st.feeder_s[:k]), "device": list(st.device[:k]), "numa_node": list(st.numa_node[:k])}


def stitch(cands, sps):
    """Host stitch of shard candidate lists (already concatenated in stream order)."""
    lib = load()
    cands = np.ascontiguousarray(cands, dtype=BURST_DTYPE).copy()
    nk = ctypes.c_int32(0)
    rc = lib.adsb_stitch(ctypes.c_void_p(cands.ctypes.data), len(cands), int(sps), ctypes.byref(nk))
    if rc != 0:
        raise AdsbError(rc, "adsb_stitch")
    return cands[:nk.value]


BURST_HEAD = 16
SHARD_DROP_OVERLONG = 1
STREAM_UNBOUNDED = 1 << 60     # stream_len of a stream whose end is not known yet
MAX_HEAD = 4096          # upper bound on head_cands callers use
EOB_NONE = -(1 << 60)


def shard_fixup(recs, sps, eob_in, inplace=False):
    """Exact kept list of a gated shard (adsb_shard_device head_cands > 0) given the previous shard's tail.
    Returns None when the head region was too short (-EAGAIN).  inplace=True compacts recs itself (e.g. a
    last_result(copy=False) view of the pinned buffer) instead of a copy."""
    lib = load()
    if not inplace:
        recs = np.ascontiguousarray(recs, dtype=BURST_DTYPE).copy()
    assert recs.dtype == BURST_DTYPE and recs.flags.c_contiguous
    nk = ctypes.c_int32(0)
    rc = lib.adsb_shard_fixup(ctypes.c_void_p(recs.ctypes.data), len(recs), int(sps), int(eob_in), ctypes.byref(nk))
    if rc == -11:
        return None
    if rc != 0:
        raise AdsbError(rc, "adsb_shard_fixup")
    return recs[:nk.value]


SYNC_ALWAYS = (1 << 62)


def gate_window(recs, sps):
    """Samples the re-trigger gate stays closed after each burst: 63*sps (framer.py:165), 119*sps for records a
    long-aware context flagged BURST_LONG_HINT."""
    return np.where((recs["flags"] & BURST_LONG_HINT) != 0, 119, 63).astype(np.int64) * int(sps)


def shard_head_sync(recs, sps):
    """The one number that tells every rank -- for ANY incoming eob -- whether this shard's fresh-state gate
    decisions (and therefore its fresh-state tail, shard_tail) are exact from some head centre on: the largest
    offset of a head-region centre that starts an independent chain, i.e. lies beyond the reach (offset + gate
    window) of every head centre before it (the first centre of a shard always does).  If that offset is beyond
    the incoming eob the centre is accepted by the true gate and by the fresh-state gate alike, both hold the
    same state from there on, adsb_shard_fixup succeeds and the tail published from the fresh-state gate is the
    true one.  SYNC_ALWAYS for a shard without any centre (its tail is EOB_NONE: the incoming state passes
    through).  A shard that lies ENTIRELY in its head region gets no special treatment: its own fix-up could not
    fail, but if all its chain heads are at or before the incoming eob its true tail depends on that eob and the
    fresh-state tail it published would be stale (round-1 bug: such shards returned SYNC_ALWAYS)."""
    if len(recs) == 0:
        return SYNC_ALWAYS
    fl = recs["flags"]
    nh = int(np.count_nonzero(fl[:MAX_HEAD] & BURST_HEAD))
    if nh == 0:
        return EOB_NONE
    off = recs["offset"][:nh]
    reach = np.maximum.accumulate(off + gate_window(recs[:nh], sps))      # how far the centres so far can hold the gate
    idx = np.flatnonzero(off[1:] > reach[:-1])
    return int(off[idx[-1] + 1]) if len(idx) else int(off[0])


def shard_tail(recs, sps):
    """End-of-burst state a gated shard hands to the next one."""
    fl = recs["flags"]
    for i in range(len(recs) - 1, -1, -1):          # behind the head region every record is KEPT: O(1) in practice
        if fl[i] & BURST_KEPT:
            return int(recs["offset"][i]) + (119 if fl[i] & BURST_LONG_HINT else 63) * sps
    return EOB_NONE


def snr_db_c(peak, median):
    return load().adsb_snr_db(float(peak), float(median))


def burst_df(recs):
    """Downlink format of every record (decoder.py:551), from the device's pre-filter bits."""
    return (recs["flags"] >> BURST_DF_SHIFT) & 31


def parity_ok(recs):
    """True where the decoder's check_parity() will pass without an aircraft table (DF 11/17/18/19, syndrome 0)."""
    return (recs["flags"] & BURST_PARITY_OK) != 0


def shard_bounds(stream_len, n_shards, g, sps, align=4096):
    """adsb_shard_bounds: (own_lo, own_hi, lo, hi) of shard g (pure host arithmetic, no device needed)."""
    v = [ctypes.c_int64(0) for _ in range(4)]
    rc = load().adsb_shard_bounds(int(stream_len), int(n_shards), int(g), int(sps), int(align), *[ctypes.byref(x) for x in v])
    if rc:
        raise AdsbError(rc, "adsb_shard_bounds")
    return tuple(x.value for x in v)


def plan_chunks(n_samples, resident_wavefronts):
    """(units, samples_per_chunk) of one call over n_samples (adsb_plan_chunks: pure host arithmetic)."""
    u, t = ctypes.c_int64(), ctypes.c_int64()
    rc = load().adsb_plan_chunks(int(n_samples), int(resident_wavefronts), ctypes.byref(u), ctypes.byref(t))
    if rc != 0:
        raise AdsbError(rc, "adsb_plan_chunks")
    return u.value, t.value


def mode_s_syndrome(bits14):
    """(syndrome, df, nbits) of one 14-byte payload via the C helper: the announced address for the
    address/parity formats (decoder.py:577,647)."""
    b = np.ascontiguousarray(bits14, dtype=np.uint8)
    assert b.size == 14
    df, nb = ctypes.c_int32(), ctypes.c_int32()
    syn = load().adsb_mode_s_syndrome(b.ctypes.data_as(ctypes.c_void_p), ctypes.byref(df), ctypes.byref(nb))
    return int(syn), df.value, nb.value


def unpack_bits(bits14):
    """[n,14] packed bytes -> [n,112] 0/1 uint8 (the u8vector layout of the reference PDU)."""
    return np.unpackbits(np.asarray(bits14, dtype=np.uint8).reshape(-1, 14), axis=1, bitorder="big")


def confidence_db(ratio):
    """demod.py:101: bit_confidence = 10*log10(bit1_amp / bit0_amp), float32 with NumPy itself (like snr_db)."""
    with np.errstate(all="ignore"):
        return (np.float32(10.0) * np.log10(np.asarray(ratio, dtype=np.float32))).astype(np.float32)


def snr_db(peak, median):
    """10*log10(peak/median)+1.6 in float32 with NumPy itself, so the bits equal the reference's
    (framer.py:157 under NumPy-2 promotion) on whatever host this runs on."""
    with np.errstate(all="ignore"):
        p = np.asarray(peak, dtype=np.float32)
        m = np.asarray(median, dtype=np.float32)
        return (np.float32(10.0) * np.log10(p / m) + np.float32(1.6)).astype(np.float32)
