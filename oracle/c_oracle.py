"""ctypes loader for oracle/adsb_oracle.c (CPU ORACLE, test infrastructure -- see that file's header)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "liboracle.so")
REC = np.dtype([("offset", "<i8"), ("peak", "<f4"), ("median", "<f4"), ("bits", "u1", (14,)), ("flags", "<u2")])


def build(force=False):
    src = os.path.join(HERE, "adsb_oracle.c")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B", "-s"])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.oracle_canonical.restype = ctypes.c_int64
        _lib.oracle_process_iq.restype = ctypes.c_int64
    return _lib


def canonical(x, sps, thr, abs_offset=0, want_cands=False):
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = len(x)
    cap = max(16, n // (60 * sps) + 16)
    c = ctypes
    while True:
        out = np.zeros(cap, dtype=REC)
        cands = np.zeros(cap * 2 if want_cands else 1, dtype=np.int64)
        nc = c.c_int64(0)
        r = lib().oracle_canonical(x.ctypes.data_as(c.c_void_p), c.c_int64(n), c.c_int(sps), c.c_float(thr),
                                   c.c_int64(abs_offset), out.ctypes.data_as(c.c_void_p), c.c_int64(cap),
                                   cands.ctypes.data_as(c.c_void_p) if want_cands else None,
                                   c.c_int64(len(cands) if want_cands else 0), c.byref(nc))
        if r < 0 or (want_cands and nc.value > len(cands)):
            cap = max(cap * 2, -r + 16, nc.value)
            continue
        return (out[:r].copy(), cands[:nc.value].copy()) if want_cands else out[:r].copy()


def process_iq(iq, sps, thr, abs_offset=0, cap=None):
    iq = np.ascontiguousarray(iq, dtype=np.complex64)
    n = len(iq)
    cap = cap or max(16, n // (60 * sps) + 16)
    c = ctypes
    while True:
        out = np.zeros(cap, dtype=REC)
        r = lib().oracle_process_iq(iq.ctypes.data_as(c.c_void_p), c.c_int64(n), c.c_int(sps), c.c_float(thr),
                                    c.c_int64(abs_offset), out.ctypes.data_as(c.c_void_p), c.c_int64(cap))
        if r < 0:
            cap = -r + 16
            continue
        return out[:r].copy()


def parity_flags(recs):
    """The Mode S parity pre-filter bits (oracle_mode_s_parity) for every demodulated record: returns
    (flags_with_parity uint16[n], syndrome uint32[n]) -- what the device writes into adsb_burst.flags."""
    c = ctypes
    L = lib()
    L.oracle_mode_s_parity.restype = c.c_uint32
    flags = recs["flags"].astype(np.uint16).copy()
    syn = np.zeros(len(recs), np.uint32)
    bits = np.ascontiguousarray(recs["bits"])
    f = c.c_uint(0)
    for i in range(len(recs)):
        if flags[i] & 1:
            syn[i] = L.oracle_mode_s_parity(bits[i].ctypes.data_as(c.c_void_p), None, None, c.byref(f))
            flags[i] |= f.value
    return flags, syn


class long_aware_gate:
    """with c_oracle.long_aware_gate(): ... -- canonical()/process_iq() restate ADSB_FLAG_LONG_AWARE_GATE (not the reference)."""
    def __enter__(self):
        lib().oracle_set_long_aware(1)

    def __exit__(self, *a):
        lib().oracle_set_long_aware(0)
