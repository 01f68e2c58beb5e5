"""ctypes wrapper around tests/sim/libadsb_sim.so (the device code run by the CPU SIMT emulator).
Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SIM_DIR = os.path.join(HERE, "sim")
SIM_SO = os.path.join(SIM_DIR, "libadsb_sim.so")


class SimOut(ctypes.Structure):
    _fields_ = [("n_rec", ctypes.c_int), ("n_kept", ctypes.c_int), ("overflow", ctypes.c_int),
                ("long_count", ctypes.c_int), ("flags", ctypes.c_uint), ("lastp", ctypes.c_longlong),
                ("last_kept", ctypes.c_longlong)]


REC_DTYPE = np.dtype([("offset", "<i8"), ("peak", "<f4"), ("median", "<f4"), ("bits", "u1", (14,)), ("flags", "<u2")])
assert REC_DTYPE.itemsize == 32


def build_sim(force=False):
    srcs = [os.path.join(SIM_DIR, "sim_driver.cpp"), os.path.join(SIM_DIR, "hipsim.h"),
            os.path.join(HERE, "..", "gr_adsb_amd", "csrc", "adsb_device.h")]
    if not force and os.path.exists(SIM_SO) and all(os.path.getmtime(SIM_SO) >= os.path.getmtime(s) for s in srcs):
        return SIM_SO
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unknown-pragmas",
                           srcs[0], "-o", SIM_SO])
    return SIM_SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build_sim())
        _lib.sim_run.restype = ctypes.c_int
        _lib.sim_slice.restype = ctypes.c_int
    return _lib


def kernel_geometry():
    """(tile, forward halo, back halo) of k_detect's sliding LDS window, from the compiled device header."""
    t, f, b = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    lib().sim_geometry(ctypes.byref(t), ctypes.byref(f), ctypes.byref(b))
    return t.value, f.value, b.value


def convert8(mode, iq8, scale):
    """body_convert<mode> (k_detect's in-register conversion) of interleaved 8-bit IQ -> float32 |IQ|^2; len(iq8) % 16 == 0."""
    raw = np.ascontiguousarray(iq8).view(np.uint32)
    out = np.empty(2 * len(raw), dtype=np.float32)
    lib().sim_convert8(ctypes.c_int(mode), raw.ctypes.data_as(ctypes.c_void_p), ctypes.c_longlong(len(raw)),
                       ctypes.c_float(scale), out.ctypes.data_as(ctypes.c_void_p))
    return out


def sim_run(mode, data, sps, thr, in0_base, scan_lo, scan_hi, fall_hi, dem_hi, origin=0, prev_in0=0.0,
            end_is_call_end=1, prev_eob_stream=None, gate=True, grid_max=6, rec_cap=0, scale=1.0):
    """mode 0: complex64[n]; 1: float32 |IQ|^2 [n]; 2: int16 / 3: int8 / 4: uint8 interleaved IQ [2n]."""
    if mode >= 2:
        data = np.ascontiguousarray(data, dtype={2: np.int16, 3: np.int8, 4: np.uint8}[mode])
        n = len(data) // 2
    else:
        data = np.ascontiguousarray(data, dtype=np.complex64 if mode == 0 else np.float32)
        n = len(data)
    if prev_eob_stream is None:
        prev_eob_stream = origin + in0_base - 1
    cap = max(16, n // 2 + 16)
    out = np.zeros(cap, dtype=REC_DTYPE)
    so = SimOut()
    c = ctypes
    rc = lib().sim_run(c.c_int(mode), data.ctypes.data_as(c.c_void_p), c.c_longlong(n), c.c_longlong(in0_base),
                       c.c_longlong(scan_lo), c.c_longlong(scan_hi), c.c_longlong(fall_hi), c.c_longlong(dem_hi),
                       c.c_longlong(origin), c.c_float(thr), c.c_float(prev_in0), c.c_int(sps), c.c_int(end_is_call_end),
                       c.c_longlong(prev_eob_stream), c.c_int(1 if gate else 0), c.c_int(0), c.c_int(grid_max), c.c_int(rec_cap),
                       c.c_float(scale),
                       out.ctypes.data_as(c.c_void_p), c.c_int(cap), c.byref(so))
    assert rc == 0
    return out[:so.n_kept].copy(), so


def sim_canonical(mode, data, fs, thr, abs_offset=0, **kw):
    sps = int(fs // 1e6)
    H = 8 * sps
    n = len(data) // 2 if mode >= 2 else len(data)
    return sim_run(mode, data, sps, thr, in0_base=-(H - 1), scan_lo=-(H - 1), scan_hi=n - (H - 1), fall_hi=n - (H - 1),
                   dem_hi=n, origin=abs_offset, **kw)


def _call(fn, args, cap, gate=True):
    out = np.zeros(cap, dtype=REC_DTYPE)
    so = SimOut()
    rc = fn(*args, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(cap), ctypes.byref(so))
    assert rc == 0, rc
    return out[:so.n_kept].copy(), so


class SimFramer:
    """framer.work() through adsb_plan.h + the emulated kernels, state carried like the library does."""

    def __init__(self, fs, thr, grid_max=6):
        self.sps = int(fs // 1e6)
        self.thr = thr
        self.prev_in0 = ctypes.c_float(0.0)
        self.prev_eob = ctypes.c_longlong(-1)
        self.grid_max = grid_max

    def work(self, in0, N, nitems_written):
        c = ctypes
        in0 = np.ascontiguousarray(in0, dtype=np.float32)
        args = (in0.ctypes.data_as(c.c_void_p), c.c_longlong(len(in0)), c.c_longlong(N), c.c_longlong(nitems_written),
                c.c_float(self.thr), c.c_int(self.sps), c.byref(self.prev_in0), c.byref(self.prev_eob), c.c_int(self.grid_max))
        return _call(lib().sim_framer_work, args, max(16, len(in0) // 2 + 16))


def sim_shard(mode, data, origin, own_lo, own_hi, stream_len, fs, thr, head_cands=0, grid_max=6):
    c = ctypes
    data = np.ascontiguousarray(data, dtype=np.complex64 if mode == 0 else np.float32)
    args = (c.c_int(mode), data.ctypes.data_as(c.c_void_p), c.c_longlong(len(data)), c.c_longlong(origin),
            c.c_longlong(own_lo), c.c_longlong(own_hi), c.c_longlong(stream_len), c.c_float(thr), c.c_int(int(fs // 1e6)),
            c.c_int(head_cands), c.c_int(grid_max))
    return _call(lib().sim_shard, args, max(16, len(data) // 2 + 16), gate=False)


def sim_slice(in0, tag_idx, sps, want_ratio=True):
    c = ctypes
    in0 = np.ascontiguousarray(in0, dtype=np.float32)
    tag_idx = np.ascontiguousarray(tag_idx, dtype=np.int64)
    nt = len(tag_idx)
    bits = np.zeros((nt, 14), dtype=np.uint8)
    ok = np.zeros(nt, dtype=np.uint8)
    ratio = np.zeros((nt, 112), dtype=np.float32)
    lib().sim_slice(in0.ctypes.data_as(c.c_void_p), c.c_longlong(len(in0)), tag_idx.ctypes.data_as(c.c_void_p), c.c_int(nt),
                    c.c_int(sps), bits.ctypes.data_as(c.c_void_p), ok.ctypes.data_as(c.c_void_p),
                    ratio.ctypes.data_as(c.c_void_p) if want_ratio else None)
    return bits, ok, ratio


def unpack_bits(bits14):
    return np.unpackbits(np.asarray(bits14, dtype=np.uint8).reshape(-1, 14), axis=1, bitorder="big")


class tail_mode:
    """with simlib.tail_mode(1): ... -- 1 = always the tail kernel chain, 2 = always the one-workgroup fused tail
    (0 = the library's own choice by pass size)."""
    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        lib().sim_set_tail_mode(self.mode)

    def __exit__(self, *a):
        lib().sim_set_tail_mode(0)


class long_aware_gate:
    """with simlib.long_aware_gate(): ... -- the emulated device code runs with ADSB_FLAG_LONG_AWARE_GATE semantics."""
    def __enter__(self):
        lib().sim_set_long_aware(1)

    def __exit__(self, *a):
        lib().sim_set_long_aware(0)
