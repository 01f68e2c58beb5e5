"""Whole-buffer and sharded front end over the C ABI: the bulk (bench) path.

FrontEnd.process_* == one canonical framer.work() + demod.work() over a buffer (SURVEY.md §8a chunk
semantics); shard_plan/process_shard/stitch tile a long stream across GPUs as independent overlapped
time shards whose candidate lists are stitched on the host -- no collective.
"""
import numpy as np

from . import _native

NOISE_BACK = 100          # framer.py:31 look-back for the SNR median


def shard_plan(stream_len, n_shards, sps, max_run=256, align=4096):
    """Split [0, stream_len) into n_shards owner ranges with the halos each one needs:
    back = 100 (noise median) + 1, forward = max_run (longest pulse followed) + 120*sps (preamble +
    112 bits).  Returns a list of dicts(own_lo, own_hi, lo, hi) in stream offsets; buffer starts are
    aligned down to 4 samples (16-byte loads)."""
    per = -(-stream_len // n_shards)
    per = -(-per // align) * align
    plans = []
    for g in range(n_shards):
        own_lo = min(stream_len, g * per)
        own_hi = min(stream_len, (g + 1) * per)
        lo = max(0, own_lo - (NOISE_BACK + 8 * sps + 4))
        lo -= lo % 4
        hi = min(stream_len, own_hi + max_run + 121 * sps)
        plans.append(dict(own_lo=own_lo, own_hi=own_hi, lo=lo, hi=hi))
    return plans


def _after_torch(ctx, t):
    """The context runs on its own HIP streams: whatever torch still has queued on ITS current stream for this tensor
    (the kernels that produce it) must have finished before our kernels read it.  A device-side dependency -- an event
    recorded on torch's stream that the context's streams wait for (adsb_wait_for_event); the host does not block, so
    a producer still running does not serialise the submit / wait pipeline.
    (adsb_set_stream / FrontEnd.use_torch_stream is the alternative: share torch's stream and skip this.)"""
    import torch
    assert t.is_cuda and t.is_contiguous()
    # a ring of eight re-recordable events per context: the wait is queued with the NEXT call on the context (which follows
    # at once), so an event is long done with by the time its turn comes again -- and none is created per call
    ring = getattr(ctx, "_torch_events", None)
    if ring is None:
        ring = ctx._torch_events = [[torch.cuda.Event() for _ in range(8)], 0]
    ev = ring[0][ring[1] & 7]
    ring[1] += 1
    ev.record(torch.cuda.current_stream(t.device))
    ctx.wait_for_event(ev.cuda_event)


class FrontEnd:
    def __init__(self, fs, threshold, device=0, timing=False, flags=0):
        self.fs = float(fs)
        self.sps = int(fs // 1e6)
        self.ctx = _native.Context(fs, threshold, device=device, flags=int(flags) | (_native.FLAG_TIMING if timing else 0))

    def set_threshold(self, thr):
        self.ctx.set_threshold(thr)

    def use_torch_stream(self, stream):
        self.ctx.set_stream(stream.cuda_stream)

    # -- host buffers -------------------------------------------------------------------------------
    def process_iq(self, iq, abs_offset=0):
        return self.ctx.process_iq(iq, abs_offset)

    def process_mag2(self, x, abs_offset=0):
        return self.ctx.process_mag2(x, abs_offset)

    def process_iq16(self, iq16, abs_offset=0, scale=None):
        """iq16: interleaved int16 I,Q; scale: float32 multiplier per component (default 1/32768)."""
        if scale is not None:
            self.ctx.set_iq16_scale(scale)
        return self.ctx.process_iq16(iq16, abs_offset)

    def process_iq8(self, iq8, abs_offset=0, scale=None):
        """iq8: interleaved 8-bit I,Q: int8 (cs8) or uint8 (cu8, offset binary around 127.5: RTL-SDR), chosen by
        the array's dtype; scale: float32 multiplier per component (defaults 1/128 resp. 1/255 per half LSB)."""
        iq8 = np.asarray(iq8)
        fmt = _native.FMT_CU8 if iq8.dtype == np.uint8 else _native.FMT_SC8
        if scale is not None:
            self.ctx.set_format_scale(fmt, scale)
        return self.ctx.process_format(fmt, iq8, abs_offset)

    def process_format(self, fmt, data, abs_offset=0):
        return self.ctx.process_format(fmt, data, abs_offset)

    # -- torch tensors already in HBM ---------------------------------------------------------------
    def process_format_tensor(self, fmt, t, abs_offset=0, fetch=True):
        """t: contiguous CUDA tensor whose first dimension is the sample count ([n,2] for the IQ formats)."""
        _after_torch(self.ctx, t)
        return self.ctx.process_format_device(fmt, t.data_ptr(), t.shape[0], abs_offset, fetch=fetch)

    def submit_format_tensor(self, fmt, t, abs_offset=0):
        _after_torch(self.ctx, t)
        return self.ctx.submit_format_device(fmt, t.data_ptr(), t.shape[0], abs_offset)

    def process_iq_tensor(self, t, abs_offset=0, fetch=True):
        """t: float32 [n,2] (or complex64 [n]) CUDA tensor, contiguous."""
        _after_torch(self.ctx, t)
        n = t.shape[0]
        return self.ctx.process_iq_device(t.data_ptr(), n, abs_offset, fetch=fetch)

    def process_mag2_tensor(self, t, abs_offset=0, fetch=True):
        _after_torch(self.ctx, t)
        return self.ctx.process_mag2_device(t.data_ptr(), t.shape[0], abs_offset, fetch=fetch)

    def submit_iq16_tensor(self, t, abs_offset=0):
        """t: int16 [n,2] CUDA tensor (interleaved I,Q)."""
        _after_torch(self.ctx, t)
        return self.ctx.submit_iq16_device(t.data_ptr(), t.shape[0], abs_offset)

    def submit_iq_tensor(self, t, abs_offset=0):
        """Queue a canonical pass over t (up to _native.MAX_IN_FLIGHT in flight); returns a ticket for wait()."""
        _after_torch(self.ctx, t)
        return self.ctx.submit_iq_device(t.data_ptr(), t.shape[0], abs_offset)

    def submit_shard_tensor(self, t, origin, own_lo, own_hi, stream_len, fmt=0, head_cands=0):
        _after_torch(self.ctx, t)
        return self.ctx.submit_shard_device(fmt, t.data_ptr(), t.shape[0], origin, own_lo, own_hi, stream_len, head_cands)

    def process_sharded_tensor(self, fmt, t, shards, abs_offset=0, out=None):
        """t as `shards` overlapped time shards on this GPU, pipelined and stitched in C (adsb_process_sharded_device):
        bit-identical to process_format_tensor over the whole tensor."""
        _after_torch(self.ctx, t)
        return self.ctx.process_sharded_device(fmt, t.data_ptr(), t.shape[0], shards, abs_offset, out=out)

    def wait(self, ticket, fetch=True, copy=True):
        return self.ctx.wait(ticket, fetch=fetch, copy=copy)

    def shard_tensor(self, t, origin, own_lo, own_hi, stream_len, fmt=0, head_cands=0):
        _after_torch(self.ctx, t)
        return self.ctx.shard_device(fmt, t.data_ptr(), t.shape[0], origin, own_lo, own_hi, stream_len, head_cands)

    def stitch(self, cand_lists):
        c = np.concatenate([np.asarray(x, dtype=_native.BURST_DTYPE) for x in cand_lists]) if len(cand_lists) else \
            np.zeros(0, dtype=_native.BURST_DTYPE)
        return _native.stitch(c, self.sps)

    def stats(self):
        return self.ctx.stats()

    @staticmethod
    def snr(bursts):
        return _native.snr_db(bursts["peak"], bursts["median"])

    @staticmethod
    def bits(bursts):
        return _native.unpack_bits(bursts["bits"])[:, :112]


class MultiDevice:
    """ONE process, N devices, ONE host ring (BASELINE config 4 / SURVEY.md §8e as the reference's own process model:
    examples/adsb_rx.py:242-268 is one process with one IQ source).  Holds one context per device; process_host() hands a
    host buffer to adsb_process_sharded_multi: the stream is tiled into len(devices) * shards_per_device overlapped time
    shards, each device's feeder thread (inside the library, on the cpus local to its GPU) uploads and runs its shards
    ADSB_MAX_IN_FLIGHT deep, the seams are stitched on the host -- the result equals FrontEnd.process_format over the whole
    buffer bit for bit.  devices: HIP ordinals, e.g. range(torch.cuda.device_count()); an ordinal may repeat (several
    contexts on one GPU: how a one-GPU box exercises this)."""

    def __init__(self, fs, threshold, devices=(0,), flags=0, scales=None):
        self.fs, self.sps = float(fs), int(fs // 1e6)
        self.contexts = [_native.Context(fs, threshold, device=int(d), flags=int(flags)) for d in devices]
        for fmt, sc in (scales or {}).items():
            for cx in self.contexts:
                cx.set_format_scale(fmt, sc)
        self.last_stats = None

    def set_threshold(self, thr):
        for cx in self.contexts:
            cx.set_threshold(thr)

    def pinned(self, n_items, dtype):
        """Page-locked host ring near the FIRST device (one ring feeds every device; on a two-socket node the far socket's
        GPUs read it over the interconnect -- a caller with one ring per socket makes two MultiDevice objects)."""
        return _native.PinnedArray(n_items, dtype, near=self.contexts[0])

    def process_host(self, fmt, data, shards_per_device=1, abs_offset=0, out=None):
        recs, self.last_stats = _native.process_sharded_multi(self.contexts, fmt, data, shards_per_device, abs_offset, out=out,
                                                              want_stats=True)
        return recs

    def close(self):
        for cx in self.contexts:
            cx.close()
        self.contexts = []
