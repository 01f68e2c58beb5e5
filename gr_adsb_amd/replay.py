"""Recorded-IQ replay (SURVEY.md §8f-3 "file-source framing"): run the front end over an IQ recording that
does not fit in HBM -- or simply arrives block by block -- with results bit-identical to ONE canonical
framer+demod call over the whole recording.

The recording is cut into overlapped time shards (the multi-GPU machinery of sharding.py used sequentially on
one GPU): block g owns the pulse rises of [own_lo, own_hi), is uploaded with its halos, detected and gated on
the device as a fresh stream (adsb_shard_device, head_cands > 0), and its head is re-gated on the host with the
end-of-burst state carried over from block g-1 (adsb_shard_fixup).  A head region that ends inside an unbroken
chain of overlapping bursts falls back to the block's ungated candidate list and the plain greedy gate
(framer.py:121-123,165).

    python -m gr_adsb_amd.replay capture.cs8 --format sc8 --fs 2e6 --threshold 0.01 --sqlite pdus.db

Raw files as SDR tools write them: fc32 (GNU Radio file sink / the reference's fc32 stream), sc16 (UHD / SDRplay),
sc8 (hackrf_transfer), cu8 (rtl_sdr).
"""
import argparse
import os

import numpy as np

from . import _native
from .frontend import shard_plan

FORMATS = {"fc32": _native.FMT_FC32, "mag2": _native.FMT_MAG2, "sc16": _native.FMT_SC16, "sc8": _native.FMT_SC8,
           "cu8": _native.FMT_CU8}


def greedy_gate(cands, sps, eob_in):
    """framer.py:121-123,165 over a block's ungated centres with the incoming end-of-burst state (fallback path)."""
    keep = np.zeros(len(cands), dtype=bool)
    eob = eob_in
    win = _native.gate_window(cands, sps)
    for i, p in enumerate(cands["offset"]):
        if p > eob:
            keep[i] = True
            eob = int(p) + int(win[i])
    out = cands[keep].copy()
    out["flags"] = (out["flags"] | _native.BURST_KEPT) & ~np.uint16(_native.BURST_HEAD)
    return out


def replay_blocks(stream_len, sps, block_samples, shard_fn, head_cands=64, head_growth=16):
    """Yield the exact kept bursts block by block.  shard_fn(plan, head_cands) -> the adsb_shard_device result
    for that block (plan: dict(own_lo, own_hi, lo, hi) in stream offsets).  Transport-free: the GPU path passes
    FileReplay._shard, the CPU tests the emulated device code."""
    n_blocks = max(1, -(-stream_len // block_samples))
    eob = _native.EOB_NONE
    for plan in shard_plan(stream_len, n_blocks, sps, align=max(4, block_samples)):
        if plan["own_lo"] >= plan["own_hi"]:
            continue
        kept = None
        for hc in (head_cands, head_growth * head_cands):
            recs = shard_fn(plan, min(hc, _native.MAX_HEAD))
            kept = _native.shard_fixup(recs, sps, eob)
            if kept is not None:
                break
        if kept is None:
            kept = greedy_gate(shard_fn(plan, 0), sps, eob)
        if len(kept):
            eob = int(kept["offset"][-1]) + int(_native.gate_window(kept[-1:], sps)[0])
        yield kept


class FileReplay:
    def __init__(self, path, fmt, fs, threshold, block_samples=1 << 26, device=0, scale=None, long_aware=False):
        self.fmt = FORMATS[fmt] if isinstance(fmt, str) else int(fmt)
        dt, per = _native.FMT_LAYOUT[self.fmt]
        self.items_per_sample = per
        self.data = np.memmap(path, dtype=dt, mode="r") if os.path.getsize(path) else np.zeros(0, dtype=dt)
        self.n = len(self.data) // per
        self.fs, self.sps = float(fs), int(fs // 1e6)
        self.block_samples = int(block_samples)
        self.device = device
        self.ctx = _native.Context(fs, threshold, device=device, flags=_native.FLAG_LONG_AWARE_GATE if long_aware else 0)
        if scale is not None:
            self.ctx.set_format_scale(self.fmt, scale)

    def _shard(self, plan, head_cands):
        per = self.items_per_sample
        hi = plan["hi"]
        while True:
            host = self.data[plan["lo"] * per:hi * per]                   # file block incl. halos -> adsb_shard_host
            try:
                return self.ctx.shard_host(self.fmt, host, plan["lo"], plan["own_lo"], plan["own_hi"], self.n, head_cands)
            except _native.AdsbError as e:
                # -EOVERFLOW: a pulse (carrier, overlapping bursts) runs past the block's forward halo: widen it
                if e.code != -75 or hi >= self.n:
                    raise
                hi = min(self.n, hi + 16 * (hi - plan["own_hi"]))

    def __iter__(self):
        if self.n == 0:
            return iter(())
        return replay_blocks(self.n, self.sps, self.block_samples, self._shard)

    def all(self):
        parts = list(self)
        return np.concatenate(parts) if parts else np.zeros(0, dtype=_native.BURST_DTYPE)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("path")
    ap.add_argument("--format", choices=sorted(FORMATS), default="fc32")
    ap.add_argument("--fs", type=float, default=2e6)
    ap.add_argument("--threshold", type=float, default=0.01)
    ap.add_argument("--scale", type=float, default=None, help="integer formats: float32 multiplier per component")
    ap.add_argument("--block-log2", type=int, default=26, help="log2 of samples per uploaded block")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--sqlite", default=None, help="record PDUs into this SQLite file (table `demodulated`)")
    ap.add_argument("--parity-only", action="store_true", help="keep only PDUs the decoder's check_parity() can pass")
    ap.add_argument("--long-aware", action="store_true", help="length-aware re-trigger gate (not the reference's behaviour)")
    args = ap.parse_args(argv)
    rp = FileReplay(args.path, args.format, args.fs, args.threshold, 1 << args.block_log2, args.device, args.scale,
                    long_aware=args.long_aware)
    sink = None
    if args.sqlite:
        from .pdu_store import PduSqliteSink
        sink = PduSqliteSink(args.sqlite)
    n_tags = n_pdus = n_ok = 0
    for recs in rp:
        n_tags += len(recs)
        dem = (recs["flags"] & _native.BURST_DEMOD) != 0
        ok = _native.parity_ok(recs)
        n_ok += int(ok.sum())
        if args.parity_only:
            known = (recs["flags"] & _native.BURST_KNOWN_DF) != 0
            pi = np.isin(_native.burst_df(recs), (11, 17, 18, 19))
            recs = recs[dem & known & (~pi | ok)]
        n_pdus += int(((recs["flags"] & _native.BURST_DEMOD) != 0).sum())
        if sink is not None:
            sink.write_bursts(recs, args.fs)
    if sink is not None:
        sink.close()
    print("%d samples, %d burst tags, %d PDUs, %d with zero parity syndrome (DF 11/17/18/19)" % (rp.n, n_tags, n_pdus, n_ok))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
