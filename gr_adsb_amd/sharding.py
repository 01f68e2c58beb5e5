"""Host stitch of overlapped time shards: no data-path collective.

Every rank detects and gates its own shard on its GPU as if the shard began a fresh stream, and also gets
the first few centres of the shard ungated (adsb_shard_device, head_cands > 0).  The only thing a shard
needs from its predecessors is their end-of-burst state -- one int64 per rank -- after which the head of
the shard is re-gated on the host (adsb_shard_fixup) and the result is bit-identical to one canonical call
over the whole stream.  If a shard's head region ends inside an unbroken chain of overlapping bursts (dense
traffic, tiny head) the ranks fall back to exchanging their full candidate lists (adsb_stitch).
"""
import numpy as np

from . import _native

HEAD_CANDS = 64


def incoming_eob(tails, rank):
    """eob a shard inherits: the tail of the nearest earlier shard that kept any burst."""
    for t in reversed(tails[:rank]):
        if t != _native.EOB_NONE:
            return int(t)
    return _native.EOB_NONE


def finish_shard(recs, sps, rank, all_gather_int, ungated_fn, all_gather_obj, inplace=False):
    """recs: this rank's adsb_shard_device(head_cands>0) output (inplace=True: compacted where it lies, e.g.
    in the context's pinned buffer).  all_gather_int(v) -> list of every rank's
    int; ungated_fn() -> this shard's ungated candidates (only called on the fallback path);
    all_gather_obj(o) -> list of every rank's object.  Returns this rank's exact kept bursts."""
    tails = all_gather_int(_native.shard_tail(recs, sps))
    kept = _native.shard_fixup(recs, sps, incoming_eob(tails, rank), inplace=inplace)
    if not any(all_gather_int(1 if kept is None else 0)):
        return kept
    mine = ungated_fn()
    whole = _native.stitch(np.concatenate(all_gather_obj(mine)), sps)
    return whole[np.isin(whole["offset"], mine["offset"])]
