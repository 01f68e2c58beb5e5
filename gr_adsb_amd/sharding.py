"""Host stitch of overlapped time shards: no data-path collective.

Every rank detects and gates its own shard on its GPU as if the shard began a fresh stream, and also gets
the first few centres of the shard ungated (adsb_shard_device, head_cands > 0).  The only thing a shard
needs from its predecessors is their end-of-burst state -- one small all_gather of two int64 per rank -- after which the head of
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


def finish_shard(recs, sps, rank, all_gather_pair, ungated_fn, all_gather_obj, inplace=False):
    """recs: this rank's adsb_shard_device(head_cands>0) output (inplace=True: compacted where it lies, e.g.
    in the context's pinned buffer).  all_gather_pair((a, b)) -> list of every rank's (a, b) int64 pair -- the
    ONE collective of the common path; ungated_fn() -> this shard's ungated candidates (fallback only);
    all_gather_obj(o) -> list of every rank's object (fallback only).  Returns this rank's exact kept bursts.

    Every rank publishes its end-of-burst tail and its head-sync offset; from those every rank can tell, for
    every rank, whether the local fix-up will succeed, so all ranks agree on the (rare) fallback without a
    second collective."""
    pairs = all_gather_pair((_native.shard_tail(recs, sps), _native.shard_head_sync(recs, sps)))
    tails = [t for t, _ in pairs]
    ok = all(sync > incoming_eob(tails, r) for r, (_, sync) in enumerate(pairs))
    if ok:
        kept = _native.shard_fixup(recs, sps, incoming_eob(tails, rank), inplace=inplace)
        assert kept is not None
        return kept
    mine = ungated_fn()
    whole = _native.stitch(np.concatenate(all_gather_obj(mine)), sps)
    return whole[np.isin(whole["offset"], mine["offset"])]
