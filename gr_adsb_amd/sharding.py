"""Host stitch of overlapped time shards: no data-path collective.

Every rank detects and gates its own shard on its GPU as if the shard began a fresh stream, and also gets
the first few centres of the shard ungated (adsb_shard_device, head_cands > 0).  The only thing a shard
needs from its predecessors is their end-of-burst state -- one small all_gather of two int64 per rank -- after which the head of
the shard is re-gated on the host (adsb_shard_fixup) and the result is bit-identical to one canonical call
over the whole stream.  If a shard's head region ends inside an unbroken chain of overlapping bursts (dense
traffic, tiny head) the ranks fall back to exchanging their full candidate lists (adsb_stitch).
"""
import mmap
import os
import time

import numpy as np

from . import _native

HEAD_CANDS = 64


class ShmPairExchange:
    """all_gather of one (int64, int64) pair per rank per pass through a shared-memory mailbox, for ranks
    of ONE node (bench.py --gpus N, one process per GPU): a gloo all_gather of 16 bytes costs 0.5-3 ms on
    loopback TCP -- several times the 0.47 ms GPU pass it accompanies; this costs microseconds.  Ranks on
    different nodes use the torch.distributed (gloo) all_gather instead -- finish_shard takes either.

    Layout (int64 words): ring of DEPTH entries x world slots x [seq, a, b, check].  Pass k: a rank stores
    a, b, check = a ^ b ^ MAGIC*(k+1), then seq = k+1 into entry k % DEPTH, and spins until every slot of the
    entry shows seq == k+1 WITH a matching check word (so a reader can never pair a new seq with stale values,
    whatever the store ordering of the host CPU).  A rank writes pass k+1 only after it has read pass k
    completely, i.e. after every rank wrote pass k, i.e. after every rank finished reading pass k-1: entry
    (k+1) % DEPTH is free for DEPTH >= 3."""
    DEPTH = 4
    WORDS = 4
    MAGIC = 0x9E3779B97F4A7C15

    def __init__(self, path, rank, world, create):
        self.rank, self.world, self.path = rank, world, path
        size = self.DEPTH * world * self.WORDS * 8
        if create:
            fd = os.open(path, os.O_CREAT | os.O_RDWR | os.O_TRUNC, 0o600)
            os.ftruncate(fd, size)
        else:
            fd = os.open(path, os.O_RDWR)
        self._mm = mmap.mmap(fd, size)
        os.close(fd)
        self._a = np.frombuffer(self._mm, dtype=np.uint64).reshape(self.DEPTH, world, self.WORDS)
        self._k = 0

    def all_gather_pair(self, pair, timeout=120.0):
        k = self._k
        m64 = (1 << 64) - 1
        e = self._a[k % self.DEPTH]
        a, b = int(pair[0]) & m64, int(pair[1]) & m64
        salt = (self.MAGIC * (k + 1)) & m64
        e[self.rank, 1] = a
        e[self.rank, 2] = b
        e[self.rank, 3] = a ^ b ^ salt
        e[self.rank, 0] = k + 1                      # publish last
        t0 = None
        while True:
            snap = e.copy()
            if np.all(snap[:, 0] == k + 1) and np.all((snap[:, 1] ^ snap[:, 2] ^ np.uint64(salt)) == snap[:, 3]):
                break
            if t0 is None:
                t0 = time.perf_counter()
            elif time.perf_counter() - t0 > timeout:
                raise TimeoutError("shard tail exchange: a rank did not publish pass %d" % k)
        self._k = k + 1
        s64 = snap[:, 1:3].astype(np.uint64).view(np.int64)
        return [(int(s64[r, 0]), int(s64[r, 1])) for r in range(self.world)]

    def close(self):
        self._a = None
        try:
            self._mm.close()
        except BufferError:
            pass


def make_pair_exchange(dist, rank, world, group=None):
    """The cheapest correct transport for finish_shard's one exchange: a shared-memory mailbox when every
    rank runs on this node (torchrun sets LOCAL_WORLD_SIZE == WORLD_SIZE; ADSB_SHARD_EXCHANGE=gloo overrides),
    else the process group's own all_gather.  group: the (host-side, gloo) process group to set the mailbox up with /
    to gather over -- None = the default group.  Returns (all_gather_pair, close)."""
    import torch
    same_node = os.environ.get("LOCAL_WORLD_SIZE") == str(world) and os.path.isdir("/dev/shm") \
        and os.environ.get("ADSB_SHARD_EXCHANGE", "shm") == "shm"
    if same_node:
        name = [None]
        if rank == 0:
            name[0] = "/dev/shm/adsb_xchg_%d_%d" % (os.getpid(), int(time.time() * 1e6) & 0xFFFFFFFF)
            ShmPairExchange(name[0], 0, world, create=True).close()
        dist.broadcast_object_list(name, src=0, group=group)
        x = ShmPairExchange(name[0], rank, world, create=False)
        dist.barrier(group=group)                    # everyone has it mapped ...
        if rank == 0:
            os.unlink(name[0])                       # ... so the name can go: nothing is left behind on a crash
        return x.all_gather_pair, x.close

    def ag(pair):
        out = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(out, torch.tensor([int(pair[0]), int(pair[1])], dtype=torch.int64), group=group)
        return [(int(t[0]), int(t[1])) for t in out]
    return ag, (lambda: None)


STATS = {"fallbacks": 0}     # finish_shard calls that needed the full-candidate exchange (this process)


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

    Every rank publishes its fresh-state end-of-burst tail and its head-sync offset (shard_head_sync: the last
    head centre that starts an independent chain).  By induction over the ranks: shard 0's incoming state is
    "none"; if shard r's sync offset lies beyond its true incoming eob, the true gate and the fresh-state gate
    agree from that centre on, so the local fix-up succeeds AND the published fresh-state tail is shard r's true
    tail -- which makes the incoming eob computed for shard r+1 the true one.  If the condition fails for any
    shard (its head region ends inside an unbroken chain, or a short shard -- at most HEAD_CANDS centres, all in
    one chain reached by the incoming eob -- whose true tail therefore depends on that eob), every rank sees
    the same failure from the same gathered pairs and all take the full-candidate fallback together, without a
    second collective.  The result equals adsb_stitch over the whole stream in every case
    (tests/test_shard_stitch_property.py)."""
    pairs = all_gather_pair((_native.shard_tail(recs, sps), _native.shard_head_sync(recs, sps)))
    tails = [t for t, _ in pairs]
    ok = all(sync > incoming_eob(tails, r) for r, (_, sync) in enumerate(pairs))
    if ok:
        kept = _native.shard_fixup(recs, sps, incoming_eob(tails, rank), inplace=inplace)
        assert kept is not None
        return kept
    STATS["fallbacks"] += 1
    mine = ungated_fn()
    whole = _native.stitch(np.concatenate(all_gather_obj(mine)), sps)
    return whole[np.isin(whole["offset"], mine["offset"])]


class ShardedRank:
    """One rank's side of the N-process deployment (one process per GPU): its overlapped time shard resident in HBM as a
    torch tensor, `depth` passes in flight; per pass one device pass over the shard, ONE 16-byte exchange of end-of-burst
    state (all_gather_pair: make_pair_exchange) and the host fix-up of the shard's head (finish_shard) -- no data-path
    collective.  step() submits the next pass and collects the oldest once `depth` are in flight; drain() collects the rest
    and returns the number of bursts this rank kept in the last pass; last_kept holds them.
    (One PROCESS with all the GPUs of a node: frontend.MultiDevice / adsb_process_sharded_multi.)"""

    def __init__(self, fe, iq, plan, stream_len, rank, all_gather_pair, all_gather_obj, depth=_native.MAX_IN_FLIGHT, fmt=_native.FMT_FC32):
        self.fe, self.iq, self.plan, self.stream_len, self.sps, self.rank = fe, iq, plan, stream_len, fe.sps, rank
        self.ag_int, self.ag_obj, self.depth, self.fmt = all_gather_pair, all_gather_obj, depth, fmt
        self.pending, self.stash = [], {}      # tickets in flight; results of tickets collected early (fallback path only)
        self.last_kept, self.last_n, self.stitch_s, self.passes = None, 0, 0.0, 0

    def step(self):
        p = self.plan
        self.pending.append(self.fe.submit_shard_tensor(self.iq, p["lo"], p["own_lo"], p["own_hi"], self.stream_len,
                                                        fmt=self.fmt, head_cands=HEAD_CANDS))
        if len(self.pending) == self.depth:
            self._collect(self.pending.pop(0))

    def drain(self):
        while self.pending:
            self._collect(self.pending.pop(0))
        return self.last_n

    def _ungated(self):
        # fallback of finish_shard: a blocking call is only allowed with no ticket pending, so collect (and keep)
        # whatever is still in flight first; every rank takes this path together
        for t in list(self.pending):
            self.stash[t] = self.fe.wait(t)
        p = self.plan
        return self.fe.shard_tensor(self.iq, p["lo"], p["own_lo"], p["own_hi"], self.stream_len, fmt=self.fmt)

    def _collect(self, ticket):
        if ticket in self.stash:
            recs, inplace = self.stash.pop(ticket), False
        else:
            recs, inplace = self.fe.wait(ticket, copy=False), True      # view of the pinned result buffer, fixed up in place
        t_x = time.perf_counter()
        kept = finish_shard(recs, self.sps, self.rank, self.ag_int, self._ungated, self.ag_obj, inplace=inplace)
        self.stitch_s += time.perf_counter() - t_x
        self.passes += 1
        self.last_kept, self.last_n = kept, len(kept)
