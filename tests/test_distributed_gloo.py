"""The N>1 path on CPU: world_size-2 gloo processes, each owning one overlapped time shard.  The GPU pass
of a rank is stood in for by the emulated device code (tests/simlib.py); everything after it -- the tail
exchange, adsb_shard_fixup, the full-candidate fallback (gr_adsb_amd/sharding.py + the C ABI's host
functions) -- is the product code that bench.py --gpus N runs."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, head, bps, q, exchange="gloo"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import simlib
    from gr_adsb_amd import modulator as M, sharding
    from gr_adsb_amd.frontend import shard_plan
    fs, n = 2e6, 1 << 16
    sps = 2
    iq = M.synth_iq(n, fs, bps, seed=5)          # every rank can rebuild the stream; it only touches its shard
    p = shard_plan(n, world, sps)[rank]

    def shard(head_cands):
        return simlib.sim_shard(0, iq[p["lo"]:p["hi"]], p["lo"], p["own_lo"], p["own_hi"], n, fs, 0.01, head_cands=head_cands)[0]

    # the transport bench.py --gpus N picks: shared-memory mailbox on one node, else the gloo all_gather
    os.environ["ADSB_SHARD_EXCHANGE"] = exchange
    os.environ["LOCAL_WORLD_SIZE"] = str(world)
    ag_int, ag_close = sharding.make_pair_exchange(dist, rank, world)
    assert (ag_int.__self__.__class__.__name__ == "ShmPairExchange") == (exchange == "shm") if hasattr(ag_int, "__self__") else exchange == "gloo"
    for k in range(300):                          # many passes through the ring, values must never mix
        got = ag_int((rank * 1000003 + k, -(k + 1) * (rank + 1)))
        assert got == [(r * 1000003 + k, -(k + 1) * (r + 1)) for r in range(world)]

    def ag_obj(o):
        out = [None] * world
        dist.all_gather_object(out, o)
        return out

    kept = sharding.finish_shard(shard(head), sps, rank, ag_int, lambda: shard(0), ag_obj)
    allk = ag_obj(kept)
    if rank == 0:
        q.put(np.concatenate(allk).tobytes())
    dist.barrier()
    ag_close()
    dist.destroy_process_group()


@pytest.mark.parametrize("head,bps,exchange", [(64, 6000, "gloo"), (1, 40000, "gloo"), (64, 6000, "shm"), (1, 40000, "shm")])
def test_two_rank_shard_stitch_equals_single_call(head, bps, exchange):    # head=1 forces the full-candidate fallback
    import torch.multiprocessing as mp
    from gr_adsb_amd import modulator as M
    from oracle import adsb_oracle as O
    from oracle import c_oracle as C
    import simlib
    simlib.build_sim()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, head, bps, q, exchange)) for r in range(2)]
    for p in procs:
        p.start()
    got = np.frombuffer(q.get(timeout=300), dtype=C.REC)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    iq = M.synth_iq(1 << 16, 2e6, bps, seed=5)
    want = C.canonical(O.mag2(iq), 2, 0.01)
    assert np.array_equal(got["offset"], want["offset"])
    assert np.array_equal(got["bits"], want["bits"])
    assert np.array_equal(got["median"].view(np.uint32), want["median"].view(np.uint32))
    assert np.array_equal(got["flags"] & 1, want["flags"] & 1)
