"""Latency of the 16-byte shard-tail exchange between the ranks of one node (sharding.ShmPairExchange, the shared-memory
mailbox bench.py --gpus N uses) at 2 / 4 / 8 ranks: every rank calls all_gather_pair K times back to back; the time per
exchange is what one pass's stitch adds on the host.  CPU only.   python tools/mailbox_latency.py [K]"""
import multiprocessing as mp
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(path, rank, world, K, q, ready, go):
    from gr_adsb_amd.sharding import ShmPairExchange
    ex = ShmPairExchange(path, rank, world, create=False)
    ready.put(rank)
    go.wait()
    t0 = time.perf_counter()
    for k in range(K):
        got = ex.all_gather_pair((k * world + rank, -k))
        assert got[(rank + 1) % world] == (k * world + (rank + 1) % world, -k)
    q.put((rank, (time.perf_counter() - t0) / K))
    ex.close()


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    from gr_adsb_amd.sharding import ShmPairExchange
    print("# sharding.ShmPairExchange.all_gather_pair: %d exchanges back to back per rank, host has %d cpus" % (K, os.cpu_count() or 1))
    ctx = mp.get_context("spawn")
    for world in (2, 4, 8):
        path = os.path.join(tempfile.gettempdir(), "adsb_mbx_%d_%d" % (os.getpid(), world))
        ShmPairExchange(path, 0, world, create=True).close()
        q, ready, go = ctx.Queue(), ctx.Queue(), ctx.Event()
        procs = [ctx.Process(target=worker, args=(path, r, world, K, q, ready, go)) for r in range(world)]
        for p in procs:
            p.start()
        for _ in procs:
            ready.get(timeout=120)
        go.set()
        res = sorted(q.get(timeout=600) for _ in procs)
        for p in procs:
            p.join()
        os.unlink(path)
        print("%d ranks: %.1f us per exchange (slowest rank %.1f us)" % (world, 1e6 * sum(t for _, t in res) / world, 1e6 * max(t for _, t in res)))


if __name__ == "__main__":
    main()
