"""The N>1 PRODUCT path on the GPU box (-m gpu): two ranks -- two processes, each with its own context on cuda:0 -- run
the overlapped-time-shard pipeline of bench.py --gpus N: device pass per shard, the shared-memory tail exchange,
adsb_shard_fixup, and the full-candidate fallback.  (The driver's box has one GPU: ADSB_BENCH_ONE_GPU / device 0 for
both ranks exercises the code path, not the scaling.)"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stream(case, n):
    from gr_adsb_amd import modulator as M
    if case == "chain":
        # clean preambles every 100 samples (< the 126-sample gate): one unbroken chain through every shard seam, so a
        # shard's first centres are decided by its predecessor's tail and no head region can re-synchronise
        x = np.full(n, 1e-4, dtype=np.float32)
        for base in range(50, n - 400, 100):
            x[base + np.array([0, 2, 7, 9])] = 1.0
        return np.sqrt(x).astype(np.complex64)
    return M.synth_iq(n, 2e6, 3000, seed=23)


def _worker(rank, world, port, case, n, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_WORLD_SIZE"] = str(world)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gr_adsb_amd import _native, sharding
    from gr_adsb_amd.frontend import shard_plan
    fs, sps = 2e6, 2
    iq = _stream(case, n)
    p = shard_plan(n, world, sps, align=4)[rank]
    ctx = _native.Context(fs, 0.01, device=0)

    def shard(head):
        return ctx.shard_host(_native.FMT_FC32, iq[p["lo"]:p["hi"]], p["lo"], p["own_lo"], p["own_hi"], n, head_cands=head)

    ag_pair, ag_close = sharding.make_pair_exchange(dist, rank, world)
    assert ag_pair.__self__.__class__.__name__ == "ShmPairExchange"

    def ag_obj(o):
        out = [None] * world
        dist.all_gather_object(out, o)
        return out

    before = sharding.STATS["fallbacks"]
    kept = sharding.finish_shard(shard(sharding.HEAD_CANDS), sps, rank, ag_pair, lambda: shard(0), ag_obj)
    allk = ag_obj((kept, sharding.STATS["fallbacks"] - before))
    if rank == 0:
        q.put((np.concatenate([k for k, _ in allk]).tobytes(), [f for _, f in allk]))
    dist.barrier()
    ag_close()
    ctx.close()
    dist.destroy_process_group()


# chain, n = 29800: shard 0 keeps its last preamble (14850), so shard 1's first centre (14950) lies inside that burst's
# gate and every later one inside its predecessor's reach: no head centre can re-synchronise -> full-candidate fallback;
# n = 30000: shard 0's last kept preamble is 14850 again but shard 1 begins at 15050, beyond it: local fix-up suffices
@pytest.mark.parametrize("case,n,world,fallback", [("synthetic", 1 << 18, 2, False), ("chain", 29800, 2, True),
                                                   ("chain", 30000, 2, False), ("chain", 1500, 3, True),
                                                   ("synthetic", 70000, 4, False), ("synthetic", 1 << 19, 8, False)])
def test_ranks_on_gpu_equal_one_canonical_call(case, n, world, fallback):
    import torch.multiprocessing as mp
    from gr_adsb_amd import _native
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    raw, fallbacks = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    got = np.frombuffer(raw, dtype=_native.BURST_DTYPE)
    want = _native.Context(2e6, 0.01).process_iq(_stream(case, n))          # ONE canonical call on the same GPU
    assert len(want) > 5
    assert np.array_equal(got["offset"], want["offset"])
    assert np.array_equal(got["bits"], want["bits"])
    assert np.array_equal(got["median"].view(np.uint32), want["median"].view(np.uint32))
    assert np.array_equal(got["flags"] & 0x1FE1, want["flags"] & 0x1FE1)
    assert len(set(fallbacks)) == 1                                          # every rank took the same decision
    assert (fallbacks[0] > 0) == fallback


def test_bench_eight_ranks_one_gpu_all_seven_seams():
    """BASELINE config 4's shape -- 20 Msps, eight overlapped time shards, host stitch -- as the driver would launch it on an
    8-GPU node, here with all eight ranks on cuda:0: every one of the seven seams must compare identical to a canonical
    call over a window that straddles it."""
    env = dict(os.environ, ADSB_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1",
           "--log2n", "22", "--fs", "20e6", "--bursts", "3000", "--min-time", "0.02", "--no-config4", "--no-one-process"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    mg = d["multi_gpu"]
    assert mg["ranks_seen"] == 8 and mg["exchange_transport"] == "shm mailbox"
    assert len(mg["seam_check"]["per_seam"]) == 7 and mg["seam_check"]["all_identical"]
    assert all(s_["bursts_compared"] > 20 for s_ in mg["seam_check"]["per_seam"])
    assert [r_["rank"] for r_ in mg["per_rank"]] == list(range(8))


def test_bench_plain_launch_starts_its_own_ranks_and_carries_config4():
    """`python bench.py --gpus 2` with NO launcher around it: the bench starts its two ranks itself (never a line that says
    n_gpus: 1) and, with N > 1, the line carries BASELINE config 4 -- one 20 Msps stream tiled as N overlapped shards -- as a
    weak and a strong scaling leg, each with per-rank roofline figures and a seam check.  Both ranks on cuda:0."""
    env = dict(os.environ, ADSB_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--log2n", "23",
           "--min-time", "0.05", "--extra-steps", "4", "--extra-min-time", "0.05", "--no-host-fed-multi", "--hostfed-log2n", "23"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["launched_by"].startswith("bench.py itself")
    # ... and the leg in which rank 0 ALONE drives every device from one host ring (adsb_process_sharded_multi; here two
    # contexts on cuda:0): identical to one blocking call, reported per shard count
    op = d["one_process"]
    assert "error" not in op and op["runs"][0]["contexts"] == 2
    assert all(l_["identical_to_one_blocking_call"] and l_["value"] > 0 for l_ in op["runs"][0]["legs"])
    assert d["config"]["one_process_2ctx_identical"] is True and d["config"]["one_process_2ctx_msps"] > 0
    assert d["config"]["seams_identical"] is True and d["config"]["rank_sync"] == "gloo"
    c4 = d["config4_20msps"]
    assert c4["fs"] == 20e6 and c4["rank_sync"] == "gloo"
    for leg, n_own in (("weak", 1 << 23), ("strong", 1 << 22)):
        L = c4[leg]
        assert L["scaling"] == leg and L["samples_per_gpu_per_step"] == n_own and L["stream_samples_per_step"] == 2 * n_own
        assert L["value"] > 0 and L["seam_check"]["all_identical"] and L["seam_check"]["per_seam"][0]["bursts_compared"] > 20
        assert [r_["rank"] for r_ in L["per_rank"]] == [0, 1]
        assert all(0.0 < r_["roofline_frac"] < 1.0 and r_["kernel_ms"] > 0 for r_ in L["per_rank"])
        assert d["config"]["cfg4_%s_seams_identical" % leg] is True and d["config"]["cfg4_%s_msps" % leg] == L["value"]


def test_bench_refuses_a_world_that_is_not_gpus():
    """Under a launcher whose world size is not --gpus the bench exits non-zero instead of printing a line for the wrong N."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 2 and "WORLD_SIZE=1 but --gpus 2" in r.stderr and not r.stdout.strip()


def test_bench_two_ranks_one_gpu_seam_check():
    """bench.py --gpus 2 under an external launcher (torch.distributed.run), both ranks on cuda:0."""
    env = dict(os.environ, ADSB_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--log2n", "23", "--min-time", "0.05", "--no-config4"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    mg = d["multi_gpu"]
    assert mg["ranks_seen"] == 2 and mg["exchange_transport"] == "shm mailbox"
    assert mg["seam_check"]["all_identical"] and mg["seam_check"]["per_seam"][0]["bursts_compared"] > 100
    assert [r_["rank"] for r_ in mg["per_rank"]] == [0, 1]


@pytest.mark.parametrize("world", [1, 2])
def test_bench_force_dist_initialises_like_a_multi_gpu_launch(world):
    """bench.py --force-dist: the process group is created the way an N-GPU launch creates it (backend cpu:gloo,cuda:nccl),
    the barrier / max-over-ranks try RCCL first and fall back to the gloo side when the communicator cannot be built.  On
    this one-GPU box: one rank -> RCCL works (a communicator of one); two ranks sharing cuda:0 -> RCCL refuses the duplicate
    device and the gloo fall-back carries the run.  Either way the line is complete and, with two ranks, the seam between
    the two shards compares identical to a canonical call."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("ADSB_BENCH_ONE_GPU", None)
    tail = ["--gpus", str(world), "--force-dist", "--steps", "4", "--warmup", "1", "--log2n", "23", "--min-time", "0.05",
            "--no-cpu", "--no-extra", "--no-hostfed", "--no-host-fed-multi", "--no-config4"]
    port = str(_free_port())
    if world == 1:
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py")] + tail
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == world and d["value"] > 0
    assert d["config"]["rank_sync"] in ("rccl", "gloo")
    if world == 1:
        assert d["config"]["rank_sync"] == "rccl", "a communicator of one rank must come up on the GPU box"
    else:
        assert d["multi_gpu"]["seam_check"]["all_identical"]
        assert all("numa_node" in r_ and "local_cpulist" in r_ for r_ in d["multi_gpu"]["per_rank"])
    assert "numa_node" in d["config"]


def test_numa_placement_is_reported_and_allocations_work():
    """adsb_numa_info / adsb_host_alloc_near: the context reports the NUMA node and cpus of its GPU's PCI device (what sysfs
    says; -1 / "" where sysfs is hidden), page-locked memory allocated near the GPU is DMA'd in place by the host-fed path,
    and the results equal a blocking call's."""
    from gr_adsb_amd import _native
    from gr_adsb_amd import modulator as M
    ctx = _native.Context(2e6, 0.01)
    info = ctx.numa_info()
    assert set(info) == {"node", "cpulist", "pci"} and info["node"] >= -1
    path = "/sys/bus/pci/devices/%s/numa_node" % info["pci"]
    if info["pci"] and os.path.exists(path):
        assert info["node"] == int(open(path).read())
        assert info["cpulist"] == open(os.path.dirname(path) + "/local_cpulist").read().strip()
    iq = M.synth_iq(1 << 20, 2e6, 3000, seed=5)
    near = _native.PinnedArray(len(iq), np.complex64, near=ctx)
    near.array[:] = iq
    got = ctx.wait(ctx.submit_format_host(_native.FMT_FC32, near.array))
    want = _native.Context(2e6, 0.01, flags=_native.FLAG_NO_NUMA_BINDING).process_iq(iq)
    assert len(want) > 500 and got.tobytes() == want.tobytes()
    pageable = ctx.wait(ctx.submit_format_host(_native.FMT_FC32, np.tile(iq, 8)))       # 64 MB: the copy threads and the ring
    assert len(pageable) > 8 * 500
    ctx.close()
