#!/usr/bin/env python
"""bench.py -- IQ Msamples/s through framer+demod on MI355X (BASELINE.json metric).

A step = one pass of the hot path (|IQ|^2 -> preamble detect/tag -> gate -> PPM slice -> burst records
back in pinned host memory) over one batch of synthetic complex64 IQ that is already resident in HBM.
N=1 workload: BASELINE.json configs[1] -- synthetic 2 Msps IQ, ~1k DF17 bursts/s.  With N>1 each rank
owns one overlapped time shard of an N-times longer stream (weak scaling): it detects and gates its own
shard, the ranks exchange only their 8-byte end-of-burst tail state, and each fixes up the head of its
shard on the host (no data-path collective).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--fs 2e6] [--log2n 30] [--bursts 1000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def gen_stream_blocks(n_local, origin, fs, bursts_per_s, seed, device, block=1 << 22):
    """Deterministic stream: block b (block samples) depends only on (seed, b), so overlapping shards
    generated on different ranks hold identical samples where they overlap."""
    import torch
    from gr_adsb_amd import modulator as M
    b0 = origin // block
    b1 = (origin + n_local + block - 1) // block
    parts = []
    for b in range(b0, b1):
        parts.append(M.synth_iq_torch(block, fs, bursts_per_s, seed * 1000003 + b, device))
    full = torch.cat(parts, dim=0) if len(parts) > 1 else parts[0]
    s = origin - b0 * block
    out = full[s:s + n_local].contiguous()
    del full, parts
    return out


def pmc_traffic(fs, log2n, bursts):
    """HBM bytes per k_detect launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json), if this
    exact workload was profiled; PMC collection needs its own rocprofv3 run, it cannot happen inside bench.py."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            for e in json.load(f)["entries"]:
                if e["fs"] == fs and e["log2n"] == log2n and e["bursts"] == bursts:
                    return int(e["traffic_bytes"])
    except (OSError, ValueError, KeyError):
        pass
    return None


def cpu_baseline(iq_host, sps, thr, reps=5):
    """Single-core C port of the reference path (oracle/adsb_oracle.c) on a bounded sample."""
    from oracle import c_oracle as C
    C.lib()
    best = None
    recs = None
    for _ in range(reps):
        t0 = time.perf_counter()
        recs = C.process_iq(iq_host, sps, thr)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return len(iq_host) / best / 1e6, recs


def cpu_baseline_threads(iq_host, sps, thr, threads):
    """The same C port on `threads` host threads, one contiguous shard each (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import c_oracle as C
    n = len(iq_host)
    per = n // threads
    shards = [iq_host[i * per:(i + 1) * per] for i in range(threads)]
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(lambda s_: C.process_iq(s_, sps, thr), shards[:threads]))        # warm
        t0 = time.perf_counter()
        list(ex.map(lambda s_: C.process_iq(s_, sps, thr), shards))
        dt = time.perf_counter() - t0
    return per * threads / dt / 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--fs", type=float, default=2e6)
    ap.add_argument("--log2n", type=int, default=30,
                    help="log2 of complex samples per GPU per step (SURVEY §8d M2: >= 2^28; 2^30 = 8 GB of IQ resident in HBM)")
    ap.add_argument("--bursts", type=float, default=1000.0, help="bursts per second of signal")
    ap.add_argument("--threshold", type=float, default=0.01)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-log2n", type=int, default=28, help="log2 of the CPU-baseline sample (default: one whole step)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--format", choices=["fc32", "sc16", "sc8", "cu8"], default="fc32",
                    help="input sample format: complex64 (BASELINE workload), int16 IQ (4 B/sample) or 8-bit IQ "
                         "(2 B/sample: int8 / RTL-SDR offset binary); integer formats N=1 only")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ADSB_BENCH_ONE_GPU=1: debugging aid -- run the N-rank code path with every rank on cuda:0 (gloo only)
    one_gpu = os.environ.get("ADSB_BENCH_ONE_GPU") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if one_gpu else "cpu:gloo,cuda:nccl", rank=rank, world_size=world)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N"
    n_gpus = world
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from gr_adsb_amd import _native, sharding
    from gr_adsb_amd.frontend import FrontEnd, shard_plan

    fs = args.fs
    sps = int(fs // 1e6)
    n_own = 1 << args.log2n
    stream_len = n_own * n_gpus
    fe = FrontEnd(fs, args.threshold, device=local_rank, timing=True)

    sc16 = args.format != "fc32"            # any integer wire format (name kept from the first one added)
    fmt = {"fc32": _native.FMT_FC32, "sc16": _native.FMT_SC16, "sc8": _native.FMT_SC8, "cu8": _native.FMT_CU8}[args.format]
    assert not (sc16 and n_gpus > 1)
    if n_gpus == 1:
        iq = gen_stream_blocks(n_own, 0, fs, args.bursts, args.seed, dev)
        plan = None
        # quantise the same stream to the integer wire format (full scale 4.0); the kernel converts with the same scale
        if fmt == _native.FMT_SC16:
            fe.ctx.set_format_scale(fmt, 4.0 / 32767.0)
            iq = torch.clamp(torch.round(iq * (32767.0 / 4.0)), -32768, 32767).to(torch.int16).contiguous()
        elif fmt == _native.FMT_SC8:
            fe.ctx.set_format_scale(fmt, 4.0 / 127.0)
            iq = torch.clamp(torch.round(iq * (127.0 / 4.0)), -128, 127).to(torch.int8).contiguous()
        elif fmt == _native.FMT_CU8:
            fe.ctx.set_format_scale(fmt, 4.0 / 255.0)
            iq = torch.clamp(torch.floor(iq * (127.5 / 4.0) + 128.0), 0, 255).to(torch.uint8).contiguous()
    else:
        plan = shard_plan(stream_len, n_gpus, sps, align=n_own)[rank]
        iq = gen_stream_blocks(plan["hi"] - plan["lo"], plan["lo"], fs, args.bursts, args.seed, dev)
    torch.cuda.synchronize()

    DEPTH = int(os.environ.get("ADSB_BENCH_DEPTH", _native.MAX_IN_FLIGHT))     # (tuning aid: needs a library built with that many slots)
    pending = []          # tickets of submitted, not yet collected passes (pipeline of DEPTH passes)

    def step():
        if n_gpus == 1:
            # submit pass i+1 before collecting pass i: the PCIe copy and host work of one pass overlap the
            # kernels of the next; every pass is collected inside the timed region (drain() below)
            pending.append(fe.submit_format_tensor(fmt, iq, 0))
            if len(pending) == DEPTH:
                return fe.wait(pending.pop(0), fetch=False)
            return 0
        # N>1: same DEPTH-deep pipeline; the host stitch of pass i (one 16-byte exchange) overlaps the GPU passes after it
        pending.append(fe.submit_shard_tensor(iq, plan["lo"], plan["own_lo"], plan["own_hi"], stream_len,
                                              head_cands=sharding.HEAD_CANDS))
        if len(pending) == DEPTH:
            return collect_shard(pending.pop(0))
        return 0

    host_t = {"stitch": 0.0}
    stash = {}            # results of tickets that had to be collected early (fallback path only)

    def ungated():
        # fallback of sharding.finish_shard: a blocking call is only allowed with no ticket pending, so
        # collect (and keep) whatever is still in flight first; every rank takes this path together
        for t in list(pending):
            stash[t] = fe.wait(t)
        return fe.shard_tensor(iq, plan["lo"], plan["own_lo"], plan["own_hi"], stream_len)

    def collect_shard(ticket):
        if ticket in stash:
            recs, inplace = stash.pop(ticket), False
        else:
            recs, inplace = fe.wait(ticket, copy=False), True    # view of the pinned result buffer, fixed up in place
        t_x = time.perf_counter()
        kept = sharding.finish_shard(recs, sps, rank, ag_int, ungated, ag_obj, inplace=inplace)
        host_t["stitch"] += time.perf_counter() - t_x
        return len(kept)

    # 16 bytes per rank per pass, host side: shared-memory mailbox on one node, gloo all_gather across nodes
    ag_int, ag_close = sharding.make_pair_exchange(dist, rank, n_gpus) if n_gpus > 1 else (None, lambda: None)

    def ag_obj(o):
        out = [None] * n_gpus
        dist.all_gather_object(out, o)
        return out

    def drain():
        n = 0
        while pending:
            n = fe.wait(pending.pop(0), fetch=False) if n_gpus == 1 else collect_shard(pending.pop(0))
        return n

    # barrier / max-over-ranks go over RCCL (backend "nccl") on the GPUs; if the communicator cannot be set up on
    # this node they fall back to the gloo side of the same process group rather than losing the run
    sync_dev = [dev]
    if n_gpus > 1 and not one_gpu:
        try:
            probe = torch.zeros(1, device=dev)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
        except Exception as e:                                   # noqa: BLE001
            if rank == 0:
                print("bench: RCCL all_reduce failed (%s); synchronising ranks over gloo" % type(e).__name__, file=sys.stderr)
            sync_dev[0] = "cpu"
    elif one_gpu:
        sync_dev[0] = "cpu"

    def sync_all():
        torch.cuda.synchronize()
        if n_gpus > 1:
            dist.all_reduce(torch.zeros(1, device=sync_dev[0]))      # barrier
            torch.cuda.synchronize()

    # the same kernel timed without a neighbour, BEFORE the timed region (it also brings a fresh box's clocks up):
    # blocking passes, nothing else on the GPU -- in the pipelined timed region below k_burst of pass i runs beside
    # k_detect of pass i+1 and takes some of its bandwidth
    iso_ms = None
    if n_gpus == 1:
        for _ in range(3):
            fe.ctx.process_format_device(fmt, iq.data_ptr(), n_own, 0, fetch=False)
        fe.ctx.reset_stats()
        for _ in range(5):
            fe.ctx.process_format_device(fmt, iq.data_ptr(), n_own, 0, fetch=False)
        st_iso = fe.stats()
        iso_ms = st_iso["detect_ms"] / max(1, st_iso["detect_launches"])

    for _ in range(args.warmup):
        step()
    drain()
    fe.ctx.reset_stats()
    sync_all()
    t0 = time.perf_counter()
    n_bursts = 0
    for _ in range(args.steps):
        n_bursts = step()
    n_bursts = drain()
    sync_all()
    elapsed = time.perf_counter() - t0
    if n_gpus > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=sync_dev[0])
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    st = fe.stats()

    result = None
    if rank == 0:
        total_samples = float(n_own) * n_gpus * args.steps
        value = total_samples / elapsed / 1e6
        kern_ms = st["detect_ms"] / max(1, st["detect_launches"])
        alg_bytes = st["detect_bytes"] / max(1, st["detect_launches"])
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        result = {
            "metric": "IQ Msamples/s through framer+demod",
            "value": round(value, 1),
            "unit": "Msamples/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.format == "fc32" else {"sc16": "i16->f32", "sc8": "i8->f32", "cu8": "u8->f32"}[args.format],
            "data": "synthetic",
            "config": {
                "workload": "synthetic %g Msps %s IQ, %g DF17-length bursts/s, AWGN 1e-3, threshold %g; "
                            "2^%d samples per GPU per step resident in HBM; one canonical framer+demod pass"
                            % (fs / 1e6, {"fc32": "complex64", "sc16": "int16", "sc8": "int8", "cu8": "uint8 offset-binary"}[args.format], args.bursts, args.threshold, args.log2n),
                "fs": fs, "samples_per_gpu_per_step": n_own, "bursts_per_step_rank0": int(n_bursts),
                "sharding": "none" if n_gpus == 1 else "%d overlapped time shards, host stitch" % n_gpus,
                "pipeline": "%d passes in flight (submit/wait)" % DEPTH,
                "detect_gap_ms_avg": round(st["detect_gap_ms"] / max(1, st["detect_gaps"]), 4),
                "stitch_ms_per_step_rank0": None if n_gpus == 1 else round(host_t["stitch"] / max(1, args.steps + args.warmup) * 1e3, 4),
                "stitch_fallbacks_rank0": None if n_gpus == 1 else sharding.STATS["fallbacks"],
                "detect_grid": int(st["detect_grid"]), "retries": int(st["retries"]), "longrun_calls": int(st["longrun_calls"]),
            },
            "roofline": {
                "bound": "hbm", "kernel": "k_detect<%s>" % args.format,
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "kernel_ms": round(kern_ms, 4), "algorithmic_bytes_per_launch": int(alg_bytes),
                "kernel_only_msamples_per_s": round(st["detect_samples"] / max(1, st["detect_launches"]) / (kern_ms * 1e-3) / 1e6, 1) if kern_ms > 0 else 0.0,
                "isolated": None if iso_ms is None else {
                    "kernel_ms": round(iso_ms, 4), "achieved": round(alg_bytes / (iso_ms * 1e-3) / 1e9, 1),
                    "frac": round(alg_bytes / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "note": "same kernel, blocking passes, no concurrent k_burst of the previous pass"},
                "traffic": pmc_traffic(fs, args.log2n, args.bursts) if (n_gpus == 1 and not sc16) else None,
                "traffic_source": "profiles/pmc_traffic.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes)",
            },
        }
        if not args.no_cpu and n_gpus == 1 and not sc16:
            n_cpu = min(n_own, 1 << args.cpu_log2n)
            host = iq[:n_cpu].cpu().numpy().view(np.complex64).reshape(-1)
            msps, crecs = cpu_baseline(host, sps, args.threshold)
            # parity in the same run: the GPU path on the same sample must match the C port bit for bit
            grecs = fe.process_iq_tensor(iq[:n_cpu].contiguous(), 0)
            match = (len(grecs) == len(crecs) and np.array_equal(grecs["offset"], crecs["offset"])
                     and np.array_equal(grecs["bits"], crecs["bits"])
                     and np.array_equal(grecs["median"].view(np.uint32), crecs["median"].view(np.uint32))
                     and np.array_equal(grecs["peak"].view(np.uint32), crecs["peak"].view(np.uint32))
                     and np.array_equal(grecs["flags"] & 1, crecs["flags"] & 1))
            result["cpu_baseline"] = {
                "value": round(msps, 1), "unit": "Msamples/s", "cores": 1, "kind": "port",
                "sample": "first 2^%d samples of the same stream, oracle/adsb_oracle.c (scalar C restatement of "
                          "the reference path incl. |IQ|^2), best of 5 (about 6 s of CPU work), host has %d cpus" % (int(np.log2(n_cpu)), os.cpu_count()),
            }
            result["bit_match"] = {"sample_bursts": int(len(crecs)), "identical": bool(match)}
            # the reference-STRUCTURED restatement (vectorised threshold/edges + per-pulse Python
            # loop, oracle/adsb_oracle.py) on one core, next to the scalar C port above
            from oracle import adsb_oracle as O
            n_np = min(n_cpu, 1 << 24)
            t_np = time.perf_counter()
            o_np = O.run_stream(O.mag2(host[:n_np]), fs, args.threshold)
            t_np = time.perf_counter() - t_np
            result["cpu_baseline"]["numpy_port"] = {
                "value": round(n_np / t_np / 1e6, 1), "unit": "Msamples/s", "cores": 1,
                "sample": "first 2^%d samples, oracle/adsb_oracle.py (NumPy restatement with the reference's structure), %d tags"
                          % (int(np.log2(n_np)), len(o_np["tag_offsets"]))}
            nthr = min(64, os.cpu_count() or 1)
            if nthr > 1:
                result["cpu_baseline"]["all_threads"] = {
                    "value": round(cpu_baseline_threads(host, sps, args.threshold, nthr), 1), "unit": "Msamples/s",
                    "threads": nthr, "note": "same C port, one contiguous shard of the sample per host thread"}
        print(json.dumps(result), flush=True)
    if n_gpus > 1:
        sync_all()
        ag_close()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
