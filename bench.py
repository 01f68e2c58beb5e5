#!/usr/bin/env python
"""bench.py -- IQ Msamples/s through framer+demod on MI355X (BASELINE.json metric).

A step = one pass of the hot path (|IQ|^2 -> preamble detect/tag -> gate -> PPM slice -> burst records
back in pinned host memory) over one batch of synthetic complex64 IQ that is already resident in HBM.
N=1 workload: BASELINE.json configs[1] -- synthetic 2 Msps IQ, ~1k DF17 bursts/s.  With N>1 each rank
owns one overlapped time shard of an N-times longer stream (weak scaling): it detects and gates its own
shard, the ranks exchange only their 16-byte end-of-burst state, and each fixes up the head of its
shard on the host (no data-path collective); the same line then carries BASELINE configs[3] -- ONE 20 Msps
stream tiled as N overlapped shards -- as a weak and a strong scaling leg (`config4_20msps`), each with per-rank
kernel time / roofline fraction and a seam check.

`--gpus N` is a promise about the line's `n_gpus`: under a launcher (WORLD_SIZE set, as the driver starts it:
python -m torch.distributed.run --nproc-per-node N ...) the world must be N; started plainly with N > 1 the bench
starts its own N ranks through the same launcher; anything else exits non-zero -- never a line that says n_gpus: 1.

The timed region is EXACTLY K steps between barrier + synchronize on both sides, max over ranks; it is
repeated until at least --min-time seconds have been timed (never fewer than 3 repeats) and the MEDIAN
repeat is reported (`timing` holds min / median / max).

One JSON line.  Beside the headline it carries (N=1): `roofline`, `cpu_baseline` (four CPU legs timed on
this box's host cores in the same run), `bit_match`, `host_fed` (PCIe-inclusive rates of the host-fed entry
point next to a plain pinned H2D copy) and `extra_configs` (BASELINE configs 3, 4, 5 at 2^28 samples).

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
# environment variables the bench knows about; any other ADSB_* variable is refused (a stray tuning knob must never
# produce an unlabelled number), the known ones are recorded in config.env
KNOWN_ENV = ("ADSB_BENCH_ONE_GPU", "ADSB_SHARD_EXCHANGE", "ADSB_HIP_LIB", "ADSB_BENCH_SPAWNED")
# reference docs/DF_histogram.txt:4-35: the formats BASELINE config 5 names (DF0/4/5/11/16/17) plus the two other long
# formats the histogram holds more than a handful of (DF20/21), each at its counted weight (DF16: 2 of 5187)
DF_MIX = ((11, 0, 4, 17, 20, 5, 21, 16), (2335, 1395, 732, 582, 61, 34, 32, 2))


def gen_stream_blocks(n_local, origin, fs, bursts_per_s, seed, device, block=1 << 22, **synth):
    """Deterministic stream: block b (block samples) depends only on (seed, b), so overlapping shards
    generated on different ranks hold identical samples where they overlap."""
    import torch
    from gr_adsb_amd import modulator as M
    b0 = origin // block
    b1 = (origin + n_local + block - 1) // block
    parts = []
    for b in range(b0, b1):
        parts.append(M.synth_iq_torch(block, fs, bursts_per_s, seed * 1000003 + b, device, **synth))
    full = torch.cat(parts, dim=0) if len(parts) > 1 else parts[0]
    s = origin - b0 * block
    out = full[s:s + n_local].contiguous()
    del full, parts
    return out


def pmc_traffic(fmt_name, fs, bursts, mixed, log2n):
    """HBM bytes per k_detect launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json), if this exact
    workload was profiled; PMC collection needs its own rocprofv3 run, it cannot happen inside bench.py."""
    key = "%s|fs=%g|bursts=%g|mixed=%d|log2n=%d" % (fmt_name, fs, bursts, int(bool(mixed)), log2n)
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            e = json.load(f)["entries"].get(key)
        if e:
            return int(e["traffic_bytes"]), e.get("source")
    except (OSError, ValueError, KeyError, AttributeError):
        pass
    return None, None


# ---- CPU baselines (oracle/: test infrastructure, used here only as the thing timed beside the GPU) -------------
def cpu_c_port(iq_host, sps, thr, reps=3):
    """Single-core scalar C port of the reference path (oracle/adsb_oracle.c) on a bounded sample."""
    from oracle import c_oracle as C
    C.lib()
    best, recs = None, None
    for _ in range(reps):
        t0 = time.perf_counter()
        recs = C.process_iq(iq_host, sps, thr)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return len(iq_host) / best / 1e6, recs


def cpu_c_port_threads(iq_host, sps, thr, threads):
    """The same C port on `threads` host threads, one contiguous shard each (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import c_oracle as C
    per = len(iq_host) // threads
    shards = [iq_host[i * per:(i + 1) * per] for i in range(threads)]
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(lambda s_: C.process_iq(s_, sps, thr), shards))        # warm
        t0 = time.perf_counter()
        list(ex.map(lambda s_: C.process_iq(s_, sps, thr), shards))
        dt = time.perf_counter() - t0
    return per * threads / dt / 1e6


def cpu_reference_structured(x, fs, thr, procs):
    """oracle/ref_structured.py -- vectorised front end + per-pulse Python loop + np.median, the reference's own cost
    structure -- on one core, and on `procs` processes as overlapped time shards + stitch."""
    import multiprocessing as mp
    from oracle import ref_structured as RS
    n1 = min(len(x), 1 << 23)
    stats = {}
    t0 = time.perf_counter()
    one = RS.run_stream(x[:n1], fs, thr, stats=stats)
    t1 = time.perf_counter() - t0
    out = {"one_core": {"value": round(n1 / t1 / 1e6, 2), "unit": "Msamples/s", "cores": 1,
                        "sample": "first 2^%d samples, %d pulses of which %d evaluated, %d tags" % (
                            int(np.log2(n1)), stats.get("pulses", 0), stats.get("evaluated", 0), len(one["tag_offsets"]))}}
    if procs > 1:
        from concurrent.futures import ProcessPoolExecutor
        try:
            # spawn, never fork: the parent holds a HIP context.  Every wait is bounded: a worker that cannot start must
            # cost the bench one record, not the run.
            with ProcessPoolExecutor(max_workers=procs, mp_context=mp.get_context("spawn")) as ex:
                list(ex.map(abs, range(procs), timeout=180))   # workers up (interpreter + numpy import) before the clock starts
                t0 = time.perf_counter()
                sh = RS.run_sharded(x, fs, thr, procs, pool_map=lambda f, jobs: ex.map(f, jobs, timeout=600))
                tp = time.perf_counter() - t0
            cut = n1 - 200 * int(fs // 1e6)
            ok = bool(np.array_equal(sh["tag_offsets"][sh["tag_offsets"] < cut], one["tag_offsets"][one["tag_offsets"] < cut]))
            out["all_cores"] = {"value": round(len(x) / tp / 1e6, 2), "unit": "Msamples/s", "processes": procs,
                                "sample": "first 2^%d samples as %d overlapped time shards (one process each, warm-up "
                                          "%d samples, host stitch), %d tags; shard pickling included" % (
                                              int(np.log2(len(x))), procs, RS.WARM, len(sh["tag_offsets"])),
                                "agrees_with_one_core_leg": ok, "serial_fallback": bool(sh["fallback"])}
        except Exception as e:                                   # noqa: BLE001
            out["all_cores"] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
    return out


def recs_match(a, b):
    return bool(len(a) == len(b) and np.array_equal(a["offset"], b["offset"]) and np.array_equal(a["bits"], b["bits"])
                and np.array_equal(a["median"].view(np.uint32), b["median"].view(np.uint32))
                and np.array_equal(a["peak"].view(np.uint32), b["peak"].view(np.uint32))
                and np.array_equal(a["flags"] & 1, b["flags"] & 1))


def timed_repeats(step, drain, sync_all, reduce_max, steps, min_time, max_repeats=400):
    """EXACTLY `steps` steps per repeat, bracketed by sync_all (barrier + synchronize); repeats until min_time seconds
    have been timed.  Every rank sees the same reduced times, so every rank stops at the same repeat."""
    times, nb = [], 0
    while True:
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        nb = drain()
        sync_all()
        times.append(reduce_max(time.perf_counter() - t0))
        if (len(times) >= 3 and sum(times) >= min_time) or len(times) >= max_repeats:
            return times, nb


def roofline_of(st, fmt_name, iso_ms=None):
    kern_ms = st["detect_ms"] / max(1, st["detect_launches"])
    alg = st["detect_bytes"] / max(1, st["detect_launches"])
    ach = alg / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    r = {"bound": "hbm", "kernel": "k_detect<%s>" % fmt_name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(ach / HBM_PEAK_GBS, 4), "kernel_ms": round(kern_ms, 4), "algorithmic_bytes_per_launch": int(alg),
         "launches_timed": int(st["detect_launches"])}
    if iso_ms:
        r["isolated"] = {"kernel_ms": round(iso_ms, 4), "achieved": round(alg / (iso_ms * 1e-3) / 1e9, 1),
                         "frac": round(alg / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "note": "same kernel, blocking passes, nothing else on the GPU"}
    return r


def run_single_gpu_config(fe, fmt, iq, n, steps, warmup, min_time, depth):
    """Pipelined canonical passes over a resident buffer; returns (median ms/step, times, stats, bursts, iso_ms)."""
    for _ in range(2):
        fe.ctx.process_format_device(fmt, iq.data_ptr(), n, 0, fetch=False)
    import torch
    pending = []

    def step():
        pending.append(fe.submit_format_tensor(fmt, iq, 0))
        if len(pending) == depth:
            fe.wait(pending.pop(0), fetch=False)

    def drain():
        nb = 0
        while pending:
            nb = fe.wait(pending.pop(0), fetch=False)
        return nb

    for _ in range(warmup):
        step()
    drain()
    fe.ctx.reset_stats()
    times, nb = timed_repeats(step, drain, torch.cuda.synchronize, lambda t: t, steps, min_time)
    st = fe.stats()
    return float(np.median(times)) / steps * 1e3, times, st, nb, isolated_kernel_ms(fe, fmt, iq, n)


def isolated_kernel_ms(fe, fmt, iq, n, launches=8):
    """k_detect alone: blocking passes, nothing else on the GPU, HIP events around the kernel.  Measured AFTER the timed
    region, on a GPU whose clocks are up (a fresh box ramps its shader clock by ~10 % over the first dozens of launches:
    profiles/r04_cu_probe.txt; measured in front of the timed region this figure was 2-5 % pessimistic)."""
    fe.ctx.reset_stats()
    for _ in range(launches):
        fe.ctx.process_format_device(fmt, iq.data_ptr(), n, 0, fetch=False)
    s0 = fe.stats()
    return s0["detect_ms"] / max(1, s0["detect_launches"])


def sharded_on_one_gpu(fe, iq, n, sps, shards, depth):
    """BASELINE config 4's decomposition on ONE GPU: the resident stream as `shards` overlapped time shards, each
    detected and gated on the device as a fresh stream, heads re-gated on the host with the carried end-of-burst
    state (the multi-GPU stitch run sequentially).  Since round 5 the whole loop -- plan, submit three deep, wait, fix-up,
    fallback -- is ONE C call (adsb_process_sharded_device): the Python loop it replaces cost ~28 us per shard on top of the
    52 us a 2^25-sample pass takes (tools/pass_cost.py).  Returns a callable running one step -> the stream's records."""
    from gr_adsb_amd import _native
    out = np.empty(max(1 << 16, n // 2048), dtype=_native.BURST_DTYPE)

    def one_step(collect):
        recs = fe.process_sharded_tensor(_native.FMT_FC32, iq, shards, out=out)
        return recs.copy() if collect else None

    return one_step


def extra_configs(args, dev, depth):
    """BASELINE configs 3, 4 and 5 after the headline leg, each on 2^28 resident samples with its own parity check."""
    import torch
    from gr_adsb_amd import _native
    from gr_adsb_amd.frontend import FrontEnd
    from oracle import c_oracle as C
    out = []
    log2n = args.extra_log2n
    n = 1 << log2n
    cpu_n = 1 << 26
    specs = [
        dict(name="config3_8msps_dense", fs=8e6, bursts=6000.0, seed=2, synth={},
             workload="synthetic 8 Msps complex64 IQ (4x oversampled), 6000 DF17-length bursts/s (72 %% duty, overlapping), AWGN 1e-3"),
        dict(name="config4_20msps_8_shards_on_one_gpu", fs=20e6, bursts=1000.0, seed=3, synth={}, shards=8,
             workload="synthetic 20 Msps complex64 IQ, 1000 bursts/s, processed as 8 overlapped time shards on this GPU, host stitch"),
        dict(name="config5_mixed_df_low_snr", fs=2e6, bursts=1000.0, seed=4,
             synth=dict(noise_power=2e-3, df_choices=DF_MIX[0], df_weights=DF_MIX[1], snr_db_range=(3.0, 25.0)),
             workload="synthetic 2 Msps complex64 IQ, 1000 bursts/s mixed DF 0/4/5/11 (56 bit) and 16/17/20/21 (112 bit) in the "
                      "proportions of docs/DF_histogram.txt, per-burst SNR 3-25 dB over noise power 2e-3"),
    ]
    for sp in specs:
        fs, sps = sp["fs"], int(sp["fs"] // 1e6)
        iq = gen_stream_blocks(n, 0, fs, sp["bursts"], sp["seed"], dev, **sp["synth"])
        torch.cuda.synchronize()
        fe = FrontEnd(fs, args.threshold, device=dev.index, timing=True)
        ms, times, st, nb, iso_ms = run_single_gpu_config(fe, _native.FMT_FC32, iq, n, args.extra_steps, 3, args.extra_min_time, depth)
        rec = {"name": sp["name"], "workload": sp["workload"] + "; 2^%d samples per step resident in HBM" % log2n,
               "fs": fs, "value": round(n / ms / 1e3, 1), "unit": "Msamples/s", "ms_per_step": round(ms, 4),
               "steps": args.extra_steps, "repeats": len(times), "bursts_per_step": int(nb),
               "longrun_calls": int(st["longrun_calls"]), "retries": int(st["retries"]),
               "long_pulses_per_step": round(st["longrun_pulses"] / max(1, st["calls"]), 2),
               "roofline": roofline_of(st, "fc32", iso_ms),
               "product_default": untimed_context_ms(args, dev.index, _native.FMT_FC32, iq, n, depth, torch.cuda.synchronize, fs=fs,
                                                     steps=args.extra_steps)}
        tr, tr_src = pmc_traffic("fc32", fs, sp["bursts"], bool(sp["synth"]), log2n)
        rec["roofline"]["traffic"] = tr
        rec["roofline"]["traffic_source"] = tr_src or "none for this workload (see profiles/)"
        # parity: the GPU pass over the first 2^26 samples against the scalar C port of the reference path
        host = iq[:cpu_n].cpu().numpy().view(np.complex64).reshape(-1)
        t0 = time.perf_counter()
        crecs = C.process_iq(host, sps, args.threshold)
        tc = time.perf_counter() - t0
        grecs = fe.process_iq_tensor(iq[:cpu_n].contiguous(), 0)
        rec["bit_match"] = {"sample_bursts": int(len(crecs)), "identical": recs_match(grecs, crecs),
                            "sample": "first 2^26 samples vs oracle/adsb_oracle.c"}
        rec["cpu_c_port_msamples_per_s"] = round(cpu_n / tc / 1e6, 1)
        if sp.get("shards"):
            # (a context of the product's default kind: no ADSB_FLAG_TIMING, the shards' passes free to overlap -- this leg
            # quotes a wall time, not a kernel duration)
            fe_t, fe = fe, FrontEnd(fs, args.threshold, device=dev.index, timing=False)
            whole = fe.process_iq_tensor(iq, 0)
            one_step = sharded_on_one_gpu(fe, iq, n, sps, sp["shards"], depth)
            stitched = one_step(True)
            for _ in range(2):
                one_step(False)
            torch.cuda.synchronize()
            tt = []
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(4):
                    one_step(False)
                torch.cuda.synchronize()
                tt.append((time.perf_counter() - t0) / 4)
            msh = float(np.median(tt)) * 1e3
            rec["sharded"] = {"shards": sp["shards"], "value": round(n / msh / 1e3, 1), "unit": "Msamples/s",
                              "ms_per_step": round(msh, 4),
                              "note": "one step = adsb_process_sharded_device: all %d shards submitted %d deep + host fix-up of every head, in C" % (sp["shards"], depth),
                              "shard_fallbacks": int(fe.stats()["shard_fallbacks"]),
                              "stitched_equals_single_call": recs_match(stitched, whole) and bool(
                                  np.array_equal(stitched["flags"] & 0x1FE1, whole["flags"] & 0x1FE1)),
                              "bursts": int(len(whole))}
            fe.ctx.close()
            fe = fe_t
        out.append(rec)
        del iq, fe
        torch.cuda.empty_cache()
    return out


def format_legs(args, dev, depth):
    """The other input formats on the headline signal (BASELINE config 2: 2 Msps, ~1 k DF17 bursts/s) at the headline's size:
    float32 |IQ|^2 (the framer's literal input), int16 IQ, int8 IQ, RTL-SDR uint8 IQ -- each with its own roofline record
    (algorithmic bytes = samples x bytes per sample of THAT format) and its own parity check against the C oracle fed
    with the oracle's exact conversion of the same bytes."""
    import torch
    from gr_adsb_amd import _native
    from gr_adsb_amd.frontend import FrontEnd
    from oracle import adsb_oracle as O
    from oracle import c_oracle as C
    log2n = args.log2n
    n, cpu_n = 1 << log2n, 1 << 25
    fs, sps = 2e6, 2
    base = gen_stream_blocks(n, 0, fs, 1000.0, args.seed, dev)
    torch.cuda.synchronize()
    out = []
    # int8 and uint8 twice each: a power-of-two scale (2^-5 / 2^-6: the usual int8 convention / the RTL-SDR one) selects the
    # dot-product instance of k_detect for the format, any other scale (4/127, 4/255: sc8g / cu8g) the generic instance.
    # int8: scale 2^-5 selects the dot-product instance of k_detect (k_detect<sc8, power-of-two scale>: v_dot4c_i32_i8,
    # two tiles in flight), scale 4/127 the generic int8 instance (convert, multiply: what any other scale runs)
    for name, fmt, scale in (("mag2", _native.FMT_MAG2, None), ("sc16", _native.FMT_SC16, 4.0 / 32767.0),
                             ("sc8", _native.FMT_SC8, 4.0 / 128.0), ("sc8g", _native.FMT_SC8, 4.0 / 127.0),
                             ("cu8", _native.FMT_CU8, 2.0 ** -6), ("cu8g", _native.FMT_CU8, 4.0 / 255.0)):
        fe = FrontEnd(fs, args.threshold, device=dev.index, timing=True)
        q = quantise_for(fmt, base, fe, scale=scale)
        torch.cuda.synchronize()
        ms, times, st, nb, iso_ms = run_single_gpu_config(fe, fmt, q, n, args.extra_steps, 3, args.extra_min_time, depth)
        rec = {"name": "format_" + name, "format": name, "bytes_per_sample": _native.FMT_BYTES[fmt],
               "workload": "BASELINE config 2's signal as %s; 2^%d samples per step resident in HBM" % (
                   {"mag2": "float32 |IQ|^2", "sc16": "int16 IQ", "sc8": "int8 IQ (scale 2^-5: dot-product instance)",
                    "sc8g": "int8 IQ (scale 4/127: generic int8 instance)", "cu8": "uint8 offset-binary IQ (scale 2^-6, the RTL-SDR convention (u8 - 127.5) / 32 at this full scale: dot-product instance)",
                    "cu8g": "uint8 offset-binary IQ (scale 4/255: generic uint8 instance)"}[name], log2n),
               "kernel_instance": {"sc8": "k_detect<int8, power-of-two scale>", "sc8g": "k_detect<int8, any scale>",
                                   "cu8": "k_detect<uint8, power-of-two scale>", "cu8g": "k_detect<uint8, any scale>"}.get(name, "k_detect<%s>" % name),
               "value": round(n / ms / 1e3, 1), "unit": "Msamples/s", "ms_per_step": round(ms, 4), "bursts_per_step": int(nb),
               "roofline": roofline_of(st, name, iso_ms),
               "product_default": untimed_context_ms(args, dev.index, fmt, q, n, depth, torch.cuda.synchronize, fs=fs, scale=scale,
                                                     steps=args.extra_steps)}
        tr, tr_src = pmc_traffic({"sc8g": "sc8", "cu8g": "cu8"}.get(name, name), fs, 1000.0, False, log2n)
        rec["roofline"]["traffic"] = tr
        rec["roofline"]["traffic_source"] = tr_src or "none for this workload (see profiles/)"
        host = q[:cpu_n].cpu().numpy()
        if name == "mag2":
            x = host.reshape(-1)
        elif name == "sc16":
            x = O.mag2_iq16(host.reshape(-1), scale)
        else:
            x = O.mag2_iq8(host.reshape(-1), float(np.float32(scale)), name in ("cu8", "cu8g"))
        crecs = C.canonical(x, sps, np.float32(args.threshold))
        grecs = fe.ctx.process_format_device(fmt, q.data_ptr(), cpu_n)
        rec["bit_match"] = {"sample_bursts": int(len(crecs)), "identical": recs_match(grecs, crecs),
                            "sample": "first 2^25 samples vs oracle/adsb_oracle.c on the oracle's conversion of the same bytes"}
        out.append(rec)
        del q, fe
        torch.cuda.empty_cache()
    return out


def quantise_for(fmt, iq, fe, scale=None):
    """The float [n,2] stream quantised to an integer wire format (full scale 4.0); sets the context's matching scale."""
    import torch
    from gr_adsb_amd import _native
    if fmt == _native.FMT_SC8 and scale is not None and scale != 4.0 / 128.0:
        # any other int8 scale (here 4/127): the generic int8 instance of k_detect
        fe.ctx.set_format_scale(fmt, scale)
        return torch.clamp(torch.round(iq * (1.0 / scale)), -128, 127).to(torch.int8).contiguous()
    if fmt == _native.FMT_SC16:
        fe.ctx.set_format_scale(fmt, 4.0 / 32767.0)
        return torch.clamp(torch.round(iq * (32767.0 / 4.0)), -32768, 32767).to(torch.int16).contiguous()
    if fmt == _native.FMT_SC8:
        # the usual int8 convention, component = i8 * 2^-k (here full scale 4.0: 2^-5): a power-of-two scale also selects
        # the library's dot-product instance of k_detect (adsb_hip.hip: launch_detect)
        fe.ctx.set_format_scale(fmt, 4.0 / 128.0)
        return torch.clamp(torch.round(iq * (128.0 / 4.0)), -128, 127).to(torch.int8).contiguous()
    if fmt == _native.FMT_CU8:
        # component = (2 u8 - 255) * scale.  Default: the RTL-SDR convention (u8 - 127.5) / 2^k -- here full scale 4.0, 2^-6 --
        # a power-of-two scale, which also selects the library's dot-product instance for offset-binary bytes; any other scale
        # (--cu8-generic: 4/255) runs the generic uint8 instance
        sc = 2.0 ** -6 if scale is None else scale
        fe.ctx.set_format_scale(fmt, sc)
        return torch.clamp(torch.floor(iq * (0.5 / sc) + 128.0), 0, 255).to(torch.uint8).contiguous()
    if fmt == _native.FMT_MAG2:
        return (iq[:, 0] * iq[:, 0] + iq[:, 1] * iq[:, 1]).contiguous()
    return iq


def host_fed_record(args, fe, iq, depth, formats=("fc32", "sc16", "sc8", "cu8")):
    """PCIe-inclusive rates of the host-fed entry point (adsb_submit_format_host) per wire format, each next to a plain
    pinned H2D copy of the same bytes measured in the same run.  Never the bench `value`."""
    import torch
    from gr_adsb_amd import _native
    FM = {"fc32": _native.FMT_FC32, "sc16": _native.FMT_SC16, "sc8": _native.FMT_SC8, "cu8": _native.FMT_CU8}
    chunk = min(iq.shape[0], 1 << args.hostfed_log2n)
    nbuf = 4
    out = {"entry_point": "adsb_submit_format_host, %d chunks in flight" % depth, "chunk_samples": chunk, "formats": {}}

    def run(fmt, srcs, reps):
        pend, nb = [], 0
        for k in range(2):                                 # warm: device input buffers of the slots, staging ring
            fe.ctx.wait(fe.ctx.submit_format_host(fmt, srcs[k % len(srcs)]), fetch=False)
        t0 = time.perf_counter()
        for k in range(reps):
            pend.append(fe.ctx.submit_format_host(fmt, srcs[k % len(srcs)]))
            if len(pend) == depth:
                nb = fe.ctx.wait(pend.pop(0), fetch=False)
        while pend:
            nb = fe.ctx.wait(pend.pop(0), fetch=False)
        dt = time.perf_counter() - t0
        return reps * chunk / dt / 1e6, nb, dt

    for name in formats:
        fmt = FM[name]
        dev = [quantise_for(fmt, iq[k * chunk:(k + 1) * chunk] if (k + 1) * chunk <= iq.shape[0] else iq[:chunk], fe) for k in range(nbuf)]
        pinned = [torch.empty(d.shape, dtype=d.dtype).pin_memory() for d in dev]
        for p_, d in zip(pinned, dev):
            p_.copy_(d)
        torch.cuda.synchronize()
        bytes_per = pinned[0].numel() * pinned[0].element_size()
        # plain pinned H2D of the same chunks, back to back on one stream
        dst = torch.empty_like(dev[0])
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for k in range(3):
                dst.copy_(pinned[k % nbuf], non_blocking=True)
            st.synchronize()
            reps, h2d = 12, 0.0
            for _ in range(2):                                 # best of two: the first burst after an allocation can be slow
                t0 = time.perf_counter()
                for k in range(reps):
                    dst.copy_(pinned[k % nbuf], non_blocking=True)
                st.synchronize()
                h2d = max(h2d, reps * bytes_per / (time.perf_counter() - t0) / 1e9)
        dt_np, per = _native.FMT_LAYOUT[fmt]
        views = [p_.numpy().reshape(-1).view(dt_np) if name != "fc32" else p_.numpy().view(np.complex64).reshape(-1) for p_ in pinned]
        # three repeats, the median reported and all three kept: a transient on the host (round 4's final file once read
        # 0.845 of the plain copy where ten repeats on one box read 0.988-0.997, profiles/r05_hostfed_repeat.txt) shows as one
        p_runs = sorted(run(fmt, views, 12) for _ in range(3))
        p_msps, nb, dt = p_runs[1]
        p_gbs = 12 * bytes_per / dt / 1e9
        pageable = [v.copy() for v in views[:2]]
        g_msps, _, dt = run(fmt, pageable, 8)
        g_gbs = 8 * bytes_per / dt / 1e9
        rec = {"bytes_per_sample": bytes_per // chunk, "bursts_per_chunk": int(nb),
               "pinned": {"value": round(p_msps, 1), "unit": "Msamples/s", "gbytes_per_s": round(p_gbs, 2),
                          "repeats_msamples_per_s": [round(r_[0], 1) for r_ in p_runs]},
               "pageable": {"value": round(g_msps, 1), "unit": "Msamples/s", "gbytes_per_s": round(g_gbs, 2),
                            "vs_pinned": round(g_gbs / p_gbs, 3)},
               "plain_pinned_h2d_gbytes_per_s": round(h2d, 2), "pinned_vs_plain_h2d": round(p_gbs / h2d, 3)}
        if name == "fc32":
            # the same pageable buffers page-locked in place (adsb_host_register: what an application does once per ring buffer)
            t0 = time.perf_counter()
            regs = [_native.RegisteredArray(a) for a in pageable]
            t_reg = (time.perf_counter() - t0) / len(regs)
            r_msps, _, dt = run(fmt, [r.array for r in regs], 8)
            for r in regs:
                r.close()
            rec["registered_in_place"] = {"value": round(r_msps, 1), "unit": "Msamples/s", "gbytes_per_s": round(8 * bytes_per / dt / 1e9, 2),
                                          "register_ms_per_buffer": round(t_reg * 1e3, 2),
                                          "note": "the same pageable buffers after adsb_host_register (once per buffer)"}
        out["formats"][name] = rec
        del pinned, dev, dst, views, pageable
        torch.cuda.empty_cache()
    out["pageable_note"] = ("pageable sources are copied into a ring of four pinned 16 MiB chunks by the context's copy threads "
                            "(adsb_set_copy_threads, default 6 incl. the caller) beside the DMA of the previous chunk")
    # the complex64 record at the top level too (the shape earlier rounds reported)
    out.update({k: v for k, v in out["formats"].get("fc32", {}).items() if k in ("pinned", "pageable", "registered_in_place",
                                                                                 "plain_pinned_h2d_gbytes_per_s", "pinned_vs_plain_h2d")})
    return out


def host_fed_all_ranks(args, fe, iq, depth, rank, n_gpus, sync_all, ag_obj):
    """--gpus N (N > 1): every rank feeds ITS GPU from page-locked host memory through adsb_submit_format_host at the
    same time (one barrier in front, one behind): the PCIe-inclusive figure of the node, per rank and in total.  A rank
    that fails here reports the error and still takes part in every collective (the headline line must not be lost)."""
    import torch
    from gr_adsb_amd import _native
    chunk = min(iq.shape[0], 1 << args.hostfed_log2n)
    reps, err, own, views = 12, None, float("nan"), None
    try:
        # page-locked on the NUMA node of this rank's GPU (adsb_host_alloc_near): what a feeder's ring should be
        pinned = [_native.PinnedArray(chunk, np.complex64, near=fe.ctx) for _ in range(3)]
        src = iq[:chunk].cpu().numpy().view(np.complex64).reshape(-1)
        for p_ in pinned:
            p_.array[:] = src
        views = [p_.array for p_ in pinned]
        for k in range(2):
            fe.ctx.wait(fe.ctx.submit_format_host(_native.FMT_FC32, views[k]), fetch=False)
    except Exception as e:                                       # noqa: BLE001
        err = "%s: %s" % (type(e).__name__, e)
    sync_all()
    t0 = time.perf_counter()
    if err is None:
        try:
            pend = []
            for k in range(reps):
                pend.append(fe.ctx.submit_format_host(_native.FMT_FC32, views[k % 3]))
                if len(pend) == depth:
                    fe.ctx.wait(pend.pop(0), fetch=False)
            while pend:
                fe.ctx.wait(pend.pop(0), fetch=False)
            own = time.perf_counter() - t0
        except Exception as e:                                   # noqa: BLE001
            err = "%s: %s" % (type(e).__name__, e)
    sync_all()
    wall = time.perf_counter() - t0
    ni = fe.ctx.numa_info()
    per = ag_obj({"rank": rank, "numa_node": ni["node"], "local_cpulist": ni["cpulist"], "pci": ni["pci"],
                  "msamples_per_s": None if err else round(reps * chunk / own / 1e6, 1),
                  "gbytes_per_s": None if err else round(reps * chunk * 8 / own / 1e9, 2), "error": err})
    walls = ag_obj(wall)
    ok = all(r["error"] is None for r in per)
    return {"entry_point": "adsb_submit_format_host (complex64, page-locked source), every rank at once", "chunk_samples": chunk,
            "chunks_per_rank": reps, "per_rank": per,
            "total": {"value": round(n_gpus * reps * chunk / max(walls) / 1e6, 1) if ok else None, "unit": "Msamples/s",
                      "gbytes_per_s": round(n_gpus * reps * chunk * 8 / max(walls) / 1e9, 2) if ok else None}}


def one_process_leg(args, devices, context_counts=None, fs=20e6, bursts=1000.0, seed=3):
    """ONE process, N devices, ONE page-locked host ring (adsb_process_sharded_multi; BASELINE config 4's stream, 20 Msps): the
    stream is tiled into overlapped time shards inside the library, one feeder thread per context, seams stitched on the
    host.  PCIe-inclusive (never the bench `value`).  devices: HIP ordinals, one context each; context_counts: also run with
    that many contexts on devices[0] (how a one-GPU box exercises the driver).  Every run is compared with ONE blocking
    canonical call over the whole ring."""
    import torch
    from gr_adsb_amd import _native
    from gr_adsb_amd.frontend import MultiDevice
    per_dev = 1 << args.hostfed_log2n
    runs = [("one context per device", list(devices))] + [("%d contexts on device %d" % (k, devices[0]), [devices[0]] * k)
                                                          for k in (context_counts or [])]
    out = {"entry_point": "adsb_process_sharded_multi (complex64, one page-locked ring, feeder thread per context)",
           "workload": "synthetic %g Msps complex64 IQ, %g bursts/s, seed %d; 2^%d samples per context" % (fs / 1e6, bursts, seed, args.hostfed_log2n),
           "runs": []}
    ring, want, n_have = None, None, 0
    for what, devs in runs:
        n = per_dev * len(devs)
        md = MultiDevice(fs, args.threshold, devices=devs)
        try:
            if n != n_have:
                ring = md.pinned(n, np.complex64)
                with torch.cuda.device(devs[0]):
                    blk = 1 << 24
                    for o in range(0, n, blk):           # generated on the device in blocks, copied down: the ring is host memory
                        m = min(blk, n - o)
                        ring.array[o:o + m] = gen_stream_blocks(m, o, fs, bursts, seed, torch.device("cuda", devs[0]))[:m] \
                            .cpu().numpy().view(np.complex64).reshape(-1)
                want = md.contexts[0].process_format(_native.FMT_FC32, ring.array)
                want["flags"] &= ~np.uint16(_native.BURST_HEAD)
                n_have = n
            rec = {"what": what, "contexts": len(devs), "devices": [int(d) for d in devs], "samples": n, "legs": []}
            for spc in (1, 4):
                got = md.process_host(_native.FMT_FC32, ring.array, spc)
                same = bool(got.tobytes() == want.tobytes())
                tt = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    md.process_host(_native.FMT_FC32, ring.array, spc, out=got if len(got) else None)
                    tt.append(time.perf_counter() - t0)
                dt = float(np.median(tt))
                st = md.last_stats
                rec["legs"].append({"shards_per_context": spc, "value": round(n / dt / 1e6, 1), "unit": "Msamples/s",
                                    "gbytes_per_s": round(n * 8 / dt / 1e9, 2), "ms": round(dt * 1e3, 3), "bursts": int(len(got)),
                                    "identical_to_one_blocking_call": same, "fallbacks": int(st["fallbacks"]),
                                    "feeder_ms": [round(x * 1e3, 3) for x in st["feeder_s"]], "numa_node": st["numa_node"]})
            out["runs"].append(rec)
        finally:
            md.close()
    del ring
    return out


def sharded_leg(args, dev, rank, n_gpus, fs, bursts, seed, n_own, steps, warmup, min_time, depth, sync_all, reduce_max,
                ag_int, ag_obj, synth, fe=None, extra_me=None, untimed=False):
    """One stream of n_own * n_gpus samples tiled as n_gpus overlapped time shards, one per rank; returns this rank's view
    (every rank gets the same dict: times are max over ranks, per_rank and the seam check are gathered)."""
    import torch
    from gr_adsb_amd import sharding
    from gr_adsb_amd.frontend import FrontEnd, shard_plan
    sps = int(fs // 1e6)
    stream_len = n_own * n_gpus
    if fe is None:
        fe = FrontEnd(fs, args.threshold, device=dev.index, timing=True)
    plan = shard_plan(stream_len, n_gpus, sps, align=n_own)[rank]
    iq = gen_stream_blocks(plan["hi"] - plan["lo"], plan["lo"], fs, bursts, seed, dev, **synth)
    torch.cuda.synchronize()
    run = sharding.ShardedRank(fe, iq, plan, stream_len, rank, ag_int, ag_obj, depth)
    fb0 = sharding.STATS["fallbacks"]
    for _ in range(warmup):
        run.step()
    run.drain()
    fe.ctx.reset_stats()
    own_times = []

    def reduce_and_keep(t):
        own_times.append(t)
        return reduce_max(t)

    times, n_bursts = timed_repeats(run.step, run.drain, sync_all, reduce_and_keep, steps, min_time)
    st = fe.stats()
    numa = fe.ctx.numa_info()
    kern_ms = st["detect_ms"] / max(1, st["detect_launches"])
    alg = st["detect_bytes"] / max(1, st["detect_launches"])
    me = {"rank": rank, "device": torch.cuda.current_device(),
          "pci_bus_id": getattr(torch.cuda.get_device_properties(dev), "pci_bus_id", None) or numa["pci"],
          "ms_per_step_min": round(min(own_times) / steps * 1e3, 4), "ms_per_step_max": round(max(own_times) / steps * 1e3, 4),
          "kernel_ms": round(kern_ms, 4),
          "roofline_frac": round(alg / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if kern_ms > 0 else None,
          "shard_samples": int(plan["hi"] - plan["lo"]),
          "stitch_ms_per_step": round(run.stitch_s / max(1, run.passes) * 1e3, 4),
          "stitch_fallbacks": sharding.STATS["fallbacks"] - fb0, "bursts_per_step": int(n_bursts),
          "numa_node": numa["node"], "local_cpulist": numa["cpulist"]}
    me.update(extra_me or {})
    per_rank = ag_obj(me)
    prod = None
    if untimed:
        # the same leg on contexts of the product's default kind (no ADSB_FLAG_TIMING: the ranks' passes free to overlap on one
        # stream per pipeline slot; the timed contexts above keep their k_detect launches in line so that every event pair
        # brackets one launch) -- a wall time only, three repeats, every rank takes part (the exchange is collective)
        fe2 = FrontEnd(fs, args.threshold, device=dev.index, timing=False)
        run2 = sharding.ShardedRank(fe2, iq, plan, stream_len, rank, ag_int, ag_obj, depth)
        for _ in range(warmup):
            run2.step()
        run2.drain()
        t2, _ = timed_repeats(run2.step, run2.drain, sync_all, reduce_max, steps, 0.0, max_repeats=3)
        e2 = float(np.median(t2))
        prod = {"ms_per_step": round(e2 / steps * 1e3, 4), "value": round(float(stream_len) * steps / e2 / 1e6, 1),
                "unit": "Msamples/s", "repeats": len(t2),
                "note": "same shards on contexts without ADSB_FLAG_TIMING (the product default: passes overlap)"}
        fe2.ctx.close()
    seam = seam_check(args, fe, dev, rank, n_gpus, sps, n_own, stream_len, run.last_kept, ag_obj, fs=fs, bursts=bursts,
                      seed=seed, synth=synth)
    elapsed = float(np.median(times))
    return {"fs": fs, "n_own": n_own, "stream_len": stream_len, "times": times, "elapsed": elapsed, "stats": st, "product_default": prod,
            "value": round(float(stream_len) * steps / elapsed / 1e6, 1), "ms_per_step": round(elapsed / steps * 1e3, 4),
            "n_bursts": int(n_bursts), "per_rank": per_rank, "seam": seam, "fe": fe, "iq": iq}


def config4_legs(args, dev, rank, n_gpus, depth, sync_all, reduce_max, ag_int, ag_obj, rank_sync):
    """BASELINE configs[3] with N ranks: ONE 20 Msps stream tiled as N overlapped time shards, host stitch.  Two legs:
    `weak` (2^log2n samples per GPU: the stream grows with N) and `strong` (2^log2n samples in total: every GPU gets 1/N),
    each with per-rank kernel time / roofline fraction and its own seam check."""
    import torch
    out = {"workload": "synthetic 20 Msps complex64 IQ, 1000 DF17-length bursts/s, AWGN 1e-3, seed 3: one stream tiled as %d "
                       "overlapped time shards (one per GPU), heads re-gated on the host after one 16-byte exchange" % n_gpus,
           "fs": 20e6, "rank_sync": rank_sync}
    fe = None
    for name, n_own in (("weak", 1 << args.log2n), ("strong", max(1 << 22, (1 << args.log2n) // n_gpus))):
        leg = sharded_leg(args, dev, rank, n_gpus, 20e6, 1000.0, 3, n_own, args.extra_steps, 3, args.extra_min_time, depth,
                          sync_all, reduce_max, ag_int, ag_obj, {}, fe=fe, untimed=True)
        fe = leg["fe"]
        out[name] = {"scaling": name, "samples_per_gpu_per_step": n_own, "stream_samples_per_step": leg["stream_len"],
                     "value": leg["value"], "unit": "Msamples/s", "ms_per_step": leg["ms_per_step"], "steps": args.extra_steps,
                     "repeats": len(leg["times"]), "bursts_per_step_rank0": leg["n_bursts"],
                     "hbm_frac_whole_job": round(8.0 * leg["stream_len"] / (leg["ms_per_step"] * 1e-3) / 1e9 / (HBM_PEAK_GBS * n_gpus), 4),
                     "per_rank": leg["per_rank"], "seam_check": leg["seam"], "product_default": leg["product_default"],
                     "stitch_fallbacks_total": int(sum(r["stitch_fallbacks"] for r in leg["per_rank"]))}
        del leg
        torch.cuda.empty_cache()
    return out


def _free_port():
    import socket
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    return port


def spawn_ranks(n):
    """`python bench.py --gpus N` started plainly (no launcher around it: WORLD_SIZE unset): start the N ranks ourselves,
    exactly as the driver's launcher would -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N on 127.0.0.1 --
    pass every argument through, hand rank 0's JSON line on and return the launcher's exit code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), ADSB_BENCH_SPAWNED="1")
    print("bench.py: --gpus %d without a launcher: starting the ranks with %s" % (n, " ".join(cmd[1:9])), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


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
    ap.add_argument("--min-time", type=float, default=0.6, help="seconds of timed region to accumulate (repeats of K steps)")
    ap.add_argument("--cpu-log2n", type=int, default=28, help="log2 of the C-port CPU-baseline sample")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baselines and the bit-match leg")
    ap.add_argument("--no-extra", action="store_true", help="skip extra_configs (BASELINE configs 3, 4, 5)")
    ap.add_argument("--no-hostfed", action="store_true", help="skip the host-fed (PCIe-inclusive) record")
    ap.add_argument("--extra-log2n", type=int, default=28)
    ap.add_argument("--extra-steps", type=int, default=20)
    ap.add_argument("--extra-min-time", type=float, default=0.25)
    ap.add_argument("--hostfed-log2n", type=int, default=26)
    ap.add_argument("--host-fed", action="store_true", help="(kept for compatibility: with --gpus N > 1 the PCIe-inclusive leg runs by default)")
    ap.add_argument("--no-host-fed-multi", action="store_true",
                    help="with --gpus N > 1: skip the PCIe-inclusive leg (every rank feeding its GPU from page-locked host memory)")
    ap.add_argument("--no-config4", action="store_true",
                    help="with --gpus N > 1: skip BASELINE config 4's legs (one 20 Msps stream tiled as N overlapped shards, weak and strong)")
    ap.add_argument("--no-one-process", action="store_true",
                    help="with --gpus N > 1: skip the leg in which rank 0 alone drives all N devices from one host ring "
                         "(adsb_process_sharded_multi)")
    ap.add_argument("--mixed-df", action="store_true",
                    help="BASELINE config 5's signal as the main workload: DF mix of docs/DF_histogram.txt, SNR 3-25 dB over noise 2e-3")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed the way an N-GPU launch does (backend cpu:gloo,cuda:nccl, RCCL barrier with "
                         "its gloo fall-back) even with one rank, and let ranks share a GPU (rank r -> device r mod device_count): "
                         "exercises the multi-GPU initialisation on a one-GPU box")
    ap.add_argument("--depth", type=int, default=0, help="passes in flight (default: the library's ADSB_MAX_IN_FLIGHT)")
    ap.add_argument("--low-latency", action="store_true", help="ADSB_FLAG_LOW_LATENCY: tail kernels beside the next pass's k_detect")
    ap.add_argument("--single-stream", action="store_true",
                    help="profiling aid (ADSB_FLAG_SINGLE_STREAM): the sparse tail of a pass behind its k_detect on one stream")
    ap.add_argument("--sc8-generic", action="store_true",
                    help="with --format sc8: scale 4/127 instead of 2^-5, i.e. k_detect's generic int8 instance (any scale) instead "
                         "of the dot-product one (power-of-two scales)")
    ap.add_argument("--cu8-generic", action="store_true",
                    help="with --format cu8: scale 4/255 instead of 2^-6, i.e. k_detect's generic uint8 instance (any scale) instead of "
                         "the dot-product one (power-of-two scales: the RTL-SDR convention)")
    ap.add_argument("--cu8-pow2", action="store_true", help=argparse.SUPPRESS)      # (the default since round 6; kept for old scripts)
    ap.add_argument("--format", choices=["fc32", "mag2", "sc16", "sc8", "cu8"], default="fc32",
                    help="input sample format: complex64 (BASELINE workload), float32 |IQ|^2 (the framer's literal input, "
                         "4 B/sample), int16 IQ (4 B/sample) or 8-bit IQ (2 B/sample: int8 / RTL-SDR offset binary); "
                         "formats other than fc32 N=1 only")
    args = ap.parse_args()

    stray = sorted(k for k in os.environ if k.startswith("ADSB_") and k not in KNOWN_ENV)
    if stray:
        print("bench.py: refusing to run with unknown ADSB_* environment variables set: %s" % ", ".join(stray), file=sys.stderr)
        return 2
    env_known = {k: os.environ[k] for k in KNOWN_ENV if k in os.environ}

    # --gpus N is a promise about the line's n_gpus.  Under a launcher (WORLD_SIZE set) the world must BE N; started plainly
    # with N > 1 the bench starts its own N ranks; anything else is an error -- never a line that says n_gpus: 1.
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench.py: launched with WORLD_SIZE=%d but --gpus %d: refusing to print a line for the wrong number of GPUs "
              "(use --nproc-per-node %d, or start `python bench.py --gpus %d` without a launcher)"
              % (world, args.gpus, args.gpus, args.gpus), file=sys.stderr)
        return 2

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ADSB_BENCH_ONE_GPU=1: debugging aid -- run the N-rank code path with every rank on cuda:0 (gloo only)
    one_gpu = os.environ.get("ADSB_BENCH_ONE_GPU") == "1"
    host_group = None     # host-side collectives (mailbox set-up, object gathers, fallbacks): always gloo, never RCCL
    dist_on = world > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="gloo" if one_gpu else "cpu:gloo,cuda:nccl", rank=rank, world_size=world)
        host_group = dist.new_group(backend="gloo")
    n_gpus = world
    if not one_gpu and not args.force_dist and n_gpus > torch.cuda.device_count():
        print("bench.py: --gpus %d but this node shows %d GPU(s) (ADSB_BENCH_ONE_GPU=1 runs every rank on cuda:0 for debugging)"
              % (n_gpus, torch.cuda.device_count()), file=sys.stderr)
        if dist_on:
            dist.destroy_process_group()
        return 2
    if one_gpu:
        local_rank = 0
    elif args.force_dist:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from gr_adsb_amd import _native, sharding
    from gr_adsb_amd.frontend import FrontEnd

    fs = args.fs
    sps = int(fs // 1e6)
    n_own = 1 << args.log2n
    fe = FrontEnd(fs, args.threshold, device=local_rank, timing=True,
                  flags=(_native.FLAG_SINGLE_STREAM if args.single_stream else 0) | (_native.FLAG_LOW_LATENCY if args.low_latency else 0))
    # one process per GPU: run this rank's host side on the cpus local to its GPU (the library already places its pinned
    # buffers and copy threads there, adsb_numa_info); a one-rank run stays wherever it was started
    numa = fe.ctx.numa_info()
    numa["bound"] = False
    if n_gpus > 1 and numa["cpulist"] and hasattr(os, "sched_setaffinity"):
        try:
            cpus = set()
            for part in numa["cpulist"].split(","):
                a_, _, b_ = part.partition("-")
                cpus.update(range(int(a_), int(b_ or a_) + 1))
            os.sched_setaffinity(0, cpus & os.sched_getaffinity(0) or os.sched_getaffinity(0))
            numa["bound"] = True
        except (OSError, ValueError):
            pass

    intfmt = args.format != "fc32"
    fmt = {"fc32": _native.FMT_FC32, "mag2": _native.FMT_MAG2, "sc16": _native.FMT_SC16, "sc8": _native.FMT_SC8,
           "cu8": _native.FMT_CU8}[args.format]
    if intfmt and n_gpus > 1:
        print("bench.py: --format %s is an N=1 leg" % args.format, file=sys.stderr)
        if dist_on:
            dist.destroy_process_group()
        return 2
    synth = dict(noise_power=2e-3, df_choices=DF_MIX[0], df_weights=DF_MIX[1], snr_db_range=(3.0, 25.0)) if args.mixed_df else {}

    DEPTH = min(args.depth, _native.MAX_IN_FLIGHT) if args.depth > 0 else _native.MAX_IN_FLIGHT

    # 16 bytes per rank per pass, host side: shared-memory mailbox on one node, gloo all_gather across nodes
    ag_int, ag_close = sharding.make_pair_exchange(dist, rank, n_gpus, group=host_group) if n_gpus > 1 else (None, lambda: None)
    transport = None if n_gpus == 1 else ("shm mailbox" if getattr(ag_int, "__self__", None) is not None else "gloo all_gather")

    def ag_obj(o):
        out = [None] * n_gpus
        dist.all_gather_object(out, o, group=host_group)
        return out

    # barrier / max-over-ranks go over RCCL (backend "nccl") on the GPUs; if the communicator cannot be set up on
    # this node they fall back to the gloo side of the same process group rather than losing the run
    sync_dev = [dev]
    if dist_on and not one_gpu:
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

    def sync_group():
        return host_group if sync_dev[0] == "cpu" else None          # RCCL (default group) or the gloo fallback

    def sync_all():
        torch.cuda.synchronize()
        if dist_on:
            dist.all_reduce(torch.zeros(1, device=sync_dev[0]), group=sync_group())      # barrier
            torch.cuda.synchronize()

    def reduce_max(t):
        if not dist_on:
            return t
        tmax = torch.tensor([t], dtype=torch.float64, device=sync_dev[0])
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=sync_group())
        return float(tmax.item())

    rank_sync = None if not dist_on else ("gloo" if sync_dev[0] == "cpu" else "rccl")
    per_rank = seam = hf_multi = cfg4 = None
    iso_ms = untimed = None
    if n_gpus == 1:
        iq = gen_stream_blocks(n_own, 0, fs, args.bursts, args.seed, dev, **synth)
        # |IQ|^2 of the same stream (separately rounded products) / the stream quantised to the integer wire
        # format (full scale 4.0; the kernel converts with the same scale), computed once outside the timed region
        alt_scale = 4.0 / 127.0 if (args.sc8_generic and args.format == "sc8") else (4.0 / 255.0 if (args.cu8_generic and args.format == "cu8") else None)
        iq = quantise_for(fmt, iq, fe, scale=alt_scale)
        torch.cuda.synchronize()
        pending = []          # tickets of submitted, not yet collected passes (pipeline of DEPTH passes)
        last_n = [0]

        def step():
            # submit pass i+1 before collecting pass i: the PCIe copy and host work of one pass overlap the
            # kernels of the next; every pass is collected inside the timed region (drain() below)
            pending.append(fe.submit_format_tensor(fmt, iq, 0))
            if len(pending) == DEPTH:
                last_n[0] = fe.wait(pending.pop(0), fetch=False)

        def drain():
            while pending:
                last_n[0] = fe.wait(pending.pop(0), fetch=False)
            return last_n[0]

        # a few blocking passes in front of everything (they also bring a fresh box's clocks up)
        for _ in range(6):
            fe.ctx.process_format_device(fmt, iq.data_ptr(), n_own, 0, fetch=False)
        for _ in range(args.warmup):
            step()
        drain()
        fe.ctx.reset_stats()
        times, n_bursts = timed_repeats(step, drain, sync_all, reduce_max, args.steps, args.min_time)
        st = fe.stats()
        # the same kernel timed without a neighbour, AFTER the timed region
        iso_ms = isolated_kernel_ms(fe, fmt, iq, n_own)
        # ... and the same pipeline on a context WITHOUT ADSB_FLAG_TIMING (the product default: no event pair between
        # consecutive k_detect launches), a few repeats right behind the timed ones
        untimed = untimed_context_ms(args, local_rank, fmt, iq, n_own, DEPTH, sync_all, scale=alt_scale)
    else:
        leg = sharded_leg(args, dev, rank, n_gpus, fs, args.bursts, args.seed, n_own, args.steps, args.warmup, args.min_time,
                          DEPTH, sync_all, reduce_max, ag_int, ag_obj, synth, fe=fe,
                          extra_me={"process_bound_to_local_cpus": numa["bound"]})
        times, n_bursts, st, per_rank, seam, iq = leg["times"], leg["n_bursts"], leg["stats"], leg["per_rank"], leg["seam"], leg["iq"]
        if not one_gpu and not args.force_dist:
            assert len({r["device"] for r in per_rank}) == n_gpus or len({r["pci_bus_id"] for r in per_rank}) == n_gpus, \
                "ranks share a GPU (set ADSB_BENCH_ONE_GPU=1 if that is intended)"
        if not args.no_host_fed_multi:
            hf_multi = host_fed_all_ranks(args, fe, iq, DEPTH, rank, n_gpus, sync_all, ag_obj)
        del leg, iq
        torch.cuda.empty_cache()
        if not args.no_config4:
            cfg4 = config4_legs(args, dev, rank, n_gpus, DEPTH, sync_all, reduce_max, ag_int, ag_obj, rank_sync)
    elapsed = float(np.median(times))

    result = None
    if rank == 0:
        total_samples = float(n_own) * n_gpus * args.steps
        value = total_samples / elapsed / 1e6
        traffic, traffic_src = pmc_traffic(args.format, fs, args.bursts, args.mixed_df, args.log2n) if n_gpus == 1 else (None, None)
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
            "dtype": "f32" if args.format in ("fc32", "mag2") else {"sc16": "i16->f32", "sc8": "i8->f32", "cu8": "u8->f32"}[args.format],
            "data": "synthetic",
            "timing": {"repeats": len(times), "steps_per_repeat": args.steps, "timed_seconds": round(sum(times), 4),
                       "ms_per_step_min": round(min(times) / args.steps * 1e3, 4),
                       "ms_per_step_median": round(elapsed / args.steps * 1e3, 4),
                       "ms_per_step_max": round(max(times) / args.steps * 1e3, 4),
                       "note": "each repeat = exactly `steps` steps between barrier+synchronize, max over ranks; value uses the median repeat"},
            "config": {
                # (kept under 130 characters: a reader that cuts long strings still sees all of it)
                "workload": "synthetic %g Msps %s IQ, %g %s bursts/s, %s, thr %g; 2^%d samples/GPU/step in HBM"
                            % (fs / 1e6, {"fc32": "complex64", "mag2": "f32 |IQ|^2", "sc16": "int16", "sc8": "int8", "cu8": "uint8"}[args.format], args.bursts,
                               "mixed-DF 3-25 dB" if args.mixed_df else "DF17",
                               "AWGN 2e-3" if args.mixed_df else "AWGN 1e-3", args.threshold, args.log2n),
                "samples_per_gpu_per_step": n_own, "fs": fs, "bursts_per_step_rank0": int(n_bursts),
                "step": "one canonical framer+demod pass (|IQ|^2, preamble detect + tag, gate, PPM slice, records to pinned host memory)",
                "sharding": "none" if n_gpus == 1 else "%d overlapped time shards, host stitch" % n_gpus,
                "rank_sync": rank_sync,
                "launched_by": "bench.py itself (torch.distributed.run)" if os.environ.get("ADSB_BENCH_SPAWNED") == "1" else (
                    "external launcher" if "WORLD_SIZE" in os.environ else "plain process"),
                "pipeline": "%d passes in flight (submit/wait)%s" % (DEPTH, (", single stream" if args.single_stream else "") + (", low-latency tail" if args.low_latency else "")),
                "detect_gap_ms_avg": round(st["detect_gap_ms"] / max(1, st["detect_gaps"]), 4),
                "detect_grid": int(st["detect_grid"]), "retries": int(st["retries"]), "longrun_calls": int(st["longrun_calls"]),
                "long_pulses_per_step": round(st["longrun_pulses"] / max(1, st["calls"]), 2),
                "env": env_known,
                "numa_node": numa["node"], "local_cpulist": numa["cpulist"],
            },
            "roofline": roofline_of(st, args.format, iso_ms),
        }
        if untimed is not None:
            result["timing"]["ms_per_step_untimed_ctx"] = untimed["ms_per_step"]
            result["timing"]["untimed_ctx"] = untimed
        result["roofline"]["kernel_only_msamples_per_s"] = round(
            st["detect_samples"] / max(1, st["detect_launches"]) / (result["roofline"]["kernel_ms"] * 1e-3) / 1e6, 1)
        result["roofline"]["traffic"] = traffic
        result["roofline"]["traffic_source"] = traffic_src or "none for this workload (see profiles/)"
        if n_gpus > 1:
            result["roofline"]["note"] = "rank 0's k_detect over its own shard; every rank's figure is in multi_gpu.per_rank"
            result["multi_gpu"] = {"ranks_seen": len(per_rank), "exchange_transport": transport, "per_rank": per_rank,
                                   "stitch_fallbacks_total": int(sum(r["stitch_fallbacks"] for r in per_rank)),
                                   "seam_check": seam}
            if hf_multi is not None:
                result["host_fed"] = hf_multi
            if cfg4 is not None:
                result["config4_20msps"] = cfg4
            if not args.no_one_process:
                # ONE process driving every GPU of the node from one host ring (the other ranks are idle at the closing barrier;
                # ADSB_BENCH_ONE_GPU=1: the same with every context on cuda:0)
                try:
                    result["one_process"] = one_process_leg(args, [0] * n_gpus if one_gpu else
                                                            list(range(min(n_gpus, torch.cuda.device_count()))))
                except Exception as e:                           # noqa: BLE001  (the headline line must not be lost)
                    result["one_process"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not args.no_cpu and n_gpus == 1 and not intfmt:
            n_cpu = min(n_own, 1 << args.cpu_log2n)
            host = iq[:n_cpu].cpu().numpy().view(np.complex64).reshape(-1)
            msps, crecs = cpu_c_port(host, sps, args.threshold)
            # parity in the same run: the GPU path on the same sample must match the C port bit for bit
            grecs = fe.process_iq_tensor(iq[:n_cpu].contiguous(), 0)
            ncpu = os.cpu_count() or 1
            result["cpu_baseline"] = {
                "value": round(msps, 1), "unit": "Msamples/s", "cores": 1, "kind": "port",
                "sample": "first 2^%d samples of the same stream, oracle/adsb_oracle.c (scalar C restatement of "
                          "the reference path incl. |IQ|^2), best of 3 (about 3 s of CPU work); host has %d cpus"
                          % (int(np.log2(n_cpu)), ncpu),
            }
            result["bit_match"] = {"sample_bursts": int(len(crecs)), "identical": recs_match(grecs, crecs)}
            nthr = min(64, ncpu)
            if nthr > 1:
                result["cpu_baseline"]["c_port_all_threads"] = {
                    "value": round(cpu_c_port_threads(host, sps, args.threshold, nthr), 1), "unit": "Msamples/s",
                    "threads": nthr, "note": "same C port, one contiguous shard of the sample per host thread, no stitch"}
            from oracle import adsb_oracle as O
            n_np = min(n_cpu, 1 << 24)
            x_np = O.mag2(host[:1 << 26]) if n_cpu >= (1 << 26) else O.mag2(host)
            t_np = time.perf_counter()
            o_np = O.run_stream(x_np[:n_np], fs, args.threshold)
            t_np = time.perf_counter() - t_np
            result["cpu_baseline"]["vectorised_numpy_oracle"] = {
                "value": round(n_np / t_np / 1e6, 1), "unit": "Msamples/s", "cores": 1,
                "sample": "first 2^%d samples of |IQ|^2, oracle/adsb_oracle.py (all pulses matched at once, Python only "
                          "over matches: faster than the reference's structure), %d tags" % (int(np.log2(n_np)), len(o_np["tag_offsets"]))}
            result["cpu_baseline"]["reference_structured"] = cpu_reference_structured(x_np, fs, args.threshold, min(64, ncpu))
            result["cpu_baseline"]["reference_structured"]["what"] = (
                "oracle/ref_structured.py: vectorised threshold/edges + per-PULSE Python loop + np.median, the cost "
                "structure of framer.py:83-174 / demod.py:67-110 (|IQ|^2 given); pinned to the goldens and the real reference")
        if n_gpus == 1 and not intfmt and not args.no_hostfed:
            result["host_fed"] = host_fed_record(args, fe, iq, DEPTH)
            result["host_fed"]["one_process"] = one_process_leg(args, [local_rank], context_counts=[3, 8])
        if n_gpus == 1 and not intfmt and not args.no_extra:
            del iq
            torch.cuda.empty_cache()
            result["extra_configs"] = extra_configs(args, dev, DEPTH)
            result["formats"] = format_legs(args, dev, DEPTH)
        # A compact, FLAT copy of every leg's verdict inside `config` (scalars only): a reader that keeps just the headline
        # keys of the line still sees each config's / format's roofline fraction, time per step and bit-match.
        cfgd = result["config"]
        if "bit_match" in result:
            cfgd["bit_match_identical"] = bool(result["bit_match"]["identical"])
            cfgd["bit_match_bursts"] = int(result["bit_match"]["sample_bursts"])
        cfgd["roofline_frac_isolated"] = (result["roofline"].get("isolated") or {}).get("frac")
        if untimed is not None:
            cfgd["ms_per_step_untimed_ctx"] = untimed["ms_per_step"]
        for rec in result.get("extra_configs", []) + result.get("formats", []):
            key = rec["name"].split("_")[0].replace("config", "cfg") if rec["name"].startswith("config") else rec["format"]
            cfgd[key + "_frac"] = rec["roofline"]["frac"]
            cfgd[key + "_frac_isolated"] = (rec["roofline"].get("isolated") or {}).get("frac")
            cfgd[key + "_ms"] = rec["ms_per_step"]
            if "product_default" in rec:
                cfgd[key + "_ms_untimed_ctx"] = rec["product_default"]["ms_per_step"]
            cfgd[key + "_identical"] = bool(rec["bit_match"]["identical"])
            if "sharded" in rec:
                cfgd[key + "_8shards_msps"] = rec["sharded"]["value"]
                cfgd[key + "_stitched_equals_single"] = bool(rec["sharded"]["stitched_equals_single_call"])
        hf = result.get("host_fed", {}).get("formats", {})
        for name, rec in hf.items():
            cfgd["hostfed_%s_pinned_msps" % name] = rec["pinned"]["value"]
            cfgd["hostfed_%s_pageable_msps" % name] = rec["pageable"]["value"]
            cfgd["hostfed_%s_pinned_vs_plain_h2d" % name] = rec["pinned_vs_plain_h2d"]
        op = (result.get("host_fed", {}).get("one_process") or result.get("one_process") or {}).get("runs", [])
        for r_ in op:
            best = max(r_["legs"], key=lambda l_: l_["value"])
            cfgd["one_process_%dctx_msps" % r_["contexts"]] = best["value"]
            cfgd["one_process_%dctx_identical" % r_["contexts"]] = all(l_["identical_to_one_blocking_call"] for l_ in r_["legs"])
        if n_gpus > 1 and result.get("multi_gpu"):
            cfgd["seams_identical"] = bool(result["multi_gpu"]["seam_check"]["all_identical"])
            cfgd["stitch_fallbacks"] = int(result["multi_gpu"]["stitch_fallbacks_total"])
            cfgd["per_rank_roofline_frac"] = [r_["roofline_frac"] for r_ in per_rank]
            if result.get("host_fed", {}).get("total"):
                cfgd["hostfed_total_msps"] = result["host_fed"]["total"]["value"]
            if cfg4 is not None:
                for leg_name in ("weak", "strong"):
                    cfgd["cfg4_%s_msps" % leg_name] = cfg4[leg_name]["value"]
                    cfgd["cfg4_%s_ms" % leg_name] = cfg4[leg_name]["ms_per_step"]
                    if cfg4[leg_name].get("product_default"):
                        cfgd["cfg4_%s_ms_untimed_ctx" % leg_name] = cfg4[leg_name]["product_default"]["ms_per_step"]
                    cfgd["cfg4_%s_seams_identical" % leg_name] = bool(cfg4[leg_name]["seam_check"]["all_identical"])
                    cfgd["cfg4_%s_per_rank_frac" % leg_name] = [r_["roofline_frac"] for r_ in cfg4[leg_name]["per_rank"]]
        result["config"] = order_config(cfgd, n_gpus)
        print(json.dumps(result), flush=True)
    if n_gpus > 1:
        sync_all()
        ag_close()
    if dist_on:
        dist.destroy_process_group()
    return result


# The per-leg verdicts a reader must not lose come FIRST in `config` (a record that keeps only the first two dozen keys
# of the line -- the driver's BENCH_rNN.json does -- still holds every BASELINE config and every input format):
# workload, size, then roofline fraction live / alone and the bit-match of configs 3, 4, 5 and of the five formats.
CONFIG_FIRST = (["workload", "samples_per_gpu_per_step"]
                + ["cfg%d_%s" % (c, k) for c in (3, 4, 5) for k in ("frac", "frac_isolated", "identical")]
                + ["cfg4_8shards_msps"]
                + ["%s_%s" % (f, k) for f in ("mag2", "sc16", "sc8", "sc8g", "cu8") for k in ("frac", "identical")]
                + ["hostfed_fc32_pinned_vs_plain_h2d"])
CONFIG_FIRST_MULTI = ["workload", "samples_per_gpu_per_step", "seams_identical", "stitch_fallbacks", "per_rank_roofline_frac",
                      "hostfed_total_msps", "cfg4_weak_msps", "cfg4_weak_ms", "cfg4_weak_seams_identical", "cfg4_weak_per_rank_frac",
                      "cfg4_strong_msps", "cfg4_strong_ms", "cfg4_strong_seams_identical", "cfg4_strong_per_rank_frac",
                      "cfg4_weak_ms_untimed_ctx", "cfg4_strong_ms_untimed_ctx", "sharding", "rank_sync", "launched_by"]
# (+ one_process_<N>ctx_msps / _identical right behind them, see order_config)


def order_config(cfgd, n_gpus):
    first = CONFIG_FIRST if n_gpus == 1 else CONFIG_FIRST_MULTI + sorted(k for k in cfgd if k.startswith("one_process_"))
    out = {k: cfgd[k] for k in first if k in cfgd}
    out.update((k, v) for k, v in cfgd.items() if k not in out)
    return out


def untimed_context_ms(args, device, fmt, iq, n, depth, sync_all, repeats=3, fs=None, scale=None, steps=None):
    """The same pipeline on a context created WITHOUT ADSB_FLAG_TIMING -- the product default.  Two differences from the timed
    context every roofline figure comes from: no HIP event pair around k_detect, and (round 5) a schedule of its own -- a timed
    context keeps its k_detect launches one behind the other so that each event pair brackets ONE launch with the machine to
    itself; the default lets consecutive passes overlap on one stream per pipeline slot (passes over more than 4 GiB of input
    excepted: gr_adsb_amd/csrc/adsb_hip.hip, enqueue).  Same buffer, same depth, `steps` steps per repeat, median of a few."""
    from gr_adsb_amd import _native
    from gr_adsb_amd.frontend import FrontEnd
    fs = args.fs if fs is None else fs
    steps = args.steps if steps is None else steps
    fe2 = FrontEnd(fs, args.threshold, device=device, timing=False, flags=0)
    if fmt not in (0, 1):
        # the integer wire formats convert with the scale the timed context was given (quantise_for)
        fe2.ctx.set_format_scale(fmt, scale if scale is not None else
                                 {_native.FMT_SC16: 4.0 / 32767.0, _native.FMT_SC8: 4.0 / 128.0, _native.FMT_CU8: 2.0 ** -6}[fmt])
    pend = []
    for _ in range(3):
        fe2.ctx.process_format_device(fmt, iq.data_ptr(), n, 0, fetch=False)
    tt, nb = [], 0
    for _ in range(repeats):
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            pend.append(fe2.submit_format_tensor(fmt, iq, 0))
            if len(pend) == depth:
                nb = fe2.wait(pend.pop(0), fetch=False)
        while pend:
            nb = fe2.wait(pend.pop(0), fetch=False)
        sync_all()
        tt.append((time.perf_counter() - t0) / steps * 1e3)
    fe2.ctx.close()
    return {"ms_per_step": round(float(np.median(tt)), 4), "repeats": repeats, "steps_per_repeat": steps,
            "bursts_per_step": int(nb), "msamples_per_s": round(n / float(np.median(tt)) / 1e3, 1),
            "note": "same buffer and depth on a context without ADSB_FLAG_TIMING: no event pair around k_detect, consecutive "
                    "passes free to overlap (one stream per pipeline slot) unless a pass reads more than 4 GiB"}


def seam_check(args, fe, dev, rank, n_gpus, sps, n_own, stream_len, kept, ag_obj, fs=None, bursts=None, seed=None, synth=None):
    """N>1 exactness evidence at full size: around every shard seam, ONE canonical call over a window that straddles
    the seam (regenerated from the deterministic stream) must report exactly the bursts the two neighbouring ranks
    reported there.  The window call starts from fresh state half a window before the compared region; it re-joins
    the true gate state at the first pulse-free stretch longer than 63*sps, long before that region."""
    W = min(1 << 21, n_own // 2)
    lo_cmp = lambda s: s - W // 2                                   # noqa: E731
    hi_cmp = lambda s: s + W - 200 * sps                            # noqa: E731
    seams = [r * n_own for r in range(1, n_gpus)]
    mine = []                                                       # my records near my two seams
    for s in seams:
        m = (kept["offset"] >= lo_cmp(s)) & (kept["offset"] < hi_cmp(s))
        mine.append(kept[m].copy())
    everyone = ag_obj(mine)
    ok, nrec = True, 0
    if rank >= 1:
        s = rank * n_own
        import torch
        win = gen_stream_blocks(2 * W, s - W, args.fs if fs is None else fs, args.bursts if bursts is None else bursts,
                                args.seed if seed is None else seed, dev, **(synth or {}))
        torch.cuda.synchronize()                                   # torch produced it; the context runs on its own stream
        recs = fe.process_iq_tensor(win, s - W)
        recs = recs[(recs["offset"] >= lo_cmp(s)) & (recs["offset"] < hi_cmp(s))]
        theirs = np.concatenate([everyone[r][rank - 1] for r in range(n_gpus)])
        theirs = theirs[np.argsort(theirs["offset"], kind="stable")]
        ok = recs_match(recs, theirs)
        nrec = len(recs)
        if not ok:
            a, b = set(recs["offset"].tolist()), set(theirs["offset"].tolist())
            print("bench: seam %d differs: window call %d bursts, ranks %d; only in window %s; only in ranks %s"
                  % (rank, len(recs), len(theirs), sorted(a - b)[:8], sorted(b - a)[:8]), file=sys.stderr)
    res = ag_obj({"rank": rank, "identical": bool(ok), "bursts_compared": int(nrec)})
    return {"window_samples": 2 * W, "per_seam": res[1:], "all_identical": all(r["identical"] for r in res)}


if __name__ == "__main__":
    rc = main()
    sys.exit(rc if isinstance(rc, int) else 0)
