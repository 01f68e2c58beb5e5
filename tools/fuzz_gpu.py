"""GPU fuzz campaign: adversarial streams through the C ABI vs the C oracle for a given number of seconds -- float |IQ|^2,
complex64 and integer entry points, canonical, block-by-block, host-fed, GNU Radio chunk schedules (paired and unpaired
blocks), confidence ratios, the length-aware gate, rates 2-100 Msps.  python tools/fuzz_gpu.py [seconds] [seed0]
A bounded run with fixed seeds is part of `-m gpu` (tests/test_gpu_fuzz.py): run(budget, seed0)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gr_adsb_amd import _native, replay          # noqa: E402
from gr_adsb_amd import modulator as M          # noqa: E402
from helpers import assert_recs_equal           # noqa: E402
from oracle import adsb_oracle as O             # noqa: E402
from oracle import c_oracle as C                # noqa: E402
from test_sim_property import adversarial_stream  # noqa: E402


def want_flags(ctx, x):
    return ctx.process_mag2(x)["flags"]


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    print("fuzz: %(cases)d cases (%(gr)d with a GNU Radio chunk schedule), %(bursts)d bursts, seeds up to %(last_seed)d, %(seconds).0f s: "
          "all identical" % run(budget, seed))


def run(budget, seed, max_n=None):
    """Cases with seeds seed, seed + 1, ... until `budget` seconds are used; raises AssertionError on the first difference.
    max_n: leave out stream lengths above it (the bounded test run skips the multi-megasample cases)."""
    _native.load()
    ctxs = {}
    la_ctxs = {}
    cf_ctxs = {}
    t0 = time.time()
    n_cases = n_bursts = 0
    n_gr = [0]
    sizes = [1, 17, 240, 1023, 1024, 1025, 1279, 1280, 1281, 4095, 4096, 4097, 4111, 4352, 8191, 8192, 8193, 12288, 20000,
             70000, 300001, 1 << 20, (1 << 22) + 5, 6_000_000]
    while time.time() - t0 < budget:
        rng = np.random.default_rng(seed)
        n = int(rng.choice(sizes, p=np.array([3] * 20 + [2, 1, 0.5, 0.25]) / (60 + 3.75)))
        if max_n is not None and n > max_n:
            n = int(max_n)
        sps = int(rng.choice([2, 4, 8, 20, 6, 10, 12, 16, 30, 100]))      # instantiated rates + run-time-stride ones
        thr = float(rng.choice([0.01, 0.0099, 0.0101, 0.004, 0.05]))
        x = adversarial_stream(rng, n, sps)
        if rng.random() < 0.2:                        # signed zeros in the noise windows (a zero median is +0.0: np.mean's sum)
            z = np.flatnonzero(x == 0)
            x[z[rng.random(len(z)) < 0.5]] = np.float32(-0.0)
        ctx = ctxs.setdefault(sps, _native.Context(sps * 1e6, thr))
        ctx.set_threshold(thr)
        what = "seed %d n %d sps %d thr %g" % (seed, n, sps, thr)
        want = C.canonical(x, sps, np.float32(thr))
        assert_recs_equal(ctx.process_mag2(x), want, what + " mag2")
        if rng.random() < 0.5:                        # complex64 entry: the oracle gets the |IQ|^2 of the same IQ
            with np.errstate(all="ignore"):
                iq = (np.sqrt(np.abs(x)) * np.exp(1j * rng.uniform(0, 6.28, n))).astype(np.complex64)
            assert_recs_equal(ctx.process_iq(iq), C.canonical(O.mag2(iq), sps, np.float32(thr)), what + " iq")
        if n >= 4096 and rng.random() < 0.5:          # block-by-block replay == one call
            blk = int(rng.choice([1500, 4096, 5000, 65536]))

            def shard_fn(plan, hc):
                try:
                    return ctx.shard_host(_native.FMT_MAG2, x[plan["lo"]:plan["hi"]], plan["lo"], plan["own_lo"], plan["own_hi"], n, hc)
                except _native.AdsbError as e:
                    if e.code != -75:
                        raise
                    return None
            try:
                parts = list(replay.replay_blocks(n, sps, blk, shard_fn))
            except (TypeError, AttributeError):
                parts = None                           # a plateau ran past a block's halo (-EOVERFLOW): not stitchable
            if parts is not None:
                got = np.concatenate(parts) if parts else np.zeros(0, _native.BURST_DTYPE)
                assert_recs_equal(got, want, what + " replay %d" % blk)
        if 240 <= n <= 20000 and rng.random() < 0.3:  # GNU Radio emulation: random chunk schedule vs the NumPy oracle
            from gr_adsb_amd import blocks, grshim
            sch = []
            while sum(sch) < n:
                sch.append(int(min(rng.choice([1, 7, 100, 700, 4096, 9000]), n - sum(sch))))
            fr = blocks.framer(sps * 1e6, thr)
            dm = blocks.demod(sps * 1e6, framer=fr if rng.random() < 0.5 else None)    # paired: PDUs from the framer's pass
            dm.start_timestamp = 0.0
            tags, msgs = grshim.drive(fr, dm, x, sch)
            o = O.run_stream(x, sps * 1e6, thr, sch)
            assert np.array_equal(np.array([t.offset for t in tags], dtype=np.int64), o["tag_offsets"]), what + " gr tags"
            snr = np.array([t.value[1] for t in tags], dtype=np.float32)
            assert np.array_equal(snr.view(np.uint32), o["tag_snr"].view(np.uint32)), what + " gr snr"
            bits = np.array([m[1] for _, m in msgs], dtype=np.uint8).reshape(-1, 112)
            assert np.array_equal(bits, o["pdu_bits"]), what + " gr pdus"
            n_gr[0] += 1
        if n >= 1 and rng.random() < 0.3:             # host-fed pipelined submission (pageable source) == blocking call
            t1 = ctx.submit_format_host(_native.FMT_MAG2, x)
            t2 = ctx.submit_format_host(_native.FMT_MAG2, x, abs_offset=77)
            assert_recs_equal(ctx.wait(t1), want, what + " host-fed")
            r2 = ctx.wait(t2)
            r2["offset"] -= 77
            assert_recs_equal(r2, want, what + " host-fed #2")
        if rng.random() < 0.3:                        # device-resident: three submissions in flight (one stream per slot) and
            import torch                              # the C sharded driver (adsb_process_sharded_device) == the blocking call
            t = torch.from_numpy(x).to("cuda:0")
            torch.cuda.synchronize()
            tk = [ctx.submit_format_device(_native.FMT_MAG2, t.data_ptr(), n, 1000 * k) for k in range(3)]
            for k, q in enumerate(tk):
                r = ctx.wait(q)
                r["offset"] -= 1000 * k
                assert_recs_equal(r, want, what + " submitted #%d" % k)
            if n >= 2048:
                try:
                    got = ctx.process_sharded_device(_native.FMT_MAG2, t.data_ptr(), n, int(rng.integers(1, 10)))
                except _native.AdsbError as e:
                    if e.code != -75:                 # a plateau ran past a shard's halo (-EOVERFLOW): not stitchable
                        raise
                    got = None
                if got is not None:
                    assert_recs_equal(got, want, what + " sharded in C")
                    assert np.array_equal(got["flags"] & 0x1FE3, want_flags(ctx, x) & 0x1FE3), what + " sharded flags"
        if n <= 20000 and rng.random() < 0.3:         # fused-path confidence ratios (ADSB_FLAG_CONFIDENCE) vs demod.py:97-101
            cf = cf_ctxs.setdefault(sps, _native.Context(sps * 1e6, thr, flags=_native.FLAG_CONFIDENCE))
            cf.set_threshold(thr)
            rc = cf.process_mag2(x)
            ratio = cf.last_confidence()
            o = O.run_stream(x, sps * 1e6, thr)
            assert_recs_equal(rc, want, what + " confidence ctx")
            dem = (rc["flags"] & 1) != 0
            gr, wr = ratio[dem], o["pdu_ratio"]
            assert gr.shape == wr.shape and np.array_equal(gr, wr, equal_nan=True), what + " ratios"
            fin = ~np.isnan(wr)
            assert np.array_equal(gr[fin].view(np.uint32), wr[fin].view(np.uint32)), what + " ratio bits"
        if rng.random() < 0.3:                        # opt-in length-aware gate vs its oracle restatement
            la = la_ctxs.setdefault(sps, _native.Context(sps * 1e6, thr, flags=_native.FLAG_LONG_AWARE_GATE))
            la.set_threshold(thr)
            with C.long_aware_gate():
                wl = C.canonical(x, sps, np.float32(thr))
            assert_recs_equal(la.process_mag2(x), wl, what + " long-aware")
        if n >= 240 and rng.random() < 0.25:           # integer wire formats: exact conversion, then identical downstream
            fmt = int(rng.choice([_native.FMT_SC16, _native.FMT_SC8, _native.FMT_CU8]))
            amp = np.sqrt(np.clip(np.nan_to_num(x, nan=0.0), 0, 1.0)).astype(np.float32)
            ph = rng.uniform(0, 6.28, n)
            iqf = (amp * np.exp(1j * ph)).astype(np.complex64)
            if fmt == _native.FMT_SC16:
                q, scale = M.quantize_iq16(iqf, full_scale=2.0), float(np.float32(2.0 / 32767.0))
                xq = O.mag2_iq16(q, scale)
            else:
                ob = fmt == _native.FMT_CU8
                q = M.quantize_iq8(iqf, full_scale=2.0, offset_binary=ob)
                # half of the cases with a power-of-two scale (the dot-product instances of k_detect: int8 2/128, uint8 2/256),
                # half with 2/127 resp. 2/255 (the generic instances)
                p2 = rng.random() < 0.5
                scale = float(np.float32((2.0 / 256.0 if p2 else 2.0 / 255.0) if ob else (2.0 / 128.0 if p2 else 2.0 / 127.0)))
                xq = O.mag2_iq8(q, scale, ob)
            ctx.set_format_scale(fmt, scale)
            assert_recs_equal(ctx.process_format(fmt, q), C.canonical(xq, sps, np.float32(thr)), what + " fmt %d" % fmt)
        if rng.random() < 0.1:                         # chunk-invariant blocks on a burst stream, random tiny chunks
            from gr_adsb_amd import blocks, grshim
            L = int(rng.choice([3000, 9000, 20000]))
            iqb = M.synth_iq(L, sps * 1e6, float(rng.choice([3000, 20000, 60000])), int(rng.integers(1 << 30)))
            xb = O.mag2(iqb)
            fr, dm = blocks.framer(sps * 1e6, 0.01, improved=True), blocks.demod(sps * 1e6, improved=True)
            dm.start_timestamp = 0.0
            pad = fr.delay + 512
            xx = np.concatenate([xb, np.zeros(pad, np.float32)])
            sch = []
            while sum(sch) < len(xx):
                sch.append(int(min(rng.choice([1, 3, 50, 257, 1024, 5000]), len(xx) - sum(sch))))
            tags, msgs = grshim.drive(fr, dm, xx, sch)
            # "one reference work() call over the whole stream": the stream the blocks saw is xx, padding included (a pulse
            # whose centre lies in the last 8*sps-1 samples of xb is only look-ahead in a call over xb alone, but is
            # evaluated -- by the reference too -- once the zeros follow it)
            wb = C.canonical(xx, sps, np.float32(0.01))
            assert np.array_equal(np.array([t.value[2] for t in tags], dtype=np.int64), wb["offset"]), what + " improved tags"
            offs = np.array([int(round(m[0]["timestamp"] * sps * 1e6)) for _, m in msgs], dtype=np.int64)
            bits = np.array([m[1] for _, m in msgs], dtype=np.uint8).reshape(-1, 112)
            # the demod sees the framer's output, i.e. the stream delayed by fr.delay: it can complete the bursts that end
            # fr.delay samples before the end of what was fed (the others wait for input that never comes)
            dem = wb["offset"] + 119 * sps + sps // 2 < len(xx) - fr.delay
            assert np.array_equal(offs, wb["offset"][dem]) and np.array_equal(np.packbits(bits, axis=1), wb["bits"][dem]), what + " improved pdus"
            n_gr[0] += 1
        n_cases += 1
        n_bursts += len(want)
        seed += 1
    for group in (ctxs, la_ctxs, cf_ctxs):
        for c_ in group.values():
            c_.close()
    return {"cases": n_cases, "gr": n_gr[0], "bursts": n_bursts, "last_seed": seed - 1, "seconds": time.time() - t0}


if __name__ == "__main__":
    main()
