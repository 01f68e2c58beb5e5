"""GPU fuzz campaign (not part of the test suite): adversarial streams through the C ABI vs the C oracle for a given
number of seconds, float |IQ|^2 and complex64 entry points, canonical and sharded.  python tools/fuzz_gpu.py [seconds] [seed0]"""
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


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    _native.load()
    ctxs = {}
    t0 = time.time()
    n_cases = n_bursts = 0
    n_gr = [0]
    sizes = [1, 17, 240, 1023, 1024, 1025, 1279, 1280, 1281, 4095, 4096, 4097, 4111, 4352, 8191, 8192, 8193, 12288, 20000,
             70000, 300001, 1 << 20, (1 << 22) + 5, 6_000_000]
    while time.time() - t0 < budget:
        rng = np.random.default_rng(seed)
        n = int(rng.choice(sizes, p=np.array([3] * 20 + [2, 1, 0.5, 0.25]) / (60 + 3.75)))
        sps = int(rng.choice([2, 4, 8, 20]))
        thr = float(rng.choice([0.01, 0.0099, 0.0101, 0.004, 0.05]))
        x = adversarial_stream(rng, n, sps)
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
            fr, dm = blocks.framer(sps * 1e6, thr), blocks.demod(sps * 1e6)
            dm.start_timestamp = 0.0
            tags, msgs = grshim.drive(fr, dm, x, sch)
            o = O.run_stream(x, sps * 1e6, thr, sch)
            assert np.array_equal(np.array([t.offset for t in tags], dtype=np.int64), o["tag_offsets"]), what + " gr tags"
            snr = np.array([t.value[1] for t in tags], dtype=np.float32)
            assert np.array_equal(snr.view(np.uint32), o["tag_snr"].view(np.uint32)), what + " gr snr"
            bits = np.array([m[1] for _, m in msgs], dtype=np.uint8).reshape(-1, 112)
            assert np.array_equal(bits, o["pdu_bits"]), what + " gr pdus"
            n_gr[0] += 1
        n_cases += 1
        n_bursts += len(want)
        seed += 1
    print("fuzz: %d cases (%d with a GNU Radio chunk schedule), %d bursts, seeds up to %d, %.0f s: all identical" % (n_cases, n_gr[0], n_bursts, seed - 1, time.time() - t0))


if __name__ == "__main__":
    main()
