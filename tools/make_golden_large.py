"""Round-3 fixtures from the REAL reference (container-only; needs /root/reference):

  tests/golden/L*.npz   large vectors, 2^20 - 2^21 samples per rate (>= 500 tags each, 20 Msps included): int8 interleaved
                        IQ (the cs8 wire format, 2 B/sample, so that megasample inputs stay small in the repository) + what
                        the unmodified framer.py / demod.py produced for |IQ|^2 of exactly those bytes under three chunk
                        schedules (single call, fixed 2048 = the deaf-state schedule of framer.py:177-179, random 1000-9000)
  tests/golden/P*.npz   pathological float32 |IQ|^2 streams (NaN / inf, thresholds <= 0, plateaus over several 1024-sample
                        tiles, streams that start / end above the threshold, exact ties) + the reference's outputs

The .npz files are data (inputs and expected outputs); no reference source is stored.   python tools/make_golden_large.py
"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_harness as R  # noqa: E402
from gr_adsb_amd import modulator as M  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
FULL_SCALE = 4.0      # LSB 0.0315: sigma of the default noise = 0.7 LSB (compresses, and makes exact ties common)

LARGE = [
    # name, fs, n, bursts/s, seed, threshold, kwargs
    ("L2msps_df17", 2e6, 1 << 20, 3000, 201, 0.01, {}),
    ("L4msps_df17", 4e6, 1 << 20, 6000, 202, 0.01, {}),
    ("L8msps_dense", 8e6, 1 << 21, 6000, 203, 0.01, {}),
    ("L20msps", 20e6, 3 << 20, 6000, 204, 0.01, {}),
    ("L2msps_mixed_lowsnr", 2e6, 1 << 20, 3000, 205, 0.01,
     dict(noise_power=2e-3, df_choices=(11, 0, 4, 17, 20, 5, 21), df_weights=tuple(np.array((2335, 1395, 732, 582, 61, 34, 32)) / 5171.0),
          snr_db_range=(3, 25))),
]


def random_schedule(n, rng, lo=1000, hi=9000):
    s, rem = [], n
    while rem > 0:
        c = int(min(rem, rng.integers(lo, hi)))
        s.append(c)
        rem -= c
    return s


def store(data, sname, sched, r, conf):
    data[sname + "_schedule"] = np.array(sched, dtype=np.int64)
    data[sname + "_tag_offsets"] = r["tag_offsets"]
    data[sname + "_tag_snr_bits"] = r["tag_snr"].view(np.uint32)
    data[sname + "_pdu_offsets"] = r["pdu_offsets"]
    data[sname + "_pdu_bits"] = np.packbits(r["pdu_bits"], axis=1)
    data[sname + "_pdu_snr_bits"] = r["pdu_snr"].view(np.uint32)
    if conf:
        data[sname + "_pdu_conf_bits"] = r["pdu_conf"].view(np.uint32)
    data[sname + "_final_prev_eob"] = np.int64(r["final_prev_eob"])
    data[sname + "_final_prev_in0_bits"] = np.float32(r["final_prev_in0"]).view(np.uint32)


def pathological():
    """(name, x, fs, thr) -- float32 |IQ|^2 given directly (NaN, inf, -0.0 are not reachable from IQ)."""
    from test_sim_property import adversarial_stream
    out = []
    base2 = M.mag2(M.synth_iq(1 << 16, 2e6, 5000, seed=301))
    x = base2.copy()
    x[[100, 5000, 5001, 20000, 33333]] = np.nan          # NaN in noise windows, as a peak, inside a burst
    x[[3000, 30000, 40001]] = np.inf
    for s in np.flatnonzero(x > 0.2)[::400][:12]:        # NaN shortly before / on strong samples
        x[max(0, s - 37)] = np.nan
    out.append(("Pnan_inf", x, 2e6, 0.01))
    out.append(("Pnan_inf_thr0", x, 2e6, 0.0))
    out.append(("Pthr0", base2, 2e6, 0.0))
    out.append(("Pthr_negative", base2, 2e6, -1.0))
    out.append(("Pthr_low", base2, 2e6, 0.001))          # every noise sample is a pulse
    # plateaus over several tiles, before / after / inside bursts; stream starts and ends above the threshold
    for sps, seed in ((2, 302), (8, 303), (20, 304)):
        fs = sps * 1e6
        b = M.mag2(M.synth_iq(1 << 16, fs, 4000 if sps < 20 else 9000, seed=seed))
        x = b.copy()
        x[:700] = 0.7
        x[-333:] = 0.4
        for s, ln in ((3000, 1023), (7000, 1024), (11000, 1025), (15000, 1279), (19000, 1280), (23000, 1281), (27000, 2500),
                      (33000, 5000), (45000, 64), (50000, 65), (52000, 3 * 1024 + 7)):
            x[s:s + ln] = np.maximum(x[s:s + ln], 0.3)
        out.append(("Pruns_%dmsps" % sps, x, fs, 0.01))
    rng = np.random.default_rng(305)
    for sps in (2, 4, 8, 20):                            # exact ties, structures on the 1024 / 1280 / 128 seams
        out.append(("Pties_%dmsps" % sps, adversarial_stream(rng, 40000, sps), sps * 1e6, 0.01))
    out.append(("Pties_thr_on_value", adversarial_stream(rng, 30000, 2), 2e6, float(np.float32(0.0099))))
    for n in (1, 15, 16, 17, 239, 240, 241, 1023, 1025):  # tiny inputs
        out.append(("Ptiny_%d" % n, base2[7000:7000 + n].copy(), 2e6, 0.01))
    out.append(("Pall_high", np.full(5000, 0.3, np.float32), 2e6, 0.01))
    out.append(("Pall_zero", np.zeros(5000, np.float32), 2e6, 0.01))
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    warnings.simplefilter("ignore")
    for name, fs, n, bps, seed, thr, kw in LARGE:
        iq = M.synth_iq(n, fs, bps, seed, **kw)
        q = M.quantize_iq8(iq, full_scale=FULL_SCALE)
        scale = np.float32(FULL_SCALE / 127.0)
        # |IQ|^2 of exactly those bytes: component = f32(int8) * scale (one rounded multiply), then re*re + im*im
        v = q.astype(np.float32) * scale
        x = (v[0::2] * v[0::2] + v[1::2] * v[1::2]).astype(np.float32)
        rng = np.random.default_rng(seed)
        data = dict(iq8=q, fs=np.float64(fs), threshold=np.float64(thr), scale=scale)
        scheds = {"single": [n], "fixed2048": [2048] * (n // 2048), "random": random_schedule(n, rng)}
        for sname, sched in scheds.items():
            r = R.run_reference(x, fs, thr, None if sname == "single" else sched)
            assert r["snr_types"] <= {"float32"}
            store(data, sname, sched, r, conf=(sname == "single"))
            print(name, sname, "tags", len(r["tag_offsets"]), "pdus", len(r["pdu_offsets"]), flush=True)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
    for name, x, fs, thr in pathological():
        n = len(x)
        rng = np.random.default_rng(len(name) + n)
        data = dict(x=np.asarray(x, dtype=np.float32), fs=np.float64(fs), threshold=np.float64(thr))
        scheds = {"single": [n]}
        if n >= 4096:
            scheds["fixed2048"] = [2048] * (n // 2048) + ([n % 2048] if n % 2048 else [])
            scheds["random"] = random_schedule(n, rng, 1, 3000)
        for sname, sched in scheds.items():
            r = R.run_reference(data["x"], fs, thr, None if sname == "single" else sched)
            store(data, sname, sched, r, conf=(sname == "single"))
            print(name, sname, "tags", len(r["tag_offsets"]), "pdus", len(r["pdu_offsets"]), flush=True)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)


if __name__ == "__main__":
    main()
