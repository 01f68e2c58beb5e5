"""Generate tests/golden/*.npz from the REAL reference (container-only; needs /root/reference).

Each fixture = int16 interleaved IQ (the modulator's output, quantised) + what the unmodified reference
framer.py/demod.py produced for it, driven by tools/ref_harness.py with the stated chunk schedules.
The .npz files are data (inputs and expected outputs); no reference source is stored.

    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import ref_harness as R  # noqa: E402
from gr_adsb_amd import modulator as M  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

FIXTURES = [
    # name, fs, n, bursts/s, seed, threshold, kwargs
    ("g2msps_df17", 2e6, 1 << 17, 3000, 101, 0.01, {}),
    ("g4msps_df17", 4e6, 1 << 17, 5000, 102, 0.01, {}),
    ("g8msps_dense", 8e6, 1 << 17, 8000, 103, 0.01, {}),
    ("g20msps", 20e6, 1 << 17, 6000, 104, 0.01, {}),
    ("g2msps_mixed_lowsnr", 2e6, 1 << 17, 4000, 105, 0.01,
     dict(noise_power=2e-3, df_choices=(0, 4, 5, 11, 16, 17), df_weights=(0.27, 0.14, 0.01, 0.45, 0.02, 0.11),
          snr_db_range=(3, 25))),
]


def schedules(n, rng):
    out = {"single": [n], "fixed4096": [4096] * (n // 4096), "fixed8192": [8192] * (n // 8192)}
    s, rem = [], n
    while rem > 0:
        c = int(min(rem, rng.integers(1000, 9000)))
        s.append(c)
        rem -= c
    out["random"] = s
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, fs, n, bps, seed, thr, kw in FIXTURES:
        iq = M.synth_iq(n, fs, bps, seed, **kw)
        q = M.quantize_iq16(iq)
        x = M.mag2(M.dequantize_iq16(q))
        rng = np.random.default_rng(seed)
        data = dict(iq16=q, fs=np.float64(fs), threshold=np.float64(thr))
        for sname, sched in schedules(n, rng).items():
            r = R.run_reference(x, fs, thr, None if sname == "single" else sched)
            assert r["snr_types"] <= {"float32"}
            data[sname + "_schedule"] = np.array(sched, dtype=np.int64)
            data[sname + "_tag_offsets"] = r["tag_offsets"]
            data[sname + "_tag_snr_bits"] = r["tag_snr"].view(np.uint32)
            data[sname + "_pdu_offsets"] = r["pdu_offsets"]
            data[sname + "_pdu_bits"] = np.packbits(r["pdu_bits"], axis=1)
            data[sname + "_pdu_snr_bits"] = r["pdu_snr"].view(np.uint32)
            if sname == "single":
                data["single_pdu_conf_bits"] = r["pdu_conf"].view(np.uint32)
            data[sname + "_final_prev_eob"] = np.int64(r["final_prev_eob"])
            print(name, sname, "tags", len(r["tag_offsets"]), "pdus", len(r["pdu_offsets"]))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)


if __name__ == "__main__":
    main()
