"""profiles/rNN_<workload>_pmc_hbm.txt (tools/profile_round.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)
-> profiles/pmc_traffic.json, the HBM bytes per k_detect launch that bench.py reports as roofline.traffic for the headline
workload, every extra_configs entry and every --format.
    python tools/update_pmc_traffic.py r03
Correction per MI355X_MICROARCH.md: the counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a
wide coalesced stream (16 B per lane -- every k_detect instance loads that way), so it is doubled; WRITE_SIZE as is."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# workload name in tools/profile_round.sh -> (format, fs, bursts/s, mixed-DF, log2 samples per launch)
WORKLOADS = {
    "cfg2_2msps_fc32": ("fc32", 2e6, 1000.0, False, 30),
    "cfg3_8msps_dense_fc32": ("fc32", 8e6, 6000.0, False, 28),
    "cfg4_20msps_fc32": ("fc32", 20e6, 1000.0, False, 28),
    "cfg5_mixed_df_fc32": ("fc32", 2e6, 1000.0, True, 28),
    "fmt_mag2": ("mag2", 2e6, 1000.0, False, 30),
    "fmt_sc16": ("sc16", 2e6, 1000.0, False, 30),
    "fmt_sc8": ("sc8", 2e6, 1000.0, False, 30),
    "fmt_cu8": ("cu8", 2e6, 1000.0, False, 30),
}
BYTES = {"fc32": 8, "mag2": 4, "sc16": 4, "sc8": 2, "cu8": 2}


def traffic_key(fmt, fs, bursts, mixed, log2n):
    return "%s|fs=%g|bursts=%g|mixed=%d|log2n=%d" % (fmt, fs, bursts, int(bool(mixed)), log2n)


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
    entries = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "%s_*_pmc_hbm.txt" % rnd))):
        name = os.path.basename(path)[len(rnd) + 1:-len("_pmc_hbm.txt")]
        if name not in WORKLOADS:
            continue
        fmt, fs, bursts, mixed, log2n = WORKLOADS[name]
        txt = open(path).read()
        vals = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            m = re.search(r"k_detect<[^\n]*\n\s*%s\s+n=\d+\s+avg=([0-9.e+]+)" % c, txt)
            if m:
                vals[c] = float(m.group(1))
        if len(vals) != 2:
            print("skipping %s: counters not found" % path)
            continue
        fetch = vals["FETCH_SIZE"] * 1024 * 2
        write = vals["WRITE_SIZE"] * 1024
        alg = BYTES[fmt] * (1 << log2n)
        entries[traffic_key(fmt, fs, bursts, mixed, log2n)] = {
            "workload": name, "kernel": "k_detect<%s>" % fmt, "FETCH_SIZE_KiB_raw": vals["FETCH_SIZE"], "fetch_bytes_corrected": fetch,
            "WRITE_SIZE_KiB": vals["WRITE_SIZE"], "traffic_bytes": fetch + write, "algorithmic_bytes": alg,
            "ratio_to_algorithmic": round((fetch + write) / alg, 4),
            "source": "profiles/%s (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes)" % os.path.basename(path)}
    out = {"_comment": "HBM traffic of k_detect per launch from rocprofv3 PMC passes on MI355X, per BASELINE config and input format. "
                       "Collected as MI355X_MICROARCH.md prescribes: separate --pmc passes with --kernel-trace only (FETCH_SIZE and "
                       "WRITE_SIZE cannot share a pass); counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes "
                       "of a wide coalesced stream (16 B per lane), so it is doubled. Collected by tools/profile_round.sh, converted "
                       "by tools/update_pmc_traffic.py.", "round": rnd, "entries": entries}
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    for k, e in entries.items():
        print("%-48s %.4f GB traffic vs %.4f GB algorithmic (%.4fx)" % (k, e["traffic_bytes"] / 1e9, e["algorithmic_bytes"] / 1e9, e["ratio_to_algorithmic"]))


if __name__ == "__main__":
    main()
