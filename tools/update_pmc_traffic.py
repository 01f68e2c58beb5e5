"""profiles/rNN_pmc_hbm_traffic_log2n*.txt (tools/profile_round.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)
-> profiles/pmc_traffic.json, the HBM bytes per k_detect launch that bench.py reports as roofline.traffic.
    python tools/update_pmc_traffic.py r02
Correction per MI355X_MICROARCH.md: the counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a
wide coalesced stream (16 B per lane), so it is doubled; WRITE_SIZE is taken as is."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
    entries = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "%s_pmc_hbm_traffic_log2n*.txt" % rnd))):
        log2n = int(re.search(r"log2n(\d+)", path).group(1))
        txt = open(path).read()
        vals = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            m = re.search(r"k_detect<0>\s*\n\s*%s\s+n=\d+\s+avg=([0-9.e+]+)" % c, txt)
            vals[c] = float(m.group(1))
        fetch = vals["FETCH_SIZE"] * 1024 * 2
        write = vals["WRITE_SIZE"] * 1024
        n = 1 << log2n
        entries.append({"fs": 2e6, "log2n": log2n, "bursts": 1000.0, "kernel": "k_detect<complex64>",
                        "FETCH_SIZE_KiB_raw": vals["FETCH_SIZE"], "fetch_bytes_corrected": fetch,
                        "WRITE_SIZE_KiB": vals["WRITE_SIZE"], "traffic_bytes": fetch + write,
                        "algorithmic_bytes": 8 * n - 8 * 15,
                        "ratio_to_algorithmic": round((fetch + write) / (8.0 * n), 4),
                        "source": "profiles/%s (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes)" % os.path.basename(path)})
    out = {"_comment": "HBM traffic of k_detect<complex64> per launch from rocprofv3 PMC passes on MI355X. Collected as "
                       "MI355X_MICROARCH.md prescribes: separate --pmc passes with --kernel-trace only (FETCH_SIZE and WRITE_SIZE "
                       "cannot share a pass); counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a "
                       "wide coalesced stream, so it is doubled. Collected by tools/profile_round.sh, converted by "
                       "tools/update_pmc_traffic.py.", "round": rnd, "entries": entries}
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    for e in entries:
        print("log2n %d: %.4f GB traffic vs %.4f GB algorithmic (%.4fx)" % (e["log2n"], e["traffic_bytes"] / 1e9, e["algorithmic_bytes"] / 1e9, e["ratio_to_algorithmic"]))


if __name__ == "__main__":
    main()
