"""The table of profiles/README.md that holds the rocprofv3 kernel traces against the unprofiled bench lines of the same box.

    python tools/reconcile_profiles.py r04        (reads profiles/<R>_<W>_{kernel_trace_blocking,kernel_trace}.txt, <R>_<W>_bench.json)

Per workload: k_detect avg / median / min of the BLOCKING trace (one pass in flight) vs `roofline.isolated.kernel_ms` of the
unprofiled line; k_detect avg of the PIPELINED trace vs `roofline.kernel_ms`; the blit copies the profiler adds to the pipeline.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = [("cfg2_2msps_fc32", "cfg2 2 Msps complex64 (2^30)"), ("cfg3_8msps_dense_fc32", "cfg3 8 Msps dense (2^28)"),
     ("cfg4_20msps_fc32", "cfg4 20 Msps (2^28)"), ("cfg5_mixed_df_fc32", "cfg5 mixed DF (2^28)"),
     ("fmt_mag2", "\\|IQ\\|² floats (2^30)"), ("fmt_sc16", "int16 (2^30)"), ("fmt_sc8", "int8 (2^30)"), ("fmt_cu8", "uint8, any scale (2^30)"),
     ("fmt_cu8p", "uint8, power-of-two scale (2^30)")]


def rows(path, key):
    for line in open(path):
        f = line.split()
        if len(f) >= 9 and key in f[0]:
            return dict(calls=int(f[1]), total=float(f[2]), avg=float(f[3]), med=float(f[4]), mn=float(f[6]))
    return None


def main(r):
    P = os.path.join(ROOT, "profiles")
    print("| workload | blocking profile: `k_detect` avg / median / min µs | unprofiled `isolated.kernel_ms`, same box | Δ | pipelined profile avg µs "
          "| unprofiled pipelined `kernel_ms` | Δ | blit copies in the pipelined trace (calls, total ms) |")
    print("|---|---|---|---|---|---|---|---|")
    for w, label in W:
        if not os.path.exists(os.path.join(P, "%s_%s_bench.json" % (r, w))):
            continue                                           # (a workload this round's set does not hold)
        b = rows(os.path.join(P, "%s_%s_kernel_trace_blocking.txt" % (r, w)), "k_detect")
        p = rows(os.path.join(P, "%s_%s_kernel_trace.txt" % (r, w)), "k_detect")
        c = rows(os.path.join(P, "%s_%s_kernel_trace.txt" % (r, w)), "copyBuffer")
        d = json.loads(open(os.path.join(P, "%s_%s_bench.json" % (r, w))).readline())
        iso = d["roofline"]["isolated"]["kernel_ms"]
        km = d["roofline"]["kernel_ms"]
        print("| %s | %.1f / %.1f / %.1f | %.4f | %+.1f %% | %.1f | %.4f | %+.1f %% | %d, %.1f |" % (
            label, b["avg"], b["med"], b["mn"], iso, 100 * (b["avg"] / 1e3 / iso - 1), p["avg"], km, 100 * (p["avg"] / 1e3 / km - 1),
            c["calls"] if c else 0, c["total"] / 1e3 if c else 0.0))
    print()
    for w, label in W:
        if not os.path.exists(os.path.join(P, "%s_%s_bench.json" % (r, w))):
            continue                                           # (a workload this round's set does not hold)
        d = json.loads(open(os.path.join(P, "%s_%s_bench.json" % (r, w))).readline())
        ro = d["roofline"]
        print("%-28s value %9.1f Msps  step %.4f ms  frac %.4f live / %.4f isolated" % (w, d["value"], d["ms_per_step"], ro["frac"], ro["isolated"]["frac"]))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r04")
