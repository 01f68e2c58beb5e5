"""Launch-by-launch comparison of k_detect durations: HIP events of a plain run, HIP events of a profiled run, and the
profiler's own start/end timestamps of that run (rocprofv3 --kernel-trace -f csv) -- plus, when a counter CSV is given, the
GRBM_GUI_ACTIVE cycles of every launch as a second clock (cycles / duration = the shader clock the launch really ran at).
    python tools/launch_hist_report.py <plain.json> <profiled.json> <kernel_trace.csv> [counter_collection.csv]"""
import csv
import json
import statistics
import sys


def hist(v, lo, hi, nb=12):
    w = (hi - lo) / nb or 1.0
    h = [0] * nb
    for x in v:
        h[min(nb - 1, max(0, int((x - lo) / w)))] += 1
    return " ".join("%3d" % c for c in h)


def describe(name, v):
    if not v:
        print("%-34s (none)" % name)
        return
    s = sorted(v)
    print("%-34s n=%-3d min %.4f  p25 %.4f  median %.4f  p75 %.4f  max %.4f  mean %.4f ms" % (
        name, len(v), s[0], s[len(s) // 4], statistics.median(s), s[(3 * len(s)) // 4], s[-1], sum(v) / len(v)))


def pattern(name, v):
    """lag-1 / lag-2 / lag-3 autocorrelation and the split first-third / rest: every-other-launch shows as lag-1 < 0 < lag-2."""
    if len(v) < 12:
        return
    m = sum(v) / len(v)
    var = sum((x - m) ** 2 for x in v) or 1e-30
    ac = [sum((v[i] - m) * (v[i + k] - m) for i in range(len(v) - k)) / var for k in (1, 2, 3)]
    a, b = v[:len(v) // 3], v[len(v) // 3:]
    print("%-34s autocorrelation lag 1/2/3: %+.2f %+.2f %+.2f   first third mean %.4f, rest %.4f" % (
        name, ac[0], ac[1], ac[2], sum(a) / len(a), sum(b) / len(b)))


def main():
    plain = json.load(open(sys.argv[1]))
    prof = json.load(open(sys.argv[2]))
    rows = []
    with open(sys.argv[3]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Dispatch_Id", "")))
    rows.sort()
    det = [(s, e, d) for s, e, k, d in rows if "k_detect" in k]
    others = [(s, e, k) for s, e, k, d in rows if "k_detect" not in k and "adsb" in k]
    n = len(prof["hip_event_ms"])
    det = det[-n:]                                            # the timed launches are the last n (4 warm-up calls before)
    kt = [(e - s) / 1e6 for s, e, _ in det]
    print("# %s 2^%d samples, %s, %d timed launches" % (plain["format"], plain["log2n"], "single stream" if plain["single_stream"] else "pipelined (3 in flight)", n))
    print("# wall ms/step: plain %.4f   profiled %.4f" % (plain["wall_ms_per_step"], prof["wall_ms_per_step"]))
    describe("plain run, HIP events", plain["hip_event_ms"])
    describe("profiled run, HIP events", prof["hip_event_ms"])
    describe("profiled run, profiler timestamps", kt)
    allv = plain["hip_event_ms"] + prof["hip_event_ms"] + kt
    lo, hi = min(allv), max(allv)
    print("# histograms, 12 bins from %.4f to %.4f ms" % (lo, hi))
    print("plain HIP events      ", hist(plain["hip_event_ms"], lo, hi))
    print("profiled HIP events   ", hist(prof["hip_event_ms"], lo, hi))
    print("profiler timestamps   ", hist(kt, lo, hi))
    pattern("plain HIP events", plain["hip_event_ms"])
    pattern("profiled HIP events", prof["hip_event_ms"])
    pattern("profiler timestamps", kt)
    # what ran while each profiled k_detect ran: tail kernels of the previous pass overlapping it
    ov = []
    for s, e, _ in det:
        ov.append(sum(max(0, min(e, oe) - max(s, os_)) for os_, oe, _ in others) / 1e6)
    if any(ov):
        m = statistics.median(kt)
        slow = [o for o, d in zip(ov, kt) if d > m]
        fast = [o for o, d in zip(ov, kt) if d <= m]
        print("# tail-kernel time overlapping a k_detect launch (ms, summed over kernels): slow half mean %.4f, fast half mean %.4f"
              % (sum(slow) / max(1, len(slow)), sum(fast) / max(1, len(fast))))
    gaps = [(det[i + 1][0] - det[i][1]) / 1e6 for i in range(len(det) - 1)]
    if gaps:
        describe("gap between consecutive k_detect", gaps)
    cyc = {}
    if len(sys.argv) > 4:
        with open(sys.argv[4]) as f:
            for r in csv.DictReader(f):
                if "k_detect" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    cyc[r["Dispatch_Id"]] = cyc.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        pairs = [((e - s) / 1e6, cyc[d]) for s, e, d in det if d in cyc]
        if pairs:
            print("# GRBM_GUI_ACTIVE per launch (all XCDs summed by the tool) vs duration: cycles / ms / 8 XCDs = effective MHz")
            pairs.sort()
            for d, c in pairs[:3] + pairs[len(pairs) // 2 - 1:len(pairs) // 2 + 2] + pairs[-3:]:
                print("   %.4f ms  %.4g cycles  -> %.0f MHz if 8 XCDs, %.0f MHz if 1" % (d, c, c / d / 8e3, c / d / 1e3))
    print("# launch order (ms): plain HIP | profiled HIP | profiler")
    for i in range(n):
        p = plain["hip_event_ms"][i] if i < len(plain["hip_event_ms"]) else float("nan")
        print("%3d  %.4f  %.4f  %.4f" % (i, p, prof["hip_event_ms"][i], kt[i] if i < len(kt) else float("nan")))


if __name__ == "__main__":
    main()
