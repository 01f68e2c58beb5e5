"""Per-slot averages of a rocprofv3 --pmc pass over tools/slot_probe.py: k_detect dispatch i (in dispatch order, from the
context's first call) writes the output buffers of pipeline slot i mod 3.  Counters with one row per hardware instance
(no _sum) are reported as sum and as max/min over the instances (channel skew).
    python tools/slot_pmc_report.py counter_collection.csv [kernel_trace.csv]"""
import csv
import sys
from collections import defaultdict

S = 3
WARM = 6          # slot_probe's warm-up calls: left out of the averages


def main():
    rows = defaultdict(lambda: defaultdict(list))      # dispatch id -> counter -> [values per instance]
    order = {}
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            if "k_detect" not in r["Kernel_Name"]:
                continue
            d = int(r["Dispatch_Id"])
            order.setdefault(d, len(order))
            rows[d][r["Counter_Name"]].append(float(r["Counter_Value"]))
    disp = sorted(rows, key=lambda d: order[d])
    dur = {}
    if len(sys.argv) > 2 and sys.argv[2]:
        try:
            with open(sys.argv[2]) as f:
                for r in csv.DictReader(f):
                    if "k_detect" in r["Kernel_Name"]:
                        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        except (OSError, KeyError, ValueError):
            dur = {}
    counters = sorted({c for d in disp for c in rows[d]})
    print("k_detect dispatches: %d (first %d left out); slot = dispatch index mod %d" % (len(disp), WARM, S))
    if dur:
        per = [[dur[d] for i, d in enumerate(disp) if i >= WARM and i % S == s and d in dur] for s in range(S)]
        print("  %-46s %s" % ("kernel duration under this pass, us", "  ".join("slot%d %10.1f" % (s, sum(v) / max(1, len(v))) for s, v in enumerate(per))))
    for c in counters:
        tot = [[sum(rows[d][c]) for i, d in enumerate(disp) if i >= WARM and i % S == s and c in rows[d]] for s in range(S)]
        line = "  %-46s %s" % (c, "  ".join("slot%d %10.4g" % (s, sum(v) / max(1, len(v))) for s, v in enumerate(tot)))
        ninst = max(len(rows[d][c]) for d in disp)
        if ninst > 1:
            skew = [[max(rows[d][c]) / max(1e-9, min(rows[d][c])) for i, d in enumerate(disp) if i >= WARM and i % S == s and c in rows[d]] for s in range(S)]
            line += "   max/min over %d instances: %s" % (ninst, " ".join("%.3f" % (sum(v) / max(1, len(v))) for v in skew))
        print(line)


if __name__ == "__main__":
    main()
