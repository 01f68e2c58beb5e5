"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files.
    python tools/pmc_summary.py gpurun_out/pmc1/p_counter_collection.csv [...]"""
import csv
import sys
from collections import defaultdict

for path in sys.argv[1:]:
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0][:60]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("#", path)
    for k, cs in acc.items():
        if "adsb" not in k:
            continue
        print(k)
        for c, v in cs.items():
            print("    %-28s n=%-3d avg=%.4g" % (c, len(v), sum(v) / len(v)))
