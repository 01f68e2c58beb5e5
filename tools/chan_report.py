"""Per-channel TCC requests of k_detect per pipeline slot, from a rocprofv3 JSON with the un-summed counters
(rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ --kernel-trace -f json -- python tools/slot_probe.py --steps 9):
16 channel instances x 8 XCCs per counter and dispatch.  Round 6, the last experiment on the slow pipeline slot.
    python tools/chan_report.py results.json[.gz]"""
import gzip
import json
import sys

import numpy as np

path = sys.argv[1]
d = json.load(gzip.open(path) if path.endswith(".gz") else open(path))
t = d["rocprofiler-sdk-tool"][0]
ks = {k["kernel_id"]: k.get("formatted_kernel_name", k.get("kernel_name", "?")) for k in t["kernel_symbols"]}
cn = {c["id"]["handle"]: c["name"] for c in t["counters"]}
det = [r for r in t["callback_records"]["counter_collection"] if "k_detect" in ks.get(r["dispatch_data"]["dispatch_info"]["kernel_id"], "")]
print("# k_detect dispatches: %d (slot = dispatch index mod 3); per counter: 128 instances = 8 XCC x 16 channels" % len(det))
print("# columns: sum over instances | min / max instance | cv over instances | cv of the per-XCC sums | busiest channel (summed over XCCs) / mean channel")
per_slot = {}
for i, r in enumerate(det):
    vals = {}
    for x in r["records"]:
        vals.setdefault(cn[x["counter_id"]["handle"]], []).append(x["value"])
    dur = (r["dispatch_data"]["end_timestamp"] - r["dispatch_data"]["start_timestamp"]) / 1e3
    per_slot.setdefault(i % 3, []).append(dur)
    out = ["#%2d slot %d %7.1f us" % (i, i % 3, dur)]
    for k, v in sorted(vals.items()):
        v = np.array(v)
        a = v.reshape(-1, 16)
        out.append("%s %.4g | %.4g / %.4g | %.4f | %.4f | %.4f" % (k, v.sum(), v.min(), v.max(), v.std() / v.mean(),
                                                                 a.sum(1).std() / a.sum(1).mean(), a.sum(0).max() / a.sum(0).mean()))
    print("   ".join(out))
print("# k_detect duration per slot, first dispatch of each slot left out: " +
      "  ".join("slot %d %.1f us" % (s, float(np.mean(v[1:]))) for s, v in sorted(per_slot.items())))
