#!/bin/bash
# round 5, call 8: GPU suite on the final queue policy, pass cost, default bench line
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05/gputest8.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/gputest8.txt
tail -12 gpurun_out/r05/gputest8.txt
timeout 300 python tools/pass_cost.py --fs 20e6 --reps 200 > gpurun_out/r05/pass_cost_final.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05/pass_cost_final.txt
timeout 900 python bench.py > gpurun_out/r05/bench_call8.json 2> gpurun_out/r05/bench_call8.err
tail -3 gpurun_out/r05/bench_call8.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05/bench_call8.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "iso", d["roofline"]["isolated"]["frac"], "untimed", d["timing"].get("ms_per_step_untimed_ctx"))
for e in d.get("extra_configs", []) + d.get("formats", []):
    print(e["name"], e["value"], e["ms_per_step"], "frac", e["roofline"]["frac"], "product", e["product_default"]["ms_per_step"], e["bit_match"]["identical"], e.get("sharded", {}).get("value"))
print({k: v for k, v in d["host_fed"]["formats"]["fc32"].items() if k in ("pinned_vs_plain_h2d",)})
PY
