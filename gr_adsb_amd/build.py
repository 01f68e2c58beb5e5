"""Build libadsb_hip.so (gfx950) in-tree with hipcc.  `python -m gr_adsb_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libadsb_hip.so")
SOURCES = ["adsb_hip.hip"]
DEPS = ["adsb_hip.hip", "adsb_device.h", "adsb_plan.h", os.path.join("..", "..", "include", "adsb_hip.h")]
# -ffp-contract=off: |IQ|^2 must be two rounded products and one rounded add
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(os.path.join(CSRC, d)) <= t for d in DEPS)


RES = os.path.join(HERE, "kernel_resources.json")


def _parse_resources(remarks):
    """-Rpass-analysis=kernel-resource-usage remarks -> {kernel: {vgprs, sgprs, scratch, lds, occupancy, ...}}."""
    import re
    out, cur = {}, None
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "TotalSGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
            "Occupancy [waves/SIMD]": "occupancy_waves_per_simd", "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills",
            "LDS Size [bytes/block]": "lds_bytes_per_block"}
    for line in remarks.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+?): (\d+)", line)
        if m and cur is not None and m.group(1) in keys:
            cur[keys[m.group(1)]] = int(m.group(2))
    return out


def build(force=False, verbose=False):
    """hipcc -> libadsb_hip.so, and beside it kernel_resources.json: the compiler's per-kernel register / LDS / scratch
    report (tests/test_abi.py holds the limits the pipeline relies on: k_detect must leave the tail kernels room)."""
    if not force and up_to_date() and os.path.exists(RES):
        return LIB
    cmd = [hipcc()] + FLAGS + ["-Rpass-analysis=kernel-resource-usage"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise subprocess.CalledProcessError(r.returncode, cmd)
    import json
    res = _parse_resources(r.stdout)
    with open(RES, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
