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
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


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


def build(force=False, verbose=False):
    if not force and up_to_date():
        return LIB
    cmd = [hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
