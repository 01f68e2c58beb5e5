"""A bounded slice of the fuzz campaign (tools/fuzz_gpu.py) inside `-m gpu`: fixed seeds, about twenty seconds, every entry
point -- float |IQ|^2, complex64, int16 / int8 / uint8, canonical calls, block-by-block replay, host-fed submission, the
drop-in blocks under random GNU Radio chunk schedules (paired and unpaired), fused-path confidence ratios, the length-aware
gate -- at rates from 2 to 100 Msps, against the C oracle (and the NumPy oracle for the chunked cases).  The long campaigns
stay outside the suite (profiles/rNN_fuzz_gpu*.txt); this keeps randomised parity in every GPUTEST record."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed0", [5000, 905000])
def test_bounded_fuzz_against_the_oracles(seed0):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_gpu
    res = fuzz_gpu.run(10.0, seed0, max_n=1 << 20)
    assert res["cases"] >= 100 and res["bursts"] > 0, res


def test_bounded_8bit_fuzz_aimed_at_the_narrow_format_kernels():
    """tools/fuzz_sim_8bit.py through the C ABI: few-level noise floors (the exact-hint median), bursts back to back and at the
    start of the stream (short / odd noise windows), int8 and offset-binary uint8 with power-of-two and other scales (all four
    8-bit instances of k_detect), streams up to 3 M samples -- against the C oracle on the oracle's |IQ|^2 of the same bytes."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_sim_8bit
    res = fuzz_sim_8bit.run(10.0, 31337, on_gpu=True)
    assert res["cases"] >= 50 and res["bursts"] > 0 and len(res["per_instance"]) == 4, res
