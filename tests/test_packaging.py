"""the drop-in boundary under the reference's own ids.  GNU Radio cannot be installed in the build
image, so the GRC side is exercised the way GRC's generated script would: run the descriptor's `imports`, evaluate its
`make` / `callbacks` templates -- against packaging/gnuradio_adsb loaded as `gnuradio.adsb` under the stub runtime.
When /root/reference is present the templates are the REFERENCE's (read at test time, never copied), and our own
descriptors must agree with them field for field.  The GPU half (blocks really running) is
tests/test_gpu_parity.py::test_gnuradio_adsb_namespace_drives_goldens."""
import glob
import inspect
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import yaml

from helpers import grc_instantiate, load_gnuradio_adsb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_GRC = "/root/reference/grc"
OURS = os.path.join(ROOT, "packaging", "gnuradio_adsb", "grc")


def _load(d, name):
    return yaml.safe_load(open(os.path.join(d, name)))


class _NoGpuContext:
    """Stands in for _native.Context so that the block constructors can run in the GPU-less container."""
    def __init__(self, fs, threshold, device=0, flags=0):
        self.args = (fs, threshold, device, flags)

    def set_threshold(self, t):
        self.thr = t


@pytest.fixture
def adsb(monkeypatch):
    from gr_adsb_amd import _native
    monkeypatch.setattr(_native, "Context", _NoGpuContext)
    return load_gnuradio_adsb()


def test_extension_descriptors_match_the_block_constructors():
    from gr_adsb_amd import blocks
    files = sorted(glob.glob(os.path.join(ROOT, "packaging", "grc", "*.block.yml")))
    assert len(files) == 2
    for f in files:
        d = yaml.safe_load(open(f))
        assert d["file_format"] == 1 and d["templates"]["imports"] == "import gr_adsb_amd.blocks as adsb_hip"
        make = d["templates"]["make"]
        cls = getattr(blocks, re.match(r"adsb_hip\.(\w+)\(", make).group(1))
        sig = inspect.signature(cls.__init__).parameters
        for kw in re.findall(r"(\w+)=\$\{", make):
            assert kw in sig, "%s: constructor has no %s" % (f, kw)
        ids = {p["id"] for p in d["parameters"]}
        assert set(re.findall(r"\$\{(\w+)\}", make)) <= ids
        for cb in d["templates"].get("callbacks", []):
            assert hasattr(cls, cb.split("(")[0])
        assert all(p["dtype"] == "float" for p in d["inputs"])


@pytest.mark.skipif(not os.path.isdir(REF_GRC), reason="/root/reference not present")
@pytest.mark.parametrize("name", ["adsb_framer.block.yml", "adsb_demod.block.yml"])
def test_our_descriptors_equal_the_references_field_for_field(name):
    ref, ours = _load(REF_GRC, name), _load(OURS, name)
    for key in ("id", "templates", "parameters", "inputs", "outputs", "file_format", "label", "category"):
        assert ours[key] == ref[key], "%s: %s differs from the reference's descriptor" % (name, key)


@pytest.mark.parametrize("src", ["reference", "ours"])
def test_grc_templates_instantiate_the_mi355x_blocks(adsb, src):
    d = REF_GRC if src == "reference" else OURS
    if not os.path.isdir(d):
        pytest.skip("/root/reference not present")
    from gr_adsb_amd import blocks
    fr, fr_cb = grc_instantiate(_load(d, "adsb_framer.block.yml"))
    dm, _ = grc_instantiate(_load(d, "adsb_demod.block.yml"))
    assert type(fr) is blocks.framer is adsb.framer and type(dm) is blocks.demod is adsb.demod
    # framer.py:37-65 / demod.py:35-54: names, history, ports, defaults
    assert fr.name() == "ADS-B Framer" and dm.name() == "demod"
    assert fr.fs == 2e6 and fr.threshold == 0.01 and fr.history() == 16 and dm.fs == 2e6
    assert fr._in_sig == [np.float32] and fr._out_sig == [np.float32] and dm._in_sig == [np.float32] and dm._out_sig == [np.float32]
    assert dm._ports == ["demodulated"]
    fr_cb(threshold=0.02)                                   # the GUI callback of examples/adsb_rx.py:210-214
    assert fr.threshold == 0.02
    # other rates through the same template; the constructor asserts like the reference's (framer.py:44)
    fr8, _ = grc_instantiate(_load(d, "adsb_framer.block.yml"), fs=8e6, threshold=0.5)
    assert fr8.sps == 8 and fr8.history() == 64
    with pytest.raises(AssertionError):
        grc_instantiate(_load(d, "adsb_framer.block.yml"), fs=2.5e6)


def test_namespace_exports_and_decoder_placeholder(adsb):
    assert {"framer", "demod", "decoder"} <= set(dir(adsb))                   # python/adsb/__init__.py:24-26
    with pytest.raises(ImportError):                                          # no gr-adsb decoder.py next to the shim here
        adsb.decoder("All Messages", "None", "None")


def test_installer_dry_run(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "packaging", "gnuradio_adsb", "install.py"), "--dry-run",
                        "--python-dir", str(tmp_path / "gnuradio"), "--grc-dir", str(tmp_path / "grc")],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "adsb/__init__.py" in r.stdout and "adsb_framer.block.yml" in r.stdout
    assert not (tmp_path / "gnuradio").exists()
