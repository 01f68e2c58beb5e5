"""packaging/grc/*.block.yml: the GRC descriptors cannot be exercised without GNU Radio, but they
can be kept honest -- valid YAML, and every keyword their `make` template passes exists on the block's constructor."""
import glob
import inspect
import os
import re

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_grc_descriptors_match_the_block_constructors():
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
