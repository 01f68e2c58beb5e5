#!/usr/bin/env python
"""Install the MI355X front end under the import path and GRC ids gr-adsb uses.

    python packaging/gnuradio_adsb/install.py [--python-dir DIR] [--grc-dir DIR] [--dry-run]

* <python-dir>/adsb/__init__.py  <- packaging/gnuradio_adsb/adsb/__init__.py   (python-dir defaults to the directory of
  the installed `gnuradio` package; an existing gr-adsb __init__.py is kept as __init__.py.gr-adsb, its decoder.py is
  left where it is and keeps being used)
* <grc-dir>/adsb_framer.block.yml, adsb_demod.block.yml  -- only when gr-adsb's own descriptors are not installed
  there: theirs already name the same ids, import path and constructor calls, and keep working unchanged
* gr_adsb_amd itself must be importable (pip install -e . of this repository, or PYTHONPATH)

Needs GNU Radio, which the build image of this repository does not have: untested there beyond --dry-run.
"""
import argparse
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--python-dir", default=None, help="directory of the `gnuradio` Python package")
    ap.add_argument("--grc-dir", default=None, help="GRC block path (default: ~/.grc_gnuradio)")
    ap.add_argument("--dry-run", action="store_true")
    a = ap.parse_args(argv)
    pydir = a.python_dir
    if pydir is None:
        try:
            import gnuradio
            pydir = os.path.dirname(gnuradio.__file__)
        except ImportError:
            print("gnuradio is not importable: pass --python-dir", file=sys.stderr)
            return 2
    grc = a.grc_dir or os.path.expanduser("~/.grc_gnuradio")
    plan = []
    dst = os.path.join(pydir, "adsb", "__init__.py")
    if os.path.exists(dst) and not os.path.exists(dst + ".gr-adsb"):
        plan.append(("backup", dst, dst + ".gr-adsb"))
    plan.append(("copy", os.path.join(HERE, "adsb", "__init__.py"), dst))
    for f in ("adsb_framer.block.yml", "adsb_demod.block.yml"):
        if not os.path.exists(os.path.join(grc, f)):
            plan.append(("copy", os.path.join(HERE, "grc", f), os.path.join(grc, f)))
    for op, src, d in plan:
        print("%-6s %s -> %s" % (op, src, d))
        if not a.dry_run:
            os.makedirs(os.path.dirname(d), exist_ok=True)
            shutil.copy2(src, d)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
