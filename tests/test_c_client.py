"""A plain C99 program against include/adsb_hip.h + libadsb_hip.so: the header is valid C (-std=c99 -pedantic
-Wall -Werror), the library links without Python/torch, argument errors and the no-device error surface through
the C ABI (CPU), and on the GPU the C client reproduces the oracle's records byte for byte."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cabi", "c_client.c")
LIBDIR = os.path.join(ROOT, "gr_adsb_amd")


def _build(tmp_path):
    exe = os.path.join(tmp_path, "c_client")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), SRC,
           "-L", LIBDIR, "-ladsb_hip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def _inputs(tmp_path):
    from gr_adsb_amd import modulator as M
    from oracle import c_oracle as C
    iq = M.synth_iq(1 << 18, 2e6, 3000, seed=8, df_choices=(11, 17, 4), df_weights=(0.3, 0.5, 0.2))
    want = C.process_iq(iq, 2, 0.01)
    a, b = os.path.join(tmp_path, "iq.f32"), os.path.join(tmp_path, "want.rec")
    iq.tofile(a)
    want.tofile(b)
    return a, b, len(want)


def test_c_client_compiles_links_and_fails_loudly_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered by the gpu test on a GPU box")
    from gr_adsb_amd import build
    build.build()
    exe = _build(tmp_path)
    a, b, _ = _inputs(tmp_path)
    r = subprocess.run([exe, "2e6", "0.01", a, b], capture_output=True, text=True)
    assert r.returncode == 3 and "ENODEV" in r.stderr          # no device -> error, never a CPU fallback


@pytest.mark.gpu
def test_c_client_matches_oracle_on_gpu(tmp_path):
    exe = _build(tmp_path)
    a, b, n = _inputs(tmp_path)
    assert n > 300
    r = subprocess.run([exe, "2e6", "0.01", a, b], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "%d bursts, identical" % n in r.stdout
