"""SURVEY.md §8f-1, downstream witness: PDUs built by the drop-in path (gr_adsb_amd.blocks.make_pdu from the
device code's bits, offsets and SNR inputs) are consumed by the UNMODIFIED reference decoder exactly like
the reference front end's own PDUs: same published messages, same aircraft table.  Container only (needs
/root/reference); the device code is executed by the CPU emulator here and by the GPU in test_gpu_parity."""
import os
import sys

import numpy as np
import pytest

REF_OK = os.path.exists("/root/reference/python/adsb/decoder.py")
pytestmark = pytest.mark.skipif(not REF_OK, reason="/root/reference not present on this machine")


def _harness():
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import ref_harness
    return ref_harness


def _decode_all(R, pdus):
    dec = R.load_reference_decoder()
    errors = []
    for i, p in enumerate(pdus):
        try:
            dec.decode_packet(p)
        except Exception as e:        # random payloads reach unfinished branches of the reference decoder
            errors.append((i, type(e).__name__, str(e)))
    dec.msgs.append(("errors", errors, None))
    msgs = [(port, repr(m)) for port, m, _ in dec.msgs]
    return msgs, {k: repr(sorted(v.items())) if isinstance(v, dict) else repr(v) for k, v in dec.plane_dict.items()}


def test_reference_decoder_sees_identical_pdus(monkeypatch):
    import time
    # the reference decoder stamps wall-clock seconds into its aircraft table (decoder.py:424,433,1124): freeze
    # the clock so the two decoder runs below cannot straddle a second boundary
    monkeypatch.setattr(time, "time", lambda: 1.7e9)
    import simlib
    from gr_adsb_amd import blocks, _native
    from gr_adsb_amd import modulator as M
    R = _harness()
    fs, thr, n = 2e6, 0.01, 1 << 16
    iq, truth = M.synth_iq(n, fs, 4000, seed=12, df_choices=(11, 17), df_weights=(0.3, 0.7), return_truth=True)
    x = M.mag2(iq)
    # (a) the reference front end -> PDUs exactly as its demod publishes them
    fr_mod, dm_mod = R.ref_modules()
    ref = R.run_reference(x, fs, thr)
    ref_pdus = [blocks.make_pdu(0.0, fs, int(o), s, b) for o, s, b in zip(ref["pdu_offsets"], ref["pdu_snr"], ref["pdu_bits"])]
    # (b) the device code (emulated) -> PDUs through the drop-in block's own builder
    recs, _ = simlib.sim_canonical(0, iq, fs, thr)
    dem = (recs["flags"] & 1) != 0
    snr = _native.snr_db(recs["peak"], recs["median"])
    bits = _native.unpack_bits(recs["bits"])[:, :112]
    our_pdus = [blocks.make_pdu(0.0, fs, int(o), s, b) for o, s, b in zip(recs["offset"][dem], snr[dem], bits[dem])]
    assert len(our_pdus) == len(ref_pdus) > 50
    m_ref, planes_ref = _decode_all(R, ref_pdus)
    m_our, planes_our = _decode_all(R, our_pdus)
    assert m_our == m_ref
    assert planes_our == planes_ref
    # sanity: the synthetic frames carry valid parity, so the decoder does decode aircraft from them
    assert len(planes_ref) > 10
    assert any(port == "decoded" for port, _ in m_ref)
