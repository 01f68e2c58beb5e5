"""ONE process, N contexts, ONE host ring: adsb_process_sharded_multi (ABI 5) through the C ABI on the GPU.

On an 8-GPU node the contexts sit on eight devices; a one-GPU box runs the same driver -- feeder thread per context,
host-fed shard passes ADSB_MAX_IN_FLIGHT deep, seams stitched in stream order on the calling thread -- with 1, 3 and 8
contexts on device 0.  Every result must equal ONE blocking canonical call over the whole buffer byte for byte (and the C
oracle), the case whose every seam falls inside an unbroken chain (each shard takes the fallback) included."""
import numpy as np
import pytest

from helpers import preamble_train_iq, assert_recs_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from gr_adsb_amd import _native
    _native.load()
    return _native


def _clear_head(native, recs):
    r = recs.copy()
    r["flags"] &= np.uint16(~native.BURST_HEAD & 0xFFFF)
    return r


@pytest.mark.parametrize("fs,bps,log2n,seed", [(2e6, 3000, 21, 31), (20e6, 1000, 22, 3), (8e6, 6000, 21, 2), (2e6, -32, 19, 5)])
@pytest.mark.parametrize("n_ctx", [1, 3, 8])
def test_one_process_n_contexts_equal_one_blocking_call(native, fs, bps, log2n, seed, n_ctx):
    from gr_adsb_amd import modulator as M
    from gr_adsb_amd.frontend import MultiDevice
    from oracle import c_oracle as C
    sps = int(fs // 1e6)
    n = (1 << log2n) - 5 * (seed % 3)                       # ragged lengths too
    iq = M.synth_iq(n, fs, bps, seed=seed) if bps > 0 else preamble_train_iq(n, spacing=-bps, sps=sps, seed=seed)
    md = MultiDevice(fs, 0.01, devices=[0] * n_ctx)
    whole = _clear_head(native, md.contexts[0].process_iq(iq))
    assert_recs_equal(whole, C.process_iq(iq, sps, 0.01), "whole")
    assert len(whole) > 3
    ring = md.pinned(n, np.complex64)
    ring.array[:] = iq
    for spc, src in ((1, ring.array), (2, iq), (5, ring.array)):        # page-locked and pageable sources
        got = md.process_host(native.FMT_FC32, src, spc)
        st = md.last_stats
        assert got.tobytes() == whole.tobytes(), "%d contexts x %d shards" % (n_ctx, spc)
        assert st["contexts"] == n_ctx and st["shards"] == n_ctx * spc and len(st["feeder_s"]) == n_ctx and st["wall_s"] > 0
        if bps < 0 and n_ctx * spc > 1:
            # every seam inside the chain: every later shard that owns anything falls back
            owning = 0
            for g in range(n_ctx * spc):
                own_lo, own_hi, _, _ = native.shard_bounds(n, n_ctx * spc, g, sps)
                owning += own_hi > own_lo
            assert owning >= 2 and st["fallbacks"] >= owning - 1
    # offsets shifted like the canonical call's abs_offset; an output array that is too small: grown by the binding
    off = md.process_host(native.FMT_FC32, ring.array, 1, abs_offset=777)
    assert np.array_equal(off["offset"], whole["offset"] + 777) and np.array_equal(off["bits"], whole["bits"])
    small = np.empty(2, dtype=native.BURST_DTYPE)
    assert md.process_host(native.FMT_FC32, ring.array, 2, out=small).tobytes() == whole.tobytes()
    # the contexts stay usable for ordinary calls, in any order
    for cx in reversed(md.contexts):
        assert _clear_head(native, cx.process_iq(iq)).tobytes() == whole.tobytes()
    md.close()


def test_one_process_other_formats_and_refusals(native):
    from gr_adsb_amd import modulator as M
    from gr_adsb_amd.frontend import MultiDevice
    from oracle import adsb_oracle as O
    fs, n = 8e6, (1 << 21) + 40
    iq = M.synth_iq(n, fs, 3000, seed=21)
    q8 = np.clip(np.round(iq.view(np.float32) * 32.0), -127, 127).astype(np.int8)
    u8 = np.clip(np.floor(iq.view(np.float32) * (127.5 / 4.0) + 128.0), 0, 255).astype(np.uint8)
    q16 = np.clip(np.round(iq.view(np.float32) * 8000.0), -32768, 32767).astype(np.int16)
    md = MultiDevice(fs, 0.01, devices=[0, 0, 0], scales={native.FMT_SC8: 1.0 / 32.0, native.FMT_CU8: 4.0 / 255.0, native.FMT_SC16: 1.0 / 8000.0})
    for fmt, data in ((native.FMT_SC8, q8), (native.FMT_CU8, u8), (native.FMT_SC16, q16), (native.FMT_MAG2, O.mag2(iq))):
        whole = _clear_head(native, md.contexts[1].process_format(fmt, data))
        assert len(whole) > 100
        for spc in (1, 3):
            assert md.process_host(fmt, data, spc).tobytes() == whole.tobytes(), (fmt, spc)
    # empty input; mismatched contexts; a context with submissions pending
    assert len(md.process_host(native.FMT_FC32, np.zeros(0, np.complex64), 1)) == 0
    md.contexts[2].set_threshold(0.02)
    with pytest.raises(native.AdsbError) as e:
        md.process_host(native.FMT_FC32, iq, 1)
    assert e.value.code == -22
    md.contexts[2].set_threshold(0.01)
    t = md.contexts[1].submit_format_host(native.FMT_FC32, iq)
    with pytest.raises(native.AdsbError) as e:
        md.process_host(native.FMT_FC32, iq, 1)
    assert e.value.code == -16
    md.contexts[1].wait(t)
    assert md.process_host(native.FMT_FC32, iq, 2).tobytes() == _clear_head(native, md.contexts[0].process_iq(iq)).tobytes()
    md.close()
    cc = native.Context(fs, 0.01, flags=native.FLAG_CONFIDENCE)
    with pytest.raises(native.AdsbError):
        native.process_sharded_multi([cc], native.FMT_FC32, iq, 1)
    cc.close()


def test_sharded_device_driver_waits_for_a_late_producer_on_every_stream(native):
    """Round 5's advisor finding: adsb_process_sharded_device puts its shard passes on three streams; every one of them
    (the fallback's re-runs too) has to wait for the caller's pending event -- here the copy that PRODUCES the tensor, queued
    on torch's stream behind a pile of other work.  And the driver refuses a confidence context (rows would be misaligned)."""
    import torch
    from gr_adsb_amd import modulator as M
    from gr_adsb_amd.frontend import FrontEnd
    fs, n = 2e6, 1 << 22
    for timing in (False, True):
        fe = FrontEnd(fs, 0.01, timing=timing)
        for iq in (M.synth_iq(n, fs, 3000, seed=77), preamble_train_iq(n // 8, spacing=24, sps=2, seed=6)):
            src = torch.from_numpy(np.ascontiguousarray(iq).view(np.float32).reshape(-1, 2).copy()).to("cuda:0")
            whole = _clear_head(native, fe.process_iq_tensor(src))
            junk = torch.randn(1 << 25, device="cuda:0")
            for rep in range(3):
                d = torch.zeros_like(src)
                for _ in range(12):
                    junk = junk * 1.0001 + 0.5
                d.copy_(src)                                   # late producer: still queued when the driver is called
                got = fe.process_sharded_tensor(native.FMT_FC32, d, 7)
                assert got.tobytes() == whole.tobytes(), "timing %s rep %d" % (timing, rep)
        fe.ctx.close()
    fc = FrontEnd(fs, 0.01, flags=native.FLAG_CONFIDENCE)
    with pytest.raises(native.AdsbError):
        fc.process_sharded_tensor(native.FMT_FC32, src, 3)
    # the refused call consumed nothing it should not have: the context still works, pending events can be dropped
    fc.ctx.clear_pending_events()
    assert len(fc.process_iq_tensor(src)) == len(whole)
    fc.ctx.close()
