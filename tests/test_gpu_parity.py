"""Parity tests proper: the HIP path, called through the C ABI (gr_adsb_amd._native -> libadsb_hip.so),
against the golden vectors from the real reference and against the oracle.  Bit-exact everywhere:
offsets, (peak, median) float bits, SNR bits, 112 hard bits, PDU set, confidence bits."""
import warnings

import numpy as np
import pytest

from helpers import (SCHEDULES, Golden, every_byte_pair_stream, preamble_train_iq, assert_recs_equal, assert_recs_match_golden, bulk_golden_names, golden_names,
                     large_golden_names, pathological_names, rate_golden_names, schedules_of, snr_bits, unpack)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from gr_adsb_amd import _native
    _native.load()          # fails loudly if the HIP extension is missing
    return _native


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    return torch


def to_dev(torch, iq):
    return torch.from_numpy(np.ascontiguousarray(iq).view(np.float32).reshape(-1, 2).copy()).to("cuda:0")


@pytest.mark.parametrize("name", golden_names())
def test_canonical_host_iq_matches_reference_goldens(native, name):
    g = Golden(name)
    ctx = native.Context(g.fs, g.thr)
    assert_recs_match_golden(ctx.process_iq(g.iq), g)
    assert_recs_match_golden(ctx.process_mag2(g.x), g)
    st = ctx.stats()
    assert st["calls"] == 2
    ctx.close()


@pytest.mark.parametrize("name", large_golden_names() + rate_golden_names())
def test_large_reference_goldens_every_entry_point(native, torch_mod, name):
    """tests/golden/L*.npz (round 3): 2^20 .. 3*2^20 samples per rate, >= 500 reference tags each (20 Msps included), stored as
    int8 IQ -- so the reference's own outputs pin the cs8 entry point (the bytes as they are), the complex64 and the
    |IQ|^2 entry points (the oracle's exact conversion of the same bytes), host, device-resident, submitted and host-fed.
    tests/golden/R*.npz (round 4): the same at 6 / 10 / 12 / 16 / 24 / 40 / 100 Msps -- the run-time-stride instances
    k_detect<fmt, 0> of three input formats (int8, complex64, |IQ|^2); int16 and uint8 in test_run_time_stride_other_formats."""
    g = Golden(name)
    assert len(g.get("single", "tag_offsets")) >= (490 if name.startswith("L") else 200)
    ctx = native.Context(g.fs, g.thr)
    ctx.set_format_scale(native.FMT_SC8, float(g.scale))
    n = len(g.x)
    assert_recs_match_golden(ctx.process_format(native.FMT_SC8, g.iq8), g)
    assert_recs_match_golden(ctx.process_iq(g.iq), g)
    assert_recs_match_golden(ctx.process_mag2(g.x), g)
    t = torch_mod.from_numpy(g.iq8.copy()).to("cuda:0")
    assert_recs_match_golden(ctx.process_format_device(native.FMT_SC8, t.data_ptr(), n), g)
    assert_recs_match_golden(ctx.wait(ctx.submit_format_device(native.FMT_SC8, t.data_ptr(), n)), g)
    assert_recs_match_golden(ctx.wait(ctx.submit_format_host(native.FMT_MAG2, g.x)), g)
    ctx.close()


def _drive_blocks_vs_golden(g, sched, improved=False):
    from gr_adsb_amd import blocks, grshim
    fr = blocks.framer(g.fs, g.thr)
    dm = blocks.demod(g.fs)
    dm.start_timestamp = 0.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tags, msgs = grshim.drive(fr, dm, g.x, None if sched == "single" else g.sched(sched))
    assert np.array_equal(np.array([t.offset for t in tags], dtype=np.int64), g.get(sched, "tag_offsets"))
    snr = np.array([t.value[1] for t in tags], dtype=np.float32)
    assert np.array_equal(snr.view(np.uint32), g.get(sched, "tag_snr_bits"))
    offs = np.array([int(round(m[0]["timestamp"] * g.fs)) for _, m in msgs], dtype=np.int64)
    assert np.array_equal(offs, g.get(sched, "pdu_offsets"))
    bits = np.array([m[1] for _, m in msgs], dtype=np.uint8).reshape(-1, 112)
    assert np.array_equal(bits, g.pdu_bits(sched))
    psnr = np.array([m[0]["snr"] for _, m in msgs], dtype=np.float32)
    assert np.array_equal(psnr.view(np.uint32), g.get(sched, "pdu_snr_bits"))
    # the framer's two words of cross-call state after the last call (framer.py:54,57)
    assert fr.prev_eob_idx == int(g.get(sched, "final_prev_eob"))
    assert np.float32(fr.prev_in0).view(np.uint32) == g.get(sched, "final_prev_in0_bits")


@pytest.mark.parametrize("name,sched", [(n_, s_) for n_ in large_golden_names() + rate_golden_names()
                                        for s_ in schedules_of(n_) if s_ != "single"])
def test_dropin_blocks_match_large_reference_goldens(native, name, sched):
    """The drop-in blocks, work() call by work() call, against the reference under the deaf-state schedule (fixed 2048,
    framer.py:177-179) and a random 1000-9000 schedule over megasample streams; the rate goldens (R*: k_pass_small<0> for
    calls of up to four units, k_detect<1, 0> + k_tail_small beyond) also under chunks of 150-1500 symbols."""
    _drive_blocks_vs_golden(Golden(name), sched)


@pytest.mark.parametrize("name", rate_golden_names())
def test_run_time_stride_other_formats(native, torch_mod, name):
    """k_detect<int16, 0> and k_detect<uint8, 0> at the rate goldens' rates: the int8 bytes of the vector re-expressed exactly
    in the other two integer wire formats (int16 = the same integers; offset binary u8 = (v + 255) / 2 for odd v, so v is
    made odd first and the reference side is the C oracle on the oracle's conversion of those bytes), plus the paired
    blocks and the stand-alone demod under the stored random schedule."""
    from oracle import adsb_oracle as O
    from oracle import c_oracle as C
    from gr_adsb_amd import blocks, grshim
    g = Golden(name)
    ctx = native.Context(g.fs, g.thr)
    ctx.set_format_scale(native.FMT_SC16, float(g.scale))
    assert_recs_match_golden(ctx.process_format(native.FMT_SC16, g.iq8.astype(np.int16)), g)
    u8 = ((g.iq8.astype(np.int16) | 1) + 255) // 2                     # 2*u8 - 255 = v | 1
    u8 = u8.astype(np.uint8)
    s8 = float(np.float32(g.scale))
    ctx.set_format_scale(native.FMT_CU8, s8)
    want = C.canonical(O.mag2_iq8(u8, s8, True), g.sps, np.float32(g.thr))
    assert len(want) >= 100
    assert_recs_equal(ctx.process_format(native.FMT_CU8, u8), want, name + " cu8")
    ctx.close()
    fr = blocks.framer(g.fs, g.thr)
    dm = blocks.demod(g.fs, framer=fr)
    dm.start_timestamp = 0.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tags, msgs = grshim.drive(fr, dm, g.x, g.sched("randombig"))
    assert np.array_equal(np.array([t.offset for t in tags], dtype=np.int64), g.get("randombig", "tag_offsets"))
    assert np.array_equal(np.array([m[1] for _, m in msgs], dtype=np.uint8).reshape(-1, 112), g.pdu_bits("randombig"))


@pytest.mark.parametrize("name", bulk_golden_names())
def test_bulk_reference_golden(native, torch_mod, name):
    """tests/golden/B*.npz (round 4): 2^28 samples at 2 / 8 / 20 Msps, generated on the device by tests/lcg_stream.py (the same
    integer hashing the build container ran in NumPy: identical bytes), against what the UNMODIFIED reference produced for them
    in one work() call -- 32 k / 43 k / 8 k tags and PDUs (8 and 20 Msps: bursts longer than the LDS window, bits taken tile by
    tile from the pending list).  This is the size at which a pass runs as eight resident rounds of short
    chunks (adsb_plan.h) and the usual-tile instance (EASY) carries all but the first and last tiles: the bulk device path
    itself pinned by the reference, through the int8, complex64 and |IQ|^2 entry points."""
    import lcg_stream
    torch = torch_mod
    g = Golden(name, lazy=True)
    n = g.gen["n"]
    assert n >= 1 << 28
    iq8 = lcg_stream.stream(g.gen, device="cuda:0")
    head = lcg_stream.stream(g.gen, lo=0, hi=1 << 16)                  # the NumPy generator and the device one agree
    assert np.array_equal(iq8[:2 << 16].cpu().numpy(), head)
    ctx = native.Context(g.fs, g.thr)
    ctx.set_format_scale(native.FMT_SC8, float(g.scale))
    assert_recs_match_golden(ctx.process_format_device(native.FMT_SC8, iq8.data_ptr(), n), g)
    st = ctx.stats()
    # (k_detect for the 8-bit formats: one wavefront per workgroup since round 5 -- blocks_per_cu = resident wavefronts per
    # CU, detect_grid = units)
    resident = torch.cuda.get_device_properties(0).multi_processor_count * st["blocks_per_cu"]
    units, per = native.plan_chunks(n, resident)
    assert units == st["detect_grid"]
    assert units * per >= n and units >= 4 * resident, "a bulk pass: several resident rounds of short chunks"
    v = iq8.to(torch.float32) * float(g.scale)                         # f32(int8) * scale: one rounded multiply
    iq = v.view(n, 2).contiguous()
    torch.cuda.synchronize()                                           # (the context runs on its own stream)
    assert_recs_match_golden(ctx.process_iq_device(iq.data_ptr(), n), g)
    prod = iq * iq                                                     # two rounded products, one rounded add
    x = (prod[:, 0] + prod[:, 1]).contiguous()
    del v, prod, iq
    torch.cuda.synchronize()
    assert_recs_match_golden(ctx.process_mag2_device(x.data_ptr(), n), g)
    ctx.close()


@pytest.mark.parametrize("scale", [1.0 / 128.0, 4.0, 2.0 ** -20, 3.0 / 128.0])
def test_int8_every_byte_pair_as_peak_and_as_median(native, scale):
    """int8 IQ, all 65536 (i, q) byte pairs: each once as the four high chips of a preamble (the record's peak) and once as
    a constant 100-sample noise window (its median), 16.7 M samples, against the C oracle on the oracle's conversion
    f32(i8) * scale squared and summed.  Power-of-two scales run k_detect<5, .>, whose |IQ|^2 is the v_dot4c_i32_i8 sum
    accumulated onto the bit pattern of 2^23 and one fused multiply-add (adsb_device.h body_convert); 3/128 the generic
    chain."""
    from oracle import adsb_oracle as O
    from oracle import c_oracle as C
    iq8, thr = every_byte_pair_stream(scale)
    want = C.canonical(O.mag2_iq8(iq8, scale), 2, thr)
    assert len(want) >= 65000
    ctx = native.Context(2e6, float(thr))
    ctx.set_format_scale(native.FMT_SC8, float(scale))
    assert_recs_equal(ctx.process_format(native.FMT_SC8, iq8), want, "every byte pair, scale %g" % scale)
    ctx.close()


@pytest.mark.parametrize("scale", [2.0 ** -8, 2.0 ** -6, 8.0, 3.0 / 256.0])
def test_uint8_every_byte_pair_as_peak_and_as_median(native, scale):
    """The same stream read as offset-binary uint8 IQ: all 65536 (i, q) byte pairs as peak and as median against the C oracle
    on the oracle's conversion f32(2 u8 - 255) * scale.  Power-of-two scales ((u8 - 127.5) / 128 = 2^-8: the RTL-SDR convention)
    run k_detect<6, .>: x.x + x.1 by two v_dot4c_i32_i8 per sample, x = u8 - 128 (adsb_device.h body_convert); 3/256 the generic chain."""
    from oracle import adsb_oracle as O
    from oracle import c_oracle as C
    iq8, _ = every_byte_pair_stream(1.0, quiet=0x8080)
    u8 = iq8.view(np.uint8)
    x = O.mag2_iq8(u8, float(np.float32(scale)), True)
    thr = np.float32(6.0) * np.float32(scale) * np.float32(scale)           # between the resting |IQ|^2 (2 s^2: bytes 127 / 128) and the next (10 s^2)
    want = C.canonical(x, 2, thr)
    assert len(want) >= 60000, len(want)
    ctx = native.Context(2e6, float(thr))
    ctx.set_format_scale(native.FMT_CU8, float(scale))
    assert_recs_equal(ctx.process_format(native.FMT_CU8, u8), want, "every byte pair, offset binary, scale %g" % scale)
    ctx.close()


@pytest.mark.parametrize("sps", [2, 8])
def test_output_stage_with_lists_of_eighty_entries(native, sps):
    """k_detect's LDS output stage (adsb_device.h: Stage; complex64, |IQ|^2 floats, int16) holds sixteen records: a train of
    bare preambles, one every 32 symbols over 2^26 samples, gives every wavefront a list of ~80 (2 Msps) / ~20 (8 Msps)
    entries on five-tile chunks -- the stage is flushed in the middle of the chunk, at 8 Msps with records whose bits are
    still arriving (their second half then goes to global memory directly) -- against the C oracle."""
    from oracle import adsb_oracle as O
    from oracle import c_oracle as C
    n = 1 << 26
    iq = preamble_train_iq(n, sps=sps)
    x = O.mag2(iq)
    want = C.canonical(x, sps, np.float32(0.01))
    assert len(want) >= n // (64 * sps) - 2
    ctx = native.Context(sps * 1e6, 0.01)
    assert_recs_equal(ctx.process_iq(iq), want, "stage, complex64")
    units, per = native.plan_chunks(n, 256 * 5 * 4)
    assert per >= 4 * 1024 and (n // (32 * sps)) / units > 16, "lists longer than the stage"
    assert_recs_equal(ctx.process_mag2(x), want, "stage, |IQ|^2")
    ctx.close()


@pytest.mark.parametrize("name", pathological_names())
def test_pathological_reference_goldens(native, name):
    """tests/golden/P*.npz (round 3): NaN / inf, thresholds <= 0, plateaus over several tiles, streams that start / end
    above the threshold, exact ties, tiny inputs -- the REFERENCE's outputs for them (not only the C oracle's), through the
    canonical |IQ|^2 entry point and through the drop-in blocks under every stored schedule."""
    g = Golden(name)
    ctx = native.Context(g.fs, g.thr)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert_recs_match_golden(ctx.process_mag2(g.x), g)
    ctx.close()
    for sched in schedules_of(name):
        _drive_blocks_vs_golden(g, sched)


@pytest.mark.parametrize("name", golden_names())
def test_canonical_int16_iq_matches_reference_goldens(native, torch_mod, name):
    g = Golden(name)
    ctx = native.Context(g.fs, g.thr)
    ctx.set_iq16_scale(2.0 / 32767.0)
    q = g.z["iq16"]
    assert_recs_match_golden(ctx.process_iq16(q), g)
    t = torch_mod.from_numpy(q.copy()).to("cuda:0")
    assert_recs_match_golden(ctx.process_iq16_device(t.data_ptr(), len(q) // 2), g)
    tk = ctx.submit_iq16_device(t.data_ptr(), len(q) // 2)
    assert_recs_match_golden(ctx.wait(tk), g)


@pytest.mark.parametrize("fmt_name,scale", [("sc8", 1.0 / 128.0), ("sc8", 1.0 / 127.0), ("cu8", 1.0 / 255.0), ("cu8", 1.0 / 256.0)])
@pytest.mark.parametrize("fs", [2e6, 8e6, 6e6])
def test_every_rise_a_tile_can_have_8bit_formats(native, torch_mod, fmt_name, scale, fs):
    """Up to 512 rises per 1024-sample tile -- every rise a tile can have -- with matched preambles all over them, through the
    8-bit formats (one wavefront per k_detect workgroup since round 5; the test was written for a variant whose rise list held
    256 of them and worked a tile off in batches: profiles/r05_ab_25_waves_8bit_rejected.txt): 2^20 samples, blocking and as
    three submissions in flight, against the C oracle on the oracle's conversion of the same bytes."""
    from helpers import rise_storm_iq8
    from oracle import adsb_oracle as O
    from oracle import c_oracle as C
    fmt = native.FMT_SC8 if fmt_name == "sc8" else native.FMT_CU8
    sps = int(fs // 1e6)
    n = (1 << 20) * max(1, sps // 4) + 77          # (the gate holds 63 * sps samples: longer streams at the higher rates)
    # (the blocks' bare preambles are spaced for THIS rate -- sps / 2 samples per chip -- so the one-wavefront path meets matched
    # centres among its rise storm at 6 and 8 Msps too, not only rises)
    q = rise_storm_iq8(n, seed=int(fs // 1e6), offset_binary=fmt_name == "cu8", half=sps // 2)
    x = O.mag2_iq8(q, float(np.float32(scale)), fmt_name == "cu8")
    want = C.canonical(x, sps, np.float32(0.01))
    assert len(want) > 2000, len(want)
    ctx = native.Context(fs, 0.01)
    ctx.set_format_scale(fmt, scale)
    assert_recs_equal(ctx.process_format(fmt, q), want, "rise storm %s %g %g" % (fmt_name, scale, fs))
    t = torch_mod.from_numpy(q.reshape(-1, 2)).to("cuda:0")
    torch_mod.cuda.synchronize()
    tk = [ctx.submit_format_device(fmt, t.data_ptr(), n, 0) for _ in range(3)]
    for k in tk:
        assert_recs_equal(ctx.wait(k), want, "rise storm, submitted")
    ctx.close()


@pytest.mark.parametrize("fs,bps", [(2e6, 2000), (8e6, 6000), (20e6, 2000), (12e6, 3000)])
@pytest.mark.parametrize("fmt", ["sc8", "cu8", "cu8p"])
def test_int8_iq_formats_vs_c_oracle(native, torch_mod, fs, bps, fmt):
    """8-bit IQ ingestion (ADSB_FMT_SC8 / ADSB_FMT_CU8) -- host, device and submitted entry points
    against the oracle fed with the oracle's own exact conversion of the same bytes.  cu8p: offset binary with a
    power-of-two scale, (2 u8 - 255) * 2^-6: the dot-product instance k_detect<6, .> (round 6)."""
    from gr_adsb_amd import modulator as M
    from oracle import adsb_oracle as O
    from oracle import c_oracle as C
    ob = fmt != "sc8"
    f = native.FMT_CU8 if ob else native.FMT_SC8
    n = (1 << 22) + 1234
    q = M.quantize_iq8(M.synth_iq(n, fs, bps, 21, noise_power=3e-3, amp2_range=(0.2, 1.0)), full_scale=4.0, offset_binary=ob)
    scale = float(np.float32({"sc8": 4.0 / 127.0, "cu8": 4.0 / 255.0, "cu8p": 2.0 ** -6}[fmt]))
    ctx = native.Context(fs, 0.03)
    ctx.set_format_scale(f, scale)
    want = C.canonical(O.mag2_iq8(q, scale, ob), int(fs // 1e6), np.float32(0.03))
    assert len(want) > 200
    assert_recs_equal(ctx.process_format(f, q), want, fmt + " host")
    t = torch_mod.from_numpy(q.copy()).to("cuda:0")
    assert_recs_equal(ctx.process_format_device(f, t.data_ptr(), n), want, fmt + " device")
    tk = ctx.submit_format_device(f, t.data_ptr(), n)
    assert_recs_equal(ctx.wait(tk), want, fmt + " submitted")
    with pytest.raises(native.AdsbError):
        ctx.set_format_scale(native.FMT_FC32, 1.0)


def test_int16_iq_large_vs_c_oracle(native, torch_mod):
    from gr_adsb_amd import modulator as M
    from oracle import adsb_oracle as O
    from oracle import c_oracle as C
    n = 1 << 22
    q = M.quantize_iq16(M.synth_iq(n, 8e6, 6000, 9), full_scale=4.0)
    scale = 4.0 / 32767.0
    ctx = native.Context(8e6, 0.01)
    ctx.set_iq16_scale(scale)
    got = ctx.process_iq16(q)
    want = C.canonical(O.mag2_iq16(q, scale), 8, 0.01)
    assert len(want) > 1000
    assert_recs_equal(got, want, "int16 IQ")


@pytest.mark.parametrize("name", golden_names())
@pytest.mark.parametrize("sched", SCHEDULES)
def test_dropin_blocks_match_reference_goldens(native, name, sched):
    """The reference's own block API, driven work() call by work() call with the golden's chunk schedule."""
    from gr_adsb_amd import blocks, grshim
    g = Golden(name)
    fr = blocks.framer(g.fs, g.thr)
    dm = blocks.demod(g.fs)
    dm.start_timestamp = 0.0
    assert fr.name() == "ADS-B Framer" and dm.name() == "demod" and fr.history() == 8 * g.sps
    confs = []                                   # demod.bit_confidence as it stands when each PDU is published
    pub = dm.message_port_pub
    dm.message_port_pub = lambda port, msg: (confs.append(np.array(dm.bit_confidence, dtype=np.float32)), pub(port, msg))[1]
    tags, msgs = grshim.drive(fr, dm, g.x, None if sched == "single" else g.sched(sched))
    assert np.array_equal(np.array([t.offset for t in tags], dtype=np.int64), g.get(sched, "tag_offsets"))
    assert all(t.key == "burst" and t.srcid == "framer" and t.value[0] == "SOB" for t in tags)
    snr = np.array([t.value[1] for t in tags], dtype=np.float32)
    assert np.array_equal(snr.view(np.uint32), g.get(sched, "tag_snr_bits"))
    assert all(port == "demodulated" for port, _ in msgs)
    offs = np.array([int(round(m[0]["timestamp"] * g.fs)) for _, m in msgs], dtype=np.int64)
    assert np.array_equal(offs, g.get(sched, "pdu_offsets"))
    bits = np.array([m[1] for _, m in msgs], dtype=np.uint8).reshape(-1, 112)
    assert np.array_equal(bits, g.pdu_bits(sched))
    psnr = np.array([m[0]["snr"] for _, m in msgs], dtype=np.float32)
    assert np.array_equal(psnr.view(np.uint32), g.get(sched, "pdu_snr_bits"))
    assert all(set(m[0].keys()) == {"timestamp", "snr"} and m[1].dtype == np.uint8 and len(m[1]) == 112 for _, m in msgs)
    # demod.py:101: a PDU's confidence depends on its own samples only, so under any schedule it equals the row the
    # reference produced for the same burst in its single call (the goldens store those)
    single = dict(zip(g.get("single", "pdu_offsets").tolist(), g.get("single", "pdu_conf_bits")))
    assert len(confs) == len(offs)
    for o, cf in zip(offs.tolist(), confs):
        if o in single:
            assert np.array_equal(cf.view(np.uint32), single[o])


@pytest.mark.parametrize("name", golden_names() + large_golden_names()[:2])
def test_paired_demod_publishes_the_framers_slices(native, name):
    """demod(fs, framer=fr): the framer's ONE device pass per chunk also slices the bits (ADSB_FLAG_FRAMER_SLICES), the demod
    publishes them without uploading / scanning the samples again.  Same tags and PDUs as the reference under every stored
    schedule, and -- framer and demod chunked differently -- the same PDUs as the unpaired blocks under the same pair of
    schedules (the drop rule is the demod's own chunk end, demod.py:82)."""
    from gr_adsb_amd import blocks, grshim
    g = Golden(name)
    calls = []
    for sched in schedules_of(name):
        fr = blocks.framer(g.fs, g.thr, min_chunk=64)
        dm = blocks.demod(g.fs, framer=fr)
        assert fr.output_multiple() == 64 and dm.output_multiple() == 1
        dm.start_timestamp = 0.0
        real = dm._ctx.demod_work
        dm._ctx.demod_work = lambda *a, **k: (calls.append(sched), real(*a, **k))[1]
        tags, msgs = grshim.drive(fr, dm, g.x, None if sched == "single" else g.sched(sched))
        assert np.array_equal(np.array([t.offset for t in tags], dtype=np.int64), g.get(sched, "tag_offsets"))
        offs = np.array([int(round(m[0]["timestamp"] * g.fs)) for _, m in msgs], dtype=np.int64)
        assert np.array_equal(offs, g.get(sched, "pdu_offsets"))
        assert np.array_equal(np.array([m[1] for _, m in msgs], dtype=np.uint8).reshape(-1, 112), g.pdu_bits(sched))
        psnr = np.array([m[0]["snr"] for _, m in msgs], dtype=np.float32)
        assert np.array_equal(psnr.view(np.uint32), g.get(sched, "pdu_snr_bits"))
    assert not calls, "same chunking on both blocks: the paired demod never needs the device"
    # different chunking: framer in 4096-sample calls, demod in random ones (and the other way round)
    rng = np.random.default_rng(11)
    n = len(g.x)
    rnd, rem = [], n
    while rem > 0:
        c = int(min(rem, rng.integers(500, 9000)))
        rnd.append(c)
        rem -= c
    fixed = [4096] * (n // 4096) + ([n % 4096] if n % 4096 else [])
    for fs_, ds_ in ((fixed, rnd), (rnd, fixed)):
        out = []
        for paired in (False, True):
            fr = blocks.framer(g.fs, g.thr)
            dm = blocks.demod(g.fs, framer=fr if paired else None)
            dm.start_timestamp = 0.0
            _, msgs = grshim.drive(fr, dm, g.x, fs_, ds_)
            out.append([(int(round(m[0]["timestamp"] * g.fs)), bytes(m[1]), np.float32(m[0]["snr"]).tobytes()) for _, m in msgs])
        assert out[0] == out[1] and len(out[0]) >= 1


def _msgs_vs_golden(g, sched, tags, msgs):
    assert np.array_equal(np.array([t.offset for t in tags], dtype=np.int64), g.get(sched, "tag_offsets"))
    offs = np.array([int(round(m[0]["timestamp"] * g.fs)) for _, m in msgs], dtype=np.int64)
    assert np.array_equal(offs, g.get(sched, "pdu_offsets"))
    assert np.array_equal(np.array([m[1] for _, m in msgs], dtype=np.uint8).reshape(-1, 112), g.pdu_bits(sched))
    psnr = np.array([m[0]["snr"] for _, m in msgs], dtype=np.float32)
    assert np.array_equal(psnr.view(np.uint32), g.get(sched, "pdu_snr_bits"))


@pytest.mark.parametrize("name", ["L2msps_df17", "L8msps_dense", "R6msps"])
@pytest.mark.parametrize("sched", ["fixed2048", "random"])
def test_paired_blocks_on_two_threads(native, name, sched):
    """The paired blocks the way GNU Radio's thread-per-block scheduler runs them (grshim.drive_threaded: framer.work on one
    thread, demod.work on another, a bounded buffer between them, the demod reading the framer's tag list while it grows):
    the framer -> demod slice hand-over (blocks._SliceStore) is shared state of two threads.  Tags and PDUs must equal the
    reference's under the same schedule -- (1) demod right behind the framer: every PDU comes from the framer's slices, the
    demod never goes to the device; (2) demod lagging 300 calls behind a store that keeps only 20 bursts: slices are
    forgotten before they are collected and the demod's device fall-back produces those PDUs -- still the reference's."""
    from gr_adsb_amd import blocks, grshim
    g = Golden(name)
    many = len(g.get(sched, "pdu_offsets")) >= 200      # (8 Msps in 2048-sample calls: nearly every burst is dropped)
    for lag, cap in ((0, None), (300, 20)):
        fr = blocks.framer(g.fs, g.thr)
        dm = blocks.demod(g.fs, framer=fr)
        dm.start_timestamp = 0.0
        if cap:
            fr._slices.cap = cap
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tags, msgs = grshim.drive_threaded(fr, dm, g.x, g.sched(sched), demod_lag=lag)
        _msgs_vs_golden(g, sched, tags, msgs)
        assert len(tags) >= 20
        if cap and many:
            assert fr._slices.evicted > 100 and dm.device_calls > 10, (fr._slices.evicted, dm.device_calls)
        elif not cap:
            assert fr._slices.evicted == 0 and dm.device_calls == 0


def test_paired_blocks_chunk_with_more_bursts_than_the_store_keeps(native, torch_mod):
    """One work() call of 2^25 samples carrying 16 k bursts -- more than the slice store's capacity (set to 8192 here): the
    newest call's slices are never forgotten, however many, so every PDU of the chunk still comes from the framer's pass;
    and with the store emptied behind the framer's back (a demod that was disconnected for a while) the device fall-back
    delivers the same PDUs.  Checked against the NumPy oracle (pinned to the reference, tests/test_oracle_vs_reference.py)."""
    import lcg_stream
    from gr_adsb_amd import blocks, grshim
    from oracle import adsb_oracle as O
    p = lcg_stream.params(1 << 25, 2, 77, 2048)
    iq8 = lcg_stream.stream(p, device="cuda:0").cpu().numpy()
    scale = np.float32(1.0 / 128.0)
    x = O.mag2_iq8(iq8, float(scale), False)
    sched = [1 << 25]
    o = O.run_stream(x, 2e6, 0.005, sched)
    assert len(o["pdu_offsets"]) > 12000
    for sabotage in (False, True):
        fr = blocks.framer(2e6, 0.005)
        dm = blocks.demod(2e6, framer=fr)
        dm.start_timestamp = 0.0
        fr._slices.cap = 8192
        if sabotage:
            fr._slices.put = lambda *a: None                      # nothing is ever stored: every chunk takes the fall-back
        tags, msgs = grshim.drive(fr, dm, x, sched)
        assert np.array_equal(np.array([t.offset for t in tags], dtype=np.int64), o["tag_offsets"])
        offs = np.array([int(round(m[0]["timestamp"] * 2e6)) for _, m in msgs], dtype=np.int64)
        assert np.array_equal(offs, o["pdu_offsets"])
        assert np.array_equal(np.array([m[1] for _, m in msgs], dtype=np.uint8).reshape(-1, 112), o["pdu_bits"])
        assert dm.device_calls == (1 if sabotage else 0) and fr._slices.evicted == 0


@pytest.mark.parametrize("name", golden_names())
@pytest.mark.parametrize("sched", ["fixed4096", "random", "tiny"])
def test_improved_blocks_are_chunk_invariant(native, name, sched):
    """SURVEY.md §8f-4 (opt-in, never the parity mode): framer(improved=True) / demod(improved=True) driven with any
    chunk schedule reproduce what the REFERENCE produces in ONE work() call over the whole stream (the golden
    "single" vectors): no pulses lost at call boundaries, no stale gate state, no bursts dropped at chunk ends."""
    from gr_adsb_amd import blocks, grshim
    g = Golden(name)
    fr = blocks.framer(g.fs, g.thr, improved=True)
    dm = blocks.demod(g.fs, improved=True)
    dm.start_timestamp = 0.0
    F = fr.delay
    assert F == 256 + 121 * g.sps and fr.history() == 100 + 8 * g.sps + 4 + F + 1
    L = len(g.x)
    pad = F + 4096                                   # flush the block's look-ahead
    x = np.concatenate([g.x, np.zeros(pad, np.float32)])
    if sched == "tiny":
        rng = np.random.default_rng(3)
        sch = []
        while sum(sch) < len(x):
            sch.append(int(min(rng.integers(1, 700), len(x) - sum(sch))))
    else:
        sch = g.sched(sched)
        sch = sch + [pad]
    tags, msgs = grshim.drive(fr, dm, x, sch)
    want_off = g.get("single", "tag_offsets")
    assert np.array_equal(np.array([t.value[2] for t in tags], dtype=np.int64), want_off)
    assert np.array_equal(np.array([t.offset for t in tags], dtype=np.int64), want_off + F)       # on the delayed stream
    snr = np.array([t.value[1] for t in tags], dtype=np.float32)
    assert np.array_equal(snr.view(np.uint32), g.get("single", "tag_snr_bits"))
    # PDUs: every burst the single call demodulates, bit for bit; the ones it drops at the end of the stream
    # (eob >= L) are completed here from the zero padding and come on top
    offs = np.array([int(round(m[0]["timestamp"] * g.fs)) for _, m in msgs], dtype=np.int64)
    bits = np.array([m[1] for _, m in msgs], dtype=np.uint8).reshape(-1, 112)
    inside = offs + 119 * g.sps + g.sps // 2 < L
    assert np.array_equal(offs[inside], g.get("single", "pdu_offsets"))
    assert np.array_equal(bits[inside], g.pdu_bits("single"))
    assert np.array_equal(np.sort(offs), offs) and len(offs) == len(want_off)
    psnr = np.array([m[0]["snr"] for _, m in msgs], dtype=np.float32)
    assert np.array_equal(psnr[inside].view(np.uint32), g.get("single", "pdu_snr_bits"))


@pytest.mark.parametrize("fs,bps", [(2e6, 3000), (8e6, 6000)])
def test_long_aware_gate(native, torch_mod, fs, bps):
    """SURVEY.md §8f-4, opt-in ADSB_FLAG_LONG_AWARE_GATE (not the reference's gate): device == the oracle's restatement
    of that rule; shards stitch to the single call; the improved blocks with long_aware reproduce it under a chunk
    schedule; a default context on the same data stays reference-exact."""
    from gr_adsb_amd import blocks, grshim, replay
    from gr_adsb_amd import modulator as M
    from gr_adsb_amd.frontend import shard_plan
    from oracle import adsb_oracle as O
    from oracle import c_oracle as C
    sps, n = int(fs // 1e6), 1 << 21
    iq = M.synth_iq(n, fs, bps, 55, df_choices=(4, 11, 17, 20), df_weights=(0.2, 0.2, 0.4, 0.2))
    x = O.mag2(iq)
    ref = C.canonical(x, sps, np.float32(0.01))
    with C.long_aware_gate():
        want = C.canonical(x, sps, np.float32(0.01))
    assert len(want) < len(ref) and (want["flags"] & native.BURST_LONG_HINT).any()
    ctx = native.Context(fs, 0.01, flags=native.FLAG_LONG_AWARE_GATE)
    got = ctx.process_iq(iq)
    assert_recs_equal(got, want, "long-aware")
    assert np.array_equal(got["flags"] & native.BURST_LONG_HINT, want["flags"] & native.BURST_LONG_HINT)
    t = to_dev(torch_mod, iq)
    assert_recs_equal(ctx.wait(ctx.submit_iq_device(t.data_ptr(), n)), want, "long-aware submitted")
    lists = [ctx.shard_host(native.FMT_FC32, iq[p["lo"]:p["hi"]], p["lo"], p["own_lo"], p["own_hi"], n) for p in shard_plan(n, 7, sps)]
    assert_recs_equal(native.stitch(np.concatenate(lists), sps), want, "long-aware stitch")
    parts = list(replay.replay_blocks(n, sps, 1 << 18, lambda p, hc: ctx.shard_host(native.FMT_FC32, iq[p["lo"]:p["hi"]], p["lo"],
                                                                                  p["own_lo"], p["own_hi"], n, hc)))
    assert_recs_equal(np.concatenate(parts), want, "long-aware replay")
    plain = native.Context(fs, 0.01)
    r0 = plain.process_iq(iq)
    assert_recs_equal(r0, ref, "default context") and None
    assert not (r0["flags"] & native.BURST_LONG_HINT).any()
    # chunk-invariant blocks with the long-aware gate == the oracle's single call under that rule
    L = 1 << 18
    fr = blocks.framer(fs, 0.01, improved=True, long_aware=True)
    dm = blocks.demod(fs, improved=True)
    dm.start_timestamp = 0.0
    pad = fr.delay + 4096
    xx = np.concatenate([x[:L], np.zeros(pad, np.float32)])
    sch = [4096] * (L // 4096) + [pad]
    tags, msgs = grshim.drive(fr, dm, xx, sch)
    with C.long_aware_gate():
        w2 = C.canonical(x[:L], sps, np.float32(0.01))
    # the zero padding completes the bursts the single call drops at the end of its stream: compare the tag set
    assert np.array_equal(np.array([tg.value[2] for tg in tags], dtype=np.int64), w2["offset"])
    with pytest.raises(ValueError):
        blocks.framer(fs, 0.01, long_aware=True)


@pytest.mark.parametrize("name", golden_names())
def test_demod_confidence_bits(native, name):
    """demod.py:97-101 on every golden (2/4/8/20 Msps, the low-SNR DF mix included): the stand-alone demod entry
    returns the exact float32 ratios; 10*log10 of them is the reference's bit_confidence, bit for bit."""
    g = Golden(name)
    ctx = native.Context(g.fs, g.thr)
    offs = g.get("single", "pdu_offsets")
    bits, ok, ratio = ctx.demod_work(g.x, 0, offs, want_ratio=True)
    assert ok.all() and len(offs) > 10
    assert np.array_equal(bits, g.pdu_bits("single"))
    assert np.array_equal(native.confidence_db(ratio).view(np.uint32), g.get("single", "pdu_conf_bits"))
    # a tag in front of the chunk (the reference's get_tags_in_range never hands one over, demod.py:67) is dropped
    bits2, ok2, _ = ctx.demod_work(g.x[1000:], 1000, np.concatenate([[1000 - 8 * g.sps - 1], offs[offs >= 1000]]))
    assert not ok2[0] and ok2[1:].all() and np.array_equal(bits2[1:], g.pdu_bits("single")[offs >= 1000])


@pytest.mark.parametrize("name", golden_names())
@pytest.mark.parametrize("entry", ["iq", "mag2", "iq16"])
def test_fused_path_confidence(native, name, entry):
    """ADSB_FLAG_CONFIDENCE: the whole-buffer (fused) path returns the same ratios for every burst it delivers a PDU
    for -- SURVEY §8b-B3's optional conf_ratio[112] -- checked against the reference's bit_confidence bits."""
    g = Golden(name)
    ctx = native.Context(g.fs, g.thr, flags=native.FLAG_CONFIDENCE)
    if entry == "iq":
        recs = ctx.process_iq(g.iq)
    elif entry == "mag2":
        recs = ctx.process_mag2(g.x)
    else:
        ctx.set_iq16_scale(2.0 / 32767.0)
        recs = ctx.process_iq16(g.z["iq16"])
    assert_recs_match_golden(recs, g)
    ratio = ctx.last_confidence()
    assert ratio.shape == (len(recs), 112)
    dem = (recs["flags"] & 1) != 0
    assert np.array_equal(native.confidence_db(ratio[dem]).view(np.uint32), g.get("single", "pdu_conf_bits"))
    assert not np.any(ratio[~dem].view(np.uint32))                     # rows without a PDU stay zero
    # pipelined form: the ratios follow the ticket
    import torch
    t = torch.from_numpy(np.ascontiguousarray(g.x)).to("cuda:0")
    tk = [ctx.submit_mag2_device(t.data_ptr(), len(g.x)) for _ in range(2)]
    for k in tk:
        r2 = ctx.wait(k)
        assert np.array_equal(r2["offset"], recs["offset"])
        assert np.array_equal(ctx.last_confidence().view(np.uint32), ratio.view(np.uint32))
    plain = native.Context(g.fs, g.thr)
    plain.process_mag2(g.x)
    with pytest.raises(native.AdsbError):
        plain.last_confidence()


def test_set_threshold_takes_effect_next_call(native):
    from oracle import c_oracle as C
    g = Golden("g2msps_mixed_lowsnr")
    ctx = native.Context(g.fs, 0.01)
    a = ctx.process_mag2(g.x)
    ctx.set_threshold(0.004)
    b = ctx.process_mag2(g.x)
    assert_recs_equal(a, C.canonical(g.x, g.sps, 0.01), "thr .01")
    assert_recs_equal(b, C.canonical(g.x, g.sps, 0.004), "thr .004")
    assert len(a) != len(b)


@pytest.mark.parametrize("fs,bps,log2n,seed", [(2e6, 1000, 22, 1), (8e6, 6000, 22, 2), (20e6, 1000, 22, 3), (4e6, 2000, 21, 5)])
def test_device_resident_synthetic_vs_c_oracle(native, torch_mod, fs, bps, log2n, seed):
    """BASELINE.json configs 1-3 at parity size (2^22 samples), device-resident input."""
    from gr_adsb_amd import modulator as M
    from oracle import c_oracle as C
    n = 1 << log2n
    iq = M.synth_iq(n, fs, bps, seed)
    sps = int(fs // 1e6)
    ctx = native.Context(fs, 0.01)
    t = to_dev(torch_mod, iq)
    got = ctx.process_iq_device(t.data_ptr(), n)
    want = C.process_iq(iq, sps, 0.01)
    assert len(want) > 100
    assert_recs_equal(got, want, "fs %g" % fs)
    # a non-zero absolute offset only shifts the tags
    got2 = ctx.process_iq_device(t.data_ptr(), n, abs_offset=123456789012)
    assert np.array_equal(got2["offset"], got["offset"] + 123456789012)


@pytest.mark.parametrize("mode", ["default", "low_latency", "single_stream"])
def test_submit_wait_pipeline_matches_blocking_calls(native, torch_mod, mode):
    """Several passes in flight (adsb_submit_* / adsb_wait) must deliver exactly what the blocking call does -- with the
    tail of a pass kept behind the next pass's k_detect (default), beside it (ADSB_FLAG_LOW_LATENCY), or everything on
    one stream (ADSB_FLAG_SINGLE_STREAM)."""
    from gr_adsb_amd import modulator as M
    n = 1 << 21
    iqs = [M.synth_iq(n, 2e6, 2000, seed) for seed in (11, 12, 13)]
    ts = [to_dev(torch_mod, iq) for iq in iqs]
    ctx = native.Context(2e6, 0.01, flags={"default": 0, "low_latency": native.FLAG_LOW_LATENCY,
                                           "single_stream": native.FLAG_SINGLE_STREAM}[mode])
    want = [ctx.process_iq_device(t.data_ptr(), n, abs_offset=1000 * i) for i, t in enumerate(ts)]
    t0 = ctx.submit_iq_device(ts[0].data_ptr(), n, 0)
    t1 = ctx.submit_iq_device(ts[1].data_ptr(), n, 1000)
    t2 = ctx.submit_iq_device(ts[2].data_ptr(), n, 2000)
    assert native.MAX_IN_FLIGHT == 3
    with pytest.raises(native.AdsbError) as e:
        ctx.submit_iq_device(ts[0].data_ptr(), n, 0)
    assert e.value.code == -16          # -EBUSY: every slot in flight
    with pytest.raises(native.AdsbError):
        ctx.process_iq_device(ts[2].data_ptr(), n)   # blocking calls refuse while calls are pending
    got0 = ctx.wait(t0)
    t3 = ctx.submit_iq_device(ts[0].data_ptr(), n, 0)     # the freed slot is reusable at once
    got1 = ctx.wait(t1)
    got2 = ctx.wait(t2)
    assert ctx.wait(t3).tobytes() == want[0].tobytes()
    for g, w in zip((got0, got1, got2), want):
        assert g.tobytes() == w.tobytes()
    with pytest.raises(native.AdsbError):
        ctx.wait(t2)                     # nothing pending on that ticket any more


def test_mixed_df_low_snr_config(native, torch_mod):
    """BASELINE.json config 5: mixed DF0/4/5/11/16/17 at 3-25 dB over noise 2e-3."""
    from gr_adsb_amd import modulator as M
    from oracle import c_oracle as C
    n = 1 << 22
    iq = M.synth_iq(n, 2e6, 2000, 4, noise_power=2e-3, df_choices=(0, 4, 5, 11, 16, 17),
                    df_weights=(0.27, 0.14, 0.01, 0.45, 0.02, 0.11), snr_db_range=(3, 25))
    ctx = native.Context(2e6, 0.01)
    got = ctx.process_iq(iq)
    assert_recs_equal(got, C.process_iq(iq, 2, 0.01), "mixed")


def test_parity_prefilter_flags_and_block_option(native):
    """the DF / length class / parity verdict attached to every PDU on the device equals the
    oracle's restatement of decoder.py:550-688 (pinned to the reference decoder by tests/golden/g_parity.npz);
    the demod block's opt-in filter drops exactly the PDUs check_parity() rejects without an aircraft table."""
    from gr_adsb_amd import blocks, grshim
    from gr_adsb_amd import modulator as M
    from oracle import adsb_oracle as O
    from oracle import c_oracle as C
    dfs = (0, 4, 5, 11, 16, 17, 18, 19, 20, 21, 24, 7, 28)
    for fs in (2e6, 8e6):
        sps = int(fs // 1e6)
        iq = M.synth_iq(1 << 21, fs, 4000, seed=78, df_choices=dfs, df_weights=[1.0 / len(dfs)] * len(dfs))
        ctx = native.Context(fs, 0.01)
        got = ctx.process_iq(iq)
        want = C.process_iq(iq, sps, 0.01)
        assert_recs_equal(got, want, "parity stream")
        wf, wsyn = C.parity_flags(want)
        assert np.array_equal(got["flags"] & 0x1FE1, wf & 0x1FE1)
        assert len(np.unique(native.burst_df(got))) >= 10 and 50 < native.parity_ok(got).sum() < len(got)
        for i in np.flatnonzero((got["flags"] & 1) != 0)[:200]:
            syn, df, nb = native.mode_s_syndrome(got["bits"][i])
            assert syn == wsyn[i] and df == native.burst_df(got)[i]
        # the stand-alone demod block reports the same verdicts through ok[]
        dem = (got["flags"] & 1) != 0
        x = M.mag2(iq)
        bits, ok, _ = ctx.demod_work(x, 0, got["offset"][dem])
        assert ok.all() and np.array_equal(ctx.last_demod_flags, (got["flags"][dem] & 0xE1).astype(np.uint8))
        # block option: publish only what can still pass check_parity()
        fr, dm = blocks.framer(fs, 0.01), blocks.demod(fs, parity_filter=True)
        dm.start_timestamp = 0.0
        tags, msgs = grshim.drive(fr, dm, x, None)
        p = O.mode_s_parity(unpack(got["bits"][dem]))
        keep = (p["nbits"] != 0) & (~np.isin(p["df"], O.DF_PI) | p["parity_ok"])
        offs = np.array([int(round(m[0]["timestamp"] * fs)) for _, m in msgs], dtype=np.int64)
        assert np.array_equal(offs, got["offset"][dem][keep]) and dm.filtered == int((~keep).sum()) > 0


def test_pathological_inputs(native):
    from gr_adsb_amd import modulator as M
    from oracle import adsb_oracle as O
    from oracle import c_oracle as C
    fs, sps = 2e6, 2
    base = O.mag2(M.synth_iq(1 << 16, fs, 5000, seed=21))
    cases = []
    x = base.copy(); x[10000:13000] = 0.5; cases.append(("long run", x, 0.01))
    x = base.copy(); x[5000:30000] = 0.5; cases.append(("run over several tiles", x, 0.01))
    x = base.copy(); x[:50] = 0.7; x[-40:] = 0.7; cases.append(("starts/ends high", x, 0.01))
    x = base.copy(); x[[100, 5000, 5001, 20000, 40000]] = np.nan; x[[3000, 30000]] = np.inf
    cases.append(("nan/inf", x, 0.01)); cases.append(("nan/inf thr 0", x, 0.0))
    cases.append(("thr 0", base, 0.0)); cases.append(("thr < 0", base, -1.0))
    cases.append(("thr at noise", base, 0.001)); cases.append(("thr below noise", base, 0.0003))
    cases.append(("zeros", np.zeros(5000, np.float32), 0.01)); cases.append(("const high", np.full(9000, 0.3, np.float32), 0.01))
    for n in (1, 5, 15, 16, 17, 100, 239, 240, 241, 300, 4095, 4096, 4097, 4111, 4112):
        cases.append(("n=%d" % n, base[7000:7000 + n].copy(), 0.01))
    ctx = native.Context(fs, 0.01)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for what, x, thr in cases:
            ctx.set_threshold(thr)
            assert_recs_equal(ctx.process_mag2(x), C.canonical(x, sps, thr), what)
    assert ctx.stats()["longrun_calls"] >= 3
    assert len(ctx.process_mag2(np.zeros(0, np.float32))) == 0


def test_dense_bursts_force_record_capacity_regrowth(native):
    from gr_adsb_amd import modulator as M
    from oracle import c_oracle as C
    fs = 2e6
    iq = M.synth_iq(1 << 20, fs, 40000, seed=31, noise_power=1e-4)
    ctx = native.Context(fs, 0.001)
    got = ctx.process_iq(iq)
    assert_recs_equal(got, C.process_iq(iq, 2, 0.001), "dense")


@pytest.mark.parametrize("fs,bps,shards", [(2e6, 3000, 8), (20e6, 2000, 8), (8e6, 6000, 3)])
def test_overlapped_shards_stitch_equals_single_call(native, torch_mod, fs, bps, shards):
    """BASELINE.json config 4: the stream tiled as overlapped time shards, host stitch, bit-exact."""
    from gr_adsb_amd import modulator as M
    from gr_adsb_amd.frontend import FrontEnd, shard_plan
    from oracle import c_oracle as C
    n = 1 << 21
    iq = M.synth_iq(n, fs, bps, seed=8)
    sps = int(fs // 1e6)
    fe = FrontEnd(fs, 0.01)
    t = to_dev(torch_mod, iq)
    lists = []
    for p in shard_plan(n, shards, sps):
        lists.append(fe.shard_tensor(t[p["lo"]:p["hi"]].contiguous(), p["lo"], p["own_lo"], p["own_hi"], n))
    got = fe.stitch(lists)
    assert_recs_equal(got, C.process_iq(iq, sps, 0.01), "stitched")
    whole = fe.process_iq_tensor(t)
    assert_recs_equal(got, whole, "stitched vs whole")
    # gated shards + 8-byte tail exchange + host fix-up of each shard's head (what bench.py --gpus N runs)
    from gr_adsb_amd import sharding
    plans = shard_plan(n, shards, sps)
    bufs = [t[p["lo"]:p["hi"]].contiguous() for p in plans]
    tickets, parts = [], []
    for p, b in zip(plans, bufs):       # two in flight, collected in order
        tickets.append(fe.submit_shard_tensor(b, p["lo"], p["own_lo"], p["own_hi"], n, head_cands=sharding.HEAD_CANDS))
        if len(tickets) == 2:
            parts.append(fe.wait(tickets.pop(0)))
    while tickets:
        parts.append(fe.wait(tickets.pop(0)))
    tails = [native.shard_tail(r, sps) for r in parts]
    outs = [native.shard_fixup(r, sps, sharding.incoming_eob(tails, g)) for g, r in enumerate(parts)]
    assert all(o is not None for o in outs)
    fixed = np.concatenate(outs)
    assert np.array_equal(fixed["offset"], whole["offset"]) and np.array_equal(fixed["bits"], whole["bits"])
    assert np.array_equal(fixed["median"].view(np.uint32), whole["median"].view(np.uint32))
    assert np.array_equal(fixed["flags"] & 3, whole["flags"] & 3)


@pytest.mark.parametrize("fs,bps,log2n,shards,seed", [(20e6, 2000, 23, 8, 8), (2e6, 3000, 21, 8, 9), (8e6, 6000, 21, 3, 10),
                                                      (2e6, -32, 20, 3, 11), (8e6, -16, 19, 4, 14), (2e6, 1000, 14, 5, 12), (20e6, 1000, 22, 1, 13)])
def test_sharded_driver_in_c_equals_single_call(native, torch_mod, fs, bps, log2n, shards, seed):
    """adsb_process_sharded_device (round 5): BASELINE config 4's decomposition with the whole submit / wait / fix-up loop in
    C.  Bit-identical to one canonical call and to the C oracle.  bps < 0: a train of bare preambles every -bps symbols -- ONE
    unbroken chain of centres inside each other's gate over the whole stream, thousands per shard -- so every shard takes the
    fallback (the largest head, then ungated + the plain greedy gate)."""
    from gr_adsb_amd import modulator as M
    from gr_adsb_amd.frontend import FrontEnd
    from oracle import c_oracle as C
    n = (1 << log2n) - 3 * (seed % 4)             # ragged lengths too
    thr = 0.01
    sps = int(fs // 1e6)
    iq = M.synth_iq(n, fs, bps, seed=seed) if bps > 0 else preamble_train_iq(n, spacing=-bps, sps=sps, seed=seed)
    fe = FrontEnd(fs, thr)
    t = to_dev(torch_mod, iq)
    whole = fe.process_iq_tensor(t)
    assert_recs_equal(whole, C.process_iq(iq, sps, thr), "whole")
    assert len(whole) > 3
    got = fe.process_sharded_tensor(native.FMT_FC32, t, shards)
    assert np.array_equal(got["flags"] & native.BURST_HEAD, np.zeros(len(got), np.uint16))
    assert_recs_equal(got, whole, "sharded in C")
    assert np.array_equal(got["flags"] & 0x1FE3, whole["flags"] & 0x1FE3)      # KEPT, DEMOD and the pre-filter bits
    if bps < 0:
        assert fe.stats()["shard_fallbacks"] >= shards - 1        # (the first shard's incoming state is "none": it may pass)
    # an output array that is too small: -ENOSPC with the number needed, then the binding grows it
    small = np.empty(2, dtype=native.BURST_DTYPE)
    again = fe.process_sharded_tensor(native.FMT_FC32, t, shards, out=small)
    assert again.tobytes() == got.tobytes()
    # stream offsets shifted like the canonical call's abs_offset; the context stays usable for ordinary calls
    off = fe.process_sharded_tensor(native.FMT_FC32, t, shards, abs_offset=12345)
    assert np.array_equal(off["offset"], whole["offset"] + 12345) and np.array_equal(off["bits"], whole["bits"])
    assert fe.process_iq_tensor(t).tobytes() == whole.tobytes()


def test_sharded_driver_other_formats(native, torch_mod):
    """The same through the int8 and |IQ|^2 entry formats (buffer starts of the shards on 16-byte boundaries of 2- and
    4-byte samples)."""
    from gr_adsb_amd import modulator as M
    from gr_adsb_amd.frontend import FrontEnd
    from oracle import adsb_oracle as O
    fs, n = 8e6, (1 << 21) + 40
    iq = M.synth_iq(n, fs, 3000, seed=21)
    fe = FrontEnd(fs, 0.01)
    x = torch_mod.from_numpy(O.mag2(iq)).to("cuda:0")
    assert fe.process_sharded_tensor(native.FMT_MAG2, x, 7).tobytes() == _clear_head(native, fe.process_mag2_tensor(x)).tobytes()
    q = np.clip(np.round(iq.view(np.float32) * 32.0), -127, 127).astype(np.int8).reshape(-1, 2)
    fe.ctx.set_format_scale(native.FMT_SC8, 1.0 / 32.0)
    t8 = torch_mod.from_numpy(q).to("cuda:0")
    whole = fe.process_format_tensor(native.FMT_SC8, t8)
    assert len(whole) > 100
    assert fe.process_sharded_tensor(native.FMT_SC8, t8, 5).tobytes() == _clear_head(native, whole).tobytes()


def _clear_head(native, recs):
    r = recs.copy()
    r["flags"] &= np.uint16(~native.BURST_HEAD & 0xFFFF)
    return r


def test_mid_size_pass_records_arrive_without_a_copy_or_with_one(native, torch_mod):
    """Round 5: a pass of up to 2^26 samples gets its first 32768 records stored straight into the pinned result buffer by
    k_compact (beside the device copy); adsb_wait copies only when the pass delivered more.  Both sides of that boundary,
    blocking and submitted three deep, against the C oracle."""
    from gr_adsb_amd import modulator as M
    from gr_adsb_amd.frontend import FrontEnd
    from oracle import c_oracle as C
    fs, sps = 2e6, 2
    n = 1 << 23                                             # 4.2 s: ~3.7 k bursts at 1 k/s; a preamble train keeps one per 128 samples
    thr = 0.01
    for bps, lo, hi in ((1000, 2000, 32768), (-32, 32769, 1 << 30)):
        iq = M.synth_iq(n, fs, bps, seed=5) if bps > 0 else preamble_train_iq(n, spacing=-bps, sps=sps, seed=5)
        ref = C.process_iq(iq, sps, thr)
        assert lo <= len(ref) <= hi, len(ref)
        fe = FrontEnd(fs, thr)
        t = to_dev(torch_mod, iq)
        assert_recs_equal(fe.process_iq_tensor(t), ref, "blocking %d" % bps)
        tickets = [fe.submit_iq_tensor(t) for _ in range(native.MAX_IN_FLIGHT)]
        for k in tickets:
            assert_recs_equal(fe.wait(k), ref, "submitted %d" % bps)


def test_full_size_properties(native, torch_mod):
    """At the bench size (2^28 samples, generated in HBM) the oracle is too slow to run on everything, so use
    size-independent properties: (1) the whole-buffer result restricted to a window equals the C oracle on
    that window away from the window edges, for several windows incl. the two ends; (2) whole == stitched
    shards; (3) offsets strictly increasing and spaced by more than 63*sps; (4) a checksum of the result is
    reproducible across calls."""
    torch = torch_mod
    from gr_adsb_amd import modulator as M
    from gr_adsb_amd.frontend import FrontEnd, shard_plan
    from oracle import c_oracle as C
    fs, sps, n = 2e6, 2, 1 << 28
    iq = M.synth_iq_torch(n, fs, 1000, 1, torch.device("cuda:0"))
    fe = FrontEnd(fs, 0.01)
    whole = fe.process_iq_tensor(iq)            # (FrontEnd waits for torch's stream: the generator's kernels produce iq)
    assert 100000 < len(whole) < 140000      # 134 s of signal at ~1000 bursts/s, some lost to collisions
    d = np.diff(whole["offset"])
    assert d.min() > 63 * sps
    again = fe.process_iq_tensor(iq)
    assert whole.tobytes() == again.tobytes()
    w = 1 << 22
    for start in (0, n // 3, n // 2 + 12345 * 4, n - w):
        host = iq[start:start + w].cpu().numpy().view(np.complex64).reshape(-1)
        ref = C.process_iq(host, sps, 0.01, abs_offset=start)
        lo = start + (2000 if start > 0 else 0)
        hi = start + w - 2000 if start + w < n else n
        a = whole[(whole["offset"] >= lo) & (whole["offset"] < hi)]
        b = ref[(ref["offset"] >= lo) & (ref["offset"] < hi)]
        # the gate state entering the window may differ; once both lists contain the same offset the
        # state is identical from there on, so everything from the first common offset must agree
        common = np.intersect1d(a["offset"][:8], b["offset"][:8])
        assert len(common), "no common burst near the window start"
        c0 = common[0]
        assert_recs_equal(a[a["offset"] >= c0], b[b["offset"] >= c0], "window @%d" % start)
        assert (a["offset"] >= c0).sum() > 1000
    lists = [fe.shard_tensor(iq[p["lo"]:p["hi"]], p["lo"], p["own_lo"], p["own_hi"], n) for p in shard_plan(n, 8, sps)]
    assert fe.stitch(lists).tobytes() == whole.tobytes()


def test_more_than_2_31_samples(native, torch_mod):
    """Maximum-size edge: one canonical call over 2^31 + 2^20 + 37 samples (int8 IQ, 4.3 GB in HBM), so every
    64-bit index path is exercised past the int32 range.  Checked by size-independent properties: windows
    (start, across the 2^31 boundary, ragged end) equal the C oracle on the same bytes; offsets increasing and
    gated; shards stitched == whole."""
    torch = torch_mod
    from gr_adsb_amd import modulator as M
    from gr_adsb_amd.frontend import FrontEnd, shard_plan
    from oracle import adsb_oracle as O
    from oracle import c_oracle as C
    fs, sps = 2e6, 2
    nb = 1 << 24
    block = M.quantize_iq8(M.synth_iq(nb, fs, 2000, 41, noise_power=3e-3, amp2_range=(0.2, 1.0)), full_scale=4.0)
    n = (1 << 31) + (1 << 20) + 37
    reps = n // nb + 1
    dev = torch.from_numpy(block.copy()).to("cuda:0").view(nb, 2)
    big = dev.repeat(reps, 1)[:n].contiguous()
    assert big.shape[0] == n and big.dtype == torch.int8
    scale = float(np.float32(4.0 / 127.0))
    fe = FrontEnd(fs, 0.03)
    fe.ctx.set_format_scale(native.FMT_SC8, scale)
    whole = fe.process_format_tensor(native.FMT_SC8, big)
    assert len(whole) > 1_000_000 and whole["offset"][-1] > (1 << 31)
    assert np.diff(whole["offset"]).min() > 63 * sps
    w = 1 << 21
    for start in (0, (1 << 31) - w // 2, n - w):
        host = big[start:start + w].cpu().numpy().reshape(-1)
        ref = C.canonical(O.mag2_iq8(host, scale), sps, np.float32(0.03), abs_offset=start)
        lo = start + (2000 if start > 0 else 0)
        hi = start + w - 2000 if start + w < n else n
        a = whole[(whole["offset"] >= lo) & (whole["offset"] < hi)]
        b = ref[(ref["offset"] >= lo) & (ref["offset"] < hi)]
        common = np.intersect1d(a["offset"][:8], b["offset"][:8])
        assert len(common), "no common burst near the window start"
        c0 = common[0]
        assert_recs_equal(a[a["offset"] >= c0], b[b["offset"] >= c0], "window @%d" % start)
        assert (a["offset"] >= c0).sum() > 500
    lists = [fe.shard_tensor(big[p["lo"]:p["hi"]], p["lo"], p["own_lo"], p["own_hi"], n, fmt=native.FMT_SC8)
             for p in shard_plan(n, 3, sps)]
    assert fe.stitch(lists).tobytes() == whole.tobytes()


@pytest.mark.parametrize("fmt", ["fc32", "sc16", "sc8", "cu8"])
def test_file_replay_equals_one_canonical_call(native, tmp_path, fmt):
    """SURVEY.md §8f-3 file-source framing: a raw IQ recording replayed block by block (gr_adsb_amd.replay: overlapped
    shards on one GPU, end-of-burst state carried on the host) == the oracle's single canonical call over the
    whole file; the CLI records the same PDUs into SQLite."""
    import os
    from gr_adsb_amd import modulator as M, pdu_store, replay
    from oracle import adsb_oracle as O
    from oracle import c_oracle as C
    fs, sps, n = 4e6, 4, (1 << 21) + 4321
    iq = M.synth_iq(n, fs, 4000, 23, noise_power=3e-3, amp2_range=(0.2, 1.0))
    scale = None
    if fmt == "fc32":
        raw, x = iq, O.mag2(iq)
    elif fmt == "sc16":
        raw = M.quantize_iq16(iq, full_scale=4.0); scale = float(np.float32(4.0 / 32767.0)); x = O.mag2_iq16(raw, scale)
    else:
        ob = fmt == "cu8"
        raw = M.quantize_iq8(iq, full_scale=4.0, offset_binary=ob)
        scale = float(np.float32(4.0 / 255.0 if ob else 4.0 / 127.0)); x = O.mag2_iq8(raw, scale, ob)
    path = os.path.join(tmp_path, "capture." + fmt)
    raw.tofile(path)
    want = C.canonical(x, sps, np.float32(0.03))
    assert len(want) > 500
    rp = replay.FileReplay(path, fmt, fs, 0.03, block_samples=1 << 18, scale=scale)
    parts = list(rp)
    assert len(parts) == 9
    assert_recs_equal(np.concatenate(parts), want, "replay " + fmt)
    db = os.path.join(tmp_path, "pdus.db")
    argv = [path, "--format", fmt, "--fs", str(fs), "--threshold", "0.03", "--block-log2", "19", "--sqlite", db]
    if scale is not None:
        argv += ["--scale", repr(scale)]
    assert replay.main(argv) == 0
    pdus = pdu_store.read_pdus(db)
    dem = (want["flags"] & 1) != 0
    assert len(pdus) == dem.sum()
    assert np.array_equal(np.array([v for _, v in pdus]), unpack(want["bits"][dem]))
    # empty recording
    open(os.path.join(tmp_path, "empty.bin"), "wb").close()
    assert len(replay.FileReplay(os.path.join(tmp_path, "empty.bin"), fmt, fs, 0.03).all()) == 0


def test_caller_stream_and_reset(native, torch_mod):
    """adsb_set_stream: the pipeline runs on the caller's HIP stream, so input produced by kernels queued on that
    stream needs no host synchronisation.  adsb_reset: the GNU Radio emulation state (framer.py:54,57) starts over."""
    torch = torch_mod
    from gr_adsb_amd import modulator as M
    from gr_adsb_amd.frontend import FrontEnd
    from oracle import adsb_oracle as O
    from oracle import c_oracle as C
    fs, sps, n = 2e6, 2, 1 << 22
    iq = M.synth_iq(n, fs, 3000, 77)
    want = C.process_iq(iq, sps, 0.01)
    fe = FrontEnd(fs, 0.01)
    st = torch.cuda.Stream()
    fe.use_torch_stream(st)
    base = to_dev(torch, iq)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        for k in range(3):
            t = (base * 3.0) / 3.0 if k else base + 0.0       # produced on `st` right before the call, no sync
            got = fe.process_iq_tensor(t)
            ref = want if k == 0 else C.process_iq(t.cpu().numpy().view(np.complex64).reshape(-1), sps, 0.01)
            assert_recs_equal(got, ref, "caller stream pass %d" % k)
    with pytest.raises(native.AdsbError):
        tk = fe.submit_iq_tensor(base)
        try:
            fe.use_torch_stream(st)                           # -EBUSY while a submitted call is pending
        finally:
            fe.wait(tk)
    # reset: two identical framer.work() sequences give identical tags only if the state starts over
    x = O.mag2(iq[:1 << 16])
    ctx = native.Context(fs, 0.01)
    H = 8 * sps

    def run():
        tags = []
        pad = np.concatenate([np.zeros(H - 1, np.float32), x])
        pos = 0
        for N in (4096,) * 16:
            tags.append(ctx.framer_work(pad[pos:pos + N + H - 1], N, pos)["offset"].copy())
            pos += N
        return np.concatenate(tags)
    a = run()
    ctx.reset()
    b = run()
    o = O.run_stream(x, fs, 0.01, [4096] * 16)
    assert np.array_equal(a, o["tag_offsets"]) and np.array_equal(b, a) and len(a) > 50


def test_contexts_are_independent_across_threads(native):
    """a context is single-threaded, different contexts run concurrently (one per GR block / per
    stream).  Six host threads, each with its own context, sample rate and data, hammering the library at once."""
    from concurrent.futures import ThreadPoolExecutor
    from gr_adsb_amd import modulator as M
    from oracle import c_oracle as C
    cases = [(2e6, 3000, 1), (4e6, 3000, 2), (8e6, 6000, 3), (20e6, 2000, 4), (2e6, 20000, 5), (8e6, 500, 6)]
    data = [(fs, M.synth_iq(1 << 19, fs, bps, seed)) for fs, bps, seed in cases]
    want = [C.process_iq(iq, int(fs // 1e6), 0.01) for fs, iq in data]

    def worker(i):
        fs, iq = data[i]
        ctx = native.Context(fs, 0.01)
        outs = [ctx.process_iq(iq) for _ in range(20)]
        x = M.mag2(iq)
        outs.append(ctx.process_mag2(x))
        return outs

    with ThreadPoolExecutor(max_workers=len(cases)) as ex:
        res = list(ex.map(worker, range(len(cases))))
    for i, outs in enumerate(res):
        assert len(want[i]) > 10
        for o in outs:
            assert_recs_equal(o, want[i], "thread %d" % i)


def test_lifecycle_stress(native, torch_mod):
    """Create/destroy many contexts, sizes growing and shrinking, every entry point interleaved on one context:
    results stay exact and device memory returns to where it started."""
    torch = torch_mod
    from gr_adsb_amd import modulator as M
    from oracle import adsb_oracle as O
    from oracle import c_oracle as C
    fs, sps = 2e6, 2
    big = M.synth_iq(1 << 21, fs, 4000, 61)
    sizes = [1 << 21, 1000, 1 << 18, 77, 1 << 20, 5, 1 << 21, 1 << 12]
    want = {n: C.process_iq(big[:n], sps, 0.01) for n in sizes}
    free = []
    for rep in range(31):
        ctx = native.Context(fs, 0.01)
        for n in (sizes if rep % 5 == 0 else sizes[rep % len(sizes):][:2]):
            assert_recs_equal(ctx.process_iq(big[:n]), want[n], "rep %d n %d" % (rep, n))
        ctx.close()
        del ctx
        torch.cuda.synchronize()
        free.append(torch.cuda.mem_get_info()[0])
    # after the first lifecycles (runtime pools, code objects) free device memory must not keep shrinking
    assert free[10] - free[30] < (16 << 20), [(f - free[0]) >> 20 for f in free]
    # one context, every kind of call interleaved
    ctx = native.Context(fs, 0.01)
    dev = to_dev(torch, big)
    x = O.mag2(big[:1 << 16])
    o = O.run_stream(x, fs, 0.01, [8192] * 8)
    H = 8 * sps
    pad = np.concatenate([np.zeros(H - 1, np.float32), x])
    tags, pos = [], 0
    for k in range(8):
        tags.append(ctx.framer_work(pad[pos:pos + 8192 + H - 1], 8192, pos)["offset"].copy())      # stateful GR call ...
        pos += 8192
        n = sizes[k]
        assert_recs_equal(ctx.process_iq(big[:n]), want[n], "interleaved host")                     # ... canonical calls between
        t = ctx.submit_iq_device(dev.data_ptr(), 1 << 21)
        bits, ok, _ = ctx.demod_work(x, 0, o["pdu_offsets"][:10])
        assert ok.all() and np.array_equal(bits, o["pdu_bits"][:10])
        assert_recs_equal(ctx.wait(t), want[1 << 21], "interleaved submit")
    assert np.array_equal(np.concatenate(tags), o["tag_offsets"])              # the framer state survived all of it
    with pytest.raises(native.AdsbError):
        ctx.wait(0)                                                            # nothing pending


def test_adversarial_streams(native):
    """The seam-hunting streams of test_sim_property.py (plateaus and bursts planted on tile / window
    boundaries, exact ties, thresholds on sample values, NaNs) through the real kernels."""
    from test_sim_property import adversarial_stream
    from oracle import c_oracle as C
    ctxs = {}
    checked = 0
    for seed in range(120):
        rng = np.random.default_rng(seed)
        n = int(rng.choice([1, 17, 240, 4095, 4096, 4097, 4111, 4352, 8191, 8192, 8193, 12288, 20000, 70000]))
        sps = int(rng.choice([2, 4, 8, 20]))
        thr = float(rng.choice([0.01, 0.0099, 0.0101, 0.004, 0.05]))
        x = adversarial_stream(rng, n, sps)
        ctx = ctxs.setdefault(sps, native.Context(sps * 1e6, thr))
        ctx.set_threshold(thr)
        want = C.canonical(x, sps, thr)
        assert_recs_equal(ctx.process_mag2(x), want, "seed %d" % seed)
        checked += len(want)
    assert checked > 100


@pytest.mark.parametrize("name", ["g2msps_df17", "g8msps_dense"])
@pytest.mark.parametrize("sched", ["single", "random"])
def test_gnuradio_adsb_namespace_drives_goldens(native, name, sched):
    """the blocks instantiated the way GRC does -- `import gnuradio.adsb as adsb`, then the descriptor's
    make template `adsb.framer(${fs}, ${threshold})` / `adsb.demod(${fs})` with the ids, parameters and callbacks of the
    reference's descriptors (packaging/gnuradio_adsb/grc; tests/test_packaging.py proves them equal to the
    reference's) -- produce the reference's tags and PDUs."""
    import os
    import yaml
    from gr_adsb_amd import grshim
    from helpers import grc_instantiate, load_gnuradio_adsb
    load_gnuradio_adsb()
    g = Golden(name)
    grc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "packaging", "gnuradio_adsb", "grc")
    fr, fr_cb = grc_instantiate(yaml.safe_load(open(os.path.join(grc, "adsb_framer.block.yml"))), fs=g.fs, threshold=0.5)
    dm, _ = grc_instantiate(yaml.safe_load(open(os.path.join(grc, "adsb_demod.block.yml"))), fs=g.fs)
    fr_cb(threshold=g.thr)                           # the descriptor's set_threshold callback, before the first work()
    dm.start_timestamp = 0.0
    tags, msgs = grshim.drive(fr, dm, g.x, None if sched == "single" else g.sched(sched))
    assert np.array_equal(np.array([t.offset for t in tags], dtype=np.int64), g.get(sched, "tag_offsets"))
    snr = np.array([t.value[1] for t in tags], dtype=np.float32)
    assert np.array_equal(snr.view(np.uint32), g.get(sched, "tag_snr_bits"))
    bits = np.array([m[1] for _, m in msgs], dtype=np.uint8).reshape(-1, 112)
    assert np.array_equal(bits, g.pdu_bits(sched))


def test_host_fed_pipeline_equals_blocking_calls(native):
    """adsb_submit_format_host: chunks in host memory (page-locked: DMA'd where they lie; pageable: through the
    pinned chunk ring, > 16 MiB so that the ring wraps), three in flight, == the blocking entry points == the oracle."""
    from gr_adsb_amd import modulator as M
    from oracle import adsb_oracle as O
    from oracle import c_oracle as C
    fs, n = 2e6, (1 << 22) + 12345
    chunks = [M.synth_iq(n, fs, 3000, seed=70 + k) for k in range(4)]
    want = [C.canonical(O.mag2(c), 2, np.float32(0.01)) for c in chunks]
    ctx = native.Context(fs, 0.01)
    pins = []
    for c in chunks:
        pa = native.PinnedArray(n, np.complex64)
        pa.array[:] = c
        pins.append(pa)
    for src in ([p.array for p in pins], chunks):            # pinned, then pageable
        pending, got = [], []
        for k in range(8):
            pending.append((k % 4, ctx.submit_format_host(native.FMT_FC32, src[k % 4], abs_offset=k * 1000)))
            if len(pending) == native.MAX_IN_FLIGHT:
                j, t = pending.pop(0)
                got.append((j, len(got) * 1000, ctx.wait(t)))
        while pending:
            j, t = pending.pop(0)
            got.append((j, len(got) * 1000, ctx.wait(t)))
        assert len(got) == 8
        for j, off, recs in got:
            r = recs.copy()
            r["offset"] -= off
            assert_recs_equal(r, want[j], "host-fed chunk %d" % j)
    # a caller-owned pageable buffer page-locked in place (adsb_host_register): same results, DMA'd where it lies
    with native.RegisteredArray(chunks[1]) as reg:
        t = ctx.submit_format_host(native.FMT_FC32, reg.array)
        assert_recs_equal(ctx.wait(t), want[1], "registered chunk")
    # int16 chunks through the same entry (4 B/sample)
    q = M.quantize_iq16(chunks[0])
    ctx.set_iq16_scale(2.0 / 32767.0)
    t = ctx.submit_format_host(native.FMT_SC16, q)
    assert_recs_equal(ctx.wait(t), ctx.process_iq16(q), "host-fed int16")
    with pytest.raises(native.AdsbError):
        for _ in range(native.MAX_IN_FLIGHT + 1):
            ctx.submit_format_host(native.FMT_FC32, chunks[0])
    ctx.close()


def test_registered_buffers_small_calls_read_in_place(native):
    """Inputs of <= 256 KB are not copied to the device: the kernels read the page-locked buffer in place through its
    DEVICE alias (hipHostGetDevicePointer).  For memory page-locked by registration (adsb_host_register) that alias is
    the only address HIP promises -- every small-call entry point on registered buffers, aligned and not, against the
    same calls on pageable copies (which go through the library's own staging buffer)."""
    from gr_adsb_amd import blocks, grshim
    from gr_adsb_amd import modulator as M
    g = Golden("g2msps_df17")
    ctx = native.Context(g.fs, g.thr)
    n = 30000                                           # 120 KB of |IQ|^2, 240 KB of complex64: the in-place branch
    x, iq = g.x[:n].copy(), g.iq[:n].copy()
    want_x, want_iq = ctx.process_mag2(x), ctx.process_iq(iq)
    assert len(want_x) > 10
    with native.RegisteredArray(x) as rx, native.RegisteredArray(iq) as riq:
        assert_recs_equal(ctx.process_mag2(rx.array), want_x, "registered |IQ|^2, in place")
        assert_recs_equal(ctx.process_iq(riq.array), want_iq, "registered complex64, in place")
        assert_recs_equal(ctx.process_mag2(rx.array[3:]), ctx.process_mag2(x[3:].copy()), "registered, 12-byte offset (staged)")
        assert_recs_equal(ctx.process_mag2(rx.array[4:]), ctx.process_mag2(x[4:].copy()), "registered, 16-byte offset (in place)")
        # the GNU Radio emulation entry points on a registered ring buffer: framer.work() / demod.work() chunk by chunk
        H = 8 * g.sps
        buf = np.concatenate([np.zeros(H - 1, np.float32), x])
        with native.RegisteredArray(buf) as rb:
            st = native.Context(g.fs, g.thr)
            ref = native.Context(g.fs, g.thr)
            pos = 0
            for N in (4096, 4096, 8192, 2048, 11568):
                a_ = st.framer_work(rb.array[pos:pos + N + H - 1], N, pos)
                b_ = ref.framer_work(buf[pos:pos + N + H - 1].copy(), N, pos)
                assert_recs_equal(a_, b_, "framer_work on a registered ring, pos %d" % pos)
                pos += N
            assert st.framer_state() == ref.framer_state()
            tags = want_x["offset"][:40]
            bits_a, ok_a, ra = st.demod_work(rx.array, 0, tags, want_ratio=True)
            bits_b, ok_b, rb_ = ref.demod_work(x.copy(), 0, tags, want_ratio=True)
            assert np.array_equal(bits_a, bits_b) and np.array_equal(ok_a, ok_b) and np.array_equal(ra.view(np.uint32), rb_.view(np.uint32))
    ctx.close()


def test_device_side_ordering_after_a_torch_producer(native, torch_mod):
    """FrontEnd tensor entry points order the context's streams behind torch's current stream with an event
    (adsb_wait_for_event), without blocking the host: a producer kernel queued on torch's stream right before the
    submit must be seen complete by k_detect."""
    from gr_adsb_amd.frontend import FrontEnd
    from gr_adsb_amd import modulator as M
    iq = M.synth_iq(1 << 22, 2e6, 3000, seed=77)
    fe = FrontEnd(2e6, 0.01)
    want = fe.process_iq(iq)
    src = to_dev(torch_mod, iq)
    for rep in range(5):
        dst = torch_mod.zeros_like(src)
        junk = torch_mod.randn(1 << 24, device="cuda:0")
        for _ in range(8):                               # keep torch's stream busy in front of the producer
            junk = junk * 1.0001 + 0.5
        dst.copy_(src)                                   # the producer: queued, not finished, when submit is called
        tk = fe.submit_iq_tensor(dst, 0)
        assert_recs_equal(fe.wait(tk), want, "rep %d" % rep)
    assert_recs_equal(fe.process_iq_tensor(src), want, "blocking entry point")


@pytest.mark.parametrize("timing", [False, True])
def test_device_side_ordering_on_every_slot_stream(native, torch_mod, timing):
    """Round 5: submitted passes run on one stream per pipeline slot (timed contexts: k_detect on a shared stream), and the
    caller's event is waited for by the stream the NEXT pass's first kernel runs on.  Three submissions in flight, each with
    its own late producer on torch's stream; then six events pending in front of ONE submission (more than the context
    remembers: the oldest are waited for by every queue at once)."""
    torch = torch_mod
    from gr_adsb_amd.frontend import FrontEnd
    from gr_adsb_amd import modulator as M
    fs, n = 2e6, 1 << 21
    iqs = [M.synth_iq(n, fs, 3000, seed=90 + k) for k in range(3)]
    fe = FrontEnd(fs, 0.01, timing=timing)
    wants = [fe.process_iq(q) for q in iqs]
    srcs = [to_dev(torch, q) for q in iqs]
    junk = torch.randn(1 << 24, device="cuda:0")
    for rep in range(4):
        dsts, tks = [], []
        for k in range(3):
            for _ in range(6):
                junk = junk * 1.0001 + 0.5
            d = torch.zeros_like(srcs[k])
            d.copy_(srcs[k])                             # late producer of THIS submission's input
            dsts.append(d)
            tks.append(fe.submit_iq_tensor(d, 0))
        for k in range(3):
            assert_recs_equal(fe.wait(tks[k]), wants[k], "rep %d slot %d" % (rep, k))
    # six producers, six events, one consumer
    parts = [torch.zeros_like(srcs[0][: n // 6 if i < 5 else n - 5 * (n // 6)]) for i in range(6)]
    evs = []
    side = [torch.cuda.Stream() for _ in range(6)]
    for i, (pt, st) in enumerate(zip(parts, side)):
        with torch.cuda.stream(st):
            j = junk * 1.0001
            for _ in range(4):
                j = j * 1.0001 + 0.5
            pt.copy_(srcs[0][i * (n // 6): i * (n // 6) + pt.shape[0]])
            ev = torch.cuda.Event()
            ev.record(st)
            evs.append(ev)
    whole = torch.empty_like(srcs[0])
    last = torch.cuda.Stream()
    with torch.cuda.stream(last):
        for ev in evs:
            last.wait_event(ev)
        torch.cat(parts, out=whole)
        done = torch.cuda.Event()
        done.record(last)
    for ev in evs + [done]:
        fe.ctx.wait_for_event(ev.cuda_event)
    got = fe.ctx.wait(fe.ctx.submit_iq_device(whole.data_ptr(), n, 0))
    assert_recs_equal(got, wants[0], "seven pending events")


@pytest.mark.parametrize("n", [6000, 6600])        # pulse inside k_detect's LDS window / longer than it (k_longrun)
def test_shard_host_drop_overlong_never_raises(native, n):
    """with ADSB_SHARD_DROP_OVERLONG a matched centre whose burst runs past the buffer is left out
    (the reference degrades, it never raises: framer.py:102-108, demod.py:130-133); without the flag it is an error."""
    from oracle import c_oracle as C
    sps, fs = 8, 8e6
    x = np.full(n, 1e-4, dtype=np.float32)
    # a pulse 600 samples long (longer than k_detect's 256-sample window: k_longrun) whose centre matches the
    # preamble template: plateau just above the threshold, spikes where chips 0, 2, 7, 9 are sampled (stride sps/2)
    r = n - 700
    x[r:r + 600] = 0.02
    p = (r + r + 600) // 2
    x[p + np.array([0, 2, 7, 9]) * (sps // 2)] = 1.0
    ctx = native.Context(fs, 0.01)
    kept = ctx.shard_host(native.FMT_MAG2, x, 0, 0, n - 64, native.STREAM_UNBOUNDED, head_cands=64, drop_overlong=True)
    assert p not in kept["offset"].tolist()
    with pytest.raises(native.AdsbError) as ei:
        ctx.shard_host(native.FMT_MAG2, x, 0, 0, n - 64, native.STREAM_UNBOUNDED, head_cands=64, drop_overlong=False)
    assert ei.value.code == -75
    # the same centre with room behind it is an ordinary record
    x2 = np.concatenate([x, np.full(2000, 1e-4, dtype=np.float32)])
    kept2 = ctx.shard_host(native.FMT_MAG2, x2, 0, 0, len(x2) - 64, native.STREAM_UNBOUNDED, head_cands=64, drop_overlong=True)
    assert p in kept2["offset"].tolist()
    want = C.canonical(x2, sps, np.float32(0.01))
    assert_recs_equal(kept2[kept2["offset"] == p], want[want["offset"] == p], "long-pulse record")


def test_c_abi_output_array_and_error_paths(native):
    """The raw C ABI the way a C caller uses it: caller-owned output array, -ENOSPC with the required count,
    pinned host buffers from adsb_host_alloc, argument validation."""
    import ctypes
    g = Golden("g2msps_df17")
    lib = native.load()
    ctx = native.Context(g.fs, g.thr)
    want = ctx.process_iq(g.iq)
    n_out = ctypes.c_int32(0)
    small = np.zeros(3, dtype=native.BURST_DTYPE)
    rc = lib.adsb_process_iq(ctx._h, ctypes.c_void_p(g.iq.ctypes.data), len(g.iq), 0, ctypes.c_void_p(small.ctypes.data), 3,
                             ctypes.byref(n_out))
    assert rc == -28 and n_out.value == len(want)            # -ENOSPC, count reported
    out = np.zeros(len(want), dtype=native.BURST_DTYPE)
    rc = lib.adsb_process_iq(ctx._h, ctypes.c_void_p(g.iq.ctypes.data), len(g.iq), 0, ctypes.c_void_p(out.ctypes.data), len(out),
                             ctypes.byref(n_out))
    assert rc == 0 and out.tobytes() == want.tobytes()
    # pinned source buffer: straight DMA, same result
    pa = native.PinnedArray(len(g.iq), np.complex64)
    pa.array[:] = g.iq
    assert ctx.process_iq(pa.array).tobytes() == want.tobytes()
    # argument validation
    assert lib.adsb_process_iq(ctx._h, None, 10, 0, None, 0, ctypes.byref(n_out)) == -22
    assert lib.adsb_process_iq_device(ctx._h, ctypes.c_void_p(8), 10, 0, None, 0, ctypes.byref(n_out)) == -22   # misaligned
    assert lib.adsb_framer_work(ctx._h, ctypes.c_void_p(g.x.ctypes.data), 100, 50, 0, None, 0, ctypes.byref(n_out)) == -22
    assert b"8*sps" in lib.adsb_last_error(ctx._h)
    assert lib.adsb_wait(ctx._h, 0, None, 0, ctypes.byref(n_out)) == -22                                         # nothing pending
    assert lib.adsb_wait(ctx._h, 7, None, 0, ctypes.byref(n_out)) == -22
    # empty input is a valid call
    assert lib.adsb_process_iq(ctx._h, ctypes.c_void_p(g.iq.ctypes.data), 0, 0, None, 0, ctypes.byref(n_out)) == 0 and n_out.value == 0


def test_polled_small_passes_never_show_stale_or_partial_records(native):
    """Scheduler-sized framer calls signal completion through memory: the kernel's last store is the pass number the host
    polls (publish_small: every thread fences its record stores to system scope in front of the workgroup barrier, then
    thread 0 publishes).  Hammer such calls, alternating between chunks with DIFFERENT bursts, with FRAMER_SLICES so that
    the records carry bits written by several wavefronts, and compare every call's records -- read the moment the poll
    returns -- byte for byte with the chunk's verified records (equal to the C oracle's for the same samples): a record
    that is stale (the previous call's) or partial (a half not yet arrived) would differ."""
    from gr_adsb_amd import modulator as M
    from oracle import c_oracle as C
    sps, H = 2, 16
    chunks, want = [], []
    ref = native.Context(2e6, 0.01, flags=native.FLAG_FRAMER_SLICES)
    for seed in range(6):
        x = M.mag2(M.synth_iq(6000, 2e6, 3000 + 1000 * seed, seed=900 + seed))      # ~9-24 bursts, some overlapping
        in0 = np.concatenate([np.zeros(H - 1, np.float32), x])
        ref.reset()
        w = ref.framer_work(in0, len(x), 0).copy()
        assert len(w) >= 2
        assert_recs_equal(w, C.canonical(x, sps, np.float32(0.01)), "small pass vs the C oracle")
        chunks.append(in0)
        want.append(w.tobytes())
    assert len(set(want)) == len(want)
    ref.close()
    ctx = native.Context(2e6, 0.01, flags=native.FLAG_FRAMER_SLICES)
    for k in range(6000):
        i = (k * 5 + k // 7) % len(chunks)
        ctx.reset()
        got = ctx.framer_work(chunks[i], len(chunks[i]) - (H - 1), 0)
        assert got.tobytes() == want[i], "call %d (chunk %d): records differ from the blocking pass" % (k, i)
    st = ctx.stats()
    assert st["poll_fallbacks"] <= 60, st          # the short spin catches (nearly) every such call on an idle GPU
    ctx.close()


def test_record_copy_of_an_in_line_pass_while_slot_two_holds_an_overlapped_pass(native, torch_mod):
    """Round 6 (advisor): the record copy of an IN-LINE pass -- here an int8 pass of 1 GiB in an untimed context -- normally rides
    on slot 2's idle stream; when slot 2 itself holds a younger, overlapped pass the copy must not queue behind it: the context
    then uses a copy stream of its own.  Big pass into slot 0, two small ones into slots 1 and 2, collected in order; every result
    equals the blocking call's."""
    from gr_adsb_amd import modulator as M
    torch = torch_mod
    fs = 2e6
    small = M.synth_iq(1 << 20, fs, 3000, seed=61)
    q_small = np.clip(np.round(small.view(np.float32) * 32.0), -127, 127).astype(np.int8).reshape(-1, 2)
    t_small = torch.from_numpy(q_small).to("cuda:0")
    reps = (1 << 29) // len(q_small)                            # 2^29 samples x 2 bytes = 1 GiB: the in-line policy of the 8-bit formats
    t_big = t_small.repeat(reps, 1).contiguous()
    ctx = native.Context(fs, 0.01)
    ctx.set_format_scale(native.FMT_SC8, 1.0 / 32.0)
    want_small = ctx.process_format_device(native.FMT_SC8, t_small.data_ptr(), len(q_small))
    want_big = ctx.process_format_device(native.FMT_SC8, t_big.data_ptr(), t_big.shape[0])
    assert len(want_small) > 1000 and len(want_big) > reps * (len(want_small) - 2)
    for rep in range(2):
        tb = ctx.submit_format_device(native.FMT_SC8, t_big.data_ptr(), t_big.shape[0])
        t1 = ctx.submit_format_device(native.FMT_SC8, t_small.data_ptr(), len(q_small))
        t2 = ctx.submit_format_device(native.FMT_SC8, t_small.data_ptr(), len(q_small))
        assert ctx.wait(tb).tobytes() == want_big.tobytes()
        assert ctx.wait(t1).tobytes() == want_small.tobytes()
        assert ctx.wait(t2).tobytes() == want_small.tobytes()
    ctx.close()


def test_work_inputs_are_page_locked_once_through_their_owning_array(native):
    """Round 6: framer.work() page-locks, once, the array its input is a slice of (blocks.pin_source), so that the library DMA's
    the samples where they lie; tags and PDUs are the same with and without, the registration ends with the owning array, and a
    slice without an ndarray owner (what the real GNU Radio gateway hands over) is left alone."""
    import gc
    from gr_adsb_amd import blocks, grshim
    from gr_adsb_amd import modulator as M
    from oracle import adsb_oracle as O
    fs = 2e6
    blocks.unpin_all()                                             # (arrays other tests keep alive may hold the eight slots)
    x = O.mag2(M.synth_iq(1 << 21, fs, 3000, seed=15))
    sched = [1 << 18] * 8
    outs = []
    for pin in (False, True):
        fr, dm = blocks.framer(fs, 0.01, pin_inputs=pin), blocks.demod(fs)
        dm.start_timestamp = 0.0
        xx = x.copy()
        before = len(blocks._pinned_roots)
        tags, msgs = grshim.drive(fr, dm, xx, sched)
        outs.append(([(t_.offset, float(t_.value[1])) for t_ in tags], [m[1].tobytes() for _, m in msgs]))
        assert len(tags) > 1000
        del fr, dm, tags, msgs, xx
        gc.collect()
        assert len(blocks._pinned_roots) == before                 # dropped with the owning array (or never made)
    assert outs[0] == outs[1]
    big = np.zeros(1 << 20, np.float32)
    assert blocks.pin_source(big[100:5000]) is True and big.ctypes.data in blocks._pinned_roots
    assert blocks.pin_source(big[7:]) is True and len([k for k in blocks._pinned_roots if k == big.ctypes.data]) == 1
    import ctypes
    raw = (ctypes.c_float * 4096)()
    assert blocks.pin_source(np.frombuffer(raw, dtype=np.float32)) is False        # no ndarray owner, too small anyway
    assert blocks.pin_source(np.zeros(100, np.float32)) is False
    del big
    gc.collect()
    assert not blocks._pinned_roots
