"""The PRODUCT device code (gr_adsb_amd/csrc/adsb_device.h + adsb_plan.h) executed by the CPU SIMT
emulator (tests/sim/hipsim.h) and checked against the goldens and the oracle.  This is how kernel
logic is verified in the GPU-less build container; the same checks run on the real GPU through the
C ABI in test_gpu_parity.py."""
import warnings

import numpy as np
import pytest

import simlib
from helpers import every_byte_pair_stream, Golden, assert_recs_equal, assert_recs_match_golden, golden_names, rate_golden_names, snr_bits, unpack
from gr_adsb_amd import modulator as M
from oracle import adsb_oracle as O
from oracle import c_oracle as C


@pytest.mark.parametrize("name", golden_names())
@pytest.mark.parametrize("mode", [0, 1])
def test_canonical_matches_goldens(name, mode):
    g = Golden(name)
    recs, so = simlib.sim_canonical(mode, g.iq if mode == 0 else g.x, g.fs, g.thr)
    assert so.overflow == 0
    assert_recs_match_golden(recs, g)
    assert np.all((recs["flags"] & 2) != 0)


@pytest.mark.parametrize("name", rate_golden_names())
def test_run_time_stride_instances_match_rate_goldens(name):
    """tests/golden/R*.npz: 6 / 10 / 12 / 16 / 24 / 40 / 100 Msps, the rates served by k_detect<fmt, 0> and k_pass_small<0>
    (tap stride sps//2 known only at run time, taps bounds-checked against the LDS window; framer.py:45,137).  A prefix of
    the generated stream through the emulated kernels: canonical int8 / complex64 / |IQ|^2 entry vs the reference's tags
    (the gate is causal, so the tags of a prefix are a prefix of the tags), and work() call by work() call under the
    stored random schedule (k_pass_small<0> for the short calls, k_detect<1, 0> + the tail for the long ones)."""
    g = Golden(name, lazy=True)
    n = min(g.gen["n"], 60 * 120 * g.sps)                       # room for ~60 back-to-back replies
    g.load_generated(0, n)
    offs = g.get("single", "tag_offsets")
    k = int(np.searchsorted(offs, n - 8 * g.sps + 1))
    assert k >= 10
    for mode, data in ((3, g.iq8), (0, g.iq), (1, g.x)):
        recs, so = simlib.sim_canonical(mode, data, g.fs, g.thr, grid_max=3, **({"scale": float(g.scale)} if mode == 3 else {}))
        assert so.overflow == 0
        assert np.array_equal(recs["offset"], offs[:k]), "mode %d" % mode
        assert np.array_equal(snr_bits(recs["peak"], recs["median"]), g.get("single", "tag_snr_bits")[:k])
        dem = (recs["flags"] & 1) != 0
        assert np.array_equal(unpack(recs["bits"][dem]), g.pdu_bits("single")[:int(dem.sum())])
    sched, pos = [], 0
    for N in g.sched("random"):
        if pos + N > n:
            break
        sched.append(N)
        pos += N
    H = 8 * g.sps
    buf = np.concatenate([np.zeros(H - 1, np.float32), g.x])
    fr = simlib.SimFramer(g.fs, g.thr)
    pos, outs = 0, []
    for N in sched:
        outs.append(fr.work(buf[pos:pos + N + H - 1], N, pos)[0])
        pos += N
    recs = np.concatenate(outs)
    want = g.get("random", "tag_offsets")
    kk = int(np.searchsorted(want, pos - 8 * g.sps + 1))
    assert kk >= 5 and np.array_equal(recs["offset"], want[:kk])
    assert np.array_equal(snr_bits(recs["peak"], recs["median"]), g.get("random", "tag_snr_bits")[:kk])


@pytest.mark.parametrize("name", golden_names())
def test_canonical_int16_iq_matches_goldens(name):
    """The fixtures ARE int16 IQ; the int16 input format must reproduce the reference on them directly."""
    g = Golden(name)
    scale = float(np.float32(2.0 / 32767.0))
    assert np.array_equal(O.mag2_iq16(g.z["iq16"], scale), g.x)
    recs, so = simlib.sim_canonical(2, g.z["iq16"], g.fs, g.thr, scale=scale)
    assert_recs_match_golden(recs, g)


@pytest.mark.parametrize("name", ["g2msps_df17", "g8msps_dense", "g2msps_mixed_lowsnr"])
@pytest.mark.parametrize("sched", ["fixed4096", "random"])
def test_framer_work_chunked_matches_goldens(name, sched):
    g = Golden(name)
    H = 8 * g.sps
    buf = np.concatenate([np.zeros(H - 1, np.float32), g.x])
    fr = simlib.SimFramer(g.fs, g.thr)
    pos, outs = 0, []
    for N in g.sched(sched):
        r, _ = fr.work(buf[pos:pos + N + H - 1], N, pos)
        outs.append(r)
        pos += N
    recs = np.concatenate(outs)
    assert np.array_equal(recs["offset"], g.get(sched, "tag_offsets"))
    from helpers import snr_bits
    assert np.array_equal(snr_bits(recs["peak"], recs["median"]), g.get(sched, "tag_snr_bits"))
    assert fr.prev_eob.value == int(g.get(sched, "final_prev_eob"))
    # demod block: k_slice per chunk for the tags inside it
    pos, got_off, got_bits = 0, [], []
    offs = g.get(sched, "tag_offsets")
    for N in g.sched(sched):
        inside = np.flatnonzero((offs >= pos) & (offs < pos + N))
        if len(inside):
            bits, ok, _ = simlib.sim_slice(g.x[pos:pos + N], offs[inside] - pos, g.sps)
            sel = ok.astype(bool)
            got_off.append(offs[inside][sel])
            got_bits.append(unpack(bits[sel]))
        pos += N
    assert np.array_equal(np.concatenate(got_off), g.get(sched, "pdu_offsets"))
    assert np.array_equal(np.concatenate(got_bits), g.pdu_bits(sched))


def test_slice_ratio_matches_reference_confidence():
    g = Golden("g2msps_df17")
    offs = g.get("single", "pdu_offsets")
    bits, ok, ratio = simlib.sim_slice(g.x, offs, g.sps)
    assert ok.all()
    with np.errstate(all="ignore"):
        conf = (np.float32(10.0) * np.log10(ratio)).astype(np.float32)
    assert np.array_equal(conf.view(np.uint32), g.get("single", "pdu_conf_bits"))


@pytest.mark.parametrize("fs,bps,shards", [(2e6, 4000, 4), (20e6, 3000, 3)])
def test_overlapped_shards_stitch_equals_single_call(fs, bps, shards):
    from gr_adsb_amd.frontend import shard_plan
    n = 1 << 17
    iq = M.synth_iq(n, fs, bps, seed=8)
    sps = int(fs // 1e6)
    want, cands = C.canonical(O.mag2(iq), sps, 0.01, want_cands=True)
    got = []
    for p in shard_plan(n, shards, sps):
        r, so = simlib.sim_shard(0, iq[p["lo"]:p["hi"]], p["lo"], p["own_lo"], p["own_hi"], n, fs, 0.01)
        assert (so.flags & 4) == 0
        got.append(r)
    c = np.concatenate(got)
    assert np.array_equal(c["offset"], cands)
    keep = O.resolve_candidates(c["offset"], sps)
    assert_recs_equal(c[keep], want, "stitched shards")


def test_record_capacity_overflow_is_reported():
    g = Golden("g2msps_df17")
    recs, so = simlib.sim_canonical(0, g.iq, g.fs, g.thr, rec_cap=2)
    assert so.overflow == 1


def test_pathological_inputs_match_oracle():
    fs, sps = 2e6, 2
    base = O.mag2(M.synth_iq(1 << 15, fs, 5000, seed=21))
    cases = []
    x = base.copy(); x[3000:7000] = 0.5; cases.append(("long run", x, 0.01))
    x = base.copy(); x[2000:22000] = 0.5; cases.append(("run over several tiles", x, 0.01))
    x = base.copy(); x[:50] = 0.7; x[-40:] = 0.7; cases.append(("starts/ends high", x, 0.01))
    x = base.copy(); x[[100, 5000, 5001, 20000]] = np.nan; x[[3000, 30000]] = np.inf
    cases.append(("nan/inf", x, 0.01)); cases.append(("nan/inf thr 0", x, 0.0))
    cases.append(("thr 0", base, 0.0)); cases.append(("thr < 0", base, -1.0)); cases.append(("thr at noise", base, 0.001))
    cases.append(("zeros", np.zeros(3000, np.float32), 0.01)); cases.append(("const high", np.full(9000, 0.3, np.float32), 0.01))
    for n in (1, 15, 16, 17, 239, 240, 241):
        cases.append(("tiny %d" % n, base[7000:7000 + n].copy(), 0.01))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for what, x, thr in cases:
            want = C.canonical(x, sps, thr)
            recs, so = simlib.sim_canonical(1, x, fs, thr)
            assert_recs_equal(recs, want, what)


def test_noise_window_at_stream_start_and_nan_median_bits():
    """A burst at the very start: the median window is clipped by the zero history (framer.py:156-157)."""
    fs, sps = 2e6, 2
    rng = np.random.default_rng(3)
    bits = M.make_frame(17, rng)
    env = M.burst_waveform(bits, sps)
    for lead in (0, 1, 7, 40, 99, 100, 101):
        x = np.full(600, 1e-4, np.float32)
        x[lead:lead + len(env)] += 0.5 * env
        want = C.canonical(x, sps, 0.01)
        recs, _ = simlib.sim_canonical(1, x, fs, 0.01)
        assert len(want) == 1
        assert_recs_equal(recs, want, "lead %d" % lead)


@pytest.mark.parametrize("fs,bps,shards,head", [(2e6, 8000, 4, 64), (2e6, 8000, 4, 2), (20e6, 3000, 3, 16), (2e6, 30000, 5, 8)])
def test_gated_shards_with_head_fixup_equal_single_call(fs, bps, shards, head):
    """adsb_shard_device(head_cands > 0) semantics: device gate with fresh state + ungated head region, then
    the host fix-up with the previous shard's 8-byte tail (gr_adsb_amd.sharding)."""
    from gr_adsb_amd import _native, sharding
    from gr_adsb_amd.frontend import shard_plan
    n = 1 << 17
    iq = M.synth_iq(n, fs, bps, seed=8)
    sps = int(fs // 1e6)
    want = C.canonical(O.mag2(iq), sps, 0.01)
    plans = shard_plan(n, shards, sps)
    parts = [simlib.sim_shard(0, iq[p["lo"]:p["hi"]], p["lo"], p["own_lo"], p["own_hi"], n, fs, 0.01, head_cands=head)[0]
             for p in plans]
    tails = [_native.shard_tail(r, sps) for r in parts]
    outs = []
    for g, r in enumerate(parts):
        k = _native.shard_fixup(r, sps, sharding.incoming_eob(tails, g))
        assert k is not None, "head region too short for this fixture"
        assert not np.any(k["flags"] & 16) and np.all(k["flags"] & 2)
        outs.append(k)
    assert_recs_equal(np.concatenate(outs), want, "gated shards + fixup")


@pytest.mark.parametrize("fs,mode", [(2e6, 0), (4e6, 1), (8e6, 2)])
def test_parity_prefilter_flags_all_formats(fs, mode):
    """DF / length class / parity verdict the device attaches to every PDU equal the oracle's
    restatement of decoder.py:550-688 (pinned against the reference decoder), for every downlink format."""
    dfs = (0, 4, 5, 11, 16, 17, 18, 19, 20, 21, 24, 7, 28)
    iq = M.synth_iq(1 << 18, fs, 5000, seed=77, df_choices=dfs, df_weights=[1.0 / len(dfs)] * len(dfs))
    if mode == 2:
        q = M.quantize_iq16(iq)
        scale = float(np.float32(2.0 / 32767.0))
        data, x = q, O.mag2_iq16(q, scale)
        recs, _ = simlib.sim_canonical(2, q, fs, 0.01, scale=scale)
    else:
        x = M.mag2(iq)
        recs, _ = simlib.sim_canonical(mode, iq if mode == 0 else x, fs, 0.01)
    want = C.canonical(x, int(fs // 1e6), np.float32(0.01))
    assert_recs_equal(recs, want, "parity stream")
    wf, _ = C.parity_flags(want)
    assert np.array_equal(recs["flags"] & 0x1FE1, wf & 0x1FE1)
    df = (recs["flags"] >> 8) & 31
    ok = (recs["flags"] & 32) != 0
    assert len(np.unique(df)) >= 10 and 10 < ok.sum() < len(recs)
    assert set(np.unique(df[ok])) <= {11, 17, 18, 19}


@pytest.mark.parametrize("name", ["g2msps_df17", "g8msps_dense", "g20msps"])
@pytest.mark.parametrize("mode", [3, 4])
def test_canonical_int8_iq_formats_match_oracle(name, mode):
    """8-bit IQ ingestion (cs8 / RTL-SDR cu8).  The fixtures' signal requantised to 8 bits;
    conversion is exact by construction, so everything downstream must equal the oracle run on the oracle's
    own |IQ|^2 of the same bytes -- bit for bit, at every sample rate."""
    g = Golden(name)
    ob = mode == 4
    q = M.quantize_iq8(g.iq, offset_binary=ob)
    scale = float(np.float32(2.0 / 255.0 if ob else 2.0 / 127.0))
    x = O.mag2_iq8(q, scale, ob)
    recs, so = simlib.sim_canonical(mode, q, g.fs, g.thr, scale=scale)
    want = C.canonical(x, g.sps, np.float32(g.thr))
    assert len(want) > 10
    assert_recs_equal(recs, want, "%s mode %d" % (name, mode))
    # and the same records come out of the float |IQ|^2 entry fed with the oracle's conversion
    recs1, _ = simlib.sim_canonical(1, x, g.fs, g.thr)
    assert_recs_equal(recs, recs1, "int8 vs mag2 entry")


@pytest.mark.parametrize("bps,block,head,growth", [(3000, 1 << 14, 64, 16), (40000, 1 << 13, 1, 1), (6000, 5000, 64, 16)])
def test_replay_blocks_equal_one_canonical_call(bps, block, head, growth, monkeypatch):
    """gr_adsb_amd.replay: a recording processed block by block (overlapped shards, host fix-up with the carried
    end-of-burst state, greedy fallback for dense traffic / tiny heads) == one canonical call over all of it."""
    from gr_adsb_amd import replay
    fs, sps, n = 2e6, 2, (1 << 16) + 777
    iq = M.synth_iq(n, fs, bps, seed=19)
    if head == 1:
        # an unbroken chain across every block boundary: clean preambles every 100 samples (< the 126-sample gate), so
        # every other one is kept and a 1-centre head region cannot re-synchronise -> greedy fallback
        x = np.full(n, 1e-4, dtype=np.float32)
        for base in range(50, n - 400, 100):
            x[base + np.array([0, 2, 7, 9])] = 1.0
        iq = np.sqrt(x).astype(np.complex64)

    def shard_fn(plan, head_cands):
        return simlib.sim_shard(0, iq[plan["lo"]:plan["hi"]], plan["lo"], plan["own_lo"], plan["own_hi"], n, fs, 0.01,
                                head_cands=head_cands)[0]

    fallbacks = []
    gate = replay.greedy_gate
    monkeypatch.setattr(replay, "greedy_gate", lambda *a: (fallbacks.append(1), gate(*a))[1])
    parts = list(replay.replay_blocks(n, sps, block, shard_fn, head_cands=head, head_growth=growth))
    assert len(parts) >= 4 and (len(fallbacks) > 0) == (head == 1)
    got = np.concatenate(parts)
    want = C.canonical(O.mag2(iq), sps, np.float32(0.01))
    assert_recs_equal(got, want, "replay")
    assert np.all((got["flags"] & 2) != 0) and not np.any(got["flags"] & 16)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("name", golden_names())
def test_both_tails_match_goldens(name, mode):
    """The tail as a chain of kernels (bulk passes) and as ONE workgroup (k_tail_small: small passes, e.g. every GNU
    Radio work() call) are the same bodies run two ways: both must reproduce the goldens, canonical and chunked."""
    g = Golden(name)
    with simlib.tail_mode(mode):
        recs, _ = simlib.sim_canonical(0, g.iq, g.fs, g.thr)
        assert_recs_match_golden(recs, g)
        fr = simlib.SimFramer(g.fs, g.thr)
        H = 8 * g.sps
        buf = np.concatenate([np.zeros(H - 1, np.float32), g.x])
        pos, offs = 0, []
        for N in g.sched("random"):
            r, _ = fr.work(buf[pos:pos + N + H - 1], N, pos)
            offs.append(r["offset"])
            pos += N
        assert np.array_equal(np.concatenate(offs), g.get("random", "tag_offsets"))
        assert fr.prev_eob.value == int(g.get("random", "final_prev_eob"))


def test_fused_tail_with_many_lists_and_shard_heads():
    """k_tail_small on its limits: a few hundred lists (several k_scan rounds), head records of a gated shard."""
    fs, n = 2e6, 170_000
    iq = M.synth_iq(n, fs, 8000, seed=33)
    want = C.canonical(O.mag2(iq), 2, np.float32(0.01))
    for mode in (1, 2):
        with simlib.tail_mode(mode):
            got, so = simlib.sim_canonical(0, iq, fs, 0.01, grid_max=160, rec_cap=40)
            assert so.overflow == 0
            assert_recs_equal(got, want, "tail mode %d" % mode)
            a = simlib.sim_shard(0, iq[:90000], 0, 0, 80000, n, fs, 0.01, head_cands=16, grid_max=40)[0]
            if mode == 1:
                ref = a
            else:
                assert a.tobytes() == ref.tobytes()


def test_many_units():
    """A grid of 40 workgroups = 160 per-wavefront lists: ordering across many units (k_scan / k_gather)."""
    fs, n = 2e6, 170_000
    iq = M.synth_iq(n, fs, 8000, seed=33)
    recs, so = simlib.sim_canonical(0, iq, fs, 0.01, grid_max=40)
    assert so.overflow == 0
    assert_recs_equal(recs, C.canonical(O.mag2(iq), 2, np.float32(0.01)), "160 lists")
    assert len(recs) > 200


@pytest.mark.parametrize("mode", [1, 3])
def test_ordering_kernel_over_several_workgroups_with_long_pulses(mode):
    """k_order (round 4: long pulses + counts -> offsets + words into stream order in ONE launch) with more lists than one
    of its workgroups owns (1024): 2600 lists = three workgroups, every one summing the counts in front of it itself; long
    pulses (plateaus longer than k_detect's LDS window) planted in lists of the first, the second and the last workgroup,
    one of them with a matching preamble behind it, so that each workgroup finishes its own placeholders before it reads
    the list; and a call with no centre at all (empty lists everywhere).  Against the C oracle, bit for bit."""
    n = 2600 * 1024 + 500
    fs, sps = 2e6, 2
    x = O.mag2(M.synth_iq(n, fs, 2500, seed=91)).copy()
    env = M.burst_waveform(M.make_frame(17, np.random.default_rng(3)), sps)
    for start, ln in ((300 * 1024 + 17, 1500), (1030 * 1024 + 999, 1281), (1500 * 1024 + 5, 3000), (2599 * 1024 - 700, 1400)):
        x[start:start + ln] = 0.31
        x[start + ln:start + ln + 40] = 0.0005                        # a quiet gap, then a reply right behind the plateau
        x[start + ln + 40:start + ln + 40 + len(env)] = np.maximum(x[start + ln + 40:start + ln + 40 + len(env)], 0.5 * env)
    # a long pulse that IS the first preamble pulse of a reply (centre matches: the record comes from global memory)
    s0 = 2000 * 1024 + 333
    x[s0 - 1400:s0 + 1] = 0.9
    x[s0 + 1:s0 + 1 + len(env) - 1] = 0.0
    # centre of [s0-1400, s0] is s0-700: plant the other preamble pulses relative to it
    c = s0 - 700
    x[c + 16:c + 16 + 240] = 0.0
    if mode == 3:
        q = M.quantize_iq8((np.sqrt(x) * np.exp(1j * 0.7)).astype(np.complex64), full_scale=4.0)
        scale = float(np.float32(4.0 / 127.0))
        data, kw = q, {"scale": scale}
        xx = O.mag2_iq8(q, scale, False)
    else:
        data, kw, xx = x, {}, x
    want = C.canonical(xx, sps, np.float32(0.01))
    with simlib.tail_mode(1):
        got, so = simlib.sim_canonical(mode, data, fs, 0.01, grid_max=650, **kw)
        assert so.overflow == 0 and so.long_count >= 4
        assert_recs_equal(got, want, "k_order, three workgroups")
        assert len(got) > 2000
        z, so = simlib.sim_canonical(1, np.zeros(1100 * 1024, np.float32), fs, 0.01, grid_max=650)
        assert len(z) == 0 and so.n_rec == 0 and so.flags == 0 and so.lastp == -(1 << 62)


@pytest.mark.parametrize("fs,bps,mode,grid_max", [(2e6, 4000, 0, 1), (8e6, 6000, 1, 2), (20e6, 9000, 3, 1), (2e6, 3000, 4, 2)])
def test_bulk_pass_in_several_rounds_of_short_chunks(fs, bps, mode, grid_max):
    """A call with many more tiles than resident wavefronts is cut into up to eight rounds of chunks of four tiles or more
    (adsb_plan.h: plan_chunks, the function the library's enqueue() uses): every chunk fills its own window head (six loads
    per lane in flight together, samples in front of the buffer and past its end are zeros), the usual-tile instance and the
    general one alternate at the ends of the call, and the tail orders far more lists than one round has."""
    n = 200_000
    T, F, B = simlib.kernel_geometry()
    ntiles = -(-n // T)
    units, per = _native_plan(ntiles * T, grid_max * 4)
    assert units > 4 * grid_max * 4 and per >= 4 * T          # several rounds, short chunks
    iq = M.synth_iq(n, fs, bps, seed=int(fs // 1e6) + mode)
    sps = int(fs // 1e6)
    if mode == 0:
        data, want = iq, C.canonical(O.mag2(iq), sps, np.float32(0.01))
        kw = {}
    elif mode == 1:
        data = O.mag2(iq)
        want = C.canonical(data, sps, np.float32(0.01))
        kw = {}
    else:
        q = M.quantize_iq8(iq, full_scale=4.0, offset_binary=(mode == 4))
        scale = float(np.float32(4.0 / (255.0 if mode == 4 else 127.0)))
        data, want = q, C.canonical(O.mag2_iq8(q, scale, mode == 4), sps, np.float32(0.01))
        kw = {"scale": scale}
    got, so = simlib.sim_canonical(mode, data, fs, 0.01, grid_max=grid_max, **kw)
    assert so.overflow == 0 and len(want) > 25
    assert_recs_equal(got, want, "rounds, mode %d" % mode)


def _native_plan(n_samples, resident):
    from gr_adsb_amd import build as b
    b.build()
    from gr_adsb_amd import _native
    return _native.plan_chunks(n_samples, resident)


@pytest.mark.parametrize("fs,bps,mode", [(2e6, 6000, 0), (4e6, 5000, 1), (8e6, 6000, 0), (20e6, 3000, 1)])
def test_long_aware_gate_matches_its_oracle(fs, bps, mode):
    """SURVEY.md §8f-4, opt-in (ADSB_FLAG_LONG_AWARE_GATE): the gate holds 119*sps after a burst whose first data bit is
    set.  Not the reference -- pinned to the oracle's restatement of that rule (NumPy and C agree); the default mode
    is untouched and never sets the hint flag; overlapped shards still stitch to the single-call result."""
    sps = int(fs // 1e6)
    n = 1 << 17
    iq = M.synth_iq(n, fs, bps, seed=44, df_choices=(4, 11, 17, 20), df_weights=(0.2, 0.2, 0.4, 0.2))
    x = M.mag2(iq)
    data = iq if mode == 0 else x
    ref = C.canonical(x, sps, np.float32(0.01))
    with C.long_aware_gate():
        want = C.canonical(x, sps, np.float32(0.01))
    o = O.run_stream(x, fs, 0.01, long_aware=True)
    assert np.array_equal(want["offset"], o["tag_offsets"]) and np.array_equal(want["bits"][(want["flags"] & 1) != 0], O.pack_bits(o["pdu_bits"]))
    assert len(want) <= len(ref) and (sps > 4 or len(want) < len(ref))      # false re-triggers inside long replies are gone
    with simlib.long_aware_gate():
        got, _ = simlib.sim_canonical(mode, data, fs, 0.01)
        assert_recs_equal(got, want, "long-aware canonical")
        assert np.array_equal(got["flags"] & 0x2000, want["flags"] & 0x2000) and (got["flags"] & 0x2000).any()
        # overlapped shards, gated per shard + host fix-up, == one call
        from gr_adsb_amd import _native, replay
        from gr_adsb_amd.frontend import shard_plan

        def shard_fn(plan, hc):
            return simlib.sim_shard(mode, data[plan["lo"]:plan["hi"]], plan["lo"], plan["own_lo"], plan["own_hi"], n, fs, 0.01,
                                    head_cands=hc)[0]
        parts = list(replay.replay_blocks(n, sps, 1 << 14, shard_fn, head_cands=8))
        assert_recs_equal(np.concatenate(parts), want, "long-aware replay")
        ung = np.concatenate([shard_fn(p_, 0) for p_ in shard_plan(n, 5, sps)])
        assert_recs_equal(_native.stitch(ung, sps), want, "long-aware stitch")
    got0, _ = simlib.sim_canonical(mode, data, fs, 0.01)
    assert_recs_equal(got0, ref, "default mode unchanged")
    assert not (got0["flags"] & 0x2000).any()


@pytest.mark.parametrize("sps", [4, 8, 20])
def test_bits_of_bursts_longer_than_the_window_arrive_later(sps):
    """At 4 Msps and up a burst is longer than what k_detect's LDS window holds behind a late centre: its bits are taken as
    the samples arrive (the wavefront's pending list, pend_step), at the end of a wavefront's chunk from global memory
    (pend_flush).  Dense overlapping bursts over many tiles, tiny grids (chunk ends everywhere) and one unit (none)."""
    from oracle import c_oracle as C
    from helpers import assert_recs_equal
    rng = np.random.default_rng(50 + sps)
    n = 30000
    x = rng.exponential(1e-3, n).astype(np.float32)
    env = M.burst_waveform(M.make_frame(17, rng), sps)
    for _ in range(40):
        s = int(rng.integers(0, n - 10))
        e = min(n, s + len(env))
        x[s:e] = np.maximum(x[s:e], np.float32(rng.uniform(0.05, 1.0)) * env[:e - s])
    want = C.canonical(x, sps, np.float32(0.01))
    assert (want["flags"] & 1).sum() >= 1
    for gm in (1, 2, 6):
        got, so = simlib.sim_canonical(1, x, sps * 1e6, 0.01, grid_max=gm)
        assert so.overflow == 0
        assert_recs_equal(got, want, "sps %d grid_max %d" % (sps, gm))


@pytest.mark.parametrize("sps", [6, 50, 200])
def test_sample_rates_without_their_own_instance(sps):
    """k_detect is instantiated for 2 / 4 / 8 / 20 Msps (preamble taps at immediate offsets); any other even rate runs the
    run-time-stride instance.  At 50 and 200 samples per microsecond a pulse is longer than a mask unit, a burst spans tens
    of tiles (pending list across many tiles) and the taps of a late centre leave the LDS window."""
    rng = np.random.default_rng(3 + sps)
    n = 120 * sps * 6 + 5000
    x = rng.exponential(1e-3, n).astype(np.float32)
    env = M.burst_waveform(M.make_frame(17, rng), sps)
    for s in (1000, 1000 + 130 * sps, n - len(env) - 300, 2500 + 260 * sps):
        e = min(n, s + len(env))
        x[s:e] = np.maximum(x[s:e], np.float32(0.5) * env[:e - s])
    want = C.canonical(x, sps, np.float32(0.01))
    assert len(want) >= 3 and (want["flags"] & 1).sum() >= 3
    for gm in (1, 4):
        got, so = simlib.sim_canonical(1, x, sps * 1e6, 0.01, grid_max=gm)
        assert so.overflow == 0
        assert_recs_equal(got, want, "sps %d grid_max %d" % (sps, gm))


@pytest.mark.parametrize("scale", [1.0 / 128.0, 4.0, 2.0 ** -20, 2.0 ** 30])
def test_int8_power_of_two_scale_conversion_every_byte_pair(scale):
    """body_convert of the power-of-two instance (k_detect<5, .>: the dot product accumulated onto the bit pattern of 2^23,
    one fused multiply-add) == the generic int8 chain == the oracle's f32(i8)*scale squared and summed, for all 65536 byte
    pairs in both sample positions of a word."""
    k = np.arange(65536, dtype=np.uint32)
    words = np.concatenate([k | (np.roll(k, 7919) << 16), (k << 16) | np.roll(k, 104729 % 65536)]).astype(np.uint32)
    iq8 = words.view(np.int8)
    want = O.mag2_iq8(iq8, scale)
    for mode in (3, 5):
        got = simlib.convert8(mode, iq8, scale)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), mode
    u8 = words.view(np.uint8)
    want_u = O.mag2_iq8(u8, scale, True)
    for mode in (4, 6):          # 6: the offset-binary power-of-two instance (two dot products per sample: x.x and x.1, x = u8 - 128)
        assert np.array_equal(simlib.convert8(mode, u8, scale).view(np.uint32), want_u.view(np.uint32)), mode


def test_int8_power_of_two_scale_every_byte_pair_through_the_kernels():
    """a slice of every_byte_pair_stream (4096 pairs as peaks, 4096 others as medians) through the emulated kernels."""
    iq8, thr = every_byte_pair_stream(1.0 / 128.0)
    lo, hi = 2 * 256 * 30000, 2 * 256 * 34096
    seg = np.ascontiguousarray(iq8[lo:hi])
    want = C.canonical(O.mag2_iq8(seg, 1.0 / 128.0), 2, thr)
    assert len(want) >= 4000
    recs, so = simlib.sim_canonical(3, seg, 2e6, float(thr), scale=1.0 / 128.0)
    assert so.overflow == 0
    assert_recs_equal(recs, want, "every byte pair")
    # the same bytes read as offset binary with the RTL-SDR scale (u8 - 127.5) / 128 = (2 u8 - 255) * 2^-8: k_detect<6, .>
    u8 = seg.view(np.uint8)
    xu = O.mag2_iq8(u8, 2.0 ** -8, True)
    thr_u = np.float32(0.75) * np.float32(2.0 ** -16) * np.float32(2.0)          # below the smallest |IQ|^2 (2 * 2^-16)
    want_u = C.canonical(xu, 2, thr_u)
    recs, so = simlib.sim_canonical(4, u8, 2e6, float(thr_u), scale=2.0 ** -8)
    assert so.overflow == 0
    assert_recs_equal(recs, want_u, "every byte pair, offset binary")


@pytest.mark.parametrize("fs,bps", [(2e6, 5000), (8e6, 6000), (12e6, 4000)])
def test_two_tiles_ahead_chunks_of_every_length(fs, bps):
    """The int8 power-of-two-scale instance keeps TWO tile bodies in flight (tile loop unrolled twice, adsb_device.h): chunks of
    odd and even numbers of tiles, streams that end in the first or the second half of the unrolled body, ragged last tiles,
    one resident round and several -- against the C oracle."""
    T, F, B = simlib.kernel_geometry()
    sps = int(fs // 1e6)
    scale = 4.0 / 128.0
    iq = M.synth_iq(9 * T + 700, fs, bps, seed=sps)
    q8 = M.quantize_iq8(iq, full_scale=4.0 * 127.0 / 128.0)
    for n in (T - 1, T, T + 1, 2 * T + F, 3 * T + F + 5, 4 * T + 17, 5 * T + F, 7 * T + 300, 9 * T + 700):
        seg = np.ascontiguousarray(q8[:2 * n])
        want = C.canonical(O.mag2_iq8(seg, scale), sps, np.float32(0.01))
        for grid_max in (1, 2, 40):
            got, so = simlib.sim_canonical(3, seg, fs, 0.01, grid_max=grid_max, scale=scale)
            assert so.overflow == 0
            assert_recs_equal(got, want, "n %d grid %d" % (n, grid_max))


@pytest.mark.parametrize("fs,bps,n", [(20e6, 9000, 3_000_000), (8e6, 9000, 1_500_000), (2e6, 20000, 400_000)])
def test_output_stage_flushes_in_the_middle_of_a_chunk(fs, bps, n):
    """Lists far longer than the sixteen-entry LDS stage (adsb_device.h: Stage): dense overlapping replies on four units only --
    the stage is flushed many times per chunk, bursts longer than the window complete their records after a flush has taken
    the first half to global memory (20 / 8 Msps), and the last partial stage goes out at the end of the chunk."""
    sps = int(fs // 1e6)
    iq = M.synth_iq(n, fs, bps, seed=17 + sps)
    x = O.mag2(iq)
    want = C.canonical(x, sps, np.float32(0.01))
    assert len(want) > 40 * 4
    got, so = simlib.sim_canonical(1, x, fs, 0.01, grid_max=1)
    assert so.overflow == 0 and so.n_rec > 40 * 4
    assert_recs_equal(got, want, "stage")


@pytest.mark.parametrize("name", ["Pnegzero_2msps", "Pnegzero_8msps"])
def test_signed_zeros_in_the_noise_window_match_the_reference(name):
    """tests/golden/Pnegzero_*.npz (tools/make_golden_negzero.py, the REAL reference's outputs): -0.0 and +0.0 mixed in the
    noise window, zero medians in most windows.  np.median is np.mean of the middle element(s) and that sum starts from
    +0.0, so a zero median is always +0.0 (SNR +inf, never NaN): the emulated device code must agree bit for bit, in one
    call and work() call by work() call (chunk starts cut the window to every odd and even length)."""
    g = Golden(name)
    assert int((g.x.view(np.uint32) == 0x80000000).sum()) > 1000
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        recs, so = simlib.sim_canonical(1, g.x, g.fs, g.thr)
        assert_recs_match_golden(recs, g)
        assert not np.any(recs["median"].view(np.uint32) == 0x80000000), "a zero median is +0.0 (framer.py:157: np.median)"
        assert int(np.isposinf(np.float32(10.0) * np.log10(recs["peak"] / recs["median"])).sum()) >= 40
        H = 8 * g.sps
        buf = np.concatenate([np.zeros(H - 1, np.float32), g.x])
        fr = simlib.SimFramer(g.fs, g.thr)
        pos, outs = 0, []
        for N in g.sched("random"):
            outs.append(fr.work(buf[pos:pos + N + H - 1], N, pos)[0])
            pos += N
        recs = np.concatenate(outs)
        assert np.array_equal(recs["offset"], g.get("random", "tag_offsets"))
        assert np.array_equal(snr_bits(recs["peak"], recs["median"]), g.get("random", "tag_snr_bits"))


@pytest.mark.parametrize("mode,scale", [(3, 1.0 / 128.0), (3, 1.0 / 127.0), (4, 1.0 / 255.0), (4, 1.0 / 256.0)])
@pytest.mark.parametrize("fs", [2e6, 4e6, 6e6])
def test_every_rise_a_tile_can_have_8bit_formats(mode, scale, fs):
    """Streams with up to 512 rises per tile -- every rise a tile can have -- and matched preambles all over them, through the
    8-bit formats: the dot-product instance (scale 2^-7), the generic int8 one and uint8, a compiled-in tap stride and a
    run-time one.  (Written for a rejected variant whose rise list held 256 rises and worked a tile off in batches; the
    bursts behind the 256th rise of a tile are what that variant's second batch had to deliver.)"""
    from helpers import rise_storm_iq8
    sps = int(fs // 1e6)
    for seed, n in ((1, 1024 * 7 + 300), (2, 1024 * 3)):
        q = rise_storm_iq8(n, seed=seed, offset_binary=mode == 4)
        x = O.mag2_iq8(q, float(np.float32(scale)), mode == 4)
        above = x >= np.float32(0.01)
        rises = above[1:] & ~above[:-1]
        per_tile = [int(rises[t:t + 1024].sum()) for t in range(0, n - 1, 1024)]
        assert max(per_tile) > 300, per_tile
        want = C.canonical(x, sps, np.float32(0.01))
        recs, so = simlib.sim_canonical(mode, q, fs, 0.01, scale=float(np.float32(scale)))
        assert_recs_equal(recs, want, "rise storm mode %d scale %g fs %g seed %d" % (mode, scale, fs, seed))
        if sps == 2:
            assert len(want) > 10
            # ... and some delivered bursts are rises number 256 and up of their tile (tiles start at the first framer sample,
            # -(8*sps - 1)): they come out of the second batch
            ridx = np.flatnonzero(rises) + 1
            late = 0
            for off in want["offset"]:
                t0 = -(8 * sps - 1) + ((int(off) + 8 * sps - 1) // 1024) * 1024
                late += int(np.count_nonzero((ridx >= t0) & (ridx < off)) >= 256)
            assert late >= 1, "no burst behind the 256th rise of a tile"
        else:
            # the same storm with the bare preambles spaced for THIS rate: matched centres among the rises at 4 and 6 Msps too
            q = rise_storm_iq8(n, seed=seed, offset_binary=mode == 4, half=sps // 2)
            x = O.mag2_iq8(q, float(np.float32(scale)), mode == 4)
            want = C.canonical(x, sps, np.float32(0.01))
            assert len(want) >= 3
            recs, so = simlib.sim_canonical(mode, q, fs, 0.01, scale=float(np.float32(scale)))
            assert_recs_equal(recs, want, "rise storm spaced for the rate, mode %d scale %g fs %g seed %d" % (mode, scale, fs, seed))
