"""C-ABI surface checks that need no GPU: the library loads, exports every symbol the header declares,
refuses to work without a device (no CPU fallback), and its pure-host helpers agree with the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    from gr_adsb_amd import build as b
    b.build()
    from gr_adsb_amd import _native
    return _native


def declared_functions():
    text = open(os.path.join(ROOT, "include", "adsb_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(adsb_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(native):
    lib = native.load()
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    assert set(names) == set(native.EXPORTS)
    assert lib.adsb_abi_version() == native.ABI_VERSION == 5


def test_struct_layouts(native):
    assert native.BURST_DTYPE.itemsize == 32
    assert native.BURST_DTYPE.fields["bits"][1] == 16 and native.BURST_DTYPE.fields["flags"][1] == 30
    assert ctypes.sizeof(native.Stats) == 112          # ABI 3: + poll_fallbacks; ABI 4: + shard_fallbacks


def test_no_cpu_fallback_without_device(native):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(native.AdsbError) as e:
        native.Context(2e6, 0.01)
    assert e.value.code == -19  # -ENODEV
    from gr_adsb_amd import blocks
    with pytest.raises(native.AdsbError):
        blocks.framer(2e6, 0.01)
    with pytest.raises(native.AdsbError):
        blocks.demod(2e6)


def test_create_rejects_bad_sample_rates(native):
    lib = native.load()
    h = ctypes.c_void_p()
    # non-integer sps (framer.py:44) / odd sps (work() would raise) / above ADSB_MAX_SPS (nothing beyond it is tested)
    for fs in (2.5e6, 0.0, -2e6, 1e6, 3e6, 102e6, 4096e6):
        assert lib.adsb_create(fs, 0.01, 0, 0, ctypes.byref(h)) == -22
        assert not h.value


def test_block_constructor_asserts_like_reference(native):
    from gr_adsb_amd import blocks
    with pytest.raises(AssertionError):
        blocks.framer(2.5e6, 0.01)
    with pytest.raises(AssertionError):
        blocks.demod(2.5e6)


def test_snr_db_c_close_to_numpy(native):
    """np.log10 on float32 is SIMD/SVML on some hosts and libm on others, so the reference's own SNR bits
    are host dependent; the Python shim therefore finalises SNR with NumPy (bit-equal to the reference on
    the same host) and the C helper (libm log10f) is only required to be within a few ULP of it."""
    rng = np.random.default_rng(0)
    peak = rng.uniform(1e-3, 2.0, 2000).astype(np.float32)
    med = rng.uniform(1e-5, 1e-2, 2000).astype(np.float32)
    want = native.snr_db(peak, med)
    got = np.array([native.snr_db_c(p, m) for p, m in zip(peak, med)], dtype=np.float32)
    ulp = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
    assert ulp.max() <= 4


def test_mode_s_syndrome_helper_matches_reference_known_answers(native):
    """adsb_mode_s_syndrome (pure host arithmetic) against the reference decoder's known answers in
    tests/golden/g_parity.npz: DF, payload length, pass/fail for DF 11/17/18/19, announced address otherwise."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g_parity.npz"))
    for i in range(len(z["bits"])):
        syn, df, nb = native.mode_s_syndrome(z["bits"][i])
        assert df == z["df"][i] and (nb if nb else -1) == z["payload_length"][i]
        if df in (11, 17, 18, 19):
            assert (syn == 0) == bool(z["parity_passed"][i])
        elif nb:
            assert syn == z["aa"][i] and z["parity_passed"][i] == 0


def test_stitch_and_fixup_honour_the_long_hint(native):
    """Host gate functions with records of a long-aware context (ADSB_BURST_LONG_HINT: 119*sps instead of 63*sps);
    records without the flag behave exactly as before.  Pure host arithmetic."""
    sps = 2
    offs = np.array([1000, 1000 + 130, 1000 + 240, 1000 + 400, 5000, 5000 + 127, 5000 + 239], dtype=np.int64)
    c = np.zeros(len(offs), dtype=native.BURST_DTYPE)
    c["offset"] = offs
    c["flags"][[0, 4]] = native.BURST_LONG_HINT          # the bursts at 1000 and 5000 are long replies
    kept = native.stitch(c, sps)
    # 1000 (long: gate to 1238) -> 1130 rejected, 1240 accepted (short: to 1366), 1400 accepted; 5000 (long: to 5238) -> 5127 rejected, 5239 accepted
    assert kept["offset"].tolist() == [1000, 1240, 1400, 5000, 5239]
    c["flags"] = 0
    assert native.stitch(c, sps)["offset"].tolist() == [1000, 1130, 1400, 5000, 5127]      # reference gate: 63*sps = 126
    # fix-up: a head whose first record is long must not sync on a centre inside its window
    h = np.zeros(5, dtype=native.BURST_DTYPE)
    h["offset"] = [100, 100 + 200, 100 + 239, 1500, 2000]
    h["flags"] = native.BURST_HEAD
    h["flags"][0] |= native.BURST_LONG_HINT | native.BURST_KEPT
    h["flags"][2] |= native.BURST_KEPT
    h["flags"][3] |= native.BURST_KEPT                   # beyond everybody's reach: the head syncs here
    h["flags"][4] = native.BURST_KEPT                    # first record behind the head region
    out = native.shard_fixup(h, sps, native.EOB_NONE)
    assert out["offset"].tolist() == [100, 339, 1500, 2000]
    # with the reference's 63*sps the centre at 300 would already be "more than a window" behind 100: not with the hint
    assert native.shard_head_sync(h, sps) == 1500 and native.shard_tail(h[:1], sps) == 100 + 119 * sps
    # (with an incoming eob past 1500 the head cannot sync any more)
    assert native.shard_fixup(h, sps, 1600) is None


def test_stitch_is_the_reference_gate(native):
    from oracle import adsb_oracle as O
    rng = np.random.default_rng(1)
    for sps in (2, 8, 20):
        off = np.unique(rng.integers(0, 200000, 3000)).astype(np.int64)
        c = np.zeros(len(off), dtype=native.BURST_DTYPE)
        c["offset"] = off
        kept = native.stitch(c, sps)
        assert np.array_equal(kept["offset"], off[O.resolve_candidates(off, sps)])
        assert np.all(kept["flags"] & 2)
    c = np.zeros(2, dtype=native.BURST_DTYPE)
    c["offset"] = [5, 5]
    with pytest.raises(native.AdsbError):
        native.stitch(c, 2)


def test_head_sync_predicts_fixup_and_fixup_is_exact(native):
    """Host-only: for random centre lists, (1) shard_head_sync() > eob_in  <=>  adsb_shard_fixup succeeds (=> only, for a shard that is all head),
    (2) when it succeeds the result equals the sequential gate started from eob_in."""
    from oracle import adsb_oracle as O
    rng = np.random.default_rng(7)
    hits = misses = 0
    for trial in range(400):
        sps = int(rng.choice([2, 8, 20]))
        gate = 63 * sps
        n = int(rng.integers(1, 200))
        off = np.cumsum(rng.integers(1, 3 * gate, n)).astype(np.int64) + 1000
        head_n = int(rng.choice([1, 2, 8, 64]))
        fresh = O.resolve_candidates(off, sps)                        # what the device gate keeps with fresh state
        sel = fresh | (np.arange(n) < head_n)                          # delivered: kept + the whole head region
        recs = np.zeros(int(sel.sum()), dtype=native.BURST_DTYPE)
        recs["offset"] = off[sel]
        recs["flags"] = (np.where(fresh[sel], 2, 0) | np.where(np.arange(n)[sel] < head_n, 16, 0)).astype(np.uint16)
        eob_in = int(rng.choice([native.EOB_NONE, 900, 1000 + int(rng.integers(0, 4 * gate)), int(off[min(n - 1, 3)]) + gate]))
        want = off[O.resolve_candidates(off, sps, prev_eob=eob_in)]
        got = native.shard_fixup(recs, sps, eob_in)
        predicted = native.shard_head_sync(recs, sps) > eob_in
        all_head = head_n >= n
        # the prediction is exact unless the whole shard lies in its head region: such a shard's fix-up cannot
        # fail, but the prediction must still say "no" when its tail depends on eob_in (test_shard_stitch_property.py)
        assert predicted == (got is not None) or (all_head and got is not None and not predicted)
        if got is not None:
            assert np.array_equal(got["offset"], want)
            assert np.all(got["flags"] & 2) and not np.any(got["flags"] & 16)
            hits += 1
        else:
            misses += 1
    assert hits > 100 and misses > 10


def test_kernel_resources_keep_the_tail_co_resident():
    """The pipelined pass rate depends on a resource fact, not only on code: k_detect fills a CU with resident wavefronts
    -- five workgroups of four (complex64, |IQ|^2 floats, int16), or 21 workgroups of one (the 8-bit formats, round 5: six
    1280-byte LDS granules per wavefront) -- and the sparse tail kernels of the PREVIOUS pass run beside them.  If they do not
    fit in what those leave free, they run when k_detect drains and the next k_detect starts with some of its workgroups in
    a second round (measured in round 2: 1.89 instead of 1.51 ms per pass when k_detect took 96 VGPRs).  The compiler's own
    report (written by gr_adsb_amd.build) is checked against those limits here."""
    import json
    from gr_adsb_amd import build as B
    B.build()
    res = json.load(open(B.RES))
    LDS_CU, VGPR_SIMD, SIMDS, GRAN = 160 * 1024, 512, 4, 1280
    alloc = lambda v: -(-v // 8) * 8                                    # noqa: E731  (allocation granule: 8 VGPRs)
    gran = lambda b: -(-b // GRAN) * GRAN                               # noqa: E731  (LDS allocation granule on gfx950)
    detect = {k: v for k, v in res.items() if "k_detect" in k}
    assert len(detect) == 35            # (5 input formats + int8 and uint8 with a power-of-two scale) x 5 samples-per-chip instances
    tail = {k: v for k, v in res.items() if any(t in k for t in ("k_order", "k_resolve", "k_count", "k_compact"))}
    assert len(tail) == 8                                               # k_order per input format + three format-blind kernels
    for name, d in detect.items():
        assert d["scratch_bytes_per_lane"] == 0 and d["vgpr_spills"] == 0, name + ": spills in the streaming kernel"
        mode = int(re.search(r"k_detectILi(\d)E", name).group(1))
        wpb = 1 if mode in (3, 4, 5, 6) else 4                          # adsb_device.h: det_waves
        wg_cu = min(LDS_CU // gran(d["lds_bytes_per_block"]), SIMDS * (VGPR_SIMD // alloc(d["vgprs"])) // wpb, 32)
        assert wg_cu == (21 if wpb == 1 else 5), (name, wg_cu)            # 8-bit: six LDS granules per wavefront
        free_lds = LDS_CU - wg_cu * gran(d["lds_bytes_per_block"])
        # resident k_detect wavefronts per SIMD: five of a workgroup of four each; 21 single ones = 6 + 5 + 5 + 5
        per_simd = [6, 5, 5, 5] if wpb == 1 else [5, 5, 5, 5]
        for tname, t in tail.items():
            assert gran(t["lds_bytes_per_block"]) <= free_lds, "%s (%d B LDS) does not fit beside %s" % (tname, t["lds_bytes_per_block"], name)
            assert t["scratch_bytes_per_lane"] == 0, tname
            # the four wavefronts of a tail workgroup need register room somewhere on the CU.  (k_order is the exception by
            # design: it is launched with 8 KB of unused dynamic LDS so that it starts when k_detect drains -- adsb_hip.hip:
            # enqueue_tail -- and the kernels behind it follow in stream order)
            slots = sum((VGPR_SIMD - w * alloc(d["vgprs"])) // alloc(t["vgprs"]) for w in per_simd)
            if "k_order" not in tname:
                assert slots >= 4, "%s (%d VGPRs) does not fit beside %s (%d)" % (tname, t["vgprs"], name, d["vgprs"])


def test_chunk_plan_covers_the_call_and_keeps_its_limits(native):
    """adsb_plan_chunks (pure host arithmetic, the function enqueue() cuts every call with): the chunks cover the call,
    one resident round is the floor, a bulk call gets at most eight rounds of chunks no shorter than four tiles, and more
    samples never mean fewer than half the chunks."""
    T = 1024
    for resident in (4, 1024, 4096, 5120):
        prev_units = 0
        for n in [1, 100, T, T + 1, 37 * T, resident * T - 1, resident * T, resident * T + 1, 3 * resident * T + 5,
                  4 * resident * T, 4 * resident * T + 1, 9 * resident * T, 31 * resident * T + 7, 32 * resident * T,
                  33 * resident * T, 200 * resident * T + 123, 1 << 30, (1 << 31) + (1 << 20) + 37]:
            units, per = native.plan_chunks(n, resident)
            ntiles = -(-n // T)
            assert per % T == 0 and per >= T
            assert units * per >= n and (units - 1) * per < ntiles * T          # covers, and no empty trailing chunk
            assert units <= max(ntiles, 1) and units <= 8 * resident
            if ntiles <= resident:
                assert units == ntiles and per == T                             # one tile per wavefront
            if units > resident:
                assert per >= 4 * T                                             # several rounds: chunks of >= 4 tiles
            if ntiles >= 32 * resident:
                assert units > 6 * resident                                     # a bulk call uses (about) eight rounds
            if n >= resident * T:                                               # (chunks are whole tiles of one length: just over a
                assert 2 * units >= prev_units                                  #  round, two tiles each need half the wavefronts)
            prev_units = units
    # the bench sizes on an MI355X context (256 CUs x 5 workgroups x 4 wavefronts)
    assert native.plan_chunks(1 << 30, 5120) == (40330, 26 * T)
    assert native.plan_chunks(1 << 26, 5120)[1] == 5 * T
    with pytest.raises(native.AdsbError):
        native.plan_chunks(-1, 5120)
    with pytest.raises(native.AdsbError):
        native.plan_chunks(1 << 20, 0)


def test_shard_bounds_is_the_python_tiling():
    """adsb_shard_bounds (round 5: the tiling the C sharded driver uses) == frontend.shard_plan (what bench.py's ranks and
    file replay use), on the CPU: pure host arithmetic."""
    import random
    from gr_adsb_amd import _native as native
    from gr_adsb_amd.frontend import shard_plan
    rnd = random.Random(5)
    for _ in range(3000):
        L = rnd.choice([0, 1, 5, 4095, 4096, 4097, rnd.randrange(1, 1 << 34)])
        k, sps, al = rnd.randrange(1, 17), rnd.choice([2, 4, 8, 20, 100]), rnd.choice([4, 4096, 1 << 20])
        for g, pl in enumerate(shard_plan(L, k, sps, align=al)):
            assert native.shard_bounds(L, k, g, sps, al) == (pl["own_lo"], pl["own_hi"], pl["lo"], pl["hi"])
    for bad in ((-1, 1, 0, 2, 4), (10, 0, 0, 2, 4), (10, 2, 2, 2, 4), (10, 2, 0, 1, 4), (10, 2, 0, 2, 0)):
        with pytest.raises(native.AdsbError):
            native.shard_bounds(*bad)
