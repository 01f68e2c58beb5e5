"""Host-only exactness of the multi-GPU stitch (gr_adsb_amd/sharding.finish_shard + the C ABI's adsb_shard_fixup /
adsb_stitch): for ANY partition of ANY matched-centre list into shards -- down to shards that hold a single
centre or none -- the per-rank results concatenated equal ONE sequential gate over the whole list
(framer.py:121-123,165).  The GPU pass of a rank is replaced by its definition (fresh-state gate over the shard's
centres + the first HEAD_CANDS centres delivered ungated), everything after it is the product code that
bench.py --gpus N runs.

Round-1 bug pinned here: a shard lying entirely inside its head region published its fresh-state tail although
its true tail depended on the incoming eob."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from gr_adsb_amd import _native as N
from gr_adsb_amd import sharding


@pytest.fixture(scope="module")
def native():
    N.load()
    return N


def device_shard(offs, long_hint, sps, head_n):
    """What adsb_shard_device(head_cands=head_n) returns for a shard whose matched centres are `offs`: the gate run
    from fresh state (KEPT), the first head_n centres delivered whether kept or not (HEAD)."""
    n = len(offs)
    gate = np.where(long_hint, 119, 63) * sps
    kept = np.zeros(n, dtype=bool)
    eob = -(1 << 61)
    for i in range(n):
        if offs[i] > eob:
            kept[i] = True
            eob = int(offs[i]) + int(gate[i])
    head = np.arange(n) < head_n
    sel = kept | head
    recs = np.zeros(int(sel.sum()), dtype=N.BURST_DTYPE)
    recs["offset"] = offs[sel]
    recs["flags"] = (np.where(kept[sel], N.BURST_KEPT, 0) | np.where(head[sel], N.BURST_HEAD, 0)
                     | np.where(long_hint[sel], N.BURST_LONG_HINT, 0)).astype(np.uint16)
    return recs


def ungated_shard(offs, long_hint):
    recs = np.zeros(len(offs), dtype=N.BURST_DTYPE)
    recs["offset"] = offs
    recs["flags"] = np.where(long_hint, N.BURST_LONG_HINT, 0).astype(np.uint16)
    return recs


def run_world(offs, long_hint, cuts, sps, head_n):
    """Partition by owner ranges (cuts = sorted stream offsets where a new shard begins), run every rank's
    finish_shard with in-process stand-ins for the two collectives, return (concatenated result, fallbacks)."""
    edges = [-(1 << 60)] + list(cuts) + [1 << 60]
    parts = [(offs >= lo) & (offs < hi) for lo, hi in zip(edges[:-1], edges[1:])]
    world = len(parts)
    dev = [device_shard(offs[m], long_hint[m], sps, head_n) for m in parts]
    ung = [ungated_shard(offs[m], long_hint[m]) for m in parts]
    pairs = [(N.shard_tail(r, sps), N.shard_head_sync(r, sps)) for r in dev]
    before = sharding.STATS["fallbacks"]
    out = []
    for rank in range(world):
        def ag_pair(pair, rank=rank):
            assert pair == pairs[rank]
            return pairs
        out.append(sharding.finish_shard(dev[rank], sps, rank, ag_pair, lambda rank=rank: ung[rank], lambda o: ung))
    fb = sharding.STATS["fallbacks"] - before
    assert fb in (0, world)                         # every rank takes the same decision
    return np.concatenate(out) if out else np.zeros(0, dtype=N.BURST_DTYPE), fb // max(1, world)


def truth(offs, long_hint, sps):
    c = ungated_shard(offs, long_hint)
    return N.stitch(c, sps)


def test_judge_repro_three_short_shards(native):
    """shards [950], [1010,1090], [1150] at sps=2 -- one call keeps 950 and 1090 (1150 lies inside
    1090's window); round 1 also kept 1150 because shard 1 published a tail computed from fresh state."""
    offs = np.array([950, 1010, 1090, 1150], dtype=np.int64)
    lh = np.zeros(4, dtype=bool)
    got, fb = run_world(offs, lh, [1000, 1100], 2, sharding.HEAD_CANDS)
    assert got["offset"].tolist() == [950, 1090] == truth(offs, lh, 2)["offset"].tolist()
    assert fb == 1                                  # detected from the gathered pairs: full-candidate fallback
    # the advisor's variant
    offs = np.array([1000, 1110, 1200, 1300], dtype=np.int64)
    got, _ = run_world(offs, lh, [1050, 1250], 2, sharding.HEAD_CANDS)
    assert got["offset"].tolist() == [1000, 1200] == truth(offs, lh, 2)["offset"].tolist()


def test_all_head_shard_with_a_chain_head_needs_no_fallback(native):
    # shard 1 = [1010 (suppressed by 950), 1500 (its own chain head, beyond the incoming eob 1076)]: fix-up + exact tail
    offs = np.array([950, 1010, 1500, 1700], dtype=np.int64)
    lh = np.zeros(4, dtype=bool)
    got, fb = run_world(offs, lh, [1000, 1550], 2, sharding.HEAD_CANDS)
    assert got["offset"].tolist() == [950, 1500, 1700] and fb == 0
    # empty shards in between pass the state through
    got, fb = run_world(offs, lh, [960, 970, 980, 1000, 1550, 1560], 2, sharding.HEAD_CANDS)
    assert got["offset"].tolist() == [950, 1500, 1700] and fb == 0


@st.composite
def worlds(draw):
    sps = draw(st.sampled_from([2, 4, 8, 20]))
    gate = 63 * sps
    n = draw(st.integers(0, 60))
    # gaps around the two gate windows, so that chains, chain breaks and exact ties all occur
    gaps = draw(st.lists(st.one_of(st.integers(1, 3), st.integers(gate - 2, gate + 2), st.integers(119 * sps - 2, 119 * sps + 2),
                                   st.integers(1, 3 * gate)), min_size=n, max_size=n))
    offs = np.cumsum(np.asarray(gaps, dtype=np.int64)) + 1000 if n else np.zeros(0, dtype=np.int64)
    long_hint = np.asarray(draw(st.lists(st.booleans(), min_size=n, max_size=n)), dtype=bool) if draw(st.booleans()) \
        else np.zeros(n, dtype=bool)
    hi = int(offs[-1]) + 10 if n else 2000
    # shard boundaries anywhere, including shards one sample long
    ncut = draw(st.integers(0, 12))
    cuts = sorted(set(draw(st.lists(st.integers(990, hi), min_size=ncut, max_size=ncut))))
    head_n = draw(st.sampled_from([1, 2, 3, 8, 64]))
    return offs, long_hint, cuts, sps, head_n


@settings(max_examples=600, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(worlds())
def test_any_partition_equals_one_sequential_gate(native, w):
    offs, long_hint, cuts, sps, head_n = w
    got, _ = run_world(offs, long_hint, cuts, sps, head_n)
    want = truth(offs, long_hint, sps)
    assert got["offset"].tolist() == want["offset"].tolist()
    assert np.all(got["flags"] & N.BURST_KEPT) and not np.any(got["flags"] & N.BURST_HEAD)


def test_one_sample_shards_exhaustively(native):
    """Every centre its own shard (and every second shard empty): the worst case for tail hand-over."""
    rng = np.random.default_rng(11)
    for sps in (2, 8):
        gate = 63 * sps
        for trial in range(60):
            n = int(rng.integers(1, 40))
            offs = np.cumsum(rng.integers(1, 2 * gate, n)).astype(np.int64) + 1000
            lh = np.zeros(n, dtype=bool)
            cuts = sorted(set(offs.tolist()) | set((offs + 1).tolist()))
            got, _ = run_world(offs, lh, cuts, sps, 64)
            assert got["offset"].tolist() == truth(offs, lh, sps)["offset"].tolist()
