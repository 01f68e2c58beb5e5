"""Host-side pieces of the drop-in blocks that need no GPU: the framer -> demod slice hand-over (blocks._SliceStore) under
two threads, the way GNU Radio's thread-per-block scheduler runs the paired blocks, and the stand-in runtime's tag lookup."""
import threading

import numpy as np

from gr_adsb_amd import grshim
from gr_adsb_amd.blocks import _SliceStore


def _call(offs):
    offs = np.asarray(offs, dtype=np.int64)
    return offs, np.tile((offs % 251).astype(np.uint8)[:, None], (1, 14)), (offs % 7).astype(np.uint16)


def test_slice_store_hands_over_in_stream_order_and_forgets_what_was_asked_for():
    st = _SliceStore(cap=10)
    st.put(*_call([10, 20, 30])); st.put(*_call([40, 50])); st.put(*_call([60, 70, 80, 90]))
    found, bits, flags = st.take(np.array([20, 25, 40], dtype=np.int64))       # 25: a tag the framer had no bits for
    assert found.tolist() == [True, False, True] and bits[:, 0].tolist() == [20, 0, 40] and flags.tolist() == [6, 0, 5]
    assert len(st) == 5                                                        # 10, 20, 30, 40 gone; 50 kept
    found, bits, _ = st.take(np.array([50, 60, 70], dtype=np.int64))           # a demod chunk that ends inside a framer call
    assert found.all() and bits[:, 13].tolist() == [50, 60, 70] and len(st) == 2
    st.put(*_call(range(100, 130)))                                            # one call larger than the cap: never dropped ...
    assert len(st) == 30 and st.evicted == 2                                   # ... but the older ones are
    st.put(*_call([200]))
    assert len(st) == 1 and st.evicted == 32
    found, _, _ = st.take(np.array([80, 100, 200], dtype=np.int64))
    assert found.tolist() == [False, False, True] and len(st) == 0


def test_slice_store_take_few_matches_take():
    """take_few (the Python-int path of a scheduler-sized call) and take (the array path) are the same hand-over: same rows,
    same forgetting, and entries written by either kind of put() serve either kind of take."""
    rng = np.random.default_rng(3)
    offs_all = np.cumsum(rng.integers(1, 50, 400))
    for mode in range(3):
        a, b = _SliceStore(cap=10000), _SliceStore(cap=10000)
        pos = 0
        while pos < len(offs_all):
            k = int(rng.integers(1, 9))
            chunk = offs_all[pos:pos + k]
            for st in (a, b):
                st.put(*_call(chunk), chunk.tolist() if (mode + pos) % 2 else None)
            pos += k
        pos = 0
        while pos < len(offs_all):
            k = int(rng.integers(1, 7))
            want = offs_all[pos:pos + k].copy()
            if mode == 2 and len(want) > 1:
                want[0] -= 1 if pos and want[0] - 1 != offs_all[pos - 1] else 0     # a tag the framer had no bits for
            found, bits, flags = a.take(want.astype(np.int64))
            few = b.take_few(want.tolist())
            assert [g is not None for g in few] == found.tolist()
            for j, g in enumerate(few):
                if g is not None:
                    assert np.array_equal(g[0], bits[j]) and g[1] == int(flags[j])
            assert len(a) == len(b)
            pos += k


def test_slice_store_under_two_threads():
    """put() on one thread, take() on another (blocks.py: framer.work / demod.work under GNU Radio's scheduler): every row
    handed over is the row that was stored, nothing raises, and whatever was not found had been evicted or not yet stored."""
    st = _SliceStore(cap=500)
    n_calls, per = 3000, 7
    done = threading.Event()
    errors = []

    def producer():
        try:
            for k in range(n_calls):
                st.put(*_call(np.arange(per) * 3 + k * 100))
        except BaseException as e:      # noqa: BLE001
            errors.append(e)
        finally:
            done.set()

    got = [0]

    def consumer():
        try:
            k = 0
            while k < n_calls:
                offs = np.arange(per, dtype=np.int64) * 3 + k * 100
                found, bits, flags = st.take(offs)
                assert np.array_equal(bits[found, 5], (offs[found] % 251).astype(np.uint8))
                assert np.array_equal(flags[found], (offs[found] % 7).astype(np.uint16))
                got[0] += int(found.sum())
                k += 1 if (found.any() or done.is_set()) else 0    # not stored yet: ask again (a demod never runs ahead)
        except BaseException as e:      # noqa: BLE001
            errors.append(e)

    tp, tc = threading.Thread(target=producer), threading.Thread(target=consumer)
    tc.start(); tp.start(); tp.join(); tc.join()
    assert not errors, errors
    assert got[0] + st.evicted + len(st) <= n_calls * per and got[0] > 0


def test_stand_in_runtime_finds_tags_by_bisection():
    blk = grshim.sync_block("b", None, None)
    blk.tags_in = [grshim.Tag(o, "burst" if o % 20 else "other", ("SOB", 1.0)) for o in range(0, 1000, 10)]
    got = blk.get_tags_in_range(0, 95, 305, "burst")
    assert [t.offset for t in got] == [o for o in range(100, 305, 10) if o % 20]
    assert [t.offset for t in blk.get_tags_in_range(0, 0, 10)] == [0] and blk.get_tags_in_range(0, 990, 5000, "burst")[0].offset == 990
    assert blk.get_tags_in_range(0, 1000, 2000) == []
