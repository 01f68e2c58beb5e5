"""Stand-in for the slice of the GNU Radio runtime the two blocks touch, used ONLY when `gnuradio`
is not importable (this image has no GNU Radio).  It lets the drop-in blocks in blocks.py be
constructed and driven work()-call by work()-call exactly the way the scheduler's Python gateway
would (SURVEY.md §8b B2 lists the surface).  Not a scheduler: drive() below calls work() with an
explicit chunk schedule.
"""
import bisect
import queue
import threading
import types

import numpy as np

TPP_ONE_TO_ONE = 1


class Tag:
    __slots__ = ("offset", "key", "value", "srcid")

    def __init__(self, offset, key, value, srcid=None):
        self.offset, self.key, self.value, self.srcid = offset, key, value, srcid


class sync_block:
    def __init__(self, name=None, in_sig=None, out_sig=None):
        self._name = name
        self._in_sig, self._out_sig = in_sig, out_sig
        self._history = 1
        self._nread = 0
        self._nwritten = 0
        self.tags_out = []      # tags this block added
        self.tags_in = []       # tags visible on its input (set by the driver)
        self.messages = []      # (port, msg) published
        self._ports = []

    def name(self):
        return self._name

    def set_history(self, n):
        self._history = int(n)

    def history(self):
        return self._history

    def set_tag_propagation_policy(self, p):
        self._tpp = p

    def set_output_multiple(self, n):
        self._output_multiple = int(n)

    def output_multiple(self):
        return getattr(self, "_output_multiple", 1)

    def nitems_written(self, port):
        return self._nwritten

    def nitems_read(self, port):
        return self._nread

    def add_item_tag(self, port, offset, key, value, srcid=None):
        self.tags_out.append(Tag(offset, key, value, srcid))

    def get_tags_in_range(self, port, start, end, key=None):
        # tags arrive in stream order (one upstream block appends them): a bisection instead of a scan of the whole list
        tin = self.tags_in
        n = len(tin)                          # (a threaded driver appends while this runs: only the first n are looked at)
        lo = bisect.bisect_left(tin, start, 0, n, key=_tag_offset)
        hi = bisect.bisect_left(tin, end, lo, n, key=_tag_offset)
        return [t for t in tin[lo:hi] if key is None or t.key == key]

    def message_port_register_out(self, name):
        self._ports.append(name)

    def message_port_pub(self, port, msg):
        self.messages.append((port, msg))


def _tag_offset(t):
    return t.offset


pmt = types.SimpleNamespace(
    to_pmt=lambda x: x,
    to_python=lambda x: x,
    cons=lambda a, b: (a, b),
    car=lambda p: p[0],
    cdr=lambda p: p[1],
)


def drive(framer_blk, demod_blk, x, schedule=None, demod_schedule=None):
    """Feed the float32 |IQ|^2 stream x through framer -> demod with explicit chunk schedules
    (None = one call each).  All framer tags are delivered to demod (SURVEY.md §7 canonical)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    L = len(x)
    schedule = [L] if schedule is None else list(schedule)
    demod_schedule = schedule if demod_schedule is None else list(demod_schedule)
    assert sum(schedule) == L and sum(demod_schedule) == L
    H = framer_blk.history()
    buf = np.concatenate([np.zeros(H - 1, dtype=np.float32), x])
    pos = 0
    y = np.empty(L, dtype=np.float32)             # the framer's output stream: what the demod is connected to
    for N in schedule:
        out0 = y[pos:pos + N]
        framer_blk._nread = framer_blk._nwritten = pos
        assert framer_blk.work([buf[pos:pos + N + H - 1]], [out0]) == N
        pos += N
    demod_blk.tags_in = [Tag(t.offset, t.key, t.value, t.srcid) for t in framer_blk.tags_out]
    pos = 0
    for N in demod_schedule:
        out0 = np.empty(N, dtype=np.float32)
        demod_blk._nread = demod_blk._nwritten = pos
        demod_blk.work([y[pos:pos + N]], [out0])
        pos += N
    return framer_blk.tags_out, demod_blk.messages


def drive_threaded(framer_blk, demod_blk, x, schedule, demod_lag=0, depth=4):
    """The two blocks the way GNU Radio's thread-per-block scheduler runs them: the framer's work() calls on one thread,
    the demod's on another, a bounded buffer of `depth` chunks between them (the scheduler's ring buffer), the demod
    seeing the framer's tags live (the same list, appended to while it is read).  demod_lag > 0: the demod does not
    start before the framer has finished that many calls (then `depth` grows to hold them) -- a demod that lags, which is
    when a paired demod finds the framer's slices forgotten and falls back to the device.  Both blocks get the same
    chunking, like a real sync-block chain.  Returns (tags, messages); an exception on either thread is re-raised."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    L = len(x)
    schedule = list(schedule)
    assert sum(schedule) == L
    H = framer_blk.history()
    buf = np.concatenate([np.zeros(H - 1, dtype=np.float32), x])
    y = np.empty(L, dtype=np.float32)
    q = queue.Queue(maxsize=max(depth, demod_lag + 1))
    started = threading.Event()
    errors = []
    demod_blk.tags_in = framer_blk.tags_out              # live: what TPP_ONE_TO_ONE propagation gives the demod

    def run_framer():
        try:
            pos = 0
            for k, N in enumerate(schedule):
                framer_blk._nread = framer_blk._nwritten = pos
                assert framer_blk.work([buf[pos:pos + N + H - 1]], [y[pos:pos + N]]) == N
                q.put((pos, N))
                pos += N
                if k + 1 >= demod_lag:
                    started.set()
        except BaseException as e:          # noqa: BLE001 - handed to the caller
            errors.append(e)
        finally:
            started.set()
            q.put(None)

    def run_demod():
        try:
            started.wait()
            while True:
                item = q.get()
                if item is None:
                    return
                pos, N = item
                demod_blk._nread = demod_blk._nwritten = pos
                demod_blk.work([y[pos:pos + N]], [np.empty(N, dtype=np.float32)])
        except BaseException as e:          # noqa: BLE001
            errors.append(e)
            while q.get() is not None:      # let the framer finish
                pass

    tf, td = threading.Thread(target=run_framer), threading.Thread(target=run_demod)
    tf.start(); td.start()
    tf.join(); td.join()
    if errors:
        raise errors[0]
    return framer_blk.tags_out, demod_blk.messages
