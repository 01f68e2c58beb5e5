"""the SQLite PDU record format of the reference's playback graph (examples/adsb_playback.grc:
table `demodulated`, columns `timestamp` / `data`).  Host logic only; records come from the emulated device code."""
import os
import sqlite3

import numpy as np

import simlib
from gr_adsb_amd import _native, blocks, pdu_store
from gr_adsb_amd import modulator as M


def _records():
    iq = M.synth_iq(1 << 16, 2e6, 4000, seed=31, df_choices=(11, 17), df_weights=(0.4, 0.6))
    recs, _ = simlib.sim_canonical(0, iq, 2e6, 0.01)
    return recs


def test_round_trip_and_schema(tmp_path):
    recs = _records()
    fn = os.path.join(tmp_path, "adsb.db")
    sink = pdu_store.PduSqliteSink(fn)
    n = sink.write_bursts(recs, 2e6, start_timestamp=1.5e9)
    sink.close()
    dem = (recs["flags"] & 1) != 0
    assert n == dem.sum() > 50
    conn = sqlite3.connect(fn)
    cols = [r[1] for r in conn.execute("PRAGMA table_info(demodulated)")]
    assert cols[0] == "timestamp" and "data" in cols                # the names adsb_playback.grc:186-203 configures
    assert conn.execute("SELECT COUNT(*) FROM demodulated").fetchone()[0] == n
    conn.close()
    back = pdu_store.read_pdus(fn)
    snr = _native.snr_db(recs["peak"], recs["median"])[dem]
    bits = _native.unpack_bits(recs["bits"])[dem][:, :112]
    want = [blocks.make_pdu(1.5e9, 2e6, int(o), s, b) for o, s, b in zip(recs["offset"][dem], snr, bits)]
    assert len(back) == len(want)
    for (m0, v0), (m1, v1) in zip(back, want):
        assert m0["timestamp"] == m1["timestamp"] and np.float32(m0["snr"]) == np.float32(m1["snr"])
        assert v0.dtype == np.uint8 and np.array_equal(v0, v1)


def test_block_pdus_can_be_recorded(tmp_path):
    recs = _records()
    dem = (recs["flags"] & 1) != 0
    fn = os.path.join(tmp_path, "pdus.db")
    sink = pdu_store.PduSqliteSink(fn)
    snr = _native.snr_db(recs["peak"], recs["median"])
    for o, s, b in zip(recs["offset"][dem], snr[dem], _native.unpack_bits(recs["bits"])[dem][:, :112]):
        sink.write_pdu(blocks.make_pdu(0.0, 2e6, int(o), s, b))
    sink.close()
    assert len(pdu_store.read_pdus(fn)) == dem.sum()
