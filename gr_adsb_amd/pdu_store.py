"""Record / replay of demodulated PDUs in the SQLite layout the reference's playback flowgraph reads
(SURVEY.md §8f-3): table `demodulated`, one row per PDU, metadata keys as columns (`timestamp`, `snr`) and the
112-entry u8vector as a BLOB column `data` -- the names `examples/adsb_playback.grc:163-203` configures on its
sqlite_timed_source (table_name "demodulated", timestamp_column_name "timestamp", vector_column_name "data").

The block that writes/reads those files in the reference setup is the third-party gr-sqlite project (README.md:
134-138), which is not vendored in the reference: the on-disk encoding below (REAL columns, BLOB of 112 bytes
of 0/1) follows the playback graph's column names only -- parity UNPINNED beyond that.  Host-side convenience
around the hot path, no GPU involved: PDUs come from gr_adsb_amd.blocks.demod / FrontEnd results.
"""
import sqlite3

import numpy as np

from . import _native
from .blocks import make_pdu, pmt

TABLE = "demodulated"
TIMESTAMP_COLUMN = "timestamp"
VECTOR_COLUMN = "data"
INSERTS_PER_TRANSACTION = 50      # the reference decoder's own batching constant (decoder.py:261)


class PduSqliteSink:
    def __init__(self, filename, table_name=TABLE):
        self.table = table_name
        self.conn = sqlite3.connect(filename)
        self.conn.execute("CREATE TABLE IF NOT EXISTS %s (%s REAL, snr REAL, %s BLOB)" % (self.table, TIMESTAMP_COLUMN, VECTOR_COLUMN))
        self._pending = 0

    def write(self, timestamp, snr, bits112):
        b = np.ascontiguousarray(bits112, dtype=np.uint8)
        assert b.size == 112
        self.conn.execute("INSERT INTO %s VALUES (?, ?, ?)" % self.table, (float(timestamp), float(snr), b.tobytes()))
        self._pending += 1
        if self._pending >= INSERTS_PER_TRANSACTION:
            self.flush()

    def write_pdu(self, pdu):
        """pdu: (meta dict {timestamp, snr}, u8 vector[112]) as published on the `demodulated` port (demod.py:104-110)."""
        meta, vec = pmt.to_python(pmt.car(pdu)), pmt.to_python(pmt.cdr(pdu))
        self.write(meta["timestamp"], meta["snr"], vec)

    def write_bursts(self, bursts, fs, start_timestamp=0.0):
        """Every burst of a FrontEnd / C-ABI result that carries a PDU (ADSB_BURST_DEMOD)."""
        dem = (bursts["flags"] & _native.BURST_DEMOD) != 0
        snr = _native.snr_db(bursts["peak"], bursts["median"])
        bits = _native.unpack_bits(bursts["bits"])[:, :112]
        for o, s, b in zip(bursts["offset"][dem], snr[dem], bits[dem]):
            self.write(start_timestamp + int(o) / fs, s, b)
        return int(dem.sum())

    def flush(self):
        self.conn.commit()
        self._pending = 0

    def close(self):
        self.flush()
        self.conn.close()


def read_pdus(filename, table_name=TABLE):
    """PDUs in timestamp order, in the shape the decoder's `demodulated` handler takes (decoder.py:330-335)."""
    conn = sqlite3.connect(filename)
    rows = conn.execute("SELECT %s, snr, %s FROM %s ORDER BY %s, rowid" % (TIMESTAMP_COLUMN, VECTOR_COLUMN, table_name, TIMESTAMP_COLUMN)).fetchall()
    conn.close()
    return [make_pdu(0.0, 1.0, ts, snr, np.frombuffer(blob, dtype=np.uint8).copy()) for ts, snr, blob in rows]
