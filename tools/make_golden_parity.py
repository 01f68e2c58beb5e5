"""Container-only: generate tests/golden/g_parity.npz -- known answers of the REFERENCE decoder's header /
parity stage (decoder.py:550-556 decode_header, :560-688 check_parity, empty aircraft table) for a set of
112-bit PDUs: valid synthetic replies of every downlink format, the same with 1-3 flipped bits, random
bits, and the PDUs of the committed front-end goldens.  Data only: inputs (packed bits) and the reference's
outputs (df, payload_length, check_parity()'s return value, the announced address `aa`)."""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_harness as R                     # noqa: E402
from gr_adsb_amd import modulator as M      # noqa: E402


def main():
    rng = np.random.default_rng(20240926)
    rows = []
    for df in list(range(32)):
        for _ in range(6):
            f = M.make_frame(df, rng)
            if df == 19:                      # the decoder treats DF19 as a parity/interrogator format
                f[88:] = [(M.crc24(f[:88]) >> (23 - k)) & 1 for k in range(24)]
            bits = np.zeros(112, np.uint8)
            bits[:len(f)] = f
            if len(f) < 112:
                bits[len(f):] = rng.integers(0, 2, 112 - len(f))
            rows.append(bits.copy())
            g = bits.copy()
            g[rng.integers(0, 112, rng.integers(1, 4))] ^= 1
            rows.append(g)
    rows += list(rng.integers(0, 2, (200, 112)).astype(np.uint8))
    for fn in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "g*msps*.npz"))):
        z = np.load(fn)
        rows += list(np.unpackbits(z["single_pdu_bits"][:150], axis=1)[:, :112])
    bits = np.array(rows, dtype=np.uint8)
    dec = R.load_reference_decoder("All Messages", "None", "None")
    df = np.zeros(len(bits), np.int32); plen = np.zeros(len(bits), np.int32)
    passed = np.zeros(len(bits), np.int32); aa = np.zeros(len(bits), np.int64)
    for i, b in enumerate(bits):
        dec.reset()
        dec.bits = b.astype(int)
        dec.datetime = ""; dec.snr = 0.0; dec.timestamp = 0.0
        dec.decode_header()
        passed[i] = dec.check_parity()
        df[i] = dec.df; plen[i] = dec.payload_length; aa[i] = dec.aa
    out = os.path.join(ROOT, "tests", "golden", "g_parity.npz")
    np.savez_compressed(out, bits=np.packbits(bits, axis=1), df=df, payload_length=plen, parity_passed=passed, aa=aa)
    print(out, len(bits), "pdus; passed", int(passed.sum()), "; df hist", np.bincount(df, minlength=32).tolist())


if __name__ == "__main__":
    main()
