/* adsb_oracle.c -- CPU ORACLE (test infrastructure, NOT product code): a scalar C restatement of the
 * gr-adsb framer + demod hot path in canonical whole-buffer mode (one framer.work() call over a
 * fresh stream, one demod.work() call over the same samples, all tags delivered).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; nothing under
 * gr_adsb_amd/ does.  It is pinned against oracle/adsb_oracle.py (which is pinned against the real
 * reference, see that file's header) by tests/test_oracle_golden.py, and used where the NumPy oracle is
 * too slow: full-size parity checks and the single-core CPU baseline ("kind": "port").
 *
 * Citations are to /root/reference/python/adsb/.  |IQ|^2 (GNU Radio complex_to_mag_squared,
 * examples/adsb_rx.py:180) is parity-unpinned by the reference's tests and defined as float32
 * re*re + im*im with separately rounded products; build with -ffp-contract=off.
 *
 * Output record layout (32 bytes) matches include/adsb_hip.h's adsb_burst so results compare bytewise.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int64_t offset;
  float peak;
  float median;
  uint8_t bits[14];
  uint16_t flags; /* 1 = demodulated (PDU published), 2 = kept (tag emitted) */
} orec;

static const int TEMPLATE[16] = {1, 0, 1, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0}; /* framer.py:50 */

/* np.median of w[0:n] in float32: odd -> middle, even -> f32(a+b)/2, NaN if n == 0 or any NaN.  A median of zero is
 * +0.0 whatever the signs of the zeros in the window: np.median takes np.mean of the middle element(s), whose sum starts
 * from +0.0 (0.0 + -0.0 = +0.0) -- so -0.0 vs +0.0 never depends on how the partition ordered them. */
static float median_f32(const float* w, int n) {
  float s[100];
  if (n <= 0) { union { uint32_t u; float f; } q; q.u = 0xFFC00000u; return q.f; }  /* np.median([]) = 0/0 */
  for (int i = 0; i < n; ++i) {
    if (w[i] != w[i]) return NAN;
    float v = w[i];
    int j = i;
    while (j > 0 && s[j - 1] > v) { s[j] = s[j - 1]; --j; }
    s[j] = v;
  }
  volatile float zero = 0.0f;
  if (n & 1) { volatile float m = s[n / 2] + zero; return m; }
  volatile float sum = s[n / 2 - 1] + s[n / 2];
  volatile float h = sum / 2.0f;
  volatile float m2 = h + zero;
  return m2;
}

void oracle_mag2(const float* iq, int64_t n, float* out) {
  for (int64_t i = 0; i < n; ++i) {
    volatile float a = iq[2 * i] * iq[2 * i];
    volatile float b = iq[2 * i + 1] * iq[2 * i + 1];
    out[i] = a + b;
  }
}

/* Opt-in, NOT the reference: the length-aware gate of ADSB_FLAG_LONG_AWARE_GATE -- 119*sps after a
 * burst whose first data bit is set (demod.py:87-95, k = 0; samples past the end read as 0), 63*sps otherwise.  Such
 * records carry flag 0x2000 like the device's. */
static int g_long_aware = 0;
void oracle_set_long_aware(int v) { g_long_aware = v; }

/* x: float32 |IQ|^2 stream of length n.  Returns number of tags written (or -needed if cap too small;
 * the first `cap` are still written).  If cands != NULL every matched centre's stream offset (before the
 * re-trigger gate) is stored there, up to cand_cap, count in *n_cands. */
int64_t oracle_canonical(const float* x, int64_t n, int sps, float thr, int64_t abs_offset, orec* out, int64_t cap,
                         int64_t* cands, int64_t cand_cap, int64_t* n_cands) {
  const int64_t H = 8 * (int64_t)sps;       /* framer.py:61 */
  const int half = sps / 2;
  const int64_t N = n;                      /* one call: N = len(out0) */
  float* in0 = (float*)calloc((size_t)(n + H), sizeof(float));
  if (!in0) return -1;
  memcpy(in0 + (H - 1), x, (size_t)n * sizeof(float));   /* history of H-1 zeros, then the stream */
  int64_t ntags = 0, ncand = 0;
  int64_t eob = -1;                          /* framer.py:57 */
  int prev = (0.0f >= thr);                  /* framer.py:54,84: prev_in0 = 0 */
  int64_t rise = -1;
  for (int64_t j = 0; j < N; ++j) {
    const int a = (in0[j] >= thr);           /* framer.py:84 (float32 compare; NaN -> 0) */
    if (a && !prev) rise = j;                /* framer.py:92 */
    if (!a && prev && rise >= 0) {           /* framer.py:93,98-100: a fall with no rise in this call is dropped */
      const int64_t p = (rise + j) / 2;      /* framer.py:113 */
      rise = -1;
      if (p > eob) {                         /* framer.py:121 */
        eob = -1;                            /* framer.py:123 */
        const float hp = in0[p] / 2.0f;      /* framer.py:141 */
        int ok = 1;
        for (int k = 0; k < 16; ++k) {       /* framer.py:137-147 */
          const int chip = in0[p + (int64_t)k * half] > hp;
          if (chip != TEMPLATE[k]) { ok = 0; break; }
        }
        if (ok) {
          const int64_t lo = p < 100 ? 0 : p - 100;               /* framer.py:156-159 */
          const float med = median_f32(in0 + lo, (int)(p - lo));
          eob = p + 63 * (int64_t)sps;                            /* framer.py:165 */
          int lng = 0;
          if (g_long_aware) {
            const int64_t i1 = p + 8 * (int64_t)sps, i0 = i1 + half;
            const float b1 = i1 < n + H - 1 ? in0[i1] : 0.0f, b0 = i0 < n + H - 1 ? in0[i0] : 0.0f;
            if (b1 > b0) { lng = 1; eob = p + 119 * (int64_t)sps; }
          }
          const int64_t off = abs_offset - (H - 1) + p;           /* framer.py:170 */
          if (ntags < cap) {
            orec* r = &out[ntags];
            memset(r, 0, sizeof(*r));
            r->offset = off; r->peak = in0[p]; r->median = med; r->flags = (uint16_t)(2 | (lng ? 0x2000 : 0));
            /* demod.py:75-95 on the stream itself (its in0 has no history) */
            const int64_t s_off = off - abs_offset;               /* index into x */
            const double eobd = (double)s_off + 119.0 * sps + sps / 2.0;   /* demod.py:76,80 */
            if (eobd < (double)n) {                               /* demod.py:82 */
              const int64_t sob = s_off + 8 * (int64_t)sps;
              for (int k = 0; k < 112; ++k) {
                const float b1 = x[sob + (int64_t)k * sps];
                const float b0 = x[sob + half + (int64_t)k * sps];
                if (b1 > b0) r->bits[k >> 3] |= (uint8_t)(0x80u >> (k & 7));
              }
              r->flags |= 1;
            }
          }
          ++ntags;
        }
      }
      if (cands) {                           /* every matched centre, gated or not (shard-stitch checks) */
        const float hp = in0[p] / 2.0f;
        int ok = 1;
        for (int k = 0; k < 16; ++k) {
          const int chip = in0[p + (int64_t)k * half] > hp;
          if (chip != TEMPLATE[k]) { ok = 0; break; }
        }
        if (ok) { if (ncand < cand_cap) cands[ncand] = abs_offset - (H - 1) + p; ++ncand; }
      }
    }
    prev = a;
  }
  free(in0);
  if (n_cands) *n_cands = ncand;
  return ntags <= cap ? ntags : -ntags;
}

/* complex64 in -> tags, the whole CPU path as the GPU library runs it (baseline timing entry point) */
int64_t oracle_process_iq(const float* iq, int64_t n, int sps, float thr, int64_t abs_offset, orec* out, int64_t cap) {
  float* x = (float*)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
  if (!x) return -1;
  oracle_mag2(iq, n, x);
  int64_t r = oracle_canonical(x, n, sps, thr, abs_offset, out, cap, 0, 0, 0);
  free(x);
  return r;
}

/* ---- Mode S parity pre-filter: decoder.py:551 (DF), :560-688 (check_parity), :693-714
 * (compute_crc: bit-serial division by the 25-coefficient generator of decoder.py:269).  bits14 = the 112
 * PDU bits packed MSB first.  Returns crc(bits[0:L-24]) ^ bits[L-24:L] for the DF's length L (56-bit
 * reading for DFs the decoder does not know); *flags_out = PARITY_OK 32 | LONG 64 | KNOWN_DF 128 | DF<<8. */
uint32_t oracle_mode_s_parity(const uint8_t* bits14, int* df_out, int* nbits_out, unsigned* flags_out) {
  static const int POLY[25] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 1};
  int b[112 + 24];
  for (int k = 0; k < 112; ++k) b[k] = (bits14[k >> 3] >> (7 - (k & 7))) & 1;
  int df = 0;
  for (int k = 0; k < 5; ++k) df = (df << 1) | b[k];
  const int is_long = df == 16 || df == 17 || df == 18 || df == 19 || df == 20 || df == 21 || df == 24;
  const int is_short = df == 0 || df == 4 || df == 5 || df == 11;
  const int L = is_long ? 112 : 56;
  int w[112];
  for (int k = 0; k < L - 24; ++k) w[k] = b[k];
  for (int k = L - 24; k < L; ++k) w[k] = 0;                       /* decoder.py:704 */
  for (int i = 0; i < L - 24; ++i)                                 /* decoder.py:706-711 */
    if (w[i]) for (int k = 0; k < 25; ++k) w[i + k] ^= POLY[k];
  uint32_t syn = 0;
  for (int k = L - 24; k < L; ++k) syn = (syn << 1) | (uint32_t)(w[k] ^ b[k]);
  const int pi = df == 11 || df == 17 || df == 18 || df == 19;
  if (df_out) *df_out = df;
  if (nbits_out) *nbits_out = is_long ? 112 : (is_short ? 56 : 0);
  if (flags_out) *flags_out = ((unsigned)df << 8) | (is_long ? 64u : 0u) | ((is_long || is_short) ? 128u : 0u) | ((pi && syn == 0) ? 32u : 0u);
  return syn;
}
