// adsb_device.h -- CDNA4 (gfx950) device code for the ADS-B front end.
//
//   k_detect   streams the IQ once: |IQ|^2 -> threshold bitmask (one __ballot per 64 samples) -> rises by
//              mask algebra -> pulse centre -> 16-chip preamble test -> for every match, while its samples are
//              still in the wavefront's LDS window, the complete burst record (peak, median-of-100 noise,
//              112-bit PPM slice, Mode S parity pre-filter): the stream is never read a second time
//   k_longrun  (rare) pulses longer than the LDS window
//   k_scan / k_gather / k_resolve / k_count / k_compact / k_publish
//              order the centres and their records, apply the re-trigger gate as parallel chain walks, compact
//              the survivors' records, hand the summary to the host
//   k_slice    PPM slice (+ confidence ratio) for a tag list: the stand-alone demod block, and the opt-in
//              confidence output of the fused path
//
// Written for 64-wide wavefronts: the threshold mask of 64 samples IS a ballot, rises/falls are 64-bit
// mask algebra, compaction is ballot + prefix-popcount, per-burst work is done by a whole wavefront.
// Memory-bound integer/compare work: no MFMA anywhere.
//
// Behaviour restated from the reference (citations to /root/reference/python/adsb/):
//   threshold / edges / pairing   framer.py:83-113      preamble test     framer.py:137-147
//   SNR inputs (peak, median)     framer.py:156-159     re-trigger gate   framer.py:121-123,165
//   PPM slice                     demod.py:75-95        confidence ratio  demod.py:101
//
// This header contains device code only and includes nothing.  The includer provides the HIP device
// environment plus `adsb_wave_sync()` (a wavefront-level execution/LDS ordering point), `adsb_uniform(int)`
// (marks a wavefront-uniform value so it lives in a scalar register), `adsb_readlane(int, lane)` and
// `adsb_bitrep32(u32) -> u64` (every bit doubled: s_bitreplicate_b64_b32), `adsb_opaque(int)` (returns its argument
// through an empty asm statement, so that nothing derived from it is treated as loop invariant) and
// `adsb_ld_stream<Q>(const char*)` (one Q-sized load of streamed, single-use data), and the macro ADSB_LDS (the
// address-space qualifier of workgroup-local memory, empty for the emulator): the
// product translation unit adsb_hip.hip maps them to __builtin_amdgcn_wave_barrier() / _readfirstlane() /
// _readlane(); tests/sim/sim_driver.cpp includes the test-only SIMT emulator instead, so the very same
// kernels run on a machine without a GPU.
#pragma once

namespace adsb {

constexpr int kThreads = 256;            // 4 wavefronts per workgroup
constexpr int kWaves = kThreads / 64;
constexpr int kWTile = 1024;             // samples a wavefront owns per tile iteration (16 ballot words)
constexpr int kFwd = 256;                // forward halo kept in LDS behind every tile
constexpr int kBack = 128;               // back halo kept in LDS in front of every tile: the <=100-sample noise window (framer.py:156)
constexpr int kWWin = kWTile + kFwd;
constexpr int kWWords = kWWin / 64;      // 20
constexpr int kWOwn = kWTile / 64;       // 16: word owners are lanes 0..15
constexpr int kHeadWords = kFwd / 64;    // 4
constexpr unsigned kTemplate = 0x285u;   // chips 0,2,7,9 high (framer.py:50)
constexpr int kNoise = 100;              // framer.py:31
constexpr long long kNoIndex = -(1ll << 62);
static_assert(kWOwn == 16 && kFwd % 64 == 0, "word owners are lanes 0..15 of each wavefront");
static_assert(kBack % 64 == 0 && kBack >= 100, "the noise window of a centre in the tile must lie in the LDS window");

// k_detect is latency bound per workgroup: 5 resident workgroups per CU (<= 96 VGPRs, no spills) measured
// 15-20 % faster than 4; 6 would need spills to scratch.
// Number of prefetch registers (of Span::ITER) reloaded inside the commit loop, right after their samples were
// converted; the others are reloaded after the commit.  More = less time without a fetch in flight, but the
// reloaded registers stay live through the rest of the commit (VGPR pressure: 8 costs the fifth wavefront per SIMD).
#ifndef ADSB_EARLY_ISSUE
#define ADSB_EARLY_ISSUE 4
#endif
constexpr int kEarlyIssue = ADSB_EARLY_ISSUE;
#ifndef ADSB_MIN_WAVES
#define ADSB_MIN_WAVES 5
#endif
// how k_detect turns a tile's rise mask into its ordered rise list: 1 = every lane walks its own 16 samples (shipped),
// 0 = word by word with one lane per bit (round 1)
#ifndef ADSB_RISE_BY_LANE
#define ADSB_RISE_BY_LANE 1
#endif

enum RecFlags : unsigned {
  kDemod = 1u,     // eob inside the demod input: bits valid (demod.py:82)
  kKept = 2u,      // passed the re-trigger gate (framer.py:121)
  kNoMatch = 4u,   // placeholder whose long pulse did not match the preamble
  kPending = 8u,   // placeholder waiting for k_longrun
  kHead = 16u,     // shard mode: one of the first centres of the shard, delivered whether gated or not
  kLongHint = 32u, // CANDIDATE words only (long-aware gate, opt-in): the burst's first data bit is set = 112-bit reply
  // Mode S parity pre-filter (decoder.py:550-556 DF, :560-688 check_parity), kDemod only:
  kParityOk = 32u,   // DF 11/17/18/19 (parity/interrogator field) and the 24-bit syndrome is zero
  kLongFmt = 64u,    // DF 16/17/18/19/20/21/24: 112-bit reply (else the decoder reads 56 bits)
  kKnownDf = 128u,   // DF is one the reference decoder checks parity for (0,4,5,11,16-21,24)
  kDfShift = 8u,     // bits 8..12: the downlink format (first five bits, MSB first)
  kRecLongHint = 0x2000u,   // records of a long-aware context: this burst holds the gate for 119*sps, not 63*sps
};

// x^j mod G, j = 0..111, G = x^24 + 0xFFF409 (decoder.py:268-269: the 25 coefficients spell 0x1FFF409).  The
// Mode S syndrome is linear in the message bits: bit i of an L-bit reply contributes x^(L-1-i) mod G, so
// compute_crc(bits[0:L-24]) ^ bits[L-24:L] (decoder.py:693-714 and its callers) is one XOR-reduction.
struct CrcTab { unsigned r[112]; };
constexpr CrcTab make_crc_tab() {
  CrcTab t{};
  unsigned v = 1;
  for (int j = 0; j < 112; ++j) {
    t.r[j] = v;
    v <<= 1;
    if (v & 0x1000000u) v ^= 0x1FFF409u;
  }
  return t;
}
// DF membership sets of decoder.py:565,604,636,669 as bit masks over the 5-bit DF
constexpr unsigned kDfShortSet = (1u << 0) | (1u << 4) | (1u << 5) | (1u << 11);
constexpr unsigned kDfLongSet = (1u << 16) | (1u << 17) | (1u << 18) | (1u << 19) | (1u << 20) | (1u << 21) | (1u << 24);
constexpr unsigned kDfPiSet = (1u << 11) | (1u << 17) | (1u << 18) | (1u << 19);

// 32-byte burst record: w0 = stream offset (int64); w1 = peak | median<<32 (float bits);
// w2 = bits 0..63 as bytes 0..7 (first bit = MSB of byte 0); w3 = bytes 8..13 | flags<<48.
struct alignas(16) Rec { unsigned long long w[4]; };
struct alignas(16) RecHalf { unsigned long long a, b; };

// A matched centre travels between kernels as one 64-bit word: flags<<56 | (local index + kBias).
constexpr long long kBias = 1ll << 40;
__device__ __forceinline__ unsigned long long cand_make(long long p, unsigned flags) {
  return ((unsigned long long)flags << 56) | (unsigned long long)(p + kBias);
}
__device__ __forceinline__ long long cand_p(unsigned long long c) {
  return (long long)(c & 0x00FFFFFFFFFFFFFFull) - kBias;
}
__device__ __forceinline__ unsigned cand_flags(unsigned long long c) { return (unsigned)(c >> 56); }

struct LongRise { long long rise; int blk; int slot; };

struct Summary {
  int n_rec;        // centres (incl. placeholders) in stream order
  int n_kept;       // survivors (after the gate, or all real centres when the gate is off)
  int overflow;     // some workgroup exceeded rec_cap
  int long_count;   // entries in the long-rise list
  unsigned flags;   // bit0 any rise, bit1 any fall, bit2 halo exceeded
  int pad_;
  long long lastp;  // largest paired pulse centre (local index) or kNoIndex
  long long last_kept_p;  // local index of the last survivor or kNoIndex
};

struct DetectArgs {
  const void* data;      // MODE 0: float2[n] complex64 IQ; 1: float[n] |IQ|^2; 2: short2[n] int16 IQ; 3: char2[n] int8 IQ; 4: uchar2[n] offset-binary IQ
  long long n;           // samples present; x(i) = 0 for i < 0 or i >= n
  long long in0_base;    // local index of the framer's in0[0] (-(8*sps-1) on a fresh stream)
  long long scan_lo;     // rises (and falls) are owned / counted in [scan_lo, scan_hi)
  long long scan_hi;
  long long fall_hi;     // a pulse needs its fall at an index < fall_hi
  long long dem_hi;      // PPM-slice iff p + 119*sps + sps/2 < dem_hi
  long long origin;      // stream offset of local index 0
  long long chunk;       // samples per unit (= wavefront), multiple of kWTile
  float thr;
  float prev_in0;        // value compared for the sample before in0[0] (framer.py:84)
  float scale;           // MODE 2/3/4: float32 multiplier applied to every integer component
  int sps;
  int end_is_call_end;   // 1: pulse still high at fall_hi is discarded (framer.py:102-108); 0: halo error
  int long_aware;        // opt-in extension: matched centres carry kLongHint for a length-aware gate
  int rec_cap;           // centres per unit
  int long_cap;
  unsigned long long* cands;  // [units][rec_cap]
  Rec* recs;             // [units][rec_cap]: the burst record of every matched centre, same slot as its list word
  int* blk_count;        // [units]
  long long* blk_lastp;  // [units]
  unsigned* blk_flags;   // [units]
  LongRise* longlist;
  int* long_count;
  unsigned long long* long_lastp;   // biased: centre + 2^62, 0 = none
};

__device__ __forceinline__ float mag2f(float re, float im) {
  // two rounded products, one rounded add: never contracted into an FMA
  return __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
}

// int16 IQ: component -> float32 exactly, times `scale` (one rounded multiply), then |.|^2 as above
__device__ __forceinline__ float mag2_iq16(unsigned iq, float scale) {
  const float re = __fmul_rn((float)(short)(iq & 0xFFFFu), scale);
  const float im = __fmul_rn((float)(short)(iq >> 16), scale);
  return mag2f(re, im);
}

// 8-bit IQ (I in the low byte): MODE 3 = two's complement (cs8: HackRF, SDRplay): component = f32(int8) * scale;
// MODE 4 = offset binary (cu8: RTL-SDR): component = f32(2*u8 - 255) * scale, i.e. (u8 - 127.5) * 2*scale with
// the subtraction done exactly in integers.  One rounded multiply per component, then |.|^2 as above.
template <int MODE>
__device__ __forceinline__ float mag2_iq8(unsigned iq, float scale) {
  int i8 = (int)(iq & 0xFFu), q8 = (int)((iq >> 8) & 0xFFu);
  if (MODE == 3) { i8 = (int)(signed char)i8; q8 = (int)(signed char)q8; }
  else { i8 = 2 * i8 - 255; q8 = 2 * q8 - 255; }
  return mag2f(__fmul_rn((float)i8, scale), __fmul_rn((float)q8, scale));
}

constexpr bool mode_is_iq8(int mode) { return mode == 3 || mode == 4; }
constexpr int mode_bytes(int mode) { return mode == 0 ? 8 : mode_is_iq8(mode) ? 2 : 4; }   // bytes per sample

// One sample as it lies in memory (RawSel<MODE>::type) and its conversion to |IQ|^2, kept apart so that a
// gather can be issued long before its value is needed.
template <int MODE> struct RawSel { using type = float; };
template <> struct RawSel<0> { using type = float2; };
template <> struct RawSel<2> { using type = unsigned; };
template <> struct RawSel<3> { using type = unsigned short; };
template <> struct RawSel<4> { using type = unsigned short; };

template <int MODE>
__device__ __forceinline__ typename RawSel<MODE>::type load_raw(const void* data, long long i) {
  return reinterpret_cast<const typename RawSel<MODE>::type*>(data)[i];
}
template <int MODE>
__device__ __forceinline__ float raw_mag2(typename RawSel<MODE>::type r, float scale) {
  if constexpr (MODE == 0) return mag2f(r.x, r.y);
  else if constexpr (MODE == 2) return mag2_iq16(r, scale);
  else if constexpr (mode_is_iq8(MODE)) return mag2_iq8<MODE>(r, scale);
  else return r;
}

template <int MODE>
__device__ __forceinline__ float load_sample(const void* data, long long i, float scale) {
  return raw_mag2<MODE>(load_raw<MODE>(data, i), scale);
}

template <int MODE>
__device__ __forceinline__ float xg(const void* data, long long n, long long i, float scale) {
  if (i < 0 || i >= n) return 0.0f;
  return load_sample<MODE>(data, i, scale);
}

// Branch-free variant for gathers: the load is always issued (from a clamped, always valid index) and the
// result is selected afterwards, so a wavefront can keep many of them in flight.  Needs n >= 1.
template <int MODE>
__device__ __forceinline__ float xg_nb(const void* data, long long n, long long i, float scale) {
  const bool ok = (i >= 0) & (i < n);
  const long long ci = ok ? i : 0;
  const float v = load_sample<MODE>(data, ci, scale);
  return ok ? v : 0.0f;
}

template <int MODE>
__device__ __forceinline__ bool above_at(const DetectArgs& a, long long i) {
  if (i >= 0) return xg<MODE>(a.data, a.n, i, a.scale) >= a.thr;
  if (i == a.in0_base - 1) return a.prev_in0 >= a.thr;
  return 0.0f >= a.thr;
}

__device__ __forceinline__ unsigned long long bit_range(long long lo, long long hi) {
  // bits [lo, hi) of a 64-bit word, arguments clamped
  if (hi <= 0 || lo >= 64 || hi <= lo) return 0ull;
  if (lo < 0) lo = 0;
  if (hi > 64) hi = 64;
  unsigned long long m = (hi == 64) ? ~0ull : ((1ull << hi) - 1ull);
  return m & ~((1ull << lo) - 1ull);
}

__device__ __forceinline__ int lanes_below(unsigned long long m, int lane) {
  return __popcll(m & ((1ull << lane) - 1ull));      // prefix popcount (v_mbcnt)
}

// ---- one wavefront builds one burst record ---------------------------------------------------------
// peak = x[p]; median of x[max(in0_base, p-100) : p] with np.median semantics (framer.py:156-159);
// 112 hard bits b1 > b0 at stride sps (demod.py:87-95).  All 64 lanes must be active.
// The median is an exact MSB-first radix select over order-preserving integer keys (two window
// samples per lane, counts by ballot + popcount): 32 wave-uniform steps instead of a sort.
__device__ __forceinline__ unsigned f32_key(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);        // monotone: key(a) < key(b) <=> a < b
}
__device__ __forceinline__ float key_f32(unsigned k) {
  return __builtin_bit_cast(float, k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

// Mode S parity pre-filter for one sliced burst held by a wavefront.  ma / mb = ballots of message bits 0..63 /
// 64..111 (wave-uniform).  The syndrome is linear in the message bits, so syndrome bit b is the parity of the
// message ANDed with a fixed 112-bit mask: lane b (b < 24) owns bit b of the 112-bit reading, lane 32+b bit b of
// the 56-bit reading, each with its mask in registers (ParityConsts, fetched once per thread; no cross-lane
// traffic, no per-burst memory access); one ballot collects both syndromes.  Returns kParityOk | kLongFmt | kKnownDf |
// DF << kDfShift (wave-uniform).
struct ParityConsts { unsigned long long m1, m2; };
struct ParityTab { unsigned long long m1[64], m2[64]; };
constexpr ParityTab make_parity_tab() {
  ParityTab t{};
  unsigned v = 1;
  for (int j = 0; j < 112; ++j) {                    // v = x^j mod G: the contribution of message bit L-1-j
    for (int b = 0; b < 24; ++b) {
      const unsigned long long bit = (v >> b) & 1u;
      const int i = 111 - j;                         // 112-bit reading: lane b
      if (i < 64) t.m1[b] |= bit << i; else t.m2[b] |= bit << (i - 64);
      if (j < 56) t.m1[32 + b] |= bit << (55 - j);   // 56-bit reading: lane 32+b
    }
    v <<= 1;
    if (v & 0x1000000u) v ^= 0x1FFF409u;
  }
  return t;
}
// one 16-byte read per thread per kernel launch (loop invariant for every burst the thread's wavefront handles)
__device__ __forceinline__ ParityConsts parity_consts(int lane) {
  static constexpr ParityTab tab = make_parity_tab();
  return ParityConsts{tab.m1[lane], tab.m2[lane]};
}
__device__ __forceinline__ unsigned parity_prefilter(unsigned long long ma, unsigned long long mb, const ParityConsts& pc) {
  const int par = (__popcll(ma & pc.m1) + __popcll(mb & pc.m2)) & 1;
  const unsigned long long syn = __ballot(par != 0);      // bits 0..23: 112-bit reading, bits 32..55: 56-bit reading
  const unsigned df = (unsigned)(__brevll(ma & 31ull) >> 59);                              // decoder.py:551
  const unsigned dfb = 1u << df;
  const bool lng = (dfb & kDfLongSet) != 0, known = lng || (dfb & kDfShortSet) != 0;
  unsigned flags = df << kDfShift;
  if (lng) flags |= kLongFmt;
  if (known) flags |= kKnownDf;
  if ((dfb & kDfPiSet) && (unsigned)(lng ? syn : (syn >> 32)) == 0) flags |= kParityOk;      // decoder.py:625,679
  return flags;
}

// The seven gathers of one burst.  Fast path (wave-uniform; every burst but the few within ~100 samples of the
// start or ~136*sps of the end of the buffer): the whole footprint [p-100, p+135.5*sps] lies inside the buffer,
// so the loads are a scalar base plus 32-bit lane offsets, nothing is clamped and nothing needs re-validating.
// Slow path: clamped (always valid) indices, burst_finish re-derives which were in range.  Needs n >= 1.
template <int MODE>
struct BurstFetch {
  using R = typename RawSel<MODE>::type;
  long long p;
  unsigned xflags;
  bool fast;
  R peak, w0, w1, x1, x0, y1, y0;
};
__device__ __forceinline__ long long clamp_idx(long long i, long long n) { return ((i >= 0) & (i < n)) ? i : 0; }
__device__ __forceinline__ long long uniform64(long long v) {
  const unsigned lo = (unsigned)adsb_uniform((int)(unsigned)(unsigned long long)v);
  const unsigned hi = (unsigned)adsb_uniform((int)(unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}

template <int MODE>
__device__ __forceinline__ BurstFetch<MODE> burst_issue(const DetectArgs& a, unsigned long long cand, int lane) {
  using R = typename RawSel<MODE>::type;
  BurstFetch<MODE> f;
  const void* d = a.data;
  const long long n = a.n, p = uniform64(cand_p(cand));      // the list word is the same in every lane
  const int sps = a.sps, half = sps >> 1;
  f.p = p;
  f.xflags = (cand_flags(cand) & kLongHint) ? kRecLongHint : 0u;      // the tail adds kKept / kHead (k_compact)
  const long long w100 = p - kNoise;
  f.fast = w100 >= 0 && w100 >= a.in0_base && p + 136ll * sps < n;
  if (f.fast) {
    const R* b = reinterpret_cast<const R*>(d) + w100;        // scalar base; every offset below is >= 0
    const unsigned l = (unsigned)lane, us = (unsigned)sps, o0 = (unsigned)kNoise + 8u * us + l * us;
    f.peak = b[kNoise];
    f.w0 = b[l];
    f.w1 = b[l + 64u];
    f.x1 = b[o0];                                             // demod.py:75,87
    f.x0 = b[o0 + (unsigned)half];                            // demod.py:91
    f.y1 = b[o0 + 64u * us];
    f.y0 = b[o0 + 64u * us + (unsigned)half];
    return f;
  }
  long long wlo = w100;
  if (wlo < a.in0_base) wlo = a.in0_base;
  const long long s0 = p + 8ll * sps + (long long)lane * sps;
  const long long s1 = s0 + 64ll * sps;
  f.peak = load_raw<MODE>(d, clamp_idx(p, n));
  f.w0 = load_raw<MODE>(d, clamp_idx(wlo + lane, n));
  f.w1 = load_raw<MODE>(d, clamp_idx(wlo + lane + 64, n));
  f.x1 = load_raw<MODE>(d, clamp_idx(s0, n));
  f.x0 = load_raw<MODE>(d, clamp_idx(s0 + half, n));
  f.y1 = load_raw<MODE>(d, clamp_idx(s1, n));
  f.y0 = load_raw<MODE>(d, clamp_idx(s1 + half, n));
  return f;
}

// The part of a burst record every path shares, once the wavefront holds the burst's samples: lane l has window
// samples l and l+64 (v0, v1; valid iff val0, val1; nwin = window length, framer.py:156), bit-pair l (x1, x0) and,
// for l < 48, bit-pair 64+l (y1, y0) (demod.py:87-92); peak = in0[pulse_idx].  `dem` = the burst ends inside the demod
// input (demod.py:82).  xflags: record flags already known (kRecLongHint; the tail adds kKept / kHead).  Lane 0
// stores the 32-byte record.
__device__ __forceinline__ void burst_reduce(long long offset, int nwin, bool val0, bool val1, float peak, float v0,
                                             float v1, bool dem, float x1, float x0, float y1, float y0, unsigned xflags,
                                             Rec* out, int lane, const ParityConsts& pc) {
  const unsigned long long nanm = __ballot((val0 && v0 != v0) || (val1 && v1 != v1));
  const unsigned k0 = val0 ? f32_key(v0) : 0xFFFFFFFFu;      // lanes outside the window sort last
  const unsigned k1 = val1 ? f32_key(v1) : 0xFFFFFFFFu;
  // exact MSB-first select of the lower middle (the middle for odd n): the answer's known high bits are `prefix`;
  // bit b is set iff at most kt keys lie below prefix | 1<<b.  One compare per key per step; prefix, c, kt are
  // wave-uniform and live in scalar registers.
  const int kt = adsb_uniform((nwin - 1) >> 1);
  unsigned prefix = 0;
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned T = prefix | (1u << bit);
    const int c = adsb_uniform(__popcll(__ballot(k0 < T)) + __popcll(__ballot(k1 < T)));
    if (c <= kt) prefix = T;
  }
  const unsigned A = prefix;                                 // key of the lower middle
  float med;
  if (nwin == 0) med = __builtin_bit_cast(float, 0xFFC00000u);       // np.median([]) == 0/0: default NaN, sign set
  else if (nanm) med = __builtin_bit_cast(float, 0x7FC00000u);        // a NaN in the window propagates
  else if (nwin & 1) med = key_f32(A);
  else {
    // upper middle: A again if it occurs often enough, else the smallest key above A
    const int cle = __popcll(__ballot(k0 <= A)) + __popcll(__ballot(k1 <= A));
    unsigned m = 0xFFFFFFFFu;
    if (k0 > A) m = k0;
    if (k1 > A && k1 < m) m = k1;
    for (int d2 = 32; d2 >= 1; d2 >>= 1) {
      const unsigned o = __shfl_xor(m, d2);
      if (o < m) m = o;
    }
    const unsigned B = (cle >= ((nwin - 1) >> 1) + 2) ? A : m;
    med = __fmul_rn(__fadd_rn(key_f32(A), key_f32(B)), 0.5f);          // f32(a+b)/2
  }
  const bool bitA = dem && x1 > x0, bitB = dem && lane < 48 && y1 > y0;                     // demod.py:95
  const unsigned long long ma = __ballot(bitA), mb = __ballot(bitB);
  const unsigned pflags = parity_prefilter(ma, mb, pc);
  if (lane == 0) {
    const unsigned long long ra = __builtin_bswap64(__brevll(ma));
    const unsigned long long rb = __builtin_bswap64(__brevll(mb)) & 0xFFFFFFFFFFFFull;
    const unsigned flags = (dem ? (kDemod | pflags) : 0u) | xflags;
    Rec r;
    r.w[0] = (unsigned long long)offset;
    r.w[1] = (unsigned long long)__builtin_bit_cast(unsigned, peak) |
             ((unsigned long long)__builtin_bit_cast(unsigned, med) << 32);
    r.w[2] = ra;
    r.w[3] = rb | ((unsigned long long)flags << 48);
    *out = r;
  }
}

// Record of a centre whose samples come from global memory (k_longrun: pulses longer than the LDS window).
template <int MODE>
__device__ void burst_finish(const DetectArgs& a, const BurstFetch<MODE>& f, Rec* out, int lane, const ParityConsts& pc) {
  const long long n = a.n, p = f.p;
  const int sps = a.sps, half = sps >> 1;
  const bool dem = p + 119ll * sps + half < a.dem_hi;   // demod.py:76,82 (sps even)
  int nwin;
  float peak, v0, v1, x1, x0, y1, y0;
  bool val0, val1;
  if (f.fast) {                                          // wave-uniform
    nwin = kNoise; val0 = true; val1 = lane < kNoise - 64;
    peak = raw_mag2<MODE>(f.peak, a.scale);
    v0 = raw_mag2<MODE>(f.w0, a.scale);
    v1 = val1 ? raw_mag2<MODE>(f.w1, a.scale) : 0.0f;
    x1 = raw_mag2<MODE>(f.x1, a.scale); x0 = raw_mag2<MODE>(f.x0, a.scale);
    y1 = raw_mag2<MODE>(f.y1, a.scale); y0 = raw_mag2<MODE>(f.y0, a.scale);
  } else {
    long long wlo = p - kNoise;
    if (wlo < a.in0_base) wlo = a.in0_base;
    nwin = (int)(p - wlo);
    val0 = lane < nwin; val1 = lane + 64 < nwin;
    const long long s0 = p + 8ll * sps + (long long)lane * sps;
    const long long s1 = s0 + 64ll * sps;
    auto val = [&](typename RawSel<MODE>::type r, long long i) -> float {     // x(i) = 0 outside the buffer
      const float v = raw_mag2<MODE>(r, a.scale);
      return ((i >= 0) & (i < n)) ? v : 0.0f;
    };
    peak = val(f.peak, p);
    v0 = val0 ? val(f.w0, wlo + lane) : 0.0f;
    v1 = val1 ? val(f.w1, wlo + lane + 64) : 0.0f;
    x1 = val(f.x1, s0); x0 = val(f.x0, s0 + half);
    y1 = val(f.y1, s1); y0 = val(f.y0, s1 + half);
  }
  burst_reduce(a.origin + p, nwin, val0, val1, peak, v0, v1, dem, x1, x0, y1, y0, f.xflags, out, lane, pc);
}

// Record of a centre found by k_detect, built by the whole wavefront while the centre's tile is still in its LDS
// window s_x[-kBack .. kWWin) (index j <-> sample t0 + j; samples outside the buffer are zeros there too): the noise
// window always lies inside it, the bit samples do for all of a 2 Msps burst that starts in the tile and for the
// beginning of longer ones -- samples past the window come from global memory (they are the next thing this
// wavefront streams anyway, so that fetch is served by the caches a moment later).  p = centre relative to t0.
// Inlined into k_detect's match loop (ADSB_INLINE_RECORDS=1, the shipped form: 75 VGPRs, no call).  It can also be
// built as a real call (=0: how it was introduced, when the streaming loop could not spare the registers); for that
// form every input is passed by value -- a by-reference DetectArgs would force the kernel arguments into scratch
// memory -- and the window arrives as a generic pointer.  A call costs the callee's entry wait for ALL outstanding
// memory operations, i.e. for the prefetch of the next tile, once per match: measured 1.54 vs 1.48-1.52 ms per pass.
struct WinArgs {
  const void* data; long long n, in0_base, dem_hi, origin; float scale; int sps;
};
// s_x_generic: the window as a generic pointer (what crosses a real call); it is cast back to the LDS address space at
// once, so the reads are ds_read (ordered by lgkmcnt) and not flat loads -- a flat load counts on vmcnt, and waiting for
// it would also wait for the prefetch of the next tile, which is older and still in flight.  For the same reason the
// parity masks arrive in registers (pc: loaded once per kernel) instead of being fetched here.
#ifndef ADSB_INLINE_RECORDS
#define ADSB_INLINE_RECORDS 1
#endif
#if ADSB_INLINE_RECORDS
#define ADSB_RECORD_FN __forceinline__
#else
#define ADSB_RECORD_FN __attribute__((noinline))
#endif
template <int MODE>
__device__ ADSB_RECORD_FN void burst_from_window(WinArgs a, const float* s_x_generic, long long t0, int p,
                                                 unsigned xflags, Rec* out, int lane, unsigned long long pc_m1,
                                                 unsigned long long pc_m2) {
  const ParityConsts pc{pc_m1, pc_m2};
  const ADSB_LDS float* s_x = (const ADSB_LDS float*)s_x_generic;
  const int sps = a.sps, half = sps >> 1;
  const long long P = t0 + p;
  long long wlo = P - kNoise;                                // framer.py:156: in0[max(0, pulse_idx-100) : pulse_idx]
  if (wlo < a.in0_base) wlo = a.in0_base;
  const int nwin = (int)(P - wlo);
  const int wl = (int)(wlo - t0);                            // >= p - 100 >= -kBack
  const bool val0 = lane < nwin, val1 = lane + 64 < nwin;
  const float v0 = val0 ? s_x[wl + lane] : 0.0f;
  const float v1 = val1 ? s_x[wl + lane + 64] : 0.0f;
  const float peak = s_x[p];
  const bool dem = P + 119ll * sps + half < a.dem_hi;        // demod.py:76,82 (sps even)
  float x1 = 0.0f, x0 = 0.0f, y1 = 0.0f, y0 = 0.0f;
  if (dem) {                                                 // wave-uniform
    const int j0 = p + 8 * sps + lane * sps;                 // demod.py:75,87
    if (p + 119 * sps + half < kWWin) {
      // wave-uniform fast path (every 2 Msps burst that starts in the tile): all 224 bit samples lie in the LDS window
      x1 = s_x[j0]; x0 = s_x[j0 + half];                     // demod.py:91
      if (lane < 48) { y1 = s_x[j0 + 64 * sps]; y0 = s_x[j0 + 64 * sps + half]; }
    } else {
      auto smp = [&](int j) -> float { return (j < kWWin) ? s_x[j] : xg<MODE>(a.data, a.n, t0 + j, a.scale); };
      x1 = smp(j0); x0 = smp(j0 + half);
      if (lane < 48) { y1 = smp(j0 + 64 * sps); y0 = smp(j0 + 64 * sps + half); }
    }
  }
  burst_reduce(a.origin + P, nwin, val0, val1, peak, v0, v1, dem, x1, x0, y1, y0, xflags, out, lane, pc);
}

// ---- global -> register -> LDS staging of one wavefront's share of a span ----------------------------
// A span of COUNT samples is split into kWaves contiguous shares; each wavefront fetches its share with
// 16-byte loads (1 KiB contiguous per wave instruction) into registers (span_issue) and later turns it
// into |IQ|^2 floats in LDS AND into the natural-order threshold bitmask words of those samples
// (span_commit) -- so the fetch of tile k+1 is in flight while tile k is processed, and the threshold
// masks cost no LDS re-read.  The fast path (whole span inside the buffer) has no per-load branches.
// One wave-wide 16-byte (8-byte for the 8-bit formats) load of the stream, marked non-temporal (`global_load ... nt`):
// every sample is fetched exactly once, so it should not displace anything in the caches on its way.  Measured on
// MI355X, 2^30 complex64 samples per pass, three passes in flight: 1.51-1.54 ms per pass with nt against 1.63-1.64
// with the default policy (k_detect alone 1.54-1.57 against 1.62-1.63).  The includer provides adsb_ld_stream<V>().
template <class Q>
__device__ __forceinline__ Q ld_stream(const char* p) {
  return adsb_ld_stream<Q>(p);
}
template <bool IQ8> struct SelectQ { using type = float4; };
template <> struct SelectQ<true> { using type = float2; };
template <int MODE, int COUNT, int NWAVES = kWaves>
struct Span {
  static constexpr int PER = (MODE == 0) ? 2 : 4;             // samples per lane load (complex64: 2 in 16 B; float / int16 IQ: 4 in 16 B; int8 IQ: 4 in 8 B)
  static constexpr int SHARE = COUNT / NWAVES;                // samples per wavefront (NWAVES share the span)
  static constexpr int GROUP = 64 * PER;                      // samples per wave-wide load
  static constexpr int ITER = (SHARE + GROUP - 1) / GROUP;    // (a partial last group only for the head span)
  static constexpr int LANES = (SHARE < GROUP) ? SHARE / PER : 64;   // active lanes when SHARE < GROUP
  using Q = typename SelectQ<mode_is_iq8(MODE)>::type;
  Q q[ITER];
};

// Returns false (wave- and block-uniform) when the span is not entirely inside the buffer: the caller
// then stages it with span_fill_ragged instead (at most one tile per call ends ragged).
template <int MODE, int COUNT, int NWAVES>
__device__ __forceinline__ bool span_issue(Span<MODE, COUNT, NWAVES>& sp, const DetectArgs& a, long long src, int wave, int lane) {
  using S = Span<MODE, COUNT, NWAVES>;
  if (src + COUNT > a.n) return false;
  // scalar base + 32-bit lane offset: one address VGPR for all loads of the span
  const long long wsrc = src + (long long)wave * S::SHARE;
  constexpr int BPS = mode_bytes(MODE);                       // bytes per sample
  const char* ub = reinterpret_cast<const char*>(a.data) + wsrc * BPS;
  using Q = typename S::Q;
  const unsigned lo = (unsigned)lane * (unsigned)sizeof(Q);
#pragma unroll
  for (int k = 0; k < S::ITER; ++k) {
    if (S::LANES == 64 || lane < S::LANES) sp.q[k] = ld_stream<Q>(ub + (k * 64 * (int)sizeof(Q) + lo));
    else sp.q[k] = Q{};
  }
  return true;
}

// Slow path for the ragged end of the buffer: scalar reads (zeros past the end) straight into LDS, then the
// mask words by ballot over LDS.  One wavefront stages its own span.
template <int MODE, int COUNT>
__device__ __forceinline__ void span_fill_ragged_w(float* sx, unsigned long long* smask, int dst,
                                                   const DetectArgs& a, long long src, int lane) {
  for (int i = lane; i < COUNT; i += 64) sx[dst + i] = xg<MODE>(a.data, a.n, src + i, a.scale);
  adsb_wave_sync();
  for (int w = 0; w < COUNT / 64; ++w) {
    const unsigned long long m = __ballot(sx[dst + 64 * w + lane] >= a.thr);
    if (lane == 0) smask[(dst >> 6) + w] = m;
  }
}

// dst = LDS sample index of the span's first sample (multiple of 64).  Every lane of the wavefront
// must call this (ballots); a wavefront whose share is one 64-sample word (the head span) uses only
// its low lanes for data and writes one mask word.
// REISSUE: `next` = address of this wavefront's share of the NEXT span (entirely inside the buffer); every
// register is loaded again the moment its samples have been converted, so the only time a wavefront has no
// fetch in flight is one |IQ|^2 computation per register -- not the whole commit (LDS writes, ballots, mask
// interleave), which now runs under the next tile's loads.
template <int MODE, int COUNT, int NWAVES, int REISSUE = 0>
__device__ __forceinline__ void span_commit(Span<MODE, COUNT, NWAVES>& sp, float* sx, unsigned long long* smask,
                                            int dst, float thr, float scale, int wave, int lane,
                                            const char* next = nullptr) {
  using S = Span<MODE, COUNT, NWAVES>;
  using Q = typename S::Q;
  const int wdst = dst + wave * S::SHARE;
  const bool act = (S::LANES == 64) || lane < S::LANES;
  const unsigned lo = (unsigned)lane * (unsigned)sizeof(Q);
#pragma unroll
  for (int k = 0; k < S::ITER; ++k) {
    const int g = wdst + k * S::GROUP;                        // first LDS sample of this wave-wide group
    if constexpr (MODE == 0) {
      float2 m;
      m.x = mag2f(sp.q[k].x, sp.q[k].y);
      m.y = mag2f(sp.q[k].z, sp.q[k].w);
      if (REISSUE > 0 && k < REISSUE) sp.q[k] = ld_stream<Q>(next + (k * 64 * (int)sizeof(Q) + lo));
      if (act) *reinterpret_cast<float2*>(&sx[g + 2 * lane]) = m;
      {
        // lane l holds samples 2l, 2l+1: the even/odd threshold masks (framer.py:83-84) are interleaved into
        // natural-order words on the SCALAR unit: s_bitreplicate doubles every bit, the masks pick the slot
        const unsigned long long E = __ballot(act && m.x >= thr), O = __ballot(act && m.y >= thr);
        const unsigned long long w0 = (adsb_bitrep32((unsigned)E) & 0x5555555555555555ull) |
                                      (adsb_bitrep32((unsigned)O) & 0xAAAAAAAAAAAAAAAAull);
        if (S::SHARE >= 128) {
          const unsigned long long w1 = (adsb_bitrep32((unsigned)(E >> 32)) & 0x5555555555555555ull) |
                                        (adsb_bitrep32((unsigned)(O >> 32)) & 0xAAAAAAAAAAAAAAAAull);
          if (lane == 0) { smask[g >> 6] = w0; smask[(g >> 6) + 1] = w1; }
        } else if (lane == 0) {
          smask[g >> 6] = w0;
        }
      }
    } else {
      float4 m;
      if constexpr (mode_is_iq8(MODE)) {                      // 8 int8 -> 4 |IQ|^2
        const unsigned u0 = __builtin_bit_cast(unsigned, sp.q[k].x), u1 = __builtin_bit_cast(unsigned, sp.q[k].y);
        m.x = mag2_iq8<MODE>(u0 & 0xFFFFu, scale);
        m.y = mag2_iq8<MODE>(u0 >> 16, scale);
        m.z = mag2_iq8<MODE>(u1 & 0xFFFFu, scale);
        m.w = mag2_iq8<MODE>(u1 >> 16, scale);
      } else if constexpr (MODE == 2) {                       // 8 int16 -> 4 |IQ|^2
        m.x = mag2_iq16(__builtin_bit_cast(unsigned, sp.q[k].x), scale);
        m.y = mag2_iq16(__builtin_bit_cast(unsigned, sp.q[k].y), scale);
        m.z = mag2_iq16(__builtin_bit_cast(unsigned, sp.q[k].z), scale);
        m.w = mag2_iq16(__builtin_bit_cast(unsigned, sp.q[k].w), scale);
      } else {
        m = sp.q[k];
      }
      if (REISSUE > 0 && k < REISSUE) sp.q[k] = ld_stream<Q>(next + (k * 64 * (int)sizeof(Q) + lo));
      if (act) *reinterpret_cast<float4*>(&sx[g + 4 * lane]) = m;
      {
        const unsigned long long A = __ballot(act && m.x >= thr), B = __ballot(act && m.y >= thr);
        const unsigned long long C = __ballot(act && m.z >= thr), D = __ballot(act && m.w >= thr);
        const int c = lane & 3;
        const unsigned long long sel = (c == 0) ? A : (c == 1) ? B : (c == 2) ? C : D;
        const int sh = lane >> 2;
        const unsigned long long w0 = __ballot((sel >> sh) & 1ull);
        if (lane == 0) smask[g >> 6] = w0;
        if (S::SHARE >= 256) {
          const unsigned long long w1 = __ballot((sel >> (16 + sh)) & 1ull);
          const unsigned long long w2 = __ballot((sel >> (32 + sh)) & 1ull);
          const unsigned long long w3 = __ballot((sel >> (48 + sh)) & 1ull);
          if (lane == 0) { smask[(g >> 6) + 1] = w1; smask[(g >> 6) + 2] = w2; smask[(g >> 6) + 3] = w3; }
        }
      }
    }
  }
  if constexpr (REISSUE > 0) {                               // the registers consumed last are reloaded last
#pragma unroll
    for (int k = REISSUE; k < S::ITER; ++k) sp.q[k] = ld_stream<Q>(next + (k * 64 * (int)sizeof(Q) + lo));
  }
}

// ---- k_detect: the streaming kernel, one independent stream segment per WAVEFRONT -------------------
// Each wavefront ("unit" = blockIdx*4 + wave) walks its own contiguous chunk of the stream tile by tile with
// its own sliding LDS window (kWTile + kFwd |IQ|^2 floats and their threshold mask words: the forward halo
// of one tile is the head of the next, so every sample is fetched from HBM exactly once), its own rise list
// and its own output list.  Nothing is shared between the wavefronts of a workgroup: there is no workgroup
// barrier, and a wavefront that meets a burst does not hold up three others (a first version with four
// wavefronts sharing a 4096-sample tile and three barriers per tile was 5 % slower).  a.chunk is the chunk of
// ONE unit (multiple of kWTile); matched centres are appended to the unit's slice of `cands` in stream order;
// ordering across units is by unit index (k_scan / k_gather).
template <int MODE>
__global__ void __launch_bounds__(kThreads, ADSB_MIN_WAVES) k_detect(DetectArgs a) {
  __shared__ __attribute__((aligned(16))) float s_xa[kWaves][kBack + kWWin];
  __shared__ unsigned long long s_maska[kWaves][kWWords];
  __shared__ unsigned s_risea[kWaves][kWTile / 2];

  const int lane = threadIdx.x & 63, wave = adsb_uniform((int)(threadIdx.x >> 6));
  float* s_x = s_xa[wave] + kBack;                           // s_x[j] <-> sample t0 + j, j in [-kBack, kWWin)
  unsigned long long* s_mask = s_maska[wave];
  unsigned* s_rise = s_risea[wave];
  const long long unit = (long long)blockIdx.x * kWaves + wave;
  const long long c0 = unit * a.chunk;
  long long c1 = c0 + a.chunk;
  if (c1 > a.scan_hi) c1 = a.scan_hi;
  const int half = a.sps >> 1;
  long long lastp_g = kNoIndex;
  int nrec = 0;                                              // wave-uniform running count of this unit's list
  unsigned uflags = 0u;
  int pred = adsb_uniform(above_at<MODE>(a, c0 - 1) ? 1 : 0);
  unsigned long long* my_cands = a.cands + unit * a.rec_cap;
  Rec* my_recs = a.recs + unit * a.rec_cap;

  // virtual rise in the zero history in front of a fresh stream: only possible when 0 >= thr
  if (unit == 0 && a.scan_lo < 0 && (0.0f >= a.thr) && !(a.prev_in0 >= a.thr)) {
    uflags |= 1u;
    if (lane == 0 && 0 < a.rec_cap) {
      my_cands[0] = cand_make(a.scan_lo, kPending | kNoMatch);
      const int li = atomicAdd(a.long_count, 1);
      if (li < a.long_cap) { LongRise e; e.rise = a.scan_lo; e.blk = 0; e.slot = 0; a.longlist[li] = e; }
    }
    nrec = 1;
  }

  Span<MODE, kWTile, 1> body;
  bool body_ok = true;
  if (c0 < c1) {
    Span<MODE, kFwd, 1> head;
    const bool head_ok = span_issue(head, a, c0, 0, lane);
    body_ok = span_issue(body, a, c0 + kFwd, 0, lane);
    if (head_ok) span_commit(head, s_x, s_mask, 0, a.thr, a.scale, 0, lane);
    else span_fill_ragged_w<MODE, kFwd>(s_x, s_mask, 0, a, c0, lane);
    // back halo of the first tile (later tiles inherit it from their predecessor): once per unit and launch
    for (int i = lane; i < kBack; i += 64) s_x[i - kBack] = xg<MODE>(a.data, a.n, c0 - kBack + i, a.scale);
  }

  const int lane_outer = lane;
  const ParityConsts pc = parity_consts(lane);               // four registers for the whole kernel, see burst_from_window
  for (long long t0 = c0; t0 < c1; t0 += kWTile) {
    // The lane number through an opaque copy, renewed every tile: otherwise a dozen lane-derived addresses and masks
    // (64-bit offsets of the loads, LDS addresses, 64*j + lane ...) are hoisted out of this loop as loop-invariant
    // registers, which the 96-VGPR budget of five resident wavefronts per SIMD cannot hold (they were spilled to
    // scratch memory, and every reload waited for the prefetch of the next tile with it).
    const int lane = adsb_opaque(lane_outer);
    // -- A: commit this tile's body (floats + mask words 4..19), start fetching the next one
    if (!body_ok) {
      span_fill_ragged_w<MODE, kWTile>(s_x, s_mask, kFwd, a, t0 + kFwd, lane);
      if (t0 + kWTile < c1) body_ok = span_issue(body, a, t0 + kWTile + kFwd, 0, lane);
    } else if (kEarlyIssue > 0 && t0 + kWTile < c1 && t0 + 2 * kWTile + kFwd <= a.n) {
      // the usual case: the next tile's body lies inside the buffer -> reload every register as soon as it is consumed
      const char* nb = reinterpret_cast<const char*>(a.data) + (t0 + kWTile + kFwd) * (long long)mode_bytes(MODE);
      span_commit<MODE, kWTile, 1, kEarlyIssue>(body, s_x, s_mask, kFwd, a.thr, a.scale, 0, lane, nb);
    } else {
      span_commit(body, s_x, s_mask, kFwd, a.thr, a.scale, 0, lane);
      if (t0 + kWTile < c1) body_ok = span_issue(body, a, t0 + kWTile + kFwd, 0, lane);
    }
    adsb_wave_sync();

    // -- B.1 rises / falls by mask algebra (framer.py:91-93)
#if ADSB_RISE_BY_LANE
    // lane l owns samples [16 l, 16 l + 16) of the tile: its 16 bits of the threshold mask and the bit in front of them
    const int w = lane >> 2, sh16 = (lane & 3) << 4;
    const unsigned long long Mw = s_mask[w];
    const unsigned pbit = sh16 ? ((unsigned)(Mw >> (sh16 - 1)) & 1u)
                               : (w ? (unsigned)(s_mask[w - 1] >> 63) : (unsigned)pred);
    const unsigned m16 = (unsigned)(Mw >> sh16) & 0xFFFFu;
    const long long sbase = t0 + 16ll * lane;
    const unsigned own16 = (unsigned)bit_range(a.scan_lo - sbase, a.scan_hi - sbase) & 0xFFFFu;
    const unsigned prev16 = (m16 << 1) | pbit;
    unsigned piece = m16 & ~prev16 & own16;                 // rises among my 16 samples
    const unsigned long long anyr = __ballot(piece != 0u), anyf = __ballot((~m16 & prev16 & own16) != 0u);
#else
    // lanes 0..15 own one word each
    const int word = (lane < kWWords) ? lane : 0;          // lanes 0..19 hold words 0..19 (16..19 = forward halo)
    const unsigned long long M = s_mask[word];
    const unsigned long long pb = (word > 0) ? (s_mask[word - 1] >> 63) : (unsigned long long)pred;
    const unsigned long long sh = (M << 1) | pb;
    const long long wbase = t0 + 64ll * word;
    const unsigned long long own = (lane < 16) ? bit_range(a.scan_lo - wbase, a.scan_hi - wbase) : 0ull;
    unsigned long long R = M & ~sh & own;
    const unsigned long long Fm = ~M & sh & own;
    const unsigned long long anyr = __ballot(R != 0ull), anyf = __ballot(Fm != 0ull);
#endif
    uflags |= (anyr ? 1u : 0u) | (anyf ? 2u : 0u);
    int nm = 0;
    if (anyr) {                                            // wave-uniform: quiet stretches skip everything below
      // Ordered rise list.  Each entry carries the rise index and, when the pulse ends within this or the next mask
      // word (always, for real Mode-S pulses), its fall index -- found here from the mask words, which saves B.2 a
      // dependent LDS round trip; 0xFFFF = not found yet (B.2 searches the LDS mask words).
#if ADSB_RISE_BY_LANE
      // Lane l extracts the rises among its samples [16 l, 16 l + 16) of the tile (`piece`, at most 8: a rise needs a
      // sub-threshold sample in front of it) in a short per-lane loop; its slots
      // follow from an exclusive prefix sum of the per-lane counts, built from four ballots (the counts have 4 bits).
      // The instruction count follows the densest 16 samples of the tile instead of the number of non-empty mask
      // words x a fixed cost: 2-3x fewer instructions on a tile that holds a 2 Msps burst, 6-8x fewer on 8 Msps dense
      // traffic, where every word has rises.
      int nr = 0;
      {
        const unsigned long long Mn = s_mask[w + 1];                            // w + 1 <= 16: a forward-halo word
        const int cnt = __builtin_popcount(piece);
        int slot = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const unsigned long long bm = __ballot(((cnt >> b) & 1) != 0);
          slot += lanes_below(bm, lane) << b;
          nr += __popcll(bm) << b;
        }
        while (piece) {
          const int pos = sh16 + __builtin_ctz(piece);
          piece &= piece - 1u;
          const unsigned long long inv0 = (pos == 63) ? 0ull : (~Mw & (~0ull << (pos + 1)));
          unsigned f = 0xFFFFu;
          if (inv0) f = (unsigned)(64 * w + __builtin_ctzll(inv0));
          else if (~Mn) f = (unsigned)(64 * (w + 1) + __builtin_ctzll(~Mn));
          s_rise[slot++] = (unsigned)(64 * w + pos) | (f << 16);
        }
      }
#else
      // word by word (wave-uniform loop over the 16 words, empty ones skipped): lane l takes bit l of the word and
      // its slot from the prefix popcount
      const int rlo = (int)(unsigned)R, rhi = (int)(unsigned)(R >> 32);
      const int mlo = (int)(unsigned)M, mhi = (int)(unsigned)(M >> 32);
      int nr = 0;
#pragma unroll
      for (int j = 0; j < kWOwn; ++j) {
        const unsigned long long Rj = (unsigned long long)(unsigned)adsb_readlane(rlo, j) |
                                      ((unsigned long long)(unsigned)adsb_readlane(rhi, j) << 32);
        if (Rj) {
          const unsigned long long Mj = (unsigned long long)(unsigned)adsb_readlane(mlo, j) |
                                        ((unsigned long long)(unsigned)adsb_readlane(mhi, j) << 32);
          const unsigned long long Mn = (unsigned long long)(unsigned)adsb_readlane(mlo, j + 1) |
                                        ((unsigned long long)(unsigned)adsb_readlane(mhi, j + 1) << 32);
          if ((Rj >> lane) & 1ull) {
            const unsigned long long inv0 = (lane == 63) ? 0ull : (~Mj & (~0ull << (lane + 1)));
            unsigned f = 0xFFFFu;
            if (inv0) f = (unsigned)(64 * j + __builtin_ctzll(inv0));
            else if (~Mn) f = (unsigned)(64 * (j + 1) + __builtin_ctzll(~Mn));
            s_rise[nr + lanes_below(Rj, lane)] = (unsigned)(64 * j + lane) | (f << 16);
          }
          nr += __popcll(Rj);
        }
      }
#endif
      adsb_wave_sync();

      // -- B.2 per rise: fall, centre, 16-chip test (framer.py:113,137-147)
      int lp = -1, lp2 = -1, hflag = 0;
      for (int i = lane; i < nr; i += 64) {
        const unsigned e0 = s_rise[i];
        const int r = (int)(e0 & 0xFFFFu);
        int f = (int)(e0 >> 16);
        if (f == 0xFFFF) {                                 // run longer than a word: search the LDS mask words
          int w = (r >> 6) + 2;
          unsigned long long inv = 0ull;
          while (w < kWWords && (inv = ~s_mask[w]) == 0ull) ++w;
          f = (w < kWWords) ? w * 64 + __builtin_ctzll(inv) : -1;
        }
        unsigned res = 0;
        if (f < 0) {
          if (t0 + kWWin < a.fall_hi) res = 0xFFFFu;       // pulse longer than the window: k_longrun
          else if (!a.end_is_call_end) hflag = 4;
        } else if (t0 + f < a.fall_hi) {
          const int p = (r + f) >> 1;                      // framer.py:113
          if (i == nr - 1) lp = p;                         // centres increase with i; only the last rise of
          else if (i == nr - 2) lp2 = p;                   // a tile can be left without a fall
          // 16-chip test as a chain of booleans: a per-lane boolean is a lane mask in scalar registers, so every chip
          // costs one vector compare and one scalar AND (assembling a 16-bit chip word per lane cost three vector
          // instructions per chip)
          bool match = true;
          if (p + 15 * half < kWWin) {                // all 16 taps inside the LDS window: one LDS round trip
            const float* tp = s_x + p;
            float tap[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) tap[k] = tp[k * half];
            const float hp = __fmul_rn(tap[0], 0.5f);      // tap 0 IS in0[pulse_idx]; /2 is exact
#pragma unroll
            for (int k = 0; k < 16; ++k) match = match & ((tap[k] > hp) == (((kTemplate >> k) & 1u) != 0u));   // framer.py:140-147
          } else {                                         // rare: taps past the window come from global memory
            const float hp = __fmul_rn(s_x[p], 0.5f);
            unsigned chips = 0;
#pragma unroll 1
            for (int k = 0; k < 16; ++k) {
              const int idx = p + k * half;
              const float v = (idx < kWWin) ? s_x[idx] : xg<MODE>(a.data, a.n, t0 + idx, a.scale);
              chips |= (v > hp ? 1u : 0u) << k;
            }
            match = chips == kTemplate;
          }
          if (match) {
            res = 0x8000u | (unsigned)p;
            if (a.long_aware) {                              // first data bit (demod.py:87-95, k = 0): DF >= 16 = long reply
              const int i1 = p + 16 * half, i0 = i1 + half;
              const float v1 = (i1 < kWWin) ? s_x[i1] : xg<MODE>(a.data, a.n, t0 + i1, a.scale);
              const float v0 = (i0 < kWWin) ? s_x[i0] : xg<MODE>(a.data, a.n, t0 + i0, a.scale);
              if (v1 > v0) res |= 0x10000u;
            }
          }
        } else if (!a.end_is_call_end) {
          hflag = 4;
        }
        s_rise[i] = res;
      }
      {
        const unsigned long long m1 = __ballot(lp >= 0), m2 = __ballot(lp2 >= 0), mh = __ballot(hflag != 0);
        const int v1 = __shfl(lp, m1 ? __builtin_ctzll(m1) : 0);
        const int v2 = __shfl(lp2, m2 ? __builtin_ctzll(m2) : 0);
        const int wl = m1 ? v1 : (m2 ? v2 : -1);
        if (wl >= 0) lastp_g = t0 + wl;
        if (mh) uflags |= 4u;
      }
      adsb_wave_sync();

      // -- B.3 ordered in-place compaction of the matched centres
      for (int base = 0; base < nr; base += 64) {
        const int i = base + lane;
        const unsigned e = (i < nr) ? s_rise[i] : 0u;
        const unsigned long long mb = __ballot(e != 0);
        if (e) s_rise[nm + lanes_below(mb, lane)] = e;
        nm += __popcll(mb);
      }
      adsb_wave_sync();

      // -- C: this tile's matches, in stream order: the list word and, built by the whole wavefront from the LDS
      //    window while it still holds the burst's samples, the burst record (wave-uniform loop: a tile rarely has
      //    more than one or two matches)
      for (int m = 0; m < nm; ++m) {
        const int slot = nrec + m;
        if (slot >= a.rec_cap) break;                        // overflow: reported through the count, call is re-run
        const unsigned e = (unsigned)adsb_uniform((int)s_rise[m]);
        if (e == 0xFFFFu) {
          if (lane == 0) {
            // the long pulse is the last rise of its tile: recover its index from the masks
            long long rg = kNoIndex;
            for (int w2 = kWOwn - 1; w2 >= 0 && rg == kNoIndex; --w2) {
              const unsigned long long M2 = s_mask[w2];
              const unsigned long long pb2 = (w2 > 0) ? (s_mask[w2 - 1] >> 63) : (unsigned long long)pred;
              const long long wb = t0 + 64ll * w2;
              const unsigned long long R2 = M2 & ~((M2 << 1) | pb2) & bit_range(a.scan_lo - wb, a.scan_hi - wb);
              if (R2) rg = wb + (63 - __builtin_clzll(R2));
            }
            my_cands[slot] = cand_make(rg, kPending | kNoMatch);
            const int li = atomicAdd(a.long_count, 1);
            if (li < a.long_cap) { LongRise le; le.rise = rg; le.blk = (int)unit; le.slot = slot; a.longlist[li] = le; }
          }
        } else {
          const int p = (int)(e & 0x7FFFu);
          const bool lh = (e & 0x10000u) != 0;
          if (lane == 0) my_cands[slot] = cand_make(t0 + (long long)p, lh ? kLongHint : 0u);
          burst_from_window<MODE>(WinArgs{a.data, a.n, a.in0_base, a.dem_hi, a.origin, a.scale, a.sps}, s_x, t0, p,
                                  lh ? kRecLongHint : 0u, my_recs + slot, lane, pc.m1, pc.m2);
        }
      }
      nrec += nm;
    }

    // what the next tile inherits: back + forward halo (floats; mask words of the forward halo) and the last threshold bit
    // (sources [kWTile-kBack, kWWin) and destinations [-kBack, kFwd) are disjoint: two short batches keep the
    // register footprint of this step at four values)
    const unsigned long long keepm = (lane < kHeadWords) ? s_mask[kWOwn + lane] : 0ull;
    const int keepp = adsb_uniform((int)(s_mask[kWOwn - 1] >> 63));
    {
      float keep[kFwd / 64];
#pragma unroll
      for (int j = 0; j < kFwd / 64; ++j) keep[j] = s_x[kWTile + lane + 64 * j];
      adsb_wave_sync();
#pragma unroll
      for (int j = 0; j < kFwd / 64; ++j) s_x[lane + 64 * j] = keep[j];
      if (lane < kHeadWords) s_mask[lane] = keepm;
    }
    {
      float keepb[kBack / 64];
#pragma unroll
      for (int j = 0; j < kBack / 64; ++j) keepb[j] = s_x[kWTile - kBack + lane + 64 * j];
#pragma unroll
      for (int j = 0; j < kBack / 64; ++j) s_x[lane + 64 * j - kBack] = keepb[j];
    }
    pred = keepp;
    // (the next iteration's commit writes s_x[kFwd..] and mask words >= 4; its wave_sync orders all of it)
  }
  if (lane == 0) {
    a.blk_count[unit] = nrec;
    a.blk_lastp[unit] = lastp_g;
    a.blk_flags[unit] = uflags;
  }
}

// ---- k_longrun: pulses whose run leaves the LDS window (or starts in the zero history) -------------
// One workgroup per entry scans forward cooperatively for the fall, then wave 0 finishes the pulse
// with global-memory taps and overwrites the placeholder.  Rare (CW / overload, or dense overlapping
// bursts): it is launched after every k_detect and returns at once when the list is empty.
template <int MODE>
__device__ __forceinline__ void longrun_body(int bid, int nb, const DetectArgs& a) {
  __shared__ unsigned long long s_found;   // fall index relative to rise+1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int n_entries = *a.long_count;                 // written by k_detect, usually 0: then this is a no-op
  if (n_entries > a.long_cap) n_entries = a.long_cap;
  for (int e = bid; e < n_entries; e += nb) {
    const LongRise le = a.longlist[e];
    const long long limit = a.fall_hi;
    const long long start = le.rise + 1;
    const unsigned long long none = ~0ull;
    if (tid == 0) s_found = none;
    __syncthreads();
    for (long long base = start; base < limit; base += 4ll * kThreads) {
      unsigned long long mine = none;
      for (int q = 3; q >= 0; --q) {
        const long long i = base + 4ll * tid + q;
        if (i < limit && !above_at<MODE>(a, i)) mine = (unsigned long long)(i - start);
      }
      if (mine != none) atomicMin(&s_found, mine);
      __syncthreads();
      const bool done = s_found != none;
      __syncthreads();
      if (done) break;
    }
    __syncthreads();
    const long long f = (s_found == none) ? limit : start + (long long)s_found;
    unsigned long long* out = a.cands + (long long)le.blk * a.rec_cap + le.slot;
    if (wave == 0) {
      if (f < limit) {
        const long long p = (le.rise + f) >> 1;            // floor, also for negative indices
        if (lane == 0) atomicMax(a.long_lastp, (unsigned long long)(p + (1ll << 62)));
        const float hp = __fmul_rn(xg<MODE>(a.data, a.n, p, a.scale), 0.5f);
        const float v = (lane < 16) ? xg<MODE>(a.data, a.n, p + (long long)lane * (a.sps >> 1), a.scale) : 0.0f;
        const unsigned long long cm = __ballot(lane < 16 && v > hp);
        unsigned hint = 0u;
        if (a.long_aware) {
          const long long i1 = p + 8ll * a.sps;
          if (xg<MODE>(a.data, a.n, i1, a.scale) > xg<MODE>(a.data, a.n, i1 + (a.sps >> 1), a.scale)) hint = kLongHint;
        }
        const bool match = (unsigned)cm == kTemplate;          // wave-uniform
        if (lane == 0) *out = match ? cand_make(p, hint) : cand_make(p, kNoMatch);
        if (match) {                                           // the centre's burst record, samples from global memory
          const BurstFetch<MODE> bf = burst_issue<MODE>(a, cand_make(p, hint), lane);
          burst_finish<MODE>(a, bf, a.recs + (long long)le.blk * a.rec_cap + le.slot, lane, parity_consts(lane));
        }
      } else if (lane == 0) {
        *out = cand_make(le.rise, kNoMatch);
        if (!a.end_is_call_end) atomicOr(a.blk_flags + le.blk, 4u);
      }
    }
    __syncthreads();
  }
}
template <int MODE>
__global__ void __launch_bounds__(kThreads) k_longrun(DetectArgs a) {
  longrun_body<MODE>((int)blockIdx.x, (int)gridDim.x, a);
}

// Exclusive prefix sum of one int per thread over a 256-thread workgroup (wave shuffles + 4 LDS words).
// Every thread must call it; *total receives the sum on every thread.
__device__ __forceinline__ int block_excl_scan(int v, int* total) {
  __shared__ int s_wsum[kWaves];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, (unsigned)d);
    if (lane >= d) incl += t;
  }
  if (lane == 63) s_wsum[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    const int c = s_wsum[w];
    if (w < wave) base += c;
    tot += c;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

// ---- k_scan: per-unit counts -> offsets, totals (single workgroup) ---------------------------------
// Global memory is touched only with COALESCED wave accesses (list b by thread b mod 256): the counts are staged in
// LDS round by round (kScanRound lists each), the per-thread contiguous ranges of the prefix sum are walked there,
// and the offsets go back through LDS the same way.  (A version whose threads read their 20 contiguous lists straight
// from global memory -- 64 different cache lines per wave instruction -- was measured to lengthen the CONCURRENT
// k_detect of the next pass by 0.03 ms; the same kernel with coalesced accesses costs it nothing.)  2 KB of LDS and
// few registers on purpose: this workgroup is meant to fit on a CU BESIDE five resident k_detect workgroups (which
// leave 3.8 KB of LDS -- in whole 1280-byte allocation granules -- and 112 VGPRs per SIMD free).
#ifndef ADSB_SCAN_ROUND
#define ADSB_SCAN_ROUND 512
#endif
constexpr int kScanRound = ADSB_SCAN_ROUND;
constexpr int kScanPer = kScanRound / kThreads;      // consecutive lists per thread and round
__device__ __forceinline__ void scan_body(const int* blk_count, const long long* blk_lastp,
                                          const unsigned* blk_flags, int nblk, int rec_cap,
                                          const int* long_count, const unsigned long long* long_lastp,
                                          int* blk_off, Summary* sum) {
  __shared__ long long s_lp[kWaves];
  __shared__ unsigned s_fl[kWaves];
  __shared__ int s_cnt[kScanRound];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  long long lp = kNoIndex; unsigned fl = 0;
  int carry = 0;                                               // lists before this round hold `carry` centres
  for (int base = 0; base < nblk; base += kScanRound) {
    for (int k = tid; k < kScanRound; k += kThreads) {         // coalesced: counts to LDS, max / OR reduced on the fly
      const int b = base + k;
      int c = 0;
      if (b < nblk) {
        c = blk_count[b];
        if (c > rec_cap) { fl |= 0x80000000u; c = rec_cap; }  // bit 31: some unit overflowed its list
        const long long l = blk_lastp[b];
        if (l > lp) lp = l;
        fl |= blk_flags[b];
      }
      s_cnt[k] = c;
    }
    __syncthreads();
    int acc = 0;
#pragma unroll
    for (int q = 0; q < kScanPer; ++q) acc += s_cnt[tid * kScanPer + q];
    int total = 0;
    int run = carry + block_excl_scan(acc, &total);            // (barriers inside)
#pragma unroll
    for (int q = 0; q < kScanPer; ++q) { const int c = s_cnt[tid * kScanPer + q]; s_cnt[tid * kScanPer + q] = run; run += c; }
    __syncthreads();
    for (int k = tid; k < kScanRound; k += kThreads)           // coalesced
      if (base + k < nblk) blk_off[base + k] = s_cnt[k];
    __syncthreads();
    carry += total;
  }
  // workgroup reductions: max of lastp, OR of flags (wave shuffles, then 4 words)
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const long long ol = __shfl_xor(lp, d);
    const unsigned of = __shfl_xor(fl, d);
    if (ol > lp) lp = ol;
    fl |= of;
  }
  if (lane == 0) { s_lp[wave] = lp; s_fl[wave] = fl; }
  __syncthreads();
  if (tid == 0) {
    long long L = (*long_lastp == 0ull) ? kNoIndex : (long long)(*long_lastp) - (1ll << 62);
    unsigned F = 0;
    for (int w = 0; w < kWaves; ++w) { if (s_lp[w] > L) L = s_lp[w]; F |= s_fl[w]; }
    sum->n_rec = carry; sum->overflow = (F >> 31) & 1u; sum->flags = F & 0x7FFFFFFFu; sum->lastp = L;
    sum->long_count = *long_count; sum->n_kept = 0; sum->last_kept_p = kNoIndex;
  }
}
__global__ void __launch_bounds__(kThreads) k_scan(const int* blk_count, const long long* blk_lastp,
                                                   const unsigned* blk_flags, int nblk, int rec_cap,
                                                   const int* long_count, const unsigned long long* long_lastp,
                                                   int* blk_off, Summary* sum) {
  scan_body(blk_count, blk_lastp, blk_flags, nblk, rec_cap, long_count, long_lastp, blk_off, sum);
}

// ---- k_gather: per-unit lists (centre words + burst records) -> one list each in stream order --------
// (the tail kernels are bodies with the workgroup's index and the grid size as parameters: k_tail_small runs them all
// in ONE workgroup for small passes)
__device__ __forceinline__ void gather_body(int bid, int nb, const unsigned long long* cands, const Rec* recs,
                                            const int* blk_count, const int* blk_off, int nblk, int rec_cap,
                                            unsigned long long* sorted, Rec* sorted_recs) {
  for (int b = bid; b < nblk; b += nb) {
    int c = blk_count[b];
    if (c > rec_cap) c = rec_cap;
    const int off = blk_off[b];
    const long long src = (long long)b * rec_cap;
    for (int j = threadIdx.x; j < c; j += kThreads) sorted[off + j] = cands[src + j];
    // records as 16-byte halves: consecutive threads touch consecutive 16 bytes
    const RecHalf* rs = reinterpret_cast<const RecHalf*>(recs + src);
    RecHalf* rd = reinterpret_cast<RecHalf*>(sorted_recs + off);
    for (int j = threadIdx.x; j < 2 * c; j += kThreads) rd[j] = rs[j];
  }
}
__global__ void __launch_bounds__(kThreads) k_gather(const unsigned long long* cands, const Rec* recs, const int* blk_count,
                                                     const int* blk_off, int nblk, int rec_cap,
                                                     unsigned long long* sorted, Rec* sorted_recs) {
  gather_body((int)blockIdx.x, (int)gridDim.x, cands, recs, blk_count, blk_off, nblk, rec_cap, sorted, sorted_recs);
}

// ---- k_resolve: the re-trigger gate (framer.py:121-123,165) as parallel chain walks -----------------
// Sequentially: accept a matched centre p iff p > eob, then eob = p + 63*sps.  A centre more than
// 63*sps past its predecessor is accepted whatever happened before it, so it starts an independent
// chain; each chain head walks its own (short) chain.  Words flagged kNoMatch are skipped.
__device__ __forceinline__ void resolve_body(int bid, int nb, unsigned long long* sorted, const Summary* sum, long long gate,
                                             long long gate_long, long long prev_eob) {
  // gate = 63*sps; gate_long = what a centre flagged kLongHint holds the gate for (119*sps with the long-aware gate,
  // else == gate: flags are never set then); prev_eob = carried eob as a local index (or very negative).
  // Chain heads are found with the LARGER window, which is always safe.
  const int n = sum->n_rec;
  const int nseg = (n + kThreads - 1) / kThreads;
  for (int seg = bid; seg < nseg; seg += nb) {
    const int i = seg * kThreads + threadIdx.x;
    if (i < n && !(cand_flags(sorted[i]) & kNoMatch)) {
      const long long p = cand_p(sorted[i]);
      int j = i - 1;
      while (j >= 0 && (cand_flags(sorted[j]) & kNoMatch)) --j;
      const bool head = (j < 0) || ((p - cand_p(sorted[j]) > gate_long) && (p > prev_eob));
      if (head) {
        long long eob = (j < 0) ? prev_eob : (p - 1);   // a head with a predecessor is always accepted
        int k = i;
        while (k < n) {
          const unsigned long long ck = sorted[k];
          if (!(cand_flags(ck) & kNoMatch)) {
            const long long pk = cand_p(ck);
            if (k != i) {
              // stop at the next head: it owns the rest
              int jj = k - 1;
              while (jj >= 0 && (cand_flags(sorted[jj]) & kNoMatch)) --jj;
              if (jj >= 0 && pk - cand_p(sorted[jj]) > gate_long && pk > prev_eob) break;
            }
            if (pk > eob) {
              sorted[k] = ck | ((unsigned long long)kKept << 56);
              eob = pk + ((cand_flags(ck) & kLongHint) ? gate_long : gate);
            }
          }
          ++k;
        }
      }
    }
  }
}
__global__ void __launch_bounds__(kThreads) k_resolve(unsigned long long* sorted, const Summary* sum, long long gate,
                                                      long long gate_long, long long prev_eob) {
  resolve_body((int)blockIdx.x, (int)gridDim.x, sorted, sum, gate, gate_long, prev_eob);
}

// ---- k_count / k_compact: survivors -> dense list ---------------------------------------------------
// A real centre survives when (flags & fmask) == fwant -- gate on: (kKept, kKept); gate off: (0, 0) -- or
// when it is one of the first head_n entries of the list (shard mode: the head of a shard is delivered
// whole so that the host can re-gate it against the previous shard's tail).
__device__ __forceinline__ bool survives(unsigned long long c, int i, unsigned fmask, unsigned fwant, int head_n) {
  const unsigned f = cand_flags(c);
  if (f & (kNoMatch | kPending)) return false;
  return (f & fmask) == fwant || i < head_n;
}

__device__ __forceinline__ void count_body(int bid, int nb, const unsigned long long* sorted, const Summary* sum,
                                           unsigned fmask, unsigned fwant, int head_n, int* seg_count) {
  __shared__ int s_c[kWaves];
  const int n = sum->n_rec;
  const int nseg = (n + kThreads - 1) / kThreads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int seg = bid; seg < nseg; seg += nb) {
    const int i = seg * kThreads + threadIdx.x;
    const bool kept = i < n && survives(sorted[i], i, fmask, fwant, head_n);
    const unsigned long long m = __ballot(kept);
    if (lane == 0) s_c[wave] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) seg_count[seg] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
    __syncthreads();
  }
}
__global__ void __launch_bounds__(kThreads) k_count(const unsigned long long* sorted, const Summary* sum,
                                                    unsigned fmask, unsigned fwant, int head_n, int* seg_count) {
  count_body((int)blockIdx.x, (int)gridDim.x, sorted, sum, fmask, fwant, head_n, seg_count);
}

// k_compact also does what a separate single-workgroup scan kernel used to: every workgroup sums the segment counts
// in front of its segment itself (a few hundred ints, L2 resident) -- one launch fewer on the tail of every pass.
// Emits the survivors' burst records (built by k_detect / k_longrun, ordered by k_gather) with kKept / kHead added.
__device__ __forceinline__ void compact_body(int bid, int nb, const unsigned long long* sorted, const Rec* sorted_recs,
                                             Summary* sum, const int* seg_count, unsigned fmask, unsigned fwant, int head_n,
                                             Rec* out, int out_cap, int* long_count, unsigned long long* long_lastp) {
  __shared__ int s_c[kWaves];
  __shared__ int s_pre[kWaves], s_tot[kWaves];
  const int n = sum->n_rec;
  const int nseg = (n + kThreads - 1) / kThreads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (bid == 0 && threadIdx.x == 0) {
    *long_count = 0;        // k_scan has consumed the long-pulse list: leave it empty for the slot's next pass
    *long_lastp = 0ull;
    if (nseg == 0) sum->n_kept = 0;
  }
  for (int seg = bid; seg < nseg; seg += nb) {
    // exclusive prefix of this segment and the grand total
    int pre = 0, tot = 0;
    for (int j = threadIdx.x; j < nseg; j += kThreads) {
      const int cj = seg_count[j];
      tot += cj;
      if (j < seg) pre += cj;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { pre += __shfl_xor(pre, d); tot += __shfl_xor(tot, d); }
    const int i = seg * kThreads + threadIdx.x;
    unsigned long long c = (i < n) ? sorted[i] : 0ull;
    const bool k = i < n && survives(c, i, fmask, fwant, head_n);
    if (i < head_n) c |= (unsigned long long)kHead << 56;
    const unsigned long long m = __ballot(k);
    if (lane == 0) { s_c[wave] = __popcll(m); s_pre[wave] = pre; s_tot[wave] = tot; }
    __syncthreads();
    int off = s_pre[0] + s_pre[1] + s_pre[2] + s_pre[3];
    const int total = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
    for (int w = 0; w < wave; ++w) off += s_c[w];
    off += lanes_below(m, lane);
    if (seg == 0 && threadIdx.x == 0) sum->n_kept = total;
    if (k && off < out_cap) {
      Rec r = sorted_recs[i];
      r.w[3] |= (unsigned long long)(cand_flags(c) & (kKept | kHead)) << 48;
      out[off] = r;
      if (off == total - 1) sum->last_kept_p = cand_p(c);
    }
    __syncthreads();
  }
}
__global__ void __launch_bounds__(kThreads) k_compact(const unsigned long long* sorted, const Rec* sorted_recs, Summary* sum,
                                                      const int* seg_count, unsigned fmask, unsigned fwant, int head_n,
                                                      Rec* out, int out_cap, int* long_count,
                                                      unsigned long long* long_lastp) {
  compact_body((int)blockIdx.x, (int)gridDim.x, sorted, sorted_recs, sum, seg_count, fmask, fwant, head_n, out, out_cap,
               long_count, long_lastp);
}

// ---- k_publish: the pass's 48-byte summary, final once k_compact is done, stored straight into the caller-visible
// (pinned, mapped) host copy -- no separate copy operation on the tail of the pass
__global__ void k_publish(const Summary* sum, Summary* host_sum) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *host_sum = *sum;
}

// ---- k_tail_small: the whole tail of a SMALL pass (a GNU Radio work() call: a few lists, a few hundred centres at most)
// in one workgroup and one launch -- long pulses, scan, gather, gate, count, compact, publish, with workgroup barriers
// where the multi-kernel chain has kernel boundaries.  Such a pass is bound by the number of GPU operations, not by their size:
// seven launches fewer.  (Writes of one phase are read by other wavefronts of the SAME workgroup in the next: the
// workgroup-scope ordering of __syncthreads() is what that needs.)
struct TailArgs {
  const unsigned long long* cands; const Rec* recs; const int* blk_count; const long long* blk_lastp; const unsigned* blk_flags;
  int* blk_off; int nblk, rec_cap; int* long_count; unsigned long long* long_lastp;
  unsigned long long* sorted; Rec* sorted_recs; int* seg_count; Summary* sum; Summary* host_sum; Rec* out; int out_cap;
  int gate_on, head_n; long long gate, gate_long, prev_eob;
};
template <int MODE>
__global__ void __launch_bounds__(kThreads) k_tail_small(DetectArgs a, TailArgs t) {
  longrun_body<MODE>(0, 1, a);                                // pulses longer than k_detect's LDS window (usually none)
  __syncthreads();
  scan_body(t.blk_count, t.blk_lastp, t.blk_flags, t.nblk, t.rec_cap, t.long_count, t.long_lastp, t.blk_off, t.sum);
  __syncthreads();
  gather_body(0, 1, t.cands, t.recs, t.blk_count, t.blk_off, t.nblk, t.rec_cap, t.sorted, t.sorted_recs);
  __syncthreads();
  unsigned fmask = 0u, fwant = 0u;
  if (t.gate_on) {
    resolve_body(0, 1, t.sorted, t.sum, t.gate, t.gate_long, t.prev_eob);
    fmask = kKept; fwant = kKept;
    __syncthreads();
  }
  count_body(0, 1, t.sorted, t.sum, fmask, fwant, t.head_n, t.seg_count);
  __syncthreads();
  compact_body(0, 1, t.sorted, t.sorted_recs, t.sum, t.seg_count, fmask, fwant, t.head_n, t.out, t.out_cap, t.long_count,
               t.long_lastp);
  __syncthreads();
  if (threadIdx.x == 0) *t.host_sum = *t.sum;
}

// ---- k_slice: PPM slice (+ optional confidence ratio) for a caller-supplied tag list ---------------
// The stand-alone demod block (demod.py:57-136): one wavefront per tag.  in0 is the demod's input
// chunk (|IQ|^2 floats, MODE 1), tag_idx[] are sob-8*sps positions local to in0.
template <int MODE>
__global__ void __launch_bounds__(kThreads) k_slice(const void* data, long long n, const long long* tag_idx,
                                                    int ntags, int sps, unsigned char* bits14,
                                                    unsigned char* ok, float* ratio) {
  const int lane = threadIdx.x & 63;
  const int wave_g = (int)((blockIdx.x * (unsigned)kThreads + threadIdx.x) >> 6);
  const int nwave = (int)((gridDim.x * (unsigned)kThreads) >> 6);
  const int half = sps >> 1;
  const ParityConsts pc = parity_consts(lane);
  for (int t = wave_g; t < ntags; t += nwave) {
    const long long p = tag_idx[t];
    // demod.work() only ever sees tags inside its own chunk (demod.py:67): a tag in front of it is dropped, not sliced
    const bool dem = p >= 0 && p + 119ll * sps + half < n;
    bool b0 = false, b1 = false;
    float q0 = 0.0f, q1 = 0.0f;
    if (dem) {
      const long long s0 = p + 8ll * sps + (long long)lane * sps;
      const float x1 = xg<MODE>(data, n, s0, 1.0f), x0 = xg<MODE>(data, n, s0 + half, 1.0f);
      b0 = x1 > x0; q0 = __fdiv_rn(x1, x0);
      if (lane < 48) {
        const long long s1 = s0 + 64ll * sps;
        const float y1 = xg<MODE>(data, n, s1, 1.0f), y0 = xg<MODE>(data, n, s1 + half, 1.0f);
        b1 = y1 > y0; q1 = __fdiv_rn(y1, y0);
      }
    }
    const unsigned long long ma = __ballot(b0), mb = __ballot(b1);
    const unsigned pflags = parity_prefilter(ma, mb, pc);
    if (lane == 0) {
      const unsigned long long ra = __builtin_bswap64(__brevll(ma));
      const unsigned long long rb = __builtin_bswap64(__brevll(mb));
      for (int k = 0; k < 8; ++k) bits14[(long long)t * 14 + k] = (unsigned char)(ra >> (8 * k));
      for (int k = 0; k < 6; ++k) bits14[(long long)t * 14 + 8 + k] = (unsigned char)(rb >> (8 * k));
      ok[t] = dem ? (unsigned char)(1u | (pflags & 0xE0u)) : 0;     // kDemod | kParityOk | kLongFmt | kKnownDf
    }
    if (ratio && dem) {
      ratio[(long long)t * 112 + lane] = q0;
      if (lane < 48) ratio[(long long)t * 112 + 64 + lane] = q1;
    }
  }
}

// ---- k_confidence: opt-in confidence output of the fused path (ADSB_FLAG_CONFIDENCE) -----------------
// demod.py:97-101 keeps bit_confidence = 10*log10(bit1_amp / bit0_amp) for the burst it just sliced; this kernel
// returns the float32 RATIO of every delivered record that has a PDU (the host applies 10*log10 with NumPy, like the
// SNR): one wavefront per record, samples from global memory, row t of ratio[n_kept][112] belongs to record t
// (rows of records without ADSB_BURST_DEMOD are left untouched).
template <int MODE>
__global__ void __launch_bounds__(kThreads) k_confidence(DetectArgs a, const Rec* out, const Summary* sum, int out_cap,
                                                         float* ratio) {
  const int lane = threadIdx.x & 63;
  const int wave_g = (int)((blockIdx.x * (unsigned)kThreads + threadIdx.x) >> 6);
  const int nwave = (int)((gridDim.x * (unsigned)kThreads) >> 6);
  int n = sum->n_kept;
  if (n > out_cap) n = out_cap;
  const int sps = a.sps, half = sps >> 1;
  for (int t = wave_g; t < n; t += nwave) {
    const Rec r = out[t];
    if (!((unsigned)(r.w[3] >> 48) & kDemod)) continue;        // wave-uniform
    const long long p = (long long)r.w[0] - a.origin;
    const long long s0 = p + 8ll * sps + (long long)lane * sps;
    const float x1 = xg<MODE>(a.data, a.n, s0, a.scale), x0 = xg<MODE>(a.data, a.n, s0 + half, a.scale);
    ratio[(long long)t * 112 + lane] = __fdiv_rn(x1, x0);
    if (lane < 48) {
      const long long s1 = s0 + 64ll * sps;
      const float y1 = xg<MODE>(a.data, a.n, s1, a.scale), y0 = xg<MODE>(a.data, a.n, s1 + half, a.scale);
      ratio[(long long)t * 112 + 64 + lane] = __fdiv_rn(y1, y0);
    }
  }
}

}  // namespace adsb
