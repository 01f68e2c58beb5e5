// adsb_device.h -- CDNA4 (gfx950) device code for the ADS-B front end.
//
//   k_detect   streams the IQ once: |IQ|^2 -> threshold bitmask (one __ballot per 64 samples) -> rises by
//              mask algebra -> pulse centre -> 16-chip preamble test -> for every match, while its samples are
//              still in the wavefront's LDS window, the complete burst record (peak, median-of-100 noise,
//              112-bit PPM slice, Mode S parity pre-filter): the stream is never read a second time
//   k_order    finishes the (rare) pulses longer than the LDS window, turns the per-unit counts into offsets and puts the
//              centre words into stream order -- one launch
//   k_resolve / k_count / k_compact
//              apply the re-trigger gate as parallel chain walks, compact the survivors' records; the last workgroup of
//              k_compact hands the summary to the host
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
// `adsb_ld_stream<Q>(const char*)` (one Q-sized load of streamed, single-use data), `adsb_sdot4(int a, int b, int c)` (c + the sum of the
// four products of the signed bytes of a and b: v_dot4c_i32_i8), `adsb_cold(const DetectArgs&)` (a pointer
// to the same argument block as it lies in the kernel's argument memory, opaque to the optimiser: rarely needed fields are
// loaded where they are used instead of occupying scalar registers across the tile loop), and the macro ADSB_LDS (the
// address-space qualifier of workgroup-local memory, empty for the emulator): the
// product translation unit adsb_hip.hip maps them to __builtin_amdgcn_wave_barrier() / _readfirstlane() /
// _readlane(); tests/sim/sim_driver.cpp includes the test-only SIMT emulator instead, so the very same
// kernels run on a machine without a GPU.
#pragma once


namespace adsb {

constexpr int kThreads = 256;            // 4 wavefronts per workgroup: the tail kernels and the one-launch small pass
constexpr int kWaves = kThreads / 64;
// k_detect's wavefronts share nothing and meet at no barrier; a workgroup of four only ties four of them together -- the
// slot of a wavefront that has finished its chunk stays empty until the slowest of its three neighbours has finished too
// (their burst counts differ), and LDS is handed out four windows at a time.  For the 8-bit formats, which are bound by the
// instructions their wavefronts issue, k_detect therefore runs ONE wavefront per workgroup (round 5): a slot is refilled
// the moment its wavefront ends and 21 instead of 20 wavefronts fit on a CU: +6.6 % int8, +6.9 % uint8.  The HBM-bound
// formats lose 2-5 % that way (complex64, |IQ|^2 floats; int16 +-0) and keep four (profiles/r05_ab_one_wave_workgroups.txt).
constexpr int kWTile = 1024;             // samples a wavefront owns per tile iteration (16 ballot words)
constexpr int kFwd = 256;                // forward halo kept in LDS behind every tile
constexpr int kBack = 128;               // back halo kept in LDS in front of every tile: the <=100-sample noise window (framer.py:156)
constexpr int kWWin = kWTile + kFwd;
constexpr int kMaxCentre = (kWTile - 1 + kWWin - 1) / 2;   // largest centre relative to its tile: rise in the tile, fall inside the window
template <bool B> struct BoolC { static constexpr bool value = B; };
constexpr int kWWords = kWWin / 64;      // 20
constexpr int kWOwn = kWTile / 64;       // 16: word owners are lanes 0..15
constexpr int kHeadWords = kFwd / 64;    // 4
constexpr unsigned kTemplate = 0x285u;   // chips 0,2,7,9 high (framer.py:50)
constexpr int kNoise = 100;              // framer.py:31
constexpr long long kNoIndex = -(1ll << 62);
static_assert(kWOwn == 16 && kFwd % 64 == 0, "word owners are lanes 0..15 of each wavefront");
static_assert(kBack % 64 == 0 && kBack >= 100, "the noise window of a centre in the tile must lie in the LDS window");

// k_detect: five resident workgroups per CU (<= 96 VGPRs, 31 KB of LDS each); the tail kernels of the previous pass
// are sized to fit beside them (tests/test_abi.py holds the limits).
constexpr int kMinWaves = 5;

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
// MODE 5 = MODE 3 whose scale is a power of two (the usual int8 convention: x / 128): every intermediate of the
// float chain is then exact -- f32(i) * 2^-k, its square, the sum of two squares (< 2^15 * 2^-2k) -- so |IQ|^2 equals
// f32(i*i + q*q) * scale^2 bit for bit, and the tile loop takes the integer sum of squares from ONE v_dot4_i32_i8 per
// sample (body_convert); everything outside the tile loop converts as MODE 3 does.
constexpr int kModeSc8Pow2 = 5;
// MODE 6 = MODE 4 whose scale is a power of two (the RTL-SDR convention (u8 - 127.5) / 128 = (2*u8 - 255) * 2^-8): again every
// intermediate of the float chain is exact -- odd integers |c| <= 255 times 2^-k, their squares, the sum of two (< 2^18 *
// 2^-2k) -- and with x = u8 - 128 (the byte with its top bit flipped, read as a signed byte) c = 2x + 1, so
// c_i^2 + c_q^2 = 4 (x_i^2 + x_q^2 + x_i + x_q) + 2: two v_dot4c_i32_i8 per sample (x.x and x.1) in the tile loop (body_convert).
constexpr int kModeCu8Pow2 = 6;
template <int MODE>
__device__ __forceinline__ float mag2_iq8(unsigned iq, float scale) {
  if constexpr (MODE == 3 || MODE == kModeSc8Pow2) {
    const int i8 = (int)(signed char)(iq & 0xFFu), q8 = (int)(signed char)((iq >> 8) & 0xFFu);
    return mag2f(__fmul_rn((float)i8, scale), __fmul_rn((float)q8, scale));
  } else {
    // f32(2*u8 - 255) as 2*f32(u8) - 255: every step is exact (integers below 2^24), and it is one byte-to-float
    // conversion (v_cvt_f32_ubyteN, no extraction) and half a packed multiply-add per component instead of three
    // integer instructions and a conversion
    const float fi = (float)(iq & 0xFFu), fq = (float)((iq >> 8) & 0xFFu);
    const float ci = __builtin_fmaf(fi, 2.0f, -255.0f), cq = __builtin_fmaf(fq, 2.0f, -255.0f);
    return mag2f(__fmul_rn(ci, scale), __fmul_rn(cq, scale));
  }
}

constexpr bool mode_is_iq8(int mode) { return mode == 3 || mode == 4 || mode == 5 || mode == 6; }
constexpr int det_waves(int mode) { return mode_is_iq8(mode) ? 1 : kWaves; }      // wavefronts per k_detect workgroup
constexpr int mode_bytes(int mode) { return mode == 0 ? 8 : mode_is_iq8(mode) ? 2 : 4; }   // bytes per sample

// One sample as it lies in memory (RawSel<MODE>::type) and its conversion to |IQ|^2, kept apart so that a
// gather can be issued long before its value is needed.
template <int MODE> struct RawSel { using type = float; };
template <> struct RawSel<0> { using type = float2; };
template <> struct RawSel<2> { using type = unsigned; };
template <> struct RawSel<3> { using type = unsigned short; };
template <> struct RawSel<4> { using type = unsigned short; };
template <> struct RawSel<5> { using type = unsigned short; };
template <> struct RawSel<6> { using type = unsigned short; };

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

template <int MODE, class A>
__device__ __forceinline__ bool above_at(const A& a, long long i) {
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

// np.median of the noise window held by a wavefront (framer.py:156-159): lane l has window samples l and l + 64 (v0, v1;
// valid iff val0, val1; nwin = window length).  Wave-uniform result.  `hint` (wave-uniform, in and out): the key this
// function found for the wavefront's previous burst, 0 = none -- a first guess that is proved before it is used.
template <bool QUANT = false>
__device__ __forceinline__ float noise_median(int nwin, bool val0, bool val1, float v0, float v1, unsigned& hint) {
  const unsigned long long nanm = __ballot((val0 && v0 != v0) || (val1 && v1 != v1));
  const unsigned k0 = val0 ? f32_key(v0) : 0xFFFFFFFFu;      // lanes outside the window sort last
  const unsigned k1 = val1 ? f32_key(v1) : 0xFFFFFFFFu;
  // Exact select of the lower middle (the middle for odd n), rank kt: binary search for the largest T with at most kt
  // keys below it, one bit per step (one compare per key per step; lo, the count and kt are wave-uniform scalars).
  // The interval [lo, lo + 2^bit) always holds the key of rank kt.
  const int kt = adsb_uniform((nwin - 1) >> 1);
  unsigned lo = 0u, span = 0u;
  bool single = false;
  // Steps of one bit (from the full range: 32; from a proved guess: 23).  After 16, 20 and 24 decided bits: what does the
  // interval [lo, lo + 2^bit) hold?  Exactly
  // one key: that key IS the answer -- the usual case for float noise after 16 bits (half the steps).  Four or more keys
  // that are all EQUAL (minimum == maximum of the keys inside): that value is the answer -- the usual case for streams
  // quantised to a few levels (8-bit IQ), whose median sits in a crowd of duplicates.  Otherwise on with the search.
  unsigned A = 0u;
  bool have = false;
  auto below = [&](unsigned T) -> int { return adsb_uniform(__popcll(__ballot(k0 < T)) + __popcll(__ballot(k1 < T))); };
  // The key of the previous burst of this wavefront (`hint`) as a first guess: the noise floor is the same a few hundred
  // samples on, so the new key almost always lies within 2^23 key units (half to all of its own value) of the old
  // one.  Two counts PROVE on which side and within that distance it lies (nothing is assumed: a guess that fails
  // costs its two counts and the search starts from the full range), and replace the first nine steps.
  const unsigned h = (unsigned)adsb_uniform((int)hint);
  bool full = true;
  // QUANT (streams quantised to a few levels: the 8-bit formats): the noise floor of consecutive bursts has, more often than
  // not, the very SAME median value.  Two counts prove it -- at most kt keys below h, more than kt keys not above it: h IS the
  // key of rank kt -- and the second one also answers the even window's question (does the value occur again above rank
  // kt?).  A hint that fails costs those two counts; the search below then starts as it always did.
  bool exact = false, exact_again = false;
  if (QUANT && h >= 0x00800000u && h < 0xFF000000u) {
    const int c_lt = below(h), c_le = below(h + 1u);
    if (c_lt <= kt && c_le > kt) { exact = true; exact_again = c_le >= kt + 2; }
  }
  if (exact) {
    A = h; have = true; full = false;
  } else
  if (h >= 0x00800000u && h < 0xFF000000u) {                 // wave-uniform; h +- 2^23 cannot wrap
    const int c1 = below(h);                                   // <= kt: the key is >= h
    const unsigned T2 = c1 <= kt ? h + 0x00800000u : h - 0x00800000u;
    const int c2 = below(T2);                                  // <= kt: the key is >= T2
    if ((c1 <= kt) != (c2 <= kt)) {                            // the key lies between h and T2: in [lo, lo + 2^23)
      lo = c1 <= kt ? h : T2;
      full = false;
    }
  }
  if (full) {                                                // no guess, or not near it: bits 31 .. 23 from the full range
#pragma unroll
    for (int b = 31; b >= 23; --b) {
      const unsigned T = lo | (1u << b);
      if (below(T) <= kt) lo = T;
    }
  }
  if (!exact) {
#pragma unroll
  for (int b = 22; b >= 20; --b) {                           // the key lies in [lo, lo + 2^(b+1)); lo need not be aligned
    const unsigned T = lo + (1u << b);
    if (below(T) <= kt) lo = T;
  }
  for (int bit = 19; bit >= 0; bit -= 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned T = lo + (1u << (bit - j));
      if (below(T) <= kt) lo = T;
    }
    if (bit <= 19 && bit >= 11) {
      const unsigned sp = 1u << (bit - 3);
      const bool in0 = k0 - lo < sp, in1 = k1 - lo < sp;
      const int inside = adsb_uniform(__popcll(__ballot(in0)) + __popcll(__ballot(in1)));
      if (inside == 1) { single = true; span = sp; break; }
      if (inside >= 4) {                                      // wave-uniform
        unsigned mn = in0 ? k0 : 0xFFFFFFFFu, mx = in0 ? k0 : 0u;
        if (in1) { mn = k1 < mn ? k1 : mn; mx = k1 > mx ? k1 : mx; }
        mn = adsb_wave_min_u32(mn);
        mx = adsb_wave_max_u32(mx);
        if (mn == mx) { A = mn; have = true; break; }
      }
    }
  }
  }
  if (single) {                                              // wave-uniform: fetch the one key inside [lo, lo + span)
    const unsigned long long m0 = __ballot(k0 - lo < span), m1 = __ballot(k1 - lo < span);
    const unsigned a0 = (unsigned)adsb_readlane((int)k0, m0 ? __builtin_ctzll(m0) : 0);
    const unsigned a1 = (unsigned)adsb_readlane((int)k1, m1 ? __builtin_ctzll(m1) : 0);
    A = m0 ? a0 : a1;
    have = true;
  }
  if (!have) A = lo;                                         // key of the lower middle
  hint = A;
  float med;
  if (nwin == 0) med = __builtin_bit_cast(float, 0xFFC00000u);       // np.median([]) == 0/0: default NaN, sign set
  else if (nanm) med = __builtin_bit_cast(float, 0x7FC00000u);        // a NaN in the window propagates
  // (a median of zero is +0.0 whatever the signs of the zeros in the window: np.median is np.mean of the middle element(s),
  // whose sum starts from +0.0 -- 0.0 + -0.0 = +0.0 -- so the result never depends on how -0.0 and +0.0 were ordered; the
  // key order here puts -0.0 below +0.0, the rounded add of +0.0 below makes that invisible.  x + 0.0 == x for every other x)
  else if (nwin & 1) med = __fadd_rn(key_f32(A), 0.0f);
  else {
    // upper middle: A again if it occurs often enough (impossible when A was alone in its interval), else the
    // smallest key above A
    bool again = false;
    if (exact) again = exact_again;
    else if (!single) again = __popcll(__ballot(k0 <= A)) + __popcll(__ballot(k1 <= A)) >= kt + 2;
    unsigned B = A;
    if (!again) {                                            // wave-uniform
      unsigned m = (k0 > A) ? k0 : 0xFFFFFFFFu;
      if (k1 > A && k1 < m) m = k1;
      B = adsb_wave_min_u32(m);
    }
    med = __fadd_rn(__fmul_rn(__fadd_rn(key_f32(A), key_f32(B)), 0.5f), 0.0f);          // f32(a+b)/2
  }
  return med;
}

// the two halves of a 32-byte burst record, stored by lane 0: offset + (peak, median), and 112 bits + flags
__device__ __forceinline__ void rec_store_head(Rec* out, long long offset, float peak, float med, int lane) {
  if (lane == 0) {
    out->w[0] = (unsigned long long)offset;
    out->w[1] = (unsigned long long)__builtin_bit_cast(unsigned, peak) |
                ((unsigned long long)__builtin_bit_cast(unsigned, med) << 32);
  }
}
// ma / mb = ballots of message bits 0..63 / 64..111; flags = kDemod | kRecLongHint (the tail adds kKept / kHead and the
// parity pre-filter bits)
__device__ __forceinline__ void rec_store_bits(Rec* out, unsigned long long ma, unsigned long long mb, unsigned flags, int lane) {
  if (lane == 0) {
    out->w[2] = __builtin_bswap64(__brevll(ma));
    out->w[3] = (__builtin_bswap64(__brevll(mb)) & 0xFFFFFFFFFFFFull) | ((unsigned long long)flags << 48);
  }
}

// The record of a burst whose samples a wavefront holds in registers (k_longrun): lane l has window samples l and l+64
// (v0, v1; valid iff val0, val1; nwin = window length, framer.py:156), bit-pair l (x1, x0) and, for l < 48, bit-pair 64+l
// (y1, y0) (demod.py:87-92); peak = in0[pulse_idx].  `dem` = the burst ends inside the demod input (demod.py:82).
// xflags: record flags already known (kRecLongHint; the tail adds kKept / kHead).  Lane 0 stores the 32-byte record.
__device__ __forceinline__ void burst_reduce(long long offset, int nwin, bool val0, bool val1, float peak, float v0,
                                             float v1, bool dem, float x1, float x0, float y1, float y0, unsigned xflags,
                                             Rec* out, int lane) {
  unsigned no_hint = 0u;
  const float med = noise_median(nwin, val0, val1, v0, v1, no_hint);
  const bool bitA = dem && x1 > x0, bitB = dem && lane < 48 && y1 > y0;                     // demod.py:95
  const unsigned long long ma = __ballot(bitA), mb = __ballot(bitB);
  rec_store_head(out, offset, peak, med, lane);
  rec_store_bits(out, ma, mb, (dem ? kDemod : 0u) | xflags, lane);
}

// Mode S parity pre-filter of one finished record (one THREAD per record, in the tail: k_compact), from its packed
// bits: the same verdict as parity_prefilter above.  Syndrome = message polynomial mod G (decoder.py:693-714: CRC of
// the first L-24 bits XOR the last 24), byte-wise with the 256-entry table of v(x) * x^24 mod G.
struct Crc8Tab { unsigned t[256]; };
constexpr Crc8Tab make_crc8_tab() {
  Crc8Tab tab{};
  for (unsigned v = 0; v < 256; ++v) {
    unsigned c = v << 16;
    for (int k = 0; k < 8; ++k) {
      c <<= 1;
      if (c & 0x1000000u) c ^= 0x1FFF409u;
    }
    tab.t[v] = c & 0xFFFFFFu;
  }
  return tab;
}
__device__ __forceinline__ unsigned parity_flags_of(unsigned long long w2, unsigned long long w3) {
  static constexpr Crc8Tab tab = make_crc8_tab();
  const unsigned df = (unsigned)(w2 & 0xFFu) >> 3;                                         // decoder.py:551
  const unsigned dfb = 1u << df;
  const bool lng = (dfb & kDfLongSet) != 0, known = lng || (dfb & kDfShortSet) != 0;
  unsigned flags = df << kDfShift;
  if (lng) flags |= kLongFmt;
  if (known) flags |= kKnownDf;
  if (dfb & kDfPiSet) {                                                                     // decoder.py:625,679
    const int nb = lng ? 14 : 7;
    unsigned crc = 0u, last = 0u;
    for (int k = 0; k < nb; ++k) {
      const unsigned byte = (unsigned)((k < 8 ? (w2 >> (8 * k)) : (w3 >> (8 * (k - 8)))) & 0xFFu);
      if (k < nb - 3) crc = ((crc << 8) ^ tab.t[((crc >> 16) ^ byte) & 0xFFu]) & 0xFFFFFFu;
      else last = (last << 8) | byte;
    }
    if ((crc ^ last) == 0u) flags |= kParityOk;
  }
  return flags;
}

// Record of a centre whose samples come from global memory (k_longrun: pulses longer than the LDS window).
template <int MODE>
__device__ void burst_finish(const DetectArgs& a, const BurstFetch<MODE>& f, Rec* out, int lane) {
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
  burst_reduce(a.origin + p, nwin, val0, val1, peak, v0, v1, dem, x1, x0, y1, y0, f.xflags, out, lane);
}

// Record of a centre found by k_detect, built by the whole wavefront while the centre's tile is still in its LDS
// window s_x[-kBack .. kWWin) (index j <-> sample t0 + j; samples outside the buffer are zeros there too): the noise
// window always lies inside it, the bit samples do for all of a 2 Msps burst that starts in the tile and for the
// beginning of longer ones (4 Msps and up).  The bits of such a burst are sliced as their samples ARRIVE in the window:
// the record gets its first half now, the burst goes on the wavefront's pending list (PendList, in LDS) with the bits it
// already has, and pend_step() takes the rest from the next tile(s) -- every sample still comes from LDS, nothing is
// gathered from global memory.  (Only when the list is full, or at the very end of the wavefront's chunk, the missing
// samples are read from global memory instead: pend_flush / the smp path below.)
// Inlined into k_detect's hit loop: a real call would cost the callee's entry wait for ALL outstanding memory
// operations, i.e. for the prefetch of the next tile, once per hit.
// (the rarely needed fields -- data, n, in0_base, dem_hi -- are read through `c`, the kernel's argument block in memory,
// where they are needed: see adsb_cold below)
template <class CP>
struct WinArgs {
  CP c; long long origin; float scale; int sps;
};
constexpr int kMaxPend = 8;
struct PendEntry { int slot; int p; unsigned flags; int pad_; unsigned long long ma, mb; };    // p = centre relative to the CURRENT t0
struct PendList { PendEntry e[kMaxPend]; };

// Bit pair k of a burst centred at p: samples j1 = p + 8*sps + k*sps and j1 + half (demod.py:75-91).  A pair is taken by
// the first window that holds both samples: pairs with j1 + half in [lo, kWWin) -- lo = 0 when the burst is met, kFwd
// (= kWWin - kWTile) on every later tile.  Returns the ballots of the bits taken now (bits 0..63 / 64..111).
__device__ __forceinline__ void slice_window(const float* s_x, int p, int sps, int half, int lo, int lane,
                                             unsigned long long* ma, unsigned long long* mb) {
  const int ja = p + 8 * sps + lane * sps, jb = ja + 64 * sps;
  const bool ta = ja + half >= lo && ja + half < kWWin, tb = lane < 48 && jb + half >= lo && jb + half < kWWin;
  float x1 = 0.0f, x0 = 0.0f, y1 = 0.0f, y0 = 0.0f;
  if (ta) { x1 = s_x[ja]; x0 = s_x[ja + half]; }
  if (tb) { y1 = s_x[jb]; y0 = s_x[jb + half]; }
  *ma = __ballot(ta && x1 > x0);                             // demod.py:95
  *mb = __ballot(tb && y1 > y0);
}

// EASY (wave-uniform, known per tile): the whole 100-sample noise window of every centre of the tile lies inside the
// framer's input and every burst that starts in the tile ends inside the demod's -- the usual tile; nothing is clipped.
template <int MODE, bool EASY, class CP>
__device__ __forceinline__ void burst_from_window(WinArgs<CP> a, const float* s_x, long long t0, int p, unsigned xflags,
                                                  Rec* out, int slot, int lane, PendList* pend, int* n_pend,
                                                  unsigned& med_hint) {
  const int sps = a.sps, half = sps >> 1;
  const long long P = t0 + p;
  int nwin = kNoise, wl = p - kNoise;
  bool dem = true;
  if constexpr (!EASY) {
    long long wlo = P - kNoise;                              // framer.py:156: in0[max(0, pulse_idx-100) : pulse_idx]
    const long long in0_base = a.c->in0_base;
    if (wlo < in0_base) wlo = in0_base;
    nwin = (int)(P - wlo);
    wl = (int)(wlo - t0);                                    // >= p - 100 >= -kBack
    dem = P + 119ll * sps + half < a.c->dem_hi;              // demod.py:76,82 (sps even)
  }
  const bool val0 = EASY || lane < nwin, val1 = lane + 64 < nwin;
  const float v0 = val0 ? s_x[wl + lane] : 0.0f;
  const float v1 = val1 ? s_x[wl + lane + 64] : 0.0f;
  const float peak = s_x[p];
  const float med = noise_median<mode_is_iq8(MODE)>(nwin, val0, val1, v0, v1, med_hint);
  rec_store_head(out, a.origin + P, peak, med, lane);
  const unsigned flags = (dem ? kDemod : 0u) | xflags;
  unsigned long long ma = 0ull, mb = 0ull;
  if (dem) {                                                 // wave-uniform
    if (p + 119 * sps + half < kWWin) {
      // every 2 Msps burst that starts in the tile: all 224 bit samples lie in the LDS window
      slice_window(s_x, p, sps, half, 0, lane, &ma, &mb);
    } else if (*n_pend < kMaxPend) {
      slice_window(s_x, p, sps, half, 0, lane, &ma, &mb);    // what is here already; the rest when it arrives
      if (lane == 0) {
        PendEntry e;
        e.slot = slot; e.p = p; e.flags = flags; e.pad_ = 0; e.ma = ma; e.mb = mb;
        pend->e[*n_pend] = e;
      }
      *n_pend += 1;
      return;
    } else {
      const int j0 = p + 8 * sps + lane * sps;               // demod.py:75,87
      auto smp = [&](int j) -> float { return (j < kWWin) ? s_x[j] : xg<MODE>(a.c->data, a.c->n, t0 + j, a.scale); };
      const float x1 = smp(j0), x0 = smp(j0 + half);
      float y1 = 0.0f, y0 = 0.0f;
      if (lane < 48) { y1 = smp(j0 + 64 * sps); y0 = smp(j0 + 64 * sps + half); }
      ma = __ballot(x1 > x0);
      mb = __ballot(lane < 48 && y1 > y0);
    }
  }
  rec_store_bits(out, ma, mb, flags, lane);
}

// A wavefront's output stage.  Records and list words are not stored to global memory where they are made (lane 0, 16 + 16 + 8
// bytes per burst, three store instructions at three different moments): they are collected in LDS and written kStage at a
// time -- 512 + 128 contiguous bytes, one store instruction each, whole 128-byte lines -- or at the end of the chunk (the
// usual list of a bulk pass, ~12 entries: once).  Why: HBM pays for every scattered partial-line write with a bus
// turn-around in the middle of the read stream, and this kernel is HBM-bound: with the per-burst stores the complex64
// headline ran 2.5-8 % slower than with none at all, by an amount that depended on where the pass's output buffers happened
// to lie (profiles/r04_ab_output_stage.txt).
constexpr int kStage = 16;
struct alignas(16) StageBuf { Rec rec[kStage]; unsigned long long cand[kStage]; };
struct Stage {
  StageBuf* buf;               // LDS: slots [base, base + kStage) of this unit's list
  Rec* recs;                   // the unit's list in global memory (a.recs + unit * rec_cap)
  unsigned long long* cands;
  int base;                    // wave-uniform: first slot still in the stage
  int cap;                     // slots of the list (rec_cap)
};
// everything in front of slot `upto` (<= base + kStage) goes to global memory
__device__ __forceinline__ void stage_flush(Stage& st, int upto, int lane) {
  const int cnt = upto - st.base;                            // wave-uniform
  if (cnt > 0) {
    adsb_wave_sync();                                        // lane 0's stage writes lie in front of the reads below
    // Whole 128-byte lines only: four records, sixteen list words (the host makes rec_cap a multiple of 16, so every list
    // and every stage-full of it starts on a line).  What lies behind the last entry of a list is never read, and a line
    // that is written completely needs no fill first (measured: -0.3 ... -0.9 % against writing exactly cnt entries).
    const int room = st.cap - st.base;                         // (a list that is not a multiple of 16 long: never past its end)
    const int nr = ((cnt + 3) & ~3) < room ? ((cnt + 3) & ~3) : room;
    const int nc = kStage < room ? kStage : room;
    // NON-TEMPORAL stores (round 5): the lines are written once and read once, much later, by the tail -- written through
    // instead of sitting dirty in the L2 until the read stream evicts them.  Measured on one box, same process layout
    // (profiles/r05_slot_variants_nt_arena.txt): k_detect 1.283-1.288 -> 1.270-1.279 ms per 2^30 samples, and in a process
    // whose output buffers had landed badly (two of three slots at 1.332) 1.308.
    if (lane < 2 * nr) {
      const adsb_u64x2 v = reinterpret_cast<const adsb_u64x2*>(st.buf->rec)[lane];
      adsb_st_stream(reinterpret_cast<adsb_u64x2*>(st.recs + st.base) + lane, v);
    }
    if (lane < nc) adsb_st_stream(st.cands + st.base + lane, st.buf->cand[lane]);
    adsb_wave_sync();                                        // ... and these reads in front of the stage's next use
    st.base = upto;
  }
}

// second half of a record that is still in the stage (the pointer is typed as LDS on purpose: a store through a generic
// pointer next to the global-memory one of the other branch is merged with it into ONE flat store by the compiler)
__device__ __forceinline__ void stage_store_bits(const Stage& st, int slot, unsigned long long ma, unsigned long long mb,
                                                 unsigned flags, int lane) {
  if (lane == 0) {
    ADSB_LDS Rec* o = (ADSB_LDS Rec*)(st.buf->rec + (slot - st.base));
    o->w[2] = __builtin_bswap64(__brevll(ma));
    o->w[3] = (__builtin_bswap64(__brevll(mb)) & 0xFFFFFFFFFFFFull) | ((unsigned long long)flags << 48);
  }
}

// Once per tile, after its body is in the window: every pending burst takes the bit pairs that have arrived; a burst
// whose last pair is in is finished (second half of its record stored) and leaves the list.  t_next = false on the
// tile the entries were made (nothing to do yet); the list is compacted in place, order kept.
template <bool STAGED>
__device__ __forceinline__ void pend_step(PendList* pend, int* n_pend, const float* s_x, const Stage& st, int sps, int half, int lane) {
  int keep = 0;
  for (int i = 0; i < *n_pend; ++i) {                        // wave-uniform
    PendEntry e = pend->e[i];
    e.p -= kWTile;                                           // the window has moved on by one tile
    unsigned long long ma, mb;
    slice_window(s_x, e.p, sps, half, kFwd, lane, &ma, &mb);
    e.ma |= ma; e.mb |= mb;
    adsb_wave_sync();                                        // every lane has read entry i before lane 0 rewrites the list
    if (e.p + 119 * sps + half < kWWin) {                    // the last pair (k = 111) has been taken
      // (its record is normally still in the stage; a list that has been flushed since takes the half record directly)
      if (STAGED && e.slot >= st.base) stage_store_bits(st, e.slot, e.ma, e.mb, e.flags, lane);
      else rec_store_bits(st.recs + e.slot, e.ma, e.mb, e.flags, lane);
    } else {
      if (lane == 0) pend->e[keep] = e;
      ++keep;
    }
    adsb_wave_sync();
  }
  *n_pend = keep;
}

// End of the wavefront's chunk: what is still pending takes its missing samples from global memory.
template <int MODE, bool STAGED, class CP>
__device__ __forceinline__ void pend_flush(const PendList* pend, int n_pend, WinArgs<CP> a, const float* s_x, long long t0,
                                           const Stage& st, int lane) {
  const int sps = a.sps, half = sps >> 1;
  for (int i = 0; i < n_pend; ++i) {
    const PendEntry e = pend->e[i];                          // p relative to t0 (the last tile's start)
    const int ja = e.p + 8 * sps + lane * sps, jb = ja + 64 * sps;
    auto smp = [&](int j) -> float { return (j < kWWin) ? s_x[j] : xg<MODE>(a.c->data, a.c->n, t0 + j, a.scale); };
    const bool ta = ja + half >= kWWin, tb = lane < 48 && jb + half >= kWWin;     // pairs no window has held
    float x1 = 0.0f, x0 = 0.0f, y1 = 0.0f, y0 = 0.0f;
    if (ta) { x1 = smp(ja); x0 = smp(ja + half); }
    if (tb) { y1 = smp(jb); y0 = smp(jb + half); }
    const unsigned long long ma = e.ma | __ballot(ta && x1 > x0), mb = e.mb | __ballot(tb && y1 > y0);
    if (STAGED && e.slot >= st.base) stage_store_bits(st, e.slot, ma, mb, e.flags, lane);
    else rec_store_bits(st.recs + e.slot, ma, mb, e.flags, lane);
  }
}

// ---- k_detect: the streaming kernel, one independent stream segment per WAVEFRONT -------------------
// Each wavefront ("unit" = blockIdx*4 + wave) walks its own contiguous chunk of the stream tile by tile (kWTile
// samples) with its own sliding LDS window s_x[-kBack .. kWWin) of |IQ|^2 floats (the forward halo of one tile is
// the head of the next, so every sample is fetched from HBM exactly once), its own threshold masks, rise list and
// output list.  Nothing is shared between the wavefronts of a workgroup: there is no workgroup barrier.
//
// The kernel is bound by the number of instructions a wavefront issues per tile (measured: rate x instructions per
// tile is the same for every input format), so the tile loop is built to issue as few as possible:
//   * commit: 16-byte loads one tile ahead -> |IQ|^2 -> LDS, and the running integer maximum of the bit patterns (one
//     v_max3 per two samples).  A body whose maximum stays below the threshold's bit pattern ("quiet": half of the
//     tiles at 1 k bursts/s) costs nothing more than a zero written to its mask units.
//   * only an active body pays for its threshold masks (adsb_above4: compare + add-with-carry, 2 instructions per sample)
//     -- mask unit u covers samples [16u, 16u+16) of the window; units are 16-bit halves of the dwords the fall search
//     reads.  The 8-bit formats build them in body_commit from the registers that still hold the converted samples (a
//     lane's load holds eight consecutive samples = one mask byte; round 6); the other formats' lanes read their 16
//     consecutive floats back from LDS (unit_mask).
//   * rises by 16-bit mask algebra per lane, the ordered rise list by a DPP prefix sum and a short per-lane loop that
//     stores positions only; fall, centre and the 16 chip taps per rise with one lane per rise.
//   * the tile counter and every per-tile condition are 32-bit scalars computed once per unit.
constexpr int kUnit = 16;                        // samples per mask unit = one lane's share of a tile
constexpr int kUnits = kWWin / kUnit;            // 80 units in the window
constexpr int kTileUnits = kWTile / kUnit;       // 64: unit l of the tile belongs to lane l
constexpr int kHeadUnits = kFwd / kUnit;         // 16 units of forward halo
constexpr int kMaskDwords = kUnits / 2;          // the same masks read as 32-bit words
static_assert(kTileUnits == 64 && kUnits % 2 == 0 && kFwd == 256 && kBack == 128, "lane <-> unit mapping and the slide copies");
typedef unsigned short __attribute__((may_alias)) u16_alias;
typedef unsigned char __attribute__((may_alias)) u8_alias;
typedef unsigned __attribute__((may_alias)) u32_alias;

// entries of the per-tile hit list (16 bits, written over the rise list in place)
constexpr unsigned kHitValid = 0x8000u;          // matched centre: low 11 bits = centre relative to the tile
constexpr unsigned kHitLongHint = 0x4000u;       // ... whose first data bit is set (long-aware gate only)
constexpr unsigned kHitLongPulse = 0x2000u;      // a pulse that leaves the LDS window: low 11 bits = its RISE

// The body of one tile (kWTile samples) in flight between HBM and LDS: ITER 16-byte loads per lane, 1 KiB contiguous
// per wave instruction; lane l's load k holds samples k*64*SPL + SPL*l ... (+SPL) of the body.
template <int MODE>
struct Body {
  static constexpr int BPS = mode_bytes(MODE);
  static constexpr int ITER = kWTile * BPS / 1024;          // complex64: 8, float / int16 IQ: 4, 8-bit IQ: 2
  static constexpr int SPL = 16 / BPS;                       // samples per lane and load: 2 / 4 / 8
  float4 q[ITER];
};

template <int MODE>
__device__ __forceinline__ void body_issue(Body<MODE>& b, const char* src, int lane) {
  const unsigned lo = (unsigned)lane * 16u;                   // scalar base + one 32-bit lane offset
#pragma unroll
  for (int k = 0; k < Body<MODE>::ITER; ++k) b.q[k] = adsb_ld_stream<float4>(src + (k * 1024 + lo));
}

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imax3(int a, int b, int c) { return imax(imax(a, b), c); }     // v_max3_i32

// |IQ|^2 of the SPL samples of one 16-byte load (each format's exact arithmetic, see mag2f / mag2_iq16 / mag2_iq8)
template <int MODE>
__device__ __forceinline__ void body_convert(const float4& q, float scale, float* m) {
  if constexpr (MODE == 0) {
    m[0] = adsb_mag2(q.x, q.y);                               // (the same arithmetic as mag2f, spelled for the streaming loop)
    m[1] = adsb_mag2(q.z, q.w);
  } else if constexpr (MODE == 1) {
    m[0] = q.x; m[1] = q.y; m[2] = q.z; m[3] = q.w;
  } else if constexpr (MODE == 2) {
    m[0] = mag2_iq16(__builtin_bit_cast(unsigned, q.x), scale);
    m[1] = mag2_iq16(__builtin_bit_cast(unsigned, q.y), scale);
    m[2] = mag2_iq16(__builtin_bit_cast(unsigned, q.z), scale);
    m[3] = mag2_iq16(__builtin_bit_cast(unsigned, q.w), scale);
  } else if constexpr (MODE == kModeSc8Pow2) {
    // power-of-two scale: i*i + q*q as an integer (v_dot4c_i32_i8 on the packed bytes) -- 3 instead of 4.5 vector
    // instructions per sample, every step exact (integers below 2^16, then a power of two)
    const unsigned u[4] = {__builtin_bit_cast(unsigned, q.x), __builtin_bit_cast(unsigned, q.y),
                           __builtin_bit_cast(unsigned, q.z), __builtin_bit_cast(unsigned, q.w)};
    const float s2 = __fmul_rn(scale, scale);
    // The first sample of a word: the other sample's bytes masked out of one operand.  The dot product ACCUMULATES (v_dot4c_i32_i8 adds onto its destination): started from the bit pattern of 2^23 the sum IS
    // the float 2^23 + (i*i + q*q) (sum < 2^16), and one fused multiply-add per sample -- (2^23 + m) * s2 - 2^23 * s2, exact
    // for a power-of-two s2 -- replaces the conversion and the multiply.  The second sample of a word: all four byte
    // products onto (2^23 - m0), i.e. 2 * bits(2^23) minus the first result, instead of a second masked operand.
    constexpr unsigned kTwo23 = 0x4B000000u;
    const float c23 = -__fmul_rn(8388608.0f, s2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned a0 = (unsigned)adsb_sdot4((int)u[j], (int)(u[j] & 0x0000FFFFu), (int)kTwo23);
      const unsigned a1 = (unsigned)adsb_sdot4((int)u[j], (int)u[j], (int)(2u * kTwo23 - a0));
      m[2 * j] = __builtin_fmaf(__builtin_bit_cast(float, a0), s2, c23);
      m[2 * j + 1] = __builtin_fmaf(__builtin_bit_cast(float, a1), s2, c23);
    }
  } else if constexpr (MODE == kModeCu8Pow2) {
    // offset binary, power-of-two scale: t = x.x + x.1 over the sample's two bytes (x = byte ^ 0x80 as a signed byte;
    // x^2 + x = x (x + 1) >= 0, t <= 2 * 128 * 127), accumulated onto the bit pattern of 2^23 like the int8 instance;
    // |IQ|^2 = (4 t + 2) s2 = fma(2^23 + t, 4 s2, (2 - 2^25) s2): 2^25 - 2 has 24 significant bits, every step is exact.
    // The second sample of a word takes all four bytes onto 2 * bits(2^23) minus the first result.
    const unsigned u[4] = {__builtin_bit_cast(unsigned, q.x), __builtin_bit_cast(unsigned, q.y),
                           __builtin_bit_cast(unsigned, q.z), __builtin_bit_cast(unsigned, q.w)};
    const float s2 = __fmul_rn(scale, scale);
    const float s4 = __fmul_rn(4.0f, s2), c25 = __fmul_rn(-33554430.0f, s2);
    constexpr unsigned kTwo23 = 0x4B000000u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned x = u[j] ^ 0x80808080u;
      unsigned a0 = (unsigned)adsb_sdot4((int)x, (int)(x & 0x0000FFFFu), (int)kTwo23);
      a0 = (unsigned)adsb_sdot4((int)x, 0x00000101, (int)a0);
      unsigned a1 = (unsigned)adsb_sdot4((int)x, (int)x, (int)(2u * kTwo23 - a0));
      a1 = (unsigned)adsb_sdot4((int)x, 0x01010101, (int)a1);
      m[2 * j] = __builtin_fmaf(__builtin_bit_cast(float, a0), s4, c25);
      m[2 * j + 1] = __builtin_fmaf(__builtin_bit_cast(float, a1), s4, c25);
    }
  } else {
    const unsigned u[4] = {__builtin_bit_cast(unsigned, q.x), __builtin_bit_cast(unsigned, q.y),
                           __builtin_bit_cast(unsigned, q.z), __builtin_bit_cast(unsigned, q.w)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      m[2 * j] = mag2_iq8<MODE>(u[j] & 0xFFFFu, scale);
      m[2 * j + 1] = mag2_iq8<MODE>(u[j] >> 16, scale);
    }
  }
}

// Registers -> |IQ|^2 floats at sx_body[0 .. kWTile) (= s_x + kFwd); returns the lane's maximum over the BIT PATTERNS
// of its samples as signed integers: for a threshold > 0, "no sample of the body >= thr" <=> every lane's maximum is
// below the threshold's bit pattern (non-negative floats order like their bits; negative values and -0.0 are negative
// integers; a NaN with the sign clear is larger than everything and only sends the body down the exact path).
// REISSUE: every register is loaded again from `next` (this wavefront's next body, entirely inside the buffer) the
// moment its samples have been converted.
// MASKS (the 8-bit formats: a lane's load holds EIGHT consecutive samples): when the body is active -- some lane's maximum
// reaches thr_bits, or !thr_pos -- the threshold masks of the lane's 2 x 8 samples are built right here from the registers
// (*bm = mask byte of load 0 | mask byte of load 1 << 8; mask byte k covers body samples [512 k + 8 lane, + 8)) instead of
// from floats read back from LDS (unit_mask: four ds_read_b128 at a 64-byte lane stride = 4-way bank conflicts, and a
// round trip through the LDS in the middle of every active tile); *active tells which.
template <int MODE, bool REISSUE, bool MASKS = false>
__device__ __forceinline__ int body_commit(Body<MODE>& b, float* sx_body, float scale, int lane, const char* next,
                                           float thr = 0.0f, int thr_bits = 0, bool thr_pos = true, unsigned* bm = nullptr,
                                           bool* active = nullptr) {
  using B = Body<MODE>;
  int mx = (int)0x80000000u;
  const unsigned lo = (unsigned)lane * 16u;
  // convert everything, THEN reload all registers back to back (one contiguous 8 / 4 / 2 KB request burst per wavefront:
  // measured 1.5-4 % faster for complex64 than reloading each register right after its conversion), then store.  The
  // reload addresses are made to depend on the last converted sample, so the loads go into the registers just consumed
  // instead of a second set that would be copied over at the end of the tile
  float m[B::ITER][B::SPL];
#pragma unroll
  for (int k = 0; k < B::ITER; ++k) body_convert<MODE>(b.q[k], scale, m[k]);
  if (REISSUE) {
    const unsigned lo2 = adsb_after(lo, m[B::ITER - 1][B::SPL - 1]);
#pragma unroll
    for (int k = 0; k < B::ITER; ++k) b.q[k] = adsb_ld_stream<float4>(next + (k * 1024 + lo2));
  }
#pragma unroll
  for (int k = 0; k < B::ITER; ++k) {
    float* dst = sx_body + (k * 64 * B::SPL + B::SPL * lane);
    if constexpr (B::SPL == 2) {
      *reinterpret_cast<float2*>(dst) = float2{m[k][0], m[k][1]};
      mx = imax3(mx, __builtin_bit_cast(int, m[k][0]), __builtin_bit_cast(int, m[k][1]));
    } else {
#pragma unroll
      for (int j = 0; j < B::SPL; j += 4) {
        *reinterpret_cast<float4*>(dst + j) = float4{m[k][j], m[k][j + 1], m[k][j + 2], m[k][j + 3]};
        mx = imax3(mx, __builtin_bit_cast(int, m[k][j]), __builtin_bit_cast(int, m[k][j + 1]));
        mx = imax3(mx, __builtin_bit_cast(int, m[k][j + 2]), __builtin_bit_cast(int, m[k][j + 3]));
      }
    }
  }
  if constexpr (MASKS) {
    static_assert(B::SPL == 8 && B::ITER == 2, "mask bytes from registers: eight consecutive samples per lane and load");
    const bool act = !thr_pos || __ballot(mx >= thr_bits) != 0ull;      // wave-uniform
    unsigned v = 0u;
    if (act) {
      unsigned a0 = adsb_above4(0u, m[0][7], m[0][6], m[0][5], m[0][4], thr);
      a0 = adsb_above4(a0, m[0][3], m[0][2], m[0][1], m[0][0], thr);
      unsigned a1 = adsb_above4(0u, m[1][7], m[1][6], m[1][5], m[1][4], thr);
      a1 = adsb_above4(a1, m[1][3], m[1][2], m[1][1], m[1][0], thr);
      v = a0 | (a1 << 8);
    }
    *bm = v;
    *active = act;
  }
  return mx;
}

// The 16-bit threshold mask (framer.py:83-84) of the 16 consecutive floats at p (16-byte aligned): bit k <-> p[k].
__device__ __forceinline__ unsigned unit_mask(const float* p, float thr) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  const float4 c = reinterpret_cast<const float4*>(p)[2], d = reinterpret_cast<const float4*>(p)[3];
  unsigned acc = 0;
  acc = adsb_above4(acc, d.w, d.z, d.y, d.x, thr);
  acc = adsb_above4(acc, c.w, c.z, c.y, c.x, thr);
  acc = adsb_above4(acc, b.w, b.z, b.y, b.x, thr);
  acc = adsb_above4(acc, a.w, a.z, a.y, a.x, thr);
  return acc;
}

// The 16-chip preamble test (framer.py:137-147) for a centre whose taps all lie in the LDS window: chip k is
// tp[k*half] > tp[0]/2 and the chips must spell 1010000101000000.  Chips 0, 2, 7, 9 are compared one by one (a NaN
// tap fails); the twelve others must all be "not above", i.e. their maximum is not above: five v_max3 and one v_max
// (a NaN among them counts as a low chip and is skipped by the maximum, which is the same thing) and ONE compare.
// HALF > 0: samples per chip known at compile time -- the taps become ds_read2_b32 with immediate offsets.
template <int HALF>
__device__ __forceinline__ bool chips_match(const float* tp, int half_rt) {
  const int h = HALF ? HALF : half_rt;
  float tap[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) tap[k] = tp[k * h];
  const float hp = __fmul_rn(tap[0], 0.5f);                  // tap 0 IS in0[pulse_idx]; /2 is exact
  const bool hi = (tap[0] > hp) & (tap[2] > hp) & (tap[7] > hp) & (tap[9] > hp);
  float mx = adsb_fmax3(tap[1], tap[3], tap[4]);
  mx = adsb_fmax3(mx, tap[5], tap[6]);
  mx = adsb_fmax3(mx, tap[8], tap[10]);
  mx = adsb_fmax3(mx, tap[11], tap[12]);
  mx = adsb_fmax3(mx, tap[13], tap[14]);
  mx = adsb_fmax3(mx, tap[15], tap[15]);
  return hi & !(mx > hp);
}

// HALF = samples per chip (sps/2) when it is one of the instantiated rates (2, 4, 8, 20 Msps), else 0 = run-time value.
// WPB = wavefronts per workgroup of the kernel this is the body of: 1 for k_detect, kWaves for k_pass_small.
template <int MODE, int HALF, int WPB = kWaves>
__device__ __forceinline__ void detect_body(const DetectArgs& a, const int block) {
  __shared__ __attribute__((aligned(16))) float s_xa[WPB][kBack + kWWin];
  __shared__ __attribute__((aligned(16))) unsigned s_ma[WPB][kMaskDwords];
  __shared__ __attribute__((aligned(4))) unsigned short s_risea[WPB][kWTile / 2];
  __shared__ PendList s_penda[WPB];
  // The output stage is for the formats whose tile loop is bound by HBM (complex64, |IQ|^2 floats, int16: +1-5 %); the 8-bit
  // formats are bound by their instruction stream, for which the stage is more instructions (-2 ... -5 %): they keep the
  // store per burst.
  constexpr bool STAGED = !mode_is_iq8(MODE);
  __shared__ StageBuf s_stagea[STAGED ? WPB : 1];

  const int lane = threadIdx.x & 63, wave = adsb_uniform((int)(threadIdx.x >> 6));
  float* s_x = s_xa[wave] + kBack;                           // s_x[j] <-> sample t0 + j, j in [-kBack, kWWin)
  u32_alias* s_m32 = reinterpret_cast<u32_alias*>(s_ma[wave]);
  u16_alias* s_m16 = reinterpret_cast<u16_alias*>(s_ma[wave]);     // unit u <-> samples [16u, 16u + 16) of the window
  u16_alias* s_rise = reinterpret_cast<u16_alias*>(s_risea[wave]);
  PendList* pend = &s_penda[wave];
  int n_pend = 0;                                            // wave-uniform: bursts whose last bit samples have not arrived yet
  // SGPR budget: the argument block is ~45 scalar registers, most of them needed once per chunk or in rare branches only.
  // Those are read through cold() -- the same block as it lies in (kernarg) memory, one scalar load where it is used --
  // instead of living in registers across the tile loop (every k_detect instance spilled 60-90 SGPRs into vector lanes,
  // 8-15 v_readlane reloads per tile).  Hot fields (thr, scale, sps, origin, rec_cap, long_aware) stay by-value.
  auto cold = [&]() { return adsb_cold(a); };
  const long long unit = (long long)block * WPB + wave;
  const long long c0 = unit * cold()->chunk;
  long long c1 = c0 + cold()->chunk;
  if (c1 > cold()->scan_hi) c1 = cold()->scan_hi;
  const int half = HALF ? HALF : (a.sps >> 1);
  int nrec = 0;                                              // wave-uniform running count of this unit's list
  unsigned uflags = 0u;
  int lp = -1;                                               // per lane: largest paired pulse centre so far, relative to c0
  unsigned med_hint = 0u;                                    // wave-uniform: median key of this wavefront's previous burst
  int pred = adsb_uniform(above_at<MODE>(*cold(), c0 - 1) ? 1 : 0);
  Stage st{&s_stagea[STAGED ? wave : 0], a.recs + unit * a.rec_cap, a.cands + unit * a.rec_cap, 0, a.rec_cap};
  const float thr = a.thr;
  const bool thr_pos = thr > 0.0f;                           // else (thr <= 0 or NaN): every body takes the exact path
  const int thr_bits = __builtin_bit_cast(int, thr);

  // virtual rise in the zero history in front of a fresh stream: only possible when 0 >= thr
  if (unit == 0 && cold()->scan_lo < 0 && (0.0f >= a.thr) && !(cold()->prev_in0 >= a.thr)) {
    uflags |= 1u;
    if (lane == 0 && 0 < a.rec_cap) {
      *(STAGED ? &st.buf->cand[0] : &st.cands[0]) = cand_make(cold()->scan_lo, kPending | kNoMatch);
      const int li = atomicAdd(cold()->long_count, 1);
      if (li < cold()->long_cap) { LongRise e; e.rise = cold()->scan_lo; e.blk = 0; e.slot = 0; cold()->longlist[li] = e; }
    }
    nrec = 1;
  }

  // Tile counters of this unit (32-bit, wave-uniform), computed once: tile `it` covers [c0 + it*kWTile, +kWTile).
  //   it < it_rag  : the tile's own body lies inside the buffer (else: scalar reads, zeros past the end)
  //   it < it_re   : the NEXT tile's body lies inside the buffer and is needed -> that is what the registers reload
  //   it_i0 <= it < it_i1 : "interior": the whole tile is owned ([scan_lo, scan_hi)) and the window ends in front of
  //                  fall_hi -> no per-lane 64-bit ownership / end-of-call arithmetic
  constexpr int BPS = mode_bytes(MODE);
  int ntile = 0, it_re = 0, it_i0 = 0, it_i1 = 0, it_rag = 0, it_e0 = 0, it_e1 = 0;
  if (c0 < c1) {
    auto tiles_below = [](long long x) -> long long {          // number of it >= 0 with it*kWTile < x
      return x <= 0 ? 0 : (x + kWTile - 1) / kWTile;
    };
    auto clampi = [](long long v, long long hi) -> int { return (int)(v < 0 ? 0 : (v > hi ? hi : v)); };
    const long long nt = tiles_below(c1 - c0);
    ntile = (int)nt;
    // c0 + it*T + T < c1  and  c0 + it*T + 2T + kFwd <= n
    long long re = tiles_below(c1 - c0 - kWTile);
    const auto C = cold();
    const long long re2 = tiles_below(C->n - c0 - 2 * kWTile - kFwd + 1);
    if (re2 < re) re = re2;
    it_re = clampi(re, nt);
    it_rag = clampi(tiles_below(C->n - c0 - kFwd - kWTile + 1), nt);      // c0 + it*T + kFwd + T <= n: body inside the buffer
    // scan_lo <= c0 + it*T,  c0 + it*T + T <= scan_hi,  c0 + it*T + kWWin < fall_hi
    const long long i0 = tiles_below(C->scan_lo - c0);
    long long i1 = tiles_below(C->scan_hi - c0 - kWTile + 1);
    const long long i1f = tiles_below(C->fall_hi - c0 - kWWin);
    if (i1f < i1) i1 = i1f;
    it_i0 = clampi(i0, nt);
    it_i1 = clampi(i1, nt);
    // "easy" tiles (burst_from_window<.., EASY>): interior, c0 + it*T - kNoise >= in0_base, and the latest centre a tile
    // can hold (kMaxCentre) still has its last bit sample in front of dem_hi
    long long e0 = tiles_below(C->in0_base + kNoise - c0);
    long long e1 = tiles_below(C->dem_hi - c0 - kMaxCentre - 119ll * a.sps - (a.sps >> 1));
    if (e0 < i0) e0 = i0;
    if (e1 > i1) e1 = i1;
    it_e0 = clampi(e0, nt);
    it_e1 = clampi(e1, nt);
  }
  ntile = adsb_uniform(ntile); it_re = adsb_uniform(it_re); it_i0 = adsb_uniform(it_i0); it_i1 = adsb_uniform(it_i1);
  it_rag = adsb_uniform(it_rag); it_e0 = adsb_uniform(it_e0); it_e1 = adsb_uniform(it_e1);

  // Head of the window and back halo of the first tile (later tiles inherit both): once per chunk -- and a bulk pass has
  // tens of thousands of chunks: all six loads of a lane are in flight together (clamped index, value selected afterwards),
  // and the streaming loop issues the first body's loads BEFORE it fills the head: one memory latency at the start of a
  // chunk instead of seven in a row.
  auto head_fill = [&](const int lane) {
    constexpr int NQ = (kBack + kFwd) / 64;
    const auto C = cold();
    if (C->n > 0) {                                            // wave-uniform
      float hv[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) hv[q] = xg_nb<MODE>(C->data, C->n, c0 - kBack + lane + 64 * q, a.scale);
#pragma unroll
      for (int q = 0; q < NQ; ++q) s_x[lane + 64 * q - kBack] = hv[q];
    } else {
      for (int i = lane; i < kBack + kFwd; i += 64) s_x[i - kBack] = 0.0f;
    }
    adsb_wave_sync();
    s_m16[lane & (kHeadUnits - 1)] = (unsigned short)unit_mask(s_x + kUnit * (lane & (kHeadUnits - 1)), thr);
  };
  bool prev_active = true;                                   // the head units were computed exactly

  // -- B.2 one lane per rise: fall, centre, 16-chip test (framer.py:113,137-147); hits are compacted in order over the
  //    rise list in place (a hit's slot never lies beyond its rise's).  Then C: the tile's hits in stream order.
  auto rises_to_records = [&](auto easy_c, const int it, const long long t0, const int nr, const int lane) {
    constexpr bool EASY = decltype(easy_c)::value;
    const bool interior = EASY || (it >= it_i0 && it < it_i1);
    int nm = 0;
    bool hflag = false;
    const int trel = it * kWTile;
    for (int base = 0; base < nr; base += 64) {
      const int i = base + lane;
      unsigned res = 0u;
      if (i < nr) {
        const int r = (int)s_rise[i];
        // the first sub-threshold sample after r: 32 mask bits from r on (two aligned words), else word by word
        const int d = r >> 5, sh = r & 31;
        const unsigned long long two = ((unsigned long long)s_m32[d + 1] << 32) | s_m32[d];
        const unsigned inv = ~(unsigned)(two >> sh);      // bit 0 (the rise itself) is clear
        int f = inv ? r + __builtin_ctz(inv) : -1;
        if (inv == 0u) {                                    // rare: the pulse is longer than those 32 - sh samples
          int wd = d + 1;
          unsigned cur = ~s_m32[wd] & (~0u << sh);
          while (cur == 0u && ++wd < kMaskDwords) cur = ~s_m32[wd];
          if (wd < kMaskDwords && cur) f = wd * 32 + __builtin_ctz(cur);
        }
        if (f < 0) {
          if (EASY || interior || t0 + kWWin < cold()->fall_hi) res = kHitValid | kHitLongPulse | (unsigned)r;    // long pulse: k_order
          else if (!cold()->end_is_call_end) hflag = true;
        } else if (EASY || interior || t0 + f < cold()->fall_hi) {
          const int p = (r + f) >> 1;                    // framer.py:113
          lp = imax(lp, trel + p);                        // (centres increase along the stream)
          bool match = true;
          // all 16 taps inside the LDS window (one LDS round trip): always, at the instantiated rates up to 8 Msps
          if ((HALF >= 1 && kMaxCentre + 15 * HALF < kWWin) || p + 15 * half < kWWin) {
            match = chips_match<HALF>(s_x + p, half);
          } else {                                       // rare: taps past the window come from global memory
            const float hp = __fmul_rn(s_x[p], 0.5f);
            unsigned chips = 0;
#pragma unroll 1
            for (int k = 0; k < 16; ++k) {
              const int idx = p + k * half;
              const float v = (idx < kWWin) ? s_x[idx] : xg<MODE>(cold()->data, cold()->n, t0 + idx, a.scale);
              chips |= (v > hp ? 1u : 0u) << k;
            }
            match = chips == kTemplate;
          }
          if (match) {
            res = kHitValid | (unsigned)p;
            if (a.long_aware) {                          // first data bit (demod.py:87-95, k = 0): DF >= 16 = long reply
              const int i1 = p + 16 * half, i0 = i1 + half;
              const float v1 = (i1 < kWWin) ? s_x[i1] : xg<MODE>(cold()->data, cold()->n, t0 + i1, a.scale);
              const float v0 = (i0 < kWWin) ? s_x[i0] : xg<MODE>(cold()->data, cold()->n, t0 + i0, a.scale);
              if (v1 > v0) res |= kHitLongHint;
            }
          }
        } else if (!cold()->end_is_call_end) {
          hflag = true;
        }
      }
      const unsigned long long mb = __ballot(res != 0u);
      if (mb) {                                          // wave-uniform
        if (res) s_rise[nm + lanes_below(mb, lane)] = (unsigned short)res;
        nm += __popcll(mb);
      }
    }
    if constexpr (!EASY) {
      if (!interior && __ballot(hflag)) uflags |= 4u;
    }
    adsb_wave_sync();

    // -- C: this tile's hits, in stream order: the list word and, built by the whole wavefront from the LDS
    //    window while it still holds the burst's samples, the burst record (wave-uniform loop: a tile rarely has
    //    more than one or two hits)
    for (int m = 0; m < nm; ++m) {
      const int slot2 = nrec + m;
      if (slot2 >= a.rec_cap) break;                     // overflow: reported through the count, call is re-run
      if (STAGED && slot2 - st.base >= kStage) stage_flush(st, slot2, lane);     // the stage is full: sixteen entries to global memory
      unsigned long long* const cand_out = STAGED ? &st.buf->cand[slot2 - st.base] : &st.cands[slot2];
      Rec* const rec_out = STAGED ? &st.buf->rec[slot2 - st.base] : &st.recs[slot2];
      const unsigned e = (unsigned)adsb_uniform((int)s_rise[m]);
      const int v = (int)(e & 0x7FFu);
      if (e & kHitLongPulse) {
        if (lane == 0) {
          const long long rg = t0 + v;
          *cand_out = cand_make(rg, kPending | kNoMatch);
          const int li = atomicAdd(cold()->long_count, 1);
          if (li < cold()->long_cap) { LongRise le; le.rise = rg; le.blk = (int)unit; le.slot = slot2; cold()->longlist[li] = le; }
        }
      } else {
        const bool lh = (e & kHitLongHint) != 0;
        if (lane == 0) *cand_out = cand_make(t0 + (long long)v, lh ? kLongHint : 0u);
        burst_from_window<MODE, EASY>(WinArgs<decltype(cold())>{cold(), a.origin, a.scale, a.sps}, s_x, t0, v,
                                      lh ? kRecLongHint : 0u, rec_out, slot2, lane, pend, &n_pend, med_hint);
      }
    }
    nrec += nm;
  };

  // Everything a tile needs once its body is in the window (floats at s_x[kFwd ..)): mask units, pending bursts, rises,
  // hits, records, the slide.  One body of code, used by the streaming loop and by the loop for tiny inputs below.
  // The body's threshold masks.  MASKS formats (8-bit: mask BYTES, byte k of lane l covers body samples [512 k + 8 l, + 8)):
  // the streaming loop brings them along (have_bm: built by body_commit from the registers); the rare callers without them
  // (a tile rewritten by the scalar loop, inputs shorter than a tile) have them built here from the floats in the window,
  // in the same layout.  Other formats: lane l builds the 16-bit mask of body unit l from the window.
  auto process_tile = [&](const int it, const long long t0, const bool active, const int lane, const bool have_bm, unsigned bm) {
    constexpr bool MASKS = mode_is_iq8(MODE);
    adsb_wave_sync();
    if constexpr (MASKS) {
      if (!have_bm) {                                                  // wave-uniform
        bm = 0u;
        if (active) {
          const float* q0 = s_x + kFwd + 8 * lane;
          const float4 a = reinterpret_cast<const float4*>(q0)[0], b = reinterpret_cast<const float4*>(q0)[1];
          const float4 c = reinterpret_cast<const float4*>(q0 + 512)[0], d = reinterpret_cast<const float4*>(q0 + 512)[1];
          unsigned m0 = adsb_above4(0u, b.w, b.z, b.y, b.x, thr), m1 = adsb_above4(0u, d.w, d.z, d.y, d.x, thr);
          m0 = adsb_above4(m0, a.w, a.z, a.y, a.x, thr);
          m1 = adsb_above4(m1, c.w, c.z, c.y, c.x, thr);
          bm = m0 | (m1 << 8);
        }
      }
      u8_alias* s_m8 = reinterpret_cast<u8_alias*>(s_ma[wave]);
      s_m8[2 * kHeadUnits + lane] = (unsigned char)bm;
      s_m8[2 * kHeadUnits + 64 + lane] = (unsigned char)(bm >> 8);
    } else {
      // mask units of the body (units kHeadUnits .. kUnits): lane l owns body unit l
      bm = 0u;
      if (active) bm = unit_mask(s_x + kFwd + kUnit * lane, thr);      // wave-uniform branch
      s_m16[kHeadUnits + lane] = (unsigned short)bm;
    }
    adsb_wave_sync();
    // -- P: bursts met in earlier tiles take the bit samples that have arrived with this body
    if (n_pend > 0) pend_step<STAGED>(pend, &n_pend, s_x, st, a.sps, half, lane);

    // -- B: rises among the tile's own samples (units 0 .. 63); a tile can hold one only if this or the previous body
    //       had a sample above the threshold (or the sample in front of the tile was)
    int keepp = 0;
    if (active || prev_active || pred) {
      const bool interior = it >= it_i0 && it < it_i1;
      // -- B.1 rises / falls by mask algebra (framer.py:91-93): lane l owns samples [16 l, 16 l + 16)
      const unsigned W = (unsigned)s_m16[lane] | ((unsigned)s_m16[lane + 1] << 16);   // own unit, and the next one
      const unsigned pW = adsb_lane_up1(W, (unsigned)pred << 15);
      const unsigned m16 = W & 0xFFFFu, prev16 = ((m16 << 1) | ((pW >> 15) & 1u)) & 0xFFFFu;
      unsigned own16 = 0xFFFFu;
      if (!interior) {
        const long long sbase = t0 + 16ll * lane;
        own16 = (unsigned)bit_range(cold()->scan_lo - sbase, cold()->scan_hi - sbase) & 0xFFFFu;
      }
      unsigned piece = m16 & ~prev16 & own16;                 // rises among my 16 samples
      const unsigned long long anyr = __ballot(piece != 0u), anyf = __ballot((~m16 & prev16 & own16) != 0u);
      keepp = (adsb_readlane((int)W, 63) >> 15) & 1;
      uflags |= (anyr ? 1u : 0u) | (anyf ? 2u : 0u);
      if (anyr) {                                            // wave-uniform
        // The wavefront that has met rises runs its per-rise / per-record chain at raised issue priority (s_setprio) and drops
        // back when the tile's records are done: its SIMD neighbours are mostly streaming quiet tiles and lose nothing
        // measurable -- uint8 +3.5 %, int8 +1.1 %, int16 +0.6 %, |IQ|^2 floats +0.2 %, complex64 0
        // (profiles/r05_ab_wave_priority_cu8.txt, r05_ab_wave_priority_all.txt)
        adsb_setprio<3>();
        // Ordered rise list: the slots of a lane follow from the prefix sum of the per-lane counts; every lane then
        // stores the positions of its (at most 8) rises in a short loop.
        const unsigned cnt = (unsigned)__builtin_popcount(piece);
        const unsigned incl = adsb_wave_incl_scan(cnt);
        const int nr = adsb_readlane((int)incl, 63);
        unsigned slot = incl - cnt;
        while (piece) {
          s_rise[slot++] = (unsigned short)(kUnit * lane + __builtin_ctz(piece));
          piece &= piece - 1u;
        }
        adsb_wave_sync();

        // -- B.2 + C for this tile's rise list, in two instances: EASY (wave-uniform; the usual tile) = the tile is
        //    interior, every centre's noise window lies inside the framer's input and every burst that starts here ends
        //    inside the demod's -- no per-lane ownership / end-of-call arithmetic, no clipping; else the general code
        const bool easy = it >= it_e0 && it < it_e1;
        if (easy) rises_to_records(BoolC<true>{}, it, t0, nr, lane);
        else rises_to_records(BoolC<false>{}, it, t0, nr, lane);
        adsb_setprio<0>();
      }
    }

    if (n_pend > 0 && it + 1 == ntile) {                     // end of this wavefront's chunk
      adsb_wave_sync();
      pend_flush<MODE, STAGED>(pend, n_pend, WinArgs<decltype(cold())>{cold(), a.origin, a.scale, a.sps}, s_x, t0, st, lane);
      n_pend = 0;
    }
    // what the next tile inherits: back + forward halo (floats), the mask units of the forward halo and the last
    // threshold bit (sources and destinations are disjoint; all reads are issued before the writes)
    {
      const float4 keep = *reinterpret_cast<const float4*>(s_x + kWTile + 4 * lane);
      const float2 keepb = *reinterpret_cast<const float2*>(s_x + kWTile - kBack + 2 * lane);
      const unsigned short keepm = s_m16[kTileUnits + (lane & (kHeadUnits - 1))];
      adsb_wave_sync();
      *reinterpret_cast<float4*>(s_x + 4 * lane) = keep;
      *reinterpret_cast<float2*>(s_x - kBack + 2 * lane) = keepb;
      s_m16[lane & (kHeadUnits - 1)] = keepm;
    }
    pred = keepp;
    prev_active = active;
    // (the next tile's commit writes s_x[kFwd..] and mask units >= kHeadUnits; its wave_sync orders all of it)
  };

  const int lane_outer = lane;
  if (cold()->n >= kWTile) {
    // -- the streaming loop.  The body of tile `it` sits in registers, fetched one tile ahead; every register is reloaded
    // the moment its samples are converted -- from the next tile's body or, when there is none inside the buffer (last
    // tile of the chunk, ragged end of the buffer: one tile in ~200), from `clamp`, the last whole tile of the buffer: a
    // load nobody uses, but the loop has ONE unconditional path through commit and reload.  (With a second path the
    // prefetch registers were copied at the join, 32 moves per tile that had to wait for the loads just issued.)
    // A tile whose own body is not entirely inside the buffer (it >= it_rag: at most two per launch) was "prefetched"
    // from `clamp` too; its floats are then written by the scalar loop, zeros past the end.
    const char* const clamp = reinterpret_cast<const char*>(cold()->data) + (((cold()->n - kWTile) * (long long)BPS) & ~15ll);
    const char* nb = reinterpret_cast<const char*>(cold()->data) + (c0 + kFwd) * (long long)BPS;      // this tile's body
    // Prefetch depth D (tiles in flight per wavefront); the loop is unrolled D times so that each body is a fixed set of
    // registers and the compiler's s_waitcnt in front of a commit counts exactly the younger bodies' loads (a rotating set
    // meets a wait for everything at the loop head: round 3's attempt).  Measured with exact waits (round 4,
    // profiles/r04_ab_prefetch_depth.txt): two tiles ahead +2.2 % for int8 IQ with the dot-product conversion (the format
    // with the least arithmetic per byte), +0.8 % int16, 0 for uint8 / |IQ|^2 floats / complex64, three no better than
    // two -- the narrow formats are NOT short of bytes in flight; kept where it pays for its 50 % more code (measured again
    // in round 6 on the kernels with register-built masks: generic int8 -1 %, uint8 -1.5 % with two tiles ahead:
    // profiles/r06_ab_8bit_masks_median.txt).
    constexpr int D = (MODE == kModeSc8Pow2 || MODE == kModeCu8Pow2) ? 2 : 1;
    Body<MODE> body[D];
    if (ntile > 0) {
      body_issue(body[0], it_rag > 0 ? nb : clamp, lane);
#pragma unroll
      for (int d = 1; d < D; ++d) {                            // tile d: loadable iff d - 1 < it_re
        nb += (long long)kWTile * BPS;
        body_issue(body[d], d - 1 < it_re ? nb : clamp, lane);
      }
      head_fill(lane);
    }
    // nb: the body of tile it + D - 1 when step `it` starts
    auto step = [&](Body<MODE>& b, const int it, const bool real) {
      // The lane number through an opaque copy, renewed every tile: otherwise lane-derived addresses are hoisted out of
      // this loop as loop-invariant registers, which the register budget of five wavefronts per SIMD cannot hold.
      const int lane = adsb_opaque(lane_outer);
      const long long t0 = c0 + (long long)it * kWTile;        // (only the rare paths use it)
      nb += (long long)kWTile * BPS;                           // now: the body of tile it + D, which these registers hold next
      constexpr bool MASKS = mode_is_iq8(MODE);
      unsigned bm_regs = 0u;
      bool active = false;
      const int mx = body_commit<MODE, true, MASKS>(b, s_x + kFwd, a.scale, lane, it + D - 1 < it_re ? nb : clamp, thr, thr_bits,
                                                    thr_pos, &bm_regs, &active);
      if (real) {
        if constexpr (!MASKS) active = !thr_pos || __ballot(mx >= thr_bits) != 0ull;
        if (it >= it_rag) {
          adsb_wave_sync();                                    // every lane's commit stores lie in front of the rewrite below
          for (int i = lane; i < kWTile; i += 64) s_x[kFwd + i] = xg<MODE>(cold()->data, cold()->n, t0 + kFwd + i, a.scale);
          active = true;
        }
        process_tile(it, t0, active, lane, MASKS && it < it_rag, bm_regs);
      }
    };
    // A chunk with a number of tiles that is not a multiple of D ends inside the unrolled body: the steps past its end still
    // COMMIT (floats nobody reads, a reload from `clamp`) and skip the tile's work.  Leaving the loop there instead gives
    // the compiler a second path to the loop head on which the OTHER body's loads are the younger ones, and it then waits
    // for all loads in front of every commit (measured: that is what made the first attempt at two tiles ahead, in round
    // 3, worthless).
    for (int it = 0; it < ntile; it += D) {
      step(body[0], it, true);
#pragma unroll
      for (int d = 1; d < D; ++d) step(body[d], it + d, it + d < ntile);
    }
  } else {
    // -- inputs shorter than one tile (a GNU Radio work() call of a few hundred samples): scalar reads only
    if (ntile > 0) head_fill(lane_outer);
    for (int it = 0; it < ntile; ++it) {
      const int lane = adsb_opaque(lane_outer);
      const long long t0 = c0 + (long long)it * kWTile;
      for (int i = lane; i < kWTile; i += 64) s_x[kFwd + i] = xg<MODE>(cold()->data, cold()->n, t0 + kFwd + i, a.scale);
      process_tile(it, t0, true, lane, false, 0u);
    }
  }
  // what the stage still holds (the whole list of a usual chunk): to global memory, in one go
  if (STAGED) stage_flush(st, nrec < a.rec_cap ? nrec : a.rec_cap, lane_outer);
  // largest paired pulse centre of the unit: once per unit
#pragma unroll
  for (int d2 = 32; d2 >= 1; d2 >>= 1) lp = imax(lp, __shfl_xor(lp, d2));
  if (lane == 0) {
    cold()->blk_count[unit] = nrec;
    cold()->blk_lastp[unit] = lp >= 0 ? c0 + lp : kNoIndex;
    cold()->blk_flags[unit] = uflags;
  }
}
template <int MODE, int HALF>
__global__ void __launch_bounds__(64 * det_waves(MODE), kMinWaves) k_detect(DetectArgs a) {
  detect_body<MODE, HALF, det_waves(MODE)>(a, (int)blockIdx.x);
}

// ---- k_longrun: pulses whose run leaves the LDS window (or starts in the zero history) -------------
// One workgroup per entry scans forward cooperatively for the fall, then wave 0 finishes the pulse
// with global-memory taps and overwrites the placeholder.  Rare (CW / overload, or dense overlapping
// bursts): it is launched after every k_detect and returns at once when the list is empty.
template <int MODE>
__device__ __forceinline__ void longrun_entry(const DetectArgs& a, int e) {
  __shared__ unsigned long long s_found;   // fall index relative to rise+1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {
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
          burst_finish<MODE>(a, bf, a.recs + (long long)le.blk * a.rec_cap + le.slot, lane);
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
__device__ __forceinline__ void longrun_body(int bid, int nb, const DetectArgs& a) {
  int n_entries = *a.long_count;                 // written by k_detect, usually 0: then this is a no-op
  if (n_entries > a.long_cap) n_entries = a.long_cap;
  for (int e = bid; e < n_entries; e += nb) longrun_entry<MODE>(a, e);
}
// (bulk passes: k_order's workgroups take the entries whose list they own; small passes: longrun_body in k_tail_small /
// k_pass_small)

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
// A bulk pass with thousands of lists (k_detect runs several resident rounds of short chunks, adsb_hip.hip: enqueue)
// stages kScanRoundBig lists per round in the 8 KB of dynamic LDS the launch carries anyway (its padding, see
// enqueue_tail): a round costs about a microsecond of load latency and barriers whatever its size.
constexpr int kScanRound = 512;
constexpr int kScanRoundBig = 2048;
template <int ROUND>
__device__ __forceinline__ void scan_body(const int* blk_count, const long long* blk_lastp,
                                          const unsigned* blk_flags, int nblk, int rec_cap,
                                          const int* long_count, const unsigned long long* long_lastp,
                                          int* blk_off, Summary* sum, int* s_cnt) {
  constexpr int kScanRound = ROUND;
  constexpr int kScanPer = ROUND / kThreads;         // consecutive lists per thread and round
  __shared__ long long s_lp[kWaves];
  __shared__ unsigned s_fl[kWaves];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  long long lp = kNoIndex; unsigned fl = 0;
  int carry = 0;                                               // lists before this round hold `carry` centres
  for (int base = 0; base < nblk; base += kScanRound) {
    // coalesced: counts to LDS, max / OR reduced on the fly.  All loads of the round are issued before any is used
    // (clamped index, result masked): one memory latency per round, not one per 256 lists
    int cc[kScanPer];
    long long ll[kScanPer];
    unsigned ff[kScanPer];
#pragma unroll
    for (int q = 0; q < kScanPer; ++q) {
      const int b = base + q * kThreads + tid, bc = b < nblk ? b : nblk - 1;
      cc[q] = blk_count[bc]; ll[q] = blk_lastp[bc]; ff[q] = blk_flags[bc];
    }
#pragma unroll
    for (int q = 0; q < kScanPer; ++q) {
      const bool in = base + q * kThreads + tid < nblk;
      int c = in ? cc[q] : 0;
      if (c > rec_cap) { fl |= 0x80000000u; c = rec_cap; }    // bit 31: some unit overflowed its list
      if (in && ll[q] > lp) lp = ll[q];
      if (in) fl |= ff[q];
      s_cnt[q * kThreads + tid] = c;
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
// ---- k_gather: per-unit lists of centre words -> one list in stream order; the burst records stay where k_detect wrote
// them (their list slot travels with the word): round 2 copied them too, 32 MB more of scattered traffic per 2^30-sample
// pass beside a k_detect that is bound by HBM ------------------------------------------------------------------------
// (the tail kernels are bodies with the workgroup's index and the grid size as parameters: k_tail_small runs them all
// in ONE workgroup for small passes)
__device__ __forceinline__ void gather_body(int bid, int nb, const unsigned long long* cands, const int* blk_count,
                                            const int* blk_off, int nblk, int rec_cap, unsigned long long* sorted,
                                            unsigned* sorted_src) {
  // one WAVEFRONT per list: a bulk pass has tens of thousands of short lists (a dozen centres each), and a list costs
  // its two dependent reads whatever its length
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b = bid * kWaves + wave; b < nblk; b += nb * kWaves) {
    int c = blk_count[b];
    if (c > rec_cap) c = rec_cap;
    const int off = blk_off[b];
    const long long src = (long long)b * rec_cap;
    for (int j = lane; j < c; j += 64) {
      sorted[off + j] = cands[src + j];
      sorted_src[off + j] = (unsigned)(src + j);           // where its 32-byte record lies: it is read once, by k_compact
    }
  }
}

// ---- k_order: the first tail kernel of a BULK pass -- long pulses, per-unit counts -> offsets, centre words into stream
// order -- what k_longrun + k_scan + k_gather did in three launches (one of them a single workgroup, 42 us for the 40 k lists
// of a 2^30-sample pass).  Workgroup g owns kOrderLists consecutive lists, one per thread:
//   0. the long pulses k_detect listed for one of ITS lists (usually none): their placeholder word / record is finished by
//      this workgroup before it reads the list (longrun_entry);
//   1. count, last paired centre, flags of its lists; maximum / OR go to the pass's accumulator with ONE atomic per workgroup;
//   2. the number of centres in front of its first list: every workgroup sums those counts itself (<= 160 KB of L2-resident
//      ints, 16-byte loads, all in flight together) -- no second launch, no look-back chain;
//   3. each thread copies the words of its list (a dozen) to their final places, four loads in flight at a time.
// The summary's fields are completed by k_compact (OrderAcc -> Summary), which also hands the summary to the host.
constexpr int kOrderLists = kThreads;
struct OrderAcc { unsigned long long lastp_biased; unsigned flags; unsigned pad_; };
template <int MODE>
__global__ void __launch_bounds__(kThreads) k_order(DetectArgs a, int nblk, unsigned long long* sorted, unsigned* sorted_src,
                                                   Summary* sum, OrderAcc* acc) {
  __shared__ long long s_lp[kWaves];
  __shared__ unsigned s_fl[kWaves];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, bid = (int)blockIdx.x;
  const int rec_cap = a.rec_cap;
  // 0. long pulses of this workgroup's lists (workgroup-uniform loop: every thread sees the same list)
  int n_long = *a.long_count;
  if (n_long > a.long_cap) n_long = a.long_cap;
  for (int e = 0; e < n_long; ++e)
    if (a.longlist[e].blk / kOrderLists == bid) longrun_entry<MODE>(a, e);
  __syncthreads();
  // 1. my list
  const int b = bid * kOrderLists + tid;
  const bool in = b < nblk;
  const int bc = in ? b : nblk - 1;
  int cnt = a.blk_count[bc];
  long long lp = a.blk_lastp[bc];
  unsigned fl = a.blk_flags[bc];
  if (!in) { cnt = 0; lp = kNoIndex; fl = 0u; }
  if (cnt > rec_cap) { fl |= 0x80000000u; cnt = rec_cap; }       // bit 31: some unit overflowed its list
  // 2. centres in front of this workgroup's first list
  int before = 0;
  {
    struct alignas(16) I4 { int x, y, z, w; };
    const int nfront = bid * kOrderLists;                       // a multiple of 4: whole 16-byte groups only
    const I4* c4 = reinterpret_cast<const I4*>(a.blk_count);
    auto clipped = [&](const I4& v) {
      return (v.x < rec_cap ? v.x : rec_cap) + (v.y < rec_cap ? v.y : rec_cap) + (v.z < rec_cap ? v.z : rec_cap) +
             (v.w < rec_cap ? v.w : rec_cap);
    };
    // (eight loads in flight per thread: the last workgroup of a headline pass has forty groups per thread to add, and one
    // L2 round trip each made this loop the longest part of the kernel)
    const int n4 = nfront / 4;
    int i = tid;
    for (; i + 7 * kThreads < n4; i += 8 * kThreads) {
      I4 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = c4[i + q * kThreads];
#pragma unroll
      for (int q = 0; q < 8; ++q) before += clipped(v[q]);
    }
    for (; i < n4; i += kThreads) before += clipped(c4[i]);
  }
  int front = 0, total = 0;
  (void)block_excl_scan(before, &front);                        // (barriers inside) front = sum over the workgroup
  const int run = front + block_excl_scan(cnt, &total);
  // 3. words to their final places
  {
    const long long src = (long long)b * rec_cap;
    constexpr int kAtOnce = 8;                                  // loads in flight per thread (a list of the headline pass: ~12 words)
    for (int j = 0; j < cnt; j += kAtOnce) {
      unsigned long long w[kAtOnce];
#pragma unroll
      for (int q = 0; q < kAtOnce; ++q) w[q] = a.cands[src + (j + q < cnt ? j + q : j)];
#pragma unroll
      for (int q = 0; q < kAtOnce; ++q)
        if (j + q < cnt) {
          sorted[run + j + q] = w[q];
          sorted_src[run + j + q] = (unsigned)(src + j + q);   // where its 32-byte record lies: read once, by k_compact
        }
    }
  }
  // workgroup reductions: max of lastp, OR of flags (wave shuffles, then 4 words), one atomic each
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
    long long L = kNoIndex;
    unsigned F = 0u;
    for (int w = 0; w < kWaves; ++w) { if (s_lp[w] > L) L = s_lp[w]; F |= s_fl[w]; }
    if (L != kNoIndex) atomicMax(&acc->lastp_biased, (unsigned long long)(L + (1ll << 62)));
    if (F) atomicOr(&acc->flags, F);
    if (bid == (int)gridDim.x - 1) sum->n_rec = front + total;
  }
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
template <bool HOSTCOPY = false>
__device__ __forceinline__ void compact_body(int bid, int nb, const unsigned long long* sorted, const Rec* recs, const unsigned* sorted_src,
                                             Summary* sum, const int* seg_count, unsigned fmask, unsigned fwant, int head_n,
                                             Rec* out, int out_cap, int* long_count, unsigned long long* long_lastp,
                                             OrderAcc* acc = nullptr, Summary* host_sum = nullptr, Rec* host_out = nullptr,
                                             int host_cap = 0) {
  __shared__ int s_c[kWaves];
  __shared__ int s_pre[kWaves], s_tot[kWaves];
  const int n = sum->n_rec;
  const int nseg = (n + kThreads - 1) / kThreads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (bid == 0 && host_sum) {
    // Bulk pass: workgroup 0 completes the pass's 48-byte summary ALONE and stores it straight into the caller-visible
    // (pinned, mapped) host copy -- no publishing kernel, no cross-workgroup hand-over (a "last workgroup done" counter needs a
    // device-scope fence per workgroup: an L2 write-back on a part with one L2 per XCD, measured 21 -> 41 us for this kernel).
    // k_order left the maximum / OR of the lists in the accumulator; the number of survivors is the sum of the segment
    // counts every workgroup computes anyway; the last survivor is found by looking at the END of the ordered list.
    int tot = 0;
    for (int j = threadIdx.x; j < nseg; j += kThreads) tot += seg_count[j];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d);
    if (lane == 0) s_tot[wave] = tot;
    __syncthreads();
    tot = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
    __syncthreads();
    long long lastk = kNoIndex;
    if (tot > 0) {
      for (int hi = n; hi > 0 && lastk == kNoIndex; hi -= kThreads) {      // workgroup-uniform loop
        const int i = hi - 1 - (int)threadIdx.x;
        const unsigned long long c = i >= 0 ? sorted[i] : 0ull;
        const bool k = i >= 0 && survives(c, i, fmask, fwant, head_n);
        const unsigned long long m = __ballot(k);
        if (lane == 0) s_c[wave] = m ? __builtin_ctzll(m) : 64;            // smallest lane = largest index
        __syncthreads();
        int w = 0;
        while (w < kWaves && s_c[w] == 64) ++w;
        if (w < kWaves) lastk = cand_p(sorted[hi - 1 - (w * 64 + s_c[w])]);
        __syncthreads();
      }
    }
    if (threadIdx.x == 0) {
      const unsigned F = acc->flags;
      const unsigned long long lb = acc->lastp_biased > *long_lastp ? acc->lastp_biased : *long_lastp;
      Summary f;
      f.n_rec = n; f.n_kept = tot; f.overflow = (int)((F >> 31) & 1u); f.long_count = *long_count; f.flags = F & 0x7FFFFFFFu;
      f.pad_ = 0; f.lastp = lb ? (long long)lb - (1ll << 62) : kNoIndex; f.last_kept_p = lastk;
      *host_sum = f;
      *sum = f;
      acc->flags = 0u; acc->lastp_biased = 0ull;             // empty again for the slot's next pass
    }
  }
  if (bid == 0 && threadIdx.x == 0) {
    *long_count = 0;        // the long-pulse list has been consumed: leave it empty for the slot's next pass
    *long_lastp = 0ull;
    if (nseg == 0 && !host_sum) sum->n_kept = 0;
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
    if (seg == 0 && threadIdx.x == 0 && !host_sum) sum->n_kept = total;
    if (k && off < out_cap) {
      // (the record as two 16-byte VECTOR values: as an aggregate -- one Rec or two RecHalf -- with a second store behind the
      // parity loop the compiler parked half of it in 4 KB of LDS, which no longer fits beside five k_detect workgroups:
      // tests/test_abi.py)
      typedef unsigned long long u64x2 __attribute__((vector_size(16)));      // (clang and g++: the emulator build)
      const u64x2* src = reinterpret_cast<const u64x2*>(recs + sorted_src[i]);
      const u64x2 h0 = src[0];
      u64x2 h1 = src[1];
      unsigned fl = cand_flags(c) & (kKept | kHead);
      if ((unsigned)(h1[1] >> 48) & kDemod) fl |= parity_flags_of(h1[0], h1[1]);
      h1[1] |= (unsigned long long)fl << 48;
      u64x2* dst = reinterpret_cast<u64x2*>(out + off);
      dst[0] = h0; dst[1] = h1;
      // mid-size pass: the first host_cap records ALSO go straight into the pinned result buffer (adsb_hip.hip: enqueue) --
      // when the pass delivers no more than that, adsb_wait has them without a copy of its own
      if (HOSTCOPY && off < host_cap) {
        u64x2* hd = reinterpret_cast<u64x2*>(host_out + off);
        hd[0] = h0; hd[1] = h1;
      }
      if (off == total - 1 && !host_sum) sum->last_kept_p = cand_p(c);
    }
    __syncthreads();
  }
}
__global__ void __launch_bounds__(kThreads) k_compact(const unsigned long long* sorted, const Rec* recs, const unsigned* sorted_src, Summary* sum,
                                                      const int* seg_count, unsigned fmask, unsigned fwant, int head_n,
                                                      Rec* out, int out_cap, int* long_count,
                                                      unsigned long long* long_lastp, OrderAcc* acc, Summary* host_sum,
                                                      Rec* host_out, int host_cap) {
  compact_body<true>((int)blockIdx.x, (int)gridDim.x, sorted, recs, sorted_src, sum, seg_count, fmask, fwant, head_n, out, out_cap,
                     long_count, long_lastp, acc, host_sum, host_out, host_cap);
}

// ---- k_tail_small: the whole tail of a SMALL pass (a GNU Radio work() call: a few lists, a few hundred centres at most)
// in one workgroup and one launch -- long pulses, scan, gather, gate, count, compact, publish, with workgroup barriers
// where the multi-kernel chain has kernel boundaries.  Such a pass is bound by the number of GPU operations, not by their size:
// seven launches fewer.  (Writes of one phase are read by other wavefronts of the SAME workgroup in the next: the
// workgroup-scope ordering of __syncthreads() is what that needs.)
struct TailArgs {
  const unsigned long long* cands; const Rec* recs; const int* blk_count; const long long* blk_lastp; const unsigned* blk_flags;
  int* blk_off; int nblk, rec_cap; int* long_count; unsigned long long* long_lastp;
  unsigned long long* sorted; unsigned* sorted_src; int* seg_count; Summary* sum; Summary* host_sum; Rec* out; int out_cap;
  int gate_on, head_n; long long gate, gate_long, prev_eob;
  int seq;               // this pass's number: stored LAST, into host_sum->pad_ -- the host polls it (adsb_hip.hip: finish)
};
// End of a one-workgroup pass: the summary into the caller-visible (pinned, mapped) host copy, then -- behind a system-scope
// fence, so that the records compact_body stored straight into pinned host memory and the summary's fields are visible
// first -- the pass number.  The host does not wait for the kernel's completion signal (end-of-kernel cache maintenance,
// signal, the runtime's wake-up: several microseconds of a call that costs 25): it polls that word.
// The records were stored by EVERY wavefront of the workgroup: each thread fences its own stores to system scope (the
// fence waits for the thread's outstanding stores to be acknowledged and writes them through) BEFORE the workgroup
// barrier -- a workgroup barrier alone orders them at workgroup scope only, and thread 0's fence below would cover
// thread 0's stores, not the other wavefronts'.
__device__ __forceinline__ void publish_small(const TailArgs& t) {
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    Summary f = *t.sum;
    f.pad_ = 0;
    Summary* h = t.host_sum;
    h->n_rec = f.n_rec; h->n_kept = f.n_kept; h->overflow = f.overflow; h->long_count = f.long_count; h->flags = f.flags;
    h->lastp = f.lastp; h->last_kept_p = f.last_kept_p;
    __threadfence_system();
    *reinterpret_cast<volatile int*>(&h->pad_) = t.seq;
  }
}
template <int MODE>
__global__ void __launch_bounds__(kThreads) k_tail_small(DetectArgs a, TailArgs t) {
  longrun_body<MODE>(0, 1, a);                                // pulses longer than k_detect's LDS window (usually none)
  __syncthreads();
  __shared__ int s_scan_cnt[kScanRound];
  scan_body<kScanRound>(t.blk_count, t.blk_lastp, t.blk_flags, t.nblk, t.rec_cap, t.long_count, t.long_lastp, t.blk_off, t.sum, s_scan_cnt);
  __syncthreads();
  gather_body(0, 1, t.cands, t.blk_count, t.blk_off, t.nblk, t.rec_cap, t.sorted, t.sorted_src);
  __syncthreads();
  unsigned fmask = 0u, fwant = 0u;
  if (t.gate_on) {
    resolve_body(0, 1, t.sorted, t.sum, t.gate, t.gate_long, t.prev_eob);
    fmask = kKept; fwant = kKept;
    __syncthreads();
  }
  count_body(0, 1, t.sorted, t.sum, fmask, fwant, t.head_n, t.seg_count);
  __syncthreads();
  compact_body(0, 1, t.sorted, t.recs, t.sorted_src, t.sum, t.seg_count, fmask, fwant, t.head_n, t.out, t.out_cap, t.long_count,
               t.long_lastp);
  __syncthreads();
  publish_small(t);
}

// ---- k_pass_small: a WHOLE small pass -- the one pass over the samples and its tail -- in one workgroup and ONE launch.
// A GNU Radio work() call of a few thousand samples is four units at most (one workgroup of k_detect) and a handful of
// centres: its cost is the number of GPU operations, and this is one instead of two.  |IQ|^2 float input only (the
// framer's input type, adsb_framer_work); the workgroup barrier between the two halves also makes the lists the four
// wavefronts wrote to global memory visible to each other.
template <int HALF>
__global__ void __launch_bounds__(kThreads) k_pass_small(DetectArgs a, TailArgs t) {
  detect_body<1, HALF>(a, 0);
  __syncthreads();
  longrun_body<1>(0, 1, a);
  __syncthreads();
  __shared__ int s_scan_cnt[kScanRound];
  scan_body<kScanRound>(t.blk_count, t.blk_lastp, t.blk_flags, t.nblk, t.rec_cap, t.long_count, t.long_lastp, t.blk_off, t.sum, s_scan_cnt);
  __syncthreads();
  gather_body(0, 1, t.cands, t.blk_count, t.blk_off, t.nblk, t.rec_cap, t.sorted, t.sorted_src);
  __syncthreads();
  unsigned fmask = 0u, fwant = 0u;
  if (t.gate_on) {
    resolve_body(0, 1, t.sorted, t.sum, t.gate, t.gate_long, t.prev_eob);
    fmask = kKept; fwant = kKept;
    __syncthreads();
  }
  count_body(0, 1, t.sorted, t.sum, fmask, fwant, t.head_n, t.seg_count);
  __syncthreads();
  compact_body(0, 1, t.sorted, t.recs, t.sorted_src, t.sum, t.seg_count, fmask, fwant, t.head_n, t.out, t.out_cap, t.long_count,
               t.long_lastp);
  __syncthreads();
  publish_small(t);
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
