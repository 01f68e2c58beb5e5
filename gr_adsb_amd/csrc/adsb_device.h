// adsb_device.h -- CDNA4 (gfx950) device code for the ADS-B front end: |IQ|^2 -> threshold/edge ->
// pulse centre -> 16-chip preamble test -> (peak, noise median, 112-bit PPM slice) per matched centre,
// then ordering, the re-trigger gate and compaction.  Written for 64-wide wavefronts: the threshold
// bitmask of a 64-sample word IS one __ballot, rises are found with scalar mask algebra, and per-burst
// work (median-of-100, 224 PPM gathers) is done by one whole wavefront per burst.
//
// Behaviour restated from the reference (citations to /root/reference/python/adsb/):
//   threshold / edges / pairing   framer.py:83-113      preamble test     framer.py:137-147
//   SNR inputs (peak, median)     framer.py:156-159     re-trigger gate   framer.py:121-123,165
//   PPM slice                     demod.py:75-95        confidence ratio  demod.py:101
//
// This header contains device code only and includes nothing: the product translation unit
// (adsb_hip.hip) includes <hip/hip_runtime.h> first; tests/sim/sim_driver.cpp includes the test-only
// SIMT emulator first so the very same kernels can be checked on a machine without a GPU.
#pragma once

namespace adsb {

constexpr int kThreads = 256;            // 4 wavefronts per workgroup
constexpr int kWaves = kThreads / 64;
constexpr int kTile = 4096;              // samples owned per tile iteration (64 ballot words)
constexpr int kFwd = 256;                // forward halo kept in LDS behind every tile
constexpr int kWin = kTile + kFwd;
constexpr int kWords = kWin / 64;        // 68
constexpr int kOwnWords = kTile / 64;    // 64 == one wavefront of word owners
constexpr unsigned kTemplate = 0x285u;   // chips 0,2,7,9 high (framer.py:50)
constexpr int kNoise = 100;              // framer.py:31
constexpr long long kNoIndex = -(1ll << 62);
static_assert(kFwd == kThreads, "the halo shift moves one float per thread");

enum RecFlags : unsigned {
  kDemod = 1u,     // eob inside the demod input: bits valid (demod.py:82)
  kKept = 2u,      // passed the re-trigger gate (framer.py:121)
  kNoMatch = 4u,   // placeholder whose long pulse did not match the preamble
  kPending = 8u,   // placeholder waiting for k_longrun
};

// 32-byte burst record: w0 = stream offset (int64); w1 = peak | median<<32 (float bits);
// w2 = bits 0..63 as bytes 0..7 (first bit = MSB of byte 0); w3 = bytes 8..13 | flags<<48.
struct Rec { unsigned long long w[4]; };

struct LongRise { long long rise; int blk; int slot; };

struct Summary {
  int n_rec;        // records (matched centres + placeholders) in sorted order
  int n_kept;       // after gate
  int overflow;     // some block exceeded rec_cap
  int long_count;   // entries in the long-rise list
  unsigned flags;   // bit0 any rise, bit1 any fall, bit2 halo exceeded
  int pad_;
  long long lastp;  // largest paired pulse centre (local index) or kNoIndex
  long long last_kept_p;  // centre (local) of the last kept record or kNoIndex
};

struct DetectArgs {
  const void* data;      // float2[n] (MODE 0, complex64 IQ) or float[n] (MODE 1, |IQ|^2)
  long long n;           // samples present; x(i) = 0 for i < 0 or i >= n
  long long in0_base;    // local index of the framer's in0[0] (-(8*sps-1) on a fresh stream)
  long long scan_lo;     // rises (and falls) are owned / counted in [scan_lo, scan_hi)
  long long scan_hi;
  long long fall_hi;     // a pulse needs its fall at an index < fall_hi
  long long dem_hi;      // PPM-slice iff p + 119*sps + sps/2 < dem_hi
  long long origin;      // stream offset of local index 0
  long long chunk;       // samples per workgroup (multiple of kTile)
  float thr;
  float prev_in0;        // value compared for the sample before in0[0] (framer.py:84)
  int sps;
  int end_is_call_end;   // 1: pulse still high at fall_hi is discarded (framer.py:102-108); 0: halo error
  int rec_cap;           // records per workgroup
  int long_cap;
  Rec* recs;             // [grid][rec_cap]
  int* blk_count;        // [grid]
  long long* blk_lastp;  // [grid]
  unsigned* blk_flags;   // [grid]
  LongRise* longlist;
  int* long_count;
  unsigned long long* long_lastp;   // biased: centre + 2^62, 0 = none
};

__device__ __forceinline__ float mag2f(float re, float im) {
  // two rounded products, one rounded add: never contracted into an FMA
  return __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
}

template <int MODE>
__device__ __forceinline__ float xg(const void* data, long long n, long long i) {
  if (i < 0 || i >= n) return 0.0f;
  if (MODE == 0) {
    float2 q = reinterpret_cast<const float2*>(data)[i];
    return mag2f(q.x, q.y);
  }
  return reinterpret_cast<const float*>(data)[i];
}

template <int MODE>
__device__ __forceinline__ bool above_at(const DetectArgs& a, long long i) {
  if (i >= 0) return xg<MODE>(a.data, a.n, i) >= a.thr;
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

// ---- window accessor: LDS when the sample is inside the tile window, global otherwise -------------
template <int MODE>
struct WinAcc {
  const float* sx;
  long long t0;
  const void* data;
  long long n;
  __device__ __forceinline__ float get(long long i) const {
    long long li = i - t0;
    if (li >= 0 && li < kWin) return sx[li];
    return xg<MODE>(data, n, i);
  }
};
template <int MODE>
struct GlobAcc {
  const void* data;
  long long n;
  __device__ __forceinline__ float get(long long i) const { return xg<MODE>(data, n, i); }
};

// ---- one wavefront builds one burst record ---------------------------------------------------------
// peak = x[p]; median of x[max(in0_base, p-100) : p] with np.median semantics (framer.py:156-159);
// 112 hard bits b1 > b0 at stride sps (demod.py:87-95).  All 64 lanes must be active.
template <class Acc>
__device__ void emit_record(const Acc& acc, const DetectArgs& a, long long p, Rec* out, int lane) {
  const float peak = acc.get(p);
  long long wlo = p - kNoise;
  if (wlo < a.in0_base) wlo = a.in0_base;
  const int nwin = (int)(p - wlo);
  const bool val0 = lane < nwin, val1 = lane + 64 < nwin;
  const float v0 = val0 ? acc.get(wlo + lane) : 0.0f;
  const float v1 = val1 ? acc.get(wlo + lane + 64) : 0.0f;
  const unsigned long long nanm = __ballot((val0 && v0 != v0) || (val1 && v1 != v1));
  int r0 = 0, r1 = 0;
  for (int j = 0; j < nwin; ++j) {
    const float e = __shfl(j < 64 ? v0 : v1, j & 63);
    r0 += (e < v0 || (e == v0 && j < lane)) ? 1 : 0;
    r1 += (e < v1 || (e == v1 && j < lane + 64)) ? 1 : 0;
  }
  float med;
  {
    const int khi = nwin >> 1;               // upper middle (the middle for odd n)
    const int klo = (nwin & 1) ? khi : khi - 1;
    const unsigned long long h0 = __ballot(val0 && r0 == khi), h1 = __ballot(val1 && r1 == khi);
    const unsigned long long l0 = __ballot(val0 && r0 == klo), l1 = __ballot(val1 && r1 == klo);
    const float sa = __shfl(v0, h0 ? __builtin_ctzll(h0) : 0);
    const float sb = __shfl(v1, h1 ? __builtin_ctzll(h1) : 0);
    const float sc = __shfl(v0, l0 ? __builtin_ctzll(l0) : 0);
    const float sd = __shfl(v1, l1 ? __builtin_ctzll(l1) : 0);
    const float hi = h0 ? sa : sb;
    const float lo = l0 ? sc : sd;
    if (nwin == 0) med = __builtin_bit_cast(float, 0xFFC00000u);   // np.median([]) == 0/0: default NaN, sign set
    else if (nanm) med = __builtin_bit_cast(float, 0x7FC00000u);    // a NaN in the window propagates
    else if (nwin & 1) med = hi;
    else med = __fmul_rn(__fadd_rn(lo, hi), 0.5f);     // f32(a+b)/2
  }
  const int sps = a.sps, half = sps >> 1;
  const bool dem = p + 119ll * sps + half < a.dem_hi;   // demod.py:76,82 (sps even)
  bool b0 = false, b1 = false;
  if (dem) {
    const long long s0 = p + 8ll * sps + (long long)lane * sps;           // demod.py:75,87
    b0 = acc.get(s0) > acc.get(s0 + half);                                // demod.py:91,95
    if (lane < 48) {
      const long long s1 = s0 + 64ll * sps;
      b1 = acc.get(s1) > acc.get(s1 + half);
    }
  }
  const unsigned long long ma = __ballot(b0), mb = __ballot(b1);
  if (lane == 0) {
    const unsigned long long ra = __builtin_bswap64(__brevll(ma));
    const unsigned long long rb = __builtin_bswap64(__brevll(mb)) & 0xFFFFFFFFFFFFull;
    const unsigned flags = dem ? kDemod : 0u;
    Rec r;
    r.w[0] = (unsigned long long)(a.origin + p);
    r.w[1] = (unsigned long long)__builtin_bit_cast(unsigned, peak) |
             ((unsigned long long)__builtin_bit_cast(unsigned, med) << 32);
    r.w[2] = ra;
    r.w[3] = rb | ((unsigned long long)flags << 48);
    *out = r;
  }
}

// ---- global -> register -> LDS tile staging ---------------------------------------------------------
// A tile body (COUNT samples) is fetched with 16-byte loads into registers (span_issue) and turned into
// |IQ|^2 floats in LDS later (span_commit), so that the fetch of tile k+1 is in flight while tile k is
// being processed.  The fast path (whole span inside the buffer) has no per-load branches: all loads of a
// thread are issued back to back and waited for once.
template <int MODE, int COUNT>
struct Span {
  static constexpr int PER = (MODE == 0) ? 2 : 4;             // samples per float4
  static constexpr int NV = COUNT / PER;
  static constexpr int ITER = (NV + kThreads - 1) / kThreads;
  float4 q[ITER];
};

template <int MODE, int COUNT>
__device__ __forceinline__ void span_issue(Span<MODE, COUNT>& sp, const DetectArgs& a, long long src, int tid) {
  using S = Span<MODE, COUNT>;
  const float4* base = reinterpret_cast<const float4*>(a.data);
  if (src + COUNT <= a.n) {                                    // wave-uniform: no bounds checks needed
    const float4* p = base + src / S::PER + tid;
#pragma unroll
    for (int k = 0; k < S::ITER; ++k) {
      if (S::NV % kThreads == 0 || tid + k * kThreads < S::NV) sp.q[k] = p[k * kThreads];
    }
  } else {                                                     // ragged end of the buffer (at most one tile per call)
    const float* fb = reinterpret_cast<const float*>(a.data);
#pragma unroll
    for (int k = 0; k < S::ITER; ++k) {
      const long long i = src + (long long)S::PER * (tid + k * kThreads);
      float e[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (MODE == 0) {
        if (i < a.n) { e[0] = fb[2 * i]; e[1] = fb[2 * i + 1]; }
        if (i + 1 < a.n) { e[2] = fb[2 * i + 2]; e[3] = fb[2 * i + 3]; }
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) if (i + c < a.n) e[c] = fb[i + c];
      }
      sp.q[k].x = e[0]; sp.q[k].y = e[1]; sp.q[k].z = e[2]; sp.q[k].w = e[3];
    }
  }
}

template <int MODE, int COUNT>
__device__ __forceinline__ void span_commit(const Span<MODE, COUNT>& sp, float* sx, int dst, int tid) {
  using S = Span<MODE, COUNT>;
#pragma unroll
  for (int k = 0; k < S::ITER; ++k) {
    const int v = tid + k * kThreads;
    if (S::NV % kThreads == 0 || v < S::NV) {
      if (MODE == 0) {
        float2 m;
        m.x = mag2f(sp.q[k].x, sp.q[k].y);
        m.y = mag2f(sp.q[k].z, sp.q[k].w);
        *reinterpret_cast<float2*>(&sx[dst + 2 * v]) = m;
      } else {
        *reinterpret_cast<float4*>(&sx[dst + 4 * v]) = sp.q[k];
      }
    }
  }
}

// ---- k_detect: the streaming kernel ----------------------------------------------------------------
// One workgroup walks a contiguous chunk of the stream tile by tile with a sliding LDS window of
// kTile + kFwd |IQ|^2 floats (the forward halo of one tile is the head of the next, so every sample
// is fetched from HBM once).  Records are appended to the workgroup's own slice of `recs` in stream
// order; ordering across workgroups is by workgroup index (k_scan / k_gather).
template <int MODE>
__global__ void __launch_bounds__(kThreads) k_detect(DetectArgs a) {
  __shared__ __attribute__((aligned(16))) float s_x[kWin];
  __shared__ unsigned long long s_mask[kWords];
  __shared__ unsigned short s_list[kTile / 2];
  __shared__ int s_nrise, s_ncand, s_nrec, s_pred, s_lastp, s_lastp2;
  __shared__ unsigned s_flags;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long c0 = (long long)blockIdx.x * a.chunk;
  long long c1 = c0 + a.chunk;
  if (c1 > a.scan_hi) c1 = a.scan_hi;
  const int half = a.sps >> 1;
  long long lastp_g = kNoIndex;

  if (tid == 0) { s_nrec = 0; s_flags = 0u; }
  __syncthreads();

  // virtual rise in the zero history in front of a fresh stream: only possible when 0 >= thr
  if (blockIdx.x == 0 && wave == 0 && a.scan_lo < 0) {
    const bool vr = (0.0f >= a.thr) && !(a.prev_in0 >= a.thr);
    if (vr) {
      if (lane == 0) {
        s_flags |= 1u;
        const int slot = s_nrec;
        if (slot < a.rec_cap) {
          Rec r; r.w[0] = (unsigned long long)a.scan_lo; r.w[1] = 0; r.w[2] = 0;
          r.w[3] = (unsigned long long)(kPending | kNoMatch) << 48;
          a.recs[(long long)blockIdx.x * a.rec_cap + slot] = r;
          const int li = atomicAdd(a.long_count, 1);
          if (li < a.long_cap) { LongRise e; e.rise = a.scan_lo; e.blk = 0; e.slot = slot; a.longlist[li] = e; }
        }
        s_nrec = slot + 1;
      }
    }
  }
  __syncthreads();

  // prologue: forward-halo head of the first tile straight into LDS, body of the first tile into registers
  Span<MODE, kTile> body;
  if (c0 < c1) {
    Span<MODE, kFwd> head;
    span_issue<MODE, kFwd>(head, a, c0, tid);
    span_issue<MODE, kTile>(body, a, c0 + kFwd, tid);
    span_commit<MODE, kFwd>(head, s_x, 0, tid);
  }
  for (long long t0 = c0; t0 < c1; t0 += kTile) {
    // -- window [t0, t0+kWin): sx[0..kFwd) already holds the head; commit the body, then start fetching
    //    the next tile's body so that it is in flight while this tile is processed
    span_commit<MODE, kTile>(body, s_x, kFwd, tid);
    if (t0 + kTile < c1) span_issue<MODE, kTile>(body, a, t0 + kTile + kFwd, tid);
    if (tid == 0) { s_pred = above_at<MODE>(a, t0 - 1) ? 1 : 0; s_lastp = -1; s_lastp2 = -1; }
    __syncthreads();

    // -- B1: one ballot per 64-sample word (framer.py:83-84)
    for (int w = wave; w < kWords; w += kWaves) {
      const float v = s_x[w * 64 + lane];
      const unsigned long long m = __ballot(v >= a.thr);
      if (lane == 0) s_mask[w] = m;
    }
    __syncthreads();

    // -- B2: rises / falls by mask algebra, ordered rise list (framer.py:91-93)
    if (wave == 0) {
      const unsigned long long M = s_mask[lane];
      const unsigned long long pb = (lane > 0) ? (s_mask[lane - 1] >> 63) : (unsigned long long)s_pred;
      const unsigned long long sh = (M << 1) | pb;
      const long long wbase = t0 + 64ll * lane;
      const unsigned long long own = bit_range(a.scan_lo - wbase, a.scan_hi - wbase);
      unsigned long long R = M & ~sh & own;
      const unsigned long long Fm = ~M & sh & own;
      const unsigned long long anyr = __ballot(R != 0ull), anyf = __ballot(Fm != 0ull);
      const int cnt = __popcll(R);
      int incl = cnt;
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, (unsigned)d);
        if (lane >= d) incl += t;
      }
      int pos = incl - cnt;
      const int total = __shfl(incl, 63);
      while (R) {
        const int b = __builtin_ctzll(R);
        R &= R - 1ull;
        s_list[pos++] = (unsigned short)(64 * lane + b);
      }
      if (lane == 0) {
        s_nrise = total;
        s_flags |= (anyr ? 1u : 0u) | (anyf ? 2u : 0u);
      }
    }
    __syncthreads();

    // -- B3: per rise: fall, centre, 16-chip test (framer.py:113,137-147)
    const int nr = s_nrise;
    for (int i = tid; i < nr; i += kThreads) {
      const int r = s_list[i];
      int w = r >> 6;
      const int b = r & 63;
      unsigned long long inv = ~s_mask[w];
      inv = (b == 63) ? 0ull : (inv & (~0ull << (b + 1)));
      while (inv == 0ull && ++w < kWords) inv = ~s_mask[w];
      unsigned short res = 0;
      if (inv == 0ull) {
        if (t0 + kWin < a.fall_hi) res = 0xFFFFu;        // pulse longer than the window: k_longrun
        else if (!a.end_is_call_end) atomicOr(&s_flags, 4u);
      } else {
        const int f = w * 64 + __builtin_ctzll(inv);
        if (t0 + f < a.fall_hi) {
          const int p = (r + f) >> 1;                    // framer.py:113
          if (i == nr - 1) s_lastp = p;                  // centres increase with i; only the last rise
          else if (i == nr - 2) s_lastp2 = p;            // of a tile can be left without a fall
          const float hp = __fmul_rn(s_x[p], 0.5f);      // in0[pulse_idx]/2, exact
          unsigned chips = 0;
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int idx = p + k * half;
            const float v = (idx < kWin) ? s_x[idx] : xg<MODE>(a.data, a.n, t0 + idx);
            chips |= (v > hp ? 1u : 0u) << k;
          }
          if (chips == kTemplate) res = (unsigned short)(0x8000u | (unsigned)p);
        } else if (!a.end_is_call_end) {
          atomicOr(&s_flags, 4u);
        }
      }
      s_list[i] = res;
    }
    __syncthreads();

    // -- B4: ordered in-place compaction of the matched centres
    if (wave == 0) {
      int nc = 0;
      for (int base = 0; base < nr; base += 64) {
        const int i = base + lane;
        const unsigned short e = (i < nr) ? s_list[i] : (unsigned short)0;
        const unsigned long long mb = __ballot(e != 0);
        const int pos = nc + __popcll(mb & ((1ull << lane) - 1ull));
        if (e) s_list[pos] = e;
        nc += __popcll(mb);
      }
      if (lane == 0) s_ncand = nc;
    }
    __syncthreads();

    // -- C: one wavefront per matched centre builds the record
    const int nc = s_ncand;
    const int rec_base = s_nrec;
    WinAcc<MODE> acc{s_x, t0, a.data, a.n};
    for (int m = wave; m < nc; m += kWaves) {
      const unsigned short e = s_list[m];
      const int slot = rec_base + m;
      if (slot < a.rec_cap) {
        Rec* out = a.recs + (long long)blockIdx.x * a.rec_cap + slot;
        if (e == 0xFFFFu) {
          if (lane == 0) {
            // the long pulse is the last rise of its tile; its rise index is the last set bit's
            // successor search start: recover it from the masks (last rise in the owned words)
            long long rg = kNoIndex;
            for (int w2 = kOwnWords - 1; w2 >= 0 && rg == kNoIndex; --w2) {
              const unsigned long long M = s_mask[w2];
              const unsigned long long pb = (w2 > 0) ? (s_mask[w2 - 1] >> 63) : (unsigned long long)s_pred;
              const long long wb = t0 + 64ll * w2;
              const unsigned long long R = M & ~((M << 1) | pb) & bit_range(a.scan_lo - wb, a.scan_hi - wb);
              if (R) rg = wb + (63 - __builtin_clzll(R));
            }
            Rec r; r.w[0] = (unsigned long long)rg; r.w[1] = 0; r.w[2] = 0;
            r.w[3] = (unsigned long long)(kPending | kNoMatch) << 48;
            *out = r;
            const int li = atomicAdd(a.long_count, 1);
            if (li < a.long_cap) { LongRise le; le.rise = rg; le.blk = (int)blockIdx.x; le.slot = slot; a.longlist[li] = le; }
          }
        } else {
          emit_record(acc, a, t0 + (long long)(e & 0x7FFFu), out, lane);
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      s_nrec = rec_base + nc;
      const int lp = s_lastp >= 0 ? s_lastp : s_lastp2;
      if (lp >= 0) lastp_g = t0 + lp;
    }
    // the forward halo of this tile is the head of the next one (kFwd == kThreads)
    const float keep = s_x[kTile + tid];
    __syncthreads();
    s_x[tid] = keep;
  }
  __syncthreads();
  if (tid == 0) {
    a.blk_count[blockIdx.x] = s_nrec;
    a.blk_lastp[blockIdx.x] = lastp_g;
    a.blk_flags[blockIdx.x] = s_flags;
  }
}

// ---- k_longrun: pulses whose run leaves the LDS window (or starts in the zero history) -------------
// One workgroup per entry scans forward cooperatively for the fall, then wave 0 finishes the pulse
// with global-memory taps.  Rare (CW / overload); correctness path, not a fast path.
template <int MODE>
__global__ void __launch_bounds__(kThreads) k_longrun(DetectArgs a, int n_entries) {
  __shared__ unsigned long long s_found;   // fall index relative to rise+1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = blockIdx.x; e < n_entries; e += gridDim.x) {
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
    Rec* out = a.recs + (long long)le.blk * a.rec_cap + le.slot;
    if (wave == 0) {
      if (f < limit) {
        const long long p = (le.rise + f) >> 1;            // floor, also for negative indices
        if (lane == 0) atomicMax(a.long_lastp, (unsigned long long)(p + (1ll << 62)));
        const float hp = __fmul_rn(xg<MODE>(a.data, a.n, p), 0.5f);
        const float v = (lane < 16) ? xg<MODE>(a.data, a.n, p + (long long)lane * (a.sps >> 1)) : 0.0f;
        const unsigned long long cm = __ballot(lane < 16 && v > hp);
        if ((unsigned)cm == kTemplate) {
          GlobAcc<MODE> acc{a.data, a.n};
          emit_record(acc, a, p, out, lane);
        } else if (lane == 0) {
          out->w[3] = (unsigned long long)kNoMatch << 48;
        }
      } else {
        if (lane == 0) {
          out->w[3] = (unsigned long long)kNoMatch << 48;
          if (!a.end_is_call_end) atomicOr(a.blk_flags + le.blk, 4u);
        }
      }
    }
    __syncthreads();
  }
}

// ---- k_scan: per-workgroup counts -> offsets, totals (single workgroup) ----------------------------
__global__ void __launch_bounds__(kThreads) k_scan(const int* blk_count, const long long* blk_lastp,
                                                   const unsigned* blk_flags, int nblk, int rec_cap,
                                                   const int* long_count, const unsigned long long* long_lastp,
                                                   int* blk_off, Summary* sum) {
  __shared__ int s_part[kThreads];
  __shared__ long long s_lp[kThreads];
  __shared__ unsigned s_fl[kThreads];
  __shared__ int s_ovf[kThreads];
  const int tid = threadIdx.x;
  const int per = (nblk + kThreads - 1) / kThreads;
  const int b0 = tid * per;
  int b1 = b0 + per; if (b1 > nblk) b1 = nblk;
  int acc = 0, ovf = 0; long long lp = kNoIndex; unsigned fl = 0;
  for (int b = b0; b < b1; ++b) {
    int c = blk_count[b];
    if (c > rec_cap) { ovf = 1; c = rec_cap; }
    acc += c;
    const long long l = blk_lastp[b];
    if (l > lp) lp = l;
    fl |= blk_flags[b];
  }
  s_part[tid] = acc; s_lp[tid] = lp; s_fl[tid] = fl; s_ovf[tid] = ovf;
  __syncthreads();
  if (tid == 0) {
    int run = 0; long long L = (*long_lastp == 0ull) ? kNoIndex : (long long)(*long_lastp) - (1ll << 62); unsigned F = 0; int O = 0;
    for (int t = 0; t < kThreads; ++t) {
      const int c = s_part[t]; s_part[t] = run; run += c;
      if (s_lp[t] > L) L = s_lp[t];
      F |= s_fl[t]; O |= s_ovf[t];
    }
    sum->n_rec = run; sum->overflow = O; sum->flags = F; sum->lastp = L;
    sum->long_count = *long_count; sum->n_kept = 0; sum->last_kept_p = kNoIndex;
  }
  __syncthreads();
  int run = s_part[tid];
  for (int b = b0; b < b1; ++b) {
    int c = blk_count[b];
    if (c > rec_cap) c = rec_cap;
    blk_off[b] = run;
    run += c;
  }
}

// ---- k_gather: per-workgroup slices -> one list in stream order ------------------------------------
__global__ void __launch_bounds__(kThreads) k_gather(const Rec* recs, const int* blk_count, const int* blk_off,
                                                     int nblk, int rec_cap, Rec* sorted) {
  for (int b = blockIdx.x; b < nblk; b += gridDim.x) {
    int c = blk_count[b];
    if (c > rec_cap) c = rec_cap;
    const int off = blk_off[b];
    for (int j = threadIdx.x; j < c; j += kThreads) sorted[off + j] = recs[(long long)b * rec_cap + j];
  }
}

// ---- k_resolve: the re-trigger gate (framer.py:121-123,165) as parallel chain walks -----------------
// Sequentially: accept a matched centre p iff p > eob, then eob = p + 63*sps.  A centre more than
// 63*sps past its predecessor is accepted whatever happened before it, so it starts an independent
// chain; each chain head walks its own (short) chain.  Records flagged kNoMatch are skipped.
__device__ __forceinline__ long long rec_p(const Rec& r) { return (long long)r.w[0]; }
__device__ __forceinline__ unsigned rec_flags(const Rec& r) { return (unsigned)(r.w[3] >> 48); }

__global__ void __launch_bounds__(kThreads) k_resolve(Rec* sorted, const Summary* sum, long long gate,
                                                      long long prev_eob_stream, int* seg_count) {
  // gate = 63*sps; prev_eob_stream = carried eob expressed as a stream offset (or very negative)
  __shared__ int s_cnt[kWaves];
  const int n = sum->n_rec;
  const int nseg = (n + kThreads - 1) / kThreads;
  for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    const int i = seg * kThreads + threadIdx.x;
    bool head = false;
    if (i < n && !(rec_flags(sorted[i]) & kNoMatch)) {
      const long long p = rec_p(sorted[i]);
      int j = i - 1;
      while (j >= 0 && (rec_flags(sorted[j]) & kNoMatch)) --j;
      if (j < 0) head = true;                         // first real centre: walks from the carried eob
      else head = (p - rec_p(sorted[j]) > gate) && (p > prev_eob_stream);
      if (head) {
        long long eob = (j < 0) ? prev_eob_stream : (p - 1);   // a head with a predecessor is always accepted
        int k = i;
        while (k < n) {
          const unsigned fl = rec_flags(sorted[k]);
          if (!(fl & kNoMatch)) {
            const long long pk = rec_p(sorted[k]);
            if (k != i) {
              // stop at the next head: it owns the rest
              int jj = k - 1;
              while (jj >= 0 && (rec_flags(sorted[jj]) & kNoMatch)) --jj;
              if (jj >= 0 && pk - rec_p(sorted[jj]) > gate && pk > prev_eob_stream) break;
            }
            if (pk > eob) {
              sorted[k].w[3] |= (unsigned long long)kKept << 48;
              eob = pk + gate;
            }
          }
          ++k;
        }
      }
    }
  }
  (void)s_cnt; (void)seg_count;
}

// ---- k_count / k_compact: kept records -> dense output ---------------------------------------------
__global__ void __launch_bounds__(kThreads) k_count(const Rec* sorted, const Summary* sum, int* seg_count) {
  __shared__ int s_c[kWaves];
  const int n = sum->n_rec;
  const int nseg = (n + kThreads - 1) / kThreads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    const int i = seg * kThreads + threadIdx.x;
    const bool kept = i < n && (rec_flags(sorted[i]) & kKept);
    const unsigned long long m = __ballot(kept);
    if (lane == 0) s_c[wave] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) seg_count[seg] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
    __syncthreads();
  }
}

__global__ void __launch_bounds__(kThreads) k_scan2(int* seg_count, Summary* sum, const Rec* sorted) {
  // exclusive scan of seg_count in place (single workgroup) + totals
  __shared__ int s_part[kThreads];
  const int n = sum->n_rec;
  const int nseg = (n + kThreads - 1) / kThreads;
  const int tid = threadIdx.x;
  const int per = (nseg + kThreads - 1) / kThreads;
  const int b0 = tid * per;
  int b1 = b0 + per; if (b1 > nseg) b1 = nseg;
  int acc = 0;
  for (int b = b0; b < b1; ++b) acc += seg_count[b];
  s_part[tid] = acc;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int t = 0; t < kThreads; ++t) { const int c = s_part[t]; s_part[t] = run; run += c; }
    sum->n_kept = run;
    long long lk = kNoIndex;
    for (int i = n - 1; i >= 0; --i) {
      if (rec_flags(sorted[i]) & kKept) { lk = rec_p(sorted[i]); break; }
    }
    sum->last_kept_p = lk;
  }
  __syncthreads();
  int run = s_part[tid];
  for (int b = b0; b < b1; ++b) { const int c = seg_count[b]; seg_count[b] = run; run += c; }
}

__global__ void __launch_bounds__(kThreads) k_compact(const Rec* sorted, const Summary* sum, const int* seg_off,
                                                      Rec* out, int out_cap) {
  __shared__ int s_c[kWaves];
  const int n = sum->n_rec;
  const int nseg = (n + kThreads - 1) / kThreads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    const int i = seg * kThreads + threadIdx.x;
    Rec r;
    r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0;
    if (i < n) r = sorted[i];
    const bool kept = i < n && (rec_flags(r) & kKept);
    const unsigned long long m = __ballot(kept);
    if (lane == 0) s_c[wave] = __popcll(m);
    __syncthreads();
    int off = seg_off[seg];
    for (int w = 0; w < wave; ++w) off += s_c[w];
    off += __popcll(m & ((1ull << lane) - 1ull));
    if (kept && off < out_cap) out[off] = r;
    __syncthreads();
  }
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
  for (int t = wave_g; t < ntags; t += nwave) {
    const long long p = tag_idx[t];
    const bool dem = p + 119ll * sps + half < n;
    bool b0 = false, b1 = false;
    float q0 = 0.0f, q1 = 0.0f;
    if (dem) {
      const long long s0 = p + 8ll * sps + (long long)lane * sps;
      const float x1 = xg<MODE>(data, n, s0), x0 = xg<MODE>(data, n, s0 + half);
      b0 = x1 > x0; q0 = __fdiv_rn(x1, x0);
      if (lane < 48) {
        const long long s1 = s0 + 64ll * sps;
        const float y1 = xg<MODE>(data, n, s1), y0 = xg<MODE>(data, n, s1 + half);
        b1 = y1 > y0; q1 = __fdiv_rn(y1, y0);
      }
    }
    const unsigned long long ma = __ballot(b0), mb = __ballot(b1);
    if (lane == 0) {
      const unsigned long long ra = __builtin_bswap64(__brevll(ma));
      const unsigned long long rb = __builtin_bswap64(__brevll(mb));
      for (int k = 0; k < 8; ++k) bits14[(long long)t * 14 + k] = (unsigned char)(ra >> (8 * k));
      for (int k = 0; k < 6; ++k) bits14[(long long)t * 14 + 8 + k] = (unsigned char)(rb >> (8 * k));
      ok[t] = dem ? 1 : 0;
    }
    if (ratio && dem) {
      ratio[(long long)t * 112 + lane] = q0;
      if (lane < 48) ratio[(long long)t * 112 + 64 + lane] = q1;
    }
  }
}

}  // namespace adsb
