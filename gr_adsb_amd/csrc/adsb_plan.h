// adsb_plan.h -- host-side index bookkeeping shared by libadsb_hip.so and the test emulator driver:
// how one reference work() call (or one overlapped shard) maps onto k_detect's local index space, and
// how the framer's two words of cross-call state evolve.  Pure C++, no HIP, no arithmetic on samples.
//
// Local index i of the device buffer <-> framer in0 index j = i - in0_base <-> stream offset origin + i.
// Citations: /root/reference/python/adsb/framer.py, demod.py.
#pragma once

namespace adsb {

struct Plan {
  int mode;                 // 0 complex64 IQ, 1 float |IQ|^2
  const void* d_data;
  long long n;              // samples in the buffer
  long long in0_base;       // local index of in0[0]
  long long scan_lo, scan_hi;   // thresholded range == in0[0:N] (framer.py:83-84)
  long long fall_hi;        // falls must lie below (framer.py:102-108 drops a pulse still high at N)
  long long dem_hi;         // end of the demod input (demod.py:82)
  long long origin;         // stream offset of local index 0
  float prev_in0;           // framer.py:54,84,87
  int end_is_call_end;
  long long prev_eob_stream;    // framer.py:57 expressed as a stream offset
  bool gate;                // apply framer.py:121-123 on the device
  int head_n = 0;           // shard mode: deliver the first head_n centres whether gated or not
  bool long_aware = false;  // opt-in length-aware gate (never in GNU Radio emulation): set by the caller from the context
};

// How a pass over `ntiles` tiles (1024 samples each) is cut into chunks, one per wavefront, for a device that keeps
// `resident` wavefronts of k_detect resident at a time.  One resident round is the floor (units == ntiles while there
// are fewer tiles than resident wavefronts); a bulk pass is cut into up to kChunkRounds rounds of shorter chunks, none
// shorter than kMinChunkTiles tiles: the wavefronts of ONE round, each with 1/resident of the stream, finish up to 10 %
// apart (their burst counts differ) and the kernel ends with its slowest wavefront; with several rounds the dispatcher
// evens that out.  Measured on MI355X (tools/r3_variants.sh): 2^30 complex64 samples 0.74 -> 0.79-0.82 of the HBM peak
// with 8 rounds (16: the same for complex64, -6 % for int8 / int16; 4: -3 %), 2^28 samples 0.72 -> 0.79 with chunks
// down to 4 tiles (8: 0.77), 2^26 samples +1 %.
// Out: *units wavefronts with work, each *tiles_per tiles long (the last may be shorter): units * tiles_per >= ntiles.
constexpr int kChunkRounds = 8, kMinChunkTiles = 4;
inline void plan_chunks(long long ntiles, long long resident, long long* units_out, long long* tiles_per_out) {
  if (ntiles < 1) ntiles = 1;
  if (resident < 1) resident = 1;
  long long rounds = ntiles / (resident * kMinChunkTiles);
  if (rounds < 1) rounds = 1;
  if (rounds > kChunkRounds) rounds = kChunkRounds;
  const long long umax = resident * rounds;
  long long units = ntiles < umax ? ntiles : umax;
  const long long tiles_per = (ntiles + units - 1) / units;
  units = (ntiles + tiles_per - 1) / tiles_per;
  *units_out = units;
  *tiles_per_out = tiles_per;
}

struct FramerState {
  float prev_in0 = 0.0f;    // framer.py:54
  long long prev_eob = -1;  // framer.py:57 (index into the NEXT call's in0)
};

// One canonical whole-buffer call on a fresh stream: history = 8*sps-1 zeros, N = n, then demod over
// the same n samples (SURVEY.md §8a "chunk semantics").
inline Plan plan_canonical(int mode, const void* d, long long n, long long abs_offset, int sps) {
  const long long H = 8ll * sps;
  Plan p;
  p.mode = mode; p.d_data = d; p.n = n;
  p.in0_base = -(H - 1); p.scan_lo = p.in0_base; p.scan_hi = n - (H - 1); p.fall_hi = p.scan_hi;
  p.dem_hi = n; p.origin = abs_offset; p.prev_in0 = 0.0f; p.end_is_call_end = 1;
  p.prev_eob_stream = abs_offset + p.in0_base - 1;      // prev_eob_idx = -1
  p.gate = true;
  return p;
}

// framer.work() as GNU Radio calls it: the buffer IS in0 (N + 8*sps - 1 floats, history first).
inline Plan plan_framer_work(const void* d_in0, long long n_in0, long long N, long long nitems_written, int sps,
                             const FramerState& st) {
  const long long H = 8ll * sps;
  Plan p;
  p.mode = 1; p.d_data = d_in0; p.n = n_in0; p.in0_base = 0; p.scan_lo = 0; p.scan_hi = N; p.fall_hi = N;
  p.dem_hi = 0;                                          // framer does not slice bits
  p.origin = nitems_written - (H - 1);                   // framer.py:170
  p.prev_in0 = st.prev_in0; p.end_is_call_end = 1;
  p.prev_eob_stream = p.origin + st.prev_eob;
  p.gate = true;
  return p;
}

// State after a framer.work() call (framer.py:87,95,121-123,165,177-179).
//   flags bit0/bit1: the call saw >=1 rise / >=1 fall in in0[0:N]; lastp: in0 index of the last
//   paired pulse centre or `none`; last_kept: in0 index of the last accepted centre (n_kept > 0).
inline void framer_state_update(FramerState& st, float last_sample, long long N, int sps, unsigned flags,
                                long long lastp, long long none, int n_kept, long long last_kept) {
  st.prev_in0 = last_sample;                             // framer.py:87
  if ((flags & 1u) && (flags & 2u)) {                    // framer.py:95
    long long eob = st.prev_eob;
    if (n_kept > 0) eob = last_kept + 63ll * sps;        // framer.py:165
    if (lastp != none && lastp > eob) eob = -1;          // framer.py:121-123
    if (eob >= N) eob -= N;                              // framer.py:177-179
    st.prev_eob = eob;
  }
}

// One overlapped time shard of a canonical whole-stream call.  The buffer holds stream samples
// [origin, origin+n); this shard owns rises with stream offset in [own_lo, own_hi).
inline Plan plan_shard(int mode, const void* d, long long n, long long origin, long long own_lo, long long own_hi,
                       long long stream_len, int sps, int head_n = 0) {
  const long long H = 8ll * sps;
  const long long scan_end = stream_len - (H - 1);       // framer scans stream offsets [-(H-1), stream_len-(H-1))
  Plan p;
  p.mode = mode; p.d_data = d; p.n = n; p.origin = origin;
  p.in0_base = -(H - 1) - origin;
  long long lo = own_lo, hi = own_hi;
  if (lo <= 0 && origin == 0) lo = -(H - 1);
  if (hi > scan_end) hi = scan_end;
  p.scan_lo = lo - origin; p.scan_hi = hi - origin;
  if (p.scan_lo < p.in0_base) p.scan_lo = p.in0_base;
  const bool has_end = origin + n >= stream_len;
  p.fall_hi = has_end ? scan_end - origin : n;
  p.end_is_call_end = has_end ? 1 : 0;
  p.dem_hi = stream_len - origin;
  p.prev_in0 = 0.0f;
  p.prev_eob_stream = -(1ll << 61);
  p.gate = head_n > 0;                                   // gated with fresh state + whole head, or not gated at all
  p.head_n = head_n;
  return p;
}

}  // namespace adsb

