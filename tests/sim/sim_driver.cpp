// sim_driver.cpp -- TEST INFRASTRUCTURE ONLY.  Compiles the product's device code
// (gr_adsb_amd/csrc/adsb_device.h) against the SIMT emulator in hipsim.h and runs the same kernel
// sequence the library launches, on host memory, so that kernel logic can be checked against the
// oracle in the GPU-less build container.  Never linked into libadsb_hip.so.
#include "hipsim.h"

#include "../../gr_adsb_amd/csrc/adsb_device.h"
#include "../../gr_adsb_amd/csrc/adsb_plan.h"

#include <vector>

using namespace adsb;

extern "C" {

struct SimOut {
  int n_rec, n_kept, overflow, long_count;
  unsigned flags;
  long long lastp, last_kept;
};

static float* g_conf_out = nullptr; // sim_set_confidence_out: [n_kept][112] ratios of the next run (k_confidence), or null
static int g_tail_mode = 0;         // sim_set_tail_mode
static int g_long_aware = 0;      // opt-in length-aware gate (ADSB_FLAG_LONG_AWARE_GATE): set by sim_set_long_aware

#define SIM_BY_MODE(mode, K, ...)                          \
  switch (mode) {                                          \
    case 0: hipsim::launch(K<0>, __VA_ARGS__); break;      \
    case 1: hipsim::launch(K<1>, __VA_ARGS__); break;      \
    case 2: hipsim::launch(K<2>, __VA_ARGS__); break;      \
    case 3: hipsim::launch(K<3>, __VA_ARGS__); break;      \
    default: hipsim::launch(K<4>, __VA_ARGS__); break;     \
  }

// mode = ADSB_FMT_* (0 complex64, 1 float |IQ|^2, 2 int16 IQ, 3 int8 IQ, 4 uint8 offset-binary IQ).
// Returns 0 or -1 (overflow of out).
int sim_run(int mode, const float* data, long long n, long long in0_base, long long scan_lo, long long scan_hi,
            long long fall_hi, long long dem_hi, long long origin, float thr, float prev_in0, int sps,
            int end_is_call_end, long long prev_eob_stream, int gate, int head_n, int grid_max, int rec_cap_in, float scale,
            unsigned long long* out_recs /* 4 words each */, int out_cap, SimOut* so) {
  // same unit geometry as adsb_hip.hip: enqueue()  (grid_max = resident workgroups available)
  const long long span = scan_hi > 0 ? scan_hi : 0;
  const int upb = det_waves(mode);          // wavefronts per k_detect workgroup: four, or one (8-bit formats)
  const int tile = kWTile;
  long long ntiles = (span + tile - 1) / tile;
  if (ntiles < 1) ntiles = 1;
  long long units = 0, tiles_per = 0;
  plan_chunks(ntiles, (long long)grid_max * kWaves, &units, &tiles_per);  // adsb_plan.h: the library's own chunk / round policy
  const long long chunk = tiles_per * tile;
  // like adsb_hip.hip: enqueue(): k_detect runs one wavefront per workgroup (grid = units); a call of at most four units of
  // |IQ|^2 floats gets the four lists of the one-launch small pass
  const bool can_fuse = units <= kWaves && mode == 1;
  const int grid = can_fuse ? 1 : (int)((units + upb - 1) / upb);
  const int nlists = can_fuse ? kWaves : grid * upb;
  const int rec_cap = rec_cap_in > 0 ? rec_cap_in : (int)(chunk / 2 + 8);
  const long long tot = (long long)nlists * rec_cap;

  // data must be 16-byte aligned like a device allocation
  const size_t nby = (size_t)n * (size_t)mode_bytes(mode);
  float* dbuf = (float*)aligned_alloc(64, ((nby + 63) / 64 + 1) * 64);
  memcpy(dbuf, data, nby);

  std::vector<unsigned long long> cands(tot), sorted(tot);
  std::vector<Rec> recs(tot), outv(tot);
  std::vector<unsigned> sorted_src(tot);
  std::vector<int> blk_count(nlists), blk_off(nlists), seg(tot / kThreads + 2);
  std::vector<long long> blk_lastp(nlists);
  std::vector<unsigned> blk_flags(nlists);
  std::vector<LongRise> longlist(ntiles + 1);
  int long_count = 0;
  unsigned long long long_lastp = 0;
  Summary sum, sum_host;

  DetectArgs a;
  a.data = dbuf; a.n = n; a.in0_base = in0_base; a.scan_lo = scan_lo; a.scan_hi = scan_hi; a.fall_hi = fall_hi;
  a.dem_hi = dem_hi; a.origin = origin; a.chunk = chunk; a.thr = thr; a.prev_in0 = prev_in0; a.scale = scale; a.sps = sps;
  a.long_aware = g_long_aware;
  a.end_is_call_end = end_is_call_end; a.rec_cap = rec_cap; a.long_cap = (int)(ntiles + 1);
  a.cands = cands.data(); a.recs = recs.data(); a.blk_count = blk_count.data(); a.blk_lastp = blk_lastp.data();
  a.blk_flags = blk_flags.data(); a.longlist = longlist.data(); a.long_count = &long_count;
  a.long_lastp = &long_lastp;

  // same choice as adsb_hip.hip: enqueue(): a one-workgroup pass over |IQ|^2 floats whose tail is the fused one runs as ONE
  // kernel, k_pass_small (g_tail_mode 1 = always the kernel chain)
  const bool one_launch = can_fuse && (g_tail_mode == 2 || (g_tail_mode == 0 && tot <= 16384));
  // the instance the library would launch (per format and samples per chip), like adsb_hip.hip: launch_detect()
#define SIM_DETECT(MODE)                                                                   \
  switch (sps) {                                                                           \
    case 2: hipsim::launch(k_detect<MODE, 1>, grid, 64 * det_waves(MODE), a); break;       \
    case 4: hipsim::launch(k_detect<MODE, 2>, grid, 64 * det_waves(MODE), a); break;       \
    case 8: hipsim::launch(k_detect<MODE, 4>, grid, 64 * det_waves(MODE), a); break;       \
    case 20: hipsim::launch(k_detect<MODE, 10>, grid, 64 * det_waves(MODE), a); break;     \
    default: hipsim::launch(k_detect<MODE, 0>, grid, 64 * det_waves(MODE), a); break;      \
  }
  // like adsb_hip.hip: launch_detect(): int8 IQ with a power-of-two scale runs the dot-product instance
  int fe = 0;
  const bool pow2 = (mode == 3 || mode == 4) && frexpf(scale, &fe) == 0.5f && fe > -50 && fe < 50;
  if (!one_launch) switch (pow2 ? (mode == 3 ? 5 : 6) : mode) {
    case 0: SIM_DETECT(0) break;
    case 1: SIM_DETECT(1) break;
    case 2: SIM_DETECT(2) break;
    case 3: SIM_DETECT(3) break;
    case 5: SIM_DETECT(5) break;
    case 6: SIM_DETECT(6) break;
    default: SIM_DETECT(4) break;
  }
#undef SIM_DETECT
  // same choice as adsb_hip.hip: enqueue_tail(): small passes take the one-workgroup tail (g_tail_mode: 0 = as the
  // library decides, 1 = always the kernel chain, 2 = always the fused tail)
  const bool fused = g_tail_mode == 2 || (g_tail_mode == 0 && tot <= 16384);
  if (fused) {
    TailArgs t;
    t.cands = cands.data(); t.recs = recs.data(); t.blk_count = blk_count.data(); t.blk_lastp = blk_lastp.data();
    t.blk_flags = blk_flags.data(); t.blk_off = blk_off.data(); t.nblk = nlists; t.rec_cap = rec_cap;
    t.long_count = &long_count; t.long_lastp = &long_lastp; t.sorted = sorted.data(); t.sorted_src = sorted_src.data();
    t.seg_count = seg.data(); t.sum = &sum; t.host_sum = &sum_host; t.out = outv.data(); t.out_cap = (int)tot;
    t.gate_on = gate ? 1 : 0; t.head_n = head_n; t.gate = 63ll * sps; t.gate_long = (long long)(g_long_aware ? 119 : 63) * sps;
    t.prev_eob = prev_eob_stream - origin;
    t.seq = 7;
    if (one_launch) {
      switch (sps) {
        case 2: hipsim::launch(k_pass_small<1>, 1, kThreads, a, t); break;
        case 4: hipsim::launch(k_pass_small<2>, 1, kThreads, a, t); break;
        case 8: hipsim::launch(k_pass_small<4>, 1, kThreads, a, t); break;
        case 20: hipsim::launch(k_pass_small<10>, 1, kThreads, a, t); break;
        default: hipsim::launch(k_pass_small<0>, 1, kThreads, a, t); break;
      }
    } else {
      SIM_BY_MODE(mode, k_tail_small, 1, kThreads, a, t);
    }
    if (sum_host.pad_ != 7) return -8;               // the pass number the host polls for
    sum = sum_host;
    if (g_conf_out) SIM_BY_MODE(mode, k_confidence, 2, kThreads, a, (const Rec*)outv.data(), (const Summary*)&sum, (int)tot, g_conf_out);
  } else {
    OrderAcc acc{};
    SIM_BY_MODE(mode, k_order, (nlists + kOrderLists - 1) / kOrderLists, kThreads, a, nlists, sorted.data(), sorted_src.data(), &sum, &acc);
    unsigned fmask = 0u, fwant = 0u;
    if (gate) {
      hipsim::launch(k_resolve, 3, kThreads, sorted.data(), (const Summary*)&sum, (long long)63 * sps,
                     (long long)(g_long_aware ? 119 : 63) * sps, prev_eob_stream - origin);
      fmask = kKept; fwant = kKept;
    }
    hipsim::launch(k_count, 3, kThreads, (const unsigned long long*)sorted.data(), (const Summary*)&sum, fmask, fwant,
                   head_n, seg.data());
    // mid-size passes of the library: the first host_cap records also go to the pinned result buffer (here: three, so that
    // vectors with more survivors than that cross the boundary)
    constexpr int kHostCap = 3;
    std::vector<Rec> host_first(kHostCap);
    memset(host_first.data(), 0xEE, kHostCap * sizeof(Rec));
    hipsim::launch(k_compact, 3, kThreads, (const unsigned long long*)sorted.data(), (const Rec*)recs.data(), (const unsigned*)sorted_src.data(), &sum,
                   (const int*)seg.data(), fmask, fwant, head_n, outv.data(), (int)tot, &long_count, &long_lastp, &acc, &sum_host,
                   host_first.data(), kHostCap);
    if (acc.flags != 0u || acc.lastp_biased != 0ull) return -7;     // left clean for the slot's next pass
    sum = sum_host;
    {
      const int nh = sum.n_kept < kHostCap ? sum.n_kept : kHostCap;
      if (nh > 0 && memcmp(host_first.data(), outv.data(), (size_t)nh * sizeof(Rec)) != 0) return -9;
      for (int i = nh; i < kHostCap; ++i)
        for (size_t b = 0; b < sizeof(Rec); ++b)
          if (((const unsigned char*)&host_first[i])[b] != 0xEE) return -9;     // nothing stored past the survivors
    }
    if (g_conf_out) SIM_BY_MODE(mode, k_confidence, 2, kThreads, a, (const Rec*)outv.data(), (const Summary*)&sum, (int)tot, g_conf_out);
  }
  so->n_rec = sum.n_rec; so->n_kept = sum.n_kept; so->overflow = sum.overflow; so->long_count = sum.long_count;
  so->flags = sum.flags; so->lastp = sum.lastp; so->last_kept = sum.last_kept_p;
  const int nres = sum.n_kept;
  const Rec* src = outv.data();
  free(dbuf);
  if (nres > out_cap) return -1;
  memcpy(out_recs, src, (size_t)nres * sizeof(Rec));
  return 0;
}

void sim_set_long_aware(int v) { g_long_aware = v; }
// the kernel's tile geometry, so that seam-targeted tests follow it
void sim_geometry(int* tile, int* fwd, int* back) { *tile = kWTile; *fwd = kFwd; *back = kBack; }
void sim_set_confidence_out(float* p) { g_conf_out = p; }
void sim_set_tail_mode(int m) { g_tail_mode = m; }

// k_detect's own conversion of one 16-byte load (body_convert) for the 8-bit formats: words[4*k .. 4*k+4) -> out[8*k .. 8*k+8).
// mode 3 int8, 4 offset-binary uint8, 5 / 6 their power-of-two-scale instances (v_dot4c_i32_i8, see adsb_device.h).
void sim_convert8(int mode, const unsigned* words, long long nwords, float scale, float* out) {
  for (long long k = 0; k + 4 <= nwords; k += 4) {
    float4 q;
    memcpy(&q, words + k, 16);
    if (mode == 3) body_convert<3>(q, scale, out + 2 * k);
    else if (mode == 4) body_convert<4>(q, scale, out + 2 * k);
    else if (mode == 6) body_convert<kModeCu8Pow2>(q, scale, out + 2 * k);
    else body_convert<kModeSc8Pow2>(q, scale, out + 2 * k);
  }
}

// --- the library's three call shapes, through the same adsb_plan.h the library uses ------------------
int sim_canonical(int mode, const float* data, long long n, long long abs_offset, float thr, int sps, int grid_max,
                  int rec_cap, float scale, unsigned long long* out, int out_cap, SimOut* so) {
  Plan p = plan_canonical(mode, data, n, abs_offset, sps);
  return sim_run(mode, data, n, p.in0_base, p.scan_lo, p.scan_hi, p.fall_hi, p.dem_hi, p.origin, thr, p.prev_in0, sps,
                 p.end_is_call_end, p.prev_eob_stream, 1, 0, grid_max, rec_cap, scale, out, out_cap, so);
}

// state[0] = prev_in0 (float bits in a double), state[1] = prev_eob
int sim_framer_work(const float* in0, long long n_in0, long long N, long long nitems_written, float thr, int sps,
                    float* prev_in0, long long* prev_eob, int grid_max, unsigned long long* out, int out_cap,
                    SimOut* so) {
  FramerState st;
  st.prev_in0 = *prev_in0; st.prev_eob = *prev_eob;
  Plan p = plan_framer_work(in0, n_in0, N, nitems_written, sps, st);
  int rc = sim_run(1, in0, n_in0, p.in0_base, p.scan_lo, p.scan_hi, p.fall_hi, p.dem_hi, p.origin, thr, p.prev_in0, sps,
                   p.end_is_call_end, p.prev_eob_stream, 1, 0, grid_max, 0, 1.0f, out, out_cap, so);
  if (rc) return rc;
  framer_state_update(st, in0[N - 1], N, sps, so->flags, so->lastp, kNoIndex, so->n_kept,
                      so->n_kept > 0 ? so->last_kept : 0);
  *prev_in0 = st.prev_in0; *prev_eob = st.prev_eob;
  return 0;
}

int sim_shard(int mode, const float* data, long long n, long long origin, long long own_lo, long long own_hi,
              long long stream_len, float thr, int sps, int head_n, int grid_max, unsigned long long* out, int out_cap,
              SimOut* so) {
  Plan p = plan_shard(mode, data, n, origin, own_lo, own_hi, stream_len, sps, head_n);
  return sim_run(mode, data, n, p.in0_base, p.scan_lo, p.scan_hi, p.fall_hi, p.dem_hi, p.origin, thr, p.prev_in0, sps,
                 p.end_is_call_end, p.prev_eob_stream, p.gate ? 1 : 0, p.head_n, grid_max, 0, 1.0f, out, out_cap, so);
}

// k_slice for a tag list (demod block emulation)
int sim_slice(const float* in0, long long n, const long long* tag_idx, int ntags, int sps, unsigned char* bits14,
              unsigned char* ok, float* ratio) {
  float* dbuf = (float*)aligned_alloc(64, (((size_t)n * 4 + 63) / 64 + 1) * 64);
  memcpy(dbuf, in0, (size_t)n * 4);
  int nb = (ntags + kWaves - 1) / kWaves;
  if (nb > 4) nb = 4;
  if (nb < 1) nb = 1;
  hipsim::launch(k_slice<1>, nb, kThreads, (const void*)dbuf, n, tag_idx, ntags, sps, bits14, ok, ratio);
  free(dbuf);
  return 0;
}
}
