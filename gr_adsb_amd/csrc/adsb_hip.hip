// adsb_hip.hip -- host side of libadsb_hip.so (C ABI in include/adsb_hip.h) for gfx950.
// Owns device memory, pinned staging and the launch sequence
//   k_detect -> k_scan -> k_gather -> k_resolve -> k_count -> k_scan2 -> k_compact -> k_burst  (+ k_longrun when needed)
// There is deliberately no CPU implementation of the path in this library.
#include <hip/hip_runtime.h>

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

// wavefront-level ordering point used by adsb_device.h: LDS traffic of one wavefront is executed in
// program order by the hardware, this only stops the compiler from moving LDS accesses across it
__device__ __forceinline__ void adsb_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int adsb_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

#include "adsb_device.h"
#include "adsb_plan.h"
#include "../../include/adsb_hip.h"

using namespace adsb;

static_assert(sizeof(adsb_burst) == 32 && sizeof(Rec) == 32, "record layout");

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

}  // namespace

struct adsb_ctx {
  int device = 0;
  double fs = 0;
  int sps = 0;
  float thr = 0;
  uint32_t flags = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int n_cu = 256;
  int bpc[2] = {4, 4};  // resident k_detect workgroups per CU (occupancy query), per input mode
  // framer state (framer.py:54,57)
  FramerState st;
  // device buffers
  DevBuf d_in, d_cands, d_kept, d_blk_count, d_blk_lastp, d_blk_flags, d_blk_off, d_long, d_misc, d_sorted, d_seg, d_out,
      d_tags, d_bits, d_ok, d_ratio;
  int rec_cap_shift = 0;  // rec_cap multiplier (grows on overflow)
  // pinned host
  Summary* h_sum = nullptr;
  void* h_out = nullptr;
  size_t h_out_cap = 0;
  void* h_stage = nullptr;
  size_t h_stage_cap = 0;
  int32_t last_n = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  adsb_stats stats{};
  char err[256] = {0};
};

namespace {

int fail(adsb_ctx* c, int code, const char* what, hipError_t he = hipSuccess) {
  if (c) {
    if (he != hipSuccess) snprintf(c->err, sizeof(c->err), "%s: %s", what, hipGetErrorString(he));
    else snprintf(c->err, sizeof(c->err), "%s", what);
  }
  return code;
}

#define HIPCHK(c, call)                                   \
  do {                                                    \
    hipError_t e_ = (call);                               \
    if (e_ != hipSuccess) return fail((c), -EIO, #call, e_); \
  } while (0)

int ensure(adsb_ctx* c, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return 0;
  if (b.p) { HIPCHK(c, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
  size_t want = bytes + bytes / 4 + 256;
  HIPCHK(c, hipMalloc(&b.p, want));
  b.cap = want;
  return 0;
}

int ensure_pinned(adsb_ctx* c, void*& p, size_t& cap, size_t bytes) {
  if (bytes <= cap) return 0;
  if (p) { HIPCHK(c, hipHostFree(p)); p = nullptr; cap = 0; }
  size_t want = bytes + bytes / 4 + 4096;
  HIPCHK(c, hipHostMalloc(&p, want, hipHostMallocDefault));
  cap = want;
  return 0;
}

// d_misc layout: [0] int long_count, [8] u64 long_lastp, [64] Summary
struct Misc {
  int long_count;
  int pad;
  unsigned long long long_lastp;
  char fill[48];
  Summary sum;
};

template <int MODE>
void launch_detect(adsb_ctx* c, const DetectArgs& a, int grid) {
  // ADSB_DEBUG_DYNLDS (bytes): tuning knob that pads the workgroup's LDS to lower occupancy on purpose
  static const int dyn = getenv("ADSB_DEBUG_DYNLDS") ? atoi(getenv("ADSB_DEBUG_DYNLDS")) : 0;
  hipLaunchKernelGGL((k_detect<MODE>), dim3(grid), dim3(kThreads), dyn, c->stream, a);
}
template <int MODE>
void launch_burst(adsb_ctx* c, const DetectArgs& a, const unsigned long long* kept, const Summary* sum, unsigned orflags,
                  Rec* out, int cap) {
  hipLaunchKernelGGL((k_burst<MODE>), dim3(c->n_cu * 8), dim3(kThreads), 0, c->stream, a, kept, sum, orflags, out, cap);
}
template <int MODE>
void launch_longrun(adsb_ctx* c, const DetectArgs& a, int n) {
  int g = n < 256 ? n : 256;
  hipLaunchKernelGGL((k_longrun<MODE>), dim3(g), dim3(kThreads), 0, c->stream, a, n);
}

// Runs the whole device pipeline for one plan.  On return the kept (or all matched) records are in
// c->h_out (pinned), count in *n_res, and *sum holds the device summary.
int run_pipeline(adsb_ctx* c, const Plan& pl, Summary* sum, int32_t* n_res) {
  HIPCHK(c, hipSetDevice(c->device));
  c->stats.calls++;
  const long long span = pl.scan_hi > 0 ? pl.scan_hi : 0;
  long long ntiles = (span + kTile - 1) / kTile;
  if (ntiles < 1) ntiles = 1;
  // exactly one resident round of workgroups: a partial second round costs ~20 % (tail effect)
  static const int bpc_env = getenv("ADSB_DEBUG_BPC") ? atoi(getenv("ADSB_DEBUG_BPC")) : 0;
  const long long gmax = (long long)c->n_cu * (bpc_env > 0 ? bpc_env : c->bpc[pl.mode]);
  int grid = (int)(ntiles < gmax ? ntiles : gmax);
  const long long tiles_per = (ntiles + grid - 1) / grid;
  grid = (int)((ntiles + tiles_per - 1) / tiles_per);
  const long long chunk = tiles_per * kTile;

  for (int attempt = 0; attempt < 12; ++attempt) {
    long long rc = (chunk / 256 + 64) << c->rec_cap_shift;
    if (rc > chunk / 2 + 8) rc = chunk / 2 + 8;   // can never have more rises than that
    const int rec_cap = (int)rc;
    const long long tot = (long long)grid * rec_cap;
    const int long_cap = grid + 1;  // at most one long pulse per tile that ends a chunk... bounded by tiles
    const long long long_cap_ll = ntiles + 1;
    int rcx;
    if ((rcx = ensure(c, c->d_cands, (size_t)tot * 8))) return rcx;
    if ((rcx = ensure(c, c->d_sorted, (size_t)tot * 8))) return rcx;
    if ((rcx = ensure(c, c->d_kept, (size_t)tot * 8))) return rcx;
    if ((rcx = ensure(c, c->d_out, (size_t)tot * sizeof(Rec)))) return rcx;
    if ((rcx = ensure(c, c->d_seg, (size_t)(tot / kThreads + 2) * sizeof(int)))) return rcx;
    if ((rcx = ensure(c, c->d_blk_count, (size_t)grid * sizeof(int)))) return rcx;
    if ((rcx = ensure(c, c->d_blk_lastp, (size_t)grid * sizeof(long long)))) return rcx;
    if ((rcx = ensure(c, c->d_blk_flags, (size_t)grid * sizeof(unsigned)))) return rcx;
    if ((rcx = ensure(c, c->d_blk_off, (size_t)grid * sizeof(int)))) return rcx;
    if ((rcx = ensure(c, c->d_long, (size_t)long_cap_ll * sizeof(LongRise)))) return rcx;
    if ((rcx = ensure(c, c->d_misc, sizeof(Misc)))) return rcx;
    (void)long_cap;
    Misc* misc = (Misc*)c->d_misc.p;

    DetectArgs a;
    a.data = pl.d_data; a.n = pl.n; a.in0_base = pl.in0_base; a.scan_lo = pl.scan_lo; a.scan_hi = pl.scan_hi;
    a.fall_hi = pl.fall_hi; a.dem_hi = pl.dem_hi; a.origin = pl.origin; a.chunk = chunk; a.thr = c->thr;
    a.prev_in0 = pl.prev_in0; a.sps = c->sps; a.end_is_call_end = pl.end_is_call_end; a.rec_cap = rec_cap;
    a.long_cap = (int)long_cap_ll; a.cands = (unsigned long long*)c->d_cands.p; a.blk_count = (int*)c->d_blk_count.p;
    a.blk_lastp = (long long*)c->d_blk_lastp.p; a.blk_flags = (unsigned*)c->d_blk_flags.p;
    a.longlist = (LongRise*)c->d_long.p; a.long_count = &misc->long_count; a.long_lastp = &misc->long_lastp;

    HIPCHK(c, hipMemsetAsync(misc, 0, 16, c->stream));
    const bool timing = (c->flags & ADSB_FLAG_TIMING) != 0;
    if (timing) HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (pl.mode == 0) launch_detect<0>(c, a, grid); else launch_detect<1>(c, a, grid);
    c->stats.detect_grid = (uint64_t)grid; c->stats.blocks_per_cu = (uint64_t)c->bpc[pl.mode];
    if (timing) HIPCHK(c, hipEventRecord(c->ev1, c->stream));

    bool did_long = false;
    for (;;) {
      hipLaunchKernelGGL(k_scan, dim3(1), dim3(kThreads), 0, c->stream, (const int*)a.blk_count,
                         (const long long*)a.blk_lastp, (const unsigned*)a.blk_flags, grid, rec_cap,
                         (const int*)a.long_count, (const unsigned long long*)a.long_lastp, (int*)c->d_blk_off.p,
                         &misc->sum);
      int gg = grid < 1024 ? grid : 1024;
      unsigned long long* sorted = (unsigned long long*)c->d_sorted.p;
      unsigned long long* kept = (unsigned long long*)c->d_kept.p;
      hipLaunchKernelGGL(k_gather, dim3(gg), dim3(kThreads), 0, c->stream, (const unsigned long long*)a.cands,
                         (const int*)a.blk_count, (const int*)c->d_blk_off.p, grid, rec_cap, sorted);
      const int ag = 512;
      unsigned fmask = kNoMatch | kPending, fwant = 0u, orflags = 0u;
      if (pl.gate) {
        hipLaunchKernelGGL(k_resolve, dim3(ag), dim3(kThreads), 0, c->stream, sorted, (const Summary*)&misc->sum,
                           (long long)63 * c->sps, pl.prev_eob_stream - pl.origin);
        fmask = kKept; fwant = kKept; orflags = kKept;
      }
      hipLaunchKernelGGL(k_count, dim3(ag), dim3(kThreads), 0, c->stream, (const unsigned long long*)sorted,
                         (const Summary*)&misc->sum, fmask, fwant, (int*)c->d_seg.p);
      hipLaunchKernelGGL(k_scan2, dim3(1), dim3(kThreads), 0, c->stream, (int*)c->d_seg.p, &misc->sum);
      hipLaunchKernelGGL(k_compact, dim3(ag), dim3(kThreads), 0, c->stream, (const unsigned long long*)sorted,
                         &misc->sum, (const int*)c->d_seg.p, fmask, fwant, kept, (int)tot);
      if (pl.mode == 0) launch_burst<0>(c, a, kept, &misc->sum, orflags, (Rec*)c->d_out.p, (int)tot);
      else launch_burst<1>(c, a, kept, &misc->sum, orflags, (Rec*)c->d_out.p, (int)tot);
      HIPCHK(c, hipMemcpyAsync(c->h_sum, &misc->sum, sizeof(Summary), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      HIPCHK(c, hipGetLastError());
      if (c->h_sum->long_count > 0 && !did_long && !c->h_sum->overflow) {
        // rare: pulses longer than the LDS window (or starting in the zero history)
        did_long = true;
        c->stats.longrun_calls++;
        int nl = c->h_sum->long_count;
        if (nl > (int)long_cap_ll) return fail(c, -EIO, "long-rise list overflow");
        if (pl.mode == 0) launch_longrun<0>(c, a, nl); else launch_longrun<1>(c, a, nl);
        continue;
      }
      break;
    }
    if (timing) {
      float ms = 0;
      HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
      c->stats.detect_launches++;
      c->stats.detect_ms += ms;
      c->stats.detect_samples += (uint64_t)(span > 0 ? span : 0);
      c->stats.detect_bytes += (uint64_t)(span > 0 ? span : 0) * (pl.mode == 0 ? 8u : 4u);
    }
    if (c->h_sum->overflow) {
      c->rec_cap_shift++;
      c->stats.retries++;
      if (rc >= chunk / 2 + 8) return fail(c, -EIO, "record capacity overflow at maximum size");
      continue;
    }
    *sum = *c->h_sum;
    const int nres = sum->n_kept;
    int rcx2;
    if ((rcx2 = ensure_pinned(c, c->h_out, c->h_out_cap, (size_t)(nres > 0 ? nres : 1) * sizeof(Rec)))) return rcx2;
    if (nres > 0) {
      const void* src = c->d_out.p;
      HIPCHK(c, hipMemcpyAsync(c->h_out, src, (size_t)nres * sizeof(Rec), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    *n_res = nres;
    c->last_n = nres;
    return 0;
  }
  return fail(c, -EIO, "record capacity did not converge");
}

int deliver(adsb_ctx* c, int32_t nres, adsb_burst* out, int32_t cap, int32_t* n_out) {
  if (n_out) *n_out = nres;
  if (out) {
    if (nres > cap) return fail(c, -ENOSPC, "output array too small");
    if (nres > 0) memcpy(out, c->h_out, (size_t)nres * sizeof(adsb_burst));
  }
  return 0;
}

int canonical(adsb_ctx* c, int mode, const void* d_data, int64_t n, int64_t abs_offset, adsb_burst* out,
              int32_t cap, int32_t* n_out) {
  if (!c || n < 0) return -EINVAL;
  if (((uintptr_t)d_data & 15u) != 0) return fail(c, -EINVAL, "device pointer must be 16-byte aligned");
  Plan pl = plan_canonical(mode, d_data, n, abs_offset, c->sps);
  Summary s;
  int32_t nres = 0;
  if (n == 0) { c->last_n = 0; if (n_out) *n_out = 0; return 0; }
  int rc = run_pipeline(c, pl, &s, &nres);
  if (rc) return rc;
  return deliver(c, nres, out, cap, n_out);
}

int upload(adsb_ctx* c, const void* host, size_t bytes, void** d_out) {
  int rc;
  if ((rc = ensure(c, c->d_in, bytes + 64))) return rc;
  if ((rc = ensure_pinned(c, c->h_stage, c->h_stage_cap, bytes))) return rc;
  memcpy(c->h_stage, host, bytes);
  HIPCHK(c, hipMemcpyAsync(c->d_in.p, c->h_stage, bytes, hipMemcpyHostToDevice, c->stream));
  *d_out = c->d_in.p;
  return 0;
}

}  // namespace

extern "C" {

int adsb_abi_version(void) { return ADSB_ABI_VERSION; }

float adsb_snr_db(float peak, float median) {
  // framer.py:157 under NumPy-2 promotion: every operation in float32
  volatile float q = peak / median;
  volatile float l = log10f(q);
  volatile float m = 10.0f * l;
  volatile float r = m + 1.6f;
  return r;
}

int adsb_create(double fs, float threshold, int device, uint32_t flags, adsb_ctx** out) {
  if (!out) return -EINVAL;
  *out = nullptr;
  if (!(fs > 0) || fmod(fs, 1e6) != 0.0) return -EINVAL;        // framer.py:44, demod.py:42
  long long sps = (long long)(fs / 1e6);
  if (sps < 2 || (sps & 1) || sps > 4096) return -EINVAL;        // odd sps crashes the reference's work()
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return -ENODEV;   // no CPU fallback, by design
  if (device < 0 || device >= ndev) return -ENODEV;
  adsb_ctx* c = new (std::nothrow) adsb_ctx();
  if (!c) return -ENOMEM;
  c->device = device; c->fs = fs; c->sps = (int)sps; c->thr = threshold; c->flags = flags;
  if (hipSetDevice(device) != hipSuccess) { delete c; return -ENODEV; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
  {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_detect<0>, kThreads, 0) == hipSuccess && nb > 0) c->bpc[0] = nb;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_detect<1>, kThreads, 0) == hipSuccess && nb > 0) c->bpc[1] = nb;
  }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return -EIO; }
  c->own_stream = true;
  if (hipHostMalloc((void**)&c->h_sum, sizeof(Summary), hipHostMallocDefault) != hipSuccess) { adsb_destroy(c); return -ENOMEM; }
  if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) { adsb_destroy(c); return -EIO; }
  *out = c;
  return 0;
}

void adsb_destroy(adsb_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  DevBuf* bufs[] = {&c->d_in, &c->d_cands, &c->d_kept, &c->d_blk_count, &c->d_blk_lastp, &c->d_blk_flags, &c->d_blk_off, &c->d_long,
                    &c->d_misc, &c->d_sorted, &c->d_seg, &c->d_out, &c->d_tags, &c->d_bits, &c->d_ok, &c->d_ratio};
  for (DevBuf* b : bufs) if (b->p) (void)hipFree(b->p);
  if (c->h_sum) (void)hipHostFree(c->h_sum);
  if (c->h_out) (void)hipHostFree(c->h_out);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int adsb_set_threshold(adsb_ctx* c, float threshold) {
  if (!c) return -EINVAL;
  c->thr = threshold;
  return 0;
}

int adsb_set_stream(adsb_ctx* c, void* hip_stream) {
  if (!c) return -EINVAL;
  if (c->own_stream && c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
  c->stream = (hipStream_t)hip_stream;
  c->own_stream = false;
  return 0;
}

int adsb_reset(adsb_ctx* c) {
  if (!c) return -EINVAL;
  c->st = FramerState();
  return 0;
}

int adsb_process_iq_device(adsb_ctx* c, const void* d_iq, int64_t n, int64_t abs_offset, adsb_burst* out,
                           int32_t cap, int32_t* n_out) {
  return canonical(c, 0, d_iq, n, abs_offset, out, cap, n_out);
}

int adsb_process_mag2_device(adsb_ctx* c, const void* d_mag2, int64_t n, int64_t abs_offset, adsb_burst* out,
                             int32_t cap, int32_t* n_out) {
  return canonical(c, 1, d_mag2, n, abs_offset, out, cap, n_out);
}

int adsb_process_iq(adsb_ctx* c, const float* iq_host, int64_t n, int64_t abs_offset, adsb_burst* out,
                    int32_t cap, int32_t* n_out) {
  if (!c || n < 0 || (n > 0 && !iq_host)) return -EINVAL;
  if (n == 0) { if (n_out) *n_out = 0; c->last_n = 0; return 0; }
  HIPCHK(c, hipSetDevice(c->device));
  void* d = nullptr;
  int rc = upload(c, iq_host, (size_t)n * 8, &d);
  if (rc) return rc;
  return canonical(c, 0, d, n, abs_offset, out, cap, n_out);
}

int adsb_process_mag2(adsb_ctx* c, const float* mag2_host, int64_t n, int64_t abs_offset, adsb_burst* out,
                      int32_t cap, int32_t* n_out) {
  if (!c || n < 0 || (n > 0 && !mag2_host)) return -EINVAL;
  if (n == 0) { if (n_out) *n_out = 0; c->last_n = 0; return 0; }
  HIPCHK(c, hipSetDevice(c->device));
  void* d = nullptr;
  int rc = upload(c, mag2_host, (size_t)n * 4, &d);
  if (rc) return rc;
  return canonical(c, 1, d, n, abs_offset, out, cap, n_out);
}

int adsb_last_result(adsb_ctx* c, const adsb_burst** bursts, int32_t* n) {
  if (!c) return -EINVAL;
  if (bursts) *bursts = (const adsb_burst*)c->h_out;
  if (n) *n = c->last_n;
  return 0;
}

int adsb_framer_work(adsb_ctx* c, const float* in0, int64_t n_in0, int64_t N, int64_t nitems_written,
                     adsb_burst* tags, int32_t cap, int32_t* n_out) {
  if (!c || !in0 || N < 1) return -EINVAL;
  const long long H = 8ll * c->sps;
  if (n_in0 != N + H - 1) return fail(c, -EINVAL, "framer input must hold N + 8*sps - 1 items");
  HIPCHK(c, hipSetDevice(c->device));
  void* d = nullptr;
  int rc = upload(c, in0, (size_t)n_in0 * 4, &d);
  if (rc) return rc;
  Plan pl = plan_framer_work(d, n_in0, N, nitems_written, c->sps, c->st);
  Summary s;
  int32_t nres = 0;
  rc = run_pipeline(c, pl, &s, &nres);
  if (rc) return rc;
  // cross-call state, exactly as framer.py:87,121-123,165,177-179 (in0 index == local index here)
  framer_state_update(c->st, in0[N - 1], N, c->sps, s.flags, s.lastp, kNoIndex, nres,
                      nres > 0 ? s.last_kept_p : 0);
  return deliver(c, nres, tags, cap, n_out);
}

int adsb_demod_work(adsb_ctx* c, const float* in0, int64_t n, int64_t nitems_read, const int64_t* tag_offsets,
                    int32_t ntags, uint8_t* bits112, uint8_t* ok, float* ratio) {
  if (!c || n < 0 || ntags < 0 || (n > 0 && !in0) || (ntags > 0 && (!tag_offsets || !bits112 || !ok))) return -EINVAL;
  if (ntags == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  void* d = nullptr;
  int rc = upload(c, in0, (size_t)n * 4, &d);
  if (rc) return rc;
  if ((rc = ensure(c, c->d_tags, (size_t)ntags * 8))) return rc;
  if ((rc = ensure(c, c->d_bits, (size_t)ntags * 14))) return rc;
  if ((rc = ensure(c, c->d_ok, (size_t)ntags))) return rc;
  if (ratio && (rc = ensure(c, c->d_ratio, (size_t)ntags * 112 * 4))) return rc;
  // local positions of the tags inside in0 (demod.py:79: offset - nitems_written)
  long long* loc = (long long*)malloc((size_t)ntags * 8);
  if (!loc) return -ENOMEM;
  for (int t = 0; t < ntags; ++t) loc[t] = tag_offsets[t] - nitems_read;
  hipError_t he = hipMemcpyAsync(c->d_tags.p, loc, (size_t)ntags * 8, hipMemcpyHostToDevice, c->stream);
  if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
  free(loc);
  if (he != hipSuccess) return fail(c, -EIO, "tag upload", he);
  int nb = (ntags + kWaves - 1) / kWaves;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL((k_slice<1>), dim3(nb), dim3(kThreads), 0, c->stream, (const void*)d, (long long)n,
                     (const long long*)c->d_tags.p, (int)ntags, c->sps, (unsigned char*)c->d_bits.p,
                     (unsigned char*)c->d_ok.p, ratio ? (float*)c->d_ratio.p : (float*)nullptr);
  unsigned char* packed = (unsigned char*)malloc((size_t)ntags * 14);
  if (!packed) return -ENOMEM;
  he = hipMemcpyAsync(packed, c->d_bits.p, (size_t)ntags * 14, hipMemcpyDeviceToHost, c->stream);
  if (he == hipSuccess) he = hipMemcpyAsync(ok, c->d_ok.p, (size_t)ntags, hipMemcpyDeviceToHost, c->stream);
  if (he == hipSuccess && ratio) he = hipMemcpyAsync(ratio, c->d_ratio.p, (size_t)ntags * 112 * 4, hipMemcpyDeviceToHost, c->stream);
  if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
  if (he == hipSuccess) he = hipGetLastError();
  if (he != hipSuccess) { free(packed); return fail(c, -EIO, "k_slice", he); }
  for (int t = 0; t < ntags; ++t)
    for (int k = 0; k < 112; ++k) bits112[(size_t)t * 112 + k] = (packed[(size_t)t * 14 + (k >> 3)] >> (7 - (k & 7))) & 1u;
  free(packed);
  return 0;
}

int adsb_shard_device(adsb_ctx* c, int fmt, const void* d_data, int64_t n, int64_t origin, int64_t own_lo,
                      int64_t own_hi, int64_t stream_len, adsb_burst* out, int32_t cap, int32_t* n_out) {
  if (!c || n < 0 || (fmt != 0 && fmt != 1)) return -EINVAL;
  if (((uintptr_t)d_data & 15u) != 0) return fail(c, -EINVAL, "device pointer must be 16-byte aligned");
  Plan pl = plan_shard(fmt, d_data, n, origin, own_lo, own_hi, stream_len, c->sps);
  if (origin > 0 && pl.scan_lo < 1) return fail(c, -EINVAL, "shard needs at least one sample of back halo");
  Summary s;
  int32_t nres = 0;
  int rc = run_pipeline(c, pl, &s, &nres);
  if (rc) return rc;
  if (s.flags & 4u) return fail(c, -EOVERFLOW, "pulse runs past the shard's forward halo");
  // drop placeholders that did not match; verify the demod window of every record was inside the shard
  Rec* r = (Rec*)c->h_out;
  int w = 0;
  for (int i = 0; i < nres; ++i) {
    const unsigned fl = (unsigned)(r[i].w[3] >> 48);
    const long long off = (long long)r[i].w[0];
    const long long eob = off + 119ll * c->sps + c->sps / 2;
    if (!(fl & kDemod) && eob < stream_len) return fail(c, -EOVERFLOW, "internal: demod flag");
    if ((fl & kDemod) && eob >= origin + n) return fail(c, -EOVERFLOW, "burst runs past the shard's forward halo");
    if (off - 100 < origin && origin > 0) return fail(c, -EOVERFLOW, "noise window runs past the shard's back halo");
    r[w++] = r[i];
  }
  c->last_n = w;
  return deliver(c, w, out, cap, n_out);
}

int adsb_stitch(adsb_burst* cands, int32_t n, int sps, int32_t* n_kept) {
  if (n < 0 || (n > 0 && !cands) || sps < 2) return -EINVAL;
  long long eob = -(1ll << 61);
  int w = 0;
  for (int i = 0; i < n; ++i) {
    if (i > 0 && cands[i].offset <= cands[i - 1].offset) return -EINVAL;  // must be in stream order
    if (cands[i].offset > eob) {                                           // framer.py:121
      eob = cands[i].offset + 63ll * sps;                                  // framer.py:165
      adsb_burst b = cands[i];
      b.flags |= ADSB_BURST_KEPT;
      cands[w++] = b;
    }
  }
  if (n_kept) *n_kept = w;
  return 0;
}

int adsb_get_stats(adsb_ctx* c, adsb_stats* out) {
  if (!c || !out) return -EINVAL;
  *out = c->stats;
  return 0;
}

int adsb_reset_stats(adsb_ctx* c) {
  if (!c) return -EINVAL;
  memset(&c->stats, 0, sizeof(c->stats));
  return 0;
}

const char* adsb_last_error(adsb_ctx* c) { return c ? c->err : "null context"; }

}  // extern "C"
