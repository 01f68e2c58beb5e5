// adsb_hip.hip -- host side of libadsb_hip.so (C ABI in include/adsb_hip.h) for gfx950.
// Owns device memory, pinned staging and the launch sequence
//   k_detect (the one pass over the samples: centres AND their burst records)
//   -> k_order (long pulses, counts -> offsets, words in stream order) -> k_resolve -> k_count -> k_compact (+ summary to the host)
// There is deliberately no CPU implementation of the path in this library.
//
// The library reads no environment variable.
#include <hip/hip_runtime.h>

#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

// wavefront-level ordering point used by adsb_device.h: LDS traffic of one wavefront is executed in
// program order by the hardware, this only stops the compiler from moving LDS accesses across it
__device__ __forceinline__ void adsb_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int adsb_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int adsb_readlane(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
// `v`, made to depend on `after` without an instruction: whatever uses the result cannot be scheduled before `after` exists
__device__ __forceinline__ unsigned adsb_after(unsigned v, float after) {
  asm volatile("" : "+v"(v) : "v"(after));
  return v;
}
// the value itself, but opaque to the optimiser (per-lane: a vector register)
__device__ __forceinline__ int adsb_opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
// c + the sum of the four products of the signed bytes of a and b (v_dot4c_i32_i8)
__device__ __forceinline__ int adsb_sdot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }
// streamed single-use data: non-temporal load (every sample is fetched exactly once)
template <class Q>
__device__ __forceinline__ Q adsb_ld_stream(const char* p) {
  using V = float __attribute__((ext_vector_type(sizeof(Q) / 4)));
  return __builtin_bit_cast(Q, __builtin_nontemporal_load(reinterpret_cast<const V*>(p)));
}
// issue priority of the calling wavefront among the wavefronts of its SIMD (0 = default ... 3)
template <int P>
__device__ __forceinline__ void adsb_setprio() { __builtin_amdgcn_s_setprio(P); }
// written once, read once much later (k_detect's burst lists): non-temporal store
typedef unsigned long long adsb_u64x2 __attribute__((ext_vector_type(2)));
template <class T>
__device__ __forceinline__ void adsb_st_stream(T* p, T v) { __builtin_nontemporal_store(v, p); }
// acc = 16*acc + the four threshold bits (x >= thr, NaN -> 0) of a, b, c, d (a highest): four compares into four
// scalar pairs, then a chain of add-with-carry (acc = 2*acc + bit).  Batched by four because gfx950 wants two wait
// states between a vector instruction that writes a scalar pair and a vector instruction that reads it: the three
// other compares are that distance, no s_nop is spent.
__device__ __forceinline__ unsigned adsb_above4(unsigned acc, float a, float b, float c, float d, float thr) {
  unsigned long long ca, cb, cc, cd;
  asm("v_cmp_le_f32 %1, %6, %7\n\t"
      "v_cmp_le_f32 %2, %6, %8\n\t"
      "v_cmp_le_f32 %3, %6, %9\n\t"
      "v_cmp_le_f32 %4, %6, %10\n\t"
      "v_addc_co_u32 %0, vcc, %5, %5, %1\n\t"
      "v_addc_co_u32 %0, vcc, %0, %0, %2\n\t"
      "v_addc_co_u32 %0, vcc, %0, %0, %3\n\t"
      "v_addc_co_u32 %0, vcc, %0, %0, %4"
      : "=&v"(acc), "=&s"(ca), "=&s"(cb), "=&s"(cc), "=&s"(cd)
      : "v"(acc), "s"(thr), "v"(a), "v"(b), "v"(c), "v"(d)
      : "vcc");
  return acc;
}
// x*x + y*y with two rounded products and one rounded add: one packed multiply on the register pair the
// sample was loaded into, one add.  As asm statements because the compiler's own packing of re*re + im*im wants (re0, re1) /
// (im0, im1) pairs and pays for them with three or four register moves per 16-byte load.
__device__ __forceinline__ float adsb_mag2(float x, float y) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f a = {x, y}, r;
  float m;
  asm("v_pk_mul_f32 %0, %1, %1" : "=v"(r) : "v"(a));
  asm("v_add_f32 %0, %1, %2" : "=v"(m) : "v"(r.x), "v"(r.y));
  return m;
}
// maximum of three floats with the hardware's own NaN rule (a quiet NaN operand is skipped); as an asm statement so that
// no canonicalising instruction is spent on operands that come straight from memory
__device__ __forceinline__ float adsb_fmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// the value of the lane below (lane 0: `fill`): DPP wave_shr:1
__device__ __forceinline__ unsigned adsb_lane_up1(unsigned v, unsigned fill) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xF, 0xF, false);
}
// inclusive prefix sum over the 64 lanes: four row_shr steps inside each row of 16, then row_bcast:15 / row_bcast:31
__device__ __forceinline__ unsigned adsb_wave_incl_scan(unsigned x) {
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);
  x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);
  return x;
}
// minimum over the 64 lanes (wave-uniform result), same DPP ladder
__device__ __forceinline__ unsigned adsb_wave_min_u32(unsigned v) {
#define ADSB_MIN_STEP(ctrl, rows)                                                                  \
  {                                                                                                \
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, ctrl, rows, 0xF, false);  \
    v = o < v ? o : v;                                                                             \
  }
  ADSB_MIN_STEP(0x111, 0xF) ADSB_MIN_STEP(0x112, 0xF) ADSB_MIN_STEP(0x114, 0xF) ADSB_MIN_STEP(0x118, 0xF)
  ADSB_MIN_STEP(0x142, 0xA) ADSB_MIN_STEP(0x143, 0xC)
#undef ADSB_MIN_STEP
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// maximum over the 64 lanes (wave-uniform result)
__device__ __forceinline__ unsigned adsb_wave_max_u32(unsigned v) {
#define ADSB_MAX_STEP(ctrl, rows)                                                                 \
  {                                                                                                \
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xF, false);   \
    v = o > v ? o : v;                                                                             \
  }
  ADSB_MAX_STEP(0x111, 0xF) ADSB_MAX_STEP(0x112, 0xF) ADSB_MAX_STEP(0x114, 0xF) ADSB_MAX_STEP(0x118, 0xF)
  ADSB_MAX_STEP(0x142, 0xA) ADSB_MAX_STEP(0x143, 0xC)
#undef ADSB_MAX_STEP
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// the dynamic LDS of the launch as an int array
#define ADSB_DYN_LDS_INT(name) extern __shared__ int name[]
// workgroup-local (LDS) address space qualifier for pointers that crossed a function call as generic pointers
#define ADSB_LDS __attribute__((address_space(3)))
// bit i of x -> bits 2i and 2i+1 (scalar unit; the argument must be wave-uniform)
__device__ __forceinline__ unsigned long long adsb_bitrep32(unsigned x) {
  unsigned long long r;
  asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(r) : "s"(x));
  return r;
}

// The kernel's own argument block in (kernarg) memory: DetectArgs is the FIRST parameter of every kernel that runs
// detect_body, so the block starts with it.  Through an empty asm statement, so that loads through it are neither hoisted
// out of the tile loop nor merged with the by-value copy: a field read through this pointer costs one s_load where it is
// used and no scalar register anywhere else.
namespace adsb { struct DetectArgs; }
typedef const __attribute__((address_space(4))) adsb::DetectArgs* adsb_cold_ptr;
__device__ __forceinline__ adsb_cold_ptr adsb_cold(const adsb::DetectArgs&) {
  auto p = __builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return (adsb_cold_ptr)p;
}

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

// d_misc layout: [0] int long_count, [8] u64 long_lastp, [16] OrderAcc (k_order's accumulator), [64] Summary
struct Misc {
  int long_count;
  int pad;
  unsigned long long long_lastp;
  OrderAcc acc;
  char fill[32];
  Summary sum;
};
static_assert(sizeof(OrderAcc) == 16, "Misc layout");

// Everything one in-flight call needs on the device and in pinned host memory.  Several slots let call i+1
// run on the GPU while the records of call i travel over PCIe (adsb_submit_* / adsb_wait), and -- host-fed -- while
// the samples of call i+2 travel the other way.
struct Slot {
  DevBuf d_cands, d_recs, d_sorted, d_sorted_src, d_out, d_seg, d_blk_count, d_blk_lastp, d_blk_flags, d_blk_off, d_long, d_misc;
  DevBuf d_in;                   // host-fed submissions: this call's samples (adsb_submit_format_host)
  DevBuf d_ratio;                // ADSB_FLAG_CONFIDENCE: [n_kept][112] bit1/bit0 ratios
  Summary* h_sum = nullptr;      // pinned
  void* h_out = nullptr;         // pinned burst records of the finished call
  size_t h_out_cap = 0;
  void* h_ratio = nullptr;       // pinned confidence ratios of the finished call
  size_t h_ratio_cap = 0;
  hipEvent_t done = nullptr, ev0 = nullptr, ev1 = nullptr, det_done = nullptr, h2d_done = nullptr;
  hipStream_t stream = nullptr;  // this slot's own in-order queue: a SUBMITTED pass runs on it from k_detect to k_compact (see enqueue)
  hipStream_t ds = nullptr;      // the stream this pass's k_detect was queued on ...
  hipStream_t cs = nullptr;      // ... and the one its tail, its summary and its record copy run on (the same, or behind an event)
  bool busy = false;
  bool direct = false;           // small pass: k_compact writes the records straight into h_out (no device->host copy)
  bool fused = false;            // ... and the whole pass is ONE launch (k_pass_small)
  int host_cap = 0;              // mid-size pass: k_compact stores the first host_cap records into h_out as well (see enqueue)
  bool is_shard = false;
  bool submitted = false;        // queued by adsb_submit_* / the sharded driver (other passes may be in flight beside it)
  bool ev1_valid = false;
  bool h2d_pending = false;      // host-fed submission: the upload's event has to be waited for by the k_detect stream
  int seq = 0;                   // direct passes: the number the kernel stores into h_sum->pad_ when everything is out
  bool polled = false;           // ... and finish() polls for instead of waiting for an event
  Plan plan{};
  DetectArgs args{};
  int grid = 0, nlists = 0, rec_cap = 0;
  long long tot = 0, ntiles = 0, chunk = 0, span = 0;
  int32_t nres = 0;
};

}  // namespace

// Host threads that copy a pageable source into the pinned staging ring of a host-fed submission: one host core moves
// ~10 GB/s, the DMA behind it 57 -- so the copy of every chunk is split over a few threads (created on the first pageable
// submission, parked on a condition variable in between).
struct CopyPool {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv_job, cv_done;
  const char* src = nullptr;
  char* dst = nullptr;
  size_t bytes = 0;
  unsigned gen = 0;
  int pending = 0;
  bool stop = false;
  int nthreads = 0;      // workers besides the caller
  cpu_set_t cpus;        // the host cpus local to the context's GPU (NUMA node of its PCI device); workers run there
  bool have_cpus = false;

  static void slice(size_t bytes, int parts, int i, size_t* lo, size_t* hi) {
    const size_t per = ((bytes / (size_t)parts) + 4095) & ~(size_t)4095;
    *lo = per * (size_t)i < bytes ? per * (size_t)i : bytes;
    *hi = (i == parts - 1) ? bytes : (*lo + per < bytes ? *lo + per : bytes);
  }
  void worker(int id) {
    unsigned seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(m);
      cv_job.wait(lk, [&] { return stop || gen != seen; });
      if (stop) return;
      seen = gen;
      const char* s_ = src; char* d_ = dst; const size_t b_ = bytes;
      lk.unlock();
      size_t lo, hi;
      slice(b_, nthreads + 1, id + 1, &lo, &hi);
      if (hi > lo) memcpy(d_ + lo, s_ + lo, hi - lo);
      lk.lock();
      if (--pending == 0) cv_done.notify_one();
    }
  }
  void start(int n) {
    nthreads = n;
    for (int i = 0; i < n; ++i) {
      th.emplace_back([this, i] { worker(i); });
      // the copy of a pageable source into the pinned ring is bound by host memory bandwidth: keep it on the socket the
      // ring lives on and the GPU hangs off (best effort: an error leaves the thread where the scheduler puts it)
      if (have_cpus) (void)pthread_setaffinity_np(th.back().native_handle(), sizeof(cpus), &cpus);
    }
  }
  // blocking: returns when all of [src, src + bytes) is in dst
  void copy(char* d_, const char* s_, size_t b_) {
    if (nthreads == 0 || b_ < ((size_t)1 << 20)) { memcpy(d_, s_, b_); return; }
    {
      std::lock_guard<std::mutex> lk(m);
      src = s_; dst = d_; bytes = b_; pending = nthreads; ++gen;
    }
    cv_job.notify_all();
    size_t lo, hi;
    slice(b_, nthreads + 1, 0, &lo, &hi);
    memcpy(d_ + lo, s_ + lo, hi - lo);
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] { return pending == 0; });
  }
  ~CopyPool() {
    { std::lock_guard<std::mutex> lk(m); stop = true; }
    cv_job.notify_all();
    for (std::thread& t : th) t.join();
  }
};

struct adsb_ctx {
  int device = 0;
  double fs = 0;
  int sps = 0;
  float thr = 0;
  uint32_t flags = 0;
  // Queues.  A SUBMITTED pass (adsb_submit_*, the sharded driver) runs on its pipeline slot's own stream, k_detect and its
  // tail back to back: passes in different slots share nothing, so the hardware overlaps them as resources allow -- the
  // tail of pass i runs beside k_detect of pass i+1, and k_detect i+1 fills the CUs k_detect i drains from -- without one
  // event between them (round 5; until then: k_detect of every pass on `stream`, every tail on `tail_stream` behind an
  // event per pass: 9 us of the 27 us of host time a submission cost, tools/micro/launch_cost.hip).  A BLOCKING call runs
  // alone, on `stream`, as one queue.  A caller-owned stream (adsb_set_stream) keeps the old arrangement: k_detect on it,
  // tails on `tail_stream`; so do timed passes and passes over more than 4 GiB, with the slot streams in those roles (enqueue).
  // Few streams on purpose: the runtime multiplexes streams onto four hardware queues by default, and two streams that share
  // one serialise.  A context of its own uses three (the slots') plus the upload stream of host-fed submissions;
  // the record copy of a finished pass goes onto that pass's own -- by then idle -- stream.
  hipStream_t stream = nullptr;       // compute stream of the blocking entry points: slot 0's stream, or the caller's (adsb_set_stream)
  hipStream_t copy_stream = nullptr;  // caller-owned compute stream only: device -> pinned host result copies
  hipStream_t tail_stream = nullptr;  // caller-owned compute stream only: everything after k_detect
  // adsb_wait_for_event: the caller's events the NEXT pass has to wait for (applied to the stream its first kernel runs on)
  static constexpr int kMaxExt = 4;
  hipEvent_t ext_ev[kMaxExt] = {nullptr, nullptr, nullptr, nullptr};
  int n_ext = 0;
  hipStream_t h2d_stream = nullptr;   // host-fed submissions: sample uploads, back to back on their own stream
  hipStream_t d2h_stream = nullptr;   // record copies of in-line passes while slot 2's stream holds an overlapped pass (created on first use)
  bool split_tail = false;
  bool own_stream = false;
  int n_cu = 256;
  int bpc[ADSB_FMT_COUNT] = {4, 4, 4, 4, 4};  // resident k_detect workgroups per CU (occupancy query), per input format
  // Unused dynamic LDS per k_detect workgroup = how many workgroups share a CU (28.3 KB static: five fit; 6 KB of padding
  // would lift a workgroup over the 32 KB that five per CU allow).  Measured per format on MI355X (tools/r3_variants.sh,
  // 2^30 samples): five per CU for every format since round 4 -- float |IQ|^2 ran best with four until the SGPR spills went
  // (five: +1.3 % pipelined, 0.78 instead of 0.72 of its roofline isolated; profiles/r04_ab_workgroups_per_cu.txt); three:
  // -24 % for the 8-bit formats.  Six would need 1.4 KB less LDS per workgroup: a 256-entry rise list worked off in two
  // halves was built and measured -- six workgroups gave the 8-bit formats +5 %, the second code path took 4 % back.
  unsigned det_dyn_lds[ADSB_FMT_COUNT] = {0, 0, 0, 0, 0};
  unsigned lds_beside[ADSB_FMT_COUNT] = {0, 0, 0, 0, 0};    // LDS a CU has left beside its resident k_detect workgroups
  // integer IQ component -> float32 multiplier per format (adsb_set_format_scale); unused for the float formats
  float scale[ADSB_FMT_COUNT] = {1.0f, 1.0f, 1.0f / 32768.0f, 1.0f / 128.0f, 1.0f / 255.0f};
  FramerState st;       // framer.py:54,57
  Slot slot[ADSB_MAX_IN_FLIGHT];
  int next_slot = 0;
  int last_slot = 0;
  DevBuf d_in;            // device copy of the host input of the blocking entry points
  int rec_cap_shift = 0;  // rec_cap multiplier (grows on overflow)
  void* h_stage = nullptr;   // pinned staging for pageable inputs of the blocking entry points
  size_t h_stage_cap = 0;
  void* h_dm = nullptr;      // pinned scratch of adsb_demod_work: tag positions in, bits / ok / ratio out
  size_t h_dm_cap = 0;
  static constexpr int kRing = 4;
  void* h_ring[kRing] = {nullptr, nullptr, nullptr, nullptr};   // pinned chunks for pageable host-fed submissions
  hipEvent_t ring_done[kRing] = {nullptr, nullptr, nullptr, nullptr};
  bool ring_used[kRing] = {false, false, false, false};
  unsigned ring_k = 0;
  // NUMA placement of the host side (adsb_numa_info): the node and cpus local to the GPU's PCI device, from sysfs
  int numa_node = -1;
  char pci_bdf[32] = {0};
  char cpulist[256] = {0};
  cpu_set_t local_cpus;
  bool have_local_cpus = false;
  CopyPool* pool = nullptr;    // host copy threads of the pageable path (adsb_set_copy_threads; created on first use)
  int copy_threads = -1;       // -1 = default
  adsb_stats stats{};
  // per-launch k_detect durations of the timed calls (ADSB_FLAG_TIMING), a ring of the last kHist: adsb_detect_history
  static constexpr int kHist = 4096;
  std::vector<float> det_hist;
  uint64_t det_hist_n = 0;
  char err[256] = {0};
};

namespace {

// one polite iteration of a host spin loop
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield" ::: "memory");
#else
  std::this_thread::yield();
#endif
}

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

// Page-locked host memory on the NUMA node of the context's GPU: the calling thread's memory policy is set to prefer that
// node for the duration of the allocation (raw set_mempolicy: no libnuma in the image) and hipHostMallocNumaUser tells HIP
// to honour it.  Without a known node (numa_node < 0: single-socket host, or sysfs not visible) a plain allocation.
constexpr int kMpolDefault = 0, kMpolPreferred = 1;
// `coherent`: the buffer is read by the HOST while the kernel that writes it may still be running (the polled summary and
// the records of a one-workgroup pass): asked for explicitly as fine-grained memory (hipHostMallocCoherent) instead of
// relying on the runtime's default / HIP_HOST_COHERENT.
hipError_t host_alloc_near(adsb_ctx* c, void** p, size_t bytes, bool coherent = false) {
  const unsigned co = coherent ? hipHostMallocCoherent : 0u;
  if (!c || c->numa_node < 0 || c->numa_node >= 1024) {
    hipError_t e0 = hipHostMalloc(p, bytes, hipHostMallocDefault | co);
    if (e0 != hipSuccess && co) { (void)hipGetLastError(); e0 = hipHostMalloc(p, bytes, hipHostMallocDefault); }
    return e0;
  }
  unsigned long mask[16] = {0}, old_mask[16] = {0};
  mask[c->numa_node / (8 * sizeof(unsigned long))] |= 1ul << (c->numa_node % (8 * sizeof(unsigned long)));
  // the caller's own policy is put back afterwards (whatever it was)
  int old_mode = kMpolDefault;
  const bool have_old = syscall(SYS_get_mempolicy, &old_mode, old_mask, sizeof(old_mask) * 8, nullptr, 0) == 0;
  const long rc = syscall(SYS_set_mempolicy, kMpolPreferred, mask, sizeof(mask) * 8);
  hipError_t e = hipHostMalloc(p, bytes, (rc == 0 ? hipHostMallocNumaUser : hipHostMallocDefault) | co);
  if (e != hipSuccess && co) {                       // (a runtime that refuses the combination: placement first)
    (void)hipGetLastError();
    e = hipHostMalloc(p, bytes, rc == 0 ? hipHostMallocNumaUser : hipHostMallocDefault);
  }
  if (rc == 0) {
    if (!have_old || old_mode == kMpolDefault || syscall(SYS_set_mempolicy, old_mode, old_mask, sizeof(old_mask) * 8) != 0)
      (void)syscall(SYS_set_mempolicy, kMpolDefault, nullptr, 0);
  }
  if (e != hipSuccess && rc == 0) { (void)hipGetLastError(); e = hipHostMalloc(p, bytes, hipHostMallocDefault); }
  return e;
}

// "0-63,128-191" -> cpu set
bool parse_cpulist(const char* sl, cpu_set_t* set) {
  CPU_ZERO(set);
  int any = 0;
  const char* p = sl;
  while (*p) {
    char* e = nullptr;
    long a = strtol(p, &e, 10);
    if (e == p) break;
    long b = a;
    p = e;
    if (*p == '-') { b = strtol(p + 1, &e, 10); if (e == p + 1) break; p = e; }
    for (long k = a; k <= b && k < CPU_SETSIZE; ++k) { if (k >= 0) { CPU_SET((int)k, set); ++any; } }
    if (*p == ',') ++p; else break;
  }
  return any > 0;
}

bool read_line(const char* path, char* out, size_t cap) {
  FILE* f = fopen(path, "r");
  if (!f) return false;
  const bool ok = fgets(out, (int)cap, f) != nullptr;
  fclose(f);
  if (ok) { size_t n = strlen(out); while (n && (out[n - 1] == '\n' || out[n - 1] == ' ')) out[--n] = 0; }
  return ok;
}

// numa_node / local_cpulist of the GPU's PCI device (/sys/bus/pci/devices/<bdf>/): which host memory and which cpus are
// local to it.  Best effort: a container may hide sysfs, a single-socket host reports node -1 or 0.
void probe_numa(adsb_ctx* c) {
  char bdf[32] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), c->device) != hipSuccess) { (void)hipGetLastError(); return; }
  for (char* q = bdf; *q; ++q) if (*q >= 'A' && *q <= 'F') *q = (char)(*q - 'A' + 'a');      // sysfs spells it in lower case
  snprintf(c->pci_bdf, sizeof(c->pci_bdf), "%s", bdf);
  char path[128], line[256];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
  if (read_line(path, line, sizeof(line))) c->numa_node = atoi(line);
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bdf);
  if (read_line(path, line, sizeof(line)) && line[0]) {
    snprintf(c->cpulist, sizeof(c->cpulist), "%s", line);
    c->have_local_cpus = parse_cpulist(line, &c->local_cpus);
  }
}

int ensure_pinned(adsb_ctx* c, void*& p, size_t& cap, size_t bytes, bool coherent = false) {
  if (bytes <= cap) return 0;
  if (p) { HIPCHK(c, hipHostFree(p)); p = nullptr; cap = 0; }
  size_t want = bytes + bytes / 4 + 4096;
  HIPCHK(c, host_alloc_near(c, &p, want, coherent));
  cap = want;
  return 0;
}

// one instantiation of every sample-reading kernel per input format (ADSB_FMT_*)
#define ADSB_BY_MODE(mode, F, ...)            \
  switch (mode) {                             \
    case 0: F<0>(__VA_ARGS__); break;         \
    case 1: F<1>(__VA_ARGS__); break;         \
    case 2: F<2>(__VA_ARGS__); break;         \
    case 3: F<3>(__VA_ARGS__); break;         \
    default: F<4>(__VA_ARGS__); break;        \
  }

// k_detect is instantiated per input format and per samples-per-chip of the common rates (2, 4, 8, 20 Msps: the
// preamble taps become immediate offsets); any other even rate runs the run-time-stride instance
template <int KMODE>
void launch_detect_k(hipStream_t st, unsigned dyn, const DetectArgs& a, int grid) {
  switch (a.sps) {
    case 2: hipLaunchKernelGGL((k_detect<KMODE, 1>), dim3(grid), dim3(64 * det_waves(KMODE)), dyn, st, a); break;
    case 4: hipLaunchKernelGGL((k_detect<KMODE, 2>), dim3(grid), dim3(64 * det_waves(KMODE)), dyn, st, a); break;
    case 8: hipLaunchKernelGGL((k_detect<KMODE, 4>), dim3(grid), dim3(64 * det_waves(KMODE)), dyn, st, a); break;
    case 20: hipLaunchKernelGGL((k_detect<KMODE, 10>), dim3(grid), dim3(64 * det_waves(KMODE)), dyn, st, a); break;
    default: hipLaunchKernelGGL((k_detect<KMODE, 0>), dim3(grid), dim3(64 * det_waves(KMODE)), dyn, st, a); break;
  }
}
// a power of two whose square, times any integer below 2^15, is an exact float32: what the int8 dot-product instance needs
bool scale_is_pow2(float s) {
  int e = 0;
  const float m = frexpf(s, &e);
  return m == 0.5f && e > -50 && e < 50;
}
template <int MODE>
void launch_detect(adsb_ctx* c, hipStream_t st, const DetectArgs& a, int grid) {
  const unsigned dyn = c->det_dyn_lds[MODE];
  // int8 IQ with a power-of-two scale (x / 128 and the like): the instance whose tile loop squares with v_dot4_i32_i8
  if (MODE == ADSB_FMT_SC8 && scale_is_pow2(a.scale)) launch_detect_k<kModeSc8Pow2>(st, dyn, a, grid);
  // uint8 IQ with a power-of-two scale ((u8 - 127.5) / 128 and the like): the same for offset-binary bytes
  else if (MODE == ADSB_FMT_CU8 && scale_is_pow2(a.scale)) launch_detect_k<kModeCu8Pow2>(st, dyn, a, grid);
  else launch_detect_k<MODE>(st, dyn, a, grid);
}
template <int MODE>
unsigned detect_static_lds() {
  hipFuncAttributes at;
  if (hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_detect<MODE, 0>)) != hipSuccess) { (void)hipGetLastError(); return 32768u; }
  return (unsigned)at.sharedSizeBytes;
}
template <int MODE>
int detect_occupancy(unsigned dyn) {
  // (the instances of one format differ only in tap addressing: the same resources decide)
  int nb = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_detect<MODE, 0>, 64 * det_waves(MODE), dyn);
  if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
  return nb;
}
template <int MODE>
void launch_order(hipStream_t st, int grid, unsigned dyn, const DetectArgs& a, int nblk, unsigned long long* sorted, unsigned* sorted_src,
                  Summary* sum, OrderAcc* acc) {
  hipLaunchKernelGGL((k_order<MODE>), dim3(grid), dim3(kThreads), dyn, st, a, nblk, sorted, sorted_src, sum, acc);
}
template <int MODE>
void launch_tail_small(hipStream_t st, const DetectArgs& a, const TailArgs& t) {
  hipLaunchKernelGGL((k_tail_small<MODE>), dim3(1), dim3(kThreads), 0, st, a, t);
}
// the whole small pass in one launch (|IQ|^2 float input, one workgroup)
void launch_pass_small(hipStream_t st, const DetectArgs& a, const TailArgs& t) {
  switch (a.sps) {
    case 2: hipLaunchKernelGGL((k_pass_small<1>), dim3(1), dim3(kThreads), 0, st, a, t); break;
    case 4: hipLaunchKernelGGL((k_pass_small<2>), dim3(1), dim3(kThreads), 0, st, a, t); break;
    case 8: hipLaunchKernelGGL((k_pass_small<4>), dim3(1), dim3(kThreads), 0, st, a, t); break;
    case 20: hipLaunchKernelGGL((k_pass_small<10>), dim3(1), dim3(kThreads), 0, st, a, t); break;
    default: hipLaunchKernelGGL((k_pass_small<0>), dim3(1), dim3(kThreads), 0, st, a, t); break;
  }
}
template <int MODE>
void launch_confidence(hipStream_t st, int grid, const DetectArgs& a, const Rec* out, const Summary* sum, int cap, float* ratio) {
  hipLaunchKernelGGL((k_confidence<MODE>), dim3(grid), dim3(kThreads), 0, st, a, out, sum, cap, ratio);
}

// adsb_wait_for_event: the caller's pending events are waited for by `st`, the stream the next call's first operation runs on
int apply_ext(adsb_ctx* c, hipStream_t st) {
  for (int i = 0; i < c->n_ext; ++i) HIPCHK(c, hipStreamWaitEvent(st, c->ext_ev[i], 0));
  c->n_ext = 0;
  return 0;
}

// Everything after k_detect (and after k_longrun on the rare second pass): order, gate, compact, records.
int enqueue_tail(adsb_ctx* c, Slot& s) {
  const DetectArgs& a = s.args;
  const Plan& pl = s.plan;
  Misc* misc = (Misc*)s.d_misc.p;
  hipStream_t ts = s.cs;
  if (s.direct) {
    ts = s.ds;
    // a small pass (few lists, at most kDirectRecs centres): the whole tail in one workgroup and one launch, on the
    // compute stream right behind its k_detect (nothing to overlap: the pass is a few microseconds of GPU time) -- or,
    // when k_detect itself is a single workgroup of |IQ|^2 float input (a GNU Radio work() call), the whole pass in ONE
    // launch (s.fused: k_detect was not launched)
    TailArgs t;
    t.cands = a.cands; t.recs = a.recs; t.blk_count = a.blk_count; t.blk_lastp = a.blk_lastp; t.blk_flags = a.blk_flags;
    t.blk_off = (int*)s.d_blk_off.p; t.nblk = s.nlists; t.rec_cap = s.rec_cap; t.long_count = a.long_count;
    t.long_lastp = a.long_lastp; t.sorted = (unsigned long long*)s.d_sorted.p; t.sorted_src = (unsigned*)s.d_sorted_src.p;
    t.seg_count = (int*)s.d_seg.p; t.sum = &misc->sum; t.host_sum = s.h_sum; t.out = (Rec*)s.h_out; t.out_cap = (int)s.tot;
    t.gate_on = pl.gate ? 1 : 0; t.head_n = pl.head_n; t.gate = 63ll * c->sps;
    t.gate_long = (long long)(pl.long_aware ? 119 : 63) * c->sps; t.prev_eob = pl.prev_eob_stream - pl.origin;
    s.seq = s.seq >= 0x7FFFFFF0 ? 1 : s.seq + 1;
    t.seq = s.seq;
    if (s.fused) launch_pass_small(ts, a, t);
    else ADSB_BY_MODE(pl.mode, launch_tail_small, ts, a, t);
    // the host polls the pass number in the pinned summary (finish): no completion event on the stream
    s.polled = true;
    return 0;
  }
  if (s.ds != s.cs) {
    // k_detect ran on a stream it shares with the k_detect launches of other passes: the tail, on its own stream, is ordered
    // after this pass's k_detect only -- the next k_detect can start while it runs.
    // With timing on, the event that closes k_detect's bracket doubles as the dependency.
    hipEvent_t dep = s.ev1_valid ? s.ev1 : s.det_done;
    if (!s.ev1_valid) HIPCHK(c, hipEventRecord(dep, s.ds));
    HIPCHK(c, hipStreamWaitEvent(s.cs, dep, 0));
  }
  // Every tail kernel is small enough to run on a CU BESIDE five resident k_detect workgroups (tests/test_abi.py holds
  // the limits).  Whether it should is a choice: beside the next pass's k_detect the tail finishes ~0.25 ms after its own
  // k_detect (results one pass earlier, two passes in flight suffice) but costs that k_detect 1-2 %; kept out -- k_order,
  // the first kernel of the chain, is launched with 8 KB of unused dynamic LDS, more than five k_detect workgroups leave
  // free on a CU -- it runs when that k_detect drains.  Throughput is the default, ADSB_FLAG_LOW_LATENCY selects the other.
  // (8 KB: measured.  A padding sized to what k_detect leaves free keeps the chain out more strictly and was SLOWER on
  // every workload: int16 791 vs 1039 Gsamples/s)
  const unsigned order_pad = (c->flags & ADSB_FLAG_LOW_LATENCY) ? 0u : 8192u;
  unsigned long long* sorted = (unsigned long long*)s.d_sorted.p;
  unsigned* sorted_src = (unsigned*)s.d_sorted_src.p;
  ADSB_BY_MODE(pl.mode, launch_order, ts, (s.nlists + kOrderLists - 1) / kOrderLists, order_pad, a, s.nlists, sorted, sorted_src,
               &misc->sum, &misc->acc);
  // k_resolve / k_count / k_compact work in segments of 256 list words; a workgroup past the last segment returns at once.
  // 2048 workgroups (eight per CU) give every segment of a headline pass (≈1900) its own workgroup: one round instead of
  // four (512 workgroups: k_compact 22 us, k_resolve 11, k_count 5 for 477 k centres)
  const int ag = 2048;
  unsigned fmask = 0u, fwant = 0u;
  if (pl.gate) {
    hipLaunchKernelGGL(k_resolve, dim3(ag), dim3(kThreads), 0, ts, sorted, (const Summary*)&misc->sum,
                       (long long)63 * c->sps, (long long)(pl.long_aware ? 119 : 63) * c->sps, pl.prev_eob_stream - pl.origin);
    fmask = kKept; fwant = kKept;
  }
  hipLaunchKernelGGL(k_count, dim3(ag), dim3(kThreads), 0, ts, (const unsigned long long*)sorted,
                     (const Summary*)&misc->sum, fmask, fwant, pl.head_n, (int*)s.d_seg.p);
  // k_compact's workgroup 0 stores the summary straight into s.h_sum (pinned host memory, device-visible): visible to the
  // host once the `done` event below has completed
  hipLaunchKernelGGL(k_compact, dim3(ag), dim3(kThreads), 0, ts, (const unsigned long long*)sorted, (const Rec*)a.recs, (const unsigned*)sorted_src,
                     &misc->sum, (const int*)s.d_seg.p, fmask, fwant, pl.head_n, (Rec*)(s.direct ? s.h_out : s.d_out.p), (int)s.tot,
                     a.long_count, a.long_lastp, &misc->acc, s.h_sum, (Rec*)(s.host_cap > 0 ? s.h_out : nullptr), s.host_cap);
  HIPCHK(c, hipEventRecord(s.done, ts));
  return 0;
}

// Queue the whole device pipeline of one plan on the compute stream; nothing here waits for the GPU.
int enqueue(adsb_ctx* c, Slot& s, const Plan& pl, bool submitted) {
  HIPCHK(c, hipSetDevice(c->device));
  s.plan = pl;
  // the queue(s) of this pass (see adsb_ctx): a submitted pass on the slot's own stream; a blocking one alone on the
  // compute stream; a submitted pass on a caller-owned stream: k_detect there, the tail on the tail stream
  s.submitted = submitted;
  s.ds = s.cs = c->stream;
  if (submitted && c->split_tail) {
    if (!c->own_stream) {
      s.cs = c->tail_stream;                     // caller-owned compute stream: k_detect there, every tail on the tail stream
    } else {
      s.ds = s.cs = s.stream;
      // Three kinds of submitted passes keep their k_detect launches one behind the other on a stream of their own (the tail
      // follows on the slot's stream behind an event -- the arrangement of rounds 2-4):
      //  * timed ones (ADSB_FLAG_TIMING): the HIP events around k_detect are meant to bracket ONE launch that has the
      //    machine to itself -- two launches that overlap share the CUs and each reads twice as long;
      //  * passes over more than 4 GiB of input: a launch of over half a millisecond loses < 1 % in the hand-over to the next,
      //    and two HBM-bound launches that run side by side cost about that in DRAM locality (complex64, 2^30 samples:
      //    1.354-1.372 ms chained, 1.375-1.377 overlapped).  Everything smaller gains from the overlap: 2^28 samples +4-5 %,
      //    2^26 +20 %, 2^24 +38 %; the instruction-bound formats +2-6 % at any size (profiles/r05_ab_queue_arrangements.txt).
      //  * passes of the 8-bit formats over 1 GiB or more: their one-wavefront workgroups (adsb_device.h) leave no ragged
      //    end for the next launch to fill, and two instruction-bound launches side by side slow each other down (int8,
      //    2^30 samples: 0.568 ms in line, 0.607-0.617 overlapped; at 2^28 samples the same either way, below that the
      //    overlap wins by up to 20 %: profiles/r05_ab_8bit_workgroup_shape_and_schedule.txt).
      // No stream is added for that: the three slot streams take the three roles -- slot 0's every k_detect, slot 1's every
      // tail, slot 2's the record copies (finish) -- because the runtime multiplexes all streams of a process onto FOUR
      // hardware queues (GPU_MAX_HW_QUEUES), and a k_detect stream that shares its queue with a stream whose tail waits for
      // that k_detect stalls behind it: with a fourth stream for k_detect the timed 2^28-sample legs ran 10 % slower than
      // in round 4 (profiles/r05_pass_cost_timed_with_a_fourth_stream.txt).
      const long long in_bytes = (pl.scan_hi > 0 ? pl.scan_hi : 0) * (long long)mode_bytes(pl.mode);
      if ((c->flags & ADSB_FLAG_TIMING) || in_bytes > (4ll << 30) || (mode_is_iq8(pl.mode) && in_bytes >= (1ll << 30))) {
        s.ds = c->slot[0].stream;
        s.cs = c->slot[1].stream;
      }
    }
  }
  s.span = pl.scan_hi > 0 ? pl.scan_hi : 0;
  // A "unit" (one wavefront) walks one contiguous chunk and owns one output list.
  const int upb = det_waves(pl.mode);                      // units per k_detect workgroup: four, or one (8-bit formats)
  const int tile = kWTile;
  long long ntiles = (s.span + tile - 1) / tile;
  if (ntiles < 1) ntiles = 1;
  const int bpc = c->bpc[pl.mode];
  // chunks: one resident round of wavefronts is the floor, a bulk pass runs up to eight rounds of shorter ones (adsb_plan.h)
  long long units = 0, tiles_per = 0;
  plan_chunks(ntiles, (long long)c->n_cu * bpc * upb, &units, &tiles_per);
  const long long chunk = tiles_per * tile;
  // k_detect keeps pulse centres relative to the start of a unit's chunk in 32 bits
  if (chunk >= (1ll << 30)) return fail(c, -EINVAL, "input too long for one call on this device (chunk per wavefront >= 2^30 samples)");
  // a call of at most four units of |IQ|^2 floats may run as ONE launch of one four-wavefront workgroup (k_pass_small, below):
  // it gets four lists whatever it needs (units past `units` own nothing and report empty lists)
  const bool timing = (c->flags & ADSB_FLAG_TIMING) != 0;
  const bool can_fuse = units <= kWaves && pl.mode == ADSB_FMT_MAG2 && !timing;
  const int grid = can_fuse ? 1 : (int)((units + upb - 1) / upb);
  const int nlists = can_fuse ? kWaves : grid * upb;
  long long rc = (chunk / 256 + 64) << c->rec_cap_shift;
  if (rc > chunk / 2 + 8) rc = chunk / 2 + 8;   // there can never be more rises than that
  rc = (rc + 15) & ~15ll;                       // every list starts on a 128-byte line (k_detect's output stage writes whole lines)
  s.grid = grid; s.nlists = nlists; s.rec_cap = (int)rc; s.tot = (long long)nlists * rc; s.ntiles = ntiles; s.chunk = chunk;
  const long long long_cap = ntiles + 1;        // at most one long pulse per tile, plus the virtual rise
  int r;
  if ((r = ensure(c, s.d_cands, (size_t)s.tot * 8))) return r;
  if ((r = ensure(c, s.d_recs, (size_t)s.tot * sizeof(Rec)))) return r;
  if ((r = ensure(c, s.d_sorted, (size_t)s.tot * 8))) return r;
  if (s.tot >= (1ll << 32)) return fail(c, -EINVAL, "input too long for one call (list slots >= 2^32)");
  if ((r = ensure(c, s.d_sorted_src, (size_t)s.tot * sizeof(unsigned)))) return r;
  // A pass that can deliver only a few records (the GNU Radio work() calls: a few thousand samples) writes them from
  // k_compact straight into the pinned, device-visible result buffer: no device->host copy and no second
  // synchronisation at adsb_wait (the 48-byte summary travels the same way); bulk passes keep the DMA copy.
  const long long kDirectRecs = 16384;
  s.direct = s.tot <= kDirectRecs;
  // A MID-SIZE pass (up to 2^26 samples: tens of microseconds of k_detect, a few thousand records) gets both: k_compact
  // stores every record into d_out and the first kHostRecs of them ALSO straight into the pinned result buffer.  When the
  // pass delivers no more than that (the usual case) adsb_wait returns as soon as the tail's event has completed -- no
  // device->host copy and no second synchronisation (~12 us of a pass whose host side costs 40, tools/pass_cost.py); when
  // it delivers more, the copy runs as for a bulk pass (d_out is always complete).  Bulk passes are left alone: their
  // records are megabytes, the DMA engine moves them beside the next pass's kernels.
  const long long kMidTiles = 65536, kHostRecs = 32768;
  s.host_cap = 0;
  if (s.direct) { if ((r = ensure_pinned(c, s.h_out, s.h_out_cap, (size_t)s.tot * sizeof(Rec), true))) return r; }
  else {
    if ((r = ensure(c, s.d_out, (size_t)s.tot * sizeof(Rec)))) return r;
    if (ntiles <= kMidTiles) {
      if ((r = ensure_pinned(c, s.h_out, s.h_out_cap, (size_t)kHostRecs * sizeof(Rec), true))) return r;
      s.host_cap = (int)kHostRecs;
    }
  }
  if ((r = ensure(c, s.d_seg, (size_t)(s.tot / kThreads + 2) * sizeof(int)))) return r;
  if ((r = ensure(c, s.d_blk_count, (size_t)nlists * sizeof(int)))) return r;
  if ((r = ensure(c, s.d_blk_lastp, (size_t)nlists * sizeof(long long)))) return r;
  if ((r = ensure(c, s.d_blk_flags, (size_t)nlists * sizeof(unsigned)))) return r;
  if ((r = ensure(c, s.d_blk_off, (size_t)nlists * sizeof(int)))) return r;
  if ((r = ensure(c, s.d_long, (size_t)long_cap * sizeof(LongRise)))) return r;
  if (!s.d_misc.p) {
    if ((r = ensure(c, s.d_misc, sizeof(Misc)))) return r;
    // on the compute stream (the context's streams are non-blocking: a legacy-stream memset would not be ordered
    // before the first k_detect); afterwards k_compact re-zeroes the list head every pass
    HIPCHK(c, hipMemsetAsync(s.d_misc.p, 0, sizeof(Misc), s.ds));
  }
  Misc* misc = (Misc*)s.d_misc.p;

  DetectArgs& a = s.args;
  a.data = pl.d_data; a.n = pl.n; a.in0_base = pl.in0_base; a.scan_lo = pl.scan_lo; a.scan_hi = pl.scan_hi;
  a.fall_hi = pl.fall_hi; a.dem_hi = pl.dem_hi; a.origin = pl.origin; a.chunk = chunk; a.thr = c->thr;
  a.prev_in0 = pl.prev_in0; a.scale = c->scale[pl.mode]; a.sps = c->sps; a.end_is_call_end = pl.end_is_call_end; a.rec_cap = s.rec_cap;
  a.long_aware = pl.long_aware ? 1 : 0;
  a.long_cap = (int)long_cap; a.cands = (unsigned long long*)s.d_cands.p; a.recs = (Rec*)s.d_recs.p; a.blk_count = (int*)s.d_blk_count.p;
  a.blk_lastp = (long long*)s.d_blk_lastp.p; a.blk_flags = (unsigned*)s.d_blk_flags.p;
  a.longlist = (LongRise*)s.d_long.p; a.long_count = &misc->long_count; a.long_lastp = &misc->long_lastp;

  s.fused = s.direct && can_fuse;
  // what this pass's first kernel has to wait for: its own upload (host-fed submission), the caller's events (adsb_wait_for_event)
  if (s.h2d_pending) { HIPCHK(c, hipStreamWaitEvent(s.ds, s.h2d_done, 0)); s.h2d_pending = false; }
  { int r_ = apply_ext(c, s.ds); if (r_) return r_; }
  if (!s.fused) {
    if (timing) HIPCHK(c, hipEventRecord(s.ev0, s.ds));
    ADSB_BY_MODE(pl.mode, launch_detect, c, s.ds, a, grid);
    if (timing) HIPCHK(c, hipEventRecord(s.ev1, s.ds));
  }
  c->stats.detect_grid = (uint64_t)grid; c->stats.blocks_per_cu = (uint64_t)c->bpc[pl.mode];
  c->stats.calls++;
  s.busy = true;
  s.ev1_valid = timing;
  s.polled = false;
  return enqueue_tail(c, s);
}

// Wait for a queued call; handle the two rare outcomes that need a second pass (pulses longer than the
// LDS window; per-workgroup list overflow); bring the records to pinned host memory.
int finish(adsb_ctx* c, Slot& s, Summary* sum, int32_t* n_res) {
  // every error exit releases the slot: a failed call must not leave its ticket busy for good
#define FINCHK(call)                                                           \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) { s.busy = false; return fail(c, -EIO, #call, e_); } \
  } while (0)
  FINCHK(hipSetDevice(c->device));
  const bool timing = (c->flags & ADSB_FLAG_TIMING) != 0;
  for (int attempt = 0; attempt < 16; ++attempt) {
    if (s.polled) {
      // a one-workgroup pass: its last store is the pass number (publish_small); spin on it -- bounded: a kernel that died
      // never stores it, and the stream synchronisation below reports why
      // The spin is SHORT (the pass itself takes ~10 us; ~50 us covers a pass queued behind one or two others): a pass that
      // sits behind more work than that -- a caller-owned stream, several submissions in flight -- is waited for by
      // blocking on the stream instead of burning the calling thread's core (in GNU Radio: the framer's work() thread).
      const volatile int* seqp = &s.h_sum->pad_;
      const auto t0 = std::chrono::steady_clock::now();
      unsigned spins = 0;
      while (*seqp != s.seq) {
        cpu_relax();
        if ((++spins & 0x3Fu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(50)) break;
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      if (*seqp != s.seq) {
        FINCHK(hipStreamSynchronize(s.ds));
        FINCHK(hipGetLastError());
        std::atomic_thread_fence(std::memory_order_acquire);
        if (*seqp != s.seq) { s.busy = false; return fail(c, -EIO, "small pass finished without publishing its summary"); }
        c->stats.poll_fallbacks++;
      }
    } else {
      FINCHK(hipEventSynchronize(s.done));
      FINCHK(hipGetLastError());
    }
    if (timing) {
      float ms = 0;
      FINCHK(hipEventElapsedTime(&ms, s.ev0, s.ev1));
      c->stats.detect_launches++;
      c->stats.detect_ms += ms;
      if (c->det_hist.size() < (size_t)adsb_ctx::kHist) c->det_hist.resize(adsb_ctx::kHist);
      c->det_hist[c->det_hist_n++ % adsb_ctx::kHist] = ms;
      // idle time on the compute stream between the previous pass's k_detect and this one (pipelined use)
      // idle time on the compute stream between this pass's k_detect and the NEXT pass's (already queued in
      // pipelined use; its events are valid if that pass has been collected or is complete -- best effort)
      const int me = (int)(&s - c->slot);
      Slot& nxt = c->slot[(me + 1) % ADSB_MAX_IN_FLIGHT];
      if (nxt.busy && nxt.ev1_valid) {
        float gap = 0;
        if (hipEventElapsedTime(&gap, s.ev1, nxt.ev0) == hipSuccess && gap >= 0 && gap < 100.0f) {
          c->stats.detect_gap_ms += gap; c->stats.detect_gaps++;
        } else (void)hipGetLastError();
      }
      c->stats.detect_samples += (uint64_t)s.span;
      c->stats.detect_bytes += (uint64_t)s.span * (uint64_t)mode_bytes(s.plan.mode);
    }
    if (s.h_sum->overflow) {
      if ((long long)s.rec_cap >= s.chunk / 2 + 8) { s.busy = false; return fail(c, -EIO, "centre list overflow at maximum size"); }
      c->rec_cap_shift++;
      c->stats.retries++;
      int r = enqueue(c, s, s.plan, s.submitted);
      if (r) { s.busy = false; return r; }
      c->stats.calls--;
      continue;
    }
    if (s.h_sum->long_count > 0) { c->stats.longrun_calls++; c->stats.longrun_pulses += (uint64_t)s.h_sum->long_count; }
    if (s.h_sum->long_count > s.args.long_cap) { s.busy = false; return fail(c, -EIO, "long-rise list overflow"); }
    *sum = *s.h_sum;
    const int nres = sum->n_kept;
    // the records are already in h_out: a one-workgroup pass (all of them), or a mid-size pass that delivered no more than
    // k_compact stored there beside d_out (enqueue)
    const bool in_host = (s.direct || nres <= s.host_cap) && s.h_out != nullptr;
    int r = in_host ? 0 : ensure_pinned(c, s.h_out, s.h_out_cap, (size_t)(nres > 0 ? nres : 1) * sizeof(Rec), true);
    if (r) { s.busy = false; return r; }
    const Rec* recs_dev = (const Rec*)(s.direct ? s.h_out : s.d_out.p);
    if (nres > 0 && (!in_host || (c->flags & ADSB_FLAG_CONFIDENCE))) {
      // on the pass's own stream (idle: its last kernel has completed); never on a caller-owned one
      // (a submitted pass that shares its stream with the passes behind it -- kernels in line, or ADSB_FLAG_SINGLE_STREAM --
      // copies on the context's record-copy stream: on its own one the copy would wait for everything queued since)
      // Slot 2's stream is free for that in a context whose passes are all in line (timed contexts; bulk 8-bit / > 4 GiB
      // streams) -- and it keeps the context at three busy streams, one hardware queue each (a fourth busy stream shares a
      // queue with one of the three and serialises behind it: the 8-bit legs ran 6 % slower with a copy stream of their own,
      // profiles/r06_ab_record_copy_stream.txt).  Only when slot 2 itself holds an overlapped pass (a context that mixes in-line
      // and overlapped passes) would the copy of an older pass wait for that younger one: then a copy stream of its own.
      const bool shared = s.ds != s.cs || (s.submitted && !c->split_tail);
      const Slot& s2 = c->slot[2];
      const bool slot2_taken = &s2 != &s && s2.busy && s2.ds == s2.stream;
      if (c->own_stream && shared && slot2_taken && !c->d2h_stream) FINCHK(hipStreamCreateWithFlags(&c->d2h_stream, hipStreamNonBlocking));
      const hipStream_t xs = !c->own_stream ? c->copy_stream : (shared ? (slot2_taken ? c->d2h_stream : c->slot[2].stream) : s.cs);
      if (!in_host)
        FINCHK(hipMemcpyAsync(s.h_out, s.d_out.p, (size_t)nres * sizeof(Rec), hipMemcpyDeviceToHost, xs));
      if (c->flags & ADSB_FLAG_CONFIDENCE) {
        // opt-in (demod.py:97-101): bit1/bit0 ratios of the delivered records, computed now that their number is
        // known -- one more small kernel and copy on the copy stream, paid only by callers who ask for it
        const size_t rb = (size_t)nres * 112 * sizeof(float);
        if ((r = ensure(c, s.d_ratio, rb)) || (r = ensure_pinned(c, s.h_ratio, s.h_ratio_cap, rb))) { s.busy = false; return r; }
        FINCHK(hipMemsetAsync(s.d_ratio.p, 0, rb, xs));
        int cg = (nres + kWaves - 1) / kWaves;
        if (cg > c->n_cu * 8) cg = c->n_cu * 8;
        ADSB_BY_MODE(s.plan.mode, launch_confidence, xs, cg, s.args, recs_dev,
                     (const Summary*)&((Misc*)s.d_misc.p)->sum, nres, (float*)s.d_ratio.p);
        FINCHK(hipMemcpyAsync(s.h_ratio, s.d_ratio.p, rb, hipMemcpyDeviceToHost, xs));
      }
      FINCHK(hipStreamSynchronize(xs));
      FINCHK(hipGetLastError());
    }
    s.nres = nres;
    *n_res = nres;
    s.busy = false;
    return 0;
  }
  s.busy = false;
  return fail(c, -EIO, "centre list capacity did not converge");
#undef FINCHK
}

// Synchronous form used by every blocking entry point.
int run_pipeline(adsb_ctx* c, const Plan& pl, Summary* sum, int32_t* n_res) {
  for (const Slot& sl : c->slot) if (sl.busy) return fail(c, -EBUSY, "a submitted call is still pending (adsb_wait first)");
  Slot& s = c->slot[0];
  c->last_slot = 0;
  int r = enqueue(c, s, pl, false);
  if (r) { s.busy = false; return r; }
  return finish(c, s, sum, n_res);
}

int deliver(adsb_ctx* c, int32_t nres, adsb_burst* out, int32_t cap, int32_t* n_out) {
  if (n_out) *n_out = nres;
  if (out) {
    if (nres > cap) return fail(c, -ENOSPC, "output array too small");
    if (nres > 0) memcpy(out, c->slot[c->last_slot].h_out, (size_t)nres * sizeof(adsb_burst));
  }
  return 0;
}

int canonical(adsb_ctx* c, int mode, const void* d_data, int64_t n, int64_t abs_offset, adsb_burst* out,
              int32_t cap, int32_t* n_out) {
  if (!c || n < 0) return -EINVAL;
  if (((uintptr_t)d_data & 15u) != 0) return fail(c, -EINVAL, "device pointer must be 16-byte aligned");
  Plan pl = plan_canonical(mode, d_data, n, abs_offset, c->sps);
  pl.long_aware = (c->flags & ADSB_FLAG_LONG_AWARE_GATE) != 0;
  Summary s;
  int32_t nres = 0;
  if (n == 0) { c->slot[c->last_slot].nres = 0; if (n_out) *n_out = 0; return 0; }
  int rc = run_pipeline(c, pl, &s, &nres);
  if (rc) return rc;
  return deliver(c, nres, out, cap, n_out);
}

bool is_pinned_host(const void* p) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // pageable
  return at.type == hipMemoryTypeHost;
}

// The address under which the DEVICE sees a page-locked host buffer.  For hipHostMalloc'ed memory it equals the host
// address on ROCm; for memory page-locked in place (adsb_host_register -> hipHostRegister) HIP only promises access
// through the alias hipHostGetDevicePointer returns.  nullptr: not mapped -- the caller stages instead.
const void* device_alias(const void* host) {
  void* dev = nullptr;
  if (hipHostGetDevicePointer(&dev, const_cast<void*>(host), 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return dev;
}

int ensure_pool(adsb_ctx* c) {
  if (c->pool) return 0;
  c->pool = new (std::nothrow) CopyPool();
  if (!c->pool) return fail(c, -ENOMEM, "copy pool");
  const unsigned hw = std::thread::hardware_concurrency();
  const int nt = c->copy_threads >= 0 ? c->copy_threads : (hw >= 16 ? 5 : (hw >= 4 ? 2 : 0));     // workers besides the caller
  if (c->have_local_cpus && !(c->flags & ADSB_FLAG_NO_NUMA_BINDING)) {
    // the GPU's local cpus, but never outside the mask the PROCESS was given (taskset, numactl --physcpubind, a cgroup):
    // the workers run on the intersection, or stay unbound when that is empty
    cpu_set_t mine, both;
    CPU_ZERO(&mine);
    if (sched_getaffinity(0, sizeof(mine), &mine) == 0) {
      CPU_AND(&both, &mine, &c->local_cpus);
      if (CPU_COUNT(&both) > 0) { c->pool->cpus = both; c->pool->have_cpus = true; }
    }
  }
  c->pool->start(nt);
  return 0;
}

// Pageable host memory -> device memory on `stream` through the ring of pinned chunks: the host copy of chunk k+1 (split
// over the context's copy threads) runs beside the DMA of chunk k.  Returns once the last DMA is QUEUED.
int staged_copy(adsb_ctx* c, void* d_dst, const void* host, size_t bytes, hipStream_t stream) {
  constexpr size_t kRingChunk = (size_t)16 << 20;
  for (void*& r : c->h_ring)
    if (!r) HIPCHK(c, host_alloc_near(c, &r, kRingChunk));      // the staging ring: on the GPU's NUMA node
  int rc = ensure_pool(c);
  if (rc) return rc;
  // the host copy of piece k+1 runs beside the DMA of piece k: a source of a few megabytes (a GNU Radio work() call of a
  // megasample) is cut into at least four pieces, a bulk one into whole 16 MiB ring chunks
  size_t kChunk = kRingChunk;
  while (kChunk > ((size_t)1 << 20) && bytes < 4 * kChunk) kChunk >>= 1;
  for (size_t off = 0; off < bytes; off += kChunk) {
    const size_t m = bytes - off < kChunk ? bytes - off : kChunk;
    const int b = (int)(c->ring_k++ % (unsigned)adsb_ctx::kRing);
    if (c->ring_used[b]) HIPCHK(c, hipEventSynchronize(c->ring_done[b]));      // the chunk's previous DMA has read it
    c->pool->copy((char*)c->h_ring[b], (const char*)host + off, m);
    HIPCHK(c, hipMemcpyAsync((char*)d_dst + off, c->h_ring[b], m, hipMemcpyHostToDevice, stream));
    HIPCHK(c, hipEventRecord(c->ring_done[b], stream));
    c->ring_used[b] = true;
  }
  return 0;
}

// Host buffer -> something the kernels can read, for the blocking entry points (the buffer only has to stay valid until
// the call returns).  Large inputs are copied to the device: page-locked sources (adsb_host_alloc, hipHostMalloc, torch
// pin_memory) go straight over PCIe, pageable ones through the context's pinned staging buffer.  Small inputs (the GNU
// Radio work() calls: a few thousand samples) are not copied to the device at all: the kernels read the page-locked
// copy in place over PCIe, once -- one operation fewer in a call whose cost is operations, not bytes.
int upload(adsb_ctx* c, const void* host, size_t bytes, void** d_out) {
  const size_t kZeroCopyBytes = (size_t)256 << 10;
  int rc;
  if ((rc = apply_ext(c, c->stream))) return rc;
  const void* src = host;
  const bool pinned = is_pinned_host(host);
  const bool small = bytes <= kZeroCopyBytes;
  const void* alias = (pinned && small) ? device_alias(host) : nullptr;     // only the in-place path needs it
  if (!small && !pinned && bytes >= ((size_t)4 << 20)) {
    // multi-megabyte pageable input (a GNU Radio block run with large chunks): chunked through the pinned ring
    if ((rc = ensure(c, c->d_in, bytes + 64))) return rc;
    if ((rc = staged_copy(c, c->d_in.p, host, bytes, c->stream))) return rc;
    *d_out = c->d_in.p;
    return 0;
  }
  if (!pinned || (small && (!alias || ((uintptr_t)alias & 15u) != 0))) {
    if ((rc = ensure_pinned(c, c->h_stage, c->h_stage_cap, bytes))) return rc;
    memcpy(c->h_stage, host, bytes);
    src = c->h_stage;
    alias = c->h_stage;                                   // hipHostMalloc'ed: one address on both sides
  }
  if (small) {
    *d_out = const_cast<void*>(alias);
    return 0;
  }
  if ((rc = ensure(c, c->d_in, bytes + 64))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->d_in.p, src, bytes, hipMemcpyHostToDevice, c->stream));
  *d_out = c->d_in.p;
  return 0;
}

}  // namespace

extern "C" {

int adsb_abi_version(void) { return ADSB_ABI_VERSION; }

int32_t adsb_plan_chunks(int64_t n_samples, int64_t resident_wavefronts, int64_t* units, int64_t* samples_per_chunk) {
  if (n_samples < 0 || resident_wavefronts < 1 || !units || !samples_per_chunk) return -EINVAL;
  long long u = 0, t = 0;
  plan_chunks((n_samples + kWTile - 1) / kWTile, resident_wavefronts, &u, &t);
  *units = u; *samples_per_chunk = t * kWTile;
  return 0;
}

uint32_t adsb_mode_s_syndrome(const uint8_t bits[14], int32_t* df_out, int32_t* nbits_out) {
  // decoder.py:551 (DF), :565,604,636,669 (format sets), :693-714 (compute_crc); same table as the device
  static constexpr CrcTab tab = make_crc_tab();
  const unsigned df = bits[0] >> 3, dfb = 1u << df;
  const bool lng = (dfb & kDfLongSet) != 0, known = lng || (dfb & kDfShortSet) != 0;
  const int L = lng ? 112 : 56;
  uint32_t syn = 0;
  for (int i = 0; i < L; ++i)
    if ((bits[i >> 3] >> (7 - (i & 7))) & 1) syn ^= tab.r[L - 1 - i];
  if (df_out) *df_out = (int32_t)df;
  if (nbits_out) *nbits_out = known ? L : 0;
  return syn;
}

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
  if (sps < 2 || (sps & 1) || sps > ADSB_MAX_SPS) return -EINVAL;   // odd sps crashes the reference's work(); above the maximum: untested
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return -ENODEV;   // no CPU fallback, by design
  if (device < 0 || device >= ndev) return -ENODEV;
  adsb_ctx* c = new (std::nothrow) adsb_ctx();
  if (!c) return -ENOMEM;
  c->device = device; c->fs = fs; c->sps = (int)sps; c->thr = threshold; c->flags = flags;
  if (hipSetDevice(device) != hipSuccess) { delete c; return -ENODEV; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
  probe_numa(c);
  if (flags & ADSB_FLAG_NO_NUMA_BINDING) c->numa_node = -1;    // (the cpu list stays readable through adsb_numa_info)
  {
    int nb;
    if ((nb = detect_occupancy<0>(c->det_dyn_lds[0])) > 0) c->bpc[0] = nb;
    if ((nb = detect_occupancy<1>(c->det_dyn_lds[1])) > 0) c->bpc[1] = nb;
    if ((nb = detect_occupancy<2>(c->det_dyn_lds[2])) > 0) c->bpc[2] = nb;
    if ((nb = detect_occupancy<3>(c->det_dyn_lds[3])) > 0) c->bpc[3] = nb;
    if ((nb = detect_occupancy<4>(c->det_dyn_lds[4])) > 0) c->bpc[4] = nb;
    const unsigned st[ADSB_FMT_COUNT] = {detect_static_lds<0>(), detect_static_lds<1>(), detect_static_lds<2>(),
                                         detect_static_lds<3>(), detect_static_lds<4>()};
    for (int m = 0; m < ADSB_FMT_COUNT; ++m) {
      const unsigned per = (st[m] + c->det_dyn_lds[m] + 1279u) / 1280u * 1280u;             // LDS allocation granule on gfx950
      const unsigned used = per * (unsigned)c->bpc[m];
      c->lds_beside[m] = used < 163840u ? 163840u - used : 0u;
    }
  }
  c->own_stream = true;
  // submitted passes overlap on the slots' streams unless the caller opts out (ADSB_FLAG_SINGLE_STREAM: everything on ONE stream)
  c->split_tail = (flags & ADSB_FLAG_SINGLE_STREAM) == 0;
  for (hipEvent_t& e : c->ring_done)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { adsb_destroy(c); return -EIO; }
  for (Slot& sl : c->slot) {
    if (host_alloc_near(c, (void**)&sl.h_sum, sizeof(Summary), true) != hipSuccess) { adsb_destroy(c); return -ENOMEM; }
    memset(sl.h_sum, 0, sizeof(Summary));                        // (pad_ is the pass number finish() polls: starts at zero)
    if (hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking) != hipSuccess) { adsb_destroy(c); return -EIO; }
    if (hipEventCreate(&sl.ev0) != hipSuccess || hipEventCreate(&sl.ev1) != hipSuccess ||
        hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sl.det_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sl.h2d_done, hipEventDisableTiming) != hipSuccess) { adsb_destroy(c); return -EIO; }
  }
  c->stream = c->slot[0].stream;         // blocking calls always run in slot 0 (run_pipeline)
  if (hipStreamCreateWithFlags(&c->h2d_stream, hipStreamNonBlocking) != hipSuccess) { adsb_destroy(c); return -EIO; }
  *out = c;
  return 0;
}

void adsb_destroy(adsb_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (!c->own_stream && c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
  if (c->h2d_stream) (void)hipStreamSynchronize(c->h2d_stream);
  if (c->d2h_stream) (void)hipStreamSynchronize(c->d2h_stream);
  if (c->tail_stream) (void)hipStreamSynchronize(c->tail_stream);
  for (Slot& sl : c->slot) if (sl.stream) (void)hipStreamSynchronize(sl.stream);
  DevBuf* bufs[] = {&c->d_in};
  for (DevBuf* b : bufs) if (b->p) (void)hipFree(b->p);
  for (Slot& sl : c->slot) {
    DevBuf* sb[] = {&sl.d_cands, &sl.d_recs, &sl.d_sorted, &sl.d_sorted_src, &sl.d_out, &sl.d_seg, &sl.d_blk_count,
                    &sl.d_blk_lastp, &sl.d_blk_flags, &sl.d_blk_off, &sl.d_long, &sl.d_misc, &sl.d_in, &sl.d_ratio};
    for (DevBuf* b : sb) if (b->p) (void)hipFree(b->p);
    if (sl.h_sum) (void)hipHostFree(sl.h_sum);
    if (sl.h_out) (void)hipHostFree(sl.h_out);
    if (sl.h_ratio) (void)hipHostFree(sl.h_ratio);
    if (sl.h2d_done) (void)hipEventDestroy(sl.h2d_done);
    if (sl.ev0) (void)hipEventDestroy(sl.ev0);
    if (sl.ev1) (void)hipEventDestroy(sl.ev1);
    if (sl.done) (void)hipEventDestroy(sl.done);
    if (sl.det_done) (void)hipEventDestroy(sl.det_done);
    if (sl.stream) (void)hipStreamDestroy(sl.stream);
  }
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->h_dm) (void)hipHostFree(c->h_dm);
  for (void* r : c->h_ring) if (r) (void)hipHostFree(r);
  delete c->pool;
  for (hipEvent_t e : c->ring_done) if (e) (void)hipEventDestroy(e);
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->h2d_stream) (void)hipStreamDestroy(c->h2d_stream);
  if (c->d2h_stream) (void)hipStreamDestroy(c->d2h_stream);
  if (c->tail_stream) (void)hipStreamDestroy(c->tail_stream);
  delete c;
}

int adsb_set_threshold(adsb_ctx* c, float threshold) {
  if (!c) return -EINVAL;
  c->thr = threshold;
  return 0;
}

int adsb_set_stream(adsb_ctx* c, void* hip_stream) {
  if (!c) return -EINVAL;
  for (const Slot& sl : c->slot) if (sl.busy) return fail(c, -EBUSY, "calls pending");
  if (c->own_stream && c->stream) (void)hipStreamSynchronize(c->stream);      // (slot 0's: it stays the slot's)
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  if (!c->tail_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->tail_stream, hipStreamNonBlocking));
  c->stream = (hipStream_t)hip_stream;
  c->own_stream = false;
  return 0;
}

int adsb_set_copy_threads(adsb_ctx* c, int32_t threads) {
  if (!c || threads < 1 || threads > 64) return -EINVAL;
  if (c->pool) return fail(c, -EBUSY, "copy threads already running (set before the first pageable submission)");
  c->copy_threads = threads - 1;                            // the calling thread is one of them
  return 0;
}

int adsb_numa_info(adsb_ctx* c, int32_t* node, char* cpulist, size_t cap, char* pci_bdf, size_t bdf_cap) {
  if (!c) return -EINVAL;
  if (node) *node = c->numa_node;
  if (cpulist && cap) snprintf(cpulist, cap, "%s", c->cpulist);
  if (pci_bdf && bdf_cap) snprintf(pci_bdf, bdf_cap, "%s", c->pci_bdf);
  return 0;
}

int adsb_host_alloc_near(adsb_ctx* c, void** p, size_t bytes) {
  if (!c || !p || bytes == 0) return -EINVAL;
  *p = nullptr;
  return host_alloc_near(c, p, bytes) == hipSuccess ? 0 : -ENOMEM;
}

int adsb_host_copy(adsb_ctx* c, void* dst, const void* src, size_t bytes) {
  if (!c || (bytes > 0 && (!dst || !src))) return -EINVAL;
  int rc = ensure_pool(c);
  if (rc) return rc;
  c->pool->copy((char*)dst, (const char*)src, bytes);
  return 0;
}

int adsb_wait_for_event(adsb_ctx* c, void* hip_event) {
  if (!c || !hip_event) return -EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  // The NEXT call -- blocking, submitted or host-fed -- runs after the event: the wait is queued with that call, on the stream
  // its first operation runs on (which one is decided there; a host-fed submission: the upload stream).  Nothing is queued
  // now: a stream that is never used never takes one of the process's four hardware queues.  The event has to stay alive
  // until that call; a later submission that depends on the same producer asks again (FrontEnd does, per tensor call).
  if (c->n_ext == adsb_ctx::kMaxExt) {           // more producers than remembered: the oldest is waited for by every queue now
    for (Slot& sl : c->slot) HIPCHK(c, hipStreamWaitEvent(sl.stream, c->ext_ev[0], 0));
    if (!c->own_stream) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ext_ev[0], 0));
    HIPCHK(c, hipStreamWaitEvent(c->h2d_stream, c->ext_ev[0], 0));
    for (int i = 1; i < c->n_ext; ++i) c->ext_ev[i - 1] = c->ext_ev[i];
    --c->n_ext;
  }
  c->ext_ev[c->n_ext++] = (hipEvent_t)hip_event;
  return 0;
}

int adsb_clear_pending_events(adsb_ctx* c) {
  if (!c) return -EINVAL;
  c->n_ext = 0;
  return 0;
}

int adsb_reset(adsb_ctx* c) {
  if (!c) return -EINVAL;
  c->st = FramerState();
  c->n_ext = 0;           // a fresh stream starts without remembered producers
  return 0;
}

int adsb_framer_state(adsb_ctx* c, float* prev_in0, int64_t* prev_eob_idx) {
  if (!c) return -EINVAL;
  if (prev_in0) *prev_in0 = c->st.prev_in0;
  if (prev_eob_idx) *prev_eob_idx = (int64_t)c->st.prev_eob;
  return 0;
}

int adsb_process_iq_device(adsb_ctx* c, const void* d_iq, int64_t n, int64_t abs_offset, adsb_burst* out,
                           int32_t cap, int32_t* n_out) {
  return canonical(c, 0, d_iq, n, abs_offset, out, cap, n_out);
}

int adsb_set_iq16_scale(adsb_ctx* c, float scale) { return adsb_set_format_scale(c, ADSB_FMT_SC16, scale); }

int adsb_set_format_scale(adsb_ctx* c, int format, float scale) {
  if (!c || format < ADSB_FMT_SC16 || format >= ADSB_FMT_COUNT) return -EINVAL;
  c->scale[format] = scale;
  return 0;
}

int adsb_process_format_device(adsb_ctx* c, int format, const void* d_data, int64_t n, int64_t abs_offset,
                               adsb_burst* out, int32_t cap, int32_t* n_out) {
  if (!c || format < 0 || format >= ADSB_FMT_COUNT) return -EINVAL;
  return canonical(c, format, d_data, n, abs_offset, out, cap, n_out);
}

int adsb_process_format(adsb_ctx* c, int format, const void* host, int64_t n, int64_t abs_offset, adsb_burst* out,
                        int32_t cap, int32_t* n_out) {
  if (!c || format < 0 || format >= ADSB_FMT_COUNT || n < 0 || (n > 0 && !host)) return -EINVAL;
  if (n == 0) { if (n_out) *n_out = 0; c->slot[c->last_slot].nres = 0; return 0; }
  HIPCHK(c, hipSetDevice(c->device));
  void* d = nullptr;
  int rc = upload(c, host, (size_t)n * (size_t)mode_bytes(format), &d);
  if (rc) return rc;
  return canonical(c, format, d, n, abs_offset, out, cap, n_out);
}

int adsb_process_iq16_device(adsb_ctx* c, const void* d_iq16, int64_t n, int64_t abs_offset, adsb_burst* out,
                             int32_t cap, int32_t* n_out) {
  return canonical(c, 2, d_iq16, n, abs_offset, out, cap, n_out);
}

int adsb_process_iq16(adsb_ctx* c, const int16_t* iq16_host, int64_t n, int64_t abs_offset, adsb_burst* out,
                      int32_t cap, int32_t* n_out) {
  if (!c || n < 0 || (n > 0 && !iq16_host)) return -EINVAL;
  if (n == 0) { if (n_out) *n_out = 0; c->slot[c->last_slot].nres = 0; return 0; }
  HIPCHK(c, hipSetDevice(c->device));
  void* d = nullptr;
  int rc = upload(c, iq16_host, (size_t)n * 4, &d);
  if (rc) return rc;
  return canonical(c, 2, d, n, abs_offset, out, cap, n_out);
}

int adsb_process_mag2_device(adsb_ctx* c, const void* d_mag2, int64_t n, int64_t abs_offset, adsb_burst* out,
                             int32_t cap, int32_t* n_out) {
  return canonical(c, 1, d_mag2, n, abs_offset, out, cap, n_out);
}

int adsb_process_iq(adsb_ctx* c, const float* iq_host, int64_t n, int64_t abs_offset, adsb_burst* out,
                    int32_t cap, int32_t* n_out) {
  if (!c || n < 0 || (n > 0 && !iq_host)) return -EINVAL;
  if (n == 0) { if (n_out) *n_out = 0; c->slot[c->last_slot].nres = 0; return 0; }
  HIPCHK(c, hipSetDevice(c->device));
  void* d = nullptr;
  int rc = upload(c, iq_host, (size_t)n * 8, &d);
  if (rc) return rc;
  return canonical(c, 0, d, n, abs_offset, out, cap, n_out);
}

int adsb_process_mag2(adsb_ctx* c, const float* mag2_host, int64_t n, int64_t abs_offset, adsb_burst* out,
                      int32_t cap, int32_t* n_out) {
  if (!c || n < 0 || (n > 0 && !mag2_host)) return -EINVAL;
  if (n == 0) { if (n_out) *n_out = 0; c->slot[c->last_slot].nres = 0; return 0; }
  HIPCHK(c, hipSetDevice(c->device));
  void* d = nullptr;
  int rc = upload(c, mag2_host, (size_t)n * 4, &d);
  if (rc) return rc;
  return canonical(c, 1, d, n, abs_offset, out, cap, n_out);
}

static int shard_post(adsb_ctx* c, Slot& s, const Summary& sum, int32_t* nres_io, bool drop_overlong = false);

static int submit_canonical(adsb_ctx* c, int mode, const void* d_data, int64_t n, int64_t abs_offset, int32_t* ticket) {
  if (!c || n < 1 || !ticket) return -EINVAL;
  if (((uintptr_t)d_data & 15u) != 0) return fail(c, -EINVAL, "device pointer must be 16-byte aligned");
  Slot& s = c->slot[c->next_slot];
  if (s.busy) return fail(c, -EBUSY, "every pipeline slot is in flight (adsb_wait first)");
  Plan pl = plan_canonical(mode, d_data, n, abs_offset, c->sps);
  pl.long_aware = (c->flags & ADSB_FLAG_LONG_AWARE_GATE) != 0;
  int r = enqueue(c, s, pl, true);
  if (r) { s.busy = false; return r; }
  s.is_shard = false;
  *ticket = c->next_slot;
  c->next_slot = (c->next_slot + 1) % ADSB_MAX_IN_FLIGHT;
  return 0;
}

// Host buffer -> the slot's own device input buffer on the upload stream; only this slot's k_detect waits for it.
// Page-locked sources are DMA'd where they lie; pageable ones go through a ring of four pinned chunks: the host copy of
// chunk k+1 (split over the context's copy threads) runs beside the DMA of chunk k.
static int upload_async(adsb_ctx* c, Slot& s, const void* host, size_t bytes) {
  int rc;
  if ((rc = apply_ext(c, c->h2d_stream))) return rc;     // (the kernels come behind the upload: they need no wait of their own)
  if ((rc = ensure(c, s.d_in, bytes + 64))) return rc;
  if (is_pinned_host(host)) {
    HIPCHK(c, hipMemcpyAsync(s.d_in.p, host, bytes, hipMemcpyHostToDevice, c->h2d_stream));
  } else if ((rc = staged_copy(c, s.d_in.p, host, bytes, c->h2d_stream))) {
    return rc;
  }
  HIPCHK(c, hipEventRecord(s.h2d_done, c->h2d_stream));
  s.h2d_pending = true;          // enqueue() makes the stream this pass's k_detect runs on wait for it
  return 0;
}

int adsb_submit_format_host(adsb_ctx* c, int format, const void* host, int64_t n, int64_t abs_offset, int32_t* ticket) {
  if (!c || format < 0 || format >= ADSB_FMT_COUNT || n < 1 || !host || !ticket) return -EINVAL;
  Slot& s = c->slot[c->next_slot];
  if (s.busy) return fail(c, -EBUSY, "every pipeline slot is in flight (adsb_wait first)");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = upload_async(c, s, host, (size_t)n * (size_t)mode_bytes(format));
  if (rc) return rc;
  return submit_canonical(c, format, s.d_in.p, n, abs_offset, ticket);
}

int adsb_last_confidence(adsb_ctx* c, const float** ratio, int32_t* n) {
  if (!c) return -EINVAL;
  if (!(c->flags & ADSB_FLAG_CONFIDENCE)) return fail(c, -EINVAL, "context created without ADSB_FLAG_CONFIDENCE");
  const Slot& s = c->slot[c->last_slot];
  if (ratio) *ratio = s.nres > 0 ? (const float*)s.h_ratio : nullptr;
  if (n) *n = s.nres;
  return 0;
}

int adsb_submit_format_device(adsb_ctx* c, int format, const void* d_data, int64_t n, int64_t abs_offset, int32_t* ticket) {
  if (!c || format < 0 || format >= ADSB_FMT_COUNT) return -EINVAL;
  return submit_canonical(c, format, d_data, n, abs_offset, ticket);
}

int adsb_submit_iq_device(adsb_ctx* c, const void* d_iq, int64_t n, int64_t abs_offset, int32_t* ticket) {
  return submit_canonical(c, 0, d_iq, n, abs_offset, ticket);
}

int adsb_submit_iq16_device(adsb_ctx* c, const void* d_iq16, int64_t n, int64_t abs_offset, int32_t* ticket) {
  return submit_canonical(c, 2, d_iq16, n, abs_offset, ticket);
}

int adsb_submit_mag2_device(adsb_ctx* c, const void* d_mag2, int64_t n, int64_t abs_offset, int32_t* ticket) {
  return submit_canonical(c, 1, d_mag2, n, abs_offset, ticket);
}

int adsb_wait(adsb_ctx* c, int32_t ticket, adsb_burst* out, int32_t cap, int32_t* n_out) {
  if (!c || ticket < 0 || ticket >= ADSB_MAX_IN_FLIGHT) return -EINVAL;
  Slot& s = c->slot[ticket];
  if (!s.busy) return fail(c, -EINVAL, "no call pending on this ticket");
  Summary sum;
  int32_t nres = 0;
  int r = finish(c, s, &sum, &nres);
  if (r) return r;
  c->last_slot = ticket;
  if (s.is_shard && (r = shard_post(c, s, sum, &nres))) return r;
  return deliver(c, nres, out, cap, n_out);
}

int adsb_last_result(adsb_ctx* c, const adsb_burst** bursts, int32_t* n) {
  if (!c) return -EINVAL;
  if (bursts) *bursts = (const adsb_burst*)c->slot[c->last_slot].h_out;
  if (n) *n = c->slot[c->last_slot].nres;
  return 0;
}

int adsb_framer_work(adsb_ctx* c, const float* in0, int64_t n_in0, int64_t N, int64_t nitems_written,
                     adsb_burst* tags, int32_t cap, int32_t* n_out) {
  return adsb_framer_work_passthrough(c, in0, n_in0, N, nitems_written, nullptr, tags, cap, n_out);
}

int adsb_framer_work_passthrough(adsb_ctx* c, const float* in0, int64_t n_in0, int64_t N, int64_t nitems_written, float* out0,
                                 adsb_burst* tags, int32_t cap, int32_t* n_out) {
  if (!c || !in0 || N < 1) return -EINVAL;
  const long long H = 8ll * c->sps;
  if (n_in0 != N + H - 1) return fail(c, -EINVAL, "framer input must hold N + 8*sps - 1 items");
  HIPCHK(c, hipSetDevice(c->device));
  void* d = nullptr;
  int rc = upload(c, in0, (size_t)n_in0 * 4, &d);
  if (rc) return rc;
  Plan pl = plan_framer_work(d, n_in0, N, nitems_written, c->sps, c->st);
  if (c->flags & ADSB_FLAG_FRAMER_SLICES) pl.dem_hi = n_in0;   // bursts that end inside this call's input get their bits
  Summary s;
  int32_t nres = 0;
  for (const Slot& sl : c->slot) if (sl.busy) return fail(c, -EBUSY, "a submitted call is still pending (adsb_wait first)");
  {
    // the device pass is queued, THEN the block's pass-through copy (framer.py:181: out0[:] = in0[history:]) runs on the
    // host -- beside the upload's DMA and the kernels instead of behind them -- then the pass is waited for
    Slot& sl0 = c->slot[0];
    c->last_slot = 0;
    rc = enqueue(c, sl0, pl, false);
    if (rc) { sl0.busy = false; return rc; }
    if (out0) {
      const size_t pb = (size_t)N * sizeof(float);
      if (pb >= ((size_t)1 << 20) && ensure_pool(c) == 0) c->pool->copy((char*)out0, (const char*)(in0 + (H - 1)), pb);
      else memcpy(out0, in0 + (H - 1), pb);
    }
    rc = finish(c, sl0, &s, &nres);
  }
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
  // (independent of submitted calls still in flight: own buffers, ordered behind them on the compute stream)
  // Tag positions, packed bits, ok flags and ratios live in the context's pinned, device-visible scratch (layout: tag
  // positions | packed bits | ok | ratios): the kernel reads and writes them in place, so the call is one sample upload,
  // one kernel and ONE synchronisation.  Every buffer is acquired BEFORE anything is queued: no error path leaves work
  // in flight.
  const size_t nt = (size_t)ntags;
  const size_t o_bits = nt * 8, o_ok = o_bits + nt * 14, o_ratio = (o_ok + nt + 15) & ~(size_t)15;
  const size_t total = o_ratio + (ratio ? nt * 112 * sizeof(float) : 0);
  int rc;
  if ((rc = ensure_pinned(c, c->h_dm, c->h_dm_cap, total))) return rc;
  char* h = (char*)c->h_dm;
  long long* loc = (long long*)h;
  // local positions of the tags inside in0 (demod.py:79: offset - nitems_written); a tag outside the chunk -- the
  // reference's get_tags_in_range never returns one (demod.py:67) -- is dropped by the kernel (ok = 0)
  for (int t = 0; t < ntags; ++t) loc[t] = tag_offsets[t] - nitems_read;
  void* d = nullptr;
  if ((rc = upload(c, in0, (size_t)n * 4, &d))) return rc;
  int nb = (ntags + kWaves - 1) / kWaves;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL((k_slice<1>), dim3(nb), dim3(kThreads), 0, c->stream, (const void*)d, (long long)n,
                     (const long long*)loc, (int)ntags, c->sps, (unsigned char*)(h + o_bits),
                     (unsigned char*)(h + o_ok), ratio ? (float*)(h + o_ratio) : (float*)nullptr);
  hipError_t he = hipStreamSynchronize(c->stream);            // always: nothing stays queued behind an error return
  if (he == hipSuccess) he = hipGetLastError();
  if (he != hipSuccess) return fail(c, -EIO, "k_slice", he);
  const unsigned char* packed = (const unsigned char*)(h + o_bits);
  for (size_t t = 0; t < nt; ++t)
    for (int k = 0; k < 112; ++k) bits112[t * 112 + k] = (packed[t * 14 + (k >> 3)] >> (7 - (k & 7))) & 1u;
  memcpy(ok, h + o_ok, nt);
  if (ratio) {
    memcpy(ratio, h + o_ratio, nt * 112 * sizeof(float));
    for (size_t t = 0; t < nt; ++t)                            // the kernel leaves the rows of dropped tags untouched
      if (!ok[t]) memset(ratio + t * 112, 0, 112 * sizeof(float));
  }
  return 0;
}

// Halo checks of a finished shard call (the records are in the slot's pinned buffer).  drop_overlong
// (ADSB_SHARD_DROP_OVERLONG): a centre whose pulse or burst runs past the buffer's forward halo is left out of the
// result instead of failing the call -- the reference degrades the same way (a pulse still high at the end of a
// call is never evaluated, framer.py:102-108; a burst past the end of the chunk is dropped, demod.py:130-133).
static int shard_post(adsb_ctx* c, Slot& s, const Summary& sum, int32_t* nres_io, bool drop_overlong) {
  if ((sum.flags & 4u) && !drop_overlong) return fail(c, -EOVERFLOW, "pulse runs past the shard's forward halo");
  Rec* r = (Rec*)s.h_out;
  const int32_t nres = *nres_io;
  const long long origin = s.plan.origin, n = s.plan.n, stream_len = s.plan.origin + s.plan.dem_hi;
  int32_t w = 0;
  for (int i = 0; i < nres; ++i) {
    const unsigned fl = (unsigned)(r[i].w[3] >> 48);
    const long long off = (long long)r[i].w[0];
    const long long eob = off + 119ll * c->sps + c->sps / 2;
    if (!(fl & kDemod) && eob < stream_len) return fail(c, -EOVERFLOW, "internal: demod flag");
    if ((fl & kDemod) && eob >= origin + n) {
      if (drop_overlong) continue;
      return fail(c, -EOVERFLOW, "burst runs past the shard's forward halo");
    }
    if (off - 100 < origin && origin > 0) return fail(c, -EOVERFLOW, "noise window runs past the shard's back halo");
    if (w != i) {
      r[w] = r[i];
      // row t of adsb_last_confidence belongs to record t of the delivered list: rows move with their records
      if ((c->flags & ADSB_FLAG_CONFIDENCE) && s.h_ratio) memcpy((float*)s.h_ratio + (size_t)w * 112, (float*)s.h_ratio + (size_t)i * 112, 112 * sizeof(float));
    }
    ++w;
  }
  s.nres = w;
  *nres_io = w;
  return 0;
}

static int shard_plan_checked(adsb_ctx* c, int fmt, const void* d_data, int64_t n, int64_t origin, int64_t own_lo,
                              int64_t own_hi, int64_t stream_len, int32_t head_cands, Plan* pl) {
  if (!c || n < 0 || fmt < 0 || fmt >= ADSB_FMT_COUNT || head_cands < 0) return -EINVAL;
  if (((uintptr_t)d_data & 15u) != 0) return fail(c, -EINVAL, "device pointer must be 16-byte aligned");
  *pl = plan_shard(fmt, d_data, n, origin, own_lo, own_hi, stream_len, c->sps, head_cands);
  pl->long_aware = (c->flags & ADSB_FLAG_LONG_AWARE_GATE) != 0;
  if (origin > 0 && pl->scan_lo < 1) return fail(c, -EINVAL, "shard needs at least one sample of back halo");
  return 0;
}

int adsb_shard_device(adsb_ctx* c, int fmt, const void* d_data, int64_t n, int64_t origin, int64_t own_lo,
                      int64_t own_hi, int64_t stream_len, int32_t head_cands, adsb_burst* out, int32_t cap,
                      int32_t* n_out) {
  Plan pl;
  int rc = shard_plan_checked(c, fmt, d_data, n, origin, own_lo, own_hi, stream_len, head_cands, &pl);
  if (rc) return rc;
  Summary s;
  int32_t nres = 0;
  rc = run_pipeline(c, pl, &s, &nres);
  if (rc) return rc;
  if ((rc = shard_post(c, c->slot[c->last_slot], s, &nres))) return rc;
  return deliver(c, nres, out, cap, n_out);
}

int adsb_shard_host(adsb_ctx* c, int fmt, const void* host, int64_t n, int64_t origin, int64_t own_lo, int64_t own_hi,
                    int64_t stream_len, int32_t head_cands, uint32_t shard_flags, adsb_burst* out, int32_t cap,
                    int32_t* n_out) {
  if (!c || fmt < 0 || fmt >= ADSB_FMT_COUNT || n < 1 || !host) return -EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  void* d = nullptr;
  int rc = upload(c, host, (size_t)n * (size_t)mode_bytes(fmt), &d);
  if (rc) return rc;
  Plan pl;
  if ((rc = shard_plan_checked(c, fmt, d, n, origin, own_lo, own_hi, stream_len, head_cands, &pl))) return rc;
  Summary s;
  int32_t nres = 0;
  if ((rc = run_pipeline(c, pl, &s, &nres))) return rc;
  if ((rc = shard_post(c, c->slot[c->last_slot], s, &nres, (shard_flags & ADSB_SHARD_DROP_OVERLONG) != 0))) return rc;
  return deliver(c, nres, out, cap, n_out);
}

int adsb_submit_shard_device(adsb_ctx* c, int fmt, const void* d_data, int64_t n, int64_t origin, int64_t own_lo,
                             int64_t own_hi, int64_t stream_len, int32_t head_cands, int32_t* ticket) {
  if (!ticket) return -EINVAL;
  Plan pl;
  int rc = shard_plan_checked(c, fmt, d_data, n, origin, own_lo, own_hi, stream_len, head_cands, &pl);
  if (rc) return rc;
  Slot& s = c->slot[c->next_slot];
  if (s.busy) return fail(c, -EBUSY, "every pipeline slot is in flight (adsb_wait first)");
  rc = enqueue(c, s, pl, true);
  if (rc) { s.busy = false; return rc; }
  s.is_shard = true;
  *ticket = c->next_slot;
  c->next_slot = (c->next_slot + 1) % ADSB_MAX_IN_FLIGHT;
  return 0;
}

int32_t adsb_shard_bounds(int64_t stream_len, int32_t n_shards, int32_t g, int sps, int64_t align, int64_t* own_lo,
                          int64_t* own_hi, int64_t* lo, int64_t* hi) {
  // The tiling of gr_adsb_amd/frontend.py: shard_plan (one rule for every caller: bench.py's ranks, file replay, the C
  // driver below): equal owner ranges of a multiple of `align` samples; back halo = noise window + preamble span + 4
  // (framer.py:31,137-147), buffer start on a 16-byte boundary of every format; forward halo = longest pulse followed (256)
  // + preamble and 112 bits (121*sps: framer.py:165, demod.py:76).
  if (stream_len < 0 || n_shards < 1 || g < 0 || g >= n_shards || sps < 2 || align < 1 || !own_lo || !own_hi || !lo || !hi) return -EINVAL;
  long long per = (stream_len + n_shards - 1) / n_shards;
  per = (per + align - 1) / align * align;
  const long long olo = (long long)g * per < stream_len ? (long long)g * per : stream_len;
  const long long ohi = (long long)(g + 1) * per < stream_len ? (long long)(g + 1) * per : stream_len;
  long long l = olo - (kNoise + 8ll * sps + 4);
  if (l < 0) l = 0;
  l -= l % 4;
  long long h = ohi + 256 + 121ll * sps;
  if (h > stream_len) h = stream_len;
  *own_lo = olo; *own_hi = ohi; *lo = l; *hi = h;
  return 0;
}

// the plain greedy gate (framer.py:121-123,165) over a shard's UNGATED centres with the incoming end-of-burst state: the
// fallback for a shard whose head region ends inside an unbroken chain of overlapping bursts
static int32_t gate_from(adsb_burst* recs, int32_t n, int sps, long long* eob_io) {
  long long eob = *eob_io;
  int32_t w = 0;
  for (int32_t i = 0; i < n; ++i) {
    if (recs[i].offset > eob) {
      eob = recs[i].offset + ((recs[i].flags & ADSB_BURST_LONG_HINT) ? 119ll : 63ll) * sps;
      adsb_burst b = recs[i];
      b.flags = (uint16_t)((b.flags | ADSB_BURST_KEPT) & ~ADSB_BURST_HEAD);
      recs[w++] = b;
    }
  }
  *eob_io = eob;
  return w;
}

int adsb_process_sharded_device(adsb_ctx* c, int fmt, const void* d_data, int64_t n, int64_t abs_offset, int32_t shards,
                                adsb_burst* out, int32_t cap, int32_t* n_out) {
  if (!c || fmt < 0 || fmt >= ADSB_FMT_COUNT || n < 0 || shards < 1 || cap < 0 || (cap > 0 && !out) || !n_out) return -EINVAL;
  if (((uintptr_t)d_data & 15u) != 0) return fail(c, -EINVAL, "device pointer must be 16-byte aligned");
  for (const Slot& sl : c->slot) if (sl.busy) return fail(c, -EBUSY, "a submitted call is still pending (adsb_wait first)");
  // the rows of adsb_last_confidence belong to the records of ONE pass in the order that pass delivered them; this driver
  // re-gates and concatenates the records of many passes on the host
  if (c->flags & ADSB_FLAG_CONFIDENCE) return fail(c, -EINVAL, "adsb_process_sharded_device: not for ADSB_FLAG_CONFIDENCE contexts");
  *n_out = 0;
  if (n == 0) return 0;
  // adsb_wait_for_event: EVERY shard pass reads the caller's buffer and the passes run on different streams, so every
  // one of them (the re-runs of the fallback included) waits for the pending events -- enqueue() applies and clears what
  // is pending, the driver puts the same events back in front of each of its passes
  hipEvent_t ext_snap[adsb_ctx::kMaxExt];
  const int n_ext_snap = c->n_ext;
  for (int i = 0; i < n_ext_snap; ++i) ext_snap[i] = c->ext_ev[i];
  auto rearm = [&]() { for (int i = 0; i < n_ext_snap; ++i) c->ext_ev[i] = ext_snap[i]; c->n_ext = n_ext_snap; };
  const int bps = mode_bytes(fmt);
  struct Pend { int ticket; long long own_lo, own_hi, lo, hi; };
  Pend pend[ADSB_MAX_IN_FLIGHT];
  int n_pend = 0, head = 0;
  long long eob = -(1ll << 60);
  int64_t total = 0;
  int rc = 0;
  const int kHead = 64;

  // collect the oldest pass: fix up its head with the carried state, append what it kept
  auto collect = [&]() -> int {
    const Pend p = pend[head];
    head = (head + 1) % ADSB_MAX_IN_FLIGHT; --n_pend;
    Slot& s = c->slot[p.ticket];
    Summary sum;
    int32_t nres = 0;
    int r = finish(c, s, &sum, &nres);
    if (r) return r;
    c->last_slot = p.ticket;
    if ((r = shard_post(c, s, sum, &nres))) return r;
    adsb_burst* recs = (adsb_burst*)s.h_out;
    int32_t kept = 0;
    r = adsb_shard_fixup(recs, nres, c->sps, eob, &kept);
    if (r == -EAGAIN) {
      // the head region ended inside a chain: this shard again on the slot that has just become free -- first with the
      // largest head, then ungated with the plain greedy gate (exact in every case)
      for (int attempt = 0; attempt < 2 && r == -EAGAIN; ++attempt) {
        Plan pl;
        if ((r = shard_plan_checked(c, fmt, (const char*)d_data + (size_t)p.lo * bps, p.hi - p.lo, p.lo, p.own_lo, p.own_hi, n,
                                    attempt == 0 ? 4096 : 0, &pl))) return r;
        rearm();
        if ((r = enqueue(c, s, pl, true))) { s.busy = false; return r; }
        if ((r = finish(c, s, &sum, &nres))) return r;
        if ((r = shard_post(c, s, sum, &nres))) return r;
        recs = (adsb_burst*)s.h_out;
        if (attempt == 0) r = adsb_shard_fixup(recs, nres, c->sps, eob, &kept);
        else { long long e = eob; kept = gate_from(recs, nres, c->sps, &e); r = 0; }
      }
      if (r) return r;
      c->stats.shard_fallbacks++;
    } else if (r) return fail(c, r, "adsb_shard_fixup");
    if (kept > 0) {
      const adsb_burst& last = recs[kept - 1];
      eob = last.offset + ((last.flags & ADSB_BURST_LONG_HINT) ? 119ll : 63ll) * c->sps;
      if (total + kept <= cap) {
        memcpy(out + total, recs, (size_t)kept * sizeof(adsb_burst));
        if (abs_offset) for (int32_t i = 0; i < kept; ++i) out[total + i].offset += abs_offset;
      }
    }
    total += kept;
    s.nres = kept;
    return 0;
  };

  for (int32_t g = 0; g < shards && !rc; ++g) {
    int64_t own_lo, own_hi, lo, hi;
    if ((rc = adsb_shard_bounds(n, shards, g, c->sps, 4096, &own_lo, &own_hi, &lo, &hi))) break;
    if (own_hi <= own_lo) continue;
    if (n_pend == ADSB_MAX_IN_FLIGHT && (rc = collect())) break;
    Plan pl;
    if ((rc = shard_plan_checked(c, fmt, (const char*)d_data + (size_t)lo * bps, hi - lo, lo, own_lo, own_hi, n, kHead, &pl))) break;
    Slot& s = c->slot[c->next_slot];
    rearm();
    if ((rc = enqueue(c, s, pl, true))) { s.busy = false; break; }
    s.is_shard = true;
    pend[(head + n_pend) % ADSB_MAX_IN_FLIGHT] = Pend{c->next_slot, own_lo, own_hi, lo, hi};
    ++n_pend;
    c->next_slot = (c->next_slot + 1) % ADSB_MAX_IN_FLIGHT;
  }
  while (n_pend > 0) {
    const int r = collect();          // (after an error too: no pass stays in flight behind this call)
    if (r && !rc) rc = r;
  }
  c->n_ext = 0;                       // consumed by this call, whether a pass was queued or not
  if (rc) return rc;
  *n_out = (int32_t)(total > 0x7FFFFFFF ? 0x7FFFFFFF : total);
  if (total > cap) return fail(c, -ENOSPC, "output array too small");
  return 0;
}

// ---- one process, N devices, one host ring (BASELINE config 4) --------------------------------------------
// The stream lies in HOST memory; context k (one per device, or several on one) takes `shards_per_ctx` consecutive
// overlapped time shards of it.  One feeder thread per context, inside the cpus local to its GPU: upload of shard i+1 on
// the context's upload stream beside the shard pass of i and the record download of i-1, ADSB_MAX_IN_FLIGHT deep -- the
// host-fed pipeline of adsb_submit_format_host with a shard plan.  The calling thread takes the finished shards in STREAM
// order as they arrive and re-gates each head with the end-of-burst state carried over the seam (adsb_shard_fixup); a head
// that ends inside an unbroken chain has its shard run again on its own context (largest head, then ungated + the plain
// greedy gate), after that context's feeder has finished.  No interpreter, no torch.distributed, no mailbox: the only
// thing that crosses a seam is one int64.
namespace {

struct MultiShard {
  long long own_lo = 0, own_hi = 0, lo = 0, hi = 0;
  std::vector<adsb_burst> recs;      // the shard's records as its pass delivered them (head + what a fresh-state gate kept)
  int rc = 0;
  bool done = false;
};

// queue one shard of a host-resident stream on the context's next slot: upload -> shard pass
int submit_shard_host(adsb_ctx* c, int fmt, const char* host, const MultiShard& sh, long long stream_len, int head, int32_t* ticket) {
  Slot& s = c->slot[c->next_slot];
  if (s.busy) return fail(c, -EBUSY, "every pipeline slot is in flight");
  HIPCHK(c, hipSetDevice(c->device));
  const int bps = mode_bytes(fmt);
  int rc = upload_async(c, s, host + (size_t)sh.lo * bps, (size_t)(sh.hi - sh.lo) * bps);
  if (rc) return rc;
  Plan pl;
  if ((rc = shard_plan_checked(c, fmt, s.d_in.p, sh.hi - sh.lo, sh.lo, sh.own_lo, sh.own_hi, stream_len, head, &pl))) return rc;
  if ((rc = enqueue(c, s, pl, true))) { s.busy = false; return rc; }
  s.is_shard = true;
  *ticket = c->next_slot;
  c->next_slot = (c->next_slot + 1) % ADSB_MAX_IN_FLIGHT;
  return 0;
}

int collect_shard(adsb_ctx* c, int32_t ticket, std::vector<adsb_burst>* out) {
  Slot& s = c->slot[ticket];
  Summary sum;
  int32_t nres = 0;
  int r = finish(c, s, &sum, &nres);
  if (r) return r;
  c->last_slot = ticket;
  if ((r = shard_post(c, s, sum, &nres))) return r;
  const adsb_burst* recs = (const adsb_burst*)s.h_out;
  try {
    out->assign(recs, recs + nres);
  } catch (...) {                                     // (no exception crosses the ABI or ends a feeder thread)
    return fail(c, -ENOMEM, "shard records");
  }
  return 0;
}

}  // namespace

int adsb_process_sharded_multi(adsb_ctx* const* ctxs, int32_t n_ctx, int fmt, const void* host, int64_t n, int64_t abs_offset,
                               int32_t shards_per_ctx, adsb_burst* out, int32_t cap, int32_t* n_out, adsb_multi_stats* stats) {
  if (!ctxs || n_ctx < 1 || n_ctx > ADSB_MULTI_MAX_CTX || fmt < 0 || fmt >= ADSB_FMT_COUNT || n < 0 || shards_per_ctx < 1 ||
      cap < 0 || (cap > 0 && !out) || !n_out || (n > 0 && !host)) return -EINVAL;
  adsb_ctx* c0 = ctxs[0];
  if (!c0) return -EINVAL;
  for (int k = 0; k < n_ctx; ++k) {
    adsb_ctx* c = ctxs[k];
    if (!c) return -EINVAL;
    for (int j = 0; j < k; ++j) if (ctxs[j] == c) return fail(c0, -EINVAL, "adsb_process_sharded_multi: a context listed twice");
    // one stream, one set of rules: every context must have been created with the same rate, threshold, gate and scale
    if (c->sps != c0->sps || !(c->thr == c0->thr) || ((c->flags ^ c0->flags) & ADSB_FLAG_LONG_AWARE_GATE) ||
        !(c->scale[fmt] == c0->scale[fmt]))
      return fail(c0, -EINVAL, "adsb_process_sharded_multi: contexts differ in rate, threshold, gate or format scale");
    if (c->flags & ADSB_FLAG_CONFIDENCE) return fail(c0, -EINVAL, "adsb_process_sharded_multi: not for ADSB_FLAG_CONFIDENCE contexts");
    if (!c->own_stream) return fail(c0, -EINVAL, "adsb_process_sharded_multi: not for contexts on a caller-owned stream");
    for (const Slot& sl : c->slot) if (sl.busy) return fail(c0, -EBUSY, "a submitted call is still pending on one of the contexts");
  }
  *n_out = 0;
  if (stats) memset(stats, 0, sizeof(*stats));
  if (n == 0) return 0;
  const int sps = c0->sps, kHead = 64;
  const int G = n_ctx * shards_per_ctx;
  const auto t_start = std::chrono::steady_clock::now();

  std::vector<MultiShard> sh;
  std::vector<double> feed_s;
  std::vector<std::thread> th;
  std::vector<char> joined;
  try {
    sh.resize((size_t)G); feed_s.assign((size_t)n_ctx, 0.0); th.reserve((size_t)n_ctx); joined.assign((size_t)n_ctx, 0);
  } catch (...) {
    return fail(c0, -ENOMEM, "adsb_process_sharded_multi: shard table");
  }
  for (int g = 0; g < G; ++g) {
    int64_t olo, ohi, lo, hi;
    const int rc = adsb_shard_bounds(n, G, g, sps, 4096, &olo, &ohi, &lo, &hi);
    if (rc) return rc;
    sh[(size_t)g].own_lo = olo; sh[(size_t)g].own_hi = ohi; sh[(size_t)g].lo = lo; sh[(size_t)g].hi = hi;
  }
  std::mutex m;
  std::condition_variable cv;

  // context k's feeder: its shards, ADSB_MAX_IN_FLIGHT deep.  An error ends this feeder only (its remaining shards are
  // marked done with the error) -- and nothing of its context stays in flight behind it.
  auto feeder = [&](int k) {
    adsb_ctx* c = ctxs[k];
    if (c->have_local_cpus && !(c->flags & ADSB_FLAG_NO_NUMA_BINDING)) {
      cpu_set_t mine, both;
      CPU_ZERO(&mine);
      if (sched_getaffinity(0, sizeof(mine), &mine) == 0) {
        CPU_AND(&both, &mine, &c->local_cpus);
        if (CPU_COUNT(&both) > 0) (void)pthread_setaffinity_np(pthread_self(), sizeof(both), &both);
      }
    }
    const auto t0 = std::chrono::steady_clock::now();
    const int g0 = k * shards_per_ctx, g1 = g0 + shards_per_ctx;
    struct Fly { int g; int32_t ticket; };
    Fly fly[ADSB_MAX_IN_FLIGHT];
    int n_fly = 0, head = 0, rc = 0;
    auto mark = [&](int g, int r) {
      { std::lock_guard<std::mutex> lk(m); sh[(size_t)g].rc = r; sh[(size_t)g].done = true; }
      cv.notify_all();
    };
    auto collect_oldest = [&]() {
      const Fly f = fly[head];
      head = (head + 1) % ADSB_MAX_IN_FLIGHT; --n_fly;
      std::vector<adsb_burst> recs;
      const int r = collect_shard(c, f.ticket, &recs);
      if (r && !rc) rc = r;
      { std::lock_guard<std::mutex> lk(m); sh[(size_t)f.g].recs.swap(recs); }
      mark(f.g, r ? r : rc);
    };
    int g = g0;
    for (; g < g1 && !rc; ++g) {
      if (sh[(size_t)g].own_hi <= sh[(size_t)g].own_lo) { mark(g, 0); continue; }     // (a stream shorter than the tiling: nothing owned)
      if (n_fly == ADSB_MAX_IN_FLIGHT) collect_oldest();
      if (rc) break;
      int32_t ticket = -1;
      const int r = submit_shard_host(c, fmt, (const char*)host, sh[(size_t)g], n, kHead, &ticket);
      if (r) { rc = r; break; }
      fly[(head + n_fly) % ADSB_MAX_IN_FLIGHT] = Fly{g, ticket};
      ++n_fly;
    }
    while (n_fly > 0) collect_oldest();
    for (; g < g1; ++g) mark(g, rc ? rc : -EIO);                                     // what an error left unsubmitted
    feed_s[(size_t)k] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  };

  for (int k = 0; k < n_ctx; ++k) {
    try {
      th.emplace_back(feeder, k);
    } catch (...) {
      // no thread for this context (resource limit): its shards fail, the others still run and are collected
      th.emplace_back();                                     // (keeps th[k] <-> context k; reserve() above: cannot throw)
      joined[(size_t)k] = 1;
      for (int g = k * shards_per_ctx; g < (k + 1) * shards_per_ctx; ++g) {
        { std::lock_guard<std::mutex> lk(m); sh[(size_t)g].rc = -ENOMEM; sh[(size_t)g].done = true; }
      }
      (void)fail(ctxs[k], -ENOMEM, "adsb_process_sharded_multi: could not start the feeder thread");
    }
  }
  auto join_one = [&](int k) { if (!joined[(size_t)k]) { if (th[(size_t)k].joinable()) th[(size_t)k].join(); joined[(size_t)k] = 1; } };

  // the calling thread: finished shards in stream order, head fix-up with the carried state, records to `out`
  long long eob = -(1ll << 60);
  int64_t total = 0;
  int rc = 0, fallbacks = 0;
  for (int g = 0; g < G; ++g) {
    MultiShard& S = sh[(size_t)g];
    {
      std::unique_lock<std::mutex> lk(m);
      cv.wait(lk, [&] { return S.done; });
    }
    if (S.rc) { if (!rc) rc = S.rc; continue; }
    if (rc || S.own_hi <= S.own_lo) continue;
    adsb_ctx* c = ctxs[g / shards_per_ctx];
    int32_t kept = 0;
    int r = adsb_shard_fixup(S.recs.data(), (int32_t)S.recs.size(), sps, eob, &kept);
    if (r == -EAGAIN) {
      // the head region ended inside a chain: this shard again on its own context, once its feeder is through with it
      join_one(g / shards_per_ctx);
      for (int attempt = 0; attempt < 2 && r == -EAGAIN; ++attempt) {
        int32_t ticket = -1;
        if ((r = submit_shard_host(c, fmt, (const char*)host, S, n, attempt == 0 ? 4096 : 0, &ticket))) break;
        if ((r = collect_shard(c, ticket, &S.recs))) break;
        if (attempt == 0) r = adsb_shard_fixup(S.recs.data(), (int32_t)S.recs.size(), sps, eob, &kept);
        else { long long e = eob; kept = gate_from(S.recs.data(), (int32_t)S.recs.size(), sps, &e); r = 0; }
      }
      if (!r) { ++fallbacks; c->stats.shard_fallbacks++; }
    }
    if (r) { rc = r < 0 ? r : -EIO; if (r == -EAGAIN) rc = -EIO; continue; }
    if (kept > 0) {
      const adsb_burst& last = S.recs[(size_t)kept - 1];
      eob = last.offset + ((last.flags & ADSB_BURST_LONG_HINT) ? 119ll : 63ll) * sps;
      if (total + kept <= cap) {
        memcpy(out + total, S.recs.data(), (size_t)kept * sizeof(adsb_burst));
        if (abs_offset) for (int32_t i = 0; i < kept; ++i) out[total + i].offset += abs_offset;
      }
    }
    total += kept;
    std::vector<adsb_burst>().swap(S.recs);
  }
  for (int k = 0; k < n_ctx; ++k) join_one(k);
  if (stats) {
    stats->contexts = n_ctx; stats->shards = G; stats->fallbacks = fallbacks;
    stats->wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    for (int k = 0; k < n_ctx; ++k) {
      stats->feeder_s[k] = feed_s[(size_t)k];
      stats->device[k] = ctxs[k]->device;
      stats->numa_node[k] = ctxs[k]->numa_node;
    }
  }
  if (rc) {
    for (int k = 0; k < n_ctx; ++k)
      if (ctxs[k] != c0 && ctxs[k]->err[0] && !c0->err[0]) snprintf(c0->err, sizeof(c0->err), "context %d: %s", k, ctxs[k]->err);
    return rc;
  }
  *n_out = (int32_t)(total > 0x7FFFFFFF ? 0x7FFFFFFF : total);
  if (total > cap) return fail(c0, -ENOSPC, "output array too small");
  return 0;
}

// ---- plain device memory for callers that do not link HIP (C / ctypes clients of the *_device entry points) ---------------
int adsb_device_alloc(adsb_ctx* c, void** d, size_t bytes) {
  if (!c || !d || bytes == 0) return -EINVAL;
  *d = nullptr;
  HIPCHK(c, hipSetDevice(c->device));
  if (hipMalloc(d, bytes) != hipSuccess) { (void)hipGetLastError(); *d = nullptr; return fail(c, -ENOMEM, "hipMalloc"); }
  return 0;
}

int adsb_device_free(adsb_ctx* c, void* d) {
  if (!c) return -EINVAL;
  if (!d) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipFree(d));
  return 0;
}

int adsb_device_upload(adsb_ctx* c, void* d, const void* host, size_t bytes) {
  if (!c || (bytes > 0 && (!d || !host))) return -EINVAL;
  if (bytes == 0) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));      // blocking: the buffer is ready when this returns
  return 0;
}

int adsb_shard_fixup(adsb_burst* recs, int32_t n, int sps, int64_t eob_in, int32_t* n_kept) {
  // recs: output of adsb_shard_device(head_cands > 0): every centre of the shard's head (ADSB_BURST_HEAD,
  // complete, gated or not) followed by the centres a fresh-state gate kept.  Re-gate the head with the
  // true incoming eob (framer.py:121-123,165) until the first centre that starts an independent chain
  // -- beyond the reach (offset + gate window) of every head centre before it (trivially true for the first one)
  // and beyond eob_in: it is accepted whatever came before, so from there on the fresh-state decisions are exact.
  // A shard that lies entirely in its head region is re-gated completely and never fails; whether its END-OF-BURST
  // state equals the fresh-state one is a separate question the caller answers with the same criterion
  // (gr_adsb_amd/_native.py shard_head_sync; sharding.finish_shard falls back to adsb_stitch when it does not).
  // The window of a record is 63*sps (framer.py:165), or 119*sps for records flagged ADSB_BURST_LONG_HINT by a long-aware context.
  if (n < 0 || (n > 0 && !recs) || sps < 2 || !n_kept) return -EINVAL;
  int i = 0, w = 0;
  long long eob = eob_in, reach = -(1ll << 61);
  bool synced = false;
  for (; i < n && (recs[i].flags & ADSB_BURST_HEAD); ++i) {
    const long long p = recs[i].offset;
    const long long gate = ((recs[i].flags & ADSB_BURST_LONG_HINT) ? 119ll : 63ll) * sps;
    if (p > reach && p > eob_in) { synced = true; break; }   // also i == 0: the first centre lies beyond eob_in
    if (p + gate > reach) reach = p + gate;
    if (p > eob) {
      eob = p + gate;
      adsb_burst b = recs[i];
      b.flags = (uint16_t)((b.flags | ADSB_BURST_KEPT) & ~ADSB_BURST_HEAD);
      recs[w++] = b;
    }
  }
  if (!synced && i < n) return -EAGAIN;   // the head region ended inside a chain: ask for a larger head
  for (; i < n; ++i) {
    if (!(recs[i].flags & ADSB_BURST_KEPT)) continue;
    adsb_burst b = recs[i];
    b.flags = (uint16_t)(b.flags & ~ADSB_BURST_HEAD);
    recs[w++] = b;
  }
  *n_kept = w;
  return 0;
}

int adsb_stitch(adsb_burst* cands, int32_t n, int sps, int32_t* n_kept) {
  if (n < 0 || (n > 0 && !cands) || sps < 2) return -EINVAL;
  long long eob = -(1ll << 61);
  int w = 0;
  for (int i = 0; i < n; ++i) {
    if (i > 0 && cands[i].offset <= cands[i - 1].offset) return -EINVAL;  // must be in stream order
    if (cands[i].offset > eob) {                                           // framer.py:121
      eob = cands[i].offset + ((cands[i].flags & ADSB_BURST_LONG_HINT) ? 119ll : 63ll) * sps;   // framer.py:165 (+ §8f-4)
      adsb_burst b = cands[i];
      b.flags |= ADSB_BURST_KEPT;
      cands[w++] = b;
    }
  }
  if (n_kept) *n_kept = w;
  return 0;
}

int adsb_host_alloc(void** p, size_t bytes) {
  if (!p || bytes == 0) return -EINVAL;
  *p = nullptr;
  return hipHostMalloc(p, bytes, hipHostMallocDefault) == hipSuccess ? 0 : -ENOMEM;
}

int adsb_host_free(void* p) {
  if (!p) return 0;
  return hipHostFree(p) == hipSuccess ? 0 : -EINVAL;
}

int adsb_host_register(void* p, size_t bytes) {
  if (!p || bytes == 0) return -EINVAL;
  const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
  if (e != hipSuccess) { (void)hipGetLastError(); return e == hipErrorHostMemoryAlreadyRegistered ? -EEXIST : -ENOMEM; }
  return 0;
}

int adsb_host_unregister(void* p) {
  if (!p) return -EINVAL;
  if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return -EINVAL; }
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
  c->det_hist_n = 0;
  return 0;
}

int adsb_detect_history(adsb_ctx* c, float* ms, int32_t cap, int32_t* n) {
  if (!c || cap < 0 || (cap > 0 && !ms) || !n) return -EINVAL;
  const uint64_t have = c->det_hist_n < (uint64_t)adsb_ctx::kHist ? c->det_hist_n : (uint64_t)adsb_ctx::kHist;
  const uint64_t take = have < (uint64_t)cap ? have : (uint64_t)cap;
  for (uint64_t i = 0; i < take; ++i) ms[i] = c->det_hist[(c->det_hist_n - take + i) % adsb_ctx::kHist];   // oldest first
  *n = (int32_t)take;
  return 0;
}

const char* adsb_last_error(adsb_ctx* c) { return c ? c->err : "null context"; }

}  // extern "C"
