// hipsim.h -- TEST INFRASTRUCTURE ONLY: a tiny single-process SIMT emulator that lets the device
// code in gr_adsb_amd/csrc/adsb_device.h be compiled with g++ and executed on the CPU (no GPU in the
// build container).  Every HIP thread of a block is a ucontext fiber; __syncthreads() and the
// wave-level intrinsics (__ballot, __shfl, __shfl_up) are rendezvous points for 256 / 64 fibers.
// It is used by tests/sim/sim_driver.cpp to check kernel logic against the oracle in `-m "not gpu"`
// tests.  It is NOT a fallback: nothing under gr_adsb_amd/ includes it and the shipped library
// contains no CPU path.
//
// Restrictions (asserted where possible): blockDim.x is a multiple of 64; wave intrinsics are called
// by all 64 lanes of a wave in convergent control flow; no thread exits before its wave-mates'
// last wave intrinsic; blocks run one after another (a `__shared__` array is a function-level static).
#pragma once
#include <ucontext.h>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct short4 { short x, y, z, w; };
struct hipsim_dim3 { unsigned x = 1, y = 1, z = 1; };

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

namespace hipsim {
constexpr int kWave = 64;
constexpr size_t kStack = 256 * 1024;

struct WaveState {
  int arrived = 0;
  unsigned gen = 0;
  uint64_t slot[2][kWave];
};

struct Block {
  int nthreads = 0;
  int bar_arrived = 0;
  unsigned bar_gen = 0;
  std::vector<WaveState> waves;
  std::vector<ucontext_t> ctx;
  std::vector<char*> stacks;
  std::vector<char> done;
  ucontext_t sched;
  int cur = -1;
  int ndone = 0;
  std::function<void()> body;
};

inline Block*& cur_block() { static Block* b = nullptr; return b; }
inline hipsim_dim3& tIdx() { static hipsim_dim3 v; return v; }
inline hipsim_dim3& bIdx() { static hipsim_dim3 v; return v; }
inline hipsim_dim3& bDim() { static hipsim_dim3 v; return v; }
inline hipsim_dim3& gDim() { static hipsim_dim3 v; return v; }

inline void yield() {
  Block* b = cur_block();
  int me = b->cur;
  swapcontext(&b->ctx[me], &b->sched);
  // resumed: restore my thread index
  tIdx().x = (unsigned)me;
}

inline void trampoline() {
  Block* b = cur_block();
  int me = b->cur;
  tIdx().x = (unsigned)me;
  b->body();
  b->done[me] = 1;
  b->ndone++;
  swapcontext(&b->ctx[me], &b->sched);
}

inline void run_block(Block& b) {
  cur_block() = &b;
  b.ndone = 0;
  b.bar_arrived = 0;
  for (int t = 0; t < b.nthreads; ++t) {
    b.done[t] = 0;
    getcontext(&b.ctx[t]);
    b.ctx[t].uc_stack.ss_sp = b.stacks[t];
    b.ctx[t].uc_stack.ss_size = kStack;
    b.ctx[t].uc_link = nullptr;
    makecontext(&b.ctx[t], (void (*)())trampoline, 0);
  }
  for (auto& w : b.waves) { w.arrived = 0; }
  long spins = 0;
  while (b.ndone < b.nthreads) {
    for (int t = 0; t < b.nthreads; ++t) {
      if (b.done[t]) continue;
      b.cur = t;
      swapcontext(&b.sched, &b.ctx[t]);
    }
    if (++spins > 100000000L) { fprintf(stderr, "hipsim: deadlock (divergent barrier?)\n"); abort(); }
  }
  // a thread that exited while others still wait at a barrier is a bug in the kernel under test
  assert(b.bar_arrived == 0 && "hipsim: block ended with threads parked at __syncthreads");
}

inline void wave_sync(WaveState& w) {
  unsigned g = w.gen;
  if (++w.arrived == kWave) { w.arrived = 0; w.gen++; }
  else while (w.gen == g) yield();
}

template <class K, class... A>
void launch(K kernel, unsigned grid, unsigned block, A... args) {
  assert(block % kWave == 0);
  Block b;
  b.nthreads = (int)block;
  b.waves.resize(block / kWave);
  b.ctx.resize(block);
  b.done.resize(block);
  b.stacks.resize(block);
  for (unsigned t = 0; t < block; ++t) b.stacks[t] = (char*)malloc(kStack);
  bDim().x = block;
  gDim().x = grid;
  b.body = [&]() { kernel(args...); };
  for (unsigned g = 0; g < grid; ++g) {
    bIdx().x = g;
    run_block(b);
  }
  for (unsigned t = 0; t < block; ++t) free(b.stacks[t]);
  cur_block() = nullptr;
}
}  // namespace hipsim

#define threadIdx (hipsim::tIdx())
#define blockIdx (hipsim::bIdx())
#define blockDim (hipsim::bDim())
#define gridDim (hipsim::gDim())

inline void __syncthreads() {
  hipsim::Block* b = hipsim::cur_block();
  unsigned g = b->bar_gen;
  if (++b->bar_arrived == b->nthreads) { b->bar_arrived = 0; b->bar_gen++; }
  else while (b->bar_gen == g) hipsim::yield();
}

inline unsigned long long __ballot(int pred) {
  hipsim::Block* b = hipsim::cur_block();
  int me = b->cur, lane = me & 63;
  hipsim::WaveState& w = b->waves[me >> 6];
  unsigned par = w.gen & 1;
  w.slot[par][lane] = pred ? 1 : 0;
  hipsim::wave_sync(w);
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l) m |= (unsigned long long)(w.slot[par][l] & 1) << l;
  return m;
}

template <class T>
inline T hipsim_shfl_idx(T v, int src) {
  static_assert(sizeof(T) <= 8, "shfl type too wide");
  hipsim::Block* b = hipsim::cur_block();
  int me = b->cur, lane = me & 63;
  hipsim::WaveState& w = b->waves[me >> 6];
  unsigned par = w.gen & 1;
  uint64_t raw = 0;
  memcpy(&raw, &v, sizeof(T));
  w.slot[par][lane] = raw;
  hipsim::wave_sync(w);
  uint64_t r = w.slot[par][src & 63];
  T out;
  memcpy(&out, &r, sizeof(T));
  return out;
}
template <class T> inline T __shfl(T v, int src) { return hipsim_shfl_idx(v, src); }
template <class T> inline T __shfl_up(T v, unsigned d) {
  int lane = hipsim::cur_block()->cur & 63;
  int src = lane - (int)d;
  T r = hipsim_shfl_idx(v, src < 0 ? lane : src);
  return r;
}

template <class T> inline T __shfl_xor(T v, int mask) {
  int lane = hipsim::cur_block()->cur & 63;
  return hipsim_shfl_idx(v, lane ^ mask);
}

// wavefront-level ordering point required by adsb_device.h (device: compiler fence + wave_barrier)
inline void adsb_wave_sync() {
  hipsim::Block* b = hipsim::cur_block();
  hipsim::wave_sync(b->waves[b->cur >> 6]);
}

inline int adsb_uniform(int v) { return v; }
inline int adsb_opaque(int v) { return v; }
#define ADSB_DYN_LDS_INT(name) static int name[2048]
inline unsigned adsb_after(unsigned v, float) { return v; }
#define ADSB_LDS
template <class Q> inline Q adsb_ld_stream(const char* p) { Q q; memcpy(&q, p, sizeof(Q)); return q; }
template <int P> inline void adsb_setprio() {}
typedef unsigned long long adsb_u64x2 __attribute__((vector_size(16)));
template <class T> inline void adsb_st_stream(T* p, T v) { memcpy(p, &v, sizeof(T)); }
inline int adsb_readlane(int v, int lane) { return hipsim_shfl_idx(v, lane); }
inline unsigned adsb_above4(unsigned acc, float a, float b, float c, float d, float thr) {
  return (acc << 4) | ((a >= thr) ? 8u : 0u) | ((b >= thr) ? 4u : 0u) | ((c >= thr) ? 2u : 0u) | ((d >= thr) ? 1u : 0u);
}
inline float adsb_mag2(float x, float y) {
  volatile float a = x * x, b = y * y;
  volatile float m = a + b;
  return m;
}
inline float adsb_fmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
inline unsigned adsb_lane_up1(unsigned v, unsigned fill) {
  const int lane = hipsim::cur_block()->cur & 63;
  const unsigned r = hipsim_shfl_idx(v, lane ? lane - 1 : 0);
  return lane ? r : fill;
}
inline unsigned adsb_wave_incl_scan(unsigned x) {
  const int lane = hipsim::cur_block()->cur & 63;
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned t = hipsim_shfl_idx(x, lane >= d ? lane - d : lane);
    if (lane >= d) x += t;
  }
  return x;
}
inline unsigned adsb_wave_min_u32(unsigned v) {
  const int lane = hipsim::cur_block()->cur & 63;
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned o = hipsim_shfl_idx(v, lane ^ d);
    if (o < v) v = o;
  }
  return v;
}
inline unsigned adsb_wave_max_u32(unsigned v) {
  const int lane = hipsim::cur_block()->cur & 63;
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned o = hipsim_shfl_idx(v, lane ^ d);
    if (o > v) v = o;
  }
  return v;
}
inline unsigned long long adsb_bitrep32(unsigned x) {
  unsigned long long r = 0;
  for (int i = 0; i < 32; ++i) if ((x >> i) & 1u) r |= 3ull << (2 * i);
  return r;
}

inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline unsigned long long __brevll(unsigned long long v) {
  unsigned long long r = 0;
  for (int i = 0; i < 64; ++i) r |= ((v >> i) & 1ull) << (63 - i);
  return r;
}
inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }

template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
inline int adsb_sdot4(int a, int b, int c) {
  unsigned r = (unsigned)c;
  for (int k = 0; k < 4; ++k) r += (unsigned)((int)(signed char)((unsigned)a >> (8 * k)) * (int)(signed char)((unsigned)b >> (8 * k)));
  return (int)r;
}
inline void __threadfence() {}
inline void __threadfence_system() {}
// the kernel's argument block "as it lies in memory": for the emulator simply the by-value copy
template <class A> inline const A* adsb_cold(const A& a) { return &a; }
