// valu_rate.hip -- what do the instructions of k_detect's tile loop COST on gfx950?  One workgroup on one CU; per instruction
// kind a loop of 32 instances per iteration, (a) independent (eight accumulators: issue rate) and (b) dependent (one
// accumulator: latency), with 1 wavefront and with 1, 2 and 4 wavefronts per SIMD.  Time base: s_memtime (shader cycles).
// Also: one count-and-branch step of the median search (noise_median in adsb_device.h) as the compiler builds it.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define X8(OP) OP("%0") OP("%1") OP("%2") OP("%3") OP("%4") OP("%5") OP("%6") OP("%7")
#define D8(OP) OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0")

#define DEFK(name, OP, T, CA, REP)                                                                          \
  __global__ void __launch_bounds__(1024) k_##name(unsigned long long* out, int iters, unsigned seed) {      \
    T a0 = (T)(seed + CA##_TID), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,          \
      a6 = a0 + 6, a7 = a0 + 7;                                                                               \
    unsigned b = seed | 1u, c = threadIdx.x * 2654435761u + seed;                                             \
    __syncthreads();                                                                                          \
    const unsigned long long t0 = wall_clock64(), c0 = clock64();                                             \
    for (int i = 0; i < iters; ++i)                                                                           \
      asm volatile(REP(OP) REP(OP) REP(OP) REP(OP)                                                            \
                   : "+" CA(a0), "+" CA(a1), "+" CA(a2), "+" CA(a3), "+" CA(a4), "+" CA(a5), "+" CA(a6), "+" CA(a7) \
                   : "v"(b), "v"(c)                                                                           \
                   : "vcc", "scc");                                                                           \
    const unsigned long long t1 = wall_clock64(), c1 = clock64();                                             \
    T s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                                              \
    if (s == (T)0x12345677u) out[1] = (unsigned long long)s;                                                  \
    __syncthreads();                                                                                          \
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[2] = c1 - c0; }                                             \
  }
#define CV(x) "v"(x)
#define CS(x) "s"(x)
#define CV_TID threadIdx.x
#define CS_TID 0u
typedef unsigned u32;
typedef unsigned long long u64;

#define OP_FMA(a) "v_fma_f32 " a ", %8, %9, " a "\n"
#define OP_PKFMA(a) "v_pk_fma_f32 " a ", " a ", " a ", " a "\n"
#define OP_PKMUL(a) "v_pk_mul_f32 " a ", " a ", " a "\n"
#define OP_CVT(a) "v_cvt_f32_i32 " a ", " a "\n"
#define OP_CVTUB(a) "v_cvt_f32_ubyte1 " a ", " a "\n"
#define OP_DOT4C(a) "v_dot4c_i32_i8 " a ", %8, %9\n"
#define OP_MAX3(a) "v_max3_i32 " a ", " a ", %8, %9\n"
#define OP_AND(a) "v_and_b32 " a ", %8, " a "\n"
#define OP_SUB(a) "v_sub_u32 " a ", %8, " a "\n"
#define OP_MOV(a) "v_mov_b32 " a ", 0x4b000000\n"
#define OP_CMP(a) "v_cmp_lt_u32 " a ", %8, %9\n"
#define OP_DPP(a) "v_add_u32_dpp " a ", " a ", " a " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_BCAST(a) "v_mov_b32_dpp " a ", " a " row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define OP_PERM(a) "v_perm_b32 " a ", " a ", %8, %9\n"
#define OP_CNDMASK(a) "v_cndmask_b32 " a ", " a ", %8, vcc\n"
#define OP_LSHL64(a) "v_lshlrev_b64 " a ", 1, " a "\n"
#define OP_MBCNT(a) "v_mbcnt_lo_u32_b32 " a ", %8, " a "\n"
#define OP_BFREV(a) "v_bfrev_b32 " a ", " a "\n"
#define OP_FFBL(a) "v_ffbl_b32 " a ", " a "\n"
#define OP_ALIGNBIT(a) "v_alignbit_b32 " a ", " a ", %8, 7\n"
#define OP_ADDCO(a) "v_add_co_u32 " a ", vcc, %8, " a "\n"
#define OP_MAXF3(a) "v_max3_f32 " a ", " a ", %8, %9\n"
#define OP_SBCNT(a) "s_bcnt1_i32_b64 " a ", exec\n"
#define OP_SADD(a) "s_add_i32 " a ", " a ", 7\n"
#define OP_SCSEL(a) "s_cselect_b32 " a ", " a ", 5\n"
#define OP_READLANE(a) "v_readlane_b32 " a ", %8, 3\n"
#define OP_READFIRST(a) "v_readfirstlane_b32 " a ", %8\n"

DEFK(fma, OP_FMA, u32, CV, X8)
DEFK(fma_dep, OP_FMA, u32, CV, D8)
DEFK(pk_fma, OP_PKFMA, u64, CV, X8)
DEFK(pk_mul, OP_PKMUL, u64, CV, X8)
DEFK(cvt_f32_i32, OP_CVT, u32, CV, X8)
DEFK(cvt_f32_ubyte, OP_CVTUB, u32, CV, X8)
DEFK(dot4c_i8, OP_DOT4C, u32, CV, X8)
DEFK(dot4c_i8_dep, OP_DOT4C, u32, CV, D8)
DEFK(max3_i32, OP_MAX3, u32, CV, X8)
DEFK(max3_f32, OP_MAXF3, u32, CV, X8)
DEFK(and_b32, OP_AND, u32, CV, X8)
DEFK(and_b32_dep, OP_AND, u32, CV, D8)
DEFK(sub_u32, OP_SUB, u32, CV, X8)
DEFK(mov_lit, OP_MOV, u32, CV, X8)
DEFK(cmp_to_sgpr, OP_CMP, u64, CS, X8)
DEFK(add_dpp_row_shr, OP_DPP, u32, CV, X8)
DEFK(add_dpp_row_shr_dep, OP_DPP, u32, CV, D8)
DEFK(mov_dpp_bcast, OP_BCAST, u32, CV, X8)
DEFK(perm_b32, OP_PERM, u32, CV, X8)
DEFK(cndmask, OP_CNDMASK, u32, CV, X8)
DEFK(lshlrev_b64, OP_LSHL64, u64, CV, X8)
DEFK(mbcnt, OP_MBCNT, u32, CV, X8)
DEFK(bfrev, OP_BFREV, u32, CV, X8)
DEFK(ffbl, OP_FFBL, u32, CV, X8)
DEFK(alignbit, OP_ALIGNBIT, u32, CV, X8)
DEFK(add_co, OP_ADDCO, u32, CV, X8)
DEFK(s_bcnt1_b64, OP_SBCNT, u32, CS, X8)
DEFK(s_add, OP_SADD, u32, CS, X8)
DEFK(s_add_dep, OP_SADD, u32, CS, D8)
DEFK(s_cselect, OP_SCSEL, u32, CS, X8)
DEFK(readlane, OP_READLANE, u32, CS, X8)
DEFK(readfirstlane, OP_READFIRST, u32, CS, X8)

// LDS reads: 32 per iteration, (a) one float per lane, consecutive; (b) 16 bytes per lane at a 64-byte lane stride (the mask
// read-back of k_detect); (c) 16 bytes per lane, consecutive (the slide)
template <int KIND>
__global__ void __launch_bounds__(1024) k_lds(unsigned long long* out, int iters, unsigned seed) {
  __shared__ __attribute__((aligned(16))) float s[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) s[i] = (float)(i ^ seed);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  const unsigned long long t0 = wall_clock64(), c0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const int base = ((wave * 1024 + k * 16 + (i & 3) * 4) & 8191);
      if (KIND == 0) acc += s[base + lane];
      else if (KIND == 1) { const float4 q = *reinterpret_cast<const float4*>(&s[(base & ~3) + 16 * lane]); acc += q.x + q.w; }
      else { const float4 q = *reinterpret_cast<const float4*>(&s[(base & ~3) + 4 * lane]); acc += q.x + q.w; }
    }
  }
  const unsigned long long t1 = wall_clock64(), c1 = clock64();
  if (acc == 1.2345f) out[1] = 1;
  __syncthreads();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[2] = c1 - c0; }
}

// one step of the median search as noise_median() spells it: compare two keys per lane against a wave-uniform threshold,
// count, branch-free select of the new lower bound.  32 steps per iteration (bits 31..0 of a full search).
__global__ void __launch_bounds__(1024) k_median_step(unsigned long long* out, int iters, unsigned seed) {
  const unsigned k0 = (threadIdx.x * 2654435761u) ^ seed, k1 = (threadIdx.x * 40503u + 77u) * seed;
  unsigned sink = 0;
  __syncthreads();
  const unsigned long long t0 = wall_clock64(), c0 = clock64();
  for (int i = 0; i < iters; ++i) {
    unsigned lo = 0u;
    const int kt = 49 + (i & 1);
#pragma unroll
    for (int b = 31; b >= 0; --b) {
      const unsigned T = lo + (1u << b);
      const int c = __builtin_amdgcn_readfirstlane(__popcll(__ballot(k0 < T)) + __popcll(__ballot(k1 < T)));
      if (c <= kt) lo = T;
    }
    sink ^= lo;
  }
  const unsigned long long t1 = wall_clock64(), c1 = clock64();
  if (sink == 0x1234567u) out[1] = sink;
  __syncthreads();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[2] = c1 - c0; }
}

typedef void (*KFn)(unsigned long long*, int, unsigned);
struct Entry { const char* name; KFn fn; int per_iter; };

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  unsigned long long* d;
  (void)hipMalloc(&d, 64);
#define E(n) {#n, k_##n, 32}
  std::vector<Entry> es = {E(fma), E(fma_dep), E(pk_fma), E(pk_mul), E(cvt_f32_i32), E(cvt_f32_ubyte), E(dot4c_i8), E(dot4c_i8_dep),
                           E(max3_i32), E(max3_f32), E(and_b32), E(and_b32_dep), E(sub_u32), E(mov_lit), E(cmp_to_sgpr), E(add_dpp_row_shr),
                           E(add_dpp_row_shr_dep), E(mov_dpp_bcast), E(perm_b32), E(cndmask), E(lshlrev_b64), E(mbcnt), E(bfrev), E(ffbl),
                           E(alignbit), E(add_co), E(s_bcnt1_b64), E(s_add), E(s_add_dep), E(s_cselect), E(readlane), E(readfirstlane),
                           {"lds_read_b32", k_lds<0>, 32}, {"lds_read_b128_stride64B", k_lds<1>, 32}, {"lds_read_b128_consecutive", k_lds<2>, 32},
                           {"median_step", k_median_step, 32}};
  const int wgs[4] = {64, 256, 512, 1024};            // 1 wavefront; 1, 2, 4 wavefronts per SIMD (4, 8, 16 on the CU)
  printf("# one workgroup on one CU, %d iterations x 32 instructions.  Columns: s_memtime counts (= shader cycles at the 2.4 GHz this part\n", iters);
  printf("# runs at) per instruction of ONE wavefront while 1 / 4 / 8 / 16 wavefronts run the same loop on the CU (0 / 1 / 2 / 4 per SIMD):\n");
  printf("# a constant row = the wavefronts do not slow each other down (the SIMD has issue slots to spare); a row that doubles = shared unit saturated\n");
  printf("%-28s %10s %10s %10s %10s\n", "# instruction", "1 wave", "1/SIMD", "2/SIMD", "4/SIMD");
  for (size_t e = 0; e < es.size(); ++e) {
    double cyc[4];
    for (int w = 0; w < 4; ++w) {
      unsigned long long best = ~0ull;
      for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(es[e].fn, dim3(1), dim3(wgs[w]), 0, 0, d, iters, 12345u + rep);
        unsigned long long t[3] = {0, 0, 0};
        (void)hipMemcpy(t, d, 24, hipMemcpyDeviceToHost);
        if (t[2] < best) best = t[2];
      }
      cyc[w] = (double)best / ((double)iters * es[e].per_iter);
    }
    printf("%-28s %10.2f %10.2f %10.2f %10.2f\n", es[e].name, cyc[0], cyc[1], cyc[2], cyc[3]);
    fflush(stdout);
  }
  {
    hipLaunchKernelGGL(k_fma, dim3(1), dim3(64), 0, 0, d, iters, 1u);
    unsigned long long t[3] = {0, 0, 0};
    (void)hipMemcpy(t, d, 24, hipMemcpyDeviceToHost);
    printf("# s_memtime: %llu counts per %llu ticks of the constant 100 MHz counter = %.0f MHz\n", t[2], t[0], (double)t[2] / (double)t[0] * 100.0);
  }
  (void)hipFree(d);
  return 0;
}
