// hbm_read_steps.hip -- the read-only ceiling of k_detect's access pattern PER INPUT FORMAT: every wavefront walks its own
// contiguous chunk in steps of L x 1 KiB (L = 16-byte loads per lane and step: 8 = complex64's tile, 4 = int16 / |IQ|^2 floats,
// 2 = the 8-bit formats), D steps ahead, with k_detect's residency (LDS-limited: 20 wavefronts per CU in workgroups of four,
// 21 in workgroups of one) -- and nothing else to do.  hbm_read.hip (round 3) measured L = 8, D = 1 only.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/hbm_read_steps.hip -o /tmp/hbm_read_steps && /tmp/hbm_read_steps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int L, int D, int WPB>
__global__ void __launch_bounds__(64 * WPB, 5) k_read(const v4f* __restrict__ p, long long chunk_f4, long long n_f4, float* out) {
  extern __shared__ float s_pad[];
  const int lane = threadIdx.x & 63;
  const long long unit = (long long)blockIdx.x * WPB + (threadIdx.x >> 6);
  long long i = unit * chunk_f4, e = i + chunk_f4;
  if (e > n_f4) e = n_f4;
  constexpr int S = 64 * L;                                    // float4 per step
  float m = 0.f;
  v4f q[D][L];
  const long long last = e - S;                                // start of the last whole step
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const long long a = i + (long long)d * S <= last ? i + (long long)d * S : (last >= i ? last : i);
#pragma unroll
    for (int k = 0; k < L; ++k) q[d][k] = __builtin_nontemporal_load(p + a + k * 64 + lane);
  }
  for (; i + (long long)D * S <= e; i += (long long)D * S) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      v4f r[L];
#pragma unroll
      for (int k = 0; k < L; ++k) r[k] = q[d][k];
      long long nx = i + (long long)(D + d) * S;
      if (nx > last) nx = last;
#pragma unroll
      for (int k = 0; k < L; ++k) q[d][k] = __builtin_nontemporal_load(p + nx + k * 64 + lane);
#pragma unroll
      for (int k = 0; k < L; ++k) m = fmaxf(fmaxf(m, fmaxf(r[k].x, r[k].y)), fmaxf(r[k].z, r[k].w));
    }
  }
  if (m == 12345.f) { out[unit] = m; s_pad[lane] = m; }
}

template <int L, int D, int WPB>
static void run(const v4f* d, float* o, size_t bytes, int cus, int rounds) {
  const long long n_f4 = bytes / 16;
  const size_t lds = WPB == 4 ? 30848 : 7072;                  // k_detect's LDS per workgroup: 5 x 4 resp. 21 x 1 wavefronts per CU
  const int per_cu = WPB == 4 ? 5 : 21;
  const int grid = cus * per_cu * rounds;
  const long long units = (long long)grid * WPB;
  long long chunk = (n_f4 + units - 1) / units;
  const long long S = 64 * L * D;
  chunk = (chunk + S - 1) / S * S;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_read<L, D, WPB>), dim3(grid), dim3(64 * WPB), lds, 0, d, chunk, n_f4, o);
  float best = 1e9f, sum = 0.f;
  const int reps = 10;
  for (int r = 0; r < reps; ++r) {                              // one launch per event pair: what bench.py's per-launch kernel time sees
    hipEventRecord(a);
    hipLaunchKernelGGL((k_read<L, D, WPB>), dim3(grid), dim3(64 * WPB), lds, 0, d, chunk, n_f4, o);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    best = ms < best ? ms : best; sum += ms;
  }
  const float ms = sum / reps;
  printf("L=%d KiB/step D=%d ahead  %d wave/WG x %2d WG/CU x %d rounds  2^%d B: avg %.4f ms = %.3f of 8 TB/s (best %.4f = %.3f)\n", L, D, WPB, per_cu,
         rounds, (int)__builtin_ctzll(bytes), ms, bytes / (ms * 1e-3) / 8e12, best, bytes / (best * 1e-3) / 8e12);
  hipEventDestroy(a); hipEventDestroy(b);
}

int main(int argc, char** argv) {
  const size_t cap = (size_t)1 << 33;
  v4f* d; float* o;
  hipMalloc(&d, cap); hipMalloc(&o, 1 << 22);
  hipMemset(d, 0, cap);
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  for (int rounds : {1, 8}) {
    // complex64: 2^30 samples = 2^33 bytes
    run<8, 1, 4>(d, o, (size_t)1 << 33, cus, rounds);
    run<8, 2, 4>(d, o, (size_t)1 << 33, cus, rounds);
    // int16 / |IQ|^2 floats: 2^32 bytes
    run<4, 1, 4>(d, o, (size_t)1 << 32, cus, rounds);
    run<4, 2, 4>(d, o, (size_t)1 << 32, cus, rounds);
    run<4, 3, 4>(d, o, (size_t)1 << 32, cus, rounds);
    run<8, 1, 4>(d, o, (size_t)1 << 32, cus, rounds);
    run<4, 1, 1>(d, o, (size_t)1 << 32, cus, rounds);
    run<4, 2, 1>(d, o, (size_t)1 << 32, cus, rounds);
    // 8-bit: 2^31 bytes
    run<2, 1, 1>(d, o, (size_t)1 << 31, cus, rounds);
    run<2, 2, 1>(d, o, (size_t)1 << 31, cus, rounds);
    run<2, 4, 1>(d, o, (size_t)1 << 31, cus, rounds);
    run<2, 2, 4>(d, o, (size_t)1 << 31, cus, rounds);
    run<8, 1, 4>(d, o, (size_t)1 << 31, cus, rounds);
    // config 3/4/5: 2^28 complex64 = 2^31 bytes
    run<8, 2, 4>(d, o, (size_t)1 << 31, cus, rounds);
  }
  return 0;
}
