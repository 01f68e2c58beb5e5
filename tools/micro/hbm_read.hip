// hbm_read.hip -- what a READ-ONLY stream reaches on this GPU with k_detect's access pattern and nothing else to do: every
// wavefront walks its own contiguous chunk in 8 KiB steps (8 x global_load_dwordx4 nt per lane, one step ahead), keeps the
// running maximum, writes one word.  The ceiling to hold k_detect's burst-free rate against.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/hbm_read.hip -o /tmp/hbm_read && /tmp/hbm_read [log2 bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256, 5) k_read(const v4f* __restrict__ p, long long chunk_f4, long long n_f4, float* out) {
  const int lane = threadIdx.x & 63;
  const long long unit = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  long long i = unit * chunk_f4, e = i + chunk_f4;
  if (e > n_f4) e = n_f4;
  float m = 0.f;
  v4f q[8];
  if (i + 512 <= e)
    for (int k = 0; k < 8; ++k) q[k] = __builtin_nontemporal_load(p + i + k * 64 + lane);
  for (; i + 512 <= e; i += 512) {
    v4f r[8];
    for (int k = 0; k < 8; ++k) r[k] = q[k];
    const long long nx = (i + 1024 <= e) ? i + 512 : i;
    for (int k = 0; k < 8; ++k) q[k] = __builtin_nontemporal_load(p + nx + k * 64 + lane);
    for (int k = 0; k < 8; ++k) m = fmaxf(fmaxf(m, fmaxf(r[k].x, r[k].y)), fmaxf(r[k].z, r[k].w));
  }
  if (m == 12345.f) out[unit] = m;
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 33;
  const size_t bytes = (size_t)1 << lg;
  v4f* d; float* o;
  hipMalloc(&d, bytes); hipMalloc(&o, 1 << 20);
  hipMemset(d, 0, bytes);
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const long long n_f4 = bytes / 16;
  for (int wg_per_cu = 3; wg_per_cu <= 8; ++wg_per_cu) {
    const int grid = pr.multiProcessorCount * wg_per_cu;
    const long long units = (long long)grid * 4;
    long long chunk = (n_f4 + units - 1) / units;
    chunk = (chunk + 511) / 512 * 512;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, d, chunk, n_f4, o);
    hipEventRecord(a);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, d, chunk, n_f4, o);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("read-only stream, 2^%d bytes, %d workgroups per CU (grid %d, launch bound 5/SIMD): %.3f ms per pass = %.0f GB/s = %.3f of 8 TB/s\n",
           lg, wg_per_cu, grid, ms / reps, bytes / (ms / reps * 1e-3) / 1e9, bytes / (ms / reps * 1e-3) / 8e12);
  }
  return 0;
}
