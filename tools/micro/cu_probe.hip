// cu_probe.hip -- what changes on the GPU when a run is slower under `rocprofv3 --kernel-trace`?  A long, instruction-bound
// kernel with k_detect's launch shape (tens of thousands of 256-thread workgroups, five per CU, ~0.6 ms per launch) launched
// back to back; every workgroup records where it ran (XCC, SE / SH / CU from HW_ID) and when (the constant 100 MHz counter).
// Per launch: HIP-event duration, distinct CUs and XCCs used, median workgroup duration, mean number of workgroups resident
// at once.  Same per-workgroup time with fewer resident workgroups = CUs / occupancy taken away; longer per-workgroup time at
// the same residency = a lower shader clock (or time slicing).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/cu_probe.hip -o /tmp/cu_probe && /tmp/cu_probe [launches] [iters]
//   rocprofv3 --kernel-trace -d /tmp/x -- /tmp/cu_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

struct Rec { unsigned long long t0, t1; unsigned hwid, xcc; };

__global__ void __launch_bounds__(256, 5) k_busy(Rec* out, int iters, float seed) {
  __shared__ float pad[7000];                      // 28 KB: five workgroups per CU, like k_detect
  const unsigned long long t0 = wall_clock64();
  float a = seed + threadIdx.x, b = 1.0001f, c = 0.5f;
  for (int i = 0; i < iters; ++i) {                // dependent FMAs: bound by instruction issue
    a = __builtin_fmaf(a, b, c); b = __builtin_fmaf(b, 0.9999f, 1e-6f); c = __builtin_fmaf(c, a, -b);
    a = __builtin_fmaf(a, 0.5f, c); c = __builtin_fmaf(c, 0.25f, a);
  }
  pad[threadIdx.x] = a + c;
  __syncthreads();
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    Rec r; r.t0 = t0; r.t1 = t1; r.hwid = hwid; r.xcc = xcc & 15u;
    if (pad[17] == 123.456f) r.t0 = 0;            // keep the arithmetic alive
    out[blockIdx.x] = r;
  }
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 40;
  const int iters = argc > 2 ? atoi(argv[2]) : 3000;
  const int nwg = 40960;
  Rec* d;
  hipMalloc(&d, sizeof(Rec) * (size_t)nwg * launches);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  std::vector<hipEvent_t> ev(launches + 1);
  for (auto& e : ev) hipEventCreate(&e);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_busy, dim3(nwg), dim3(256), 0, st, d, iters, 1.0f);
  hipStreamSynchronize(st);
  hipEventRecord(ev[0], st);
  for (int l = 0; l < launches; ++l) {
    hipLaunchKernelGGL(k_busy, dim3(nwg), dim3(256), 0, st, d + (size_t)l * nwg, iters, 1.0f);
    hipEventRecord(ev[l + 1], st);
  }
  hipStreamSynchronize(st);
  std::vector<Rec> h((size_t)nwg * launches);
  hipMemcpy(h.data(), d, h.size() * sizeof(Rec), hipMemcpyDeviceToHost);
  printf("# %d launches of %d workgroups x 256 threads, %d iterations; counter = 100 MHz\n", launches, nwg, iters);
  printf("# launch  event_ms  span_ms  xccs  cus  wg_median_us  wg_p95_us  resident_wgs\n");
  for (int l = 0; l < launches; ++l) {
    float ms = 0; hipEventElapsedTime(&ms, ev[l], ev[l + 1]);
    const Rec* r = h.data() + (size_t)l * nwg;
    std::set<unsigned> cus, xccs;
    std::vector<double> dur(nwg);
    unsigned long long lo = ~0ull, hi = 0; double sum = 0;
    for (int i = 0; i < nwg; ++i) {
      cus.insert((r[i].xcc << 16) | (r[i].hwid & 0xFF00u));            // XCC, SE / SH / CU id
      xccs.insert(r[i].xcc);
      dur[i] = (double)(r[i].t1 - r[i].t0) / 100.0;                    // microseconds
      sum += dur[i];
      lo = std::min(lo, r[i].t0); hi = std::max(hi, r[i].t1);
    }
    std::sort(dur.begin(), dur.end());
    const double span_us = (double)(hi - lo) / 100.0;
    printf("%4d  %8.4f  %8.4f  %3zu  %4zu  %10.2f  %10.2f  %8.1f\n", l, ms, span_us / 1e3, xccs.size(), cus.size(), dur[nwg / 2],
           dur[(size_t)(nwg * 0.95)], sum / span_us);
  }
  return 0;
}
