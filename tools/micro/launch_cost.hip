// launch_cost.hip -- what the HOST pays per pass for the launch sequence of a bulk pass, and what a captured hipGraph would
// pay for the same sequence (round-5 question: "one hipGraph per (slot, plan shape)").
//
// The sequence of adsb_hip.hip: enqueue / enqueue_tail, with empty kernels of the same argument sizes:
//   compute stream:  k1(args 216 B)  record(dep)
//   tail stream:     wait(dep)  k2(216 B + 40 B)  k3(40 B)  k4(40 B)  k5(100 B)  record(done)
// Measured per pass, three passes in flight (the host never waits for the GPU except through the oldest pass's `done`):
//   A  direct launches (what the library does)
//   B  hipGraphLaunch of the instantiated sequence, arguments UNCHANGED between launches
//   C  like B, with the kernel arguments of all five nodes replaced before every launch (hipGraphExecKernelNodeSetParams):
//      what a pass needs -- the sample pointer, the length and the stream offsets change with every call
// and the pieces: one launch, one event record, one stream wait.
//
//   hipcc --offload-arch=gfx950 -O2 tools/micro/launch_cost.hip -o /tmp/launch_cost && /tmp/launch_cost
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                              \
  do {                                                                                      \
    hipError_t e_ = (x);                                                                    \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
  } while (0)

struct Big { long long v[27]; };     // 216 bytes: DetectArgs
struct Mid { long long v[12]; };     // ~100 bytes: k_compact's arguments
struct Small { long long v[5]; };    // 40 bytes

__global__ void k1(Big a, int* sink) { if (a.v[0] == -1) *sink = 1; }
__global__ void k2(Big a, Small b, int* sink) { if (a.v[0] + b.v[0] == -1) *sink = 1; }
__global__ void k3(Small a, int* sink) { if (a.v[0] == -1) *sink = 1; }
__global__ void k5(Mid a, int* sink) { if (a.v[0] == -1) *sink = 1; }

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  CHK(hipSetDevice(0));
  hipStream_t cs, ts;
  CHK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  CHK(hipStreamCreateWithFlags(&ts, hipStreamNonBlocking));
  int* sink = nullptr;
  CHK(hipMalloc(&sink, 4));
  constexpr int kSlots = 3;
  hipEvent_t dep[kSlots], done[kSlots];
  for (int i = 0; i < kSlots; ++i) {
    CHK(hipEventCreateWithFlags(&dep[i], hipEventDisableTiming));
    CHK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
  }
  Big big{}; Mid mid{}; Small sm{};
  const int reps = 3000;
  const dim3 g1(2048), g2(32), gt(2048), b(256);

  auto pass_direct = [&](int s) {
    hipLaunchKernelGGL(k1, g1, b, 0, cs, big, sink);
    CHK(hipEventRecord(dep[s], cs));
    CHK(hipStreamWaitEvent(ts, dep[s], 0));
    hipLaunchKernelGGL(k2, g2, b, 8192, ts, big, sm, sink);
    hipLaunchKernelGGL(k3, gt, b, 0, ts, sm, sink);
    hipLaunchKernelGGL(k3, gt, b, 0, ts, sm, sink);
    hipLaunchKernelGGL(k5, gt, b, 0, ts, mid, sink);
    CHK(hipEventRecord(done[s], ts));
  };

  // --- A: direct, three in flight
  auto run = [&](const char* name, auto&& submit, auto&& wait) {
    for (int w = 0; w < 2; ++w) {
      int inflight = 0, head = 0, tail = 0;
      double t_sub = 0, t_wait = 0;
      CHK(hipDeviceSynchronize());
      const double t0 = now_us();
      for (int i = 0; i < reps; ++i) {
        const double a = now_us();
        submit(head); head = (head + 1) % kSlots; ++inflight;
        const double c = now_us();
        t_sub += c - a;
        if (inflight == kSlots) { wait(tail); tail = (tail + 1) % kSlots; --inflight; t_wait += now_us() - c; }
      }
      while (inflight) { wait(tail); tail = (tail + 1) % kSlots; --inflight; }
      const double t1 = now_us();
      if (w == 1) printf("%-58s wall %6.2f us/pass   submit %6.2f   wait %6.2f\n", name, (t1 - t0) / reps, t_sub / reps, t_wait / reps);
    }
  };
  run("A direct: 5 launches + 2 records + 1 stream wait", pass_direct, [&](int s) { CHK(hipEventSynchronize(done[s])); });

  // --- graphs: one per slot, captured from the same sequence (two streams: the tail forks off the compute stream)
  hipGraph_t graph[kSlots];
  hipGraphExec_t exec[kSlots];
  std::vector<hipGraphNode_t> knodes[kSlots];
  for (int s = 0; s < kSlots; ++s) {
    CHK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k1, g1, b, 0, cs, big, sink);
    CHK(hipEventRecord(dep[s], cs));
    CHK(hipStreamWaitEvent(ts, dep[s], 0));
    hipLaunchKernelGGL(k2, g2, b, 8192, ts, big, sm, sink);
    hipLaunchKernelGGL(k3, gt, b, 0, ts, sm, sink);
    hipLaunchKernelGGL(k3, gt, b, 0, ts, sm, sink);
    hipLaunchKernelGGL(k5, gt, b, 0, ts, mid, sink);
    CHK(hipEventRecord(done[s], ts));
    CHK(hipStreamWaitEvent(cs, done[s], 0));              // join (a capture must end on its origin stream)
    CHK(hipStreamEndCapture(cs, &graph[s]));
    CHK(hipGraphInstantiate(&exec[s], graph[s], nullptr, nullptr, 0));
    size_t nn = 0;
    CHK(hipGraphGetNodes(graph[s], nullptr, &nn));
    std::vector<hipGraphNode_t> all(nn);
    CHK(hipGraphGetNodes(graph[s], all.data(), &nn));
    for (hipGraphNode_t n : all) {
      hipGraphNodeType t;
      CHK(hipGraphNodeGetType(n, &t));
      if (t == hipGraphNodeTypeKernel) knodes[s].push_back(n);
    }
    if (s == 0) printf("graph: %zu nodes, %zu kernel nodes\n", nn, knodes[s].size());
  }
  // a launched graph serialises the whole pass on the launch stream: the next pass's k1 cannot start beside this pass's tail
  // unless consecutive passes go to different streams -- one launch stream per slot
  hipStream_t gs[kSlots];
  for (int s = 0; s < kSlots; ++s) CHK(hipStreamCreateWithFlags(&gs[s], hipStreamNonBlocking));
  run("B hipGraphLaunch, arguments unchanged", [&](int s) { CHK(hipGraphLaunch(exec[s], gs[s])); },
      [&](int s) { CHK(hipStreamSynchronize(gs[s])); });
  run("C hipGraphLaunch + 5 x hipGraphExecKernelNodeSetParams", [&](int s) {
        for (hipGraphNode_t n : knodes[s]) {
          hipKernelNodeParams p;
          CHK(hipGraphKernelNodeGetParams(n, &p));
          CHK(hipGraphExecKernelNodeSetParams(exec[s], n, &p));
        }
        CHK(hipGraphLaunch(exec[s], gs[s]));
      },
      [&](int s) { CHK(hipStreamSynchronize(gs[s])); });

  // --- the pieces (host time per call, stream kept busy so nothing blocks)
  auto piece = [&](const char* name, auto&& f) {
    CHK(hipDeviceSynchronize());
    const double t0 = now_us();
    for (int i = 0; i < reps; ++i) { f(i); if ((i & 255) == 255) CHK(hipStreamSynchronize(cs)); }
    const double t1 = now_us();
    CHK(hipDeviceSynchronize());
    printf("%-58s %6.2f us/call (incl. a stream synchronise every 256)\n", name, (t1 - t0) / reps);
  };
  piece("hipLaunchKernelGGL, 2048 workgroups, 216-byte arguments", [&](int) { hipLaunchKernelGGL(k1, g1, b, 0, cs, big, sink); });
  piece("hipLaunchKernelGGL, 32 workgroups, 40-byte arguments", [&](int) { hipLaunchKernelGGL(k3, g2, b, 0, cs, sm, sink); });
  piece("hipEventRecord (timing disabled)", [&](int i) { CHK(hipEventRecord(dep[i % kSlots], cs)); });
  piece("hipEventRecord + hipStreamWaitEvent on a second stream", [&](int i) {
    CHK(hipEventRecord(dep[i % kSlots], cs));
    CHK(hipStreamWaitEvent(ts, dep[i % kSlots], 0));
  });
  {
    hipEvent_t tev[2];
    CHK(hipEventCreate(&tev[0])); CHK(hipEventCreate(&tev[1]));
    piece("hipEventRecord (timing enabled)", [&](int i) { CHK(hipEventRecord(tev[i & 1], cs)); });
  }
  CHK(hipDeviceSynchronize());
  // completion latency: one empty kernel, then how long until the host sees it (event synchronise vs a polled pinned word)
  {
    const int n = 500;
    double t = 0;
    for (int i = 0; i < n; ++i) {
      hipLaunchKernelGGL(k3, dim3(1), b, 0, cs, sm, sink);
      CHK(hipEventRecord(done[0], cs));
      const double a = now_us();
      CHK(hipEventSynchronize(done[0]));
      t += now_us() - a;
    }
    printf("%-58s %6.2f us (launch of an empty kernel -> hipEventSynchronize returns, measured from after the record)\n",
           "completion through an event", t / n);
  }
  return 0;
}
