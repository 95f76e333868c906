// When do the kernels of several linear graphs really start?  Every kernel
// spins for ~5 us and stamps wall_clock64() (100 MHz) at entry and exit, so the
// schedule can be read without a profiler.
//   hipcc --offload-arch=gfx950 -O3 -o graph_sched graph_sched.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin(long long* stamps, int id, int ticks)
{
  if (threadIdx.x == 0 && blockIdx.x == 0)
  {
    const long long t0 = wall_clock64();
    stamps[2 * id] = t0;
    while (wall_clock64() - t0 < ticks)
      ;
    stamps[2 * id + 1] = wall_clock64();
  }
}
__global__ void stamp0(long long* stamps) { stamps[0] = wall_clock64(); }

int main()
{
  long long* d;
  hipMalloc(&d, 4096);
  hipStream_t s, s2;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t e1, e2;
  hipEventCreateWithFlags(&e1, hipEventDisableTiming);
  hipEventCreateWithFlags(&e2, hipEventDisableTiming);
  const int NK = 6, TICKS = 500;  // 6 kernels of 5 us per segment
  hipGraph_t g[4];
  hipGraphExec_t ge[4];
  hipStream_t ss[4] = {s, s2, s, s};
  for (int k = 0; k < 4; ++k)
  {
    hipStreamBeginCapture(ss[k], hipStreamCaptureModeRelaxed);
    for (int i = 0; i < NK; ++i)
      spin<<<1, 64, 0, ss[k]>>>(d, 1 + k * NK + i, TICKS);
    hipStreamEndCapture(ss[k], &g[k]);
    hipGraphInstantiate(&ge[k], g[k], nullptr, nullptr, 0);
  }
  auto seg_plain = [&](int k) {
    for (int i = 0; i < NK; ++i)
      spin<<<1, 64, 0, ss[k]>>>(d, 1 + k * NK + i, TICKS);
  };
  auto report = [&](const char* name) {
    hipDeviceSynchronize();
    std::vector<long long> h(2 + 2 * 4 * NK);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    long long t0 = h[2];
    for (size_t i = 2; i < h.size(); ++i)
      if (h[i] && h[i] < t0)
        t0 = h[i];
    printf("%-44s", name);
    for (int k = 0; k < 4; ++k)
      printf(" seg%d %6.1f..%6.1f", k, (h[2 * (1 + k * NK)] - t0) / 100.0,
             (h[2 * (1 + k * NK + NK - 1) + 1] - t0) / 100.0);
    printf("\n");
    hipMemset(d, 0, 4096);
    hipDeviceSynchronize();
  };
  for (int rep = 0; rep < 2; ++rep)
  {
    // 1. two linear graphs back to back on one stream
    hipGraphLaunch(ge[0], s);
    hipGraphLaunch(ge[2], s);
    hipGraphLaunch(ge[3], s);
    report("graphs 0,2,3 on one stream");
    hipGraphLaunch(ge[0], s);
    hipEventRecord(e1, s);
    hipGraphLaunch(ge[2], s);
    hipGraphLaunch(ge[3], s);
    report("graphs 0,record,2,3 on one stream");
    hipGraphLaunch(ge[0], s);
    hipEventRecord(e1, s);
    hipStreamWaitEvent(s2, e1, 0);
    hipGraphLaunch(ge[1], s2);
    hipEventRecord(e2, s2);
    hipGraphLaunch(ge[2], s);
    hipStreamWaitEvent(s, e2, 0);
    hipGraphLaunch(ge[3], s);
    report("graphs 0 | 1 (s2) , 2 | 3  with events");
    seg_plain(0);
    hipEventRecord(e1, s);
    hipStreamWaitEvent(s2, e1, 0);
    seg_plain(1);
    hipEventRecord(e2, s2);
    seg_plain(2);
    hipStreamWaitEvent(s, e2, 0);
    seg_plain(3);
    report("plain launches, same pattern");
    hipGraphLaunch(ge[0], s);
    hipEventRecord(e1, s);
    hipGraphLaunch(ge[2], s);
    hipStreamWaitEvent(s2, e1, 0);
    hipGraphLaunch(ge[1], s2);
    hipEventRecord(e2, s2);
    hipStreamWaitEvent(s, e2, 0);
    hipGraphLaunch(ge[3], s);
    report("graphs 0, 2 first, then 1 (s2), 3");
  }
  return 0;
}
