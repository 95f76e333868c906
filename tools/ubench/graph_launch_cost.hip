// Host cost of hipGraphLaunch for linear graphs (one captured stream) of N
// small kernels against a forked graph of the same kernels, plain launches,
// and event record / wait.  One frame per call is bound by what the host can
// enqueue: this decides between one forked graph and several linear ones.
//   hipcc --offload-arch=gfx950 -O3 -o graph_launch_cost graph_launch_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void tiny(float* p, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    p[i] = p[i] * 1.0001f + 1.f;
}

static double now_us()
{
  return std::chrono::duration<double, std::micro>(
             std::chrono::steady_clock::now().time_since_epoch())
      .count();
}

int main()
{
  float* d;
  hipMalloc(&d, 1 << 22);
  hipStream_t s, s2, s3;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s3, hipStreamNonBlocking);
  hipEvent_t e1, e2, e3;
  hipEventCreateWithFlags(&e1, hipEventDisableTiming);
  hipEventCreateWithFlags(&e2, hipEventDisableTiming);
  hipEventCreateWithFlags(&e3, hipEventDisableTiming);
  const int reps = 200;
  for (int n : {1, 2, 4, 8, 16, 32})
  {
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
    for (int i = 0; i < n; ++i)
      tiny<<<64, 256, 0, s>>>(d, 16384);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 10; ++i)
      hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    double host = 0, total = 0;
    for (int r = 0; r < reps; ++r)
    {
      const double t0 = now_us();
      hipGraphLaunch(ge, s);
      const double t1 = now_us();
      hipStreamSynchronize(s);
      const double t2 = now_us();
      host += t1 - t0;
      total += t2 - t0;
    }
    // forked: the same kernels alternating on two captured streams
    hipGraph_t gf;
    hipGraphExec_t gef;
    hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
    tiny<<<64, 256, 0, s>>>(d, 16384);
    hipEventRecord(e1, s);
    hipStreamWaitEvent(s2, e1, 0);
    for (int i = 1; i < n; ++i)
      tiny<<<64, 256, 0, (i & 1) ? s2 : s>>>(d + (i & 1) * 65536, 16384);
    hipEventRecord(e2, s2);
    hipStreamWaitEvent(s, e2, 0);
    hipStreamEndCapture(s, &gf);
    hipGraphInstantiate(&gef, gf, nullptr, nullptr, 0);
    for (int i = 0; i < 10; ++i)
      hipGraphLaunch(gef, s);
    hipStreamSynchronize(s);
    double hostf = 0, totalf = 0;
    for (int r = 0; r < reps; ++r)
    {
      const double t0 = now_us();
      hipGraphLaunch(gef, s);
      const double t1 = now_us();
      hipStreamSynchronize(s);
      const double t2 = now_us();
      hostf += t1 - t0;
      totalf += t2 - t0;
    }
    // plain launches
    double hostp = 0, totalp = 0;
    for (int r = 0; r < reps; ++r)
    {
      const double t0 = now_us();
      for (int i = 0; i < n; ++i)
        tiny<<<64, 256, 0, s>>>(d, 16384);
      const double t1 = now_us();
      hipStreamSynchronize(s);
      const double t2 = now_us();
      hostp += t1 - t0;
      totalp += t2 - t0;
    }
    printf("n=%2d  linear graph: host %6.1f total %6.1f | forked graph: host %6.1f total %6.1f | plain: host %6.1f total %6.1f us\n",
           n, host / reps, total / reps, hostf / reps, totalf / reps, hostp / reps, totalp / reps);
  }
  // several linear graphs in a row on two streams with event hand-offs
  {
    hipGraph_t g[4];
    hipGraphExec_t ge[4];
    hipStream_t ss[4] = {s, s2, s, s};
    for (int k = 0; k < 4; ++k)
    {
      hipStreamBeginCapture(ss[k], hipStreamCaptureModeRelaxed);
      for (int i = 0; i < 8; ++i)
        tiny<<<64, 256, 0, ss[k]>>>(d + k * 65536, 16384);
      hipStreamEndCapture(ss[k], &g[k]);
      hipGraphInstantiate(&ge[k], g[k], nullptr, nullptr, 0);
    }
    auto run = [&]() {
      hipGraphLaunch(ge[0], s);
      hipEventRecord(e1, s);
      hipStreamWaitEvent(s2, e1, 0);
      hipGraphLaunch(ge[1], s2);
      hipEventRecord(e2, s2);
      hipGraphLaunch(ge[2], s);
      hipStreamWaitEvent(s, e2, 0);
      hipGraphLaunch(ge[3], s);
    };
    for (int i = 0; i < 10; ++i)
      run();
    hipStreamSynchronize(s);
    double host = 0, total = 0;
    for (int r = 0; r < reps; ++r)
    {
      const double t0 = now_us();
      run();
      const double t1 = now_us();
      hipStreamSynchronize(s);
      const double t2 = now_us();
      host += t1 - t0;
      total += t2 - t0;
    }
    printf("4 linear graphs x 8 kernels on 2 streams + 2 records + 2 waits: host %6.1f total %6.1f us\n",
           host / reps, total / reps);
    double ev = 0;
    for (int r = 0; r < reps; ++r)
    {
      const double t0 = now_us();
      hipEventRecord(e3, s);
      hipStreamWaitEvent(s3, e3, 0);
      ev += now_us() - t0;
    }
    hipDeviceSynchronize();
    printf("event record + wait: %5.1f us\n", ev / reps);
  }
  return 0;
}
