// Can uploads and read-backs cross PCIe at the same time on this platform?
// The host-to-host SIFT step moves 0.53 GB up (64 float frames) and 0.14 GB
// down (keypoints + descriptors); round 2 measured 9.26 + 2.52 = 11.78 ms with
// both in flight through the copy engines, i.e. no overlap at all.  This probe
// times every pairing of {copy engine, copy kernel} for the two directions.
//   hipcc --offload-arch=gfx950 -O3 -o pcie_duplex pcie_duplex.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <chrono>

__global__ __launch_bounds__(256) void copy16(const uint4* __restrict__ s,
                                              uint4* __restrict__ d, size_t n)
{
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    d[i] = s[i];
}

static double now_ms()
{
  return std::chrono::duration<double, std::milli>(
             std::chrono::steady_clock::now().time_since_epoch())
      .count();
}

int main()
{
  const size_t up = size_t(64) * 1920 * 1080 * 4, down = size_t(64) * 4400 * 568;
  void *h_up, *h_down, *d_up, *d_down;
  hipHostMalloc(&h_up, up, hipHostMallocDefault);
  hipHostMalloc(&h_down, down, hipHostMallocDefault);
  hipMalloc(&d_up, up);
  hipMalloc(&d_down, down);
  hipMemset(d_down, 1, down);
  std::memset(h_up, 2, up);
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  auto up_dma = [&]() { hipMemcpyAsync(d_up, h_up, up, hipMemcpyHostToDevice, s1); };
  auto down_dma = [&]() { hipMemcpyAsync(h_down, d_down, down, hipMemcpyDeviceToHost, s2); };
  int blocks = 64;
  auto up_k = [&]() { copy16<<<blocks, 256, 0, s1>>>((const uint4*) h_up, (uint4*) d_up, up / 16); };
  auto down_k = [&]() { copy16<<<blocks, 256, 0, s2>>>((const uint4*) d_down, (uint4*) h_down, down / 16); };
  auto timeit = [&](const char* name, auto f, auto g, bool both) {
    for (int w = 0; w < 2; ++w)
    {
      f();
      if (both)
        g();
      hipDeviceSynchronize();
    }
    const int reps = 5;
    const double t0 = now_ms();
    for (int r = 0; r < reps; ++r)
    {
      f();
      if (both)
        g();
      hipStreamSynchronize(s1);
      hipStreamSynchronize(s2);
    }
    const double ms = (now_ms() - t0) / reps;
    printf("%-48s %7.2f ms\n", name, ms);
  };
  timeit("up, copy engine (0.53 GB)", up_dma, down_dma, false);
  timeit("down, copy engine (0.14 GB)", down_dma, up_dma, false);
  timeit("up + down, copy engines", up_dma, down_dma, true);
  for (int b : {8, 32, 64, 256})
  {
    blocks = b;
    char name[96];
    snprintf(name, sizeof name, "up, copy kernel, %d blocks", b);
    timeit(name, up_k, down_k, false);
    snprintf(name, sizeof name, "down, copy kernel, %d blocks", b);
    timeit(name, down_k, up_k, false);
    snprintf(name, sizeof name, "up kernel + down copy engine, %d blocks", b);
    timeit(name, up_k, down_dma, true);
    snprintf(name, sizeof name, "up copy engine + down kernel, %d blocks", b);
    timeit(name, up_dma, down_k, true);
    snprintf(name, sizeof name, "up kernel + down kernel, %d blocks", b);
    timeit(name, up_k, down_k, true);
  }
  // Does an upload in flight slow an HBM-streaming kernel down?  (The SIFT
  // step's streaming kernels ran 1.5-2.7x longer under the float32 upload.)
  {
    void *a, *b;
    const size_t n = size_t(1) << 30;
    hipMalloc(&a, n);
    hipMalloc(&b, n);
    hipStream_t s3;
    hipStreamCreateWithFlags(&s3, hipStreamNonBlocking);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto stream_ms = [&](int reps) {
      hipEventRecord(e0, s3);
      for (int r = 0; r < reps; ++r)
        copy16<<<8192, 256, 0, s3>>>((const uint4*) a, (uint4*) b, n / 16);
      hipEventRecord(e1, s3);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      return ms / reps;
    };
    stream_ms(2);
    printf("1 GiB device copy alone                          %7.3f ms (%.0f GB/s)\n", stream_ms(10),
           2.0 * n / stream_ms(10) * 1e-6);
    up_dma();
    const float m1 = stream_ms(10);
    hipDeviceSynchronize();
    printf("1 GiB device copy under a copy-engine upload     %7.3f ms\n", m1);
    down_dma();
    const float m2 = stream_ms(4);
    hipDeviceSynchronize();
    printf("1 GiB device copy under a copy-engine read-back  %7.3f ms\n", m2);
    blocks = 32;
    up_k();
    const float m3 = stream_ms(10);
    hipDeviceSynchronize();
    printf("1 GiB device copy under a copy-KERNEL upload     %7.3f ms\n", m3);
    // upload from hipHostRegister'ed memory instead of hipHostMalloc
    void* reg = aligned_alloc(4096, up);
    std::memset(reg, 3, up);
    hipHostRegister(reg, up, hipHostRegisterDefault);
    hipMemcpyAsync(d_up, reg, up, hipMemcpyHostToDevice, s1);
    const float m4 = stream_ms(10);
    hipDeviceSynchronize();
    printf("1 GiB device copy under an upload from registered memory %7.3f ms\n", m4);
    const double t0 = now_ms();
    hipMemcpyAsync(d_up, reg, up, hipMemcpyHostToDevice, s1);
    hipStreamSynchronize(s1);
    printf("upload from registered memory                    %7.2f ms\n", now_ms() - t0);
  }
  return 0;
}
