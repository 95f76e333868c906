// Micro-benchmark: cost of LDS atomics on gfx950 by address pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int MODE>
__global__ void k(T* out, long long* cyc, int iters)
{
  __shared__ T s[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = T(0);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  int idx;
  if (MODE == 0) idx = wave * 64 + lane;               // distinct, conflict free
  else if (MODE == 1) idx = wave * 64;                   // all lanes same address
  else if (MODE == 2) idx = wave * 64 + (lane & 7);      // 8 addresses
  else if (MODE == 3) idx = wave * 64 + (lane >> 3) * 8 * 0 + (lane & 31); // 32 addr, pairs
  else idx = wave * 1024 + lane * 16;                    // distinct, 16-float stride (bank conflicts)
  T v = T(lane + 1);
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i)
  {
    atomicAdd(&s[idx], v);
    atomicAdd(&s[idx + 64 * 4], v);
    atomicAdd(&s[idx + 64 * 8], v);
    atomicAdd(&s[idx + 64 * 12], v);
  }
  __syncthreads();
  long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[threadIdx.x];
}
template <typename T, int MODE>
void run(const char* name, int waves)
{
  T* out; long long* cyc;
  hipMalloc(&out, sizeof(T) * 1024 * 256); hipMalloc(&cyc, 8 * 1024);
  const int iters = 1000;
  k<T, MODE><<<256, waves * 64>>>(out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  k<T, MODE><<<256, waves * 64>>>(out, cyc, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-28s waves/blk=%d: %8.1f clk per wave-atomic (per CU: %.1f clk/instr), %.3f ms\n", name, waves,
         double(c) / (iters * 4), double(c) / (iters * 4) / waves, ms);
  hipFree(out); hipFree(cyc);
}
int main()
{
  for (int w : {1, 4}) {
    if (w == 1) {
      run<float, 0>("f32 distinct", 1); run<float, 1>("f32 same addr", 1); run<float, 2>("f32 8 addrs", 1);
      run<float, 3>("f32 32 addrs", 1); run<float, 4>("f32 distinct stride16", 1);
      run<unsigned, 0>("u32 distinct", 1); run<unsigned, 1>("u32 same addr", 1); run<unsigned, 2>("u32 8 addrs", 1);
      run<unsigned long long, 0>("u64 distinct", 1); run<unsigned long long, 2>("u64 8 addrs", 1);
    } else {
      run<float, 0>("f32 distinct", 4); run<float, 1>("f32 same addr", 4); run<float, 2>("f32 8 addrs", 4);
      run<unsigned, 0>("u32 distinct", 4); run<unsigned, 2>("u32 8 addrs", 4);
    }
  }
  return 0;
}
