// Correctness probe: do LDS atomics with same-address conflicts inside one
// wave instruction sum correctly (u32 vs u64, with carries across bit 32)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <typename T>
__global__ void k(T* out, int naddr, int iters, unsigned long long big)
{
  __shared__ T s[64];
  if (threadIdx.x < 64) s[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  for (int i = 0; i < iters; ++i)
  {
    T v = T(big) + T(lane * 7 + i);
    atomicAdd(&s[lane % naddr], v);
  }
  __syncthreads();
  if (threadIdx.x < 64) out[threadIdx.x] = s[threadIdx.x];
}
template <typename T> void run(const char* name, int naddr, unsigned long long big)
{
  T* d; hipMalloc(&d, 64 * sizeof(T));
  const int iters = 1000;
  k<T><<<1, 64>>>(d, naddr, iters, big);
  std::vector<T> h(64); hipMemcpy(h.data(), d, 64 * sizeof(T), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int a = 0; a < naddr; ++a) {
    T want = 0;
    for (int lane = a; lane < 64; lane += naddr) for (int i = 0; i < iters; ++i) want += T(big) + T(lane * 7 + i);
    if (h[a] != want) { if (bad < 3) printf("  addr %d got %llx want %llx\n", a, (unsigned long long) h[a], (unsigned long long) want); ++bad; }
  }
  printf("%s naddr=%d big=%llx: %s (%d bad)\n", name, naddr, big, bad ? "WRONG" : "ok", bad);
  hipFree(d);
}
int main()
{
  for (int n : {64, 32, 8, 3, 1}) {
    run<unsigned>("u32", n, 0x00f00000ull);
    run<unsigned long long>("u64", n, 0x00f00000ull);
    run<unsigned long long>("u64", n, 0xfffffffffff00000ull);  // negative hi, carries
    run<unsigned long long>("u64", n, 0x00000000fff00000ull);
  }
}
