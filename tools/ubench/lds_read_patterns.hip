// Micro-benchmark: LDS cycles of ds_read_b64 / ds_read_b32 / ds_or_b64 on
// gfx950 by address pattern and active-lane count - the per-keypoint kernels
// are co-bound by the LDS data path, so the patterns of
// their inner loops are chosen from these numbers.
//   per-CU cost of one wave instruction = elapsed cycles / (instructions
//   issued by all waves of the workgroup), with 8 waves per workgroup so that
//   the LDS pipe, not one wave's latency, is what is measured.
#include <hip/hip_runtime.h>
#include <cstdio>

enum Pattern
{
  kContiguous,   // lane l reads slot l
  kSameAddress,  // every lane reads slot 0
  kStride2,      // slot 2 l   (16-byte stride for b64)
  kStride4,      // slot 4 l   (32-byte stride for b64)
  kStride4Skew,  // slot 4 l + 2 (l >> 3)
  kRandom,       // a fixed pseudo-random slot per lane
  kFourActive,   // contiguous, only lanes 0..3 active (exec mask)
  kFourDistinctRestSame,  // lanes 0..3 distinct, the other 60 read one slot
  k36Contiguous, // contiguous, lanes 0..35 active
};

template <typename T, int OP>
__global__ __launch_bounds__(512) void k(T* out, long long* cyc, int iters, int pattern)
{
  __shared__ T s[8][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = lane; i < 1024; i += 64)
    s[wave][i] = T(i);
  __syncthreads();
  int idx = lane;
  bool active = true;
  switch (pattern)
  {
  case kContiguous: idx = lane; break;
  case kSameAddress: idx = 0; break;
  case kStride2: idx = 2 * lane; break;
  case kStride4: idx = 4 * lane; break;
  case kStride4Skew: idx = 4 * lane + 2 * (lane >> 3); break;
  case kRandom: idx = (lane * 2654435761u >> 7) & 255; break;
  case kFourActive: idx = lane; active = lane < 4; break;
  case kFourDistinctRestSame: idx = lane < 4 ? lane : 7; break;
  case k36Contiguous: idx = lane; active = lane < 36; break;
  }
  T acc = T(0);
  T* p = &s[wave][idx];
  const long long t0 = clock64();
  if (active)
  {
    for (int i = 0; i < iters; ++i)
    {
      if (OP == 0)
      {
        // four independent reads per round (volatile: the compiler may not
        // hoist them); the offsets keep the pattern
        const unsigned addr = unsigned(reinterpret_cast<size_t>(p));
        if (sizeof(T) == 8)
        {
          unsigned long long a0, a1, a2, a3, a4, a5, a6, a7;
          asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:2048\n\t"
                       "ds_read_b64 %2, %8 offset:4096\n\tds_read_b64 %3, %8 offset:6144\n\t"
                       "ds_read_b64 %4, %8 offset:8\n\tds_read_b64 %5, %8 offset:2056\n\t"
                       "ds_read_b64 %6, %8 offset:4104\n\tds_read_b64 %7, %8 offset:6152\n\t"
                       "s_waitcnt lgkmcnt(0)"
                       : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5),
                         "=&v"(a6), "=&v"(a7)
                       : "v"(addr));
          acc += T(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
        }
        else
        {
          unsigned a0, a1, a2, a3, a4, a5, a6, a7;
          asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:1024\n\t"
                       "ds_read_b32 %2, %8 offset:2048\n\tds_read_b32 %3, %8 offset:3072\n\t"
                       "ds_read_b32 %4, %8 offset:4\n\tds_read_b32 %5, %8 offset:1028\n\t"
                       "ds_read_b32 %6, %8 offset:2052\n\tds_read_b32 %7, %8 offset:3076\n\t"
                       "s_waitcnt lgkmcnt(0)"
                       : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5),
                         "=&v"(a6), "=&v"(a7)
                       : "v"(addr));
          acc += T(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
        }
      }
      else
      {
        atomicOr(&p[0], T(1) << (lane & 31));
        atomicOr(&p[256], T(1) << (lane & 31));
        atomicOr(&p[512], T(1) << (lane & 31));
        atomicOr(&p[768 - 256 * (i & 1)], T(1) << (lane & 31));
      }
    }
  }
  __syncthreads();
  const long long t1 = clock64();
  if (threadIdx.x == 0)
    cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + s[wave][lane];
}

template <typename T, int OP>
void run(const char* name, int pattern)
{
  T* out;
  long long* cyc;
  hipMalloc(&out, sizeof(T) * 512 * 256);
  hipMalloc(&cyc, 8 * 256);
  const int iters = 2000;
  k<T, OP><<<256, 512>>>(out, cyc, iters, pattern);
  hipDeviceSynchronize();
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipEventRecord(a);
  k<T, OP><<<256, 512>>>(out, cyc, iters, pattern);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  // one workgroup per CU: elapsed time / instructions issued per CU
  const double instr = iters * (OP == 0 ? 8.0 : 4.0) * 8;
  printf("%-40s %7.2f clock64 ticks, %6.2f ns per wave instruction per CU (%5.1f clk at 2.4 GHz)\n",
         name, double(c) / instr, 1e6 * ms / instr, 2.4e6 * ms / instr);
  hipFree(out);
  hipFree(cyc);
}

int main()
{
  const char* names[] = {"contiguous", "same address", "stride 2 slots", "stride 4 slots",
                         "stride 4 slots + skew", "random", "4 lanes active",
                         "4 distinct + 60 same", "36 lanes contiguous"};
  for (int p = 0; p < 9; ++p)
  {
    char buf[96];
    snprintf(buf, sizeof buf, "ds_read_b64  %s", names[p]);
    run<unsigned long long, 0>(buf, p);
  }
  for (int p = 0; p < 9; ++p)
  {
    char buf[96];
    snprintf(buf, sizeof buf, "ds_read_b32  %s", names[p]);
    run<unsigned, 0>(buf, p);
  }
  for (int p = 0; p < 9; ++p)
  {
    char buf[96];
    snprintf(buf, sizeof buf, "ds_or_b64    %s", names[p]);
    run<unsigned long long, 1>(buf, p);
  }
  return 0;
}
