// Micro-benchmark: unfused f32 mul->add issue rate on gfx950 as a function of
// (a) resident waves per SIMD and (b) the distance, in instructions, between a
// v_mul_f32 and the v_add_f32 that consumes it.  Guides the hand-scheduled
// column pass of the marching Gaussian blur.
#include <hip/hip_runtime.h>
#include <cstdio>

// DIST = 1: mul a; add a; mul b; add b ...     (consumer right behind producer)
// DIST = 2: mul a; mul b; add a; add b ...
// DIST = 4: mul a..d; add a..d
// DIST = 8: mul a..h; add a..h
template <int DIST>
__global__ __launch_bounds__(64) void k(float* out, float a, int iters)
{
  float x[8], p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
    x[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it)
  {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
    {
      if (DIST == 1)
      {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          asm volatile("v_mul_f32 %0, %2, %1\n\tv_add_f32 %1, %1, %0"
                       : "=&v"(p[i]), "+v"(x[i]) : "s"(a));
      }
      else
      {
#pragma unroll
        for (int g = 0; g < 8; g += DIST)
        {
#pragma unroll
          for (int i = g; i < g + DIST; ++i)
            asm volatile("v_mul_f32 %0, %2, %1" : "=&v"(p[i]) : "v"(x[i]), "s"(a));
#pragma unroll
          for (int i = g; i < g + DIST; ++i)
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(p[i]));
        }
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int DIST>
void run(int waves_per_simd)
{
  const int blocks = 1024 * waves_per_simd;
  float* out;
  hipMalloc(&out, size_t(blocks) * 64 * 4);
  const int iters = 4000;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e9f;
  for (int r = 0; r < 3; ++r)
  {
    hipEventRecord(a);
    k<DIST><<<blocks, 64>>>(out, 0.999f, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    best = ms < best ? ms : best;
  }
  const double instr = double(blocks) * iters * 64;  // wave-instructions
  printf("dist=%d waves/SIMD=%d: %.3f ms  %.3f T wave-instr/s  (%.2f cyc/instr/SIMD @2.4GHz)\n",
         DIST, waves_per_simd, best, instr / best / 1e9,
         2.4e9 * 1024 / (instr / (best * 1e-3)));
  hipFree(out);
}

int main()
{
  for (int w : {1, 2, 3, 4, 5, 6, 8})
  {
    run<1>(w);
    run<2>(w);
    run<4>(w);
    run<8>(w);
  }
  return 0;
}
