// Micro-benchmark: issue cost per SIMD of the orientation replay's dependent
// chain hist = float(double(hist) + c) in two forms, 8 waves per SIMD:
//   A  v_cvt_f64_f32, v_add_f64, v_cvt_f32_f64
//   B  v_add_f64, v_add_f64 (+ C), v_add_f64 (- C)   (rounding through a magic
//      constant C = 1.5 * 2^(e+29) of the sum's binade)
// and of the single instructions.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(64) void k(double* out, double c, double C, int iters)
{
  float h = 1.0f + threadIdx.x * 1e-3f;
  double hd = h, t = 0.;
  for (int it = 0; it < iters; ++it)
  {
#pragma unroll
    for (int r = 0; r < 16; ++r)
    {
      if (OP == 0)
        asm volatile("v_cvt_f64_f32 %1, %0\n\tv_add_f64 %1, %1, %2\n\tv_cvt_f32_f64 %0, %1"
                     : "+v"(h), "+v"(t) : "v"(c));
      if (OP == 1)
        asm volatile("v_add_f64 %0, %0, %1\n\tv_add_f64 %0, %0, %2\n\tv_add_f64 %0, %0, -%2"
                     : "+v"(hd) : "v"(c), "v"(C));
      if (OP == 2)
        asm volatile("v_cvt_f64_f32 %1, %0" : "+v"(h), "+v"(t));
      if (OP == 3)
        asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(h), "+v"(t));
      if (OP == 4)
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(hd) : "v"(c));
      if (OP == 5)
        asm volatile("v_cmp_ge_f64 vcc, %0, %1" : : "v"(hd), "v"(c) : "vcc");
      if (OP == 6)
        asm volatile("v_add_f32 %0, %0, %0" : "+v"(h));
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = hd + h + t;
}

template <int OP>
void run(const char* name, int per_round)
{
  const int blocks = 1024 * 8;
  double* out;
  hipMalloc(&out, size_t(blocks) * 64 * 8);
  const int iters = 1000;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e9f;
  for (int r = 0; r < 3; ++r)
  {
    hipEventRecord(a);
    k<OP><<<blocks, 64>>>(out, 1e-3, 1.5 * 9007199254740992.0 / 8388608.0 * 2.0, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    best = ms < best ? ms : best;
  }
  const double instr = double(blocks) * iters * 16 * per_round;
  printf("%-34s %8.3f ms  %6.2f ns per instruction per SIMD (%.1f clk at 2.4 GHz)\n", name, best,
         best * 1e6 / (instr / 1024), 2.4 * best * 1e6 / (instr / 1024));
  hipFree(out);
}

int main()
{
  run<0>("cvt_f64_f32 + add_f64 + cvt_f32_f64", 3);
  run<1>("add_f64 x3 (magic rounding)", 3);
  run<2>("v_cvt_f64_f32", 1);
  run<3>("v_cvt_f32_f64", 1);
  run<4>("v_add_f64", 1);
  run<5>("v_cmp_ge_f64", 1);
  run<6>("v_add_f32", 1);
  return 0;
}
