// Micro-benchmark: issue rate of unfused f32 mul+add, scalar vs packed
// (v_pk_mul_f32 / v_pk_add_f32), and fma, on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters)
{
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
  f2 pa = {a, a}, pb = {b, b};
  for (int i = 0; i < iters; ++i)
  {
    if (MODE == 0) {  // scalar mul + add (8 independent chains, 16 instr)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(x0) : "v"(a), "v"(b));
        asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(x1) : "v"(a), "v"(b));
        asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(x2) : "v"(a), "v"(b));
        asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(x3) : "v"(a), "v"(b));
        asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(x4) : "v"(a), "v"(b));
        asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(x5) : "v"(a), "v"(b));
        asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(x6) : "v"(a), "v"(b));
        asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(x7) : "v"(a), "v"(b));
      }
    } else if (MODE == 1) {  // packed mul + add: same 8 values as 4 pairs
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %2" : "+v"(p0) : "v"(pa), "v"(pb));
        asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %2" : "+v"(p1) : "v"(pa), "v"(pb));
        asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %2" : "+v"(p2) : "v"(pa), "v"(pb));
        asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %2" : "+v"(p3) : "v"(pa), "v"(pb));
      }
    } else {  // scalar fma
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x4) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x5) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x6) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x7) : "v"(a), "v"(b));
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE> void run(const char* name, int blocks)
{
  float* out; hipMalloc(&out, 4 * 256 * blocks);
  const int iters = 4000;
  k<MODE><<<blocks, 256>>>(out, 0.999f, 0.001f, iters);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a); k<MODE><<<blocks, 256>>>(out, 0.999f, 0.001f, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double muladds = double(blocks) * 256 * iters * 32;  // 32 mul-add pairs per thread per iter
  printf("%-22s blocks=%5d: %.3f ms  %.2f T mul-add/s\n", name, blocks, ms, muladds / ms / 1e9);
  hipFree(out);
}
int main()
{
  for (int blocks : {1024, 4096}) {
    run<0>("v_mul + v_add", blocks);
    run<1>("v_pk_mul + v_pk_add", blocks);
    run<2>("v_fma", blocks);
  }
}
