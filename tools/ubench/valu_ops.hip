// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the
// VALU operations the feature kernels are made of, at 4 and 8 resident waves
// per SIMD, 8 independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(64) void k(float* out, float a, float b, int iters)
{
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
    x[i] = threadIdx.x * 0.001f + i + 1.0f;
  unsigned long long m = 0x5555555555555555ull;
  for (int it = 0; it < iters; ++it)
  {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
    {
#define DO(i)                                                                          \
  if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "s"(a), "v"(b)); \
  if (OP == 1) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[i]) : "s"(a));             \
  if (OP == 2) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b) : ); \
  if (OP == 3) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(b), "s"(m)); \
  if (OP == 4) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(x[i]), "v"(b) : "vcc"); \
  if (OP == 5) asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(m) : "v"(x[i]), "v"(b));  \
  if (OP == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));                          \
  if (OP == 7) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[i]));                         \
  if (OP == 8) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(x[i]) : "v"(b) : "vcc"); \
  if (OP == 9) asm volatile("v_div_fmas_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b) : ); \
  if (OP == 10) asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b));  \
  if (OP == 11) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x[i]) : "s"(a), "v"(b)); \
  if (OP == 12) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x[i]) : "s"(a));            \
  if (OP == 13) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i])); \
  if (OP == 14) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));            \
  if (OP == 15) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));            \
  if (OP == 16) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x[i]));                     \
  if (OP == 17) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));                         \
  if (OP == 18) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));         \
  if (OP == 19) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x[i]) : "v"(b));    \
  if (OP == 20) asm volatile("v_floor_f32 %0, %0" : "+v"(x[i]));                       \
  if (OP == 21) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i & 1]) : "v"(x[i]));     \
  if (OP == 22) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[i & 1]));             \
  if (OP == 23) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d[i & 1]));                 \
  if (OP == 24) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[i & 1]));                 \
  if (OP == 30) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[i]) : "v"(a));            \
  if (OP == 31) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)); \
  if (OP == 32) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[i]) : "s"(a));            \
  if (OP == 33) { if (i & 1) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[i]) : "s"(a)); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b)); } \
  if (OP == 34) { if (i & 1) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[i]) : "v"(a)); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b)); } \
  if (OP == 35) asm volatile("v_max_f32 %0, %0, %0" : "+v"(x[i]));                     \
  if (OP == 36) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));            \
  if (OP == 37) { if (i & 1) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(b), "s"(m)); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b)); } \
  if (OP == 38) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(d[i & 1]));              \
  if (OP == 39) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(d[i & 1]));              \
  if (OP == 40) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(d[i & 1]));          \
  if (OP == 43) { if (i & 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b)); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b)); } \
  if (OP == 44) { if (i & 1) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b) : "vcc"); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b)); } \
  if (OP == 45) { if (i & 1) asm volatile("v_cmp_gt_f32 %2, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(b), "s"(m)); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b)); } \
  if (OP == 46) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));            \
  if (OP == 47) { if (i & 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b)); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b)); } \
  if (OP == 48) { if (i & 1) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x[i]) : "v"(b)); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b)); } \
  if (OP == 49) { if (i & 1) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x[i]) ); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b)); } \
  if (OP == 50) { if (i & 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]) ); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b)); } \
  if (OP == 51) { if ((i & 3) == 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]) ); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b)); } \
  if (OP == 52) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x[i]) : "v"(b));            \
  if (OP == 53) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));        \
  if (OP == 41) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(x[i]));                    \
  if (OP == 42) asm volatile("v_mul_f32 %0, 0x3f7fbe77, %0" : "+v"(x[i]));
      double d[2] = {1.0, 2.0};
      REP8(DO)
      if (OP >= 21 && OP < 30 || OP >= 38 && OP <= 40)
        x[0] += float(d[0] + d[1]);
    }
  }
  float s = float(m & 1);
#pragma unroll
  for (int i = 0; i < 8; ++i)
    s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name)
{
  for (int w : {2, 4, 8})
  {
    const int blocks = 1024 * w;
    float* out;
    hipMalloc(&out, size_t(blocks) * 64 * 4);
    const int iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < 3; ++r)
    {
      hipEventRecord(a);
      k<OP><<<blocks, 64>>>(out, 0.999f, 0.5f, iters);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      best = ms < best ? ms : best;
    }
    const double instr = double(blocks) * iters * 32;
    printf("%-22s waves/SIMD=%d: %8.3f ms  %6.2f ns/instr/SIMD\n", name, w, best,
           best * 1e6 / (instr / 1024));
    hipFree(out);
  }
}

int main()
{
  run<30>("v_mul_f32 vgpr");
  run<41>("v_mul_f32 inline const");
  run<42>("v_mul_f32 literal");
  run<31>("v_fma_f32 vgpr");
  run<32>("v_add_f32 sgpr");
  run<36>("v_sub_f32 vgpr");
  run<33>("mul(sgpr)/add alt");
  run<34>("mul(vgpr)/add alt");
  run<37>("cndmask(sgpr)/add alt");
  run<35>("v_max_f32 x,x");
  run<38>("v_pk_mul_f32");
  run<39>("v_pk_add_f32");
  run<40>("v_pk_fma_f32");
  run<43>("cndmask(vcc)/add alt");
  run<44>("cmp+nop+cndmask vcc/add");
  run<45>("cmp+nop+cndmask sgpr/add");
  run<46>("v_max_f32 vgpr");
  run<47>("max/add alt");
  run<52>("v_and_b32 vgpr");
  run<48>("and/add alt");
  run<49>("cvt_i32/add alt");
  run<50>("rcp/add alt");
  run<51>("rcp/3 add");
  run<53>("v_fma_f32 x,b,x");
  run<0>("v_fma_f32");
  run<1>("v_mul_f32");
  run<14>("v_add_f32");
  run<15>("v_max_f32");
  run<2>("v_cndmask vcc");
  run<3>("v_cndmask sgpr");
  run<4>("v_cmp -> vcc");
  run<5>("v_cmp -> sgpr");
  run<6>("v_rcp_f32");
  run<7>("v_sqrt_f32");
  run<17>("v_exp_f32");
  run<8>("v_div_scale_f32");
  run<9>("v_div_fmas_f32");
  run<10>("v_div_fixup_f32");
  run<11>("v_bfi_b32");
  run<12>("v_and_b32");
  run<13>("v_mov_b32_dpp");
  run<16>("v_cvt_i32_f32");
  run<20>("v_floor_f32");
  run<18>("v_mul_lo_u32");
  run<19>("v_lshl_add_u32");
  run<21>("v_cvt_f64_f32");
  run<22>("v_fma_f64");
  run<23>("v_mul_f64");
  run<24>("v_add_f64");
  return 0;
}
