// Exhaustive check of the short correctly-rounded sqrt / division sequences
// against the compiler's IEEE ones, over every non-negative float:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o fast_math_check fast_math_check.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

__device__ inline float sqrt_fast(float x)
{
  const float y = __builtin_amdgcn_rsqf(x);
  const float g = x * y;
  const float h = 0.5f * y;
  const float d = __builtin_fmaf(-g, g, x);
  const float r = __builtin_fmaf(d, h, g);
  return fmaxf(r, 0.f);
}

__device__ inline float sqrt_fast2(float x)
{
  // v_sqrt seed, reciprocal of the seed for the half-inverse
  const float g = __builtin_amdgcn_sqrtf(x);
  const float h = 0.5f * __builtin_amdgcn_rcpf(g);
  const float d = __builtin_fmaf(-g, g, x);
  const float r = __builtin_fmaf(d, h, g);
  return fmaxf(r, 0.f);
}

__device__ inline float sqrt_fast3(float x)
{
  // rsq seed with one coupled refinement of g and h (Goldschmidt), then Markstein
  const float y = __builtin_amdgcn_rsqf(x);
  float g = x * y;
  float h = 0.5f * y;
  const float e = __builtin_fmaf(-h, g, 0.5f);
  g = __builtin_fmaf(g, e, g);
  h = __builtin_fmaf(h, e, h);
  const float d = __builtin_fmaf(-g, g, x);
  const float r = __builtin_fmaf(d, h, g);
  return fmaxf(r, 0.f);
}

__device__ inline float div_fast(float n, float d)
{
  float r = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  float q = n * r;
  float rem = __builtin_fmaf(-d, q, n);
  q = __builtin_fmaf(rem, r, q);
  rem = __builtin_fmaf(-d, q, n);
  return __builtin_fmaf(rem, r, q);
}

__device__ inline float div_fast1(float n, float d)
{
  float r = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  float q = n * r;
  float rem = __builtin_fmaf(-d, q, n);
  return __builtin_fmaf(rem, r, q);
}

__global__ void check_sqrt(unsigned long long* bad, unsigned* first, int variant, unsigned* hist)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  for (uint32_t hi = 0; hi < 128; ++hi)
  {
    const uint32_t bits = hi * (1u << 24) + i;
    if (bits > 0x7f800000u)
      continue;
    float x;
    memcpy(&x, &bits, 4);
    const float a = sqrtf(x), b = variant == 0 ? sqrt_fast(x) : (variant == 1 ? sqrt_fast2(x) : sqrt_fast3(x));
    uint32_t ua, ub;
    memcpy(&ua, &a, 4);
    memcpy(&ub, &b, 4);
    if (ua != ub)
    {
      atomicAdd(bad, 1ull);
      atomicMin(first, bits);
      atomicMax(first + 1, bits);
      atomicAdd(hist + (bits >> 23), 1u);
    }
  }
}

// the quotients of the atanf argument reduction: (a x + b) / (c x + d)
__global__ void check_div(unsigned long long* bad, unsigned* first, int variant)
{
  const float A[5] = {1.f, 2.f, 1.f, 1.f, 0.f};
  const float B[5] = {0.f, -1.f, -1.f, -1.5f, -1.f};
  const float C[5] = {0.f, 1.f, 1.f, 1.5f, 1.f};
  const float D[5] = {1.f, 2.f, 1.f, 1.f, 0.f};
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  for (uint32_t hi = 0; hi < 128; ++hi)
  {
    const uint32_t bits = hi * (1u << 24) + i;
    if (bits >= 0x7f800000u)
      continue;
    float x;
    memcpy(&x, &bits, 4);
    const int id = int(bits >= 0x3ee00000u) + int(bits >= 0x3f300000u) +
                   int(bits >= 0x3f980000u) + int(bits >= 0x401c0000u);
    const float num = A[id] * x + B[id];
    const float den = C[id] * x + D[id];
    const float a = num / den;
    const float b = variant ? div_fast1(num, den) : div_fast(num, den);
    uint32_t ua, ub;
    memcpy(&ua, &a, 4);
    memcpy(&ub, &b, 4);
    if (ua != ub)
    {
      atomicAdd(bad + 1 + id, 1ull);
      atomicAdd(bad, 1ull);
      atomicMin(first, bits);
      atomicMax(first + 1, bits);
    }
  }
}

int main()
{
  unsigned long long* bad;
  unsigned* first;
  hipMalloc(&bad, 8 * 8);
  hipMalloc(&first, 8);
  auto run = [&](const char* name, auto launch) {
    unsigned long long hb[8] = {};
    unsigned hf[2] = {0xffffffffu, 0};
    hipMemcpy(bad, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipMemcpy(first, hf, sizeof(hf), hipMemcpyHostToDevice);
    launch();
    hipDeviceSynchronize();
    hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost);
    hipMemcpy(hf, first, sizeof(hf), hipMemcpyDeviceToHost);
    printf("%s: mismatches %llu (per id %llu %llu %llu %llu %llu) first 0x%08x last 0x%08x\n",
           name, hb[0], hb[1], hb[2], hb[3], hb[4], hb[5], hf[0], hf[1]);
  };
  unsigned* hist;
  hipMalloc(&hist, 256 * 4);
  for (int v = 0; v < 3; ++v)
  {
    hipMemset(hist, 0, 256 * 4);
    run(v == 0 ? "sqrt_fast (rsq, 1 step)" : v == 1 ? "sqrt_fast2 (sqrt+rcp, 1 step)" : "sqrt_fast3 (rsq, 2 steps)",
        [&] { hipLaunchKernelGGL(check_sqrt, dim3((1u << 24) / 256), dim3(256), 0, 0, bad, first, v, hist); });
    unsigned hh[256];
    hipMemcpy(hh, hist, sizeof(hh), hipMemcpyDeviceToHost);
    printf("  mismatches by exponent:");
    for (int e = 0; e < 256; ++e)
      if (hh[e])
        printf(" %d:%u", e, hh[e]);
    printf("\n");
  }
  run("div_fast (2 refinements)", [&] { hipLaunchKernelGGL(check_div, dim3((1u << 24) / 256), dim3(256), 0, 0, bad, first, 0); });
  run("div_fast1 (1 refinement)", [&] { hipLaunchKernelGGL(check_div, dim3((1u << 24) / 256), dim3(256), 0, 0, bad, first, 1); });
  return 0;
}
