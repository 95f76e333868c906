// Does gfx950 take 8-/16-byte global loads and stores (and 4-byte loads of
// bytes) at addresses that are only element-aligned?  The marching kernels use
// float4 / float2 / uchar4 accesses on rows of any width (odd widths: rows
// start at any multiple of 4 bytes).  Prints PASS / FAIL and the copy rate of
// aligned against misaligned float4 streams.
//   hipcc --offload-arch=gfx950 -O3 -o unaligned_check unaligned_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void copy4(const float* __restrict__ s, float* __restrict__ d, size_t n4)
{
  const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n4)
    *reinterpret_cast<float4*>(d + 4 * i) = *reinterpret_cast<const float4*>(s + 4 * i);
}
__global__ void copy2(const float* __restrict__ s, float* __restrict__ d, size_t n2)
{
  const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n2)
    *reinterpret_cast<float2*>(d + 2 * i) = *reinterpret_cast<const float2*>(s + 2 * i);
}
__global__ void bytes4(const unsigned char* __restrict__ s, float* __restrict__ d, size_t n4)
{
  const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n4)
  {
    const uchar4 q = *reinterpret_cast<const uchar4*>(s + 4 * i);
    *reinterpret_cast<float4*>(d + 4 * i) = make_float4(q.x, q.y, q.z, q.w);
  }
}

int main()
{
  const size_t n = size_t(64) << 20;  // floats
  float *s, *d;
  hipMalloc(&s, (n + 16) * 4);
  hipMalloc(&d, (n + 16) * 4);
  std::vector<float> h(n + 16);
  for (size_t i = 0; i < n + 16; ++i)
    h[i] = float(i % 1000003);
  hipMemcpy(s, h.data(), (n + 16) * 4, hipMemcpyHostToDevice);
  std::vector<float> back(n + 16);
  bool all = true;
  for (int so = 0; so < 4; ++so)
    for (int dof = 0; dof < 4; ++dof)
    {
      hipMemset(d, 0, (n + 16) * 4);
      const size_t n4 = n / 4;
      hipEvent_t a, b;
      hipEventCreate(&a);
      hipEventCreate(&b);
      copy4<<<dim3((n4 + 255) / 256), dim3(256)>>>(s + so, d + dof, n4);
      hipEventRecord(a);
      for (int r = 0; r < 5; ++r)
        copy4<<<dim3((n4 + 255) / 256), dim3(256)>>>(s + so, d + dof, n4);
      hipEventRecord(b);
      hipError_t e = hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      hipMemcpy(back.data(), d, (n + 16) * 4, hipMemcpyDeviceToHost);
      bool ok = e == hipSuccess;
      for (size_t i = 0; i < n && ok; ++i)
        ok = back[i + dof] == h[i + so];
      printf("float4 src+%d dst+%d: %s  %.0f GB/s\n", so, dof, ok ? "PASS" : "FAIL",
             5 * 8.0 * n / ms * 1e-6);
      all &= ok;
    }
  for (int so = 0; so < 2; ++so)
  {
    hipMemset(d, 0, (n + 16) * 4);
    copy2<<<dim3((n / 2 + 255) / 256), dim3(256)>>>(s + so, d + 1 - so, n / 2);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(back.data(), d, (n + 16) * 4, hipMemcpyDeviceToHost);
    bool ok = e == hipSuccess;
    for (size_t i = 0; i < n && ok; ++i)
      ok = back[i + 1 - so] == h[i + so];
    printf("float2 src+%d dst+%d: %s\n", so, 1 - so, ok ? "PASS" : "FAIL");
    all &= ok;
  }
  {
    std::vector<unsigned char> hb(n / 4 + 16);
    for (size_t i = 0; i < hb.size(); ++i)
      hb[i] = (unsigned char) (i * 7u);
    unsigned char* sb;
    hipMalloc(&sb, hb.size());
    hipMemcpy(sb, hb.data(), hb.size(), hipMemcpyHostToDevice);
    for (int so = 0; so < 4; ++so)
    {
      const size_t n4 = n / 16;
      bytes4<<<dim3((n4 + 255) / 256), dim3(256)>>>(sb + so, d + 1, n4);
      hipError_t e = hipDeviceSynchronize();
      hipMemcpy(back.data(), d, (n + 16) * 4, hipMemcpyDeviceToHost);
      bool ok = e == hipSuccess;
      for (size_t i = 0; i < 4 * n4 && ok; ++i)
        ok = back[i + 1] == float(hb[i + so]);
      printf("uchar4 src+%d: %s\n", so, ok ? "PASS" : "FAIL");
      all &= ok;
    }
  }
  printf(all ? "ALL PASS\n" : "SOME FAIL\n");
  return all ? 0 : 1;
}
