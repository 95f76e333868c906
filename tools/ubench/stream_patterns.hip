// Micro-benchmark: achievable HBM bandwidth of (a) a flat float4 copy and
// (b) the strip-marching access pattern of the pyramid kernels (one wave per
// 256-column strip, rows in sequence, PF rows of loads in flight), no compute.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void flat_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n)
{
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
    b[i] = a[i];
}
template <int PF, int WPB>
__global__ __launch_bounds__(64 * WPB) void march_copy(const float* __restrict__ src, float* __restrict__ dst,
                                                  int w, int h, int seg_rows, int nstrips)
{
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * WPB + (threadIdx.x >> 6);
  const int strip = wid % nstrips, seg = wid / nstrips;
  const size_t b = blockIdx.y;
  src += b * size_t(w) * h; dst += b * size_t(w) * h;
  const int col = strip * 256 + 4 * lane;
  if (col >= w) return;
  const int y0 = seg * seg_rows, y1 = min(h, y0 + seg_rows);
  if (y0 >= h) return;
  float4 pm[PF];
#pragma unroll
  for (int q = 0; q < PF; ++q) pm[q] = *reinterpret_cast<const float4*>(src + size_t(min(y0 + q, h - 1)) * w + col);
  for (int y = y0; y < y1; y += PF)
  {
#pragma unroll
    for (int i = 0; i < PF; ++i)
    {
      const int yy = y + i;
      const float4 v = pm[i];
      pm[i] = *reinterpret_cast<const float4*>(src + size_t(min(yy + PF, h - 1)) * w + col);
      if (yy < y1) *reinterpret_cast<float4*>(dst + size_t(yy) * w + col) = v;
    }
  }
}
// the wide blurs' pattern: 2 columns per lane (strips of 128), 8-byte loads and stores
template <int PF>
__global__ __launch_bounds__(64) void march_copy2(const float* __restrict__ src, float* __restrict__ dst,
                                                  int w, int h, int seg_rows, int nstrips)
{
  const int lane = threadIdx.x;
  const int wid = blockIdx.x;
  const int strip = wid % nstrips, seg = wid / nstrips;
  const size_t b = blockIdx.y;
  src += b * size_t(w) * h; dst += b * size_t(w) * h;
  const int col = strip * 128 + 2 * lane;
  if (col >= w) return;
  const int y0 = seg * seg_rows, y1 = min(h, y0 + seg_rows);
  if (y0 >= h) return;
  float2 pm[PF];
#pragma unroll
  for (int q = 0; q < PF; ++q) pm[q] = *reinterpret_cast<const float2*>(src + size_t(min(y0 + q, h - 1)) * w + col);
  for (int y = y0; y < y1; y += PF)
  {
#pragma unroll
    for (int i = 0; i < PF; ++i)
    {
      const int yy = y + i;
      const float2 v = pm[i];
      pm[i] = *reinterpret_cast<const float2*>(src + size_t(min(yy + PF, h - 1)) * w + col);
      if (yy < y1) *reinterpret_cast<float2*>(dst + size_t(yy) * w + col) = v;
    }
  }
}
template <typename F> float timeit(F f, int reps = 5)
{
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main()
{
  const int w = 1920, h = 1080, B = 64;
  const size_t n = size_t(w) * h * B;
  float *a, *b; hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
  hipMemset(a, 1, n * 4); hipMemset(b, 0, n * 4);
  const double gb = 2.0 * n * 4 / 1e9;
  for (int blocks : {2048, 8192, 65536}) {
    float ms = timeit([&] { flat_copy<<<blocks, 256>>>((float4*) a, (float4*) b, n / 4); });
    printf("flat float4 copy  blocks=%6d: %.3f ms  %.2f TB/s\n", blocks, ms, gb / ms);
  }
  const int nstrips = (w + 255) / 256;
  for (int nseg : {4, 6, 8, 12, 18, 36}) {
    const int seg_rows = (h + nseg - 1) / nseg;
    {
      float ms = timeit([&] { march_copy<4, 1><<<dim3(nstrips * nseg, B), 64>>>(a, b, w, h, seg_rows, nstrips); });
      printf("march PF=4 WPB=1 nseg=%2d (waves %5d): %.3f ms  %.2f TB/s\n", nseg, nstrips * nseg * B, ms, gb / ms);
    }
    {
      float ms = timeit([&] { march_copy<8, 1><<<dim3(nstrips * nseg, B), 64>>>(a, b, w, h, seg_rows, nstrips); });
      printf("march PF=8 WPB=1 nseg=%2d (waves %5d): %.3f ms  %.2f TB/s\n", nseg, nstrips * nseg * B, ms, gb / ms);
    }
    {
      float ms = timeit([&] { march_copy<4, 4><<<dim3((nstrips * nseg + 3) / 4, B), 256>>>(a, b, w, h, seg_rows, nstrips); });
      printf("march PF=4 WPB=4 nseg=%2d (waves %5d): %.3f ms  %.2f TB/s\n", nseg, nstrips * nseg * B, ms, gb / ms);
    }
  }
  for (int nseg : {2, 3, 4, 6}) {
    const int seg_rows = (h + nseg - 1) / nseg;
    const int ns128 = (w + 127) / 128;
    float ms = timeit([&] { march_copy2<4><<<dim3(ns128 * nseg, B), 64>>>(a, b, w, h, seg_rows, ns128); });
    printf("march2 (128-col strips, float2) PF=4 nseg=%2d (waves %5d): %.3f ms  %.2f TB/s\n", nseg, ns128 * nseg * B, ms, gb / ms);
    ms = timeit([&] { march_copy2<8><<<dim3(ns128 * nseg, B), 64>>>(a, b, w, h, seg_rows, ns128); });
    printf("march2 (128-col strips, float2) PF=8 nseg=%2d (waves %5d): %.3f ms  %.2f TB/s\n", nseg, ns128 * nseg * B, ms, gb / ms);
  }
  return 0;
}
