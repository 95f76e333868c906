// Micro-benchmark: strip-marching READ of NP planes (the extremum scan's
// pattern) with 8-byte vs 16-byte loads per lane; writes almost nothing.
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename V, int NP, int PF>
__global__ __launch_bounds__(64) void k(const float* __restrict__ src, float* __restrict__ out, int w, int h,
                                      int seg_rows, int nstrips, size_t plane, size_t frame_stride)
{
  constexpr int CPL = sizeof(V) / 4;
  const int lane = threadIdx.x;
  const int strip = blockIdx.x % nstrips, seg = blockIdx.x / nstrips;
  const float* g = src + blockIdx.y * frame_stride;
  const int col = min(strip * 64 * CPL + CPL * lane, w - CPL);
  const int y0 = seg * seg_rows, y1 = min(h, y0 + seg_rows);
  V pg[PF][NP];
  float acc = 0.f;
#pragma unroll
  for (int q = 0; q < PF; ++q)
#pragma unroll
    for (int l = 0; l < NP; ++l)
      pg[q][l] = *reinterpret_cast<const V*>(g + l * plane + size_t(min(y0 + q, h - 1)) * w + col);
  for (int y = y0; y < y1; y += PF)
  {
#pragma unroll
    for (int i = 0; i < PF; ++i)
    {
#pragma unroll
      for (int l = 0; l < NP; ++l)
      {
        acc += pg[i][l].x;
        pg[i][l] = *reinterpret_cast<const V*>(g + l * plane + size_t(min(y + i + PF, h - 1)) * w + col);
      }
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}
template <typename V, int NP, int PF> void run(const char* name, int nseg)
{
  const int w = 1920, h = 1080, B = 64;
  const size_t plane = size_t(w) * h, fs = plane * NP;
  float *a, *o; hipMalloc(&a, fs * B * 4); hipMalloc(&o, 64); hipMemset(a, 0, fs * B * 4);
  constexpr int CPL = sizeof(V) / 4;
  const int nstrips = (w + 64 * CPL - 1) / (64 * CPL);
  const int seg_rows = (h + nseg - 1) / nseg;
  auto f = [&] { k<V, NP, PF><<<dim3(nstrips * nseg, B), 64>>>(a, o, w, h, seg_rows, nstrips, plane, fs); };
  f(); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); for (int i = 0; i < 5; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%-34s nseg=%2d waves=%6d: %.3f ms  %.2f TB/s\n", name, nseg, nstrips * nseg * B, ms, fs * B * 4 / 1e9 / ms);
  hipFree(a); hipFree(o);
}
int main()
{
  for (int nseg : {2, 4, 8}) {
    run<float2, 6, 3>("6 planes float2 PF3", nseg);
    run<float4, 6, 2>("6 planes float4 PF2", nseg);
    run<float4, 6, 3>("6 planes float4 PF3", nseg);
    run<float4, 1, 4>("1 plane  float4 PF4", nseg * 4);
  }
}
