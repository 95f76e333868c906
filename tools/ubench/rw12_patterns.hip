// Micro-benchmark: achievable HBM bandwidth of a 1 read : 2 write stream (the
// polar-gradient kernel's traffic: 4 B in, 8 B out per pixel) under different
// access patterns, no compute.  192 planes of 1920 x 1080 (octave 0, 64 frames).
#include <hip/hip_runtime.h>
#include <cstdio>
// MODE 0: flat, each lane 16 B in -> 32 B out (two float4 at stride 32 B)
// MODE 1: flat, contiguous stores (out[lane], out[64 + lane] per wave)
template <int MODE>
__global__ __launch_bounds__(256) void flat(const float4* __restrict__ a, float4* __restrict__ b, size_t n)
{
  const int lane = threadIdx.x & 63;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
  {
    const float4 v = a[i];
    if (MODE == 0) { b[2 * i] = v; b[2 * i + 1] = v; }
    else { const size_t w0 = (i - lane) * 2; b[w0 + lane] = v; b[w0 + 64 + lane] = v; }
  }
}
// marching: one wave per (strip of W columns, segment of rows)
// MODE 0: float4 load, two float4 stores per lane at stride 32 B (current kernel)
// MODE 1: float4 load, contiguous stores
// MODE 2: two float2 loads (cols 2l, 128 + 2l), two contiguous float4 stores
// MODE 3: W = 512: two float4 loads, four contiguous float4 stores
template <int MODE, int PF>
__global__ __launch_bounds__(64) void march(const float* __restrict__ src, float* __restrict__ dst,
                                            int w, int h, int seg_rows, int nstrips, int nseg)
{
  constexpr int W = MODE == 3 ? 512 : 256;
  const int lane = threadIdx.x;
  const int item = blockIdx.x;
  const int strip = item % nstrips, seg = (item / nstrips) % nseg;
  const size_t plane = item / (nstrips * nseg);
  src += plane * size_t(w) * h; dst += plane * size_t(w) * h * 2;
  const int x0 = strip * W;
  const int y0 = seg * seg_rows, y1 = min(h, y0 + seg_rows);
  if (y0 >= h) return;
  constexpr int NV = MODE == 3 ? 2 : 1;
  float4 pm[PF][NV];
  auto load = [&](int y, float4 (&r)[NV]) {
    const float* row = src + size_t(min(y, h - 1)) * w;
    if (MODE == 2) {
      const int c0 = min(x0 + 2 * lane, w - 2), c1 = min(x0 + 128 + 2 * lane, w - 2);
      const float2 p = *reinterpret_cast<const float2*>(row + c0);
      const float2 q = *reinterpret_cast<const float2*>(row + c1);
      r[0] = make_float4(p.x, p.y, q.x, q.y);
    } else if (MODE == 3) {
      r[0] = *reinterpret_cast<const float4*>(row + min(x0 + 4 * lane, w - 4));
      r[1] = *reinterpret_cast<const float4*>(row + min(x0 + 256 + 4 * lane, w - 4));
    } else r[0] = *reinterpret_cast<const float4*>(row + min(x0 + 4 * lane, w - 4));
  };
#pragma unroll
  for (int q = 0; q < PF; ++q) load(y0 + q, pm[q]);
  for (int y = y0; y < y1; y += PF)
  {
#pragma unroll
    for (int i = 0; i < PF; ++i)
    {
      const int yy = y + i;
      float4 v[NV];
#pragma unroll
      for (int k = 0; k < NV; ++k) v[k] = pm[i][k];
      load(yy + PF, pm[i]);
      if (yy >= y1) continue;
      float* orow = dst + size_t(yy) * w * 2;
      if (MODE == 0) {
        const int col = x0 + 4 * lane;
        if (col < w) { float4* op = reinterpret_cast<float4*>(orow + 2 * col); op[0] = v[0]; op[1] = v[0]; }
      } else if (MODE == 1) {
        float4* ob = reinterpret_cast<float4*>(orow + 2 * x0);
        if (x0 + 2 * lane < w) ob[lane] = v[0];
        if (x0 + 128 + 2 * lane < w) ob[64 + lane] = v[0];
      } else if (MODE == 2) {
        float4* ob = reinterpret_cast<float4*>(orow + 2 * x0);
        if (x0 + 2 * lane < w) ob[lane] = v[0];
        if (x0 + 128 + 2 * lane < w) ob[64 + lane] = v[0];
      } else {
        float4* ob = reinterpret_cast<float4*>(orow + 2 * x0);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (x0 + 128 * k + 2 * lane < w) ob[64 * k + lane] = v[k & 1];
      }
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
  const int w = 1920, h = 1080, P = 192;
  const size_t n = size_t(w) * h * P;
  float *a, *b; hipMalloc(&a, n * 4); hipMalloc(&b, n * 8);
  hipMemset(a, 1, n * 4); hipMemset(b, 0, n * 8);
  const double gb = 3.0 * n * 4 / 1e9;
  for (int blocks : {4096, 16384}) {
    float ms = timeit([&] { flat<0><<<blocks, 256>>>((float4*) a, (float4*) b, n / 4); });
    printf("flat pair-stores   blocks=%6d: %.3f ms  %.2f TB/s\n", blocks, ms, gb / ms);
    ms = timeit([&] { flat<1><<<blocks, 256>>>((float4*) a, (float4*) b, n / 4); });
    printf("flat contig-stores blocks=%6d: %.3f ms  %.2f TB/s\n", blocks, ms, gb / ms);
  }
  for (int nseg : {6, 12, 24}) {
    const int seg_rows = (h + nseg - 1) / nseg;
    const int ns256 = (w + 255) / 256, ns512 = (w + 511) / 512;
#define RUN(MODE, PF, NS, name) { float ms = timeit([&] { march<MODE, PF><<<dim3(NS * nseg * P), 64>>>(a, b, w, h, seg_rows, NS, nseg); }); \
    printf("%-34s nseg=%2d waves=%6d: %.3f ms  %.2f TB/s\n", name, nseg, NS * nseg * P, ms, gb / ms); }
    RUN(0, 4, ns256, "march f4 load, pair stores PF4");
    RUN(0, 8, ns256, "march f4 load, pair stores PF8");
    RUN(1, 4, ns256, "march f4 load, contig stores PF4");
    RUN(2, 4, ns256, "march 2xf2 load, contig stores PF4");
    RUN(2, 8, ns256, "march 2xf2 load, contig stores PF8");
    RUN(3, 4, ns512, "march W512 PF4");
  }
  return 0;
}
