// Calibration of the TCC FETCH_SIZE counter on gfx950: kernels that read a
// 1 GiB buffer exactly once with 4-, 8- and 16-byte loads per lane, in the
// two access shapes of the pipeline (flat grid-stride; one wave marching down
// a strip of rows).  Run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d D -o c -- ./fetch_calib
// and compare the counter (KiB) with 1 048 576 KiB per launch.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename T>
__global__ __launch_bounds__(256) void flat_read(const T* __restrict__ a, size_t n, float* out)
{
  float acc = 0.f;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += size_t(gridDim.x) * blockDim.x)
  {
    const T v = a[i];
    const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
    for (int q = 0; q < int(sizeof(T) / 4); ++q)
      acc += f[q];
  }
  if (acc == 12345.678f)
    out[0] = acc;
}

// one wave per strip of 64 * sizeof(T) / 4 columns, rows in sequence
template <typename T>
__global__ __launch_bounds__(64) void march_read(const float* __restrict__ a, int w, int h,
                                                 int seg_rows, int nstrips, float* out)
{
  constexpr int C = sizeof(T) / 4;
  const int strip = blockIdx.x % nstrips, seg = blockIdx.x / nstrips;
  const float* p = a + size_t(blockIdx.y) * w * h;
  const int col = strip * 64 * C + C * threadIdx.x;
  const int y0 = seg * seg_rows, y1 = min(h, y0 + seg_rows);
  float acc = 0.f;
  if (col < w)
    for (int y = y0; y < y1; ++y)
    {
      const T v = *reinterpret_cast<const T*>(p + size_t(y) * w + col);
      const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
      for (int q = 0; q < C; ++q)
        acc += f[q];
    }
  if (acc == 12345.678f)
    out[0] = acc;
}

int main()
{
  const size_t bytes = size_t(1) << 30;
  float *buf, *out;
  hipMalloc(&buf, bytes);
  hipMalloc(&out, 4);
  hipMemset(buf, 0, bytes);
  hipDeviceSynchronize();
  // flat: 1 GiB
  hipLaunchKernelGGL(flat_read<float>, dim3(8192), dim3(256), 0, 0, buf, bytes / 4, out);
  hipLaunchKernelGGL(flat_read<float2>, dim3(8192), dim3(256), 0, 0,
                     reinterpret_cast<const float2*>(buf), bytes / 8, out);
  hipLaunchKernelGGL(flat_read<float4>, dim3(8192), dim3(256), 0, 0,
                     reinterpret_cast<const float4*>(buf), bytes / 16, out);
  // marching: 128 planes of 2048 x 1024 floats = 1 GiB, 4 segments of 256 rows
  const int w = 2048, h = 1024, planes = 128, seg = 256;
  hipLaunchKernelGGL(march_read<float>, dim3((w / 64) * (h / seg), planes), dim3(64), 0, 0, buf, w,
                     h, seg, w / 64, out);
  hipLaunchKernelGGL(march_read<float2>, dim3((w / 128) * (h / seg), planes), dim3(64), 0, 0, buf,
                     w, h, seg, w / 128, out);
  hipLaunchKernelGGL(march_read<float4>, dim3((w / 256) * (h / seg), planes), dim3(64), 0, 0, buf,
                     w, h, seg, w / 256, out);
  hipDeviceSynchronize();
  printf("done\n");
  return 0;
}
