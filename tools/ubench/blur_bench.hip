// Micro-benchmark of the pyramid's blur kernels: the PRODUCT kernels, included
// from sara_amd/csrc/pyramid_kernels.hip (no copy), launched through
// launch_gaussian_blur() on one plane of `batch` synthetic w x h frames per
// radius of the default pyramid; reports us per launch and the fraction of the
// 8 TB/s HBM roofline (8 bytes per pixel).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off \
//         -I../../sara_amd/csrc -o blur_bench blur_bench.hip
//   ./blur_bench [w h batch [fma]]
#include "../../sara_amd/csrc/pyramid_kernels.hip"

#include <cstdio>
#include <vector>

// pyramid_kernels.hip only needs these two from the rest of the library
namespace sara_hip {
  sara_hip_status set_error(sara_hip_status code, const char*) { return code; }
  std::recursive_mutex& runtime_mutex()
  {
    static std::recursive_mutex m;
    return m;
  }
}  // namespace sara_hip

using namespace sara_hip;

static Taps taps_for(float sigma)
{
  int size = int(2 * 4.f * sigma + 1);
  size = size < 3 ? 3 : size;
  if (size % 2 == 0)
    ++size;
  Taps t{};
  t.size = size;
  float sum = 0.f;
  for (int i = 0; i < size; ++i)
  {
    const float d = float(i) - float(size / 2);
    t.k[i] = std::exp(-(d * d) / (2 * sigma * sigma));
    sum += t.k[i];
  }
  for (int i = 0; i < size; ++i)
    t.k[i] /= sum;
  return t;
}

int main(int argc, char** argv)
{
  const int w = argc > 3 ? atoi(argv[1]) : 1920, h = argc > 3 ? atoi(argv[2]) : 1080;
  const int batch = argc > 3 ? atoi(argv[3]) : 64;
  const bool fma = argc > 4 && atoi(argv[4]) != 0;
  const size_t plane = size_t(w) * h;
  float *src = nullptr, *dst = nullptr;
  hipMalloc(&src, plane * batch * sizeof(float));
  hipMalloc(&dst, plane * batch * sizeof(float));
  std::vector<float> hsrc(plane);
  unsigned s = 12345u;
  for (float& v : hsrc)
  {
    s = s * 1664525u + 1013904223u;
    v = float(s >> 8) / 16777216.f;
  }
  for (int b = 0; b < batch; ++b)
    hipMemcpy(src + plane * b, hsrc.data(), plane * sizeof(float), hipMemcpyHostToDevice);
  // sigma of the default pyramid's six blurs (k = 2^(1/3), sigma0 = 1.6, camera 0.5)
  const float sigmas[6] = {1.5198685f, 1.2262735f, 1.5450078f, 1.9465878f, 2.4525471f,
                           3.0900156f};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::printf("%d x %d x %d frames%s\n", w, h, batch, fma ? " (FMA option)" : "");
  for (float sigma : sigmas)
  {
    const Taps t = taps_for(sigma);
    for (int i = 0; i < 3; ++i)
      launch_gaussian_blur(src, plane, dst, plane, nullptr, 0, w, h, batch, t, nullptr,
                           nullptr, 0, fma);
    hipEventRecord(e0, nullptr);
    const int reps = 10;
    for (int i = 0; i < reps; ++i)
      launch_gaussian_blur(src, plane, dst, plane, nullptr, 0, w, h, batch, t, nullptr,
                           nullptr, 0, fma);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = 1e3 * ms / reps;
    std::printf("R = %2d: %8.1f us per launch, %6.0f GB/s = %.3f of 8 TB/s\n", t.size / 2,
                us, 8.0 * plane * batch / 1e3 / us, 8.0 * plane * batch / 1e3 / us / 8000.0);
  }
  return 0;
}
