// Synthetic benchmark frames (include/sara_synth.h).
// Host-only helper library, libsara_synth.so: no HIP, no dependency on the
// SIFT library.
#include "../../include/sara_synth.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <thread>
#include <vector>

namespace {

  struct SplitMix64
  {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next()
    {
      uint64_t z = (s += 0x9e3779b97f4a7c15ull);
      z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
      z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
      return z ^ (z >> 31);
    }
    //! uniform in [0, 1) with 53 random bits
    double uniform() { return double(next() >> 11) * (1.0 / 9007199254740992.0); }
    //! uniform integer in [0, n)
    int below(int n) { return int(uniform() * double(n)); }
  };

  void generate(int w, int h, uint64_t seed, float* out)
  {
    SplitMix64 rng(seed);
    const size_t px = size_t(w) * h;
    std::vector<double> img(px, 0.0);

    // ---- blobs ------------------------------------------------------------
    const int n = std::max<int>(1, int(px / 400));
    const double log_ratio = std::log(16.0 / 1.2);
    std::vector<double> ex;
    for (int i = 0; i < n; ++i)
    {
      const int cx = rng.below(w);
      const int cy = rng.below(h);
      const double rho = 1.2 * std::exp(rng.uniform() * log_ratio);
      double amp = 0.15 + 0.35 * rng.uniform();
      if (rng.next() & 1ull)
        amp = -amp;
      const int r = int(std::ceil(4.0 * rho));
      ex.resize(size_t(2 * r + 1));
      const double inv = 1.0 / (2.0 * rho * rho);
      for (int d = -r; d <= r; ++d)
        ex[size_t(d + r)] = std::exp(-double(d * d) * inv);
      const int y0 = std::max(0, cy - r), y1 = std::min(h - 1, cy + r);
      const int x0 = std::max(0, cx - r), x1 = std::min(w - 1, cx + r);
      for (int y = y0; y <= y1; ++y)
      {
        const double ay = amp * ex[size_t(y - cy + r)];
        double* row = img.data() + size_t(y) * w;
        for (int x = x0; x <= x1; ++x)
          row[x] += ay * ex[size_t(x - cx + r)];
      }
    }

    // ---- band-limited noise: N(0,1) (Box-Muller), Gaussian sigma = 1 ---------
    std::vector<double> nz(px), tmp(px);
    for (size_t i = 0; i < px; i += 2)
    {
      double u1 = rng.uniform();
      const double u2 = rng.uniform();
      if (u1 < 1e-300)
        u1 = 1e-300;
      const double m = std::sqrt(-2.0 * std::log(u1));
      const double a = 6.283185307179586476925286766559 * u2;
      nz[i] = m * std::cos(a);
      if (i + 1 < px)
        nz[i + 1] = m * std::sin(a);
    }
    double k[9];
    {
      double sum = 0.0;
      for (int d = -4; d <= 4; ++d)
        sum += (k[d + 4] = std::exp(-0.5 * double(d * d)));
      for (double& v : k)
        v /= sum;
    }
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x)
      {
        double s = 0.0;
        for (int d = -4; d <= 4; ++d)
        {
          const int xx = std::min(std::max(x + d, 0), w - 1);
          s += k[d + 4] * nz[size_t(y) * w + xx];
        }
        tmp[size_t(y) * w + x] = s;
      }
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x)
      {
        double s = 0.0;
        for (int d = -4; d <= 4; ++d)
        {
          const int yy = std::min(std::max(y + d, 0), h - 1);
          s += k[d + 4] * tmp[size_t(yy) * w + x];
        }
        nz[size_t(y) * w + x] = s;
      }
    double mean = 0.0;
    for (size_t i = 0; i < px; ++i)
      mean += nz[i];
    mean /= double(px);
    double var = 0.0;
    for (size_t i = 0; i < px; ++i)
      var += (nz[i] - mean) * (nz[i] - mean);
    const double sd = std::max(std::sqrt(var / double(px)), 1e-12);

    for (size_t i = 0; i < px; ++i)
    {
      const double v = 0.5 + img[i] + 0.02 * (nz[i] / sd);
      out[i] = float(std::min(std::max(v, 0.0), 1.0));
    }
  }

}  // namespace

extern "C" {

__attribute__((visibility("default"))) int sara_synth_frame(int width, int height,
                                                            uint64_t seed,
                                                            float* out)
{
  if (width < 1 || height < 1 || !out)
    return -1;
  generate(width, height, seed, out);
  return 0;
}

__attribute__((visibility("default"))) int sara_synth_batch(int width, int height,
                                                            int count,
                                                            uint64_t first_seed,
                                                            float* out,
                                                            int threads)
{
  if (width < 1 || height < 1 || count < 0 || (count > 0 && !out))
    return -1;
  if (threads <= 0)
    threads = int(std::thread::hardware_concurrency());
  threads = std::max(1, std::min(threads, count));
  const size_t px = size_t(width) * height;
  std::atomic<int> next{0};
  auto work = [&] {
    for (int i = next.fetch_add(1); i < count; i = next.fetch_add(1))
      generate(width, height, first_seed + uint64_t(i), out + size_t(i) * px);
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; ++t)
    pool.emplace_back(work);
  work();
  for (auto& t : pool)
    t.join();
  return 0;
}

}  // extern "C"
