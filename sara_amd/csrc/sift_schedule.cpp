// Host arithmetic of the C-ABI (include/sara_hip_sift.h): the parameter schedule
//   gaussian_pyramid           ImageProcessing/GaussianPyramid.hpp:33-125
//   make_gaussian_kernel       ImageProcessing/LinearFiltering.hpp:171-203
// kernel selection, the host-only entry points, the error text.
#include "sift_host.hpp"

using namespace sara_hip;
using namespace sara_hip::host;

namespace sara_hip { namespace host {
  thread_local std::string g_error = "";

  sara_hip_status fail(sara_hip_status code, const std::string& msg)
  {
    g_error = msg;
    return code;
  }


  // ---- kernel selection (sift_kernels.hpp) ---------------------------------
  thread_local const KernelSelection* t_selection = nullptr;
}}  // namespace sara_hip::host

namespace sara_hip {
  const KernelSelection& environment_selection()
  {
    static const KernelSelection env = [] {
      KernelSelection k;
      auto is = [](const char* name, const char* value) {
        const char* e = getenv(name);
        return e && std::string(e) == value;
      };
      k.blur_march = !is("SARA_HIP_BLUR", "tile");
      k.feature_march = !is("SARA_HIP_FEATURES", "tile");
      if (const char* e = getenv("SARA_HIP_MARCH_WAVES"))
        k.march_waves = std::max(64, atoi(e));
      if (const char* e = getenv("SARA_HIP_MARCH2_WAVES"))
        k.march2_waves = std::max(64, atoi(e));
      if (const char* e = getenv("SARA_HIP_MARCH_MIN_PIXELS"))
        k.march_min_pixels = size_t(atoll(e));
      if (const char* e = getenv("SARA_HIP_STRIP_GROUP"))
        k.strip_group = atoi(e);
      if (const char* e = getenv("SARA_HIP_GRAD_TILE_PIXELS"))
        k.grad_tile_pixels = atoll(e);
      if (const char* e = getenv("SARA_HIP_TILE_GEOMETRY"))
        k.tile_geometry = atoi(e);
      k.xcd_map = !is("SARA_HIP_XCD_MAP", "0");
      k.level_merge = !is("SARA_HIP_LEVELS", "0");
      return k;
    }();
    return env;
  }
  const KernelSelection& selection()
  {
    return t_selection ? *t_selection : environment_selection();
  }
  ScopedSelection::ScopedSelection(const KernelSelection* s)
    : before{t_selection}
  {
    t_selection = s;
  }
  ScopedSelection::~ScopedSelection() { t_selection = before; }
}  // namespace sara_hip

namespace sara_hip { namespace host {
  // ---- host restatement of the parameter schedule --------------------------

  // make_gaussian_kernel, ImageProcessing/LinearFiltering.hpp:171-203, is three
  // Eigen expressions; exp() and sum() are the two operations in them that are
  // not one correctly rounded IEEE operation, so their result depends on the
  // path Eigen takes in the reference's build (SARA_HIP_TAPS_*):
  //  * a scalar build: expf per tap, left-to-right sum;
  //  * the Release build (x86-64 baseline = SSE2, Packet4f): the dense
  //    assignment loop sends taps [0, 4*(n/4)) through pexp<Packet4f> and the
  //    rest through the scalar functor (expf); sum() keeps two packet
  //    accumulators over even / odd packets, adds them, adds the odd packet out,
  //    reduces as (a0 + a2) + (a1 + a3) and finishes with the scalar tail.
  // pexp is written from the published algorithm (Cephes: m = floor(x log2 e +
  // 1/2), r = x - m ln 2 in two parts, degree-5 polynomial, times 2^m); on SSE2
  // pmadd is a multiply and an add, each rounded (this file is compiled with
  // -ffp-contract=off).

  //! Eigen 3.4 pexp_float, one lane.
  float pexp_eigen34(float x0)
  {
    const float x = std::max(std::min(x0, 88.723f), -88.723f);
    const float m = std::floor(x * 1.44269504088896341f + 0.5f);
    float r = m * -0.693359375f + x;
    r = m * 2.12194440e-4f + r;
    const float r2 = r * r, r3 = r2 * r;
    float y = 1.9875691500E-4f * r + 1.3981999507E-3f;
    float y1 = 4.1665795894E-2f * r + 1.6666665459E-1f;
    const float y2 = r + 1.0f;
    y = y * r + 8.3334519073E-3f;
    y1 = y1 * r + 5.0000001201E-1f;
    y = y * r3 + y1;
    y = y * r2 + y2;
    return std::max(std::ldexp(y, int(m)), x0);
  }

  //! Eigen 3.3 pexp<Packet4f>, one lane: Horner form, (P(r) r^2 + r) + 1.
  float pexp_eigen33(float x0)
  {
    float x = std::max(std::min(x0, 88.3762626647950f), -88.3762626647949f);
    const float fx = std::floor(x * 1.44269504088896341f + 0.5f);
    const float hi = fx * 0.693359375f;
    float z = fx * -2.12194440e-4f;
    x = x - hi;
    x = x - z;
    z = x * x;
    float y = 1.9875691500E-4f;
    const float p[5] = {1.3981999507E-3f, 8.3334519073E-3f, 4.1665795894E-2f,
                        1.6666665459E-1f, 5.0000001201E-1f};
    for (float pi : p)
      y = y * x + pi;
    y = y * z + x;
    y = y + 1.0f;
    return std::max(std::ldexp(y, int(fx)), x0);
  }

  //! VectorXf::sum() on SSE2 (Redux.h, LinearVectorizedTraversal).
  float sum_eigen_sse2(const float* v, int n)
  {
    const int n4 = (n / 4) * 4, n8 = (n / 8) * 8;
    if (n4 == 0)
    {
      float res = v[0];
      for (int i = 1; i < n; ++i)
        res = res + v[i];
      return res;
    }
    float a[4] = {v[0], v[1], v[2], v[3]};
    if (n4 > 4)
    {
      float b[4] = {v[4], v[5], v[6], v[7]};
      for (int i = 8; i < n8; i += 8)
        for (int j = 0; j < 4; ++j)
        {
          a[j] = a[j] + v[i + j];
          b[j] = b[j] + v[i + 4 + j];
        }
      for (int j = 0; j < 4; ++j)
        a[j] = a[j] + b[j];
      if (n4 > n8)
        for (int j = 0; j < 4; ++j)
          a[j] = a[j] + v[n8 + j];
    }
    float res = (a[0] + a[2]) + (a[1] + a[3]);
    for (int i = n4; i < n; ++i)
      res = res + v[i];
    return res;
  }

  //! make_gaussian_kernel, ImageProcessing/LinearFiltering.hpp:171-203.
  std::vector<float> gaussian_taps(float sigma, float gauss_truncate,
                                   int arithmetic)
  {
    int size = int(2 * gauss_truncate * sigma + 1);
    size = std::max(3, size);
    if (size % 2 == 0)
      ++size;
    const int c = size / 2;
    std::vector<float> k(size);
    const float denom = 2 * (sigma * sigma);
    const bool packets = arithmetic != SARA_HIP_TAPS_LIBM_SERIAL;
    const int packets_end = packets ? (size / 4) * 4 : 0;
    for (int i = 0; i < size; ++i)
    {
      const float d = float(i) - float(c);
      const float x = -(d * d) / denom;
      if (i >= packets_end)
        k[i] = std::exp(x);
      else
        k[i] = arithmetic == SARA_HIP_TAPS_EIGEN34_SSE2 ? pexp_eigen34(x)
                                                        : pexp_eigen33(x);
    }
    float sum = 0.f;
    if (packets)
      sum = sum_eigen_sse2(k.data(), size);
    else
      for (int i = 0; i < size; ++i)
        sum += k[i];
    for (int i = 0; i < size; ++i)
      k[i] /= sum;
    return k;
  }

  bool to_taps(const std::vector<float>& k, Taps& t)
  {
    if (int(k.size()) > kMaxTaps)
      return false;
    t.size = int(k.size());
    std::memset(t.k, 0, sizeof(t.k));
    std::memcpy(t.k, k.data(), sizeof(float) * k.size());
    return true;
  }

  //! Geometry part of gaussian_pyramid(), GaussianPyramid.hpp:43-122.
  Schedule make_schedule(const sara_pyramid_params& p, int w, int h,
                         bool downscale_at_double_sigma)
  {
    Schedule s;
    s.resize_factor = std::pow(2.f, -static_cast<float>(p.first_octave_index));
    const float camera_sigma = p.scale_camera * s.resize_factor;
    const float init_sigma = p.scale_initial;
    if (p.first_octave_index < 0)
    {
      s.base_w = int(double(w) * double(s.resize_factor));
      s.base_h = int(double(h) * double(s.resize_factor));
    }
    else
    {
      if (camera_sigma < init_sigma)
      {
        s.init_blur = true;
        s.init_sigma =
            std::sqrt(init_sigma * init_sigma - camera_sigma * camera_sigma);
      }
      if (p.first_octave_index > 0)
      {
        const int f = int(std::round(1 / s.resize_factor));
        s.base_w = f > 0 ? w / f : 0;
        s.base_h = f > 0 ? h / f : 0;
      }
      else
      {
        s.base_w = w;
        s.base_h = h;
      }
    }
    const int l = std::min(s.base_w, s.base_h);
    const int b = p.image_padding_size;
    int n = 0;
    if (l > 0 && b > 0)
      n = std::min(static_cast<int>(std::log(double(float(l) / (2.f * float(b)))) /
                                    std::log(double(2.f))),
                   p.num_octaves_max);
    s.num_octaves = std::max(n, 0);
    // GaussianPyramid.hpp:97-100: floor(); round() is the scale at 2 sigma_0
    // the float value of k misses (SARA_HIP_OPT_DOWNSCALE_AT_DOUBLE_SIGMA).
    const double per_doubling =
        std::log(double(2.f)) / std::log(double(p.scale_geometric_factor));
    s.downscale_index = static_cast<int>(
        downscale_at_double_sigma ? std::round(per_doubling) : std::floor(per_doubling));
    s.oct.resize(s.num_octaves);
    for (int o = 0; o < s.num_octaves; ++o)
    {
      s.oct[o].factor = (o == 0) ? 1 / s.resize_factor : s.oct[o - 1].factor * 2;
      s.oct[o].w = (o == 0) ? s.base_w : s.oct[o - 1].w / 2;
      s.oct[o].h = (o == 0) ? s.base_h : s.oct[o - 1].h / 2;
    }
    return s;
  }

  sara_hip_status validate(const sara_pyramid_params& p, int padding)
  {
    if (p.scale_count_per_octave < 4)
      return fail(SARA_HIP_INVALID_PARAMS,
                  "Error: The extraction of DoG extrema needs (1 + 3) = 4 "
                  "scales per octave at the very minimum!");
    if (p.scale_count_per_octave > kMaxScales)
      return fail(SARA_HIP_INVALID_PARAMS, "scale_count_per_octave > 16");
    if (!(p.scale_geometric_factor > 1.f))
      return fail(SARA_HIP_INVALID_PARAMS, "scale_geometric_factor must be > 1");
    if (p.image_padding_size < 1)
      return fail(SARA_HIP_INVALID_PARAMS, "image_padding_size must be >= 1");
    if (padding < 1)
      return fail(SARA_HIP_INVALID_PARAMS,
                  "the extremum border padding must be >= 1 (the reference "
                  "reads out of bounds below that)");
    if (!(p.scale_initial > 0.f) || !(p.scale_camera >= 0.f))
      return fail(SARA_HIP_INVALID_PARAMS, "scales must be positive");
    return SARA_HIP_OK;
  }


}}  // namespace sara_hip::host

namespace sara_hip { namespace host {
//! ScaleTable::ori_bin_thr: thr[k] = smallest float >= 0 whose histogram bin
//! int(floor(double(a / float(2 pi) * 36))) (Orientation.hpp:118-119) is >= k,
//! by bisection on the bit patterns with the expression itself (+inf where no
//! angle of [0, 2 pi] gets there).
void orientation_bin_thresholds(float thr_out[40])
{
  auto bin_of = [](float a) {
    return int(std::floor(double(a / float(2 * M_PI) * 36)));
  };
  for (int kk = 0; kk < 40; ++kk)
  {
    uint32_t lo = 0u, hi = 0x40c91000u;  // [0, a little above float(2 pi)]
    float thr = std::numeric_limits<float>::infinity();
    float top;
    std::memcpy(&top, &hi, 4);
    if (bin_of(top) >= kk)
    {
      while (lo < hi)  // first bit pattern (= first float >= 0) with bin >= kk
      {
        const uint32_t mid = lo + (hi - lo) / 2;
        float a;
        std::memcpy(&a, &mid, 4);
        if (bin_of(a) >= kk)
          hi = mid;
        else
          lo = mid + 1;
      }
      std::memcpy(&thr, &lo, 4);
    }
    thr_out[kk] = thr;
  }
}

}}  // namespace sara_hip::host

extern "C" {


const char* sara_hip_last_error(void) { return g_error.c_str(); }

int sara_hip_version(void) { return 100; }

int sara_hip_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

void sara_hip_default_pyramid_params(sara_pyramid_params* p)
{
  p->first_octave_index = -1;
  p->scale_count_per_octave = 3 + 3;
  p->scale_geometric_factor = std::pow(2.f, 1.f / 3.f);
  p->image_padding_size = 1;
  p->scale_camera = 0.5f;
  p->scale_initial = 1.6f;
  p->num_octaves_max = INT_MAX;
}

void sara_hip_default_sift_params(sara_sift_params* p)
{
  sara_hip_default_pyramid_params(&p->pyramid);
  p->gauss_truncate = 4.f;
  p->extremum_thres = 0.01f;
  p->edge_ratio_thres = 10.f;
  p->extremum_refinement_iter = 5;
}

int sara_hip_pyramid_octave_count(const sara_pyramid_params* p, int width,
                                  int height)
{
  if (!p)
    return 0;
  return make_schedule(*p, width, height).num_octaves;
}

sara_hip_status sara_hip_pyramid_octave_info(const sara_pyramid_params* p,
                                             int width, int height, int octave,
                                             int* ow, int* oh, float* factor)
{
  if (!p)
    return fail(SARA_HIP_INVALID_PARAMS, "null params");
  const Schedule s = make_schedule(*p, width, height);
  if (octave < 0 || octave >= s.num_octaves)
    return fail(SARA_HIP_OUT_OF_RANGE, "octave index out of range");
  if (ow)
    *ow = s.oct[octave].w;
  if (oh)
    *oh = s.oct[octave].h;
  if (factor)
    *factor = s.oct[octave].factor;
  return SARA_HIP_OK;
}

int sara_hip_make_gaussian_kernel(float sigma, float gauss_truncate, float* taps,
                                  int capacity)
{
  const auto k = gaussian_taps(sigma, gauss_truncate);
  if (int(k.size()) > capacity || !taps)
    return -int(k.size());
  std::memcpy(taps, k.data(), sizeof(float) * k.size());
  return int(k.size());
}

int sara_hip_make_gaussian_kernel_with(int arithmetic, float sigma,
                                       float gauss_truncate, float* taps,
                                       int capacity)
{
  if (arithmetic < SARA_HIP_TAPS_LIBM_SERIAL || arithmetic > SARA_HIP_TAPS_EIGEN33_SSE2)
    return 0;
  const auto k = gaussian_taps(sigma, gauss_truncate, arithmetic);
  if (int(k.size()) > capacity || !taps)
    return -int(k.size());
  std::memcpy(taps, k.data(), sizeof(float) * k.size());
  return int(k.size());
}

}  // extern "C"
