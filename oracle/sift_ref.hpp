// ========================================================================== //
// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Plain C++17 CPU restatement of Sara's default (non-Halide) CPU SIFT path.
// It is the parity checker for the HIP product and the "port" CPU baseline of
// bench.py.  Nothing under sara_amd/ may include, link or call this code: only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
//
// Every function cites the reference file:line it follows (paths relative to
// /root/reference/cpp/src/DO/Sara).  No reference source is copied: the
// reference is Eigen-based templates, this is flat loops over float buffers.
//
// PARITY PIN STATUS
//   * pinned by the reference's own tests (restated in tests/test_oracle_*.py):
//     convolve_array, row/column filter, Gaussian kernel (L2 1e-5), downscale,
//     enlarge, gradient, 2-D hessian, scale-space extremum predicates, pyramid
//     octave count, DoG blob test, histogram smoothing, hard binning, single
//     peak recovery, descriptor API consistency.
//   * "parity unpinned" (the reference holds no known-answer test, and the
//     reference itself cannot be built here: Eigen >= 3.4 is neither vendored
//     nor installed).  Round 2 restates the PUBLISHED Eigen 3.4.0 algorithms
//     for the two pieces that decide something: the float
//     SelfAdjointEigenSolver<Matrix3f> behind refine_extremum's definiteness
//     test, and the Packet4f reduction order of normalize()'s squaredNorm();
//     both are compared with round 1's stand-ins (Sylvester minors in double,
//     left-to-right sum) site by site / row by row, see DESIGN.md section 2.
//     Still at the last-ulp level only: make_gaussian_kernel's packet exp()
//     and sum().  Round 6: the arithmetic is a selectable variant
//     (tap_variant(): expf + serial sum, the Eigen 3.4 / 3.3 SSE2 models,
//     one-ulp perturbations) and tests/golden/sensitivity.json holds what each
//     variant does to the keypoints of the real-image pack and of 64 synthetic
//     1080p frames (the reference's own test accepts 1e-5 on the taps).
// ========================================================================== //
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <vector>

#ifdef _OPENMP
#  include <omp.h>
#endif

namespace sara_ref {

  //! Switches of the "corrected" detector mode;
  //! 0 = the reference's default (non-Halide) build, which everything pinned in
  //! this file describes.
  //!  * kModeSignedExtremumType: the host loop of the DO_SARA_USE_HALIDE branch
  //!    (RefineExtremum.cpp:226-361): the map is int8, so minima reach
  //!    refine_extremum as type -1 and are refined like maxima (quirk Q2 gone),
  //!    and a refined scale outside (sigma(s) / 4, 4 sigma(s)) rejects the
  //!    site (:307-325).  The map comes from the restated Halide classifier
  //!    (halide_is_dog_extremum below): every pixel, replicated borders, strict
  //!    contrast test, Halide's own hessian.
  //!  * kModeDownscaleAtDoubleSigma: octave o + 1 is sub-sampled from the
  //!    scale whose sigma is 2 sigma_0, round(log 2 / log k), instead of
  //!    floor(...) which float rounding of k turns into one scale lower
  //!    (quirk Q3, GaussianPyramid.hpp:97-100).
  //! PARITY UNPINNED: no build of the reference produces this combination.
  enum
  {
    kModeSignedExtremumType = 1,
    kModeDownscaleAtDoubleSigma = 2
  };
  inline int& detector_mode()
  {
    static int mode = 0;
    return mode;
  }


  // ------------------------------------------------------------------------ //
  // Containers.  Image.hpp:45-181: x-fastest storage, pixel (x,y) at y*w+x.
  // ------------------------------------------------------------------------ //
  struct Image
  {
    int w = 0, h = 0;
    std::vector<float> d;
    Image() = default;
    Image(int w_, int h_) : w(w_), h(h_), d(size_t(w_) * h_) {}
    float& operator()(int x, int y) { return d[size_t(y) * w + x]; }
    const float& operator()(int x, int y) const { return d[size_t(y) * w + x]; }
  };

  //! Image<Vector2f>: (mag, ori) interleaved, 8 B per pixel.
  struct Image2
  {
    int w = 0, h = 0;
    std::vector<float> d;
    Image2() = default;
    Image2(int w_, int h_) : w(w_), h(h_), d(size_t(w_) * h_ * 2) {}
    float* at(int x, int y) { return &d[(size_t(y) * w + x) * 2]; }
    const float* at(int x, int y) const { return &d[(size_t(y) * w + x) * 2]; }
  };

  //! ImageProcessing/ImagePyramid.hpp:29-198.
  struct PyramidParams
  {
    int first_octave_index = -1;
    int scale_count_per_octave = 6;
    float scale_geometric_factor = std::pow(2.f, 1.f / 3.f);
    int image_padding_size = 1;
    float scale_camera = 0.5f;
    float scale_initial = 1.6f;
    int num_octaves_max = std::numeric_limits<int>::max();
  };

  //! ImageProcessing/ImagePyramid.hpp:206-340.
  template <typename Img>
  struct Pyramid
  {
    std::vector<std::vector<Img>> octaves;
    std::vector<float> oct_scaling_factors;
    float scale_initial = 0.f;
    float scale_geometric_factor = 0.f;

    void reset(int num_octaves, int num_scales, float s0, float k)
    {
      octaves.assign(num_octaves, std::vector<Img>(num_scales));
      oct_scaling_factors.assign(num_octaves, 0.f);
      scale_initial = s0;
      scale_geometric_factor = k;
    }
    int octave_count() const { return int(octaves.size()); }
    int scale_count_per_octave() const { return int(octaves.front().size()); }
    Img& operator()(int s, int o) { return octaves[o][s]; }
    const Img& operator()(int s, int o) const { return octaves[o][s]; }
    //! ImagePyramid.hpp:316-319 — std::pow(float, int) promotes to double.
    double scale_relative_to_octave(int s) const
    {
      return std::pow(double(scale_geometric_factor), double(s)) *
             double(scale_initial);
    }
  };

  //! Features/Feature.hpp:40-179 — 48-byte layout (Matrix2f is 16-B aligned).
  struct alignas(16) OERegion
  {
    float coords[2] = {0, 0};
    float _pad0[2] = {0, 0};
    float shape_matrix[4] = {0, 0, 0, 0};  // column-major 2x2
    float orientation = 0;
    float extremum_value = 0;
    std::uint8_t type = 11;          // Type::Undefined
    std::int8_t extremum_type = -2;  // ExtremumType::Undefined
    std::uint8_t _pad1[6] = {0, 0, 0, 0, 0, 0};
  };
  static_assert(sizeof(OERegion) == 48, "OERegion must be 48 bytes");

  //! Feature.hpp:79-83: shape = I * float(pow(double(scale), -2.0)).
  inline OERegion make_oeregion(float x, float y, float scale)
  {
    OERegion f;
    f.coords[0] = x;
    f.coords[1] = y;
    const float c = static_cast<float>(std::pow(double(scale), -2.0));
    f.shape_matrix[0] = c;
    f.shape_matrix[3] = c;
    return f;
  }

  //! Features/Feature.cpp:28-39 for an isotropic shape matrix c*I: JacobiSVD
  //! returns singular values (c, c), U = I; radius = 1/sqrt(c) (float ops).
  //! General anisotropic matrices are not produced on this path.
  inline float oeregion_scale(const OERegion& f)
  {
    const float c = f.shape_matrix[0];
    const float r0 = 1.f / std::sqrt(c);
    const float x = r0 * 1.f;
    const float y = 0.f;
    return std::sqrt(x * x + y * y);
  }

  // ------------------------------------------------------------------------ //
  // Colour conversion (config 1 input only).
  // Core/Pixel/SmartColorConversion.hpp:237-246,
  // Core/Pixel/ChannelConversion.hpp:40-54, Core/Pixel/ColorConversion.hpp:26-34.
  // ------------------------------------------------------------------------ //
  inline float rgb8_to_gray32f(std::uint8_t r, std::uint8_t g, std::uint8_t b)
  {
    const double rd = (double(r) - 0.0) / 255.0;
    const double gd = (double(g) - 0.0) / 255.0;
    const double bd = (double(b) - 0.0) / 255.0;
    const double gray = 0.2125 * rd + 0.7154 * gd + 0.0721 * bd;
    return static_cast<float>(gray);
  }

  // ------------------------------------------------------------------------ //
  // Linear filtering.
  // ------------------------------------------------------------------------ //

  //! ImageProcessing/LinearFiltering.hpp:43-63 — correlation, in place,
  //! taps accumulated left to right from 0.f (mul then add, no FMA: the
  //! reference is built -O3 for baseline x86-64).
  inline void convolve_array(float* signal, const float* kernel,
                             int signal_size, int kernel_size)
  {
    for (int i = 0; i < signal_size; ++i)
    {
      float sum = 0.f;
      for (int j = 0; j < kernel_size; ++j)
        sum += signal[i + j] * kernel[j];
      signal[i] = sum;
    }
  }

  //! ImageProcessing/LinearFiltering.hpp:76-107 (replicate borders; the omp
  //! pragma is at :90).
  inline void apply_row_based_filter(const Image& src, Image& dst,
                                     const float* kernel, int kernel_size)
  {
    if (src.w != dst.w || src.h != dst.h)
      throw std::domain_error{
          "Source and destination image sizes are not equal!"};
    const int w = src.w, h = src.h, half = kernel_size / 2;
#pragma omp parallel for
    for (int y = 0; y < h; ++y)
    {
      std::vector<float> buffer(w + half * 2);
      for (int x = 0; x < half; ++x)
        buffer[x] = src(0, y);
      for (int x = 0; x < w; ++x)
        buffer[half + x] = src(x, y);
      for (int x = 0; x < half; ++x)
        buffer[w + half + x] = src(w - 1, y);
      convolve_array(buffer.data(), kernel, w, kernel_size);
      for (int x = 0; x < w; ++x)
        dst(x, y) = buffer[x];
    }
  }

  //! ImageProcessing/LinearFiltering.hpp:118-149 (omp pragma at :132).
  inline void apply_column_based_filter(const Image& src, Image& dst,
                                        const float* kernel, int kernel_size)
  {
    if (src.w != dst.w || src.h != dst.h)
      throw std::domain_error{
          "Source and destination image sizes are not equal!"};
    const int w = src.w, h = src.h, half = kernel_size / 2;
#pragma omp parallel for
    for (int x = 0; x < w; ++x)
    {
      std::vector<float> buffer(h + half * 2);
      for (int y = 0; y < half; ++y)
        buffer[y] = src(x, 0);
      for (int y = 0; y < h; ++y)
        buffer[half + y] = src(x, y);
      for (int y = 0; y < half; ++y)
        buffer[h + half + y] = src(x, h - 1);
      convolve_array(buffer.data(), kernel, h, kernel_size);
      for (int y = 0; y < h; ++y)
        dst(x, y) = buffer[y];
    }
  }

  // ------------------------------------------------------------------------ //
  // make_gaussian_kernel and the arithmetic the reference's build gives it.
  //
  // ImageProcessing/LinearFiltering.hpp:171-203 is three Eigen expressions:
  //   kernel = LinSpaced(n, 0, n - 1);                    (exact integers)
  //   kernel.array() = (-(kernel.array() - c).square() / (2 sigma^2)).exp();
  //   kernel /= kernel.sum();
  // Everything but exp() and sum() is a correctly rounded IEEE operation on the
  // same operands whatever the evaluation path.  exp() and sum() are not: in
  // the reference's Release build (x86-64 baseline => SSE2, Packet4f; Eigen
  // 3.4.0 on the CI image) the dense assignment loop
  // (AssignEvaluator.h, LinearVectorizedTraversal) sends the first 4*(n/4) taps
  // through pexp<Packet4f> = pexp_float (GenericPacketMathFunctions.h: the
  // Cephes range reduction with a degree-5 polynomial) and the n % 4 trailing
  // ones through the scalar functor, i.e. libm's expf; sum() goes through
  // redux_impl<LinearVectorizedTraversal, NoUnrolling> (Redux.h): two Packet4f
  // accumulators over even / odd packets, acc0 + acc1, one more packet when
  // their number is odd, predux = (a0 + a2) + (a1 + a3), then the scalar tail.
  //
  // No Eigen on this image => none of this can be compiled against the real
  // thing: PARITY UNPINNED.  The oracle therefore offers the arithmetic as a
  // selectable VARIANT and tests/golden/make_sensitivity.py measures what each
  // choice does to the keypoints (tests/golden/sensitivity.json, DESIGN.md
  // section 5): the distance oracle -> reference gets a number instead of an
  // assumption.
  //   kTapsExpfSerial   expf + left-to-right sum (rounds 1-5; the default the
  //                     product's host code shares)
  //   kTapsEigen34Sse   the model above, Eigen 3.4.0's pexp_float written from
  //                     the published algorithm
  //   kTapsEigen33Sse   the same with Eigen 3.3's pexp (plain Horner form)
  //   kTapsUlp*         the default taps moved by one ulp each: seeded random
  //                     signs (-1 / 0 / +1), all up, all down, alternating,
  //                     centre up + tails down ("narrow") and its opposite -
  //                     envelopes of what ANY exp()/sum() within 1 ulp per tap
  //                     could do.
  // ------------------------------------------------------------------------ //
  enum
  {
    kTapsExpfSerial = 0,
    kTapsEigen34Sse = 1,
    kTapsEigen33Sse = 2,
    kTapsUlpRandom = 3,
    kTapsUlpPlus = 4,
    kTapsUlpMinus = 5,
    kTapsUlpAlternate = 6,
    kTapsUlpNarrow = 7,
    kTapsUlpWide = 8,
    kTapsVariantCount = 9
  };
  struct TapVariant
  {
    int kind = kTapsExpfSerial;
    std::uint32_t seed = 0;
  };
  inline TapVariant& tap_variant()
  {
    static TapVariant v;
    return v;
  }

  //! Eigen 3.4.0 pexp_float (GenericPacketMathFunctions.h), one lane of it,
  //! without FMA (pmadd = multiply, round, add, round on SSE2).  pldexp_generic
  //! multiplies by exact powers of two: exact for the normal results here.
  inline float eigen34_pexp_lane(float x0)
  {
    const float x = std::max(std::min(x0, 88.723f), -88.723f);
    const float m = std::floor(x * 1.44269504088896341f + 0.5f);
    float r = m * -0.693359375f + x;
    r = m * 2.12194440e-4f + r;
    const float r2 = r * r;
    const float r3 = r2 * r;
    float y = 1.9875691500E-4f * r + 1.3981999507E-3f;
    float y1 = 4.1665795894E-2f * r + 1.6666665459E-1f;
    const float y2 = r + 1.0f;
    y = y * r + 8.3334519073E-3f;
    y1 = y1 * r + 5.0000001201E-1f;
    y = y * r3 + y1;
    y = y * r2 + y2;
    return std::max(std::ldexp(y, int(m)), x0);
  }

  //! Eigen 3.3 pexp<Packet4f> (arch/SSE/MathFunctions.h): same reduction,
  //! Horner evaluation, y = (P(r) r^2 + r) + 1.
  inline float eigen33_pexp_lane(float x0)
  {
    float x = std::max(std::min(x0, 88.3762626647950f), -88.3762626647949f);
    const float fx = std::floor(x * 1.44269504088896341f + 0.5f);
    const float tmp = fx * 0.693359375f;
    float z = fx * -2.12194440e-4f;
    x = x - tmp;
    x = x - z;
    z = x * x;
    float y = 1.9875691500E-4f;
    y = y * x + 1.3981999507E-3f;
    y = y * x + 8.3334519073E-3f;
    y = y * x + 4.1665795894E-2f;
    y = y * x + 1.6666665459E-1f;
    y = y * x + 5.0000001201E-1f;
    y = y * z + x;
    y = y + 1.0f;
    return std::max(std::ldexp(y, int(fx)), x0);
  }

  //! VectorXf::sum() through redux_impl<LinearVectorizedTraversal,
  //! NoUnrolling> with Packet4f (Redux.h; the data of a dynamic vector is
  //! 16-byte aligned, so the packets start at coefficient 0).
  inline float eigen_sum_sse(const float* v, int n)
  {
    const int aligned = (n / 4) * 4, aligned2 = (n / 8) * 8;
    if (aligned == 0)
    {
      float res = v[0];
      for (int i = 1; i < n; ++i)
        res = res + v[i];
      return res;
    }
    float a0[4] = {v[0], v[1], v[2], v[3]};
    if (aligned > 4)
    {
      float a1[4] = {v[4], v[5], v[6], v[7]};
      for (int i = 8; i < aligned2; i += 8)
        for (int j = 0; j < 4; ++j)
        {
          a0[j] = a0[j] + v[i + j];
          a1[j] = a1[j] + v[i + 4 + j];
        }
      for (int j = 0; j < 4; ++j)
        a0[j] = a0[j] + a1[j];
      if (aligned > aligned2)
        for (int j = 0; j < 4; ++j)
          a0[j] = a0[j] + v[aligned2 + j];
    }
    float res = (a0[0] + a0[2]) + (a0[1] + a0[3]);
    for (int i = aligned; i < n; ++i)
      res = res + v[i];
    return res;
  }

  //! One ulp up (+1) or down (-1) of a positive finite float.
  inline float step_ulp(float v, int dir)
  {
    if (dir == 0)
      return v;
    return std::nextafter(v, dir > 0 ? std::numeric_limits<float>::infinity()
                                     : -std::numeric_limits<float>::infinity());
  }

  //! ImageProcessing/LinearFiltering.hpp:171-203 under tap_variant().
  inline std::vector<float> make_gaussian_kernel(float sigma,
                                                 float gauss_truncate = 4.f)
  {
    const TapVariant variant = tap_variant();
    int kernel_size = int(2 * gauss_truncate * sigma + 1);
    kernel_size = std::max(3, kernel_size);
    if (kernel_size % 2 == 0)
      ++kernel_size;
    const int c = kernel_size / 2;
    std::vector<float> kernel(kernel_size);
    const float denom = 2 * (sigma * sigma);
    const bool eigen = variant.kind == kTapsEigen34Sse ||
                       variant.kind == kTapsEigen33Sse;
    const int packets_end = eigen ? (kernel_size / 4) * 4 : 0;
    for (int i = 0; i < kernel_size; ++i)
    {
      const float d = float(i) - float(c);
      const float x = -(d * d) / denom;
      if (i < packets_end)
        kernel[i] = variant.kind == kTapsEigen34Sse ? eigen34_pexp_lane(x)
                                                    : eigen33_pexp_lane(x);
      else
        kernel[i] = std::exp(x);
    }
    float sum = 0.f;
    if (eigen)
      sum = eigen_sum_sse(kernel.data(), kernel_size);
    else
      for (int i = 0; i < kernel_size; ++i)
        sum += kernel[i];
    for (int i = 0; i < kernel_size; ++i)
      kernel[i] /= sum;

    if (variant.kind >= kTapsUlpRandom && variant.kind <= kTapsUlpWide)
    {
      std::uint32_t sigma_bits;
      std::memcpy(&sigma_bits, &sigma, sizeof(sigma_bits));
      for (int i = 0; i < kernel_size; ++i)
      {
        int dir = 0;
        switch (variant.kind)
        {
        case kTapsUlpRandom:
        {
          // SplitMix64 of (seed, sigma, tap): the same sigma gets the same
          // taps in every octave, as any deterministic exp() would give
          std::uint64_t z = (std::uint64_t(variant.seed) << 32 | sigma_bits) +
                            0x9E3779B97F4A7C15ull * std::uint64_t(i + 1);
          z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
          z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
          z ^= z >> 31;
          dir = int(z % 3) - 1;
          break;
        }
        case kTapsUlpPlus: dir = 1; break;
        case kTapsUlpMinus: dir = -1; break;
        case kTapsUlpAlternate: dir = (i & 1) ? 1 : -1; break;
        case kTapsUlpNarrow: dir = (i == c) ? 1 : -1; break;
        case kTapsUlpWide: dir = (i == c) ? -1 : 1; break;
        default: break;
        }
        kernel[i] = step_ulp(kernel[i], dir);
      }
    }
    return kernel;
  }

  //! ImageProcessing/LinearFiltering.cpp:30-68 (#else branch): rows src->dst,
  //! then columns dst->dst in place.
  inline void apply_gaussian_filter(const Image& src, Image& dst, float sigma,
                                    float gauss_truncate = 4.f)
  {
    if (src.w != dst.w || src.h != dst.h)
      throw std::domain_error{
          "Source and destination image sizes are not equal!"};
    const auto kernel = make_gaussian_kernel(sigma, gauss_truncate);
    apply_row_based_filter(src, dst, kernel.data(), int(kernel.size()));
    apply_column_based_filter(dst, dst, kernel.data(), int(kernel.size()));
  }

  //! ImageProcessing/LinearFiltering.hpp:445-454.
  inline Image gaussian(const Image& src, float sigma,
                        float gauss_truncate = 4.f)
  {
    Image dst(src.w, src.h);
    apply_gaussian_filter(src, dst, sigma, gauss_truncate);
    return dst;
  }

  // ------------------------------------------------------------------------ //
  // Resize.
  // ------------------------------------------------------------------------ //

  //! ImageProcessing/Resize.cpp:31-62 (#else branch): nearest neighbour with
  //! float index arithmetic.
  inline void scale(const Image& src, Image& dst)
  {
    const float sx = float(src.w) / float(dst.w);
    const float sy = float(src.h) / float(dst.h);
    const int w = dst.w;
    const int wh = dst.w * dst.h;
#pragma omp parallel for
    for (int xy = 0; xy < wh; ++xy)
    {
      const int y = xy / w;
      const int x = xy - y * w;
      const int xi = int(float(x) * sx);
      const int yi = int(float(y) * sy);
      dst(x, y) = src(xi, yi);
    }
  }

  //! ImageProcessing/Resize.cpp:64-84.
  inline Image downscale(const Image& src, int fact)
  {
    Image dst(src.w / fact, src.h / fact);
    scale(src, dst);
    return dst;
  }

  //! ImageProcessing/Interpolation.hpp:33-78 specialised to 2-D float images:
  //! bilinear in double, far border replicated, x-fastest accumulation.
  inline double interpolate(const Image& image, double px, double py)
  {
    if (px < 0 || px >= image.w || py < 0 || py >= image.h)
      throw std::out_of_range{
          "Cannot interpolate: position is out of image domain"};
    double ipx, ipy;
    const double fx = std::modf(px, &ipx);
    const double fy = std::modf(py, &ipy);
    const int sx = int(ipx), sy = int(ipy);
    double value = 0.;
    for (int yy = sy; yy < sy + 2; ++yy)
      for (int xx = sx; xx < sx + 2; ++xx)
      {
        double weight = 1.;
        weight *= (xx == sx) ? (1. - fx) : fx;
        weight *= (yy == sy) ? (1. - fy) : fy;
        const int ox = xx < image.w ? 0 : -1;
        const int oy = yy < image.h ? 0 : -1;
        value += weight * double(image(xx + ox, yy + oy));
      }
    return value;
  }

  //! ImageProcessing/Resize.cpp:86-128 (#else branch).
  inline void enlarge(const Image& src, Image& dst)
  {
    if (dst.w < src.w || dst.h < src.h)
      throw std::range_error{"The destination image must have smaller sizes "
                             "than the source image!"};
    if (std::min(dst.w, dst.h) <= 0)
      throw std::range_error{
          "The sizes of the destination image must be positive!"};
    const int wh = dst.w * dst.h;
    const double sx = double(src.w) / double(dst.w);
    const double sy = double(src.h) / double(dst.h);
#pragma omp parallel for
    for (int xy = 0; xy < wh; ++xy)
    {
      const int w = dst.w;
      const int y = xy / w;
      const int x = xy - y * w;
      dst(x, y) =
          static_cast<float>(interpolate(src, double(x) * sx, double(y) * sy));
    }
  }

  //! ImageProcessing/Resize.hpp:190-216: enlarge(image, double fact).
  inline Image enlarge(const Image& src, double fact)
  {
    Image dst(int(double(src.w) * fact), int(double(src.h) * fact));
    enlarge(src, dst);
    return dst;
  }

  // ------------------------------------------------------------------------ //
  // Gaussian pyramid / DoG pyramid.
  // ------------------------------------------------------------------------ //

  //! ImageProcessing/GaussianPyramid.hpp:33-125, with quirks Q3-Q6 of
  //! SURVEY.md: unqualified log()/sqrt() resolve to the double C functions.
  inline Pyramid<Image> gaussian_pyramid(const Image& image,
                                         const PyramidParams& params,
                                         float gauss_truncate = 4.f)
  {
    const float resize_factor =
        std::pow(2.f, -static_cast<float>(params.first_octave_index));
    const float camera_sigma = params.scale_camera * resize_factor;
    const float init_sigma = params.scale_initial;

    Image I;
    if (params.first_octave_index < 0)
      I = enlarge(image, double(resize_factor));
    else if (params.first_octave_index > 0)
    {
      if (camera_sigma < init_sigma)
      {
        const float sigma = std::sqrt(init_sigma * init_sigma -
                                      camera_sigma * camera_sigma);
        I = gaussian(image, sigma, gauss_truncate);
      }
      else
        I = image;
      I = downscale(I, int(std::round(1 / resize_factor)));
    }
    else
    {
      if (camera_sigma < init_sigma)
      {
        const float sigma = std::sqrt(init_sigma * init_sigma -
                                      camera_sigma * camera_sigma);
        I = gaussian(image, sigma);  // default truncate 4 (Q4)
      }
      else
        I = image;
    }

    const int l = std::min(I.w, I.h);
    const int b = params.image_padding_size;
    const int num_octaves = std::min(
        static_cast<int>(std::log(double(float(l) / (2.f * float(b)))) /
                         std::log(double(2.f))),
        params.num_octaves_max);

    const float k = params.scale_geometric_factor;
    const int num_scales = params.scale_count_per_octave;
    const double scales_per_doubling = std::log(double(2.f)) / std::log(double(k));
    const int downscale_index =
        (detector_mode() & kModeDownscaleAtDoubleSigma)
            ? static_cast<int>(std::round(scales_per_doubling))
            : static_cast<int>(std::floor(scales_per_doubling));

    Pyramid<Image> G;
    G.reset(std::max(num_octaves, 0), num_scales, init_sigma, k);

    for (int o = 0; o < num_octaves; ++o)
    {
      G.oct_scaling_factors[o] =
          (o == 0) ? 1 / resize_factor : G.oct_scaling_factors[o - 1] * 2;

      float sigma_s_1 = init_sigma;
      if (o == 0)
        G(0, o) = std::move(I);
      else
        G(0, o) = downscale(G(downscale_index, o - 1), 2);

      for (int s = 1; s < num_scales; ++s)
      {
        const float ks = k * sigma_s_1;
        const double sigma = std::sqrt(double(ks * ks - sigma_s_1 * sigma_s_1));
        G(s, o) = gaussian(G(s - 1, o), static_cast<float>(sigma));
        sigma_s_1 *= k;
      }
    }
    return G;
  }

  //! ImageProcessing/GaussianPyramid.cpp:23-51 (#else branch).
  inline Pyramid<Image> difference_of_gaussians_pyramid(const Pyramid<Image>& G)
  {
    Pyramid<Image> D;
    D.reset(G.octave_count(), G.scale_count_per_octave() - 1, G.scale_initial,
            G.scale_geometric_factor);
    for (int o = 0; o < D.octave_count(); ++o)
    {
      D.oct_scaling_factors[o] = G.oct_scaling_factors[o];
      for (int s = 0; s < D.scale_count_per_octave(); ++s)
      {
        const Image& a = G(s + 1, o);
        const Image& b = G(s, o);
        Image& d = D(s, o);
        d = Image(b.w, b.h);
        const size_t n = d.d.size();
        for (size_t i = 0; i < n; ++i)
          d.d[i] = a.d[i] - b.d[i];
      }
    }
    return D;
  }

  // ------------------------------------------------------------------------ //
  // Differential operators.
  // ------------------------------------------------------------------------ //

  //! ImageProcessing/Differential.hpp:46-61: central difference / 2, one-sided
  //! (f1 - f0)/2 on the borders.
  inline void gradient_at(const Image& f, int x, int y, float& gx, float& gy)
  {
    if (x == 0)
      gx = (f(x + 1, y) - f(x, y)) / 2;
    else if (x == f.w - 1)
      gx = (f(x, y) - f(x - 1, y)) / 2;
    else
      gx = (f(x + 1, y) - f(x - 1, y)) / 2;
    if (y == 0)
      gy = (f(x, y + 1) - f(x, y)) / 2;
    else if (y == f.h - 1)
      gy = (f(x, y) - f(x, y - 1)) / 2;
    else
      gy = (f(x, y + 1) - f(x, y - 1)) / 2;
  }

  //! ImageProcessing/Differential.hpp:191-226: 2-D Hessian with clamped
  //! neighbours.  H = [hxx hxy; hxy hyy].
  inline void hessian_at(const Image& f, int x, int y, float& hxx, float& hxy,
                         float& hyy)
  {
    const int nx = (x == f.w - 1) ? 0 : 1, px = (x == 0) ? 0 : -1;
    const int ny = (y == f.h - 1) ? 0 : 1, py = (y == 0) ? 0 : -1;
    hxx = f(x + nx, y) - 2.f * f(x, y) + f(x + px, y);
    hyy = f(x, y + ny) - 2.f * f(x, y) + f(x, y + py);
    hxy = (f(x + nx, y + ny) - f(x + px, y + ny) - f(x + nx, y + py) +
           f(x + px, y + py)) /
          4.f;
  }

  //! FeatureDescriptors/Orientation.cpp:24-56 (#else branch):
  //! (2*||grad||, atan2f(gy, gx)) in float.
  inline Image2 gradient_polar_coordinates(const Image& f)
  {
    Image2 out(f.w, f.h);
    for (int y = 0; y < f.h; ++y)
      for (int x = 0; x < f.w; ++x)
      {
        float gx, gy;
        gradient_at(f, x, y, gx, gy);
        const float r = 2 * std::sqrt(gx * gx + gy * gy);
        const float theta = std::atan2(gy, gx);
        float* p = out.at(x, y);
        p[0] = r;
        p[1] = theta;
      }
    return out;
  }

  //! FeatureDescriptors/Orientation.hpp:69-86.
  inline Pyramid<Image2> gradient_polar_coordinates(const Pyramid<Image>& G)
  {
    Pyramid<Image2> P;
    P.reset(G.octave_count(), G.scale_count_per_octave(), G.scale_initial,
            G.scale_geometric_factor);
    for (int o = 0; o < G.octave_count(); ++o)
    {
      P.oct_scaling_factors[o] = G.oct_scaling_factors[o];
      for (int s = 0; s < G.scale_count_per_octave(); ++s)
        P(s, o) = gradient_polar_coordinates(G(s, o));
    }
    return P;
  }

  // ------------------------------------------------------------------------ //
  // Extrema.
  // ------------------------------------------------------------------------ //

  //! ImageProcessing/Extrema.hpp:28-46.
  template <typename Compare>
  inline bool compare_with_neighborhood3(float val, int x, int y,
                                         const Image& I, bool with_center,
                                         Compare cmp)
  {
    for (int v = -1; v <= 1; ++v)
      for (int u = -1; u <= 1; ++u)
      {
        if (u == 0 && v == 0 && !with_center)
          continue;
        if (!cmp(val, I(x + u, y + v)))
          return false;
      }
    return true;
  }

  //! ImageProcessing/Extrema.hpp:63-75.
  template <typename Compare>
  inline bool local_scale_space_extremum(int x, int y, int s, int o,
                                         const Pyramid<Image>& I, Compare cmp)
  {
    const float val = I(s, o)(x, y);
    return compare_with_neighborhood3(val, x, y, I(s - 1, o), true, cmp) &&
           compare_with_neighborhood3(val, x, y, I(s, o), false, cmp) &&
           compare_with_neighborhood3(val, x, y, I(s + 1, o), true, cmp);
  }

  // ------------------------------------------------------------------------ //
  // The classifier of the reference's DO_SARA_USE_HALIDE build:
  // scale_space_dog_extremum_map() (ImageProcessing/LocalExtremum.cpp:23-37)
  // calls the AOT function shakti_scale_space_dog_extremum_32f_cpu, generated
  // from v2::LocalScaleSpaceExtremum
  // (Shakti/Halide/Generators/LocalExtremumGeneratorsV2.cpp:103-220) =
  // is_dog_extremum(prev, curr, next, ...) of
  // Shakti/Halide/Components/DoGExtremum.hpp:59-78 on repeat_edge() inputs:
  //   * EVERY pixel is classified, neighbours outside the image replicate the
  //     border (no img_padding_sz);
  //   * extremum = value equals the maximum / minimum of the 3 x 3 x 3 block
  //     (LocalExtremum.hpp:42-52), i.e. non-strict like the default build;
  //     a flat block is a maximum (the select tries is_max first);
  //   * contrast: abs(curr) > 0.8f * extremum_thres - STRICT, where the
  //     default build keeps abs >= 0.8 thres (RefineExtremum.cpp:426);
  //   * on_edge (DoGExtremum.hpp:29-36) through the Halide hessian
  //     (Components/Differential.hpp:33-49): dxx = in(x+1) + in(x-1) - 2 in(x)
  //     (another operation order than Differential.hpp of Sara), and
  //     dxy = (in(x+1,y+1) - in(x-1,y-1) - in(x+1,y-1) + in(x-1,y-1)) / 4 -
  //     the second term reads (x-1, y-1) where the formula wants (x-1, y+1);
  //     restated as written.  pow(., 2) with a constant integer exponent is
  //     expanded by Halide into a product (IROperator.cpp,
  //     raise_to_integer_power).
  // PARITY UNPINNED: Halide cannot be built here, and whether its x86 code
  // generator contracts a*b+c is not knowable from the sources; plain IEEE
  // operations in the written order are assumed.
  // ------------------------------------------------------------------------ //
  inline float clamped(const Image& I, int x, int y)
  {
    x = x < 0 ? 0 : (x > I.w - 1 ? I.w - 1 : x);
    y = y < 0 ? 0 : (y > I.h - 1 ? I.h - 1 : y);
    return I(x, y);
  }

  inline bool halide_on_edge(const Image& in, int x, int y, float edge_ratio)
  {
    const float c = clamped(in, x, y);
    const float dxx = clamped(in, x + 1, y) + clamped(in, x - 1, y) - 2 * c;
    const float dyy = clamped(in, x, y + 1) + clamped(in, x, y - 1) - 2 * c;
    const float dxy = (clamped(in, x + 1, y + 1) - clamped(in, x - 1, y - 1) -
                       clamped(in, x + 1, y - 1) + clamped(in, x - 1, y - 1)) /
                      4;
    const float tr = dxx + dyy;
    const float det = dxx * dyy - dxy * dxy;
    return (tr * tr) * edge_ratio >=
           ((1 + edge_ratio) * (1 + edge_ratio)) * std::abs(det);
  }

  //! is_dog_extremum(prev, curr, next, edge_ratio, extremum_thres, x, y) -> int8.
  inline int halide_is_dog_extremum(const Image& prev, const Image& curr,
                                    const Image& next, int x, int y,
                                    float edge_ratio, float extremum_thres)
  {
    const float v = curr(x, y);
    float mx = v, mn = v;
    for (int dv = -1; dv <= 1; ++dv)
      for (int du = -1; du <= 1; ++du)
      {
        const float a = clamped(prev, x + du, y + dv);
        const float b = clamped(curr, x + du, y + dv);
        const float c = clamped(next, x + du, y + dv);
        mx = std::max(mx, std::max(a, std::max(b, c)));
        mn = std::min(mn, std::min(a, std::min(b, c)));
      }
    const bool is_max = mx == v, is_min = mn == v;
    const bool is_strong = std::abs(v) > 0.8f * extremum_thres;
    const bool is_not_on_edge = !halide_on_edge(curr, x, y, edge_ratio);
    if (is_max && is_strong && is_not_on_edge)
      return 1;
    if (is_min && is_strong && is_not_on_edge)
      return -1;
    return 0;
  }

  //! FeatureDetectors/RefineExtremum.cpp:24-30.
  inline bool on_edge(const Image& I, int x, int y, float edge_ratio)
  {
    float hxx, hxy, hyy;
    hessian_at(I, x, y, hxx, hxy, hyy);
    const float tr = hxx + hyy;
    const float det = hxx * hyy - hxy * hxy;
    return (tr * tr) * edge_ratio >=
           ((edge_ratio + 1.f) * (edge_ratio + 1.f)) * std::abs(det);
  }

  //! ImageProcessing/GaussianPyramid.hpp:183-197.
  inline void gradient3(const Pyramid<Image>& I, int x, int y, int s, int o,
                        float d[3])
  {
    if (x < 1 || x >= I(s, o).w - 1 || y < 1 || y >= I(s, o).h - 1 || s < 1 ||
        s >= int(I.octaves[o].size()) - 1)
      throw std::out_of_range{"Computing gradient out of image range!"};
    d[0] = (I(s, o)(x + 1, y) - I(s, o)(x - 1, y)) / 2.f;
    d[1] = (I(s, o)(x, y + 1) - I(s, o)(x, y - 1)) / 2.f;
    d[2] = (I(s + 1, o)(x, y) - I(s - 1, o)(x, y)) / 2.f;
  }

  //! ImageProcessing/GaussianPyramid.hpp:200-233.  H is symmetric 3x3,
  //! order (x, y, s).
  inline void hessian3(const Pyramid<Image>& I, int x, int y, int s, int o,
                       float H[3][3])
  {
    if (x < 1 || x >= I(s, o).w - 1 || y < 1 || y >= I(s, o).h - 1 || s < 1 ||
        s >= int(I.octaves[o].size()) - 1)
      throw std::out_of_range{"Computing Hessian matrix out of image range!"};
    const Image& c = I(s, o);
    const Image& n = I(s + 1, o);
    const Image& p = I(s - 1, o);
    H[0][0] = c(x + 1, y) - 2.f * c(x, y) + c(x - 1, y);
    H[1][1] = c(x, y + 1) - 2.f * c(x, y) + c(x, y - 1);
    H[2][2] = n(x, y) - 2.f * c(x, y) + p(x, y);
    H[0][1] = H[1][0] =
        (c(x + 1, y + 1) - c(x - 1, y + 1) - c(x + 1, y - 1) + c(x - 1, y - 1)) /
        4.f;
    H[0][2] = H[2][0] =
        (n(x + 1, y) - n(x - 1, y) - p(x + 1, y) + p(x - 1, y)) / 4.f;
    H[1][2] = H[2][1] =
        (n(x, y + 1) - n(x, y - 1) - p(x, y + 1) + p(x, y - 1)) / 4.f;
  }

  // ------------------------------------------------------------------------ //
  // Definiteness of the 3x3 Hessian (RefineExtremum.cpp:74-77):
  //   SelfAdjointEigenSolver<Matrix3f> solver(D_second);
  //   lambda = solver.eigenvalues();
  //   if ((lambda * float(type)).maxCoeff() >= 0) -> do not refine
  // The constructor runs compute() (the iterative solver, not computeDirect())
  // in float.  Eigen (>= 3.4, CMakeLists.txt:94-96; the CI's distro package is
  // 3.4.0) is absent from this image, so its PUBLISHED algorithm is restated
  // below from Eigen 3.4.0's SelfAdjointEigenSolver.h / Tridiagonalization.h /
  // Jacobi.h / MathFunctions.h, operation by operation, in float:
  //   1. lower triangle copied, divided by scale = max |coefficient| (1 if 0);
  //   2. tridiagonalization_inplace_selector<Matrix3f, 3, false>::run: one
  //      closed-form Householder step;
  //   3. computeFromTridiagonal_impl: deflation test, implicit symmetric QR
  //      step with Wilkinson shift (tridiagonal_qr_step, makeGivens), at most
  //      30 * n iterations;
  //   4. eigenvalues * scale, sorted ascending.
  // kDefEigen33 differs only in the deflation test (Eigen 3.3.x:
  // |e| <= 2 eps (|d_i| + |d_i+1|)), kept to measure how much the decision
  // depends on the solver's version.  kDefSylvesterDouble is round 1's
  // stand-in (exact signs of the leading minors in double).
  // PARITY: still unpinned against a real Eigen build (none here), but the
  // three rules are compared site by site in tests/test_oracle_definiteness.py
  // and the counts are in DESIGN.md section 2.
  // ------------------------------------------------------------------------ //
  enum
  {
    kDefEigen34 = 0,         // default: the reference's build
    kDefEigen33 = 1,
    kDefSylvesterDouble = 2
  };
  inline int& definiteness_rule()
  {
    static int rule = kDefEigen34;
    return rule;
  }

  //! numext::hypot -> positive_real_hypot(abs(x), abs(y)), MathFunctions.h.
  inline float eigen_hypot(float x, float y)
  {
    x = std::abs(x);
    y = std::abs(y);
    if (std::isinf(x) || std::isinf(y))
      return std::numeric_limits<float>::infinity();
    if (std::isnan(x) || std::isnan(y))
      return std::numeric_limits<float>::quiet_NaN();
    const float p = std::max(x, y);
    if (p == 0.f)
      return 0.f;
    const float qp = std::min(y, x) / p;
    return p * std::sqrt(1.f + qp * qp);
  }

  //! JacobiRotation<float>::makeGivens(p, q) (real case), Jacobi.h.
  inline void eigen_make_givens(float p, float q, float& c, float& s)
  {
    if (q == 0.f)
    {
      c = p < 0.f ? -1.f : 1.f;
      s = 0.f;
    }
    else if (p == 0.f)
    {
      c = 0.f;
      s = q < 0.f ? 1.f : -1.f;
    }
    else if (std::abs(p) > std::abs(q))
    {
      const float t = q / p;
      float u = std::sqrt(1.f + t * t);
      if (p < 0.f)
        u = -u;
      c = 1.f / u;
      s = -t * c;
    }
    else
    {
      const float t = p / q;
      float u = std::sqrt(1.f + t * t);
      if (q < 0.f)
        u = -u;
      s = -1.f / u;
      c = -t * s;
    }
  }

  //! tridiagonal_qr_step (eigenvalues only), SelfAdjointEigenSolver.h.
  inline void eigen_tridiagonal_qr_step(float* diag, float* subdiag, int start,
                                        int end)
  {
    const float td = (diag[end - 1] - diag[end]) * 0.5f;
    const float e = subdiag[end - 1];
    float mu = diag[end];
    if (td == 0.f)
      mu -= std::abs(e);
    else if (e != 0.f)
    {
      const float e2 = e * e;
      const float h = eigen_hypot(td, e);
      if (e2 == 0.f)
        mu -= e / ((td + (td > 0.f ? h : -h)) / e);
      else
        mu -= e2 / (td + (td > 0.f ? h : -h));
    }
    float x = diag[start] - mu;
    float z = subdiag[start];
    for (int k = start; k < end && z != 0.f; ++k)
    {
      float c, s;
      eigen_make_givens(x, z, c, s);
      const float sdk = s * diag[k] + c * subdiag[k];
      const float dkp1 = s * subdiag[k] + c * diag[k + 1];
      diag[k] = c * (c * diag[k] - s * subdiag[k]) -
                s * (c * subdiag[k] - s * diag[k + 1]);
      diag[k + 1] = s * sdk + c * dkp1;
      subdiag[k] = c * sdk - s * dkp1;
      if (k > start)
        subdiag[k - 1] = c * subdiag[k - 1] - s * z;
      x = subdiag[k];
      if (k < end - 1)
      {
        z = -s * subdiag[k + 1];
        subdiag[k + 1] = c * subdiag[k + 1];
      }
    }
  }

  //! SelfAdjointEigenSolver<Matrix3f>(H).eigenvalues(); returns false when the
  //! iteration limit is hit (info() == NoConvergence; the reference does not
  //! check it and uses the unsorted values).
  inline bool eigen_selfadjoint_eigenvalues3(const float H[3][3], float lambda[3],
                                             int rule = kDefEigen34)
  {
    // lower triangle, mapped to [-1, 1]
    float m00 = H[0][0], m10 = H[1][0], m11 = H[1][1], m20 = H[2][0],
          m21 = H[2][1], m22 = H[2][2];
    float scale = std::max(
        std::max(std::max(std::abs(m00), std::abs(m10)),
                 std::max(std::abs(m11), std::abs(m20))),
        std::max(std::abs(m21), std::abs(m22)));
    if (scale == 0.f)
      scale = 1.f;
    m00 /= scale;
    m10 /= scale;
    m11 /= scale;
    m20 /= scale;
    m21 /= scale;
    m22 /= scale;

    // tridiagonalization, 3x3 real specialisation
    float diag[3], subdiag[2];
    const float tol = std::numeric_limits<float>::min();
    diag[0] = m00;
    const float v1norm2 = m20 * m20;
    if (v1norm2 <= tol)
    {
      diag[1] = m11;
      diag[2] = m22;
      subdiag[0] = m10;
      subdiag[1] = m21;
    }
    else
    {
      const float beta = std::sqrt(m10 * m10 + v1norm2);
      const float inv_beta = 1.f / beta;
      const float m01 = m10 * inv_beta;
      const float m02 = m20 * inv_beta;
      const float q = 2.f * m01 * m21 + m02 * (m22 - m11);
      diag[1] = m11 + m02 * q;
      diag[2] = m22 - m02 * q;
      subdiag[0] = beta;
      subdiag[1] = m21 - m01 * q;
    }

    // computeFromTridiagonal_impl
    const int n = 3, max_iterations = 30;
    int end = n - 1, start = 0, iter = 0;
    const float consider_as_zero = std::numeric_limits<float>::min();
    const float eps = std::numeric_limits<float>::epsilon();
    const float precision_inv = 1.f / eps;
    while (end > 0)
    {
      for (int i = start; i < end; ++i)
      {
        if (rule == kDefEigen33)
        {
          // isMuchSmallerThan(|e|, |d_i| + |d_i+1|, 2 eps) || |e| <= min
          if (std::abs(subdiag[i]) <=
                  (std::abs(diag[i]) + std::abs(diag[i + 1])) * (2.f * eps) ||
              std::abs(subdiag[i]) <= consider_as_zero)
            subdiag[i] = 0.f;
        }
        else if (std::abs(subdiag[i]) < consider_as_zero)
          subdiag[i] = 0.f;
        else
        {
          // |e| <= eps * sqrt(|d_i| + |d_i+1|), scaled against underflow
          const float scaled_subdiag = precision_inv * subdiag[i];
          if (scaled_subdiag * scaled_subdiag <=
              (std::abs(diag[i]) + std::abs(diag[i + 1])))
            subdiag[i] = 0.f;
        }
      }
      while (end > 0 && subdiag[end - 1] == 0.f)
        end--;
      if (end <= 0)
        break;
      iter++;
      if (iter > max_iterations * n)
        break;
      start = end - 1;
      while (start > 0 && subdiag[start - 1] != 0.f)
        start--;
      eigen_tridiagonal_qr_step(diag, subdiag, start, end);
    }
    const bool converged = iter <= max_iterations * n;
    if (converged)
      for (int i = 0; i < n - 1; ++i)
      {
        int k = i;
        for (int j = i + 1; j < n; ++j)
          if (diag[j] < diag[k])
            k = j;
        if (k > i)
          std::swap(diag[i], diag[k]);
      }
    for (int i = 0; i < 3; ++i)
      lambda[i] = diag[i] * scale;
    return converged;
  }

  //! Round 1's stand-in: +1 if all eigenvalues > 0, -1 if all < 0, 0 otherwise,
  //! by Sylvester's criterion in double on the float entries.
  inline int definiteness3_sylvester(const float Hf[3][3])
  {
    double H[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        H[i][j] = double(Hf[i][j]);
    const double m1 = H[0][0];
    const double m2 = H[0][0] * H[1][1] - H[0][1] * H[1][0];
    const double m3 = H[0][0] * (H[1][1] * H[2][2] - H[1][2] * H[2][1]) -
                      H[0][1] * (H[1][0] * H[2][2] - H[1][2] * H[2][0]) +
                      H[0][2] * (H[1][0] * H[2][1] - H[1][1] * H[2][0]);
    if (m1 > 0 && m2 > 0 && m3 > 0)
      return +1;
    if (m1 < 0 && m2 > 0 && m3 < 0)
      return -1;
    return 0;
  }

  //! (lambda * float(type)).maxCoeff() >= 0, RefineExtremum.cpp:76-77.
  inline bool not_definite_enough3(const float H[3][3], int type, int rule)
  {
    if (rule == kDefSylvesterDouble)
    {
      const int def = definiteness3_sylvester(H);
      return type > 0 ? (def != -1) : (def != +1);
    }
    float lambda[3];
    eigen_selfadjoint_eigenvalues3(H, lambda, rule);
    const float t = float(type);
    return std::max(std::max(lambda[0] * t, lambda[1] * t), lambda[2] * t) >= 0.f;
  }

  //! Site-by-site comparison of the three rules (filled by refine_extremum
  //! when enabled; tests/test_oracle_definiteness.py reads it).
  struct DefinitenessAudit
  {
    bool enabled = false;
    long long sites = 0;           // Hessians examined
    long long eigen34_vs_sylvester = 0;
    long long eigen34_vs_eigen33 = 0;
    long long not_converged = 0;
  };
  inline DefinitenessAudit& definiteness_audit()
  {
    static DefinitenessAudit a;
    return a;
  }

  //! Cofactor (i,j) of a 3x3 matrix the way Eigen's 3x3 inverse forms it:
  //! m(i1,j1)*m(i2,j2) - m(i1,j2)*m(i2,j1), i1=(i+1)%3, i2=(i+2)%3 (idem j).
  inline float cofactor3(const float m[3][3], int i, int j)
  {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
    const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1];
  }

  //! Eigen's fixed-size 3-term reduction is a binary split: a0 + (a1 + a2).
  inline float sum3(float a0, float a1, float a2) { return a0 + (a1 + a2); }

  //! h = -inverse(H) * g, standing in for Matrix3f::inverse() at
  //! RefineExtremum.cpp:85: Eigen 3.4 forms the 3x3 inverse from cofactors
  //! times 1/det in float (det = cofactor column 0 . matrix column 0), then a
  //! coefficient-wise product.  "Parity unpinned" at the ulp level.
  inline void newton_step3(const float H[3][3], const float g[3], float h[3])
  {
    const float c0 = cofactor3(H, 0, 0);
    const float c1 = cofactor3(H, 1, 0);
    const float c2 = cofactor3(H, 2, 0);
    const float det = sum3(c0 * H[0][0], c1 * H[1][0], c2 * H[2][0]);
    const float invdet = 1.f / det;
    float inv[3][3];
    inv[0][0] = c0 * invdet;
    inv[0][1] = c1 * invdet;
    inv[0][2] = c2 * invdet;
    inv[1][0] = cofactor3(H, 0, 1) * invdet;
    inv[1][1] = cofactor3(H, 1, 1) * invdet;
    inv[1][2] = cofactor3(H, 2, 1) * invdet;
    inv[2][0] = cofactor3(H, 0, 2) * invdet;
    inv[2][1] = cofactor3(H, 1, 2) * invdet;
    inv[2][2] = cofactor3(H, 2, 2) * invdet;
    for (int i = 0; i < 3; ++i)
      h[i] = sum3((-inv[i][0]) * g[0], (-inv[i][1]) * g[1], (-inv[i][2]) * g[2]);
  }

  //! FeatureDetectors/RefineExtremum.cpp:32-130.  `type` is the raw map value:
  //! 1 for maxima, 255 for minima (quirk Q2: the uint8 map stores -1 as 255,
  //! so minima are never refined and never take the `type == -1` branch).
  inline bool refine_extremum(const Pyramid<Image>& I, int x, int y, int s,
                              int o, int type, float pos[3], float& val,
                              int border_sz, int num_iter)
  {
    float D_prime[3] = {0, 0, 0};
    float D_second[3][3];
    float h[3] = {0, 0, 0};

    pos[0] = float(x);
    pos[1] = float(y);
    pos[2] = static_cast<float>(I.scale_relative_to_octave(s));

    int i = 0;
    for (; i < num_iter; ++i)
    {
      if (x < border_sz || x >= I(s, o).w - border_sz || y < border_sz ||
          y >= I(s, o).h - border_sz || s < 1 ||
          s >= int(I.octaves[o].size()) - 1)
        break;

      gradient3(I, x, y, s, o, D_prime);
      hessian3(I, x, y, s, o, D_second);

      // (lambda * float(type)).maxCoeff() >= 0
      const bool not_definite_enough =
          not_definite_enough3(D_second, type, definiteness_rule());
      if (definiteness_audit().enabled)
      {
        DefinitenessAudit& au = definiteness_audit();
        const bool e34 = not_definite_enough3(D_second, type, kDefEigen34);
        const bool e33 = not_definite_enough3(D_second, type, kDefEigen33);
        const bool syl = not_definite_enough3(D_second, type, kDefSylvesterDouble);
        float lam[3];
        const bool conv = eigen_selfadjoint_eigenvalues3(D_second, lam);
#pragma omp critical(sara_ref_def_audit)
        {
          au.sites += 1;
          au.eigen34_vs_sylvester += (e34 != syl);
          au.eigen34_vs_eigen33 += (e34 != e33);
          au.not_converged += !conv;
        }
      }
      if (not_definite_enough)
      {
        h[0] = h[1] = h[2] = 0.f;
        break;
      }

      newton_step3(D_second, D_prime, h);

      if (std::max(std::abs(h[0]), std::abs(h[1])) > 1.5f)
        return false;

      if (std::min(std::abs(h[0]), std::abs(h[1])) > 0.6f)
      {
        x += h[0] > 0 ? 1 : -1;
        y += h[1] > 0 ? 1 : -1;
        continue;
      }
      break;
    }

    pos[0] = float(x);
    pos[1] = float(y);
    pos[2] = static_cast<float>(I.scale_relative_to_octave(s));
    const float oldval = I(s, o)(x, y);
    const float newval =
        oldval +
        0.5f * sum3(D_prime[0] * h[0], D_prime[1] * h[1], D_prime[2] * h[2]);

    if ((type == 1 && oldval <= newval) || (type == -1 && oldval >= newval))
    {
      pos[0] += h[0];
      pos[1] += h[1];
      pos[2] *= std::pow(I.scale_geometric_factor, h[2]);
      val = newval;
    }
    return true;
  }

  //! One entry of the intermediate extremum list (raster order per (s,o)).
  struct ExtremumRecord
  {
    int x, y, s, o;
    int type;  // +1 max, -1 min
    OERegion region;
  };

  //! FeatureDetectors/RefineExtremum.cpp:363-521 (#else branch).
  inline std::vector<OERegion>
  local_scale_space_extrema(const Pyramid<Image>& I, int s, int o,
                            float extremum_thres, float edge_ratio_thres,
                            int img_padding_sz, int refine_iterations,
                            std::vector<ExtremumRecord>* records = nullptr)
  {
    const int w = I(s, o).w;
    const int h = I(s, o).h;
    const int wh = w * h;

    std::vector<std::uint8_t> map(size_t(wh), 0);
    const bool halide_map = (detector_mode() & kModeSignedExtremumType) != 0;

#pragma omp parallel for
    for (int xy = 0; xy < wh; ++xy)
    {
      const int y = xy / w;
      const int x = xy - y * w;
      if (halide_map)
      {
        // RefineExtremum.cpp:246-262: the whole map comes from the Halide
        // classifier (int8: -1 stays -1)
        map[xy] = static_cast<std::uint8_t>(static_cast<std::int8_t>(
            halide_is_dog_extremum(I(s - 1, o), I(s, o), I(s + 1, o), x, y,
                                   edge_ratio_thres, extremum_thres)));
        continue;
      }
      const bool in_domain = img_padding_sz <= x && x < w - img_padding_sz &&
                             img_padding_sz <= y && y < h - img_padding_sz;
      if (!in_domain)
        continue;

      int type = 0;
      if (local_scale_space_extremum(x, y, s, o, I, std::greater_equal<float>{}))
        type = 1;
      else if (local_scale_space_extremum(x, y, s, o, I,
                                          std::less_equal<float>{}))
        type = -1;
      else
        continue;

      if (std::abs(I(s, o)(x, y)) < 0.8f * extremum_thres)
        continue;
      if (on_edge(I(s, o), x, y, edge_ratio_thres))
        continue;

      map[xy] = static_cast<std::uint8_t>(type);  // -1 -> 255 (Q2)
    }
    const bool signed_type = (detector_mode() & kModeSignedExtremumType) != 0;

    std::vector<float> location_refined(static_cast<size_t>(wh) * 3, 0.f);
    std::vector<float> extremum_value(static_cast<size_t>(wh), 0.f);
#pragma omp parallel for
    for (int xy = 0; xy < wh; ++xy)
    {
      const int y = xy / w;
      const int x = xy - y * w;
      const std::uint8_t type = map[xy];
      if (type == 0)
        continue;
      float* pos = &location_refined[size_t(xy) * 3];
      float& val = extremum_value[xy];
      val = I(s, o)(x, y);
      refine_extremum(I, x, y, s, o,
                      signed_type ? int(static_cast<std::int8_t>(type)) : int(type),
                      pos, val, img_padding_sz, refine_iterations);
      bool scale_implausible = false;
      if (signed_type)
      {
        // RefineExtremum.cpp:307-318
        const float scale_approx =
            static_cast<float>(I.scale_relative_to_octave(s));
        const float ratio_max = 4.f, ratio_min = 1 / ratio_max;
        scale_implausible =
            !(ratio_min * scale_approx < pos[2] && pos[2] < ratio_max * scale_approx);
      }
      if (std::abs(val) < extremum_thres || scale_implausible)
        map[xy] = 0;
    }

    std::vector<OERegion> extrema;
    extrema.reserve(10000);
    for (int xy = 0; xy < wh; ++xy)
    {
      const int y = xy / w;
      const int x = xy - y * w;
      const std::uint8_t type = map[xy];
      if (type == 0)
        continue;
      const float* pos = &location_refined[size_t(xy) * 3];
      OERegion dog = make_oeregion(pos[0], pos[1], pos[2]);
      dog.extremum_value = extremum_value[xy];
      dog.extremum_type = (type == 1) ? 1 : -1;
      extrema.push_back(dog);
      if (records)
        records->push_back({x, y, s, o, type == 1 ? 1 : -1, dog});
    }
    return extrema;
  }

  // ------------------------------------------------------------------------ //
  // Dominant orientations.
  // ------------------------------------------------------------------------ //

  inline int int_round(double x) { return static_cast<int>(std::round(x)); }
  inline int int_round(float x) { return static_cast<int>(std::round(x)); }

  //! FeatureDescriptors/Orientation.hpp:91-135 with T=float.  The unqualified
  //! exp()/floor() bind to the double C functions, so the weight
  //! and the accumulation are evaluated in double and rounded per pixel.
  template <int N>
  inline void compute_orientation_histogram(float* hist, const Image2& grad,
                                            float x, float y, float s,
                                            float patch_truncation_factor = 3.f,
                                            float blur_factor = 1.5f)
  {
    for (int i = 0; i < N; ++i)
      hist[i] = 0.f;
    const int rounded_x = int_round(x);
    const int rounded_y = int_round(y);
    const float sigma = s * blur_factor;
    const int patch_radius = int_round(sigma * patch_truncation_factor);

    for (int v = -patch_radius; v <= patch_radius; ++v)
      for (int u = -patch_radius; u <= patch_radius; ++u)
      {
        if (rounded_x + u < 0 || rounded_x + u >= grad.w ||
            rounded_y + v < 0 || rounded_y + v >= grad.h)
          continue;
        const float* g = grad.at(rounded_x + u, rounded_y + v);
        const float mag = g[0];
        float ori = g[1];
        ori = ori < 0 ? ori + float(2. * M_PI) : ori;
        int bin_index = int(std::floor(double(ori / float(2 * M_PI) * N)));
        bin_index %= N;
        const double weight =
            std::exp(double(-(u * u + v * v) / (2.f * sigma * sigma)));
        hist[bin_index] =
            static_cast<float>(double(hist[bin_index]) + weight * double(mag));
      }
  }

  //! FeatureDescriptors/Orientation.hpp:143-163.
  template <int N>
  inline void lowe_smooth_histogram(float* hist, int num_iters = 6)
  {
    for (int iter = 0; iter < num_iters; ++iter)
    {
      const float first = hist[0];
      float prev = hist[N - 1];
      for (int i = 0; i < N - 1; ++i)
      {
        const float val = (prev + hist[i] + hist[i + 1]) / 3.f;
        prev = hist[i];
        hist[i] = val;
      }
      hist[N - 1] = (prev + hist[N - 1] + first) / 3.f;
    }
  }

  //! FeatureDescriptors/Orientation.hpp:173-186.
  template <int N>
  inline std::vector<int> find_peaks(const float* hist,
                                     float peak_ratio_thres = 0.8f)
  {
    float max = hist[0];
    for (int i = 1; i < N; ++i)
      max = std::max(max, hist[i]);
    std::vector<int> peaks;
    peaks.reserve(N);
    for (int i = 0; i < N; ++i)
      if (hist[i] >= peak_ratio_thres * max && hist[i] > hist[(i - 1 + N) % N] &&
          hist[i] > hist[(i + 1) % N])
        peaks.push_back(i);
    return peaks;
  }

  //! FeatureDescriptors/Orientation.hpp:190-212.
  template <int N>
  inline float refine_peak(const float* hist, int i)
  {
    const float y0 = hist[(i - 1 + N) % N];
    const float y1 = hist[i];
    const float y2 = hist[(i + 1) % N];
    const float fprime = (y2 - y0) / 2.f;
    const float fsecond = y0 - 2.f * y1 + y2;
    const float h = -fprime / fsecond;
    return float(i) + 0.5f + h;
  }

  //! FeatureDescriptors/Orientation.cpp:90-118.
  inline std::vector<float>
  dominant_orientations(const Image2& grad, float x, float y, float sigma,
                        float peak_ratio_thres = 0.8f,
                        float patch_truncation_factor = 3.f,
                        float blur_factor = 1.5f, float* hist_out = nullptr)
  {
    constexpr int O = 36;
    float hist[O];
    compute_orientation_histogram<O>(hist, grad, x, y, sigma,
                                     patch_truncation_factor, blur_factor);
    lowe_smooth_histogram<O>(hist);
    if (hist_out)
      std::memcpy(hist_out, hist, sizeof(hist));
    const auto peak_indices = find_peaks<O>(hist, peak_ratio_thres);
    std::vector<float> peaks(peak_indices.size());
    for (size_t i = 0; i != peaks.size(); ++i)
    {
      peaks[i] = refine_peak<O>(hist, peak_indices[i]);
      peaks[i] *= static_cast<float>(2 * M_PI) / O;
      if (peaks[i] > float(M_PI))
        peaks[i] -= 2.f * float(M_PI);
    }
    return peaks;
  }

  //! FeatureDescriptors/Orientation.cpp:120-166: expands the list, one copy of
  //! the extremum per peak, serial loop.
  inline void assign_dominant_orientations(const Pyramid<Image2>& pyramid,
                                           std::vector<OERegion>& extrema,
                                           std::vector<int>& so_pairs)
  {
    std::vector<OERegion> e2;
    std::vector<int> so2;
    e2.reserve(extrema.size() * 2);
    so2.reserve(extrema.size() * 4);
    for (size_t i = 0; i != extrema.size(); ++i)
    {
      const int s_index = so_pairs[2 * i], o_index = so_pairs[2 * i + 1];
      const float s =
          static_cast<float>(pyramid.scale_relative_to_octave(s_index));
      const auto orientations = dominant_orientations(
          pyramid(s_index, o_index), extrema[i].coords[0], extrema[i].coords[1],
          s);
      for (size_t k = 0; k != orientations.size(); ++k)
      {
        so2.push_back(s_index);
        so2.push_back(o_index);
        e2.push_back(extrema[i]);
        e2.back().orientation = orientations[k];
      }
    }
    e2.swap(extrema);
    so2.swap(so_pairs);
  }

  // ------------------------------------------------------------------------ //
  // SIFT descriptor, N=4, O=8.
  // ------------------------------------------------------------------------ //

  //! FeatureDescriptors/SIFT.hpp:204-238: std::modf truncates toward zero
  //! (quirk Q13).
  inline void sift_accumulate(float* h, float px, float py, float ori,
                              float weight, float mag)
  {
    constexpr int N = 4, O = 8;
    float xif, yif, oriif;
    const float xfrac = std::modf(px, &xif);
    const float yfrac = std::modf(py, &yif);
    const float orifrac = std::modf(ori, &oriif);
    const int xi = int(xif), yi = int(yif), orii = int(oriif);
    for (int dy = 0; dy < 2; ++dy)
    {
      const int y = yi + dy;
      if (y < 0 || y >= N)
        continue;
      const float wy = (dy == 0) ? 1 - yfrac : yfrac;
      for (int dx = 0; dx < 2; ++dx)
      {
        const int x = xi + dx;
        if (x < 0 || x >= N)
          continue;
        const float wx = (dx == 0) ? 1 - xfrac : xfrac;
        for (int dori = 0; dori < 2; ++dori)
        {
          const int o = (orii + dori) % O;
          const float wo = (dori == 0) ? 1 - orifrac : orifrac;
          h[N * O * y + x * O + o] += wy * wx * wo * weight * mag;
        }
      }
    }
  }

  //! Matrix<float,128,1>::squaredNorm() as Eigen 3.4 evaluates it in the
  //! reference's default build (x86-64 baseline => SSE2, Packet4f; the
  //! -march=native line of cmake/sara_configure_cxx_compiler.cmake:37 is
  //! commented out): cwiseAbs2().sum() goes through redux_impl<...,
  //! LinearVectorizedTraversal, NoUnrolling> (Redux.h; 128 coefficients exceed
  //! the unrolling limit), i.e. two Packet4f accumulators over the even / odd
  //! packets, initialised with packets 0 and 1, then acc0 + acc1, then
  //! predux<Packet4f> (SSE/PacketMath.h): (a0 + a2) + (a1 + a3).
  inline float eigen_squared_norm128(const float* h)
  {
    float acc0[4], acc1[4];
    for (int j = 0; j < 4; ++j)
    {
      acc0[j] = h[j] * h[j];
      acc1[j] = h[4 + j] * h[4 + j];
    }
    for (int i = 8; i < 128; i += 8)
      for (int j = 0; j < 4; ++j)
      {
        acc0[j] = acc0[j] + h[i + j] * h[i + j];
        acc1[j] = acc1[j] + h[i + 4 + j] * h[i + 4 + j];
      }
    float a[4];
    for (int j = 0; j < 4; ++j)
      a[j] = acc0[j] + acc1[j];
    return (a[0] + a[2]) + (a[1] + a[3]);
  }

  //! 0: Eigen's packet order (default, see above); 1: plain left-to-right sum
  //! (round 1's stand-in, kept to measure the difference:
  //! tests/test_oracle_definiteness.py).
  inline int& squared_norm_order()
  {
    static int order = 0;
    return order;
  }

  //! Eigen's MatrixBase::normalize() (Dot.h): z = squaredNorm(); if (z > 0)
  //! *this /= sqrt(z) (a true division per coefficient).
  inline void l2_normalize128(float* h)
  {
    float z = 0.f;
    if (squared_norm_order() == 0)
      z = eigen_squared_norm128(h);
    else
      for (int i = 0; i < 128; ++i)
        z += h[i] * h[i];
    if (z > 0.f)
    {
      const float n = std::sqrt(z);
      for (int i = 0; i < 128; ++i)
        h[i] /= n;
    }
  }

  //! FeatureDescriptors/RootSIFT.hpp:45-53: the descriptor of the base operator
  //! (already normalised and scaled to 0..255) divided by its L1 norm, then the
  //! square root of every bin.  The reference's two lines use Eigen 2's
  //! `.cwise()` and no longer compile (nothing includes the header), so this is
  //! a restatement of their intent - PARITY UNPINNED; left-to-right float sum
  //! for lpNorm<1>(); an all-zero descriptor is left as it is (the reference
  //! would divide 0 by 0).  The base descriptor has negative bins (modf() of a
  //! patch coordinate in (-1, 0) gives a negative fraction, SIFT.hpp:209-232,
  //! about one bin in ten), whose plain square root would be NaN: the root is
  //! taken of the magnitude and the sign kept, so that <root(a), root(a)> = 1
  //! still holds.
  inline void root_sift(float* h, int dim)
  {
    float l1 = 0.f;
    for (int i = 0; i < dim; ++i)
      l1 += std::abs(h[i]);
    if (!(l1 > 0.f))
      return;
    for (int i = 0; i < dim; ++i)
      h[i] = std::copysign(std::sqrt(std::abs(h[i]) / l1), h[i]);
  }

  //! FeatureDescriptors/SIFT.hpp:62-145 (+ normalize :241-252).  The
  //! unqualified sqrt/cos/sin bind to the double C functions; T's entries are
  //! rounded to float by Eigen's comma initialiser.
  inline void compute_sift_descriptor(float* h, float x, float y, float s,
                                      float theta, const Image2& grad,
                                      bool do_normalization = true,
                                      float bin_scale_unit_length = 3.f,
                                      float max_bin_value = 0.2f)
  {
    constexpr int N = 4, O = 8, Dim = 128;
    constexpr float pi = static_cast<float>(M_PI);
    const float lambda = bin_scale_unit_length;
    const float l = lambda * s;
    const double r = std::sqrt(double(2.f)) * double(l) * (N + 1) / double(2.f);

    float T00 = static_cast<float>(std::cos(double(theta)));
    float T01 = static_cast<float>(std::sin(double(theta)));
    float T10 = static_cast<float>(-std::sin(double(theta)));
    float T11 = static_cast<float>(std::cos(double(theta)));
    T00 /= l;
    T01 /= l;
    T10 /= l;
    T11 /= l;

    for (int i = 0; i < Dim; ++i)
      h[i] = 0.f;

    const int rounded_r = int_round(r);
    const int rounded_x = int_round(x);
    const int rounded_y = int_round(y);

    for (int v = -rounded_r; v <= rounded_r; ++v)
      for (int u = -rounded_r; u <= rounded_r; ++u)
      {
        float px = T00 * float(u) + T01 * float(v);
        float py = T10 * float(u) + T11 * float(v);

        if (rounded_x + u < 0 || rounded_x + u >= grad.w ||
            rounded_y + v < 0 || rounded_y + v >= grad.h)
          continue;

        constexpr float sigma = N * N * 0.25f;
        const float weight = std::exp(-(px * px + py * py) / (2.f * sigma));

        const float* g = grad.at(rounded_x + u, rounded_y + v);
        const float mag = g[0];
        float ori = g[1] - theta;
        ori = ori < 0.f ? ori + 2.f * pi : ori;
        ori *= static_cast<float>(O) / (2.f * pi);

        px += N / 2.f - 0.5f;
        py += N / 2.f - 0.5f;

        if (std::min(px, py) <= -1.f || std::max(px, py) >= static_cast<float>(N))
          continue;

        sift_accumulate(h, px, py, ori, weight, mag);
      }

    if (do_normalization)
    {
      l2_normalize128(h);
      for (int i = 0; i < Dim; ++i)
        h[i] = std::min(h[i], max_bin_value);
      l2_normalize128(h);
      for (int i = 0; i < Dim; ++i)
        h[i] = std::min(h[i] * 512.f, 255.f);
    }
  }

  // ------------------------------------------------------------------------ //
  // Drivers.
  // ------------------------------------------------------------------------ //

  struct StageTimes
  {
    double gaussian_pyramid_ms = 0, dog_pyramid_ms = 0, dog_extrema_ms = 0;
    double gradient_ms = 0, orientation_ms = 0, descriptors_ms = 0;
    double total_ms = 0;
  };

  struct Timer
  {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void restart() { t0 = std::chrono::steady_clock::now(); }
    double elapsed_ms() const
    {
      return std::chrono::duration<double, std::milli>(
                 std::chrono::steady_clock::now() - t0)
          .count();
    }
  };

  //! FeatureDetectors/DoG.hpp:72-90 + DoG.cpp:23-87.
  struct ComputeDoGExtrema
  {
    PyramidParams pyramid_params;
    float gauss_truncate = 4.f;
    float extremum_thres = 0.01f;
    float edge_ratio_thres = 10.f;
    int img_padding_sz = 1;
    int extremum_refinement_iter = 5;

    Pyramid<Image> gaussians;
    Pyramid<Image> diff_of_gaussians;
    std::vector<ExtremumRecord> records;

    ComputeDoGExtrema(const PyramidParams& p = PyramidParams(),
                      float gauss_truncate_ = 4.f, float extremum_thres_ = 0.01f,
                      float edge_ratio_thres_ = 10.f, int img_padding_sz_ = 1,
                      int extremum_refinement_iter_ = 5)
      : pyramid_params(p)
      , gauss_truncate(gauss_truncate_)
      , extremum_thres(extremum_thres_)
      , edge_ratio_thres(edge_ratio_thres_)
      , img_padding_sz(img_padding_sz_)
      , extremum_refinement_iter(extremum_refinement_iter_)
    {
      if (pyramid_params.scale_count_per_octave < 4)
        throw std::runtime_error{
            "Error: The extraction of DoG extrema needs (1 + 3) = 4 scales per "
            "octave at the very minimum!"};
    }

    //! so_pairs: flat (s, o) pairs.  stop_after: 0 = everything, 1 = stop
    //! after the Gaussian pyramid, 2 = after the DoG pyramid.
    std::vector<OERegion> operator()(const Image& image,
                                     std::vector<int>* so_pairs,
                                     StageTimes* times = nullptr,
                                     int stop_after = 0)
    {
      Timer timer;
      gaussians = gaussian_pyramid(image, pyramid_params, gauss_truncate);
      if (times)
        times->gaussian_pyramid_ms = timer.elapsed_ms();
      std::vector<OERegion> extrema;
      if (so_pairs)
        so_pairs->clear();
      records.clear();
      if (stop_after == 1)
        return extrema;

      timer.restart();
      diff_of_gaussians = difference_of_gaussians_pyramid(gaussians);
      if (times)
        times->dog_pyramid_ms = timer.elapsed_ms();
      if (stop_after == 2)
        return extrema;

      timer.restart();
      const auto& D = diff_of_gaussians;
      extrema.reserve(10000);
      for (int o = 0; o < D.octave_count(); ++o)
        for (int s = 1; s < D.scale_count_per_octave() - 1; ++s)
        {
          const auto e = local_scale_space_extrema(
              D, s, o, extremum_thres, edge_ratio_thres, img_padding_sz,
              extremum_refinement_iter, &records);
          extrema.insert(extrema.end(), e.begin(), e.end());
          if (so_pairs)
            for (size_t i = 0; i != e.size(); ++i)
            {
              so_pairs->push_back(s);
              so_pairs->push_back(o);
            }
        }
      if (times)
        times->dog_extrema_ms = timer.elapsed_ms();
      return extrema;
    }
  };

  struct SiftResult
  {
    ComputeDoGExtrema dog;
    Pyramid<Image2> nabla_G;
    std::vector<OERegion> extrema;     // before orientation assignment
    std::vector<int> extrema_so;       // flat (s,o)
    std::vector<OERegion> features;    // final, rescaled
    std::vector<int> so_pairs;         // flat (s,o) of the final list
    std::vector<float> descriptors;    // N x 128 row-major
    StageTimes times;
  };

  //! FeatureDetectors/SIFT.cpp:27-108, including the argument shift Q1:
  //! extremum_refinement_iter lands in img_padding_sz.
  //! stop_after: 0 full, 1 pyramid, 2 DoG, 3 extrema, 4 gradients,
  //! 5 orientations.
  inline void compute_sift_keypoints(SiftResult& R, const Image& image,
                                     const PyramidParams& pyramid_params,
                                     float gauss_truncate = 4.f,
                                     float extremum_thres = 0.01f,
                                     float edge_ratio_thres = 10.f,
                                     int extremum_refinement_iter = 5,
                                     bool parallel = false, int stop_after = 0)
  {
    Timer timer;
    R.times = StageTimes{};
    R.dog = ComputeDoGExtrema{pyramid_params, gauss_truncate, extremum_thres,
                              edge_ratio_thres, extremum_refinement_iter};
    R.features.clear();
    R.so_pairs.clear();
    R.descriptors.clear();
    R.extrema = R.dog(image, &R.extrema_so, &R.times,
                      (stop_after >= 1 && stop_after <= 2) ? stop_after : 0);
    double elapsed = timer.elapsed_ms();
    if (stop_after >= 1 && stop_after <= 3)
    {
      R.times.total_ms = elapsed;
      return;
    }

    timer.restart();
    R.nabla_G = gradient_polar_coordinates(R.dog.gaussians);
    R.times.gradient_ms = timer.elapsed_ms();
    elapsed += R.times.gradient_ms;
    if (stop_after == 4)
    {
      R.times.total_ms = elapsed;
      return;
    }

    timer.restart();
    R.features = R.extrema;
    R.so_pairs = R.extrema_so;
    assign_dominant_orientations(R.nabla_G, R.features, R.so_pairs);
    R.times.orientation_ms = timer.elapsed_ms();
    elapsed += R.times.orientation_ms;
    if (stop_after == 5)
    {
      R.times.total_ms = elapsed;
      return;
    }

#ifdef _OPENMP
    if (parallel)
      omp_set_num_threads(omp_get_max_threads());
#endif

    timer.restart();
    const int n = int(R.features.size());
    R.descriptors.assign(size_t(n) * 128, 0.f);
    if (parallel)
    {
#pragma omp parallel for
      for (int i = 0; i < n; ++i)
      {
        const auto& f = R.features[i];
        compute_sift_descriptor(&R.descriptors[size_t(i) * 128], f.coords[0],
                                f.coords[1], oeregion_scale(f), f.orientation,
                                R.nabla_G(R.so_pairs[2 * i], R.so_pairs[2 * i + 1]));
      }
    }
    else
    {
      for (int i = 0; i < n; ++i)
      {
        const auto& f = R.features[i];
        compute_sift_descriptor(&R.descriptors[size_t(i) * 128], f.coords[0],
                                f.coords[1], oeregion_scale(f), f.orientation,
                                R.nabla_G(R.so_pairs[2 * i], R.so_pairs[2 * i + 1]));
      }
    }
    R.times.descriptors_ms = timer.elapsed_ms();

    for (int i = 0; i < n; ++i)
    {
      const float factor = R.nabla_G.oct_scaling_factors[R.so_pairs[2 * i + 1]];
      R.features[i].coords[0] *= factor;
      R.features[i].coords[1] *= factor;
      const float f2 = factor * factor;
      for (int j = 0; j < 4; ++j)
        R.features[i].shape_matrix[j] /= f2;
    }
    elapsed += R.times.descriptors_ms;
    R.times.total_ms = elapsed;
  }


  // ======================================================================== //
  // Descriptor matching.
  //
  // Reference: AnnMatcher (both constructors), compute_matches and
  // append_nearest_neighbors, FeatureMatching/AnnMatcher.cpp:59-268, called by
  // match(), SfM/Helpers/KeypointMatching.cpp:19-25; KeyProximity,
  // FeatureMatching/KeyProximity.cpp:17-30 with SquaredRefDistance,
  // Geometry/Tools/Metric.hpp:47-50.  The reference queries FLANN kd-trees
  // (un-vendored search heuristics: 8 randomised trees, 32 checks), whose
  // answers are *approximations* of the exact nearest neighbours and depend on
  // FLANN's random seeds.  This restatement replaces the tree queries by
  // exhaustive ones - what FLANN converges to with unlimited checks - and
  // keeps everything else:
  //   * FLANN's squared L2 distance in its own summation order
  //     (flann/algorithms/dist.h:150-178: groups of four, float accumulator);
  //   * knnSearch(3) -> top1_score = d[top1] / d[top1 + 1]
  //     (AnnMatcher.cpp:126-130), top1 = 1 when self-matching (rank 0 is taken
  //     to be the query itself, :124-125);
  //   * squared ratio > 1 (the DEFAULT, sift_ratio_thres = 1.2f,
  //     AnnMatcher.hpp:36-46): the adaptive radius search of :133-138 -
  //     radius = d[top1] * thres^2, FLANN's RadiusResultSet keeps dist < radius
  //     (flann/util/result_set.h, strict), sorted by (dist, index); ranks
  //     top1 .. K - 1 are emitted with score d[rank] / d[top1] until one
  //     exceeds the threshold (:141-154); a zero best distance gives radius 0,
  //     hence K = 0 and NO match at all - kept;
  //   * squared ratio <= 1: K = 1, i.e. only the best neighbour - and nothing
  //     at all when self-matching (the loop starts at rank 1) - kept;
  //   * the boundary cases :80-120 (one candidate; two keys self-matching);
  //   * KeyProximity on self-matches (:160-161);
  //   * both directions, sort by (x, y, score), unique, final sort by score.
  // Match::operator== compares the two OERegions BY VALUE (Match.hpp:161-164
  // with Feature.hpp:140-146), so std::unique also drops consecutive matches
  // between distinct keypoints with equal (coords, shape, orientation, type);
  // after the (x_index, y_index) sort that can only merge equal-valued
  // duplicates of a keypoint.  The two-set entry below has no features and
  // compares indices; the self-matching entry compares values as the
  // reference does.
  // PINNED (round 4) for the exact part: the reference's vendored FLANN is
  // header-only and compiles here (oracle/Makefile `_ref`,
  // oracle/flann_ref_harness.cpp calls it as AnnMatcher.cpp does).  With
  // flann::LinearIndexParams - FLANN's exact index - knnSearch(3),
  // radiusSearch(d_best * 1.44) and the whole compute_matches() list (ratios
  // 0.6 / 0.8 / 1.0 / 1.2, and the self-matching constructor) equal this
  // restatement BIT FOR BIT on the SIFT descriptors of two overlapping 1080p
  // frames and of the sunflower crop: neighbours, float distances, tie order,
  // strict radius (tests/test_oracle_flann_pins.py, fixtures
  // tests/golden/flann_pins.npz).  What stays different BY DESIGN is the index:
  // the reference asks KDTreeIndexParams(8) with 32 checks (AnnMatcher.cpp:227),
  // an approximate search; the same test records how its lists relate to the
  // exact ones (ratio 0.6: 4138 of 4140 matches common; default ratio 1.2: the
  // kd-trees find 2790 matches, the exact radius search 7281).  Also pinned by
  // the reference's only matcher test (test_featurematching_matching.cpp:29-62)
  // and test_featurematching_key_proximity.cpp.
  // ======================================================================== //
  struct Match
  {
    int32_t x_index;   //!< index in the first key set
    int32_t y_index;   //!< index in the second key set
    float score;       //!< squared-distance ratio
    int32_t rank;      //!< 1 = best neighbour, 2.. = further ones in the radius
    int32_t direction; //!< 0 = SourceToTarget, 1 = TargetToSource
  };

  //! flann::L2<float>::operator() (dist.h:150-178) without the early exit.
  inline float flann_l2(const float* a, const float* b, int size)
  {
    float result = 0.f;
    int i = 0;
    for (; i + 3 < size; i += 4)
    {
      const float d0 = a[i] - b[i], d1 = a[i + 1] - b[i + 1];
      const float d2 = a[i + 2] - b[i + 2], d3 = a[i + 3] - b[i + 3];
      result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    for (; i < size; ++i)
    {
      const float d0 = a[i] - b[i];
      result += d0 * d0;
    }
    return result;
  }

  //! The part of an OERegion the matcher looks at (Feature.hpp:155-177).
  struct MatchFeature
  {
    float x, y;
    float m00, m10, m01, m11;  // shape matrix, column-major
    float orientation;
    int type;
    bool operator==(const MatchFeature& o) const  // Feature.hpp:140-146
    {
      return x == o.x && y == o.y && m00 == o.m00 && m10 == o.m10 &&
             m01 == o.m01 && m11 == o.m11 && orientation == o.orientation &&
             type == o.type;
    }
  };

  //! KeyProximity::operator() (KeyProximity.cpp:17-30).
  struct KeyProximity
  {
    float squared_metric_dist = 0.5f * 0.5f;
    float squared_dist_thres = 10.f * 10.f;
    KeyProximity() = default;
    KeyProximity(float metric_dist_thres, float pixel_dist_thres)
      : squared_metric_dist(metric_dist_thres * metric_dist_thres)
      , squared_dist_thres(pixel_dist_thres * pixel_dist_thres)
    {
    }
    //! (b - a).dot(M (b - a)), Metric.hpp:47-50 (2 x 2 coefficient products).
    static float metric(const MatchFeature& f, float vx, float vy)
    {
      const float r0 = f.m00 * vx + f.m01 * vy;
      const float r1 = f.m10 * vx + f.m11 * vy;
      return vx * r0 + vy * r1;
    }
    bool operator()(const MatchFeature& f1, const MatchFeature& f2) const
    {
      const float sd1 = metric(f1, f2.x - f1.x, f2.y - f1.y);
      const float sd2 = metric(f2, f2.x - f1.x, f2.y - f1.y);
      const float dx = f1.x - f2.x, dy = f1.y - f2.y;
      const float squared_pixel_dist = dx * dx + dy * dy;
      return squared_pixel_dist < squared_dist_thres ||
             sd1 < squared_metric_dist || sd2 < squared_metric_dist;
    }
  };

  //! What tree.knnSearch / tree.radiusSearch return for an exact index
  //! (flann::LinearIndexParams: every point is examined, the result set orders
  //! by (distance, index) - flann/util/result_set.h - and RadiusResultSet keeps
  //! dist < radius, strict): all nt candidates of one query in that order.
  //! Pinned against the reference's vendored FLANN by
  //! tests/test_oracle_flann_pins.py (oracle/flann_ref_harness.cpp).
  inline void exhaustive_neighbours(const float* query, const float* t, int nt,
                                    int dim, std::vector<std::pair<float, int>>& nn)
  {
    nn.resize(static_cast<size_t>(nt));
    for (int j = 0; j < nt; ++j)
      nn[size_t(j)] = {flann_l2(query, t + size_t(j) * dim, dim), j};
    std::sort(nn.begin(), nn.end());
  }

  //! append_nearest_neighbors for every row of `q` against `t`
  //! (AnnMatcher.cpp:59-170).  fq / ft: features (self-matching only).
  inline void append_matches(const float* q, int nq, const float* t, int nt,
                             int dim, float squared_ratio_thres, int direction,
                             bool self_matching, const KeyProximity& is_redundant,
                             const MatchFeature* fq, const MatchFeature* ft,
                             std::vector<Match>& matches)
  {
    if (nt == 0)
      return;
    auto push = [&](int i1, int i2, float score, int rank) {
      matches.push_back(direction == 0 ? Match{i1, i2, score, rank, direction}
                                       : Match{i2, i1, score, rank, direction});
    };
    std::vector<std::pair<float, int>> nn(static_cast<size_t>(nt));
    for (int i = 0; i < nq; ++i)
    {
      if (nt == 1 && !self_matching)
      {
        // AnnMatcher.cpp:87-101: a single candidate gets score 1.
        if (1.f < squared_ratio_thres)
          push(i, 0, 1.f, 1);
        continue;
      }
      // FLANN's result sets order by (distance, index)
      exhaustive_neighbours(q + size_t(i) * dim, t, nt, dim, nn);
      if (nt == 2 && self_matching)
      {
        // AnnMatcher.cpp:103-120: the second neighbour, score 1, no proximity test
        if (1.f < squared_ratio_thres)
          push(i, nn[1].second, 1.f, 1);
        continue;
      }
      const int top1 = self_matching ? 1 : 0;
      if (top1 + 1 >= nt)
        continue;  // a single key self-matching: knnSearch has nothing to rank
      const float d_top1 = nn[size_t(top1)].first;
      const float top1_score =
          nn[size_t(top1) + 1].first > 0.f ? d_top1 / nn[size_t(top1) + 1].first : 0.f;
      int K = 1;
      if (squared_ratio_thres > 1.f)
      {
        const float radius = d_top1 * squared_ratio_thres;
        K = 0;
        while (K < nt && nn[size_t(K)].first < radius)
          ++K;
      }
      for (int rank = top1; rank < K; ++rank)
      {
        float score = 0.f;
        if (rank == top1)
          score = top1_score;
        else if (d_top1)
          score = nn[size_t(rank)].first / d_top1;
        if (score > squared_ratio_thres)
          break;
        const int i2 = nn[size_t(rank)].second;
        if (self_matching && is_redundant(fq[i], ft[i2]))
          continue;
        push(i, i2, score, top1 == 0 ? rank + 1 : rank);
      }
    }
  }

  inline void finish_matches(std::vector<Match>& matches, const MatchFeature* f1,
                             const MatchFeature* f2)
  {
    // AnnMatcher.cpp:239-258.
    std::sort(matches.begin(), matches.end(), [](const Match& a, const Match& b) {
      if (a.x_index != b.x_index)
        return a.x_index < b.x_index;
      if (a.y_index != b.y_index)
        return a.y_index < b.y_index;
      if (a.score != b.score)
        return a.score < b.score;
      // equal scores from the two directions (e.g. bit-identical
      // descriptors, score 0): std::sort leaves their order unspecified in
      // the reference; SourceToTarget first, lower rank first is one valid
      // outcome
      if (a.direction != b.direction)
        return a.direction < b.direction;
      return a.rank < b.rank;
    });
    matches.erase(std::unique(matches.begin(), matches.end(),
                              [&](const Match& a, const Match& b) {
                                if (f1 && f2)
                                  return f1[a.x_index] == f1[b.x_index] &&
                                         f2[a.y_index] == f2[b.y_index];
                                return a.x_index == b.x_index &&
                                       a.y_index == b.y_index;
                              }),
                  matches.end());
    // The reference's final std::sort by score leaves equal scores in an
    // unspecified order; (score, x, y) is one valid outcome of it.
    std::sort(matches.begin(), matches.end(), [](const Match& a, const Match& b) {
      if (a.score != b.score)
        return a.score < b.score;
      if (a.x_index != b.x_index)
        return a.x_index < b.x_index;
      return a.y_index < b.y_index;
    });
  }

  //! AnnMatcher{keys1, keys2, ratio}.compute_matches().
  inline std::vector<Match> compute_matches(const float* desc1, int n1,
                                            const float* desc2, int n2, int dim,
                                            float sift_ratio_thres)
  {
    // create_flann_matrix, AnnMatcher.cpp:42-54.
    if (n1 == 0 || n2 == 0)
      throw std::runtime_error{"Error: the list of key-points is empty!"};
    const float thres2 = sift_ratio_thres * sift_ratio_thres;
    std::vector<Match> matches;
    const KeyProximity unused;
    append_matches(desc1, n1, desc2, n2, dim, thres2, 0, false, unused, nullptr,
                   nullptr, matches);
    append_matches(desc2, n2, desc1, n1, dim, thres2, 1, false, unused, nullptr,
                   nullptr, matches);
    finish_matches(matches, nullptr, nullptr);
    return matches;
  }

  //! AnnMatcher{keys, ratio, min_max_metric_dist_thres, pixel_dist_thres}
  //! .compute_matches() (the self-matching constructor, AnnMatcher.cpp:199-215).
  inline std::vector<Match> compute_self_matches(
      const float* desc, const MatchFeature* features, int n, int dim,
      float sift_ratio_thres, float min_max_metric_dist_thres,
      float pixel_dist_thres)
  {
    if (n == 0)
      throw std::runtime_error{"Error: the list of key-points is empty!"};
    const float thres2 = sift_ratio_thres * sift_ratio_thres;
    const KeyProximity too_close{min_max_metric_dist_thres, pixel_dist_thres};
    std::vector<Match> matches;
    append_matches(desc, n, desc, n, dim, thres2, 0, true, too_close, features,
                   features, matches);
    append_matches(desc, n, desc, n, dim, thres2, 1, true, too_close, features,
                   features, matches);
    finish_matches(matches, features, features);
    return matches;
  }

}  // namespace sara_ref
