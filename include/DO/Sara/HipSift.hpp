// ========================================================================== //
// DO::Sara shim over the C-ABI of include/sara_hip_sift.h.
//
// Gives the MI355X SIFT front-end the exact C++ surface of the reference path
// so that its callers compile unchanged:
//
//   compute_sift_keypoints()        FeatureDetectors/SIFT.hpp:24-33
//   ComputeDoGExtrema               FeatureDetectors/DoG.hpp:72-165
//   ImagePyramidParams/ImagePyramid ImageProcessing/ImagePyramid.hpp:29-340
//   OERegion, KeypointList, features(), descriptors()
//                                   Features/Feature.hpp:40-179,
//                                   Features/KeypointList.hpp:35-96
//
// Two modes:
//  * standalone (default): minimal stand-ins with the same member names, no
//    Eigen required, and the functions / classes below live in DO::Sara under
//    the reference's own names.  This is what tests/cpp/test_shim.cpp builds.
//  * inside Sara (define SARA_HIP_WITH_SARA_HEADERS): Sara's own Image /
//    OERegion / Tensor_ / KeypointList / Match types are used (sara_oeregion
//    is byte-compatible with OERegion: 48 bytes, same member offsets - checked
//    by static_assert below - so results are memcpy'd), and because Sara
//    already defines DO::Sara::compute_sift_keypoints, ComputeDoGExtrema,
//    AnnMatcher, match and from_rgb8_to_gray32f, the GPU versions live in the
//    nested namespace DO::Sara::hip with the same names and signatures:
//    a call site switches with one `using`/qualification.
//    tests/cpp/test_shim_in_sara.cpp compiles this branch against a MOCK of
//    the five Sara headers it includes (tests/cpp/mock_sara/ - it proves that
//    the branch is well-formed, not parity).
//
// Error convention: C status codes are re-thrown as the exception classes the
// reference throws (std::runtime_error for the scale-count check DoG.hpp:86-89,
// std::domain_error for size mismatches LinearFiltering.cpp:33-35,
// std::out_of_range GaussianPyramid.hpp:187-190).
// ========================================================================== //
#pragma once

#include <sara_hip_sift.h>
#ifdef SARA_HIP_WITH_HDF5
#  include <sara_keypoint_h5.h>
#endif

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <type_traits>
#include <vector>

#ifdef SARA_HIP_WITH_SARA_HEADERS
#  include <DO/Sara/Core/Image.hpp>
#  include <DO/Sara/Core/Tensor.hpp>
#  include <DO/Sara/Features/KeypointList.hpp>
#  include <DO/Sara/ImageProcessing/ImagePyramid.hpp>
#  include <DO/Sara/Match/Match.hpp>
#  define SARA_HIP_SHIM_NS_BEGIN namespace hip {
#  define SARA_HIP_SHIM_NS_END }
#else
#  define SARA_HIP_SHIM_NS_BEGIN
#  define SARA_HIP_SHIM_NS_END
#endif

namespace DO::Sara {

#ifndef SARA_HIP_WITH_SARA_HEADERS
  // ---- standalone stand-ins (same names / members as Sara's types) --------

  //! Core/Image/Image.hpp:45-181 restricted to float, 2-D: borrowed buffer,
  //! pixel (x, y) at y * width + x.
  template <typename T>
  class ImageView
  {
  public:
    ImageView() = default;
    ImageView(T* data, int width, int height)
      : _data{data}, _w{width}, _h{height}
    {
    }
    auto width() const -> int { return _w; }
    auto height() const -> int { return _h; }
    auto data() const -> T* { return _data; }
    auto operator()(int x, int y) const -> T& { return _data[size_t(y) * _w + x]; }

  protected:
    T* _data = nullptr;
    int _w = 0, _h = 0;
  };

  template <typename T>
  class Image : public ImageView<T>
  {
  public:
    Image() = default;
    Image(int width, int height)
      : _storage(size_t(width) * height)
    {
      this->_data = _storage.data();
      this->_w = width;
      this->_h = height;
    }
    Image(const Image& o) { *this = o; }
    Image& operator=(const Image& o)
    {
      _storage = o._storage;
      this->_data = _storage.data();
      this->_w = o._w;
      this->_h = o._h;
      return *this;
    }

  private:
    std::vector<T> _storage;
  };

  //! Core/Pixel/Typedefs.hpp Rgb8: three contiguous bytes.
  struct Rgb8
  {
    std::uint8_t r, g, b;
  };

  //! ImageProcessing/ImagePyramid.hpp:29-198.
  class ImagePyramidParams
  {
  public:
    ImagePyramidParams(const int first_octave_index = -1,
                       const int scale_count_per_octave = 3 + 3,
                       const float scale_geometric_factor =
                           std::pow(2.f, 1.f / 3.f),
                       const int image_padding_size = 1,
                       const float scale_camera = 0.5f,
                       const float scale_initial = 1.6f,
                       const int num_octaves_max =
                           std::numeric_limits<int>::max())
      : _p{first_octave_index, scale_count_per_octave, scale_geometric_factor,
           image_padding_size, scale_camera, scale_initial, num_octaves_max}
    {
    }
    float scale_camera() const { return _p.scale_camera; }
    float scale_initial() const { return _p.scale_initial; }
    float scale_geometric_factor() const { return _p.scale_geometric_factor; }
    int scale_count_per_octave() const { return _p.scale_count_per_octave; }
    int image_padding_size() const { return _p.image_padding_size; }
    int first_octave_index() const { return _p.first_octave_index; }
    int num_octaves_max() const { return _p.num_octaves_max; }
    const sara_pyramid_params& c_params() const { return _p; }

  private:
    sara_pyramid_params _p;
  };

  //! ImageProcessing/ImagePyramid.hpp:206-340 (host mirror, float pixels).
  template <typename Pixel, int N = 2>
  class ImagePyramid
  {
  public:
    using image_type = Image<Pixel>;
    using octave_type = std::vector<image_type>;
    void reset(int num_octaves, int num_scales, float scale_initial,
               float scale_geometric_factor)
    {
      _octaves.assign(num_octaves, octave_type(num_scales));
      _oct_scaling_factors.assign(num_octaves, 0.f);
      _scale_initial = scale_initial;
      _scale_geometric_factor = scale_geometric_factor;
    }
    image_type& operator()(int s, int o) { return _octaves[o][s]; }
    const image_type& operator()(int s, int o) const { return _octaves[o][s]; }
    const octave_type& operator()(int o) const { return _octaves[o]; }
    Pixel operator()(int x, int y, int s, int o) const
    {
      return _octaves[o][s](x, y);
    }
    float& octave_scaling_factor(int o) { return _oct_scaling_factors[o]; }
    float octave_scaling_factor(int o) const { return _oct_scaling_factors[o]; }
    int octave_count() const { return int(_octaves.size()); }
    int scale_count_per_octave() const { return int(_octaves.front().size()); }
    int scale_count() const { return octave_count() * scale_count_per_octave(); }
    float scale_initial() const { return _scale_initial; }
    float scale_geometric_factor() const { return _scale_geometric_factor; }
    double scale_relative_to_octave(int s) const
    {
      return std::pow(_scale_geometric_factor, s) * _scale_initial;
    }
    double scale(int s, int o) const
    {
      return _oct_scaling_factors[o] * scale_relative_to_octave(s);
    }

  private:
    float _scale_initial = 0.f, _scale_geometric_factor = 0.f;
    std::vector<octave_type> _octaves;
    std::vector<float> _oct_scaling_factors;
  };

  //! Features/Feature.hpp:40-179: same members, same 48-byte layout.
  struct alignas(16) OERegion
  {
    enum class Type : std::uint8_t
    {
      Harris, HarAff, HarLap, FAST, SUSAN, DoG, LoG, DoH, MSER, HesAff, HesLap,
      Undefined
    };
    enum class ExtremumType : std::int8_t
    {
      Min = -1, Saddle = 0, Max = 1, Undefined = -2
    };
    std::array<float, 2> coords{{0.f, 0.f}};
    alignas(16) std::array<float, 4> shape_matrix{{0.f, 0.f, 0.f, 0.f}};
    float orientation{0};
    float extremum_value{0};
    Type type{Type::Undefined};
    ExtremumType extremum_type{ExtremumType::Undefined};

    float x() const { return coords[0]; }
    float y() const { return coords[1]; }
    const std::array<float, 2>& center() const { return coords; }
    //! Features/Feature.cpp:28-39 for the isotropic regions this path emits.
    float radius(float = 0.f) const { return 1.f / std::sqrt(shape_matrix[0]); }
    float scale(float radian = 0.f) const { return radius(radian); }
    bool operator==(const OERegion& o) const
    {
      return coords == o.coords && shape_matrix == o.shape_matrix &&
             orientation == o.orientation && type == o.type;
    }
  };

  //! Core/Tensor.hpp:41-45, row-major N x D matrix of descriptors.
  template <typename T, int N>
  class Tensor_
  {
    static_assert(N == 2, "only matrices are needed on this path");

  public:
    Tensor_() = default;
    Tensor_(int rows, int cols) { resize(rows, cols); }
    void resize(int rows, int cols)
    {
      _rows = rows;
      _cols = cols;
      _d.assign(size_t(rows) * cols, T{});
    }
    //! rows x cols values copied from `src` in one pass (resize() + memcpy
    //! would write the 2.2 MB of a frame's descriptors twice)
    void assign(const T* src, int rows, int cols)
    {
      _rows = rows;
      _cols = cols;
      _d.assign(src, src + size_t(rows) * cols);
    }
    int rows() const { return _rows; }
    int cols() const { return _cols; }
    std::array<int, 2> sizes() const { return {{_rows, _cols}}; }
    T* data() { return _d.data(); }
    const T* data() const { return _d.data(); }
    T& operator()(int i, int j) { return _d[size_t(i) * _cols + j]; }
    const T& operator()(int i, int j) const { return _d[size_t(i) * _cols + j]; }
    const T* operator[](int i) const { return &_d[size_t(i) * _cols]; }

  private:
    int _rows = 0, _cols = 0;
    std::vector<T> _d;
  };

  //! Core/EigenExtension.hpp:139 (Eigen::Vector2i): constructor (x, y),
  //! operator[] / operator() / x() / y().
  struct Point2i
  {
    int v[2] = {0, 0};
    Point2i() = default;
    Point2i(int x, int y) : v{x, y} {}
    int& operator[](int i) { return v[i]; }
    int operator[](int i) const { return v[i]; }
    int& operator()(int i) { return v[i]; }
    int operator()(int i) const { return v[i]; }
    int x() const { return v[0]; }
    int y() const { return v[1]; }
    bool operator==(const Point2i& o) const { return v[0] == o.v[0] && v[1] == o.v[1]; }
  };

  //! Features/KeypointList.hpp:35-96.
  template <typename F, typename T>
  using KeypointList = std::tuple<std::vector<F>, Tensor_<T, 2>>;
  template <typename F, typename T>
  inline auto features(const KeypointList<F, T>& k) -> const std::vector<F>&
  {
    return std::get<0>(k);
  }
  template <typename F, typename T>
  inline auto descriptors(const KeypointList<F, T>& k) -> const Tensor_<T, 2>&
  {
    return std::get<1>(k);
  }
  template <typename F, typename T>
  inline auto size(const KeypointList<F, T>& k)
  {
    return descriptors(k).rows();
  }
  template <typename F, typename T>
  inline auto size_consistency_predicate(const KeypointList<F, T>& k)
  {
    return int(features(k).size()) == descriptors(k).rows();
  }
  // ---- Features/IO.hpp:77-143: the readable keypoint format (SURVEY.md
  // section 8f, row f3).  One header line "N dim", then per keypoint
  //   x y m00 m10 m01 m11 orientation type d0 ... d(dim-1)
  // where the four shape coefficients (column-major, written through
  // Map<RowVector4f>) and the descriptor row are Eigen expressions: Eigen's
  // operator<< right-aligns every coefficient of an expression to the widest
  // one (Eigen/src/Core/IO.h, default IOFormat).  Reading goes through
  // OERegion's operator>> (Features/Feature.cpp:88-95), which fills the shape
  // matrix ROW-major - the reference's own transposition quirk, harmless for
  // the symmetric matrices it stores.
  namespace hip_detail {
    //! Eigen's operator<< with the default IOFormat (Eigen/src/Core/IO.h):
    //! ONE width for the whole expression = the widest coefficient as the
    //! stream would print it, every coefficient right-aligned to it, " "
    //! between columns, "\n" between rows.  `v` is column-major (Eigen's
    //! default storage).  The reference's own examples/Sara/Features/
    //! test.dogkey holds 578 such 2 x 2 blocks (tests/test_keypoint_text_pins.py).
    template <typename T>
    inline void print_eigen_block(std::ostream& os, const T* v, int rows,
                                  int cols)
    {
      std::size_t width = 0;
      for (int i = 0; i < rows * cols; ++i)
      {
        std::stringstream sstr;
        sstr.copyfmt(os);
        sstr << v[i];
        width = std::max(width, sstr.str().length());
      }
      for (int r = 0; r < rows; ++r)
      {
        if (r)
          os << "\n";
        for (int c = 0; c < cols; ++c)
        {
          if (c)
            os << " ";
          if (width)
            os.width(std::streamsize(width));
          os << v[c * rows + r];
        }
      }
    }
    template <typename T>
    inline void print_eigen_row(std::ostream& os, const T* v, int n)
    {
      print_eigen_block(os, v, 1, n);
    }
  }  // namespace hip_detail

  template <typename T>
  inline bool write_keypoints(const std::vector<OERegion>& features,
                              const Tensor_<T, 2>& descriptors,
                              const std::string& name)
  {
    std::ofstream file{name.c_str()};
    if (!file.is_open())
    {
      std::cerr << "Can't open file" << std::endl;
      return false;
    }
    file << features.size() << " " << descriptors.cols() << std::endl;
    for (std::size_t i = 0; i < features.size(); ++i)
    {
      const OERegion& feat = features[i];
      file << feat.x() << ' ' << feat.y() << ' ';
      hip_detail::print_eigen_row(file, feat.shape_matrix.data(), 4);
      file << ' ';
      file << feat.orientation << ' ';
      file << int(feat.type) << ' ';
      hip_detail::print_eigen_row(file, descriptors[int(i)], descriptors.cols());
      file << std::endl;
    }
    return true;
  }

  template <typename T>
  inline bool read_keypoints(std::vector<OERegion>& features,
                             Tensor_<T, 2>& descriptors, const std::string& name)
  {
    std::ifstream file{name.c_str()};
    if (!file.is_open())
    {
      std::cerr << "Can't open file " << name << std::endl;
      return false;
    }
    int num_features = 0, descriptor_dim = 0;
    file >> num_features >> descriptor_dim;
    features.assign(std::size_t(num_features), OERegion{});
    descriptors.resize(num_features, descriptor_dim);
    for (int i = 0; i < num_features; ++i)
    {
      OERegion& f = features[std::size_t(i)];
      int feature_type = 0;
      file >> f.coords[0] >> f.coords[1];
      // row-major fill of the column-major 2x2 (Core/EigenExtension.hpp:163-170)
      file >> f.shape_matrix[0] >> f.shape_matrix[2] >> f.shape_matrix[1] >>
          f.shape_matrix[3];
      file >> f.orientation >> feature_type;
      f.type = static_cast<decltype(f.type)>(feature_type);
      for (int j = 0; j < descriptor_dim; ++j)
        file >> descriptors(i, j);
    }
    return true;
  }
#endif  // !SARA_HIP_WITH_SARA_HEADERS

  static_assert(sizeof(OERegion) == sizeof(sara_oeregion),
                "OERegion must stay byte-compatible with sara_oeregion");

#if defined(SARA_HIP_WITH_HDF5) && !defined(SARA_HIP_WITH_SARA_HEADERS)
  //! Core/HDF5.hpp:160-193 reduced to what keypoint files need: a file name
  //! and an access mode (HDF5's own flag values); every call below opens and
  //! closes the file through libsara_keypoint_h5.so.
  struct H5File
  {
    enum : unsigned { AccRdOnly = 0x0u, AccRdWr = 0x1u, AccTrunc = 0x2u };
    H5File(const std::string& filename_, unsigned flags_)
      : filename{filename_}
      , flags{flags_}
      , truncate_pending{(flags_ & AccTrunc) != 0}
    {
    }
    std::string filename;
    unsigned flags;
    bool truncate_pending;
  };

  //! Features/IO.hpp:146-157.
  inline auto read_keypoints(H5File& h5_file, const std::string& group_name)
      -> KeypointList<OERegion, float>
  {
    int n = 0, dim = 0;
    if (sara_h5_keypoints_sizes(h5_file.filename.c_str(), group_name.c_str(), &n,
                                &dim))
      throw std::runtime_error{sara_h5_last_error()};
    auto features = std::vector<OERegion>(std::size_t(n));
    auto descriptors = Tensor_<float, 2>{n, dim};
    if (sara_h5_read_keypoints(h5_file.filename.c_str(), group_name.c_str(),
                               reinterpret_cast<sara_oeregion*>(features.data()),
                               descriptors.data()))
      throw std::runtime_error{sara_h5_last_error()};
    return {features, descriptors};
  }

  //! Features/IO.hpp:159-167.
  inline auto write_keypoints(H5File& h5_file, const std::string& group_name,
                              const KeypointList<OERegion, float>& keys,
                              bool overwrite = false) -> void
  {
    const auto& f = features(keys);
    const auto& v = descriptors(keys);
    if (h5_file.flags == H5File::AccRdOnly)
      throw std::runtime_error{"Error: the file is opened read-only"};
    if (sara_h5_write_keypoints(
            h5_file.filename.c_str(), h5_file.truncate_pending ? 1 : 0,
            group_name.c_str(), reinterpret_cast<const sara_oeregion*>(f.data()),
            int(f.size()), v.data(), v.cols(), overwrite ? 1 : 0))
      throw std::runtime_error{sara_h5_last_error()};
    h5_file.truncate_pending = false;
  }
#endif

  namespace hip_detail {

    [[noreturn]] inline void rethrow(sara_hip_status st)
    {
      const std::string msg = sara_hip_last_error();
      switch (st)
      {
      case SARA_HIP_SIZE_MISMATCH:
        throw std::domain_error{msg};
      case SARA_HIP_OUT_OF_RANGE:
        throw std::out_of_range{msg};
      default:
        throw std::runtime_error{msg};
      }
    }

    inline void check(sara_hip_status st)
    {
      if (st != SARA_HIP_OK)
        rethrow(st);
    }

    inline sara_pyramid_params to_c(const ImagePyramidParams& p)
    {
      sara_pyramid_params c;
      c.first_octave_index = p.first_octave_index();
      c.scale_count_per_octave = p.scale_count_per_octave();
      c.scale_geometric_factor = p.scale_geometric_factor();
      c.image_padding_size = p.image_padding_size();
      c.scale_camera = p.scale_camera();
      c.scale_initial = p.scale_initial();
      c.num_octaves_max = p.num_octaves_max();
      return c;
    }

    struct Deleter
    {
      void operator()(sara_hip_sift* c) const { sara_hip_sift_destroy(c); }
    };
    using Context = std::unique_ptr<sara_hip_sift, Deleter>;

    //! compute_sift_keypoints() is a free function that callers invoke once
    //! per video frame (OdometryPipeline::detect_keypoints,
    //! SfM/Odometry/OdometryPipeline.cpp:82-90).  Creating a context costs tens
    //! of milliseconds (pyramid, gradient and list buffers in HBM) against
    //! less than one for the detection itself, so every thread keeps the few
    //! contexts it used last, keyed by everything a context is built from.
    struct ContextKey
    {
      sara_sift_params params;
      int width, height, device;
      bool operator==(const ContextKey& o) const
      {
        // field-wise: the C struct may have padding
        const sara_pyramid_params &a = params.pyramid, &b = o.params.pyramid;
        return a.first_octave_index == b.first_octave_index &&
               a.scale_count_per_octave == b.scale_count_per_octave &&
               a.scale_geometric_factor == b.scale_geometric_factor &&
               a.image_padding_size == b.image_padding_size &&
               a.scale_camera == b.scale_camera &&
               a.scale_initial == b.scale_initial &&
               a.num_octaves_max == b.num_octaves_max &&
               params.gauss_truncate == o.params.gauss_truncate &&
               params.extremum_thres == o.params.extremum_thres &&
               params.edge_ratio_thres == o.params.edge_ratio_thres &&
               params.extremum_refinement_iter ==
                   o.params.extremum_refinement_iter &&
               width == o.width && height == o.height && device == o.device;
      }
    };

    inline sara_hip_sift* cached_context(const ContextKey& key)
    {
      constexpr std::size_t kMaxCached = 4;
      thread_local std::vector<std::pair<ContextKey, Context>> cache;
      for (std::size_t i = 0; i < cache.size(); ++i)
        if (cache[i].first == key)
        {
          if (i != 0)
            std::rotate(cache.begin(), cache.begin() + i, cache.begin() + i + 1);
          return cache.front().second.get();
        }
      sara_hip_sift* raw = nullptr;
      check(sara_hip_sift_create(&key.params, key.width, key.height, 1, 0,
                                 key.device, &raw));
      if (cache.size() >= kMaxCached)
        cache.pop_back();
      cache.emplace(cache.begin(), key, Context{raw});
      return raw;
    }

    //! The reference keeps its extrema in std::vectors (reserve(10000), then
    //! push_back: FeatureDetectors/RefineExtremum.cpp:496-514), so no image has
    //! "too many" keypoints.  The context's lists have a capacity; when a frame
    //! overflows them, the callers below grow the lists to twice what the frame
    //! asked for and run it again (a list that overflowed starves the ones
    //! behind it, hence a loop; every step at least doubles).  The grown
    //! context stays where it is cached: a video pays for the growth once.
    constexpr int kMaxGrowthSteps = 8;
    inline void grow_for_last_batch(sara_hip_sift* ctx)
    {
      int cap = 0, need = 0;
      check(sara_hip_sift_capacity(ctx, &cap, &need));
      check(sara_hip_sift_reserve(ctx, 2 * std::max(cap, need)));
    }

  }  // namespace hip_detail

  static_assert(sizeof(Rgb8) == 3, "Rgb8 must be three packed bytes");
  // results are memcpy'd from sara_oeregion records (also inside Sara)
  static_assert(sizeof(OERegion) == sizeof(sara_oeregion) &&
                    alignof(OERegion) == 16,
                "OERegion must be the 48-byte record of the C-ABI");

  // Inside Sara these names already exist in DO::Sara (FeatureDetectors/
  // SIFT.hpp:24-33, DoG.hpp:72-165, FeatureMatching/AnnMatcher.hpp:32-86,
  // ImageProcessing/FastColorConversion.hpp:22-23): there the GPU versions
  // are DO::Sara::hip::<same name>.
  SARA_HIP_SHIM_NS_BEGIN

  //! ImageProcessing/FastColorConversion.hpp:22-23 (.cpp:42-66): same
  //! signature, same std::domain_error on a size mismatch.
  inline auto from_rgb8_to_gray32f(const ImageView<Rgb8>& src,
                                   ImageView<float>& dst, int device = 0) -> void
  {
    if (src.width() != dst.width() || src.height() != dst.height())
      throw std::domain_error{
          "Color conversion error: image sizes are not equal!"};
    hip_detail::check(sara_hip_from_rgb8_to_gray32f(
        reinterpret_cast<const std::uint8_t*>(src.data()), dst.data(),
        src.width(), src.height(), device));
  }

  //! FeatureDetectors/SIFT.hpp:24-33.  `parallel` is accepted for signature
  //! compatibility; `device` selects the GPU (extension, defaulted).
  inline auto compute_sift_keypoints(
      const ImageView<float>& image,
      const ImagePyramidParams& pyramid_params = ImagePyramidParams(),
      float gauss_truncate = 4.f, float extremum_thres = 0.01f,
      float edge_ratio_thres = 10.f, int extremum_refinement_iter = 5,
      bool /*parallel*/ = false, int device = 0)
      -> KeypointList<OERegion, float>
  {
    sara_sift_params p;
    p.pyramid = hip_detail::to_c(pyramid_params);
    p.gauss_truncate = gauss_truncate;
    p.extremum_thres = extremum_thres;
    p.edge_ratio_thres = edge_ratio_thres;
    p.extremum_refinement_iter = extremum_refinement_iter;
    // the context (and all its HBM buffers) is reused across calls
    sara_hip_sift* ctx = hip_detail::cached_context(
        hip_detail::ContextKey{p, image.width(), image.height(), device});
    // submit / collect: upload, kernels, and the read-back into pinned memory
    // owned by the context; one copy from there into the returned containers
    const sara_oeregion* f = nullptr;
    const float* d = nullptr;
    int total = 0;
    for (int step = 0;; ++step)
    {
      int ticket = -1;
      hip_detail::check(sara_hip_sift_submit(ctx, image.data(), 0, 0, 1,
                                             image.width(), image.height(), 0,
                                             SARA_HIP_STAGE_DESCRIPTOR, &ticket));
      const sara_hip_status st =
          sara_hip_sift_collect(ctx, ticket, &f, &d, nullptr, nullptr, &total);
      if (st != SARA_HIP_CAPACITY_EXCEEDED || step + 1 >= hip_detail::kMaxGrowthSteps)
      {
        hip_detail::check(st);
        break;
      }
      hip_detail::grow_for_last_batch(ctx);
    }
#ifndef SARA_HIP_WITH_SARA_HEADERS
    // the stand-in containers are filled in one pass (no value-initialisation
    // in front of the copy: 0.25 -> 0.2 ms of a 0.8 ms call)
    static_assert(std::is_trivially_copyable<OERegion>::value, "record copy");
    const OERegion* fr = reinterpret_cast<const OERegion*>(f);
    auto feats = total > 0 ? std::vector<OERegion>(fr, fr + total)
                           : std::vector<OERegion>{};
    auto desc = Tensor_<float, 2>{};
    if (total > 0)
      desc.assign(d, total, 128);
    else
      desc.resize(0, 128);
#else
    auto feats = std::vector<OERegion>(size_t(total));
    auto desc = Tensor_<float, 2>{};
    desc.resize(total, 128);
    if (total > 0)
    {
      std::memcpy(static_cast<void*>(feats.data()), f,
                  sizeof(sara_oeregion) * size_t(total));
      std::memcpy(desc.data(), d, sizeof(float) * 128 * size_t(total));
    }
#endif
    return {std::move(feats), std::move(desc)};
  }

#ifndef SARA_HIP_WITH_SARA_HEADERS
  //! Match/Match.hpp:29-174 reduced to what AnnMatcher fills in: pointers and
  //! indices of the two keypoints, score (squared-distance ratio), rank and
  //! direction.
  class Match
  {
  public:
    enum class Direction
    {
      SourceToTarget = 0,
      TargetToSource = 1
    };
    Match() = default;
    Match(const OERegion* x, const OERegion* y, float score, Direction dir,
          int x_index, int y_index)
      : _x{x}, _y{y}, _x_index{x_index}, _y_index{y_index}, _score{score}
      , _matching_dir{dir}
    {
    }
    const OERegion& x() const { return *_x; }
    const OERegion& y() const { return *_y; }
    int x_index() const { return _x_index; }
    int y_index() const { return _y_index; }
    int rank() const { return _rank; }
    int& rank() { return _rank; }
    float score() const { return _score; }
    Direction matching_direction() const { return _matching_dir; }
    //! Match.hpp:161-164: equality of the two keypoints BY VALUE.
    bool operator==(const Match& m) const { return x() == m.x() && y() == m.y(); }

  private:
    const OERegion* _x = nullptr;
    const OERegion* _y = nullptr;
    int _x_index = -1, _y_index = -1, _rank = -1;
    float _score = std::numeric_limits<float>::max();
    Direction _matching_dir = Direction::SourceToTarget;
  };
#endif  // !SARA_HIP_WITH_SARA_HEADERS (inside Sara: DO/Sara/Match/Match.hpp)

  //! FeatureMatching/AnnMatcher.hpp:32-86, both constructors and their
  //! defaults (sift_ratio_thres = 1.2f: the adaptive radius search,
  //! AnnMatcher.cpp:133-138).  The neighbour searches run on the GPU, exact,
  //! with FLANN's distance arithmetic; `device` is an extension, defaulted.
  class AnnMatcher
  {
  public:
    AnnMatcher(const KeypointList<OERegion, float>& keys1,
               const KeypointList<OERegion, float>& keys2,
               float sift_ratio_thres = 1.2f, int device = 0)
      : _keys1{keys1}, _keys2{keys2}, _ratio{sift_ratio_thres}, _device{device}
    {
      if (!size_consistency_predicate(_keys1) ||
          !size_consistency_predicate(_keys2))
        throw std::runtime_error{
            "The list of keypoints are inconsistent in size!"};
    }

    //! Self-matching (AnnMatcher.hpp:42-46, .cpp:199-215).
    AnnMatcher(const KeypointList<OERegion, float>& keys,
               float sift_ratio_thres = 1.2f,
               float min_max_metric_dist_thres = 0.5f,
               float pixel_dist_thres = 10.f, int device = 0)
      : _keys1{keys}, _keys2{keys}, _ratio{sift_ratio_thres}
      , _metric_dist_thres{min_max_metric_dist_thres}
      , _pixel_dist_thres{pixel_dist_thres}, _self_matching{true}
      , _device{device}
    {
      if (!size_consistency_predicate(_keys1))
        throw std::runtime_error{
            "The list of keypoints are inconsistent in size!"};
    }

    auto keys1() const -> const KeypointList<OERegion, float>& { return _keys1; }
    auto keys2() const -> const KeypointList<OERegion, float>& { return _keys2; }

    auto compute_matches() -> std::vector<Match>
    {
      const auto& f1 = features(_keys1);
      const auto& f2 = features(_keys2);
      const auto& d1 = descriptors(_keys1);
      const auto& d2 = descriptors(_keys2);
      // above ratio 1 a key can have several matches: retry with the count
      // the library reports
      auto raw = std::vector<sara_match>(2 * (f1.size() + f2.size()) + 16);
      int count = 0;
      for (int attempt = 0; attempt < 2; ++attempt)
      {
        const auto st =
            _self_matching
                ? sara_hip_self_match_descriptors(
                      d1.data(), reinterpret_cast<const sara_oeregion*>(f1.data()),
                      int(f1.size()), int(d1.cols()), _ratio, _metric_dist_thres,
                      _pixel_dist_thres, 0, raw.data(), int(raw.size()), &count,
                      _device)
                : sara_hip_match_descriptors(
                      d1.data(), int(f1.size()), d2.data(), int(f2.size()),
                      f1.empty() ? int(d2.cols()) : int(d1.cols()), _ratio, 0,
                      raw.data(), int(raw.size()), &count, _device);
        if (st == SARA_HIP_CAPACITY_EXCEEDED && attempt == 0)
        {
          raw.resize(size_t(count));
          continue;
        }
        hip_detail::check(st);
        break;
      }
      auto matches = std::vector<Match>{};
      matches.reserve(size_t(count));
      for (int i = 0; i < count; ++i)
      {
        const auto& r = raw[size_t(i)];
        auto m = Match{&f1[size_t(r.x_index)], &f2[size_t(r.y_index)], r.score,
                       static_cast<Match::Direction>(r.direction), r.x_index,
                       r.y_index};
        m.rank() = r.rank;
        matches.push_back(m);
      }
      return matches;
    }

    auto compute_self_matches() -> std::vector<Match> { return compute_matches(); }

  private:
    const KeypointList<OERegion, float>& _keys1;
    const KeypointList<OERegion, float>& _keys2;
    float _ratio;
    float _metric_dist_thres = 0.5f, _pixel_dist_thres = 10.f;
    bool _self_matching = false;
    int _device;
  };

  //! SfM/Helpers/KeypointMatching.cpp:19-25.
  inline auto match(const KeypointList<OERegion, float>& keys1,
                    const KeypointList<OERegion, float>& keys2, float lowe_ratio)
      -> std::vector<Match>
  {
    AnnMatcher matcher{keys1, keys2, lowe_ratio};
    return matcher.compute_matches();
  }

  //! FeatureDescriptors/RootSIFT.hpp:45-53 applied to the descriptor matrix of a
  //! keypoint list (ComputeRootSIFTDescriptor wraps the base operator and
  //! post-processes every descriptor the same way): rows /= L1 norm, then sqrt.
  inline void root_sift(Tensor_<float, 2>& descriptors, int device = 0)
  {
    hip_detail::check(sara_hip_root_sift(descriptors.data(), descriptors.rows(),
                                         descriptors.cols(), 0, device));
  }

  //! FeatureDetectors/DoG.hpp:72-165.  The pyramids stay in HBM; gaussians()
  //! and diff_of_gaussians() copy them to the host on first use.
  class ComputeDoGExtrema
  {
  public:
    ComputeDoGExtrema(
        const ImagePyramidParams& pyramid_params = ImagePyramidParams(),
        float gauss_truncate = 4.f, float extremum_thres = 0.01f,
        float edge_ratio_thres = 10.f, int img_padding_sz = 1,
        int extremum_refinement_iter = 5, int device = 0)
      : _pyramid_params{pyramid_params}
      , _gauss_truncate{gauss_truncate}
      , _extremum_thres{extremum_thres}
      , _edge_ratio_thres{edge_ratio_thres}
      , _img_padding_sz{img_padding_sz}
      , _extremum_refinement_iter{extremum_refinement_iter}
      , _device{device}
    {
      if (_pyramid_params.scale_count_per_octave() < 4)
        throw std::runtime_error{
            "Error: The extraction of DoG extrema needs (1 + 3) = 4 scales per "
            "octave at the very minimum!"};
    }

    std::vector<OERegion> operator()(const ImageView<float>& I,
                                     std::vector<Point2i>* scale_octave_pairs = 0)
    {
      const auto cp = hip_detail::to_c(_pyramid_params);
      if (!_ctx || _ctx_w != I.width() || _ctx_h != I.height())
      {
        sara_hip_sift* raw = nullptr;
        hip_detail::check(sara_hip_sift_create_dog(
            &cp, _gauss_truncate, _extremum_thres, _edge_ratio_thres,
            _img_padding_sz, _extremum_refinement_iter, I.width(), I.height(), 1,
            0, _device, &raw));
        _ctx.reset(raw);
        _ctx_w = I.width();
        _ctx_h = I.height();
      }
      _have_g = _have_d = false;
      int total = 0;
      for (int step = 0;; ++step)
      {
        hip_detail::check(sara_hip_sift_detect(_ctx.get(), I.data(), 0, 1,
                                               I.width(), I.height(), 0,
                                               SARA_HIP_STAGE_EXTREMA, nullptr));
        const sara_hip_status st =
            sara_hip_sift_extrema_counts(_ctx.get(), nullptr, &total);
        if (st != SARA_HIP_CAPACITY_EXCEEDED ||
            step + 1 >= hip_detail::kMaxGrowthSteps)
        {
          hip_detail::check(st);
          break;
        }
        hip_detail::grow_for_last_batch(_ctx.get());
      }
      auto extrema = std::vector<OERegion>(size_t(total));
      auto xyso = std::vector<int32_t>(size_t(total) * 5);
      if (total > 0)
        hip_detail::check(sara_hip_sift_fetch_extrema(
            _ctx.get(), reinterpret_cast<sara_oeregion*>(extrema.data()),
            xyso.data()));
      // extrema(s, o): the per-(scale, octave) lists of DoG.hpp:160-163.
      const int no = sara_hip_sift_octave_count(_ctx.get());
      const int nd = _pyramid_params.scale_count_per_octave() - 1;
      _extrema.assign(size_t(std::max(no, 0)) * nd, {});
      if (scale_octave_pairs)
        scale_octave_pairs->clear();
      for (int i = 0; i < total; ++i)
      {
        const int s = xyso[5 * i + 2], o = xyso[5 * i + 3];
        _extrema[size_t(o) * nd + s].push_back(extrema[i]);
        if (scale_octave_pairs)
          scale_octave_pairs->push_back(Point2i(s, o));
      }
      return extrema;
    }

    auto gaussians() const -> const ImagePyramid<float>&
    {
      if (!_have_g)
      {
        fill(_gaussians, _pyramid_params.scale_count_per_octave(),
             sara_hip_sift_copy_gaussian);
        _have_g = true;
      }
      return _gaussians;
    }

    auto diff_of_gaussians() const -> const ImagePyramid<float>&
    {
      if (!_have_d)
      {
        fill(_diff_of_gaussians, _pyramid_params.scale_count_per_octave() - 1,
             sara_hip_sift_copy_dog);
        _have_d = true;
      }
      return _diff_of_gaussians;
    }

    auto extrema(int s, int o) const -> const std::vector<OERegion>&
    {
      return _extrema[size_t(o) * (_pyramid_params.scale_count_per_octave() - 1) +
                      s];
    }

  private:
    template <typename CopyFn>
    void fill(ImagePyramid<float>& P, int scales, CopyFn copy) const
    {
      if (!_ctx)
        throw std::runtime_error{"ComputeDoGExtrema: no image processed yet"};
      const int no = sara_hip_sift_octave_count(_ctx.get());
      P.reset(no, scales, _pyramid_params.scale_initial(),
              _pyramid_params.scale_geometric_factor());
      for (int o = 0; o < no; ++o)
      {
        int w = 0, h = 0;
        float f = 0.f;
        hip_detail::check(sara_hip_sift_octave_info(_ctx.get(), o, &w, &h, &f));
        P.octave_scaling_factor(o) = f;
        for (int s = 0; s < scales; ++s)
        {
          P(s, o) = Image<float>{w, h};
          hip_detail::check(copy(_ctx.get(), 0, s, o, P(s, o).data()));
        }
      }
    }

    ImagePyramidParams _pyramid_params;
    float _gauss_truncate, _extremum_thres, _edge_ratio_thres;
    int _img_padding_sz, _extremum_refinement_iter, _device;
    hip_detail::Context _ctx;
    int _ctx_w = 0, _ctx_h = 0;  // size the context was created for
    mutable ImagePyramid<float> _gaussians, _diff_of_gaussians;
    mutable bool _have_g = false, _have_d = false;
    std::vector<std::vector<OERegion>> _extrema;
  };

  SARA_HIP_SHIM_NS_END

}  // namespace DO::Sara

#undef SARA_HIP_SHIM_NS_BEGIN
#undef SARA_HIP_SHIM_NS_END
