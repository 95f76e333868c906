// MOCK - see ../README.md.  DO/Sara/ImageProcessing/ImagePyramid.hpp:29-340:
// ImagePyramidParams (constructor argument order and accessors) and the
// members of ImagePyramid the shim fills.
#pragma once
#include <DO/Sara/Core/Image.hpp>

#include <cmath>
#include <limits>
#include <vector>

namespace DO::Sara {

  class ImagePyramidParams
  {
  public:
    ImagePyramidParams(int first_octave_index = -1, int scale_count_per_octave = 3 + 3,
                       float scale_geometric_factor = std::pow(2.f, 1.f / 3.f),
                       int image_padding_size = 1, float scale_camera = 0.5f,
                       float scale_initial = 1.6f,
                       int num_octaves_max = std::numeric_limits<int>::max())
      : _a{first_octave_index}, _b{scale_count_per_octave}
      , _k{scale_geometric_factor}, _p{image_padding_size}, _c{scale_camera}
      , _i{scale_initial}, _n{num_octaves_max}
    {
    }
    int first_octave_index() const { return _a; }
    int scale_count_per_octave() const { return _b; }
    float scale_geometric_factor() const { return _k; }
    int image_padding_size() const { return _p; }
    float scale_camera() const { return _c; }
    float scale_initial() const { return _i; }
    int num_octaves_max() const { return _n; }

  private:
    int _a, _b;
    float _k;
    int _p;
    float _c, _i;
    int _n;
  };

  template <typename Pixel, int N = 2>
  class ImagePyramid
  {
  public:
    using image_type = Image<Pixel>;
    void reset(int num_octaves, int num_scales_per_octave, float scale_initial,
               float scale_geometric_factor)
    {
      _o.assign(std::size_t(num_octaves),
                std::vector<image_type>(std::size_t(num_scales_per_octave)));
      _f.assign(std::size_t(num_octaves), 0.f);
      _s0 = scale_initial;
      _k = scale_geometric_factor;
    }
    image_type& operator()(int s, int o) { return _o[o][s]; }
    const image_type& operator()(int s, int o) const { return _o[o][s]; }
    float& octave_scaling_factor(int o) { return _f[o]; }
    float octave_scaling_factor(int o) const { return _f[o]; }
    int octave_count() const { return int(_o.size()); }
    int scale_count_per_octave() const { return int(_o.front().size()); }

  private:
    std::vector<std::vector<image_type>> _o;
    std::vector<float> _f;
    float _s0 = 0, _k = 0;
  };

}  // namespace DO::Sara
