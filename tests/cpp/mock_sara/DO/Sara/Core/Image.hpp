// MOCK - see ../README.md.  Public surface of DO/Sara/Core/Image.hpp as used
// by include/DO/Sara/HipSift.hpp: ImageView / Image (Core/Image/Image.hpp:
// 45-181), Rgb8 (Core/Pixel/Typedefs.hpp), and the Eigen typedefs of
// Core/EigenExtension.hpp:139-143 replaced by Eigen-free structs with Eigen's
// constructors and accessors.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace DO::Sara {

  template <typename T>
  struct MockVector2
  {
    T v[2] = {T{}, T{}};
    MockVector2() = default;
    MockVector2(T x, T y) : v{x, y} {}
    T& operator()(int i) { return v[i]; }
    T operator()(int i) const { return v[i]; }
    T& operator[](int i) { return v[i]; }
    T operator[](int i) const { return v[i]; }
    T x() const { return v[0]; }
    T y() const { return v[1]; }
    bool operator==(const MockVector2& o) const { return v[0] == o.v[0] && v[1] == o.v[1]; }
  };
  using Vector2i = MockVector2<int>;
  using Point2i = Vector2i;
  using Vector2f = MockVector2<float>;
  using Point2f = Vector2f;

  struct alignas(16) Matrix2f  // column-major 2 x 2, 16-byte aligned like Eigen's
  {
    float m[4] = {0, 0, 0, 0};
    float& operator()(int r, int c) { return m[c * 2 + r]; }
    float operator()(int r, int c) const { return m[c * 2 + r]; }
    const float* data() const { return m; }
    bool operator==(const Matrix2f& o) const
    {
      return m[0] == o.m[0] && m[1] == o.m[1] && m[2] == o.m[2] && m[3] == o.m[3];
    }
  };

  struct Rgb8  // Pixel<unsigned char, Rgb>: three packed bytes
  {
    std::uint8_t r, g, b;
  };

  template <typename T, int N = 2>
  class ImageView
  {
  public:
    using vector_type = Vector2i;
    using pointer = T*;
    ImageView() = default;
    ImageView(pointer data, const vector_type& sizes) : _d{data}, _s{sizes} {}
    int width() const { return _s(0); }
    int height() const { return _s(1); }
    const vector_type& sizes() const { return _s; }
    T* data() { return _d; }
    const T* data() const { return _d; }
    T& operator()(int x, int y) { return _d[std::size_t(y) * width() + x]; }
    const T& operator()(int x, int y) const { return _d[std::size_t(y) * width() + x]; }

  protected:
    T* _d = nullptr;
    vector_type _s;
  };

  template <typename T, int N = 2>
  class Image : public ImageView<T, N>
  {
  public:
    Image() = default;
    Image(int width, int height) : _store(std::size_t(width) * height)
    {
      this->_d = _store.data();
      this->_s = Vector2i{width, height};
    }
    Image(const Image& o) : ImageView<T, N>{}, _store{o._store}
    {
      this->_d = _store.data();
      this->_s = o._s;
    }
    Image& operator=(const Image& o)
    {
      _store = o._store;
      this->_d = _store.data();
      this->_s = o._s;
      return *this;
    }

  private:
    std::vector<T> _store;
  };

}  // namespace DO::Sara
