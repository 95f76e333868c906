// MOCK - see ../README.md.  Tensor_<T, 2> of DO/Sara/Core/Tensor.hpp:41-45
// (= MultiArray<T, 2, RowMajor>; resize(rows, cols) MultiArray.hpp:162-166).
#pragma once
#include <DO/Sara/Core/Image.hpp>

namespace DO::Sara {

  template <typename T, int N>
  class Tensor_
  {
  public:
    Tensor_() = default;
    void resize(int rows, int cols)
    {
      _r = rows;
      _c = cols;
      _d.assign(std::size_t(rows) * cols, T{});
    }
    int rows() const { return _r; }
    int cols() const { return _c; }
    std::size_t size() const { return _d.size(); }
    T* data() { return _d.data(); }
    const T* data() const { return _d.data(); }

  private:
    int _r = 0, _c = 0;
    std::vector<T> _d;
  };

}  // namespace DO::Sara
