// MOCK - see ../README.md.  DO/Sara/Features/KeypointList.hpp:35-96 and the
// OERegion it pulls in (Features/Feature.hpp:40-179): same members in the same
// order, hence the same 48-byte layout as sara_oeregion.
#pragma once
#include <DO/Sara/Core/Tensor.hpp>

#include <tuple>
#include <vector>

namespace DO::Sara {

  class OERegion
  {
  public:
    enum class Type : std::uint8_t
    {
      Harris, HarAff, HarLap, FAST, SUSAN, DoG, LoG, DoH, MSER, HesAff, HesLap,
      Undefined
    };
    enum class ExtremumType : std::int8_t
    {
      Min = -1, Saddle = 0, Max = 1, Undefined = -2
    };
    OERegion() = default;
    float x() const { return coords(0); }
    float y() const { return coords(1); }
    const Point2f& center() const { return coords; }
    bool operator==(const OERegion& o) const
    {
      return coords == o.coords && shape_matrix == o.shape_matrix &&
             orientation == o.orientation && type == o.type;
    }
    Point2f coords;
    Matrix2f shape_matrix;
    float orientation{0};
    float extremum_value{0};
    Type type{Type::Undefined};
    ExtremumType extremum_type{ExtremumType::Undefined};
  };

  template <typename F, typename T>
  using KeypointList = std::tuple<std::vector<F>, Tensor_<T, 2>>;

  template <typename F, typename T>
  inline auto features(const KeypointList<F, T>& keys) -> const std::vector<F>&
  {
    return std::get<0>(keys);
  }
  template <typename F, typename T>
  inline auto descriptors(const KeypointList<F, T>& keys) -> const Tensor_<T, 2>&
  {
    return std::get<1>(keys);
  }
  template <typename F, typename T>
  inline auto size(const KeypointList<F, T>& keys)
  {
    return descriptors(keys).rows();
  }
  template <typename F, typename T>
  inline auto size_consistency_predicate(const KeypointList<F, T>& keys)
  {
    return int(features(keys).size()) == descriptors(keys).rows();
  }

}  // namespace DO::Sara
