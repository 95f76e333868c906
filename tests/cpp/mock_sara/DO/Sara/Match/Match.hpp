// MOCK - see ../README.md.  DO/Sara/Match/Match.hpp:25-174: constructor
// argument order, rank(), indices, equality by keypoint VALUE (:161-164).
#pragma once
#include <DO/Sara/Features/KeypointList.hpp>

#include <limits>

namespace DO::Sara {

  class Match
  {
  public:
    enum class Direction : std::uint8_t
    {
      SourceToTarget,
      TargetToSource
    };
    Match() = default;
    Match(const OERegion* x, const OERegion* y,
          float score = std::numeric_limits<float>::max(),
          Direction matching_dir = Direction::SourceToTarget, int x_index = -1,
          int y_index = -1)
      : _x{x}, _y{y}, _xi{x_index}, _yi{y_index}, _score{score}, _dir{matching_dir}
    {
    }
    const OERegion& x() const { return *_x; }
    const OERegion& y() const { return *_y; }
    int x_index() const { return _xi; }
    int y_index() const { return _yi; }
    int rank() const { return _rank; }
    int& rank() { return _rank; }
    float score() const { return _score; }
    Direction matching_direction() const { return _dir; }
    bool operator==(const Match& m) const { return x() == m.x() && y() == m.y(); }

  private:
    const OERegion* _x = nullptr;
    const OERegion* _y = nullptr;
    int _xi = -1, _yi = -1, _rank = -1;
    float _score = std::numeric_limits<float>::max();
    Direction _dir = Direction::SourceToTarget;
  };

}  // namespace DO::Sara
