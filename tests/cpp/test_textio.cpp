// Host-only check of the readable keypoint format of include/DO/Sara/HipSift.hpp
// (Features/IO.hpp:77-143) on a file the reference wrote
// (examples/Sara/Features/test.dogkey, read at
// examples/Sara/Features/features_read_write_example.cpp:121): read it through
// read_keypoints, then re-emit every record's head - coordinates, the 2 x 2
// shape matrix through the writer's Eigen alignment routine, orientation, type -
// so that tests/test_keypoint_text_pins.py can compare it with the file's bytes.
#include <DO/Sara/HipSift.hpp>

#include <cstdio>
#include <fstream>
#include <iostream>

namespace sara = DO::Sara;

int main(int argc, char** argv)
{
  if (argc < 3)
  {
    std::fprintf(stderr, "usage: test_textio in.dogkey out.txt\n");
    return 2;
  }
  auto features = std::vector<sara::OERegion>{};
  auto descriptors = sara::Tensor_<float, 2>{};
  if (!sara::read_keypoints(features, descriptors, argv[1]))
    return 1;
  std::cout << features.size() << " " << descriptors.cols() << "\n";
  if (features.empty())
    return 1;
  std::cout << "first " << features[0].x() << " " << features[0].y() << " "
            << features[0].orientation << " " << int(features[0].type) << "\n";

  std::ofstream out{argv[2]};
  for (const auto& f : features)
  {
    out << f.x() << ' ' << f.y() << "\n";
    sara::hip_detail::print_eigen_block(out, f.shape_matrix.data(), 2, 2);
    out << "\n" << f.orientation << "\n" << int(f.type) << "\n";
  }
  // descriptors: integers 0..255 in this file
  double sum = 0;
  for (int i = 0; i < descriptors.rows(); ++i)
    for (int j = 0; j < descriptors.cols(); ++j)
      sum += descriptors(i, j);
  std::cout << "descriptor sum " << sum << "\n";
  return 0;
}
