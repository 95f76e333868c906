// HDF5 keypoint files through the C++ shim (include/DO/Sara/HipSift.hpp with
// SARA_HIP_WITH_HDF5): the reference's test_features_hdf5.cpp:27-73 restated -
// four dummy features written to "0/features", read back, compared - plus the
// descriptor matrix and the overwrite rule.  Host code only.
#define SARA_HIP_WITH_HDF5
#include <DO/Sara/HipSift.hpp>

#include <cstdio>
#include <string>

namespace sara = DO::Sara;

int main(int argc, char** argv)
{
  if (argc < 2)
    return 2;
  const std::string filepath = argv[1];

  auto features = std::vector<sara::OERegion>(4);
  auto descriptors = sara::Tensor_<float, 2>{4, 128};
  for (int i = 0; i < 4; ++i)
  {
    auto& f = features[size_t(i)];
    f.coords[0] = f.coords[1] = float(i);
    for (int j = 0; j < 4; ++j)
      f.shape_matrix[j] = i + 0.5f;
    f.orientation = 30.f * i;
    f.extremum_value = 10.f * i;
    for (int j = 0; j < 128; ++j)
      descriptors(i, j) = float(128 * i + j);
  }
  const auto keys = sara::KeypointList<sara::OERegion, float>{features, descriptors};

  // Write.
  {
    auto h5file = sara::H5File{filepath, sara::H5File::AccTrunc};
    sara::write_keypoints(h5file, "0", keys);
    // a second write without permission fails like Core/HDF5.hpp:266-268
    bool thrown = false;
    try
    {
      sara::write_keypoints(h5file, "0", keys);
    }
    catch (const std::runtime_error& e)
    {
      thrown = std::string(e.what()).find("exists but overwriting is not permitted") !=
               std::string::npos;
    }
    if (!thrown)
      return 3;
    sara::write_keypoints(h5file, "0", keys, true);
    sara::write_keypoints(h5file, "sfm/frame/1", keys);
  }

  // Read.
  {
    auto h5file = sara::H5File{filepath, sara::H5File::AccRdOnly};
    for (const char* group : {"0", "sfm/frame/1"})
    {
      const auto back = sara::read_keypoints(h5file, group);
      const auto& f = sara::features(back);
      const auto& d = sara::descriptors(back);
      if (f.size() != 4 || d.rows() != 4 || d.cols() != 128)
        return 4;
      for (int i = 0; i < 4; ++i)
      {
        const auto& r = f[size_t(i)];
        if (r.coords[0] != float(i) || r.coords[1] != float(i) ||
            r.orientation != 30.f * i || r.extremum_value != 10.f * i)
          return 5;
        for (int j = 0; j < 4; ++j)
          if (r.shape_matrix[j] != i + 0.5f)
            return 6;
        for (int j = 0; j < 128; ++j)
          if (d(i, j) != float(128 * i + j))
            return 7;
      }
    }
    bool thrown = false;
    try
    {
      sara::read_keypoints(h5file, "missing");
    }
    catch (const std::runtime_error&)
    {
      thrown = true;
    }
    if (!thrown)
      return 8;
  }
  std::puts("ok");
  return 0;
}
