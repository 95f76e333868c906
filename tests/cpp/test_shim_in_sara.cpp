// The in-Sara build mode of include/DO/Sara/HipSift.hpp
// (-DSARA_HIP_WITH_SARA_HEADERS) compiled against the MOCK of Sara's headers
// in tests/cpp/mock_sara/ (see its README: a syntax / collision check, not
// parity) next to mock declarations of the DO::Sara names Sara itself defines.
// The GPU versions must live in DO::Sara::hip and the two sets must coexist.
//
//   test_shim_in_sara <in.f32> <w> <h> <num_octaves_max> <out.bin>
// writes: int32 n, n x 48 B OERegion, n x 128 float  (same as test_shim's head)
#include <DO/Sara/Colliding.hpp>  // mock: what Sara defines
#include <DO/Sara/HipSift.hpp>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <type_traits>

namespace sara = DO::Sara;

// the shim's results come back in Sara's own types
static_assert(std::is_same<decltype(sara::hip::compute_sift_keypoints(
                               std::declval<sara::ImageView<float>>())),
                           sara::KeypointList<sara::OERegion, float>>::value,
              "hip::compute_sift_keypoints returns Sara's KeypointList");
static_assert(!std::is_same<sara::hip::AnnMatcher, sara::AnnMatcher>::value &&
                  !std::is_same<sara::hip::ComputeDoGExtrema,
                                sara::ComputeDoGExtrema>::value,
              "the GPU classes are distinct types in DO::Sara::hip");

int main(int argc, char** argv)
{
  if (argc < 6)
  {
    std::puts("compiled: in-Sara mode is well-formed");
    return 0;
  }
  const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
  const int noct = std::atoi(argv[4]);
  std::vector<float> buf(size_t(w) * h);
  {
    std::ifstream in(argv[1], std::ios::binary);
    in.read(reinterpret_cast<char*>(buf.data()), buf.size() * sizeof(float));
    if (!in)
      return 3;
  }
  const auto image = sara::ImageView<float>{buf.data(), sara::Vector2i{w, h}};
  const auto params = sara::ImagePyramidParams(0, 6, std::pow(2.f, 1.f / 3.f), 1,
                                               0.5f, 1.6f, noct);
  // Sara's own function is still there (and is the CPU path) ...
  if (sara::size(sara::compute_sift_keypoints(image, params)) != 0)
    return 4;
  // ... the GPU one is one qualification away, same arguments
  const auto keys = sara::hip::compute_sift_keypoints(image, params);
  const auto& f = sara::features(keys);
  const auto& d = sara::descriptors(keys);
  // ComputeDoGExtrema with Sara's Point2i
  sara::hip::ComputeDoGExtrema dog{params, 4.f, 0.01f, 10.f, 5, 5};
  std::vector<sara::Point2i> so;
  const auto extrema = dog(image, &so);
  if (extrema.size() != so.size() || dog.gaussians().octave_count() != noct)
    return 5;
  // matching a list against itself with the reference's defaults
  sara::hip::AnnMatcher self{keys};
  const auto sm = self.compute_matches();
  const auto mm = sara::hip::match(keys, keys, 1.0f);
  if (mm.size() < f.size())  // every key finds itself at distance 0
    return 6;
  std::ofstream out(argv[5], std::ios::binary);
  const std::int32_t n = std::int32_t(f.size());
  out.write(reinterpret_cast<const char*>(&n), 4);
  out.write(reinterpret_cast<const char*>(f.data()), std::streamsize(48) * n);
  out.write(reinterpret_cast<const char*>(d.data()), std::streamsize(512) * n);
  std::printf("{\"keypoints\": %d, \"extrema\": %d, \"self_matches\": %d, "
              "\"matches\": %d}\n",
              n, int(extrema.size()), int(sm.size()), int(mm.size()));
  return out ? 0 : 7;
}
