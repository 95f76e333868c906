// MOCK - see README.md.  The names Sara itself defines in DO::Sara on this
// path, with the reference's signatures: FeatureDetectors/SIFT.hpp:24-33,
// FeatureDetectors/DoG.hpp:72-165, FeatureMatching/AnnMatcher.hpp:32-86,
// SfM/Helpers/KeypointMatching.hpp, ImageProcessing/FastColorConversion.hpp:
// 22-23.  Including this next to HipSift.hpp (in-Sara mode) proves that the
// shim defines none of them again: its versions are DO::Sara::hip::*.
// The bodies mark the CPU path so that the test can tell the two apart.
#pragma once
#include <DO/Sara/Features/KeypointList.hpp>
#include <DO/Sara/ImageProcessing/ImagePyramid.hpp>
#include <DO/Sara/Match/Match.hpp>

namespace DO::Sara {

  inline auto compute_sift_keypoints(
      const ImageView<float>&, const ImagePyramidParams& = ImagePyramidParams(),
      float = 4.f, float = 0.01f, float = 10.f, int = 5, bool = false)
      -> KeypointList<OERegion, float>
  {
    return {};  // "Sara's CPU SIFT"
  }

  class ComputeDoGExtrema
  {
  public:
    ComputeDoGExtrema(const ImagePyramidParams& = ImagePyramidParams(), float = 4.f,
                      float = 0.01f, float = 10.f, int = 1, int = 5)
    {
    }
    std::vector<OERegion> operator()(const ImageView<float>&,
                                     std::vector<Point2i>* = nullptr)
    {
      return {};
    }
  };

  class AnnMatcher
  {
  public:
    AnnMatcher(const KeypointList<OERegion, float>&,
               const KeypointList<OERegion, float>&, float = 1.2f)
    {
    }
    AnnMatcher(const KeypointList<OERegion, float>&, float = 1.2f, float = 0.5f,
               float = 10.f)
    {
    }
    std::vector<Match> compute_matches() { return {}; }
  };

  inline auto match(const KeypointList<OERegion, float>&,
                    const KeypointList<OERegion, float>&, float)
      -> std::vector<Match>
  {
    return {};
  }

  inline auto from_rgb8_to_gray32f(const ImageView<Rgb8>&, ImageView<float>&) -> void {}

}  // namespace DO::Sara
