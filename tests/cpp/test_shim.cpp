// Builds against include/DO/Sara/HipSift.hpp (standalone mode) the way a Sara
// caller would use the reference API (cf. cpp/examples/Sara/FeatureDescriptors/
// sift_example.cpp:36-92 and SfM/Odometry/OdometryPipeline.cpp:82-90), and
// dumps the results for tests/test_gpu_cpp_shim.py to compare with the oracle.
//
//   test_shim <in.f32> <w> <h> <num_octaves_max> <out.bin>
#include <DO/Sara/HipSift.hpp>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <fstream>
#include <iostream>

namespace sara = DO::Sara;

int main(int argc, char** argv)
{
  if (argc < 6)
    return 2;
  const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
  const int noct = std::atoi(argv[4]);
  std::vector<float> buf(size_t(w) * h);
  {
    std::ifstream in(argv[1], std::ios::binary);
    in.read(reinterpret_cast<char*>(buf.data()), buf.size() * sizeof(float));
    if (!in)
      return 3;
  }
  const auto image = sara::ImageView<float>{buf.data(), w, h};

  // Error convention: the scale-count check throws std::runtime_error like
  // DoG.hpp:86-89.
  bool threw = false;
  try
  {
    sara::ComputeDoGExtrema bad{sara::ImagePyramidParams(0, 3)};
  }
  catch (const std::runtime_error&)
  {
    threw = true;
  }
  if (!threw)
    return 4;

  // from_rgb8_to_gray32f: size check + the conversion of gray triples.
  {
    std::vector<sara::Rgb8> rgb(16 * 4);
    for (size_t i = 0; i < rgb.size(); ++i)
      rgb[i] = {std::uint8_t(4 * i), std::uint8_t(255 - i), std::uint8_t(i)};
    std::vector<float> g(16 * 4), g2(8);
    auto src = sara::ImageView<sara::Rgb8>{rgb.data(), 16, 4};
    auto dst = sara::ImageView<float>{g.data(), 16, 4};
    auto bad = sara::ImageView<float>{g2.data(), 4, 2};
    bool caught = false;
    try
    {
      sara::from_rgb8_to_gray32f(src, bad);
    }
    catch (const std::domain_error&)
    {
      caught = true;
    }
    if (!caught)
      return 9;
    sara::from_rgb8_to_gray32f(src, dst);
    for (size_t i = 0; i < rgb.size(); ++i)
    {
      const double want = 0.2125 * (rgb[i].r / 255.0) +
                          0.7154 * (rgb[i].g / 255.0) +
                          0.0721 * (rgb[i].b / 255.0);
      if (g[i] != float(want))
        return 10;
    }
  }

  // 1. the OdometryPipeline call: compute_sift_keypoints(image, params).
  const auto pyr_params = sara::ImagePyramidParams(
      0, 6, std::pow(2.f, 1.f / 3.f), 1, 0.5f, 1.6f, noct);
  const auto keys = sara::compute_sift_keypoints(image, pyr_params);
  const auto& f = sara::features(keys);
  const auto& d = sara::descriptors(keys);
  if (!sara::size_consistency_predicate(keys) || d.cols() != 128)
    return 5;

  // matching the keypoints against themselves: every keypoint whose
  // descriptor is unique finds itself with score 0 in both directions, kept
  // once (AnnMatcher.cpp:239-254).
  if (f.size() >= 3)
  {
    const auto matches = sara::match(keys, keys, 0.6f);
    if (matches.empty())
      return 11;
    for (const auto& m : matches)
      if (m.x_index() != m.y_index() && m.score() != 0.f)
        return 12;
    if (&matches.front().x() != &f[size_t(matches.front().x_index())])
      return 13;
  }

  // readable keypoint format (Features/IO.hpp:77-143): write, read back;
  // the file is also compared byte for byte with the Python writer.
  {
    const std::string txt = std::string(argv[5]) + ".txt";
    if (!sara::write_keypoints(f, d, txt))
      return 14;
    auto f2 = std::vector<sara::OERegion>{};
    auto d2 = sara::Tensor_<float, 2>{};
    if (!sara::read_keypoints(f2, d2, txt) || f2.size() != f.size() ||
        d2.rows() != d.rows() || d2.cols() != 128)
      return 15;
    for (size_t i = 0; i < f.size(); ++i)
    {
      // six significant digits survive the text round trip
      if (std::abs(f2[i].x() - f[i].x()) > 1e-3f * (1.f + std::abs(f[i].x())) ||
          f2[i].type != f[i].type ||
          std::abs(d2(int(i), 5) - d(int(i), 5)) > 1e-3f * (1.f + d(int(i), 5)))
        return 16;
    }
  }

  // RootSIFT post-processing (FeatureDescriptors/RootSIFT.hpp:45-53): every
  // non-zero row has unit L2 norm afterwards.
  {
    auto r = d;
    sara::root_sift(r);
    for (int i = 0; i < r.rows(); ++i)
    {
      double l1 = 0., z = 0.;
      for (int j = 0; j < r.cols(); ++j)
      {
        l1 += d(i, j);
        z += double(r(i, j)) * r(i, j);
      }
      if (l1 > 0. && std::abs(z - 1.) > 1e-5)
        return 17;
    }
  }

  // 2. the functor API with its pyramid accessors.
  auto compute_dogs = sara::ComputeDoGExtrema{pyr_params, 4.f, 0.01f, 10.f, 5, 5};
  auto so = std::vector<sara::Point2i>{};
  const auto extrema = compute_dogs(image, &so);
  const auto& G = compute_dogs.gaussians();
  const auto& D = compute_dogs.diff_of_gaussians();
  if (G.octave_count() != D.octave_count() || G.scale_count_per_octave() != 6 ||
      D.scale_count_per_octave() != 5 || so.size() != extrema.size())
    return 6;
  // D(s) = G(s+1) - G(s) at a few sites, through the pixel getters.
  for (int o = 0; o < G.octave_count(); ++o)
    for (int s = 0; s < 5; ++s)
    {
      const int x = G(s, o).width() / 3, y = G(s, o).height() / 2;
      if (D(x, y, s, o) != G(x, y, s + 1, o) - G(x, y, s, o))
        return 7;
    }
  size_t per_so = 0;
  for (int o = 0; o < D.octave_count(); ++o)
    for (int s = 0; s < D.scale_count_per_octave(); ++s)
      per_so += compute_dogs.extrema(s, o).size();
  if (per_so != extrema.size())
    return 8;

  std::ofstream out(argv[5], std::ios::binary);
  const int32_t n = int32_t(f.size()), ne = int32_t(extrema.size());
  out.write(reinterpret_cast<const char*>(&n), 4);
  out.write(reinterpret_cast<const char*>(&ne), 4);
  out.write(reinterpret_cast<const char*>(f.data()), sizeof(sara::OERegion) * n);
  out.write(reinterpret_cast<const char*>(d.data()), sizeof(float) * 128 * n);
  out.write(reinterpret_cast<const char*>(extrema.data()),
            sizeof(sara::OERegion) * ne);
  // The per-frame call pattern of OdometryPipeline::detect_keypoints
  // (SfM/Odometry/OdometryPipeline.cpp:82-90): the same free function once per
  // frame.  The shim keeps its context (no re-allocation): time per call, and
  // the results stay identical.
  double ms_per_call = 0.;
  {
    const int reps = 30;
    for (int i = 0; i < 3; ++i)
      (void) sara::compute_sift_keypoints(image, pyr_params);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i)
    {
      const auto again = sara::compute_sift_keypoints(image, pyr_params);
      if (sara::features(again).size() != f.size())
        return 14;
      if (i == reps - 1 &&
          std::memcmp(sara::descriptors(again).data(), d.data(),
                      sizeof(float) * 128 * f.size()) != 0)
        return 15;
    }
    ms_per_call = std::chrono::duration<double, std::milli>(
                      std::chrono::steady_clock::now() - t0)
                      .count() /
                  reps;
  }
  // where a call spends its time: the same three steps through the C-ABI
  double ms_submit = 0., ms_collect = 0., ms_fill = 0.;
  {
    sara_sift_params p;
    p.pyramid = sara::hip_detail::to_c(pyr_params);
    p.gauss_truncate = 4.f;
    p.extremum_thres = 0.01f;
    p.edge_ratio_thres = 10.f;
    p.extremum_refinement_iter = 5;
    sara_hip_sift* ctx = sara::hip_detail::cached_context(
        sara::hip_detail::ContextKey{p, image.width(), image.height(), 0});
    using clk = std::chrono::steady_clock;
    const int reps = 30;
    for (int i = 0; i < reps; ++i)
    {
      const auto t0 = clk::now();
      int ticket = -1, total = 0;
      if (sara_hip_sift_submit(ctx, image.data(), 0, 0, 1, image.width(),
                               image.height(), 0, SARA_HIP_STAGE_DESCRIPTOR,
                               &ticket) != SARA_HIP_OK)
        return 16;
      const auto t1 = clk::now();
      const sara_oeregion* pf = nullptr;
      const float* pd = nullptr;
      if (sara_hip_sift_collect(ctx, ticket, &pf, &pd, nullptr, nullptr, &total) !=
          SARA_HIP_OK)
        return 17;
      const auto t2 = clk::now();
      // as compute_sift_keypoints() fills them: one pass, no value-initialisation
      const sara::OERegion* fr = reinterpret_cast<const sara::OERegion*>(pf);
      auto feats = std::vector<sara::OERegion>(fr, fr + total);
      auto desc = sara::Tensor_<float, 2>{};
      desc.assign(pd, total, 128);
      const auto t3 = clk::now();
      ms_submit += std::chrono::duration<double, std::milli>(t1 - t0).count() / reps;
      ms_collect += std::chrono::duration<double, std::milli>(t2 - t1).count() / reps;
      ms_fill += std::chrono::duration<double, std::milli>(t3 - t2).count() / reps;
    }
  }
  std::fprintf(stderr, "phases: submit %.3f ms, collect %.3f ms, containers %.3f ms\n",
               ms_submit, ms_collect, ms_fill);
  std::printf("{\"keypoints\": %d, \"extrema\": %d, \"octaves\": %d, "
              "\"factor1\": %g, \"ms_per_call\": %.4f}\n",
              n, ne, G.octave_count(),
              G.octave_count() > 1 ? G.octave_scaling_factor(1) : 0.f,
              ms_per_call);
  return 0;
}
