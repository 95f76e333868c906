// ORACLE — TEST INFRASTRUCTURE ONLY (see sift_ref.hpp).
// C-ABI over the CPU restatement so tests/ and bench.py's cpu_baseline leg can
// drive it through ctypes.  Never linked into the product library.
#include "sift_ref.hpp"

#include <cstdio>
#include <string>

using namespace sara_ref;

namespace {
  thread_local std::string g_last_error;

  PyramidParams to_params(const int* ip, const float* fp)
  {
    // ip = {first_octave_index, scale_count_per_octave, image_padding_size,
    //       num_octaves_max}; fp = {scale_geometric_factor, scale_camera,
    //       scale_initial}
    PyramidParams p;
    p.first_octave_index = ip[0];
    p.scale_count_per_octave = ip[1];
    p.image_padding_size = ip[2];
    p.num_octaves_max = ip[3];
    p.scale_geometric_factor = fp[0];
    p.scale_camera = fp[1];
    p.scale_initial = fp[2];
    return p;
  }

  Image wrap(const float* data, int w, int h)
  {
    Image I(w, h);
    std::memcpy(I.d.data(), data, sizeof(float) * size_t(w) * h);
    return I;
  }

  Image2 wrap2(const float* data, int w, int h)
  {
    Image2 I(w, h);
    std::memcpy(I.d.data(), data, sizeof(float) * size_t(w) * h * 2);
    return I;
  }
}  // namespace

extern "C" {

const char* ref_last_error() { return g_last_error.c_str(); }

int ref_omp_max_threads()
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void ref_omp_set_threads(int n)
{
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void) n;
#endif
}

// ---- unit-level entry points ------------------------------------------- //

void ref_convolve_array(float* signal, const float* kernel, int signal_size,
                        int kernel_size)
{
  convolve_array(signal, kernel, signal_size, kernel_size);
}

//! Arithmetic variant of make_gaussian_kernel (sift_ref.hpp: kTaps*), applied
//! by every later call of this library; returns the previous kind.
int ref_set_tap_variant(int kind, unsigned seed)
{
  const int before = tap_variant().kind;
  if (kind >= 0 && kind < kTapsVariantCount)
  {
    tap_variant().kind = kind;
    tap_variant().seed = seed;
  }
  return before;
}

float ref_eigen34_pexp(float x) { return eigen34_pexp_lane(x); }
float ref_eigen33_pexp(float x) { return eigen33_pexp_lane(x); }
float ref_eigen_sum_sse(const float* v, int n) { return eigen_sum_sse(v, n); }

int ref_make_gaussian_kernel(float sigma, float gauss_truncate, float* out,
                             int capacity)
{
  const auto k = make_gaussian_kernel(sigma, gauss_truncate);
  if (int(k.size()) > capacity)
    return -int(k.size());
  std::memcpy(out, k.data(), sizeof(float) * k.size());
  return int(k.size());
}

int ref_apply_row_based_filter(const float* src, float* dst, int w, int h,
                               const float* kernel, int ksz)
{
  Image s = wrap(src, w, h), d(w, h);
  apply_row_based_filter(s, d, kernel, ksz);
  std::memcpy(dst, d.d.data(), sizeof(float) * size_t(w) * h);
  return 0;
}

int ref_apply_column_based_filter(const float* src, float* dst, int w, int h,
                                  const float* kernel, int ksz)
{
  Image s = wrap(src, w, h), d(w, h);
  apply_column_based_filter(s, d, kernel, ksz);
  std::memcpy(dst, d.d.data(), sizeof(float) * size_t(w) * h);
  return 0;
}

int ref_apply_gaussian_filter(const float* src, float* dst, int w, int h,
                              float sigma, float gauss_truncate)
{
  Image s = wrap(src, w, h), d(w, h);
  apply_gaussian_filter(s, d, sigma, gauss_truncate);
  std::memcpy(dst, d.d.data(), sizeof(float) * size_t(w) * h);
  return 0;
}

void ref_downscale(const float* src, int w, int h, int fact, float* dst)
{
  const Image d = downscale(wrap(src, w, h), fact);
  std::memcpy(dst, d.d.data(), sizeof(float) * d.d.size());
}

int ref_enlarge(const float* src, int w, int h, float* dst, int dw, int dh)
{
  try
  {
    Image s = wrap(src, w, h), d(dw, dh);
    enlarge(s, d);
    std::memcpy(dst, d.d.data(), sizeof(float) * d.d.size());
    return 0;
  }
  catch (const std::exception& e)
  {
    g_last_error = e.what();
    return 1;
  }
}

void ref_gradient(const float* src, int w, int h, float* gxgy)
{
  const Image s = wrap(src, w, h);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      gradient_at(s, x, y, gxgy[(size_t(y) * w + x) * 2],
                  gxgy[(size_t(y) * w + x) * 2 + 1]);
}

void ref_hessian(const float* src, int w, int h, float* hxx_hxy_hyy)
{
  const Image s = wrap(src, w, h);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
    {
      float* o = &hxx_hxy_hyy[(size_t(y) * w + x) * 3];
      hessian_at(s, x, y, o[0], o[1], o[2]);
    }
}

void ref_gradient_polar(const float* src, int w, int h, float* mag_ori)
{
  const Image2 g = gradient_polar_coordinates(wrap(src, w, h));
  std::memcpy(mag_ori, g.d.data(), sizeof(float) * g.d.size());
}

//! layers = 3 images (s-1, s, s+1) of size w x h.  strict: 0 -> >=/<=,
//! 1 -> >/<.  Returns +1 (max), -1 (min), 0.
int ref_scale_space_extremum(const float* layers, int w, int h, int x, int y,
                             int strict)
{
  Pyramid<Image> P;
  P.reset(1, 3, 1.6f, std::pow(2.f, 1.f / 3.f));
  for (int i = 0; i < 3; ++i)
    P(i, 0) = wrap(layers + size_t(i) * w * h, w, h);
  if (strict)
  {
    if (local_scale_space_extremum(x, y, 1, 0, P, std::greater<float>{}))
      return 1;
    if (local_scale_space_extremum(x, y, 1, 0, P, std::less<float>{}))
      return -1;
  }
  else
  {
    if (local_scale_space_extremum(x, y, 1, 0, P, std::greater_equal<float>{}))
      return 1;
    if (local_scale_space_extremum(x, y, 1, 0, P, std::less_equal<float>{}))
      return -1;
  }
  return 0;
}

int ref_on_edge(const float* img, int w, int h, int x, int y, float edge_ratio)
{
  return on_edge(wrap(img, w, h), x, y, edge_ratio) ? 1 : 0;
}

//! layers: n_layers images of one octave (a DoG octave).  Returns the bool of
//! refine_extremum; pos[3], val in/out as in the reference.
int ref_refine_extremum(const float* layers, int n_layers, int w, int h,
                        float scale_initial, float k, int x, int y, int s,
                        int type, float* pos, float* val, int border_sz,
                        int num_iter)
{
  Pyramid<Image> P;
  P.reset(1, n_layers, scale_initial, k);
  for (int i = 0; i < n_layers; ++i)
    P(i, 0) = wrap(layers + size_t(i) * w * h, w, h);
  try
  {
    return refine_extremum(P, x, y, s, 0, type, pos, *val, border_sz, num_iter)
               ? 1
               : 0;
  }
  catch (const std::exception& e)
  {
    g_last_error = e.what();
    return -1;
  }
}

void ref_orientation_histogram36(const float* grad, int w, int h, float x,
                                 float y, float s, float* hist36)
{
  compute_orientation_histogram<36>(hist36, wrap2(grad, w, h), x, y, s);
}

//! The reference's unit test instantiates N = 24 bins.
void ref_orientation_histogram24(const float* grad, int w, int h, float x,
                                 float y, float s, float* hist24)
{
  compute_orientation_histogram<24>(hist24, wrap2(grad, w, h), x, y, s);
}

void ref_lowe_smooth_histogram36(float* hist36, int iters)
{
  lowe_smooth_histogram<36>(hist36, iters);
}

int ref_dominant_orientations(const float* grad, int w, int h, float x, float y,
                              float sigma, float* out, int capacity,
                              float* hist36_out)
{
  const auto p = dominant_orientations(wrap2(grad, w, h), x, y, sigma, 0.8f,
                                       3.f, 1.5f, hist36_out);
  for (int i = 0; i < int(p.size()) && i < capacity; ++i)
    out[i] = p[i];
  return int(p.size());
}

void ref_sift_descriptor(const float* grad, int w, int h, float x, float y,
                         float s, float theta, int normalize, float* out128)
{
  compute_sift_descriptor(out128, x, y, s, theta, wrap2(grad, w, h),
                          normalize != 0);
}

float ref_oeregion_scale(float scale)
{
  return oeregion_scale(make_oeregion(0.f, 0.f, scale));
}

float ref_rgb8_to_gray32f(unsigned char r, unsigned char g, unsigned char b)
{
  return rgb8_to_gray32f(r, g, b);
}

// ---- descriptor matching (row f2) ---------------------------------------- //
//! -> number of matches (written up to `capacity`), or -1 with ref_last_error.
int ref_compute_matches(const float* desc1, int n1, const float* desc2, int n2,
                        int dim, float sift_ratio_thres, Match* out,
                        int capacity)
{
  try
  {
    const auto m = compute_matches(desc1, n1, desc2, n2, dim, sift_ratio_thres);
    for (int i = 0; i < int(m.size()) && i < capacity; ++i)
      out[i] = m[i];
    return int(m.size());
  }
  catch (const std::exception& e)
  {
    g_last_error = e.what();
    return -1;
  }
}

//! Self-matching constructor.  features: n x 8 floats (x, y, m00, m10, m01, m11,
//! orientation, type).
int ref_compute_self_matches(const float* desc, const float* features, int n,
                             int dim, float sift_ratio_thres,
                             float min_max_metric_dist_thres,
                             float pixel_dist_thres, Match* out, int capacity)
{
  try
  {
    std::vector<MatchFeature> f(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i)
    {
      const float* r = features + size_t(i) * 8;
      f[size_t(i)] = MatchFeature{r[0], r[1], r[2], r[3], r[4], r[5], r[6], int(r[7])};
    }
    const auto m = compute_self_matches(desc, f.data(), n, dim, sift_ratio_thres,
                                        min_max_metric_dist_thres, pixel_dist_thres);
    for (int i = 0; i < int(m.size()) && i < capacity; ++i)
      out[i] = m[i];
    return int(m.size());
  }
  catch (const std::exception& e)
  {
    g_last_error = e.what();
    return -1;
  }
}

int ref_key_proximity(const float* f1, const float* f2, float metric_dist_thres,
                      float pixel_dist_thres)
{
  const MatchFeature a{f1[0], f1[1], f1[2], f1[3], f1[4], f1[5], f1[6], int(f1[7])};
  const MatchFeature b{f2[0], f2[1], f2[2], f2[3], f2[4], f2[5], f2[6], int(f2[7])};
  return KeyProximity{metric_dist_thres, pixel_dist_thres}(a, b) ? 1 : 0;
}

float ref_flann_l2(const float* a, const float* b, int size)
{
  return flann_l2(a, b, size);
}

//! The neighbour lists the matcher ranks (exhaustive_neighbours): the k nearest
//! of every query row in (distance, index) order; idx / dist: nq x k.
void ref_exhaustive_knn(const float* t, int nt, int dim, const float* q, int nq,
                        int k, int* idx, float* dist)
{
  std::vector<std::pair<float, int>> nn;
  for (int i = 0; i < nq; ++i)
  {
    exhaustive_neighbours(q + size_t(i) * dim, t, nt, dim, nn);
    for (int r = 0; r < k && r < nt; ++r)
    {
      dist[size_t(i) * k + r] = nn[size_t(r)].first;
      idx[size_t(i) * k + r] = nn[size_t(r)].second;
    }
  }
}

//! ... and the members of the strict radius search dist < radius[i] in the same
//! order, rows of max_nn entries; count[i] = members.
void ref_exhaustive_radius(const float* t, int nt, int dim, const float* q, int nq,
                           const float* radius, int max_nn, int* idx, float* dist,
                           int* count)
{
  std::vector<std::pair<float, int>> nn;
  for (int i = 0; i < nq; ++i)
  {
    exhaustive_neighbours(q + size_t(i) * dim, t, nt, dim, nn);
    int K = 0;
    while (K < nt && K < max_nn && nn[size_t(K)].first < radius[i])
    {
      dist[size_t(i) * max_nn + K] = nn[size_t(K)].first;
      idx[size_t(i) * max_nn + K] = nn[size_t(K)].second;
      ++K;
    }
    count[i] = K;
  }
}

void ref_root_sift(float* desc, int n, int dim)
{
  for (int i = 0; i < n; ++i)
    root_sift(desc + size_t(i) * dim, dim);
}

//! Detector mode bits (kMode*, sift_ref.hpp); returns the previous value.
int ref_set_detector_mode(int mode)
{
  const int old = detector_mode();
  detector_mode() = mode;
  return old;
}

//! Definiteness rule of refine_extremum (kDef*, sift_ref.hpp); returns the
//! previous value.
int ref_set_definiteness_rule(int rule)
{
  const int old = definiteness_rule();
  definiteness_rule() = rule;
  return old;
}

//! Audit of the three definiteness rules: enable (resets the counters) /
//! read {sites, eigen34 != sylvester, eigen34 != eigen33, not converged}.
void ref_definiteness_audit_enable(int on)
{
  definiteness_audit() = DefinitenessAudit{};
  definiteness_audit().enabled = on != 0;
}
void ref_definiteness_audit_read(long long* out)
{
  const DefinitenessAudit& a = definiteness_audit();
  out[0] = a.sites;
  out[1] = a.eigen34_vs_sylvester;
  out[2] = a.eigen34_vs_eigen33;
  out[3] = a.not_converged;
}

//! Eigenvalues of `count` symmetric 3x3 float matrices (row-major 9 floats
//! each) by the restated SelfAdjointEigenSolver<Matrix3f>; converged[i] = 0/1.
void ref_selfadjoint_eigenvalues3(const float* mats, int count, int rule,
                                  float* lambda, int* converged)
{
  for (int i = 0; i < count; ++i)
  {
    float H[3][3];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        H[r][c] = mats[size_t(i) * 9 + r * 3 + c];
    const bool ok = eigen_selfadjoint_eigenvalues3(H, lambda + size_t(i) * 3, rule);
    if (converged)
      converged[i] = ok ? 1 : 0;
  }
}

//! not_definite_enough3 for `count` matrices: out[i] = 0/1.
void ref_not_definite_enough3(const float* mats, int count, int type, int rule,
                              int* out)
{
  for (int i = 0; i < count; ++i)
  {
    float H[3][3];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        H[r][c] = mats[size_t(i) * 9 + r * 3 + c];
    out[i] = not_definite_enough3(H, type, rule) ? 1 : 0;
  }
}

//! Reduction order of normalize()'s squaredNorm() (0 Eigen packets, 1 left to
//! right); returns the previous value.
int ref_set_squared_norm_order(int order)
{
  const int old = squared_norm_order();
  squared_norm_order() = order;
  return old;
}

float ref_eigen_squared_norm128(const float* h) { return eigen_squared_norm128(h); }

//! The Halide build's classifier on three w x h layers -> int8 map.
void ref_halide_dog_extremum_map(const float* a, const float* b, const float* c,
                                 int w, int h, float edge_ratio,
                                 float extremum_thres, signed char* out)
{
  Image A(w, h), B(w, h), Cc(w, h);
  std::copy(a, a + size_t(w) * h, A.d.begin());
  std::copy(b, b + size_t(w) * h, B.d.begin());
  std::copy(c, c + size_t(w) * h, Cc.d.begin());
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      out[size_t(y) * w + x] = static_cast<signed char>(
          halide_is_dog_extremum(A, B, Cc, x, y, edge_ratio, extremum_thres));
}

// ---- whole-pipeline handle ---------------------------------------------- //

struct ref_sift
{
  SiftResult R;
};

//! ip = {first_octave_index, scale_count_per_octave, image_padding_size,
//!       num_octaves_max}; fp = {scale_geometric_factor, scale_camera,
//!       scale_initial}.  Mirrors compute_sift_keypoints' argument order
//!       (FeatureDetectors/SIFT.hpp:24-33).  stop_after: see sift_ref.hpp.
ref_sift* ref_sift_run(const float* image, int w, int h, const int* ip,
                       const float* fp, float gauss_truncate,
                       float extremum_thres, float edge_ratio_thres,
                       int extremum_refinement_iter, int parallel,
                       int stop_after)
{
  try
  {
    auto* r = new ref_sift;
    compute_sift_keypoints(r->R, wrap(image, w, h), to_params(ip, fp),
                           gauss_truncate, extremum_thres, edge_ratio_thres,
                           extremum_refinement_iter, parallel != 0, stop_after);
    return r;
  }
  catch (const std::exception& e)
  {
    g_last_error = e.what();
    return nullptr;
  }
}

//! gaussian_pyramid() (+ difference_of_gaussians_pyramid() when with_dog)
//! alone, without ComputeDoGExtrema's scale-count check
//! (ImageProcessing/GaussianPyramid.hpp:33-125, GaussianPyramid.cpp:23-51).
ref_sift* ref_pyramid_run(const float* image, int w, int h, const int* ip,
                          const float* fp, float gauss_truncate, int with_dog)
{
  try
  {
    auto* r = new ref_sift;
    r->R.dog.gaussians =
        gaussian_pyramid(wrap(image, w, h), to_params(ip, fp), gauss_truncate);
    if (with_dog && r->R.dog.gaussians.octave_count() > 0)
      r->R.dog.diff_of_gaussians =
          difference_of_gaussians_pyramid(r->R.dog.gaussians);
    return r;
  }
  catch (const std::exception& e)
  {
    g_last_error = e.what();
    return nullptr;
  }
}

void ref_sift_free(ref_sift* r) { delete r; }

int ref_sift_octave_count(const ref_sift* r)
{
  return r->R.dog.gaussians.octave_count();
}

void ref_sift_octave_info(const ref_sift* r, int o, int* w, int* h,
                          float* factor)
{
  const auto& G = r->R.dog.gaussians;
  *w = G(0, o).w;
  *h = G(0, o).h;
  *factor = G.oct_scaling_factors[o];
}

const float* ref_sift_gaussian(const ref_sift* r, int s, int o)
{
  return r->R.dog.gaussians(s, o).d.data();
}

const float* ref_sift_dog(const ref_sift* r, int s, int o)
{
  if (r->R.dog.diff_of_gaussians.octaves.empty())
    return nullptr;
  return r->R.dog.diff_of_gaussians(s, o).d.data();
}

const float* ref_sift_gradient(const ref_sift* r, int s, int o)
{
  if (r->R.nabla_G.octaves.empty())
    return nullptr;
  return r->R.nabla_G(s, o).d.data();
}

int ref_sift_extrema_count(const ref_sift* r) { return int(r->R.extrema.size()); }

//! out_regions: N x 48 B; out_xyso_type: N x 5 ints (x, y, s, o, type).
void ref_sift_extrema(const ref_sift* r, void* out_regions, int* out_xyso_type)
{
  const auto& rec = r->R.dog.records;
  for (size_t i = 0; i < rec.size(); ++i)
  {
    std::memcpy(static_cast<char*>(out_regions) + 48 * i, &rec[i].region, 48);
    out_xyso_type[5 * i + 0] = rec[i].x;
    out_xyso_type[5 * i + 1] = rec[i].y;
    out_xyso_type[5 * i + 2] = rec[i].s;
    out_xyso_type[5 * i + 3] = rec[i].o;
    out_xyso_type[5 * i + 4] = rec[i].type;
  }
}

int ref_sift_keypoint_count(const ref_sift* r)
{
  return int(r->R.features.size());
}

//! out_regions: N x 48 B; out_so: N x 2 ints; out_desc: N x 128 floats (may be
//! null when the run stopped before descriptors).
void ref_sift_keypoints(const ref_sift* r, void* out_regions, int* out_so,
                        float* out_desc)
{
  const auto& f = r->R.features;
  if (out_regions && !f.empty())
    std::memcpy(out_regions, f.data(), 48 * f.size());
  if (out_so && !r->R.so_pairs.empty())
    std::memcpy(out_so, r->R.so_pairs.data(), sizeof(int) * r->R.so_pairs.size());
  if (out_desc && !r->R.descriptors.empty())
    std::memcpy(out_desc, r->R.descriptors.data(),
                sizeof(float) * r->R.descriptors.size());
}

//! times[7] = gaussian pyramid, DoG pyramid, DoG extrema, gradient,
//! orientation, descriptors, total (ms) — the reference's stage names
//! (FeatureDetectors/DoG.cpp:40-83, FeatureDetectors/SIFT.cpp:56-105).
void ref_sift_times(const ref_sift* r, double* times)
{
  const auto& t = r->R.times;
  times[0] = t.gaussian_pyramid_ms;
  times[1] = t.dog_pyramid_ms;
  times[2] = t.dog_extrema_ms;
  times[3] = t.gradient_ms;
  times[4] = t.orientation_ms;
  times[5] = t.descriptors_ms;
  times[6] = t.total_ms;
}

}  // extern "C"
