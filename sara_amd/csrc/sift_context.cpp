// The context behind the C-ABI (include/sara_hip_sift.h): HBM buffer ownership,
// create / destroy / options / reserve.  Mirrors what
//   compute_sift_keypoints     FeatureDetectors/SIFT.cpp:27-108
//   ComputeDoGExtrema::op()    FeatureDetectors/DoG.cpp:23-87
// allocate per call, once per context (sift_host.hpp has the map of the host side).
#include "sift_host.hpp"

using namespace sara_hip;
using namespace sara_hip::host;

namespace sara_hip { namespace host {

  //! Everything whose size is a multiple of the per-frame list capacity
  //! `c->cap` (sara_hip_sift_reserve re-allocates exactly this set).
  sara_hip_status alloc_lists(sara_hip_sift* c)
  {
    const size_t rows = size_t(c->max_batch) * c->cap;
#define TRY_ST(expr)                                                           \
  do                                                                           \
  {                                                                            \
    const sara_hip_status st_ = (expr);                                        \
    if (st_ != SARA_HIP_OK)                                                    \
      return st_;                                                              \
  } while (0)
    c->cand.cap = c->cap;
    TRY_ST(c->alloc(c->cand.key, rows));
    TRY_ST(c->alloc(c->cand.data, rows));
    TRY_ST(c->alloc(c->cand.order, rows));
    TRY_ST(c->alloc(c->cand.skey, rows));
    TRY_ST(c->alloc(c->cand.sdata, rows));
    TRY_ST(c->alloc(c->d_grouped, rows));
    c->sites.cap = 4 * c->cap;
    TRY_ST(c->alloc(c->sites.key, size_t(c->max_batch) * c->sites.cap));
    TRY_ST(c->alloc(c->sites.nb, size_t(c->max_batch) * c->sites.cap * kSiteNb));
    TRY_ST(c->alloc(c->ori.peak_count, rows));
    TRY_ST(c->alloc(c->ori.peak_theta, rows * kMaxPeaks));
    TRY_ST(c->alloc(c->ori.offset, rows));
    TRY_ST(c->alloc(c->ori.record, rows));
    TRY_ST(c->alloc(c->ori.item, rows));
    for (int k = 0; k < (c->has_slot1 ? 2 : 1); ++k)
    {
      TRY_ST(c->alloc(c->d_feat_s[k], rows));
      TRY_ST(c->alloc(c->d_so_s[k], rows * 2));
      TRY_ST(c->alloc(c->d_desc_s[k], rows * 128));
    }
    c->d_feat = c->d_feat_s[c->write_slot];
    c->d_so = c->d_so_s[c->write_slot];
    c->d_desc = c->d_desc_s[c->write_slot];
    TRY_ST(c->alloc(c->d_ex_regions, rows));
    TRY_ST(c->alloc(c->d_ex_xyso, rows * 5));
#undef TRY_ST
    return SARA_HIP_OK;
  }

  //! Bytes alloc_lists() allocates per list entry (= per keypoint of capacity)
  //! and frame.
  size_t list_bytes_per_entry(const sara_hip_sift* c)
  {
    const size_t cand = 8 + 16 + 4 + 8 + 16 + 4;              // key data order skey sdata grouped
    const size_t sites = 4 * (8 + sizeof(float) * kSiteNb);   // 4 sites per entry
    const size_t ori = 4 + 4 * kMaxPeaks + 4 + sizeof(KeypointRecord) + sizeof(KeypointItem);
    const size_t results = (c->has_slot1 ? 2 : 1) * (sizeof(sara_oeregion) + 8 + 512);
    const size_t extrema = sizeof(sara_oeregion) + 20;
    return cand + sites + ori + results + extrema;
  }

  //! Frees what alloc_lists() allocated (pointers that are still null are
  //! skipped).
  void free_lists(sara_hip_sift* c)
  {
    auto drop = [&](auto*& p) {
      if (!p)
        return;
      auto it = std::find(c->allocations.begin(), c->allocations.end(),
                          static_cast<void*>(p));
      if (it != c->allocations.end())
        c->allocations.erase(it);
      (void) hipFree(p);
      p = nullptr;
    };
    drop(c->cand.key);
    drop(c->cand.data);
    drop(c->cand.order);
    drop(c->cand.skey);
    drop(c->cand.sdata);
    drop(c->d_grouped);
    drop(c->sites.key);
    drop(c->sites.nb);
    drop(c->ori.peak_count);
    drop(c->ori.peak_theta);
    drop(c->ori.offset);
    drop(c->ori.record);
    drop(c->ori.item);
    for (int k = 0; k < 2; ++k)
    {
      drop(c->d_feat_s[k]);
      drop(c->d_so_s[k]);
      drop(c->d_desc_s[k]);
    }
    drop(c->d_ex_regions);
    drop(c->d_ex_xyso);
  }

  //! Taps of the initial blur and of the S - 1 incremental blurs under the
  //! context's tap arithmetic; nullptr or what is wrong.
  const char* compute_taps(sara_hip_sift* c)
  {
    const sara_pyramid_params& pyr = c->pyr;
    const float k = pyr.scale_geometric_factor;
    if (c->max_sched.init_blur)
    {
      const float trunc = pyr.first_octave_index > 0 ? c->gauss_truncate : 4.f;
      if (!to_taps(gaussian_taps(c->max_sched.init_sigma, trunc, c->tap_arithmetic),
                   c->init_taps))
        return "initial Gaussian needs more than 113 taps";
      c->have_init_taps = true;
    }
    c->taps.resize(c->S);
    float sigma_s_1 = pyr.scale_initial;
    for (int s = 1; s < c->S; ++s)
    {
      const float ks = k * sigma_s_1;
      const double sigma = std::sqrt(double(ks * ks - sigma_s_1 * sigma_s_1));
      if (!to_taps(gaussian_taps(static_cast<float>(sigma), 4.f, c->tap_arithmetic),
                   c->taps[s]))
        return "a pyramid Gaussian needs more than 113 taps";
      sigma_s_1 *= k;
    }
    return nullptr;
  }

  sara_hip_status create_impl(const sara_pyramid_params& pyr, float gauss_truncate,
                              float extremum_thres, float edge_ratio_thres,
                              int img_padding_sz, int refine_iters, int max_w,
                              int max_h, int max_batch, int max_keypoints,
                              int device, sara_hip_sift** out)
  {
    if (!out)
      return fail(SARA_HIP_INVALID_PARAMS, "out is null");
    *out = nullptr;
    const sara_hip_status v = validate(pyr, img_padding_sz);
    if (v != SARA_HIP_OK)
      return v;
    if (max_w < 2 || max_h < 2 || max_batch < 1)
      return fail(SARA_HIP_INVALID_PARAMS, "max_width/max_height/max_batch");
    if (max_w >= (1 << 20) || max_h >= (1 << 20))
      return fail(SARA_HIP_INVALID_PARAMS, "image side must be < 2^20");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      return fail(SARA_HIP_NO_DEVICE,
                  "no HIP device: the SIFT front-end has no CPU fallback");
    if (device < 0 || device >= ndev)
      return fail(SARA_HIP_INVALID_PARAMS, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    std::lock_guard<std::recursive_mutex> runtime_lock(runtime_mutex());

    auto* c = new sara_hip_sift;
    c->device = device;
    c->pyr = pyr;
    c->gauss_truncate = gauss_truncate;
    c->extremum_thres = extremum_thres;
    c->edge_ratio = edge_ratio_thres;
    c->img_padding = img_padding_sz;
    c->refine_iters = refine_iters;
    c->max_w = max_w;
    c->max_h = max_h;
    c->max_batch = max_batch;
    c->S = pyr.scale_count_per_octave;
    c->max_sched = make_schedule(pyr, max_w, max_h);
    if (c->max_sched.num_octaves > 16)
    {
      delete c;
      return fail(SARA_HIP_INVALID_PARAMS, "more than 16 octaves");
    }
    if (c->max_sched.downscale_index >= c->S)
    {
      delete c;
      return fail(SARA_HIP_INVALID_PARAMS,
                  "downscale index floor(log 2 / log k) >= scale count");
    }
    c->cap = max_keypoints > 0
                 ? max_keypoints
                 : std::max(1024, int((size_t(c->max_sched.base_w) *
                                       size_t(c->max_sched.base_h)) /
                                      128));

    auto cleanup = [&](sara_hip_status st) {
      sara_hip_sift_destroy(c);
      return st;
    };
#define TRY_ST(expr)                                                           \
  do                                                                           \
  {                                                                            \
    const sara_hip_status st_ = (expr);                                        \
    if (st_ != SARA_HIP_OK)                                                    \
      return cleanup(st_);                                                     \
  } while (0)
#define TRY_HIP(expr)                                                          \
  do                                                                           \
  {                                                                            \
    const hipError_t e_ = (expr);                                              \
    if (e_ != hipSuccess)                                                      \
      return cleanup(fail(SARA_HIP_RUNTIME_ERROR,                              \
                          std::string(#expr) + ": " + hipGetErrorString(e_))); \
  } while (0)

    auto make_stream = [&](hipStream_t* st) -> hipError_t {
      return pooled_stream_acquire(c->device, false, st);
    };
    TRY_HIP(make_stream(&c->own_stream));
  for (auto& e : c->ev)
      TRY_HIP(hipEventCreate(&e));
    for (int o = 0; o < 16; ++o)
    {
      TRY_HIP(hipEventCreateWithFlags(&c->oct_ready[o], hipEventDisableTiming));
      TRY_HIP(hipEventCreateWithFlags(&c->oct_done[o], hipEventDisableTiming));
      TRY_HIP(hipEventCreateWithFlags(&c->scan_done[o], hipEventDisableTiming));
      if (o > 0 && o < c->max_sched.num_octaves)
        TRY_HIP(make_stream(&c->oct_stream[o]));
    }
    if (const char* e = getenv("SARA_HIP_STREAMS"))
      c->multi_stream = std::string(e) != "1";
    if (const char* e = getenv("SARA_HIP_SIDE_GRADIENT"))
      c->side_gradient = std::string(e) != "0";
    if (const char* e = getenv("SARA_HIP_OCTAVE_PIPELINE"))
      c->octave_pipeline = std::string(e) != "0" ? 1 : 0;
    TRY_HIP(make_stream(&c->aux_stream));
    for (int k = 0; k < 3; ++k)
    {
      TRY_HIP(make_stream(&c->filler_stream[k]));
      TRY_HIP(hipEventCreateWithFlags(&c->filler_done[k], hipEventDisableTiming));
    }
    TRY_HIP(hipEventCreateWithFlags(&c->aux_fork, hipEventDisableTiming));
    TRY_HIP(hipEventCreateWithFlags(&c->aux_join, hipEventDisableTiming));
    if (const char* e = getenv("SARA_HIP_GRAPH"))
      c->use_graph = std::string(e) != "0";
    if (const char* e = getenv("SARA_HIP_GRAPH_MAX_BATCH"))
      c->graph_max_batch = atoi(e);

    // ---- taps and tables (host arithmetic as in GaussianPyramid.hpp:106-121)
    const float k = pyr.scale_geometric_factor;
    if (const char* msg = compute_taps(c))
      return cleanup(fail(SARA_HIP_INVALID_PARAMS, msg));
    std::vector<double> oriw;
    for (int s = 0; s < c->S; ++s)
    {
      // ImagePyramid.hpp:316-319 / Orientation.hpp:105-108,127.
      const float sigma = static_cast<float>(std::pow(double(k), double(s)) *
                                             double(pyr.scale_initial));
      c->h_tab.sigma[s] = sigma;
      c->h_tab.sigma_d[s] = std::pow(double(k), double(s)) * double(pyr.scale_initial);
      const float sw = sigma * 1.5f;
      c->h_tab.ori_sigma[s] = sw;
      const int R = static_cast<int>(std::round(sw * 3.f));
      c->h_tab.ori_radius[s] = R;
      c->h_tab.ori_woff[s] = int(oriw.size());
      const bool used = s >= 1 && s <= c->S - 3;
      if (used)
        for (int d2 = 0; d2 <= 2 * R * R; ++d2)
          oriw.push_back(std::exp(double(float(-d2) / (2.f * sw * sw))));
    }
    orientation_bin_thresholds(c->h_tab.ori_bin_thr);
    TRY_ST(c->alloc(c->d_tab, 1));
    TRY_HIP(hipMemcpy(c->d_tab, &c->h_tab, sizeof(ScaleTable),
                      hipMemcpyHostToDevice));
    TRY_ST(c->alloc(c->d_oriw, oriw.size()));
    c->n_oriw = int(oriw.size());
    if (!oriw.empty())
      TRY_HIP(hipMemcpy(c->d_oriw, oriw.data(), sizeof(double) * oriw.size(),
                        hipMemcpyHostToDevice));
    TRY_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_grad),
                          sizeof(GradPyramidView)));
    std::memset(c->h_grad, 0, sizeof(GradPyramidView));
    TRY_ST(c->alloc(c->d_grad, 1));
    TRY_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_counts),
                          sizeof(int) * counters_read(max_batch)));

    // ---- HBM: pyramids [frame][scale][h][w] per octave
    const int no = c->max_sched.num_octaves;
    c->G.assign(no, nullptr);
    c->GR.assign(no, nullptr);
    c->CM.assign(no, nullptr);
    for (int o = 0; o < no; ++o)
    {
      const size_t pl = size_t(c->max_sched.oct[o].w) * c->max_sched.oct[o].h;
      TRY_ST(c->alloc(c->G[o], pl * c->S * max_batch));
      TRY_ST(c->alloc(c->GR[o], pl * c->S * max_batch * 2));
      const size_t cpl = size_t((c->max_sched.oct[o].w + 15) / 16) *
                         ((c->max_sched.oct[o].h + 15) / 16);
      TRY_ST(c->alloc(c->CM[o], cpl * c->S * max_batch));
    }
    TRY_ST(c->alloc(c->d_dog_plane, size_t(c->max_sched.base_w) *
                                        std::max(c->max_sched.base_h, 1)));
    TRY_ST(c->alloc(c->d_input, size_t(max_w) * max_h * max_batch));
    if (pyr.first_octave_index > 0)
      TRY_ST(c->alloc(c->d_full, size_t(max_w) * max_h * max_batch));

    // ---- candidate / keypoint lists
    // the four per-frame counters share one block: one memset per detect()
    // + 1 for frame_offset[batch], + 1 arrival counter of the peak scan
    // padded to whole 256-byte blocks: the runtime then zeroes it with one
    // fill kernel instead of an aligned part and a tail
    TRY_ST(c->alloc(c->d_counters, counters_padded(max_batch) + 64));
    TRY_HIP(hipMemset(c->d_counters, 0, sizeof(int) * (counters_padded(max_batch) + 64)));
    c->d_epoch = reinterpret_cast<unsigned*>(c->d_counters + counters_padded(max_batch));
    c->cand.count = c->d_counters;
    c->sites.count = c->d_counters + max_batch;
    c->ori.kp_count = c->d_counters + 2 * size_t(max_batch);
    c->ori.frame_offset = c->d_counters + 3 * size_t(max_batch);  // max_batch + 1
    c->cand.error = c->sites.error = c->d_counters + error_flag_index(max_batch);
    {
      // one bucket per image row of every plane of the largest schedule
      int total = 0;
      for (int o = 0; o < c->max_sched.num_octaves; ++o)
        total += c->S * c->max_sched.oct[o].h;
      c->bucket_stride = total + 1;
      TRY_ST(c->alloc(c->d_bucket_hist, size_t(max_batch) * c->bucket_stride));
      TRY_ST(c->alloc(c->d_bucket_cursor, size_t(max_batch) * c->bucket_stride));
    }
    TRY_ST(c->alloc(c->d_ex_offset, size_t(max_batch) + 1));
    TRY_ST(alloc_lists(c));
#undef TRY_ST
#undef TRY_HIP
    *out = c;
    return SARA_HIP_OK;
  }

  sara_hip_status require_result(const sara_hip_sift* ctx, sara_hip_stage need)
  {
    if (!ctx)
      return fail(SARA_HIP_INVALID_PARAMS, "null context");
    if (!ctx->has_result)
      return fail(SARA_HIP_NOT_READY, "no detect() has run on this context");
    if (ctx->last_stage < need)
      return fail(SARA_HIP_NOT_READY,
                  "the last detect() stopped before the requested stage");
    return SARA_HIP_OK;
  }

  //! A list counter the step did not zero (seen once, with a graph
  //! captured from one stream on the ROCm 7.0 runtime, before the counters were
  //! cleared by a kernel of the library): a kernel met a negative counter and
  //! raised the flag, or a counter is negative now.  The lists of such a step
  //! are incomplete - the call fails instead of returning them.
  bool counters_corrupt(sara_hip_sift* c, const int* h, int mb, int batch, int lists,
                        unsigned expected_step)
  {
    if (h[error_flag_index(mb)] != 0)
      return true;
    for (int l = 0; l < lists; ++l)
      for (int b = 0; b < batch; ++b)
        if (h[size_t(l) * mb + b] < 0)
          return true;
    // a counter that was not cleared holds the previous step's
    // POSITIVE count and passes the test above.  The clearing kernel stamps the
    // block with the number of the step; the host knows which step it is
    // reading (after a failed enqueue it does not: the next read-back adopts the
    // device's number).
    const unsigned stamp = unsigned(h[step_stamp_index(mb)]);
    if (!c->epoch_synced)
    {
      c->epoch_host += stamp - expected_step;
      c->epoch_synced = true;
      return false;
    }
    return stamp != expected_step;
  }
  sara_hip_status corrupt_counters_error()
  {
    return fail(SARA_HIP_RUNTIME_ERROR,
                "the keypoint-list counters were not cleared before this batch "
                "ran (negative counter or a stale step stamp): the lists are "
                "incomplete");
  }

  //! Remembers what list capacity a batch asked for: per frame the largest of
  //! the extremum count, the keypoint count and a quarter of the classified
  //! sites (their list holds 4 * max_keypoints).  A list that overflowed
  //! starves the ones behind it, so the figure is a lower bound then.
  void note_required(sara_hip_sift* c, const int* h_ex, const int* h_sites,
                     const int* h_kp, int batch)
  {
    int need = 0;
    for (int b = 0; b < batch; ++b)
    {
      if (h_ex)
        need = std::max(need, h_ex[b]);
      if (h_sites)
        need = std::max(need, (h_sites[b] + 3) / 4);
      if (h_kp)
        need = std::max(need, h_kp[b]);
    }
    c->required_cap = need;
  }


  sara_hip_status select_device(int device)
  {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      return fail(SARA_HIP_NO_DEVICE,
                  "no HIP device: the SIFT front-end has no CPU fallback");
    if (device < 0 || device >= ndev)
      return fail(SARA_HIP_INVALID_PARAMS, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    return SARA_HIP_OK;
  }
}}  // namespace sara_hip::host

extern "C" {

sara_hip_status sara_hip_sift_create(const sara_sift_params* params, int max_width,
                                     int max_height, int max_batch,
                                     int max_keypoints, int device,
                                     sara_hip_sift** out)
{
  if (!params)
    return fail(SARA_HIP_INVALID_PARAMS, "null params");
  // FeatureDetectors/SIFT.cpp:45-51: the 5th constructor argument of
  // ComputeDoGExtrema is img_padding_sz, so extremum_refinement_iter becomes
  // the border padding and the iteration count keeps its default of 5.
  return create_impl(params->pyramid, params->gauss_truncate,
                     params->extremum_thres, params->edge_ratio_thres,
                     params->extremum_refinement_iter, 5, max_width, max_height,
                     max_batch, max_keypoints, device, out);
}

sara_hip_status sara_hip_sift_create_dog(const sara_pyramid_params* pyramid,
                                         float gauss_truncate,
                                         float extremum_thres,
                                         float edge_ratio_thres,
                                         int img_padding_sz,
                                         int extremum_refinement_iter,
                                         int max_width, int max_height,
                                         int max_batch, int max_keypoints,
                                         int device, sara_hip_sift** out)
{
  if (!pyramid)
    return fail(SARA_HIP_INVALID_PARAMS, "null params");
  return create_impl(*pyramid, gauss_truncate, extremum_thres, edge_ratio_thres,
                     img_padding_sz, extremum_refinement_iter, max_width,
                     max_height, max_batch, max_keypoints, device, out);
}

sara_hip_status sara_hip_sift_destroy(sara_hip_sift* c)
{
  if (!c)
    return SARA_HIP_OK;
  (void) hipSetDevice(c->device);
  if (c->last_stream)
    (void) hipStreamSynchronize(c->last_stream);
  if (c->own_stream)
    (void) hipStreamSynchronize(c->own_stream);
  // Graph calls belong to the launcher thread.  NOT under runtime_mutex(): the
  // launcher may be inside another thread's replay, which takes that mutex - a
  // caller that waits for the launcher while holding it would deadlock.
  if (c->graph_exec_s[0] || c->graph_s[0] || c->graph_exec_s[1] || c->graph_s[1])
    graph_launcher().run([&] {
      (void) hipSetDevice(c->device);
      std::lock_guard<std::recursive_mutex> lock(runtime_mutex());
      for (int k = 0; k < 2; ++k)
      {
        if (c->graph_exec_s[k])
          (void) hipGraphExecDestroy(c->graph_exec_s[k]);
        if (c->graph_s[k])
          (void) hipGraphDestroy(c->graph_s[k]);
        c->graph_exec_s[k] = nullptr;
        c->graph_s[k] = nullptr;
      }
    });
  std::lock_guard<std::recursive_mutex> runtime_lock(runtime_mutex());
  for (void* p : c->allocations)
    (void) hipFree(p);
  if (c->h_grad)
    (void) hipHostFree(c->h_grad);
  if (c->h_counts)
    (void) hipHostFree(c->h_counts);
  for (auto& e : c->ev)
    if (e)
      (void) hipEventDestroy(e);
  for (auto& r : c->launch_rec)
  {
    if (r.begin)
      (void) hipEventDestroy(r.begin);
    if (r.end)
      (void) hipEventDestroy(r.end);
  }
  for (int o = 0; o < 16; ++o)
  {
    pooled_stream_release(c->device, false, c->oct_stream[o]);
    if (c->oct_ready[o])
      (void) hipEventDestroy(c->oct_ready[o]);
    if (c->oct_done[o])
      (void) hipEventDestroy(c->oct_done[o]);
    if (c->scan_done[o])
      (void) hipEventDestroy(c->scan_done[o]);
  }
  for (int k = 0; k < 2; ++k)
  {
    if (c->stage_ready[k])
      (void) hipEventDestroy(c->stage_ready[k]);
    if (c->stage_free[k])
      (void) hipEventDestroy(c->stage_free[k]);
  }
  pooled_stream_release(c->device, false, c->copy_stream);
  for (int k = 0; k < 2; ++k)
  {
    sara_hip_sift::RingSlot& r = c->ring[k];
    if (r.done)
      (void) hipEventDestroy(r.done);
    if (r.h_counters)
      (void) hipHostFree(r.h_counters);
    if (r.h_feat)
      (void) hipHostFree(r.h_feat);
    if (r.h_desc)
      (void) hipHostFree(r.h_desc);
    if (r.h_so)
      (void) hipHostFree(r.h_so);
  }
  pooled_stream_release(c->device, true, c->d2h_stream);
  for (int k = 0; k < 3; ++k)
  {
    pooled_stream_release(c->device, false, c->filler_stream[k]);
    if (c->filler_done[k])
      (void) hipEventDestroy(c->filler_done[k]);
  }
  pooled_stream_release(c->device, false, c->aux_stream);
  if (c->aux_fork)
    (void) hipEventDestroy(c->aux_fork);
  if (c->aux_join)
    (void) hipEventDestroy(c->aux_join);
  pooled_stream_release(c->device, false, c->own_stream);
  delete c;
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_sift_set_option(sara_hip_sift* c, int option, int value)
{
  if (!c)
    return fail(SARA_HIP_INVALID_PARAMS, "null context");
  switch (option)
  {
  case SARA_HIP_OPT_ALL_GRADIENT_SCALES:
    c->all_gradient_scales = value != 0;
    c->graph_stage_s[0] = c->graph_stage_s[1] = -1;  // the captured launch sequence depends on it
    return SARA_HIP_OK;
  case SARA_HIP_OPT_STAGE_TIMERS:
    c->timers = value != 0;
    return SARA_HIP_OK;
  case SARA_HIP_OPT_SINGLE_STREAM:
    c->multi_stream = value == 0;
    c->graph_stage_s[0] = c->graph_stage_s[1] = -1;
    return SARA_HIP_OK;
  case SARA_HIP_OPT_LAUNCH_TIMERS:
    c->launch_timers = value != 0;
    c->launch_count = 0;
    return SARA_HIP_OK;
  case SARA_HIP_OPT_ROOT_SIFT:
    c->root_sift = value != 0;
    c->graph_stage_s[0] = c->graph_stage_s[1] = -1;
    return SARA_HIP_OK;
  case SARA_HIP_OPT_SIGNED_EXTREMUM_TYPE:
    c->signed_type = value != 0;
    c->graph_stage_s[0] = c->graph_stage_s[1] = -1;
    return SARA_HIP_OK;
  case SARA_HIP_OPT_FMA_BLUR:
    c->fma_blur = value != 0;
    c->graph_stage_s[0] = c->graph_stage_s[1] = -1;
    return SARA_HIP_OK;
  case SARA_HIP_OPT_GRAPH_REPLAY:
    c->use_graph = value != 0;
    c->graph_stage_s[0] = c->graph_stage_s[1] = -1;
    return SARA_HIP_OK;
  case SARA_HIP_OPT_KERNEL_SELECTION:
  {
    KernelSelection k;  // the shipped defaults
    switch (value)
    {
    case SARA_HIP_SELECT_ENVIRONMENT: k = environment_selection(); break;
    case SARA_HIP_SELECT_SHIPPED: break;
    case SARA_HIP_SELECT_FORCED_MARCH:
      k.march_min_pixels = 0;
      k.strip_group = 8;
      break;
    case SARA_HIP_SELECT_TILED:
      k.blur_march = false;
      k.feature_march = false;
      break;
    case SARA_HIP_SELECT_TILED_BLUR: k.blur_march = false; break;
    default: return fail(SARA_HIP_INVALID_PARAMS, "unknown kernel selection");
    }
    c->sel = k;
    c->graph_stage_s[0] = c->graph_stage_s[1] = -1;  // launches are captured
    return SARA_HIP_OK;
  }
  case SARA_HIP_OPT_TILE_GEOMETRY:
    if (value < 0 || value > 3)
      return fail(SARA_HIP_INVALID_PARAMS, "tile geometry is 0 (auto) .. 3");
    c->sel.tile_geometry = value;
    c->graph_stage_s[0] = c->graph_stage_s[1] = -1;
    return SARA_HIP_OK;
  case SARA_HIP_OPT_MARCH_WAVES:
    if (value != 0 && value < 64)
      return fail(SARA_HIP_INVALID_PARAMS, "waves per marching launch >= 64");
    c->sel.march_waves = value ? value : KernelSelection{}.march_waves;
    c->sel.march2_waves = value ? value : KernelSelection{}.march2_waves;
    c->graph_stage_s[0] = c->graph_stage_s[1] = -1;
    return SARA_HIP_OK;
  case SARA_HIP_OPT_MARCH2_WAVES:
    if (value != 0 && value < 64)
      return fail(SARA_HIP_INVALID_PARAMS, "waves per marching launch >= 64");
    c->sel.march2_waves = value ? value : KernelSelection{}.march2_waves;
    c->graph_stage_s[0] = c->graph_stage_s[1] = -1;
    return SARA_HIP_OK;
  case SARA_HIP_OPT_TAP_ARITHMETIC:
  {
    if (value < SARA_HIP_TAPS_LIBM_SERIAL || value > SARA_HIP_TAPS_EIGEN33_SSE2)
      return fail(SARA_HIP_INVALID_PARAMS, "unknown tap arithmetic");
    if (c->last_stream)
      HIP_TRY(hipStreamSynchronize(c->last_stream));
    const int before = c->tap_arithmetic;
    c->tap_arithmetic = value;
    if (const char* msg = compute_taps(c))
    {
      c->tap_arithmetic = before;
      (void) compute_taps(c);
      return fail(SARA_HIP_INVALID_PARAMS, msg);
    }
    c->graph_stage_s[0] = c->graph_stage_s[1] = -1;  // taps are kernel arguments
    return SARA_HIP_OK;
  }
  case SARA_HIP_OPT_DOWNSCALE_AT_DOUBLE_SIGMA:
  {
    const bool on = value != 0;
    if (make_schedule(c->pyr, c->max_sched.base_w, c->max_sched.base_h, on)
            .downscale_index >= c->S)
      return fail(SARA_HIP_INVALID_PARAMS,
                  "downscale index round(log 2 / log k) >= scale count");
    if (c->last_stream)
      HIP_TRY(hipStreamSynchronize(c->last_stream));
    c->downscale_at_double_sigma = on;
    c->max_sched = make_schedule(c->pyr, c->max_w, c->max_h, on);  // same sizes
    c->cur_w = c->cur_h = -1;  // rebuild the schedule on the next detect
    c->graph_stage_s[0] = c->graph_stage_s[1] = -1;
    return SARA_HIP_OK;
  }
  default:
    return fail(SARA_HIP_INVALID_PARAMS, "unknown option");
  }
}

sara_hip_status sara_hip_sift_capacity(const sara_hip_sift* c, int* max_keypoints,
                                       int* required)
{
  if (!c)
    return fail(SARA_HIP_INVALID_PARAMS, "null context");
  if (max_keypoints)
    *max_keypoints = c->cap;
  if (required)
    *required = c->required_cap;
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_sift_reserve(sara_hip_sift* c, int max_keypoints)
{
  if (!c)
    return fail(SARA_HIP_INVALID_PARAMS, "null context");
  if (max_keypoints <= c->cap)
    return SARA_HIP_OK;  // the lists never shrink
  if (c->ring[0].pending || c->ring[1].pending)
    return fail(SARA_HIP_NOT_READY,
                "reserve() with a batch in flight: collect() its ticket first");
  if (max_keypoints > INT_MAX / 4 - 1)  // sites.cap = 4 * cap is an int
    return fail(SARA_HIP_CAPACITY_EXCEEDED, "max_keypoints too large");
  HIP_TRY(hipSetDevice(c->device));
  std::lock_guard<std::recursive_mutex> runtime_lock(runtime_mutex());
  for (hipStream_t st : {c->last_stream, c->own_stream, c->aux_stream,
                         c->d2h_stream, c->copy_stream})
    if (st)
      HIP_TRY(hipStreamSynchronize(st));
  const int old_cap = c->cap;
  {
    // Refuse what cannot fit BEFORE touching anything: the lists take
    // list_bytes_per_entry() bytes per keypoint and frame (the old ones are
    // freed first), and a request of hundreds of gigabytes that fails half-way
    // would first have taken most of the device's memory from everybody else.
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const size_t per = list_bytes_per_entry(c) * size_t(c->max_batch);
    const size_t have = per * size_t(old_cap);
    const size_t want = per * size_t(max_keypoints);
    if (want > free_b + have)
      return fail(SARA_HIP_CAPACITY_EXCEEDED,
                  "reserve(): not enough device memory for max_keypoints (" +
                      std::to_string(want >> 20) + " MiB of lists, " +
                      std::to_string((free_b + have) >> 20) + " MiB available)");
  }
  free_lists(c);
  c->cap = max_keypoints;
  sara_hip_status st = alloc_lists(c);
  if (st != SARA_HIP_OK)
  {
    // out of HBM: back to the old lists, which fitted before.  The failed
    // hipMalloc also left its code in the thread's sticky last-error slot,
    // where the next hipGetLastError() after a launch would find it
    (void) hipGetLastError();
    free_lists(c);
    c->cap = old_cap;
    const std::string why = g_error;
    if (alloc_lists(c) != SARA_HIP_OK)
      return fail(SARA_HIP_RUNTIME_ERROR,
                  "reserve(): the keypoint lists could not be re-allocated; "
                  "the context is unusable (" + why + ")");
    return fail(SARA_HIP_CAPACITY_EXCEEDED,
                "reserve(): not enough device memory for max_keypoints (" + why +
                    ")");
  }
  // the list pointers and capacities are kernel arguments baked into a
  // captured graph: capture again on the next detect()
  c->graph_stage_s[0] = c->graph_stage_s[1] = -1;
  c->has_result = false;
  return SARA_HIP_OK;
}

}  // extern "C"
