// detect(): the launch sequence of one batch - upload, Gaussian pyramid, extrema,
// polar gradients, orientations, descriptors - on the context's streams, as
// plain launches or as a captured HIP graph (batches <= 8).  Mirrors the control
// flow of compute_sift_keypoints (FeatureDetectors/SIFT.cpp:27-108) and
// gaussian_pyramid (ImageProcessing/GaussianPyramid.hpp:33-125), batched over
// frames and with every stage resident in HBM.
#include "sift_host.hpp"

using namespace sara_hip;
using namespace sara_hip::host;

extern "C" {

sara_hip_status sara_hip_sift_detect(sara_hip_sift* c, const float* images,
                                     size_t frame_stride, int batch, int width,
                                     int height, int images_on_device,
                                     sara_hip_stage last_stage, void* hip_stream)
{
  if (!c || !images)
    return fail(SARA_HIP_INVALID_PARAMS, "null context or images");
  const ScopedSelection selection_of_this_context(&c->sel);
  if (batch < 1 || batch > c->max_batch)
    return fail(SARA_HIP_CAPACITY_EXCEEDED, "batch exceeds max_batch");
  if (width < 2 || height < 2)
    return fail(SARA_HIP_INVALID_PARAMS, "image smaller than 2x2");
  if (width > c->max_w || height > c->max_h)
    return fail(SARA_HIP_CAPACITY_EXCEEDED,
                "image larger than the context's max_width/max_height");
  if (last_stage < SARA_HIP_STAGE_PYRAMID || last_stage > SARA_HIP_STAGE_DESCRIPTOR)
    return fail(SARA_HIP_INVALID_PARAMS, "last_stage");
  if (frame_stride == 0)
    frame_stride = size_t(width) * height;
  if (frame_stride < size_t(width) * height)
    return fail(SARA_HIP_SIZE_MISMATCH, "frame_stride < width*height");

  HIP_TRY(hipSetDevice(c->device));
  hipStream_t stream =
      hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
  if (c->last_stream && c->last_stream != stream)
    HIP_TRY(hipStreamSynchronize(c->last_stream));

  const bool dims_changed = (width != c->cur_w || height != c->cur_h);
  if (dims_changed)
  {
    if (c->last_stream)
      HIP_TRY(hipStreamSynchronize(c->last_stream));
    c->cur = make_schedule(c->pyr, width, height, c->downscale_at_double_sigma);
    c->cur_w = width;
    c->cur_h = height;
    {
      RowBuckets& rb = c->row_buckets;
      std::memset(&rb, 0, sizeof(rb));
      int at = 0;
      for (int o = 0; o < c->cur.num_octaves && o < 16; ++o)
        for (int sidx = 0; sidx < c->S; ++sidx)
        {
          rb.base[o * kMaxScales + sidx] = at;
          at += c->cur.oct[o].h;
        }
      rb.total = at;
      rb.stride = c->bucket_stride;
      // the fused counting sort writes rb.total + 1 ints per frame into rows
      // of bucket_stride: the current schedule's buckets must be a subset of
      // the largest schedule's (they are for every image <= max_width x
      // max_height; checked, not assumed)
      if (rb.total >= c->bucket_stride || c->cur.num_octaves > 16)
      {
        c->cur_w = c->cur_h = -1;
        return fail(SARA_HIP_CAPACITY_EXCEEDED,
                    "the image's pyramid has more rows than the context's "
                    "largest schedule");
      }
    }
    GradPyramidView& gv = *c->h_grad;
    std::memset(&gv, 0, sizeof(gv));
    gv.octaves = c->cur.num_octaves;
    for (int o = 0; o < c->cur.num_octaves; ++o)
    {
      gv.base[o] = c->GR[o];
      gv.w[o] = c->cur.oct[o].w;
      gv.h[o] = c->cur.oct[o].h;
      gv.plane[o] = size_t(gv.w[o]) * gv.h[o];
      gv.frame_stride[o] = gv.plane[o] * 2 * c->S;
      gv.factor[o] = c->cur.oct[o].factor;
      gv.cmax[o] = c->CM[o];
      gv.cw[o] = (gv.w[o] + 15) / 16;
      gv.ch[o] = (gv.h[o] + 15) / 16;
      gv.cmax_frame_stride[o] = size_t(gv.cw[o]) * gv.ch[o] * c->S;
    }
    HIP_TRY(hipMemcpyAsync(c->d_grad, c->h_grad, sizeof(GradPyramidView),
                           hipMemcpyHostToDevice, stream));
  }
  c->last_stream = stream;
  c->cur_batch = batch;
  c->last_stage = last_stage;
  c->has_result = false;
  std::fill(std::begin(c->ev_recorded), std::end(c->ev_recorded), false);
  static const bool debug_sync = getenv("SARA_HIP_DEBUG_SYNC") != nullptr;
  const Schedule& sc = c->cur;
  const int S = c->S;
  const size_t in_plane = size_t(width) * height;

  // Graph replay: own stream, small batch, no stage timers inside a capture.
  // (up to 8 frames of any size; 9 .. graph_max_batch frames while the call
  // stays in the launch-bound regime - 16 x 1080p - where the replay still wins)
  const bool graph_sized =
      batch <= std::min(c->graph_max_batch, 8) ||
      (batch <= c->graph_max_batch && size_t(width) * height * batch <= (size_t(40) << 20));
  const bool graph_mode = c->use_graph && !c->graph_broken && !hip_stream &&
                          graph_sized && !debug_sync &&
                          (!graphs_need_one_thread() || first_graph_thread());
  const bool multi_stream = c->multi_stream;
  const bool side_gradient = c->side_gradient;
  const bool timing = c->timers && !graph_mode;
  // 8-bit gray frames not converted yet (detect_u8): the first blur of the
  // pyramid reads them directly when it is the marching blur of octave 0 and
  // no graph is replayed (a captured graph has the float source baked in);
  // otherwise they are converted into d_input now.
  const unsigned char* gray8 = c->gray8_src;
  const size_t gray8_stride = c->gray8_stride;
  c->gray8_src = nullptr;
  bool gray8_fused = gray8 && !graph_mode && c->pyr.first_octave_index == 0 &&
                     sc.init_blur && images_on_device && !c->fma_blur;
  if (gray8 && !gray8_fused)
  {
    launch_u8_to_gray32f(gray8, gray8_stride, 1, c->d_input, in_plane, in_plane,
                         batch, stream);
    HIP_TRY(hipGetLastError());
  }
  auto mark = [&](int i) -> hipError_t {
    if (debug_sync)
    {
      std::fprintf(stderr, "[sara_hip] stage mark %d: syncing...\n", i);
      const hipError_t e = hipStreamSynchronize(stream);
      std::fprintf(stderr, "[sara_hip] stage mark %d: %s\n", i,
                   hipGetErrorString(e));
      if (e != hipSuccess)
        return e;
    }
    if (!timing)
      return hipSuccess;
    c->ev_recorded[i] = true;
    return hipEventRecord(c->ev[i], stream);
  };

  if (graph_mode && c->timers)
  {
    c->ev_recorded[0] = true;  // total only: ev[0] .. ev[TOTAL] around the graph
    HIP_TRY(hipEventRecord(c->ev[0], stream));
  }
  HIP_TRY(mark(0));
  // ---- upload -------------------------------------------------------------
  const float* src = images;
  size_t src_stride = frame_stride;
  bool src_in_place = false;
  if (!images_on_device)
  {
    if (frame_stride == in_plane)  // contiguous frames: one linear copy
      HIP_TRY(hipMemcpyAsync(c->d_input, images, in_plane * sizeof(float) * batch,
                             hipMemcpyHostToDevice, stream));
    else
      HIP_TRY(hipMemcpy2DAsync(c->d_input, in_plane * sizeof(float), images,
                               frame_stride * sizeof(float),
                               in_plane * sizeof(float), batch,
                               hipMemcpyHostToDevice, stream));
    src = c->d_input;
    src_stride = in_plane;
  }
  else if (graph_mode && images != c->d_input && c->graph_inplace && !gray8)
  {
    // the graph reads the caller's frames where they are (see graph_inplace)
    src_in_place = true;
  }
  else if (graph_mode && images != c->d_input)
  {
    // the graph's first kernel reads a fixed address
    if (frame_stride == in_plane)
      HIP_TRY(hipMemcpyAsync(c->d_input, images, in_plane * sizeof(float) * batch,
                             hipMemcpyDeviceToDevice, stream));
    else
      HIP_TRY(hipMemcpy2DAsync(c->d_input, in_plane * sizeof(float), images,
                               frame_stride * sizeof(float),
                               in_plane * sizeof(float), batch,
                               hipMemcpyDeviceToDevice, stream));
    src = c->d_input;
    src_stride = in_plane;
  }
  HIP_TRY(mark(1));

  // SARA_HIP_OPT_LAUNCH_TIMERS (plain launches only): an event pair per launch
  const bool time_launches = c->launch_timers && !graph_mode;
  c->launch_count = 0;
  auto launch_begin = [&](int o, int s, int ntaps, size_t pixels,
                          hipStream_t st) -> int {
    if (!time_launches)
      return -1;
    if (size_t(c->launch_count) >= c->launch_rec.size())
    {
      sara_hip_sift::LaunchRecord r;
      if (hipEventCreate(&r.begin) != hipSuccess || hipEventCreate(&r.end) != hipSuccess)
        return -1;
      c->launch_rec.push_back(r);
    }
    sara_hip_sift::LaunchRecord& r = c->launch_rec[size_t(c->launch_count)];
    r.octave = o;
    r.scale = s;
    r.taps = ntaps;
    r.pixels = (long long) pixels;
    (void) hipEventRecord(r.begin, st);
    return c->launch_count++;
  };
  auto launch_end = [&](int rec, hipStream_t st) {
    if (rec >= 0)
      (void) hipEventRecord(c->launch_rec[size_t(rec)].end, st);
  };

  auto enqueue = [&]() -> sara_hip_status {
  // also on the launcher thread, where a graph capture runs this lambda
  const ScopedSelection selection_of_this_context(&c->sel);

  const bool want_gradients = last_stage >= SARA_HIP_STAGE_GRADIENT;
  const bool side = side_gradient && want_gradients && !debug_sync;
  // see SiftContext::octave_pipeline
  const bool pipe = multi_stream && sc.num_octaves > 1 &&
                    last_stage >= SARA_HIP_STAGE_EXTREMA && !debug_sync &&
                    (!want_gradients || side) &&
                    (c->octave_pipeline < 0 ? graph_mode : c->octave_pipeline != 0);
  // the stream the extrema .. descriptor stages are enqueued on
  hipStream_t tail = stream;

  // polar gradients of one octave (the planes the later stages read)
  auto enqueue_gradient = [&](int o, hipStream_t gs) -> sara_hip_status {
    const int s_lo = c->all_gradient_scales ? 0 : 1;
    const int s_n = c->all_gradient_scales ? S : S - 3;
    const int w = sc.oct[o].w, h = sc.oct[o].h;
    const size_t pl = size_t(w) * h;
    const size_t cpl = size_t((w + 15) / 16) * ((h + 15) / 16);
    if (gradient_polar_needs_zeroed_cmax(c->G[o] + pl * s_lo, pl * S,
                                         c->GR[o] + pl * 2 * s_lo, pl * 2 * S,
                                         w, h, batch))
      HIP_TRY(hipMemsetAsync(c->CM[o], 0, cpl * S * batch * sizeof(unsigned), gs));
    launch_gradient_polar(c->G[o] + pl * s_lo, pl * S, c->GR[o] + pl * 2 * s_lo,
                          pl * 2 * S, w, h, s_n, batch, gs,
                          c->CM[o] + cpl * s_lo, cpl * S);
    return SARA_HIP_OK;
  };
  ExtremaParams ep;
  ep.extremum_thres = c->extremum_thres;
  ep.edge_ratio_thres = c->edge_ratio;
  ep.img_padding_sz = c->img_padding;
  ep.refine_iters = c->refine_iters;
  ep.scale_geometric_factor = c->pyr.scale_geometric_factor;
  ep.signed_type = c->signed_type ? 1 : 0;
  // extremum scan of one octave
  auto enqueue_scan = [&](int o, hipStream_t ss) -> sara_hip_status {
    OctaveView dv;  // the Gaussian octave; DoG layers are formed on the fly
    dv.base = c->G[o];
    dv.w = sc.oct[o].w;
    dv.h = sc.oct[o].h;
    dv.scales = S;
    dv.plane = size_t(dv.w) * dv.h;
    dv.frame_stride = dv.plane * S;
    // the Halide-branch classifier looks at every pixel, whatever the padding
    if (c->signed_type || (dv.w > 2 * c->img_padding && dv.h > 2 * c->img_padding))
      launch_extrema_scan(dv, o, batch, ep, c->d_tab, c->cand, c->sites, ss);
    return SARA_HIP_OK;
  };
  if (pipe)  // the scans start before the pyramid is complete
    launch_zero_counters(c->d_counters, counters_padded(c->max_batch), c->d_epoch,
                         int(step_stamp_index(c->max_batch)), tail);

  bool linear = false;  // the fork-free schedule of one-frame calls ran (below)
  // ---- Gaussian pyramid + fused DoG ---------------------------------------
  if (sc.num_octaves > 0)
  {
    const size_t pl0 = size_t(sc.oct[0].w) * sc.oct[0].h;
    float* G00 = c->G[0];
    const size_t g_stride0 = pl0 * S;
    if (c->pyr.first_octave_index < 0)
    {
      launch_enlarge(src, src_stride, width, height, G00, g_stride0, sc.oct[0].w,
                     sc.oct[0].h, batch, stream);
    }
    else if (c->pyr.first_octave_index > 0)
    {
      const float* blurred = src;
      size_t bstride = src_stride;
      if (sc.init_blur)
      {
        launch_gaussian_blur(src, src_stride, c->d_full, in_plane, nullptr, 0,
                             width, height, batch, c->init_taps, stream, nullptr,
                             0, c->fma_blur);
        blurred = c->d_full;
        bstride = in_plane;
      }
      launch_scale(blurred, bstride, width, height, G00, g_stride0, sc.oct[0].w,
                   sc.oct[0].h, batch, stream);
    }
    else if (sc.init_blur)
    {
      bool done = false;
      const int rec = launch_begin(0, 0, c->init_taps.size, pl0 * batch, stream);
      if (gray8_fused)
      {
        done = launch_gaussian_blur_gray8(gray8, gray8_stride, G00, g_stride0, width,
                                          height, batch, c->init_taps, stream);
        if (!done)  // shape / radius the marching kernel does not take
          launch_u8_to_gray32f(gray8, gray8_stride, 1, c->d_input, in_plane,
                               in_plane, batch, stream);
      }
      if (!done)
        launch_gaussian_blur(src, src_stride, G00, g_stride0, nullptr, 0, width,
                             height, batch, c->init_taps, stream, nullptr, 0,
                             c->fma_blur);
      launch_end(rec, stream);
    }
    else
    {
      launch_copy_planes(src, src_stride, G00, g_stride0, pl0, batch, stream);
    }

    // nothing reads the caller's / staged frames beyond this point
    if (c->consumed_event && !graph_mode)
    {
      HIP_TRY(hipEventRecord(c->consumed_event, stream));
      c->consumed_recorded = true;
    }

    // Octave o+1 starts from G(downscale_index, o): its chain runs on its
    // own stream as soon as that plane exists and is joined at the end.
    const bool ms = multi_stream && sc.num_octaves > 1;
    const int dsi = sc.downscale_index;
    const int last = sc.num_octaves - 1;
    bool base_ready = true;  // G(0, o) already written by the previous octave
    // blur G(s-1, o) -> G(s, o); the one that produces G(downscale_index, o)
    // also emits its nearest-neighbour half, i.e. G(0, o+1), on the fast path
    auto enqueue_blur = [&](int o, int s, hipStream_t st) {
      const int w = sc.oct[o].w, h = sc.oct[o].h;
      const size_t pl = size_t(w) * h;
      const size_t gs = pl * S;
      float* dec = nullptr;
      size_t dec_stride = 0;
      if (o < last && s == dsi)
      {
        dec = c->G[o + 1];
        dec_stride = size_t(sc.oct[o + 1].w) * sc.oct[o + 1].h * S;
      }
      const int rec = launch_begin(o, s, c->taps[s].size, pl * batch, st);
      const bool fused = launch_gaussian_blur(
          c->G[o] + pl * (s - 1), gs, c->G[o] + pl * s, gs, nullptr, 0, w, h,
          batch, c->taps[s], st, dec, dec_stride, c->fma_blur);
      launch_end(rec, st);
      if (dec)
        base_ready = fused;
    };
    auto enqueue_base = [&](int o, hipStream_t st) {
      // G(0, o) from G(downscale_index, o-1) when no blur has written it
      if (o > 0 && !base_ready)
      {
        const int pw = sc.oct[o - 1].w, ph = sc.oct[o - 1].h;
        const size_t ppl = size_t(pw) * ph;
        const int rec = launch_begin(o, 0, 0,
                                     size_t(sc.oct[o].w) * sc.oct[o].h * batch, st);
        launch_scale(c->G[o - 1] + ppl * dsi, ppl * S, pw, ph, c->G[o],
                     size_t(sc.oct[o].w) * sc.oct[o].h * S, sc.oct[o].w,
                     sc.oct[o].h, batch, st);
        launch_end(rec, st);
      }
      base_ready = false;
    };
    auto enqueue_blurs = [&](int o, int s_lo, int s_hi, hipStream_t st) {
      for (int s = s_lo; s <= s_hi; ++s)
        enqueue_blur(o, s, st);
    };
    // One frame per call that stops at the keypoint sites (the reference's
    // own call pattern, SfM/Odometry/OdometryPipeline.cpp:82-90): the
    // fork-free schedule (round 6).  The graph runtime starts a node whose
    // predecessor sits on another hardware queue only when that queue has
    // drained what the graph put there before it (measured on four forked
    // layouts, docs/experiments.md): every fork of the spine layout below
    // costs tens of microseconds.  This schedule has none: same-depth blurs of
    // different octaves are ONE launch (gaussian_blur_level_kernel: 24 blur
    // nodes -> S - 1 + (octaves - 1) dsi), then the scans of all octaves in
    // one launch - one chain on one queue (0.186 ms against 0.193 ms for one
    // 1080p frame).  With gradients and the per-keypoint stages behind it the
    // forked layout wins (0.261 against 0.281 ms: the gradients of octave 0
    // overlap the later octaves' blurs there), so those calls keep it.
    // Conditions: every blur is tiled-kernel sized and has one of the level
    // kernel's radii, no level has more members than a launch takes, default
    // scan kernels.
    bool levels = pipe && graph_mode && selection().level_merge && !want_gradients &&
                  dsi >= 1 && last >= 1 && last < kScanMultiMax && !c->fma_blur && S == 6 &&
                  !c->signed_type && selection().feature_march &&
                  (S - 1 + dsi - 1) / dsi <= kBlurLevelMaxMembers &&
                  (!selection().blur_march ||
                   size_t(sc.oct[0].w) * sc.oct[0].h * batch <
                       selection().march_min_pixels);
    for (int s = 1; s < S && levels; ++s)
      levels = blur_level_radius_ok(c->taps[s].size);
    for (int o = 0; o <= last && levels; ++o)
      levels = sc.oct[o].w >= 4 && sc.oct[o].h >= 2;
    if (levels)
    {
      linear = true;
      // blur s of octave o runs at level o * dsi + s: the level that writes
      // G(dsi, o) also writes its half, G(0, o + 1)
      const int n_levels = last * dsi + S - 1;
      for (int L = 1; L <= n_levels; ++L)
      {
        BlurLevelBlur mem[kBlurLevelMaxMembers];
        int n = 0;
        for (int o = 0; o <= last; ++o)
        {
          const int s = L - o * dsi;
          if (s < 1 || s > S - 1)
            continue;
          const int w = sc.oct[o].w, h = sc.oct[o].h;
          const size_t pl = size_t(w) * h;
          BlurLevelBlur& m = mem[n++];
          m.src = c->G[o] + pl * (s - 1);
          m.dst = c->G[o] + pl * s;
          m.src_stride = m.dst_stride = pl * S;
          m.dec = nullptr;
          m.dec_stride = 0;
          if (o < last && s == dsi)
          {
            m.dec = c->G[o + 1];
            m.dec_stride = size_t(sc.oct[o + 1].w) * sc.oct[o + 1].h * S;
          }
          m.w = w;
          m.h = h;
          m.taps = &c->taps[s];
        }
        if (!launch_blur_level(mem, n, batch, tail))
          return fail(SARA_HIP_RUNTIME_ERROR, "level blur refused its members");
      }
      OctaveView views[kScanMultiMax];
      int octs[kScanMultiMax];
      int n_scan = 0;
      for (int o = 0; o <= last && last_stage >= SARA_HIP_STAGE_EXTREMA; ++o)
      {
        OctaveView dv;
        dv.base = c->G[o];
        dv.w = sc.oct[o].w;
        dv.h = sc.oct[o].h;
        dv.scales = S;
        dv.plane = size_t(dv.w) * dv.h;
        dv.frame_stride = dv.plane * S;
        if (dv.w > 2 * c->img_padding && dv.h > 2 * c->img_padding)
        {
          views[n_scan] = dv;
          octs[n_scan++] = o;
        }
      }
      if (n_scan > 0 &&
          !launch_extrema_scan_multi(views, octs, n_scan, batch, ep, c->sites, tail))
        return fail(SARA_HIP_RUNTIME_ERROR, "multi-octave scan refused an octave");
    }
    else if (pipe)
    {
      // Small batches are bound by the chain of dependent launches, and a
      // dependency that crosses hardware queues costs ~12 us against ~0 on
      // one queue.  The longest chain (the spine) - the blurs up to
      // G(downscale_index, o) of every octave, the whole last octave, its
      // scan, and then the per-keypoint stages - is enqueued on `stream`; the
      // rest of octave o (remaining blurs, scan, gradients) forks to
      // oct_stream[o + 1].
      // Capture order matters under graph replay: ROCm 7.2 hands the graph's
      // nodes to the queues in a depth-first order that follows each node's
      // first captured successor, and puts the k-th successor on queue
      // (queue of the node) + k - 1.  Octave 0's side chain (the heaviest) is
      // therefore captured BEFORE the spine goes on: it keeps queue 0 and is
      // in it by the time octave 0's third blur ends, the spine hops to queue
      // 1 once and stays there; filler nodes (4-byte memsets of spare
      // counters) in front of octave 1's .. side chains push each of them to
      // a queue of its own.  With plain streams the same order simply works.
      hipStream_t side0 = c->oct_stream[1];
      enqueue_base(0, stream);
      enqueue_blurs(0, 1, dsi, stream);
      HIP_TRY(hipEventRecord(c->oct_ready[0], stream));
      // the last octave's gradients go behind the side chain of octave last-2
      // (done early, and not the queue finish_sites is waiting for)
      const int grad_last_side = std::max(0, last - 2);
      auto enqueue_side = [&](int o, hipStream_t so) -> sara_hip_status {
        enqueue_blurs(o, dsi + 1, S - 1, so);
        const sara_hip_status sst = enqueue_scan(o, so);
        if (sst != SARA_HIP_OK)
          return sst;
        HIP_TRY(hipEventRecord(c->scan_done[o], so));
        if (want_gradients)
        {
          const sara_hip_status gst = enqueue_gradient(o, so);
          if (gst != SARA_HIP_OK)
            return gst;
        }
        return SARA_HIP_OK;
      };
      {
        HIP_TRY(hipStreamWaitEvent(side0, c->oct_ready[0], 0));
        const sara_hip_status st0 = enqueue_side(0, side0);
        if (st0 != SARA_HIP_OK)
          return st0;
      }
      // the spine
      for (int o = 1; o <= last; ++o)
      {
        enqueue_base(o, tail);
        const int s_hi = o == last ? S - 1 : dsi;
        enqueue_blurs(o, 1, s_hi, tail);
        if (o < last)
          HIP_TRY(hipEventRecord(c->oct_ready[o], tail));
      }
      HIP_TRY(hipEventRecord(c->aux_fork, tail));  // the last octave's planes
      {
        const sara_hip_status sst = enqueue_scan(last, tail);
        if (sst != SARA_HIP_OK)
          return sst;
      }
      if (want_gradients && grad_last_side == 0)
      {
        HIP_TRY(hipStreamWaitEvent(side0, c->aux_fork, 0));
        const sara_hip_status lst = enqueue_gradient(last, side0);
        if (lst != SARA_HIP_OK)
          return lst;
      }
      HIP_TRY(hipEventRecord(c->oct_done[0], side0));
      for (int o = 1; o < last; ++o)
      {
        hipStream_t so = c->oct_stream[o + 1];
        int fillers = 0;
        if (graph_mode)
          for (; fillers < last - 1 - o && fillers < 3; ++fillers)
          {
            hipStream_t fs = c->filler_stream[fillers];
            HIP_TRY(hipStreamWaitEvent(fs, c->oct_ready[o], 0));
            HIP_TRY(hipMemsetAsync(
                c->d_counters + counters_padded(c->max_batch) - 1 - fillers, 0,
                sizeof(int), fs));
            HIP_TRY(hipEventRecord(c->filler_done[fillers], fs));
          }
        {
          HIP_TRY(hipStreamWaitEvent(so, c->oct_ready[o], 0));
          const sara_hip_status sto = enqueue_side(o, so);
          if (sto != SARA_HIP_OK)
            return sto;
        }
        if (want_gradients && o == grad_last_side)
        {
          HIP_TRY(hipStreamWaitEvent(so, c->aux_fork, 0));
          const sara_hip_status lst = enqueue_gradient(last, so);
          if (lst != SARA_HIP_OK)
            return lst;
        }
        for (int k = 0; k < fillers; ++k)  // the filler streams join here
          HIP_TRY(hipStreamWaitEvent(so, c->filler_done[k], 0));
        HIP_TRY(hipEventRecord(c->oct_done[o], so));
      }
      for (int o = 0; o < last; ++o)
        HIP_TRY(hipStreamWaitEvent(tail, c->scan_done[o], 0));
    }
    else
    {
      for (int o = 0; o <= last; ++o)
      {
        hipStream_t so = (ms && o > 0) ? c->oct_stream[o] : stream;
        if (o > 0 && ms)
          HIP_TRY(hipStreamWaitEvent(so, c->oct_ready[o - 1], 0));
        enqueue_base(o, so);
        if (ms && dsi == 0 && o < last)
          HIP_TRY(hipEventRecord(c->oct_ready[o], so));
        for (int s = 1; s < S; ++s)
        {
          enqueue_blur(o, s, so);
          if (ms && s == dsi && o < last)
            HIP_TRY(hipEventRecord(c->oct_ready[o], so));
        }
        if (ms && o > 0)
          HIP_TRY(hipEventRecord(c->oct_done[o], so));
      }
      if (ms)
        for (int o = 1; o <= last; ++o)
          HIP_TRY(hipStreamWaitEvent(stream, c->oct_done[o], 0));
    }
  }
  HIP_TRY(mark(2));

  // ---- polar gradients on the side stream, next to the extrema stage --------
  auto enqueue_gradients = [&](hipStream_t gs) -> sara_hip_status {
    for (int o = 0; o < sc.num_octaves; ++o)
    {
      const sara_hip_status gst = enqueue_gradient(o, gs);
      if (gst != SARA_HIP_OK)
        return gst;
    }
    return SARA_HIP_OK;
  };
  if (side && !pipe)
  {
    HIP_TRY(hipEventRecord(c->aux_fork, stream));
    HIP_TRY(hipStreamWaitEvent(c->aux_stream, c->aux_fork, 0));
    const sara_hip_status gst = enqueue_gradients(c->aux_stream);
    if (gst != SARA_HIP_OK)
      return gst;
    HIP_TRY(hipEventRecord(c->aux_join, c->aux_stream));
  }

  // ---- extrema ------------------------------------------------------------
  if (!pipe)
    launch_zero_counters(c->d_counters, counters_padded(c->max_batch), c->d_epoch,
                         int(step_stamp_index(c->max_batch)), stream);
  if (last_stage >= SARA_HIP_STAGE_EXTREMA)
  {
    if (!pipe)
      for (int o = 0; o < sc.num_octaves; ++o)
      {
        const sara_hip_status sst = enqueue_scan(o, stream);
        if (sst != SARA_HIP_OK)
          return sst;
      }
    {
      OctavePyramidView pv{};
      pv.scales = S;
      pv.octaves = sc.num_octaves;
      for (int o = 0; o < sc.num_octaves; ++o)
      {
        pv.base[o] = c->G[o];
        pv.w[o] = sc.oct[o].w;
        pv.h[o] = sc.oct[o].h;
        pv.plane[o] = size_t(pv.w[o]) * pv.h[o];
        pv.frame_stride[o] = pv.plane[o] * S;
      }
      launch_finish_sites(pv, batch, ep, c->d_tab, c->sites, c->cand, tail);
    }
    // row_buckets.total < bucket_stride: checked where the schedule is built
    launch_rank_candidates_bucketed(c->cand, c->row_buckets, c->d_bucket_hist,
                                    c->d_bucket_cursor, c->d_grouped, batch, tail);
  }
  HIP_TRY(mark(3));

  // ---- polar gradients ----------------------------------------------------
  if (pipe && linear)
  {
    // one chain: nothing to join
  }
  else if (pipe)
  {
    // join the side chains (their gradients follow their scans)
    for (int o = 0; o + 1 < sc.num_octaves; ++o)
      HIP_TRY(hipStreamWaitEvent(tail, c->oct_done[o], 0));
  }
  else if (side)
    HIP_TRY(hipStreamWaitEvent(stream, c->aux_join, 0));
  else if (want_gradients)
  {
    const sara_hip_status gst = enqueue_gradients(stream);
    if (gst != SARA_HIP_OK)
      return gst;
  }
  HIP_TRY(mark(4));

  // ---- orientations -------------------------------------------------------
  if (last_stage >= SARA_HIP_STAGE_ORIENTATION)
  {
    launch_orientations(c->d_grad, c->d_tab, c->d_oriw, c->n_oriw, c->cand,
                        c->ori, batch, tail);
    launch_scan_peaks(c->cand, c->ori, c->d_counters + 4 * size_t(c->max_batch) + 1,
                      batch, tail);
  }
  HIP_TRY(mark(5));

  // ---- descriptors --------------------------------------------------------
  if (last_stage >= SARA_HIP_STAGE_ORIENTATION)
    launch_descriptors(*c->h_grad, c->cand, c->ori, batch, c->d_feat, c->d_so,
                       c->d_desc, last_stage >= SARA_HIP_STAGE_DESCRIPTOR ? 1 : 0,
                       c->root_sift ? 1 : 0, tail);
  HIP_TRY(mark(6));
  HIP_TRY(hipGetLastError());
  return SARA_HIP_OK;
  };

  if (!graph_mode)
  {
    const sara_hip_status est = enqueue();
    if (est != SARA_HIP_OK)
    {
      c->epoch_synced = false;  // some of the step's launches may have run
      return est;
    }
    ++c->epoch_host;
    c->has_result = true;
    return SARA_HIP_OK;
  }
  // everything below touches graphs: on the launcher thread (GraphLauncher)
  auto graph_section = [&]() -> sara_hip_status {
  HIP_TRY(hipSetDevice(c->device));
  std::lock_guard<std::recursive_mutex> graph_lock(runtime_mutex());
  const int gs = c->write_slot;
  hipGraph_t& graph = c->graph_s[gs];
  hipGraphExec_t& graph_exec = c->graph_exec_s[gs];
  // a graph captured on the caller's frames serves other frames after an
  // argument update; one captured on d_input serves only d_input (and v.v.)
  const bool src_kind_ok =
      src_in_place ? (c->graph_src_s[gs] != nullptr &&
                      c->graph_src_stride_s[gs] == src_stride &&
                      (c->graph_src_s[gs] == static_cast<const void*>(src) ||
                       !c->graph_src_nodes_s[gs].empty()))
                   : c->graph_src_s[gs] == nullptr;
  const bool cached = graph_exec && c->graph_w_s[gs] == width &&
                      c->graph_h_s[gs] == height &&
                      c->graph_batch_s[gs] == batch &&
                      c->graph_stage_s[gs] == int(last_stage) && src_kind_ok;
  if (cached && src_in_place && c->graph_src_s[gs] != static_cast<const void*>(src))
  {
    // new frame address: rewrite the first argument of the kernels that read it
    bool ok = true;
    for (hipGraphNode_t node : c->graph_src_nodes_s[gs])
    {
      hipKernelNodeParams kp;
      ok = ok && hipGraphKernelNodeGetParams(node, &kp) == hipSuccess &&
           kp.kernelParams != nullptr;
      if (!ok)
        break;
      *static_cast<const void**>(kp.kernelParams[0]) = src;
      ok = hipGraphKernelNodeSetParams(node, &kp) == hipSuccess &&
           hipGraphExecKernelNodeSetParams(graph_exec, node, &kp) == hipSuccess;
    }
    if (!ok)
    {
      // this runtime cannot do it: copy to a fixed address from now on
      (void) hipGetLastError();
      c->graph_inplace = false;
      c->graph_stage_s[gs] = -1;
      return sara_hip_sift_detect(c, images, frame_stride, batch, width, height,
                                  images_on_device, last_stage, hip_stream);
    }
    c->graph_src_s[gs] = src;
  }
  if (!cached)
  {
    c->graph_src_s[gs] = nullptr;
    c->graph_src_nodes_s[gs].clear();
    if (graph_exec)
      (void) hipGraphExecDestroy(graph_exec);
    if (graph)
      (void) hipGraphDestroy(graph);
    graph_exec = nullptr;
    graph = nullptr;
    if (!graph_budget_left())
    {
      // see kOldRuntimeGraphBudget: plain launches from now on
      c->graph_broken = true;
      const sara_hip_status est = enqueue();
      if (est != SARA_HIP_OK)
      {
        c->epoch_synced = false;
        return est;
      }
      ++c->epoch_host;
      c->has_result = true;
      return SARA_HIP_OK;
    }
    g_graph_instantiations.fetch_add(1, std::memory_order_relaxed);
    bool ok = hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed) ==
              hipSuccess;
    if (ok)
    {
      const sara_hip_status est = enqueue();
      const hipError_t ee = hipStreamEndCapture(stream, &graph);
      ok = est == SARA_HIP_OK && ee == hipSuccess && graph != nullptr;
    }
    if (ok)
      ok = hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0) ==
           hipSuccess;
    if (!ok)
    {
      // fall back to plain launches for good; clear the sticky error
      (void) hipGetLastError();
      if (graph)
        (void) hipGraphDestroy(graph);
      graph = nullptr;
      graph_exec = nullptr;
      c->graph_broken = true;
      const sara_hip_status est = enqueue();
      if (est != SARA_HIP_OK)
      {
        c->epoch_synced = false;
        return est;
      }
      ++c->epoch_host;
      c->has_result = true;
      return SARA_HIP_OK;
    }
    c->graph_w_s[gs] = width;
    c->graph_h_s[gs] = height;
    c->graph_batch_s[gs] = batch;
    c->graph_stage_s[gs] = int(last_stage);
    if (src_in_place)
    {
      // the kernel nodes whose first argument is the frame pointer
      c->graph_src_s[gs] = src;
      c->graph_src_stride_s[gs] = src_stride;
      size_t n_nodes = 0;
      if (hipGraphGetNodes(graph, nullptr, &n_nodes) == hipSuccess && n_nodes > 0)
      {
        std::vector<hipGraphNode_t> nodes(n_nodes);
        if (hipGraphGetNodes(graph, nodes.data(), &n_nodes) == hipSuccess)
          for (size_t i = 0; i < n_nodes; ++i)
          {
            hipGraphNodeType type;
            hipKernelNodeParams kp;
            if (hipGraphNodeGetType(nodes[i], &type) != hipSuccess ||
                type != hipGraphNodeTypeKernel ||
                hipGraphKernelNodeGetParams(nodes[i], &kp) != hipSuccess ||
                !kp.kernelParams || !kp.kernelParams[0])
              continue;
            if (*static_cast<const void* const*>(kp.kernelParams[0]) ==
                static_cast<const void*>(src))
              c->graph_src_nodes_s[gs].push_back(nodes[i]);
          }
      }
      (void) hipGetLastError();
      // no such node found: the graph stays valid for this address only, and
      // the next address makes src_kind_ok false -> fall back to the copy
      if (c->graph_src_nodes_s[gs].empty())
        c->graph_inplace = false;
    }
  }
  {
    const hipError_t ge = hipGraphLaunch(graph_exec, stream);
    if (ge != hipSuccess)
    {
      c->epoch_synced = false;
      return fail(SARA_HIP_RUNTIME_ERROR,
                  std::string("hipGraphLaunch: ") + hipGetErrorString(ge));
    }
    ++c->epoch_host;
  }
  if (c->timers)
  {
    c->ev_recorded[SARA_HIP_TIME_TOTAL] = true;
    HIP_TRY(hipEventRecord(c->ev[SARA_HIP_TIME_TOTAL], stream));
  }
  c->has_result = true;
  return SARA_HIP_OK;
  };
  sara_hip_status gst = SARA_HIP_OK;
  std::string gmsg;
  graph_launcher().run([&] {
    gst = graph_section();
    if (gst != SARA_HIP_OK)
      gmsg = g_error;  // the launcher thread's message
  });
  if (gst != SARA_HIP_OK)
    return fail(gst, gmsg);
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_sift_detect_u8(sara_hip_sift* c, const uint8_t* images,
                                        size_t frame_stride, int channels,
                                        int batch, int width, int height,
                                        int images_on_device,
                                        sara_hip_stage last_stage,
                                        void* hip_stream)
{
  if (!c || !images)
    return fail(SARA_HIP_INVALID_PARAMS, "null context or images");
  if (channels != 1 && channels != 3)
    return fail(SARA_HIP_INVALID_PARAMS, "channels must be 1 (gray8) or 3 (RGB8)");
  if (batch < 1 || batch > c->max_batch)
    return fail(SARA_HIP_CAPACITY_EXCEEDED, "batch exceeds max_batch");
  if (width < 2 || height < 2)
    return fail(SARA_HIP_INVALID_PARAMS, "image smaller than 2x2");
  if (width > c->max_w || height > c->max_h)
    return fail(SARA_HIP_CAPACITY_EXCEEDED,
                "image larger than the context's max_width/max_height");
  const size_t px = size_t(width) * height;
  if (frame_stride == 0)
    frame_stride = px * channels;
  if (frame_stride < px * channels)
    return fail(SARA_HIP_SIZE_MISMATCH, "frame_stride < width*height*channels");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t stream =
      hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
  if (c->last_stream && c->last_stream != stream)
    HIP_TRY(hipStreamSynchronize(c->last_stream));
  const unsigned char* src = images;
  size_t src_stride = frame_stride;
  if (!images_on_device)
  {
    if (!c->d_u8)
    {
      const sara_hip_status st =
          c->alloc(c->d_u8, size_t(c->max_w) * c->max_h * 3 * c->max_batch);
      if (st != SARA_HIP_OK)
        return st;
    }
    HIP_TRY(hipMemcpy2DAsync(c->d_u8, px * channels, images, frame_stride,
                             px * channels, batch, hipMemcpyHostToDevice,
                             stream));
    src = c->d_u8;
    src_stride = px * channels;
  }
  if (channels == 1)
  {
    // gray8: detect() lets the first blur read the bytes itself when it can
    // (and converts into d_input otherwise)
    c->gray8_src = src;
    c->gray8_stride = src_stride;
  }
  else
  {
    launch_u8_to_gray32f(src, src_stride, channels, c->d_input, px, px, batch,
                         stream);
    HIP_TRY(hipGetLastError());
  }
  // the caller's handle (possibly null), not the resolved stream: a null
  // handle keeps the HIP-graph replay of small batches available
  const sara_hip_status st = sara_hip_sift_detect(
      c, c->d_input, px, batch, width, height, 1, last_stage, hip_stream);
  c->gray8_src = nullptr;
  return st;
}

}  // extern "C"
