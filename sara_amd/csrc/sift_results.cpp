// Everything around detect(): staging of host frames, submit / collect with two
// batches in flight, counts, fetch, the plane accessors of ComputeDoGExtrema
// (FeatureDetectors/DoG.hpp:115-165), stage timers.
#include "sift_host.hpp"

using namespace sara_hip;
using namespace sara_hip::host;

extern "C" {

namespace {
  //! The read-back stream.
  sara_hip_status ensure_d2h_stream(sara_hip_sift* c)
  {
    if (c->d2h_stream)
      return SARA_HIP_OK;
    // highest priority: the read-back kernel's few workgroups should not
    // queue behind the next batch's launches
    std::lock_guard<std::recursive_mutex> runtime_lock(runtime_mutex());
    HIP_TRY(pooled_stream_acquire(c->device, true, &c->d2h_stream));
    return SARA_HIP_OK;
  }
}  // namespace

sara_hip_status sara_hip_sift_stage(sara_hip_sift* c, const void* images,
                                    size_t frame_stride, int channels, int batch,
                                    int width, int height)
{
  if (!c || !images)
    return fail(SARA_HIP_INVALID_PARAMS, "null context or images");
  if (channels != 0 && channels != 1 && channels != 3)
    return fail(SARA_HIP_INVALID_PARAMS,
                "channels must be 0 (float), 1 (gray8) or 3 (RGB8)");
  if (batch < 1 || batch > c->max_batch)
    return fail(SARA_HIP_CAPACITY_EXCEEDED, "batch exceeds max_batch");
  if (width < 2 || height < 2)
    return fail(SARA_HIP_INVALID_PARAMS, "image smaller than 2x2");
  if (width > c->max_w || height > c->max_h)
    return fail(SARA_HIP_CAPACITY_EXCEEDED,
                "image larger than the context's max_width/max_height");
  const size_t px = size_t(width) * height;
  const size_t elem = channels == 0 ? sizeof(float) : size_t(channels);
  if (frame_stride == 0)
    frame_stride = channels == 0 ? px : px * channels;
  const size_t stride_bytes = channels == 0 ? frame_stride * sizeof(float)
                                            : frame_stride;
  if (stride_bytes < px * elem)
    return fail(SARA_HIP_SIZE_MISMATCH, "frame_stride smaller than a frame");
  HIP_TRY(hipSetDevice(c->device));
  if (!c->copy_stream)
  {
    const sara_hip_status ds = ensure_d2h_stream(c);  // before the first upload
    if (ds != SARA_HIP_OK)
      return ds;
    std::lock_guard<std::recursive_mutex> runtime_lock(runtime_mutex());
    HIP_TRY(pooled_stream_acquire(c->device, false, &c->copy_stream));
    for (int k = 0; k < 2; ++k)
    {
      HIP_TRY(hipEventCreateWithFlags(&c->stage_ready[k], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&c->stage_free[k], hipEventDisableTiming));
      unsigned char* p = nullptr;
      const sara_hip_status st =
          c->alloc(p, size_t(c->max_w) * c->max_h * sizeof(float) * c->max_batch);
      if (st != SARA_HIP_OK)
        return st;
      c->d_stage[k] = p;
    }
  }
  const int k = c->stage_next;
  // The host waits here for the upload BEFORE this one, so that never more
  // than one upload is bound to a copy engine when the read-back of an older
  // batch asks for one.  Measured with stage(i + 1); collect(i - 1);
  // submit_staged(i + 1) on 64 x 1080p float32 frames:
  // under the ROCm 7.2 runtime 9.4 ms per step in every process with the wait,
  // 8.9 or 12.4 ms without (which of the two depends on what the process
  // copied first); under the 7.0.2 runtime (the one inside the torch wheel)
  // 12.5 ms with the wait and 9.2-9.4 ms without.  Hence the default follows
  // the runtime's version; SARA_HIP_STAGE_WAIT=0 / 1 forces it.
  static const bool stage_wait = [] {
    if (const char* e = getenv("SARA_HIP_STAGE_WAIT"))
      return e[0] == '1';
    int v = 0;
    return hipRuntimeGetVersion(&v) == hipSuccess && v >= 70200000;
  }();
  if (stage_wait)
    HIP_TRY(hipStreamSynchronize(c->copy_stream));
  // the pipeline that last read this buffer must be done with it
  if (c->stage_used[k])
    HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->stage_free[k], 0));
  if (stride_bytes == px * elem)  // contiguous frames: one linear copy
    HIP_TRY(hipMemcpyAsync(c->d_stage[k], images, px * elem * batch,
                           hipMemcpyHostToDevice, c->copy_stream));
  else
    HIP_TRY(hipMemcpy2DAsync(c->d_stage[k], px * elem, images, stride_bytes,
                             px * elem, batch, hipMemcpyHostToDevice,
                             c->copy_stream));
  HIP_TRY(hipEventRecord(c->stage_ready[k], c->copy_stream));
  c->staged = k;
  c->stage_next = 1 - k;
  c->staged_channels = channels;
  c->staged_batch = batch;
  c->staged_w = width;
  c->staged_h = height;
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_sift_detect_staged(sara_hip_sift* c,
                                            sara_hip_stage last_stage,
                                            void* hip_stream)
{
  if (!c)
    return fail(SARA_HIP_INVALID_PARAMS, "null context");
  if (c->staged < 0)
    return fail(SARA_HIP_NOT_READY, "no batch has been staged");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t stream =
      hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
  const int k = c->staged;
  c->staged = -1;
  HIP_TRY(hipStreamWaitEvent(stream, c->stage_ready[k], 0));
  const size_t px = size_t(c->staged_w) * c->staged_h;
  sara_hip_status st;
  // the staging buffer is handed back when its frames have been consumed
  // (detect() records the event behind the first blur), not when the batch is
  // complete: stage(i + 2) can then follow upload(i + 1) on the copy engine
  // without waiting for the kernels of batch i
  c->consumed_event = c->stage_free[k];
  c->consumed_recorded = false;
  if (c->staged_channels == 0)
    st = sara_hip_sift_detect(c, static_cast<const float*>(c->d_stage[k]), px,
                              c->staged_batch, c->staged_w, c->staged_h, 1,
                              last_stage, hip_stream);
  else
    st = sara_hip_sift_detect_u8(c, static_cast<const uint8_t*>(c->d_stage[k]),
                                 px * c->staged_channels, c->staged_channels,
                                 c->staged_batch, c->staged_w, c->staged_h, 1,
                                 last_stage, hip_stream);
  c->consumed_event = nullptr;
  if (st != SARA_HIP_OK)
    return st;
  if (!c->consumed_recorded)  // graph replay: at the end of the batch
    HIP_TRY(hipEventRecord(c->stage_free[k], stream));
  c->stage_used[k] = true;
  return SARA_HIP_OK;
}

namespace {
  //! Points the pipeline's outputs at result slot `slot` (allocating slot 1 on
  //! first use).
  sara_hip_status select_result_slot(sara_hip_sift* c, int slot)
  {
    if (!c->d_feat_s[slot])
    {
      const size_t rows = size_t(c->max_batch) * c->cap;
      sara_hip_status st = c->alloc(c->d_feat_s[slot], rows);
      if (st == SARA_HIP_OK)
        st = c->alloc(c->d_so_s[slot], rows * 2);
      if (st == SARA_HIP_OK)
        st = c->alloc(c->d_desc_s[slot], rows * 128);
      if (st != SARA_HIP_OK)
        return st;
      if (slot == 1)
        c->has_slot1 = true;
    }
    c->write_slot = slot;
    c->d_feat = c->d_feat_s[slot];
    c->d_so = c->d_so_s[slot];
    c->d_desc = c->d_desc_s[slot];
    return SARA_HIP_OK;
  }
}  // namespace

namespace {
  sara_hip_status submit_impl(sara_hip_sift* c, const void* images,
                              size_t frame_stride, int channels, int batch,
                              int width, int height, int images_on_device,
                              sara_hip_stage last_stage, int* ticket);
}

sara_hip_status sara_hip_sift_submit(sara_hip_sift* c, const void* images,
                                     size_t frame_stride, int channels,
                                     int batch, int width, int height,
                                     int images_on_device,
                                     sara_hip_stage last_stage, int* ticket)
{
  if (!c || !images || !ticket)
    return fail(SARA_HIP_INVALID_PARAMS, "null context, images or ticket");
  return submit_impl(c, images, frame_stride, channels, batch, width, height,
                     images_on_device, last_stage, ticket);
}

sara_hip_status sara_hip_sift_submit_staged(sara_hip_sift* c,
                                            sara_hip_stage last_stage, int* ticket)
{
  if (!c || !ticket)
    return fail(SARA_HIP_INVALID_PARAMS, "null context or ticket");
  if (c->staged < 0)
    return fail(SARA_HIP_NOT_READY, "no batch has been staged");
  return submit_impl(c, nullptr, 0, 0, c->staged_batch, c->staged_w, c->staged_h, 0,
                     last_stage, ticket);
}

namespace {
sara_hip_status submit_impl(sara_hip_sift* c, const void* images,
                            size_t frame_stride, int channels, int batch,
                            int width, int height, int images_on_device,
                            sara_hip_stage last_stage, int* ticket)
{
  if (last_stage < SARA_HIP_STAGE_ORIENTATION)
    return fail(SARA_HIP_INVALID_PARAMS,
                "submit() delivers keypoints: last_stage must be >= ORIENTATION");
  if (channels != 0 && channels != 1 && channels != 3)
    return fail(SARA_HIP_INVALID_PARAMS,
                "channels must be 0 (float), 1 (gray8) or 3 (RGB8)");
  HIP_TRY(hipSetDevice(c->device));
  const int slot = c->next_ticket & 1;
  sara_hip_sift::RingSlot& r = c->ring[slot];
  if (r.pending)
    return fail(SARA_HIP_NOT_READY,
                "two batches are in flight: collect() the older ticket first");
  if (!r.done)
  {
    HIP_TRY(hipEventCreateWithFlags(&r.done, hipEventDisableTiming));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&r.h_counters),
                          sizeof(int) * counters_read(c->max_batch)));
  }
  {
    const sara_hip_status ds = ensure_d2h_stream(c);
    if (ds != SARA_HIP_OK)
      return ds;
  }
  sara_hip_status st = select_result_slot(c, slot);
  if (st != SARA_HIP_OK)
    return st;
  if (!images)  // submit_staged(): the batch is on its way already
    st = sara_hip_sift_detect_staged(c, last_stage, nullptr);
  else if (!images_on_device)
  {
    // upload on the copy stream (double-buffered staging), then the pipeline
    st = sara_hip_sift_stage(c, images, frame_stride, channels, batch, width,
                             height);
    if (st == SARA_HIP_OK)
      st = sara_hip_sift_detect_staged(c, last_stage, nullptr);
  }
  else if (channels == 0)
    st = sara_hip_sift_detect(c, static_cast<const float*>(images), frame_stride,
                              batch, width, height, 1, last_stage, nullptr);
  else
    st = sara_hip_sift_detect_u8(c, static_cast<const uint8_t*>(images),
                                 frame_stride, channels, batch, width, height, 1,
                                 last_stage, nullptr);
  if (st != SARA_HIP_OK)
    return st;
  // the counters of this batch travel to pinned memory in stream order: the
  // next batch may reset them before collect() looks
  HIP_TRY(hipMemcpyAsync(r.h_counters, c->d_counters,
                         sizeof(int) * counters_read(c->max_batch),
                         hipMemcpyDeviceToHost, c->last_stream));
  HIP_TRY(hipEventRecord(r.done, c->last_stream));
  r.ticket = c->next_ticket;
  r.step = c->epoch_host;
  r.pending = true;
  r.batch = batch;
  r.stage = last_stage;
  *ticket = c->next_ticket++;
  return SARA_HIP_OK;
}
}  // namespace

sara_hip_status sara_hip_sift_collect(sara_hip_sift* c, int ticket,
                                      const sara_oeregion** features,
                                      const float** descriptors,
                                      const int32_t** scale_octave,
                                      const int32_t** frame_offsets, int* total)
{
  if (!c)
    return fail(SARA_HIP_INVALID_PARAMS, "null context");
  sara_hip_sift::RingSlot& r = c->ring[ticket & 1];
  if (ticket < 0 || !r.pending || r.ticket != ticket)
    return fail(SARA_HIP_NOT_READY, "unknown or already collected ticket");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipEventSynchronize(r.done));
  const int mb = c->max_batch;
  const int* h_ex = r.h_counters;
  const int* h_sites = r.h_counters + mb;
  const int* h_kp = r.h_counters + 2 * size_t(mb);
  const int* h_off = r.h_counters + 3 * size_t(mb);
  const int n = h_off[r.batch];
  if (descriptors && r.stage < SARA_HIP_STAGE_DESCRIPTOR)
    return fail(SARA_HIP_NOT_READY,
                "descriptors requested, but the ticket was submitted with "
                "last_stage < DESCRIPTOR (collect it with descriptors = NULL)");
  sara_hip_status status = SARA_HIP_OK;
  if (counters_corrupt(c, r.h_counters, mb, r.batch, 3, r.step))
  {
    r.pending = false;
    return corrupt_counters_error();
  }
  note_required(c, h_ex, h_sites, h_kp, r.batch);
  for (int b = 0; b < r.batch && status == SARA_HIP_OK; ++b)
    if (h_kp[b] > c->cap || h_ex[b] > c->cap || h_sites[b] > c->sites.cap)
      status = fail(SARA_HIP_CAPACITY_EXCEEDED,
                    "a frame produced more extrema / keypoints than "
                    "max_keypoints: the lists are truncated");
  if (size_t(n) > r.h_cap)
  {
    if (r.h_feat)
      (void) hipHostFree(r.h_feat);
    if (r.h_desc)
      (void) hipHostFree(r.h_desc);
    if (r.h_so)
      (void) hipHostFree(r.h_so);
    r.h_feat = nullptr;
    r.h_desc = nullptr;
    r.h_so = nullptr;
    r.h_cap = 0;
    const size_t want = std::min(size_t(mb) * c->cap, size_t(n) + size_t(n) / 2 + 1024);
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&r.h_feat),
                          sizeof(sara_oeregion) * want));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&r.h_desc),
                          sizeof(float) * 128 * want));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&r.h_so),
                          sizeof(int32_t) * 2 * want));
    r.h_cap = want;
  }
  const int slot = ticket & 1;
  if (n > 0)
  {
    // the batch is complete (event): the copies need no further ordering and
    // run beside the next batch's kernels
    {
      HIP_TRY(hipMemcpyAsync(r.h_feat, c->d_feat_s[slot], sizeof(sara_oeregion) * n,
                             hipMemcpyDeviceToHost, c->d2h_stream));
      HIP_TRY(hipMemcpyAsync(r.h_so, c->d_so_s[slot], sizeof(int32_t) * 2 * n,
                             hipMemcpyDeviceToHost, c->d2h_stream));
      if (descriptors)
        HIP_TRY(hipMemcpyAsync(r.h_desc, c->d_desc_s[slot],
                               sizeof(float) * 128 * size_t(n),
                               hipMemcpyDeviceToHost, c->d2h_stream));
    }
    HIP_TRY(hipStreamSynchronize(c->d2h_stream));
  }
  r.pending = false;
  if (features)
    *features = r.h_feat;
  if (descriptors)
    *descriptors = r.h_desc;
  if (scale_octave)
    *scale_octave = r.h_so;
  if (frame_offsets)
    *frame_offsets = h_off;
  if (total)
    *total = n;
  return status;
}

sara_hip_status sara_hip_sift_ticket_counts(sara_hip_sift* c, int ticket,
                                           int32_t* frame_offsets, int* batch,
                                           int* total)
{
  sara_hip::TicketResults res;
  const sara_hip_status st = sara_hip::ticket_results(c, ticket, &res);
  if (st != SARA_HIP_OK)
    return st;
  if (frame_offsets)
    std::copy(res.h_offsets, res.h_offsets + res.batch + 1, frame_offsets);
  if (batch)
    *batch = res.batch;
  if (total)
    *total = res.total;
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_sift_collect_into(sara_hip_sift* c, int ticket,
                                           sara_oeregion* features,
                                           float* descriptors,
                                           int32_t* scale_octave)
{
  sara_hip::TicketResults res;
  const sara_hip_status st = sara_hip::ticket_results(c, ticket, &res);
  if (st != SARA_HIP_OK)
    return st;
  if (descriptors && res.last_stage < SARA_HIP_STAGE_DESCRIPTOR)
    return fail(SARA_HIP_NOT_READY,
                "descriptors requested, but the ticket was submitted with "
                "last_stage < DESCRIPTOR");
  const size_t n = size_t(res.total);
  if (n > 0)
  {
    if (features)
      HIP_TRY(hipMemcpyAsync(features, res.d_feat, sizeof(sara_oeregion) * n,
                             hipMemcpyDeviceToHost, c->d2h_stream));
    if (scale_octave)
      HIP_TRY(hipMemcpyAsync(scale_octave, res.d_so, sizeof(int32_t) * 2 * n,
                             hipMemcpyDeviceToHost, c->d2h_stream));
    if (descriptors)
      HIP_TRY(hipMemcpyAsync(descriptors, res.d_desc, sizeof(float) * 128 * n,
                             hipMemcpyDeviceToHost, c->d2h_stream));
    HIP_TRY(hipStreamSynchronize(c->d2h_stream));
  }
  sara_hip::ticket_release(c, ticket);
  if (res.capacity_exceeded)
    return fail(SARA_HIP_CAPACITY_EXCEEDED,
                "a frame produced more extrema / keypoints than "
                "max_keypoints: the lists are truncated");
  return SARA_HIP_OK;
}

}  // extern "C"

namespace sara_hip {
  sara_hip_status set_error(sara_hip_status code, const char* msg)
  {
    return fail(code, msg);
  }

  sara_hip_status ticket_results(sara_hip_sift* c, int ticket, TicketResults* out)
  {
    if (!c || !out)
      return fail(SARA_HIP_INVALID_PARAMS, "null context");
    sara_hip_sift::RingSlot& r = c->ring[ticket & 1];
    if (ticket < 0 || !r.pending || r.ticket != ticket)
      return fail(SARA_HIP_NOT_READY, "unknown or already collected ticket");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(r.done));
    const int mb = c->max_batch;
    if (counters_corrupt(c, r.h_counters, mb, r.batch, 3, r.step))
    {
      r.pending = false;
      return corrupt_counters_error();
    }
    const int* h_off = r.h_counters + 3 * size_t(mb);
    out->device = c->device;
    out->batch = r.batch;
    out->total = h_off[r.batch];
    out->h_offsets = h_off;
    out->d_feat = c->d_feat_s[ticket & 1];
    out->d_desc = c->d_desc_s[ticket & 1];
    out->d_so = c->d_so_s[ticket & 1];
    out->capacity_exceeded = false;
    out->last_stage = r.stage;
    note_required(c, r.h_counters, r.h_counters + mb,
                  r.h_counters + 2 * size_t(mb), r.batch);
    for (int b = 0; b < r.batch; ++b)
      if (r.h_counters[2 * size_t(mb) + b] > c->cap || r.h_counters[b] > c->cap ||
          r.h_counters[mb + b] > c->sites.cap)
        out->capacity_exceeded = true;
    return SARA_HIP_OK;
  }

  void ticket_release(sara_hip_sift* c, int ticket)
  {
    if (!c || ticket < 0)
      return;
    sara_hip_sift::RingSlot& r = c->ring[ticket & 1];
    if (r.pending && r.ticket == ticket)
    {
      r.pending = false;
    }
  }
}  // namespace sara_hip

extern "C" {

sara_hip_status sara_hip_sift_synchronize(sara_hip_sift* c)
{
  if (!c)
    return fail(SARA_HIP_INVALID_PARAMS, "null context");
  HIP_TRY(hipSetDevice(c->device));
  if (c->last_stream)
    HIP_TRY(hipStreamSynchronize(c->last_stream));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_sift_counts(sara_hip_sift* c, int* per_frame, int* total)
{
  const sara_hip_status st = require_result(c, SARA_HIP_STAGE_ORIENTATION);
  if (st != SARA_HIP_OK)
    return st;
  HIP_TRY(hipSetDevice(c->device));
  // cand.count | sites.count | ori.kp_count are contiguous in d_counters: one
  // round trip brings all three (h_counts holds counters_read(max_batch) ints: the three per-frame
  // lists, the frame offsets and the error flag)
  int* h_ex = c->h_counts;
  int* h_sites = c->h_counts + c->max_batch;
  int* h_kp = c->h_counts + 2 * size_t(c->max_batch);
  HIP_TRY(hipMemcpyAsync(c->h_counts, c->d_counters,
                         sizeof(int) * counters_read(c->max_batch),
                         hipMemcpyDeviceToHost, c->last_stream));
  HIP_TRY(hipStreamSynchronize(c->last_stream));
  if (counters_corrupt(c, c->h_counts, c->max_batch, c->cur_batch, 3, c->epoch_host))
    return corrupt_counters_error();
  note_required(c, h_ex, h_sites, h_kp, c->cur_batch);
  int sum = 0;
  bool overflow = false;
  for (int b = 0; b < c->cur_batch; ++b)
  {
    const int n = h_kp[b];
    overflow = overflow || n > c->cap;
    if (per_frame)
      per_frame[b] = std::min(n, c->cap);
    sum += std::min(n, c->cap);
  }
  if (total)
    *total = sum;
  if (overflow)
    return fail(SARA_HIP_CAPACITY_EXCEEDED,
                "a frame produced more keypoints than max_keypoints");
  // the extremum list and the list of classified sites can also overflow
  // without the keypoint list doing so (keypoints would be missing silently)
  for (int b = 0; b < c->cur_batch; ++b)
  {
    if (h_ex[b] > c->cap)
      return fail(SARA_HIP_CAPACITY_EXCEEDED,
                  "a frame produced more extrema than max_keypoints");
    if (h_sites[b] > c->sites.cap)
      return fail(SARA_HIP_CAPACITY_EXCEEDED,
                  "a frame produced more classified sites than 4*max_keypoints");
  }
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_sift_fetch(sara_hip_sift* c, sara_oeregion* features,
                                    float* descriptors, int32_t* scale_octave,
                                    int dst_on_device)
{
  const sara_hip_status st = require_result(c, SARA_HIP_STAGE_ORIENTATION);
  if (st != SARA_HIP_OK)
    return st;
  if (descriptors && c->last_stage < SARA_HIP_STAGE_DESCRIPTOR)
    return fail(SARA_HIP_NOT_READY, "descriptors were not computed");
  HIP_TRY(hipSetDevice(c->device));
  int total = 0;
  HIP_TRY(hipMemcpyAsync(&c->h_counts[0], c->ori.frame_offset + c->cur_batch,
                         sizeof(int), hipMemcpyDeviceToHost, c->last_stream));
  HIP_TRY(hipStreamSynchronize(c->last_stream));
  total = c->h_counts[0];
  if (total == 0)
    return SARA_HIP_OK;
  const hipMemcpyKind kind =
      dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  if (features)
    HIP_TRY(hipMemcpyAsync(features, c->d_feat, sizeof(sara_oeregion) * total,
                           kind, c->last_stream));
  if (descriptors)
    HIP_TRY(hipMemcpyAsync(descriptors, c->d_desc,
                           sizeof(float) * 128 * size_t(total), kind,
                           c->last_stream));
  if (scale_octave)
    HIP_TRY(hipMemcpyAsync(scale_octave, c->d_so, sizeof(int32_t) * 2 * total,
                           kind, c->last_stream));
  if (!dst_on_device)
    HIP_TRY(hipStreamSynchronize(c->last_stream));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_sift_device_results(sara_hip_sift* c,
                                             const sara_oeregion** features,
                                             const float** descriptors,
                                             const int32_t** scale_octave,
                                             const int32_t** frame_offsets)
{
  const sara_hip_status st = require_result(c, SARA_HIP_STAGE_ORIENTATION);
  if (st != SARA_HIP_OK)
    return st;
  if (features)
    *features = c->d_feat;
  if (descriptors)
    *descriptors = c->d_desc;
  if (scale_octave)
    *scale_octave = c->d_so;
  if (frame_offsets)
    *frame_offsets = c->ori.frame_offset;
  return SARA_HIP_OK;
}

int sara_hip_sift_octave_count(const sara_hip_sift* c)
{
  return (c && c->cur_w > 0) ? c->cur.num_octaves : 0;
}

sara_hip_status sara_hip_sift_octave_info(const sara_hip_sift* c, int octave,
                                          int* w, int* h, float* factor)
{
  if (!c || c->cur_w <= 0)
    return fail(SARA_HIP_NOT_READY, "no detect() has run on this context");
  if (octave < 0 || octave >= c->cur.num_octaves)
    return fail(SARA_HIP_OUT_OF_RANGE, "octave index out of range");
  if (w)
    *w = c->cur.oct[octave].w;
  if (h)
    *h = c->cur.oct[octave].h;
  if (factor)
    *factor = c->cur.oct[octave].factor;
  return SARA_HIP_OK;
}

static sara_hip_status copy_plane(sara_hip_sift* c, std::vector<float*>& pyr,
                                  int frame, int s, int o, int scales, int chans,
                                  float* dst, sara_hip_stage need)
{
  const sara_hip_status st = require_result(c, need);
  if (st != SARA_HIP_OK)
    return st;
  if (!dst)
    return fail(SARA_HIP_INVALID_PARAMS, "null destination");
  if (frame < 0 || frame >= c->cur_batch || o < 0 || o >= c->cur.num_octaves ||
      s < 0 || s >= scales)
    return fail(SARA_HIP_OUT_OF_RANGE, "frame/scale/octave index out of range");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->last_stream));
  const size_t pl = size_t(c->cur.oct[o].w) * c->cur.oct[o].h * chans;
  HIP_TRY(hipMemcpy(dst, c->plane(pyr, o, frame, s, chans, scales),
                    pl * sizeof(float), hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_sift_copy_gaussian(sara_hip_sift* c, int frame, int s,
                                            int o, float* dst)
{
  return copy_plane(c, c->G, frame, s, o, c ? c->S : 0, 1, dst,
                    SARA_HIP_STAGE_PYRAMID);
}

sara_hip_status sara_hip_sift_copy_dog(sara_hip_sift* c, int frame, int s, int o,
                                       float* dst)
{
  // diff_of_gaussians()(s, o) = gaussians()(s+1, o) - gaussians()(s, o)
  // (GaussianPyramid.cpp:44-46), formed on demand.
  const sara_hip_status st = require_result(c, SARA_HIP_STAGE_PYRAMID);
  if (st != SARA_HIP_OK)
    return st;
  if (!dst)
    return fail(SARA_HIP_INVALID_PARAMS, "null destination");
  if (frame < 0 || frame >= c->cur_batch || o < 0 || o >= c->cur.num_octaves ||
      s < 0 || s >= c->S - 1)
    return fail(SARA_HIP_OUT_OF_RANGE, "frame/scale/octave index out of range");
  HIP_TRY(hipSetDevice(c->device));
  const size_t pl = size_t(c->cur.oct[o].w) * c->cur.oct[o].h;
  launch_subtract(c->plane(c->G, o, frame, s + 1, 1, c->S),
                  c->plane(c->G, o, frame, s, 1, c->S), c->d_dog_plane, pl,
                  c->last_stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(dst, c->d_dog_plane, pl * sizeof(float),
                         hipMemcpyDeviceToHost, c->last_stream));
  HIP_TRY(hipStreamSynchronize(c->last_stream));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_sift_copy_gradient(sara_hip_sift* c, int frame, int s,
                                            int o, float* dst)
{
  if (c && !c->all_gradient_scales && (s < 1 || s > c->S - 3))
    return fail(SARA_HIP_OUT_OF_RANGE,
                "only scales 1..S-3 are materialised; set "
                "SARA_HIP_OPT_ALL_GRADIENT_SCALES for the others");
  return copy_plane(c, c->GR, frame, s, o, c ? c->S : 0, 2, dst,
                    SARA_HIP_STAGE_GRADIENT);
}

sara_hip_status sara_hip_sift_extrema_counts(sara_hip_sift* c, int* per_frame,
                                             int* total)
{
  const sara_hip_status st = require_result(c, SARA_HIP_STAGE_EXTREMA);
  if (st != SARA_HIP_OK)
    return st;
  HIP_TRY(hipSetDevice(c->device));
  // cand.count | sites.count are contiguous in d_counters
  int* h_ex = c->h_counts;
  int* h_sites = c->h_counts + c->max_batch;
  HIP_TRY(hipMemcpyAsync(c->h_counts, c->d_counters,
                         sizeof(int) * counters_read(c->max_batch),
                         hipMemcpyDeviceToHost, c->last_stream));
  HIP_TRY(hipStreamSynchronize(c->last_stream));
  if (counters_corrupt(c, c->h_counts, c->max_batch, c->cur_batch, 2, c->epoch_host))
    return corrupt_counters_error();
  note_required(c, h_ex, h_sites, nullptr, c->cur_batch);
  int sum = 0;
  bool overflow = false;
  for (int b = 0; b < c->cur_batch; ++b)
  {
    const int n = h_ex[b];
    overflow = overflow || n > c->cap;
    if (per_frame)
      per_frame[b] = std::min(n, c->cap);
    sum += std::min(n, c->cap);
  }
  if (total)
    *total = sum;
  if (overflow)
    return fail(SARA_HIP_CAPACITY_EXCEEDED,
                "a frame produced more extrema than max_keypoints");
  for (int b = 0; b < c->cur_batch; ++b)
    if (h_sites[b] > c->sites.cap)
      return fail(SARA_HIP_CAPACITY_EXCEEDED,
                  "a frame produced more classified sites than 4*max_keypoints");
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_sift_fetch_extrema(sara_hip_sift* c,
                                            sara_oeregion* regions,
                                            int32_t* xyso_type)
{
  int total = 0;
  sara_hip_status st = sara_hip_sift_extrema_counts(c, nullptr, &total);
  if (st != SARA_HIP_OK && st != SARA_HIP_CAPACITY_EXCEEDED)
    return st;
  if (total == 0)
    return st;
  launch_extrema_offsets(c->cand, c->d_ex_offset, c->cur_batch, c->last_stream);
  launch_gather_extrema(c->cand, c->d_ex_offset, c->cur_batch, c->d_ex_regions,
                        c->d_ex_xyso, c->last_stream);
  if (regions)
    HIP_TRY(hipMemcpyAsync(regions, c->d_ex_regions,
                           sizeof(sara_oeregion) * total, hipMemcpyDeviceToHost,
                           c->last_stream));
  if (xyso_type)
    HIP_TRY(hipMemcpyAsync(xyso_type, c->d_ex_xyso, sizeof(int32_t) * 5 * total,
                           hipMemcpyDeviceToHost, c->last_stream));
  HIP_TRY(hipStreamSynchronize(c->last_stream));
  return st;
}

sara_hip_status sara_hip_sift_stage_times(sara_hip_sift* c, float* ms)
{
  const sara_hip_status st = require_result(c, SARA_HIP_STAGE_PYRAMID);
  if (st != SARA_HIP_OK)
    return st;
  if (!ms)
    return fail(SARA_HIP_INVALID_PARAMS, "null destination");
  if (!c->timers)
    return fail(SARA_HIP_NOT_READY, "stage timers are disabled");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->last_stream));
  for (int i = 0; i < SARA_HIP_TIME_TOTAL; ++i)
  {
    ms[i] = 0.f;
    if (c->ev_recorded[i] && c->ev_recorded[i + 1])
      HIP_TRY(hipEventElapsedTime(&ms[i], c->ev[i], c->ev[i + 1]));
  }
  ms[SARA_HIP_TIME_TOTAL] = 0.f;
  if (c->ev_recorded[0] && c->ev_recorded[SARA_HIP_TIME_TOTAL])
    HIP_TRY(hipEventElapsedTime(&ms[SARA_HIP_TIME_TOTAL], c->ev[0],
                                c->ev[SARA_HIP_TIME_TOTAL]));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_sift_pyramid_launches(sara_hip_sift* c,
                                               sara_hip_launch_time* out,
                                               int capacity, int* count)
{
  const sara_hip_status st = require_result(c, SARA_HIP_STAGE_PYRAMID);
  if (st != SARA_HIP_OK)
    return st;
  if (!count || (capacity > 0 && !out))
    return fail(SARA_HIP_INVALID_PARAMS, "null destination");
  if (!c->launch_timers)
    return fail(SARA_HIP_NOT_READY, "SARA_HIP_OPT_LAUNCH_TIMERS is off");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->last_stream));
  *count = c->launch_count;
  for (int i = 0; i < c->launch_count && i < capacity; ++i)
  {
    const sara_hip_sift::LaunchRecord& r = c->launch_rec[size_t(i)];
    out[i].octave = r.octave;
    out[i].scale = r.scale;
    out[i].taps = r.taps;
    out[i].pixels = r.pixels;
    out[i].ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&out[i].ms, r.begin, r.end));
  }
  return SARA_HIP_OK;
}

}  // extern "C"
