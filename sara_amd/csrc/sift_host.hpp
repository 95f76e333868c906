// Internal header of the host side of the C-ABI (include/sara_hip_sift.h): what
// the translation units behind it share.  Round 6 split the former 3 600-line
// sift_context.cpp along its seams:
//   graph_launcher.cpp   the HIP-graph rules of the ROCm 7 runtimes: launcher
//                        thread, first-thread rule and graph budget of < 7.2
//   sift_schedule.cpp    host arithmetic of the parameter schedule (Gaussian
//                        taps, octave geometry, bin thresholds), kernel
//                        selection, the host-only entry points, error text
//   sift_context.cpp     the context: HBM buffers, create / destroy / options /
//                        reserve
//   sift_detect.cpp      detect(): launch sequence, streams, graph capture / replay
//   sift_results.cpp     staging, submit / collect, counts, fetch, plane accessors
//   sift_operators.cpp   the operator-level seams and device self-checks
#pragma once

#include "sift_kernels.hpp"

#include "device_math.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <limits>
#include <mutex>
#include <pthread.h>
#include <string>
#include <thread>
#include <vector>

namespace sara_hip { namespace host {


  //! Every HIP-graph call of the process - capture, instantiation, argument
  //! update, hipGraphLaunch - runs on ONE thread, the graph launcher.  With
  //! graphs captured and launched from several host threads the ROCm 7 runtime
  //! crashed in hip::Graph::UpdateStreams (under hipGraphLaunch) even with every
  //! graph call serialised by runtime_mutex() and every graph used only by the
  //! thread that captured it (rocgdb backtrace; tests/test_gpu_pipeline.py::
  //! test_compute_sift_keypoints_keeps_its_context was the reproducer).  Round 3
  //! therefore gave graph replay to the first thread that asked and left every
  //! other thread on plain launches (+ 0.15 ms per 1080p frame).  Now - on
  //! ROCm 7.2 and later; older runtimes keep round 3's rule, see
  //! graphs_need_one_thread() below - a caller of any thread hands the graph
  //! part of its detect() to the launcher and waits for it: the caller is
  //! blocked for the duration anyway (the host side
  //! of a replay is what detect() consists of), so nothing is lost but the
  //! hand-over.  Both sides wait cooperatively: a short run of `pause`
  //! instructions (the answer is usually microseconds away), then
  //! sched_yield() between looks - so that a process with more threads than
  //! cores hands the core to whoever it is waiting for - then a condition
  //! variable.  The launcher only polls at all while calls keep coming (the
  //! previous job arrived within a millisecond of the one before: a video
  //! loop); an occasional caller finds it asleep and pays one wake-up.
  //! After fork() the child has no launcher thread: a pthread_atfork handler
  //! gives it a fresh launcher (graph_launcher()).
  class GraphLauncher
  {
  public:
    //! Runs fn() on the launcher thread and returns when it has finished.
    template <typename F>
    void run(F&& fn)
    {
      if (std::this_thread::get_id() == thread_id_.load(std::memory_order_acquire))
      {
        fn();  // a nested call from inside a job
        return;
      }
      Job job;
      job.fn = [&fn] { fn(); };
      {
        std::lock_guard<std::mutex> lock(m_);
        if (!started_)
        {
          started_ = true;
          worker_ = std::thread([this] { loop(); });
        }
        queue_.push_back(&job);
        ++posted_;
      }
      if (sleeping_.load(std::memory_order_acquire))
        cv_.notify_one();
      // the job is tens of microseconds of host work: look before sleeping
      if (!wait_briefly([&] { return job.done.load(std::memory_order_acquire); },
                        std::chrono::microseconds(2000)))
      {
        std::unique_lock<std::mutex> lock(job.m);
        job.waiting = true;
        job.cv.wait(lock, [&] { return job.done.load(std::memory_order_acquire); });
      }
      // the launcher may still be inside the notification of job.cv
      std::lock_guard<std::mutex> lock(job.m);
    }

    ~GraphLauncher()
    {
      {
        std::lock_guard<std::mutex> lock(m_);
        stop_ = true;
      }
      cv_.notify_all();
      if (worker_.joinable())
        worker_.join();
    }

  private:
    struct Job
    {
      std::function<void()> fn;
      std::atomic<bool> done{false};
      std::mutex m;
      std::condition_variable cv;
      bool waiting = false;
    };

    void loop()
    {
      thread_id_.store(std::this_thread::get_id(), std::memory_order_release);
      for (;;)
      {
        Job* job = nullptr;
        {
          std::unique_lock<std::mutex> lock(m_);
          if (queue_.empty())
          {
            // in a hot loop the caller's next detect() is a few hundred
            // microseconds away: look for it before sleeping
            lock.unlock();
            const bool found =
                hot_ && wait_briefly(
                            [&] {
                              return posted_.load(std::memory_order_acquire) != taken_;
                            },
                            std::chrono::microseconds(200));
            lock.lock();
            if (!found && queue_.empty())
            {
              sleeping_.store(true, std::memory_order_release);
              cv_.wait(lock, [&] { return stop_ || !queue_.empty(); });
              sleeping_.store(false, std::memory_order_release);
            }
          }
          if (queue_.empty())
          {
            if (stop_)
              return;
            continue;
          }
          job = queue_.front();
          queue_.pop_front();
          ++taken_;
        }
        {
          const auto now = std::chrono::steady_clock::now();
          hot_ = now - last_job_ < std::chrono::milliseconds(1);
          last_job_ = now;
        }
        job->fn();
        {
          std::lock_guard<std::mutex> lock(job->m);
          job->done.store(true, std::memory_order_release);
          if (job->waiting)
            job->cv.notify_one();
        }
      }
    }

    //! Waits for ready() for at most `limit` without monopolising a core:
    //! ~2 us of pause instructions, then sched_yield() between looks.
    template <typename Ready>
    static bool wait_briefly(Ready ready, std::chrono::microseconds limit)
    {
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 64; ++i)
      {
        if (ready())
          return true;
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#elif defined(__aarch64__)
        asm volatile("yield" ::: "memory");
#endif
      }
      while (!ready())
      {
        if (std::chrono::steady_clock::now() - t0 > limit)
          return false;
        std::this_thread::yield();
      }
      return true;
    }

    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Job*> queue_;
    // launcher thread only: arrival of the previous job, and whether the one
    // before it was less than a millisecond earlier
    std::chrono::steady_clock::time_point last_job_{};
    bool hot_ = false;
    std::atomic<unsigned long long> posted_{0};
    unsigned long long taken_ = 0;  // launcher thread only
    std::atomic<bool> sleeping_{false};
    std::atomic<std::thread::id> thread_id_{std::thread::id()};
    std::thread worker_;
    bool started_ = false, stop_ = false;
  };

  GraphLauncher& graph_launcher();
  //! ROCm < 7.2: see graph_launcher.cpp.
  bool graphs_need_one_thread();
  bool first_graph_thread();
  bool graph_budget_left();
  extern std::atomic<int> g_graph_instantiations;

  //! Streams of closed contexts are kept for the next context on the device
  //! (graph_launcher.cpp): creating nine of them is the slowest part of opening
  //! a context, and a process that keeps creating and destroying streams next
  //! to graph launches is where the ROCm 7.0 runtime came apart.
  hipError_t pooled_stream_acquire(int device, bool high_priority, hipStream_t* out);
  void pooled_stream_release(int device, bool high_priority, hipStream_t stream);

  // ---- error text of the calling thread (sift_schedule.cpp) -----------------
  extern thread_local std::string g_error;
  sara_hip_status fail(sara_hip_status code, const std::string& msg);

#define HIP_TRY(expr)                                                          \
  do                                                                           \
  {                                                                            \
    const hipError_t e_ = (expr);                                              \
    if (e_ != hipSuccess)                                                      \
      return fail(SARA_HIP_RUNTIME_ERROR, std::string(#expr) + ": " +          \
                                              hipGetErrorString(e_));          \
  } while (0)


  // ---- host restatement of the parameter schedule (sift_schedule.cpp) --------
  struct OctaveGeom
  {
    int w = 0, h = 0;
    float factor = 0.f;
  };

  struct Schedule
  {
    int base_w = 0, base_h = 0;  // octave 0 size
    float resize_factor = 1.f;
    int num_octaves = 0;
    int downscale_index = 0;
    bool init_blur = false;
    float init_sigma = 0.f;
    std::vector<OctaveGeom> oct;
  };


  std::vector<float> gaussian_taps(float sigma, float gauss_truncate,
                                   int arithmetic = SARA_HIP_TAPS_LIBM_SERIAL);
  bool to_taps(const std::vector<float>& k, Taps& t);
  Schedule make_schedule(const sara_pyramid_params& p, int w, int h,
                         bool downscale_at_double_sigma = false);
  sara_hip_status validate(const sara_pyramid_params& p, int padding);
  void orientation_bin_thresholds(float thr_out[40]);

//! Ints in d_counters (4 * max_batch + 4 used: the per-frame counters, the
//! frame offsets, the peak scan's arrival counter, the error flag, the step
//! stamp), in whole
//! 256-byte blocks; the last three ints are the graph's filler targets.
inline size_t counters_padded(int max_batch)
{
  return (4 * size_t(max_batch) + 2 + 8 + 63) / 64 * 64;  // >= 8 spare ints
}
//! Ints of d_counters that travel to the host with a batch's counts.
inline size_t counters_read(int max_batch)
{
  return 4 * size_t(max_batch) + 4;
}
//! The step stamp zero_counters_kernel leaves (the context's step number).
inline size_t step_stamp_index(int max_batch)
{
  return 4 * size_t(max_batch) + 3;
}
inline size_t error_flag_index(int max_batch)
{
  return 4 * size_t(max_batch) + 2;
}


  struct DeviceScratch
  {
    std::vector<void*> ptrs;
    ~DeviceScratch()
    {
      for (void* p : ptrs)
        (void) hipFree(p);
    }
    template <typename T>
    hipError_t get(T*& p, size_t count)
    {
      void* q = nullptr;
      const hipError_t e = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
      if (e == hipSuccess)
      {
        ptrs.push_back(q);
        p = static_cast<T*>(q);
      }
      return e;
    }
  };


}}  // namespace sara_hip::host

// (internal header: only the host translation units listed above include it)
using namespace sara_hip;
using namespace sara_hip::host;

struct sara_hip_sift
{
  int device = 0;
  sara_pyramid_params pyr{};
  float gauss_truncate = 4.f, extremum_thres = 0.01f, edge_ratio = 10.f;
  int img_padding = 1, refine_iters = 5;
  int max_w = 0, max_h = 0, max_batch = 0, cap = 0;
  int S = 6;

  hipStream_t own_stream = nullptr;
  hipStream_t last_stream = nullptr;
  // one auxiliary stream per octave > 0: the small octaves' launch-bound
  // chains overlap the big octave's bandwidth-bound kernels
  hipStream_t oct_stream[16] = {};
  hipEvent_t oct_ready[16] = {};  // G(downscale_index, o) is complete
  hipEvent_t oct_done[16] = {};   // octave o's chain is complete
  hipEvent_t scan_done[16] = {};  // octave o's extremum scan is complete
  // Octave pipelining: the extremum scan and the polar gradients of octave o
  // follow its last blur on the octave's own stream instead of waiting for
  // the whole pyramid.  -1 = automatic (graph replay, i.e. small batches,
  // where the dependent-launch chain is the bound), 0 / 1 = SARA_HIP_OCTAVE_PIPELINE
  int octave_pipeline = -1;
  bool multi_stream = true;
  // The polar gradients read the Gaussian pyramid only, like the extremum
  // scan: they are enqueued first, on a side stream, so that the short
  // latency-bound kernels of the extrema stage (refinement, ordering) run
  // next to them (3.14 -> 2.99 ms for the two stages; SARA_HIP_SIDE_GRADIENT=0
  // restores the sequential order and the separate stage times).
  bool side_gradient = true;
  hipStream_t aux_stream = nullptr;
  // set by detect_u8 for the duration of one detect(): the frames are 8-bit
  // gray in device memory and have NOT been converted into d_input yet
  const unsigned char* gray8_src = nullptr;
  size_t gray8_stride = 0;
  // graph replay only: streams / events of the filler nodes that steer the
  // runtime's node -> queue assignment (see the spine layout in detect)
  hipStream_t filler_stream[3] = {};
  hipEvent_t filler_done[3] = {};
  hipEvent_t aux_fork = nullptr, aux_join = nullptr;

  Schedule max_sched;
  Schedule cur;
  int cur_w = -1, cur_h = -1, cur_batch = 0;
  sara_hip_stage last_stage = SARA_HIP_STAGE_PYRAMID;
  bool has_result = false;
  bool all_gradient_scales = false;
  bool root_sift = false;
  bool signed_type = false;
  bool downscale_at_double_sigma = false;
  bool fma_blur = false;
  int tap_arithmetic = SARA_HIP_TAPS_LIBM_SERIAL;  // SARA_HIP_OPT_TAP_ARITHMETIC
  //! which kernels this context's launches take (SARA_HIP_OPT_KERNEL_SELECTION,
  //! _TILE_GEOMETRY, _MARCH_WAVES); a new context starts from the environment's
  KernelSelection sel = environment_selection();
  bool timers = true;

  // pyramids, one allocation per octave (sized for max dims / max batch).
  // The DoG pyramid is never materialised (consumers subtract on the fly);
  // d_dog_plane is the scratch of the diff_of_gaussians() accessor.
  std::vector<float*> G, GR;
  std::vector<unsigned*> CM;  // coarse 16x16 gradient-magnitude maxima
  float* d_dog_plane = nullptr;
  float* d_input = nullptr;  // staged host frames, or enlarge/blur scratch
  unsigned char* d_u8 = nullptr;  // staged 8-bit host frames (lazy)
  // double-buffered upload (sara_hip_sift_stage / _detect_staged), lazy
  void* d_stage[2] = {nullptr, nullptr};
  hipStream_t copy_stream = nullptr;
  hipEvent_t stage_ready[2] = {nullptr, nullptr};  // copy into buffer k done
  hipEvent_t stage_free[2] = {nullptr, nullptr};   // last pipeline using k done
  bool stage_used[2] = {false, false};
  int stage_next = 0;      // buffer the next stage() writes
  int staged = -1;         // buffer holding the batch detect_staged() will run
  int staged_channels = 0, staged_batch = 0, staged_w = 0, staged_h = 0;
  float* d_full = nullptr;   // first_octave > 0: blurred full-size frames

  // schedule constants
  bool have_init_taps = false;
  Taps init_taps{};
  std::vector<Taps> taps;  // per scale s = 1..S-1
  int* d_counters = nullptr;  // cand.count | sites.count | ori.kp_count | ori.frame_offset
  //! steps this context has run: bumped by zero_counters_kernel on the device
  //! (d_epoch, behind the cleared block) and by detect() on the host
  unsigned* d_epoch = nullptr;
  unsigned epoch_host = 0;
  bool epoch_synced = false;  // false: adopt the device's number at the next read-back
  ScaleTable h_tab{};
  ScaleTable* d_tab = nullptr;
  double* d_oriw = nullptr;
  int n_oriw = 0;
  GradPyramidView* h_grad = nullptr;  // pinned
  GradPyramidView* d_grad = nullptr;

  CandidateLists cand{};
  SiteLists sites{};
  OrientationLists ori{};
  int* d_ex_offset = nullptr;
  // Result buffers of the current detect().  detect()/fetch() always use slot
  // 0; the pipelined submit()/collect() pair alternates between two slots so
  // that batch i can be copied out while batch i + 1 is computed (slot 1 is
  // allocated on the first submit()).
  sara_oeregion* d_feat = nullptr;
  int32_t* d_so = nullptr;
  float* d_desc = nullptr;
  sara_oeregion* d_feat_s[2] = {nullptr, nullptr};
  int32_t* d_so_s[2] = {nullptr, nullptr};
  float* d_desc_s[2] = {nullptr, nullptr};
  int write_slot = 0;
  bool has_slot1 = false;  // the second result slot exists (first submit())
  // largest per-frame list length, in units of max_keypoints, that the last
  // examined batch asked for (sara_hip_sift_capacity)
  int required_cap = 0;
  struct RingSlot
  {
    int ticket = -1;
    bool pending = false;
    int batch = 0;
    sara_hip_stage stage = SARA_HIP_STAGE_DESCRIPTOR;  // last_stage of the submit()
    hipEvent_t done = nullptr;   // counters of the batch are in h_counters
    int* h_counters = nullptr;   // pinned copy of d_counters (counters_read())
    unsigned step = 0;           // the context's step number of this batch
    sara_oeregion* h_feat = nullptr;  // pinned result arrays, grown on demand
    float* h_desc = nullptr;
    int32_t* h_so = nullptr;
    size_t h_cap = 0;            // keypoints the pinned arrays hold
  } ring[2];
  // detect_staged(): recorded by detect() as soon as the last kernel that
  // reads the input frames has been enqueued (the staging buffer is free for
  // the next upload long before the batch is complete)
  hipEvent_t consumed_event = nullptr;
  bool consumed_recorded = false;
  hipStream_t d2h_stream = nullptr;
  int next_ticket = 0;
  sara_oeregion* d_ex_regions = nullptr;
  int32_t* d_ex_xyso = nullptr;

  // counting sort of the extrema (launch_rank_candidates_bucketed)
  int* d_bucket_hist = nullptr;    // [max_batch][bucket_stride]
  int* d_bucket_cursor = nullptr;  // [max_batch][bucket_stride]
  int* d_grouped = nullptr;        // [max_batch][cap]
  int bucket_stride = 0;
  RowBuckets row_buckets{};        // of the current schedule
  int* h_counts = nullptr;  // pinned, counters_read(max_batch)
  // Small batches are launch-bound (about 60 launches in 0.7 ms for one 1080p
  // frame): the enqueue sequence of detect() is captured once per (size,
  // batch, stage) into a HIP graph and replayed (SARA_HIP_GRAPH=0 disables,
  // SARA_HIP_GRAPH_MAX_BATCH, default 16, bounds the batch sizes that use it:
  // 1080p frames resident in HBM, 9 / 12 / 16 / 24 / 32 frames per call replayed against
  // launched: 1.03 / 1.31 / 1.65 / 2.43 / 3.20 against 1.11 / 1.39 / 1.73 / 2.46 / 3.13 ms).
  bool use_graph = true;
  int graph_max_batch = 16;
  // one captured graph per result slot (the result pointers are kernel
  // arguments baked into the capture)
  hipGraph_t graph_s[2] = {nullptr, nullptr};
  hipGraphExec_t graph_exec_s[2] = {nullptr, nullptr};
  int graph_w_s[2] = {0, 0}, graph_h_s[2] = {0, 0}, graph_batch_s[2] = {0, 0},
      graph_stage_s[2] = {-1, -1};
  bool graph_broken = false;  // a capture failed once: stay on plain launches
  // Round 3: device-resident frames are read IN PLACE by the replayed graph.
  // The captured kernels that take the frames as their first argument are
  // remembered per slot; when the caller's pointer changes, their argument is
  // rewritten in the executable graph (hipGraphExecKernelNodeSetParams)
  // instead of copying the frames to a fixed address first (8.3 MB and one
  // more enqueue per 1080p call); cleared for good when the runtime cannot
  // rewrite a captured kernel's argument (the copy comes back).
  bool graph_inplace = true;
  const void* graph_src_s[2] = {nullptr, nullptr};     // pointer baked into the slot's graph
  size_t graph_src_stride_s[2] = {0, 0};
  std::vector<hipGraphNode_t> graph_src_nodes_s[2];    // kernels reading it
  hipEvent_t ev[SARA_HIP_TIME_COUNT + 1] = {};
  bool ev_recorded[SARA_HIP_TIME_COUNT + 1] = {};
  // SARA_HIP_OPT_LAUNCH_TIMERS: one event pair around every launch of the
  // pyramid stage (plain launches only), read by sara_hip_sift_pyramid_launches
  bool launch_timers = false;
  struct LaunchRecord
  {
    hipEvent_t begin = nullptr, end = nullptr;
    int octave = 0, scale = 0, taps = 0;
    long long pixels = 0;
  };
  std::vector<LaunchRecord> launch_rec;
  int launch_count = 0;

  std::vector<void*> allocations;

  template <typename T>
  sara_hip_status alloc(T*& p, size_t count)
  {
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
    allocations.push_back(q);
    p = static_cast<T*>(q);
    return SARA_HIP_OK;
  }

  float* plane(std::vector<float*>& pyr_, int o, int frame, int s, int chans,
               int scales) const
  {
    const size_t pl = size_t(cur.oct[o].w) * cur.oct[o].h * chans;
    return pyr_[o] + (size_t(frame) * scales + s) * pl;
  }
};

namespace sara_hip { namespace host {
  // ---- context helpers (sift_context.cpp) -------------------------------------
  sara_hip_status alloc_lists(sara_hip_sift* c);
  void free_lists(sara_hip_sift* c);
  size_t list_bytes_per_entry(const sara_hip_sift* c);
  const char* compute_taps(sara_hip_sift* c);
  sara_hip_status require_result(const sara_hip_sift* ctx, sara_hip_stage need);
  bool counters_corrupt(sara_hip_sift* c, const int* h, int mb, int batch, int lists,
                        unsigned expected_step);
  sara_hip_status corrupt_counters_error();
  void note_required(sara_hip_sift* c, const int* h_ex, const int* h_sites,
                     const int* h_kp, int batch);
  sara_hip_status select_device(int device);
}}  // namespace sara_hip::host
