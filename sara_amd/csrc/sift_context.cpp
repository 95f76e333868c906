// Host side of the C-ABI (include/sara_hip_sift.h): parameter schedule,
// HBM buffer ownership, stage sequencing.  Mirrors the control flow of
//   compute_sift_keypoints     FeatureDetectors/SIFT.cpp:27-108
//   ComputeDoGExtrema::op()    FeatureDetectors/DoG.cpp:23-87
//   gaussian_pyramid           ImageProcessing/GaussianPyramid.hpp:33-125
// but batched over frames and with every stage resident in HBM.
#include "sift_kernels.hpp"

#include "device_math.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <limits>
#include <mutex>
#include <pthread.h>
#include <string>
#include <thread>
#include <vector>

using namespace sara_hip;

namespace sara_hip {
  // Contexts of different host threads are independent - except inside the
  // ROCm 7 runtime.  hipGraphLaunch keeps the streams its parallel branches
  // run on in per-device state that is not protected against other threads
  // creating / destroying streams, graphs and executables, or capturing and
  // launching graphs themselves: rocgdb shows the segmentation fault in
  // hip::Graph::UpdateStreams <- hip::GraphExec::Run <- hipGraphLaunch, with
  // the other threads inside context creation / destruction (seen with one
  // host thread per logical rank in sara_hip_sift_group_* and with the
  // per-thread context caches of compute_sift_keypoints).  Captures, graph
  // launches (host side only: tens of microseconds), and the creation /
  // destruction of contexts, streams and graphs therefore exclude each other
  // process-wide.  Plain kernel launches and copies take no lock.
  std::recursive_mutex& runtime_mutex()
  {
    static std::recursive_mutex m;
    return m;
  }
}  // namespace sara_hip

namespace {

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

  std::atomic<GraphLauncher*> g_launcher{nullptr};

  GraphLauncher& graph_launcher()
  {
    // leaked on purpose: at process exit the HIP runtime may already be gone
    // when static destructors run, and the launcher only ever sleeps by then
    static const bool once = [] {
      g_launcher.store(new GraphLauncher, std::memory_order_release);
      // fork(): the child inherits the launcher's state (started, perhaps a
      // locked mutex) but not its thread - the first run() would wait for
      // ever.  The child gets a fresh launcher; the old one is abandoned.
      // This only keeps a forked child from HANGING inside the library: the
      // HIP runtime itself does not survive fork(), GPU work in the child is
      // not supported (spawn, or fork before the first call).
      pthread_atfork(nullptr, nullptr, [] {
        g_launcher.store(new GraphLauncher, std::memory_order_release);
      });
      return true;
    }();
    (void) once;
    return *g_launcher.load(std::memory_order_acquire);
  }

  //! ROCm runtimes before 7.2 (the 7.0 runtime bundled with torch 2.10 is what
  //! a Python caller that imported torch first runs on) crash in
  //! hip::Graph::UpdateStreams - hipGraphLaunch reads a stale entry of the
  //! executable's parallel-stream list - once contexts with graphs are
  //! created, replayed and destroyed by several host threads, even with every
  //! graph call on the launcher thread; serialising every call of the library
  //! does not prevent it, a wide dummy graph launched first does not either
  //! (tools/churn_repro.py: 3 of 3 runs die; none on ROCm 7.2).  On those
  //! runtimes graph replay therefore stays with the first host thread that
  //! asks for it, as in round 3, and the other threads' contexts run plain
  //! launches (+ 0.15 ms of host time per 1080p frame); on ROCm >= 7.2 every
  //! thread replays graphs through the launcher.
  bool graphs_need_one_thread()
  {
    // fail closed: a runtime that does not say what it is counts as old
    static const bool old_runtime = [] {
      int v = 0;
      return hipRuntimeGetVersion(&v) != hipSuccess || v < 70200000;
    }();
    return old_runtime;
  }
  //! Second rule for those runtimes (round 6).  One thread is not enough: a
  //! single thread that keeps creating contexts, capturing and destroying
  //! graphs dies in the same place (hipGraphLaunch -> hip::Graph::UpdateStreams,
  //! rocgdb backtrace on the launcher thread) once enough graphs have come and
  //! gone in the process - the full GPU test suite did, deterministically, after
  //! 215 instantiations when round 6 added 60 contexts to it, after about 290
  //! with other tests left out, and earlier still when destroyed executables
  //! were kept alive instead (so it is not the destruction).  A process on such
  //! a runtime therefore instantiates at most kOldRuntimeGraphBudget graphs
  //! (SARA_HIP_GRAPH_MAX_INSTANTIATIONS overrides); contexts that need a new
  //! graph after that run plain launches (+ 0.15 ms of host time per 1080p
  //! frame), contexts that have theirs keep replaying it.  A video pipeline
  //! uses one or two graphs; the budget only matters to processes that see
  //! hundreds of frame sizes or parameter sets.  ROCm >= 7.2: no limit.
  constexpr int kOldRuntimeGraphBudget = 128;
  std::atomic<int> g_graph_instantiations{0};
  bool graph_budget_left()
  {
    if (!graphs_need_one_thread())
      return true;
    static const int limit = [] {
      const char* e = getenv("SARA_HIP_GRAPH_MAX_INSTANTIATIONS");
      return e ? atoi(e) : kOldRuntimeGraphBudget;
    }();
    return g_graph_instantiations.load(std::memory_order_relaxed) < limit;
  }
  bool first_graph_thread()
  {
    static std::atomic<std::thread::id> first{std::thread::id()};
    std::thread::id none, me = std::this_thread::get_id();
    if (first.compare_exchange_strong(none, me))
      return true;
    return first.load() == me;
  }

  thread_local std::string g_error = "";

  sara_hip_status fail(sara_hip_status code, const std::string& msg)
  {
    g_error = msg;
    return code;
  }

#define HIP_TRY(expr)                                                          \
  do                                                                           \
  {                                                                            \
    const hipError_t e_ = (expr);                                              \
    if (e_ != hipSuccess)                                                      \
      return fail(SARA_HIP_RUNTIME_ERROR, std::string(#expr) + ": " +          \
                                              hipGetErrorString(e_));          \
  } while (0)

  // ---- kernel selection (sift_kernels.hpp) ---------------------------------
  thread_local const KernelSelection* t_selection = nullptr;
}  // namespace

namespace sara_hip {
  const KernelSelection& environment_selection()
  {
    static const KernelSelection env = [] {
      KernelSelection k;
      auto is = [](const char* name, const char* value) {
        const char* e = getenv(name);
        return e && std::string(e) == value;
      };
      k.blur_march = !is("SARA_HIP_BLUR", "tile");
      k.feature_march = !is("SARA_HIP_FEATURES", "tile");
      if (const char* e = getenv("SARA_HIP_MARCH_WAVES"))
        k.march_waves = std::max(64, atoi(e));
      if (const char* e = getenv("SARA_HIP_MARCH2_WAVES"))
        k.march2_waves = std::max(64, atoi(e));
      if (const char* e = getenv("SARA_HIP_MARCH_MIN_PIXELS"))
        k.march_min_pixels = size_t(atoll(e));
      if (const char* e = getenv("SARA_HIP_STRIP_GROUP"))
        k.strip_group = atoi(e);
      if (const char* e = getenv("SARA_HIP_GRAD_TILE_PIXELS"))
        k.grad_tile_pixels = atoll(e);
      if (const char* e = getenv("SARA_HIP_TILE_GEOMETRY"))
        k.tile_geometry = atoi(e);
      k.xcd_map = !is("SARA_HIP_XCD_MAP", "0");
      return k;
    }();
    return env;
  }
  const KernelSelection& selection()
  {
    return t_selection ? *t_selection : environment_selection();
  }
  ScopedSelection::ScopedSelection(const KernelSelection* s)
    : before{t_selection}
  {
    t_selection = s;
  }
  ScopedSelection::~ScopedSelection() { t_selection = before; }
}  // namespace sara_hip

namespace {
  // ---- host restatement of the parameter schedule --------------------------

  // make_gaussian_kernel, ImageProcessing/LinearFiltering.hpp:171-203, is three
  // Eigen expressions; exp() and sum() are the two operations in them that are
  // not one correctly rounded IEEE operation, so their result depends on the
  // path Eigen takes in the reference's build (SARA_HIP_TAPS_*):
  //  * a scalar build: expf per tap, left-to-right sum;
  //  * the Release build (x86-64 baseline = SSE2, Packet4f): the dense
  //    assignment loop sends taps [0, 4*(n/4)) through pexp<Packet4f> and the
  //    rest through the scalar functor (expf); sum() keeps two packet
  //    accumulators over even / odd packets, adds them, adds the odd packet out,
  //    reduces as (a0 + a2) + (a1 + a3) and finishes with the scalar tail.
  // pexp is written from the published algorithm (Cephes: m = floor(x log2 e +
  // 1/2), r = x - m ln 2 in two parts, degree-5 polynomial, times 2^m); on SSE2
  // pmadd is a multiply and an add, each rounded (this file is compiled with
  // -ffp-contract=off).

  //! Eigen 3.4 pexp_float, one lane.
  float pexp_eigen34(float x0)
  {
    const float x = std::max(std::min(x0, 88.723f), -88.723f);
    const float m = std::floor(x * 1.44269504088896341f + 0.5f);
    float r = m * -0.693359375f + x;
    r = m * 2.12194440e-4f + r;
    const float r2 = r * r, r3 = r2 * r;
    float y = 1.9875691500E-4f * r + 1.3981999507E-3f;
    float y1 = 4.1665795894E-2f * r + 1.6666665459E-1f;
    const float y2 = r + 1.0f;
    y = y * r + 8.3334519073E-3f;
    y1 = y1 * r + 5.0000001201E-1f;
    y = y * r3 + y1;
    y = y * r2 + y2;
    return std::max(std::ldexp(y, int(m)), x0);
  }

  //! Eigen 3.3 pexp<Packet4f>, one lane: Horner form, (P(r) r^2 + r) + 1.
  float pexp_eigen33(float x0)
  {
    float x = std::max(std::min(x0, 88.3762626647950f), -88.3762626647949f);
    const float fx = std::floor(x * 1.44269504088896341f + 0.5f);
    const float hi = fx * 0.693359375f;
    float z = fx * -2.12194440e-4f;
    x = x - hi;
    x = x - z;
    z = x * x;
    float y = 1.9875691500E-4f;
    const float p[5] = {1.3981999507E-3f, 8.3334519073E-3f, 4.1665795894E-2f,
                        1.6666665459E-1f, 5.0000001201E-1f};
    for (float pi : p)
      y = y * x + pi;
    y = y * z + x;
    y = y + 1.0f;
    return std::max(std::ldexp(y, int(fx)), x0);
  }

  //! VectorXf::sum() on SSE2 (Redux.h, LinearVectorizedTraversal).
  float sum_eigen_sse2(const float* v, int n)
  {
    const int n4 = (n / 4) * 4, n8 = (n / 8) * 8;
    if (n4 == 0)
    {
      float res = v[0];
      for (int i = 1; i < n; ++i)
        res = res + v[i];
      return res;
    }
    float a[4] = {v[0], v[1], v[2], v[3]};
    if (n4 > 4)
    {
      float b[4] = {v[4], v[5], v[6], v[7]};
      for (int i = 8; i < n8; i += 8)
        for (int j = 0; j < 4; ++j)
        {
          a[j] = a[j] + v[i + j];
          b[j] = b[j] + v[i + 4 + j];
        }
      for (int j = 0; j < 4; ++j)
        a[j] = a[j] + b[j];
      if (n4 > n8)
        for (int j = 0; j < 4; ++j)
          a[j] = a[j] + v[n8 + j];
    }
    float res = (a[0] + a[2]) + (a[1] + a[3]);
    for (int i = n4; i < n; ++i)
      res = res + v[i];
    return res;
  }

  //! make_gaussian_kernel, ImageProcessing/LinearFiltering.hpp:171-203.
  std::vector<float> gaussian_taps(float sigma, float gauss_truncate,
                                   int arithmetic = SARA_HIP_TAPS_LIBM_SERIAL)
  {
    int size = int(2 * gauss_truncate * sigma + 1);
    size = std::max(3, size);
    if (size % 2 == 0)
      ++size;
    const int c = size / 2;
    std::vector<float> k(size);
    const float denom = 2 * (sigma * sigma);
    const bool packets = arithmetic != SARA_HIP_TAPS_LIBM_SERIAL;
    const int packets_end = packets ? (size / 4) * 4 : 0;
    for (int i = 0; i < size; ++i)
    {
      const float d = float(i) - float(c);
      const float x = -(d * d) / denom;
      if (i >= packets_end)
        k[i] = std::exp(x);
      else
        k[i] = arithmetic == SARA_HIP_TAPS_EIGEN34_SSE2 ? pexp_eigen34(x)
                                                        : pexp_eigen33(x);
    }
    float sum = 0.f;
    if (packets)
      sum = sum_eigen_sse2(k.data(), size);
    else
      for (int i = 0; i < size; ++i)
        sum += k[i];
    for (int i = 0; i < size; ++i)
      k[i] /= sum;
    return k;
  }

  bool to_taps(const std::vector<float>& k, Taps& t)
  {
    if (int(k.size()) > kMaxTaps)
      return false;
    t.size = int(k.size());
    std::memset(t.k, 0, sizeof(t.k));
    std::memcpy(t.k, k.data(), sizeof(float) * k.size());
    return true;
  }

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

  //! Geometry part of gaussian_pyramid(), GaussianPyramid.hpp:43-122.
  Schedule make_schedule(const sara_pyramid_params& p, int w, int h,
                         bool downscale_at_double_sigma = false)
  {
    Schedule s;
    s.resize_factor = std::pow(2.f, -static_cast<float>(p.first_octave_index));
    const float camera_sigma = p.scale_camera * s.resize_factor;
    const float init_sigma = p.scale_initial;
    if (p.first_octave_index < 0)
    {
      s.base_w = int(double(w) * double(s.resize_factor));
      s.base_h = int(double(h) * double(s.resize_factor));
    }
    else
    {
      if (camera_sigma < init_sigma)
      {
        s.init_blur = true;
        s.init_sigma =
            std::sqrt(init_sigma * init_sigma - camera_sigma * camera_sigma);
      }
      if (p.first_octave_index > 0)
      {
        const int f = int(std::round(1 / s.resize_factor));
        s.base_w = f > 0 ? w / f : 0;
        s.base_h = f > 0 ? h / f : 0;
      }
      else
      {
        s.base_w = w;
        s.base_h = h;
      }
    }
    const int l = std::min(s.base_w, s.base_h);
    const int b = p.image_padding_size;
    int n = 0;
    if (l > 0 && b > 0)
      n = std::min(static_cast<int>(std::log(double(float(l) / (2.f * float(b)))) /
                                    std::log(double(2.f))),
                   p.num_octaves_max);
    s.num_octaves = std::max(n, 0);
    // GaussianPyramid.hpp:97-100: floor(); round() is the scale at 2 sigma_0
    // the float value of k misses (SARA_HIP_OPT_DOWNSCALE_AT_DOUBLE_SIGMA).
    const double per_doubling =
        std::log(double(2.f)) / std::log(double(p.scale_geometric_factor));
    s.downscale_index = static_cast<int>(
        downscale_at_double_sigma ? std::round(per_doubling) : std::floor(per_doubling));
    s.oct.resize(s.num_octaves);
    for (int o = 0; o < s.num_octaves; ++o)
    {
      s.oct[o].factor = (o == 0) ? 1 / s.resize_factor : s.oct[o - 1].factor * 2;
      s.oct[o].w = (o == 0) ? s.base_w : s.oct[o - 1].w / 2;
      s.oct[o].h = (o == 0) ? s.base_h : s.oct[o - 1].h / 2;
    }
    return s;
  }

  sara_hip_status validate(const sara_pyramid_params& p, int padding)
  {
    if (p.scale_count_per_octave < 4)
      return fail(SARA_HIP_INVALID_PARAMS,
                  "Error: The extraction of DoG extrema needs (1 + 3) = 4 "
                  "scales per octave at the very minimum!");
    if (p.scale_count_per_octave > kMaxScales)
      return fail(SARA_HIP_INVALID_PARAMS, "scale_count_per_octave > 16");
    if (!(p.scale_geometric_factor > 1.f))
      return fail(SARA_HIP_INVALID_PARAMS, "scale_geometric_factor must be > 1");
    if (p.image_padding_size < 1)
      return fail(SARA_HIP_INVALID_PARAMS, "image_padding_size must be >= 1");
    if (padding < 1)
      return fail(SARA_HIP_INVALID_PARAMS,
                  "the extremum border padding must be >= 1 (the reference "
                  "reads out of bounds below that)");
    if (!(p.scale_initial > 0.f) || !(p.scale_camera >= 0.f))
      return fail(SARA_HIP_INVALID_PARAMS, "scales must be positive");
    return SARA_HIP_OK;
  }

}  // namespace

//! ScaleTable::ori_bin_thr: thr[k] = smallest float >= 0 whose histogram bin
//! int(floor(double(a / float(2 pi) * 36))) (Orientation.hpp:118-119) is >= k,
//! by bisection on the bit patterns with the expression itself (+inf where no
//! angle of [0, 2 pi] gets there).
static void orientation_bin_thresholds(float thr_out[40])
{
  auto bin_of = [](float a) {
    return int(std::floor(double(a / float(2 * M_PI) * 36)));
  };
  for (int kk = 0; kk < 40; ++kk)
  {
    uint32_t lo = 0u, hi = 0x40c91000u;  // [0, a little above float(2 pi)]
    float thr = std::numeric_limits<float>::infinity();
    float top;
    std::memcpy(&top, &hi, 4);
    if (bin_of(top) >= kk)
    {
      while (lo < hi)  // first bit pattern (= first float >= 0) with bin >= kk
      {
        const uint32_t mid = lo + (hi - lo) / 2;
        float a;
        std::memcpy(&a, &mid, 4);
        if (bin_of(a) >= kk)
          hi = mid;
        else
          lo = mid + 1;
      }
      std::memcpy(&thr, &lo, 4);
    }
    thr_out[kk] = thr;
  }
}

//! Ints in d_counters (4 * max_batch + 4 used: the per-frame counters, the
//! frame offsets, the peak scan's arrival counter, the error flag, the step
//! stamp), in whole
//! 256-byte blocks; the last three ints are the graph's filler targets.
static inline size_t counters_padded(int max_batch)
{
  return (4 * size_t(max_batch) + 2 + 8 + 63) / 64 * 64;  // >= 8 spare ints
}
//! Ints of d_counters that travel to the host with a batch's counts.
static inline size_t counters_read(int max_batch)
{
  return 4 * size_t(max_batch) + 4;
}
//! The step stamp zero_counters_kernel leaves (the context's step number).
static inline size_t step_stamp_index(int max_batch)
{
  return 4 * size_t(max_batch) + 3;
}
static inline size_t error_flag_index(int max_batch)
{
  return 4 * size_t(max_batch) + 2;
}

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
  // SARA_HIP_GRAPH_MAX_BATCH, default 8, bounds the batch sizes that use it).
  bool use_graph = true;
  int graph_max_batch = 8;
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

namespace {

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
      return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
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
      TRY_HIP(hipStreamCreateWithFlags(&c->filler_stream[k], hipStreamNonBlocking));
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

}  // namespace

namespace {
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
}  // namespace

extern "C" {

const char* sara_hip_last_error(void) { return g_error.c_str(); }

int sara_hip_version(void) { return 100; }

int sara_hip_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

void sara_hip_default_pyramid_params(sara_pyramid_params* p)
{
  p->first_octave_index = -1;
  p->scale_count_per_octave = 3 + 3;
  p->scale_geometric_factor = std::pow(2.f, 1.f / 3.f);
  p->image_padding_size = 1;
  p->scale_camera = 0.5f;
  p->scale_initial = 1.6f;
  p->num_octaves_max = INT_MAX;
}

void sara_hip_default_sift_params(sara_sift_params* p)
{
  sara_hip_default_pyramid_params(&p->pyramid);
  p->gauss_truncate = 4.f;
  p->extremum_thres = 0.01f;
  p->edge_ratio_thres = 10.f;
  p->extremum_refinement_iter = 5;
}

int sara_hip_pyramid_octave_count(const sara_pyramid_params* p, int width,
                                  int height)
{
  if (!p)
    return 0;
  return make_schedule(*p, width, height).num_octaves;
}

sara_hip_status sara_hip_pyramid_octave_info(const sara_pyramid_params* p,
                                             int width, int height, int octave,
                                             int* ow, int* oh, float* factor)
{
  if (!p)
    return fail(SARA_HIP_INVALID_PARAMS, "null params");
  const Schedule s = make_schedule(*p, width, height);
  if (octave < 0 || octave >= s.num_octaves)
    return fail(SARA_HIP_OUT_OF_RANGE, "octave index out of range");
  if (ow)
    *ow = s.oct[octave].w;
  if (oh)
    *oh = s.oct[octave].h;
  if (factor)
    *factor = s.oct[octave].factor;
  return SARA_HIP_OK;
}

int sara_hip_make_gaussian_kernel(float sigma, float gauss_truncate, float* taps,
                                  int capacity)
{
  const auto k = gaussian_taps(sigma, gauss_truncate);
  if (int(k.size()) > capacity || !taps)
    return -int(k.size());
  std::memcpy(taps, k.data(), sizeof(float) * k.size());
  return int(k.size());
}

int sara_hip_make_gaussian_kernel_with(int arithmetic, float sigma,
                                       float gauss_truncate, float* taps,
                                       int capacity)
{
  if (arithmetic < SARA_HIP_TAPS_LIBM_SERIAL || arithmetic > SARA_HIP_TAPS_EIGEN33_SSE2)
    return 0;
  const auto k = gaussian_taps(sigma, gauss_truncate, arithmetic);
  if (int(k.size()) > capacity || !taps)
    return -int(k.size());
  std::memcpy(taps, k.data(), sizeof(float) * k.size());
  return int(k.size());
}

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
    if (c->oct_stream[o])
    {
      (void) hipStreamSynchronize(c->oct_stream[o]);
      (void) hipStreamDestroy(c->oct_stream[o]);
    }
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
  if (c->copy_stream)
  {
    (void) hipStreamSynchronize(c->copy_stream);
    (void) hipStreamDestroy(c->copy_stream);
  }
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
  if (c->d2h_stream)
  {
    (void) hipStreamSynchronize(c->d2h_stream);
    (void) hipStreamDestroy(c->d2h_stream);
  }
  for (int k = 0; k < 3; ++k)
  {
    if (c->filler_stream[k])
      (void) hipStreamDestroy(c->filler_stream[k]);
    if (c->filler_done[k])
      (void) hipEventDestroy(c->filler_done[k]);
  }
  if (c->aux_stream)
  {
    (void) hipStreamSynchronize(c->aux_stream);
    (void) hipStreamDestroy(c->aux_stream);
  }
  if (c->aux_fork)
    (void) hipEventDestroy(c->aux_fork);
  if (c->aux_join)
    (void) hipEventDestroy(c->aux_join);
  if (c->own_stream)
    (void) hipStreamDestroy(c->own_stream);
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
  const bool graph_mode = c->use_graph && !c->graph_broken && !hip_stream &&
                          batch <= c->graph_max_batch && !debug_sync &&
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
    if (pipe)
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
  if (pipe)
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

namespace {
  //! The read-back stream.
  sara_hip_status ensure_d2h_stream(sara_hip_sift* c)
  {
    if (c->d2h_stream)
      return SARA_HIP_OK;
    // highest priority: the read-back kernel's few workgroups should not
    // queue behind the next batch's launches
    int lo = 0, hi = 0;
    std::lock_guard<std::recursive_mutex> runtime_lock(runtime_mutex());
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIP_TRY(hipStreamCreateWithPriority(&c->d2h_stream, hipStreamNonBlocking, hi));
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
    HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
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

// ---- operator-level seams --------------------------------------------------

sara_hip_status sara_hip_apply_gaussian_filter(const float* src, float* dst,
                                               int w, int h, float sigma,
                                               float gauss_truncate, int device)
{
  if (!src || !dst || w < 1 || h < 1)
    return fail(SARA_HIP_SIZE_MISMATCH,
                "Source and destination image sizes are not equal!");
  Taps taps;
  if (!to_taps(gaussian_taps(sigma, gauss_truncate), taps))
    return fail(SARA_HIP_INVALID_PARAMS, "Gaussian needs more than 113 taps");
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  DeviceScratch sc;
  float *ds = nullptr, *dd = nullptr;
  const size_t n = size_t(w) * h;
  HIP_TRY(sc.get(ds, n));
  HIP_TRY(sc.get(dd, n));
  HIP_TRY(hipMemcpy(ds, src, n * sizeof(float), hipMemcpyHostToDevice));
  launch_gaussian_blur(ds, n, dd, n, nullptr, 0, w, h, 1, taps, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(dst, dd, n * sizeof(float), hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_scale(const float* src, int sw, int sh, float* dst,
                               int dw, int dh, int device)
{
  if (!src || !dst || sw < 1 || sh < 1 || dw < 1 || dh < 1)
    return fail(SARA_HIP_INVALID_PARAMS, "bad image sizes");
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  DeviceScratch sc;
  float *ds = nullptr, *dd = nullptr;
  HIP_TRY(sc.get(ds, size_t(sw) * sh));
  HIP_TRY(sc.get(dd, size_t(dw) * dh));
  HIP_TRY(hipMemcpy(ds, src, size_t(sw) * sh * sizeof(float),
                    hipMemcpyHostToDevice));
  launch_scale(ds, 0, sw, sh, dd, 0, dw, dh, 1, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(dst, dd, size_t(dw) * dh * sizeof(float),
                    hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_enlarge(const float* src, int sw, int sh, float* dst,
                                 int dw, int dh, int device)
{
  if (!src || !dst)
    return fail(SARA_HIP_INVALID_PARAMS, "null image");
  if (dw < sw || dh < sh)
    return fail(SARA_HIP_OUT_OF_RANGE,
                "The destination image must have smaller sizes than the source "
                "image!");
  if (std::min(dw, dh) <= 0 || sw < 1 || sh < 1)
    return fail(SARA_HIP_OUT_OF_RANGE,
                "The sizes of the destination image must be positive!");
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  DeviceScratch sc;
  float *ds = nullptr, *dd = nullptr;
  HIP_TRY(sc.get(ds, size_t(sw) * sh));
  HIP_TRY(sc.get(dd, size_t(dw) * dh));
  HIP_TRY(hipMemcpy(ds, src, size_t(sw) * sh * sizeof(float),
                    hipMemcpyHostToDevice));
  launch_enlarge(ds, 0, sw, sh, dd, 0, dw, dh, 1, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(dst, dd, size_t(dw) * dh * sizeof(float),
                    hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_subtract(const float* a, const float* b, float* out,
                                  size_t count, int device)
{
  if (!a || !b || !out)
    return fail(SARA_HIP_INVALID_PARAMS, "null operand");
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  DeviceScratch sc;
  float *da = nullptr, *db = nullptr, *dout = nullptr;
  HIP_TRY(sc.get(da, count));
  HIP_TRY(sc.get(db, count));
  HIP_TRY(sc.get(dout, count));
  HIP_TRY(hipMemcpy(da, a, count * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(db, b, count * sizeof(float), hipMemcpyHostToDevice));
  launch_subtract(da, db, dout, count, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dout, count * sizeof(float), hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

static sara_hip_status u8_to_gray(const uint8_t* src, float* gray, int w, int h,
                                  int channels, int device)
{
  if (!src || !gray || w < 1 || h < 1)
    return fail(SARA_HIP_SIZE_MISMATCH,
                "Color conversion error: image sizes are not equal!");
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  DeviceScratch sc;
  unsigned char* ds = nullptr;
  float* dd = nullptr;
  const size_t n = size_t(w) * h;
  HIP_TRY(sc.get(ds, n * channels));
  HIP_TRY(sc.get(dd, n));
  HIP_TRY(hipMemcpy(ds, src, n * channels, hipMemcpyHostToDevice));
  launch_u8_to_gray32f(ds, 0, channels, dd, 0, n, 1, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(gray, dd, n * sizeof(float), hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_from_rgb8_to_gray32f(const uint8_t* rgb, float* gray,
                                              int w, int h, int device)
{
  return u8_to_gray(rgb, gray, w, h, 3, device);
}

sara_hip_status sara_hip_from_gray8_to_gray32f(const uint8_t* src, float* gray,
                                               int w, int h, int device)
{
  return u8_to_gray(src, gray, w, h, 1, device);
}

sara_hip_status sara_hip_root_sift(float* desc, int n, int dim, int on_device,
                                   int device)
{
  if (!desc || n < 0 || dim < 1)
    return fail(SARA_HIP_INVALID_PARAMS, "null pointer, negative count or empty rows");
  if (n == 0)
    return SARA_HIP_OK;
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  DeviceScratch sc;
  float* d = desc;
  const size_t bytes = size_t(n) * dim * sizeof(float);
  if (!on_device)
  {
    HIP_TRY(sc.get(d, size_t(n) * dim));
    HIP_TRY(hipMemcpy(d, desc, bytes, hipMemcpyHostToDevice));
  }
  launch_root_sift(d, n, dim, nullptr);
  HIP_TRY(hipGetLastError());
  if (!on_device)
    HIP_TRY(hipMemcpy(desc, d, bytes, hipMemcpyDeviceToHost));
  else
    HIP_TRY(hipStreamSynchronize(nullptr));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_gradient_polar_coordinates(const float* src, int w,
                                                    int h, float* mag_ori,
                                                    int device)
{
  if (!src || !mag_ori || w < 2 || h < 2)
    return fail(SARA_HIP_INVALID_PARAMS, "image must be at least 2x2");
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  DeviceScratch sc;
  float *ds = nullptr, *dd = nullptr;
  const size_t n = size_t(w) * h;
  HIP_TRY(sc.get(ds, n));
  HIP_TRY(sc.get(dd, 2 * n));
  HIP_TRY(hipMemcpy(ds, src, n * sizeof(float), hipMemcpyHostToDevice));
  launch_gradient_polar(ds, n, dd, 2 * n, w, h, 1, 1, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(mag_ori, dd, 2 * n * sizeof(float), hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_scale_space_dog_extremum_map(
    const float* a, const float* b, const float* c, int w, int h,
    float edge_ratio_thres, float extremum_thres, int img_padding_sz,
    int8_t* out, int device)
{
  if (!a || !b || !c || !out || w < 3 || h < 3)
    return fail(SARA_HIP_INVALID_PARAMS, "layers must be at least 3x3");
  if (img_padding_sz < 0)
    return fail(SARA_HIP_INVALID_PARAMS, "img_padding_sz must be >= 0");
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  DeviceScratch sc;
  float *da = nullptr, *db = nullptr, *dc = nullptr;
  int8_t* dout = nullptr;
  const size_t n = size_t(w) * h;
  HIP_TRY(sc.get(da, n));
  HIP_TRY(sc.get(db, n));
  HIP_TRY(sc.get(dc, n));
  HIP_TRY(sc.get(dout, n));
  HIP_TRY(hipMemcpy(da, a, n * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(db, b, n * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dc, c, n * sizeof(float), hipMemcpyHostToDevice));
  launch_extremum_map(da, db, dc, w, h, edge_ratio_thres, extremum_thres,
                      img_padding_sz, dout, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dout, n, hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

void sara_hip_selfcheck_atan2f(const float* y, const float* x, float* out,
                               size_t count)
{
  for (size_t i = 0; i < count; ++i)
  {
    // both restatements the kernels use must agree; a mismatch is reported
    // as NaN so that the comparison with libm fails
    static const float tab[sara_hip::kAtanTableFloats] = SARA_ATAN_TABLE_INIT;
    static const std::vector<float> lut = [] {
      std::vector<float> l(sara_hip::kAtanLutFloats);
      for (int j = 0; j < sara_hip::kAtanLutRows; ++j)
        for (int q = 0; q < 8; ++q)
          l[size_t(8 * j + q)] = tab[8 * sara_hip::atan_lut_source_row(j) + q];
      return l;
    }();
    const float a = sara_hip::fdlibm_atan2f_fast(y[i], x[i]);
    const float b = sara_hip::fdlibm_atan2f_table(y[i], x[i], tab);
    const float c = sara_hip::fdlibm_atan2f_lut(y[i], x[i], lut.data());
    const bool same = std::memcmp(&a, &b, sizeof(float)) == 0 &&
                      std::memcmp(&a, &c, sizeof(float)) == 0;
    out[i] = same ? a : std::nanf("");
  }
}

void sara_hip_selfcheck_sincos(const float* theta, float* out_sin, float* out_cos,
                               size_t count)
{
  for (size_t i = 0; i < count; ++i)
  {
    double s, c;
    sara_hip::sincos_reduced_f64_host(double(theta[i]), s, c);
    out_sin[i] = float(s);
    out_cos[i] = float(c);
  }
}

sara_hip_status sara_hip_selfcheck_device_math(unsigned long long* mismatches,
                                               int device)
{
  if (!mismatches)
    return fail(SARA_HIP_INVALID_PARAMS, "null pointer");
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  DeviceScratch sc;
  unsigned long long* d = nullptr;
  HIP_TRY(sc.get(d, 2));
  HIP_TRY(hipMemset(d, 0, 2 * sizeof(unsigned long long)));
  launch_device_math_selfcheck(d, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(mismatches, d, 2 * sizeof(unsigned long long),
                    hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_selfcheck_orientation_bins(unsigned long long* mismatches,
                                                   int device)
{
  if (!mismatches)
    return fail(SARA_HIP_INVALID_PARAMS, "null pointer");
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  float thr[40];
  orientation_bin_thresholds(thr);
  DeviceScratch sc;
  float* d_thr = nullptr;
  unsigned long long* d_bad = nullptr;
  HIP_TRY(sc.get(d_thr, 40));
  HIP_TRY(sc.get(d_bad, 1));
  HIP_TRY(hipMemcpy(d_thr, thr, sizeof(thr), hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(d_bad, 0, sizeof(unsigned long long)));
  launch_orientation_bin_selfcheck(d_thr, d_bad, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(mismatches, d_bad, sizeof(unsigned long long),
                    hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_selfcheck_definiteness(const float* hessians,
                                                const int* types, size_t count,
                                                unsigned char* out, int device)
{
  if (!hessians || !types || !out)
    return fail(SARA_HIP_INVALID_PARAMS, "null pointer");
  if (count == 0)
    return SARA_HIP_OK;
  if (count > (size_t(1) << 28))
    return fail(SARA_HIP_CAPACITY_EXCEEDED, "too many matrices");
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  DeviceScratch sc;
  float* dH = nullptr;
  int* dT = nullptr;
  unsigned char* dO = nullptr;
  HIP_TRY(sc.get(dH, 9 * count));
  HIP_TRY(sc.get(dT, count));
  HIP_TRY(sc.get(dO, count));
  HIP_TRY(hipMemcpy(dH, hessians, 9 * count * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dT, types, count * sizeof(int), hipMemcpyHostToDevice));
  launch_definiteness_selfcheck(dH, dT, int(count), dO, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dO, count, hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

}  // extern "C"
