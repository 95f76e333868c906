// The HIP-graph rules of the ROCm 7 runtimes (sift_host.hpp has the map of the
// host side).
#include "sift_host.hpp"

using namespace sara_hip;
using namespace sara_hip::host;

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

namespace sara_hip { namespace host {

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
  namespace {
    // SARA_HIP_DEBUG_COUNTS=1: what the process asked of the runtime, at exit
    std::atomic<int> g_streams_created{0}, g_streams_reused{0};
    struct CountsAtExit
    {
      ~CountsAtExit()
      {
        if (getenv("SARA_HIP_DEBUG_COUNTS"))
        {
          int version = 0;
          (void) hipRuntimeGetVersion(&version);
          std::fprintf(stderr,
                       "[sara_hip] HIP runtime %d: graphs instantiated %d, streams created "
                       "%d, reused %d\n",
                       version, g_graph_instantiations.load(), g_streams_created.load(),
                       g_streams_reused.load());
        }
      }
    } g_counts_at_exit;
  }
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
  namespace {
    struct StreamPool
    {
      std::mutex m;
      std::vector<hipStream_t> idle[64][2];  // [device][high priority]
    };
    StreamPool& stream_pool()
    {
      static StreamPool* p = new StreamPool;  // never destroyed: no teardown order to get wrong
      return *p;
    }
    bool pool_streams()
    {
      static const bool on = [] {
        const char* e = getenv("SARA_HIP_STREAM_POOL");
        return !e || e[0] != '0';
      }();
      return on;
    }
  }
  hipError_t pooled_stream_acquire(int device, bool high_priority, hipStream_t* out)
  {
    if (pool_streams())
    {
      StreamPool& p = stream_pool();
      std::lock_guard<std::mutex> lock(p.m);
      auto& idle = p.idle[device & 63][high_priority ? 1 : 0];
      if (!idle.empty())
      {
        *out = idle.back();
        idle.pop_back();
        g_streams_reused.fetch_add(1, std::memory_order_relaxed);
        return hipSuccess;
      }
    }
    g_streams_created.fetch_add(1, std::memory_order_relaxed);
    if (!high_priority)
      return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
    int lo = 0, hi = 0;
    const hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (e != hipSuccess)
      return e;
    return hipStreamCreateWithPriority(out, hipStreamNonBlocking, hi);
  }
  void pooled_stream_release(int device, bool high_priority, hipStream_t stream)
  {
    if (!stream)
      return;
    (void) hipStreamSynchronize(stream);
    if (!pool_streams())
    {
      (void) hipStreamDestroy(stream);
      return;
    }
    StreamPool& p = stream_pool();
    std::lock_guard<std::mutex> lock(p.m);
    p.idle[device & 63][high_priority ? 1 : 0].push_back(stream);
  }
  bool first_graph_thread()
  {
    static std::atomic<std::thread::id> first{std::thread::id()};
    std::thread::id none, me = std::this_thread::get_id();
    if (first.compare_exchange_strong(none, me))
      return true;
    return first.load() == me;
  }

}}  // namespace sara_hip::host
