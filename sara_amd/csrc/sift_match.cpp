// Host side of descriptor matching (include/sara_hip_sift.h, "descriptor
// matching").
//
// Restates AnnMatcher (FeatureMatching/AnnMatcher.hpp:32-86, .cpp:59-268) with
// both constructors: two key sets, and one key set matched against itself with
// the KeyProximity filter (FeatureMatching/KeyProximity.cpp:17-30).  The
// neighbour queries - FLANN's knnSearch(3) and radiusSearch - run on the GPU
// (match_kernels.hip: exhaustive, or an MFMA prefilter with exact re-ranking);
// the decision logic of append_nearest_neighbors (.cpp:59-170) and the sort /
// unique / sort of compute_matches (.cpp:239-258) run here on the few thousand
// records that come back.
#include "sift_kernels.hpp"

#include <algorithm>
#include <climits>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace sara_hip;

namespace {

#define HIPM_TRY(expr)                                                         \
  do                                                                           \
  {                                                                            \
    const hipError_t e_ = (expr);                                              \
    if (e_ != hipSuccess)                                                      \
      return set_error(SARA_HIP_RUNTIME_ERROR,                                 \
                       (std::string(#expr) + ": " + hipGetErrorString(e_))     \
                           .c_str());                                          \
  } while (0)

  sara_hip_status use_device(int device)
  {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      return set_error(SARA_HIP_NO_DEVICE,
                       "no HIP device: the SIFT front-end has no CPU fallback");
    if (device < 0 || device >= ndev)
      return set_error(SARA_HIP_INVALID_PARAMS, "device ordinal out of range");
    HIPM_TRY(hipSetDevice(device));
    return SARA_HIP_OK;
  }

  // Grow-only scratch of the calling thread, per device: hipMalloc / hipFree
  // per call cost more than the whole search of two 4 k key sets.
  struct Workspace
  {
    enum Slot
    {
      kDescA,
      kDescB,
      kPartD,
      kPartI,
      kTopD,
      kTopI,
      kOut,
      kRadius,
      kAux0,
      kAux1,
      kAux2,
      kAux3,
      kFinish,
      kBatchZero,   // the arenas and the pair table of a batched search
      kBatchWork,
      kBatchOut,
      kBatchTable,
      kBatchDense,  // the batch's lists back to back + their offsets
      kSlots
    };
    int device = -1;
    void* dev[kSlots] = {};
    size_t dev_bytes[kSlots] = {};
    void* pinned = nullptr;
    size_t pinned_bytes = 0;

    //! Frees everything on the workspace's device and leaves the thread's
    //! current device as it found it.
    void release()
    {
      if (device < 0)
        return;
      int current = -1;
      (void) hipGetDevice(&current);
      (void) hipSetDevice(device);
      for (int k = 0; k < kSlots; ++k)
      {
        if (dev[k])
          (void) hipFree(dev[k]);
        dev[k] = nullptr;
        dev_bytes[k] = 0;
      }
      if (pinned)
        (void) hipHostFree(pinned);
      pinned = nullptr;
      pinned_bytes = 0;
      if (done)
        (void) hipEventDestroy(done);
      done = nullptr;
      if (stream)
      {
        std::lock_guard<std::recursive_mutex> lock(sara_hip::runtime_mutex());
        (void) hipStreamDestroy(stream);
      }
      stream = nullptr;
      device = -1;
      if (current >= 0)
        (void) hipSetDevice(current);
    }
    ~Workspace() { release(); }

    template <typename T>
    hipError_t get(Slot k, size_t count, T*& p)
    {
      const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
      if (bytes > dev_bytes[k])
      {
        if (dev[k])
          (void) hipFree(dev[k]);
        dev[k] = nullptr;
        dev_bytes[k] = 0;
        const size_t want = bytes + bytes / 4;
        const hipError_t e = hipMalloc(&dev[k], want);
        if (e != hipSuccess)
          return e;
        dev_bytes[k] = want;
      }
      p = static_cast<T*>(dev[k]);
      return hipSuccess;
    }
    //! Waits for everything enqueued on `stream` by polling an event:
    //! hipStreamSynchronize parks the thread and its wake-up alone cost
    //! about 0.3 ms per call - more than the whole search.
    //! Everything the matcher enqueues goes to this non-blocking stream: on
    //! the legacy NULL stream - which orders itself against every blocking
    //! stream of the process - the dozen launches of a search took 0.53 ms on
    //! the GPU although the kernels add up to 0.26 ms.
    hipStream_t stream = nullptr;
    hipError_t ensure_stream()
    {
      if (stream)
        return hipSuccess;
      std::lock_guard<std::recursive_mutex> lock(sara_hip::runtime_mutex());
      return hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
    }
    hipEvent_t done = nullptr;
    hipError_t wait(hipStream_t stream)
    {
      if (!done)
      {
        const hipError_t e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
        if (e != hipSuccess)
          return e;
      }
      hipError_t e = hipEventRecord(done, stream);
      if (e != hipSuccess)
        return e;
      while ((e = hipEventQuery(done)) == hipErrorNotReady)
        ;
      return e;
    }
    //! mark(): an event behind what `stream` holds now; wait_mark() polls it.
    hipError_t mark(hipStream_t stream)
    {
      if (!done)
      {
        const hipError_t e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
        if (e != hipSuccess)
          return e;
      }
      return hipEventRecord(done, stream);
    }
    hipError_t wait_mark()
    {
      hipError_t e;
      while ((e = hipEventQuery(done)) == hipErrorNotReady)
        ;
      return e;
    }
    hipError_t host(size_t bytes, void*& p)
    {
      if (bytes > pinned_bytes)
      {
        if (pinned)
          (void) hipHostFree(pinned);
        pinned = nullptr;
        pinned_bytes = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        const hipError_t e = hipHostMalloc(&pinned, want);
        if (e != hipSuccess)
          return e;
        pinned_bytes = want;
      }
      p = pinned;
      return hipSuccess;
    }
  };

  //! One workspace per (thread, device): a thread that walks over all GPUs of
  //! a node keeps one each, nothing is ever evicted behind the caller's back.
  std::map<int, Workspace>& workspaces()
  {
    thread_local std::map<int, Workspace> ws;
    return ws;
  }

  Workspace& workspace(int device)
  {
    Workspace& w = workspaces()[device];
    w.device = device;
    return w;
  }

  //! The batched entry point keeps kBatchLanes searches in flight: one more
  //! set of workspaces per (thread, device), each with its own stream, scratch
  //! and pinned read-back buffer.
  constexpr int kBatchLanes = 4;
  std::map<int, Workspace>& lane_workspaces()
  {
    thread_local std::map<int, Workspace> ws;  // key: device * kBatchLanes + lane
    return ws;
  }
  Workspace& lane_workspace(int device, int lane)
  {
    Workspace& w = lane_workspaces()[device * kBatchLanes + lane];
    w.device = device;
    return w;
  }

  //! OERegion fields KeyProximity and Match::operator== read.
  struct Feat
  {
    float x, y, m00, m10, m01, m11, orientation;
    int type;
    bool operator==(const Feat& o) const  // Feature.hpp:140-146
    {
      return x == o.x && y == o.y && m00 == o.m00 && m10 == o.m10 &&
             m01 == o.m01 && m11 == o.m11 && orientation == o.orientation &&
             type == o.type;
    }
  };

  Feat feat_of(const sara_oeregion& r)
  {
    return Feat{r.coords[0],       r.coords[1],       r.shape_matrix[0],
                r.shape_matrix[1], r.shape_matrix[2], r.shape_matrix[3],
                r.orientation,     int(r.type)};
  }

  //! KeyProximity::operator(), KeyProximity.cpp:17-30 (SquaredRefDistance,
  //! Geometry/Tools/Metric.hpp:47-50: (b - a).dot(M (b - a))).
  struct KeyProximity
  {
    float squared_metric_dist, squared_dist_thres;
    static float metric(const Feat& f, float vx, float vy)
    {
      const float r0 = f.m00 * vx + f.m01 * vy;
      const float r1 = f.m10 * vx + f.m11 * vy;
      return vx * r0 + vy * r1;
    }
    bool operator()(const Feat& f1, const Feat& f2) const
    {
      const float sd1 = metric(f1, f2.x - f1.x, f2.y - f1.y);
      const float sd2 = metric(f2, f2.x - f1.x, f2.y - f1.y);
      const float dx = f1.x - f2.x, dy = f1.y - f2.y;
      const float squared_pixel_dist = dx * dx + dy * dy;
      return squared_pixel_dist < squared_dist_thres || sd1 < squared_metric_dist ||
             sd2 < squared_metric_dist;
    }
  };

  //! What the searches of one direction returned, on the host.
  struct Neighbours
  {
    int nq = 0, nt = 0;
    std::vector<float> top_d;  // [3][nq]
    std::vector<int> top_i;    // [3][nq]
    std::vector<MatchNeighbour> radius;  // sorted by (query, distance, index)
    std::vector<int> begin;              // [nq + 1] into radius
  };

  //! append_nearest_neighbors (AnnMatcher.cpp:59-170) for every query of one
  //! direction.
  void append_matches(const Neighbours& nb, float thres2, int direction,
                      bool self_matching, const KeyProximity& is_redundant,
                      const Feat* fq, const Feat* ft, std::vector<sara_match>& out)
  {
    const int nq = nb.nq, nt = nb.nt;
    if (nt == 0)
      return;
    auto push = [&](int i1, int i2, float score, int rank) {
      sara_match m;
      m.x_index = direction == 0 ? i1 : i2;
      m.y_index = direction == 0 ? i2 : i1;
      m.score = score;
      m.rank = rank;
      m.direction = direction;
      out.push_back(m);
    };
    for (int i = 0; i < nq; ++i)
    {
      if (nt == 1 && !self_matching)
      {
        if (1.f < thres2)  // :87-101
          push(i, 0, 1.f, 1);
        continue;
      }
      if (nt == 2 && self_matching)
      {
        if (1.f < thres2)  // :103-120
          push(i, nb.top_i[size_t(nq) + i], 1.f, 1);
        continue;
      }
      const int top1 = self_matching ? 1 : 0;
      if (top1 + 1 >= nt)
        continue;
      const float d_top1 = nb.top_d[size_t(top1) * nq + i];
      const float d_next = nb.top_d[size_t(top1 + 1) * nq + i];
      const float top1_score = d_next > 0.f ? d_top1 / d_next : 0.f;
      if (!(thres2 > 1.f))
      {
        // K = 1: only rank top1 = 0 is visited (nothing when self-matching)
        if (top1 == 0 && !(top1_score > thres2))
          push(i, nb.top_i[i], top1_score, 1);
        continue;
      }
      const int lo = nb.begin[size_t(i)], K = nb.begin[size_t(i) + 1] - lo;
      for (int rank = top1; rank < K; ++rank)
      {
        const MatchNeighbour& n = nb.radius[size_t(lo + rank)];
        float score = 0.f;
        if (rank == top1)
          score = top1_score;
        else if (d_top1)
          score = n.distance / d_top1;
        if (score > thres2)
          break;
        if (self_matching && is_redundant(fq[i], ft[n.index]))
          continue;
        push(i, n.index, score, top1 == 0 ? rank + 1 : rank);
      }
    }
  }

  //! compute_matches' tail, AnnMatcher.cpp:239-258: sort by (x, y, score, ..),
  //! drop duplicates of (x, y), sort by (score, x, y).
  //! Round 3: two comparison sorts of the 17 k matches of a 4.3 k x 4.3 k pair at
  //! the default ratio took 0.8 ms.  The first order is a counting sort on x
  //! with the two to four entries of an x ordered by insertion; the second a
  //! stable radix sort on the score's bits (scores are >= 0, so their bit
  //! patterns order like the floats; -0.f does not occur: a score is 0.f or a
  //! quotient of non-negative distances) of a list that is in (x, y) order.
  void finish_matches(std::vector<sara_match>& m, const Feat* f1, const Feat* f2)
  {
    const auto before = [](const sara_match& a, const sara_match& b) {
      if (a.x_index != b.x_index)
        return a.x_index < b.x_index;
      if (a.y_index != b.y_index)
        return a.y_index < b.y_index;
      if (a.score != b.score)
        return a.score < b.score;
      if (a.direction != b.direction)
        return a.direction < b.direction;
      return a.rank < b.rank;
    };
    const size_t n = m.size();
    if (n == 0)
      return;
    int max_x = 0;
    bool plain = true;  // indices >= 0, scores >= +0.f and not NaN
    for (const sara_match& e : m)
    {
      max_x = std::max(max_x, e.x_index);
      plain = plain && e.x_index >= 0 && e.score >= 0.f && !std::signbit(e.score);
    }
    std::vector<sara_match> tmp(n);
    if (!plain)
      std::sort(m.begin(), m.end(), before);
    else
    {
      std::vector<int> begin(size_t(max_x) + 2, 0);
      for (const sara_match& e : m)
        ++begin[size_t(e.x_index) + 1];
      for (int x = 0; x <= max_x; ++x)
        begin[size_t(x) + 1] += begin[size_t(x)];
      std::vector<int> cursor(begin.begin(), begin.end() - 1);
      for (const sara_match& e : m)
      {
        const int lo = begin[size_t(e.x_index)];
        int at = cursor[size_t(e.x_index)]++;
        while (at > lo && before(e, tmp[size_t(at - 1)]))
        {
          tmp[size_t(at)] = tmp[size_t(at - 1)];
          --at;
        }
        tmp[size_t(at)] = e;
      }
      m.swap(tmp);
    }
    m.erase(std::unique(m.begin(), m.end(),
                        [&](const sara_match& a, const sara_match& b) {
                          if (f1 && f2)  // Match::operator== compares by value
                            return f1[a.x_index] == f1[b.x_index] &&
                                   f2[a.y_index] == f2[b.y_index];
                          return a.x_index == b.x_index && a.y_index == b.y_index;
                        }),
            m.end());
    if (!plain)
    {
      std::sort(m.begin(), m.end(), [](const sara_match& a, const sara_match& b) {
        if (a.score != b.score)
          return a.score < b.score;
        if (a.x_index != b.x_index)
          return a.x_index < b.x_index;
        return a.y_index < b.y_index;
      });
      return;
    }
    // m is in (x, y) order: a stable sort on the score alone gives (score, x, y)
    const size_t k = m.size();
    tmp.resize(k);
    sara_match* a = m.data();
    sara_match* b = tmp.data();
    for (int pass = 0; pass < 3; ++pass)
    {
      const int shift = 11 * pass;
      const unsigned mask = pass == 2 ? 0x3ffu : 0x7ffu;
      unsigned hist[2049] = {0};
      auto digit = [&](const sara_match& e) {
        uint32_t u;
        std::memcpy(&u, &e.score, 4);
        return (u >> shift) & mask;
      };
      for (size_t i = 0; i < k; ++i)
        ++hist[digit(a[i]) + 1];
      for (int d = 0; d < 2048; ++d)
        hist[d + 1] += hist[d];
      for (size_t i = 0; i < k; ++i)
        b[hist[digit(a[i])]++] = a[i];
      std::swap(a, b);
    }
    if (a != m.data())  // three passes: the result is in tmp
      m.swap(tmp);
  }

  //! Which producer answers the neighbour queries: "mfma" = MFMA prefilter +
  //! exact re-ranking (match_mfma.hip), "exhaustive" = every distance in
  //! FLANN's arithmetic (match_kernels.hip).  Same answers, entry by entry.
  //! Default: mfma from 256 keys per set on; SARA_HIP_MATCH forces one.
  bool use_mfma(int n1, int n2)
  {
    static const int forced = [] {
      const char* e = getenv("SARA_HIP_MATCH");
      if (!e)
        return 0;
      return std::string(e) == "mfma" ? 1 : std::string(e) == "exhaustive" ? 2 : 0;
    }();
    if (forced)
      return forced == 1;
    return std::min(n1, n2) >= 256;
  }

  //! knnSearch(3) - and radiusSearch when thres2 > 1 - of both directions,
  //! results in device memory of the workspace.
  struct DeviceSearch
  {
    float* top_d[2] = {nullptr, nullptr};  // [3][nq] of direction 0 / 1
    int* top_i[2] = {nullptr, nullptr};
    MatchNeighbour* radius[2] = {nullptr, nullptr};
    int radius_found[2] = {0, 0};
    int* count = nullptr;       // device: members found per direction
    size_t cap[2] = {0, 0};     // entries of radius[dir]
    int nq[2] = {0, 0}, nt[2] = {0, 0};
    bool have[2] = {false, false};
  };

  //! Entries of a direction's radius-member list.
  size_t radius_list_capacity(int nq, bool have, size_t min_cap)
  {
    return have ? std::max<size_t>(std::max<size_t>(8 * size_t(nq), 1 << 15), min_cap)
                : 1;
  }

  //! keep_on_device: the radius members stay on the device and their counts
  //! are not read back (the caller finishes there, see
  //! launch_finish_radius_matches); r.count points at the two device counters.
  sara_hip_status device_search(Workspace& ws, const float* d1, int n1,
                                const float* d2, int n2, int dim, float thres2,
                                int top1, bool self_matching, DeviceSearch* out,
                                bool keep_on_device = false, size_t min_cap = 0,
                                const ZeroRanges* also_clear = nullptr)
  {
    DeviceSearch& r = *out;
    r.nq[0] = n1;
    r.nt[0] = n2;
    r.nq[1] = n2;
    r.nt[1] = n1;
    // a direction has something to rank when its target set has >= 2 keys
    r.have[0] = n2 >= 2;
    r.have[1] = !self_matching && n1 >= 2;
    float* top_d = nullptr;
    int* top_i = nullptr;
    HIPM_TRY(ws.get(Workspace::kTopD, 3 * (size_t(n1) + n2), top_d));
    HIPM_TRY(ws.get(Workspace::kTopI, 3 * (size_t(n1) + n2), top_i));
    r.top_d[0] = top_d;
    r.top_d[1] = top_d + 3 * size_t(n1);
    r.top_i[0] = top_i;
    r.top_i[1] = top_i + 3 * size_t(n1);
    const bool radius_on = thres2 > 1.f;
    int* d_count = nullptr;
    HIPM_TRY(ws.get(Workspace::kAux0, 4, d_count));
    const bool mfma = use_mfma(n1, n2) && r.have[0] && (self_matching || r.have[1]);
    size_t cap[2] = {0, 0};
    if (radius_on)
      for (int dir = 0; dir < 2; ++dir)
        cap[dir] = radius_list_capacity(r.nq[dir], r.have[dir], min_cap);
    r.count = d_count;
    r.cap[0] = cap[0];
    r.cap[1] = cap[1];
    for (int attempt = 0; attempt < 2; ++attempt)
    {
      MatchNeighbour* list = nullptr;
      if (radius_on)
      {
        HIPM_TRY(ws.get(Workspace::kRadius, cap[0] + cap[1], list));
      }
      // everything the call expects to be zero, in one launch (the MFMA
      // producer adds its own arrays to the same launch)
      ZeroRanges zr;
      if (also_clear && attempt == 0)
        zr = *also_clear;
      if (radius_on)
        zr.add(d_count, 4);
      r.radius[0] = list;
      r.radius[1] = list ? list + cap[0] : nullptr;
      if (mfma)
      {
        const int slots = radius_on ? 32 : 8;  // 64 radius slots: fewer fall-backs, same time
        float* fs = nullptr;
        int* is = nullptr;
        HIPM_TRY(ws.get(Workspace::kAux1, match_mfma_scratch_floats(n1, n2), fs));
        HIPM_TRY(ws.get(Workspace::kAux2, match_mfma_scratch_ints(n1, n2, slots), is));
        launch_match_mfma(d1, n1, d2, n2, dim, thres2, top1, r.have[1] ? 1 : 0, fs,
                          is, slots, r.top_d[0], r.top_i[0], r.top_d[1], r.top_i[1],
                          r.radius[0], int(cap[0]), d_count, r.radius[1],
                          int(cap[1]), d_count + 1, ws.stream, &zr);
      }
      else
      {
        launch_zero_ranges(zr, ws.stream);
        for (int dir = 0; dir < 2; ++dir)
        {
          if (!r.have[dir])
            continue;
          const float* q = dir == 0 ? d1 : d2;
          const float* t = dir == 0 ? d2 : d1;
          const int nq = r.nq[dir], nt = r.nt[dir];
          int chunk = 0, nchunks = 0;
          match_chunking(nq, nt, &chunk, &nchunks);
          float* part_d = nullptr;
          int* part_i = nullptr;
          HIPM_TRY(ws.get(Workspace::kPartD, 3 * size_t(nchunks) * nq, part_d));
          HIPM_TRY(ws.get(Workspace::kPartI, 3 * size_t(nchunks) * nq, part_i));
          launch_nn3_exhaustive(q, nq, t, nt, dim, part_d, part_i, r.top_d[dir],
                                r.top_i[dir], ws.stream);
          if (radius_on && nt > top1 + 1)
            launch_radius_exhaustive(q, nq, t, nt, dim, r.top_d[dir], top1, thres2,
                                     r.radius[dir], int(cap[dir]), d_count + dir,
                                     ws.stream);
        }
      }
      HIPM_TRY(hipGetLastError());
      if (!radius_on || keep_on_device)
        break;
      int found[2] = {0, 0};
      HIPM_TRY(ws.wait(ws.stream));  // the copies below use the NULL stream
      HIPM_TRY(hipMemcpy(found, d_count, 2 * sizeof(int), hipMemcpyDeviceToHost));
      r.radius_found[0] = found[0];
      r.radius_found[1] = found[1];
      if (size_t(found[0]) <= cap[0] && size_t(found[1]) <= cap[1])
        break;
      // the counts are exact: the second pass fits
      cap[0] = std::max(cap[0], size_t(found[0]));
      cap[1] = std::max(cap[1], size_t(found[1]));
    }
    return SARA_HIP_OK;
  }

  //! Both directions of a DeviceSearch on the host, radius members sorted the
  //! way FLANN's RadiusResultSet hands them out: by (distance, index).
  //! Round 3: six blocking copies into pageable vectors and a comparison sort
  //! of all radius members took 0.3 ms per direction; now every array goes
  //! through the workspace's pinned buffer behind ONE wait, and the members
  //! are grouped by query with a counting sort (a query has a handful of
  //! them, ordered by insertion).
  sara_hip_status to_host(Workspace& ws, const DeviceSearch& r, Neighbours nb[2])
  {
    size_t off[2][3], total = 0;
    for (int dir = 0; dir < 2; ++dir)
    {
      const size_t nq = size_t(r.nq[dir]);
      const size_t found = r.have[dir] ? size_t(std::max(r.radius_found[dir], 0)) : 0;
      off[dir][0] = total;
      total += sizeof(float) * 3 * nq;
      off[dir][1] = total;
      total += sizeof(int) * 3 * nq;
      off[dir][2] = total;
      total += sizeof(MatchNeighbour) * found;
      total = (total + 15) & ~size_t(15);
    }
    void* hv = nullptr;
    HIPM_TRY(ws.host(total + 16, hv));
    unsigned char* h = static_cast<unsigned char*>(hv);
    for (int dir = 0; dir < 2; ++dir)
    {
      if (!r.have[dir])
        continue;
      const size_t nq = size_t(r.nq[dir]);
      const size_t found = size_t(std::max(r.radius_found[dir], 0));
      if (nq)
      {
        HIPM_TRY(hipMemcpyAsync(h + off[dir][0], r.top_d[dir], sizeof(float) * 3 * nq,
                                hipMemcpyDeviceToHost, ws.stream));
        HIPM_TRY(hipMemcpyAsync(h + off[dir][1], r.top_i[dir], sizeof(int) * 3 * nq,
                                hipMemcpyDeviceToHost, ws.stream));
      }
      if (found)
        HIPM_TRY(hipMemcpyAsync(h + off[dir][2], r.radius[dir],
                                sizeof(MatchNeighbour) * found, hipMemcpyDeviceToHost,
                                ws.stream));
    }
    HIPM_TRY(ws.wait(ws.stream));
    for (int dir = 0; dir < 2; ++dir)
    {
      Neighbours& n = nb[dir];
      const int nq = r.nq[dir];
      n.nq = nq;
      n.nt = r.nt[dir];
      n.top_d.assign(3 * size_t(nq), 0.f);
      n.top_i.assign(3 * size_t(nq), -1);
      n.radius.clear();
      n.begin.assign(size_t(nq) + 1, 0);
      if (!r.have[dir])
        continue;
      std::memcpy(n.top_d.data(), h + off[dir][0], sizeof(float) * 3 * size_t(nq));
      std::memcpy(n.top_i.data(), h + off[dir][1], sizeof(int) * 3 * size_t(nq));
      const int found = std::max(r.radius_found[dir], 0);
      if (found == 0)
        continue;
      const MatchNeighbour* src =
          reinterpret_cast<const MatchNeighbour*>(h + off[dir][2]);
      for (int i = 0; i < found; ++i)
        ++n.begin[size_t(src[i].query) + 1];
      for (int i = 0; i < nq; ++i)
        n.begin[size_t(i) + 1] += n.begin[size_t(i)];
      n.radius.resize(size_t(found));
      std::vector<int> cursor(n.begin.begin(), n.begin.end() - 1);
      for (int i = 0; i < found; ++i)
      {
        // insertion into the query's (distance, index)-ordered run
        const MatchNeighbour& e = src[i];
        const int lo = n.begin[size_t(e.query)];
        int at = cursor[size_t(e.query)]++;
        while (at > lo && (n.radius[size_t(at - 1)].distance > e.distance ||
                           (n.radius[size_t(at - 1)].distance == e.distance &&
                            n.radius[size_t(at - 1)].index > e.index)))
        {
          n.radius[size_t(at)] = n.radius[size_t(at - 1)];
          --at;
        }
        n.radius[size_t(at)] = e;
      }
    }
    return SARA_HIP_OK;
  }

  sara_hip_status deliver(std::vector<sara_match>& m, sara_match* matches,
                          int capacity, int* count)
  {
    *count = int(m.size());
    if (int(m.size()) > capacity)
      return set_error(SARA_HIP_CAPACITY_EXCEEDED,
                       "more matches than `capacity` (*count holds the number "
                       "needed)");
    std::copy(m.begin(), m.end(), matches);
    return SARA_HIP_OK;
  }

  // ---- ratios <= 1, in two halves ----------------------------------------
  // Only the best neighbour can pass (K = 1, AnnMatcher.cpp:131): the ratio
  // test of both directions, the removal of (x, y) duplicates and the sort by
  // score all run on the device; one copy brings the finished list back.  The
  // tail's buffers exist before the search, so that ONE launch clears the
  // counters of both.  Nothing in this path waits for the device before the
  // final copy, so it splits into an asynchronous half - everything enqueued on
  // the workspace's stream, the read-back into its pinned buffer included - and
  // a half that waits and hands the list over: the batched entry point keeps
  // several workspaces between the two.
  struct BestMatchHeader
  {
    int count, pad[3];
  };

  sara_hip_status enqueue_best_match_search(Workspace& ws, const float* d1, int n1,
                                            const float* d2, int n2, int dim,
                                            float thres2)
  {
    using Header = BestMatchHeader;
    const int cap_dev = n1 + n2;
    unsigned char *d_out = nullptr, *d_tmp = nullptr;
    const size_t list_bytes = sizeof(sara_match) * size_t(cap_dev);
    const size_t out_bytes = sizeof(Header) + list_bytes;
    ZeroRanges tail;
    HIPM_TRY(ws.get(Workspace::kOut, out_bytes, d_out));
    HIPM_TRY(ws.get(Workspace::kAux3, list_bytes + sizeof(int) * size_t(cap_dev), d_tmp));
    tail.add(d_out, sizeof(Header) / sizeof(int));
    tail.add(d_tmp + list_bytes, size_t(cap_dev));
    DeviceSearch ds;
    const sara_hip_status ss = device_search(ws, d1, n1, d2, n2, dim, thres2, 0, false,
                                             &ds, false, 0, &tail);
    if (ss != SARA_HIP_OK)
      return ss;
    launch_finish_matches(ds.top_d[0], ds.top_i[0], n1, ds.top_d[1], ds.top_i[1], n2,
                          ds.have[0] ? 1 : 0, ds.have[1] ? 1 : 0, thres2,
                          reinterpret_cast<sara_match*>(d_tmp),
                          reinterpret_cast<int*>(d_tmp + list_bytes),
                          reinterpret_cast<int*>(d_out),
                          reinterpret_cast<sara_match*>(d_out + sizeof(Header)),
                          ws.stream, true);
    HIPM_TRY(hipGetLastError());
    void* h = nullptr;
    HIPM_TRY(ws.host(out_bytes, h));
    HIPM_TRY(hipMemcpyAsync(h, d_out, out_bytes, hipMemcpyDeviceToHost, ws.stream));
    HIPM_TRY(ws.mark(ws.stream));
    return SARA_HIP_OK;
  }

  //! Waits for the search enqueue_best_match_search() left in `ws`;
  //! *count = matches found, copied to `matches` when they fit.
  sara_hip_status collect_best_match_search(Workspace& ws, int n1, int n2,
                                            sara_match* matches, int capacity,
                                            int* count)
  {
    using Header = BestMatchHeader;
    HIPM_TRY(ws.wait_mark());
    const unsigned char* h = static_cast<const unsigned char*>(ws.pinned);
    const int found = std::min(reinterpret_cast<const Header*>(h)->count, n1 + n2);
    *count = found;
    if (found > capacity)
      return set_error(SARA_HIP_CAPACITY_EXCEEDED,
                       "more matches than `capacity` (*count holds the number "
                       "needed)");
    std::memcpy(matches, h + sizeof(Header), sizeof(sara_match) * size_t(found));
    return SARA_HIP_OK;
  }

  //! Host descriptors -> the workspace's device buffers (asynchronous).
  sara_hip_status upload_pair(Workspace& ws, const float*& d1, int n1,
                              const float*& d2, int n2, int dim)
  {
    float *a = nullptr, *b = nullptr;
    HIPM_TRY(ws.get(Workspace::kDescA, size_t(n1) * dim, a));
    HIPM_TRY(ws.get(Workspace::kDescB, size_t(n2) * dim, b));
    HIPM_TRY(hipMemcpyAsync(a, d1, size_t(n1) * dim * sizeof(float),
                            hipMemcpyHostToDevice, ws.stream));
    HIPM_TRY(hipMemcpyAsync(b, d2, size_t(n2) * dim * sizeof(float),
                            hipMemcpyHostToDevice, ws.stream));
    d1 = a;
    d2 = b;
    return SARA_HIP_OK;
  }

  // ---- a batch in ONE set of launches (squared ratio <= 1) -------------------
  // Every kernel of the search takes the pair as a grid dimension and reads the
  // pair's pointers from a table in HBM (MatchBatchPair, match_mfma.hip): the
  // tile grid of a batch of 16 pairs of 4.3 k keys is 18 496 workgroups instead
  // of 16 grids of 1 156 that fill the 512 workgroup slots 2.26 times each, the
  // small kernels (norms, selection, re-ranking, tail) are launched once per
  // batch instead of once per pair, and ONE copy brings every list home.
  constexpr int kTrueBatchMax = 32;                   // pairs per set of launches
  constexpr size_t kTrueBatchBytes = size_t(3) << 30;  // scratch of one set

  //! Pairs [first, last) (all eligible: both sets >= 256 keys) -> their lists
  //! appended at `at`; counts[] per pair.
  sara_hip_status true_batch(Workspace& ws, const sara_match_pair* pairs, int first,
                             int last, int dim, float thres2, bool on_device,
                             sara_match* matches, int capacity, int* at,
                             long long* need, bool* overflow, int* offsets)
  {
    const int P = last - first;
    MatchBatchLayout lay;
    size_t desc_floats = 0;
    int n1_max = 0, n2_max = 0;
    for (int p = first; p < last; ++p)
    {
      match_batch_carve(pairs[p].n1, pairs[p].n2, dim, nullptr, nullptr, nullptr, &lay,
                        nullptr);
      desc_floats += (size_t(pairs[p].n1) + pairs[p].n2) * dim;
      n1_max = std::max(n1_max, pairs[p].n1);
      n2_max = std::max(n2_max, pairs[p].n2);
    }
    unsigned char *d_zero = nullptr, *d_work = nullptr, *d_out = nullptr;
    MatchBatchPair* d_table = nullptr;
    float* d_desc = nullptr;
    HIPM_TRY(ws.get(Workspace::kBatchZero, lay.zero_bytes + 64, d_zero));
    HIPM_TRY(ws.get(Workspace::kBatchWork, lay.work_bytes + 64, d_work));
    HIPM_TRY(ws.get(Workspace::kBatchOut, lay.out_bytes + 64, d_out));
    HIPM_TRY(ws.get(Workspace::kBatchTable, size_t(P), d_table));
    if (!on_device)
      HIPM_TRY(ws.get(Workspace::kDescA, desc_floats, d_desc));
    // dense copy of the lists (what travels home) behind their P + 1 offsets
    size_t records = 0;
    for (int p = first; p < last; ++p)
      records += size_t(pairs[p].n1) + pairs[p].n2;
    const size_t heads_bytes = (sizeof(int) * size_t(P + 1) + 63) & ~size_t(63);
    unsigned char* d_dense = nullptr;
    HIPM_TRY(ws.get(Workspace::kBatchDense, heads_bytes + sizeof(sara_match) * records,
                    d_dense));
    int* d_heads = reinterpret_cast<int*>(d_dense);
    sara_match* d_lists = reinterpret_cast<sara_match*>(d_dense + heads_bytes);
    // pinned: the table on its way up, offsets and lists on their way down
    const size_t table_bytes = (sizeof(MatchBatchPair) * size_t(P) + 63) & ~size_t(63);
    void* h = nullptr;
    HIPM_TRY(ws.host(table_bytes + heads_bytes + sizeof(sara_match) * records + 64, h));
    MatchBatchPair* h_table = static_cast<MatchBatchPair*>(h);
    int* h_heads = reinterpret_cast<int*>(static_cast<unsigned char*>(h) + table_bytes);
    sara_match* h_lists = reinterpret_cast<sara_match*>(
        static_cast<unsigned char*>(h) + table_bytes + heads_bytes);
    MatchBatchLayout run;
    size_t desc_at = 0;
    for (int p = first; p < last; ++p)
    {
      MatchBatchPair& e = h_table[p - first];
      e = MatchBatchPair{};
      e.d1 = pairs[p].desc1;
      e.d2 = pairs[p].desc2;
      if (!on_device)
      {
        float* a = d_desc + desc_at;
        float* b = a + size_t(pairs[p].n1) * dim;
        desc_at += (size_t(pairs[p].n1) + pairs[p].n2) * dim;
        HIPM_TRY(hipMemcpyAsync(a, pairs[p].desc1, size_t(pairs[p].n1) * dim * sizeof(float),
                                hipMemcpyHostToDevice, ws.stream));
        HIPM_TRY(hipMemcpyAsync(b, pairs[p].desc2, size_t(pairs[p].n2) * dim * sizeof(float),
                                hipMemcpyHostToDevice, ws.stream));
        e.d1 = a;
        e.d2 = b;
      }
      match_batch_carve(pairs[p].n1, pairs[p].n2, dim, d_zero, d_work, d_out, &run, &e);
    }
    HIPM_TRY(hipMemcpyAsync(d_table, h_table, sizeof(MatchBatchPair) * size_t(P),
                            hipMemcpyHostToDevice, ws.stream));
    launch_match_batch(d_table, h_table, P, n1_max, n2_max, dim, thres2, d_zero,
                       run.zero_bytes, ws.stream);
    launch_compact_lists(d_table, P, n1_max + n2_max, d_lists, d_heads, ws.stream);
    HIPM_TRY(hipGetLastError());
    // the offsets first (their last entry is the number of records), then
    // exactly the records there are
    HIPM_TRY(hipMemcpyAsync(h_heads, d_heads, sizeof(int) * size_t(P + 1),
                            hipMemcpyDeviceToHost, ws.stream));
    HIPM_TRY(ws.wait(ws.stream));
    const size_t total = size_t(std::max(h_heads[P], 0));
    if (total > records)
      return set_error(SARA_HIP_RUNTIME_ERROR, "batched matcher: list offsets out of range");
    if (total)
    {
      HIPM_TRY(hipMemcpyAsync(h_lists, d_lists, sizeof(sara_match) * total,
                              hipMemcpyDeviceToHost, ws.stream));
      HIPM_TRY(ws.wait(ws.stream));
    }
    for (int p = first; p < last; ++p)
    {
      const int begin = h_heads[p - first], found = h_heads[p - first + 1] - begin;
      *need += found;
      if (!*overflow && found > capacity - *at)
        *overflow = true;
      if (!*overflow)
      {
        std::memcpy(matches + *at, h_lists + begin, sizeof(sara_match) * size_t(found));
        *at += found;
      }
      offsets[p + 1] = int(std::min<long long>(*need, INT_MAX));
    }
    return SARA_HIP_OK;
  }

}  // namespace

extern "C" {

sara_hip_status sara_hip_match_descriptors(const float* desc1, int n1,
                                           const float* desc2, int n2, int dim,
                                           float sift_ratio_thres, int on_device,
                                           sara_match* matches, int capacity,
                                           int* count, int device)
{
  if (count)
    *count = 0;
  if (!desc1 || !desc2 || !matches || !count || n1 < 0 || n2 < 0 || capacity < 0)
    return set_error(SARA_HIP_INVALID_PARAMS, "null pointer or negative size");
  if (n1 == 0 || n2 == 0)
    return set_error(SARA_HIP_RUNTIME_ERROR,
                     "Error: the list of key-points is empty!");
  if (dim < 1 || dim > 128)
    return set_error(SARA_HIP_INVALID_PARAMS,
                     "descriptor dimension must be in 1..128");
  if (sift_ratio_thres != sift_ratio_thres)
    return set_error(SARA_HIP_INVALID_PARAMS, "the ratio threshold is NaN");
  static const bool prof = getenv("SARA_HIP_MATCH_PROF") != nullptr;
  const auto tp0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (prof)
      std::fprintf(stderr, "[match prof] %-10s %8.1f us\n", what,
                   std::chrono::duration<double, std::micro>(
                       std::chrono::steady_clock::now() - tp0).count());
  };
  const sara_hip_status st = use_device(device);
  if (st != SARA_HIP_OK)
    return st;
  Workspace& ws = workspace(device);
  HIPM_TRY(ws.ensure_stream());
  lap("device");
  static hipEvent_t pe0 = nullptr, pe1 = nullptr;
  if (prof)
  {
    if (!pe0)
    {
      (void) hipEventCreate(&pe0);
      (void) hipEventCreate(&pe1);
    }
    (void) hipEventRecord(pe0, ws.stream);
  }
  const float thres2 = sift_ratio_thres * sift_ratio_thres;
  const float *d1 = desc1, *d2 = desc2;
  if (!on_device)
  {
    float *a = nullptr, *b = nullptr;
    HIPM_TRY(ws.get(Workspace::kDescA, size_t(n1) * dim, a));
    HIPM_TRY(ws.get(Workspace::kDescB, size_t(n2) * dim, b));
    HIPM_TRY(hipMemcpyAsync(a, desc1, size_t(n1) * dim * sizeof(float),
                            hipMemcpyHostToDevice, ws.stream));
    HIPM_TRY(hipMemcpyAsync(b, desc2, size_t(n2) * dim * sizeof(float),
                            hipMemcpyHostToDevice, ws.stream));
    d1 = a;
    d2 = b;
  }
  std::vector<sara_match> m;
  DeviceSearch ds;
  // Default ratio (squared threshold > 1), both sets with something to rank:
  // the members of the radius searches never leave the device - ranks, scores,
  // the (x, y) duplicates of the two directions and the final order are worked
  // out there (launch_finish_radius_matches) and ONE copy brings the finished
  // list back.  (Round 3 read the member lists back and finished on the host:
  // 0.37 of the call's 0.79 ms for a 4.3 k x 4.3 k pair.)  A member list that
  // overflows its capacity is reported by the device; the call then falls
  // through to the host path below, which retries with exact capacities.
  if (thres2 > 1.f && n1 >= 3 && n2 >= 3)
  {
    struct Header
    {
      int count, overflow, pad[2];
    };
    for (int attempt = 0; attempt < 2; ++attempt)
    {
      // the tail's buffers first: their zero-initialised parts are cleared by
      // the same launch as the search's counters
      const size_t min_cap = attempt ? size_t(1) << 19 : 0;
      const int cap0 = int(radius_list_capacity(n1, n2 >= 2, min_cap));
      const int cap1 = int(radius_list_capacity(n2, n1 >= 2, min_cap));
      const size_t total = size_t(cap0) + cap1;
      unsigned char *d_out = nullptr, *d_tmp = nullptr;
      int* d_int = nullptr;
      const size_t list_bytes = sizeof(sara_match) * total;
      const size_t out_bytes = sizeof(Header) + list_bytes;
      HIPM_TRY(ws.get(Workspace::kOut, out_bytes, d_out));
      HIPM_TRY(ws.get(Workspace::kAux3, list_bytes, d_tmp));
      HIPM_TRY(ws.get(Workspace::kFinish, finish_radius_scratch_ints(cap0, cap1), d_int));
      ZeroRanges tail;
      tail.add(d_out, sizeof(Header) / sizeof(int));
      tail.add(d_int, finish_radius_cleared_ints(cap0, cap1));
      const sara_hip_status ss = device_search(ws, d1, n1, d2, n2, dim, thres2, 0, false,
                                               &ds, true, min_cap, &tail);
      if (ss != SARA_HIP_OK)
        return ss;
      if (int(ds.cap[0]) != cap0 || int(ds.cap[1]) != cap1)
        return set_error(SARA_HIP_RUNTIME_ERROR, "radius list capacities disagree");
      launch_finish_radius_matches(ds.radius[0], ds.count, cap0, ds.radius[1],
                                   ds.count + 1, cap1, ds.top_d[0], n1, ds.top_d[1], n2,
                                   thres2, d_int, reinterpret_cast<sara_match*>(d_tmp),
                                   reinterpret_cast<int*>(d_out),
                                   reinterpret_cast<sara_match*>(d_out + sizeof(Header)),
                                   ws.stream, true);
      HIPM_TRY(hipGetLastError());
      // the header first (a few hundred bytes would do, but the list's length
      // is only known from it): header + as many records as a pair of sets
      // with a few members per key has, the rest in a second copy if needed
      void* h = nullptr;
      HIPM_TRY(ws.host(out_bytes, h));
      const size_t first = std::min(total, 3 * (size_t(n1) + n2));
      HIPM_TRY(hipMemcpyAsync(h, d_out, sizeof(Header) + sizeof(sara_match) * first,
                              hipMemcpyDeviceToHost, ws.stream));
      HIPM_TRY(ws.wait(ws.stream));
      const Header hd = *static_cast<Header*>(h);
      if (hd.overflow)
        continue;  // once more with room for 2^19 members per direction
      const size_t found = size_t(std::max(hd.count, 0));
      if (found > first)
      {
        HIPM_TRY(hipMemcpyAsync(static_cast<unsigned char*>(h) + sizeof(Header) +
                                    sizeof(sara_match) * first,
                                d_out + sizeof(Header) + sizeof(sara_match) * first,
                                sizeof(sara_match) * (found - first),
                                hipMemcpyDeviceToHost, ws.stream));
        HIPM_TRY(ws.wait(ws.stream));
      }
      lap("device finish");
      *count = int(found);
      if (found > size_t(capacity))
        return set_error(SARA_HIP_CAPACITY_EXCEEDED,
                         "more matches than `capacity` (*count holds the number "
                         "needed)");
      std::memcpy(matches, static_cast<unsigned char*>(h) + sizeof(Header),
                  sizeof(sara_match) * found);
      return SARA_HIP_OK;
    }
    // both attempts overflowed (more than half a million radius members per
    // direction): the host path below sizes the lists exactly
  }
  // Ratios <= 1: enqueue_best_match_search / collect_best_match_search above.
  if (!(thres2 > 1.f))
  {
    const sara_hip_status es =
        enqueue_best_match_search(ws, d1, n1, d2, n2, dim, thres2);
    if (es != SARA_HIP_OK)
      return es;
    lap("enqueue");
    const sara_hip_status cs = collect_best_match_search(ws, n1, n2, matches, capacity,
                                                         count);
    lap("collect");
    return cs;
  }
  const sara_hip_status ss = device_search(ws, d1, n1, d2, n2, dim, thres2, 0, false,
                                           &ds, false, 0, nullptr);
  if (ss != SARA_HIP_OK)
    return ss;
  lap("search");
  {
    const KeyProximity unused{0.f, 0.f};
    Neighbours nb[2];
    const sara_hip_status hs = to_host(ws, ds, nb);
    if (hs != SARA_HIP_OK)
      return hs;
    lap("to_host");
    for (int dir = 0; dir < 2; ++dir)
      append_matches(nb[dir], thres2, dir, false, unused, nullptr, nullptr, m);
    lap("append");
  }
  finish_matches(m, nullptr, nullptr);
  lap("finish");
  return deliver(m, matches, capacity, count);
}

sara_hip_status sara_hip_match_descriptors_batch(
    const sara_match_pair* pairs, int n_pairs, int dim, float sift_ratio_thres,
    int on_device, sara_match* matches, int capacity, int* offsets, int device)
{
  if (!pairs || n_pairs < 0 || !matches || !offsets || capacity < 0)
    return set_error(SARA_HIP_INVALID_PARAMS, "null pointer or negative size");
  for (int p = 0; p <= n_pairs; ++p)
    offsets[p] = 0;
  if (dim < 1 || dim > 128)
    return set_error(SARA_HIP_INVALID_PARAMS,
                     "descriptor dimension must be in 1..128");
  if (sift_ratio_thres != sift_ratio_thres)
    return set_error(SARA_HIP_INVALID_PARAMS, "the ratio threshold is NaN");
  for (int p = 0; p < n_pairs; ++p)
  {
    if (!pairs[p].desc1 || !pairs[p].desc2 || pairs[p].n1 < 0 || pairs[p].n2 < 0)
      return set_error(SARA_HIP_INVALID_PARAMS, "null pointer or negative size");
    if (pairs[p].n1 == 0 || pairs[p].n2 == 0)
      return set_error(SARA_HIP_RUNTIME_ERROR,
                       "Error: the list of key-points is empty!");
  }
  const float thres2 = sift_ratio_thres * sift_ratio_thres;
  int at = 0;         // records written so far
  long long need = 0; // records all pairs hold (reported on overflow)
  bool overflow = false;
  if (thres2 > 1.f)
  {
    // The adaptive radius search sizes its lists from what it finds (and
    // repeats itself when they overflow): pair by pair.
    for (int p = 0; p < n_pairs; ++p)
    {
      int found = 0;
      const sara_hip_status st = sara_hip_match_descriptors(
          pairs[p].desc1, pairs[p].n1, pairs[p].desc2, pairs[p].n2, dim,
          sift_ratio_thres, on_device, matches + at, overflow ? 0 : capacity - at,
          &found, device);
      if (st == SARA_HIP_CAPACITY_EXCEEDED)
        overflow = true;
      else if (st != SARA_HIP_OK)
        return st;
      need += found;
      if (!overflow)
        at += found;
      offsets[p + 1] = int(std::min<long long>(need, INT_MAX));
    }
    if (overflow)
      return set_error(SARA_HIP_CAPACITY_EXCEEDED,
                       "more matches than `capacity` (offsets[n_pairs] holds the "
                       "number needed)");
    return SARA_HIP_OK;
  }
  const sara_hip_status st = use_device(device);
  if (st != SARA_HIP_OK)
    return st;
  // Eligible for ONE set of launches: every pair large enough for the MFMA
  // prefilter (the producer a single call would pick).  Groups of at most
  // kTrueBatchMax pairs / kTrueBatchBytes of scratch.
  {
    bool eligible = n_pairs >= 2;
    for (int p = 0; p < n_pairs && eligible; ++p)
      eligible = pairs[p].n1 >= 2 && pairs[p].n2 >= 2 &&
                 use_mfma(pairs[p].n1, pairs[p].n2);
    static const bool lanes_only = getenv("SARA_HIP_MATCH_BATCH") &&
                                   std::string(getenv("SARA_HIP_MATCH_BATCH")) == "lanes";
    if (eligible && !lanes_only)
    {
      Workspace& ws = workspace(device);
      HIPM_TRY(ws.ensure_stream());
      int first = 0;
      while (first < n_pairs)
      {
        MatchBatchLayout lay;
        int last = first;
        while (last < n_pairs && last - first < kTrueBatchMax)
        {
          MatchBatchLayout next = lay;
          match_batch_carve(pairs[last].n1, pairs[last].n2, dim, nullptr, nullptr,
                            nullptr, &next, nullptr);
          if (last > first &&
              next.zero_bytes + next.work_bytes + next.out_bytes > kTrueBatchBytes)
            break;
          lay = next;
          ++last;
        }
        const sara_hip_status bs = true_batch(ws, pairs, first, last, dim, thres2,
                                              on_device != 0, matches, capacity, &at,
                                              &need, &overflow, offsets);
        if (bs != SARA_HIP_OK)
          return bs;
        first = last;
      }
      if (overflow)
        return set_error(SARA_HIP_CAPACITY_EXCEEDED,
                         "more matches than `capacity` (offsets[n_pairs] holds the "
                         "number needed)");
      return SARA_HIP_OK;
    }
  }
  // Otherwise kBatchLanes searches in flight: while the device works on pair p the host
  // enqueues pairs p + 1 .. p + 3 on the other lanes' streams - the small
  // kernels of one search (norms, selection, re-ranking, the tail) run beside
  // the tile pass of another, and no pair waits for a read-back but its own.
  // Results are collected in pair order, so the concatenated list is the one
  // n_pairs single calls would have produced.
  auto collect = [&](int p) -> sara_hip_status {
    Workspace& ws = lane_workspace(device, p % kBatchLanes);
    int found = 0;
    const sara_hip_status cs = collect_best_match_search(
        ws, pairs[p].n1, pairs[p].n2, matches + at, overflow ? 0 : capacity - at,
        &found);
    if (cs == SARA_HIP_CAPACITY_EXCEEDED)
      overflow = true;
    else if (cs != SARA_HIP_OK)
      return cs;
    need += found;
    if (!overflow)
      at += found;
    offsets[p + 1] = int(std::min<long long>(need, INT_MAX));
    return SARA_HIP_OK;
  };
  sara_hip_status result = SARA_HIP_OK;
  int enqueued = 0, collected = 0;
  for (; enqueued < n_pairs && result == SARA_HIP_OK; ++enqueued)
  {
    if (enqueued - collected == kBatchLanes)  // the lane is still busy
      result = collect(collected++);
    if (result != SARA_HIP_OK)
      break;
    const int p = enqueued;
    Workspace& ws = lane_workspace(device, p % kBatchLanes);
    HIPM_TRY(ws.ensure_stream());
    const float *d1 = pairs[p].desc1, *d2 = pairs[p].desc2;
    if (!on_device)
      result = upload_pair(ws, d1, pairs[p].n1, d2, pairs[p].n2, dim);
    if (result == SARA_HIP_OK)
      result = enqueue_best_match_search(ws, d1, pairs[p].n1, d2, pairs[p].n2, dim,
                                         thres2);
    if (result != SARA_HIP_OK)
      break;
  }
  // drain (also after an error: nothing of this call stays in flight)
  for (; collected < enqueued; ++collected)
  {
    if (result == SARA_HIP_OK)
      result = collect(collected);
    else
      (void) lane_workspace(device, collected % kBatchLanes).wait_mark();
  }
  if (result != SARA_HIP_OK)
    return result;
  if (overflow)
    return set_error(SARA_HIP_CAPACITY_EXCEEDED,
                     "more matches than `capacity` (offsets[n_pairs] holds the "
                     "number needed)");
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_self_match_descriptors(
    const float* desc, const sara_oeregion* features, int n, int dim,
    float sift_ratio_thres, float min_max_metric_dist_thres,
    float pixel_dist_thres, int desc_on_device, sara_match* matches, int capacity,
    int* count, int device)
{
  if (count)
    *count = 0;
  if (!desc || !features || !matches || !count || n < 0 || capacity < 0)
    return set_error(SARA_HIP_INVALID_PARAMS, "null pointer or negative size");
  if (n == 0)
    return set_error(SARA_HIP_RUNTIME_ERROR,
                     "Error: the list of key-points is empty!");
  if (dim < 1 || dim > 128)
    return set_error(SARA_HIP_INVALID_PARAMS,
                     "descriptor dimension must be in 1..128");
  if (sift_ratio_thres != sift_ratio_thres)
    return set_error(SARA_HIP_INVALID_PARAMS, "the ratio threshold is NaN");
  const sara_hip_status st = use_device(device);
  if (st != SARA_HIP_OK)
    return st;
  Workspace& ws = workspace(device);
  HIPM_TRY(ws.ensure_stream());
  const float thres2 = sift_ratio_thres * sift_ratio_thres;
  const float* d = desc;
  if (!desc_on_device)
  {
    float* a = nullptr;
    HIPM_TRY(ws.get(Workspace::kDescA, size_t(n) * dim, a));
    HIPM_TRY(hipMemcpyAsync(a, desc, size_t(n) * dim * sizeof(float),
                            hipMemcpyHostToDevice, ws.stream));
    d = a;
  }
  std::vector<Feat> f(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i)
    f[size_t(i)] = feat_of(features[i]);
  const KeyProximity too_close{min_max_metric_dist_thres * min_max_metric_dist_thres,
                               pixel_dist_thres * pixel_dist_thres};
  // both trees of compute_matches index the same descriptors
  // (AnnMatcher.cpp:199-215): one search serves the two directions
  DeviceSearch ds;
  const sara_hip_status ss = device_search(ws, d, n, d, n, dim, thres2, 1, true, &ds);
  if (ss != SARA_HIP_OK)
    return ss;
  Neighbours nbs[2];
  const sara_hip_status hs = to_host(ws, ds, nbs);
  if (hs != SARA_HIP_OK)
    return hs;
  const Neighbours& nb = nbs[0];
  std::vector<sara_match> m;
  append_matches(nb, thres2, 0, true, too_close, f.data(), f.data(), m);
  append_matches(nb, thres2, 1, true, too_close, f.data(), f.data(), m);
  finish_matches(m, f.data(), f.data());
  return deliver(m, matches, capacity, count);
}

sara_hip_status sara_hip_match_release_workspace(int device)
{
  // only an existing workspace of the calling thread is released: looking one
  // up must not create (or evict) anything, and the current device is kept
  auto& all = workspaces();
  const auto it = all.find(device);
  if (it != all.end())
    all.erase(it);  // ~Workspace() frees on its device, restores the current one
  auto& lanes = lane_workspaces();
  for (int lane = 0; lane < kBatchLanes; ++lane)
  {
    const auto lt = lanes.find(device * kBatchLanes + lane);
    if (lt != lanes.end())
      lanes.erase(lt);
  }
  return SARA_HIP_OK;
}

}  // extern "C"
