// Multi-GPU side of the C-ABI (include/sara_hip_sift.h).
//
// Frames are independent (compute_sift_keypoints is a pure function of one
// image, FeatureDetectors/SIFT.cpp:27-108), so they are sharded in contiguous
// blocks and the ONLY exchange is the gather of the variable-length keypoint
// arrays to a root device: per-rank counts through one ncclAllGather, then one
// group of ncclSend / ncclRecv that lands every rank's OERegion[], descriptors
// and (s, o) pairs at its global offset on the root - a gatherv, which RCCL
// does not provide natively.  xGMI is point to point: each peer reaches the
// root over its own link.
//
// Two ways to form the group:
//   * one process per GPU (torchrun and the like): sara_hip_comm_unique_id /
//     sara_hip_comm_create (ncclCommInitRank; the caller ships the 128-byte id
//     between its processes by whatever transport it has);
//   * one process, one host thread per GPU: sara_hip_sift_group_* (ncclCommInitAll),
//     the call pattern a C++ consumer such as OdometryPipeline would use.
//
// Two transports carry the same exchange code:
//   * RCCL (librccl is loaded on first use, dlopen, so that single-GPU users of
//     the library do not depend on it);
//   * "loopback" (SARA_HIP_COMM_TRANSPORT=loopback): the ranks are threads of
//     ONE process - on one device or several - and AllGather / Send / Recv are
//     device copies through a process-local mailbox.  It exists so that every
//     branch of the N > 1 exchange (offsets, empty ranks, root != 0, failing
//     ranks) runs on a one-GPU box; it is not a data path anyone should ship.
//
// Error discipline: a collective in which one rank returns early leaves its
// peers blocked for ever.  Every rank therefore always takes part in every
// collective of a gather; a rank-local failure travels as a count of -1
// through the header AllGather, all ranks then skip the transfers together,
// GroupEnd is always called, the ticket is always released, and only then the
// rank-local error is returned.
#include "sift_kernels.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace sara_hip;

namespace {

  // ---- the few RCCL entry points used, resolved at run time ------------------
  typedef struct ncclComm* ncclComm_t;
  struct ncclUniqueId
  {
    char internal[SARA_HIP_COMM_ID_BYTES];
  };
  enum
  {
    kNcclInt8 = 0,
    kNcclInt32 = 2
  };

  struct Rccl
  {
    void* handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t,
                     hipStream_t) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    std::string error;
  };

  Rccl* rccl()
  {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
      const char* names[] = {"librccl.so.1", "librccl.so",
                             "/opt/rocm/lib/librccl.so.1"};
      // a copy the process has already loaded (e.g. a framework's) wins
      for (const char* n : names)
        if (!r.handle)
          r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
      for (const char* n : names)
        if (!r.handle)
          r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (!r.handle)
      {
        r.error = std::string("cannot load librccl: ") + dlerror();
        return;
      }
      auto sym = [&](const char* name) {
        void* p = dlsym(r.handle, name);
        if (!p && r.error.empty())
          r.error = std::string("librccl lacks ") + name;
        return p;
      };
#define SARA_RCCL_SYM(field, name)                                             \
  r.field = reinterpret_cast<decltype(r.field)>(sym(name))
      SARA_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
      SARA_RCCL_SYM(CommInitRank, "ncclCommInitRank");
      SARA_RCCL_SYM(CommInitAll, "ncclCommInitAll");
      SARA_RCCL_SYM(CommDestroy, "ncclCommDestroy");
      SARA_RCCL_SYM(GetErrorString, "ncclGetErrorString");
      SARA_RCCL_SYM(GroupStart, "ncclGroupStart");
      SARA_RCCL_SYM(GroupEnd, "ncclGroupEnd");
      SARA_RCCL_SYM(Send, "ncclSend");
      SARA_RCCL_SYM(Recv, "ncclRecv");
      SARA_RCCL_SYM(AllGather, "ncclAllGather");
      SARA_RCCL_SYM(GetVersion, "ncclGetVersion");
#undef SARA_RCCL_SYM
    });
    return &r;
  }

  sara_hip_status rccl_ready()
  {
    Rccl* r = rccl();
    if (!r->error.empty())
      return set_error(SARA_HIP_RCCL_ERROR, r->error.c_str());
    return SARA_HIP_OK;
  }

#define HIPC_TRY(expr)                                                         \
  do                                                                           \
  {                                                                            \
    const hipError_t e_ = (expr);                                              \
    if (e_ != hipSuccess)                                                      \
      return set_error(SARA_HIP_RUNTIME_ERROR,                                 \
                       (std::string(#expr) + ": " + hipGetErrorString(e_))     \
                           .c_str());                                          \
  } while (0)

  bool loopback_requested()
  {
    const char* e = getenv("SARA_HIP_COMM_TRANSPORT");
    return e && std::string(e) == "loopback";
  }

  // ---- transport: what the gather needs from a communicator -----------------
  struct Transport
  {
    virtual ~Transport() {}
    virtual const char* name() const = 0;
    //! `count` int32 from every rank, in rank order, into recv (device memory).
    virtual sara_hip_status all_gather_i32(const int* d_send, int* d_recv,
                                           size_t count, hipStream_t s) = 0;
    virtual sara_hip_status group_start() = 0;
    virtual sara_hip_status send(const void* d_src, size_t bytes, int peer,
                                 hipStream_t s) = 0;
    virtual sara_hip_status recv(void* d_dst, size_t bytes, int peer,
                                 hipStream_t s) = 0;
    //! Must be called after every group_start(), whatever happened in between.
    virtual sara_hip_status group_end(hipStream_t s) = 0;
  };

  struct RcclTransport final : Transport
  {
    ncclComm_t comm = nullptr;
    bool owns = true;

    ~RcclTransport() override
    {
      if (comm && owns && rccl()->CommDestroy)
        (void) rccl()->CommDestroy(comm);
    }
    const char* name() const override { return "rccl"; }
    static sara_hip_status fail(const char* what, int e)
    {
      return set_error(SARA_HIP_RCCL_ERROR,
                       (std::string(what) + ": " + rccl()->GetErrorString(e)).c_str());
    }
    sara_hip_status all_gather_i32(const int* d_send, int* d_recv, size_t count,
                                   hipStream_t s) override
    {
      const int e = rccl()->AllGather(d_send, d_recv, count, kNcclInt32, comm, s);
      return e ? fail("ncclAllGather", e) : SARA_HIP_OK;
    }
    sara_hip_status group_start() override
    {
      const int e = rccl()->GroupStart();
      return e ? fail("ncclGroupStart", e) : SARA_HIP_OK;
    }
    sara_hip_status send(const void* p, size_t bytes, int peer, hipStream_t s) override
    {
      const int e = rccl()->Send(p, bytes, kNcclInt8, peer, comm, s);
      return e ? fail("ncclSend", e) : SARA_HIP_OK;
    }
    sara_hip_status recv(void* p, size_t bytes, int peer, hipStream_t s) override
    {
      const int e = rccl()->Recv(p, bytes, kNcclInt8, peer, comm, s);
      return e ? fail("ncclRecv", e) : SARA_HIP_OK;
    }
    sara_hip_status group_end(hipStream_t) override
    {
      const int e = rccl()->GroupEnd();
      return e ? fail("ncclGroupEnd", e) : SARA_HIP_OK;
    }
  };

  // ---- loopback: the ranks are threads of this process ------------------------
  struct LoopWorld
  {
    struct Post
    {
      int peer;
      const void* src;
      size_t bytes;
    };
    int n = 1;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long generation = 0;
    bool broken = false;
    std::vector<const int*> ag_src;
    std::vector<std::vector<Post>> posted;  // [rank]: sends of the open group
    int timeout_ms = 20000;

    explicit LoopWorld(int nranks)
      : n(nranks)
      , ag_src(size_t(nranks), nullptr)
      , posted(size_t(nranks))
    {
      if (const char* e = getenv("SARA_HIP_LOOPBACK_TIMEOUT_MS"))
        timeout_ms = std::max(1, atoi(e));
    }

    //! All n ranks meet here; false when a peer never shows up (the world is
    //! then broken for good: later barriers fail at once instead of hanging).
    bool barrier()
    {
      std::unique_lock<std::mutex> lock(m);
      if (broken)
        return false;
      const unsigned long long gen = generation;
      if (++arrived == n)
      {
        arrived = 0;
        ++generation;
        cv.notify_all();
        return true;
      }
      const bool ok = cv.wait_for(lock, std::chrono::milliseconds(timeout_ms),
                                  [&] { return generation != gen || broken; });
      if (!ok || broken)
      {
        broken = true;
        cv.notify_all();
        return false;
      }
      return true;
    }
  };

  struct LoopbackTransport final : Transport
  {
    std::shared_ptr<LoopWorld> world;
    int rank = 0;
    struct PendingRecv
    {
      int peer;
      void* dst;
      size_t bytes;
    };
    std::vector<PendingRecv> recvs;
    std::vector<LoopWorld::Post> sends;

    const char* name() const override { return "loopback"; }
    static sara_hip_status lost()
    {
      return set_error(SARA_HIP_RCCL_ERROR,
                       "loopback transport: a peer rank did not reach the "
                       "collective (timeout)");
    }
    sara_hip_status all_gather_i32(const int* d_send, int* d_recv, size_t count,
                                   hipStream_t s) override
    {
      // the send buffer must be complete before a peer reads it
      HIPC_TRY(hipStreamSynchronize(s));
      {
        std::lock_guard<std::mutex> lock(world->m);
        world->ag_src[size_t(rank)] = d_send;
      }
      if (!world->barrier())
        return lost();
      sara_hip_status st = SARA_HIP_OK;
      for (int k = 0; k < world->n && st == SARA_HIP_OK; ++k)
      {
        const hipError_t e =
            hipMemcpyAsync(d_recv + size_t(k) * count, world->ag_src[size_t(k)],
                           count * sizeof(int), hipMemcpyDefault, s);
        if (e != hipSuccess)
          st = set_error(SARA_HIP_RUNTIME_ERROR, hipGetErrorString(e));
      }
      if (hipStreamSynchronize(s) != hipSuccess && st == SARA_HIP_OK)
        st = set_error(SARA_HIP_RUNTIME_ERROR, "loopback all_gather: copy failed");
      if (!world->barrier())  // the sources may be overwritten from here on
        return lost();
      return st;
    }
    sara_hip_status group_start() override
    {
      recvs.clear();
      sends.clear();
      return SARA_HIP_OK;
    }
    sara_hip_status send(const void* p, size_t bytes, int peer, hipStream_t) override
    {
      sends.push_back({peer, p, bytes});
      return SARA_HIP_OK;
    }
    sara_hip_status recv(void* p, size_t bytes, int peer, hipStream_t) override
    {
      recvs.push_back({peer, p, bytes});
      return SARA_HIP_OK;
    }
    sara_hip_status group_end(hipStream_t s) override
    {
      {
        std::lock_guard<std::mutex> lock(world->m);
        world->posted[size_t(rank)] = sends;
      }
      if (!world->barrier())
        return lost();
      // the k-th receive from a peer pairs with that peer's k-th send to me
      sara_hip_status st = SARA_HIP_OK;
      std::vector<size_t> cursor(size_t(world->n), 0);
      for (const PendingRecv& r : recvs)
      {
        const auto& theirs = world->posted[size_t(r.peer)];
        size_t& at = cursor[size_t(r.peer)];
        while (at < theirs.size() && theirs[at].peer != rank)
          ++at;
        if (at == theirs.size() || theirs[at].bytes != r.bytes)
        {
          if (st == SARA_HIP_OK)
            st = set_error(SARA_HIP_RCCL_ERROR,
                           "loopback transport: a receive has no matching send "
                           "of the same size");
          continue;
        }
        const hipError_t e = hipMemcpyAsync(r.dst, theirs[at].src, r.bytes,
                                            hipMemcpyDefault, s);
        ++at;
        if (e != hipSuccess && st == SARA_HIP_OK)
          st = set_error(SARA_HIP_RUNTIME_ERROR, hipGetErrorString(e));
      }
      if (hipStreamSynchronize(s) != hipSuccess && st == SARA_HIP_OK)
        st = set_error(SARA_HIP_RUNTIME_ERROR, "loopback group_end: copy failed");
      recvs.clear();
      sends.clear();
      if (!world->barrier())  // senders may reuse their buffers from here on
        return lost();
      return st;
    }
  };

  const char kLoopMagic[] = "SARA-LOOPBACK-ID";  // 16 bytes + counter

  std::mutex g_loop_mutex;
  std::map<std::string, std::weak_ptr<LoopWorld>> g_loop_worlds;

  std::shared_ptr<LoopWorld> loop_world_of(const unsigned char* id, int nranks)
  {
    const std::string key(reinterpret_cast<const char*>(id), SARA_HIP_COMM_ID_BYTES);
    std::lock_guard<std::mutex> lock(g_loop_mutex);
    std::shared_ptr<LoopWorld> w = g_loop_worlds[key].lock();
    if (!w)
    {
      w = std::make_shared<LoopWorld>(nranks);
      g_loop_worlds[key] = w;
    }
    return w->n == nranks ? w : nullptr;
  }

}  // namespace

struct sara_hip_comm
{
  sara_hip_sift* ctx = nullptr;
  std::unique_ptr<Transport> tr;
  int nranks = 1, rank = 0, device = 0;
  hipStream_t stream = nullptr;  // the exchange runs beside the next batch
  // header of a gather: kHdr ints per rank (count or -1, root capacity);
  // [nranks * kHdr ..] is this rank's own entry, the AllGather's send buffer;
  // [(nranks + 1) * kHdr ..] is a constant entry of -1s, sent instead when the
  // upload of the own entry fails (the collective still has to be entered)
  int* d_hdr = nullptr;
  int* h_hdr = nullptr;  // pinned, same layout
  // gather buffers on the root (grown on demand)
  sara_oeregion* d_feat = nullptr;
  float* d_desc = nullptr;
  int32_t* d_so = nullptr;
  size_t cap = 0;
};

namespace {

  constexpr int kHdr = 4;

  sara_hip_status comm_finish_create(sara_hip_comm* c)
  {
    HIPC_TRY(hipSetDevice(c->device));
    std::lock_guard<std::recursive_mutex> runtime_lock(runtime_mutex());
    HIPC_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const size_t ints = size_t(kHdr) * (size_t(c->nranks) + 2);
    HIPC_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_hdr), sizeof(int) * ints));
    HIPC_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_hdr), sizeof(int) * ints));
    HIPC_TRY(hipMemset(c->d_hdr + size_t(kHdr) * (size_t(c->nranks) + 1), 0xff,
                       sizeof(int) * kHdr));
    return SARA_HIP_OK;
  }

  //! What one rank brings to a gather.  Never touches the transport.
  struct Local
  {
    TicketResults res;
    int mine = 0;               // keypoints of this rank, -1 = this rank failed
    bool ticket_valid = false;  // the ticket is pending and must be released
    sara_hip_status status = SARA_HIP_OK;
    std::string message;
  };

  void local_fail(Local* l, sara_hip_status st)
  {
    l->status = st;
    l->message = sara_hip_last_error();
    l->mine = -1;
  }

  //! ticket < 0 with `empty_ok`: a rank of the group without frames.
  void gather_local(sara_hip_comm* c, int ticket, int root, int with_descriptors,
                    bool empty_ok, Local* l)
  {
    *l = Local();
    if (root < 0 || root >= c->nranks)
      return local_fail(l, set_error(SARA_HIP_INVALID_PARAMS, "root rank out of range"));
    if (ticket < 0 && empty_ok)
      return;
    const sara_hip_status st = ticket_results(c->ctx, ticket, &l->res);
    if (st != SARA_HIP_OK)
      return local_fail(l, st);
    l->ticket_valid = true;
    l->mine = l->res.total;
    if (with_descriptors && l->res.last_stage < SARA_HIP_STAGE_DESCRIPTOR)
      return local_fail(
          l, set_error(SARA_HIP_NOT_READY,
                       "descriptors requested, but the ticket was submitted "
                       "with last_stage < DESCRIPTOR"));
    if (hipSetDevice(c->device) != hipSuccess)
      return local_fail(l, set_error(SARA_HIP_RUNTIME_ERROR, "hipSetDevice failed"));
  }

  //! Root only: room for `sum` keypoints.
  sara_hip_status gather_reserve(sara_hip_comm* c, size_t sum)
  {
    if (sum <= c->cap)
      return SARA_HIP_OK;
    HIPC_TRY(hipSetDevice(c->device));
    if (c->d_feat)
      (void) hipFree(c->d_feat);
    if (c->d_desc)
      (void) hipFree(c->d_desc);
    if (c->d_so)
      (void) hipFree(c->d_so);
    c->d_feat = nullptr;
    c->d_desc = nullptr;
    c->d_so = nullptr;
    c->cap = 0;
    const size_t want = sum + sum / 4 + 1024;
    HIPC_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_feat),
                       sizeof(sara_oeregion) * want));
    HIPC_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_desc),
                       sizeof(float) * 128 * want));
    HIPC_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_so),
                       sizeof(int32_t) * 2 * want));
    c->cap = want;
    return SARA_HIP_OK;
  }

  //! One group of point-to-point transfers at the global offsets.  Collective:
  //! every rank calls it with the same counts; the group is always closed.
  sara_hip_status gather_exchange(sara_hip_comm* c, const Local& l,
                                  const int* counts, int root,
                                  int with_descriptors)
  {
    Transport* t = c->tr.get();
    std::vector<size_t> offset(size_t(c->nranks) + 1, 0);
    for (int k = 0; k < c->nranks; ++k)
      offset[size_t(k) + 1] = offset[size_t(k)] + size_t(counts[k]);
    sara_hip_status first = t->group_start();
    auto keep = [&](sara_hip_status st) {
      if (first == SARA_HIP_OK && st != SARA_HIP_OK)
        first = st;
    };
    if (first == SARA_HIP_OK)
    {
      if (c->rank == root)
      {
        for (int k = 0; k < c->nranks; ++k)
        {
          const size_t n = size_t(counts[k]), at = offset[size_t(k)];
          if (k == root || n == 0)
            continue;
          keep(t->recv(c->d_feat + at, n * sizeof(sara_oeregion), k, c->stream));
          keep(t->recv(c->d_so + 2 * at, n * 2 * sizeof(int32_t), k, c->stream));
          if (with_descriptors)
            keep(t->recv(c->d_desc + 128 * at, n * 128 * sizeof(float), k,
                         c->stream));
        }
      }
      else if (l.mine > 0)
      {
        const size_t n = size_t(l.mine);
        keep(t->send(l.res.d_feat, n * sizeof(sara_oeregion), root, c->stream));
        keep(t->send(l.res.d_so, n * 2 * sizeof(int32_t), root, c->stream));
        if (with_descriptors)
          keep(t->send(l.res.d_desc, n * 128 * sizeof(float), root, c->stream));
      }
      // the group is closed whatever happened above: an open group would leave
      // the peers of the transfers already posted blocked
      keep(t->group_end(c->stream));
    }
    if (c->rank == root && l.mine > 0 && first == SARA_HIP_OK)
    {
      const size_t n = size_t(l.mine), at = offset[size_t(root)];
      auto copy = [&](void* dst, const void* src, size_t bytes) {
        if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream) !=
            hipSuccess)
          keep(set_error(SARA_HIP_RUNTIME_ERROR, "root: local copy failed"));
      };
      copy(c->d_feat + at, l.res.d_feat, n * sizeof(sara_oeregion));
      copy(c->d_so + 2 * at, l.res.d_so, n * 2 * sizeof(int32_t));
      if (with_descriptors)
        copy(c->d_desc + 128 * at, l.res.d_desc, n * 128 * sizeof(float));
    }
    // the ticket's result slot may be overwritten once the transfers are done
    if (hipStreamSynchronize(c->stream) != hipSuccess)
      keep(set_error(SARA_HIP_RUNTIME_ERROR, "gather: stream synchronisation failed"));
    return first;
  }

  void gather_outputs(sara_hip_comm* c, const int* counts, int root,
                      int with_descriptors, int* counts_out,
                      const sara_oeregion** d_features, const float** d_descriptors,
                      const int32_t** d_scale_octave, int* total)
  {
    size_t sum = 0;
    for (int k = 0; k < c->nranks; ++k)
      sum += size_t(counts[k]);
    if (counts_out)
      std::copy(counts, counts + c->nranks, counts_out);
    if (total)
      *total = int(sum);
    const bool here = c->rank == root;
    if (d_features)
      *d_features = here ? c->d_feat : nullptr;
    if (d_descriptors)
      *d_descriptors = here && with_descriptors ? c->d_desc : nullptr;
    if (d_scale_octave)
      *d_scale_octave = here ? c->d_so : nullptr;
  }

  sara_hip_status capacity_status(const Local& l)
  {
    if (l.ticket_valid && l.res.capacity_exceeded)
      return set_error(SARA_HIP_CAPACITY_EXCEEDED,
                       "a frame produced more extrema / keypoints than "
                       "max_keypoints: the lists are truncated");
    return SARA_HIP_OK;
  }

  //! Header exchange of the process-per-GPU form: own entry -> everyone's.
  //! Collective: the AllGather is posted whatever happens locally (a rank that
  //! returned before it would leave its peers blocked in theirs); when the
  //! upload of the own entry fails the constant -1 entry travels instead, and
  //! the local error is returned once the collective has been entered.
  sara_hip_status exchange_header(sara_hip_comm* c, int v0, int v1)
  {
    int* mine = c->h_hdr + size_t(kHdr) * c->nranks;
    mine[0] = v0;
    mine[1] = v1;
    mine[2] = mine[3] = 0;
    const int* d_send = c->d_hdr + size_t(kHdr) * c->nranks;
    sara_hip_status local = SARA_HIP_OK;
    const hipError_t up =
        hipMemcpyAsync(c->d_hdr + size_t(kHdr) * c->nranks, mine,
                       sizeof(int) * kHdr, hipMemcpyHostToDevice, c->stream);
    if (up != hipSuccess)
    {
      local = set_error(SARA_HIP_RUNTIME_ERROR,
                        (std::string("gather header upload: ") +
                         hipGetErrorString(up)).c_str());
      d_send = c->d_hdr + size_t(kHdr) * (size_t(c->nranks) + 1);
    }
    const std::string local_msg = local != SARA_HIP_OK ? sara_hip_last_error() : "";
    const sara_hip_status st =
        c->tr->all_gather_i32(d_send, c->d_hdr, kHdr, c->stream);
    if (local != SARA_HIP_OK)
      return set_error(local, local_msg.c_str());
    if (st != SARA_HIP_OK)
      return st;
    HIPC_TRY(hipMemcpyAsync(c->h_hdr, c->d_hdr, sizeof(int) * kHdr * c->nranks,
                            hipMemcpyDeviceToHost, c->stream));
    HIPC_TRY(hipStreamSynchronize(c->stream));
    return SARA_HIP_OK;
  }

  sara_hip_status peer_failed(const sara_hip_comm* c, const Local& l, int who)
  {
    if (l.status != SARA_HIP_OK)
      return set_error(l.status, l.message.c_str());
    return set_error(SARA_HIP_RCCL_ERROR,
                     ("gather abandoned: rank " + std::to_string(who) + " of " +
                      std::to_string(c->nranks) + " reported a failure")
                         .c_str());
  }

  //! The gatherv of one ticket, one process per GPU; see the header.
  sara_hip_status comm_gather(sara_hip_comm* c, int ticket, int root,
                              int with_descriptors, int* counts_per_rank,
                              const sara_oeregion** d_features,
                              const float** d_descriptors,
                              const int32_t** d_scale_octave, int* total)
  {
    if (!c)
      return set_error(SARA_HIP_INVALID_PARAMS, "null communicator");
    Local l;
    gather_local(c, ticket, root, with_descriptors, false, &l);
    (void) hipSetDevice(c->device);
    auto release = [&] {
      if (l.ticket_valid)
        ticket_release(c->ctx, ticket);
    };
    // 1. every rank's keypoint count (-1: that rank cannot take part) and the
    //    room it would have as the root
    sara_hip_status st = exchange_header(
        c, l.mine, int(std::min<size_t>(c->cap, size_t(INT_MAX))));
    if (st != SARA_HIP_OK)
    {
      release();
      return st;
    }
    std::vector<int> counts(size_t(c->nranks), 0);
    int failed = -1;
    size_t sum = 0;
    for (int k = 0; k < c->nranks; ++k)
    {
      counts[size_t(k)] = c->h_hdr[size_t(kHdr) * k];
      if (counts[size_t(k)] < 0 && failed < 0)
        failed = k;
      sum += size_t(std::max(counts[size_t(k)], 0));
    }
    if (failed >= 0)
    {
      release();
      return peer_failed(c, l, failed);
    }
    // 2. room on the root.  Every rank sees the root's capacity in the header,
    //    so all of them know whether the root has to allocate - and only then
    //    wait for its verdict (a root that returned on a failed hipMalloc
    //    would leave the senders blocked).
    const size_t root_cap = size_t(c->h_hdr[size_t(kHdr) * root + 1]);
    if (sum > root_cap)
    {
      sara_hip_status rs = SARA_HIP_OK;
      std::string rmsg;
      if (c->rank == root)
      {
        rs = gather_reserve(c, sum);
        if (rs != SARA_HIP_OK)
          rmsg = sara_hip_last_error();
      }
      st = exchange_header(c, rs == SARA_HIP_OK ? 0 : -1, 0);
      // every entry, as in step 1: a rank whose header upload failed sent the
      // constant -1 entry and is about to return its local error - peers that
      // only looked at the root's entry would go on to the transfers and block
      // on that rank's send / recv
      int bad = -1;
      if (st == SARA_HIP_OK)
        for (int k = 0; k < c->nranks && bad < 0; ++k)
          if (c->h_hdr[size_t(kHdr) * k] < 0)
            bad = k;
      if (bad == root)
        st = rs != SARA_HIP_OK
                 ? set_error(rs, rmsg.c_str())
                 : set_error(SARA_HIP_RCCL_ERROR,
                             "gather abandoned: the root could not allocate "
                             "its receive buffers");
      else if (bad >= 0)
        st = peer_failed(c, l, bad);
      if (st != SARA_HIP_OK)
      {
        release();
        return st;
      }
    }
    // 3. the transfers
    st = gather_exchange(c, l, counts.data(), root, with_descriptors);
    release();
    gather_outputs(c, counts.data(), root, with_descriptors, counts_per_rank,
                   d_features, d_descriptors, d_scale_octave, total);
    if (st != SARA_HIP_OK)
      return st;
    return capacity_status(l);
  }

}  // namespace

extern "C" {

void sara_hip_shard_range(int n_frames, int world_size, int rank, int* lo, int* hi)
{
  // frame f belongs to rank floor(f * world_size / n_frames): contiguous blocks
  const long long n = n_frames, w = std::max(world_size, 1);
  if (lo)
    *lo = int((rank * n + w - 1) / w);
  if (hi)
    *hi = int(((rank + 1) * n + w - 1) / w);
}

sara_hip_status sara_hip_copy_to_host(void* dst, const void* src_device,
                                      size_t bytes, int device)
{
  if (!dst || !src_device)
    return set_error(SARA_HIP_INVALID_PARAMS, "null pointer");
  HIPC_TRY(hipSetDevice(device));
  HIPC_TRY(hipMemcpy(dst, src_device, bytes, hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_copy_to_device(void* dst_device, const void* src,
                                        size_t bytes, int device)
{
  if (!dst_device || !src)
    return set_error(SARA_HIP_INVALID_PARAMS, "null pointer");
  HIPC_TRY(hipSetDevice(device));
  HIPC_TRY(hipMemcpy(dst_device, src, bytes, hipMemcpyHostToDevice));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_device_alloc(void** ptr, size_t bytes, int device)
{
  if (!ptr || !bytes)
    return set_error(SARA_HIP_INVALID_PARAMS, "null pointer or zero size");
  *ptr = nullptr;
  HIPC_TRY(hipSetDevice(device));
  HIPC_TRY(hipMalloc(ptr, bytes));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_device_free(void* ptr, int device)
{
  if (!ptr)
    return set_error(SARA_HIP_INVALID_PARAMS, "null pointer");
  HIPC_TRY(hipSetDevice(device));
  HIPC_TRY(hipFree(ptr));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_host_register(void* ptr, size_t bytes)
{
  if (!ptr || !bytes)
    return set_error(SARA_HIP_INVALID_PARAMS, "null pointer or zero size");
  HIPC_TRY(hipHostRegister(ptr, bytes, hipHostRegisterPortable));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_host_unregister(void* ptr)
{
  if (!ptr)
    return set_error(SARA_HIP_INVALID_PARAMS, "null pointer");
  HIPC_TRY(hipHostUnregister(ptr));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_host_alloc(void** ptr, size_t bytes)
{
  if (!ptr || !bytes)
    return set_error(SARA_HIP_INVALID_PARAMS, "null pointer or zero size");
  *ptr = nullptr;
  HIPC_TRY(hipHostMalloc(ptr, bytes, hipHostMallocPortable));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_host_free(void* ptr)
{
  if (!ptr)
    return set_error(SARA_HIP_INVALID_PARAMS, "null pointer");
  HIPC_TRY(hipHostFree(ptr));
  return SARA_HIP_OK;
}

const char* sara_hip_comm_transport(const sara_hip_comm* c)
{
  return c && c->tr ? c->tr->name() : "";
}

int sara_hip_comm_size(const sara_hip_comm* c) { return c ? c->nranks : 0; }

sara_hip_status sara_hip_rccl_version(int* version)
{
  if (!version)
    return set_error(SARA_HIP_INVALID_PARAMS, "null output");
  *version = 0;
  const sara_hip_status st = rccl_ready();
  if (st != SARA_HIP_OK)
    return st;
  const int e = rccl()->GetVersion(version);
  return e ? RcclTransport::fail("ncclGetVersion", e) : SARA_HIP_OK;
}

sara_hip_status sara_hip_comm_unique_id(unsigned char* id)
{
  if (!id)
    return set_error(SARA_HIP_INVALID_PARAMS, "null id");
  if (loopback_requested())
  {
    static std::mutex m;
    static unsigned long long counter = 0;
    std::lock_guard<std::mutex> lock(m);
    std::memset(id, 0, SARA_HIP_COMM_ID_BYTES);
    std::memcpy(id, kLoopMagic, sizeof(kLoopMagic) - 1);
    const unsigned long long k = ++counter;
    std::memcpy(id + 16, &k, sizeof(k));
    return SARA_HIP_OK;
  }
  const sara_hip_status st = rccl_ready();
  if (st != SARA_HIP_OK)
    return st;
  ncclUniqueId u;
  const int e = rccl()->GetUniqueId(&u);
  if (e != 0)
    return RcclTransport::fail("ncclGetUniqueId", e);
  std::memcpy(id, u.internal, SARA_HIP_COMM_ID_BYTES);
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_comm_create(sara_hip_sift* ctx, const unsigned char* id,
                                     int nranks, int rank, int device,
                                     sara_hip_comm** out)
{
  if (!ctx || !id || !out)
    return set_error(SARA_HIP_INVALID_PARAMS, "null context, id or output");
  *out = nullptr;
  if (nranks < 1 || rank < 0 || rank >= nranks)
    return set_error(SARA_HIP_INVALID_PARAMS, "rank / nranks");
  const bool loop = std::memcmp(id, kLoopMagic, sizeof(kLoopMagic) - 1) == 0;
  if (!loop)
  {
    const sara_hip_status st = rccl_ready();
    if (st != SARA_HIP_OK)
      return st;
  }
  HIPC_TRY(hipSetDevice(device));
  std::unique_ptr<sara_hip_comm> c(new sara_hip_comm);
  c->ctx = ctx;
  c->nranks = nranks;
  c->rank = rank;
  c->device = device;
  if (loop)
  {
    std::unique_ptr<LoopbackTransport> t(new LoopbackTransport);
    t->world = loop_world_of(id, nranks);
    if (!t->world)
      return set_error(SARA_HIP_INVALID_PARAMS,
                       "loopback id already in use with another rank count");
    t->rank = rank;
    c->tr = std::move(t);
  }
  else
  {
    ncclUniqueId u;
    std::memcpy(u.internal, id, SARA_HIP_COMM_ID_BYTES);
    std::unique_ptr<RcclTransport> t(new RcclTransport);
    const int e = rccl()->CommInitRank(&t->comm, nranks, u, rank);
    if (e != 0)
    {
      t->comm = nullptr;
      return RcclTransport::fail("ncclCommInitRank", e);
    }
    c->tr = std::move(t);
  }
  const sara_hip_status fs = comm_finish_create(c.get());
  if (fs != SARA_HIP_OK)
  {
    sara_hip_comm_destroy(c.release());
    return fs;
  }
  *out = c.release();
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_comm_gather(sara_hip_comm* comm, int ticket, int root,
                                     int with_descriptors, int* counts_per_rank,
                                     const sara_oeregion** d_features,
                                     const float** d_descriptors,
                                     const int32_t** d_scale_octave, int* total)
{
  return comm_gather(comm, ticket, root, with_descriptors, counts_per_rank,
                     d_features, d_descriptors, d_scale_octave, total);
}

sara_hip_status sara_hip_comm_destroy(sara_hip_comm* c)
{
  if (!c)
    return SARA_HIP_OK;
  (void) hipSetDevice(c->device);
  if (c->stream)
  {
    (void) hipStreamSynchronize(c->stream);
    std::lock_guard<std::recursive_mutex> runtime_lock(runtime_mutex());
    (void) hipStreamDestroy(c->stream);
  }
  c->tr.reset();
  if (c->d_hdr)
    (void) hipFree(c->d_hdr);
  if (c->h_hdr)
    (void) hipHostFree(c->h_hdr);
  if (c->d_feat)
    (void) hipFree(c->d_feat);
  if (c->d_desc)
    (void) hipFree(c->d_desc);
  if (c->d_so)
    (void) hipFree(c->d_so);
  delete c;
  return SARA_HIP_OK;
}

}  // extern "C"

// ---- single process, one host thread per device ------------------------------
struct sara_hip_sift_group
{
  int n = 0;
  std::vector<int> devices;
  std::vector<sara_hip_sift*> ctx;
  std::vector<sara_hip_comm*> comm;
  std::vector<int> ticket;  // -1: nothing to gather, -2: empty shard
  std::vector<std::string> errors;
  // sara_hip_sift_group_collect_host(): ONE pinned buffer for the whole node,
  // every device copies its shard to its global offset over its own PCIe link
  sara_oeregion* h_feat = nullptr;
  float* h_desc = nullptr;
  int32_t* h_so = nullptr;
  size_t h_cap = 0;
};

namespace {
  constexpr int kNoTicket = -1, kEmptyShard = -2;

  //! Runs fn(i) on one host thread per device and returns the first failure.
  template <typename F>
  sara_hip_status on_every_device(sara_hip_sift_group* g, F fn)
  {
    std::vector<sara_hip_status> st(size_t(g->n), SARA_HIP_OK);
    g->errors.assign(size_t(g->n), std::string());
    std::vector<std::thread> pool;
    for (int i = 0; i < g->n; ++i)
      pool.emplace_back([&, i] {
        st[size_t(i)] = fn(i);
        if (st[size_t(i)] != SARA_HIP_OK)
          g->errors[size_t(i)] = sara_hip_last_error();  // thread-local message
      });
    for (auto& t : pool)
      t.join();
    for (int i = 0; i < g->n; ++i)
      if (st[size_t(i)] != SARA_HIP_OK)
        return set_error(st[size_t(i)], ("device " + std::to_string(g->devices[size_t(i)]) +
                                         ": " + g->errors[size_t(i)])
                                            .c_str());
    return SARA_HIP_OK;
  }

  //! Phase 1 of both group gathers: every device waits for its batch.  On a
  //! failure all pending tickets are released and the group is back to "no
  //! batch".
  sara_hip_status group_locals(sara_hip_sift_group* g, int root,
                               int with_descriptors, std::vector<Local>* locals,
                               std::vector<int>* counts)
  {
    bool any = false;
    for (int i = 0; i < g->n; ++i)
      any = any || g->ticket[size_t(i)] >= 0;
    if (!any)
      return set_error(SARA_HIP_NOT_READY, "no group_detect() to gather");
    locals->assign(size_t(g->n), Local());
    (void) on_every_device(g, [&](int i) {
      gather_local(g->comm[size_t(i)], g->ticket[size_t(i)], root,
                   with_descriptors, g->ticket[size_t(i)] == kEmptyShard,
                   &(*locals)[size_t(i)]);
      return SARA_HIP_OK;
    });
    counts->assign(size_t(g->n), 0);
    for (int i = 0; i < g->n; ++i)
    {
      const Local& l = (*locals)[size_t(i)];
      if (l.mine < 0)
      {
        const sara_hip_status st = l.status;
        const std::string msg =
            "device " + std::to_string(g->devices[size_t(i)]) + ": " + l.message;
        for (int k = 0; k < g->n; ++k)
        {
          if ((*locals)[size_t(k)].ticket_valid)
            ticket_release(g->ctx[size_t(k)], g->ticket[size_t(k)]);
          g->ticket[size_t(k)] = kNoTicket;
        }
        return set_error(st, msg.c_str());
      }
      (*counts)[size_t(i)] = l.mine;
    }
    return SARA_HIP_OK;
  }

  void group_release(sara_hip_sift_group* g, const std::vector<Local>& locals)
  {
    for (int k = 0; k < g->n; ++k)
    {
      if (locals[size_t(k)].ticket_valid)
        ticket_release(g->ctx[size_t(k)], g->ticket[size_t(k)]);
      g->ticket[size_t(k)] = kNoTicket;
    }
  }
}  // namespace

extern "C" {

sara_hip_status sara_hip_sift_group_destroy(sara_hip_sift_group* g)
{
  if (!g)
    return SARA_HIP_OK;
  for (sara_hip_comm* c : g->comm)
    (void) sara_hip_comm_destroy(c);
  for (sara_hip_sift* c : g->ctx)
    (void) sara_hip_sift_destroy(c);
  if (g->h_feat)
    (void) hipHostFree(g->h_feat);
  if (g->h_desc)
    (void) hipHostFree(g->h_desc);
  if (g->h_so)
    (void) hipHostFree(g->h_so);
  delete g;
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_sift_group_create(const sara_sift_params* params,
                                           int max_width, int max_height,
                                           int max_batch_per_device,
                                           int max_keypoints, int n_dev,
                                           const int* devices,
                                           sara_hip_sift_group** out)
{
  if (!params || !out || n_dev < 1)
    return set_error(SARA_HIP_INVALID_PARAMS, "null params / output or n_dev < 1");
  *out = nullptr;
  const int visible = sara_hip_device_count();
  if (visible <= 0)
    return set_error(SARA_HIP_NO_DEVICE,
                     "no HIP device: the SIFT front-end has no CPU fallback");
  const bool loop = loopback_requested();
  sara_hip_status st = loop ? SARA_HIP_OK : rccl_ready();
  if (st != SARA_HIP_OK)
    return st;
  auto* g = new sara_hip_sift_group;
  g->n = n_dev;
  for (int i = 0; i < n_dev; ++i)
    g->devices.push_back(devices ? devices[i] : (loop ? i % visible : i));
  for (int d : g->devices)
    if (d < 0 || d >= visible)
    {
      delete g;
      return set_error(SARA_HIP_INVALID_PARAMS, "device ordinal out of range");
    }
  g->ctx.assign(size_t(n_dev), nullptr);
  g->comm.assign(size_t(n_dev), nullptr);
  g->ticket.assign(size_t(n_dev), kNoTicket);
  st = on_every_device(g, [&](int i) {
    return sara_hip_sift_create(params, max_width, max_height,
                                max_batch_per_device, max_keypoints,
                                g->devices[size_t(i)], &g->ctx[size_t(i)]);
  });
  if (st != SARA_HIP_OK)
  {
    sara_hip_sift_group_destroy(g);
    return st;
  }
  std::vector<ncclComm_t> comms(size_t(n_dev), nullptr);
  std::shared_ptr<LoopWorld> world;
  if (loop)
    world = std::make_shared<LoopWorld>(n_dev);
  else
  {
    const int e = rccl()->CommInitAll(comms.data(), n_dev, g->devices.data());
    if (e != 0)
    {
      sara_hip_sift_group_destroy(g);
      return RcclTransport::fail("ncclCommInitAll", e);
    }
  }
  for (int i = 0; i < n_dev; ++i)
  {
    auto* c = new sara_hip_comm;
    c->ctx = g->ctx[size_t(i)];
    if (loop)
    {
      std::unique_ptr<LoopbackTransport> t(new LoopbackTransport);
      t->world = world;
      t->rank = i;
      c->tr = std::move(t);
    }
    else
    {
      std::unique_ptr<RcclTransport> t(new RcclTransport);
      t->comm = comms[size_t(i)];
      c->tr = std::move(t);
    }
    c->nranks = n_dev;
    c->rank = i;
    c->device = g->devices[size_t(i)];
    g->comm[size_t(i)] = c;
    st = comm_finish_create(c);
    if (st != SARA_HIP_OK)
    {
      // communicators not wrapped yet would leak: wrap them first
      for (int k = i + 1; k < n_dev && !loop; ++k)
        if (comms[size_t(k)])
          (void) rccl()->CommDestroy(comms[size_t(k)]);
      sara_hip_sift_group_destroy(g);
      return st;
    }
  }
  *out = g;
  return SARA_HIP_OK;
}

int sara_hip_sift_group_size(const sara_hip_sift_group* g) { return g ? g->n : 0; }

const char* sara_hip_sift_group_transport(const sara_hip_sift_group* g)
{
  return g && g->n > 0 ? sara_hip_comm_transport(g->comm[0]) : "";
}

sara_hip_status sara_hip_sift_group_context(sara_hip_sift_group* g, int index,
                                            sara_hip_sift** ctx)
{
  if (!g || !ctx || index < 0 || index >= g->n)
    return set_error(SARA_HIP_INVALID_PARAMS, "group / index");
  *ctx = g->ctx[size_t(index)];
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_sift_group_detect(sara_hip_sift_group* g,
                                           const void* const* shard_images,
                                           const int* shard_batch,
                                           size_t frame_stride, int channels,
                                           int width, int height,
                                           int images_on_device,
                                           sara_hip_stage last_stage)
{
  if (!g || !shard_images || !shard_batch)
    return set_error(SARA_HIP_INVALID_PARAMS, "null group or shards");
  bool any = false;
  for (int i = 0; i < g->n; ++i)
  {
    if (shard_batch[i] < 0 || (shard_batch[i] > 0 && !shard_images[i]))
      return set_error(SARA_HIP_INVALID_PARAMS,
                       "negative shard size or null shard pointer");
    any = any || shard_batch[i] > 0;
  }
  if (!any)
    return set_error(SARA_HIP_INVALID_PARAMS, "no frames in any shard");
  // a batch that was detected but never gathered is dropped here: its tickets
  // would otherwise stay pending for ever and block the third submit()
  for (int i = 0; i < g->n; ++i)
  {
    if (g->ticket[size_t(i)] >= 0)
      ticket_release(g->ctx[size_t(i)], g->ticket[size_t(i)]);
    g->ticket[size_t(i)] = kNoTicket;
  }
  const sara_hip_status st = on_every_device(g, [&](int i) {
    if (shard_batch[i] == 0)
    {
      g->ticket[size_t(i)] = kEmptyShard;  // fewer frames than devices
      return SARA_HIP_OK;
    }
    return sara_hip_sift_submit(g->ctx[size_t(i)], shard_images[i], frame_stride,
                                channels, shard_batch[i], width, height,
                                images_on_device, last_stage,
                                &g->ticket[size_t(i)]);
  });
  if (st != SARA_HIP_OK)
    for (int i = 0; i < g->n; ++i)
    {
      if (g->ticket[size_t(i)] >= 0)
        ticket_release(g->ctx[size_t(i)], g->ticket[size_t(i)]);
      g->ticket[size_t(i)] = kNoTicket;
    }
  return st;
}

sara_hip_status sara_hip_sift_group_gather(sara_hip_sift_group* g, int root,
                                           int with_descriptors,
                                           int* counts_per_device,
                                           const sara_oeregion** d_features,
                                           const float** d_descriptors,
                                           const int32_t** d_scale_octave,
                                           int* total)
{
  if (!g)
    return set_error(SARA_HIP_INVALID_PARAMS, "null group");
  // checked before any state changes: the batch can be gathered again
  if (root < 0 || root >= g->n)
    return set_error(SARA_HIP_INVALID_PARAMS, "root index out of range");
  // 1. every device waits for its batch; the counts are host values of this
  //    process, so no collective is needed to share them
  std::vector<Local> locals;
  std::vector<int> counts;
  sara_hip_status st = group_locals(g, root, with_descriptors, &locals, &counts);
  if (st != SARA_HIP_OK)
    return st;
  size_t sum = 0;
  for (int n : counts)
    sum += size_t(n);
  // 2. room on the root, before anyone posts a transfer
  st = gather_reserve(g->comm[size_t(root)], sum);
  if (st != SARA_HIP_OK)
  {
    const std::string msg = sara_hip_last_error();
    group_release(g, locals);
    return set_error(st, msg.c_str());
  }
  // 3. the transfers, one host thread per device
  st = on_every_device(g, [&](int i) {
    (void) hipSetDevice(g->devices[size_t(i)]);
    return gather_exchange(g->comm[size_t(i)], locals[size_t(i)], counts.data(),
                           root, with_descriptors);
  });
  const std::string msg = st != SARA_HIP_OK ? sara_hip_last_error() : "";
  group_release(g, locals);
  gather_outputs(g->comm[size_t(root)], counts.data(), root, with_descriptors,
                 counts_per_device, d_features, d_descriptors, d_scale_octave,
                 total);
  if (st != SARA_HIP_OK)
    return set_error(st, msg.c_str());
  for (const Local& l : locals)
    if (capacity_status(l) != SARA_HIP_OK)
      return SARA_HIP_CAPACITY_EXCEEDED;
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_sift_group_collect_host(
    sara_hip_sift_group* g, int with_descriptors, int* counts_per_device,
    const sara_oeregion** h_features, const float** h_descriptors,
    const int32_t** h_scale_octave, int* total)
{
  if (!g)
    return set_error(SARA_HIP_INVALID_PARAMS, "null group");
  std::vector<Local> locals;
  std::vector<int> counts;
  sara_hip_status st = group_locals(g, 0, with_descriptors, &locals, &counts);
  if (st != SARA_HIP_OK)
    return st;
  std::vector<size_t> offset(size_t(g->n) + 1, 0);
  for (int i = 0; i < g->n; ++i)
    offset[size_t(i) + 1] = offset[size_t(i)] + size_t(counts[size_t(i)]);
  const size_t sum = offset[size_t(g->n)];
  if (sum > g->h_cap)
  {
    if (g->h_feat)
      (void) hipHostFree(g->h_feat);
    if (g->h_desc)
      (void) hipHostFree(g->h_desc);
    if (g->h_so)
      (void) hipHostFree(g->h_so);
    g->h_feat = nullptr;
    g->h_desc = nullptr;
    g->h_so = nullptr;
    g->h_cap = 0;
    const size_t want = sum + sum / 4 + 1024;
    // portable: every device of the node may DMA into it
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&g->h_feat),
                                 sizeof(sara_oeregion) * want, hipHostMallocPortable);
    if (e == hipSuccess)
      e = hipHostMalloc(reinterpret_cast<void**>(&g->h_desc),
                        sizeof(float) * 128 * want, hipHostMallocPortable);
    if (e == hipSuccess)
      e = hipHostMalloc(reinterpret_cast<void**>(&g->h_so),
                        sizeof(int32_t) * 2 * want, hipHostMallocPortable);
    if (e != hipSuccess)
    {
      group_release(g, locals);
      return set_error(SARA_HIP_RUNTIME_ERROR,
                       (std::string("hipHostMalloc: ") + hipGetErrorString(e)).c_str());
    }
    g->h_cap = want;
  }
  st = on_every_device(g, [&](int i) -> sara_hip_status {
    const Local& l = locals[size_t(i)];
    if (l.mine <= 0)
      return SARA_HIP_OK;
    sara_hip_comm* c = g->comm[size_t(i)];
    const size_t n = size_t(l.mine), at = offset[size_t(i)];
    HIPC_TRY(hipSetDevice(c->device));
    HIPC_TRY(hipMemcpyAsync(g->h_feat + at, l.res.d_feat, n * sizeof(sara_oeregion),
                            hipMemcpyDeviceToHost, c->stream));
    HIPC_TRY(hipMemcpyAsync(g->h_so + 2 * at, l.res.d_so, n * 2 * sizeof(int32_t),
                            hipMemcpyDeviceToHost, c->stream));
    if (with_descriptors)
      HIPC_TRY(hipMemcpyAsync(g->h_desc + 128 * at, l.res.d_desc,
                              n * 128 * sizeof(float), hipMemcpyDeviceToHost,
                              c->stream));
    HIPC_TRY(hipStreamSynchronize(c->stream));
    return SARA_HIP_OK;
  });
  const std::string msg = st != SARA_HIP_OK ? sara_hip_last_error() : "";
  group_release(g, locals);
  if (counts_per_device)
    std::copy(counts.begin(), counts.end(), counts_per_device);
  if (total)
    *total = int(sum);
  if (h_features)
    *h_features = g->h_feat;
  if (h_descriptors)
    *h_descriptors = with_descriptors ? g->h_desc : nullptr;
  if (h_scale_octave)
    *h_scale_octave = g->h_so;
  if (st != SARA_HIP_OK)
    return set_error(st, msg.c_str());
  for (const Local& l : locals)
    if (capacity_status(l) != SARA_HIP_OK)
      return SARA_HIP_CAPACITY_EXCEEDED;
  return SARA_HIP_OK;
}

}  // extern "C"
