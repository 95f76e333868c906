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
// librccl is loaded on first use (dlopen) so that single-GPU users of the
// library do not depend on it.
#include "sift_kernels.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
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
    kNcclInt32 = 2,
    kNcclFloat32 = 7
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

#define RCCL_TRY(expr)                                                         \
  do                                                                           \
  {                                                                            \
    const int e_ = (expr);                                                     \
    if (e_ != 0)                                                               \
      return set_error(SARA_HIP_RCCL_ERROR,                                    \
                       (std::string(#expr) + ": " +                            \
                        rccl()->GetErrorString(e_))                            \
                           .c_str());                                          \
  } while (0)
#define HIPC_TRY(expr)                                                         \
  do                                                                           \
  {                                                                            \
    const hipError_t e_ = (expr);                                              \
    if (e_ != hipSuccess)                                                      \
      return set_error(SARA_HIP_RUNTIME_ERROR,                                 \
                       (std::string(#expr) + ": " + hipGetErrorString(e_))     \
                           .c_str());                                          \
  } while (0)

}  // namespace

struct sara_hip_comm
{
  sara_hip_sift* ctx = nullptr;
  ncclComm_t comm = nullptr;
  bool owns_comm = true;
  int nranks = 1, rank = 0, device = 0;
  hipStream_t stream = nullptr;  // the exchange runs beside the next batch
  int* d_counts = nullptr;       // [nranks] + 1 (this rank's count at the end)
  int* h_counts = nullptr;       // pinned, [nranks]
  // gather buffers on the root (grown on demand)
  sara_oeregion* d_feat = nullptr;
  float* d_desc = nullptr;
  int32_t* d_so = nullptr;
  size_t cap = 0;
};

namespace {

  sara_hip_status comm_finish_create(sara_hip_comm* c)
  {
    HIPC_TRY(hipSetDevice(c->device));
    HIPC_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPC_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_counts),
                       sizeof(int) * (size_t(c->nranks) + 1)));
    HIPC_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_counts),
                           sizeof(int) * size_t(c->nranks)));
    return SARA_HIP_OK;
  }

  //! The gatherv of one ticket; see the header.
  sara_hip_status comm_gather(sara_hip_comm* c, int ticket, int root,
                              int with_descriptors, int* counts_per_rank,
                              const sara_oeregion** d_features,
                              const float** d_descriptors,
                              const int32_t** d_scale_octave, int* total)
  {
    if (!c)
      return set_error(SARA_HIP_INVALID_PARAMS, "null communicator");
    if (root < 0 || root >= c->nranks)
      return set_error(SARA_HIP_INVALID_PARAMS, "root rank out of range");
    Rccl* r = rccl();
    TicketResults res;
    sara_hip_status st = ticket_results(c->ctx, ticket, &res);
    if (st != SARA_HIP_OK)
      return st;
    HIPC_TRY(hipSetDevice(c->device));
    // 1. every rank's keypoint count
    const int mine = res.total;
    HIPC_TRY(hipMemcpyAsync(c->d_counts + c->nranks, &mine, sizeof(int),
                            hipMemcpyHostToDevice, c->stream));
    RCCL_TRY(r->AllGather(c->d_counts + c->nranks, c->d_counts, 1, kNcclInt32,
                          c->comm, c->stream));
    HIPC_TRY(hipMemcpyAsync(c->h_counts, c->d_counts, sizeof(int) * c->nranks,
                            hipMemcpyDeviceToHost, c->stream));
    HIPC_TRY(hipStreamSynchronize(c->stream));
    std::vector<size_t> offset(size_t(c->nranks) + 1, 0);
    for (int k = 0; k < c->nranks; ++k)
      offset[size_t(k) + 1] = offset[size_t(k)] + size_t(c->h_counts[k]);
    const size_t sum = offset[size_t(c->nranks)];
    // 2. room on the root
    if (c->rank == root && sum > c->cap)
    {
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
    }
    // 3. one group of point-to-point transfers at the global offsets
    RCCL_TRY(r->GroupStart());
    if (c->rank == root)
    {
      for (int k = 0; k < c->nranks; ++k)
      {
        const size_t n = size_t(c->h_counts[k]);
        if (k == root || n == 0)
          continue;
        RCCL_TRY(r->Recv(c->d_feat + offset[size_t(k)], n * sizeof(sara_oeregion),
                         kNcclInt8, k, c->comm, c->stream));
        RCCL_TRY(r->Recv(c->d_so + 2 * offset[size_t(k)], n * 2, kNcclInt32, k,
                         c->comm, c->stream));
        if (with_descriptors)
          RCCL_TRY(r->Recv(c->d_desc + 128 * offset[size_t(k)], n * 128,
                           kNcclFloat32, k, c->comm, c->stream));
      }
    }
    else if (mine > 0)
    {
      const size_t n = size_t(mine);
      RCCL_TRY(r->Send(res.d_feat, n * sizeof(sara_oeregion), kNcclInt8, root,
                       c->comm, c->stream));
      RCCL_TRY(r->Send(res.d_so, n * 2, kNcclInt32, root, c->comm, c->stream));
      if (with_descriptors)
        RCCL_TRY(r->Send(res.d_desc, n * 128, kNcclFloat32, root, c->comm,
                         c->stream));
    }
    RCCL_TRY(r->GroupEnd());
    if (c->rank == root && mine > 0)
    {
      const size_t n = size_t(mine), at = offset[size_t(root)];
      HIPC_TRY(hipMemcpyAsync(c->d_feat + at, res.d_feat, n * sizeof(sara_oeregion),
                              hipMemcpyDeviceToDevice, c->stream));
      HIPC_TRY(hipMemcpyAsync(c->d_so + 2 * at, res.d_so, n * 2 * sizeof(int32_t),
                              hipMemcpyDeviceToDevice, c->stream));
      if (with_descriptors)
        HIPC_TRY(hipMemcpyAsync(c->d_desc + 128 * at, res.d_desc,
                                n * 128 * sizeof(float), hipMemcpyDeviceToDevice,
                                c->stream));
    }
    // the ticket's result slot may be overwritten once the transfers are done
    HIPC_TRY(hipStreamSynchronize(c->stream));
    ticket_release(c->ctx, ticket);
    if (counts_per_rank)
      std::copy(c->h_counts, c->h_counts + c->nranks, counts_per_rank);
    if (total)
      *total = int(sum);
    const bool here = c->rank == root;
    if (d_features)
      *d_features = here ? c->d_feat : nullptr;
    if (d_descriptors)
      *d_descriptors = here && with_descriptors ? c->d_desc : nullptr;
    if (d_scale_octave)
      *d_scale_octave = here ? c->d_so : nullptr;
    if (res.capacity_exceeded)
      return set_error(SARA_HIP_CAPACITY_EXCEEDED,
                       "a frame produced more extrema / keypoints than "
                       "max_keypoints: the lists are truncated");
    return SARA_HIP_OK;
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

sara_hip_status sara_hip_comm_unique_id(unsigned char* id)
{
  if (!id)
    return set_error(SARA_HIP_INVALID_PARAMS, "null id");
  const sara_hip_status st = rccl_ready();
  if (st != SARA_HIP_OK)
    return st;
  ncclUniqueId u;
  RCCL_TRY(rccl()->GetUniqueId(&u));
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
  const sara_hip_status st = rccl_ready();
  if (st != SARA_HIP_OK)
    return st;
  HIPC_TRY(hipSetDevice(device));
  ncclUniqueId u;
  std::memcpy(u.internal, id, SARA_HIP_COMM_ID_BYTES);
  auto* c = new sara_hip_comm;
  c->ctx = ctx;
  c->nranks = nranks;
  c->rank = rank;
  c->device = device;
  const int e = rccl()->CommInitRank(&c->comm, nranks, u, rank);
  if (e != 0)
  {
    delete c;
    return set_error(SARA_HIP_RCCL_ERROR,
                     (std::string("ncclCommInitRank: ") +
                      rccl()->GetErrorString(e))
                         .c_str());
  }
  const sara_hip_status fs = comm_finish_create(c);
  if (fs != SARA_HIP_OK)
  {
    sara_hip_comm_destroy(c);
    return fs;
  }
  *out = c;
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
    (void) hipStreamDestroy(c->stream);
  }
  if (c->comm && c->owns_comm && rccl()->CommDestroy)
    (void) rccl()->CommDestroy(c->comm);
  if (c->d_counts)
    (void) hipFree(c->d_counts);
  if (c->h_counts)
    (void) hipHostFree(c->h_counts);
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
  std::vector<int> ticket;
  std::vector<std::string> errors;
};

namespace {
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
  sara_hip_status st = rccl_ready();
  if (st != SARA_HIP_OK)
    return st;
  auto* g = new sara_hip_sift_group;
  g->n = n_dev;
  for (int i = 0; i < n_dev; ++i)
    g->devices.push_back(devices ? devices[i] : i);
  for (int d : g->devices)
    if (d < 0 || d >= visible)
    {
      delete g;
      return set_error(SARA_HIP_INVALID_PARAMS, "device ordinal out of range");
    }
  g->ctx.assign(size_t(n_dev), nullptr);
  g->comm.assign(size_t(n_dev), nullptr);
  g->ticket.assign(size_t(n_dev), -1);
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
  const int e = rccl()->CommInitAll(comms.data(), n_dev, g->devices.data());
  if (e != 0)
  {
    sara_hip_sift_group_destroy(g);
    return set_error(SARA_HIP_RCCL_ERROR, (std::string("ncclCommInitAll: ") +
                                           rccl()->GetErrorString(e))
                                              .c_str());
  }
  for (int i = 0; i < n_dev; ++i)
  {
    auto* c = new sara_hip_comm;
    c->ctx = g->ctx[size_t(i)];
    c->comm = comms[size_t(i)];
    c->nranks = n_dev;
    c->rank = i;
    c->device = g->devices[size_t(i)];
    g->comm[size_t(i)] = c;
    st = comm_finish_create(c);
    if (st != SARA_HIP_OK)
    {
      sara_hip_sift_group_destroy(g);
      return st;
    }
  }
  *out = g;
  return SARA_HIP_OK;
}

int sara_hip_sift_group_size(const sara_hip_sift_group* g) { return g ? g->n : 0; }

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
  for (int i = 0; i < g->n; ++i)
    if (shard_batch[i] < 1 || !shard_images[i])
      return set_error(SARA_HIP_INVALID_PARAMS,
                       "every device needs at least one frame");
  return on_every_device(g, [&](int i) {
    return sara_hip_sift_submit(g->ctx[size_t(i)], shard_images[i], frame_stride,
                                channels, shard_batch[i], width, height,
                                images_on_device, last_stage,
                                &g->ticket[size_t(i)]);
  });
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
  if (root < 0 || root >= g->n)
    return set_error(SARA_HIP_INVALID_PARAMS, "root index out of range");
  for (int i = 0; i < g->n; ++i)
    if (g->ticket[size_t(i)] < 0)
      return set_error(SARA_HIP_NOT_READY, "no group_detect() to gather");
  const sara_hip_status st = on_every_device(g, [&](int i) {
    const bool here = i == root;
    return comm_gather(g->comm[size_t(i)], g->ticket[size_t(i)], root,
                       with_descriptors, here ? counts_per_device : nullptr,
                       here ? d_features : nullptr, here ? d_descriptors : nullptr,
                       here ? d_scale_octave : nullptr, here ? total : nullptr);
  });
  for (int& t : g->ticket)
    t = -1;
  return st;
}

}  // extern "C"
