// Descriptor matching on gfx950.
//
// Reference: AnnMatcher::compute_matches / append_nearest_neighbors,
// FeatureMatching/AnnMatcher.cpp:59-268 (called by match(),
// SfM/Helpers/KeypointMatching.cpp:19-25).  The reference asks FLANN kd-trees
// for approximate neighbours; here every query sees every candidate - the
// answer FLANN converges to - with FLANN's own squared-L2 arithmetic
// (third-party/flann/src/cpp/flann/algorithms/dist.h:150-178: groups of four,
// float accumulator, no FMA), so scores are bit-identical to an exhaustive CPU
// search in that arithmetic (checked by tests/test_gpu_matching.py).
//
// What a query needs from the search (sift_match.cpp has the decision logic):
//   * its three nearest neighbours ordered by (distance, index) = FLANN's
//     knnSearch(3), AnnMatcher.cpp:123 (rank 0 is the query itself when a key
//     set is matched against itself);
//   * when the squared ratio threshold exceeds 1 (the reference's default,
//     1.2^2): every neighbour with distance < d_top1 * threshold = FLANN's
//     radiusSearch, :133-138.
//
// Exhaustive kernels (this file's first half):
// nn3_kernel: one wave = 64 queries (one per lane) x one chunk of candidates.
// A tile of 32 candidates is staged in LDS; per group of four dimensions the
// lane loads its own four query values (its row stays in L1/L2 across tiles;
// staging the 64 query rows in LDS as well would cap the CU at 3 waves) and
// reads every candidate's four values as an LDS broadcast; 32 per-candidate
// accumulators live in registers so that each distance is summed group by
// group in FLANN's order.  The per-chunk top-3 lists are merged in chunk
// order by merge3_kernel (ties keep the lower index).  radius_kernel repeats
// the distances and appends the neighbours inside each query's radius.
// N x M x 128 subtract/multiply/add at VALU rate: exact float32 semantics.
#include "sift_kernels.hpp"

#include <algorithm>
#include <cfloat>

namespace sara_hip {

  constexpr int kMatchTile = 32;   // candidates per LDS tile

  __device__ inline void top3_insert(float d, int j, float& d0, float& d1,
                                     float& d2, int& i0, int& i1, int& i2)
  {
    // strict comparisons: among equal distances the one seen first (the lower
    // index: candidates and chunks are visited in index order) stays ahead
    if (d < d0)
    {
      d2 = d1;
      i2 = i1;
      d1 = d0;
      i1 = i0;
      d0 = d;
      i0 = j;
    }
    else if (d < d1)
    {
      d2 = d1;
      i2 = i1;
      d1 = d;
      i1 = j;
    }
    else if (d < d2)
    {
      d2 = d;
      i2 = j;
    }
  }

  //! Stages candidates [c0, c0 + 32) in LDS and accumulates this lane's query
  //! against each of them in FLANN's order.
  __device__ inline void tile_distances(const float* __restrict__ t, int dim,
                                        int dim4, int c0, int c_end,
                                        const float* __restrict__ myq, bool vec4,
                                        float* s_t, float (&acc)[kMatchTile])
  {
    const int lane = threadIdx.x;
    const int groups = dim / 4, tail = dim - 4 * groups;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (int idx = lane; idx < kMatchTile * dim4; idx += 64)
    {
      const int r = idx / dim4, k = idx - r * dim4;
      float v = 0.f;
      if (c0 + r < c_end && k < dim)
        v = t[size_t(c0 + r) * dim + k];
      s_t[idx] = v;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < kMatchTile; ++c)
      acc[c] = 0.f;
    for (int g = 0; g < groups; ++g)
    {
      float4 a;
      if (vec4)
        a = *reinterpret_cast<const float4*>(myq + 4 * g);
      else
        a = make_float4(myq[4 * g], myq[4 * g + 1], myq[4 * g + 2], myq[4 * g + 3]);
#pragma unroll
      for (int c = 0; c < kMatchTile; ++c)
      {
        const float4 b = *reinterpret_cast<const float4*>(s_t + c * dim4 + 4 * g);
        const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
        acc[c] += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
      }
    }
    for (int k = 4 * groups; k < 4 * groups + tail; ++k)
    {
      const float a = myq[k];
#pragma unroll
      for (int c = 0; c < kMatchTile; ++c)
      {
        const float d0 = a - s_t[c * dim4 + k];
        acc[c] += d0 * d0;
      }
    }
  }

  //! part_d / part_i: [3][nchunks][nq].
  __global__ __launch_bounds__(64) void nn3_kernel(
      const float* __restrict__ q, int nq, const float* __restrict__ t, int nt,
      int dim, int chunk, float* __restrict__ part_d, int* __restrict__ part_i)
  {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int dim4 = (dim + 3) & ~3;
    const int lane = threadIdx.x;
    const int q0 = blockIdx.x * 64;
    const int c_begin = blockIdx.y * chunk;
    const int c_end = min(nt, c_begin + chunk);
    // this lane's query row (lanes past the end re-read the last row)
    const float* myq = q + size_t(min(q0 + lane, nq - 1)) * dim;
    const bool vec4 = (dim % 4 == 0) && (reinterpret_cast<uintptr_t>(q) % 16 == 0);
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    int i0 = -1, i1 = -1, i2 = -1;
    for (int c0 = c_begin; c0 < c_end; c0 += kMatchTile)
    {
      float acc[kMatchTile];
      tile_distances(t, dim, dim4, c0, c_end, myq, vec4, smem, acc);
#pragma unroll
      for (int c = 0; c < kMatchTile; ++c)
        if (c0 + c < c_end)
          top3_insert(acc[c], c0 + c, b0, b1, b2, i0, i1, i2);
    }
    if (q0 + lane < nq)
    {
      const size_t plane = size_t(gridDim.y) * nq;
      const size_t o = size_t(blockIdx.y) * nq + q0 + lane;
      part_d[o] = b0;
      part_d[plane + o] = b1;
      part_d[2 * plane + o] = b2;
      part_i[o] = i0;
      part_i[plane + o] = i1;
      part_i[2 * plane + o] = i2;
    }
  }

  //! top_d / top_i: [3][nq], the knnSearch(3) answer of every query.
  __global__ void merge3_kernel(const float* __restrict__ part_d,
                                const int* __restrict__ part_i, int nq,
                                int nchunks, float* __restrict__ top_d,
                                int* __restrict__ top_i)
  {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq)
      return;
    const size_t plane = size_t(nchunks) * nq;
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    int i0 = -1, i1 = -1, i2 = -1;
    for (int c = 0; c < nchunks; ++c)
      for (int k = 0; k < 3; ++k)
      {
        const size_t o = k * plane + size_t(c) * nq + i;
        const int j = part_i[o];
        if (j >= 0)
          top3_insert(part_d[o], j, b0, b1, b2, i0, i1, i2);
      }
    top_d[i] = b0;
    top_d[nq + i] = b1;
    top_d[2 * size_t(nq) + i] = b2;
    top_i[i] = i0;
    top_i[nq + i] = i1;
    top_i[2 * size_t(nq) + i] = i2;
  }

  //! AnnMatcher.cpp:126-147 for squared ratio <= 1, two key sets: only the best
  //! neighbour, score d0 / d1; appends {x, y, score, rank = 1, direction}.
  __global__ void ratio_filter_kernel(const float* __restrict__ top_d,
                                      const int* __restrict__ top_i, int nq,
                                      float squared_ratio_thres, int direction,
                                      sara_match* __restrict__ out, int capacity,
                                      int* __restrict__ count)
  {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq)
      return;
    const float d0 = top_d[i], d1 = top_d[nq + i];
    const int i0 = top_i[i];
    const float score = d1 > 0.f ? d0 / d1 : 0.f;
    if (i0 < 0 || score > squared_ratio_thres)
      return;
    const int slot = atomicAdd(count, 1);
    if (slot >= capacity)
      return;
    sara_match m;
    m.x_index = direction == 0 ? i : i0;
    m.y_index = direction == 0 ? i0 : i;
    m.score = score;
    m.rank = 1;
    m.direction = direction;
    out[slot] = m;
  }

  //! radiusSearch (AnnMatcher.cpp:133-138): every candidate with distance <
  //! top_d[top1][query] * squared_ratio_thres is appended as (query, index,
  //! distance); *count keeps counting past `capacity` so that the caller can
  //! retry with room.
  __global__ __launch_bounds__(64) void radius_kernel(
      const float* __restrict__ q, int nq, const float* __restrict__ t, int nt,
      int dim, int chunk, const float* __restrict__ top_d, int top1,
      float squared_ratio_thres, MatchNeighbour* __restrict__ out, int capacity,
      int* __restrict__ count)
  {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int dim4 = (dim + 3) & ~3;
    const int lane = threadIdx.x;
    const int q0 = blockIdx.x * 64;
    const int c_begin = blockIdx.y * chunk;
    const int c_end = min(nt, c_begin + chunk);
    const int me = min(q0 + lane, nq - 1);
    const float* myq = q + size_t(me) * dim;
    const bool vec4 = (dim % 4 == 0) && (reinterpret_cast<uintptr_t>(q) % 16 == 0);
    // a query without a ranked neighbour (FLT_MAX) emits nothing: the product
    // overflows to +inf only for thresholds above 1, and the caller skips
    // those queries anyway
    const float radius =
        q0 + lane < nq ? top_d[size_t(top1) * nq + me] * squared_ratio_thres : 0.f;
    for (int c0 = c_begin; c0 < c_end; c0 += kMatchTile)
    {
      float acc[kMatchTile];
      tile_distances(t, dim, dim4, c0, c_end, myq, vec4, smem, acc);
#pragma unroll
      for (int c = 0; c < kMatchTile; ++c)
        if (c0 + c < c_end && acc[c] < radius)
        {
          const int slot = atomicAdd(count, 1);
          if (slot < capacity)
            out[slot] = MatchNeighbour{me, c0 + c, acc[c]};
        }
    }
  }

  //! A pointer every lane holds alike (it came out of the pair table at
  //! blockIdx), pinned into scalar registers: the compiler then combines the
  //! wave's atomicAdd(count, 1) into one atomic per wave, as it does for a
  //! kernel argument - 4 140 same-address atomics took 45 us instead of 5.
  template <typename T>
  __device__ __forceinline__ T* wave_uniform(T* p)
  {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(v));
    const unsigned hi = __builtin_amdgcn_readfirstlane(unsigned(v >> 32));
    using G = __attribute__((address_space(1))) T*;
    return (T*) (G) ((static_cast<unsigned long long>(hi) << 32) | lo);
  }

  // ---- the whole tail of compute_matches on the device (squared ratio <= 1) ---
  //! AnnMatcher.cpp:126-147 for both directions plus the sort by (x, y, score)
  //! and the std::unique of :239-254: thread q < n1 is key q of the first set
  //! (direction 0), thread n1 + q key q of the second (direction 1).  A pair
  //! (x, y) found from both sides is kept once, with the lower score
  //! (direction 0 on equal scores - the order the restated sort leaves).
  //! have0 / have1: the direction has >= 2 candidates (a single candidate
  //! scores 1 and never passes a squared ratio <= 1, :87-101).
  template <bool BATCH>
  __global__ void mutual_filter_kernel(const float* __restrict__ top_d0_,
                                       const int* __restrict__ top_i0_, int n1_,
                                       const float* __restrict__ top_d1_,
                                       const int* __restrict__ top_i1_, int n2_,
                                       int have0, int have1,
                                       float squared_ratio_thres,
                                       sara_match* __restrict__ out_,
                                       int* __restrict__ count_,
                                       const MatchBatchPair* __restrict__ batch)
  {
    // pair blockIdx.y of a batch (both directions have >= 2 keys), or the
    // kernel's own arguments
    const MatchBatchPair* b = BATCH ? batch + blockIdx.y : nullptr;
    const int n1 = BATCH ? b->n1 : n1_;
    const int n2 = BATCH ? b->n2 : n2_;
    const float* __restrict__ top_d0 = BATCH ? b->top_d : top_d0_;
    const int* __restrict__ top_i0 = BATCH ? b->top_i : top_i0_;
    const float* __restrict__ top_d1 = BATCH ? b->top_d + 3 * size_t(n1) : top_d1_;
    const int* __restrict__ top_i1 = BATCH ? b->top_i + 3 * size_t(n1) : top_i1_;
    sara_match* __restrict__ out = BATCH ? wave_uniform(b->tmp) : out_;
    int* __restrict__ count = BATCH ? wave_uniform(b->header) : count_;
    if (BATCH)
      have0 = have1 = 1;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n1 + n2)
      return;
    auto best = [&](int dir, int q, int& idx, float& score) -> bool {
      const float* td = dir == 0 ? top_d0 : top_d1;
      const int* ti = dir == 0 ? top_i0 : top_i1;
      const int nq = dir == 0 ? n1 : n2;
      if (!(dir == 0 ? have0 : have1))
        return false;
      idx = ti[q];
      const float d0 = td[q], d1 = td[nq + q];
      score = d1 > 0.f ? d0 / d1 : 0.f;
      return idx >= 0 && !(score > squared_ratio_thres);
    };
    const int dir = t < n1 ? 0 : 1;
    const int q = dir == 0 ? t : t - n1;
    int other = -1, back = -1;
    float score = 0.f, score_back = 0.f;
    if (!best(dir, q, other, score))
      return;
    // the same pair from the other side?
    const bool twin = best(1 - dir, other, back, score_back) && back == q;
    if (twin && (dir == 0 ? score_back < score : score_back <= score))
      return;
    sara_match m;
    m.x_index = dir == 0 ? q : other;
    m.y_index = dir == 0 ? other : q;
    m.score = score;
    m.rank = 1;
    m.direction = dir;
    out[atomicAdd(count, 1)] = m;
  }

  //! The final std::sort by score (AnnMatcher.cpp:256-258; equal scores by
  //! (x, y), one of the orders it may leave) as a rank sort: every match counts
  //! the matches that precede it - the keys are distinct - and moves to that
  //! position.  2-D grid: block (bx, by) counts, for the 256 matches of tile
  //! bx, their predecessors inside tile by (a 1-D version - each thread against
  //! all n - kept 17 workgroups busy for 0.28 ms).
  __device__ inline void match_key(const sara_match& m, unsigned long long& k1,
                                   int& k2)
  {
    // scores are >= 0: their bit patterns order like unsigned integers
    k1 = (static_cast<unsigned long long>(__float_as_uint(m.score)) << 32) |
         static_cast<unsigned>(m.x_index);
    k2 = m.y_index;
  }

  //! keys per tile of predecessors: short tiles = many workgroups with a short
  //! loop each (a single 4.3 k x 4.3 k search, 4 140 matches: 256-key tiles
  //! 21.6 us - 289 workgroups of 256 dependent LDS reads - against 64-key tiles)
  constexpr int kRankKeys = 64;
  template <bool BATCH>
  __global__ __launch_bounds__(256) void rank_count_kernel(
      const sara_match* __restrict__ in_, const int* __restrict__ count_,
      int* __restrict__ rank_, const MatchBatchPair* __restrict__ batch)
  {
    __shared__ unsigned long long s_k1[kRankKeys];
    __shared__ int s_k2[kRankKeys];
    const MatchBatchPair* b = BATCH ? batch + blockIdx.z : nullptr;  // pair blockIdx.z
    const sara_match* __restrict__ in = BATCH ? wave_uniform(b->tmp) : in_;
    const int* __restrict__ count = BATCH ? wave_uniform(b->header) : count_;
    int* __restrict__ rank = BATCH ? wave_uniform(b->rank) : rank_;
    const int n = *count;
    // (the grid may be smaller than the list: tiles are walked with its stride)
    for (int tx = blockIdx.x; tx * 256 < n; tx += gridDim.x)
    {
      const int e = tx * 256 + threadIdx.x;
      unsigned long long k1 = 0ull;
      int k2 = 0;
      if (e < n)
        match_key(in[e], k1, k2);
      int before = 0;
      for (int ty = blockIdx.y; ty * kRankKeys < n; ty += gridDim.y)
      {
        const int base = ty * kRankKeys;
        __syncthreads();  // the previous tile has been consumed
        if (threadIdx.x < kRankKeys)
        {
          unsigned long long o1 = ~0ull;
          int o2 = 0x7fffffff;
          if (base + int(threadIdx.x) < n)
            match_key(in[base + threadIdx.x], o1, o2);
          s_k1[threadIdx.x] = o1;
          s_k2[threadIdx.x] = o2;
        }
        __syncthreads();
        // (keys behind the list are the largest key: they precede nothing)
#pragma unroll 8
        for (int k = 0; k < kRankKeys; ++k)
        {
          const unsigned long long p1 = s_k1[k];
          before += (p1 < k1 || (p1 == k1 && s_k2[k] < k2)) ? 1 : 0;
        }
      }
      if (e < n && before)
        atomicAdd(rank + e, before);
    }
  }

  template <bool BATCH>
  __global__ void rank_scatter_kernel(const sara_match* __restrict__ in_,
                                      const int* __restrict__ count_,
                                      const int* __restrict__ rank_,
                                      sara_match* __restrict__ out_,
                                      const MatchBatchPair* __restrict__ batch)
  {
    const MatchBatchPair* b = BATCH ? batch + blockIdx.y : nullptr;  // pair blockIdx.y
    const sara_match* __restrict__ in = BATCH ? wave_uniform(b->tmp) : in_;
    const int* __restrict__ count = BATCH ? wave_uniform(b->header) : count_;
    const int* __restrict__ rank = BATCH ? wave_uniform(b->rank) : rank_;
    sara_match* __restrict__ out = BATCH ? wave_uniform(b->out) : out_;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < *count;
         e += gridDim.x * blockDim.x)
      out[rank[e]] = in[e];
  }

  __global__ __launch_bounds__(256) void zero_ranges_kernel(ZeroRangesArg r)
  {
    int* p = r.p[blockIdx.y];
    const unsigned n = r.n[blockIdx.y];
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
      p[i] = 0;
  }

  bool launch_zero_ranges(const ZeroRanges& r, hipStream_t stream)
  {
    if (r.overflow)
    {
      // a range was not registered: whoever relies on it would read stale
      // counters - poison the stream's error state instead of running on
      (void) hipMemsetAsync(nullptr, 0, 1, stream);  // -> hipErrorInvalidValue
      return false;
    }
    for (int first = 0; first < r.count; first += kZeroRangesPerLaunch)
    {
      const int m = std::min(kZeroRangesPerLaunch, r.count - first);
      ZeroRangesArg a;
      unsigned longest = 0;
      for (int k = 0; k < m; ++k)
      {
        a.p[k] = r.p[first + k];
        a.n[k] = r.n[first + k];
        longest = std::max(longest, a.n[k]);
      }
      const unsigned blocks = std::min(512u, (longest + 1023) / 1024);
      hipLaunchKernelGGL(zero_ranges_kernel, dim3(std::max(blocks, 1u), m), dim3(256), 0,
                         stream, a);
    }
    return true;
  }

  void launch_finish_matches(const float* top_d0, const int* top_i0, int n1,
                             const float* top_d1, const int* top_i1, int n2,
                             int have0, int have1, float squared_ratio_thres,
                             sara_match* scratch, int* rank_scratch, int* count,
                             sara_match* out, hipStream_t stream, bool scratch_cleared)
  {
    const int n = n1 + n2;
    const int tiles = (n + 255) / 256;
    if (!scratch_cleared)
    {
      ZeroRanges z;
      z.add(rank_scratch, size_t(n));
      launch_zero_ranges(z, stream);
    }
    hipLaunchKernelGGL(mutual_filter_kernel<false>, dim3(tiles), dim3(256), 0, stream,
                       top_d0, top_i0, n1, top_d1, top_i1, n2, have0, have1,
                       squared_ratio_thres, scratch, count, nullptr);
    hipLaunchKernelGGL(rank_count_kernel<false>,
                       dim3(tiles, (n + kRankKeys - 1) / kRankKeys), dim3(256), 0, stream,
                       scratch, count, rank_scratch, nullptr);
    hipLaunchKernelGGL(rank_scatter_kernel<false>, dim3(tiles), dim3(256), 0, stream,
                       scratch, count, rank_scratch, out, nullptr);
  }

  //! The sorted lists of a batch back to back in `dense`, pair p's at
  //! [heads[p], heads[p + 1]) - what travels to the host is the matches that
  //! exist, not every pair's capacity (n1 + n2 records each).
  __global__ __launch_bounds__(256) void compact_lists_kernel(
      const MatchBatchPair* __restrict__ batch, int n_pairs,
      sara_match* __restrict__ dense, int* __restrict__ heads)
  {
    const int pair = blockIdx.y;
    int offset = 0;
    for (int q = 0; q < pair; ++q)
      offset += min(batch[q].header[0], batch[q].n1 + batch[q].n2);
    const MatchBatchPair& b = batch[pair];
    const int count = min(b.header[0], b.n1 + b.n2);
    const sara_match* __restrict__ in = wave_uniform(b.out);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < count;
         e += gridDim.x * blockDim.x)
      dense[offset + e] = in[e];
    if (blockIdx.x == 0 && threadIdx.x == 0)
    {
      heads[pair] = offset;
      if (pair == n_pairs - 1)
        heads[n_pairs] = offset + count;
    }
  }

  void launch_compact_lists(const MatchBatchPair* table, int n_pairs, int n_max,
                            sara_match* dense, int* heads, hipStream_t stream)
  {
    const int tiles = std::min((n_max + 255) / 256, 16);
    hipLaunchKernelGGL(compact_lists_kernel, dim3(tiles, n_pairs), dim3(256), 0, stream,
                       table, n_pairs, dense, heads);
  }

  void launch_finish_matches_batch(const MatchBatchPair* table, int n_pairs,
                                   int n_max, float squared_ratio_thres,
                                   hipStream_t stream)
  {
    // n_max = largest n1 + n2 of the batch; a pair's list is at most that long.
    // The rank sort walks its tiles with the grid's stride: a grid of 16 x 16
    // tiles per pair covers lists of any length.
    const int tiles = (n_max + 255) / 256;
    const int rt = std::min(tiles, 16);
    hipLaunchKernelGGL(mutual_filter_kernel<true>, dim3(tiles, n_pairs), dim3(256), 0, stream,
                       nullptr, nullptr, 0, nullptr, nullptr, 0, 1, 1,
                       squared_ratio_thres, nullptr, nullptr, table);
    hipLaunchKernelGGL(rank_count_kernel<true>,
                       dim3(rt, std::min((n_max + kRankKeys - 1) / kRankKeys, 64), n_pairs),
                       dim3(256), 0, stream, nullptr, nullptr, nullptr, table);
    hipLaunchKernelGGL(rank_scatter_kernel<true>, dim3(rt, n_pairs), dim3(256), 0, stream,
                       nullptr, nullptr, nullptr, nullptr, table);
  }

  // ------------------------------------------------------------------------ //
  // compute_matches' tail for squared ratios > 1 (the reference's DEFAULT,
  // sift_ratio_thres = 1.2f) on the device.  Input: the radius members of both
  // directions as unordered (query, index, distance) triples - what rerank /
  // fallback / radius_kernel append - and the knnSearch(3) tables.  Per member
  // (AnnMatcher.cpp:141-170): rank = its position in the query's members
  // ordered by (distance, index); score = d_top1 / d_next for rank 0, distance
  // / d_top1 behind it (0 when d_top1 == 0); the reference's loop stops at the
  // first score above the threshold, and scores do not decrease along a run, so
  // a member is emitted iff its own score passes.  Then :239-258: of a pair
  // (x, y) found from both sides the lower score survives (direction 0 on a
  // tie), and the list is ordered by (score, x, y).  A few thousand members per
  // direction: ranks and twins are found by comparing every member with every
  // other one, tile against tile.
  // ------------------------------------------------------------------------ //
  //! grid (tiles, tiles, 4): z = 2 * dir + role.  role 0: members of direction
  //! `dir` count their predecessors in the same run; role 1: they look for
  //! their twin (query and index swapped) among the other direction's members.
  __global__ __launch_bounds__(256) void radius_pair_kernel(
      const MatchNeighbour* __restrict__ m0, const int* __restrict__ c0, int cap0,
      const MatchNeighbour* __restrict__ m1, const int* __restrict__ c1, int cap1,
      int* __restrict__ rank0, int* __restrict__ rank1, int* __restrict__ twin0,
      int* __restrict__ twin1)
  {
    __shared__ int s_q[256], s_i[256];
    __shared__ float s_d[256];
    const int dir = blockIdx.z >> 1, role = blockIdx.z & 1;
    const MatchNeighbour* mine = dir == 0 ? m0 : m1;
    const int n_mine = min(dir == 0 ? *c0 : *c1, dir == 0 ? cap0 : cap1);
    const bool same = role == 0;
    const MatchNeighbour* other = (dir == 0) == same ? m0 : m1;
    const int n_other = (dir == 0) == same ? min(*c0, cap0) : min(*c1, cap1);
    // the grid is sized for typical lists; longer ones are walked with its stride
    for (int tx = blockIdx.x; tx * 256 < n_mine; tx += gridDim.x)
      for (int ty = blockIdx.y; ty * 256 < n_other; ty += gridDim.y)
      {
        const int e = tx * 256 + threadIdx.x;
        const int base = ty * 256;
        const int f = base + threadIdx.x;
        MatchNeighbour o{-1, -1, 0.f};
        if (f < n_other)
          o = other[f];
        __syncthreads();  // the previous tile has been consumed
        s_q[threadIdx.x] = o.query;
        s_i[threadIdx.x] = o.index;
        s_d[threadIdx.x] = o.distance;
        __syncthreads();
        if (e >= n_mine)
          continue;
        const MatchNeighbour me = mine[e];
        const int m = min(256, n_other - base);
        if (same)
        {
          int before = 0;
#pragma unroll 8
          for (int k = 0; k < m; ++k)
            before += (s_q[k] == me.query &&
                       (s_d[k] < me.distance ||
                        (s_d[k] == me.distance && s_i[k] < me.index)))
                          ? 1
                          : 0;
          if (before)
            atomicAdd((dir == 0 ? rank0 : rank1) + e, before);
        }
        else
        {
          int twin = -1;
#pragma unroll 8
          for (int k = 0; k < m; ++k)
            if (s_q[k] == me.index && s_i[k] == me.query)
              twin = base + k;
          if (twin >= 0)
            (dir == 0 ? twin0 : twin1)[e] = twin + 1;  // 0 = none (zero-filled)
        }
      }
  }

  //! score of every member (-1: not emitted) from its rank.
  __global__ void radius_score_kernel(const MatchNeighbour* __restrict__ m0,
                                      const int* __restrict__ c0, int cap0,
                                      const MatchNeighbour* __restrict__ m1,
                                      const int* __restrict__ c1, int cap1,
                                      const float* __restrict__ top_d0, int n1,
                                      const float* __restrict__ top_d1, int n2,
                                      float squared_ratio_thres,
                                      const int* __restrict__ rank0,
                                      const int* __restrict__ rank1,
                                      float* __restrict__ score0,
                                      float* __restrict__ score1)
  {
    const int dir = blockIdx.y;
    const int n = min(dir == 0 ? *c0 : *c1, dir == 0 ? cap0 : cap1);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n;
         e += gridDim.x * blockDim.x)
    {
    const MatchNeighbour me = (dir == 0 ? m0 : m1)[e];
    const float* td = dir == 0 ? top_d0 : top_d1;
    const int nq = dir == 0 ? n1 : n2;
    const int rank = (dir == 0 ? rank0 : rank1)[e];
    const float d_top1 = td[me.query], d_next = td[nq + me.query];
    float score = 0.f;
    if (rank == 0)
      score = d_next > 0.f ? d_top1 / d_next : 0.f;
    else if (d_top1)
      score = me.distance / d_top1;
    (dir == 0 ? score0 : score1)[e] = score > squared_ratio_thres ? -1.f : score;
    }
  }

  //! Appends the surviving members as matches (any order; *count counts).
  __global__ void radius_emit_kernel(const MatchNeighbour* __restrict__ m0,
                                     const int* __restrict__ c0, int cap0,
                                     const MatchNeighbour* __restrict__ m1,
                                     const int* __restrict__ c1, int cap1,
                                     const int* __restrict__ rank0,
                                     const int* __restrict__ rank1,
                                     const int* __restrict__ twin0,
                                     const int* __restrict__ twin1,
                                     const float* __restrict__ score0,
                                     const float* __restrict__ score1,
                                     sara_match* __restrict__ out, int capacity,
                                     int* __restrict__ count, int* __restrict__ overflow)
  {
    const int dir = blockIdx.y;
    const int found = dir == 0 ? *c0 : *c1;
    const int cap = dir == 0 ? cap0 : cap1;
    if (blockIdx.x == 0 && threadIdx.x == 0 && found > cap)
      *overflow = 1;  // members were dropped: the caller redoes the search
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < min(found, cap);
         e += gridDim.x * blockDim.x)
    {
    const float score = (dir == 0 ? score0 : score1)[e];
    if (score < 0.f)
      continue;
    const int twin = (dir == 0 ? twin0 : twin1)[e] - 1;
    if (twin >= 0)
    {
      const float st = (dir == 0 ? score1 : score0)[twin];
      if (st >= 0.f && (dir == 0 ? st < score : st <= score))
        continue;
    }
    const MatchNeighbour me = (dir == 0 ? m0 : m1)[e];
    sara_match m;
    m.x_index = dir == 0 ? me.query : me.index;
    m.y_index = dir == 0 ? me.index : me.query;
    m.score = score;
    m.rank = (dir == 0 ? rank0 : rank1)[e] + 1;
    m.direction = dir;
    const int at = atomicAdd(count, 1);
    if (at < capacity)
      out[at] = m;
    else
      *overflow = 1;
    }
  }

  size_t finish_radius_scratch_ints(int cap0, int cap1)
  {
    // rank, twin, score per member of both directions, then the match ranks
    return 4 * (size_t(cap0) + cap1) + 16;
  }

  size_t finish_radius_cleared_ints(int cap0, int cap1)
  {
    // ranks, twins (index + 1) and match ranks start at 0
    return 3 * (size_t(cap0) + cap1);
  }

  void launch_finish_radius_matches(const MatchNeighbour* m0, const int* c0, int cap0,
                                    const MatchNeighbour* m1, const int* c1, int cap1,
                                    const float* top_d0, int n1, const float* top_d1,
                                    int n2, float squared_ratio_thres, int* iscratch,
                                    sara_match* scratch, int* header, sara_match* out,
                                    hipStream_t stream, bool scratch_cleared)
  {
    const size_t n = size_t(cap0) + cap1;
    int* rank0 = iscratch;
    int* rank1 = rank0 + cap0;
    int* twin0 = rank1 + cap1;
    int* twin1 = twin0 + cap0;
    int* mrank = twin1 + cap1;  // n ints
    float* score0 = reinterpret_cast<float*>(mrank + n);
    float* score1 = score0 + cap0;
    // ranks, twins (index + 1) and match ranks start at 0: one fill
    if (!scratch_cleared)
    {
      ZeroRanges z;
      z.add(rank0, finish_radius_cleared_ints(cap0, cap1));
      launch_zero_ranges(z, stream);
    }
    // a key has a handful of members: grids for 2 per key, strided beyond that
    const int tiles = std::max(1, std::min(int((std::max(cap0, cap1) + 255) / 256),
                                           (2 * std::max(n1, n2) + 255) / 256));
    hipLaunchKernelGGL(radius_pair_kernel, dim3(tiles, tiles, 4), dim3(256), 0, stream,
                       m0, c0, cap0, m1, c1, cap1, rank0, rank1, twin0, twin1);
    hipLaunchKernelGGL(radius_score_kernel, dim3(tiles, 2), dim3(256), 0, stream, m0, c0,
                       cap0, m1, c1, cap1, top_d0, n1, top_d1, n2, squared_ratio_thres,
                       rank0, rank1, score0, score1);
    hipLaunchKernelGGL(radius_emit_kernel, dim3(tiles, 2), dim3(256), 0, stream, m0, c0,
                       cap0, m1, c1, cap1, rank0, rank1, twin0, twin1, score0, score1,
                       scratch, int(n), header, header + 1);
    const int mtiles = std::max(1, std::min(int((n + 255) / 256),
                                            (4 * std::max(n1, n2) + 255) / 256));
    hipLaunchKernelGGL(rank_count_kernel<false>, dim3(mtiles, mtiles), dim3(256), 0, stream,
                       scratch, header, mrank, nullptr);
    hipLaunchKernelGGL(rank_scatter_kernel<false>, dim3(mtiles), dim3(256), 0, stream, scratch,
                       header, mrank, out, nullptr);
  }

  void match_chunking(int nq, int nt, int* chunk, int* nchunks)
  {
    // enough (query block, chunk) waves to fill the chip, chunks of whole tiles
    const int qblocks = (nq + 63) / 64;
    int want = std::max(1, 2048 / std::max(qblocks, 1));
    int c = (nt + want - 1) / want;
    c = ((std::max(c, kMatchTile) + kMatchTile - 1) / kMatchTile) * kMatchTile;
    *chunk = c;
    *nchunks = (nt + c - 1) / c;
  }

  void launch_nn3_exhaustive(const float* q, int nq, const float* t, int nt,
                             int dim, float* part_d, int* part_i, float* top_d,
                             int* top_i, hipStream_t stream)
  {
    int chunk = 0, nchunks = 0;
    match_chunking(nq, nt, &chunk, &nchunks);
    const int dim4 = (dim + 3) & ~3;
    const size_t lds = size_t(kMatchTile) * dim4 * sizeof(float);
    hipLaunchKernelGGL(nn3_kernel, dim3((nq + 63) / 64, nchunks), dim3(64), lds,
                       stream, q, nq, t, nt, dim, chunk, part_d, part_i);
    hipLaunchKernelGGL(merge3_kernel, dim3((nq + 255) / 256), dim3(256), 0, stream,
                       part_d, part_i, nq, nchunks, top_d, top_i);
  }

  void launch_ratio_filter(const float* top_d, const int* top_i, int nq,
                           float squared_ratio_thres, int direction,
                           sara_match* out, int capacity, int* count,
                           hipStream_t stream)
  {
    hipLaunchKernelGGL(ratio_filter_kernel, dim3((nq + 255) / 256), dim3(256), 0,
                       stream, top_d, top_i, nq, squared_ratio_thres, direction,
                       out, capacity, count);
  }

  void launch_radius_exhaustive(const float* q, int nq, const float* t, int nt,
                                int dim, const float* top_d, int top1,
                                float squared_ratio_thres, MatchNeighbour* out,
                                int capacity, int* count, hipStream_t stream)
  {
    int chunk = 0, nchunks = 0;
    match_chunking(nq, nt, &chunk, &nchunks);
    const int dim4 = (dim + 3) & ~3;
    const size_t lds = size_t(kMatchTile) * dim4 * sizeof(float);
    hipLaunchKernelGGL(radius_kernel, dim3((nq + 63) / 64, nchunks), dim3(64), lds,
                       stream, q, nq, t, nt, dim, chunk, top_d, top1,
                       squared_ratio_thres, out, capacity, count);
  }

}  // namespace sara_hip
