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
// nn2_kernel: one wave = 64 queries (one per lane) x one chunk of candidates.
// A tile of 32 candidates is staged in LDS; per group of four dimensions the
// lane loads its own four query values (its row stays in L1/L2 across tiles;
// staging the 64 query rows in LDS as well would cap the CU at 3 waves) and
// reads every candidate's four values as an LDS broadcast; 32 per-candidate
// accumulators live in registers so that each distance is summed group by
// group in FLANN's order.  The per-chunk
// (best, second best) pairs are merged in chunk order by merge_kernel, which
// also applies Lowe's ratio test on the squared distances and appends.
// N x M x 128 subtract/multiply/add at VALU rate: exact float32 semantics,
// which an MFMA contraction (|a|^2 + |b|^2 - 2ab) would not give.
#include "sift_kernels.hpp"

#include <cfloat>

namespace sara_hip {

  constexpr int kMatchTile = 32;   // candidates per LDS tile

  __global__ __launch_bounds__(64) void nn2_kernel(
      const float* __restrict__ q, int nq, const float* __restrict__ t, int nt,
      int dim, int chunk, float* __restrict__ part_d0, float* __restrict__ part_d1,
      int* __restrict__ part_i0)
  {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int dim4 = (dim + 3) & ~3;
    float* s_t = smem;  // [kMatchTile][dim4]
    const int lane = threadIdx.x;
    const int q0 = blockIdx.x * 64;
    const int c_begin = blockIdx.y * chunk;
    const int c_end = min(nt, c_begin + chunk);
    // this lane's query row (lanes past the end re-read the last row)
    const float* myq = q + size_t(min(q0 + lane, nq - 1)) * dim;
    const bool vec4 = (dim % 4 == 0) && (reinterpret_cast<uintptr_t>(q) % 16 == 0);
    float best0 = FLT_MAX, best1 = FLT_MAX;
    int idx0 = -1;
    const int groups = dim / 4, tail = dim - 4 * groups;

    for (int c0 = c_begin; c0 < c_end; c0 += kMatchTile)
    {
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

      float acc[kMatchTile];
#pragma unroll
      for (int c = 0; c < kMatchTile; ++c)
        acc[c] = 0.f;
      for (int g = 0; g < groups; ++g)
      {
        float4 a;
        if (vec4)
          a = *reinterpret_cast<const float4*>(myq + 4 * g);
        else
          a = make_float4(myq[4 * g], myq[4 * g + 1], myq[4 * g + 2],
                          myq[4 * g + 3]);
#pragma unroll
        for (int c = 0; c < kMatchTile; ++c)
        {
          const float4 b = *reinterpret_cast<const float4*>(s_t + c * dim4 + 4 * g);
          const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z,
                      d3 = a.w - b.w;
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
#pragma unroll
      for (int c = 0; c < kMatchTile; ++c)
      {
        const float d = acc[c];
        const bool valid = c0 + c < c_end;
        if (valid && d < best0)
        {
          best1 = best0;
          best0 = d;
          idx0 = c0 + c;
        }
        else if (valid && d < best1)
          best1 = d;
      }
    }
    if (q0 + lane < nq)
    {
      const size_t o = size_t(blockIdx.y) * nq + q0 + lane;
      part_d0[o] = best0;
      part_d1[o] = best1;
      part_i0[o] = idx0;
    }
  }

  //! Merges the per-chunk candidates of every query in chunk order (ties keep
  //! the lower index), applies the ratio test of AnnMatcher.cpp:126-147 and
  //! appends {x, y, score, rank = 1, direction}.
  __global__ void merge_matches_kernel(const float* __restrict__ part_d0,
                                       const float* __restrict__ part_d1,
                                       const int* __restrict__ part_i0, int nq,
                                       int nchunks, float squared_ratio_thres,
                                       int direction, sara_match* __restrict__ out,
                                       int capacity, int* __restrict__ count)
  {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq)
      return;
    float d0 = FLT_MAX, d1 = FLT_MAX;
    int i0 = -1;
    for (int c = 0; c < nchunks; ++c)
    {
      const size_t o = size_t(c) * nq + i;
      const float e0 = part_d0[o], e1 = part_d1[o];
      if (e0 < d0)
      {
        d1 = fminf(d0, e1);
        d0 = e0;
        i0 = part_i0[o];
      }
      else
        d1 = fminf(d1, e0);
    }
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

  size_t match_partials_per_query(int nt, int* chunk, int* nchunks, int nq)
  {
    // enough (query block, chunk) waves to fill the chip, chunks of whole tiles
    const int qblocks = (nq + 63) / 64;
    int want = std::max(1, 2048 / std::max(qblocks, 1));
    int c = (nt + want - 1) / want;
    c = ((std::max(c, kMatchTile) + kMatchTile - 1) / kMatchTile) * kMatchTile;
    *chunk = c;
    *nchunks = (nt + c - 1) / c;
    return size_t(*nchunks);
  }

  void launch_match_direction(const float* q, int nq, const float* t, int nt,
                              int dim, float squared_ratio_thres, int direction,
                              float* part_d0, float* part_d1, int* part_i0,
                              sara_match* out, int capacity, int* count,
                              hipStream_t stream)
  {
    int chunk = 0, nchunks = 0;
    match_partials_per_query(nt, &chunk, &nchunks, nq);
    const int dim4 = (dim + 3) & ~3;
    const size_t lds = size_t(kMatchTile) * dim4 * sizeof(float);
    hipLaunchKernelGGL(nn2_kernel, dim3((nq + 63) / 64, nchunks), dim3(64), lds,
                       stream, q, nq, t, nt, dim, chunk, part_d0, part_d1,
                       part_i0);
    hipLaunchKernelGGL(merge_matches_kernel, dim3((nq + 255) / 256), dim3(256), 0,
                       stream, part_d0, part_d1, part_i0, nq, nchunks,
                       squared_ratio_thres, direction, out, capacity, count);
  }

}  // namespace sara_hip
