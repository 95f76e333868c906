// Descriptor matching on gfx950, fast path: MFMA prefilter + exact re-ranking
// (reference: AnnMatcher.cpp:59-170).
//
// What the matcher needs per query is (match_kernels.hip): its three nearest
// neighbours by FLANN's float32 distance, ordered by (distance, index), and -
// for squared ratio thresholds above 1 - every neighbour inside a radius.  The
// exhaustive kernels compute all n1 x n2 x 128 subtract/multiply/add triples
// in FLANN's order on the vector ALUs.  Here the n1 x n2 squared distances are
// first APPROXIMATED as |a|^2 + |b|^2 - 2 a.b with the dot products on the
// matrix cores (v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulation,
// one contraction for both matching directions), with a rigorous bound E on
// |approximation - true distance|; only the few candidates per query that the
// bound cannot exclude are then evaluated in FLANN's exact arithmetic.  The
// results (top-3 lists, radius members) are therefore the SAME floats and
// indices as the exhaustive search returns - not "close": tests compare the
// two paths entry by entry (tests/test_gpu_matching.py).
//
// Error bound (u = 2^-24, all magnitudes bounded by |a|^2 + |b|^2 =: s):
//   |a|^2, |b|^2 summed in float32          <= 1.01 (dim + 1) u s
//   dot product, any order of dim fma/adds  <= 1.01 dim u |a||b| <= .. dim u s / 2 * 2
//   the two final additions                 <= 4 u s
// => |approx - d| <= E := kGuard (2 dim + 8) u (|a|^2 + max_j |b_j|^2), and
// FLANN's float32 distance d_f = d (1 + theta), |theta| <= (dim + 4) u.
// A candidate list built as { j : approx(j) <= tau } with
//   tau = m3 + |m3| 1e-4 + 2.01 E        (m3: third smallest approximation)
// contains every j with d_f(j) <= third smallest d_f; with
//   tau_r = (m_top1 + E) thres^2 (1 + 1e-4) + E
// it contains every j with d_f(j) < d_f(top1) * thres^2 (the radius search).
//
// Pipeline (one stream, no host round trip inside):
//   row_norms                |a_i|^2, |b_j|^2, their maxima
//   mfma_tiles<MINIMA>       128 x 128 tiles: 3 smallest approximations of every
//                            row and every column of the tile
//   thresholds               global m1..m3 per query -> tau
//   mfma_tiles<EMIT>         the same tiles again: (query, index) with
//                            approx <= tau into per-query slots
//   rerank                   exact FLANN distances of the slots -> knnSearch(3)
//                            answer, radius members; queries whose slots
//                            overflowed are flagged ...
//   fallback                 ... and searched exhaustively (one wave each).
#include "sift_kernels.hpp"

#include <atomic>
#include <cfloat>
#include <climits>

namespace sara_hip {

  namespace {
    constexpr int kTile = 128;         // rows and columns of a macro tile
    constexpr int kLdsStride = 130;    // floats per staged row: conflict-free
    constexpr int kDStride = 129;      // floats per row of the distance tile
    constexpr float kGuard = 1.25f;    // slack on the error bound
    constexpr float kUnit = 5.9604645e-8f;  // 2^-24

    using f32x16 = __attribute__((ext_vector_type(16))) float;

    //! flann::L2<float>::operator() (dist.h:150-178), exact evaluation order.
    __device__ inline float flann_l2_rows(const float* __restrict__ a,
                                          const float* __restrict__ b, int dim)
    {
      float result = 0.f;
      int i = 0;
      const bool vec4 = ((reinterpret_cast<uintptr_t>(a) |
                          reinterpret_cast<uintptr_t>(b)) & 15) == 0;
      if (vec4)
        for (; i + 3 < dim; i += 4)
        {
          const float4 x = *reinterpret_cast<const float4*>(a + i);
          const float4 y = *reinterpret_cast<const float4*>(b + i);
          const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
          result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        }
      else
        for (; i + 3 < dim; i += 4)
        {
          const float d0 = a[i] - b[i], d1 = a[i + 1] - b[i + 1];
          const float d2 = a[i + 2] - b[i + 2], d3 = a[i + 3] - b[i + 3];
          result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        }
      for (; i < dim; ++i)
      {
        const float d0 = a[i] - b[i];
        result += d0 * d0;
      }
      return result;
    }

    __device__ inline void min3_update(float v, float& m1, float& m2, float& m3)
    {
      // m1 <= m2 <= m3 stay ordered: the medians pick the right survivor
      m3 = __builtin_amdgcn_fmed3f(m2, m3, v);
      m2 = __builtin_amdgcn_fmed3f(m1, m2, v);
      m1 = fminf(m1, v);
    }

    //! |x_i|^2 of every row and the maximum over the rows (norms are >= 0, so
    //! their bit patterns order like unsigned integers).
    __global__ void row_norms_kernel(const float* __restrict__ x, int n, int dim,
                                     float* __restrict__ norms,
                                     unsigned* __restrict__ max_bits)
    {
      const int i = blockIdx.x * blockDim.x + threadIdx.x;
      float s = 0.f;
      if (i < n)
      {
        const float* r = x + size_t(i) * dim;
        for (int k = 0; k < dim; ++k)
          s += r[k] * r[k];
        norms[i] = s;
      }
      // wave maximum, one atomic per wave
      float m = s;
      for (int o = 32; o > 0; o >>= 1)
        m = fmaxf(m, __shfl_xor(m, o));
      if ((threadIdx.x & 63) == 0)
        atomicMax(max_bits, __float_as_uint(m));
    }

    enum
    {
      kMinima = 0,
      kEmit = 1
    };

    //! One 128 x 128 tile of approximate squared distances between rows
    //! [row0, row0 + 128) of A and [col0, col0 + 128) of B.  256 threads = 4
    //! waves, each a 64 x 64 quadrant = 2 x 2 MFMA blocks of 32 x 32; both
    //! operand panels sit in LDS for the whole contraction (dim <= 128).
    template <int MODE>
    __global__ __launch_bounds__(256) void mfma_tiles_kernel(
        const float* __restrict__ A, int n1, const float* __restrict__ B, int n2,
        int dim, const float* __restrict__ na, const float* __restrict__ nb,
        // MINIMA: [tiles along the other axis][n][3]
        float* __restrict__ rowmin, float* __restrict__ colmin,
        // EMIT
        const float* __restrict__ tau_row, const float* __restrict__ tau_col,
        int* __restrict__ cand_row, int* __restrict__ cnt_row,
        int* __restrict__ cand_col, int* __restrict__ cnt_col, int cap,
        int with_cols)
    {
      extern __shared__ __attribute__((aligned(16))) float lds[];
      float* sA = lds;
      float* sB = lds + kTile * kLdsStride;
      const int tid = threadIdx.x;
      const int row0 = blockIdx.y * kTile, col0 = blockIdx.x * kTile;
      const int k2 = (dim + 1) / 2;  // MFMA steps of two k each

      // ---- stage the two panels (zero-padded) --------------------------------
      const bool vec4 = (dim % 4 == 0) &&
                        ((reinterpret_cast<uintptr_t>(A) |
                          reinterpret_cast<uintptr_t>(B)) % 16 == 0);
      if (vec4)
      {
        const int q = dim / 4;
        for (int idx = tid; idx < kTile * q; idx += 256)
        {
          const int r = idx / q, c = idx - r * q;
          float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
          if (row0 + r < n1)
            va = *reinterpret_cast<const float4*>(A + size_t(row0 + r) * dim + 4 * c);
          if (col0 + r < n2)
            vb = *reinterpret_cast<const float4*>(B + size_t(col0 + r) * dim + 4 * c);
          float* pa = sA + r * kLdsStride + 4 * c;
          float* pb = sB + r * kLdsStride + 4 * c;
          *reinterpret_cast<float2*>(pa) = make_float2(va.x, va.y);
          *reinterpret_cast<float2*>(pa + 2) = make_float2(va.z, va.w);
          *reinterpret_cast<float2*>(pb) = make_float2(vb.x, vb.y);
          *reinterpret_cast<float2*>(pb + 2) = make_float2(vb.z, vb.w);
        }
      }
      else
      {
        const int kk = 2 * k2;
        for (int idx = tid; idx < kTile * kk; idx += 256)
        {
          const int r = idx / kk, k = idx - r * kk;
          sA[r * kLdsStride + k] =
              (row0 + r < n1 && k < dim) ? A[size_t(row0 + r) * dim + k] : 0.f;
          sB[r * kLdsStride + k] =
              (col0 + r < n2 && k < dim) ? B[size_t(col0 + r) * dim + k] : 0.f;
        }
      }
      __syncthreads();

      // ---- the contraction -----------------------------------------------------
      const int lane = tid & 63, wave = tid >> 6;
      const int wm = wave >> 1, wn = wave & 1;
      const int li = lane & 31, half = lane >> 5;
      const float* pa0 = sA + (wm * 64 + li) * kLdsStride + half;
      const float* pa1 = pa0 + 32 * kLdsStride;
      const float* pb0 = sB + (wn * 64 + li) * kLdsStride + half;
      const float* pb1 = pb0 + 32 * kLdsStride;
      f32x16 acc00 = {}, acc01 = {}, acc10 = {}, acc11 = {};
#pragma unroll 4
      for (int s = 0; s < k2; ++s)
      {
        const float a0 = pa0[2 * s], a1 = pa1[2 * s];
        const float b0 = pb0[2 * s], b1 = pb1[2 * s];
        acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
        acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
      }
      __syncthreads();  // the panels are dead: their LDS is reused below

      // squared norms of this tile's rows / columns (+inf-like for the padding:
      // a row or column past the end can never be anyone's neighbour)
      float* sNa = lds + kTile * kDStride;  // behind the distance tile
      float* sNb = sNa + kTile;
      float* sTr = sNb + kTile;
      float* sTc = sTr + kTile;
      if (tid < kTile)
      {
        sNa[tid] = row0 + tid < n1 ? na[row0 + tid] : FLT_MAX;
        if (MODE == kEmit)
          sTr[tid] = row0 + tid < n1 ? tau_row[row0 + tid] : -FLT_MAX;
      }
      else
      {
        const int c = tid - kTile;
        sNb[c] = col0 + c < n2 ? nb[col0 + c] : FLT_MAX;
        if (MODE == kEmit)
          sTc[c] = (with_cols && col0 + c < n2) ? tau_col[col0 + c] : -FLT_MAX;
      }
      __syncthreads();

      // C/D layout of the 32 x 32 blocks: lane -> column lane & 31, register r
      // -> row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
      auto approx = [&](const f32x16& acc, int bi, int bj, int r, int& row,
                        int& col) -> float {
        row = wm * 64 + bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        col = wn * 64 + bj * 32 + li;
        const float s = sNa[row] + sNb[col];  // >= FLT_MAX for the padding
        // +inf: below no threshold (they are <= FLT_MAX), above every minimum
        return s >= FLT_MAX ? __builtin_huge_valf() : s - 2.f * acc[r];
      };

      if (MODE == kMinima)
      {
        float* sD = lds;  // [128][129]
#pragma unroll
        for (int r = 0; r < 16; ++r)
        {
          int row, col;
          float v = approx(acc00, 0, 0, r, row, col);
          sD[row * kDStride + col] = v;
          v = approx(acc01, 0, 1, r, row, col);
          sD[row * kDStride + col] = v;
          v = approx(acc10, 1, 0, r, row, col);
          sD[row * kDStride + col] = v;
          v = approx(acc11, 1, 1, r, row, col);
          sD[row * kDStride + col] = v;
        }
        __syncthreads();
        float m1 = FLT_MAX, m2 = FLT_MAX, m3 = FLT_MAX;
        if (tid < kTile)
        {
          const float* p = sD + tid * kDStride;
#pragma unroll 8
          for (int c = 0; c < kTile; ++c)
            min3_update(p[c], m1, m2, m3);
          if (row0 + tid < n1)
          {
            float* o = rowmin + (size_t(blockIdx.x) * n1 + row0 + tid) * 3;
            o[0] = m1;
            o[1] = m2;
            o[2] = m3;
          }
        }
        else if (with_cols)
        {
          const int c = tid - kTile;
          const float* p = sD + c;
#pragma unroll 8
          for (int r = 0; r < kTile; ++r)
            min3_update(p[r * kDStride], m1, m2, m3);
          if (col0 + c < n2)
          {
            float* o = colmin + (size_t(blockIdx.y) * n2 + col0 + c) * 3;
            o[0] = m1;
            o[1] = m2;
            o[2] = m3;
          }
        }
      }
      else
      {
        auto emit = [&](const f32x16& acc, int bi, int bj) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
          {
            int row, col;
            const float v = approx(acc, bi, bj, r, row, col);
            if (v <= sTr[row])
            {
              const int q = row0 + row;
              const int slot = atomicAdd(cnt_row + q, 1);
              if (slot < cap)
                cand_row[size_t(q) * cap + slot] = col0 + col;
            }
            if (v <= sTc[col])
            {
              const int q = col0 + col;
              const int slot = atomicAdd(cnt_col + q, 1);
              if (slot < cap)
                cand_col[size_t(q) * cap + slot] = row0 + row;
            }
          }
        };
        emit(acc00, 0, 0);
        emit(acc01, 0, 1);
        emit(acc10, 1, 0);
        emit(acc11, 1, 1);
      }
    }

    //! Global three smallest approximations of every query -> its threshold.
    //! top1: rank of the best real neighbour (1 when a set is matched against
    //! itself: rank 0 is the query).
    __global__ void thresholds_kernel(const float* __restrict__ partial, int ntiles,
                                      int n, const float* __restrict__ norms,
                                      const unsigned* __restrict__ other_max_bits,
                                      int dim, float squared_ratio_thres, int top1,
                                      float* __restrict__ tau)
    {
      const int i = blockIdx.x * blockDim.x + threadIdx.x;
      if (i >= n)
        return;
      float m1 = FLT_MAX, m2 = FLT_MAX, m3 = FLT_MAX;
      for (int t = 0; t < ntiles; ++t)
      {
        const float* p = partial + (size_t(t) * n + i) * 3;
        min3_update(p[0], m1, m2, m3);
        min3_update(p[1], m1, m2, m3);
        min3_update(p[2], m1, m2, m3);
      }
      const float e = kGuard * float(2 * dim + 8) * kUnit *
                      (norms[i] + __uint_as_float(*other_max_bits));
      // fewer than three candidates: everything passes
      float t = m3 >= FLT_MAX ? FLT_MAX : m3 + fabsf(m3) * 1e-4f + 2.01f * e;
      if (squared_ratio_thres > 1.f && t < FLT_MAX)
      {
        const float mt = top1 == 0 ? m1 : m2;
        const float r = (mt + e) * squared_ratio_thres;
        t = fmaxf(t, r + fabsf(r) * 1e-4f + e);
      }
      tau[i] = t;
    }

    //! Exact distances of every query's candidate slots; CAP lanes per query.
    template <int CAP>
    __global__ __launch_bounds__(256) void rerank_kernel(
        const float* __restrict__ q, int nq, const float* __restrict__ t, int dim,
        const int* __restrict__ cand, const int* __restrict__ cnt,
        float squared_ratio_thres, int top1, float* __restrict__ top_d,
        int* __restrict__ top_i, MatchNeighbour* __restrict__ radius_out,
        int radius_cap, int* __restrict__ radius_count, int* __restrict__ flagged,
        int* __restrict__ flagged_count)
    {
      const int gid = blockIdx.x * blockDim.x + threadIdx.x;
      const int qi = gid / CAP, slot = gid % CAP;
      const int lane = threadIdx.x & 63;
      const int base = lane - slot;  // first lane of this query's group
      const bool live = qi < nq;
      const int c = live ? cnt[qi] : 0;
      const bool overflow = c > CAP;
      if (live && overflow && slot == 0)
        flagged[atomicAdd(flagged_count, 1)] = qi;
      const bool valid = live && !overflow && slot < c;
      int idx = INT_MAX;
      float d = FLT_MAX;
      if (valid)
      {
        idx = cand[size_t(qi) * CAP + slot];
        d = flann_l2_rows(q + size_t(qi) * dim, t + size_t(idx) * dim, dim);
      }
      // rank by (distance, index) inside the group; d of rank top1
      int rank = 0;
      float d_top1 = FLT_MAX;
      // first the ranks ...
      for (int k = 0; k < CAP; ++k)
      {
        const float od = __shfl(d, base + k);
        const int oi = __shfl(idx, base + k);
        rank += (od < d || (od == d && oi < idx)) ? 1 : 0;
      }
      // ... then the distance the radius is built from
      for (int k = 0; k < CAP; ++k)
      {
        const float od = __shfl(d, base + k);
        const int orank = __shfl(rank, base + k);
        const int oi = __shfl(idx, base + k);
        if (orank == top1 && oi != INT_MAX)
          d_top1 = od;
      }
      if (!live || overflow)
        return;
      if (valid && rank < 3)
      {
        top_d[size_t(rank) * nq + qi] = d;
        top_i[size_t(rank) * nq + qi] = idx;
      }
      if (slot < 3 && slot >= c)  // fewer than three candidates in all
      {
        top_d[size_t(slot) * nq + qi] = FLT_MAX;
        top_i[size_t(slot) * nq + qi] = -1;
      }
      if (squared_ratio_thres > 1.f && valid && d_top1 < FLT_MAX &&
          d < d_top1 * squared_ratio_thres)
      {
        const int o = atomicAdd(radius_count, 1);
        if (o < radius_cap)
          radius_out[o] = MatchNeighbour{qi, idx, d};
      }
    }

    //! Queries whose candidate slots overflowed: exhaustive search, one wave
    //! per query, lanes stride over the candidates.
    __global__ __launch_bounds__(64) void fallback_kernel(
        const float* __restrict__ q, int nq, const float* __restrict__ t, int nt,
        int dim, const int* __restrict__ flagged,
        const int* __restrict__ flagged_count, float squared_ratio_thres, int top1,
        float* __restrict__ top_d, int* __restrict__ top_i,
        MatchNeighbour* __restrict__ radius_out, int radius_cap,
        int* __restrict__ radius_count)
    {
      const int lane = threadIdx.x;
      const int n = *flagged_count;
      for (int k = blockIdx.x; k < n; k += gridDim.x)
      {
        const int qi = flagged[k];
        const float* qr = q + size_t(qi) * dim;
        float b[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        int bi[3] = {INT_MAX, INT_MAX, INT_MAX};
        for (int j = lane; j < nt; j += 64)
        {
          const float d = flann_l2_rows(qr, t + size_t(j) * dim, dim);
          if (d < b[0])
          {
            b[2] = b[1]; bi[2] = bi[1]; b[1] = b[0]; bi[1] = bi[0]; b[0] = d; bi[0] = j;
          }
          else if (d < b[1])
          {
            b[2] = b[1]; bi[2] = bi[1]; b[1] = d; bi[1] = j;
          }
          else if (d < b[2])
          {
            b[2] = d; bi[2] = j;
          }
        }
        // three rounds: the wave's smallest (distance, index), popped from
        // the lane that holds it
        float d_top1 = FLT_MAX;
        for (int r = 0; r < 3; ++r)
        {
          float md = b[0];
          int mi = bi[0];
          for (int o = 32; o > 0; o >>= 1)
          {
            const float od = __shfl_xor(md, o);
            const int oi = __shfl_xor(mi, o);
            if (od < md || (od == md && oi < mi))
            {
              md = od;
              mi = oi;
            }
          }
          if (bi[0] == mi && mi != INT_MAX)
          {
            b[0] = b[1]; bi[0] = bi[1]; b[1] = b[2]; bi[1] = bi[2];
            b[2] = FLT_MAX; bi[2] = INT_MAX;
          }
          if (lane == 0)
          {
            top_d[size_t(r) * nq + qi] = mi == INT_MAX ? FLT_MAX : md;
            top_i[size_t(r) * nq + qi] = mi == INT_MAX ? -1 : mi;
          }
          if (r == top1 && mi != INT_MAX)
            d_top1 = md;
        }
        if (squared_ratio_thres > 1.f && d_top1 < FLT_MAX)
        {
          const float radius = d_top1 * squared_ratio_thres;
          for (int j = lane; j < nt; j += 64)
          {
            const float d = flann_l2_rows(qr, t + size_t(j) * dim, dim);
            if (d < radius)
            {
              const int o = atomicAdd(radius_count, 1);
              if (o < radius_cap)
                radius_out[o] = MatchNeighbour{qi, j, d};
            }
          }
        }
      }
    }

    template <typename K>
    void allow_big_lds(K kernel)
    {
      static std::atomic<bool> done[64];
      int dev = 0;
      (void) hipGetDevice(&dev);
      if (!done[dev & 63].load(std::memory_order_acquire))
      {
        (void) hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024);
        done[dev & 63].store(true, std::memory_order_release);
      }
    }

  }  // namespace

  size_t match_mfma_scratch_floats(int n1, int n2)
  {
    const size_t tm = (size_t(n1) + kTile - 1) / kTile, tn = (size_t(n2) + kTile - 1) / kTile;
    // norms (n1 + n2 + 2), tau (n1 + n2), row minima [tn][n1][3], column minima [tm][n2][3]
    return 2 * (size_t(n1) + n2) + 16 + 3 * (tn * n1 + tm * n2);
  }

  size_t match_mfma_scratch_ints(int n1, int n2, int cap)
  {
    // counters (n1 + n2), slots, flagged lists (n1 + n2), 4 scalars
    return (size_t(n1) + n2) * (size_t(cap) + 2) + 16;
  }

  //! See the header of this file.  d1: n1 x dim, d2: n2 x dim (device).  Fills
  //! top_d / top_i ([3][n1] for direction 0 at `top12`, [3][n2] for direction 1
  //! at `top21`; direction 1 is skipped when with_dir1 == 0) and, for squared
  //! thresholds above 1, appends the radius members of direction 0 / 1 to
  //! radius12 / radius21 (counts keep counting past the capacity).
  void launch_match_mfma(const float* d1, int n1, const float* d2, int n2, int dim,
                         float squared_ratio_thres, int top1, int with_dir1,
                         float* fscratch, int* iscratch, int cap,
                         float* top12_d, int* top12_i, float* top21_d, int* top21_i,
                         MatchNeighbour* radius12, int radius12_cap, int* radius12_count,
                         MatchNeighbour* radius21, int radius21_cap, int* radius21_count,
                         hipStream_t stream)
  {
    const int tm = (n1 + kTile - 1) / kTile, tn = (n2 + kTile - 1) / kTile;
    // ---- carve the scratch
    float* na = fscratch;
    float* nb = na + n1;
    unsigned* maxbits = reinterpret_cast<unsigned*>(nb + n2);  // [0] of A, [1] of B
    float* tau_r = nb + n2 + 16;
    float* tau_c = tau_r + n1;
    float* rowmin = tau_c + n2;
    float* colmin = rowmin + 3 * size_t(tn) * n1;
    int* cnt_r = iscratch;
    int* cnt_c = cnt_r + n1;
    int* scal = cnt_c + n2;  // [0] flagged rows, [1] flagged columns
    int* flag_r = scal + 16;
    int* flag_c = flag_r + n1;
    int* cand_r = flag_c + n2;
    int* cand_c = cand_r + size_t(n1) * cap;
    (void) hipMemsetAsync(maxbits, 0, 16 * sizeof(float), stream);
    (void) hipMemsetAsync(cnt_r, 0, sizeof(int) * (size_t(n1) + n2 + 16), stream);
    hipLaunchKernelGGL(row_norms_kernel, dim3((n1 + 255) / 256), dim3(256), 0, stream,
                       d1, n1, dim, na, maxbits);
    hipLaunchKernelGGL(row_norms_kernel, dim3((n2 + 255) / 256), dim3(256), 0, stream,
                       d2, n2, dim, nb, maxbits + 1);
    const size_t lds = sizeof(float) * 2 * kTile * kLdsStride;
    allow_big_lds(mfma_tiles_kernel<kMinima>);
    allow_big_lds(mfma_tiles_kernel<kEmit>);
    const dim3 grid(tn, tm);
    hipLaunchKernelGGL(mfma_tiles_kernel<kMinima>, grid, dim3(256), lds, stream, d1,
                       n1, d2, n2, dim, na, nb, rowmin, colmin, nullptr, nullptr,
                       nullptr, nullptr, nullptr, nullptr, cap, with_dir1);
    hipLaunchKernelGGL(thresholds_kernel, dim3((n1 + 255) / 256), dim3(256), 0, stream,
                       rowmin, tn, n1, na, maxbits + 1, dim, squared_ratio_thres,
                       top1, tau_r);
    if (with_dir1)
      hipLaunchKernelGGL(thresholds_kernel, dim3((n2 + 255) / 256), dim3(256), 0,
                         stream, colmin, tm, n2, nb, maxbits, dim,
                         squared_ratio_thres, top1, tau_c);
    hipLaunchKernelGGL(mfma_tiles_kernel<kEmit>, grid, dim3(256), lds, stream, d1, n1,
                       d2, n2, dim, na, nb, nullptr, nullptr, tau_r, tau_c, cand_r,
                       cnt_r, cand_c, cnt_c, cap, with_dir1);
    auto rerank = [&](const float* q, int nq, const float* t, int nt, const int* cand,
                      const int* cnt, float* td, int* ti, MatchNeighbour* ro, int rcap,
                      int* rcount, int* flagged, int* fcount) {
      const size_t threads = size_t(nq) * cap;
      const dim3 g(unsigned((threads + 255) / 256));
      if (cap == 8)
        hipLaunchKernelGGL(rerank_kernel<8>, g, dim3(256), 0, stream, q, nq, t, dim,
                           cand, cnt, squared_ratio_thres, top1, td, ti, ro, rcap,
                           rcount, flagged, fcount);
      else
        hipLaunchKernelGGL(rerank_kernel<32>, g, dim3(256), 0, stream, q, nq, t, dim,
                           cand, cnt, squared_ratio_thres, top1, td, ti, ro, rcap,
                           rcount, flagged, fcount);
      hipLaunchKernelGGL(fallback_kernel, dim3(std::min(nq, 512)), dim3(64), 0, stream,
                         q, nq, t, nt, dim, flagged, fcount, squared_ratio_thres,
                         top1, td, ti, ro, rcap, rcount);
    };
    rerank(d1, n1, d2, n2, cand_r, cnt_r, top12_d, top12_i, radius12, radius12_cap,
           radius12_count, flag_r, scal);
    if (with_dir1)
      rerank(d2, n2, d1, n1, cand_c, cnt_c, top21_d, top21_i, radius21, radius21_cap,
             radius21_count, flag_c, scal + 1);
  }

}  // namespace sara_hip
