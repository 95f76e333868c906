// Descriptor matching on gfx950, fast path: MFMA prefilter + exact re-ranking
// (reference: AnnMatcher.cpp:59-170).
//
// What the matcher needs per query is (match_kernels.hip): its three nearest
// neighbours by FLANN's float32 distance, ordered by (distance, index), and -
// for squared ratio thresholds above 1 - every neighbour inside a radius.  The
// exhaustive kernels compute all n1 x n2 x 128 subtract/multiply/add triples
// in FLANN's order on the vector ALUs.  Here the n1 x n2 squared distances are
// first APPROXIMATED as |a|^2 + |b|^2 - 2 a.b with the dot products on the
// matrix cores (round 5: v_mfma_f32_32x32x16_bf16 on a hi / lo split of the
// rows, three products per pair, f32 accumulation; rounds 3-4:
// v_mfma_f32_32x32x2_f32; one contraction for both matching directions), with
// a rigorous bound E on
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
// => |approx - d| <= E := kGuard (2 dim + 8) u (|a|^2 + max_j |b_j|^2) for the
// f32 contraction of rounds 3-4.  Round 5 contracts on the bf16 matrix cores
// (16 x the f32 MFMA rate) with a split representation x = hi + lo + e,
// hi = bf16(x), lo = bf16(x - hi), |e| <= 2^-16 |x| (|x - hi| <= 2^-8 |x|,
// |lo| <= 2^-8 (1 + 2^-8) |x|), and three products per pair:
//   a.b ~ sum hi hi + hi lo + lo hi      (bf16 x bf16 is exact in float32)
//   dropped: |lo lo| + |e_a b| + |a e_b| <= 3.03 2^-16 |a_i||b_i| per term
//   float32 accumulation of 3 dim terms  <= 1.01 (3 dim) u sum |terms|
//                                        <= 1.02 (3 dim) u |a||b|
//   |a||b| <= s / 2, and the approximation uses 2 a.b:
// => |approx - d| <= E := kGuard ((4.1 dim + 8) u + 3.03 2^-16) s, about four
// times the f32 bound (for SIFT rows: 40 squared-distance units at distances of
// 10^4..10^5) - the candidate lists grow by a few per cent.  And
// FLANN's float32 distance d_f = d (1 + theta), |theta| <= (dim + 4) u.
// A candidate list built as { j : approx(j) <= tau } with
//   tau = m3 + |m3| 1e-4 + 2.01 E        (m3: third smallest approximation)
// contains every j with d_f(j) <= third smallest d_f; with
//   tau_r = (m_top1 + E) thres^2 (1 + 1e-4) + E
// it contains every j with d_f(j) < d_f(top1) * thres^2 (the radius search).
//
// Round 3: for ratios <= 1 the two passes over the tiles below are ONE
// (kPacked / select_packed_kernel: minima kept with their positions); the
// radius search keeps both.
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
#include <cstdio>
#include <vector>

namespace sara_hip {

  namespace {
    constexpr int kTile = 128;         // rows and columns of a macro tile
    constexpr int kChunk = 64;         // k staged per chunk
    constexpr int kQueue = 3072;       // LDS queue of the emit pass, in hits
    constexpr int kDStride = 129;      // floats per row of the distance tile
    constexpr float kGuard = 1.25f;    // slack on the error bound
    constexpr int kFallbackSlots = 128;  // flagged queries whose distances are staged
    constexpr int kFallbackParts = 8;    // workgroups per staged query
    constexpr float kUnit = 5.9604645e-8f;  // 2^-24
    // bf16 panels: 64 k of a row = 128 bytes, rows 144 bytes apart (the 16-byte
    // fragment reads of 16 consecutive rows then fall on distinct banks)
    constexpr int kPanelBytes = 144;
    constexpr int kMaxDimPadded = 128;  // dim <= 128 (sara_hip_sift.h), padded to 64
    //! E / (|a|^2 + max |b|^2), see the header of this file.
    __host__ __device__ inline float guard_coeff(int dim)
    {
      return kGuard * ((4.1f * float(dim) + 8.f) * kUnit + 3.03f / 65536.f);
    }
    using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
    //! Round to nearest even; a finite input that would round up to infinity
    //! takes the largest finite bf16 instead (|v - hi| <= 2^-8 |v| still holds).
    __device__ inline unsigned short to_bf16(float v)
    {
      const unsigned u = __float_as_uint(v);
      unsigned short h = (unsigned short) ((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
      if ((h & 0x7fffu) == 0x7f80u && (u & 0x7fffffffu) < 0x7f800000u)
        h = (unsigned short) (h - 1u);
      return h;
    }
    __device__ inline float from_bf16(unsigned short h)
    {
      return __uint_as_float(unsigned(h) << 16);
    }

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
#pragma unroll 8
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

    //! A value every lane of the wave holds alike (read through a table entry
    //! indexed by blockIdx): pinned into SGPRs, or the compiler keeps a copy
    //! per lane and the tile kernel spills.
    __device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
    //! Pointers read from the pair table are pointers into HBM: said so (a
    //! pointer that comes out of a load is generic to the compiler, and the tile
    //! kernel - global loads into LDS, accumulator arrays it wants in registers -
    //! spilled 72 dwords per lane on generic ones).
    template <typename T>
    __device__ __forceinline__ T* uni(T* p)
    {
      using G = __attribute__((address_space(1))) T*;
      return (T*) (G) p;
    }

    //! |x_i|^2 of every row of both key sets and the maximum over each set
    //! (norms are >= 0, so their bit patterns order like unsigned integers).
    //! 16 lanes per row, 4 rows per wave; any summation order fits the bound.
    template <bool BATCH = false>
    __global__ __launch_bounds__(1024) void row_norms_kernel(
        const float* __restrict__ x1, int n1, const float* __restrict__ x2, int n2,
        int dim, float* __restrict__ norms1, float* __restrict__ norms2,
        unsigned* __restrict__ max_bits, unsigned short* __restrict__ split1,
        unsigned short* __restrict__ split2, int dimp,
        const MatchBatchPair* __restrict__ batch)
    {
      if constexpr (BATCH)  // pair blockIdx.y of a batch
      {
        const MatchBatchPair& b = batch[blockIdx.y];
        x1 = uni(b.d1);
        n1 = uni(b.n1);
        x2 = uni(b.d2);
        n2 = uni(b.n2);
        norms1 = uni(b.na);
        norms2 = uni(b.nb);
        max_bits = uni(b.maxbits);
        split1 = uni(b.split1);
        split2 = uni(b.split2);
      }
      const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;  // row of both
      const int sub = threadIdx.x & 15;
      const bool second = g >= n1;
      const int i = second ? g - n1 : g;
      const int n = second ? n2 : n1;
      const float* x = second ? x2 : x1;
      float s = 0.f;
      if (i < n)
      {
        const float* r = x + size_t(i) * dim;
        for (int k = sub; k < dim; k += 16)
          s += r[k] * r[k];
        // the row as hi / lo bf16 halves (hi: [i][0..dimp), lo: the same array
        // n rows further), zero-padded to dimp
        unsigned short* hi = (second ? split2 : split1) + size_t(i) * dimp;
        unsigned short* lo = hi + size_t(n) * dimp;
        for (int k = sub; k < dimp; k += 16)
        {
          const float v = k < dim ? r[k] : 0.f;
          const unsigned short h = to_bf16(v);
          hi[k] = h;
          lo[k] = to_bf16(v - from_bf16(h));
        }
      }
      for (int o = 8; o > 0; o >>= 1)
        s += __shfl_xor(s, o);
      if (i < n && sub == 0)
        (second ? norms2 : norms1)[i] = s;
      // maxima of the two sets: wave -> workgroup -> one atomic pair per
      // workgroup (an atomic per row serialised 8.6 k of them: 100 us)
      __shared__ unsigned s_max[2];
      if (threadIdx.x < 2)
        s_max[threadIdx.x] = 0u;
      __syncthreads();
      float m0 = (i < n && !second) ? s : 0.f, m1 = (i < n && second) ? s : 0.f;
      for (int o = 32; o > 0; o >>= 1)
      {
        m0 = fmaxf(m0, __shfl_xor(m0, o));
        m1 = fmaxf(m1, __shfl_xor(m1, o));
      }
      if ((threadIdx.x & 63) == 0)
      {
        atomicMax(&s_max[0], __float_as_uint(m0));
        atomicMax(&s_max[1], __float_as_uint(m1));
      }
      __syncthreads();
      if (threadIdx.x < 2 && s_max[threadIdx.x] != 0u)
        atomicMax(max_bits + threadIdx.x, s_max[threadIdx.x]);
    }

    enum
    {
      kMinima = 0,
      kEmit = 1,
      kPacked = 2  // round 3: minima WITH their column / row, one pass (ratios <= 1)
    };

    // ---- packed minima (kPacked) ------------------------------------------------
    // An approximate distance and the position of its column inside the tile in
    // one 32-bit key that orders like the distance: the float's bits mapped to a
    // monotone integer, the low 7 bits replaced by the position.  The value read
    // back from a key is <= the approximation and within 2^-16 of it (relative).
    __device__ inline int ordered_bits(float v)
    {
      const int b = __float_as_int(v);
      return b ^ ((b >> 31) & 0x7fffffff);
    }
    __device__ inline float from_ordered_bits(int o)
    {
      return __int_as_float(o ^ ((o >> 31) & 0x7fffffff));
    }
    constexpr int kNoKey = 0x7f7fff80;  // keys of +inf / FLT_MAX sums: padding
    __device__ inline int med3i(int a, int b, int c)
    {
      return max(min(a, b), min(max(a, b), c));
    }
    __device__ inline void min4_update(int v, int& k1, int& k2, int& k3, int& k4)
    {
      k4 = med3i(k3, k4, v);
      k3 = med3i(k2, k3, v);
      k2 = med3i(k1, k2, v);
      k1 = min(k1, v);
    }

    //! One 128 x 128 tile of approximate squared distances between rows
    //! [row0, row0 + 128) of A and [col0, col0 + 128) of B.  256 threads = 4
    //! waves, each a 64 x 64 quadrant = 2 x 2 MFMA blocks of 32 x 32.
    //! The contraction runs in chunks of 64 k: the hi / lo panels of both
    //! operands take 72 KB of LDS, so two workgroups share a CU.  Per pass over
    //! 4.3 k x 4.3 k keys: 42 us (minima) / 54 us (emit) - with the f32
    //! instruction of rounds 3-4 80 us each, of which the contraction alone was
    //! 40 (timing builds with a phase removed: staging 23, minima epilogue 16;
    //! co-resident workgroups fall into step, so the phases add up).  The bf16
    //! contraction is 12 MFMA per 16 k and quadrant at 16 x the f32 rate: what
    //! is left is staging and the epilogue.
    //! The tile at (tile_x, tile_y) by the 256 threads of the workgroup: the body
    //! of mfma_tiles_kernel (single pair: the arguments are the kernel's) and of
    //! mfma_tiles_batch_kernel (the arguments come out of the pair table).
    template <int MODE>
    __device__ __forceinline__ void mfma_tile(
        const unsigned short* __restrict__ Ah, int n1,
        const unsigned short* __restrict__ Bh, int n2, int dimp,
        const float* __restrict__ na, const float* __restrict__ nb,
        // MINIMA: [tiles along the other axis][n][3]
        float* __restrict__ rowmin, float* __restrict__ colmin,
        // EMIT
        const float* __restrict__ tau_row, const float* __restrict__ tau_col,
        int* __restrict__ cand_row, int* __restrict__ cnt_row,
        int* __restrict__ cand_col, int* __restrict__ cnt_col, int cap,
        int with_cols_and_debug, const int tile_x, const int tile_y)
    {
      extern __shared__ __attribute__((aligned(16))) float lds[];
      const int with_cols = with_cols_and_debug & 1;
      // hi / lo panels of A and B for one chunk of 64 k (bf16, rows kPanelBytes apart)
      unsigned char* sAh = reinterpret_cast<unsigned char*>(lds);
      unsigned char* sAl = sAh + kTile * kPanelBytes;
      unsigned char* sBh = sAl + kTile * kPanelBytes;
      unsigned char* sBl = sBh + kTile * kPanelBytes;
      const unsigned short* Al = Ah + size_t(n1) * dimp;
      const unsigned short* Bl = Bh + size_t(n2) * dimp;
      const int tid = threadIdx.x;
      const int row0 = tile_y * kTile, col0 = tile_x * kTile;
      const int lane = tid & 63, wave = tid >> 6;
      const int wm = wave >> 1, wn = wave & 1;
      const int li = lane & 31, half = lane >> 5;
      f32x16 acc00 = {}, acc01 = {}, acc10 = {}, acc11 = {};

      for (int k0 = 0; k0 < dimp; k0 += kChunk)
      {
        if (k0 > 0)
          __syncthreads();  // the previous chunk's panels have been consumed
        // ---- stage the four panels of this chunk: 128 rows x 8 pieces of 16
        // bytes each, 4 pieces per thread and panel, every load in flight
        // before the first LDS write (rows past the end: zeros)
        uint4 v[16];
        const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int it = 0; it < 4; ++it)
        {
          const int idx = tid + it * 256, r = idx >> 3, c = idx & 7;
          const size_t oa = size_t(row0 + r) * dimp + k0 + 8 * c;
          const size_t ob = size_t(col0 + r) * dimp + k0 + 8 * c;
          const bool ra = row0 + r < n1, rb = col0 + r < n2;
          // explicitly global-memory loads: through a generic pointer (the
          // batched kernel's come out of an argument array) they are flat
          // loads that may address LDS as far as the compiler knows, and it
          // then orders every one of them behind the LDS writes of the
          // previous chunk instead of keeping all sixteen in flight
          auto gload = [](const unsigned short* p) -> uint4 {
            using G = const __attribute__((address_space(1))) unsigned*;
            const G g = (G) p;
            return make_uint4(g[0], g[1], g[2], g[3]);
          };
          v[it] = ra ? gload(Ah + oa) : zero;
          v[4 + it] = ra ? gload(Al + oa) : zero;
          v[8 + it] = rb ? gload(Bh + ob) : zero;
          v[12 + it] = rb ? gload(Bl + ob) : zero;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it)
        {
          const int idx = tid + it * 256, r = idx >> 3, c = idx & 7;
          const int o = r * kPanelBytes + 16 * c;
          *reinterpret_cast<uint4*>(sAh + o) = v[it];
          *reinterpret_cast<uint4*>(sAl + o) = v[4 + it];
          *reinterpret_cast<uint4*>(sBh + o) = v[8 + it];
          *reinterpret_cast<uint4*>(sBl + o) = v[12 + it];
        }
        __syncthreads();

        // ---- the contraction over this chunk: 4 steps of 16 k, three products
        // each (hi hi, hi lo, lo hi).  A lane supplies row / column (lane & 31)
        // and the 8 k of its half of the step; A and B use the same k, which
        // is all a dot product needs.
        const int ra0 = (wm * 64 + li) * kPanelBytes, ra1 = ra0 + 32 * kPanelBytes;
        const int rb0 = (wn * 64 + li) * kPanelBytes, rb1 = rb0 + 32 * kPanelBytes;
#pragma unroll
        for (int t = 0; t < 4; ++t)
        {
          const int ko = 2 * (16 * t + 8 * half);
          const bf16x8 a0h = *reinterpret_cast<const bf16x8*>(sAh + ra0 + ko);
          const bf16x8 a1h = *reinterpret_cast<const bf16x8*>(sAh + ra1 + ko);
          const bf16x8 a0l = *reinterpret_cast<const bf16x8*>(sAl + ra0 + ko);
          const bf16x8 a1l = *reinterpret_cast<const bf16x8*>(sAl + ra1 + ko);
          const bf16x8 b0h = *reinterpret_cast<const bf16x8*>(sBh + rb0 + ko);
          const bf16x8 b1h = *reinterpret_cast<const bf16x8*>(sBh + rb1 + ko);
          const bf16x8 b0l = *reinterpret_cast<const bf16x8*>(sBl + rb0 + ko);
          const bf16x8 b1l = *reinterpret_cast<const bf16x8*>(sBl + rb1 + ko);
#define SARA_MFMA_BF16(x, y)                                                   \
  acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0##x, b0##y, acc00, 0, 0, 0); \
  acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0##x, b1##y, acc01, 0, 0, 0); \
  acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1##x, b0##y, acc10, 0, 0, 0); \
  acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1##x, b1##y, acc11, 0, 0, 0);
          SARA_MFMA_BF16(h, h)
          SARA_MFMA_BF16(h, l)
          SARA_MFMA_BF16(l, h)
#undef SARA_MFMA_BF16
        }
      }
      __syncthreads();  // the panels are dead: their LDS is reused below

      // squared norms of this tile's rows / columns (>= FLT_MAX for the padding:
      // a row or column past the end can never be anyone's neighbour)
      float* sNa = lds + kTile * kDStride;  // behind the distance tile
      float* sNb = sNa + kTile;
      float* sTr = sNb + kTile;
      float* sTc = sTr + kTile;
      int* sQueue = reinterpret_cast<int*>(lds);  // EMIT: [0] = count, then hits
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
      if (MODE == kEmit && tid == 0)
        sQueue[0] = 0;
      __syncthreads();

      // C/D layout of the 32 x 32 blocks: lane -> column lane & 31, register r
      // -> row (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  A lane owns 2 columns
      // (one per block column) and 32 rows.  Padding rows / columns get +inf:
      // below no threshold (they are <= FLT_MAX), above every minimum.
      const int colA = wn * 64 + li, colB = colA + 32;
      const float nbA = sNb[colA], nbB = sNb[colB];
      const float inf = __builtin_huge_valf();
      auto approx = [&](float nrow, float ncol, float dot) -> float {
        const float s = nrow + ncol;  // >= FLT_MAX for the padding
        return s >= FLT_MAX ? inf : s - 2.f * dot;
      };
      auto row_of = [&](int bi, int r) -> int {
        return wm * 64 + bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      };

      if (MODE == kPacked)
      {
        float* sD = lds;  // [128][129]
#pragma unroll
        for (int r = 0; r < 16; ++r)
        {
          const int r0 = row_of(0, r), r1 = row_of(1, r);
          const float n0 = sNa[r0], n1r = sNa[r1];
          sD[r0 * kDStride + colA] = approx(n0, nbA, acc00[r]);
          sD[r0 * kDStride + colB] = approx(n0, nbB, acc01[r]);
          sD[r1 * kDStride + colA] = approx(n1r, nbA, acc10[r]);
          sD[r1 * kDStride + colB] = approx(n1r, nbB, acc11[r]);
        }
        __syncthreads();
        int k1 = INT_MAX, k2 = INT_MAX, k3 = INT_MAX, k4 = INT_MAX;
        if (tid < kTile)
        {
          const float* p = sD + tid * kDStride;
#pragma unroll 16
          for (int c = 0; c < kTile; ++c)
            min4_update((ordered_bits(p[c]) & ~127) | c, k1, k2, k3, k4);
          if (row0 + tid < n1)
            reinterpret_cast<int4*>(rowmin)[size_t(tile_x) * n1 + row0 + tid] =
                make_int4(k1, k2, k3, k4);
        }
        else if (with_cols)
        {
          const int c = tid - kTile;
          const float* p = sD + c;
#pragma unroll 16
          for (int r = 0; r < kTile; ++r)
            min4_update((ordered_bits(p[r * kDStride]) & ~127) | r, k1, k2, k3, k4);
          if (col0 + c < n2)
            reinterpret_cast<int4*>(colmin)[size_t(tile_y) * n2 + col0 + c] =
                make_int4(k1, k2, k3, k4);
        }
      }
      else if (MODE == kMinima)
      {
        float* sD = lds;  // [128][129]
#pragma unroll
        for (int r = 0; r < 16; ++r)
        {
          const int r0 = row_of(0, r), r1 = row_of(1, r);
          const float n0 = sNa[r0], n1r = sNa[r1];
          sD[r0 * kDStride + colA] = approx(n0, nbA, acc00[r]);
          sD[r0 * kDStride + colB] = approx(n0, nbB, acc01[r]);
          sD[r1 * kDStride + colA] = approx(n1r, nbA, acc10[r]);
          sD[r1 * kDStride + colB] = approx(n1r, nbB, acc11[r]);
        }
        __syncthreads();
        float m1 = FLT_MAX, m2 = FLT_MAX, m3 = FLT_MAX;
        if (tid < kTile)
        {
          const float* p = sD + tid * kDStride;
#pragma unroll 16
          for (int c = 0; c < kTile; ++c)
            min3_update(p[c], m1, m2, m3);
          if (row0 + tid < n1)
          {
            float* o = rowmin + (size_t(tile_x) * n1 + row0 + tid) * 3;
            o[0] = m1;
            o[1] = m2;
            o[2] = m3;
          }
        }
        else if (with_cols)
        {
          const int c = tid - kTile;
          const float* p = sD + c;
#pragma unroll 16
          for (int r = 0; r < kTile; ++r)
            min3_update(p[r * kDStride], m1, m2, m3);
          if (col0 + c < n2)
          {
            float* o = colmin + (size_t(tile_y) * n2 + col0 + c) * 3;
            o[0] = m1;
            o[1] = m2;
            o[2] = m3;
          }
        }
      }
      else
      {
        // Hits are rare (about three per query in the whole matrix) but a
        // returning global atomic under a divergent branch costs the whole wave
        // hundreds of cycles each time: they are queued in LDS (row, column,
        // which list) and the workgroup claims their slots together at the end.
        const float tcA = sTc[colA], tcB = sTc[colB];
        auto claim = [&](int row, int col, int which) {
          if (which == 0)
          {
            const int q = row0 + row;
            const int slot = atomicAdd(cnt_row + q, 1);
            if (slot < cap)
              cand_row[size_t(q) * cap + slot] = col0 + col;
          }
          else
          {
            const int q = col0 + col;
            const int slot = atomicAdd(cnt_col + q, 1);
            if (slot < cap)
              cand_col[size_t(q) * cap + slot] = row0 + row;
          }
        };
        auto hit = [&](float v, float trow, float tcol, int row, int col) {
          if (v <= trow)
          {
            const int e = atomicAdd(&sQueue[0], 1);
            if (e < kQueue)
              sQueue[1 + e] = (row << 8) | col;
            else
              claim(row, col, 0);
          }
          if (v <= tcol)
          {
            const int e = atomicAdd(&sQueue[0], 1);
            if (e < kQueue)
              sQueue[1 + e] = (1 << 16) | (row << 8) | col;
            else
              claim(row, col, 1);
          }
        };
#pragma unroll
        for (int r = 0; r < 16; ++r)
        {
          const int r0 = row_of(0, r), r1 = row_of(1, r);
          const float n0 = sNa[r0], n1r = sNa[r1];
          const float t0 = sTr[r0], t1 = sTr[r1];
          const float v00 = approx(n0, nbA, acc00[r]), v01 = approx(n0, nbB, acc01[r]);
          const float v10 = approx(n1r, nbA, acc10[r]), v11 = approx(n1r, nbB, acc11[r]);
          // one test per register quad before the four exact ones
          const float lo0 = fminf(v00, v01), lo1 = fminf(v10, v11);
          if (lo0 <= t0 || lo1 <= t1 || fminf(v00, v10) <= tcA || fminf(v01, v11) <= tcB)
          {
            hit(v00, t0, tcA, r0, colA);
            hit(v01, t0, tcB, r0, colB);
            hit(v10, t1, tcA, r1, colA);
            hit(v11, t1, tcB, r1, colB);
          }
        }
        __syncthreads();
        const int queued = min(sQueue[0], kQueue);
        for (int e = tid; e < queued; e += 256)
        {
          const int w = sQueue[1 + e];
          claim((w >> 8) & 255, w & 255, w >> 16);
        }
      }
    }

    //! Global three smallest approximations of every query -> its threshold.
    //! top1: rank of the best real neighbour (1 when a set is matched against
    //! itself: rank 0 is the query).
    template <int MODE>
    __global__ __launch_bounds__(256, 2) void mfma_tiles_kernel(
        const unsigned short* __restrict__ Ah, int n1,
        const unsigned short* __restrict__ Bh, int n2, int dimp,
        const float* __restrict__ na, const float* __restrict__ nb,
        float* __restrict__ rowmin, float* __restrict__ colmin,
        const float* __restrict__ tau_row, const float* __restrict__ tau_col,
        int* __restrict__ cand_row, int* __restrict__ cnt_row,
        int* __restrict__ cand_col, int* __restrict__ cnt_col, int cap,
        int with_cols_and_debug)
    {
      mfma_tile<MODE>(Ah, n1, Bh, n2, dimp, na, nb, rowmin, colmin, tau_row, tau_col,
                      cand_row, cnt_row, cand_col, cnt_col, cap, with_cols_and_debug,
                      blockIdx.x, blockIdx.y);
    }

    //! ONE tile grid over a batch of pairs: blockIdx.z = pair; the grid covers
    //! the largest pair, the surplus workgroups of the others leave.  The pairs'
    //! pointers travel in the kernel's argument segment (scalar loads, like the
    //! single kernel's own arguments), at most kTileBatchPairs per launch.
    constexpr int kTileBatchPairs = 32;
    struct TileBatchArgs
    {
      const unsigned short* Ah[kTileBatchPairs];
      const unsigned short* Bh[kTileBatchPairs];
      const float* na[kTileBatchPairs];
      const float* nb[kTileBatchPairs];
      float* rowmin[kTileBatchPairs];
      float* colmin[kTileBatchPairs];
      int n1[kTileBatchPairs];
      int n2[kTileBatchPairs];
    };
    template <int MODE>
    __global__ __launch_bounds__(256, 2) void mfma_tiles_batch_kernel(TileBatchArgs a,
                                                                     int dimp, int cap)
    {
      const int z = blockIdx.z;
      const int n1 = a.n1[z], n2 = a.n2[z];
      if (int(blockIdx.y) * kTile >= n1 || int(blockIdx.x) * kTile >= n2)
        return;  // the whole workgroup
      mfma_tile<MODE>(a.Ah[z], n1, a.Bh[z], n2, dimp, a.na[z], a.nb[z], a.rowmin[z],
                      a.colmin[z], nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                      cap, 1, blockIdx.x, blockIdx.y);
    }

    //! blockIdx.y = direction (rows / columns of the tiles): one launch.
    struct ThresholdDir
    {
      const float* partial;
      int ntiles, n;
      const float* norms;
      const unsigned* other_max_bits;
      float* tau;
    };
    struct ThresholdArgs
    {
      ThresholdDir d[2];
    };
    __global__ void thresholds_kernel(ThresholdArgs args, int dim,
                                      float squared_ratio_thres, int top1)
    {
      const ThresholdDir& a = args.d[blockIdx.y];
      const float* __restrict__ partial = a.partial;
      const int ntiles = a.ntiles, n = a.n;
      const float* __restrict__ norms = a.norms;
      const unsigned* __restrict__ other_max_bits = a.other_max_bits;
      float* __restrict__ tau = a.tau;
      const int i = blockIdx.x * blockDim.x + threadIdx.x;
      if (i >= n)
        return;
      float m1 = FLT_MAX, m2 = FLT_MAX, m3 = FLT_MAX;
#pragma unroll 4
      for (int t = 0; t < ntiles; ++t)
      {
        const float* p = partial + (size_t(t) * n + i) * 3;
        min3_update(p[0], m1, m2, m3);
        min3_update(p[1], m1, m2, m3);
        min3_update(p[2], m1, m2, m3);
      }
      const float e = guard_coeff(dim) * (norms[i] + __uint_as_float(*other_max_bits));
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

    //! kPacked: the candidates of every query straight from the tiles' four
    //! smallest keys.  m3 = third smallest value over all tiles (read back from
    //! the keys: <= the true third smallest approximation, within 2^-16), tau =
    //! m3 + 2e-4 |m3| + 2.01 E >= the tau of thresholds_kernel, and a key passes
    //! when its value is <= tau - every approximation <= tau does, because a
    //! key's value never exceeds its approximation.  A tile lists all of its
    //! entries below tau unless its FOURTH key passes too: then a fifth might,
    //! and the query is handed to the exhaustive fall-back (count > cap).
    //! 16 lanes per query (a thread per query walked the 34 tiles of a 4.3 k set
    //! twice, 68 dependent-latency loads on 17 workgroups: 23 us per direction):
    //! a lane takes the tiles t = lane, lane + 16, ..; the three smallest keys
    //! are merged across the 16 lanes by xor shuffles, every lane then emits
    //! the passing keys of its own tiles (slots claimed with atomicAdd on the
    //! query's counter, which the launch zeroes).
    template <bool BATCH = false>
    __global__ __launch_bounds__(256) void select_packed_kernel(
        const int4* __restrict__ partial, int ntiles, int n,
        const float* __restrict__ norms, const unsigned* __restrict__ other_max_bits,
        int dim, int cap, int* __restrict__ cand, int* __restrict__ cnt,
        const MatchBatchPair* __restrict__ batch)
    {
      if constexpr (BATCH)  // blockIdx.y = 2 * pair + direction
      {
        const MatchBatchPair& b = batch[blockIdx.y >> 1];
        const bool cols = (blockIdx.y & 1) != 0;
        partial = reinterpret_cast<const int4*>(uni(cols ? b.colmin : b.rowmin));
        ntiles = (uni(cols ? b.n1 : b.n2) + kTile - 1) / kTile;
        n = uni(cols ? b.n2 : b.n1);
        norms = uni(cols ? b.nb : b.na);
        other_max_bits = uni(cols ? b.maxbits : b.maxbits + 1);
        cand = uni(cols ? b.cand_c : b.cand_r);
        cnt = uni(cols ? b.cnt_c : b.cnt_r);
      }
      const int gid = blockIdx.x * blockDim.x + threadIdx.x;
      const int i = gid >> 4, sub = gid & 15;
      const bool live = i < n;
      int m1 = INT_MAX, m2 = INT_MAX, m3 = INT_MAX;
      auto take = [&](int v) {
        m3 = med3i(m2, m3, v);
        m2 = med3i(m1, m2, v);
        m1 = min(m1, v);
      };
      if (live)
        for (int t = sub; t < ntiles; t += 16)
        {
          const int4 k = partial[size_t(t) * n + i];
          if (k.x < kNoKey) take(k.x);
          if (k.y < kNoKey) take(k.y);
          if (k.z < kNoKey) take(k.z);
          if (k.w < kNoKey) take(k.w);
        }
      // the three smallest of the 16 lanes' triples (all 16 lanes of a group
      // are in one wave and take part, live or not)
#pragma unroll
      for (int o = 1; o < 16; o <<= 1)
      {
        const int a = __shfl_xor(m1, o), b = __shfl_xor(m2, o), c = __shfl_xor(m3, o);
        take(a);
        take(b);
        take(c);
      }
      if (!live)
        return;
      const float e = guard_coeff(dim) * (norms[i] + __uint_as_float(*other_max_bits));
      int tau_key = kNoKey - 1;  // fewer than three candidates: everything passes
      if (m3 != INT_MAX)
      {
        const float m = from_ordered_bits(m3 & ~127);
        const float tau = m + fabsf(m) * 2e-4f + 2.01f * e;
        if (tau < FLT_MAX)
          tau_key = min(ordered_bits(tau) | 127, kNoKey - 1);
      }
      for (int t = sub; t < ntiles; t += 16)
      {
        const int4 k = partial[size_t(t) * n + i];
        const int v[4] = {k.x, k.y, k.z, k.w};
        int pass = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          pass += v[r] <= tau_key ? 1 : 0;  // the keys of a tile are ascending
        if (pass == 0)
          continue;
        // a passing fourth key: a fifth entry of the tile might pass as well
        const int claim = pass == 4 ? cap + 1 : pass;
        const int at = atomicAdd(cnt + i, claim);
        for (int r = 0; r < pass && r < 3; ++r)
          if (at + r < cap)
            cand[size_t(i) * cap + at + r] = t * kTile + (v[r] & 127);
      }
    }

    __device__ __forceinline__ int wave_max_int(int v)
    {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
        v = max(v, __shfl_xor(v, o));
      return v;
    }

    //! Exact distances of every query's candidate slots; CAP lanes per query.
    //! blockIdx.y = direction: one launch for both.
    struct RerankDir
    {
      const float* q;
      int nq;
      const float* t;
      const int* cand;
      const int* cnt;
      float* top_d;
      int* top_i;
      MatchNeighbour* radius_out;
      int radius_cap;
      int* radius_count;
      int* flagged;
      int* flagged_count;
    };
    struct RerankArgs
    {
      RerankDir d[2];
    };
    template <int CAP, bool BATCH = false>
    __global__ __launch_bounds__(256) void rerank_kernel(
        RerankArgs args, int dim, float squared_ratio_thres, int top1,
        const MatchBatchPair* __restrict__ batch)
    {
      const float* __restrict__ q;
      int nq;
      const float* __restrict__ t;
      const int* __restrict__ cand;
      const int* __restrict__ cnt;
      float* __restrict__ top_d;
      int* __restrict__ top_i;
      MatchNeighbour* __restrict__ radius_out;
      int radius_cap;
      int* __restrict__ radius_count;
      int* __restrict__ flagged;
      int* __restrict__ flagged_count;
      if constexpr (BATCH)  // pair blockIdx.z, direction blockIdx.y
      {
        const MatchBatchPair& b = batch[blockIdx.z];
        const bool cols = blockIdx.y != 0;
        const int n1 = uni(b.n1);
        q = uni(cols ? b.d2 : b.d1);
        nq = cols ? uni(b.n2) : n1;
        t = uni(cols ? b.d1 : b.d2);
        cand = uni(cols ? b.cand_c : b.cand_r);
        cnt = uni(cols ? b.cnt_c : b.cnt_r);
        top_d = uni(b.top_d) + (cols ? 3 * size_t(n1) : 0);
        top_i = uni(b.top_i) + (cols ? 3 * size_t(n1) : 0);
        radius_out = nullptr;
        radius_cap = 0;
        radius_count = nullptr;
        flagged = uni(cols ? b.flag_c : b.flag_r);
        flagged_count = uni(b.scal) + (cols ? 1 : 0);
      }
      else
      {
        const RerankDir& a = args.d[blockIdx.y];
        q = a.q;
        nq = a.nq;
        t = a.t;
        cand = a.cand;
        cnt = a.cnt;
        top_d = a.top_d;
        top_i = a.top_i;
        radius_out = a.radius_out;
        radius_cap = a.radius_cap;
        radius_count = a.radius_count;
        flagged = a.flagged;
        flagged_count = a.flagged_count;
      }
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
      // rank by (distance, index) inside the group; d of rank top1.  Slots
      // from the group's count on are empty (FLT_MAX, INT_MAX: they precede
      // nothing), so the loops stop at the fullest group of the wave - a query
      // has 3-4 candidates, CAP is 32 for the radius search.
      int rank = 0;
      float d_top1 = FLT_MAX;
      const int used = __builtin_amdgcn_readfirstlane(
          wave_max_int(valid ? min(c, CAP) : 0));
      // first the ranks ...
      for (int k = 0; k < used; ++k)
      {
        const float od = __shfl(d, base + k);
        const int oi = __shfl(idx, base + k);
        rank += (od < d || (od == d && oi < idx)) ? 1 : 0;
      }
      // ... then the distance the radius is built from
      for (int k = 0; k < used; ++k)
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

    __device__ inline bool closer(float d, int i, float od, int oi)
    {
      return d < od || (d == od && i < oi);
    }
    __device__ inline void top3_insert_ordered(float d, int j, float (&b)[3],
                                               int (&bi)[3])
    {
      if (closer(d, j, b[0], bi[0]))
      {
        b[2] = b[1]; bi[2] = bi[1]; b[1] = b[0]; bi[1] = bi[0]; b[0] = d; bi[0] = j;
      }
      else if (closer(d, j, b[1], bi[1]))
      {
        b[2] = b[1]; bi[2] = bi[1]; b[1] = d; bi[1] = j;
      }
      else if (closer(d, j, b[2], bi[2]))
      {
        b[2] = d; bi[2] = j;
      }
    }

    //! Round 3: the exact distances of the first kFallbackSlots flagged queries
    //! to EVERY target, kFallbackParts workgroups per query, into `staged`
    //! ([slot][nt]).  The fall-back kernel below then selects from them instead
    //! of computing 4-5 distances per thread one after the other, twice (top-3
    //! pass and radius pass): with 33 flagged queries of 8 600 it ran on 33
    //! workgroups for 130 us per direction - a latency chain, not work.
    //! Both directions in one launch (blockIdx.z): they are independent, and a
    //! fall-back is a latency chain on a few dozen workgroups - run one after
    //! the other the two cost 120 us of the default-ratio call's 350.
    struct FallbackDir
    {
      const float* q;
      int nq;
      const float* t;
      int nt;
      const int* flagged;
      const int* flagged_count;
      float* top_d;
      int* top_i;
      MatchNeighbour* radius_out;
      int radius_cap;
      int* radius_count;
      float* staged;  // [kFallbackSlots][nt], or NULL
    };
    struct FallbackArgs
    {
      FallbackDir d[2];
    };

    __global__ __launch_bounds__(256) void fallback_distances_kernel(FallbackArgs args,
                                                                     int dim)
    {
      const FallbackDir& a = args.d[blockIdx.z];
      const float* __restrict__ q = a.q;
      const float* __restrict__ t = a.t;
      const int nt = a.nt;
      const int* __restrict__ flagged = a.flagged;
      float* __restrict__ staged = a.staged;
      const int n = min(*a.flagged_count, kFallbackSlots);
      const int per = (nt + kFallbackParts - 1) / kFallbackParts;
      for (int k = blockIdx.y; k < n; k += gridDim.y)
      {
        const float* qr = q + size_t(flagged[k]) * dim;
        const int lo = blockIdx.x * per, hi = min(nt, lo + per);
        for (int j = lo + threadIdx.x; j < hi; j += 256)
          staged[size_t(k) * nt + j] = flann_l2_rows(qr, t + size_t(j) * dim, dim);
      }
    }

    //! Queries whose candidate slots overflowed: exhaustive search, one
    //! 1024-thread workgroup per query (a single wave per query spent 0.3 ms on
    //! its 67 dependent row reads per lane).
    template <bool BATCH = false>
    __global__ __launch_bounds__(1024) void fallback_kernel(
        FallbackArgs args, int dim, float squared_ratio_thres, int top1,
        const MatchBatchPair* __restrict__ batch)
    {
      const float* __restrict__ q;
      int nq;
      const float* __restrict__ t;
      int nt;
      const int* __restrict__ flagged;
      const int* __restrict__ flagged_count;
      float* __restrict__ top_d;
      int* __restrict__ top_i;
      MatchNeighbour* __restrict__ radius_out;
      int radius_cap;
      int* __restrict__ radius_count;
      const float* __restrict__ staged;
      if constexpr (BATCH)  // pair blockIdx.y, direction blockIdx.z
      {
        const MatchBatchPair& b = batch[blockIdx.y];
        const bool cols = blockIdx.z != 0;
        const int n1 = uni(b.n1), n2 = uni(b.n2);
        q = uni(cols ? b.d2 : b.d1);
        nq = cols ? n2 : n1;
        t = uni(cols ? b.d1 : b.d2);
        nt = cols ? n1 : n2;
        flagged = uni(cols ? b.flag_c : b.flag_r);
        flagged_count = uni(b.scal) + (cols ? 1 : 0);
        top_d = uni(b.top_d) + (cols ? 3 * size_t(n1) : 0);
        top_i = uni(b.top_i) + (cols ? 3 * size_t(n1) : 0);
        radius_out = nullptr;
        radius_cap = 0;
        radius_count = nullptr;
        staged = nullptr;
      }
      else
      {
        const FallbackDir& a = args.d[blockIdx.z];
        q = a.q;
        nq = a.nq;
        t = a.t;
        nt = a.nt;
        flagged = a.flagged;
        flagged_count = a.flagged_count;
        top_d = a.top_d;
        top_i = a.top_i;
        radius_out = a.radius_out;
        radius_cap = a.radius_cap;
        radius_count = a.radius_count;
        staged = a.staged;
      }
      __shared__ float s_d[1024 * 3];
      __shared__ int s_i[1024 * 3];
      __shared__ float s_radius;
      const int tid = threadIdx.x;
      const int n = *flagged_count;
      for (int k = blockIdx.x; k < n; k += gridDim.x)
      {
        const int qi = flagged[k];
        const float* qr = q + size_t(qi) * dim;
        // distances staged by fallback_distances_kernel (same function, same
        // floats), or computed here for the queries beyond its slots
        const float* sd = (staged && k < kFallbackSlots) ? staged + size_t(k) * nt : nullptr;
        float b[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        int bi[3] = {INT_MAX, INT_MAX, INT_MAX};
        for (int j = tid; j < nt; j += 1024)
          top3_insert_ordered(sd ? sd[j] : flann_l2_rows(qr, t + size_t(j) * dim, dim), j,
                              b, bi);
        for (int r = 0; r < 3; ++r)
        {
          s_d[tid * 3 + r] = b[r];
          s_i[tid * 3 + r] = bi[r];
        }
        __syncthreads();
        if (tid < 64)
        {
          float c[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
          int ci[3] = {INT_MAX, INT_MAX, INT_MAX};
          for (int e = tid; e < 1024 * 3; e += 64)
            if (s_i[e] != INT_MAX)
              top3_insert_ordered(s_d[e], s_i[e], c, ci);
          // three rounds: the wave's smallest (distance, index), popped from
          // the lane that holds it
          float d_top1 = FLT_MAX;
          for (int r = 0; r < 3; ++r)
          {
            float md = c[0];
            int mi = ci[0];
            for (int o = 32; o > 0; o >>= 1)
            {
              const float od = __shfl_xor(md, o);
              const int oi = __shfl_xor(mi, o);
              if (closer(od, oi, md, mi))
              {
                md = od;
                mi = oi;
              }
            }
            if (ci[0] == mi && mi != INT_MAX)
            {
              c[0] = c[1]; ci[0] = ci[1]; c[1] = c[2]; ci[1] = ci[2];
              c[2] = FLT_MAX; ci[2] = INT_MAX;
            }
            if (tid == 0)
            {
              top_d[size_t(r) * nq + qi] = mi == INT_MAX ? FLT_MAX : md;
              top_i[size_t(r) * nq + qi] = mi == INT_MAX ? -1 : mi;
            }
            if (r == top1 && mi != INT_MAX)
              d_top1 = md;
          }
          if (tid == 0)
            s_radius = d_top1 < FLT_MAX ? d_top1 * squared_ratio_thres : -1.f;
        }
        __syncthreads();
        const float radius = s_radius;
        if (squared_ratio_thres > 1.f && radius >= 0.f)
          for (int j = tid; j < nt; j += 1024)
          {
            const float d = sd ? sd[j] : flann_l2_rows(qr, t + size_t(j) * dim, dim);
            if (d < radius)
            {
              const int o = atomicAdd(radius_count, 1);
              if (o < radius_cap)
                radius_out[o] = MatchNeighbour{qi, j, d};
            }
          }
        __syncthreads();
      }
    }

    //! Dynamic LDS above 64 KB has to be allowed per kernel and per device
    //! (WHICH keeps one table per kernel: both instantiations share a type).
    template <int WHICH, typename K>
    void allow_big_lds(K kernel)
    {
      static std::atomic<bool> done[64];
      int dev = 0;
      (void) hipGetDevice(&dev);
      if (!done[dev & 63].load(std::memory_order_acquire))
      {
        (void) hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024 - 64);
        done[dev & 63].store(true, std::memory_order_release);
      }
    }

  }  // namespace

  size_t match_mfma_scratch_floats(int n1, int n2)
  {
    const size_t tm = (size_t(n1) + kTile - 1) / kTile, tn = (size_t(n2) + kTile - 1) / kTile;
    // norms (n1 + n2 + 2), tau (n1 + n2), row minima [tn][n1][3], column minima [tm][n2][3]
    return 2 * (size_t(n1) + n2) + 16 + 4 * (tn * n1 + tm * n2) + 8 +
           2 * size_t(kFallbackSlots) * size_t(std::max(n1, n2)) +  // staged distances
           8 + (size_t(n1) + n2) * kMaxDimPadded;  // hi / lo bf16 rows (2 x 2 bytes per k)
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
                         hipStream_t stream, const ZeroRanges* also_clear)
  {
    const int tm = (n1 + kTile - 1) / kTile, tn = (n2 + kTile - 1) / kTile;
    // ---- carve the scratch
    float* na = fscratch;
    float* nb = na + n1;
    unsigned* maxbits = reinterpret_cast<unsigned*>(nb + n2);  // [0] of A, [1] of B
    float* tau_r = nb + n2 + 16;
    float* tau_c = tau_r + n1;
    // minima: [tiles][n][3] floats, or [tiles][n] int4 keys (16-byte aligned)
    float* rowmin = tau_c + n2;
    rowmin += (4 - ((rowmin - fscratch) & 3)) & 3;
    float* colmin = rowmin + 4 * size_t(tn) * n1;
    float* staged = colmin + 4 * size_t(tm) * n2;  // [kFallbackSlots][max(n1, n2)]
    // the keys as bf16 hi / lo rows, zero-padded to dimp: [hi n x dimp][lo n x dimp]
    const int dimp = (dim + kChunk - 1) / kChunk * kChunk;
    float* split_f = staged + 2 * size_t(kFallbackSlots) * size_t(std::max(n1, n2));
    split_f += (4 - ((split_f - fscratch) & 3)) & 3;  // 16-byte aligned
    unsigned short* split1 = reinterpret_cast<unsigned short*>(split_f);
    unsigned short* split2 = split1 + 2 * size_t(n1) * dimp;
    int* cnt_r = iscratch;
    int* cnt_c = cnt_r + n1;
    int* scal = cnt_c + n2;  // [0] flagged rows, [1] flagged columns
    int* flag_r = scal + 16;
    int* flag_c = flag_r + n1;
    int* cand_r = flag_c + n2;
    int* cand_c = cand_r + size_t(n1) * cap;
    static const bool prof = getenv("SARA_HIP_MATCH_PROF") != nullptr;
    static hipEvent_t pev[12];
    static bool pev_made = false;
    int pk = 0;
    auto tick = [&]() {
      if (!prof)
        return;
      if (!pev_made)
      {
        for (auto& e : pev)
          (void) hipEventCreate(&e);
        pev_made = true;
      }
      (void) hipEventRecord(pev[pk++], stream);
    };
    tick();
    {
      // one launch clears what this search and - when the caller says so -
      // its tail expect to be zero
      ZeroRanges z;
      if (also_clear)
        z = *also_clear;
      z.add(maxbits, 16);
      z.add(cnt_r, size_t(n1) + n2 + 16);
      launch_zero_ranges(z, stream);
    }
    hipLaunchKernelGGL(row_norms_kernel<false>, dim3((n1 + n2 + 63) / 64), dim3(1024), 0,
                       stream, d1, n1, d2, n2, dim, na, nb, maxbits, split1, split2, dimp, nullptr);
    // panels of one chunk, or the distance tile + norms / thresholds (minima),
    // or the hit queue in front of the norms / thresholds (emit: the queue must
    // end before the norm arrays start at kTile * kDStride floats)
    static_assert(1 + kQueue <= kTile * kDStride, "the queue overlaps the norms");
    tick();  // 1: memsets + norms
    const size_t lds = std::max(size_t(4) * kTile * kPanelBytes,
                                sizeof(float) * (kTile * kDStride + 4 * kTile));
    allow_big_lds<kMinima>(mfma_tiles_kernel<kMinima>);
    allow_big_lds<kEmit>(mfma_tiles_kernel<kEmit>);
    const dim3 grid(tn, tm);
    const int cols_arg = with_dir1 ? 1 : 0;
    // Round 3, ratios <= 1 (only the three nearest neighbours matter): ONE pass
    // over the tiles that keeps, per row and column of a tile, the four smallest
    // approximations together with where they are (kPacked), and the candidates
    // are picked from those lists - the second contraction (emit, 82 us of the
    // 0.28 ms per 4.3 k x 4.3 k pair) is gone.  SARA_HIP_MATCH_PASSES=2 keeps the
    // two passes; the radius search (ratios > 1) always takes them: a tile can
    // hold any number of radius members.
    static const bool two_passes_env = [] {
      const char* e = getenv("SARA_HIP_MATCH_PASSES");
      return e && atoi(e) == 2;
    }();
    const bool one_pass = !(squared_ratio_thres > 1.f) && top1 == 0 && !two_passes_env;
    if (one_pass)
    {
      allow_big_lds<kPacked>(mfma_tiles_kernel<kPacked>);
      hipLaunchKernelGGL(mfma_tiles_kernel<kPacked>, grid, dim3(256), lds, stream, split1,
                         n1, split2, n2, dimp, na, nb, rowmin, colmin, nullptr, nullptr,
                         nullptr, nullptr, nullptr, nullptr, cap, cols_arg);
      tick();  // 2: minima
      hipLaunchKernelGGL(select_packed_kernel<false>, dim3((n1 + 15) / 16), dim3(256), 0,
                         stream, reinterpret_cast<const int4*>(rowmin), tn, n1, na,
                         maxbits + 1, dim, cap, cand_r, cnt_r, nullptr);
      if (with_dir1)
        hipLaunchKernelGGL(select_packed_kernel<false>, dim3((n2 + 15) / 16), dim3(256), 0,
                           stream, reinterpret_cast<const int4*>(colmin), tm, n2, nb,
                           maxbits, dim, cap, cand_c, cnt_c, nullptr);
      tick();  // 3: thresholds
      tick();  // 4: emit (none)
    }
    else
    {
    hipLaunchKernelGGL(mfma_tiles_kernel<kMinima>, grid, dim3(256), lds, stream, split1,
                       n1, split2, n2, dimp, na, nb, rowmin, colmin, nullptr, nullptr,
                       nullptr, nullptr, nullptr, nullptr, cap, cols_arg);
    tick();  // 2: minima
    {
      ThresholdArgs ta;
      ta.d[0] = ThresholdDir{rowmin, tn, n1, na, maxbits + 1, tau_r};
      ta.d[1] = ThresholdDir{colmin, tm, n2, nb, maxbits, tau_c};
      const int nmax = std::max(n1, with_dir1 ? n2 : 0);
      hipLaunchKernelGGL(thresholds_kernel, dim3((nmax + 255) / 256, with_dir1 ? 2 : 1),
                         dim3(256), 0, stream, ta, dim, squared_ratio_thres, top1);
    }
    tick();  // 3: thresholds
    hipLaunchKernelGGL(mfma_tiles_kernel<kEmit>, grid, dim3(256), lds, stream, split1, n1,
                       split2, n2, dimp, na, nb, nullptr, nullptr, tau_r, tau_c, cand_r,
                       cnt_r, cand_c, cnt_c, cap, cols_arg);
    tick();  // 4: emit
    }
    {
      RerankArgs ra;
      ra.d[0] = RerankDir{d1, n1, d2, cand_r, cnt_r, top12_d, top12_i, radius12,
                          radius12_cap, radius12_count, flag_r, scal};
      ra.d[1] = RerankDir{d2, n2, d1, cand_c, cnt_c, top21_d, top21_i, radius21,
                          radius21_cap, radius21_count, flag_c, scal + 1};
      const size_t threads = size_t(std::max(n1, with_dir1 ? n2 : 0)) * cap;
      const dim3 g(unsigned((threads + 255) / 256), with_dir1 ? 2 : 1);
      if (cap == 8)
        hipLaunchKernelGGL(rerank_kernel<8>, g, dim3(256), 0, stream, ra, dim,
                           squared_ratio_thres, top1, nullptr);
      else if (cap == 64)
        hipLaunchKernelGGL(rerank_kernel<64>, g, dim3(256), 0, stream, ra, dim,
                           squared_ratio_thres, top1, nullptr);
      else
        hipLaunchKernelGGL(rerank_kernel<32>, g, dim3(256), 0, stream, ra, dim,
                           squared_ratio_thres, top1, nullptr);
    }
    {
      // Queries whose candidate slots overflowed.  Their distances are staged
      // only for the radius search, where overflowing queries are the rule
      // (ratio <= 1: a handful of candidates per query, no query flagged on the
      // benchmark pair - not worth a launch).
      const bool stage = squared_ratio_thres > 1.f;
      FallbackArgs fa;
      fa.d[0] = FallbackDir{d1, n1, d2, n2, flag_r, scal, top12_d, top12_i, radius12,
                            radius12_cap, radius12_count, stage ? staged : nullptr};
      fa.d[1] = FallbackDir{d2, n2, d1, n1, flag_c, scal + 1, top21_d, top21_i,
                            radius21, radius21_cap, radius21_count,
                            stage ? staged + size_t(kFallbackSlots) * std::max(n1, n2)
                                  : nullptr};
      const int ndir = with_dir1 ? 2 : 1;
      const int nq_max = std::max(n1, with_dir1 ? n2 : 0);
      if (stage)
        hipLaunchKernelGGL(fallback_distances_kernel,
                           dim3(kFallbackParts, std::min(nq_max, kFallbackSlots), ndir),
                           dim3(256), 0, stream, fa, dim);
      hipLaunchKernelGGL(fallback_kernel<false>, dim3(std::min(nq_max, 256), 1, ndir),
                         dim3(1024), 0, stream, fa, dim, squared_ratio_thres, top1,
                         nullptr);
    }
    tick();  // 5: rerank + fallback
    if (prof)
    {
      (void) hipStreamSynchronize(stream);
      const char* names[] = {"norms", "minima", "thresholds", "emit", "rerank"};
      for (int k = 0; k + 1 < pk; ++k)
      {
        float ms = 0.f;
        (void) hipEventElapsedTime(&ms, pev[k], pev[k + 1]);
        std::fprintf(stderr, "[match prof]   %-10s %8.1f us\n", names[k], 1e3 * ms);
      }
    }
  }


  // ---- a batch of pairs in one set of launches (round 6) -----------------------
  namespace {
    template <typename T>
    T* carve(unsigned char* base, size_t& at, size_t count)
    {
      at = (at + 15) & ~size_t(15);
      T* p = base ? reinterpret_cast<T*>(base + at) : nullptr;
      at += count * sizeof(T);
      return p;
    }
  }  // namespace

  void match_batch_carve(int n1, int n2, int dim, unsigned char* zero_base,
                         unsigned char* work_base, unsigned char* out_base,
                         MatchBatchLayout* at, MatchBatchPair* pair)
  {
    const int tm = (n1 + kTile - 1) / kTile, tn = (n2 + kTile - 1) / kTile;
    const int dimp = (dim + kChunk - 1) / kChunk * kChunk;
    const size_t n = size_t(n1) + n2;
    MatchBatchPair p{};
    p.n1 = n1;
    p.n2 = n2;
    // cleared at the head of the batch: maxima, candidate counters, flag
    // counters, the list header and the rank sort's counters
    p.maxbits = carve<unsigned>(zero_base, at->zero_bytes, 16);
    p.cnt_r = carve<int>(zero_base, at->zero_bytes, n + 16);
    p.cnt_c = p.cnt_r ? p.cnt_r + n1 : nullptr;
    p.scal = p.cnt_r ? p.cnt_r + n : nullptr;
    p.rank = carve<int>(zero_base, at->zero_bytes, n);
    // work
    p.na = carve<float>(work_base, at->work_bytes, n);
    p.nb = p.na ? p.na + n1 : nullptr;
    p.rowmin = carve<float>(work_base, at->work_bytes, 4 * size_t(tn) * n1);
    p.colmin = carve<float>(work_base, at->work_bytes, 4 * size_t(tm) * n2);
    p.split1 = carve<unsigned short>(work_base, at->work_bytes, 2 * size_t(n1) * dimp);
    p.split2 = carve<unsigned short>(work_base, at->work_bytes, 2 * size_t(n2) * dimp);
    p.flag_r = carve<int>(work_base, at->work_bytes, n);
    p.flag_c = p.flag_r ? p.flag_r + n1 : nullptr;
    p.cand_r = carve<int>(work_base, at->work_bytes, n * kMatchBatchCap);
    p.cand_c = p.cand_r ? p.cand_r + size_t(n1) * kMatchBatchCap : nullptr;
    p.top_d = carve<float>(work_base, at->work_bytes, 3 * n);
    p.top_i = carve<int>(work_base, at->work_bytes, 3 * n);
    p.tmp = carve<sara_match>(work_base, at->work_bytes, n);
    // read back: header (cleared by its own tiny range: it sits in the out
    // arena so that ONE copy brings every list home) + sorted list
    p.header = carve<int>(out_base, at->out_bytes, 4);
    p.out = carve<sara_match>(out_base, at->out_bytes, n);
    if (pair)
    {
      p.d1 = pair->d1;
      p.d2 = pair->d2;
      *pair = p;
    }
  }

  __global__ void zero_headers_kernel(const MatchBatchPair* __restrict__ batch)
  {
    if (threadIdx.x < 4)
      batch[blockIdx.x].header[threadIdx.x] = 0;
  }

  void launch_match_batch(const MatchBatchPair* table, const MatchBatchPair* host_table,
                          int n_pairs, int n1_max, int n2_max, int dim,
                          float squared_ratio_thres, void* zero, size_t zero_bytes,
                          hipStream_t stream)
  {
    const int dimp = (dim + kChunk - 1) / kChunk * kChunk;
    const int tm = (n1_max + kTile - 1) / kTile, tn = (n2_max + kTile - 1) / kTile;
    const int cap = kMatchBatchCap;
    {
      ZeroRanges z;
      z.add(zero, zero_bytes / sizeof(int));
      launch_zero_ranges(z, stream);
      hipLaunchKernelGGL(zero_headers_kernel, dim3(n_pairs), dim3(64), 0, stream, table);
    }
    hipLaunchKernelGGL(row_norms_kernel<true>, dim3((n1_max + n2_max + 63) / 64, n_pairs),
                       dim3(1024), 0, stream, nullptr, 0, nullptr, 0, dim, nullptr,
                       nullptr, nullptr, nullptr, nullptr, dimp, table);
    const size_t lds = std::max(size_t(4) * kTile * kPanelBytes,
                                sizeof(float) * (kTile * kDStride + 4 * kTile));
    {
      static std::atomic<bool> allowed[64];
      int dev = 0;
      (void) hipGetDevice(&dev);
      if (!allowed[dev & 63].load(std::memory_order_acquire))
      {
        (void) hipFuncSetAttribute(
            reinterpret_cast<const void*>(mfma_tiles_batch_kernel<kPacked>),
            hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        allowed[dev & 63].store(true, std::memory_order_release);
      }
    }
    // ONE tile grid over all pairs: blockIdx.z = pair
    for (int first = 0; first < n_pairs; first += kTileBatchPairs)
    {
      const int m = std::min(kTileBatchPairs, n_pairs - first);
      TileBatchArgs a{};
      for (int k = 0; k < m; ++k)
      {
        const MatchBatchPair& b = host_table[first + k];
        a.Ah[k] = b.split1;
        a.Bh[k] = b.split2;
        a.na[k] = b.na;
        a.nb[k] = b.nb;
        a.rowmin[k] = b.rowmin;
        a.colmin[k] = b.colmin;
        a.n1[k] = b.n1;
        a.n2[k] = b.n2;
      }
      hipLaunchKernelGGL(mfma_tiles_batch_kernel<kPacked>, dim3(tn, tm, m), dim3(256), lds,
                         stream, a, dimp, cap);
    }
    const int nmax = std::max(n1_max, n2_max);
    hipLaunchKernelGGL(select_packed_kernel<true>, dim3((nmax + 15) / 16, 2 * n_pairs),
                       dim3(256), 0, stream, nullptr, 0, 0, nullptr, nullptr, dim, cap,
                       nullptr, nullptr, table);
    {
      const size_t threads = size_t(nmax) * cap;
      const dim3 g(unsigned((threads + 255) / 256), 2, n_pairs);
      static_assert(kMatchBatchCap == 8, "rerank_kernel<8>");
      hipLaunchKernelGGL((rerank_kernel<8, true>), g, dim3(256), 0, stream, RerankArgs{}, dim,
                         squared_ratio_thres, 0, table);
    }
    // queries whose candidate slots overflowed (none on ordinary pairs: the
    // workgroups find a zero count and leave)
    hipLaunchKernelGGL(fallback_kernel<true>, dim3(std::min(nmax, 64), n_pairs, 2), dim3(1024),
                       0, stream, FallbackArgs{}, dim, squared_ratio_thres, 0, table);
    launch_finish_matches_batch(table, n_pairs, n1_max + n2_max, squared_ratio_thres,
                                stream);
  }

}  // namespace sara_hip
