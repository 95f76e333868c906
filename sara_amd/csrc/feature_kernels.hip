// gfx950 (MI355X / CDNA4) kernels of the SIFT front-end.
//
// Built with -ffp-contract=off: the CPU reference is compiled for baseline
// x86-64 (no FMA), and every kernel below evaluates its float expressions in
// the reference's operation order so that pyramids, DoG layers, extremum
// classification, refinement and polar gradients come out bit-identical.
//
// Wave = 64 lanes everywhere; workgroups are 256 threads = 4 waves.
#include "feature_shared.hpp"

#include <atomic>

#include "device_math.hpp"

#include <cmath>
#include <cstdlib>
#include <string>
#include <type_traits>

namespace sara_hip {

  // ======================================================================== //
  // Polar gradients.  Reference: gradient_polar_coordinates,
  // FeatureDescriptors/Orientation.cpp:24-56; Gradient functor,
  // ImageProcessing/Differential.hpp:46-61 (central difference / 2, one-sided
  // (f1 - f0) / 2 on the borders).  Output (2*|g|, atan2f(gy, gx)).
  // ======================================================================== //
  __global__ void gradient_polar_kernel(const float* __restrict__ src,
                                        size_t src_stride,
                                        float2* __restrict__ dst,
                                        size_t dst_stride2, int w, int h,
                                        int nscales, unsigned* __restrict__ cmax,
                                        size_t cmax_stride)
  {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h)
      return;
    const int z = blockIdx.z;
    const size_t b = z / nscales;
    const size_t s = z - b * nscales;
    const size_t plane = size_t(w) * h;
    const float* f = src + b * src_stride + s * plane;
    float2* o = dst + b * dst_stride2 + s * plane;

    const size_t c = size_t(y) * w + x;
    float gx, gy;
    if (x == 0)
      gx = (f[c + 1] - f[c]) / 2;
    else if (x == w - 1)
      gx = (f[c] - f[c - 1]) / 2;
    else
      gx = (f[c + 1] - f[c - 1]) / 2;
    if (y == 0)
      gy = (f[c + w] - f[c]) / 2;
    else if (y == h - 1)
      gy = (f[c] - f[c - w]) / 2;
    else
      gy = (f[c + w] - f[c - w]) / 2;
    const float r = 2 * sqrtf(gx * gx + gy * gy);
    const float theta = fdlibm_atan2f_fast(gy, gx);
    o[c] = make_float2(r, theta);
    if (cmax)
    {
      const int cw = (w + 15) / 16, ch = (h + 15) / 16;
      atomicMax(cmax + b * cmax_stride + (s * ch + y / 16) * size_t(cw) + x / 16,
                __float_as_uint(r));
    }
  }

  //! Value of the previous / next lane through DPP wave shifts (one VALU op,
  //! no LDS crossbar round trip).  Lane 0 of shift_from_prev and lane 63 of
  //! shift_from_next receive their own value.
  __device__ inline float shift_from_prev(float v)
  {
    // wave_shr:1 - lane l reads lane l-1
    return __int_as_float(__builtin_amdgcn_update_dpp(
        __float_as_int(v), __float_as_int(v), 0x138, 0xf, 0xf, false));
  }
  __device__ inline float shift_from_next(float v)
  {
    // wave_shl:1 - lane l reads lane l+1
    return __int_as_float(__builtin_amdgcn_update_dpp(
        __float_as_int(v), __float_as_int(v), 0x130, 0xf, 0xf, false));
  }

  //! Fast path (w % 4 == 0): one wave marches down a strip of 256 columns
  //! (4 per lane, float4 loads, 32-byte stores).  Rows y-1, y, y+1 live in a
  //! register ring (loop unrolled 3x), horizontal neighbours come from the
  //! adjacent lane through ds_bpermute and, at the strip edges, from one extra
  //! scalar load.  HBM traffic: 4 B read + 8 B written per pixel.
  // Settled by the sweeps of rounds 2-4 (docs/experiments.md):
  constexpr int kGradWavesPerEu = 6;  // 4-8 waves per SIMD: within 5 %
  constexpr int kGradPrefetch = 2;    // rows in flight; 3-4 no faster
  // atanf reduction: 0 select chains, 1 5-row table + IEEE division and sqrt,
  // 2 look-up table + the short sqrt / division sequences of device_math.hpp
  // (1.97 -> 1.85 -> 1.63 ms per step)
  constexpr int g_atan_table = 2;

  //! Fills the look-up form of the atanf reduction table (device_math.hpp).
  __device__ inline void fill_atan_lut(float* lut, int tid, int nthreads)
  {
    const float init[kAtanTableFloats] = SARA_ATAN_TABLE_INIT;
    for (int j = tid; j < kAtanLutRows; j += nthreads)
    {
      const int src = atan_lut_source_row(j);
#pragma unroll
      for (int q = 0; q < 8; ++q)
        lut[8 * j + q] = init[8 * src + q];
    }
  }
  template <int PF>
  __global__ __launch_bounds__(64, kGradWavesPerEu) void gradient_polar_march_kernel(
      const float* __restrict__ src, size_t src_stride,
      float* __restrict__ dst, size_t dst_stride, int w, int h, int nscales,
      int seg_rows, int nstrips, int nseg, int xcd_total,
      unsigned* __restrict__ cmax, size_t cmax_stride)
  {
    // Round 2 layout.  A lane owns two pixel PAIRS of the 256-column strip,
    // columns (2l, 2l+1) and (128 + 2l, 128 + 2l + 1): the two 8-byte loads
    // and, above all, the two 16-byte stores of a row are then contiguous
    // across the wave (1 KB each).  With 4 consecutive pixels per lane every
    // store instruction wrote 16 of each 32 bytes, and the 1 read : 2 write
    // stream stopped at 4.35 TB/s whatever the kernel computed (measured
    // without any arithmetic, tools/ubench/rw12_patterns.hip); contiguous
    // stores stream at 5.0 TB/s.
    constexpr int W = 256;
    const int lane = threadIdx.x;
    // argument-reduction table of atanf (device_math.hpp) in LDS
    __shared__ __attribute__((aligned(16))) float
        s_atan[g_atan_table == 2 ? kAtanLutFloats : kAtanTableFloats];
    {
      if (g_atan_table == 2)
        fill_atan_lut(s_atan, lane, 64);
      else
      {
        const float init[kAtanTableFloats] = SARA_ATAN_TABLE_INIT;
        if (lane < kAtanTableFloats)
          s_atan[lane] = init[lane];
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    }
    int strip, seg;
    size_t zz;
    if (!march_work_item(nstrips, nseg, xcd_total, strip, seg, zz))
      return;
    const int z = int(zz);
    const size_t b = z / nscales;
    const size_t s = z - b * nscales;
    const size_t plane = size_t(w) * h;
    const float* f = src + b * src_stride + s * plane;
    float* o = dst + b * dst_stride + s * plane * 2;

    const int x0 = strip * W;
    const int colA = x0 + 2 * lane, colB = x0 + 128 + 2 * lane;
    const bool okA = colA < w, okB = colB < w;
    // odd width: the pair that starts at the last column has one pixel.  It
    // loads (w-2, w-1) and keeps the second value twice: the right neighbour
    // of column w-1 is then the column itself, which is the one-sided
    // difference the border takes (rows are only 4-byte aligned here; the
    // stores 8-byte: tools/ubench/unaligned_check.hip).
    const bool fullA = colA + 1 < w, fullB = colB + 1 < w;
    const bool tailA = okA && !fullA, tailB = okB && !fullB;
    const int mcolA = fullA ? colA : w - 2;
    const int mcolB = fullB ? colB : w - 2;
    const int y0 = seg * seg_rows;
    const int y1 = min(h, y0 + seg_rows);
    // strip-edge neighbours: lane 0 needs column x0-1, lane 63 column x0+256
    const int ecol = lane == 0 ? max(x0 - 1, 0) : min(x0 + W, w - 1);
    const bool edge_lane = (lane == 0) || (lane == 63);

    // one row: m = (A.x, A.y, B.x, B.y), e = the edge column of lanes 0 / 63
    auto load_row = [&](int yy, float4& m, float& e) {
      const int gy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
      const float* rowp = f + size_t(gy) * w;
      const float2 pa = *reinterpret_cast<const float2*>(rowp + mcolA);
      const float2 pb = *reinterpret_cast<const float2*>(rowp + mcolB);
      m = make_float4(pa.x, pa.y, pb.x, pb.y);
      e = 0.f;
      if (edge_lane)
        e = rowp[ecol];
    };

    // coarse 16x16 magnitude maxima: 8 lanes = 16 columns per pair group
    const int cw = (w + 15) / 16, ch = (h + 15) / 16;
    unsigned* cm = cmax ? cmax + b * cmax_stride + s * size_t(ch) * cw : nullptr;
    float run_maxA = 0.f, run_maxB = 0.f;

    float4 ring[3];   // source rows (n-2, n-1, n) by n % 3
    float ering[3];
    float4 pm[PF];
    float pe[PF];
    const int T = (y1 - y0) + 2;  // source rows y0-1 .. y1
#pragma unroll
    for (int q = 0; q < PF; ++q)
      load_row(y0 - 1 + q, pm[q], pe[q]);

    for (int n0 = 0; n0 < T; n0 += 3 * PF)
    {
#pragma unroll
      for (int i = 0; i < 3 * PF; ++i)
      {
        const int n = n0 + i;
        const int yy = y0 - 1 + n;  // source row arriving now
        ring[i % 3] = pm[i % PF];
        // odd tail (see fullA / fullB): here, where the prefetched row is
        // consumed, not behind the load
        ring[i % 3].x = tailA ? ring[i % 3].y : ring[i % 3].x;
        ring[i % 3].z = tailB ? ring[i % 3].w : ring[i % 3].z;
        ering[i % 3] = pe[i % PF];
        // (rows behind the segment's last one, y1, are never consumed: ask
        // for y1 again - a cache hit - instead of streaming them from HBM)
        load_row(min(yy + PF, y1), pm[i % PF], pe[i % PF]);

        const int y = yy - 1;  // output row: needs rows y-1, y, y+1
        if (n >= 2 && y < y1)
        {
          const float4 up = ring[(i + 1) % 3];   // row y-1
          const float4 mid = ring[(i + 2) % 3];  // row y
          const float4 dn = ring[i % 3];         // row y+1
          const float emid = ering[(i + 2) % 3];
          // horizontal neighbours of the two pairs
          float leftA = shift_from_prev(mid.y);
          float rightA = shift_from_next(mid.x);
          float leftB = shift_from_prev(mid.w);
          float rightB = shift_from_next(mid.z);
          // pair A ends at column x0+127, whose right neighbour is pair B's
          // first column (lane 0), and vice versa
          const float firstB = __int_as_float(
              __builtin_amdgcn_readfirstlane(__float_as_int(mid.z)));
          const float lastA = __int_as_float(
              __builtin_amdgcn_readlane(__float_as_int(mid.y), 63));
          if (lane == 0)
          {
            leftA = emid;
            leftB = lastA;
          }
          if (lane == 63)
          {
            rightA = firstB;
            rightB = emid;
          }
          // Differential.hpp:46-61: one-sided differences on the image border
          // = central differences with the missing neighbour replaced by the
          // pixel itself.  Rows and the left column get that from the clamped
          // loads (row -1 is row 0, column -1 is column 0); only the column
          // after the last one (w is even here) needs a select.
          if (colA + 2 >= w)
            rightA = mid.y;
          if (colB + 2 >= w)
            rightB = mid.w;
          // pixel order: A.x, A.y, B.x, B.y
          const float cl[4] = {leftA, mid.x, leftB, mid.z};
          const float cr[4] = {mid.y, rightA, mid.w, rightB};
          const float cu[4] = {up.x, up.y, up.z, up.w};
          const float cd[4] = {dn.x, dn.y, dn.z, dn.w};
          float res[8];
          float ss[4];
          float gxs[4], gys[4];
#pragma unroll
          for (int c = 0; c < 4; ++c)
          {
            const float gx = (cr[c] - cl[c]) / 2;
            const float gy = (cd[c] - cu[c]) / 2;
            gxs[c] = gx;
            gys[c] = gy;
            ss[c] = gx * gx + gy * gy;
            res[2 * c + 1] = g_atan_table == 2   ? atan2f_lut_nonzero_x(gy, gx, s_atan)
                             : g_atan_table == 1 ? fdlibm_atan2f_table(gy, gx, s_atan)
                                                 : fdlibm_atan2f_fast(gy, gx);
          }
          if (g_atan_table == 2)
          {
            // gx == +-0 (flat rows: common, but then usually for whole waves)
            // takes atan2f's special values; skipped when no lane needs them
            const uint32_t anyx = min(min(__float_as_uint(gxs[0]) << 1,
                                          __float_as_uint(gxs[1]) << 1),
                                      min(__float_as_uint(gxs[2]) << 1,
                                          __float_as_uint(gxs[3]) << 1));
            if (__ballot(anyx == 0u) != 0ull)
            {
#pragma unroll
              for (int c = 0; c < 4; ++c)
                if ((__float_as_uint(gxs[c]) << 1) == 0u)
                  res[2 * c + 1] = atan2f_zero_x(gys[c], gxs[c]);
            }
          }
          if (g_atan_table == 2)
          {
            // the short square root is exact for 0 and from 2^-102 up to
            // FLT_MAX; anything else (never seen on image data) sends the
            // whole wave through sqrtf()
            const int emin = min(min(sqrt_short_exponent(ss[0]), sqrt_short_exponent(ss[1])),
                                 min(sqrt_short_exponent(ss[2]), sqrt_short_exponent(ss[3])));
            const float smax = fmaxf(fmaxf(ss[0], ss[1]), fmaxf(ss[2], ss[3]));
            const bool odd = emin < kSqrtShortMinExponent ||
                             !(smax < __builtin_inff());
            if (__builtin_expect(__ballot(odd) != 0ull, 0))
            {
#pragma unroll
              for (int c = 0; c < 4; ++c)
                res[2 * c] = 2 * sqrtf(ss[c]);
            }
            else
            {
#pragma unroll
              for (int c = 0; c < 4; ++c)
                res[2 * c] = 2 * sqrt_rn_short(ss[c]);
            }
          }
          else
          {
#pragma unroll
            for (int c = 0; c < 4; ++c)
              res[2 * c] = 2 * sqrtf(ss[c]);
          }
          {
            // lane l: 16 bytes at (x0 + 2l) and at (x0 + 128 + 2l) - contiguous
            float4* ob = reinterpret_cast<float4*>(o + (size_t(y) * w + x0) * 2);
            if (fullA)
              ob[lane] = make_float4(res[0], res[1], res[2], res[3]);
            else if (okA)
              *reinterpret_cast<float2*>(ob + lane) = make_float2(res[0], res[1]);
            if (fullB)
              ob[64 + lane] = make_float4(res[4], res[5], res[6], res[7]);
            else if (okB)
              *reinterpret_cast<float2*>(ob + 64 + lane) = make_float2(res[4], res[5]);
          }
          if (cm)
          {
            float mA = okA ? fmaxf(res[0], fullA ? res[2] : 0.f) : 0.f;
            float mB = okB ? fmaxf(res[4], fullB ? res[6] : 0.f) : 0.f;
            // max over the 8 lanes of a 16-column group: the two quad
            // permutations, then the mirror of the 8-lane half row
            auto max8 = [](float m) {
              m = fmaxf(m, __int_as_float(__builtin_amdgcn_mov_dpp(
                               __float_as_int(m), 0xB1, 0xf, 0xf, true)));
              m = fmaxf(m, __int_as_float(__builtin_amdgcn_mov_dpp(
                               __float_as_int(m), 0x4E, 0xf, 0xf, true)));
              m = fmaxf(m, __int_as_float(__builtin_amdgcn_mov_dpp(
                               __float_as_int(m), 0x141, 0xf, 0xf, true)));
              return m;
            };
            // per lane over the rows of the band; the 8-lane reduction runs
            // once per band, not once per row (max is associative: the same
            // value; round 5: timing builds showed this kernel VALU-issue-bound
            // - without stores 1.41 of 1.44 ms - and the per-row reduction 6 %
            // of it)
            run_maxA = fmaxf(run_maxA, mA);
            run_maxB = fmaxf(run_maxB, mB);
            if ((y & 15) == 15 || y == y1 - 1)
            {
              run_maxA = max8(run_maxA);
              run_maxB = max8(run_maxB);
              if ((lane & 7) == 0)
              {
                // segments of whole 16-row bands (seg_rows % 16 == 0): this
                // lane is the only writer of its 16 x 16 blocks - plain
                // stores, and the map needs no zeroing before the launch
                unsigned* rowc = cm + size_t(y >> 4) * cw;
                if (okA)
                {
                  if ((seg_rows & 15) == 0)
                    rowc[colA >> 4] = __float_as_uint(run_maxA);
                  else
                    atomicMax(rowc + (colA >> 4), __float_as_uint(run_maxA));
                }
                if (okB)
                {
                  if ((seg_rows & 15) == 0)
                    rowc[colB >> 4] = __float_as_uint(run_maxB);
                  else
                    atomicMax(rowc + (colB >> 4), __float_as_uint(run_maxB));
                }
              }
              run_maxA = 0.f;
              run_maxB = 0.f;
            }
          }
        }
      }
    }
  }

  //! Pixel-parallel form for launches too small for the marching kernel (one
  //! frame per call: a 240 x 135 octave is 27 marching waves of 18 dependent
  //! row steps each, 20 us; here it is 108 workgroups, 4 us).  A workgroup of
  //! 256 threads owns a tile of 64 columns x 16 rows - exactly four 16 x 16
  //! blocks of the coarse magnitude map, which it writes with plain stores (no
  //! atomics, no zero-fill); a thread computes 4 consecutive rows of one
  //! column from a 6-row register window, horizontal neighbours from the
  //! adjacent lane (DPP) and, at the tile's edge columns, one extra load.  The
  //! arithmetic is gradient_polar_march_kernel's (same expressions, same
  //! device_math.hpp forms): bit-identical planes.
  __global__ __launch_bounds__(256) void gradient_polar_tile_kernel(
      const float* __restrict__ src, size_t src_stride,
      float* __restrict__ dst, size_t dst_stride, int w, int h, int nscales,
      unsigned* __restrict__ cmax, size_t cmax_stride)
  {
    __shared__ __attribute__((aligned(16))) float s_atan[kAtanLutFloats];
    __shared__ float s_max[4][4];
    const int tid = threadIdx.x;
    fill_atan_lut(s_atan, tid, 256);
    __syncthreads();
    const int lane = tid & 63, ty = tid >> 6;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 16;
    const int z = blockIdx.z;
    const size_t b = z / nscales;
    const size_t s = z - b * nscales;
    const size_t plane = size_t(w) * h;
    const float* f = src + b * src_stride + s * plane;
    float* o = dst + b * dst_stride + s * plane * 2;
    const int x = x0 + lane;
    const int xc = min(x, w - 1);
    const bool okx = x < w;
    const int ya = y0 + 4 * ty;  // first output row of this thread
    // rows ya-1 .. ya+4 of the own column (clamped: the border rows repeat,
    // which is the one-sided difference, Differential.hpp:46-61)
    float v[6];
#pragma unroll
    for (int r = 0; r < 6; ++r)
    {
      const int gy = min(max(ya - 1 + r, 0), h - 1);
      v[r] = f[size_t(gy) * w + xc];
    }
    // edge columns of the tile: the left neighbour of lane 0, the right one of
    // lane 63 (clamped at the image border as well)
    const bool edge = lane == 0 || lane == 63;
    const int ex = lane == 0 ? max(x0 - 1, 0) : min(x0 + 64, w - 1);
    float e[4] = {0.f, 0.f, 0.f, 0.f};
    if (edge)
    {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        e[r] = f[size_t(min(ya + r, h - 1)) * w + ex];
    }
    float run_max = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
    {
      const int y = ya + r;
      const float mid = v[r + 1];
      float left = shift_from_prev(mid);
      float right = shift_from_next(mid);
      if (lane == 0)
        left = e[r];
      if (lane == 63)
        right = e[r];
      if (x + 1 >= w)
        right = mid;
      const float gx = (right - left) / 2;
      const float gy = (v[r + 2] - v[r]) / 2;
      const float ss = gx * gx + gy * gy;
      float a = atan2f_lut_nonzero_x(gy, gx, s_atan);
      if ((__float_as_uint(gx) << 1) == 0u)
        a = atan2f_zero_x(gy, gx);
      const bool odd = sqrt_short_exponent(ss) < kSqrtShortMinExponent ||
                       !(ss < __builtin_inff());
      float m;
      if (__builtin_expect(__ballot(odd) != 0ull, 0))
        m = 2 * sqrtf(ss);
      else
        m = 2 * sqrt_rn_short(ss);
      if (okx && y < h)
      {
        *reinterpret_cast<float2*>(o + (size_t(y) * w + x) * 2) = make_float2(m, a);
        run_max = fmaxf(run_max, m);
      }
    }
    if (cmax)
    {
      // maximum over the 16 lanes of a DPP row = one 16-column block
      float m = run_max;
      m = fmaxf(m, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m), 0xB1, 0xf, 0xf, true)));
      m = fmaxf(m, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m), 0x4E, 0xf, 0xf, true)));
      m = fmaxf(m, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m), 0x141, 0xf, 0xf, true)));
      m = fmaxf(m, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m), 0x140, 0xf, 0xf, true)));
      if ((lane & 15) == 0)
        s_max[ty][lane >> 4] = m;
      __syncthreads();
      if (tid < 4)
      {
        const int cw = (w + 15) / 16, ch = (h + 15) / 16;
        const int cx = (x0 >> 4) + tid;
        if (cx < cw)
        {
          const float t = fmaxf(fmaxf(s_max[0][tid], s_max[1][tid]),
                                fmaxf(s_max[2][tid], s_max[3][tid]));
          cmax[b * cmax_stride + (s * ch + (y0 >> 4)) * size_t(cw) + cx] =
              __float_as_uint(t);
        }
      }
    }
  }

  //! Segments of the marching gradient kernel made of whole 16-row bands (one
  //! writer per entry of the coarse magnitude map, no memset) for small
  //! batches, where the saved memset launches count (one 1080p frame: -20 us);
  //! with 64 frames the aligned split runs the overlapped extrema + gradient
  //! stage 0.15 ms slower than the free split with atomicMax.
  static inline bool grad_bands(int batch) { return batch <= 8; }
  //! Target number of waves per marching launch (sweeps on MI355X with 64 x
  //! 1080p frames: the extremum scan likes long segments, the gradient kernel
  //! many short ones).
  constexpr int g_grad_waves = 18432;
  constexpr int g_extrema_waves = 4096;

  //! Planes below KernelSelection::grad_tile_pixels per launch (w x h x batch)
  //! go to the pixel-parallel gradient kernel (0 = never).
  static inline bool grad_small_launch(int w, int h, int batch)
  {
    return w >= 2 && h >= 2 && (long long) w * h * batch < selection().grad_tile_pixels;
  }

  bool gradient_polar_needs_zeroed_cmax(const float* src, size_t src_stride,
                                        const float* dst, size_t dst_stride,
                                        int w, int h, int batch)
  {
    // any width (gradient_polar_march_kernel: pairs, odd tail)
    const bool aligned4 = w >= 4 && h >= 2;
    if (grad_small_launch(w, h, batch))
      return false;  // gradient_polar_tile_kernel: one writer per entry
    // the generic kernel and the free split of big batches use atomicMax
    return !(aligned4 && selection().feature_march && grad_bands(batch));
  }

  void launch_gradient_polar(const float* src, size_t src_stride, float* dst,
                             size_t dst_stride, int w, int h, int nscales,
                             int batch, hipStream_t stream, unsigned* cmax,
                             size_t cmax_stride)
  {
    // any width (gradient_polar_march_kernel: pairs, odd tail)
    const bool aligned4 = w >= 4 && h >= 2;
    if (grad_small_launch(w, h, batch))
    {
      const dim3 grid((w + 63) / 64, (h + 15) / 16, batch * nscales);
      hipLaunchKernelGGL(gradient_polar_tile_kernel, grid, dim3(256), 0, stream, src,
                         src_stride, dst, dst_stride, w, h, nscales, cmax,
                         cmax_stride);
      return;
    }
    if (aligned4 && selection().feature_march)
    {
      const int nstrips = (w + 255) / 256;
      const int planes = batch * nscales;
      int nseg = (g_grad_waves + nstrips * planes - 1) / (nstrips * planes);
      nseg = std::max(1, std::min(nseg, (h + 15) / 16));
      // whole 16-row bands per segment: every 16 x 16 block of the coarse
      // magnitude map has exactly one writer (gradient_polar_needs_zeroed_cmax)
      const int seg_rows = grad_bands(batch) ? (((h + nseg - 1) / nseg) + 15) & ~15
                                        : (h + nseg - 1) / nseg;
      nseg = (h + seg_rows - 1) / seg_rows;
      const int total = xcd_map_enabled() ? nstrips * nseg * planes : 0;
      const dim3 grid = total ? dim3(8 * ((total + 7) / 8)) : dim3(nstrips * nseg, planes);
      hipLaunchKernelGGL((gradient_polar_march_kernel<kGradPrefetch>), grid, dim3(64), 0,
                         stream, src, src_stride, dst, dst_stride, w, h, nscales,
                         seg_rows, nstrips, nseg, total, cmax, cmax_stride);
      return;
    }
    const dim3 block(64, 4);
    const dim3 grid((w + 63) / 64, (h + 3) / 4, batch * nscales);
    hipLaunchKernelGGL(gradient_polar_kernel, grid, block, 0, stream, src,
                       src_stride, reinterpret_cast<float2*>(dst),
                       dst_stride / 2, w, h, nscales, cmax, cmax_stride);
  }

  // ======================================================================== //
  // Scale-space extrema.  Reference: local_scale_space_extrema (non-Halide
  // branch), FeatureDetectors/RefineExtremum.cpp:363-521; predicates
  // ImageProcessing/Extrema.hpp:28-75; on_edge RefineExtremum.cpp:24-30;
  // refine_extremum RefineExtremum.cpp:32-130 with the 3-D gradient/hessian of
  // ImageProcessing/GaussianPyramid.hpp:183-233.
  // ======================================================================== //

  //! One frame's DoG octave, never materialised: layer s is the difference of
  //! the Gaussian planes s+1 and s (GaussianPyramid.cpp:44-46), evaluated where
  //! it is needed.  `layers` = number of DoG layers = Gaussian scales - 1.
  struct DogOctave
  {
    const float* base;  // Gaussian planes [scale][h][w] of one frame
    int w, h;
    size_t plane;
    int layers;
    __device__ float at(int x, int y, int s) const
    {
      const size_t i = size_t(s) * plane + size_t(y) * w + x;
      return base[i + plane] - base[i];
    }
    //! BoundaryConditions::repeat_edge of the Halide classifier.
    __device__ float at_clamped(int x, int y, int s) const
    {
      x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
      y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
      return at(x, y, s);
    }
  };

  //! is_dog_extremum of the reference's DO_SARA_USE_HALIDE build
  //! (Shakti/Halide/Components/DoGExtremum.hpp:59-78 on repeat_edge inputs,
  //! generator v2::LocalScaleSpaceExtremum): every pixel, replicated borders,
  //! value == max / min of the 3 x 3 x 3 block, STRICT contrast test, on_edge
  //! through the Halide hessian (Components/Differential.hpp:33-49, whose
  //! cross term reads (x-1, y-1) twice - restated as written).  `get(x, y, ds)`
  //! reads layer s + ds with clamped coordinates.  -> +1 / -1 / 0.
  template <typename Get>
  __device__ inline int halide_is_dog_extremum(Get get, int x, int y,
                                               float edge_ratio, float thres)
  {
    const float v = get(x, y, 0);
    float mx = v, mn = v;
#pragma unroll
    for (int dv = -1; dv <= 1; ++dv)
#pragma unroll
      for (int du = -1; du <= 1; ++du)
      {
        const float a = get(x + du, y + dv, -1);
        const float b = get(x + du, y + dv, 0);
        const float c = get(x + du, y + dv, 1);
        mx = fmaxf(mx, fmaxf(a, fmaxf(b, c)));
        mn = fminf(mn, fminf(a, fminf(b, c)));
      }
    const bool is_max = mx == v, is_min = mn == v;
    if (!(is_max || is_min) || !(fabsf(v) > 0.8f * thres))
      return 0;
    const float dxx = get(x + 1, y, 0) + get(x - 1, y, 0) - 2 * v;
    const float dyy = get(x, y + 1, 0) + get(x, y - 1, 0) - 2 * v;
    const float dxy = (get(x + 1, y + 1, 0) - get(x - 1, y - 1, 0) -
                       get(x + 1, y - 1, 0) + get(x - 1, y - 1, 0)) /
                      4;
    const float tr = dxx + dyy;
    const float det = dxx * dyy - dxy * dxy;
    if ((tr * tr) * edge_ratio >= ((1 + edge_ratio) * (1 + edge_ratio)) * fabsf(det))
      return 0;
    return is_max ? 1 : -1;
  }

  __device__ inline float sum3(float a0, float a1, float a2)
  {
    return a0 + (a1 + a2);
  }

  __device__ inline float cofactor3(const float m[3][3], int i, int j)
  {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
    const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1];
  }

  // ---- definiteness of the 3x3 Hessian ---------------------------------------
  // RefineExtremum.cpp:74-77 decides with the eigenvalues of
  // SelfAdjointEigenSolver<Matrix3f> (Eigen >= 3.4): the constructor runs the
  // iterative compute() in float.  Its published algorithm (Eigen 3.4.0:
  // SelfAdjointEigenSolver.h, Tridiagonalization.h, Jacobi.h, MathFunctions.h)
  // is restated here operation by operation - scaling by the largest
  // coefficient, the closed-form 3x3 Householder tridiagonalisation, implicit
  // symmetric QR steps with Wilkinson shift and the 3.4 deflation rule -
  // because a float solver can report a tiny wrong-signed eigenvalue for a
  // nearly singular Hessian, which an exact test would not.  (Measured: on
  // 64 x 1080p + the golden photograph the decision never differs from
  // Sylvester's criterion in double)
  __device__ inline float eigen_hypot(float x, float y)
  {
    x = fabsf(x);
    y = fabsf(y);
    if (isinf(x) || isinf(y))
      return __builtin_inff();
    if (isnan(x) || isnan(y))
      return __builtin_nanf("");
    const float p = fmaxf(x, y);
    if (p == 0.f)
      return 0.f;
    const float qp = fminf(y, x) / p;
    return p * sqrtf(1.f + qp * qp);
  }

  //! JacobiRotation<float>::makeGivens(p, q), real case.
  __device__ inline void eigen_make_givens(float p, float q, float& c, float& s)
  {
    if (q == 0.f)
    {
      c = p < 0.f ? -1.f : 1.f;
      s = 0.f;
    }
    else if (p == 0.f)
    {
      c = 0.f;
      s = q < 0.f ? 1.f : -1.f;
    }
    else if (fabsf(p) > fabsf(q))
    {
      const float t = q / p;
      float u = sqrtf(1.f + t * t);
      if (p < 0.f)
        u = -u;
      c = 1.f / u;
      s = -t * c;
    }
    else
    {
      const float t = p / q;
      float u = sqrtf(1.f + t * t);
      if (q < 0.f)
        u = -u;
      s = -1.f / u;
      c = -t * s;
    }
  }

  //! (SelfAdjointEigenSolver<Matrix3f>(H).eigenvalues() * float(type))
  //! .maxCoeff() >= 0
  //! Sylvester's criterion for the symmetric matrix sign * M + shift * I.
  __device__ inline bool positive_definite3(double a00, double a10, double a11,
                                            double a20, double a21, double a22,
                                            double sign, double shift)
  {
    a00 = sign * a00 + shift;
    a11 = sign * a11 + shift;
    a22 = sign * a22 + shift;
    a10 *= sign;
    a20 *= sign;
    a21 *= sign;
    const double m2 = a00 * a11 - a10 * a10;
    const double det = a00 * (a11 * a22 - a21 * a21) -
                       a10 * (a10 * a22 - a21 * a20) +
                       a20 * (a10 * a21 - a11 * a20);
    return a00 > 0. && m2 > 0. && det > 0.;
  }

  __device__ inline bool not_definite_enough3(const float H[3][3], int type)
  {
    float m00 = H[0][0], m10 = H[1][0], m11 = H[1][1], m20 = H[2][0],
          m21 = H[2][1], m22 = H[2][2];
    float scale = fmaxf(fmaxf(fmaxf(fabsf(m00), fabsf(m10)),
                              fmaxf(fabsf(m11), fabsf(m20))),
                        fmaxf(fabsf(m21), fabsf(m22)));
    if (scale == 0.f)
      scale = 1.f;
    // The answer is the sign of an extreme eigenvalue as the float solver
    // computes it.  The solver is backward stable: its eigenvalues of the
    // scaled matrix (largest |coefficient| = 1) are off by a few tens of
    // float epsilons at most.  When the exact extreme eigenvalue is farther
    // than delta = 2^-12 (2 048 epsilons) from zero - decided with
    // Sylvester's criterion in double on the shifted matrix - the float
    // solver cannot report the other sign and its iteration is skipped;
    // otherwise (nearly singular Hessians, a handful per frame) it runs.
    // The scale guard keeps (eigenvalue * scale) * type away from underflow.
    if (scale > 1e-20f && scale < 1e20f)
    {
      const double inv = 1. / double(scale);
      const double a00 = double(m00) * inv, a10 = double(m10) * inv,
                   a11 = double(m11) * inv, a20 = double(m20) * inv,
                   a21 = double(m21) * inv, a22 = double(m22) * inv;
      constexpr double delta = 1. / 4096.;
      // type > 0: lambda_max >= 0 ?   type < 0: lambda_min <= 0 ?
      const double sg = type > 0 ? -1. : 1.;
      if (positive_definite3(a00, a10, a11, a20, a21, a22, sg, -delta))
        return false;  // every eigenvalue is beyond delta on the definite side
      if (!positive_definite3(a00, a10, a11, a20, a21, a22, sg, delta))
        return true;   // an eigenvalue is beyond delta on the wrong side
    }
    m00 /= scale;
    m10 /= scale;
    m11 /= scale;
    m20 /= scale;
    m21 /= scale;
    m22 /= scale;

    // d0..d2 / e0, e1: diagonal and sub-diagonal of the tridiagonal form, in
    // named scalars (registers; n = 3 leaves only the blocks [0,1], [1,2], [0,2])
    float d0 = m00, d1, d2, e0, e1;
    const float tiny = 1.17549435e-38f;  // numeric_limits<float>::min()
    const float v1norm2 = m20 * m20;
    if (v1norm2 <= tiny)
    {
      d1 = m11;
      d2 = m22;
      e0 = m10;
      e1 = m21;
    }
    else
    {
      const float beta = sqrtf(m10 * m10 + v1norm2);
      const float inv_beta = 1.f / beta;
      const float m01 = m10 * inv_beta;
      const float m02 = m20 * inv_beta;
      const float q = 2.f * m01 * m21 + m02 * (m22 - m11);
      d1 = m11 + m02 * q;
      d2 = m22 - m02 * q;
      e0 = beta;
      e1 = m21 - m01 * q;
    }

    const float eps = 1.1920929e-07f;
    const float precision_inv = 1.f / eps;
    float diag[3] = {d0, d1, d2};
    float sub[2] = {e0, e1};
    int end = 2, start = 0, iter = 0;
    while (end > 0)
    {
      for (int i = start; i < end; ++i)
      {
        if (fabsf(sub[i]) < tiny)
          sub[i] = 0.f;
        else
        {
          const float scaled = precision_inv * sub[i];
          if (scaled * scaled <= (fabsf(diag[i]) + fabsf(diag[i + 1])))
            sub[i] = 0.f;
        }
      }
      while (end > 0 && sub[end - 1] == 0.f)
        end--;
      if (end <= 0)
        break;
      iter++;
      if (iter > 30 * 3)
        break;
      start = end - 1;
      while (start > 0 && sub[start - 1] != 0.f)
        start--;

      // tridiagonal_qr_step(diag, sub, start, end)
      const float td = (diag[end - 1] - diag[end]) * 0.5f;
      const float e = sub[end - 1];
      float mu = diag[end];
      if (td == 0.f)
        mu -= fabsf(e);
      else if (e != 0.f)
      {
        const float e2 = e * e;
        const float h = eigen_hypot(td, e);
        if (e2 == 0.f)
          mu -= e / ((td + (td > 0.f ? h : -h)) / e);
        else
          mu -= e2 / (td + (td > 0.f ? h : -h));
      }
      float x = diag[start] - mu;
      float z = sub[start];
      for (int k = start; k < end && z != 0.f; ++k)
      {
        float c, s;
        eigen_make_givens(x, z, c, s);
        const float sdk = s * diag[k] + c * sub[k];
        const float dkp1 = s * sub[k] + c * diag[k + 1];
        diag[k] = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
        diag[k + 1] = s * sdk + c * dkp1;
        sub[k] = c * sdk - s * dkp1;
        if (k > start)
          sub[k - 1] = c * sub[k - 1] - s * z;
        x = sub[k];
        if (k < end - 1)
        {
          z = -s * sub[k + 1];
          sub[k + 1] = c * sub[k + 1];
        }
      }
    }
    const float t = float(type);
    return fmaxf(fmaxf((diag[0] * scale) * t, (diag[1] * scale) * t),
                 (diag[2] * scale) * t) >= 0.f;
  }

  //! The 19 DoG values refine_extremum and on_edge read around (x, y, s), in
  //! the order of SiteLists::nb.
  struct SiteNeighbourhood
  {
    float v[kSiteNb];
    // layer s: v[3 * (dy + 1) + (dx + 1)]; layers s -+ 1: centre, left, right,
    // up, down at v[9..13] / v[14..18]
    __device__ float mid(int dx, int dy) const { return v[3 * (dy + 1) + dx + 1]; }
    __device__ float below(int i) const { return v[9 + i]; }   // layer s - 1
    __device__ float above(int i) const { return v[14 + i]; }  // layer s + 1
    __device__ void load(const DogOctave& I, int x, int y, int s)
    {
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx)
          v[3 * (dy + 1) + dx + 1] = I.at(x + dx, y + dy, s);
      const int ox[5] = {0, -1, 1, 0, 0}, oy[5] = {0, 0, 0, -1, 1};
#pragma unroll
      for (int i = 0; i < 5; ++i)
      {
        v[9 + i] = I.at(x + ox[i], y + oy[i], s - 1);
        v[14 + i] = I.at(x + ox[i], y + oy[i], s + 1);
      }
      v[19] = 0.f;
    }
  };

  //! refine_extremum.  type: 1 maximum, 255 minimum (the reference's uint8
  //! map stores -1 as 255, so minima are never refined), or -1 minimum with
  //! SARA_HIP_OPT_SIGNED_EXTREMUM_TYPE (the int8 map of RefineExtremum.cpp:246).
  //! pos = (x, y, sigma).  nb (optional): the neighbourhood of the start site
  //! as the scan captured it - the same floats I.at() would return.
  __device__ inline void refine_extremum(const DogOctave& I, int x, int y, int s,
                                         int type, float pos[3], float& val,
                                         int border_sz, int num_iter,
                                         const ScaleTable& tab, float kfactor,
                                         const SiteNeighbourhood* nb = nullptr)
  {
    float D_prime[3] = {0.f, 0.f, 0.f};
    float H[3][3];
    float hh[3] = {0.f, 0.f, 0.f};

    pos[0] = float(x);
    pos[1] = float(y);
    pos[2] = tab.sigma[s];

    SiteNeighbourhood n;
    bool n_valid = false;  // n holds the neighbourhood of the current (x, y)
    for (int i = 0; i < num_iter; ++i)
    {
      if (x < border_sz || x >= I.w - border_sz || y < border_sz ||
          y >= I.h - border_sz || s < 1 || s >= I.layers - 1)
        break;

      if (i == 0 && nb)
        n = *nb;
      else
        n.load(I, x, y, s);
      n_valid = true;
      const float c = n.mid(0, 0);
      D_prime[0] = (n.mid(1, 0) - n.mid(-1, 0)) / 2.f;
      D_prime[1] = (n.mid(0, 1) - n.mid(0, -1)) / 2.f;
      D_prime[2] = (n.above(0) - n.below(0)) / 2.f;

      H[0][0] = n.mid(1, 0) - 2.f * c + n.mid(-1, 0);
      H[1][1] = n.mid(0, 1) - 2.f * c + n.mid(0, -1);
      H[2][2] = n.above(0) - 2.f * c + n.below(0);
      H[0][1] = H[1][0] =
          (n.mid(1, 1) - n.mid(-1, 1) - n.mid(1, -1) + n.mid(-1, -1)) / 4.f;
      H[0][2] = H[2][0] = (n.above(2) - n.above(1) - n.below(2) + n.below(1)) / 4.f;
      H[1][2] = H[2][1] = (n.above(4) - n.above(3) - n.below(4) + n.below(3)) / 4.f;

      // (lambda * float(type)).maxCoeff() >= 0: with type in {1, 255} the
      // Newton step is taken only when H is negative definite, with type -1
      // only when it is positive definite.
      if (not_definite_enough3(H, type))
      {
        hh[0] = hh[1] = hh[2] = 0.f;
        break;
      }

      // h = -inverse(H) * D' (cofactor inverse, Eigen's evaluation order).
      const float c0 = cofactor3(H, 0, 0);
      const float c1 = cofactor3(H, 1, 0);
      const float c2 = cofactor3(H, 2, 0);
      const float det = sum3(c0 * H[0][0], c1 * H[1][0], c2 * H[2][0]);
      const float invdet = 1.f / det;
      float inv[3][3];
      inv[0][0] = c0 * invdet;
      inv[0][1] = c1 * invdet;
      inv[0][2] = c2 * invdet;
      inv[1][0] = cofactor3(H, 0, 1) * invdet;
      inv[1][1] = cofactor3(H, 1, 1) * invdet;
      inv[1][2] = cofactor3(H, 2, 1) * invdet;
      inv[2][0] = cofactor3(H, 0, 2) * invdet;
      inv[2][1] = cofactor3(H, 1, 2) * invdet;
      inv[2][2] = cofactor3(H, 2, 2) * invdet;
#pragma unroll
      for (int r = 0; r < 3; ++r)
        hh[r] = sum3((-inv[r][0]) * D_prime[0], (-inv[r][1]) * D_prime[1],
                     (-inv[r][2]) * D_prime[2]);

      if (fmaxf(fabsf(hh[0]), fabsf(hh[1])) > 1.5f)
        return;  // pos keeps the start site, val the unrefined value

      if (fminf(fabsf(hh[0]), fabsf(hh[1])) > 0.6f)
      {
        x += hh[0] > 0 ? 1 : -1;
        y += hh[1] > 0 ? 1 : -1;
        n_valid = false;
        continue;
      }
      break;
    }

    pos[0] = float(x);
    pos[1] = float(y);
    pos[2] = tab.sigma[s];
    const float oldval = n_valid ? n.mid(0, 0) : I.at(x, y, s);
    const float newval = oldval + 0.5f * sum3(D_prime[0] * hh[0],
                                              D_prime[1] * hh[1],
                                              D_prime[2] * hh[2]);
    if ((type == 1 && oldval <= newval) || (type == -1 && oldval >= newval))
    {
      pos[0] += hh[0];
      pos[1] += hh[1];
      // powf(k, h_s) of the CPU path, evaluated in double and rounded.
      pos[2] *= float(exp(double(hh[2]) * log(double(kfactor))));
      val = newval;
    }
  }

  //! Classification of one site: +1 max, -1 min, 0 none (incl. the 0.8*thres
  //! and on_edge rejections).  a/b/c = layers s-1, s, s+1; p points at (x,y).
  __device__ inline int classify_site(const float* __restrict__ a,
                                      const float* __restrict__ b,
                                      const float* __restrict__ c, int w,
                                      float thres, float edge_ratio)
  {
    const float v = b[0];
    if (fabsf(v) < 0.8f * thres)
      return 0;
    bool is_max = true, is_min = true;
#pragma unroll
    for (int dv = -1; dv <= 1; ++dv)
#pragma unroll
      for (int du = -1; du <= 1; ++du)
      {
        const int off = dv * w + du;
        const float na = a[off], nc = c[off];
        is_max = is_max && (v >= na) && (v >= nc);
        is_min = is_min && (v <= na) && (v <= nc);
        if (du != 0 || dv != 0)
        {
          const float nb = b[off];
          is_max = is_max && (v >= nb);
          is_min = is_min && (v <= nb);
        }
      }
    if (!is_max && !is_min)
      return 0;
    const float hxx = b[1] - 2.f * v + b[-1];
    const float hyy = b[w] - 2.f * v + b[-w];
    const float hxy = (b[w + 1] - b[w - 1] - b[-w + 1] + b[-w - 1]) / 4.f;
    const float tr = hxx + hyy;
    const float det = hxx * hyy - hxy * hxy;
    if ((tr * tr) * edge_ratio >=
        ((edge_ratio + 1.f) * (edge_ratio + 1.f)) * fabsf(det))
      return 0;
    return is_max ? 1 : -1;
  }

  //! Edge test, refinement, contrast test and append of one classified site
  //! (RefineExtremum.cpp:429-434, :454-484, :497-515).  type: +1 / -1.
  __device__ inline void finish_candidate(const DogOctave& I, int x, int y, int s,
                                          int type, int octave, int frame,
                                          const ExtremaParams& p,
                                          const ScaleTable& tab,
                                          const CandidateLists& cand,
                                          const SiteNeighbourhood* nb = nullptr)
  {
    auto at = [&](int dx, int dy) {
      return nb ? nb->mid(dx, dy) : I.at(x + dx, y + dy, s);
    };
    const float v = at(0, 0);
    if (!p.signed_type)  // the Halide classifier has done its own edge test
    {
      const float hxx = at(1, 0) - 2.f * v + at(-1, 0);
      const float hyy = at(0, 1) - 2.f * v + at(0, -1);
      const float hxy = (at(1, 1) - at(-1, 1) - at(1, -1) + at(-1, -1)) / 4.f;
      const float tr = hxx + hyy;
      const float det = hxx * hyy - hxy * hxy;
      const float er = p.edge_ratio_thres;
      if ((tr * tr) * er >= ((er + 1.f) * (er + 1.f)) * fabsf(det))
        return;
    }

    float pos[3];
    float val = v;
    refine_extremum(I, x, y, s, type == 1 ? 1 : (p.signed_type ? -1 : 255), pos,
                    val, p.img_padding_sz, p.refine_iters, tab,
                    p.scale_geometric_factor, nb);
    if (fabsf(val) < p.extremum_thres)
      return;
    if (p.signed_type)
    {
      // RefineExtremum.cpp:307-318: refined scale within (1/4, 4) of sigma(s),
      // compared in double as there.
      const double approx = tab.sigma_d[s];
      if (!(0.25 * approx < double(pos[2]) && double(pos[2]) < 4. * approx))
        return;
    }

    const int slot = atomicAdd(&cand.count[frame], 1);
    if (slot < 0)
      *cand.error = 1;  // the counter was not zeroed for this step
    if (unsigned(slot) < unsigned(cand.cap))
    {
      const size_t i = size_t(frame) * cand.cap + slot;
      cand.key[i] =
          ((((unsigned long long) (octave * kMaxScales + s) << 20 | (unsigned) y)
            << 20 | (unsigned) x)
           << 1) |
          (unsigned) (type == 1);
      cand.data[i] = make_float4(pos[0], pos[1], pos[2], val);
    }
  }

  //! General path (any width): one thread per site, 26 neighbours read
  //! through the on-the-fly DoG accessor.
  __global__ void extrema_scan_kernel(OctaveView gauss, int octave, int nscan,
                                      ExtremaParams p,
                                      const ScaleTable* __restrict__ tabp,
                                      CandidateLists cand)
  {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int z = blockIdx.z;
    const int b = z / nscan;
    const int s = 1 + (z - b * nscan);
    const int w = gauss.w, h = gauss.h;
    const int pad = p.img_padding_sz;
    const DogOctave I{gauss.base + size_t(b) * gauss.frame_stride, w, h,
                      gauss.plane, gauss.scales - 1};
    if (p.signed_type)
    {
      // the map of the DO_SARA_USE_HALIDE branch (RefineExtremum.cpp:246-262)
      if (x >= w || y >= h)
        return;
      const int t = halide_is_dog_extremum(
          [&](int xx, int yy, int ds) { return I.at_clamped(xx, yy, s + ds); }, x,
          y, p.edge_ratio_thres, p.extremum_thres);
      if (t != 0)
        finish_candidate(I, x, y, s, t, octave, b, p, *tabp, cand);
      return;
    }
    if (!(pad <= x && x < w - pad && pad <= y && y < h - pad))
      return;

    const float v = I.at(x, y, s);
    if (fabsf(v) < 0.8f * p.extremum_thres)
      return;
    bool is_max = true, is_min = true;
#pragma unroll
    for (int ds = -1; ds <= 1; ++ds)
#pragma unroll
      for (int dv = -1; dv <= 1; ++dv)
#pragma unroll
        for (int du = -1; du <= 1; ++du)
        {
          if (ds == 0 && dv == 0 && du == 0)
            continue;
          const float nb = I.at(x + du, y + dv, s + ds);
          is_max = is_max && (v >= nb);
          is_min = is_min && (v <= nb);
        }
    if (!is_max && !is_min)
      return;
    finish_candidate(I, x, y, s, is_max ? 1 : -1, octave, b, p, *tabp, cand);
  }

  //! Fast path (even widths, ND = scales-1 DoG layers known at compile time).
  //! One wave marches down a strip of 128 columns (2 per lane, 8-byte loads of
  //! the ND+1 Gaussian planes, strips overlap by 2 columns).  The DoG rows
  //! y-1, y, y+1 of all ND layers live in a register ring.  A site is a
  //! non-strict maximum of its 26 neighbours iff it equals the maximum of the
  //! whole 3x3x3 block, and that maximum is separable: 3-row max per layer
  //! (v_max3), 3-column max through the neighbouring lane, 3-layer max - and
  //! the per-layer 3x3 maxima are shared by the ND-2 scales scanned.  The rare
  //! classified sites go through finish_candidate().
  //! HBM traffic: 4*(ND+1) B read per pixel, nothing written but candidates.
  //! A queued site: key (2 words) + neighbourhood (kSiteNb words) + 2 pad.
  constexpr int kSiteQueueWords = 24;
  constexpr int kSiteQueueCap = 64;  // entries per wave (6 KB of LDS): any
                                     // iteration's hits fit an empty queue
  //! Moves `qn` (<= kSiteQueueCap) queued sites of one wave to the frame's list.
  __device__ inline void flush_sites(const unsigned* queue, int qn, int lane,
                                     int frame, const SiteLists& sites)
  {
    int base = 0;
    if (lane == 0)
      base = atomicAdd(&sites.count[frame], qn);
    base = __builtin_amdgcn_readfirstlane(base);
    __builtin_amdgcn_wave_barrier();
    // (unsigned: a counter that was not zeroed must not turn into a negative
    // offset - seen with a single-stream graph on the ROCm 7.0 runtime; the
    // step is then reported as failed, see SiteLists::error)
    if (base < 0 && lane == 0)
      *sites.error = 1;
    const int room = unsigned(base) < unsigned(sites.cap) ? min(qn, sites.cap - base) : 0;
    if (lane < room)
      sites.key[size_t(frame) * sites.cap + base + lane] =
          (unsigned long long) queue[lane * kSiteQueueWords] |
          ((unsigned long long) queue[lane * kSiteQueueWords + 1] << 32);
    float* nb = sites.nb + (size_t(frame) * sites.cap + base) * kSiteNb;
    for (int w = lane; w < room * kSiteNb; w += 64)
      nb[w] = __uint_as_float(queue[(w / kSiteNb) * kSiteQueueWords + 2 + w % kSiteNb]);
    __builtin_amdgcn_wave_barrier();
  }

  constexpr int kExtremaWavesPerEu = 4;  // 112 VGPRs: the register ring of 5 DoG layers
  //! NW: waves per workgroup = adjacent strips of the same rows, kept within a
  //! dozen rows of each other by a barrier.  A row segment of 128 columns starts
  //! anywhere in a cache line, so neighbouring strips share the line at their
  //! seam; single-wave workgroups drift apart by more rows than the L2 keeps
  //! and fetched those lines twice (FETCH_SIZE 4.77 GB per 64 x 1080p step for
  //! 4.23 GB of planes; launched together but unsynchronised: no change; with
  //! the barrier 4.38 GB, the kernel 1 % faster).
  //! The scan of one (strip, segment, frame) by one wave: the body of
  //! extrema_march_kernel (one octave per launch) and of
  //! extrema_march_multi_kernel (every octave of a frame in one launch).
  template <int ND, int PF, int NW>
  __device__ __forceinline__ void extrema_march_body(
      const OctaveView& gauss, const int octave, const ExtremaParams& p,
      const SiteLists& sites, const int seg_rows, const int strip, const int seg,
      const int b, unsigned* s_queue)
  {
    static_assert(PF == 3, "the row loop is unrolled 3x");
    int qn = 0;  // wave-uniform fill of the queue
    constexpr int NG = ND + 1;
    constexpr int STRIDE = 126;
    const int lane = threadIdx.x & 63;
    const int w = gauss.w, h = gauss.h;
    const int pad = p.img_padding_sz;
    const float* g = gauss.base + size_t(b) * gauss.frame_stride;
    const size_t plane = gauss.plane;

    const int x0 = strip * STRIDE;
    const int col = x0 + 2 * lane;
    const int mcol = min(col, w - 2);
    const int y0 = seg * seg_rows;
    const int y1 = min(h, y0 + seg_rows);
    const float thr8 = 0.8f * p.extremum_thres;

    // odd width: the lane whose pair starts at the last column loads the pair
    // (w-2, w-1) and keeps the second value as its first (rows are then only
    // 4-byte aligned: tools/ubench/unaligned_check.hip).  The select is applied
    // where the prefetched row is consumed - touching the registers right
    // after the load would make every load wait for its data.
    const bool odd_tail = col == w - 1;
    auto load_row = [&](int yy, float2 (&r)[NG]) {
      const int gy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
      const float* rowp = g + size_t(gy) * w + mcol;
#pragma unroll
      for (int l = 0; l < NG; ++l)
        r[l] = *reinterpret_cast<const float2*>(rowp + size_t(l) * plane);
    };

    float2 pg[PF][NG];
    float2 ring[3][ND];
#pragma unroll
    for (int q = 0; q < PF; ++q)
      load_row(y0 - 1 + q, pg[q]);
    const int T = (y1 - y0) + 2;  // source rows y0-1 .. y1

    for (int n0 = 0; n0 < T; n0 += 3)
    {
      if (NW > 1 && (n0 % 12) == 0)
        __builtin_amdgcn_s_barrier();  // the strips stay within a dozen rows
#pragma unroll
      for (int i = 0; i < 3; ++i)
      {
        const int n = n0 + i;
        const int yy = y0 - 1 + n;
        float2 cur[NG];
#pragma unroll
        for (int l = 0; l < NG; ++l)
          cur[l] = make_float2(odd_tail ? pg[i][l].y : pg[i][l].x, pg[i][l].y);
#pragma unroll
        for (int l = 0; l < ND; ++l)
          ring[i][l] = make_float2(cur[l + 1].x - cur[l].x, cur[l + 1].y - cur[l].y);
        load_row(min(yy + PF, y1), pg[i]);  // see gradient_polar_march_kernel

        const int y = yy - 1;
        const int ia = (i + 1) % 3, ib = (i + 2) % 3, ic = i;  // y-1, y, y+1
        if (n < 2 || y >= y1 || y < pad || y >= h - pad)
          continue;  // wave-uniform

        float2 m[ND], mn[ND];
#pragma unroll
        for (int l = 0; l < ND; ++l)
        {
          const float vx = fmaxf(fmaxf(ring[ia][l].x, ring[ib][l].x), ring[ic][l].x);
          const float vy = fmaxf(fmaxf(ring[ia][l].y, ring[ib][l].y), ring[ic][l].y);
          const float ux = fminf(fminf(ring[ia][l].x, ring[ib][l].x), ring[ic][l].x);
          const float uy = fminf(fminf(ring[ia][l].y, ring[ib][l].y), ring[ic][l].y);
          const float vl = shift_from_prev(vy), vr = shift_from_next(vx);
          const float ul = shift_from_prev(uy), ur = shift_from_next(ux);
          m[l] = make_float2(fmaxf(fmaxf(vl, vx), vy), fmaxf(fmaxf(vx, vy), vr));
          mn[l] = make_float2(fminf(fminf(ul, ux), uy), fminf(fminf(ux, uy), ur));
        }
        // bit 2 (s - 1) + c: (scale s, pixel c of this lane) is a classified
        // site; the same bit of maxbits: it is a maximum
        unsigned hitbits = 0u, maxbits = 0u;
#pragma unroll
        for (int s = 1; s <= ND - 2; ++s)
        {
#pragma unroll
          for (int c = 0; c < 2; ++c)
          {
            const float v = c == 0 ? ring[ib][s].x : ring[ib][s].y;
            const float M = c == 0 ? fmaxf(fmaxf(m[s - 1].x, m[s].x), m[s + 1].x)
                                   : fmaxf(fmaxf(m[s - 1].y, m[s].y), m[s + 1].y);
            const float N = c == 0 ? fminf(fminf(mn[s - 1].x, mn[s].x), mn[s + 1].x)
                                   : fminf(fminf(mn[s - 1].y, mn[s].y), mn[s + 1].y);
            const int x = col + c;
            // interior columns of the strip only (the outer two are halo)
            const bool mine = (c == 0 ? lane > 0 : lane < 63) && x >= pad &&
                              x < w - pad;
            const bool is_max = (v == M), is_min = (v == N);
            const bool hit = mine && !(fabsf(v) < thr8) && (is_max || is_min);
            hitbits |= hit ? 1u << (2 * (s - 1) + c) : 0u;
            maxbits |= is_max ? 1u << (2 * (s - 1) + c) : 0u;
          }
        }
        // Classified sites go to a wave-local LDS queue first and reach the
        // frame's list in batches: one returning atomic per batch instead of
        // one per site (every wave of a frame hits the same counter, and that
        // serialisation was the kernel's bottleneck).  The edge test and the
        // refinement run later in finish_sites_kernel, from the site's DoG
        // neighbourhood (SiteLists::nb), which is written here from the rows in
        // registers.  ONE copy of this code per row step, behind a rolled loop
        // over the (scale, pixel) slots with the layer picked by selects: inside
        // the unrolled classification loops above it stood 18 times in the
        // kernel, which then no longer fitted the instruction cache (11.5 k
        // instructions; scan 827 -> 914 us per 64 x 1080p step).  About 4 % of
        // the row steps get here.
        if (__builtin_expect(__ballot(hitbits != 0u) != 0ull, 0))
        {
#pragma clang loop unroll(disable)
          for (int k = 0; k < 2 * (ND - 2); ++k)
          {
            const bool hit = ((hitbits >> k) & 1u) != 0u;
            const unsigned long long hits = __ballot(hit);
            if (hits == 0ull)
              continue;
            const int s = 1 + (k >> 1), c = k & 1;  // wave-uniform
            const int nh = __popcll(hits);
            if (qn + nh > kSiteQueueCap)
            {
              flush_sites(s_queue, qn, lane, b, sites);
              qn = 0;
            }
            unsigned* e = s_queue + (qn + __popcll(hits & ((1ull << lane) - 1ull))) *
                                        kSiteQueueWords;
            if (hit)
            {
              const unsigned long long key =
                  ((((unsigned long long) (octave * kMaxScales + s) << 20 |
                     (unsigned) y)
                    << 20 |
                    (unsigned) (col + c))
                   << 1) |
                  ((maxbits >> k) & 1u);
              e[0] = unsigned(key);
              e[1] = unsigned(key >> 32);
            }
            // layer l of ring row r (l is wave-uniform; the register ring cannot
            // be indexed at run time without going to scratch)
            auto layer = [&](int r, int l) {
              float2 v2 = ring[r][0];
#pragma unroll
              for (int q = 1; q < ND; ++q)
                v2 = l == q ? ring[r][q] : v2;
              return v2;
            };
            // the whole wave executes the lane shifts - column x-1 / x+1 of a
            // lane's first / second pixel sit in the neighbouring lane - and
            // the lanes with a hit store, value by value
            auto put_row = [&](int slot, const float2 m2) {
              const float sp = shift_from_prev(m2.y), sn = shift_from_next(m2.x);
              if (hit)
              {
                e[2 + slot + 0] = __float_as_uint(c == 0 ? sp : m2.x);
                e[2 + slot + 1] = __float_as_uint(c == 0 ? m2.x : m2.y);
                e[2 + slot + 2] = __float_as_uint(c == 0 ? m2.y : sn);
              }
            };
            put_row(0, layer(ia, s));
            put_row(3, layer(ib, s));
            put_row(6, layer(ic, s));
#pragma unroll
            for (int dl = 0; dl < 2; ++dl)
            {
              const int l = dl == 0 ? s - 1 : s + 1;
              const float2 m2 = layer(ib, l), up = layer(ia, l), dn = layer(ic, l);
              const float sp = shift_from_prev(m2.y), sn = shift_from_next(m2.x);
              if (hit)
              {
                unsigned* o = e + 2 + 9 + 5 * dl;
                o[0] = __float_as_uint(c == 0 ? m2.x : m2.y);  // centre
                o[1] = __float_as_uint(c == 0 ? sp : m2.x);    // left
                o[2] = __float_as_uint(c == 0 ? m2.y : sn);    // right
                o[3] = __float_as_uint(c == 0 ? up.x : up.y);  // up
                o[4] = __float_as_uint(c == 0 ? dn.x : dn.y);  // down
              }
            }
            if (hit)
              e[2 + 19] = 0u;
            qn += nh;
            if (qn >= kSiteQueueCap / 2)
            {
              flush_sites(s_queue, qn, lane, b, sites);
              qn = 0;
            }
          }
        }
      }
    }
    if (qn > 0)
      flush_sites(s_queue, qn, lane, b, sites);
  }

  template <int ND, int PF, int NW>
  __global__ __launch_bounds__(64 * NW, kExtremaWavesPerEu) void extrema_march_kernel(
      OctaveView gauss, int octave, ExtremaParams p, SiteLists sites,
      int seg_rows, int nstrips, int nseg, int xcd_total)
  {
    __shared__ unsigned s_queue_all[NW][kSiteQueueCap * kSiteQueueWords];
    int strip, seg;
    size_t bb;
    if (!march_work_item((nstrips + NW - 1) / NW, nseg, xcd_total, strip, seg, bb))
      return;
    strip = strip * NW + int(threadIdx.x >> 6);
    if (NW > 1 && strip >= nstrips)
      return;  // surplus wave of the row's last group (strip_group_size)
    extrema_march_body<ND, PF, NW>(gauss, octave, p, sites, seg_rows, strip, seg,
                                   int(bb), s_queue_all[threadIdx.x >> 6]);
  }

  //! One frame per call (round 6): the scans of ALL octaves in one launch.  The
  //! octaves' (strip, segment) lists lie back to back in the grid; single-wave
  //! workgroups as in the small launches of extrema_march_kernel.
  struct ScanMultiArgs
  {
    OctaveView gauss[kScanMultiMax];
    int octave[kScanMultiMax];
    int seg_rows[kScanMultiMax];
    int nstrips[kScanMultiMax];
    int first_block[kScanMultiMax];
    int n;
  };
  template <int ND, int PF>
  __global__ __launch_bounds__(64, kExtremaWavesPerEu) void extrema_march_multi_kernel(
      ScanMultiArgs a, ExtremaParams p, SiteLists sites)
  {
    __shared__ unsigned s_queue[kSiteQueueCap * kSiteQueueWords];
    int k = 0;
#pragma unroll
    for (int i = 1; i < kScanMultiMax; ++i)
      if (i < a.n && int(blockIdx.x) >= a.first_block[i])
        k = i;
    const int local = int(blockIdx.x) - a.first_block[k];
    const int nstrips = a.nstrips[k];
    const int seg = local / nstrips;
    extrema_march_body<ND, PF, 1>(a.gauss[k], a.octave[k], p, sites, a.seg_rows[k],
                                  local - seg * nstrips, seg, int(blockIdx.y), s_queue);
  }


  //! Second half of the fast path: one thread per classified site.
  __global__ __launch_bounds__(256) void finish_sites_kernel(
      OctavePyramidView pyr, ExtremaParams p,
      const ScaleTable* __restrict__ tabp, SiteLists sites, CandidateLists cand)
  {
    const int b = blockIdx.y;
    const int n = min(sites.count[b], sites.cap);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
      return;
    const unsigned long long key = sites.key[size_t(b) * sites.cap + i];
    const int o = int(key >> 41) / kMaxScales;
    const int s = int(key >> 41) % kMaxScales;
    const int y = int((key >> 21) & 0xfffff);
    const int x = int((key >> 1) & 0xfffff);
    const DogOctave I{pyr.base[o] + size_t(b) * pyr.frame_stride[o], pyr.w[o],
                      pyr.h[o], pyr.plane[o], pyr.scales - 1};
    SiteNeighbourhood nb;
    const float4* r =
        reinterpret_cast<const float4*>(sites.nb + (size_t(b) * sites.cap + i) * kSiteNb);
#pragma unroll
    for (int q = 0; q < kSiteNb / 4; ++q)
    {
      const float4 t = r[q];
      nb.v[4 * q] = t.x;
      nb.v[4 * q + 1] = t.y;
      nb.v[4 * q + 2] = t.z;
      nb.v[4 * q + 3] = t.w;
    }
    finish_candidate(I, x, y, s, (key & 1ull) ? 1 : -1, o, b, p, *tabp, cand, &nb);
  }

  void launch_finish_sites(const OctavePyramidView& pyr, int batch,
                           const ExtremaParams& p, const ScaleTable* tab,
                           const SiteLists& sites, const CandidateLists& cand,
                           hipStream_t stream)
  {
    const dim3 grid((sites.cap + 255) / 256, batch);
    hipLaunchKernelGGL(finish_sites_kernel, grid, dim3(256), 0, stream, pyr, p,
                       tab, sites, cand);
  }

  void launch_extrema_scan(const OctaveView& gauss, int octave, int batch,
                           const ExtremaParams& p, const ScaleTable* tab,
                           const CandidateLists& cand, const SiteLists& sites,
                           hipStream_t stream)
  {
    const int nscan = gauss.scales - 3;  // DoG layers 1 .. (scales-1)-2
    if (nscan <= 0)
      return;
    // any width (odd ones: extrema_march_kernel's odd_tail)
    const bool wide_enough = gauss.w >= 4;
    // the Halide-branch classifier (signed_type) also classifies the border
    // pixels: it runs on the general path
    if (wide_enough && selection().feature_march && gauss.scales == 6 && !p.signed_type)
    {
      const int nstrips = (gauss.w - 2 + 125) / 126;
      int nseg = (g_extrema_waves + nstrips * batch - 1) / (nstrips * batch);
      constexpr int min_rows = 16;
      nseg = std::max(1, std::min(nseg, (gauss.h + min_rows - 1) / min_rows));
      const int seg_rows = (gauss.h + nseg - 1) / nseg;
      nseg = (gauss.h + seg_rows - 1) / seg_rows;
      // workgroups of 8 / 4 adjacent strips where the count divides (see the
      // kernel's NW) and the launch fills the chip anyway: a small launch (one
      // frame per call) keeps single-wave workgroups, which spread over all CUs
      // (SARA_HIP_STRIP_GROUP forces the groups for the parity tests)
      const int limit = strip_group_limit(nstrips * nseg * batch);
      const int NW = strip_group_size(nstrips, limit);
      const int gstrips = (nstrips + NW - 1) / NW;
      const int total = xcd_map_enabled() ? gstrips * nseg * batch : 0;
      const dim3 grid = total ? dim3(8 * ((total + 7) / 8)) : dim3(gstrips * nseg, batch);
#define SARA_SCAN(NW_)                                                         \
  hipLaunchKernelGGL((extrema_march_kernel<5, 3, NW_>), grid, dim3(64 * NW_),  \
                     0, stream, gauss, octave, p, sites, seg_rows, nstrips,    \
                     nseg, total)
      if (NW == 8)
        SARA_SCAN(8);
      else if (NW == 4)
        SARA_SCAN(4);
      else
        SARA_SCAN(1);
#undef SARA_SCAN
      return;
    }
    const dim3 block(64, 4);
    const dim3 grid((gauss.w + 63) / 64, (gauss.h + 3) / 4, batch * nscan);
    hipLaunchKernelGGL(extrema_scan_kernel, grid, block, 0, stream, gauss, octave,
                       nscan, p, tab, cand);
  }

  bool launch_extrema_scan_multi(const OctaveView* gauss, const int* octaves, int n,
                                 int batch, const ExtremaParams& p,
                                 const SiteLists& sites, hipStream_t stream)
  {
    if (n < 1 || n > kScanMultiMax || p.signed_type || !selection().feature_march)
      return false;
    ScanMultiArgs a{};
    a.n = n;
    int blocks = 0;
    for (int k = 0; k < n; ++k)
    {
      const OctaveView& g = gauss[k];
      if (g.scales != 6 || g.w < 4)
        return false;
      // the geometry launch_extrema_scan gives a launch of this size
      const int nstrips = (g.w - 2 + 125) / 126;
      int nseg = (g_extrema_waves + nstrips * batch - 1) / (nstrips * batch);
      constexpr int min_rows = 16;
      nseg = std::max(1, std::min(nseg, (g.h + min_rows - 1) / min_rows));
      const int seg_rows = (g.h + nseg - 1) / nseg;
      nseg = (g.h + seg_rows - 1) / seg_rows;
      a.gauss[k] = g;
      a.octave[k] = octaves[k];
      a.seg_rows[k] = seg_rows;
      a.nstrips[k] = nstrips;
      a.first_block[k] = blocks;
      blocks += nstrips * nseg;
    }
    hipLaunchKernelGGL((extrema_march_multi_kernel<5, 3>), dim3(blocks, batch), dim3(64), 0,
                       stream, a, p, sites);
    return true;
  }

  __global__ void extremum_map_kernel(const float* __restrict__ a,
                                      const float* __restrict__ b,
                                      const float* __restrict__ c, int w, int h,
                                      float edge_ratio, float thres, int pad,
                                      int8_t* __restrict__ out)
  {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h)
      return;
    const size_t i = size_t(y) * w + x;
    int type = 0;
    if (pad == 0)
    {
      // the Halide seam itself (shakti_scale_space_dog_extremum_32f_cpu)
      auto get = [&](int xx, int yy, int ds) {
        xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
        yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
        const float* l = ds < 0 ? a : (ds == 0 ? b : c);
        return l[size_t(yy) * w + xx];
      };
      type = halide_is_dog_extremum(get, x, y, edge_ratio, thres);
    }
    else if (pad <= x && x < w - pad && pad <= y && y < h - pad)
      type = classify_site(a + i, b + i, c + i, w, thres, edge_ratio);
    out[i] = int8_t(type);
  }

  void launch_extremum_map(const float* a, const float* b, const float* c,
                           int w, int h, float edge_ratio, float thres, int pad,
                           int8_t* out, hipStream_t stream)
  {
    const dim3 block(64, 4);
    const dim3 grid((w + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(extremum_map_kernel, grid, block, 0, stream, a, b, c, w,
                       h, edge_ratio, thres, pad, out);
  }

  // ------------------------------------------------------------------------ //
  // Ordering: the reference emits extrema in (octave, scale, raster) order
  // (DoG.cpp:62-82, RefineExtremum.cpp:497-515).  Keys are unique per frame,
  // so rank = number of smaller keys.
  // ------------------------------------------------------------------------ //
  // Comparing every key with every other key of its frame is O(n^2) (round 1:
  // 0.55 ms per 16 4K frames).  This is a counting sort on the key's (octave, scale, y) prefix - one
  // bucket per image row of every scanned plane - followed by a rank inside
  // the bucket, which holds a handful of keys: O(n + rows) instead of O(n^2).
  __device__ inline int key_row_bucket(unsigned long long key, const RowBuckets& rb)
  {
    return rb.base[int(key >> 41)] + int((key >> 21) & 0xfffff);
  }

  __global__ void bucket_count_kernel(CandidateLists cand, RowBuckets rb,
                                      int* __restrict__ hist)
  {
    const int b = blockIdx.y;
    const int n = min(cand.count[b], cand.cap);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
      return;
    const unsigned long long key = cand.key[size_t(b) * cand.cap + i];
    atomicAdd(hist + size_t(b) * rb.stride + key_row_bucket(key, rb), 1);
  }

  //! Exclusive scan of one frame's bucket counts (in place in `start`,
  //! rb.total + 1 entries) and a copy into `cursor`.
  __global__ __launch_bounds__(1024) void bucket_scan_kernel(
      RowBuckets rb, int* __restrict__ hist, int* __restrict__ cursor)
  {
    __shared__ int s_wave[16];
    __shared__ int s_carry;
    int* h = hist + size_t(blockIdx.x) * rb.stride;
    int* cur = cursor + size_t(blockIdx.x) * rb.stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0)
      s_carry = 0;
    __syncthreads();
    for (int base = 0; base <= rb.total; base += 1024)
    {
      const int i = base + tid;
      const int v = i < rb.total ? h[i] : 0;
      int incl = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1)
      {
        const int t = __shfl_up(incl, off);
        if (lane >= off)
          incl += t;
      }
      if (lane == 63)
        s_wave[wave] = incl;
      __syncthreads();
      int woff = 0;
      for (int q = 0; q < wave; ++q)
        woff += s_wave[q];
      const int carry = s_carry;
      const int excl = carry + woff + incl - v;
      if (i <= rb.total)
      {
        h[i] = excl;
        cur[i] = excl;
      }
      __syncthreads();
      if (tid == 1023)
        s_carry = carry + woff + incl;
      __syncthreads();
    }
  }

  __global__ void bucket_scatter_kernel(CandidateLists cand, RowBuckets rb,
                                        int* __restrict__ cursor,
                                        int* __restrict__ grouped)
  {
    const int b = blockIdx.y;
    const int n = min(cand.count[b], cand.cap);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
      return;
    const unsigned long long key = cand.key[size_t(b) * cand.cap + i];
    const int pos =
        atomicAdd(cursor + size_t(b) * rb.stride + key_row_bucket(key, rb), 1);
    grouped[size_t(b) * cand.cap + pos] = i;
  }

  __global__ void bucket_rank_kernel(CandidateLists cand, RowBuckets rb,
                                     const int* __restrict__ start,
                                     const int* __restrict__ grouped)
  {
    const int b = blockIdx.y;
    const int n = min(cand.count[b], cand.cap);
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n)
      return;
    const unsigned long long* keys = cand.key + size_t(b) * cand.cap;
    const int* g = grouped + size_t(b) * cand.cap;
    const int i = g[p];
    const unsigned long long mine = keys[i];
    const int* st = start + size_t(b) * rb.stride;
    const int bucket = key_row_bucket(mine, rb);
    const int lo = st[bucket], hi = st[bucket + 1];
    int rank = lo;
    for (int q = lo; q < hi; ++q)
      rank += (keys[g[q]] < mine);
    const size_t row = size_t(b) * cand.cap;
    cand.order[row + rank] = i;
    cand.skey[row + rank] = mine;
    cand.sdata[row + rank] = cand.data[row + i];
  }

  //! The four steps above in ONE launch: one 1024-thread workgroup per frame,
  //! the row buckets in LDS (total + 1 ints: 48 KB at 1080p, 100 KB at 4K).
  //! Saves two memsets and three kernel boundaries per step - what a
  //! one-frame batch mostly consists of - and is no slower on big batches
  //! (a frame has a few thousand keys).  `grouped`: [frame][cap] scratch.
  __global__ __launch_bounds__(1024) void bucket_sort_fused_kernel(
      CandidateLists cand, RowBuckets rb, int* __restrict__ grouped)
  {
    extern __shared__ int s_bucket[];  // rb.total + 1 counters, then 1024 partials
    int* s_part = s_bucket + rb.total + 1;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = min(cand.count[b], cand.cap);
    const size_t row = size_t(b) * cand.cap;
    const unsigned long long* keys = cand.key + row;
    int* g = grouped + row;
    int* tmp = cand.order + row;  // position inside the bucket, until the end
    for (int i = tid; i <= rb.total; i += 1024)
      s_bucket[i] = 0;
    __syncthreads();
    // 1. count: position of every key inside its bucket
    for (int i = tid; i < n; i += 1024)
      tmp[i] = atomicAdd(&s_bucket[key_row_bucket(keys[i], rb)], 1);
    __syncthreads();
    // 2. exclusive scan of the bucket counts (thread t owns a run of buckets)
    const int per = (rb.total + 1 + 1023) / 1024;
    const int lo = min(tid * per, rb.total + 1), hi = min(lo + per, rb.total + 1);
    int sum = 0;
    for (int i = lo; i < hi; ++i)
      sum += s_bucket[i];
    s_part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1)
    {
      const int t = tid >= off ? s_part[tid - off] : 0;
      __syncthreads();
      s_part[tid] += t;
      __syncthreads();
    }
    int run = s_part[tid] - sum;
    for (int i = lo; i < hi; ++i)
    {
      const int c = s_bucket[i];
      s_bucket[i] = run;
      run += c;
    }
    __syncthreads();
    // 3. scatter: keys grouped by bucket
    for (int i = tid; i < n; i += 1024)
      g[s_bucket[key_row_bucket(keys[i], rb)] + tmp[i]] = i;
    __threadfence_block();
    __syncthreads();
    // 4. rank inside the bucket (a handful of keys), sorted copies.  order[]
    //    held the in-bucket positions (tmp) until the barrier above: every
    //    thread has consumed its entries in step 3, so it can be overwritten.
    for (int p = tid; p < n; p += 1024)
    {
      const int i = g[p];
      const unsigned long long mine = keys[i];
      const int bucket = key_row_bucket(mine, rb);
      const int s0 = s_bucket[bucket], s1 = s_bucket[bucket + 1];
      int rank = s0;
      for (int q = s0; q < s1; ++q)
        rank += (keys[g[q]] < mine);
      cand.order[row + rank] = i;
      cand.skey[row + rank] = mine;
      cand.sdata[row + rank] = cand.data[row + i];
    }
  }

  //! The same ordering for a call on one or two frames, where the single
  //! 1024-thread workgroup per frame of the fused counting sort is a 19 us link
  //! in a chain of dependent launches (9 us here): rank = number of smaller
  //! keys, counted directly.  A workgroup ranks 16 keys with 16 lanes each; the frame's keys
  //! pass through LDS in chunks of 4096 (a 1080p frame has ~3 500: one chunk,
  //! ~220 comparisons per lane).  O(n^2 / lanes): only where n is a few
  //! thousand and the chip is otherwise idle.
  __global__ __launch_bounds__(256) void rank_small_kernel(CandidateLists cand)
  {
    constexpr int kChunk = 4096;
    __shared__ unsigned long long s_keys[kChunk];
    const int b = blockIdx.y;
    const int n = min(cand.count[b], cand.cap);
    if (int(blockIdx.x) * 16 >= n)
      return;
    const size_t row = size_t(b) * cand.cap;
    const unsigned long long* keys = cand.key + row;
    const int tid = threadIdx.x;
    const int i = blockIdx.x * 16 + (tid >> 4), sub = tid & 15;
    const unsigned long long mine = i < n ? keys[i] : ~0ull;
    int rank = 0;
    for (int base = 0; base < n; base += kChunk)
    {
      const int m = min(kChunk, n - base);
      if (base > 0)
        __syncthreads();
      for (int j = tid; j < m; j += 256)
        s_keys[j] = keys[base + j];
      __syncthreads();
      // eight independent LDS reads in flight per lane (a plain loop waits
      // for each read: 220 x the LDS latency)
      int j = sub;
      for (; j + 16 * 7 < m; j += 16 * 8)
      {
        unsigned long long k[8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
          k[q] = s_keys[j + 16 * q];
#pragma unroll
        for (int q = 0; q < 8; ++q)
          rank += (k[q] < mine);
      }
      for (; j < m; j += 16)
        rank += (s_keys[j] < mine);
    }
    // sum over the 16 lanes of a key (one DPP row)
    rank += __builtin_amdgcn_update_dpp(0, rank, 0x111, 0xf, 0xf, false);  // row_shr:1
    rank += __builtin_amdgcn_update_dpp(0, rank, 0x112, 0xf, 0xf, false);  // row_shr:2
    rank += __builtin_amdgcn_update_dpp(0, rank, 0x114, 0xf, 0xf, false);  // row_shr:4
    rank += __builtin_amdgcn_update_dpp(0, rank, 0x118, 0xf, 0xf, false);  // row_shr:8
    if (sub == 15 && i < n)
    {
      cand.order[row + rank] = i;
      cand.skey[row + rank] = mine;
      cand.sdata[row + rank] = cand.data[row + i];
    }
  }

  void launch_rank_candidates_bucketed(const CandidateLists& cand,
                                       const RowBuckets& rb, int* hist,
                                       int* cursor, int* grouped, int batch,
                                       hipStream_t stream)
  {
    if (batch <= 2 && cand.cap <= 32768)
    {
      hipLaunchKernelGGL(rank_small_kernel, dim3((cand.cap + 15) / 16, batch), dim3(256),
                         0, stream, cand);
      return;
    }
    // fused form when the buckets fit in LDS (images up to about 6000 rows
    // per octave-0 plane set), four launches beyond that (8K frames)
    const size_t lds = sizeof(int) * (size_t(rb.total) + 1 + 1024);
    if (lds <= 150 * 1024)
    {
      // the attribute is per DEVICE (one process may drive all GPUs of a
      // node, sara_hip_sift_group_*) and several host threads may get here
      static std::atomic<bool> allowed[64];
      int dev = 0;
      (void) hipGetDevice(&dev);
      if (lds > 64 * 1024 && !allowed[dev & 63].load(std::memory_order_acquire))
      {
        (void) hipFuncSetAttribute(
            reinterpret_cast<const void*>(bucket_sort_fused_kernel),
            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        allowed[dev & 63].store(true, std::memory_order_release);
      }
      hipLaunchKernelGGL(bucket_sort_fused_kernel, dim3(batch), dim3(1024), lds,
                         stream, cand, rb, grouped);
      return;
    }
    (void) hipMemsetAsync(hist, 0, sizeof(int) * size_t(batch) * rb.stride, stream);
    const dim3 grid((cand.cap + 255) / 256, batch);
    hipLaunchKernelGGL(bucket_count_kernel, grid, dim3(256), 0, stream, cand, rb,
                       hist);
    hipLaunchKernelGGL(bucket_scan_kernel, dim3(batch), dim3(1024), 0, stream, rb,
                       hist, cursor);
    hipLaunchKernelGGL(bucket_scatter_kernel, grid, dim3(256), 0, stream, cand,
                       rb, cursor, grouped);
    hipLaunchKernelGGL(bucket_rank_kernel, grid, dim3(256), 0, stream, cand, rb,
                       hist, grouped);
  }

  // ------------------------------------------------------------------------ //
  // Exhaustive self-check of the short device forms (device_math.hpp) against
  // the IEEE ones the rest of the parity chain was established with: every
  // non-negative float through the atanf reduction (look-up table + short
  // division vs select chains + IEEE division) and through the square root as
  // the gradient kernel uses it.  out[0], out[1] = mismatch counts.
  // ------------------------------------------------------------------------ //
  __global__ __launch_bounds__(256) void device_math_selfcheck_kernel(
      unsigned long long* __restrict__ out)
  {
    __shared__ __attribute__((aligned(16))) float s_lut[kAtanLutFloats];
    fill_atan_lut(s_lut, threadIdx.x, 256);
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;  // 2^24 threads
    unsigned bad_atan = 0, bad_sqrt = 0;
    for (uint32_t hi = 0; hi < 128; ++hi)
    {
      const uint32_t bits = hi * (1u << 24) + i;
      if (bits > 0x7f800000u)
        continue;
      const float x = __uint_as_float(bits);
      if (bits < 0x7f800000u)
      {
        const float a = atanf_nonneg_select(x);
        const float b = atanf_nonneg_lut(x, s_lut);
        bad_atan += __float_as_uint(a) != __float_as_uint(b);
      }
      const bool odd = sqrt_short_exponent(x) < kSqrtShortMinExponent ||
                       !(x < __builtin_inff());
      const float r = odd ? sqrtf(x) : sqrt_rn_short(x);
      bad_sqrt += __float_as_uint(r) != __float_as_uint(sqrtf(x));
    }
    if (bad_atan)
      atomicAdd(out, (unsigned long long) bad_atan);
    if (bad_sqrt)
      atomicAdd(out + 1, (unsigned long long) bad_sqrt);
  }

  //! Every float of [0, float(2 pi)] through the orientation kernel's bin
  //! computation (estimate + one correction against thr[]) and through the
  //! reference expression; *bad counts the mismatches.
  __global__ __launch_bounds__(256) void orientation_bin_selfcheck_kernel(
      const float* __restrict__ thr, unsigned long long* __restrict__ bad)
  {
    __shared__ float s_thr[40];
    if (threadIdx.x < 40)
      s_thr[threadIdx.x] = thr[threadIdx.x];
    __syncthreads();
    constexpr uint32_t kEnd = 0x40c90fdbu + 1u;  // float(2 pi) inclusive
    unsigned n_bad = 0;
    for (uint32_t bits = blockIdx.x * 256u + threadIdx.x; bits < kEnd;
         bits += gridDim.x * 256u)
    {
      const float a = __uint_as_float(bits);
      int want = int(floor(double(a / float(2 * M_PI) * kOriBins)));
      want %= kOriBins;
      int kb = int(a * float(kOriBins / (2. * M_PI)));
      kb = min(max(kb, 0), kOriBins);
      const float t0 = s_thr[kb], t1 = s_thr[kb + 1];
      kb += int(a >= t1) - int(a < t0);
      const int got = kb == kOriBins ? 0 : kb;
      n_bad += got != want;
    }
    if (n_bad)
      atomicAdd(bad, (unsigned long long) n_bad);
  }

  void launch_orientation_bin_selfcheck(const float* thr, unsigned long long* bad,
                                        hipStream_t stream)
  {
    hipLaunchKernelGGL(orientation_bin_selfcheck_kernel, dim3(256 * 64), dim3(256),
                       0, stream, thr, bad);
  }

  //! not_definite_enough3() of n matrices (9 floats each, row-major) -> 0 / 1.
  __global__ void definiteness_selfcheck_kernel(const float* __restrict__ H,
                                                const int* __restrict__ type,
                                                int n, unsigned char* out)
  {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
      return;
    float m[3][3];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        m[r][c] = H[9 * size_t(i) + 3 * r + c];
    out[i] = not_definite_enough3(m, type[i]) ? 1 : 0;
  }

  void launch_definiteness_selfcheck(const float* H, const int* type, int n,
                                     unsigned char* out, hipStream_t stream)
  {
    hipLaunchKernelGGL(definiteness_selfcheck_kernel, dim3((n + 255) / 256),
                       dim3(256), 0, stream, H, type, n, out);
  }

  void launch_device_math_selfcheck(unsigned long long* out, hipStream_t stream)
  {
    hipLaunchKernelGGL(device_math_selfcheck_kernel, dim3((1u << 24) / 256),
                       dim3(256), 0, stream, out);
  }

}  // namespace sara_hip
