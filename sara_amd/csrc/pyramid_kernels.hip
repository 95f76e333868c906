// gfx950 (MI355X / CDNA4) kernels of the SIFT front-end.
//
// Built with -ffp-contract=off: the CPU reference is compiled for baseline
// x86-64 (no FMA), and every kernel below evaluates its float expressions in
// the reference's operation order so that pyramids, DoG layers, extremum
// classification, refinement and polar gradients come out bit-identical.
//
// Wave = 64 lanes everywhere; workgroups are 256 threads = 4 waves.
#include "sift_kernels.hpp"

#include <atomic>

#include "device_math.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>

namespace sara_hip {

  // ======================================================================== //
  // Gaussian blur: rows then columns through LDS, replicate borders.
  // Reference: apply_gaussian_filter, ImageProcessing/LinearFiltering.cpp:30-68
  //            apply_row_based_filter / apply_column_based_filter / convolve_array,
  //            ImageProcessing/LinearFiltering.hpp:43-149.
  // Optional fused epilogue: dog = dst - src (GaussianPyramid.cpp:44-46).
  //
  // One workgroup produces a TX x TY tile.  The (TY+2R) x (TX+2R) source
  // window is staged in LDS once, row-filtered into a (TY+2R) x TX LDS tile
  // (each lane: 4 adjacent outputs from ds_read_b128 windows), then
  // column-filtered (each lane: 8 outputs of one column from a register
  // window).  Taps sit in SGPRs.  Accumulation is `sum += v * k` from 0.f in
  // ascending tap order, mul and add unfused, as in convolve_array.
  // ======================================================================== //
  constexpr int TX = 64;
  constexpr int TY = 32;
  constexpr int NT = 256;

  //! Marching segments are at least max(32, 4 R) rows tall: every segment
  //! re-filters 2R halo rows.
  constexpr int g_march_minrows = 4;
  // KernelSelection (sift_kernels.hpp) holds what used to be process-wide
  // switches here; the measurements behind its defaults:
  //  * march_waves / march2_waves - target number of waves per marching launch
  //    (the launch rounds up to whole segments).  4-column kernel (R <= 6,
  //    latency-bound: VALU pipe 26-34 % busy): 4 per SIMD - 1080p x 64: 116 ->
  //    111 us (R = 5), 135 -> 128 (R = 6), 287 -> 235 (first blur) against 2
  //    per SIMD, neutral on 720p and 4K.  2-column kernel (R >= 8, 86 / 100 /
  //    114 VGPRs: 5 / 4 / 4 waves per SIMD fit): 3072 -> 3840 waves on 1080p x
  //    64, one round of resident waves.  Round 6, tools/march2_waves_probe.py
  //    (64 x 1080p, octave 0, single stream): 2048 / 2560 / 3072 / 3584 / 4096
  //    / 5120 -> R = 8: 295 / 276 / 231 / 234 / 229 / 261 us, R = 10: 295 / 277
  //    / 262 / 267 / 299 / 269, R = 12: 335 / 307 / 301 / 307 / 330 / 299 (a
  //    launch of more waves than fit runs a second, short round); the whole
  //    stage on one stream 2.34 -> 2.11 ms.  With the per-octave streams the
  //    stage does not move (1.934 / 1.938 / 1.944 / 1.948 / 1.967 / 1.975 ms):
  //    the other octaves' launches fill the same wave slots - round 4 saw the
  //    same and kept 2048; 3072 is kept now because a single stream (and every
  //    per-kernel figure) gains 10 % and the overlapped stage loses nothing.
  //  * march_min_pixels - below this many pixels per launch (width x height x
  //    batch) the marching kernels cannot fill the chip - a wave is a serial
  //    chain of row steps - and the tiled kernel is used instead (240x135 x 64
  //    frames: 12-20 us per blur against 23-59 us).
  //  * strip_group (tests only): 0 = the production rule - strip-group
  //    workgroups (NW = 8 / 4) only for launches of >= 4096 / 2048 waves; 1, 4,
  //    8 = the largest group taken at EVERY launch size, so that the parity
  //    tests, whose images are small, run the grouped kernels.

  // Round 3: the tile geometry is a template parameter and the window is staged
  // with 16-byte loads.  One frame per call is a chain of dependent launches of
  // 3-17 MB each; tools/ubench/tile_blur_b1.hip times such chains: an empty
  // kernel costs 2.8 us per launch, round 2's kernel (64 x 32 tiles, 256
  // threads, one clamped 4-byte load per window element) 13.1 / 8.2 / 7.0 / 7.4
  // us at R = 6 on 1920x1080 / 960x540 / 480x270 / 240x135, and 9.8 us when it
  // only copies its tile through LDS - staging, not arithmetic, was the cost.
  // With the rows staged as float4 from the 16-byte column below x0 - R
  // (clamped element loads only in tiles that touch the left / right border)
  // and 512 threads per 64 x 32 tile: 10.9 / 5.3 us; the small octaves want
  // more, smaller tiles (a 240 x 135 plane is 20 tiles of 64 x 32 on 256 CUs):
  // 64 x 16 / 256 threads 4.1 us at 480x270, 32 x 16 / 128 threads 3.7 us at
  // 240x135.  launch_blur_r picks the geometry from the number of tiles.
  //! LDS floats the tile body needs: staged window, row-filtered window.
  template <int R, int TX_, int TY_>
  struct BlurTileLds
  {
    static constexpr int RP = ((R + 3) / 4) * 4;     // left halo rounded up to 16 bytes
    static constexpr int IW4 = (RP + TX_ + RP) / 4;  // float4 per staged row
    static constexpr int IP = IW4 * 4 + 4;           // row pitch, over-read safe
    static constexpr int IH = TY_ + 2 * R;
    static constexpr int in_floats = IH * IP;
    static constexpr int tmp_floats = IH * TX_;
  };

  //! One TX_ x TY_ output tile at (x0, y0) of one frame, by the NT_ threads of
  //! the workgroup (two workgroup barriers; threads outside the image return
  //! after the second).  src / dst / dog / dec address the frame.
  template <int R, int TX_, int TY_, int NT_, typename TapsT>
  __device__ __forceinline__ void blur_tile(
      const float* __restrict__ src, float* __restrict__ dst,
      float* __restrict__ dog, int w, int h, const TapsT& taps,
      float* __restrict__ dec, int x0, int y0, float* __restrict__ s_in,
      float* __restrict__ s_tmp)
  {
    using L = BlurTileLds<R, TX_, TY_>;
    constexpr int K = 2 * R + 1;
    constexpr int RP = L::RP;
    constexpr int D = RP - R;
    constexpr int IW4 = L::IW4;
    constexpr int IP = L::IP;
    constexpr int IH = L::IH;
    constexpr int NQ = (D + 4 + 2 * R + 3) / 4;  // b128 reads per 4 outputs
    constexpr int CR = TX_ * TY_ / NT_;       // rows per thread, column pass
    static_assert(TX_ % 4 == 0 && (TX_ * TY_) % NT_ == 0 && NT_ % TX_ == 0, "geometry");

    const int tid = threadIdx.x;

    // Stage the clamped source window, columns x0 - RP .. x0 + TX + RP - 1.
    const bool inside = x0 - RP >= 0 && x0 + TX_ + RP <= w;  // block-uniform
    for (int idx = tid; idx < IH * IW4; idx += NT_)
    {
      const int r = idx / IW4;
      const int c4 = idx - r * IW4;
      int gy = y0 - R + r;
      gy = gy < 0 ? 0 : (gy > h - 1 ? h - 1 : gy);
      const float* rowp = src + size_t(gy) * w;
      const int gx = x0 - RP + 4 * c4;
      float4 t;
      if (inside)  // 16-byte load; rows of any width (element-aligned is enough)
        t = *reinterpret_cast<const float4*>(rowp + gx);
      else
      {
        const int xa = min(max(gx, 0), w - 1), xb = min(max(gx + 1, 0), w - 1);
        const int xc = min(max(gx + 2, 0), w - 1), xd = min(max(gx + 3, 0), w - 1);
        t = make_float4(rowp[xa], rowp[xb], rowp[xc], rowp[xd]);
      }
      *reinterpret_cast<float4*>(&s_in[r * IP + 4 * c4]) = t;
    }
    __syncthreads();

    // Row pass.
    for (int it = tid; it < IH * (TX_ / 4); it += NT_)
    {
      const int r = it / (TX_ / 4);
      const int q = it - r * (TX_ / 4);
      float v[NQ * 4];
      const float4* p = reinterpret_cast<const float4*>(&s_in[r * IP + 4 * q]);
#pragma unroll
      for (int m = 0; m < NQ; ++m)
      {
        const float4 t = p[m];
        v[4 * m + 0] = t.x;
        v[4 * m + 1] = t.y;
        v[4 * m + 2] = t.z;
        v[4 * m + 3] = t.w;
      }
      float acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
      {
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j)
          sum += v[D + i + j] * taps.k[j];
        acc[i] = sum;
      }
      *reinterpret_cast<float4*>(&s_tmp[r * TX_ + 4 * q]) =
          make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    __syncthreads();

    // Column pass: thread -> column tx, CR consecutive rows.
    const int tx = tid % TX_;
    const int yq = tid / TX_;
    constexpr int NV = CR + 2 * R;
    float v[NV];
#pragma unroll
    for (int m = 0; m < NV; ++m)
      v[m] = s_tmp[(yq * CR + m) * TX_ + tx];

    const int gx = x0 + tx;
    if (gx >= w)
      return;
#pragma unroll
    for (int i = 0; i < CR; ++i)
    {
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < K; ++j)
        sum += v[i + j] * taps.k[j];
      const int gy = y0 + yq * CR + i;
      if (gy < h)
      {
        dst[size_t(gy) * w + gx] = sum;
        if (dog)
          dog[size_t(gy) * w + gx] = sum - s_in[(yq * CR + i + R) * IP + RP + tx];
        // nearest-neighbour half of the output = first plane of the next
        // octave (Resize.cpp:45-84: int(x * (w / (w/2))) == 2x), see the
        // marching kernel's DEC
        if (dec && ((gx | gy) & 1) == 0 && (gx >> 1) < (w >> 1) && (gy >> 1) < (h >> 1))
          dec[size_t(gy >> 1) * (w >> 1) + (gx >> 1)] = sum;
      }
    }
  }

  template <int R, int TX_, int TY_, int NT_>
  __global__ __launch_bounds__(NT_) void gaussian_blur_kernel(
      const float* __restrict__ src, size_t src_stride,
      float* __restrict__ dst, size_t dst_stride, float* __restrict__ dog,
      size_t dog_stride, int w, int h, Taps taps, float* __restrict__ dec,
      size_t dec_stride)
  {
    using L = BlurTileLds<R, TX_, TY_>;
    __shared__ __attribute__((aligned(16))) float s_in[L::in_floats];
    __shared__ __attribute__((aligned(16))) float s_tmp[L::tmp_floats];
    const size_t b = blockIdx.z;
    blur_tile<R, TX_, TY_, NT_>(src + b * src_stride, dst + b * dst_stride,
                                dog ? dog + b * dog_stride : nullptr, w, h, taps,
                                dec ? dec + b * dec_stride : nullptr,
                                blockIdx.x * TX_, blockIdx.y * TY_, s_in, s_tmp);
  }

  // ------------------------------------------------------------------------ //
  // One frame per call (round 6): the blurs of DIFFERENT octaves that sit at
  // the same depth of the pyramid's dependency graph in ONE launch.
  //
  // compute_sift_keypoints once per video frame (SfM/Odometry/OdometryPipeline
  // .cpp:82-90) replays a HIP graph; the host hands the graph's ~40 kernel
  // nodes to the queues at ~3.3 us each, which is what bounded the call (a
  // context with ONE octave - the critical chain alone - takes 0.127 / 0.209
  // ms for pyramid + extrema / full SIFT, four octaves 0.193 / 0.270 although
  // octaves 1-3 could hide under octave 0's chain: tools/b1_floor.py).
  // Octave o + 1 starts from G(downscale_index, o), so blur s of octave o and
  // blur s - downscale_index of octave o + 1 are independent: they form one
  // LEVEL.  A level is one kernel node whose workgroups are dealt to the
  // level's members (tile lists back to back); dependencies stay at kernel
  // boundaries - no grid barrier, no cooperative launch (round 5's cooperative
  // kernel paid a kernel boundary per phase and more).  24 blur nodes become
  // 11 for the default schedule.  Every member runs the tile body above
  // unchanged (64 x 32 tiles, 512 threads): bit-identical planes.
  // ------------------------------------------------------------------------ //
  constexpr int kLevelMaxRadius = 12;
  struct LevelTaps
  {
    int size;
    float k[2 * kLevelMaxRadius + 1];
  };
  struct BlurLevelMember
  {
    const float* src;
    float* dst;
    float* dec;  // G(0, o + 1) when this blur produces G(downscale_index, o)
    unsigned long long src_stride, dst_stride, dec_stride;  // per frame, floats
    int w, h, tiles_x, first_block;
    LevelTaps taps;
  };
  struct BlurLevelArgs
  {
    int n;
    int pad;
    BlurLevelMember m[kBlurLevelMaxMembers];
  };

  //! Bit of a radius in a level's radius set (0: not compiled for levels).
  __host__ __device__ constexpr unsigned level_radius_bit(int R)
  {
    return R == 5 ? 1u : R == 6 ? 2u : R == 8 ? 4u : R == 10 ? 8u : R == 12 ? 16u : 0u;
  }
  //! Largest radius of a radius set (LDS and registers follow it: a level of
  //! small blurs keeps the footprint - workgroups per CU - of the plain kernels).
  __host__ __device__ constexpr int level_max_radius(unsigned mask)
  {
    return (mask & 16u) ? 12 : (mask & 8u) ? 10 : (mask & 4u) ? 8 : (mask & 2u) ? 6 : 5;
  }

  //! RMASK: the radii of the level's members (level_radius_bit); only those
  //! tile bodies are compiled into the instance.
  template <unsigned RMASK>
  __global__ __launch_bounds__(512) void gaussian_blur_level_kernel(BlurLevelArgs a)
  {
    using L = BlurTileLds<level_max_radius(RMASK), 64, 32>;
    __shared__ __attribute__((aligned(16))) float s_in[L::in_floats];
    __shared__ __attribute__((aligned(16))) float s_tmp[L::tmp_floats];
    int k = 0;
#pragma unroll
    for (int i = 1; i < kBlurLevelMaxMembers; ++i)
      if (i < a.n && int(blockIdx.x) >= a.m[i].first_block)
        k = i;
    const BlurLevelMember& m = a.m[k];  // workgroup-uniform
    const int local = int(blockIdx.x) - m.first_block;
    const int ty = local / m.tiles_x;
    const int x0 = (local - ty * m.tiles_x) * 64, y0 = ty * 32;
    const size_t b = blockIdx.y;
    const float* src = m.src + b * m.src_stride;
    float* dst = m.dst + b * m.dst_stride;
    float* dec = m.dec ? m.dec + b * m.dec_stride : nullptr;
    const int R = m.taps.size / 2;
#define SARA_LEVEL_CASE(r)                                                     \
  if constexpr ((RMASK & level_radius_bit(r)) != 0)                            \
  {                                                                            \
    if (RMASK == level_radius_bit(r) || R == r)                                \
    {                                                                          \
      blur_tile<r, 64, 32, 512>(src, dst, nullptr, m.w, m.h, m.taps, dec, x0,  \
                                y0, s_in, s_tmp);                              \
      return;                                                                  \
    }                                                                          \
  }
    SARA_LEVEL_CASE(5)
    SARA_LEVEL_CASE(6)
    SARA_LEVEL_CASE(8)
    SARA_LEVEL_CASE(10)
    SARA_LEVEL_CASE(12)
#undef SARA_LEVEL_CASE
  }

  bool blur_level_radius_ok(int taps_size)
  {
    const int R = taps_size / 2;
    return taps_size == 2 * R + 1 && (R == 5 || R == 6 || R == 8 || R == 10 || R == 12);
  }

  bool launch_blur_level(const BlurLevelBlur* blurs, int n, int batch,
                         hipStream_t stream)
  {
    if (n < 1 || n > kBlurLevelMaxMembers)
      return false;
    BlurLevelArgs a{};
    a.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i)
    {
      const BlurLevelBlur& bl = blurs[i];
      if (!blur_level_radius_ok(bl.taps->size))
        return false;
      BlurLevelMember& m = a.m[i];
      m.src = bl.src;
      m.dst = bl.dst;
      m.dec = bl.dec;
      m.src_stride = bl.src_stride;
      m.dst_stride = bl.dst_stride;
      m.dec_stride = bl.dec_stride;
      m.w = bl.w;
      m.h = bl.h;
      m.tiles_x = (bl.w + 63) / 64;
      m.first_block = blocks;
      m.taps.size = bl.taps->size;
      for (int j = 0; j < bl.taps->size; ++j)
        m.taps.k[j] = bl.taps->k[j];
      blocks += m.tiles_x * ((bl.h + 31) / 32);
    }
    unsigned mask = 0;
    for (int i = 0; i < n; ++i)
      mask |= level_radius_bit(a.m[i].taps.size / 2);
    const dim3 grid(blocks, batch);
    switch (mask)
    {
#define SARA_LEVEL_MASK(mk)                                                    \
  case mk:                                                                     \
    hipLaunchKernelGGL(gaussian_blur_level_kernel<mk>, grid, dim3(512), 0, stream, a); \
    break;
      SARA_LEVEL_MASK(1) SARA_LEVEL_MASK(2) SARA_LEVEL_MASK(3) SARA_LEVEL_MASK(4)
      SARA_LEVEL_MASK(5) SARA_LEVEL_MASK(6) SARA_LEVEL_MASK(7) SARA_LEVEL_MASK(8)
      SARA_LEVEL_MASK(9) SARA_LEVEL_MASK(10) SARA_LEVEL_MASK(11) SARA_LEVEL_MASK(12)
      SARA_LEVEL_MASK(13) SARA_LEVEL_MASK(14) SARA_LEVEL_MASK(15) SARA_LEVEL_MASK(16)
      SARA_LEVEL_MASK(17) SARA_LEVEL_MASK(18) SARA_LEVEL_MASK(19) SARA_LEVEL_MASK(20)
      SARA_LEVEL_MASK(21) SARA_LEVEL_MASK(22) SARA_LEVEL_MASK(23) SARA_LEVEL_MASK(24)
      SARA_LEVEL_MASK(25) SARA_LEVEL_MASK(26) SARA_LEVEL_MASK(27) SARA_LEVEL_MASK(28)
      SARA_LEVEL_MASK(29) SARA_LEVEL_MASK(30) SARA_LEVEL_MASK(31)
#undef SARA_LEVEL_MASK
    default:
      return false;
    }
    return true;
  }

  //! Any radius up to kMaxRadius: same structure, runtime loops, dynamic LDS.
  __global__ __launch_bounds__(NT) void gaussian_blur_generic_kernel(
      const float* __restrict__ src, size_t src_stride,
      float* __restrict__ dst, size_t dst_stride, float* __restrict__ dog,
      size_t dog_stride, int w, int h, Taps taps)
  {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    const int R = taps.size / 2;
    const int K = taps.size;
    const int IW = TX + 2 * R;
    const int IH = TY + 2 * R;
    float* s_in = s_dyn;
    float* s_tmp = s_dyn + IH * IW;

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TX;
    const int y0 = blockIdx.y * TY;
    const size_t b = blockIdx.z;
    src += b * src_stride;
    dst += b * dst_stride;
    if (dog)
      dog += b * dog_stride;

    for (int idx = tid; idx < IH * IW; idx += NT)
    {
      const int r = idx / IW;
      const int c = idx - r * IW;
      int gy = y0 - R + r;
      int gx = x0 - R + c;
      gy = gy < 0 ? 0 : (gy > h - 1 ? h - 1 : gy);
      gx = gx < 0 ? 0 : (gx > w - 1 ? w - 1 : gx);
      s_in[idx] = src[size_t(gy) * w + gx];
    }
    __syncthreads();
    for (int idx = tid; idx < IH * TX; idx += NT)
    {
      const int r = idx / TX;
      const int c = idx - r * TX;
      float sum = 0.f;
      for (int j = 0; j < K; ++j)
        sum += s_in[r * IW + c + j] * taps.k[j];
      s_tmp[idx] = sum;
    }
    __syncthreads();
    for (int idx = tid; idx < TY * TX; idx += NT)
    {
      const int r = idx / TX;
      const int c = idx - r * TX;
      float sum = 0.f;
      for (int j = 0; j < K; ++j)
        sum += s_tmp[(r + j) * TX + c] * taps.k[j];
      const int gx = x0 + c, gy = y0 + r;
      if (gx < w && gy < h)
      {
        dst[size_t(gy) * w + gx] = sum;
        if (dog)
          dog[size_t(gy) * w + gx] = sum - s_in[(r + R) * IW + c + R];
      }
    }
  }

  // ------------------------------------------------------------------------ //
  // Marching Gaussian blur (the fast path; widths that are multiples of 4).
  //
  // One wave owns a strip of 256 columns (4 per lane) and marches down
  // `seg_rows` output rows.  Per source row: the (strip + 2R halo) segment is
  // loaded PF rows ahead into registers, staged in a small LDS ring, every
  // lane row-filters its 4 adjacent columns from ds_read_b128 windows, and the
  // column filter is a ring of K = 2R+1 partial sums per column held in
  // registers: the row arriving at step n adds tap j to the output row that is
  // j steps old.  Taps therefore accumulate in ascending order from 0.f
  // exactly like convolve_array (LinearFiltering.hpp:43-63), and each source
  // row is row-filtered once per segment instead of once per 32-row tile.
  // Ring positions are compile-time constants because the row loop is
  // unrolled K times; the prefetch registers are re-aligned once per K rows.
  //
  // gfx950 counts loads and stores on one in-order counter (vmcnt), so a wave
  // that consumes a load issued one step ago also waits for that step's
  // stores to be acknowledged; the PF-deep prefetch keeps ~PF rows of loads
  // and stores in flight per wave (measured: 1.9-3.2 TB/s at depth 2).
  //
  // HBM traffic per output pixel: 4 B read (x (1 + 2R/seg_rows) halo rows),
  // 4 B write.  The DoG layers are NOT materialised: their consumers (extremum
  // scan, refinement) subtract on the fly, see feature_kernels.hip.
  // ------------------------------------------------------------------------ //
  //! DEC: also write the nearest-neighbour half of the output,
  //! dec(x, y) = dst(2x, 2y) - exactly downscale(dst, 2) (Resize.cpp:45-84:
  //! int(x * (w / (w/2))) == 2x for every x < w/2), i.e. the first plane of
  //! the next octave, without a separate pass over HBM.
  //! FMA (opt-in, SARA_HIP_OPT_FMA_BLUR): sum = fma(v, k, sum) instead of the
  //! reference's separately rounded multiply and add - half the arithmetic
  //! instructions, results within 2e-7 of the range instead of bit-exact.
  //! U8: `src` addresses 8-bit gray frames (src_stride in bytes); every value
  //! is converted as it is loaded, float(v) / 255.f (ChannelConversion.hpp:40-54)
  //! - the first blur of the pyramid then reads the uploaded frame itself and
  //! the separate conversion pass (1 B read + 4 B written per pixel) is gone.
  template <int R, int PF, bool DEC, bool FMA = false, bool U8 = false, int NW = 1>
  __global__ __launch_bounds__(64 * NW, R <= 6 ? 4 : 1) void gaussian_blur_march_kernel(
      const float* __restrict__ src, size_t src_stride,
      float* __restrict__ dst, size_t dst_stride, float* __restrict__ dec,
      size_t dec_stride, int w, int h, int seg_rows, int nstrips, int nseg,
      int xcd_total, Taps taps)
  {
    constexpr int CPL = 4;
    constexpr int K = 2 * R + 1;
    constexpr int W = 64 * CPL;
    constexpr int RP = ((R + 3) / 4) * 4;  // left halo padded for alignment
    constexpr int D = RP - R;
    constexpr int ROWF = RP + W + RP;      // floats per LDS row slot
    constexpr int NQ = (D + CPL + 2 * R + 3) / 4;  // b128 reads per window
    __shared__ __attribute__((aligned(16))) float s_row_all[NW][2 * ROWF];
    float* s_row = s_row_all[threadIdx.x >> 6];

    const int lane = threadIdx.x & 63;
    int strip, seg;
    size_t b;
    if (!march_work_item((nstrips + NW - 1) / NW, nseg, xcd_total, strip, seg, b))
      return;
    strip = strip * NW + int(threadIdx.x >> 6);
    if (NW > 1 && strip >= nstrips)
      return;  // surplus wave of the row's last group (strip_group_size)
    const unsigned char* src8 =
        reinterpret_cast<const unsigned char*>(src) + b * src_stride;
    src += b * src_stride;
    dst += b * dst_stride;
    if (DEC)
      dec += b * dec_stride;
    const int dw = w / 2, dh = h / 2;

    // the last strip is moved left so that it is full: its first columns are
    // computed twice (two waves store identical values) instead of leaving
    // lanes idle and taking the replicated-column path below
    const int x0 = (w >= W) ? min(strip * W, w - W) : strip * W;
    const int y0 = seg * seg_rows;
    const int y1 = min(h, y0 + seg_rows);
    const int col = x0 + CPL * lane;
    // w % 4 == 0: a float4 is all in or all out; any other width comes here
    // with w >= W only, where every strip is full (the last one moved left).
    // Rows and strips are then only 4-byte aligned, which gfx950's 16-byte
    // global loads / stores take at 96-100 % of the aligned rate
    // (tools/ubench/unaligned_check.hip).
    const bool col_ok = col < w;
    // halo column of this lane (lanes < 2R)
    int hcol = lane < R ? x0 - R + lane : x0 + W + (lane - R);
    hcol = hcol < 0 ? 0 : (hcol > w - 1 ? w - 1 : hcol);
    const int hslot = lane < R ? RP - R + lane : RP + W + (lane - R);
    const int mcol = col_ok ? col : w - 4;

    auto load_row = [&](int yy, float4& m, float& hv) {
      const int gy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
      if (U8)
      {
        const unsigned char* rowp = src8 + size_t(gy) * w;
        const uchar4 q = *reinterpret_cast<const uchar4*>(rowp + mcol);
        m = make_float4(float(q.x) / 255.f, float(q.y) / 255.f, float(q.z) / 255.f,
                        float(q.w) / 255.f);
        if (!col_ok)
          m = make_float4(m.w, m.w, m.w, m.w);
        hv = 0.f;
        if (lane < 2 * R)
          hv = float(rowp[hcol]) / 255.f;
        return;
      }
      const float* rowp = src + size_t(gy) * w;
      m = *reinterpret_cast<const float4*>(rowp + mcol);
      if (!col_ok)
        m = make_float4(m.w, m.w, m.w, m.w);  // replicate src(w-1, y)
      hv = 0.f;
      if (lane < 2 * R)
        hv = rowp[hcol];
    };

    float A[K][CPL];
    float4 pm[PF];
    float phv[PF];
    const int T = (y1 - y0) + 2 * R;  // source rows y0-R .. y1+R-1

#pragma unroll
    for (int q = 0; q < PF; ++q)
      load_row(y0 - R + q, pm[q], phv[q]);

    for (int n0 = 0; n0 < T; n0 += K)
    {
      if (NW > 1)
        __builtin_amdgcn_s_barrier();  // the strips of a workgroup stay within K rows
#pragma unroll
      for (int i = 0; i < K; ++i)
      {
        const int n = n0 + i;
        if (n >= T)
          return;  // wave-uniform: the last round of K steps is rarely full
        const int yy = y0 - R + n;
        // stage source row n (loaded PF steps ago) and refill its slot
        float* rowbuf = s_row + (n & 1) * ROWF;
        *reinterpret_cast<float4*>(rowbuf + RP + CPL * lane) = pm[i % PF];
        if (lane < 2 * R)
          rowbuf[hslot] = phv[i % PF];
        // (the last source row a segment consumes is y1 + R - 1: rows behind it
        // are asked for as that row again, a cache hit, not streamed from HBM)
        load_row(min(yy + PF, y1 + R - 1), pm[i % PF], phv[i % PF]);

        // row pass on source row n
        float t[CPL];
        {
          const float4* p =
              reinterpret_cast<const float4*>(rowbuf + CPL * lane);
          float v[NQ * 4];
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const float4 x = p[q];
            v[4 * q + 0] = x.x;
            v[4 * q + 1] = x.y;
            v[4 * q + 2] = x.z;
            v[4 * q + 3] = x.w;
          }
#pragma unroll
          for (int c = 0; c < CPL; ++c)
          {
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < K; ++j)
              sum = FMA ? __builtin_fmaf(v[D + c + j], taps.k[j], sum)
                        : sum + v[D + c + j] * taps.k[j];
            t[c] = sum;
          }
        }

        // column pass: tap j goes to the output that is j steps old
#pragma unroll
        for (int j = 0; j < K; ++j)
        {
          const int sl = (i + K - 1 - j) % K;
#pragma unroll
          for (int c = 0; c < CPL; ++c)
          {
            if (j == 0)
              A[sl][c] = 0.f + t[c] * taps.k[0];
            else
              A[sl][c] = FMA ? __builtin_fmaf(t[c], taps.k[j], A[sl][c])
                             : A[sl][c] + t[c] * taps.k[j];
          }
        }

        const int o = yy - R;
        if ((o >= y0) && (o < y1) && col_ok)
        {
          *reinterpret_cast<float4*>(dst + size_t(o) * w + col) =
              make_float4(A[i][0], A[i][1], A[i][2], A[i][3]);
          if (DEC && (o & 1) == 0 && (o >> 1) < dh)
          {
            float* dp = dec + size_t(o >> 1) * dw;
            if ((col & 1) == 0)  // wave-uniform: col - x0 is a multiple of 4
              *reinterpret_cast<float2*>(dp + (col >> 1)) =
                  make_float2(A[i][0], A[i][2]);
            else
            {
              // odd width, last strip (x0 = w - 256 is odd): the even source
              // columns are the lane's second and fourth, and the last one,
              // w - 1, has no place in a plane of w / 2 columns
              const int dx = (col + 1) >> 1;
              dp[dx] = A[i][1];
              if (dx + 1 < dw)
                dp[dx + 1] = A[i][3];
            }
          }
        }
      }
      // re-align the prefetch ring: the row of step n0+K+q sits in slot
      // (K+q) % PF and must be found in slot q % PF by the next round.
      if (K % PF != 0)
      {
        float4 tm[PF];
        float th[PF];
#pragma unroll
        for (int q = 0; q < PF; ++q)
        {
          tm[q] = pm[(K + q) % PF];
          th[q] = phv[(K + q) % PF];
        }
#pragma unroll
        for (int q = 0; q < PF; ++q)
        {
          pm[q] = tm[q];
          phv[q] = th[q];
        }
      }
    }
  }

  template <int R>
  static void launch_blur_march(const float* src, size_t src_stride, float* dst,
                                size_t dst_stride, float* dec, size_t dec_stride,
                                int w, int h, int batch, const Taps& taps,
                                hipStream_t stream, bool fma = false,
                                bool src_is_u8 = false)
  {
    constexpr int W = 256;
    // prefetch depth: the K x 4 partial-sum ring dominates the register
    // budget, keep the kernel at >= 3 waves/SIMD (<= 168 VGPRs)
#ifndef SARA_MARCH_PF
#define SARA_MARCH_PF 4
#endif
    constexpr int PF = R >= 12 ? 2 : ((R >= 10 || R == 6) ? 3 : SARA_MARCH_PF);
    const int nstrips = (w + W - 1) / W;
    // enough waves to fill 256 CUs x 3-4 waves/SIMD, segments >= 32 rows
    // segments: enough waves to fill the chip, but every segment re-filters
    // 2R halo rows, so keep them at least g_march_minrows * R rows tall
    int nseg = (selection().march_waves + nstrips * batch - 1) / (nstrips * batch);
    const int min_rows = std::max(32, g_march_minrows * R);
    nseg = std::max(1, std::min(nseg, (h + min_rows - 1) / min_rows));
    const int seg_rows = (h + nseg - 1) / nseg;
    nseg = (h + seg_rows - 1) / seg_rows;
    // Workgroups of 8 / 4 adjacent strips, kept within K rows of each other by
    // a barrier per round (the NW of the kernel), when the launch fills the
    // chip anyway: the halo columns a strip reads from its neighbours' lines
    // are then L2 hits.  64 x 1080p: R = 5 338 -> 308 us per step, R = 6 + half
    // size 392 -> 367 (8 waves; 4 are neutral); a small launch (one frame per
    // call) keeps single-wave workgroups, which spread over all CUs.
    // (the wave counts keep the groups to launches that fill the chip;
    // SARA_HIP_STRIP_GROUP forces them for the parity tests)
    const int limit = (src_is_u8 || fma) ? 1 : strip_group_limit(nstrips * nseg * batch);
    const int NW = strip_group_size(nstrips, limit);
    const int gstrips = (nstrips + NW - 1) / NW;
    const int total = xcd_map_enabled() ? gstrips * nseg * batch : 0;
    const dim3 grid = total ? dim3(8 * ((total + 7) / 8)) : dim3(gstrips * nseg, batch);
    if (NW == 8 || NW == 4)
    {
#define SARA_MARCH_NW(DEC_, NW_)                                               \
  hipLaunchKernelGGL((gaussian_blur_march_kernel<R, PF, DEC_, false, false, NW_>), \
                     grid, dim3(64 * NW_), 0, stream, src, src_stride, dst,    \
                     dst_stride, dec, dec_stride, w, h, seg_rows, nstrips,     \
                     nseg, total, taps)
      if (NW == 8 && dec)
        SARA_MARCH_NW(true, 8);
      else if (NW == 8)
        SARA_MARCH_NW(false, 8);
      else if (dec)
        SARA_MARCH_NW(true, 4);
      else
        SARA_MARCH_NW(false, 4);
#undef SARA_MARCH_NW
      return;
    }
#define SARA_MARCH_LAUNCH(DEC_, FMA_)                                          \
  hipLaunchKernelGGL((gaussian_blur_march_kernel<R, PF, DEC_, FMA_>), grid,    \
                     dim3(64), 0, stream, src, src_stride, dst, dst_stride,    \
                     dec, dec_stride, w, h, seg_rows, nstrips, nseg, total,    \
                     taps)
    if (src_is_u8)
    {
      // gray8 source (the base blur of the first octave): exact arithmetic,
      // no fused half-size output
      hipLaunchKernelGGL((gaussian_blur_march_kernel<R, PF, false, false, true>),
                         grid, dim3(64), 0, stream, src, src_stride, dst,
                         dst_stride, dec, dec_stride, w, h, seg_rows, nstrips,
                         nseg, total, taps);
      return;
    }
    if (dec && fma)
      SARA_MARCH_LAUNCH(true, true);
    else if (dec)
      SARA_MARCH_LAUNCH(true, false);
    else if (fma)
      SARA_MARCH_LAUNCH(false, true);
    else
      SARA_MARCH_LAUNCH(false, false);
#undef SARA_MARCH_LAUNCH
  }

  // ------------------------------------------------------------------------ //
  // Hand-scheduled marching blur for the wide kernels (R >= 8), where the
  // arithmetic, not HBM, is the bound (tools/ubench/blur_limits.hip).
  //
  // Same marching scheme as above with three changes:
  //  * 2 columns per lane (strips of 128): the K x 2 partial-sum ring leaves
  //    room for 4-5 waves per SIMD instead of 3 (on gfx950 each of a SIMD's
  //    two VALU pipes serves its own waves, so an odd count idles half a pipe);
  //  * the column pass shares the product of the symmetric taps: k[j] and
  //    k[K-1-j] are the same float (make_gaussian_kernel evaluates exp() of the
  //    same x^2 and divides by the same sum), so t * k[j] is formed once and
  //    added to the two outputs it belongs to - R+1 multiplies instead of
  //    2R+1, each output still accumulating its taps in ascending order;
  //  * both passes are written as asm volatile blocks so that the order (and
  //    with it the register pressure) is the one written here: multiplies run
  //    >= 4 instructions ahead of the add that consumes them.
  // v_mul_f32 / v_add_f32 are IEEE single operations: the results are
  // bit-identical to the compiler-scheduled kernel (checked by the GPU tests
  // against the oracle and, kernel against kernel, by blur_limits).
  // ------------------------------------------------------------------------ //
  //! s0 += sum_{q<4} v[q] k[q],  s1 += sum_{q<4} v[q+1] k[q]  (ascending q)
  __device__ __forceinline__ void row4(float& s0, float& s1, float va, float vb,
                                       float vc, float vd, float ve, float k0,
                                       float k1, float k2, float k3)
  {
    float p0, p1, p2, p3;
    asm volatile("v_mul_f32 %2, %11, %6\n\tv_mul_f32 %3, %11, %7\n\t"
                 "v_mul_f32 %4, %12, %7\n\tv_mul_f32 %5, %12, %8\n\t"
                 "v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\t"
                 "v_mul_f32 %2, %13, %8\n\tv_mul_f32 %3, %13, %9\n\t"
                 "v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5\n\t"
                 "v_mul_f32 %4, %14, %9\n\tv_mul_f32 %5, %14, %10\n\t"
                 "v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\t"
                 "v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5"
                 : "+v"(s0), "+v"(s1), "=&v"(p0), "=&v"(p1), "=&v"(p2),
                   "=&v"(p3)
                 : "v"(va), "v"(vb), "v"(vc), "v"(vd), "v"(ve), "v"(k0),
                   "v"(k1), "v"(k2), "v"(k3));
  }
  __device__ __forceinline__ void row1(float& s0, float& s1, float va, float vb,
                                       float k0)
  {
    float p0, p1;
    asm volatile("v_mul_f32 %2, %6, %4\n\tv_mul_f32 %3, %6, %5\n\t"
                 "v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3"
                 : "+v"(s0), "+v"(s1), "=&v"(p0), "=&v"(p1)
                 : "v"(va), "v"(vb), "v"(k0));
  }
  //! taps j and j+1 (neither first nor centre) of two columns: a = output j
  //! steps old, b = K-1-j steps old, c / d the same for j+1.
  __device__ __forceinline__ void col2(float& a0, float& a1, float& b0,
                                       float& b1, float& c0, float& c1,
                                       float& d0, float& d1, float t0, float t1,
                                       float kj, float kj1)
  {
    float p0, p1, p2, p3;
    asm volatile("v_mul_f32 %8, %14, %12\n\tv_mul_f32 %9, %14, %13\n\t"
                 "v_mul_f32 %10, %15, %12\n\tv_mul_f32 %11, %15, %13\n\t"
                 "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %9\n\t"
                 "v_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %9\n\t"
                 "v_add_f32 %4, %4, %10\n\tv_add_f32 %5, %5, %11\n\t"
                 "v_add_f32 %6, %6, %10\n\tv_add_f32 %7, %7, %11"
                 : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1),
                   "+v"(d0), "+v"(d1), "=&v"(p0), "=&v"(p1), "=&v"(p2),
                   "=&v"(p3)
                 : "v"(t0), "v"(t1), "v"(kj), "v"(kj1));
  }
  __device__ __forceinline__ void col1(float& a0, float& a1, float& b0,
                                       float& b1, float t0, float t1, float kj)
  {
    float p0, p1;
    asm volatile("v_mul_f32 %4, %8, %6\n\tv_mul_f32 %5, %8, %7\n\t"
                 "v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5\n\t"
                 "v_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %5"
                 : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "=&v"(p0), "=&v"(p1)
                 : "v"(t0), "v"(t1), "v"(kj));
  }
  //! first tap (new = 0.f + t k[0]; the oldest output gets the same product
  //! as its last tap) and centre tap (mid += t k[R]).
  __device__ __forceinline__ void col_ends(float& new0, float& new1,
                                           float& old0, float& old1,
                                           float& mid0, float& mid1, float t0,
                                           float t1, float k0, float kr)
  {
    float p2, p3;
    asm volatile("v_mul_f32 %0, %10, %8\n\tv_mul_f32 %1, %10, %9\n\t"
                 "v_mul_f32 %6, %11, %8\n\tv_mul_f32 %7, %11, %9\n\t"
                 "v_add_f32 %2, %2, %0\n\tv_add_f32 %3, %3, %1\n\t"
                 "v_add_f32 %0, 0, %0\n\tv_add_f32 %1, 0, %1\n\t"
                 "v_add_f32 %4, %4, %6\n\tv_add_f32 %5, %5, %7"
                 : "=&v"(new0), "=&v"(new1), "+v"(old0), "+v"(old1),
                   "+v"(mid0), "+v"(mid1), "=&v"(p2), "=&v"(p3)
                 : "v"(t0), "v"(t1), "v"(k0), "v"(kr));
  }

  // The same two passes with fused multiply-adds (SARA_HIP_OPT_FMA_BLUR: NOT
  // bit-exact), in the same hand-written form and at the same occupancy as the
  // exact kernel - 50 instead of 88 arithmetic instructions per pixel (no
  // product can be shared between outputs any more).  It exists to answer one
  // question: is the exact kernel bound by the number of instructions it
  // issues?  (Round 2's FMA variant was compiler-scheduled at another
  // occupancy and proved nothing either way.)
  __device__ __forceinline__ void rowf4(float& s0, float& s1, float va, float vb,
                                        float vc, float vd, float ve, float k0,
                                        float k1, float k2, float k3)
  {
    asm volatile("v_fma_f32 %0, %2, %7, %0\n\tv_fma_f32 %1, %3, %7, %1\n\t"
                 "v_fma_f32 %0, %3, %8, %0\n\tv_fma_f32 %1, %4, %8, %1\n\t"
                 "v_fma_f32 %0, %4, %9, %0\n\tv_fma_f32 %1, %5, %9, %1\n\t"
                 "v_fma_f32 %0, %5, %10, %0\n\tv_fma_f32 %1, %6, %10, %1"
                 : "+v"(s0), "+v"(s1)
                 : "v"(va), "v"(vb), "v"(vc), "v"(vd), "v"(ve), "v"(k0), "v"(k1),
                   "v"(k2), "v"(k3));
  }
  __device__ __forceinline__ void rowf1(float& s0, float& s1, float va, float vb,
                                        float k0)
  {
    asm volatile("v_fma_f32 %0, %2, %4, %0\n\tv_fma_f32 %1, %3, %4, %1"
                 : "+v"(s0), "+v"(s1)
                 : "v"(va), "v"(vb), "v"(k0));
  }
  //! taps j .. j+3 of two columns into the four outputs that are j .. j+3 steps old
  __device__ __forceinline__ void colf4(float& a0, float& a1, float& b0, float& b1,
                                        float& c0, float& c1, float& d0, float& d1,
                                        float t0, float t1, float ka, float kb,
                                        float kc, float kd)
  {
    asm volatile("v_fma_f32 %0, %8, %10, %0\n\tv_fma_f32 %1, %9, %10, %1\n\t"
                 "v_fma_f32 %2, %8, %11, %2\n\tv_fma_f32 %3, %9, %11, %3\n\t"
                 "v_fma_f32 %4, %8, %12, %4\n\tv_fma_f32 %5, %9, %12, %5\n\t"
                 "v_fma_f32 %6, %8, %13, %6\n\tv_fma_f32 %7, %9, %13, %7"
                 : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1),
                   "+v"(d0), "+v"(d1)
                 : "v"(t0), "v"(t1), "v"(ka), "v"(kb), "v"(kc), "v"(kd));
  }
  __device__ __forceinline__ void colf1(float& a0, float& a1, float t0, float t1,
                                        float ka)
  {
    asm volatile("v_fma_f32 %0, %2, %4, %0\n\tv_fma_f32 %1, %3, %4, %1"
                 : "+v"(a0), "+v"(a1)
                 : "v"(t0), "v"(t1), "v"(ka));
  }

  template <int R, int PF, bool FMA = false>
  __global__ __launch_bounds__(64, 1) void gaussian_blur_march2_kernel(
      const float* __restrict__ src, size_t src_stride,
      float* __restrict__ dst, size_t dst_stride, int w, int h, int seg_rows,
      int nstrips, int nseg, int xcd_total, Taps taps)
  {
    constexpr int CPL = 2;
    constexpr int K = 2 * R + 1;
    constexpr int W = 64 * CPL;
    constexpr int RP = ((R + 1) / 2) * 2;  // left halo padded to 8 bytes
    constexpr int D = RP - R;
    constexpr int ROWF = RP + W + RP;
    constexpr int NQ = (D + CPL + 2 * R + 1) / 2;  // b64 reads per window
    __shared__ __attribute__((aligned(16))) float s_row[2 * ROWF];

    const int lane = threadIdx.x;
    int strip, seg;
    size_t b;
    if (!march_work_item(nstrips, nseg, xcd_total, strip, seg, b))
      return;
    src += b * src_stride;
    dst += b * dst_stride;

    const int x0 = (w >= W) ? min(strip * W, w - W) : strip * W;  // see above
    const int y0 = seg * seg_rows;
    const int y1 = min(h, y0 + seg_rows);
    const int col = x0 + CPL * lane;
    // w % 2 == 0: a float2 is all in or all out; odd widths only with w >= W
    // (every strip full, see gaussian_blur_march_kernel)
    const bool col_ok = col < w;
    int hcol = lane < R ? x0 - R + lane : x0 + W + (lane - R);
    hcol = hcol < 0 ? 0 : (hcol > w - 1 ? w - 1 : hcol);
    const int hslot = lane < R ? RP - R + lane : RP + W + (lane - R);
    const int mcol = col_ok ? col : w - CPL;

    auto load_row = [&](int yy, float2& m, float& hv) {
      const int gy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
      const float* rowp = src + size_t(gy) * w;
      m = *reinterpret_cast<const float2*>(rowp + mcol);
      if (!col_ok)
        m.x = m.y;  // replicate src(w-1, y)
      hv = 0.f;
      if (lane < 2 * R)
        hv = rowp[hcol];
    };

    // The R+1 distinct taps live in VGPRs: a VALU instruction with an SGPR
    // source issues at half the rate of an all-VGPR one when its neighbours
    // read SGPRs too (tools/ubench/valu_ops.hip), and half of this kernel's
    // instructions are multiplies by a tap.
    float tk[R + 1];
#pragma unroll
    for (int j = 0; j <= R; ++j)
      asm volatile("v_mov_b32 %0, %1" : "=v"(tk[j]) : "s"(taps.k[j]));
#define SARA_TK(j) tk[(j) <= R ? (j) : 2 * R - (j)]
    float A[K][CPL];
    float2 pm[PF];
    float phv[PF];
    const int T = (y1 - y0) + 2 * R;  // source rows y0-R .. y1+R-1

#pragma unroll
    for (int q = 0; q < PF; ++q)
      load_row(y0 - R + q, pm[q], phv[q]);

    for (int n0 = 0; n0 < T; n0 += K)
    {
#pragma unroll
      for (int i = 0; i < K; ++i)
      {
        const int n = n0 + i;
        if (n >= T)
          return;  // wave-uniform: the last round of K steps is rarely full
        const int yy = y0 - R + n;
        float* rowbuf = s_row + (n & 1) * ROWF;
        *reinterpret_cast<float2*>(rowbuf + RP + CPL * lane) = pm[i % PF];
        if (lane < 2 * R)
          rowbuf[hslot] = phv[i % PF];
        // (the last source row a segment consumes is y1 + R - 1: rows behind it
        // are asked for as that row again, a cache hit, not streamed from HBM)
        load_row(min(yy + PF, y1 + R - 1), pm[i % PF], phv[i % PF]);

        float v[NQ * 2];
        {
          const float2* p =
              reinterpret_cast<const float2*>(rowbuf + CPL * lane);
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const float2 x = p[q];
            v[2 * q] = x.x;
            v[2 * q + 1] = x.y;
          }
        }
        // row pass on source row n
        float t0 = 0.f, t1 = 0.f;
        if constexpr (FMA)
        {
          // opt-in fused form: one v_fma per tap and output, hand-written
          // like the exact form below
          constexpr int NB = K / 4;
#pragma unroll
          for (int q = 0; q < NB; ++q)
          {
            const int j = 4 * q;
            rowf4(t0, t1, v[D + j], v[D + j + 1], v[D + j + 2], v[D + j + 3],
                  v[D + j + 4], SARA_TK(j), SARA_TK(j + 1), SARA_TK(j + 2),
                  SARA_TK(j + 3));
          }
#pragma unroll
          for (int j = 4 * NB; j < K; ++j)
            rowf1(t0, t1, v[D + j], v[D + j + 1], SARA_TK(j));
#define SARA_SLF(j) ((i + K - 1 - (j)) % K)
          // tap 0 opens a new output: 0.f + t k[0]
          A[SARA_SLF(0)][0] = 0.f;
          A[SARA_SLF(0)][1] = 0.f;
          colf1(A[SARA_SLF(0)][0], A[SARA_SLF(0)][1], t0, t1, SARA_TK(0));
#pragma unroll
          for (int q = 0; q < (K - 1) / 4; ++q)
          {
            const int j = 1 + 4 * q;
            colf4(A[SARA_SLF(j)][0], A[SARA_SLF(j)][1], A[SARA_SLF(j + 1)][0],
                  A[SARA_SLF(j + 1)][1], A[SARA_SLF(j + 2)][0], A[SARA_SLF(j + 2)][1],
                  A[SARA_SLF(j + 3)][0], A[SARA_SLF(j + 3)][1], t0, t1, SARA_TK(j),
                  SARA_TK(j + 1), SARA_TK(j + 2), SARA_TK(j + 3));
          }
          static_assert((K - 1) % 4 == 0, "radii 8, 10, 12: K - 1 is a multiple of 4");
#undef SARA_SLF
        }
        else
        {
          constexpr int NB = K / 4;
#pragma unroll
          for (int q = 0; q < NB; ++q)
          {
            const int j = 4 * q;
            row4(t0, t1, v[D + j], v[D + j + 1], v[D + j + 2], v[D + j + 3],
                 v[D + j + 4], SARA_TK(j), SARA_TK(j + 1), SARA_TK(j + 2),
                 SARA_TK(j + 3));
          }
#pragma unroll
          for (int j = 4 * NB; j < K; ++j)
            row1(t0, t1, v[D + j], v[D + j + 1], SARA_TK(j));
        // column pass: tap j goes to the output that is j steps old, and the
        // same product, as tap K-1-j, to the one that is K-1-j steps old
#define SARA_SL(j) ((i + K - 1 - (j)) % K)
        col_ends(A[SARA_SL(0)][0], A[SARA_SL(0)][1], A[SARA_SL(K - 1)][0],
                 A[SARA_SL(K - 1)][1], A[SARA_SL(R)][0], A[SARA_SL(R)][1], t0,
                 t1, SARA_TK(0), SARA_TK(R));
#pragma unroll
        for (int j = 1; j + 1 < R; j += 2)
          col2(A[SARA_SL(j)][0], A[SARA_SL(j)][1], A[SARA_SL(K - 1 - j)][0],
               A[SARA_SL(K - 1 - j)][1], A[SARA_SL(j + 1)][0],
               A[SARA_SL(j + 1)][1], A[SARA_SL(K - 2 - j)][0],
               A[SARA_SL(K - 2 - j)][1], t0, t1, SARA_TK(j), SARA_TK(j + 1));
        if ((R - 1) % 2 == 1)
          col1(A[SARA_SL(R - 1)][0], A[SARA_SL(R - 1)][1], A[SARA_SL(R + 1)][0],
               A[SARA_SL(R + 1)][1], t0, t1, SARA_TK(R - 1));
#undef SARA_SL
        }

        const int o = yy - R;
        if ((o >= y0) && (o < y1) && col_ok)
          *reinterpret_cast<float2*>(dst + size_t(o) * w + col) =
              make_float2(A[i][0], A[i][1]);
      }
      if (K % PF != 0)
      {
        float2 tm[PF];
        float th[PF];
#pragma unroll
        for (int q = 0; q < PF; ++q)
        {
          tm[q] = pm[(K + q) % PF];
          th[q] = phv[(K + q) % PF];
        }
#pragma unroll
        for (int q = 0; q < PF; ++q)
        {
          pm[q] = tm[q];
          phv[q] = th[q];
        }
      }
    }
  }

#undef SARA_TK

  template <int R>
  static void launch_blur_march2(const float* src, size_t src_stride, float* dst,
                                 size_t dst_stride, int w, int h, int batch,
                                 const Taps& taps, hipStream_t stream,
                                 bool fma = false)
  {
    constexpr int W = 128;
    constexpr int PF = 4;
    const int nstrips = (w + W - 1) / W;
    int nseg = (selection().march2_waves + nstrips * batch - 1) / (nstrips * batch);
    const int min_rows = std::max(32, g_march_minrows * R);
    nseg = std::max(1, std::min(nseg, (h + min_rows - 1) / min_rows));
    const int seg_rows = (h + nseg - 1) / nseg;
    nseg = (h + seg_rows - 1) / seg_rows;
    // (workgroups of several strips, as in launch_blur_march, slow these
    // issue-bound kernels down: R = 10 435 -> 573 us per step with 5 / 8 waves)
    const int total = xcd_map_enabled() ? nstrips * nseg * batch : 0;
    const dim3 grid = total ? dim3(8 * ((total + 7) / 8)) : dim3(nstrips * nseg, batch);
    if (fma)
      hipLaunchKernelGGL((gaussian_blur_march2_kernel<R, PF, true>), grid,
                         dim3(64), 0, stream, src, src_stride, dst, dst_stride, w,
                         h, seg_rows, nstrips, nseg, total, taps);
    else
      hipLaunchKernelGGL((gaussian_blur_march2_kernel<R, PF, false>), grid,
                         dim3(64), 0, stream, src, src_stride, dst, dst_stride, w,
                         h, seg_rows, nstrips, nseg, total, taps);
  }

  template <int R, int TX_, int TY_, int NT_>
  static void launch_blur_geom(const float* src, size_t src_stride, float* dst,
                               size_t dst_stride, float* dog, size_t dog_stride,
                               int w, int h, int batch, const Taps& taps,
                               hipStream_t stream, float* dec, size_t dec_stride)
  {
    const dim3 grid((w + TX_ - 1) / TX_, (h + TY_ - 1) / TY_, batch);
    hipLaunchKernelGGL((gaussian_blur_kernel<R, TX_, TY_, NT_>), grid, dim3(NT_), 0,
                       stream, src, src_stride, dst, dst_stride, dog, dog_stride, w,
                       h, taps, dec, dec_stride);
  }

  template <int R>
  static void launch_blur_r(const float* src, size_t src_stride, float* dst,
                            size_t dst_stride, float* dog, size_t dog_stride,
                            int w, int h, int batch, const Taps& taps,
                            hipStream_t stream, float* dec, size_t dec_stride)
  {
    // enough tiles for the 256 CUs (gaussian_blur_kernel's header has the
    // measurements behind the thresholds)
    const long long tiles6432 =
        (long long) ((w + 63) / 64) * ((h + 31) / 32) * batch;
    const int forced = selection().tile_geometry;  // config 5's tile-shape sweep
    const int geom = forced >= 1 && forced <= 3
                         ? forced - 1
                         : (tiles6432 >= 200 ? 0 : (tiles6432 >= 48 ? 1 : 2));
    if (geom == 0)
      launch_blur_geom<R, 64, 32, 512>(src, src_stride, dst, dst_stride, dog,
                                       dog_stride, w, h, batch, taps, stream, dec,
                                       dec_stride);
    else if (geom == 1)
      launch_blur_geom<R, 64, 16, 256>(src, src_stride, dst, dst_stride, dog,
                                       dog_stride, w, h, batch, taps, stream, dec,
                                       dec_stride);
    else
      launch_blur_geom<R, 32, 16, 128>(src, src_stride, dst, dst_stride, dog,
                                       dog_stride, w, h, batch, taps, stream, dec,
                                       dec_stride);
  }

  bool launch_gaussian_blur_gray8(const unsigned char* src, size_t src_stride,
                                  float* dst, size_t dst_stride, int w, int h,
                                  int batch, const Taps& taps, hipStream_t stream)
  {
    const int R = taps.size / 2;
    const bool big_enough = size_t(w) * h * batch >= selection().march_min_pixels;
    // any width from one full strip up (element-aligned vector accesses), or
    // a multiple of 4
    const bool ok = big_enough && selection().blur_march &&
                    ((w % 4 == 0 && w >= 4) || w >= 256);
    if (!ok)
      return false;
    const float* s = reinterpret_cast<const float*>(src);
    switch (R)
    {
    case 5:
      launch_blur_march<5>(s, src_stride, dst, dst_stride, nullptr, 0, w, h, batch,
                           taps, stream, false, true);
      return true;
    case 6:
      launch_blur_march<6>(s, src_stride, dst, dst_stride, nullptr, 0, w, h, batch,
                           taps, stream, false, true);
      return true;
    default:
      return false;  // other radii: the caller converts first
    }
  }

  bool launch_gaussian_blur(const float* src, size_t src_stride, float* dst,
                            size_t dst_stride, float* dog, size_t dog_stride,
                            int w, int h, int batch, const Taps& taps,
                            hipStream_t stream, float* dec, size_t dec_stride,
                            bool fma)
  {
    const int R = taps.size / 2;
    // fast path: strips of float4 / float2 columns.  A width that is not a
    // multiple of the vector leaves a partial vector at the end of the row -
    // unless the row holds at least one full strip: the last strip is moved
    // left and every vector is whole.  Alignment is not a condition: rows of
    // such widths start at any multiple of 4 bytes and gfx950 takes 8- and
    // 16-byte global accesses there at 96-100 % of the aligned rate
    // (tools/ubench/unaligned_check.hip).  Round 2 sent these widths to the
    // tiled kernel: 64 x 1366 x 768 ran the pyramid 1.9x slower than 1368.
    const bool big_enough = size_t(w) * h * batch >= selection().march_min_pixels;
    const bool base_ok = big_enough && dog == nullptr;
    const bool march4_ok = base_ok && ((w % 4 == 0 && w >= 4) || w >= 256);
    const bool march2_ok = base_ok && ((w % 4 == 0 && w >= 4) || w >= 128);
    // the hand-scheduled kernel shares the products of mirrored taps
    bool symmetric = true;
    for (int j = 0; j < R; ++j)
      symmetric &= std::memcmp(&taps.k[j], &taps.k[2 * R - j], sizeof(float)) == 0;
    if (march2_ok && selection().blur_march && dec == nullptr && symmetric)
    {
      switch (R)
      {
      case 8:
        launch_blur_march2<8>(src, src_stride, dst, dst_stride, w, h, batch,
                              taps, stream, fma);
        return false;
      case 10:
        launch_blur_march2<10>(src, src_stride, dst, dst_stride, w, h, batch,
                               taps, stream, fma);
        return false;
      case 12:
        launch_blur_march2<12>(src, src_stride, dst, dst_stride, w, h, batch,
                               taps, stream, fma);
        return false;
      default:
        break;
      }
    }
    if (march4_ok && selection().blur_march)
    {
#define SARA_MARCH_CASE(r)                                                     \
  case r:                                                                      \
    launch_blur_march<r>(src, src_stride, dst, dst_stride, dec, dec_stride, w, \
                         h, batch, taps, stream, fma);                         \
    return dec != nullptr;
      switch (R)
      {
        SARA_MARCH_CASE(5)
        SARA_MARCH_CASE(6)
        SARA_MARCH_CASE(8)
        SARA_MARCH_CASE(10)
        SARA_MARCH_CASE(12)
      default:
        break;
      }
#undef SARA_MARCH_CASE
    }
#define SARA_BLUR_CASE(r)                                                      \
  case r:                                                                      \
    launch_blur_r<r>(src, src_stride, dst, dst_stride, dog, dog_stride, w, h,  \
                     batch, taps, stream, dec, dec_stride);                    \
    return dec != nullptr;
    switch (R)
    {
      SARA_BLUR_CASE(1)
      SARA_BLUR_CASE(2)
      SARA_BLUR_CASE(3)
      SARA_BLUR_CASE(4)
      SARA_BLUR_CASE(5)
      SARA_BLUR_CASE(6)
      SARA_BLUR_CASE(7)
      SARA_BLUR_CASE(8)
      SARA_BLUR_CASE(9)
      SARA_BLUR_CASE(10)
      SARA_BLUR_CASE(11)
      SARA_BLUR_CASE(12)
      SARA_BLUR_CASE(13)
      SARA_BLUR_CASE(14)
      SARA_BLUR_CASE(15)
      SARA_BLUR_CASE(16)
    default:
      break;
    }
#undef SARA_BLUR_CASE
    const dim3 grid((w + TX - 1) / TX, (h + TY - 1) / TY, batch);
    const size_t lds =
        sizeof(float) * size_t(TY + 2 * R) * (size_t(TX + 2 * R) + TX);
    // radii above ~36 need more than the default 64 KB of dynamic LDS
    // per DEVICE, and reachable from several host threads (sara_hip_sift_group_*)
    static std::atomic<bool> lds_allowed[64];
    int dev = 0;
    (void) hipGetDevice(&dev);
    if (lds > 64 * 1024 && !lds_allowed[dev & 63].load(std::memory_order_acquire))
    {
      (void) hipFuncSetAttribute(
          reinterpret_cast<const void*>(gaussian_blur_generic_kernel),
          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      lds_allowed[dev & 63].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(gaussian_blur_generic_kernel, grid, dim3(NT), lds, stream,
                       src, src_stride, dst, dst_stride, dog, dog_stride, w, h,
                       taps);
    return false;
  }

  // ======================================================================== //
  // Resize / copy / subtract.
  // ======================================================================== //

  //! scale(): ImageProcessing/Resize.cpp:45-60.
  __global__ void scale_kernel(const float* __restrict__ src, size_t src_stride,
                               int sw, int sh, float* __restrict__ dst,
                               size_t dst_stride, int dw, int dh)
  {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= dw || y >= dh)
      return;
    const size_t b = blockIdx.z;
    const float sx = float(sw) / float(dw);
    const float sy = float(sh) / float(dh);
    const int xi = int(float(x) * sx);
    const int yi = int(float(y) * sy);
    dst[b * dst_stride + size_t(y) * dw + x] =
        src[b * src_stride + size_t(yi) * sw + xi];
  }

  void launch_scale(const float* src, size_t src_stride, int sw, int sh,
                    float* dst, size_t dst_stride, int dw, int dh, int batch,
                    hipStream_t stream)
  {
    const dim3 block(64, 4);
    const dim3 grid((dw + 63) / 64, (dh + 3) / 4, batch);
    hipLaunchKernelGGL(scale_kernel, grid, block, 0, stream, src, src_stride, sw,
                       sh, dst, dst_stride, dw, dh);
  }

  //! enlarge(): ImageProcessing/Resize.cpp:110-126 + interpolate(),
  //! ImageProcessing/Interpolation.hpp:33-78 (bilinear in double, far border
  //! replicated, x-fastest accumulation).
  __global__ void enlarge_kernel(const float* __restrict__ src,
                                 size_t src_stride, int sw, int sh,
                                 float* __restrict__ dst, size_t dst_stride,
                                 int dw, int dh)
  {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= dw || y >= dh)
      return;
    const size_t b = blockIdx.z;
    const float* s = src + b * src_stride;
    const double scx = double(sw) / double(dw);
    const double scy = double(sh) / double(dh);
    const double px = double(x) * scx;
    const double py = double(y) * scy;
    const double ipx = trunc(px), ipy = trunc(py);
    const double fx = px - ipx, fy = py - ipy;
    const int x0 = int(ipx), y0 = int(ipy);
    double value = 0.;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx)
      {
        double weight = 1.;
        weight *= (dx == 0) ? (1. - fx) : fx;
        weight *= (dy == 0) ? (1. - fy) : fy;
        const int xx = (x0 + dx < sw) ? x0 + dx : x0 + dx - 1;
        const int yy = (y0 + dy < sh) ? y0 + dy : y0 + dy - 1;
        value += weight * double(s[size_t(yy) * sw + xx]);
      }
    dst[b * dst_stride + size_t(y) * dw + x] = float(value);
  }

  void launch_enlarge(const float* src, size_t src_stride, int sw, int sh,
                      float* dst, size_t dst_stride, int dw, int dh, int batch,
                      hipStream_t stream)
  {
    const dim3 block(64, 4);
    const dim3 grid((dw + 63) / 64, (dh + 3) / 4, batch);
    hipLaunchKernelGGL(enlarge_kernel, grid, block, 0, stream, src, src_stride,
                       sw, sh, dst, dst_stride, dw, dh);
  }

  __global__ void copy_planes_kernel(const float* __restrict__ src,
                                     size_t src_stride, float* __restrict__ dst,
                                     size_t dst_stride, size_t count)
  {
    const size_t b = blockIdx.y;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < count;
         i += size_t(gridDim.x) * blockDim.x)
      dst[b * dst_stride + i] = src[b * src_stride + i];
  }

  void launch_copy_planes(const float* src, size_t src_stride, float* dst,
                          size_t dst_stride, size_t count, int batch,
                          hipStream_t stream)
  {
    const int blocks = int(std::min<size_t>((count + 255) / 256, 2048));
    hipLaunchKernelGGL(copy_planes_kernel, dim3(blocks, batch), dim3(256), 0,
                       stream, src, src_stride, dst, dst_stride, count);
  }

  __global__ void subtract_kernel(const float* __restrict__ a,
                                  const float* __restrict__ b,
                                  float* __restrict__ out, size_t count)
  {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < count;
         i += size_t(gridDim.x) * blockDim.x)
      out[i] = a[i] - b[i];
  }

  //! Clears the per-step counters and STAMPS the step: the thread that owns
  //! the int4 holding counters[stamp] bumps the context's persistent step
  //! counter *epoch (outside the cleared block; this thread is its only
  //! writer) and stores the new value there.  The host counts the steps it
  //! enqueued; a read-back whose stamp is not that number did not go through
  //! this kernel - its counters are the previous step's (sift_context.cpp:
  //! counters_corrupt).
  __global__ void zero_counters_kernel(int4* p, unsigned* epoch, int stamp)
  {
    const int i = blockIdx.x * 16 + threadIdx.x;
    int4 v = make_int4(0, 0, 0, 0);
    if (i == (stamp >> 2))
    {
      const unsigned e = *epoch + 1u;
      *epoch = e;
      const int lane = stamp & 3;
      v.x = lane == 0 ? int(e) : 0;
      v.y = lane == 1 ? int(e) : 0;
      v.z = lane == 2 ? int(e) : 0;
      v.w = lane == 3 ? int(e) : 0;
    }
    p[i] = v;
  }

  void launch_zero_counters(int* counters, size_t count, unsigned* epoch, int stamp,
                            hipStream_t stream)
  {
    // whole 256-byte blocks (counters_padded): 16 lanes x 16 bytes each
    hipLaunchKernelGGL(zero_counters_kernel, dim3(unsigned(count / 64)), dim3(16), 0,
                       stream, reinterpret_cast<int4*>(counters), epoch, stamp);
  }

  void launch_subtract(const float* a, const float* b, float* out, size_t count,
                       hipStream_t stream)
  {
    const int blocks = int(std::min<size_t>((count + 255) / 256, 2048));
    hipLaunchKernelGGL(subtract_kernel, dim3(blocks), dim3(256), 0, stream, a, b,
                       out, count);
  }


  // ======================================================================== //
  // 8-bit input frames -> gray32f on the device.
  // Reference: from_rgb8_to_gray32f, ImageProcessing/FastColorConversion.cpp:
  // 42-66 (fallback branch = DO::Sara::convert): channels to double by /255.0,
  // 0.2125 R + 0.7154 G + 0.0721 B in double, cast to float
  // (Core/Pixel/SmartColorConversion.hpp:237-246, ColorConversion.hpp:26-34);
  // gray8: float(v) / 255.f (Core/Pixel/ChannelConversion.hpp:40-54).
  // One thread converts 4 pixels: 3 aligned dwords in, one float4 out -
  // a quarter of the PCIe bytes of a float frame and no host-side pass.
  // ======================================================================== //
  __device__ inline float rgb_to_gray(unsigned r, unsigned g, unsigned b)
  {
    const double rd = double(r) / 255.0, gd = double(g) / 255.0,
                 bd = double(b) / 255.0;
    return float(0.2125 * rd + 0.7154 * gd + 0.0721 * bd);
  }

  __global__ void rgb8_to_gray32f_kernel(const unsigned char* __restrict__ src,
                                         size_t src_stride,
                                         float* __restrict__ dst,
                                         size_t dst_stride, size_t count)
  {
    const size_t b = blockIdx.y;
    const unsigned char* s = src + b * src_stride;
    float* d = dst + b * dst_stride;
    const bool aligned = ((reinterpret_cast<uintptr_t>(s) & 3) == 0) &&
                         ((reinterpret_cast<uintptr_t>(d) & 15) == 0);
    const size_t quads = count / 4;
    for (size_t q = size_t(blockIdx.x) * blockDim.x + threadIdx.x; q < quads;
         q += size_t(gridDim.x) * blockDim.x)
    {
      unsigned w0, w1, w2;
      if (aligned)
      {
        const unsigned* p = reinterpret_cast<const unsigned*>(s + 12 * q);
        w0 = p[0];
        w1 = p[1];
        w2 = p[2];
      }
      else
      {
        const unsigned char* p = s + 12 * q;
        w0 = p[0] | (p[1] << 8) | (p[2] << 16) | (unsigned(p[3]) << 24);
        w1 = p[4] | (p[5] << 8) | (p[6] << 16) | (unsigned(p[7]) << 24);
        w2 = p[8] | (p[9] << 8) | (p[10] << 16) | (unsigned(p[11]) << 24);
      }
      const float g0 = rgb_to_gray(w0 & 255, (w0 >> 8) & 255, (w0 >> 16) & 255);
      const float g1 = rgb_to_gray(w0 >> 24, w1 & 255, (w1 >> 8) & 255);
      const float g2 = rgb_to_gray((w1 >> 16) & 255, w1 >> 24, w2 & 255);
      const float g3 = rgb_to_gray((w2 >> 8) & 255, (w2 >> 16) & 255, w2 >> 24);
      if (aligned)
        *reinterpret_cast<float4*>(d + 4 * q) = make_float4(g0, g1, g2, g3);
      else
      {
        d[4 * q] = g0;
        d[4 * q + 1] = g1;
        d[4 * q + 2] = g2;
        d[4 * q + 3] = g3;
      }
    }
    // tail (count not a multiple of 4)
    if (blockIdx.x == 0 && threadIdx.x < (count & 3))
    {
      const size_t i = quads * 4 + threadIdx.x;
      d[i] = rgb_to_gray(s[3 * i], s[3 * i + 1], s[3 * i + 2]);
    }
  }

  __global__ void gray8_to_gray32f_kernel(const unsigned char* __restrict__ src,
                                          size_t src_stride,
                                          float* __restrict__ dst,
                                          size_t dst_stride, size_t count)
  {
    const size_t b = blockIdx.y;
    const unsigned char* s = src + b * src_stride;
    float* d = dst + b * dst_stride;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < count;
         i += size_t(gridDim.x) * blockDim.x)
      d[i] = float(s[i]) / 255.f;
  }

  void launch_u8_to_gray32f(const unsigned char* src, size_t src_stride,
                            int channels, float* dst, size_t dst_stride,
                            size_t count, int batch, hipStream_t stream)
  {
    if (channels == 3)
    {
      const int blocks = int(std::min<size_t>((count / 4 + 255) / 256 + 1, 4096));
      hipLaunchKernelGGL(rgb8_to_gray32f_kernel, dim3(blocks, batch), dim3(256),
                         0, stream, src, src_stride, dst, dst_stride, count);
    }
    else
    {
      const int blocks = int(std::min<size_t>((count + 255) / 256, 4096));
      hipLaunchKernelGGL(gray8_to_gray32f_kernel, dim3(blocks, batch), dim3(256),
                         0, stream, src, src_stride, dst, dst_stride, count);
    }
  }

}  // namespace sara_hip
