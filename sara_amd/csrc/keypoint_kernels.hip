// gfx950 (MI355X / CDNA4) per-keypoint kernels of the SIFT front-end: dominant
// orientations, list expansion, 128-D descriptors, RootSIFT, extremum read-out.
//
// A translation unit of its own since round 5: the streaming kernels of
// feature_kernels.hip run 4 % faster WITHOUT the SLP vectoriser (its v_pk_mul /
// v_pk_add_f32 pairs are slower than two scalar instructions on gfx950:
// extrema + gradient stage 2.22 -> 2.11 ms per 64 x 1080p step), the kernels here
// 1-2 % slower (Makefile).
//
// Built with -ffp-contract=off, like every kernel of the library.
#include "feature_shared.hpp"

#include "device_math.hpp"

#include <cmath>
#include <cstdlib>
#include <string>
#include <type_traits>

namespace sara_hip {

  // ======================================================================== //
  // Dominant orientations.  Reference: ComputeDominantOrientations,
  // FeatureDescriptors/Orientation.cpp:82-166; compute_orientation_histogram,
  // lowe_smooth_histogram, find_peaks, refine_peak,
  // FeatureDescriptors/Orientation.hpp:91-212.
  //
  // One wave per extremum.  The CPU path adds the window's pixels to the
  // 36-bin histogram one by one in raster order, rounding to float after every
  // addition (the weight is a double).  To reproduce that exactly, lanes
  // evaluate 64 pixels at a time (bin, double contribution); a 64-bit mask per
  // bin records which lanes hit it (ds_or_b64), and lanes 0..35 each own one
  // bin and replay only their own contributions, in ascending lane = raster
  // order.  Smoothing / peak search are exact lane-parallel restatements.
  // ======================================================================== //
  __device__ inline int key_octave(unsigned long long key)
  {
    return int(key >> 41) / kMaxScales;
  }
  __device__ inline int key_scale(unsigned long long key)
  {
    return int(key >> 41) % kMaxScales;
  }
  __device__ inline int key_y(unsigned long long key)
  {
    return int((key >> 21) & 0xfffff);
  }
  __device__ inline int key_x(unsigned long long key)
  {
    return int((key >> 1) & 0xfffff);
  }

  //! Workgroup -> work item remap.  A frame's (octave, scale, y, x)-sorted
  //! list is cut into 8 contiguous chunks and each chunk is served by the
  //! workgroups of one XCD (workgroup b runs on XCD b % 8 when the grid's x
  //! extent is a multiple of 8), so that neighbouring keypoints share their
  //! image rows in that XCD's L2.  The chunk -> XCD assignment rotates with the
  //! frame index: the chunks are not equally expensive (the tail of the list
  //! holds the large-scale keypoints), and a fixed assignment leaves one XCD
  //! with all the expensive chunks.  Returns the logical block index or -1.
  __device__ inline int xcd_local_block(int bx, int frame, int nblk, int run)
  {
    // runs of `run` consecutive logical blocks, dealt round-robin to the XCDs
    const int x = (bx + 3 * frame) & 7;
    const int j = bx >> 3;
    const int q = j / run, i = j - q * run;
    const int lb = (q * 8 + x) * run + i;
    return lb < nblk ? lb : -1;
  }

  __device__ inline double readlane_f64(double v, int lane)
  {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane(unsigned(u), lane);
    const unsigned hi = __builtin_amdgcn_readlane(unsigned(u >> 32), lane);
    return __longlong_as_double(((unsigned long long) hi << 32) | lo);
  }

  // Wave reductions through DPP instead of __shfl_xor: a shuffle is a
  // ds_bpermute_b32, i.e. one LDS round trip per step, and these kernels'
  // LDS queues are full of atomics (a dependent chain of 6 - 12 of them per
  // reduction is what the per-keypoint phases were waiting for).
#define SARA_DPP_F(old, v, ctrl, rmask)                                        \
  __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old),              \
                                             __float_as_int(v), ctrl, rmask,   \
                                             0xf, false))
  //! Sum over the 64 lanes, returned in every lane.
  __device__ __forceinline__ float wave_sum_dpp(float v)
  {
    v += SARA_DPP_F(0.f, v, 0x111, 0xf);  // row_shr:1
    v += SARA_DPP_F(0.f, v, 0x112, 0xf);  // row_shr:2
    v += SARA_DPP_F(0.f, v, 0x114, 0xf);  // row_shr:4
    v += SARA_DPP_F(0.f, v, 0x118, 0xf);  // row_shr:8
    v += SARA_DPP_F(0.f, v, 0x142, 0xa);  // row_bcast:15
    v += SARA_DPP_F(0.f, v, 0x143, 0xc);  // row_bcast:31
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
  }
  //! Maximum over the 64 lanes, returned in every lane.
  __device__ __forceinline__ float wave_max_dpp(float v)
  {
    v = fmaxf(v, SARA_DPP_F(v, v, 0x111, 0xf));
    v = fmaxf(v, SARA_DPP_F(v, v, 0x112, 0xf));
    v = fmaxf(v, SARA_DPP_F(v, v, 0x114, 0xf));
    v = fmaxf(v, SARA_DPP_F(v, v, 0x118, 0xf));
    v = fmaxf(v, SARA_DPP_F(v, v, 0x142, 0xa));
    v = fmaxf(v, SARA_DPP_F(v, v, 0x143, 0xc));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
  }
  __device__ __forceinline__ int wave_max_dpp(int v)
  {
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
  }
  //! Lane i receives lane i - 1 (lane 0 keeps its own value) / lane i + 1.
  __device__ __forceinline__ float wave_shr1(float v)
  {
    return SARA_DPP_F(v, v, 0x138, 0xf);  // wave_shr:1
  }
  __device__ __forceinline__ float wave_shl1(float v)
  {
    return SARA_DPP_F(v, v, 0x130, 0xf);  // wave_shl:1
  }

  //! Waves (= keypoints in flight) per workgroup of the per-keypoint kernels.
  //! A workgroup's LDS and wave slots are only released when its slowest wave
  //! retires, and the work per keypoint varies (1 to 4 orientations, patch
  //! area 1 : 2.5 between scales): with 4 waves per group the others idle.
  //! Measured (64 x 1080p): descriptor kernel 2.67 / 2.45 / 2.36 ms with
  //! 4 / 2 / 1 waves per group; the orientation kernel shares its weight
  //! tables in LDS across the group and prefers 4 (round 4, same box: 0.75 /
  //! 0.81 / 0.89 ms with 4 / 2 / 1, 0.81 with 8).

  //! Inclusive prefix sum over the 64 lanes (DPP row shifts, then the row
  //! broadcasts of gfx9: 6 steps, no LDS).
  __device__ __forceinline__ int wave_inclusive_scan(int v)
  {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31
    return v;
  }

#if defined(SARA_DESC_PROF) || defined(SARA_ORI_PROF)
  // Per-phase wave cycles (s_memtime), summed over all waves: a development
  // aid, read back through sara_hip_debug_desc_prof().
  __device__ unsigned long long g_desc_prof[8];
#define SARA_PROF_T(var) const long long var = clock64()
#define SARA_PROF_ADD(slot, a, b)                                              \
  if (lane == 0)                                                               \
  atomicAdd(&g_desc_prof[slot], (unsigned long long) ((b) - (a)))
#else
#define SARA_PROF_T(var)
#define SARA_PROF_ADD(slot, a, b)
#endif
#ifdef SARA_ORI_PROF
#define SARA_OPROF_T(var) const long long var = clock64()
#define SARA_OPROF_ADD(slot, a, b)                                             \
  if (lane == 0)                                                               \
  atomicAdd(&g_desc_prof[slot], (unsigned long long) ((b) - (a)))
#else
#define SARA_OPROF_T(var)
#define SARA_OPROF_ADD(slot, a, b)
#endif
  //! WLDS: the Gaussian weight tables of all scales sit in (dynamic) LDS.  From
  //! global memory each look-up is a vector load that shares the in-order
  //! vmcnt counter with the gathers: waiting for a weight then also waits for
  //! every gather issued ahead of it, which defeats the prefetch ring below.
  constexpr int kOriWaves = 4;       // waves per workgroup (2: 0.73 -> 0.74 ms)
  constexpr int kOriBlocksPerEu = 1;
  constexpr int kOriGroup = 2;  // chunks sorted and replayed together (4 / 6 / 8: 1.02 / 1.02 / 1.22 ms vs 0.92)
  constexpr int kOriAheadRing = 2;  // gathers in flight per lane (4 / 6 / 8 slower)
  static_assert(kOriGroup % kOriAheadRing == 0, "ring slots are static");
  template <bool WLDS>
  __global__ __launch_bounds__(64 * kOriWaves, kOriBlocksPerEu) void orientation_kernel(
      const GradPyramidView* __restrict__ gradp,
      const ScaleTable* __restrict__ tabp, const double* __restrict__ weights,
      int n_weights, CandidateLists cand, OrientationLists ori, int xcd_run)
  {
    __shared__ unsigned long long s_mask[kOriWaves][kOriGroup * kOriBins];
    // a bin's segment is followed by one slot that holds 0. (see the replay)
    __shared__ double s_contrib[kOriWaves][64 * kOriGroup + 64];
    __shared__ int s_segoff[kOriWaves][kOriGroup * kOriBins];
    extern __shared__ __attribute__((aligned(16))) double s_weights[];
    const GradPyramidView& grad = *gradp;
    const ScaleTable& tab = *tabp;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    __shared__ float s_thr[40];
    if (threadIdx.x < 40)
      s_thr[threadIdx.x] = tab.ori_bin_thr[threadIdx.x];
    // What an item needs of its octave and scale, in LDS: from global memory
    // these are a second round trip (key -> table entries -> first gather) at
    // the head of every item.  The patch geometry comes with it: D = 2R + 1,
    // 64 / D and 64 % D (a lane's pixel advances by 64 per chunk) and
    // ceil(2^16 / D), with which lane / D is one multiplication and a shift
    // (exact for lane < 64: the error of lane * magic / 2^16 stays below
    // 64 / 2^16 < 1 / D).
    struct OctaveEntry
    {
      unsigned long long base, frame_stride, plane;
      int w, h;
    };
    struct ScaleEntry
    {
      int radius, woff, dv64, du64, magic, pad[3];
    };
    __shared__ __attribute__((aligned(16))) OctaveEntry s_oct[16];
    __shared__ __attribute__((aligned(16))) ScaleEntry s_scale[kMaxScales];
    if (threadIdx.x < 16)
    {
      const int o_ = threadIdx.x;
      s_oct[o_].base = reinterpret_cast<unsigned long long>(grad.base[o_]);
      s_oct[o_].frame_stride = grad.frame_stride[o_];
      s_oct[o_].plane = grad.plane[o_];
      s_oct[o_].w = grad.w[o_];
      s_oct[o_].h = grad.h[o_];
    }
    if (threadIdx.x >= 32 && threadIdx.x < 32 + kMaxScales)
    {
      const int s_ = threadIdx.x - 32;
      const int R_ = tab.ori_radius[s_];
      const int D_ = max(2 * R_ + 1, 1);
      s_scale[s_].radius = R_;
      s_scale[s_].woff = tab.ori_woff[s_];
      s_scale[s_].dv64 = 64 / D_;
      s_scale[s_].du64 = 64 % D_;
      s_scale[s_].magic = (65536 + D_ - 1) / D_;
    }
    if (WLDS)
    {
      for (int i = threadIdx.x; i < n_weights; i += 64 * kOriWaves)
        s_weights[i] = weights[i];
    }
    __syncthreads();
    const int b = blockIdx.y;
    const int n = min(cand.count[b], cand.cap);
    // Persistent blocks: the grid holds one run-group of blocks per frame and
    // every block walks the frame's work items with that stride (a grid sized
    // for the list capacity launches ~3 empty waves for every useful one).
    const int nblk = (n + kOriWaves - 1) / kOriWaves;
    const int unit = 8 * xcd_run;
    const int positions = unit * ((nblk + unit - 1) / unit);
    // (key, data) of a work item: loaded one item ahead (a wave walks several
    // items, and the pair sits at the head of the item's dependent chain:
    // key / data -> gather addresses -> first gather)
    const size_t row = size_t(b) * cand.cap;
    auto fetch = [&](int lb, unsigned long long& key_, float4& d_) {
      const int idx = lb * kOriWaves + wave;
      key_ = 0ull;
      d_ = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lb >= 0 && idx < n)
      {
        key_ = cand.skey[row + idx];
        d_ = cand.sdata[row + idx];
      }
    };
    auto item = [&](int lb, unsigned long long key, float4 d) {
    const int idx = lb * kOriWaves + wave;
    if (idx >= n)
      return;

    SARA_OPROF_T(t_item);
    const int o = key_octave(key);
    const int s = key_scale(key);

    const int rx = int(roundf(d.x));
    const int ry = int(roundf(d.y));
    const OctaveEntry oe = s_oct[o];
    const ScaleEntry se = s_scale[s];
    const int R = se.radius;
    const int woff = se.woff;
    const int w = oe.w, h = oe.h;
    // explicitly a global-memory pointer: through a generic pointer these
    // gathers become flat_load, which counts on lgkmcnt as well, so waiting
    // for a sample would also wait for every LDS atomic still in flight
    const global_float2_ptr g =
        (global_float2_ptr) reinterpret_cast<const f32x2*>(
            reinterpret_cast<const float*>(oe.base) + size_t(b) * oe.frame_stride) +
        size_t(s) * oe.plane;
    // a pixel that is always inside the image (idle lanes gather it)
    const size_t center = size_t(min(max(ry, 0), h - 1)) * w +
                          size_t(min(max(rx, 0), w - 1));

    const int D = 2 * R + 1;
    const int npx = D * D;
    float hist = 0.f;

    const int dv64 = se.dv64, du64 = se.du64;
    // lane = lrow * D + lcol
    const int lrow = (lane * se.magic) >> 16, lcol = lane - lrow * D;
    unsigned long long* bin_mask = s_mask[wave];
    double* contrib = s_contrib[wave];
    int* seg_off = s_segoff[wave];

    // The patch is fetched from HBM once and nothing else hides that latency:
    // a ring of kOriAhead chunks (64 pixels each) of unconditional gathers runs
    // ahead of the histogram work.  (u, v) of this lane's pixel advances by 64
    // pixels per chunk, once on the issue side and once on the consumer side.
    constexpr int kOriAhead = kOriAheadRing;
    auto advance = [&](int& u_, int& v_) {
      u_ += du64;
      v_ += dv64;
      if (u_ > R)
      {
        u_ -= D;
        v_ += 1;
      }
    };
    int iu = lcol - R, iv = lrow - R;  // issue side
    auto issue = [&](int base_, float2& mo_, bool& ok_) {
      const int xx = rx + iu, yy = ry + iv;
      ok_ = (base_ + lane < npx) && xx >= 0 && xx < w && yy >= 0 && yy < h;
      mo_ = load_pair32(g, ok_ ? __umul24(unsigned(yy), unsigned(w)) + unsigned(xx)
                                : unsigned(center));
      advance(iu, iv);
    };
    float2 ring[kOriAhead];
    bool ring_ok[kOriAhead];
#pragma unroll
    for (int q = 0; q < kOriAhead; ++q)
      issue(64 * q, ring[q], ring_ok[q]);
    __builtin_amdgcn_s_waitcnt(0x0f70 | (kOriAhead - 1));  // see descriptor_kernel
    int u = lcol - R, v = lrow - R;  // consumer side

    // Chunks are processed in groups of kOriGroup: the samples of a group are
    // sorted by (bin, chunk, lane) and each bin's segment is replayed once per
    // group.  The replay is a lock-step loop of as many steps as the fullest
    // bin has samples; per 64-pixel chunk that is 104 steps for an average
    // extremum of the benchmark frames, per group of four 75 (whole patch: 53)
    // - and the scan over the bins runs once per group.
    for (int base0 = 0; base0 < npx; base0 += 64 * kOriGroup)
    {
      SARA_OPROF_T(t_c0);
      int binq[kOriGroup];
      double cq[kOriGroup];
      if (lane < kOriBins)
      {
#pragma unroll
        for (int q = 0; q < kOriGroup; ++q)
          bin_mask[q * kOriBins + lane] = 0ull;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < kOriGroup; ++q)
      {
        // Chunks beyond the patch are idle (ok is false in every lane): their
        // histogram work is skipped under a wave-uniform branch, but the slot
        // is still refilled - the compiler counts the gathers in flight only
        // along straight control flow, so the refill stays unconditional and
        // there is no break.
        const int base = base0 + 64 * q;
        binq[q] = -1;
        cq[q] = 0.;
        if (base < npx)
        {
          const float2 mo = ring[q % kOriAhead];
          const bool ok = ring_ok[q % kOriAhead];
          const int uc = u, vc = v;
          advance(u, v);
          if (ok)
          {
            float a = mo.y;
            a = a < 0 ? a + float(2. * M_PI) : a;
            // int(floor(double(a / float(2 pi) * 36))) % 36 without the
            // division: estimate, then one step of correction against the
            // exact thresholds
            int kb = int(a * float(kOriBins / (2. * M_PI)));
            kb = min(max(kb, 0), kOriBins);
            const float t0 = s_thr[kb], t1 = s_thr[kb + 1];
            kb += int(a >= t1) - int(a < t0);
            binq[q] = kb == kOriBins ? 0 : kb;
            const int wi = woff + uc * uc + vc * vc;
            cq[q] = (WLDS ? s_weights[wi] : weights[wi]) * double(mo.x);
            // which lanes of this chunk fall into which bin: one 64-bit mask
            // per (chunk, bin), built with integer LDS atomics
            atomicOr(&bin_mask[q * kOriBins + binq[q]], 1ull << lane);
          }
        }
        // refill the slot (its pair is consumed): chunk base + 64 * kOriAhead
        issue(base + 64 * kOriAhead, ring[q % kOriAhead], ring_ok[q % kOriAhead]);
      }
      __builtin_amdgcn_wave_barrier();
      SARA_OPROF_T(t_c1);
      SARA_OPROF_ADD(1, t_c0, t_c1);
      // The contributions are sorted by (bin, chunk, lane): a sample's slot is
      // the number of samples in smaller bins (exclusive scan of the masks'
      // population counts over the 36 owner lanes), plus those of its bin in
      // earlier chunks of the group, plus its rank inside its chunk's mask
      // (mbcnt); the owner lane of a bin then adds its contiguous segment in
      // that order - chunk by chunk, ascending lane = raster order, the
      // rounding sequence of the CPU loop - with a plain counted loop.
      int cnt = 0;
      int before[kOriGroup];
#pragma unroll
      for (int q = 0; q < kOriGroup; ++q)
      {
        const unsigned long long own =
            lane < kOriBins ? bin_mask[q * kOriBins + lane] : 0ull;
        before[q] = cnt;
        cnt += __popcll(own);
      }
      const int incl = wave_inclusive_scan(cnt);
      // segment of lane l: slots [seg_begin, seg_end), then a slot that stays
      // 0. - lane l's segment is shifted by l slots to make room for them
      const int seg_begin = incl - cnt + lane;
      const int seg_end = seg_begin + cnt;
      contrib[seg_end] = 0.;
      if (lane < kOriBins)
      {
#pragma unroll
        for (int q = 0; q < kOriGroup; ++q)
          seg_off[q * kOriBins + lane] = seg_begin + before[q];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < kOriGroup; ++q)
        if (binq[q] >= 0)
        {
          const unsigned long long m = bin_mask[q * kOriBins + binq[q]];
          const int r = __builtin_amdgcn_mbcnt_hi(
              unsigned(m >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(m), 0u));
          contrib[seg_off[q * kOriBins + binq[q]] + r] = cq[q];
        }
      __builtin_amdgcn_wave_barrier();
      SARA_OPROF_T(t_c2);
      SARA_OPROF_ADD(2, t_c1, t_c2);
      {
        // Lock step over the fullest bin's count, without predication: a lane
        // whose segment is exhausted keeps reading the 0. behind it, and
        // float(double(hist) + 0.) is hist (hist >= +0: the contributions are
        // weight * magnitude).  Per step: index, LDS read of the next
        // contribution, and the three dependent conversions / addition.
        const int steps = wave_max_dpp(cnt);
        const double* p = contrib + seg_begin;
        const double* const p_end = contrib + seg_end;
        double cur = *p;
#pragma unroll 2
        for (int st = 0; st < steps; ++st)
        {
          p = p + 1 < p_end ? p + 1 : p_end;
          const double nxt = *p;
          hist = float(double(hist) + cur);
          cur = nxt;
        }
      }
      __builtin_amdgcn_wave_barrier();
      SARA_OPROF_T(t_c3);
      SARA_OPROF_ADD(3, t_c2, t_c3);
    }
    SARA_OPROF_T(t_loop);

    // lowe_smooth_histogram: 6 circular box-blur iterations.
    // circular neighbours of the 36 bin lanes: DPP wave shifts, the two wrap
    // positions through readlane
    auto ring_prev = [&](float x) {
      const float wrap = __int_as_float(
          __builtin_amdgcn_readlane(__float_as_int(x), kOriBins - 1));
      const float sh = wave_shr1(x);
      return lane == 0 ? wrap : sh;
    };
    auto ring_next = [&](float x) {
      const float wrap =
          __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0));
      const float sh = wave_shl1(x);
      return lane == kOriBins - 1 ? wrap : sh;
    };
    for (int iter = 0; iter < 6; ++iter)
    {
      const float prev = ring_prev(hist);
      const float next = ring_next(hist);
      hist = (prev + hist + next) / 3.f;
    }

    // find_peaks + refine_peak.
    const float mx = wave_max_dpp(lane < kOriBins ? hist : -INFINITY);
    const float y0 = ring_prev(hist);
    const float y2 = ring_next(hist);
    const bool is_peak =
        lane < kOriBins && hist >= 0.8f * mx && hist > y0 && hist > y2;
    const unsigned long long mask = __ballot(is_peak);
    if (is_peak)
    {
      const float fprime = (y2 - y0) / 2.f;
      const float fsecond = y0 - 2.f * hist + y2;
      const float hh = -fprime / fsecond;
      float theta = float(lane) + 0.5f + hh;
      theta *= float(2 * M_PI) / kOriBins;
      if (theta > float(M_PI))
        theta -= 2.f * float(M_PI);
      const int r = __popcll(mask & ((1ull << lane) - 1ull));
      ori.peak_theta[(row + idx) * kMaxPeaks + r] = theta;
      if (r < 8)
        ori.record[row + idx].theta[r] = theta;
    }
    if (lane == 0)
    {
      ori.peak_count[row + idx] = __popcll(mask);
      KeypointRecord& rec = ori.record[row + idx];
      rec.d = d;
      rec.key = key;
      rec.npeaks = __popcll(mask);
      rec.reserved = 0;
    }
    SARA_OPROF_T(t_end);
    SARA_OPROF_ADD(4, t_loop, t_end);
    SARA_OPROF_ADD(7, t_item, t_end);
    };
    int bx = blockIdx.x;
    int lb = bx < positions ? xcd_local_block(bx, b, nblk, xcd_run) : -1;
    unsigned long long key = 0ull;
    float4 d;
    fetch(lb, key, d);
    while (bx < positions)
    {
      const int bx_next = bx + gridDim.x;
      const int lb_next =
          bx_next < positions ? xcd_local_block(bx_next, b, nblk, xcd_run) : -1;
      unsigned long long key_next;
      float4 d_next;
      fetch(lb_next, key_next, d_next);
      if (lb >= 0)
        item(lb, key, d);
      bx = bx_next;
      lb = lb_next;
      key = key_next;
      d = d_next;
    }
  }

  void launch_orientations(const GradPyramidView* grad, const ScaleTable* tab,
                           const double* ori_weights, int n_weights,
                           const CandidateLists& cand,
                           const OrientationLists& ori, int batch,
                           hipStream_t stream)
  {
    // xcd_local_block() spreads ceil(n/4) work items over 8 chunks, so the
    // grid has to be a multiple of 8 blocks
    constexpr int ori_run = g_xcd_run;
    const int unit = 8 * ori_run;
    const int needed =
        unit * (((cand.cap + kOriWaves - 1) / kOriWaves + unit - 1) / unit);
    const int units = persist_units(batch, kOriWaves);
    const dim3 grid(std::min(needed, unit * units), batch);
    // weight tables in LDS when they leave room for 8 blocks per CU
    const size_t wbytes = sizeof(double) * size_t(n_weights);
    if (n_weights > 0 && wbytes <= 14 * 1024)
      hipLaunchKernelGGL(orientation_kernel<true>, grid, dim3(64 * kOriWaves), wbytes,
                         stream, grad, tab, ori_weights, n_weights, cand, ori,
                         ori_run);
    else
      hipLaunchKernelGGL(orientation_kernel<false>, grid, dim3(64 * kOriWaves), 0, stream,
                         grad, tab, ori_weights, n_weights, cand, ori, ori_run);
  }

  // ------------------------------------------------------------------------ //
  // Output offsets: exclusive scan of the per-extremum peak counts in sorted
  // order (Orientation.cpp:146-161 expands the list in input order).
  // ------------------------------------------------------------------------ //
  //! One 1024-thread workgroup per frame: thread t owns a run of consecutive
  //! extrema (their counts summed locally, one block scan of the 1024
  //! partials), writes the offsets and expands the keypoint list
  //! (Orientation.cpp:146-161: one entry per dominant orientation, in input
  //! order).  The workgroup that finishes last (device counter `done`, zeroed
  //! with the other per-step counters) also writes the frame offsets, which
  //! saves the separate single-thread launch.
  __global__ __launch_bounds__(1024) void scan_peaks_kernel(CandidateLists cand,
                                                            OrientationLists ori,
                                                            int* done, int batch)
  {
    __shared__ int s_part[1024];
    __shared__ int s_last;
    __shared__ int s_before;
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int n = min(cand.count[b], cand.cap);
    const size_t row = size_t(b) * cand.cap;
    // Small batches (round 3): a frame is cut into gridDim.x consecutive parts,
    // one workgroup each - one frame's 4 300 extrema took 20 us on a single
    // workgroup (four dependent rounds of loads and stores per thread).  A
    // part first sums the counts in front of it (every thread a strided share,
    // one block reduction).
    const int parts = gridDim.x, part = blockIdx.x;
    const int chunk = (n + parts - 1) / parts;
    const int plo = min(part * chunk, n), phi = min(plo + chunk, n);
    int before = 0;
    if (parts > 1)
    {
      int acc = 0;
      for (int i = tid; i < plo; i += 1024)
        acc += ori.peak_count[row + i];
      acc = wave_inclusive_scan(acc);
      if ((tid & 63) == 63)
        s_part[tid >> 6] = acc;
      __syncthreads();
      if (tid == 0)
      {
        int t = 0;
        for (int k = 0; k < 16; ++k)
          t += s_part[k];
        s_before = t;
      }
      __syncthreads();
      before = s_before;
      __syncthreads();
    }
    const int per = (phi - plo + 1023) / 1024;
    const int lo = min(plo + tid * per, phi), hi = min(lo + per, phi);
    int sum = 0;
    for (int i = lo; i < hi; ++i)
      sum += ori.peak_count[row + i];
    // block scan: DPP scan inside each of the 16 waves, the 16 wave totals
    // through LDS (two barriers instead of twenty)
    const int lane = tid & 63, wave = tid >> 6;
    const int incl_w = wave_inclusive_scan(sum);
    if (lane == 63)
      s_part[wave] = incl_w;
    __syncthreads();
    int wave_base = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k)
    {
      const int t = s_part[k];
      wave_base += k < wave ? t : 0;
      total += t;
    }
    __syncthreads();
    if (tid == 1023)
      s_part[1023] = total;  // read below as the frame's keypoint count
    int at = before + wave_base + incl_w - sum;
    auto expand = [&](int i, const KeypointRecord& rec) {
      const int v = rec.npeaks;
      ori.offset[row + i] = at;
      for (int k = 0; k < v && at + k < cand.cap; ++k)
      {
        KeypointItem it;
        it.d = rec.d;
        it.key = rec.key;
        it.theta = k < 8 ? rec.theta[k] : ori.peak_theta[(row + i) * kMaxPeaks + k];
        it.reserved = 0;
        ori.item[row + at + k] = it;
      }
      at += v;
    };
    for (int i = lo; i < hi; ++i)
      expand(i, ori.record[row + i]);
    if (tid == 1023)
    {
      s_last = 0;
      if (part == parts - 1)  // the frame's last part knows the frame's total
      {
        ori.kp_count[b] = before + s_part[1023];
        __threadfence();  // the count is visible before the arrival is
        s_last = atomicAdd(done, 1) == batch - 1;
      }
    }
    __syncthreads();
    if (s_last && tid == 0)
    {
      __threadfence();
      int acc = 0;
      for (int f = 0; f < batch; ++f)
      {
        ori.frame_offset[f] = acc;
        acc += min(__hip_atomic_load(&ori.kp_count[f], __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT),
                   cand.cap);
      }
      ori.frame_offset[batch] = acc;
    }
  }

  __global__ void frame_offsets_kernel(const int* __restrict__ counts, int cap,
                                       int* __restrict__ offsets, int batch)
  {
    if (threadIdx.x == 0 && blockIdx.x == 0)
    {
      int acc = 0;
      for (int b = 0; b < batch; ++b)
      {
        offsets[b] = acc;
        acc += min(counts[b], cap);
      }
      offsets[batch] = acc;
    }
  }

  void launch_scan_peaks(const CandidateLists& cand, const OrientationLists& ori,
                         int* done_counter, int batch, hipStream_t stream)
  {
    // parts per frame: ~1 extremum per thread at the list sizes of a video
    // frame when the batch alone cannot fill the chip
    // (64 frames: 41 / 31 / 29 us with 1 / 2 / 4 parts - one workgroup per
    // frame leaves three quarters of the CUs idle)
    const int parts = (batch <= 2 || batch > 8) ? 4 : 2;
    hipLaunchKernelGGL(scan_peaks_kernel, dim3(parts, batch), dim3(1024), 0, stream,
                       cand, ori, done_counter, batch);
  }

  void launch_extrema_offsets(const CandidateLists& cand, int* ex_offset,
                              int batch, hipStream_t stream)
  {
    hipLaunchKernelGGL(frame_offsets_kernel, dim3(1), dim3(64), 0, stream,
                       cand.count, cand.cap, ex_offset, batch);
  }

  // ======================================================================== //
  // SIFT descriptors (N = 4, O = 8).  Reference: ComputeSIFTDescriptor,
  // FeatureDescriptors/SIFT.hpp:62-145 (patch loop), :204-238 (trilinear
  // accumulate with std::modf truncation), :241-252 (normalisation);
  // OERegion::scale(), Features/Feature.cpp:28-39; final rescale,
  // FeatureDetectors/SIFT.cpp:92-98.
  //
  // One wave per extremum, looping over its orientations; lanes stride the
  // patch pixels and accumulate into a 128-bin LDS histogram per wave, held
  // as 64-bit fixed point (ds_add_u64, see below).  Float tolerance vs the CPU
  // path: summation order and expf/cos/sin last-ulp differences only.
  // ======================================================================== //
  __device__ inline float wave_sum(float v) { return wave_sum_dpp(v); }

  // LDS accumulation of the 128 bins.  ds_add_f32 costs ~192 clk per wave
  // instruction on gfx950 whatever the address pattern (tools/ubench/
  // lds_atomic.hip), ds_add_u32 ~6-16, so contributions are accumulated as a
  // 64-bit two's-complement fixed point (ds_add_u64 costs about one
  // ds_add_u32), scaled per patch (see fx_scale in the kernel): the sum is
  // order-independent, so the result is deterministic.
  //
  // Round 2 structure.  Round 1 walked the patch four rows at a time in
  // lockstep: every pass paid the row set-up (60 instructions) and the exposed
  // latency of its first gather - measured with the sample loop removed, that
  // skeleton alone took 1.15 of the kernel's 2.4 ms.  Now the rows of a
  // patch are cut into chunks of 16 consecutive pixels ONCE per orientation
  // (each lane sets up one row, a wave scan numbers the chunks, the list goes
  // to LDS) and the four 16-lane groups of the wave stream through that list,
  // one chunk each per step: no per-row lockstep, equal work for the groups,
  // and a software pipeline that runs across row boundaries (chunk entry read
  // three steps ahead, gather issued two steps ahead).
  constexpr int kDescCopies = 4;  // histogram replicas per wave (2: +0.6 ms; 8 do not fit)
  // 32-bit fixed-point accumulators on a 5 x 5 cell grid.  The fifth row and
  // column are dump cells: the dx / dy = 1 neighbours of cells 3 always exist,
  // so the eight addresses of a sample are two registers plus immediates and
  // the weights need no selects; the fixed-point scale is chosen per keypoint
  // so that no bin can overflow (see fx_scale).  (Round 2a: 64-bit
  // accumulators on 4 x 4 cells, 2.04 vs 1.60 ms; in git history.)
  constexpr int kDescGrid = 5;  // cells per row of the LDS grid
  // pad 4: a neighbouring cell starts four banks on (0 / 1 / 3 / 4 / 5 / 7 words:
  // 1.396 / 1.412 / 1.386 / 1.376 / 1.389 / 1.400 ms per step)
  constexpr int kDescCellStride = 8 * kDescCopies + 4;
  constexpr int kDescHistWords = kDescGrid * kDescGrid * kDescCellStride;
  using desc_acc_t = int;
  constexpr int kDescRowsPerBlock = 64;   // one row per lane
  constexpr int kDescChunk = 8;        // pixels per chunk (16: 1.60 vs 1.53 ms)
  constexpr int kDescGroups = 64 / kDescChunk;       // chunks per step of a wave
  constexpr int kDescChunksPerPhase = 8;  // chunks of one row per table fill
  constexpr int kDescAhead = 4;  // gathers in flight per lane (2 / 6 / 8: slower)
  // The chunk list is kept as one segment per group, each followed by idle
  // entries as far as the software pipeline looks ahead (3 * kDescAhead - 1
  // steps past the last one): the stream needs no bounds checks
  constexpr int kDescSegIdle = 3 * kDescAhead;
  constexpr int kDescSeg = kDescRowsPerBlock * kDescChunksPerPhase / kDescGroups + kDescSegIdle;
  constexpr int kDescTableCap = kDescGroups * kDescSeg;

  constexpr int kDescWaves = 1;       // waves per workgroup (2 / 4: 2.45 / 2.67 vs 2.36 ms)
  constexpr int kDescWavesPerEu = 6;  // 7 / 8: no change (round 5)
  //! constants of sincos_reduced_f64 (device_math.hpp), read with scalar loads
  __constant__ double g_sincos_coef[kSincosCoefCount] = SARA_SINCOS_COEF_INIT;


  __global__ __launch_bounds__(64 * kDescWaves, kDescWavesPerEu) void descriptor_kernel(
      GradPyramidView grad, CandidateLists cand, OrientationLists ori,
      sara_oeregion* __restrict__ features, int32_t* __restrict__ scale_octave,
      float* __restrict__ descriptors, int with_descriptors, int root_sift,
      int xcd_run)
  {
    __shared__ __attribute__((aligned(16))) desc_acc_t s_acc[kDescWaves][kDescHistWords + 4];
    __shared__ unsigned s_tab[kDescWaves][kDescTableCap];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int grp = lane / kDescChunk, l16 = lane % kDescChunk;  // l16: pixel of the chunk
    const int b = blockIdx.y;
    // work items = keypoints (one dominant orientation each), in output order
    const int n = min(ori.kp_count[b], cand.cap);
    // persistent blocks, see orientation_kernel
    const int nblk = (n + kDescWaves - 1) / kDescWaves;
    const int unit = 8 * xcd_run;
    const int positions = unit * ((nblk + unit - 1) / unit);
    // The item of keypoint idx (scan_peaks_kernel): lanes 0..7 fetch one dword
    // each.  The load of the NEXT work item is issued before the current one
    // is processed.
    const size_t row = size_t(b) * cand.cap;
    const int frame_base = ori.frame_offset[b];
    auto fetch_item = [&](int lb) -> unsigned {
      const int idx = lb * kDescWaves + wave;
      unsigned wv = 0u;
      if (lb >= 0 && idx < n && lane < 8)
        wv = reinterpret_cast<const unsigned*>(ori.item + row + idx)[lane];
      return wv;
    };
    auto item = [&](int lb, unsigned wv) {
    const int idx = lb * kDescWaves + wave;
    if (idx >= n)
      return;

    SARA_PROF_T(t_item);
    auto word = [&](int i) { return unsigned(__builtin_amdgcn_readlane(int(wv), i)); };
    const float4 d = make_float4(__uint_as_float(word(0)), __uint_as_float(word(1)),
                                 __uint_as_float(word(2)), __uint_as_float(word(3)));
    const unsigned long long key =
        (unsigned long long) word(4) | ((unsigned long long) word(5) << 32);
    const float theta = __uint_as_float(word(6));
    const int o = key_octave(key);
    const int s = key_scale(key);
    const int is_max = int(key & 1ull);

    // OERegion(pos, sigma): shape = I * float(pow(double(sigma), -2)).
    const float shape = float(1.0 / (double(d.z) * double(d.z)));
    // OERegion::scale() for an isotropic shape matrix.
    const float scale = 1.f / sqrtf(shape);

    constexpr float pi = float(M_PI);
    const float l = 3.f * scale;
    const double r = sqrt(double(2.f)) * double(l) * 5 / double(2.f);
    const int rr = int(round(r));
    const int rx = int(roundf(d.x));
    const int ry = int(roundf(d.y));
    const int w = grad.w[o], h = grad.h[o];
    // explicitly a global-memory pointer: through a generic pointer these
    // gathers become flat_load, which counts on lgkmcnt as well, so waiting
    // for a sample would also wait for every LDS atomic still in flight
    const global_float2_ptr g =
        (global_float2_ptr) reinterpret_cast<const f32x2*>(
            grad.base[o] + size_t(b) * grad.frame_stride[o]) +
        size_t(s) * grad.plane[o];
    const float factor = grad.factor[o];
    desc_acc_t* hist = s_acc[wave];
    unsigned* tab = s_tab[wave];
    const int copy = lane & (kDescCopies - 1);

    // rows / columns of the patch that fall inside the image
    const int v_lo = max(-rr, -ry), v_hi = min(rr, h - 1 - ry);
    const int u_min = max(-rr, -rx), u_max = min(rr, w - 1 - rx);
    // a pixel that is always inside the image (idle lanes gather it)
    const size_t center = size_t(min(max(ry, 0), h - 1)) * w +
                          size_t(min(max(rx, 0), w - 1));

    // Fixed-point scale of the accumulation.  Every contribution is bounded by
    // |wy*wx*wo*weight*mag| < 2*2*1*1*max(mag); max(mag) over a superset of
    // the patch comes from the coarse 16x16 magnitude maxima written by the
    // gradient kernel.
    float fx_scale = 1.f;
    double fx_inv = 1.;
    if (with_descriptors)
    {
      const unsigned* cm = grad.cmax[o] + size_t(b) * grad.cmax_frame_stride[o] +
                           size_t(s) * grad.ch[o] * grad.cw[o];
      const int cx0 = (rx + u_min) >> 4, cx1 = (rx + u_max) >> 4;
      const int cy0 = (ry + v_lo) >> 4, cy1 = (ry + v_hi) >> 4;
      const int ncx = cx1 - cx0 + 1, ncy = cy1 - cy0 + 1;
      unsigned mxb = 0u;
      if (ncx > 0 && ncy > 0)
        for (int q = lane; q < ncx * ncy; q += 64)
          mxb = max(mxb, cm[size_t(cy0 + q / ncx) * grad.cw[o] + cx0 + q % ncx]);
      // magnitudes are >= 0: their bit patterns order like the floats
      mxb = unsigned(wave_max_dpp(int(mxb)));
      const float mx = __uint_as_float(mxb);
      // 32-bit accumulators: the scale is as large as the worst case allows.
      // A bin collects the samples whose patch coordinates (px, py) lie in a
      // 2 x 2 cell box.  Each of its four cell-sized quadrants (side l
      // pixels) holds at most (l + 2)^2 pixels (area + perimeter / 2 + 1 of
      // a convex region), and |wy wx| <= 4, 2, 2, 1 there (the weights
      // exceed 1 only where modf() hands out a negative fraction, for
      // coordinates in (-1, 0)); wo <= 1, weight <= 1, mag <= mx.  Hence
      // sum |contribution| <= 9 (l + 2)^2 mx scale, kept below 2^31.
      const float bound = 9.f * (l + 2.f) * (l + 2.f);
      if (mx > 0.f && mx < 3.0e38f)
        fx_scale = (2147483648.f * 0.999f) / (bound * mx);
      // a scale outside the normal range (absurd magnitudes) falls back to 1
      if (!(fx_scale > 1e-30f && fx_scale < 1e30f))
        fx_scale = 1.f;
      fx_inv = 1. / double(fx_scale);
    }

    SARA_PROF_T(t_setup);
    SARA_PROF_ADD(0, t_item, t_setup);
    {
      SARA_PROF_T(t_peak);
      const size_t out = size_t(frame_base) + idx;

      if (lane == 0)
      {
        // the 48 bytes of sara_oeregion as three 16-byte stores: every byte
        // (padding included) is written, so the records are reproducible
        // whatever the buffer held before
        const float f2 = factor * factor;
        float4* rec = reinterpret_cast<float4*>(features + out);
        rec[0] = make_float4(d.x * factor, d.y * factor, 0.f, 0.f);
        rec[1] = make_float4(shape / f2, 0.f / f2, 0.f / f2, shape / f2);
        // type = 11 (uint8 @40), extremum_type = +1 / -1 (int8 @41), padding
        const unsigned tail = 11u | ((is_max ? 0x01u : 0xffu) << 8);
        rec[2] = make_float4(theta, d.w, __uint_as_float(tail), 0.f);
        *reinterpret_cast<int2*>(scale_octave + 2 * out) = make_int2(s, o);
      }
      if (!with_descriptors)
        return;

      {
        // 16 bytes per lane and store (the array is padded to a multiple of 4)
        int4* h4 = reinterpret_cast<int4*>(hist);
  #pragma unroll
        for (int q = 0; q < (kDescHistWords + 255) / 256; ++q)
          if (q * 64 + lane < (kDescHistWords + 3) / 4)
            h4[q * 64 + lane] = make_int4(0, 0, 0, 0);
      }

      SARA_PROF_T(t_zero);
#ifdef SARA_DESC_PROF2
      SARA_PROF_ADD(5, t_peak, t_zero);
#endif
      // theta is a refined histogram peak in (-pi, pi] (orientation_kernel).
      // Anything else (never produced) is first brought back by whole turns:
      // no general-range library call - its Payne-Hanek path and constants
      // cost registers (spills reloaded from scratch in every item) for a
      // branch that is never taken.
      double td = double(theta);
      if (!(fabsf(theta) <= 4.f))
        td -= 6.28318530717958647692 * __builtin_rint(td * 0.15915494309189533577);
      double sd, cd;
      sincos_reduced_f64(td, sd, cd, g_sincos_coef);
      const float ct = float(cd);
      const float st = float(sd);
      SARA_PROF_T(t_sc);
#ifdef SARA_DESC_PROF2
      SARA_PROF_ADD(6, t_zero, t_sc);
#endif
      const float T00 = ct / l, T01 = st / l, T10 = (-st) / l, T11 = ct / l;

      // Row intervals are conservative (the exact float test in the sample
      // step decides), hence the approximate reciprocals: the window's edges
      // in a row, u = (-+2.5 - T01 v) / T00, are evaluated with a few float
      // roundings (errors ~1e-5 pixel for coordinates below 100) and widened
      // by kDescMargin.  (Round 2a widened by a whole pixel and then rounded
      // outwards: 3-4 idle samples per row of ~30.)
      constexpr float kDescMargin = 0.02f;
      const bool t00_ok = fabsf(T00) > 1e-12f, t10_ok = fabsf(T10) > 1e-12f;
      const float inv00 = t00_ok ? 1.f / T00 : 0.f;
      const float inv10 = t10_ok ? 1.f / T10 : 0.f;

      // One sample: trilinear accumulation of pixel (u, v) of the patch
      // (SIFT.hpp:204-238).  p = T (u, v) is evaluated in the reference's
      // operation order (the window test is a float comparison).
      auto accumulate = [&](int u, int v, float2 mo) {
        const float fu = float(u), fv = float(v);
        float px = T00 * fu + T01 * fv;
        float py = T10 * fu + T11 * fv;
        const float nrm2 = px * px + py * py;
        px += 1.5f;
        py += 1.5f;
        if (fminf(px, py) <= -1.f || fmaxf(px, py) >= 4.f)
          return;
        // weight * mag * 2^(25 - e), once per sample
        // exp(-nrm2 / 8) as one scaling into the hardware exp2
        const float wm = __builtin_amdgcn_exp2f(nrm2 * float(-0.125 * 1.4426950408889634)) *
                         (mo.x * fx_scale);
        float a = mo.y - theta;
        a = a < 0.f ? a + 2.f * pi : a;
        a *= 8.f / (2.f * pi);
        const float xif = truncf(px), yif = truncf(py), oif = truncf(a);
        const float xfrac = px - xif, yfrac = py - yif, ofrac = a - oif;
        const int xi = int(xif), yi = int(yif), oi = int(oif);
        const float w1 = ofrac * wm, w0 = wm - w1;
        // xi, yi are in 0..3 (p in (-1, 4), truncation): on the 5 x 5 grid the
        // dx / dy = 1 neighbours always exist (dump row / column)
        const float wx1 = xfrac, wy1 = yfrac;
        const float wy0 = 1.f - yfrac, wx0 = 1.f - xfrac;
        const float p00 = wy0 * wx0, p01 = wy0 * wx1, p10 = wy1 * wx0,
                    p11 = wy1 * wx1;
        const unsigned dxo = unsigned(kDescCellStride);
        const unsigned dyo = unsigned(kDescGrid * kDescCellStride);
        // word index of (cell, copy) through the float pipe: yi, xi are small
        // integers held in floats already, the two fused multiply-adds are
        // exact, and one conversion replaces two conversions, a 64-bit
        // multiply-add and a 24-bit multiply
        (void) xi;
        (void) yi;
        const unsigned h0 = unsigned(int(__builtin_fmaf(
            __builtin_fmaf(yif, float(kDescGrid), xif), float(kDescCellStride),
            float(copy))));
        const unsigned ia = h0 + unsigned((oi & 7) * kDescCopies);
        const unsigned ib = h0 + unsigned(((oi + 1) & 7) * kDescCopies);
        // the eight contributions, rounded to nearest (ties up) in one block
        int c0, c1, c2, c3, c4, c5, c6, c7;
        asm("v_cvt_rpi_i32_f32 %0, %8\n\tv_cvt_rpi_i32_f32 %1, %9\n\t"
            "v_cvt_rpi_i32_f32 %2, %10\n\tv_cvt_rpi_i32_f32 %3, %11\n\t"
            "v_cvt_rpi_i32_f32 %4, %12\n\tv_cvt_rpi_i32_f32 %5, %13\n\t"
            "v_cvt_rpi_i32_f32 %6, %14\n\tv_cvt_rpi_i32_f32 %7, %15"
            : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(c4), "=&v"(c5),
              "=&v"(c6), "=&v"(c7)
            : "v"(p00 * w0), "v"(p00 * w1), "v"(p01 * w0), "v"(p01 * w1),
              "v"(p10 * w0), "v"(p10 * w1), "v"(p11 * w0), "v"(p11 * w1));
#define SARA_DESC_ADD(i, val)                                                  \
  atomicAdd(&hist[i], desc_acc_t(val))
        SARA_DESC_ADD(ia, c0);
        SARA_DESC_ADD(ib, c1);
        SARA_DESC_ADD(ia + dxo, c2);
        SARA_DESC_ADD(ib + dxo, c3);
        SARA_DESC_ADD(ia + dyo, c4);
        SARA_DESC_ADD(ib + dyo, c5);
        SARA_DESC_ADD(ia + dxo + dyo, c6);
        SARA_DESC_ADD(ib + dxo + dyo, c7);
#undef SARA_DESC_ADD
      };

      SARA_PROF_T(t_trig);
      SARA_PROF_ADD(1, t_peak, t_trig);
      for (int vb = v_lo; vb <= v_hi; vb += kDescRowsPerBlock)
      {
        SARA_PROF_T(t_blk);
        // ---- this lane's row: conservative u-interval inside the window ----
        const int v = vb + lane;
        int u_first = 0, len = 0;
        if (v <= v_hi)
        {
          const float fv = float(v);
          float lo = float(u_min), hi = float(u_max);
          const float bx_ = T01 * fv, by_ = T11 * fv;
          if (t00_ok)
          {
            const float a = (-2.5f - bx_) * inv00, c = (2.5f - bx_) * inv00;
            lo = fmaxf(lo, fminf(a, c) - kDescMargin);
            hi = fminf(hi, fmaxf(a, c) + kDescMargin);
          }
          else if (fabsf(bx_) > 2.6f)
            hi = lo - 1.f;
          if (t10_ok)
          {
            const float a = (-2.5f - by_) * inv10, c = (2.5f - by_) * inv10;
            lo = fmaxf(lo, fminf(a, c) - kDescMargin);
            hi = fminf(hi, fmaxf(a, c) + kDescMargin);
          }
          else if (fabsf(by_) > 2.6f)
            hi = lo - 1.f;
          // integers of [lo, hi]: the interval already carries the margin
          u_first = max(int(ceilf(lo)), u_min);
          const int u_last = min(int(floorf(hi)), u_max);
          len = max(u_last - u_first + 1, 0);
        }
        const int nch_row = (len + kDescChunk - 1) / kDescChunk;
        const int max_ch = wave_max_dpp(nch_row);

        for (int ph = 0; ph * kDescChunksPerPhase < max_ch; ++ph)
        {
          // ---- chunk list of this (row block, phase) -> LDS -----------------
          const int c_lo = ph * kDescChunksPerPhase;
          const int nch = min(max(nch_row - c_lo, 0), kDescChunksPerPhase);
          const int incl = wave_inclusive_scan(nch);
          const int C = __builtin_amdgcn_readlane(incl, 63);
          __builtin_amdgcn_wave_barrier();  // the previous list is consumed
          // entry j of the list goes to segment j / per, slot j % per; an entry
          // carries its pixel count (1..8; 0 = idle) instead of `last`.
          // j / per as (j * M) >> 16 with M = ceil(2^16 / per): exact for
          // j < 512, per <= 64 (the error term j e / (per 2^16) < 1 / 128 is
          // below the smallest distance 1 / 64 of frac(j / per) from 1)
          const int per_grp = (C + kDescGroups - 1) / kDescGroups;
          const unsigned magic = (65536u + unsigned(per_grp) - 1u) / unsigned(max(per_grp, 1));
          for (int c = 0; c < nch; ++c)
          {
            const int u0rel = u_first - u_min + kDescChunk * (c_lo + c);
            const int cnt = min(kDescChunk, len - kDescChunk * (c_lo + c));
            const unsigned j = unsigned(incl - nch + c);
            const unsigned gq = (j * magic) >> 16;
            tab[gq * kDescSeg + (j - gq * unsigned(per_grp))] =
                unsigned(lane) | (unsigned(cnt) << 6) | (unsigned(u0rel) << 10);
          }
          {
            // idle entries behind every segment (8 lanes per segment, 3 each:
            // the segments are short of at most 7 entries in total, and the
            // pipeline reads kDescSegIdle - 1 past the last step)
            const int cnt_g = min(max(C - grp * per_grp, 0), per_grp);
            static_assert(kDescChunk == 8 && kDescSegIdle + 7 <= 24, "idle cover");
#pragma unroll
            for (int q = 0; q < 3; ++q)
            {
              const int k = cnt_g + l16 + 8 * q;
              if (k < kDescSeg)
                tab[grp * kDescSeg + k] = 0u;
            }
          }
          __builtin_amdgcn_wave_barrier();
          SARA_PROF_T(t_tab);
          SARA_PROF_ADD(2, t_blk, t_tab);

          // ---- stream the chunks: group g walks its own segment of the list --
          // (the g-th contiguous eighth: the chunks a wave works on at one time
          // are then rows apart - about one histogram cell - and the eight
          // ds_add of a step hit different cells instead of one; same-address
          // atomics serialise.  Round 5: 1.50 -> 1.39 ms per 64 x 1080p step,
          // SQ_WAIT_INST_LDS halved.)
          // kDescAhead stages in flight per lane.  A stage holds the chunk
          // entry it is working on, the gathered pair, and the entry fetched
          // for its next use; the loop is unrolled over the stages so that
          // nothing is copied (a copy would wait for the gather).  Entries
          // behind a segment's end are idle (pixel count 0): no bounds checks.
          const unsigned* seg = tab + grp * kDescSeg;
          auto fetch = [&](int step) -> unsigned { return seg[step]; };
          auto gather = [&](unsigned e) -> float2 {
            const int last = int((e >> 6) & 15u) - 1;  // idle entries: -1
            const int vv = vb + int(e & 63u);
            const int uu = u_min + int(e >> 10) + l16;
            const bool act = l16 <= last;
            // Unconditional gather (idle lanes read the keypoint's own pixel):
            // with the load under a branch the compiler cannot count it and
            // waits for vmcnt(0), i.e. also for the gathers it has just issued.
            // rows and widths are below 2^24: one 24-bit multiply-add
            return load_pair32(g, act ? __umul24(unsigned(ry + vv), unsigned(w)) +
                                            unsigned(rx + uu)
                                      : unsigned(center));
          };
          const int nsteps = (C + kDescGroups - 1) / kDescGroups;  // == per_grp
          unsigned ent[kDescAhead], ent_next[kDescAhead];
          float2 data[kDescAhead];
#pragma unroll
          for (int q = 0; q < kDescAhead; ++q)
          {
            ent[q] = fetch(q);
            ent_next[q] = fetch(q + kDescAhead);
          }
#pragma unroll
          for (int q = 0; q < kDescAhead; ++q)
            data[q] = gather(ent[q]);
          // everything older than these gathers has landed: the compiler's
          // wait-count bookkeeping enters the loop with exactly kDescAhead
          // loads pending and can wait for vmcnt(kDescAhead - 1) per step
          // (without this it drains the queue once per unrolled round)
          __builtin_amdgcn_s_waitcnt(0x0f70 | (kDescAhead - 1));
          for (int step0 = 0; step0 < nsteps; step0 += kDescAhead)
          {
#pragma unroll
            for (int q = 0; q < kDescAhead; ++q)
            {
              const int step = step0 + q;  // steps >= nsteps find idle entries
              const unsigned e = ent[q];
              const int last = int((e >> 6) & 15u) - 1;
              if (l16 <= last)
                accumulate(u_min + int(e >> 10) + l16, vb + int(e & 63u), data[q]);
              ent[q] = ent_next[q];
              data[q] = gather(ent[q]);
              ent_next[q] = fetch(step + 2 * kDescAhead);
            }
          }
          SARA_PROF_T(t_steps);
          SARA_PROF_ADD(3, t_tab, t_steps);
#if defined(SARA_DESC_PROF) && !defined(SARA_DESC_PROF2)
          if (lane == 0)
          {
            atomicAdd(&g_desc_prof[5], (unsigned long long) nsteps);
            atomicAdd(&g_desc_prof[6], 1ull);
          }
#endif
        }
      }
      SARA_PROF_T(t_rows);
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)

      double a0 = 0., a1 = 0.;
      {
        // bin = (y * 4 + x) * 8 + o of the reference's layout
        // lane -> cell (lane >> 3) = y * 4 + x of the upper half, + 8 below
        const int cy = lane >> 5, cx = (lane >> 3) & 3;
        const desc_acc_t* q0 = hist + (cy * kDescGrid + cx) * kDescCellStride +
                               (lane & 7) * kDescCopies;
        const desc_acc_t* q1 = q0 + 2 * kDescGrid * kDescCellStride;
        {
          long long s0 = 0, s1 = 0;
#pragma unroll
          for (int c = 0; c < kDescCopies; ++c)
          {
            s0 += (long long) (int) q0[c];
            s1 += (long long) (int) q1[c];
          }
          a0 = double(s0) * fx_inv;
          a1 = double(s1) * fx_inv;
        }
      }
      float h0 = float(a0), h1 = float(a1);
      // normalize(): L2, clamp at 0.2, L2; then x512, clamp at 255.
      float z = wave_sum(h0 * h0 + h1 * h1);
      if (z > 0.f)
      {
        const float nrm = sqrtf(z);
        h0 /= nrm;
        h1 /= nrm;
      }
      h0 = fminf(h0, 0.2f);
      h1 = fminf(h1, 0.2f);
      z = wave_sum(h0 * h0 + h1 * h1);
      if (z > 0.f)
      {
        const float nrm = sqrtf(z);
        h0 /= nrm;
        h1 /= nrm;
      }
      h0 = fminf(h0 * 512.f, 255.f);
      h1 = fminf(h1 * 512.f, 255.f);
      if (root_sift)
      {
        // RootSIFT.hpp:48-50: h /= lpNorm<1>(h); h = sqrt(h) - of the magnitude,
        // sign kept: the base descriptor has negative bins.
        const float l1 = wave_sum(fabsf(h0) + fabsf(h1));
        if (l1 > 0.f)
        {
          h0 = copysignf(sqrtf(fabsf(h0) / l1), h0);
          h1 = copysignf(sqrtf(fabsf(h1) / l1), h1);
        }
      }
      descriptors[out * 128 + lane] = h0;
      descriptors[out * 128 + 64 + lane] = h1;
      __builtin_amdgcn_wave_barrier();
      SARA_PROF_T(t_fin);
      SARA_PROF_ADD(4, t_rows, t_fin);
    }
    SARA_PROF_T(t_end);
    SARA_PROF_ADD(7, t_item, t_end);
    };
    int bx = blockIdx.x;
    int lb = bx < positions ? xcd_local_block(bx, b, nblk, xcd_run) : -1;
    unsigned wv = fetch_item(lb);
    while (bx < positions)
    {
      const int bx_next = bx + gridDim.x;
      const int lb_next =
          bx_next < positions ? xcd_local_block(bx_next, b, nblk, xcd_run) : -1;
      const unsigned wv_next = fetch_item(lb_next);
      if (lb >= 0)
        item(lb, wv);
      bx = bx_next;
      lb = lb_next;
      wv = wv_next;
    }
  }

  void launch_descriptors(const GradPyramidView& grad,
                          const CandidateLists& cand,
                          const OrientationLists& ori, int batch,
                          sara_oeregion* features, int32_t* scale_octave,
                          float* descriptors, int with_descriptors,
                          int root_sift, hipStream_t stream)
  {
    const int unit = 8 * g_xcd_run;
    const int needed =
        unit * (((cand.cap + kDescWaves - 1) / kDescWaves + unit - 1) / unit);
    const dim3 grid(std::min(needed, unit * persist_units(batch, kDescWaves)), batch);
    hipLaunchKernelGGL(descriptor_kernel, grid, dim3(64 * kDescWaves), 0, stream, grad, cand,
                       ori, features, scale_octave, descriptors,
                       with_descriptors, root_sift, g_xcd_run);
  }

  // ------------------------------------------------------------------------ //
  // RootSIFT on a descriptor matrix (FeatureDescriptors/RootSIFT.hpp:45-53):
  // one wave per row, row /= its L1 norm, then the signed square root of
  // every bin.
  // ------------------------------------------------------------------------ //
  __global__ __launch_bounds__(256) void root_sift_kernel(float* __restrict__ desc,
                                                          int n, int dim)
  {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n)
      return;
    float* h = desc + size_t(row) * dim;
    float part = 0.f;
    for (int i = lane; i < dim; i += 64)
      part += fabsf(h[i]);
    const float l1 = wave_sum(part);
    if (!(l1 > 0.f))
      return;
    for (int i = lane; i < dim; i += 64)
      h[i] = copysignf(sqrtf(fabsf(h[i]) / l1), h[i]);
  }

#if defined(SARA_DESC_PROF) || defined(SARA_ORI_PROF)
  extern "C" __attribute__((visibility("default"))) int sara_hip_debug_desc_prof(
      unsigned long long* out, int reset)
  {
    unsigned long long z[8] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_desc_prof), sizeof(z)) != hipSuccess)
      return -1;
    if (reset)
      (void) hipMemcpyToSymbol(HIP_SYMBOL(g_desc_prof), z, sizeof(z));
    return 0;
  }
#endif

  void launch_root_sift(float* desc, int n, int dim, hipStream_t stream)
  {
    if (n <= 0)
      return;
    hipLaunchKernelGGL(root_sift_kernel, dim3((n + 3) / 4), dim3(256), 0, stream,
                       desc, n, dim);
  }

  // ------------------------------------------------------------------------ //
  // Extrema before orientation assignment, in reference order.
  // ------------------------------------------------------------------------ //
  __global__ void gather_extrema_kernel(CandidateLists cand,
                                        const int* __restrict__ ex_offset,
                                        sara_oeregion* __restrict__ regions,
                                        int32_t* __restrict__ xyso_type)
  {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = min(cand.count[b], cand.cap);
    if (idx >= n)
      return;
    const size_t row = size_t(b) * cand.cap;
    const int slot = cand.order[row + idx];
    const unsigned long long key = cand.key[row + slot];
    const float4 d = cand.data[row + slot];
    const size_t out = size_t(ex_offset[b]) + idx;
    if (regions)
    {
      sara_oeregion f;
      f.coords[0] = d.x;
      f.coords[1] = d.y;
      f._pad0[0] = f._pad0[1] = 0.f;
      const float shape = float(1.0 / (double(d.z) * double(d.z)));
      f.shape_matrix[0] = shape;
      f.shape_matrix[1] = 0.f;
      f.shape_matrix[2] = 0.f;
      f.shape_matrix[3] = shape;
      f.orientation = 0.f;
      f.extremum_value = d.w;
      f.type = 11;
      f.extremum_type = (key & 1ull) ? 1 : -1;
      for (int q = 0; q < 6; ++q)
        f._pad1[q] = 0;
      regions[out] = f;
    }
    if (xyso_type)
    {
      xyso_type[5 * out + 0] = key_x(key);
      xyso_type[5 * out + 1] = key_y(key);
      xyso_type[5 * out + 2] = key_scale(key);
      xyso_type[5 * out + 3] = key_octave(key);
      xyso_type[5 * out + 4] = (key & 1ull) ? 1 : -1;
    }
  }

  void launch_gather_extrema(const CandidateLists& cand, const int* ex_offset,
                             int batch, sara_oeregion* regions,
                             int32_t* xyso_type, hipStream_t stream)
  {
    const dim3 grid((cand.cap + 255) / 256, batch);
    hipLaunchKernelGGL(gather_extrema_kernel, grid, dim3(256), 0, stream, cand,
                       ex_offset, regions, xyso_type);
  }

}  // namespace sara_hip
