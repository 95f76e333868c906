// Launch interface between the host side (sift_host.hpp: context, detect()) and the gfx950
// kernels (sift_kernels.hip).  Internal header, not part of the C-ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <mutex>

#include "../../include/sara_hip_sift.h"

namespace sara_hip {

  constexpr int kMaxRadius = 56;          // Gaussian taps <= 113 (LDS: 160 KB)
  constexpr int kMaxTaps = 2 * kMaxRadius + 1;
  constexpr int kMaxScales = 16;          // scale_count_per_octave upper bound
  constexpr int kMaxPeaks = 18;           // a 36-bin histogram has <= 18 peaks
  constexpr int kOriBins = 36;

  struct Taps
  {
    int size;  // odd
    float k[kMaxTaps];
  };

  //! One octave of one pyramid kind in HBM: planes [frame][scale][h][w].
  struct OctaveView
  {
    float* base;
    int w, h;
    int scales;          // planes per frame
    size_t plane;        // w*h
    size_t frame_stride; // scales*plane (x2 floats for the gradient pyramid)
  };

  //! Per-octave constants of the extremum / descriptor stages.
  struct ScaleTable
  {
    float sigma[kMaxScales];     // float(pow(k, s) * sigma0), ImagePyramid.hpp:316-319
    double sigma_d[kMaxScales];  // the same before rounding (scale plausibility test)
    int ori_radius[kMaxScales];  // int_round(sigma*1.5f*3.f), Orientation.hpp:105-108
    float ori_sigma[kMaxScales]; // sigma * 1.5f
    int ori_woff[kMaxScales];    // offset of the weight table of scale s
    // Orientation.hpp:118-119: bin = int(floor(double(ori / float(2 pi) * 36)))
    // is non-decreasing in ori: thr[k] = smallest float >= 0 whose bin is >= k
    // (k = 0 .. 37, +inf where no angle gets there), found by bisection on the
    // host with the expression itself.  The kernel estimates the bin with one
    // multiplication and corrects it by at most one against thr[].
    float ori_bin_thr[40];
  };

  struct ExtremaParams
  {
    float extremum_thres;
    float edge_ratio_thres;
    int img_padding_sz;
    int refine_iters;
    float scale_geometric_factor;
    int signed_type;  // SARA_HIP_OPT_SIGNED_EXTREMUM_TYPE
  };

  //! Unordered candidate lists of the extremum scan, per frame.
  struct CandidateLists
  {
    unsigned long long* key; // [frame][cap]  (o,s,y,x | type)
    float4* data;            // [frame][cap]  (x, y, sigma, value) refined
    int* count;              // [frame]       number appended (may exceed cap)
    int* order;              // [frame][cap]  slot of the rank-th candidate
    // copies in rank (= reference) order, written with `order`: the
    // per-keypoint kernels read them without the indirection
    unsigned long long* skey;  // [frame][cap]
    float4* sdata;             // [frame][cap]
    int cap;
    int* error;                // see SiteLists::error
  };

  //! Everything the descriptor kernel needs to know about one extremum, in
  //! rank order, written by the orientation kernel: one 64-byte load instead
  //! of a chain of dependent ones (count -> slot -> key / data -> angles).
  struct alignas(16) KeypointRecord
  {
    float4 d;                // (x, y, sigma, value), octave coordinates
    unsigned long long key;  // (octave, scale, y, x | type)
    int npeaks;
    int reserved;
    float theta[8];          // first 8 peak angles (the rest: peak_theta)
  };
  static_assert(sizeof(KeypointRecord) == 64, "record layout");

  //! One keypoint = one (extremum, dominant orientation) pair, in output order,
  //! written by the peak scan: the work item of the descriptor kernel.
  struct alignas(16) KeypointItem
  {
    float4 d;                // (x, y, sigma, value), octave coordinates
    unsigned long long key;  // (octave, scale, y, x | type)
    float theta;
    int reserved;
  };
  static_assert(sizeof(KeypointItem) == 32, "item layout");

  //! Buckets of the counting sort that orders a frame's extrema: one per image
  //! row of every (octave, scale) plane.  base[o * kMaxScales + s] = first
  //! bucket of that plane; `total` buckets per frame, rows of `stride` ints
  //! (>= total + 1) in the per-frame arrays.
  struct RowBuckets
  {
    int base[16 * kMaxScales];
    int total;
    int stride;
  };

  //! Classified extremum sites of the marching scan, before the edge test and
  //! the refinement (same key layout as CandidateLists::key).
  //! nb: what the edge test and the first refinement step of a site read - the
  //! DoG values around it, captured by the scan from the rows it holds in
  //! registers (kSiteNb floats per site: layer s as a 3 x 3 block, row-major
  //! (y-1, y, y+1) x (x-1, x, x+1); then centre, left, right, up, down of layer
  //! s-1 and the same of layer s+1; one pad).  finish_sites_kernel then reads
  //! 88 contiguous bytes per site instead of gathering 36 Gaussian values
  //! through the planes (0.87 GB per 64 x 1080p step in round 3).
  constexpr int kSiteNb = 20;
  struct SiteLists
  {
    unsigned long long* key;  // [frame][cap]
    float* nb;                // [frame][cap][kSiteNb]
    int* count;               // [frame] (may exceed cap)
    int cap;
    int* error;               // one int per context: set to 1 by a kernel that
                              // meets a NEGATIVE list counter (a counter the
                              // step did not zero); counts() / collect() then
                              // fail instead of returning truncated lists
  };

  //! Gaussian pyramid of the batch: octave o at base[o], planes
  //! [frame][scale][h][w].
  struct OctavePyramidView
  {
    const float* base[16];
    int w[16], h[16];
    size_t plane[16];
    size_t frame_stride[16];
    int scales;
    int octaves;
  };

  struct OrientationLists
  {
    int* peak_count;   // [frame][cap]
    float* peak_theta; // [frame][cap][kMaxPeaks]
    int* offset;       // [frame][cap] exclusive prefix of peak_count
    int* kp_count;     // [frame] keypoints of the frame (may exceed cap)
    int* frame_offset; // [batch+1] exclusive prefix of min(kp_count, cap)
    KeypointRecord* record;  // [frame][cap]
    KeypointItem* item;      // [frame][cap], first min(kp_count, cap) valid
  };

  inline unsigned long long make_key(int o, int s, int y, int x, int is_max)
  {
    return ((((unsigned long long) (o * kMaxScales + s) << 20 | (unsigned) y)
             << 20 | (unsigned) x)
            << 1) |
           (unsigned) is_max;
  }

  // ---- pyramid -------------------------------------------------------------
  //! dst = gaussian(src) [rows then columns, replicate borders]; when dog is
  //! not null also dog = dst - src.  Planes are w x h; frame b of each
  //! operand lives at base + b*stride.
  //! When `dec` is given and the fast path runs, the kernel also writes
  //! dec(x, y) = dst(2x, 2y) (planes of (w/2) x (h/2), frame stride
  //! dec_stride) and the function returns true; otherwise the caller has to
  //! run launch_scale itself.  fma: the marching kernels fuse multiply and add
  //! (SARA_HIP_OPT_FMA_BLUR; not bit-exact with the reference).
  bool launch_gaussian_blur(const float* src, size_t src_stride, float* dst,
                            size_t dst_stride, float* dog, size_t dog_stride,
                            int w, int h, int batch, const Taps& taps,
                            hipStream_t stream, float* dec = nullptr,
                            size_t dec_stride = 0, bool fma = false);

  //! One frame per call: several independent tiled blurs (different octaves,
  //! same depth of the dependency graph) as ONE kernel node
  //! (gaussian_blur_level_kernel, pyramid_kernels.hip).  false = nothing
  //! launched (a radius outside {5, 6, 8, 10, 12} or too many members).
  constexpr int kBlurLevelMaxMembers = 4;
  struct BlurLevelBlur
  {
    const float* src;
    float* dst;
    float* dec;  // also write the half-size plane here (nullptr: no)
    size_t src_stride, dst_stride, dec_stride;  // per frame, in floats
    int w, h;
    const Taps* taps;
  };
  bool blur_level_radius_ok(int taps_size);
  bool launch_blur_level(const BlurLevelBlur* blurs, int n, int batch,
                         hipStream_t stream);

  //! The same blur reading 8-bit gray frames (src_stride in bytes), converted
  //! on the fly as float(v) / 255.f.  Returns false (nothing launched) when the
  //! marching kernel cannot take the shape / radius: the caller then converts
  //! into a float plane first.
  bool launch_gaussian_blur_gray8(const unsigned char* src, size_t src_stride,
                                  float* dst, size_t dst_stride, int w, int h,
                                  int batch, const Taps& taps, hipStream_t stream);

  //! Nearest-neighbour resize (Resize.cpp:31-62).
  void launch_scale(const float* src, size_t src_stride, int sw, int sh,
                    float* dst, size_t dst_stride, int dw, int dh, int batch,
                    hipStream_t stream);

  //! Bilinear enlarge in double (Resize.cpp:86-128).
  void launch_enlarge(const float* src, size_t src_stride, int sw, int sh,
                      float* dst, size_t dst_stride, int dw, int dh, int batch,
                      hipStream_t stream);

  void launch_copy_planes(const float* src, size_t src_stride, float* dst,
                          size_t dst_stride, size_t count, int batch,
                          hipStream_t stream);

  //! Zeroes `count` ints (count a multiple of 64).  The per-step counters are
  //! cleared by a kernel of the library, not by hipMemsetAsync: inside a graph
  //! captured from ONE stream the ROCm 7.0 runtime let the scan run on counters
  //! its memset node had not cleared yet (round 4; the lists then came back
  //! truncated with status OK).  A kernel node is ordered like every other
  //! kernel of the chain.
  void launch_zero_counters(int* counters, size_t count, unsigned* epoch, int stamp,
                            hipStream_t stream);
  void launch_subtract(const float* a, const float* b, float* out, size_t count,
                       hipStream_t stream);

  //! 8-bit frames (channels = 3: interleaved RGB, 1: gray) -> gray32f planes;
  //! strides in bytes (src) and floats (dst), count = pixels per frame.
  void launch_u8_to_gray32f(const unsigned char* src, size_t src_stride,
                            int channels, float* dst, size_t dst_stride,
                            size_t count, int batch, hipStream_t stream);

  // ---- gradients -----------------------------------------------------------
  //! (2*|grad|, atan2(gy,gx)) of `nscales` consecutive planes per frame.
  //! cmax (optional): coarse map of the gradient magnitude, one uint32 (the
  //! float's bits, magnitudes are >= 0) per 16x16 block, planes
  //! [frame][scale][ceil(h/16)][ceil(w/16)], frame stride cmax_stride; the
  //! kernels atomicMax into it, so it has to be zeroed before the launch.
  //! Whether launch_gradient_polar with these operands accumulates the coarse
  //! maxima with atomicMax (the map then has to be zeroed first) or writes
  //! every entry exactly once (marching kernel: no memset needed).
  bool gradient_polar_needs_zeroed_cmax(const float* src, size_t src_stride,
                                        const float* dst, size_t dst_stride,
                                        int w, int h, int batch);
  void launch_gradient_polar(const float* src, size_t src_stride, float* dst,
                             size_t dst_stride, int w, int h, int nscales,
                             int batch, hipStream_t stream,
                             unsigned* cmax = nullptr, size_t cmax_stride = 0);

  //! Octaves one multi-octave scan launch takes.
  constexpr int kScanMultiMax = 8;

  // ---- extrema -------------------------------------------------------------
  //! Scans DoG scales 1..S-3 of one Gaussian octave.  The fast path only
  //! classifies and appends to `sites` (finish with launch_finish_sites once
  //! all octaves are scanned); the general path refines and appends to `cand`
  //! directly.
  void launch_extrema_scan(const OctaveView& gauss, int octave, int batch,
                           const ExtremaParams& p, const ScaleTable* tab,
                           const CandidateLists& cand, const SiteLists& sites,
                           hipStream_t stream);
  //! One frame per call: the scans of up to kScanMultiMax octaves in ONE launch
  //! (extrema_march_multi_kernel).  false = nothing launched (signed-type mode,
  //! a scale count other than 6, the general kernels selected).
  bool launch_extrema_scan_multi(const OctaveView* gauss, const int* octaves, int n,
                                 int batch, const ExtremaParams& p,
                                 const SiteLists& sites, hipStream_t stream);

  //! Edge test + refinement + contrast test of the classified sites.
  void launch_finish_sites(const OctavePyramidView& pyr, int batch,
                           const ExtremaParams& p, const ScaleTable* tab,
                           const SiteLists& sites, const CandidateLists& cand,
                           hipStream_t stream);

  void launch_extremum_map(const float* a, const float* b, const float* c,
                           int w, int h, float edge_ratio, float thres, int pad,
                           int8_t* out, hipStream_t stream);

  //! order[b][rank] = slot, rank = number of smaller keys in the frame.
  //! Same result through a counting sort on the (octave, scale, y) key prefix
  //! (hist, cursor: [batch][rb.stride] ints; grouped: [batch][cap] ints).
  void launch_rank_candidates_bucketed(const CandidateLists& cand,
                                       const RowBuckets& rb, int* hist,
                                       int* cursor, int* grouped, int batch,
                                       hipStream_t stream);

  // ---- orientation / descriptors ----------------------------------------------
  struct GradPyramidView
  {
    const float* base[16]; // per octave, planes [frame][scale][h][w][2]
    int w[16], h[16];
    size_t plane[16];        // w*h (in pixels)
    size_t frame_stride[16]; // in floats
    float factor[16];        // octave scaling factor
    int octaves;
    // coarse magnitude maxima (16x16 blocks), planes [frame][scale][ch][cw]
    const unsigned* cmax[16];
    int cw[16], ch[16];
    size_t cmax_frame_stride[16];  // in uint32
  };

  //! Which kernel a launch takes and how it is cut (round 6).  One value per
  //! context (sara_hip_sift_set_option: SARA_HIP_OPT_KERNEL_SELECTION,
  //! _TILE_GEOMETRY) instead of process-wide switches read once from the
  //! environment: the parity tests run the forced AND the shipped selection in
  //! one process, bench.py sweeps config 5's decompositions in one run.  The
  //! environment still provides the DEFAULT of a new context
  //! (environment_selection()), so tools/test_modes.sh works as before.
  struct KernelSelection
  {
    bool blur_march = true;      //!< false: tiled blur everywhere (SARA_HIP_BLUR=tile)
    bool feature_march = true;   //!< false: pixel-parallel gradient / scan (SARA_HIP_FEATURES=tile)
    int march_waves = 4096;      //!< target waves per launch, 4-column blur (SARA_HIP_MARCH_WAVES)
    int march2_waves = 3072;     //!< the same, 2-column blur (SARA_HIP_MARCH2_WAVES)
    size_t march_min_pixels = size_t(4) << 20;  //!< smaller launches take the tiled blur
    int strip_group = 0;         //!< 0: production rule; 1 / 4 / 8 forced (SARA_HIP_STRIP_GROUP)
    long long grad_tile_pixels = (long long) 4 << 20;  //!< smaller planes: pixel-parallel gradient
    int tile_geometry = 0;       //!< tiled blur: 0 by tile count, 1 = 64x32, 2 = 64x16, 3 = 32x16
    bool xcd_map = true;         //!< XCD-aware placement of marching workgroups (SARA_HIP_XCD_MAP)
    bool level_merge = true;     //!< one frame per call: same-depth blurs of different octaves in one launch (SARA_HIP_LEVELS=0: off)
  };
  //! What the process environment asks for (read once); without any variable
  //! set it equals KernelSelection{} = the shipped selection.
  const KernelSelection& environment_selection();
  //! The selection of the calling thread: the one installed by the innermost
  //! ScopedSelection, else environment_selection() (operator-level seams).
  const KernelSelection& selection();
  struct ScopedSelection
  {
    explicit ScopedSelection(const KernelSelection* s);
    ~ScopedSelection();
    ScopedSelection(const ScopedSelection&) = delete;
    ScopedSelection& operator=(const ScopedSelection&) = delete;
    const KernelSelection* before;
  };
  //! Largest strip-group workgroup (1, 4 or 8 waves) a marching launch of
  //! `waves` waves may use: 8 from 4096 waves, 4 from 2048 (the launch fills the
  //! chip anyway), else 1; KernelSelection::strip_group overrides it for tests.
  inline int strip_group_limit(int waves)
  {
    const int forced = selection().strip_group;
    if (forced > 0)
      return forced;
    return waves >= 4096 ? 8 : (waves >= 2048 ? 4 : 1);
  }

  //! Waves per strip-group workgroup for a row of `nstrips` strips: 8 or 4 when
  //! the strips fill whole groups or leave at most a quarter of the last group
  //! idle (its surplus waves leave at once; the barrier counts live waves) -
  //! 3840 columns are 15 blur strips and 31 scan strips, which a whole-groups
  //! rule left to single-wave workgroups (16 x 4K, same box: 6.46 / 6.55 against
  //! 6.51 / 6.60 ms per step).
  inline int strip_group_size(int nstrips, int limit)
  {
    auto fits = [&](int g) {
      return nstrips >= g && 4 * ((g - nstrips % g) % g) <= g;
    };
    return (limit >= 8 && fits(8)) ? 8 : ((limit >= 4 && fits(4)) ? 4 : 1);
  }

  //! grad / tab are device pointers (indexed per wave, so they live in HBM
  //! rather than in the kernel argument segment).
  void launch_orientations(const GradPyramidView* grad, const ScaleTable* tab,
                           const double* ori_weights, int n_weights,
                           const CandidateLists& cand,
                           const OrientationLists& ori, int batch,
                           hipStream_t stream);

  //! Per-frame exclusive scan of peak counts, keypoint list, frame offsets
  //! (done_counter: one int zeroed before the launch).
  void launch_scan_peaks(const CandidateLists& cand, const OrientationLists& ori,
                         int* done_counter, int batch, hipStream_t stream);

  void launch_descriptors(const GradPyramidView& grad,
                          const CandidateLists& cand,
                          const OrientationLists& ori, int batch,
                          sara_oeregion* features, int32_t* scale_octave,
                          float* descriptors, int with_descriptors,
                          int root_sift, hipStream_t stream);
  void launch_root_sift(float* desc, int n, int dim, hipStream_t stream);
  void launch_device_math_selfcheck(unsigned long long* out, hipStream_t stream);
  void launch_orientation_bin_selfcheck(const float* thr, unsigned long long* bad,
                                        hipStream_t stream);
  void launch_definiteness_selfcheck(const float* H, const int* type, int n,
                                     unsigned char* out, hipStream_t stream);

  //! Sorted extrema (before orientation assignment) as OERegion + site.
  void launch_gather_extrema(const CandidateLists& cand, const int* ex_offset,
                             int batch, sara_oeregion* regions,
                             int32_t* xyso_type, hipStream_t stream);

  //! ex_offset[b] = exclusive prefix of min(count[b], cap), b in [0, batch].
  void launch_extrema_offsets(const CandidateLists& cand, int* ex_offset,
                              int batch, hipStream_t stream);

  // ---- descriptor matching (match_kernels.hip, sift_match.cpp) ---------------
  //! One neighbour of a query inside its search radius.
  struct MatchNeighbour
  {
    int32_t query;
    int32_t index;
    float distance;
  };
  //! Up to six int arrays cleared by ONE launch (launch_zero_ranges): the
  //! matcher's counters, flags and rank arrays of a call.  Round 4 cleared them
  //! with a hipMemsetAsync each, which the runtime turns into up to three fill
  //! kernels per call site - 17 fills per match call, a fifth of its GPU time.
  //! Integer ranges one launch clears.  kZeroRangesPerLaunch of them ride in the
  //! kernel's argument; a caller that registers more gets further launches
  //! (launch_zero_ranges) - a range is never dropped: callers tell their tails
  //! "your scratch is cleared" on the strength of having added it here.
  constexpr int kZeroRangesPerLaunch = 6;
  struct ZeroRangesArg
  {
    int* p[kZeroRangesPerLaunch] = {};
    unsigned n[kZeroRangesPerLaunch] = {};  // ints
  };
  struct ZeroRanges
  {
    static constexpr int kMax = 4 * kZeroRangesPerLaunch;
    int* p[kMax] = {};
    unsigned n[kMax] = {};  // ints
    int count = 0;
    bool overflow = false;  // more than kMax ranges: launch_zero_ranges fails loudly
    void add(void* ptr, size_t ints)
    {
      if (!ptr || !ints)
        return;
      if (count == kMax)
      {
        overflow = true;
        return;
      }
      p[count] = static_cast<int*>(ptr);
      n[count] = unsigned(ints);
      ++count;
    }
  };
  //! false (nothing launched, the thread's error text set by the caller's
  //! HIPM_TRY through hipErrorInvalidValue) when ranges were lost.
  bool launch_zero_ranges(const ZeroRanges& r, hipStream_t stream);

  // ---- a batch of independent pairs in ONE set of launches (round 6) ---------
  //! Everything the kernels of the best-match search (squared ratio <= 1) need
  //! to know about one pair: a table of these lives in HBM and the pair is a
  //! grid dimension of every launch.  Scratch pointers address the batch's
  //! arenas (match_batch_layout).
  struct MatchBatchPair
  {
    const float* d1;
    const float* d2;
    int n1, n2;
    float* na;             // |row|^2 of the first set
    float* nb;             // ... of the second
    unsigned* maxbits;     // [0] max of na, [1] max of nb (bit patterns)
    unsigned short* split1;  // hi | lo bf16 rows of the first set
    unsigned short* split2;
    float* rowmin;         // packed minima [tiles along B][n1] int4
    float* colmin;         // ... [tiles along A][n2]
    int* cnt_r;            // candidates claimed per query, then cnt_c, then scal[16]
    int* cnt_c;
    int* scal;             // [0] flagged rows, [1] flagged columns
    int* flag_r;
    int* flag_c;
    int* cand_r;           // [n1][cap]
    int* cand_c;           // [n2][cap]
    float* top_d;          // [3][n1] then [3][n2]
    int* top_i;
    sara_match* tmp;       // unsorted list (n1 + n2 records)
    int* rank;             // (n1 + n2)
    int* header;           // [0] = matches found (4 ints)
    sara_match* out;       // sorted list, right behind the header
  };
  //! Bytes of the three arenas of a batch: `zero` (cleared by one launch at the
  //! head of the batch), `work`, and `out` (headers + lists, read back in one
  //! copy); the table itself is P * sizeof(MatchBatchPair) more.
  struct MatchBatchLayout
  {
    size_t zero_bytes = 0, work_bytes = 0, out_bytes = 0;
  };
  //! Candidate slots per query of the batched search (ratios <= 1).
  constexpr int kMatchBatchCap = 8;
  //! Carves pair p of the batch out of the arenas (pass nullptr bases to get
  //! the sizes only); the arenas grow by this pair's share.
  void match_batch_carve(int n1, int n2, int dim, unsigned char* zero_base,
                         unsigned char* work_base, unsigned char* out_base,
                         MatchBatchLayout* at, MatchBatchPair* pair);
  //! The whole search + tail for `n_pairs` pairs: `table` is the DEVICE copy of
  //! the carved pairs (`host_table` the host's: the tile kernel takes its pointers
  //! as kernel arguments), n1_max / n2_max the largest set sizes, `zero` the zero
  //! arena (cleared here).  AnnMatcher::compute_matches for squared ratios <= 1.
  void launch_match_batch(const MatchBatchPair* table, const MatchBatchPair* host_table,
                          int n_pairs, int n1_max, int n2_max, int dim,
                          float squared_ratio_thres, void* zero, size_t zero_bytes,
                          hipStream_t stream);
  //! The batch's sorted lists back to back in `dense` (room for every pair's
  //! n1 + n2 records), heads[p] = where pair p's starts, heads[n_pairs] = total.
  void launch_compact_lists(const MatchBatchPair* table, int n_pairs, int n_max,
                            sara_match* dense, int* heads, hipStream_t stream);
  //! mutual filter + rank sort of every pair (match_kernels.hip).
  void launch_finish_matches_batch(const MatchBatchPair* table, int n_pairs,
                                   int n_max, float squared_ratio_thres,
                                   hipStream_t stream);

  //! (query block, candidate chunk) decomposition of an exhaustive search.
  void match_chunking(int nq, int nt, int* chunk, int* nchunks);
  //! knnSearch(3) of every row of `q` in `t` (exhaustive, FLANN's squared L2,
  //! ordered by (distance, index)): top_d / top_i = [3][nq].  part_d / part_i:
  //! scratch of 3 * nchunks * nq entries.
  void launch_nn3_exhaustive(const float* q, int nq, const float* t, int nt,
                             int dim, float* part_d, int* part_i, float* top_d,
                             int* top_i, hipStream_t stream);
  //! Ratio test for squared thresholds <= 1 on the device (best neighbour only).
  void launch_ratio_filter(const float* top_d, const int* top_i, int nq,
                           float squared_ratio_thres, int direction,
                           sara_match* out, int capacity, int* count,
                           hipStream_t stream);
  //! compute_matches' tail for squared ratios <= 1, entirely on the device:
  //! ratio test of both directions, duplicates (x, y) dropped, sorted by
  //! (score, x, y).  scratch: n1 + n2 records, rank_scratch: n1 + n2 ints;
  //! *count: number of matches.
  void launch_finish_matches(const float* top_d0, const int* top_i0, int n1,
                             const float* top_d1, const int* top_i1, int n2,
                             int have0, int have1, float squared_ratio_thres,
                             sara_match* scratch, int* rank_scratch, int* count,
                             sara_match* out, hipStream_t stream,
                             bool scratch_cleared = false);
  //! compute_matches' tail for squared ratios > 1 on the device: the radius
  //! members of both directions (unordered triples, counts on the device, lists
  //! of cap0 / cap1 entries) -> matches ordered by (score, x, y) in `out`
  //! (cap0 + cap1 entries), header[0] = their number, header[1] != 0 when a
  //! member list overflowed (nothing delivered is trustworthy then).  iscratch:
  //! finish_radius_scratch_ints() ints, scratch: cap0 + cap1 records.
  size_t finish_radius_scratch_ints(int cap0, int cap1);
  void launch_finish_radius_matches(const MatchNeighbour* m0, const int* c0, int cap0,
                                    const MatchNeighbour* m1, const int* c1, int cap1,
                                    const float* top_d0, int n1, const float* top_d1,
                                    int n2, float squared_ratio_thres, int* iscratch,
                                    sara_match* scratch, int* header, sara_match* out,
                                    hipStream_t stream, bool scratch_cleared = false);
  //! The ints of `iscratch` launch_finish_radius_matches() expects to be zero.
  size_t finish_radius_cleared_ints(int cap0, int cap1);
  //! radiusSearch of every query: neighbours with distance <
  //! top_d[top1][query] * squared_ratio_thres, appended in no particular order.
  void launch_radius_exhaustive(const float* q, int nq, const float* t, int nt,
                                int dim, const float* top_d, int top1,
                                float squared_ratio_thres, MatchNeighbour* out,
                                int capacity, int* count, hipStream_t stream);

  // MFMA prefilter + exact re-ranking (match_mfma.hip): the same answers as the
  // exhaustive kernels for both directions at once.
  size_t match_mfma_scratch_floats(int n1, int n2);
  size_t match_mfma_scratch_ints(int n1, int n2, int cap);
  void launch_match_mfma(const float* d1, int n1, const float* d2, int n2, int dim,
                         float squared_ratio_thres, int top1, int with_dir1,
                         float* fscratch, int* iscratch, int cap,
                         float* top12_d, int* top12_i, float* top21_d, int* top21_i,
                         MatchNeighbour* radius12, int radius12_cap, int* radius12_count,
                         MatchNeighbour* radius21, int radius21_cap, int* radius21_count,
                         hipStream_t stream, const ZeroRanges* also_clear = nullptr);

  // ---- hand-off between the context and the RCCL gather (sift_comm.cpp) ------
  //! Device-resident results of one submit() ticket.
  struct TicketResults
  {
    int device = 0;
    int batch = 0;
    int total = 0;                 // keypoints of the batch
    const int* h_offsets = nullptr;  // batch + 1, pinned host memory
    const sara_oeregion* d_feat = nullptr;
    const float* d_desc = nullptr;
    const int32_t* d_so = nullptr;
    bool capacity_exceeded = false;
    sara_hip_stage last_stage = SARA_HIP_STAGE_DESCRIPTOR;  // of the submit()
  };
  //! Waits for the batch of `ticket` and describes where its results are in
  //! HBM; the ticket stays pending until ticket_release().
  sara_hip_status ticket_results(sara_hip_sift* ctx, int ticket, TicketResults* out);
  void ticket_release(sara_hip_sift* ctx, int ticket);
  //! Process-wide lock around graph capture / launch and the creation /
  //! destruction of contexts, streams and graphs (graph_launcher.cpp says why).
  std::recursive_mutex& runtime_mutex();
  //! Records the error message sara_hip_last_error() returns on this thread.
  sara_hip_status set_error(sara_hip_status code, const char* msg);

#if defined(__HIPCC__)
  //! (strip, segment, frame) of a marching workgroup.  Workgroups are handed
  //! to the 8 XCDs round-robin in launch order and every XCD has its own L2, so
  //! with the plain order (strip fastest) neighbouring strips - which share
  //! their halo columns - and neighbouring segments - which share 2R halo rows
  //! - always sit on different XCDs and every halo line is fetched once per
  //! XCD.  With xcd_total > 0 (1-D grid of 8 * ceil(total / 8) groups) XCD k
  //! takes the k-th contiguous eighth of the (frame, segment, strip) list, so
  //! neighbours meet in one L2.  xcd_total == 0: plain 2-D grid.
  __device__ inline bool march_work_item(int nstrips, int nseg, int xcd_total,
                                         int& strip, int& seg, size_t& b)
  {
    if (xcd_total > 0)
    {
      const int per = (xcd_total + 7) >> 3;
      const int item = int(blockIdx.x & 7u) * per + int(blockIdx.x >> 3);
      if (item >= xcd_total)
        return false;
      strip = item % nstrips;
      const int t = item / nstrips;
      seg = t % nseg;
      b = size_t(t / nseg);
      return true;
    }
    strip = blockIdx.x % nstrips;
    seg = blockIdx.x / nstrips;
    b = blockIdx.y;
    return true;
  }

#endif

  //! XCD-aware placement of the marching workgroups (march_work_item);
  //! KernelSelection::xcd_map = false restores the plain launch order.
  inline bool xcd_map_enabled() { return selection().xcd_map; }

}  // namespace sara_hip
