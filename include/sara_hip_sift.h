/* ========================================================================== *
 * sara_hip_sift.h — C-ABI of the MI355X-native SIFT front-end.
 *
 * Drop-in boundary for Sara's CPU SIFT path (reference paths are relative to
 * /root/reference/cpp/src/DO/Sara):
 *
 *   compute_sift_keypoints()            FeatureDetectors/SIFT.hpp:24-33
 *   ComputeDoGExtrema                   FeatureDetectors/DoG.hpp:72-165
 *   ImagePyramidParams / ImagePyramid   ImageProcessing/ImagePyramid.hpp:29-340
 *   OERegion / KeypointList             Features/Feature.hpp:40-179,
 *                                       Features/KeypointList.hpp:35-96
 *   from_rgb8_to_gray32f()              ImageProcessing/FastColorConversion.cpp:42-66
 *   AnnMatcher (both constructors)      FeatureMatching/AnnMatcher.hpp:32-86,
 *                                       AnnMatcher.cpp:59-268
 *
 * Plain C: pointers, sizes and PODs only.  No exceptions cross this boundary;
 * every entry point returns a sara_hip_status and the message is available
 * from sara_hip_last_error().  The C++ shim include/DO/Sara/HipSift.hpp maps
 * the codes back onto the exception classes the reference throws.
 *
 * Threading: one context per (host thread, device).  A context is not
 * thread-safe; distinct contexts are independent.
 * ========================================================================== */
#ifndef SARA_HIP_SIFT_H
#define SARA_HIP_SIFT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#  define SARA_HIP_API __attribute__((visibility("default")))
#else
#  define SARA_HIP_API
#endif

/* -------------------------------------------------------------------------- */
/* Status codes.  The shim re-throws: INVALID_PARAMS -> std::runtime_error     */
/* (DoG.hpp:86-89), SIZE_MISMATCH -> std::domain_error                         */
/* (LinearFiltering.cpp:33-35), OUT_OF_RANGE -> std::out_of_range              */
/* (GaussianPyramid.hpp:187-190).                                              */
/* -------------------------------------------------------------------------- */
typedef enum sara_hip_status
{
  SARA_HIP_OK = 0,
  SARA_HIP_INVALID_PARAMS = 1,
  SARA_HIP_SIZE_MISMATCH = 2,
  SARA_HIP_OUT_OF_RANGE = 3,
  SARA_HIP_CAPACITY_EXCEEDED = 4, /* image/batch/keypoints above the ctx capacity */
  SARA_HIP_RUNTIME_ERROR = 5,     /* a HIP call failed, message has the detail    */
  SARA_HIP_NO_DEVICE = 6,
  SARA_HIP_NOT_READY = 7,         /* results requested before any detect()        */
  SARA_HIP_RCCL_ERROR = 8         /* librccl missing or a collective call failed  */
} sara_hip_status;

/* ImagePyramidParams — ImageProcessing/ImagePyramid.hpp:29-52 (same member   */
/* meaning and defaults; sara_hip_default_pyramid_params() fills them).        */
typedef struct sara_pyramid_params
{
  int32_t first_octave_index;     /* default -1 */
  int32_t scale_count_per_octave; /* default 6 = 3 + 3 */
  float scale_geometric_factor;   /* default powf(2, 1/3) */
  int32_t image_padding_size;     /* default 1 */
  float scale_camera;             /* default 0.5 */
  float scale_initial;            /* default 1.6 */
  int32_t num_octaves_max;        /* default INT_MAX */
} sara_pyramid_params;

/* Arguments of compute_sift_keypoints — FeatureDetectors/SIFT.hpp:24-33.      */
/* NB the reference forwards extremum_refinement_iter into                     */
/* ComputeDoGExtrema's img_padding_sz slot (SIFT.cpp:45-51 vs DoG.hpp:72-78):  */
/* the border padding becomes extremum_refinement_iter and the refinement      */
/* always runs 5 iterations.  sara_hip_sift_create() reproduces that;          */
/* sara_hip_sift_create_dog() takes the ComputeDoGExtrema constructor          */
/* arguments unshifted.                                                        */
typedef struct sara_sift_params
{
  sara_pyramid_params pyramid;
  float gauss_truncate;             /* default 4 (only used when first octave > 0) */
  float extremum_thres;             /* default 0.01 */
  float edge_ratio_thres;           /* default 10 */
  int32_t extremum_refinement_iter; /* default 5 */
} sara_sift_params;

/* OERegion — Features/Feature.hpp:155-177, byte-compatible with the reference */
/* class (and with its HDF5 compound type, Features/IO.hpp:58-73): 48 bytes,   */
/* shape_matrix column-major and 16-byte aligned.                              */
typedef struct sara_oeregion
{
  float coords[2];       /* offset 0  */
  float _pad0[2];
  float shape_matrix[4]; /* offset 16, column-major 2x2 */
  float orientation;     /* offset 32 */
  float extremum_value;  /* offset 36 */
  uint8_t type;          /* offset 40, OERegion::Type, 11 = Undefined */
  int8_t extremum_type;  /* offset 41, -1 Min, 1 Max */
  uint8_t _pad1[6];
} sara_oeregion;

/* Stages, in pipeline order; detect() runs every stage up to `last_stage`.    */
typedef enum sara_hip_stage
{
  SARA_HIP_STAGE_PYRAMID = 1,     /* gaussian_pyramid + DoG pyramid           */
  SARA_HIP_STAGE_EXTREMA = 2,     /* + local_scale_space_extrema (refined)    */
  SARA_HIP_STAGE_GRADIENT = 3,    /* + gradient_polar_coordinates             */
  SARA_HIP_STAGE_ORIENTATION = 4, /* + ComputeDominantOrientations            */
  SARA_HIP_STAGE_DESCRIPTOR = 5   /* + ComputeSIFTDescriptor (full SIFT)      */
} sara_hip_stage;

/* Indices into the array filled by sara_hip_sift_stage_times().               */
enum
{
  SARA_HIP_TIME_UPLOAD = 0,
  SARA_HIP_TIME_PYRAMID = 1, /* Gaussian pyramid + fused DoG                  */
  SARA_HIP_TIME_EXTREMA = 2,
  SARA_HIP_TIME_GRADIENT = 3,
  SARA_HIP_TIME_ORIENTATION = 4,
  SARA_HIP_TIME_DESCRIPTOR = 5,
  SARA_HIP_TIME_TOTAL = 6,
  SARA_HIP_TIME_COUNT = 7
};

typedef struct sara_hip_sift sara_hip_sift; /* opaque context */

/* Message of the last failing call on this thread (never NULL). */
SARA_HIP_API const char* sara_hip_last_error(void);

/* Library/ABI version: major*10000 + minor*100 + patch. */
SARA_HIP_API int sara_hip_version(void);

/* Number of visible HIP devices (0 when there is no GPU). */
SARA_HIP_API int sara_hip_device_count(void);

/* ImagePyramidParams() defaults (ImagePyramid.hpp:33-40) and the defaults of  */
/* compute_sift_keypoints (SIFT.hpp:27-33).                                    */
SARA_HIP_API void sara_hip_default_pyramid_params(sara_pyramid_params* p);
SARA_HIP_API void sara_hip_default_sift_params(sara_sift_params* p);

/* Number of octaves gaussian_pyramid() builds for a w x h image               */
/* (GaussianPyramid.hpp:80-94) and the size/scaling factor of octave o.        */
/* Host-only arithmetic: usable without a GPU.                                 */
SARA_HIP_API int sara_hip_pyramid_octave_count(const sara_pyramid_params* p,
                                              int width, int height);
SARA_HIP_API sara_hip_status sara_hip_pyramid_octave_info(
    const sara_pyramid_params* p, int width, int height, int octave,
    int* octave_width, int* octave_height, float* octave_scaling_factor);

/* make_gaussian_kernel (LinearFiltering.hpp:171-203): writes the taps, returns */
/* the tap count (or -needed when capacity is too small).  Host-only.          */
SARA_HIP_API int sara_hip_make_gaussian_kernel(float sigma, float gauss_truncate,
                                              float* taps, int capacity);

/* The same under a named arithmetic for the two operations of that function    */
/* which are not single IEEE operations - Eigen's array exp() and sum()          */
/* (LinearFiltering.hpp:196-200).  Which one the reference evaluates depends on  */
/* how it was built; tests/golden/sensitivity.json holds what the choice does    */
/* to the keypoints (about 3 in 10 000 change).             */
enum
{
  SARA_HIP_TAPS_LIBM_SERIAL = 0, /* expf() per tap, left-to-right sum: a scalar  */
                                 /* (EIGEN_DONT_VECTORIZE / non-SSE) build; the   */
                                 /* default of every context                      */
  SARA_HIP_TAPS_EIGEN34_SSE2 = 1,/* Eigen 3.4, x86-64 baseline (the reference's   */
                                 /* Release flags): the first 4*(n/4) taps through*/
                                 /* pexp<Packet4f> (Cephes reduction, degree-5    */
                                 /* polynomial, no FMA), the n%4 trailing ones    */
                                 /* through expf; sum() = two Packet4f            */
                                 /* accumulators, (a0+a2)+(a1+a3), scalar tail    */
  SARA_HIP_TAPS_EIGEN33_SSE2 = 2 /* the same with Eigen 3.3's Horner-form pexp    */
};
/* Values of SARA_HIP_OPT_KERNEL_SELECTION. */
enum
{
  SARA_HIP_SELECT_ENVIRONMENT = 0, /* what the SARA_HIP_* variables of the       */
                                   /* process ask for;    */
                                   /* none set = SHIPPED.  Default of a context  */
  SARA_HIP_SELECT_SHIPPED = 1,     /* the production thresholds, whatever the    */
                                   /* environment says                           */
  SARA_HIP_SELECT_FORCED_MARCH = 2,/* marching kernels + 8-strip groups at every */
                                   /* launch size (small test images then run    */
                                   /* the kernels big batches run)               */
  SARA_HIP_SELECT_TILED = 3,       /* LDS-tiled blur, pixel-parallel gradient /  */
                                   /* scan everywhere                            */
  SARA_HIP_SELECT_TILED_BLUR = 4   /* LDS-tiled blur only                        */
};
SARA_HIP_API int sara_hip_make_gaussian_kernel_with(int arithmetic, float sigma,
                                                   float gauss_truncate,
                                                   float* taps, int capacity);

/* -------------------------------------------------------------------------- */
/* Whole-pipeline context == compute_sift_keypoints over a batch of frames.    */
/* All device buffers are owned by the context and sized at creation.          */
/*   max_width/max_height : largest input frame                                */
/*   max_batch            : frames per detect() call                           */
/*   max_keypoints        : per-frame capacity of the extremum and keypoint    */
/*                          lists (0 -> max_width*max_height/128)              */
/*   device               : HIP device ordinal                                 */
/* -------------------------------------------------------------------------- */
SARA_HIP_API sara_hip_status sara_hip_sift_create(const sara_sift_params* params,
                                                 int max_width, int max_height,
                                                 int max_batch,
                                                 int max_keypoints, int device,
                                                 sara_hip_sift** out);

/* Same, with ComputeDoGExtrema's constructor arguments (DoG.hpp:72-78):       */
/* img_padding_sz and extremum_refinement_iter given separately.               */
SARA_HIP_API sara_hip_status sara_hip_sift_create_dog(
    const sara_pyramid_params* pyramid, float gauss_truncate,
    float extremum_thres, float edge_ratio_thres, int img_padding_sz,
    int extremum_refinement_iter, int max_width, int max_height, int max_batch,
    int max_keypoints, int device, sara_hip_sift** out);

SARA_HIP_API sara_hip_status sara_hip_sift_destroy(sara_hip_sift* ctx);

/* Runs the pipeline on `batch` frames of width x height float32 gray pixels,  */
/* row-major (pixel (x,y) at y*width+x, Core/Image/Image.hpp:99-108), frame b  */
/* at images + b*frame_stride (in floats; 0 -> width*height).  `images` is a   */
/* host pointer, or a device pointer when images_on_device != 0.  The input is */
/* borrowed and never modified.  Work is enqueued on `hip_stream` (a           */
/* hipStream_t, NULL = the context's own stream) and NOT waited for: results   */
/* stay in HBM until fetched.  Keypoints of frame b come out in the            */
/* reference's order: octave, scale, raster (y*w+x), ascending orientation bin */
/* (DoG.cpp:70-82, RefineExtremum.cpp:497-515, Orientation.cpp:146-161).       */
SARA_HIP_API sara_hip_status sara_hip_sift_detect(
    sara_hip_sift* ctx, const float* images, size_t frame_stride, int batch,
    int width, int height, int images_on_device, sara_hip_stage last_stage,
    void* hip_stream);

/* Same as sara_hip_sift_detect() for 8-bit frames, converted to gray32f on the  */
/* device: channels = 3 for interleaved RGB8     */
/* (from_rgb8_to_gray32f, ImageProcessing/FastColorConversion.cpp:42-66: /255.0 */
/* and 0.2125 R + 0.7154 G + 0.0721 B in double, cast to float), channels = 1   */
/* for gray8 (v / 255.f).  frame_stride is in BYTES (0 -> width*height*channels).*/
/* 3/4 (RGB8) or 1/4 (gray8) of the PCIe bytes of a float frame.                */
SARA_HIP_API sara_hip_status sara_hip_sift_detect_u8(
    sara_hip_sift* ctx, const uint8_t* images, size_t frame_stride, int channels,
    int batch, int width, int height, int images_on_device,
    sara_hip_stage last_stage, void* hip_stream);

/* Double-buffered upload (the "upload path" of SURVEY.md section 8f, row f1):   */
/* stage() copies the NEXT batch of host frames into one of two staging buffers */
/* on a copy stream and returns at once; detect_staged() runs the pipeline on   */
/* the batch staged last.  Calling stage(i+1) right after detect_staged(i) lets */
/* the PCIe transfer run under the kernels of batch i, so a stream of host      */
/* frames is bounded by max(compute, upload) instead of their sum.              */
/* channels: 0 = float frames (frame_stride in floats), 1 = gray8, 3 = RGB8     */
/* (frame_stride in bytes; converted on the device as sara_hip_sift_detect_u8). */
/* `images` must stay valid (ideally pinned) until detect_staged() returns.     */
SARA_HIP_API sara_hip_status sara_hip_sift_stage(sara_hip_sift* ctx,
                                                const void* images,
                                                size_t frame_stride, int channels,
                                                int batch, int width, int height);
SARA_HIP_API sara_hip_status sara_hip_sift_detect_staged(
    sara_hip_sift* ctx, sara_hip_stage last_stage, void* hip_stream);

/* Waits for the last detect() on this context. */
SARA_HIP_API sara_hip_status sara_hip_sift_synchronize(sara_hip_sift* ctx);

/* Per-frame keypoint counts of the last detect() (host array of `batch`       */
/* ints) and their sum.  Synchronises.  When a frame overflowed max_keypoints  */
/* the call fails with SARA_HIP_CAPACITY_EXCEEDED.                             */
SARA_HIP_API sara_hip_status sara_hip_sift_counts(sara_hip_sift* ctx,
                                                 int* per_frame_counts,
                                                 int* total);

/* Keypoint-list capacity.  The reference has none: it reserves 10000 extrema  */
/* per scale and push_backs beyond (FeatureDetectors/RefineExtremum.cpp:        */
/* 496-514), so compute_sift_keypoints() returns whatever an image produces.    */
/* The context's lists hold max_keypoints entries per frame (4 * max_keypoints */
/* classified sites); a frame that needs more is reported - never silently      */
/* truncated - as SARA_HIP_CAPACITY_EXCEEDED by counts() / collect() /          */
/* extrema_counts(), and the caller grows the lists and runs the frame again:   */
/*   capacity(): *max_keypoints = the current per-frame capacity, *required =   */
/*     the capacity the last batch whose counts were read asked for (largest    */
/*     per-frame max(extrema, keypoints, sites / 4)).  A list that overflowed   */
/*     starves the lists behind it, so after an overflow `required` is a lower  */
/*     bound: reserve more (the shims take 2 x) and loop.                       */
/*   reserve(): grows the per-frame capacity to max_keypoints (never shrinks;   */
/*     a no-op when it is already that large), waits for the context's work,    */
/*     re-allocates the list and result arrays in HBM, forgets the captured     */
/*     graph and the last result.  SARA_HIP_NOT_READY while a ticket is         */
/*     pending; SARA_HIP_CAPACITY_EXCEEDED when HBM cannot hold the request     */
/*     (the old capacity stays in place).                                       */
/* This is what DO::Sara::compute_sift_keypoints / ComputeDoGExtrema of         */
/* include/DO/Sara/HipSift.hpp and sara_amd do on their callers' behalf; the    */
/* grown context stays cached, so a video pays for the growth once.             */
SARA_HIP_API sara_hip_status sara_hip_sift_capacity(const sara_hip_sift* ctx,
                                                   int* max_keypoints,
                                                   int* required);
SARA_HIP_API sara_hip_status sara_hip_sift_reserve(sara_hip_sift* ctx,
                                                  int max_keypoints);

/* Copies the keypoints of the whole batch, frames concatenated in order:      */
/*   features     : total x sara_oeregion (rescaled by the octave factor,      */
/*                  SIFT.cpp:92-98)                                            */
/*   descriptors  : total x 128 float, row-major (FeatureDescriptors/          */
/*                  SIFT.hpp:166-200), NULL to skip                            */
/*   scale_octave : total x 2 int32 (s, o) (DoG.cpp:75-81), NULL to skip       */
/* Destinations are host pointers, or device pointers when dst_on_device != 0  */
/* (device copies are enqueued on the detect stream and not waited for).       */
SARA_HIP_API sara_hip_status sara_hip_sift_fetch(sara_hip_sift* ctx,
                                                sara_oeregion* features,
                                                float* descriptors,
                                                int32_t* scale_octave,
                                                int dst_on_device);

/* Device-resident results of the last detect(): pointers into the context's   */
/* own buffers (valid until the next detect()/destroy()), for zero-copy hand   */
/* off to a collective.  Frame b occupies [frame_offsets[b], frame_offsets[b+1]) */
/* rows; frame_offsets is a device array of batch+1 int32.                     */
SARA_HIP_API sara_hip_status sara_hip_sift_device_results(
    sara_hip_sift* ctx, const sara_oeregion** features,
    const float** descriptors, const int32_t** scale_octave,
    const int32_t** frame_offsets);

/* ---- ComputeDoGExtrema accessors (DoG.hpp:115-165), device -> host -------- */

/* Octaves built by the last detect(). */
SARA_HIP_API int sara_hip_sift_octave_count(const sara_hip_sift* ctx);
SARA_HIP_API sara_hip_status sara_hip_sift_octave_info(const sara_hip_sift* ctx,
                                                      int octave, int* width,
                                                      int* height,
                                                      float* scaling_factor);

/* gaussians()(s, o), diff_of_gaussians()(s, o) of frame `frame`: w_o*h_o      */
/* floats.  gradient: (2*|grad|, atan2) interleaved, 2*w_o*h_o floats          */
/* (Orientation.cpp:24-56); only scales 1..S-3 are materialised unless the     */
/* context was told to keep all (sara_hip_sift_set_option).                    */
SARA_HIP_API sara_hip_status sara_hip_sift_copy_gaussian(sara_hip_sift* ctx,
                                                        int frame, int s, int o,
                                                        float* dst);
SARA_HIP_API sara_hip_status sara_hip_sift_copy_dog(sara_hip_sift* ctx, int frame,
                                                   int s, int o, float* dst);
SARA_HIP_API sara_hip_status sara_hip_sift_copy_gradient(sara_hip_sift* ctx,
                                                        int frame, int s, int o,
                                                        float* dst);

/* Scale-space extrema before orientation assignment == the concatenation of   */
/* ComputeDoGExtrema::extrema(s, o) in (o, s, raster) order                    */
/* (DoG.cpp:62-82).  xyso_type: n x 5 int32 (x, y, s, o, type +1/-1), the      */
/* integer detection site.  Either output may be NULL.                         */
SARA_HIP_API sara_hip_status sara_hip_sift_extrema_counts(sara_hip_sift* ctx,
                                                         int* per_frame_counts,
                                                         int* total);
SARA_HIP_API sara_hip_status sara_hip_sift_fetch_extrema(sara_hip_sift* ctx,
                                                        sara_oeregion* regions,
                                                        int32_t* xyso_type);

/* Device time of each stage of the last detect() in milliseconds (hipEvent),  */
/* indexed by SARA_HIP_TIME_*.  Synchronises.  Batches of up to 16 frames (up  */
/* to 8 when they hold more pixels than 16 x 1080p) on the context's own        */
/* stream replay a captured HIP graph (launch-bound regime); then               */
/* only SARA_HIP_TIME_TOTAL is measured and the per-stage entries are 0         */
/* (environment SARA_HIP_GRAPH=0 restores plain launches and stage times).      */
SARA_HIP_API sara_hip_status sara_hip_sift_stage_times(sara_hip_sift* ctx,
                                                      float* ms);

/* Per-launch device times of the Gaussian-pyramid stage of the last detect()   */
/* (SARA_HIP_OPT_LAUNCH_TIMERS): which plane (octave, scale; scale 0 = the base  */
/* blur of octave 0 or a separate octave hand-over), the tap count of the blur   */
/* (0 for a hand-over), the pixels it wrote (batch included) and its duration.   */
/* With SARA_HIP_OPT_SINGLE_STREAM the launches do not overlap and the times     */
/* are kernel durations; otherwise launches of different octaves overlap.        */
/* *count = launches recorded (may exceed capacity).  Synchronises.              */
typedef struct sara_hip_launch_time
{
  int32_t octave, scale, taps;
  int32_t _pad;
  int64_t pixels;
  float ms;
  float _pad2;
} sara_hip_launch_time;
SARA_HIP_API sara_hip_status sara_hip_sift_pyramid_launches(
    sara_hip_sift* ctx, sara_hip_launch_time* out, int capacity, int* count);

/* Options. */
enum
{
  SARA_HIP_OPT_ALL_GRADIENT_SCALES = 1, /* 1: polar gradients of all S scales  */
                                        /* as the reference does (default 0:   */
                                        /* only the consumed ones)             */
  SARA_HIP_OPT_STAGE_TIMERS = 2,        /* 1: record per-stage hipEvents        */
  SARA_HIP_OPT_ROOT_SIFT = 3,           /* 1: descriptors leave the descriptor  */
                                        /* kernel as RootSIFT (see              */
                                        /* sara_hip_root_sift; default 0)       */
  /* The two switches of the "corrected" detector mode (default 0 = what the    */
  /* reference's default build computes):                                       */
  SARA_HIP_OPT_SIGNED_EXTREMUM_TYPE = 4,      /* 1: host loop of the             */
                                        /* DO_SARA_USE_HALIDE branch            */
                                        /* (RefineExtremum.cpp:226-361): int8    */
                                        /* map, so DoG minima are refined like  */
                                        /* maxima, and sites whose refined      */
                                        /* scale leaves (sigma(s)/4, 4 sigma(s))*/
                                        /* are rejected (:307-325)              */
  SARA_HIP_OPT_DOWNSCALE_AT_DOUBLE_SIGMA = 5, /* 1: octave o+1 is sub-sampled    */
                                        /* from scale round(log 2 / log k), the */
                                        /* one at 2 sigma_0, instead of floor() */
                                        /* (GaussianPyramid.hpp:97-100), which   */
                                        /* is one lower for k = float(2^(1/3))  */
  SARA_HIP_OPT_SINGLE_STREAM = 7,       /* 1: every launch on one stream (the     */
                                        /* per-octave streams of the pyramid off): */
                                        /* kernel durations then add up to the     */
                                        /* stage times (measurement aid)           */
  SARA_HIP_OPT_LAUNCH_TIMERS = 8,       /* 1: a hipEvent pair around every launch  */
                                        /* of the Gaussian-pyramid stage, read by  */
                                        /* sara_hip_sift_pyramid_launches();       */
                                        /* ignored under HIP-graph replay          */
  SARA_HIP_OPT_GRAPH_REPLAY = 13,       /* 0: small batches run plain launches   */
                                        /* instead of replaying a captured HIP   */
                                        /* graph (default 1; the process-wide    */
                                        /* switch is SARA_HIP_GRAPH=0)           */
  SARA_HIP_OPT_KERNEL_SELECTION = 10,   /* SARA_HIP_SELECT_*: which kernels the  */
                                        /* context's launches take.  Results do */
                                        /* not depend on it (every selection is */
                                        /* bit-identical: tests); speed does.   */
                                        /* Setting it also puts _TILE_GEOMETRY  */
                                        /* and _MARCH(2)_WAVES back to their    */
                                        /* defaults: set those after it         */
  SARA_HIP_OPT_TILE_GEOMETRY = 11,      /* LDS tile of the tiled blur kernel:    */
                                        /* 0 by tile count (default), 1 = 64x32 */
                                        /* / 512 threads, 2 = 64x16 / 256,       */
                                        /* 3 = 32x16 / 128 (BASELINE config 5's  */
                                        /* tile-size sweep, bench.py)            */
  SARA_HIP_OPT_MARCH_WAVES = 12,        /* target waves per marching blur launch */
                                        /* (0 = shipped: 4096 / 2048); the other */
                                        /* axis of that sweep                    */
  SARA_HIP_OPT_MARCH2_WAVES = 14,       /* the same for the R >= 8 blurs only    */
                                        /* (set after _MARCH_WAVES)              */
  SARA_HIP_OPT_TAP_ARITHMETIC = 9,      /* SARA_HIP_TAPS_*: the arithmetic of    */
                                        /* make_gaussian_kernel's exp() / sum()  */
                                        /* (default SARA_HIP_TAPS_LIBM_SERIAL).  */
                                        /* Kernels with unequal mirrored taps    */
                                        /* (the Eigen models give some) run on   */
                                        /* the general marching kernel           */
  SARA_HIP_OPT_FMA_BLUR = 6             /* 1: the Gaussian blurs fuse multiply   */
                                        /* and add (v_fma_f32): half the         */
                                        /* arithmetic, pyramids within 3e-7 of   */
                                        /* the range of the reference's instead  */
                                        /* of bit-identical (the reference is    */
                                        /* built without FMA).  Default 0; never */
                                        /* used by the benchmark's headline line */
};
SARA_HIP_API sara_hip_status sara_hip_sift_set_option(sara_hip_sift* ctx,
                                                     int option, int value);

/* -------------------------------------------------------------------------- */
/* Operator-level seams: the five entry points where the reference swaps in    */
/* its Halide AOT C functions under DO_SARA_USE_HALIDE.  Host pointers in and  */
/* out, synchronous, one image; they exist for unit parity and for callers     */
/* that use a single operator.  The pipeline context above is the fast path.   */
/* -------------------------------------------------------------------------- */

/* apply_gaussian_filter(src, dst, sigma, gauss_truncate)                      */
/* LinearFiltering.cpp:30-68 (seam: shakti_separable_convolution_2d_cpu :50-54) */
SARA_HIP_API sara_hip_status sara_hip_apply_gaussian_filter(
    const float* src, float* dst, int width, int height, float sigma,
    float gauss_truncate, int device);

/* scale(src, dst): nearest-neighbour resize, Resize.cpp:31-62                 */
/* (seam: shakti_scale_32f_cpu :42-43)                                         */
SARA_HIP_API sara_hip_status sara_hip_scale(const float* src, int src_width,
                                           int src_height, float* dst,
                                           int dst_width, int dst_height,
                                           int device);

/* enlarge(src, dst): bilinear, Resize.cpp:86-128 (seam: shakti_enlarge_cpu)   */
SARA_HIP_API sara_hip_status sara_hip_enlarge(const float* src, int src_width,
                                             int src_height, float* dst,
                                             int dst_width, int dst_height,
                                             int device);

/* out = a - b, GaussianPyramid.cpp:37-46 (seam: shakti_subtract_32f_cpu)      */
SARA_HIP_API sara_hip_status sara_hip_subtract(const float* a, const float* b,
                                              float* out, size_t count,
                                              int device);

/* gradient_polar_coordinates(f): (2*|grad f|, atan2(gy, gx)) interleaved,     */
/* Orientation.cpp:24-56 (seam: shakti_polar_gradient_2d_32f_cpu,              */
/* Differential.cpp:72-79)                                                     */
SARA_HIP_API sara_hip_status sara_hip_gradient_polar_coordinates(
    const float* src, int width, int height, float* mag_ori, int device);

/* Extremum map of DoG layers (a = s-1, b = s, c = s+1): +1 max, -1 min, 0,    */
/* including the 0.8*thres and edge tests.  img_padding_sz >= 1: the rules of  */
/* the default build, RefineExtremum.cpp:407-437 (sites inside the padding     */
/* only).  img_padding_sz == 0: the seam itself,                               */
/* shakti_scale_space_dog_extremum_32f_cpu (LocalExtremum.cpp:23-37) =          */
/* is_dog_extremum of Shakti/Halide/Components/DoGExtremum.hpp:59-78 on        */
/* repeat_edge inputs: every pixel, strict contrast test, Halide's hessian.    */
SARA_HIP_API sara_hip_status sara_hip_scale_space_dog_extremum_map(
    const float* a, const float* b, const float* c, int width, int height,
    float edge_ratio_thres, float extremum_thres, int img_padding_sz,
    int8_t* out, int device);

/* from_rgb8_to_gray32f(src, dst) - ImageProcessing/FastColorConversion.cpp:    */
/* 42-66 (seam: shakti_rgb8u_to_gray32f_cpu) - and the gray8 analogue.          */
SARA_HIP_API sara_hip_status sara_hip_from_rgb8_to_gray32f(const uint8_t* rgb,
                                                          float* gray, int width,
                                                          int height, int device);
SARA_HIP_API sara_hip_status sara_hip_from_gray8_to_gray32f(const uint8_t* src,
                                                           float* gray, int width,
                                                           int height,
                                                           int device);

/* ---- descriptor matching -------------------- */
/* One match of AnnMatcher::compute_matches (FeatureMatching/AnnMatcher.cpp:    */
/* 203-268; Match/Match.hpp:166-173): indices into the first (x) and second (y) */
/* key set, score = squared-distance ratio best / second best, rank, direction  */
/* (0 = SourceToTarget, 1 = TargetToSource).                                    */
typedef struct sara_match
{
  int32_t x_index;
  int32_t y_index;
  float score;
  int32_t rank;
  int32_t direction;
} sara_match;

/* ComputeRootSIFTDescriptor's post-processing (FeatureDescriptors/RootSIFT.hpp */
/* :45-53) on an n x dim row-major descriptor matrix, in place: every row is     */
/* divided by its L1 norm, then every bin replaced by the square root of its     */
/* magnitude with its sign kept (Sara's descriptor has negative bins, see        */
/* DESIGN.md section 9; all-zero rows are left untouched), so every non-zero row */
/* has unit L2 norm afterwards.  desc: host pointer, or device pointer when      */
/* on_device != 0.  The reference's two lines are written against Eigen 2 and    */
/* no longer compile, so the parity of this entry is pinned on the restatement   */
/* of their intent only.                                                         */
SARA_HIP_API sara_hip_status sara_hip_root_sift(float* desc, int n, int dim,
                                               int on_device, int device);

/* match(keys1, keys2, lowe_ratio) - SfM/Helpers/KeypointMatching.cpp:19-25 ->  */
/* AnnMatcher{keys1, keys2, ratio}.compute_matches() (FeatureMatching/          */
/* AnnMatcher.cpp:173-268): for every key of either set FLANN's knnSearch(3)     */
/* in the other set, score = squared-distance ratio best / second best          */
/* (:126-130); then                                                             */
/*   ratio^2 <= 1 : the best neighbour, if its score passes (rank 1);           */
/*   ratio^2 >  1 : (the reference's DEFAULT, sift_ratio_thres = 1.2f,          */
/*                  AnnMatcher.hpp:36-46) the adaptive radius search of         */
/*                  :133-138 - every neighbour closer than d_best * ratio^2,    */
/*                  ranks 1..K in (distance, index) order, score d_rank /       */
/*                  d_best, until one exceeds ratio^2; a best distance of       */
/*                  exactly 0 gives radius 0 and hence no match for that key    */
/*                  (kept);                                                     */
/* both directions, duplicates (x, y) removed (best score kept), sorted by      */
/* score.  The neighbour searches are exact (what the reference's FLANN         */
/* kd-trees approximate), with FLANN's squared-L2 arithmetic.  desc1: n1 x dim, */
/* desc2: n2 x dim row-major floats (host pointers, or device pointers when     */
/* on_device != 0); dim <= 128.  matches: host array of `capacity` records      */
/* (n1 + n2 always suffices for ratio <= 1; above 1 a key can have several      */
/* matches: on SARA_HIP_CAPACITY_EXCEEDED *count holds the number needed).      */
/* Empty key sets: SARA_HIP_RUNTIME_ERROR, like the reference's                 */
/* std::runtime_error (AnnMatcher.cpp:44-45).                                   */
/* Ordering: the search runs on a private non-blocking stream of the calling    */
/* thread and is NOT ordered after any other stream.  Device descriptors        */
/* (on_device != 0) must therefore be complete before the call - e.g. those of  */
/* sara_hip_sift_device_results() only after sara_hip_sift_counts() /           */
/* sara_hip_sift_synchronize() (the key counts n1 / n2 come from counts()       */
/* anyway) - and must stay unmodified until it returns.  The call itself        */
/* returns with the match list complete.                                        */
SARA_HIP_API sara_hip_status sara_hip_match_descriptors(
    const float* desc1, int n1, const float* desc2, int n2, int dim,
    float sift_ratio_thres, int on_device, sara_match* matches, int capacity,
    int* count, int device);

/* The same for a STREAM of independent pairs in one call - the consumer matches */
/* consecutive frames, pair after pair (SfM/Helpers/KeypointMatching.cpp:19-25   */
/* in the loops of SfM/BuildingBlocks): pair p is (desc1, n1) x (desc2, n2) of   */
/* pairs[p], all with the same `dim`, ratio and on_device.  For ratios <= 1      */
/* (the consumer's: match(keys1, keys2, 0.6f)) four searches are kept in flight  */
/* on private streams - the host enqueues pair p + 1.. while the device works on */
/* pair p, the small kernels of one search run beside the tile pass of another,  */
/* and nothing waits for a read-back but the pair it belongs to; ratios > 1 run  */
/* pair by pair.  Lists are byte-identical to n_pairs single calls: `matches`    */
/* receives them back to back, pair p's at [offsets[p], offsets[p + 1])          */
/* (offsets: n_pairs + 1 ints).  SARA_HIP_CAPACITY_EXCEEDED: offsets[n_pairs]    */
/* holds the total needed.  One empty key set fails the whole call               */
/* (SARA_HIP_RUNTIME_ERROR) before anything runs.                                */
typedef struct sara_match_pair
{
  const float* desc1;
  const float* desc2;
  int32_t n1, n2;
} sara_match_pair;
SARA_HIP_API sara_hip_status sara_hip_match_descriptors_batch(
    const sara_match_pair* pairs, int n_pairs, int dim, float sift_ratio_thres,
    int on_device, sara_match* matches, int capacity, int* offsets, int device);

/* AnnMatcher{keys, ratio, min_max_metric_dist_thres, pixel_dist_thres}         */
/* .compute_matches() - the self-matching constructor (AnnMatcher.hpp:42-46,    */
/* .cpp:199-215): one key set matched against itself.  Rank 0 of every search   */
/* is taken to be the key itself (:124-125), so ranks start at the second       */
/* entry, and a neighbour is dropped when KeyProximity finds the two keys too   */
/* close (FeatureMatching/KeyProximity.cpp:17-30: squared pixel distance below  */
/* pixel_dist_thres^2, or either key's shape-matrix metric below                */
/* min_max_metric_dist_thres^2).  With ratio^2 <= 1 the reference emits nothing */
/* in this mode (its loop runs over ranks [1, K = 1)) - kept.  Duplicates are   */
/* removed by Match::operator==, which compares the OERegions by value          */
/* (Match/Match.hpp:161-164).  features: host array of n sara_oeregion; desc: n */
/* x dim floats on the host, or in HBM when desc_on_device != 0.                */
SARA_HIP_API sara_hip_status sara_hip_self_match_descriptors(
    const float* desc, const sara_oeregion* features, int n, int dim,
    float sift_ratio_thres, float min_max_metric_dist_thres,
    float pixel_dist_thres, int desc_on_device, sara_match* matches, int capacity,
    int* count, int device);

/* The matcher keeps grow-only scratch buffers per (calling thread, device)     */
/* between calls (allocating them costs more than a search); this frees those   */
/* of the calling thread on `device`, if any (other devices' buffers and the     */
/* thread's current HIP device are left alone).                                  */
SARA_HIP_API sara_hip_status sara_hip_match_release_workspace(int device);

/* -------------------------------------------------------------------------- */
/* Pipelined host-to-host operation (the metric of SURVEY.md section 8d: frames */
/* in host memory -> OERegion[] + descriptors in host memory, as               */
/* compute_sift_keypoints returns them, FeatureDetectors/SIFT.cpp:27-33,107).   */
/* Up to two batches are in flight: submit(i + 1) may be called before          */
/* collect(i); the upload of batch i + 1 (copy stream), its kernels and the     */
/* read-back of batch i (a third stream) then overlap.                          */
/*   channels: 0 = float32 gray frames (frame_stride in floats), 1 = gray8,     */
/*             3 = interleaved RGB8 (frame_stride in bytes); 0 -> densely packed*/
/* Lifetime of `images` (host frames): the upload is asynchronous - with pinned  */
/* memory submit() returns before the copy engine has read the frames - so the  */
/* buffer must stay valid AND UNMODIFIED until collect() / collect_into() /     */
/* gather of that ticket has returned (decode the next frames into another      */
/* buffer; two alternating buffers suffice).  Device frames (images_on_device)  */
/* are read by the kernels of the batch: same rule.                             */
/* submit_staged() is submit() for the batch that sara_hip_sift_stage() put on   */
/* its way: stage(i + 1); collect(i - 1); submit_staged(i + 1) starts upload     */
/* i + 1 BEFORE the host waits for the read-back of batch i - 1, so that the     */
/* copy engine goes from one upload straight into the next (stage() only waits  */
/* - on the device - until the first blur of the batch that used its buffer has */
/* read the frames).  With float32 frames, whose upload is longer than the       */
/* kernels, the step is then the upload alone instead of kernels + read-back.    */
/* From HIP runtime 7.2 on stage() first waits (on the host) for the upload      */
/* staged before it: one upload at a time on the copy engines keeps an engine    */
/* free for the read-back (9.4 ms per 64 x 1080p float frames in every process;  */
/* SARA_HIP_STAGE_WAIT=0 / 1 overrides).                    */
/* collect() blocks until the batch of `ticket` is in pinned host memory owned  */
/* by the context and returns pointers into it.  features / descriptors /       */
/* scale_octave AND frame_offsets all live in the ticket's ring slot: they stay */
/* valid until the second submit() after this ticket's (the submit of ticket +  */
/* 2 reuses the slot and overwrites all four arrays, frame_offsets included).   */
/* frame_offsets has batch + 1 entries (frame b = [frame_offsets[b],            */
/* frame_offsets[b + 1])).  Pass descriptors = NULL to skip their read-back;    */
/* asking for descriptors of a ticket submitted with last_stage < DESCRIPTOR    */
/* returns SARA_HIP_NOT_READY and leaves the ticket pending (as fetch() does).  */
/* SARA_HIP_CAPACITY_EXCEEDED is reported by collect() (the truncated lists are */
/* still delivered).                                                            */
/* -------------------------------------------------------------------------- */
SARA_HIP_API sara_hip_status sara_hip_sift_submit_staged(
    sara_hip_sift* ctx, sara_hip_stage last_stage, int* ticket);
SARA_HIP_API sara_hip_status sara_hip_sift_submit(
    sara_hip_sift* ctx, const void* images, size_t frame_stride, int channels,
    int batch, int width, int height, int images_on_device,
    sara_hip_stage last_stage, int* ticket);
SARA_HIP_API sara_hip_status sara_hip_sift_collect(
    sara_hip_sift* ctx, int ticket, const sara_oeregion** features,
    const float** descriptors, const int32_t** scale_octave,
    const int32_t** frame_offsets, int* total);

/* The same read-back into memory the CALLER owns (any host memory; pinned or   */
/* sara_hip_host_register()-ed memory is copied by DMA without a bounce):       */
/* ticket_counts() waits for the batch and reports its size (frame_offsets:     */
/* batch + 1 entries copied out, may be NULL) and leaves the ticket pending;    */
/* collect_into() copies total x sara_oeregion / total x 128 float / total x 2  */
/* int32 (NULL to skip) and consumes the ticket.  This is what lets several     */
/* processes - one per GPU - deliver their shards into ONE host array (a shared */
/* mapping registered by every process) at their global offsets, each over its  */
/* own PCIe link: SURVEY.md section 8d's "results in host memory on the root"   */
/* without funnelling every byte through the root GPU's link.                   */
SARA_HIP_API sara_hip_status sara_hip_sift_ticket_counts(
    sara_hip_sift* ctx, int ticket, int32_t* frame_offsets, int* batch,
    int* total);
SARA_HIP_API sara_hip_status sara_hip_sift_collect_into(
    sara_hip_sift* ctx, int ticket, sara_oeregion* features, float* descriptors,
    int32_t* scale_octave);
/* hipHostRegister / hipHostUnregister (portable: usable from every device) for */
/* callers without a HIP toolchain.                                             */
SARA_HIP_API sara_hip_status sara_hip_host_register(void* ptr, size_t bytes);
SARA_HIP_API sara_hip_status sara_hip_host_unregister(void* ptr);
/* hipHostMalloc / hipHostFree (portable).  Frames decoded INTO memory of this   */
/* kind upload faster than from registered memory: measured on 64 x 1080p float */
/* frames per step, 8.7 ms against 10.7 ms host -> host.   */
SARA_HIP_API sara_hip_status sara_hip_host_alloc(void** ptr, size_t bytes);
SARA_HIP_API sara_hip_status sara_hip_host_free(void* ptr);

/* -------------------------------------------------------------------------- */
/* Multi-GPU.  Frames are independent                    */
/* (compute_sift_keypoints is a pure function of one image,                     */
/* FeatureDetectors/SIFT.cpp:27-108): they are sharded in contiguous blocks,    */
/* one context per GPU, and the only exchange is the gather of the              */
/* variable-length keypoint arrays to a root device over RCCL: counts through   */
/* one ncclAllGather, then one group of ncclSend / ncclRecv at the global       */
/* offsets, so that the root holds the keypoints in (frame, octave, scale, y,   */
/* x, bin) order.  librccl is loaded on first use.                              */
/* Failure handling: every rank takes part in every collective of a gather; a   */
/* rank that cannot (unknown ticket, descriptors that were not computed, bad    */
/* root) says so through the count exchange, all ranks then skip the transfers  */
/* together and return an error - nobody is left blocked, and the ticket is     */
/* released on every path.                                                      */
/* Test transport: with SARA_HIP_COMM_TRANSPORT=loopback in the environment     */
/* sara_hip_comm_unique_id() / sara_hip_sift_group_create() build communicators */
/* whose ranks are THREADS OF ONE PROCESS (on one device or several) and whose  */
/* AllGather / Send / Recv are device copies through a process-local mailbox:   */
/* the N > 1 exchange then runs on a one-GPU box.  Not a production path.       */
/* -------------------------------------------------------------------------- */

/* Contiguous block [*lo, *hi) of `n_frames` frames owned by `rank`.  Host-only. */
SARA_HIP_API void sara_hip_shard_range(int n_frames, int world_size, int rank,
                                      int* lo, int* hi);

/* Blocking copy of `bytes` from HBM of `device` to host memory: lets a caller   */
/* without a HIP toolchain (ctypes, cgo ...) read the gathered arrays.          */
SARA_HIP_API sara_hip_status sara_hip_copy_to_host(void* dst,
                                                  const void* src_device,
                                                  size_t bytes, int device);

/* The other direction, and HBM buffers, for the same kind of caller: frames    */
/* that several calls reuse (images_on_device = 1) are uploaded once.  A context */
/* replaying its HIP graph (batches <= 16) reads such frames IN PLACE; they must */
/* stay unmodified until the results of the call have been fetched / collected. */
SARA_HIP_API sara_hip_status sara_hip_copy_to_device(void* dst_device,
                                                    const void* src, size_t bytes,
                                                    int device);
SARA_HIP_API sara_hip_status sara_hip_device_alloc(void** ptr, size_t bytes,
                                                  int device);
SARA_HIP_API sara_hip_status sara_hip_device_free(void* ptr, int device);

/* --- one process per GPU --------------------------------------------------- */
#define SARA_HIP_COMM_ID_BYTES 128
typedef struct sara_hip_comm sara_hip_comm; /* one rank of a gather group */

/* Rank 0 makes the id; the caller ships its 128 bytes to the other ranks by    */
/* whatever transport it has (a file, MPI, a torch.distributed store ...).       */
SARA_HIP_API sara_hip_status sara_hip_comm_unique_id(unsigned char* id);
/* Collective over the `nranks` processes: joins rank `rank` (context `ctx`,    */
/* HIP device `device`) to the communicator of `id` (ncclCommInitRank).         */
SARA_HIP_API sara_hip_status sara_hip_comm_create(sara_hip_sift* ctx,
                                                 const unsigned char* id,
                                                 int nranks, int rank,
                                                 int device,
                                                 sara_hip_comm** out);
/* Collective: gathers the keypoints of every rank's submit() ticket on `root`  */
/* (rank order = global frame order when the shards are contiguous).  Blocks    */
/* until this rank's part of the exchange is complete; the exchange runs on its */
/* own stream, beside the kernels of a batch submitted after `ticket`.  The      */
/* ticket is consumed (do not collect() it as well).  counts_per_rank: nranks    */
/* entries, filled on every rank.  On the root the three device pointers address */
/* arrays owned by the communicator (valid until the next gather); elsewhere    */
/* they are set to NULL.                                                         */
SARA_HIP_API sara_hip_status sara_hip_comm_gather(
    sara_hip_comm* comm, int ticket, int root, int with_descriptors,
    int* counts_per_rank, const sara_oeregion** d_features,
    const float** d_descriptors, const int32_t** d_scale_octave, int* total);
SARA_HIP_API sara_hip_status sara_hip_comm_destroy(sara_hip_comm* comm);
/* "rccl" or "loopback". */
SARA_HIP_API const char* sara_hip_comm_transport(const sara_hip_comm* comm);
/* Ranks of the communicator. */
SARA_HIP_API int sara_hip_comm_size(const sara_hip_comm* comm);
/* ncclGetVersion() of the librccl the library bound (e.g. 22703 = 2.27.3), so  */
/* that a benchmark line can say which RCCL carried the gather.                 */
SARA_HIP_API sara_hip_status sara_hip_rccl_version(int* version);

/* --- one process, one host thread per GPU ---------------------------------- */
typedef struct sara_hip_sift_group sara_hip_sift_group;

/* n_dev contexts (devices[i], or 0..n_dev-1 when NULL) sized like              */
/* sara_hip_sift_create(), and their communicator (ncclCommInitAll).            */
SARA_HIP_API sara_hip_status sara_hip_sift_group_create(
    const sara_sift_params* params, int max_width, int max_height,
    int max_batch_per_device, int max_keypoints, int n_dev, const int* devices,
    sara_hip_sift_group** out);
SARA_HIP_API int sara_hip_sift_group_size(const sara_hip_sift_group* group);
SARA_HIP_API sara_hip_status sara_hip_sift_group_context(
    sara_hip_sift_group* group, int index, sara_hip_sift** ctx);
/* Device i runs sara_hip_sift_submit() on shard_images[i] (shard_batch[i]      */
/* frames; same frame_stride / channels / on-device meaning), all devices at    */
/* once.  Returns when everything is enqueued.  shard_batch[i] == 0 is allowed  */
/* (fewer frames than devices): that device contributes nothing.  A batch that  */
/* was detected but never gathered is dropped by the next group_detect().       */
SARA_HIP_API sara_hip_status sara_hip_sift_group_detect(
    sara_hip_sift_group* group, const void* const* shard_images,
    const int* shard_batch, size_t frame_stride, int channels, int width,
    int height, int images_on_device, sara_hip_stage last_stage);
/* Gathers the results of the last group_detect() on device `root` (index into  */
/* the group); outputs as for sara_hip_comm_gather on the root.  An invalid     */
/* `root` is rejected before anything changes (gather again with a valid one);  */
/* every other failure releases the batch.  The per-device counts are host      */
/* values of this one process, so no count collective runs here - only the      */
/* grouped Send / Recv.                                                         */
SARA_HIP_API sara_hip_status sara_hip_sift_group_gather(
    sara_hip_sift_group* group, int root, int with_descriptors,
    int* counts_per_device, const sara_oeregion** d_features,
    const float** d_descriptors, const int32_t** d_scale_octave, int* total);
/* SURVEY.md section 8d's ending for a whole node: instead of funnelling the    */
/* results through the root GPU (whose single PCIe link would then carry every  */
/* device's share), every device copies its shard straight into ONE pinned      */
/* (portable) host array owned by the group, at its global offset, over its own */
/* PCIe link, all devices at once.  Same order as group_gather.  The pointers    */
/* stay valid until the next collect_host() / destroy().  Consumes the batch    */
/* (call either this or group_gather, not both).                                */
SARA_HIP_API sara_hip_status sara_hip_sift_group_collect_host(
    sara_hip_sift_group* group, int with_descriptors, int* counts_per_device,
    const sara_oeregion** h_features, const float** h_descriptors,
    const int32_t** h_scale_octave, int* total);
SARA_HIP_API const char* sara_hip_sift_group_transport(
    const sara_hip_sift_group* group);
SARA_HIP_API sara_hip_status sara_hip_sift_group_destroy(sara_hip_sift_group* group);

/* Host self-check: evaluates, on the CPU, the float atan2 sequence the polar-  */
/* gradient kernels execute on the GPU (a restatement of glibc 2.35's          */
/* atan2f), so that tests can prove it bit-identical to the host libm the CPU  */
/* reference links against.  Not a compute path.                               */
SARA_HIP_API void sara_hip_selfcheck_atan2f(const float* y, const float* x,
                                           float* out, size_t count);

/* Host self-check: float(sin), float(cos) of double(theta) through the short   */
/* double-precision sequence the descriptor kernel uses for its rotation       */
/* matrix (FeatureDescriptors/SIFT.hpp:84-89 calls std::cos / std::sin on a    */
/* double), for theta in [-4, 4].  Not a compute path.                          */
SARA_HIP_API void sara_hip_selfcheck_sincos(const float* theta, float* out_sin,
                                           float* out_cos, size_t count);

/* Device self-check: the polar-gradient kernel replaces the compiler's IEEE    */
/* sqrt and one of its divisions by shorter correctly-rounded sequences and the */
/* range selection of atanf by a table look-up.  This runs, on the GPU, every   */
/* non-negative float through both forms: mismatches[0] = atanf reduction,      */
/* mismatches[1] = square root (both must be 0).  Not a compute path.           */
SARA_HIP_API sara_hip_status sara_hip_selfcheck_device_math(
    unsigned long long* mismatches, int device);

/* Device self-check: the orientation kernel computes a sample's histogram bin,  */
/* int(floor(double(ori / float(2 pi) * 36))) % 36 (Orientation.hpp:118-119), by */
/* one multiplication and at most one correction against thresholds bisected on */
/* the host from that expression.  This runs every float of [0, float(2 pi)]    */
/* through both forms on the GPU; *mismatches must be 0.  Not a compute path.    */
SARA_HIP_API sara_hip_status sara_hip_selfcheck_orientation_bins(
    unsigned long long* mismatches, int device);

/* Device self-check: the definiteness test of refine_extremum                  */
/* (RefineExtremum.cpp:74-77: (SelfAdjointEigenSolver<Matrix3f>(H).eigenvalues()*/
/* * float(type)).maxCoeff() >= 0) as the extrema kernels evaluate it - Eigen   */
/* 3.4's float solver restated, skipped when Sylvester's criterion in double on */
/* the shifted matrix already fixes the sign.  hessians = count x 9 floats      */
/* (row-major 3 x 3, symmetric), types = 1 / 255 / -1, out[i] = 0 / 1 (host     */
/* pointers).  Not a compute path.                                              */
SARA_HIP_API sara_hip_status sara_hip_selfcheck_definiteness(
    const float* hessians, const int* types, size_t count, unsigned char* out,
    int device);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* SARA_HIP_SIFT_H */
