// Helpers shared by feature_kernels.hip (the streaming kernels: polar gradients,
// extremum scan, refinement, ordering) and keypoint_kernels.hip (orientations,
// descriptors).
#pragma once
#include "sift_kernels.hpp"

#include <algorithm>

namespace sara_hip {

  //! (magnitude, angle) pairs through a pointer that is explicitly in the
  //! global address space (see orientation_kernel).
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef const f32x2 __attribute__((address_space(1))) * global_float2_ptr;
  __device__ __forceinline__ float2 load_pair(global_float2_ptr p, size_t i)
  {
    const f32x2 v = p[i];
    return make_float2(v.x, v.y);
  }
  //! The same with a 32-bit pixel index (a plane of one frame has far fewer
  //! than 2^28 pixels): wave-uniform base + 32-bit byte offset, i.e. the
  //! scalar-base form of global_load instead of 64-bit address arithmetic per
  //! lane (v_mad_i64_i32 + v_lshl_add_u64 per sample).
  __device__ __forceinline__ float2 load_pair32(global_float2_ptr p, unsigned i)
  {
    typedef const char __attribute__((address_space(1))) * global_bytes;
    __builtin_assume(i < (1u << 28));
    const unsigned off = i << 3;
    const f32x2 v = *reinterpret_cast<global_float2_ptr>(
        reinterpret_cast<global_bytes>(p) + off);
    return make_float2(v.x, v.y);
  }


  //! Consecutive workgroups (one work item each) of the per-keypoint kernels
  //! that stay on one XCD (xcd_local_block).
  constexpr int g_xcd_run = 128;
  //! Run-groups (8 * g_xcd_run blocks) in the grid of the per-keypoint kernels
  //! per frame; the blocks loop over the rest.  One run-group per frame fills
  //! the chip when there are many frames; a small batch gets as many groups as
  //! it takes to put ~8 waves on every SIMD (one 1080p frame: 4 300 keypoints
  //! on 1 024 one-wave blocks would walk 4 keypoints each, one after the other).
  static inline int persist_units(int batch, int waves_per_block)
  {
    const int per_unit = 8 * g_xcd_run * waves_per_block * std::max(batch, 1);
    return std::max(1, (8192 + per_unit - 1) / per_unit);
  }
}  // namespace sara_hip
