/* HDF5 keypoint files of Sara as a small C-ABI
 * over libhdf5's C API.  Replaces, for this path, the H5File wrapper and the
 * two functions of the reference:
 *   read_keypoints(H5File&, group)                    Features/IO.hpp:146-157
 *   write_keypoints(H5File&, group, keys, overwrite)  Features/IO.hpp:159-167
 * with the compound type of OERegion declared in Features/IO.hpp:58-73 and
 * the dataset conventions of Core/HDF5.hpp:250-279 (write) / :498-545 (read).
 *
 * A separate library (libsara_keypoint_h5.so) so that the SIFT library itself
 * does not depend on HDF5.  Host code only.
 */
#ifndef SARA_KEYPOINT_H5_H
#define SARA_KEYPOINT_H5_H

#include "sara_hip_sift.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Writes datasets "<group>/features" (rank 1, n records of the compound type   */
/* {coords float[2] @0, shape_matrix float[2][2] @16, orientation float @32,    */
/* extremum_value float @36, type uint8 @40, extremum_type int8 @41}, 48 bytes: */
/* the memory layout of OERegion / sara_oeregion) and "<group>/descriptors"     */
/* (rank 2, n x dim float).  truncate != 0 creates the file anew (H5F_ACC_TRUNC */
/* as in the reference's tests), otherwise it is opened read-write (created if  */
/* missing).  The group is created when absent (H5File::get_group).  An         */
/* existing dataset is replaced when overwrite != 0, otherwise the call fails   */
/* with the reference's message (Core/HDF5.hpp:266-268).  0 = ok.               */
SARA_HIP_API int sara_h5_write_keypoints(const char* path, int truncate,
                                         const char* group,
                                         const sara_oeregion* features, int n,
                                         const float* descriptors, int dim,
                                         int overwrite);

/* Sizes of the two datasets of a group (n records; descriptors n x dim).       */
SARA_HIP_API int sara_h5_keypoints_sizes(const char* path, const char* group,
                                         int* n, int* dim);

/* Reads both datasets into caller-allocated arrays of the sizes above.         */
SARA_HIP_API int sara_h5_read_keypoints(const char* path, const char* group,
                                        sara_oeregion* features,
                                        float* descriptors);

/* Message of the last failed call on this thread.                              */
SARA_HIP_API const char* sara_h5_last_error(void);

#ifdef __cplusplus
}
#endif

#endif
