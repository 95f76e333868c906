// HDF5 keypoint files (include/sara_keypoint_h5.h): the compound type of
// OERegion (Features/IO.hpp:58-73) and the two datasets of a keypoint list
// (Features/IO.hpp:146-167) through libhdf5's C API.  Host code only.
#include "../../include/sara_keypoint_h5.h"

#include <hdf5.h>

#include <cstddef>
#include <string>
#include <vector>

namespace {

  thread_local std::string g_error;

  int fail(const std::string& msg)
  {
    g_error = msg;
    return 1;
  }

  // Closes an HDF5 identifier when it goes out of scope.
  struct Handle
  {
    hid_t id = -1;
    herr_t (*close)(hid_t) = nullptr;
    Handle(hid_t i, herr_t (*c)(hid_t)) : id(i), close(c) {}
    Handle(const Handle&) = delete;
    Handle& operator=(const Handle&) = delete;
    ~Handle()
    {
      if (id >= 0 && close)
        close(id);
    }
    operator hid_t() const { return id; }
    bool ok() const { return id >= 0; }
  };

  // CalculateH5Type<OERegion> (Features/IO.hpp:58-73): members in declaration
  // order at their offsets in the 48-byte record; vectors are 1-D arrays,
  // matrices 2-D arrays (Core/HDF5.hpp:124-141).
  hid_t make_oeregion_type()
  {
    const hid_t t = H5Tcreate(H5T_COMPOUND, sizeof(sara_oeregion));
    if (t < 0)
      return t;
    const hsize_t d1[1] = {2};
    const hsize_t d2[2] = {2, 2};
    Handle coords(H5Tarray_create2(H5T_NATIVE_FLOAT, 1, d1), H5Tclose);
    Handle shape(H5Tarray_create2(H5T_NATIVE_FLOAT, 2, d2), H5Tclose);
    bool ok = coords.ok() && shape.ok();
    ok = ok && H5Tinsert(t, "coords", offsetof(sara_oeregion, coords), coords) >= 0;
    ok = ok && H5Tinsert(t, "shape_matrix", offsetof(sara_oeregion, shape_matrix),
                         shape) >= 0;
    ok = ok && H5Tinsert(t, "orientation", offsetof(sara_oeregion, orientation),
                         H5T_NATIVE_FLOAT) >= 0;
    ok = ok && H5Tinsert(t, "extremum_value",
                         offsetof(sara_oeregion, extremum_value),
                         H5T_NATIVE_FLOAT) >= 0;
    ok = ok && H5Tinsert(t, "type", offsetof(sara_oeregion, type),
                         H5T_NATIVE_UINT8) >= 0;
    ok = ok && H5Tinsert(t, "extremum_type", offsetof(sara_oeregion, extremum_type),
                         H5T_NATIVE_INT8) >= 0;
    if (!ok)
    {
      H5Tclose(t);
      return -1;
    }
    return t;
  }

  // H5File::write_dataset (Core/HDF5.hpp:250-279).
  int write_dataset(hid_t file, const std::string& name, hid_t type, int rank,
                    const hsize_t* dims, const void* data, bool overwrite)
  {
    const bool exists = H5Lexists(file, name.c_str(), H5P_DEFAULT) > 0;
    if (exists && overwrite)
      if (H5Ldelete(file, name.c_str(), H5P_DEFAULT) < 0)
        return fail("Error: could not delete dataset: " + name);
    if (exists && !overwrite)
      return fail("Error: dataset \"" + name +
                  "\" exists but overwriting is not permitted!");
    Handle space(H5Screate_simple(rank, dims, nullptr), H5Sclose);
    if (!space.ok())
      return fail("could not create the data space of " + name);
    Handle ds(H5Dcreate2(file, name.c_str(), type, space, H5P_DEFAULT, H5P_DEFAULT,
                         H5P_DEFAULT),
              H5Dclose);
    if (!ds.ok())
      return fail("could not create dataset " + name);
    bool empty = false;
    for (int i = 0; i < rank; ++i)
      empty = empty || dims[i] == 0;
    if (!empty && H5Dwrite(ds, type, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0)
      return fail("could not write dataset " + name);
    return 0;
  }

  int dataset_dims(hid_t file, const std::string& name, int rank, hsize_t* dims)
  {
    Handle ds(H5Dopen2(file, name.c_str(), H5P_DEFAULT), H5Dclose);
    if (!ds.ok())
      return fail("could not open dataset " + name);
    Handle space(H5Dget_space(ds), H5Sclose);
    if (!space.ok() || H5Sget_simple_extent_ndims(space) != rank)
      return fail("dataset " + name + " does not have rank " + std::to_string(rank));
    H5Sget_simple_extent_dims(space, dims, nullptr);
    return 0;
  }

  int read_dataset(hid_t file, const std::string& name, hid_t type, void* data)
  {
    Handle ds(H5Dopen2(file, name.c_str(), H5P_DEFAULT), H5Dclose);
    if (!ds.ok())
      return fail("could not open dataset " + name);
    Handle space(H5Dget_space(ds), H5Sclose);
    if (space.ok() && H5Sget_simple_extent_npoints(space) == 0)
      return 0;
    if (H5Dread(ds, type, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0)
      return fail("could not read dataset " + name);
    return 0;
  }

  // Failures are reported through the return value and sara_h5_last_error();
  // HDF5's own error-stack printing is switched off for the duration of a
  // call and the caller's handler restored afterwards.
  struct QuietErrors
  {
    H5E_auto2_t func = nullptr;
    void* data = nullptr;
    QuietErrors()
    {
      H5Eget_auto2(H5E_DEFAULT, &func, &data);
      H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);
    }
    ~QuietErrors() { H5Eset_auto2(H5E_DEFAULT, func, data); }
    QuietErrors(const QuietErrors&) = delete;
    QuietErrors& operator=(const QuietErrors&) = delete;
  };

}  // namespace

extern "C" {

int sara_h5_write_keypoints(const char* path, int truncate, const char* group,
                            const sara_oeregion* features, int n,
                            const float* descriptors, int dim, int overwrite)
{
  QuietErrors quiet;
  if (!path || !group || n < 0 || dim < 0 || (n > 0 && (!features || !descriptors)))
    return fail("null pointer or negative size");
  hid_t fid = -1;
  if (truncate)
    fid = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
  else
  {
    fid = H5Fopen(path, H5F_ACC_RDWR, H5P_DEFAULT);
    if (fid < 0)
      fid = H5Fcreate(path, H5F_ACC_EXCL, H5P_DEFAULT, H5P_DEFAULT);
  }
  Handle file(fid, H5Fclose);
  if (!file.ok())
    return fail(std::string("could not open ") + path);
  // H5File::get_group (Core/HDF5.hpp:186-193); intermediate groups as well.
  const std::string g(group);
  if (!g.empty() && g != "/")
  {
    Handle lcpl(H5Pcreate(H5P_LINK_CREATE), H5Pclose);
    H5Pset_create_intermediate_group(lcpl, 1);
    if (H5Lexists(file, g.c_str(), H5P_DEFAULT) <= 0)
    {
      Handle grp(H5Gcreate2(file, g.c_str(), lcpl, H5P_DEFAULT, H5P_DEFAULT),
                 H5Gclose);
      if (!grp.ok())
        return fail("could not create group " + g);
    }
  }
  Handle type(make_oeregion_type(), H5Tclose);
  if (!type.ok())
    return fail("could not build the OERegion compound type");
  const hsize_t fd[1] = {hsize_t(n)};
  if (write_dataset(file, g + "/features", type, 1, fd, features, overwrite != 0))
    return 1;
  const hsize_t dd[2] = {hsize_t(n), hsize_t(dim)};
  return write_dataset(file, g + "/descriptors", H5T_NATIVE_FLOAT, 2, dd,
                       descriptors, overwrite != 0);
}

int sara_h5_keypoints_sizes(const char* path, const char* group, int* n, int* dim)
{
  QuietErrors quiet;
  if (!path || !group || !n || !dim)
    return fail("null pointer");
  Handle file(H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT), H5Fclose);
  if (!file.ok())
    return fail(std::string("could not open ") + path);
  const std::string g(group);
  hsize_t fd[1] = {0}, dd[2] = {0, 0};
  if (dataset_dims(file, g + "/features", 1, fd) ||
      dataset_dims(file, g + "/descriptors", 2, dd))
    return 1;
  if (fd[0] != dd[0])
    return fail("features and descriptors of " + g + " differ in count");
  *n = int(fd[0]);
  *dim = int(dd[1]);
  return 0;
}

int sara_h5_read_keypoints(const char* path, const char* group,
                           sara_oeregion* features, float* descriptors)
{
  QuietErrors quiet;
  if (!path || !group)
    return fail("null pointer");
  Handle file(H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT), H5Fclose);
  if (!file.ok())
    return fail(std::string("could not open ") + path);
  const std::string g(group);
  Handle type(make_oeregion_type(), H5Tclose);
  if (!type.ok())
    return fail("could not build the OERegion compound type");
  if (read_dataset(file, g + "/features", type, features))
    return 1;
  return read_dataset(file, g + "/descriptors", H5T_NATIVE_FLOAT, descriptors);
}

const char* sara_h5_last_error(void) { return g_error.c_str(); }

}  // extern "C"
