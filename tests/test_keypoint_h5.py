"""HDF5 keypoint files (include/sara_keypoint_h5.h)
- host code, runs without a GPU.  The reference's own test
(test_features_hdf5.cpp:27-73) restated on the Python mirror and on the C++
shim, plus the layout of the file as h5dump sees it: the compound type
Features/IO.hpp:58-73 declares."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import sara_amd

HERE = os.path.dirname(os.path.abspath(__file__))
H5_HEADER = "/opt/conda/include/hdf5.h"

pytestmark = pytest.mark.skipif(not os.path.exists(H5_HEADER),
                                reason="libhdf5 C headers are not in this image")


def dummy_keys(n=4, dim=128):
    reg = np.zeros(n, sara_amd.OEREGION_DTYPE)
    for i in range(n):
        reg["coords"][i] = (i, i)
        reg["shape_matrix"][i] = (i + 0.5,) * 4
        reg["orientation"][i] = 30.0 * i
        reg["extremum_value"][i] = 10.0 * i
        reg["type"][i] = 11
        reg["extremum_type"][i] = 1 if i % 2 else -1
    desc = np.arange(n * dim, dtype=np.float32).reshape(n, dim)
    return sara_amd.KeypointList(reg, desc)


def test_library_exports_the_declared_symbols():
    import ctypes as C
    import re
    import __graft_entry__
    from sara_amd import capi
    path = os.path.join(os.path.dirname(capi.LIB_PATH), "libsara_keypoint_h5.so")
    if not os.path.exists(path):
        __graft_entry__.build()
    lib = C.CDLL(path)
    hdr = open(os.path.join(HERE, "..", "include", "sara_keypoint_h5.h")).read()
    names = re.findall(r"SARA_HIP_API[^;(]*?\b(sara_h5_\w+)\s*\(", hdr)
    assert len(names) == 4
    for n in names:
        assert hasattr(lib, n), n


def test_reference_round_trip(tmp_path):
    keys = dummy_keys()
    path = str(tmp_path / "sfm_dummy_data.h5")
    f = sara_amd.H5File(path, "w")
    sara_amd.write_keypoints(f, "0", keys)
    back = sara_amd.read_keypoints(sara_amd.H5File(path, "r"), "0")
    assert back.regions.tobytes() == keys.regions.tobytes()
    assert np.array_equal(back.descriptor_matrix, keys.descriptor_matrix)
    for i in range(4):
        assert tuple(back.regions["coords"][i]) == (i, i)
        assert np.all(back.regions["shape_matrix"][i] == i + 0.5)
        assert back.regions["orientation"][i] == 30 * i
        assert back.regions["extremum_value"][i] == 10 * i


def test_overwrite_rule_groups_and_empty_lists(tmp_path):
    keys = dummy_keys()
    path = str(tmp_path / "k.h5")
    f = sara_amd.H5File(path, "w")
    sara_amd.write_keypoints(f, "0", keys)
    with pytest.raises(RuntimeError, match="exists but overwriting is not permitted"):
        sara_amd.write_keypoints(f, "0", keys)          # Core/HDF5.hpp:266-268
    small = sara_amd.KeypointList(keys.regions[:2], keys.descriptor_matrix[:2])
    sara_amd.write_keypoints(f, "0", small, overwrite=True)
    assert len(sara_amd.read_keypoints(f, "0")) == 2
    # more groups in the same file ("a": read-write), nested names, empty list
    g = sara_amd.H5File(path, "a")
    sara_amd.write_keypoints(g, "frames/17", keys)
    empty = sara_amd.KeypointList(keys.regions[:0], keys.descriptor_matrix[:0])
    sara_amd.write_keypoints(g, "frames/18", empty)
    assert len(sara_amd.read_keypoints(g, "0")) == 2
    assert len(sara_amd.read_keypoints(g, "frames/17")) == 4
    e = sara_amd.read_keypoints(g, "frames/18")
    assert len(e) == 0 and e.descriptor_matrix.shape == (0, 128)
    with pytest.raises(RuntimeError):
        sara_amd.read_keypoints(g, "nope")
    with pytest.raises(RuntimeError):
        sara_amd.write_keypoints(sara_amd.H5File(path, "r"), "1", keys)


@pytest.mark.skipif(shutil.which("h5dump") is None
                    and not os.path.exists("/opt/conda/bin/h5dump"),
                    reason="h5dump not available")
def test_file_layout_is_the_reference_compound(tmp_path):
    """Features/IO.hpp:58-73 + Core/HDF5.hpp:124-141: member names, order,
    array shapes and scalar types of the stored OERegion; descriptors N x dim."""
    path = str(tmp_path / "k.h5")
    sara_amd.write_keypoints(sara_amd.H5File(path, "w"), "0", dummy_keys())
    exe = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    out = subprocess.run([exe, "-H", path], capture_output=True, text=True).stdout
    flat = " ".join(out.split())
    assert ('DATATYPE H5T_COMPOUND { H5T_ARRAY { [2] H5T_IEEE_F32LE } "coords"; '
            'H5T_ARRAY { [2][2] H5T_IEEE_F32LE } "shape_matrix"; '
            'H5T_IEEE_F32LE "orientation"; H5T_IEEE_F32LE "extremum_value"; '
            'H5T_STD_U8LE "type"; H5T_STD_I8LE "extremum_type"; }') in flat
    assert 'DATASET "features"' in flat and "SIMPLE { ( 4 ) / ( 4 ) }" in flat
    assert ('DATASET "descriptors" { DATATYPE H5T_IEEE_F32LE DATASPACE SIMPLE '
            '{ ( 4, 128 ) / ( 4, 128 ) }') in flat
    # values as stored: the shape matrix in storage (column-major) order
    data = subprocess.run([exe, "-d", "/0/features", path], capture_output=True,
                          text=True).stdout
    assert "[ 1.5, 1.5, 1.5, 1.5 ]" in " ".join(data.split())


def test_cpp_shim_round_trip(tmp_path):
    """tests/cpp/test_h5.cpp = test_features_hdf5.cpp through DO::Sara names;
    the file it leaves is readable from Python."""
    import __graft_entry__
    from sara_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        __graft_entry__.build()
    cpp = os.path.join(HERE, "cpp")
    subprocess.check_call(["make", "-s", "-C", cpp, "test_h5"])
    path = str(tmp_path / "cpp.h5")
    res = subprocess.run([os.path.join(cpp, "test_h5"), path], capture_output=True,
                         text=True)
    assert res.returncode == 0, (res.returncode, res.stdout, res.stderr)
    back = sara_amd.read_keypoints(sara_amd.H5File(path), "sfm/frame/1")
    assert len(back) == 4 and back.descriptor_matrix[3, 127] == 511
