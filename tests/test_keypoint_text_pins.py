"""SURVEY.md section 8f, row f3, pinned on a file the REFERENCE wrote:
tests/golden/test.dogkey = /root/reference/cpp/examples/Sara/Features/test.dogkey
(data: 578 DoG keypoints x 128 bins, read by
examples/Sara/Features/features_read_write_example.cpp:121 through
read_keypoints, Features/IO.hpp:77-108).  The file predates the one-line-per-
keypoint writer of Features/IO.hpp:110-143: coordinates, the 2 x 2 shape
matrix through Eigen's operator<< (two aligned rows), orientation, type and the
descriptor in lines of 20 - the token stream read_keypoints consumes is the
same.  It pins (a) the reader, token for token, and (b) the writer's Eigen
alignment rule on 578 blocks the reference's ostream produced."""
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DOGKEY = os.path.join(HERE, "golden", "test.dogkey")
# same example, Harris-affine regions: 226 anisotropic shape matrices with
# negative off-diagonal coefficients (leading-blank alignment), type 1
HARAFFKEY = os.path.join(HERE, "golden", "test.haraffkey")
CPP = os.path.join(HERE, "cpp")


def _records(path=DOGKEY, n=578):
    """The file cut into its per-keypoint text records, no number parsing:
    (coordinate line, two matrix lines, orientation line, type line,
    descriptor tokens)."""
    lines = open(path).read().split("\n")
    assert lines[0] == "%d 128" % n
    recs, at = [], 1
    for _ in range(n):
        head = lines[at:at + 5]
        at += 5
        toks = []
        while len(toks) < 128:
            toks += lines[at].split()
            at += 1
        assert len(toks) == 128
        recs.append((head, toks))
    assert all(not l.strip() for l in lines[at:])
    return recs


def test_reader_on_the_reference_file():
    import sara_amd
    keys = sara_amd.read_keypoints(DOGKEY)
    reg, desc = keys.regions, keys.descriptor_matrix
    assert len(reg) == 578 and desc.shape == (578, 128)
    # the first keypoint as features_read_write_example.cpp prints it
    assert np.array_equal(reg["coords"][0], np.float32([105.461, 481.567]))
    assert np.array_equal(reg["shape_matrix"][0],
                          np.float32([0.000364329, 0, 0, 0.000364329]))
    assert reg["orientation"][0] == np.float32(2.66816)
    assert reg["type"][0] == 5  # OERegion::Type::DoG, Features/Feature.hpp:48-56
    assert np.array_equal(desc[0, :20], np.float32(
        [0, 0, 0, 0, 0, 0, 0, 0, 0, 33, 13, 5, 0, 0, 0, 0, 2, 57, 6, 0]))
    # every record, token for token
    for i, (head, toks) in enumerate(_records()):
        x, y = head[0].split()
        assert reg["coords"][i][0] == np.float32(x)
        assert reg["coords"][i][1] == np.float32(y)
        m00, m01 = head[1].split()
        m10, m11 = head[2].split()
        # column-major storage: (m00, m10, m01, m11)
        assert np.array_equal(reg["shape_matrix"][i],
                              np.float32([m00, m10, m01, m11]))
        assert reg["orientation"][i] == np.float32(head[3])
        assert reg["type"][i] == int(head[4]) == 5
        assert np.array_equal(desc[i], np.float32(toks))
    # the fields the format does not carry keep OERegion's defaults
    assert np.all(reg["extremum_type"] == -2)
    assert np.all(reg["extremum_value"] == 0)
    # SIFT descriptors as Sara quantises them: integers in [0, 255]
    assert desc.min() >= 0 and desc.max() <= 255
    assert np.array_equal(desc, np.round(desc))


def test_writer_alignment_rule_reproduces_the_reference_bytes():
    """Every coefficient of one Eigen expression is right-aligned to the widest
    one: re-emitting the 578 shape matrices (and the scalars next to them)
    through the writer's routines gives the file's own bytes."""
    import sara_amd
    from sara_amd import _eigen_block, _ostream_float
    keys = sara_amd.read_keypoints(DOGKEY)
    widths = set()
    for r, (head, _) in zip(keys.regions, _records()):
        assert "%s %s" % (_ostream_float(r["coords"][0]),
                          _ostream_float(r["coords"][1])) == head[0]
        block = _eigen_block(r["shape_matrix"], 2, 2)
        assert block == head[1] + "\n" + head[2], (block, head[1:3])
        assert _ostream_float(r["orientation"]) == head[3]
        widths.add(len(head[1]))
    # the pack exercises more than one width (e.g. "0.000364329" vs "1.5e-05")
    assert len(widths) > 1


def test_one_line_writer_round_trips_the_reference_file(tmp_path):
    """Today's writer (Features/IO.hpp:110-143) on the reference's keypoints:
    reading its output gives the same floats (6 significant digits are what the
    file holds) and the descriptor row follows the same alignment rule."""
    import sara_amd
    keys = sara_amd.read_keypoints(DOGKEY)
    path = str(tmp_path / "again.dogkey")
    assert sara_amd.write_keypoints(keys.regions, keys.descriptor_matrix, path)
    back = sara_amd.read_keypoints(path)
    for f in ("coords", "shape_matrix", "orientation", "type"):
        assert np.array_equal(back.regions[f], keys.regions[f]), f
    assert np.array_equal(back.descriptor_matrix, keys.descriptor_matrix)
    line = open(path).read().split("\n")[1]
    d0 = keys.descriptor_matrix[0]
    w = max(len("%g" % v) for v in d0)
    assert line.endswith(" 5 " + " ".join(("%g" % v).rjust(w) for v in d0))
    assert line.startswith("105.461 481.567 0.000364329           0"
                           "           0 0.000364329 2.66816 5 ")


def test_cpp_shim_reader_and_alignment_on_the_reference_file(tmp_path):
    """The same two pins through include/DO/Sara/HipSift.hpp
    (tests/cpp/test_textio.cpp; host code only, no GPU call)."""
    import __graft_entry__
    from sara_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        __graft_entry__.build()
    subprocess.check_call(["make", "-s", "-C", CPP, "test_textio"])
    out = str(tmp_path / "reemitted.txt")
    res = subprocess.run([os.path.join(CPP, "test_textio"), DOGKEY, out],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert "578 128" in res.stdout and "first 105.461 481.567" in res.stdout
    # the program re-emits "coords \n block \n orientation \n type" per record
    want = "\n".join("\n".join(head) for head, _ in _records()) + "\n"
    assert open(out).read() == want


def test_haraff_file_reader_and_alignment():
    """The Harris-affine file of the same example: non-diagonal matrices, so
    the ROW-major read of OERegion's operator>> (Features/Feature.cpp:88-95,
    Core/EigenExtension.hpp:163-170) and the alignment of negative
    coefficients are both exercised."""
    import sara_amd
    from sara_amd import _eigen_block, _ostream_float
    keys = sara_amd.read_keypoints(HARAFFKEY)
    reg = keys.regions
    assert len(reg) == 226 and keys.descriptor_matrix.shape == (226, 128)
    assert np.array_equal(reg["shape_matrix"][0], np.float32(
        [0.00761515, -0.00105516, -0.00105516, 0.00807131]))
    assert np.all(reg["type"] == 1)  # OERegion::Type::HarAff
    lead_blank = msvc = 0
    for r, (head, toks) in zip(reg, _records(HARAFFKEY, 226)):
        m00, m01 = head[1].split()
        m10, m11 = head[2].split()
        assert np.array_equal(r["shape_matrix"],
                              np.float32([m00, m10, m01, m11]))
        block = head[1] + "\n" + head[2]
        if "e" in block:
            # this file came from an MSVC runtime (three exponent digits,
            # "1.43639e-005"): the numbers' own text differs from glibc's, the
            # alignment rule - one width, the widest coefficient - does not
            toks4 = [m00, m01, m10, m11]
            wd = max(len(t) for t in toks4)
            assert block == "%s %s\n%s %s" % tuple(t.rjust(wd) for t in toks4)
            msvc += 1
        else:
            assert _eigen_block(r["shape_matrix"], 2, 2) == block
        assert "%s %s" % (_ostream_float(r["coords"][0]),
                          _ostream_float(r["coords"][1])) == head[0]
        assert _ostream_float(r["orientation"]) == head[3]
        lead_blank += head[1].startswith(" ")
    assert lead_blank > 100 and 0 < msvc < 60
