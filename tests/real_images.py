"""Access to the real-image parity pack (tests/golden/real, real_images.npz)."""
import functools
import hashlib
import os

import numpy as np

from common import GOLDEN

NAMES = ("All", "GuardOnBlonde", "sift_edge", "ksmall", "dog", "stinkbug",
         "image-pinhole", "image-omni")
TAGS = ("default", "bench")
PAIR = ("All", "GuardOnBlonde")
RATIOS = (0.6, 1.0, 1.2)


@functools.lru_cache(maxsize=None)
def pack():
    return np.load(os.path.join(GOLDEN, "real_images.npz"))


@functools.lru_cache(maxsize=None)
def rgb(name):
    from PIL import Image
    return np.array(Image.open(os.path.join(GOLDEN, "real", name + ".png")))


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def gray(rb, name):
    """Rgb8 -> float as Image<Rgb8>::convert<float>() does
    (Core/Pixel/SmartColorConversion.hpp:236-245), checked against the pack."""
    g = rb.rgb8_to_gray32f(rgb(name))
    assert sha(g) == str(pack()[name + "_gray_sha256"])
    return g


def ref_params(rb, tag):
    """`default`: ImagePyramidParams() of the C++ API (first octave -1, every
    octave); `bench`: first octave 0, 4 octaves."""
    return (rb.PyramidParams() if tag == "default"
            else rb.PyramidParams(0, 6, None, 1, 0.5, 1.6, 4))


def hip_params(tag):
    import sara_amd
    if tag == "default":   # the C++ default; the Python class defaults to +1
        return sara_amd.ImagePyramidParams(first_octave_index=-1)
    return sara_amd.ImagePyramidParams(0, 6, image_padding_size=1,
                                       scale_camera=0.5, scale_initial=1.6,
                                       num_octaves_max=4)


@functools.lru_cache(maxsize=None)
def _pair_descriptors():
    import refbind as rb
    d = tuple(rb.RefSift(gray(rb, n), ref_params(rb, "default"),
                         parallel=True).keypoints()[2] for n in PAIR)
    assert sha(*d) == str(pack()["pair_sha256"])
    return d


def pair_descriptors(rb):
    """The oracle's `default` descriptors of the examples' matching pair (what
    the FLANN lists of the pack were computed from)."""
    return _pair_descriptors()
