"""ctypes binding of the CPU oracle (oracle/_build/libplh_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libplh_oracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
KL_DTYPE = np.dtype([("angle", "<f4"), ("class_id", "<i4"), ("octave", "<i4"), ("pt_x", "<f4"), ("pt_y", "<f4"),
                     ("response", "<f4"), ("size", "<f4"), ("startPointX", "<f4"), ("startPointY", "<f4"),
                     ("endPointX", "<f4"), ("endPointY", "<f4"), ("sPointInOctaveX", "<f4"),
                     ("sPointInOctaveY", "<f4"), ("ePointInOctaveX", "<f4"), ("ePointInOctaveY", "<f4"),
                     ("lineLength", "<f4"), ("numOfPixels", "<i4")])
assert KP_DTYPE.itemsize == 28 and KL_DTYPE.itemsize == 68


def _stale():
    if not os.path.exists(_SO):
        return True
    t = os.path.getmtime(_SO)
    deps = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cc", ".h", "Makefile"))]
    deps.append(os.path.join(_HERE, "..", "include", "plh_orb_pattern.inc"))
    return any(os.path.getmtime(f) > t for f in deps if os.path.exists(f))


def build(force=False):
    """Compile the oracle with g++ (make).  Rebuilds when a source is newer than the .so.  Safe to call from several processes
    at once (the ranks of a multi-GPU bench.py each verify their own batch): the check and the make run under an exclusive file
    lock, so a second caller waits and then finds the library fresh instead of dlopen()ing a half-written file."""
    if not (force or _stale()):
        return _SO
    import fcntl
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    with open(os.path.join(os.path.dirname(_SO), ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if force or _stale():
                subprocess.check_call(["make", "-C", _HERE, "-s"])
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return _SO


_lib = None

_V, _I, _F, _D, _Z, _U = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t, C.c_uint
_SIGS = {
    "plo_cv_round_f": ([_F], _I),
    "plo_fast_atan2": ([_F, _F], _F),
    "plo_resize_linear_u8": ([_V, _I, _I, _Z, _V, _I, _I, _Z], None),
    "plo_gaussian_kernel_q8": ([_I, _D, _V], None),
    "plo_gaussian_blur_u8": ([_V, _I, _I, _Z, _V, _Z, _I, _D], None),
    "plo_fast_score": ([_V, _Z, _I], _I),
    "plo_fast9_16": ([_V, _I, _I, _Z, _I, _I, _V, _I], _I),
    "plo_sobel3_s16": ([_V, _I, _I, _Z, _V, _V], None),
    "plo_undistort_maps": ([_V, _V, _I, _I, _V, _V], None),
    "plo_remap_linear_u8": ([_V, _I, _I, _Z, _V, _V, _V, _Z], None),
    "plo_orb_create": ([_I, _F, _I, _I, _I], _V),
    "plo_orb_destroy": ([_V], None),
    "plo_orb_levels": ([_V], _I),
    "plo_orb_scale_table": ([_V, _I, _V], None),
    "plo_orb_features_per_level": ([_V, _V], None),
    "plo_orb_umax": ([_V, _V], None),
    "plo_orb_extract": ([_V, _V, _I, _I, _Z, _V, _V, _I], _I),
    "plo_orb_level_size": ([_V, _I, _V, _V], _I),
    "plo_orb_read_level": ([_V, _I, _V], _I),
    "plo_orb_read_blurred": ([_V, _I, _V], _I),
    "plo_orb_read_candidates": ([_V, _I, _V, _I], _I),
    "plo_descriptor_distance": ([_V, _V], _I),
    "plo_knn2": ([_V, _I, _V, _I, _V, _V], None),
    "plo_line_mad": ([_V, _I, _V, _V], None),
    "plo_line_bfmatch": ([_V, _I, _V, _I, _F, _F, _V], None),
    "plo_line_search_double": ([_V, _I, _V, _I, _F, _F, _V], _I),
    "plo_line_bfmatch_new": ([_V, _I, _V, _I, _V, _V, _V, _V, _F, _F, _V], None),
    "plo_line_search_for_triangulation_new": ([_V, _I, _V, _I, _V, _V, _V, _V, _V, _V, _V, _V, _F, _F, _I, _V], _I),
    "plo_orb_search_by_bow": ([_V, _V, _V, _V, _I, _V, _V, _V, _I, _I, _F, _I, _V], _I),
    "plo_lsd_detect": ([_V, _I, _I, _Z, _V, _I], _I),
    "plo_lsd_detect_ex": ([_V, _I, _I, _Z, _V, _I, _I], _I),
    "plo_lsd_nfa": ([_I, _I, _I, _I, _D], _D),
    "plo_lsd_log_gamma": ([_D], _D),
    "plo_keylines_from_segments": ([_V, _I, _I, _I, _V, _Z, _V], _I),
    "plo_lbd_compute": ([_V, _I, _I, _Z, _V, _I, _V, _V], None),
    "plo_line_extract": ([_V, _I, _I, _Z, _V, _U, _D, _V, _V, _V, _I], _I),
    "plo_line_extract_ex": ([_V, _I, _I, _Z, _V, _U, _D, _V, _V, _V, _I, _I], _I),
    "plo_line_extract_oct": ([_V, _I, _I, _Z, _V, _I, _F, _U, _D, _V, _V, _V, _I, _I], _I),
    "plo_pyr_down_u8": ([_V, _I, _I, _Z, _V, _I, _I, _Z], _I),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        for name, (args, res) in _SIGS.items():
            if hasattr(_lib, name):
                f = getattr(_lib, name)
                f.argtypes = args
                f.restype = res
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OrbOracle:
    """plo_orb wrapper: restatement of ORB_SLAM2::ORBextractor (reference src/ORBextractor.cc)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.h = self.L.plo_orb_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        if not self.h:
            raise ValueError("bad ORB parameters")
        self.nlevels = nlevels
        self.nfeatures = nfeatures

    def __del__(self):
        if getattr(self, "h", None):
            self.L.plo_orb_destroy(self.h)
            self.h = None

    def scale_table(self, which=0):
        out = np.zeros(self.nlevels, np.float32)
        self.L.plo_orb_scale_table(self.h, which, _p(out))
        return out

    def features_per_level(self):
        out = np.zeros(self.nlevels, np.int32)
        self.L.plo_orb_features_per_level(self.h, _p(out))
        return out

    def umax(self):
        out = np.zeros(16, np.int32)
        self.L.plo_orb_umax(self.h, _p(out))
        return out

    def extract(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        rows, cols = img.shape
        cap = self.nfeatures + 16 * self.nlevels + 64
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = self.L.plo_orb_extract(self.h, _p(img), rows, cols, cols, _p(kps), _p(desc), cap)
        if n < 0:
            raise RuntimeError("oracle capacity")
        return kps[:n].copy(), desc[:n].copy()

    def level(self, l, blurred=False):
        r, c = C.c_int(), C.c_int()
        self.L.plo_orb_level_size(self.h, l, C.byref(r), C.byref(c))
        out = np.zeros((r.value, c.value), np.uint8)
        rc = (self.L.plo_orb_read_blurred if blurred else self.L.plo_orb_read_level)(self.h, l, _p(out))
        return out if rc == 0 else None

    def candidates(self, l, cap=200000):
        out = np.zeros(cap, KP_DTYPE)
        n = self.L.plo_orb_read_candidates(self.h, l, _p(out), cap)
        return out[:n].copy()


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().plo_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.shape[1], _p(dst), dw, dh, dw)
    return dst


def gaussian_blur(src, ksize, sigma):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros_like(src)
    lib().plo_gaussian_blur_u8(_p(src), src.shape[1], src.shape[0], src.shape[1], _p(dst), src.shape[1], ksize, sigma)
    return dst


def gaussian_kernel_q8(ksize, sigma):
    k = np.zeros(ksize, np.int32)
    lib().plo_gaussian_kernel_q8(ksize, sigma, _p(k))
    return k


def fast9_16(img, threshold, nonmax=True):
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size
    out = np.zeros(cap, KP_DTYPE)
    n = lib().plo_fast9_16(_p(img), img.shape[1], img.shape[0], img.shape[1], threshold, int(nonmax), _p(out), cap)
    return out[:n].copy()


def knn2(q, t):
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    idx = np.zeros((len(q), 2), np.int32)
    dist = np.zeros((len(q), 2), np.int32)
    lib().plo_knn2(_p(q), len(q), _p(t), len(t), _p(idx), _p(dist))
    return idx, dist


# The refine level of the cv::LineSegmentDetector the reference runs: src/LineExtractor.cpp:39-40 creates the SYSTEM
# opencv_contrib LSDDetector (include/auxiliar.h:11-16 includes <opencv2/line_descriptor/descriptor.hpp>; the twin under
# Thirdparty/line_descriptor is commented out there), and the published 3.x module creates its detector with
# cv::LSD_REFINE_ADV.  `refine=None` below means this level; 0 / 1 select LSD_REFINE_STD (what the un-linked twin creates,
# LSDDetector_custom.cpp:149 -- the goldens ref_line_* come from that twin) / LSD_REFINE_ADV explicitly.
REFERENCE_REFINE = 1


def _refine(level):
    return REFERENCE_REFINE if level is None else int(level)


def lsd_detect(img, cap=20000, refine=None):
    """cv::LineSegmentDetector(refine).detect -> float32 [n,4] (x1,y1,x2,y2); refine=None: REFERENCE_REFINE (LSD_REFINE_ADV)."""
    refine = _refine(refine)
    img = np.ascontiguousarray(img, np.uint8)
    segs = np.zeros((cap, 4), np.float32)
    n = lib().plo_lsd_detect_ex(_p(img), img.shape[1], img.shape[0], img.shape[1], _p(segs), cap, int(refine))
    return segs[:min(n, cap)].copy()


def lsd_stage_taps(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    sw, sh = int(np.rint(w * 0.8)), int(np.rint(h * 0.8))
    scaled = np.zeros((sh, sw), np.uint8)
    ang = np.zeros((sh, sw), np.float64)
    mod = np.zeros((sh, sw), np.float64)
    order = np.zeros((sh - 1) * (sw - 1), np.int32)
    a, b = C.c_int(), C.c_int()
    _SIGS_L = lib().plo_lsd_stage_taps
    _SIGS_L.argtypes = [_V, _I, _I, _Z, _V, _V, _V, _V, _V, _V]
    _SIGS_L.restype = _I
    n = _SIGS_L(_p(img), w, h, w, _p(scaled), _p(ang), _p(mod), _p(order), C.byref(a), C.byref(b))
    assert (a.value, b.value) == (sw, sh) and n == len(order)
    return scaled, ang, mod, order


class ReferenceThrows(RuntimeError):
    """The reference raises (cv::pyrDown's size assertion) or runs into undefined behaviour for this configuration."""


def line_extract(img, n_lsd_feature=200, min_line_length=0.0, mask=None, refine=None, num_octaves=1, scale=1.2):
    """LINEextractor(num_octaves, scale, ...)::operator() -> (keylines[KL_DTYPE], desc[n,32], linefn[n,3]); refine=None:
    REFERENCE_REFINE (LSD_REFINE_ADV, see above)."""
    refine = _refine(refine)
    img = np.ascontiguousarray(img, np.uint8)
    cap = n_lsd_feature + 1
    kl = np.zeros(cap, KL_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    fn = np.zeros((cap, 3), np.float64)
    if mask is not None:
        mask = np.ascontiguousarray(mask, np.uint8)
        if mask.shape != img.shape:
            raise ValueError("Mask error while detecting lines")
    n = lib().plo_line_extract_oct(_p(img), img.shape[0], img.shape[1], img.shape[1], _p(mask), int(num_octaves), float(scale),
                                   n_lsd_feature, float(min_line_length), _p(kl), _p(desc), _p(fn), cap, int(refine))
    if n in (-3, -4):
        raise ReferenceThrows("LINEextractor(numOctaves %d, scale %g): %s" % (
            num_octaves, scale, "cv::pyrDown's size assertion fails" if n == -3 else "undefined behaviour in BinaryDescriptor::computeImpl"))
    if n < 0:
        raise RuntimeError("oracle line capacity")
    return kl[:n].copy(), desc[:n].copy(), fn[:n].copy()


def lbd_compute(img, keylines):
    img = np.ascontiguousarray(img, np.uint8)
    keylines = np.ascontiguousarray(keylines)
    n = len(keylines)
    desc = np.zeros((n, 32), np.uint8)
    f72 = np.zeros((n, 72), np.float32)
    lib().plo_lbd_compute(_p(img), img.shape[1], img.shape[0], img.shape[1], _p(keylines), n, _p(desc), _p(f72))
    return desc, f72
