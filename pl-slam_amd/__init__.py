"""plslam_amd -- Python host side of the MI355X-native PL-SLAM front end.

The product is the C-ABI shared library `libplslam_hip.so` (hand-written HIP kernels for gfx950,
see include/plslam_hip.h).  This module is the thin host mirror of the reference's operator
interface -- `ORBextractor`, `LINEextractor`, `ORBmatcher`, `LSDmatcher` with the reference's
constructor arguments and call semantics (include/ORBextractor.h:45-111, LineExtractor.h:20-62,
ORBmatcher.h:37-102, LSDmatcher.h:22-76) -- implemented with ctypes over that library.

There is NO CPU fallback: if the HIP library is missing or no GPU is visible every operator
raises.  (oracle/ is test infrastructure and is never imported from here.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libplslam_hip.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
KL_DTYPE = np.dtype([("angle", "<f4"), ("class_id", "<i4"), ("octave", "<i4"), ("pt_x", "<f4"), ("pt_y", "<f4"),
                     ("response", "<f4"), ("size", "<f4"), ("startPointX", "<f4"), ("startPointY", "<f4"),
                     ("endPointX", "<f4"), ("endPointY", "<f4"), ("sPointInOctaveX", "<f4"),
                     ("sPointInOctaveY", "<f4"), ("ePointInOctaveX", "<f4"), ("ePointInOctaveY", "<f4"),
                     ("lineLength", "<f4"), ("numOfPixels", "<i4")])
assert KP_DTYPE.itemsize == 28 and KL_DTYPE.itemsize == 68


class PlhError(RuntimeError):
    pass


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32)]


class LineParams(C.Structure):
    _fields_ = [("num_octaves", C.c_int32), ("scale", C.c_float), ("n_lsd_feature", C.c_uint32),
                ("min_line_length", C.c_double)]


_V, _I, _F, _Z = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_SIGS = {
    "plh_last_error": ([], C.c_char_p),
    "plh_version": ([], C.c_char_p),
    "plh_device_count": ([], _I),
    "plh_orb_create": ([_V, _I, _I, _I, _I, _V], _I),
    "plh_orb_destroy": ([_V], _I),
    "plh_orb_scale_table": ([_V, _I, _V], _I),
    "plh_orb_levels": ([_V], _I),
    "plh_orb_features_per_level": ([_V, _V], _I),
    "plh_orb_capacity": ([_V], _I),
    "plh_orb_extract": ([_V, _V, _I, _I, _Z, _V, _V, _I, _V], _I),
    "plh_orb_extract_batch": ([_V, _V, _I, _Z, _V, _V, _V], _I),
    "plh_orb_extract_batch_dev": ([_V, _V, _I, _Z, _V, _V, _V, _V], _I),
    "plh_orb_set_profiling": ([_V, _I], _I),
    "plh_orb_kernel_ms": ([_V, _I, _V, _V], _I),
    "plh_orb_pyramid_dev": ([_V, _I, _I, _V, _V, _V, _V], _I),
    "plh_orb_read_level": ([_V, _I, _I, _V, _Z], _I),
    "plh_orb_read_candidates": ([_V, _I, _I, _V, _I, _V], _I),
    "plh_descriptor_distance": ([_V, _V], _I),
    "plh_hamming_knn2_dev": ([_V, _I, _V, _I, _V, _V, _V], _I),
    "plh_hamming_knn2": ([_V, _I, _V, _I, _V, _V, _I], _I),
    "plh_hamming_knn2_batch_dev": ([_V, _V, _I, _V, _V, _I, _I, _V, _V, _V], _I),
    "plh_line_bfmatch_batch_dev": ([_V, _V, _V, _V, _I, _I, _F, _F, _V, _V], _I),
    "plh_line_search_double_batch_dev": ([_V, _V, _V, _V, _I, _I, _F, _F, _V, _V, _V, _Z, _V], _I),
    "plh_line_search_double_workspace": ([_I, _I], _Z),
    "plh_orb_search_by_bow_batch_dev": ([_V, _V, _V, _V, _V, _V, _V, _V, _V, _I, _I, _I, _F, _I, _V, _V, _V], _I),
    "plh_line_create": ([_V, _I, _I, _I, _I, _V], _I),
    "plh_line_destroy": ([_V], _I),
    "plh_line_capacity": ([_V], _I),
    "plh_line_set_undistort": ([_V, _V, _V], _I),
    "plh_line_extract": ([_V, _V, _I, _I, _Z, _V, _V, _V, _V, _I, _V], _I),
    "plh_line_extract_batch_dev": ([_V, _V, _I, _Z, _V, _V, _V, _V, _V, _V], _I),
    "plh_line_read_segments": ([_V, _I, _V, _I, _V], _I),
}

_libs = {}


def load(path=None):
    """Load the C-ABI library.  Raises PlhError (never falls back) when it is missing."""
    path = os.path.abspath(path or os.environ.get("PLSLAM_HIP_LIB", DEFAULT_LIB))
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise PlhError("HIP library not found: %s -- run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
    lib = C.CDLL(path)
    for name, (args, res) in _SIGS.items():
        if hasattr(lib, name):
            f = getattr(lib, name)
            f.argtypes = args
            f.restype = res
    _libs[path] = lib
    return lib


def exported_symbols():
    """Entry points include/plslam_hip.h declares (parsed from the header)."""
    import re
    hdr = open(os.path.join(_HERE, "..", "include", "plslam_hip.h")).read()
    return sorted(set(re.findall(r"PLH_API\s+[\w\s\*]+?\b(plh_\w+)\s*\(", hdr)))


def _check(lib, st, what):
    if st != 0:
        msg = lib.plh_last_error()
        raise PlhError("%s failed: status %d: %s" % (what, st, msg.decode() if msg else ""))


def _p(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if hasattr(a, "data_ptr"):   # torch tensor (device memory)
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(int(a))


class ORBextractor:
    """ORB_SLAM2::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) on the GPU.

    `__call__(image)` is the reference's operator() (ORBextractor.cc:1043-1105): returns
    (keypoints[KP_DTYPE], descriptors[n,32] u8).  The mask argument is ignored, as in the reference.
    `extract_batch_dev` is the batch path the throughput numbers are measured on.
    """

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, rows=480, cols=640, max_batch=1,
                 device=0, lib=None):
        self.lib = load(lib)
        self.params = OrbParams(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
        self.rows, self.cols, self.max_batch, self.device = rows, cols, max_batch, device
        h = C.c_void_p()
        _check(self.lib, self.lib.plh_orb_create(C.byref(self.params), device, rows, cols, max_batch, C.byref(h)),
               "plh_orb_create")
        self.h = h
        self.nlevels = nlevels
        self.capacity = self.lib.plh_orb_capacity(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.plh_orb_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # reference getters (ORBextractor.h:62-84)
    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return float(self.params.scale_factor)

    def _table(self, which):
        out = np.zeros(self.nlevels, np.float32)
        _check(self.lib, self.lib.plh_orb_scale_table(self.h, which, _p(out)), "plh_orb_scale_table")
        return out

    def GetScaleFactors(self):
        return self._table(0)

    def GetInverseScaleFactors(self):
        return self._table(1)

    def GetScaleSigmaSquares(self):
        return self._table(2)

    def GetInverseScaleSigmaSquares(self):
        return self._table(3)

    def features_per_level(self):
        out = np.zeros(self.nlevels, np.int32)
        _check(self.lib, self.lib.plh_orb_features_per_level(self.h, _p(out)), "plh_orb_features_per_level")
        return out

    def __call__(self, image, mask=None):
        image = np.asarray(image)
        if image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2, "image.type() == CV_8UC1"
        image = np.ascontiguousarray(image)
        kps = np.zeros(self.capacity, KP_DTYPE)
        desc = np.zeros((self.capacity, 32), np.uint8)
        n = C.c_int(0)
        _check(self.lib, self.lib.plh_orb_extract(self.h, _p(image), image.shape[0], image.shape[1], image.strides[0],
                                                  _p(kps), _p(desc), self.capacity, C.byref(n)), "plh_orb_extract")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, images):
        images = np.ascontiguousarray(images, np.uint8)
        b = images.shape[0]
        kps = np.zeros((b, self.capacity), KP_DTYPE)
        desc = np.zeros((b, self.capacity, 32), np.uint8)
        n = np.zeros(b, np.int32)
        _check(self.lib, self.lib.plh_orb_extract_batch(self.h, _p(images), b, images.strides[0], _p(kps), _p(desc), _p(n)),
               "plh_orb_extract_batch")
        return kps, desc, n

    def extract_batch_dev(self, d_imgs, batch, frame_stride, d_kps, d_desc, d_n, stream=0):
        """Device pointers (ints or torch tensors); asynchronous on `stream` (hipStream_t as int)."""
        _check(self.lib, self.lib.plh_orb_extract_batch_dev(self.h, _p(d_imgs), batch, frame_stride, _p(d_kps), _p(d_desc),
                                                            _p(d_n), C.c_void_p(stream)), "plh_orb_extract_batch_dev")

    def set_profiling(self, on=True):
        _check(self.lib, self.lib.plh_orb_set_profiling(self.h, int(on)), "plh_orb_set_profiling")

    def kernel_ms(self, kernel):
        """(total_ms, intervals) of kernel group 0 pyramid / 1 FAST / 2 quad-tree / 3 orient+rBRIEF."""
        ms, n = C.c_double(0), C.c_int(0)
        _check(self.lib, self.lib.plh_orb_kernel_ms(self.h, kernel, C.byref(ms), C.byref(n)), "plh_orb_kernel_ms")
        return ms.value, n.value

    # parity taps
    def read_level(self, b, level, shape):
        out = np.zeros(shape, np.uint8)
        _check(self.lib, self.lib.plh_orb_read_level(self.h, b, level, _p(out), out.size), "plh_orb_read_level")
        return out

    def read_candidates(self, b, level, cap=200000):
        out = np.zeros(cap, KP_DTYPE)
        n = C.c_int(0)
        _check(self.lib, self.lib.plh_orb_read_candidates(self.h, b, level, _p(out), cap, C.byref(n)), "plh_orb_read_candidates")
        return out[:n.value].copy()
