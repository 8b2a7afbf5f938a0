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


class FrontendParams(C.Structure):   # plh_frontend_params
    _fields_ = [("struct_size", C.c_uint32), ("rows", C.c_int32), ("cols", C.c_int32), ("orb", OrbParams), ("line", LineParams), ("undistort", C.c_int32),
                ("K", C.c_float * 4), ("D", C.c_float * 5), ("bow_levelsup", C.c_int32), ("orb_th_low", C.c_int32),
                ("orb_nnratio", C.c_float), ("orb_check_orientation", C.c_int32), ("line_th", C.c_float), ("line_nnratio", C.c_float),
                ("external_records", C.c_int32), ("lsd_refine", C.c_int32)]


class FrontendRecords(C.Structure):   # plh_frontend_records
    _fields_ = [("first", C.c_int32), ("frames", C.c_int32), ("orb_capacity", C.c_int32), ("line_capacity", C.c_int32)] + \
               [(k, C.c_void_p) for k in ("kps", "desc", "n", "nid", "word", "bow_word", "bow_value", "bow_n", "kl", "ldesc", "lfn", "nl",
                                          "m_orb", "nm_orb", "m_line", "nm_line")]


FRONTEND_GATHERED = 7   # PLH_FRONTEND_GATHERED: n, kps, desc, nl, kl, ldesc, lfn


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
    "plh_orb_status": ([_V, _V], _I),
    "plh_line_status": ([_V, _V], _I),
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
    "plh_bow_transform_batch_dev": ([_V, _V, _I, _I, _V, _V, _V, _V, _V, _I, _I, _V, _V, _V], _I),
    "plh_bow_vector_batch_dev": ([_V, _V, _I, _I, _V, _I, _I, _V, _V, _V, _V], _I),
    "plh_vocab_load_text": ([C.c_char_p, _I, _V], _I),
    "plh_vocab_load_binary": ([C.c_char_p, _I, _V], _I),
    "plh_vocab_create": ([_I, _I, _I, _I, _I, _V, _V, _V, _V, _I, _V], _I),
    "plh_vocab_save_binary": ([_V, C.c_char_p], _I),
    "plh_vocab_destroy": ([_V], _I),
    "plh_vocab_get_info": ([_V, _V], _I),
    "plh_vocab_device_arrays": ([_V, _V, _V, _V, _V, _V, _V], _I),
    "plh_vocab_read": ([_V, _V, _V, _V, _V, _V, _V], _I),
    "plh_vocab_transform_batch_dev": ([_V, _V, _V, _I, _I, _I, _V, _V, _V, _V, _V, _V], _I),
    "plh_comm_unique_id": ([_V], _I),
    "plh_comm_create": ([_V, _I, _I, _I, _V], _I),
    "plh_comm_wrap": ([_V, _I, _I, _V], _I),
    "plh_comm_destroy": ([_V], _I),
    "plh_comm_info": ([_V, _V, _V, _V], _I),
    "plh_gather_records": ([_V, _V, _I, _I, _V], _I),
    "plh_line_create": ([_V, _I, _I, _I, _I, _V], _I),
    "plh_line_destroy": ([_V], _I),
    "plh_line_capacity": ([_V], _I),
    "plh_line_set_undistort": ([_V, _V, _V], _I),
    "plh_line_extract": ([_V, _V, _I, _I, _Z, _V, _V, _V, _V, _I, _V], _I),
    "plh_line_extract_batch_dev": ([_V, _V, _I, _Z, _V, _V, _V, _V, _V, _V], _I),
    "plh_line_read_segments": ([_V, _I, _V, _I, _V], _I),
    "plh_line_set_grow_events": ([_V, _V, _V], _I),
    "plh_line_set_grow_waves": ([_V, _I], _I),
    "plh_line_set_refine": ([_V, _I], _I),
    "plh_line_reserve": ([_V, _I], _I),
    "plh_line_set_screen": ([_V, _I], _I),
    "plh_line_set_grow_tuning": ([_V, _I, _I], _I),
    "plh_lsd_refine_default": ([], _I),
    "plh_orb_search_by_bow_kfkf": ([_V, _V, _V, _V, _I, _V, _V, _V, _V, _I, _I, _F, _I, _V, _V, _I], _I),
    "plh_orb_search_for_triangulation": ([_V, _V, _V, _V, _I, _V, _V, _V, _V, _I, _V, _F, _F, _V, _V, _I, _I, _I, _V, _V, _I], _I),
    "plh_line_frame_bfmatch": ([_V, _I, _V, _I, _F, _F, _V, _I], _I),
    "plh_line_frame_bfmatch_new": ([_V, _I, _V, _I, _V, _V, _V, _V, _F, _F, _V, _I], _I),
    "plh_line_search_for_triangulation_new": ([_V, _I, _V, _I, _V, _V, _V, _V, _V, _V, _V, _V, _F, _F, _I, _V, _V, _I], _I),
    "plh_line_fuse_search": ([_V, _V, _I, _V, _I, _I, _V, _V, _V, _V, _F, _F, _I, _V, _V, _I], _I),
    "plh_orb_search_by_sim3": ([_V, _V, _I, _V, _V, _I, _V, _V, _I, _V, _V, _V, _V, _V, _V, _V, _V, _F, _I, _V, _V, _I], _I),
    "plh_orb_search_by_projection_kf": ([_V, _V, _I, _V, _V, _I, _V, _I, _V, _V, _V, _V, _V, _F, _I, _I, _V, _V, _I], _I),
    "plh_orb_search_by_projection_sim3": ([_V, _V, _I, _V, _V, _I, _V, _I, _V, _V, _V, _V, _V, _F, _I, _V, _V, _I], _I),
    "plh_orb_fuse_search": ([_V, _V, _I, _V, _V, _V, _I, _I, _V, _V, _V, _V, _F, _I, _V, _V, _I], _I),
    "plh_selftest": ([_I, _V, _V, _I], _I),
    "plh_selftest_shims": ([], _I),
    "plh_box_probe": ([_I, _I, _V], _I),
    "plh_frontend_create": ([_V, _V, _I, _I, _I, _V], _I),
    "plh_frontend_destroy": ([_V], _I),
    "plh_frontend_parts": ([_V], _I),
    "plh_frontend_handles": ([_V, _I, _V, _V], _I),
    "plh_frontend_records_of": ([_V, _I, _V], _I),
    "plh_frontend_bind_records": ([_V, _I, _V], _I),
    "plh_frontend_step": ([_V, _V, _Z, _V, _I], _I),
    "plh_frontend_join": ([_V, _V], _I),
    "plh_frontend_set_overlap": ([_V, _I], _I),
    "plh_frontend_gather": ([_V, _V, _I, _V, _V], _I),
    "plh_frontend_gather_bytes": ([_V, _I, _V], _I),
    "plh_frontend_status": ([_V, _V], _I),
    "plh_orb_search_for_triangulation_batch_dev": ([_V] * 10 + [_I, _I, _V, _F, _F, _V, _V, _I, _I, _I, _V, _V, _V], _I),
    "plh_orb_search_by_bow_kfkf_batch_dev": ([_V] * 10 + [_I, _I, _I, _F, _I, _V, _V, _V], _I),
    "plh_orb_search_by_projection_kf_batch_dev": ([_V, _V, _V, _I, _I, _V, _V, _V, _V, _I, _V, _V, _I] + [_V] * 6 +
                                                  [_F, _I, _I, _V, _V, _V], _I),
    "plh_line_fuse_search_batch_dev": ([_V, _V, _V, _I, _I, _V, _I, _V, _I, _V, _V, _V, _V, _F, _F, _I, _V, _V, _V], _I),
    "plh_orb_fuse_search_batch_dev": ([_V, _V, _V, _I, _I, _V, _V, _V, _V, _V, _I, _V, _I, _V, _V, _V, _V, _F, _I, _V, _V, _V], _I),
    "plh_orb_search_by_projection_sim3_batch_dev": ([_V, _V, _V, _I, _I, _V, _V, _V, _V, _I, _V, _V, _I] + [_V] * 5 +
                                                    [_F, _I, _V, _V, _V], _I),
    "plh_orb_search_by_sim3_batch_dev": ([_V] * 10 + [_I, _I, _V, _V, _I] + [_V] * 8 + [_F, _I, _V, _V, _V, _V, _V], _I),
    "plh_undistort_keypoints_batch_dev": ([_V, _V, _I, _I, _V, _V, _V, _V], _I),
    "plh_distinctive_descriptor_batch_dev": ([_V, _V, _I, _V, _V], _I),
    "plh_frame_project_points_batch_dev": ([_V, _I, _V, _I, _V, _I, _V, _V, _V], _I),
    "plh_frame_is_in_frustum_points_batch_dev": ([_V, _I, _V, _I, _V, _V, _V, _V, C.c_float, _V, _V, _V, _V, _V], _I),
    "plh_frame_is_in_frustum_lines_batch_dev": ([_V, _I, _V, _I, _V, _V, _V, _V, C.c_float, _V, _V, _V, _V, _V], _I),
    "plh_frame_assign_grid_batch_dev": ([_V, _V, _I, _I, _V, _V, _V, _V], _I),
    "plh_frame_assign_grid_lines_batch_dev": ([_V, _V, _I, _I, _V, _V, _V, _I, _V], _I),
    "plh_orb_search_for_initialization_batch_dev": ([_V] * 6 + [_I, _I, _V, _V, _V, _V, _I, _F, _I, _V, _V, _V], _I),
    "plh_orb_search_by_projection_mp_batch_dev": ([_V, _V, _V, _I, _I, _V, _V, _V, _V, _I, _V, _V, _I] + [_V] * 6 +
                                                  [_F, _F, _V, _V, _V], _I),
    "plh_orb_search_by_projection_frame_batch_dev": ([_V, _V, _V, _I, _I, _V, _V, _V, _V, _I, _V, _V, _I] + [_V] * 6 +
                                                     [_F, _I, _I, _V, _V, _V], _I),
    "plh_line_search_by_projection_frame_batch_dev": ([_V, _V, _V, _V, _I, _I, _V, _V, _V, _I, _V, _V, _I] + [_V] * 5 +
                                                      [_F, _V, _V, _V], _I),
    "plh_line_search_by_projection_ml_batch_dev": ([_V, _V, _V, _V, _I, _I, _V, _V, _V, _I, _V, _V, _I] + [_V] * 5 +
                                                   [_F, _F, _V, _V, _V], _I),
}

_libs = {}


def load(path=None):
    """Load the C-ABI library.  Raises PlhError (never falls back) when it is missing."""
    path = os.path.abspath(path or os.environ.get("PLSLAM_HIP_LIB", DEFAULT_LIB))
    if path in _libs:
        return _libs[path]
    if "hipemu" not in path:
        try:   # share PyTorch's HIP runtime instance when it is present (it provides device memory / streams)
            import torch  # noqa: F401
        except Exception:
            pass
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

    def status(self):
        """Capacity flags of the most recent extract call (0 = nothing was truncated); waits for that call."""
        f = C.c_int(0)
        _check(self.lib, self.lib.plh_orb_status(self.h, C.byref(f)), "plh_orb_status")
        return f.value

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


# ------------------------------------------------------------------------------------------------
# device buffers: torch tensors on the GPU (PyTorch is only the allocator / stream provider);
# under the hipemu test build "device" memory is host memory, so numpy arrays are passed directly.
# ------------------------------------------------------------------------------------------------
def is_emulated(lib):
    return b"hipemu" in lib.plh_version()


class _Dev:
    def __init__(self, lib, device=0):
        self.emu = is_emulated(lib)
        self.device = device
        if not self.emu:
            import torch
            if not torch.cuda.is_available():
                raise PlhError("no GPU visible and no CPU fallback exists")
            self.torch = torch
            self.dev = torch.device("cuda", device)

    def put(self, a):
        a = np.ascontiguousarray(a)
        if self.emu:
            return a.copy()
        return self.torch.from_numpy(a.view(np.uint8).reshape(-1) if a.dtype.names else a).to(self.dev)

    def empty(self, shape, dtype):
        if self.emu:
            return np.zeros(shape, dtype)
        tmap = {np.int32: self.torch.int32, np.uint8: self.torch.uint8, np.float32: self.torch.float32,
                np.float64: self.torch.float64}
        return self.torch.zeros(shape, dtype=tmap[dtype], device=self.dev)

    def get(self, d):
        if self.emu:
            return d
        self.torch.cuda.synchronize(self.dev)
        return d.cpu().numpy()

    def stream(self):
        return 0 if self.emu else self.torch.cuda.current_stream(self.dev).cuda_stream


def descriptor_distance(a, b, lib=None):
    """ORBmatcher::DescriptorDistance / LSDmatcher::DescriptorDistance (256-bit Hamming)."""
    L = load(lib)
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return L.plh_descriptor_distance(_p(a), _p(b))


def hamming_knn2(q, t, device=0, lib=None):
    """cv::BFMatcher(NORM_HAMMING).knnMatch(q, t, k=2) -> (idx[nq,2], dist[nq,2])."""
    L = load(lib)
    D = _Dev(L, device)
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    nq, nt = len(q), len(t)
    if nq == 0:
        return np.zeros((0, 2), np.int32), np.zeros((0, 2), np.int32)
    dq, dt = D.put(q), D.put(t if nt else np.zeros((1, 32), np.uint8))
    di, dd = D.empty((nq, 2), np.int32), D.empty((nq, 2), np.int32)
    _check(L, L.plh_hamming_knn2_dev(_p(dq), nq, _p(dt), nt, _p(di), _p(dd), C.c_void_p(D.stream())), "plh_hamming_knn2_dev")
    return D.get(di), D.get(dd)


def _pad_sets(sets, cap, width, dtype):
    out = np.zeros((len(sets), cap) + ((width,) if width else ()), dtype)
    n = np.zeros(len(sets), np.int32)
    for i, s in enumerate(sets):
        s = np.asarray(s, dtype)
        n[i] = len(s)
        if len(s):
            out[i, :len(s)] = s
    return out, n



GRID_COLS, GRID_ROWS = 64, 48          # FRAME_GRID_COLS / FRAME_GRID_ROWS, include/Frame.h:44-45
GRID_CELLS = GRID_COLS * GRID_ROWS


class GridParams(C.Structure):
    _fields_ = [("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float),
                ("inv_w", C.c_float), ("inv_h", C.c_float)]


def grid_params(cols, rows, min_x=0.0, min_y=0.0, max_x=None, max_y=None):
    """Frame::ComputeImageBounds + the static grid members (Frame.cc:113-117): bounds of the undistorted image and
    mfGridElementWidthInv / HeightInv = 64 / (maxX - minX), 48 / (maxY - minY), all in float32."""
    max_x = np.float32(cols if max_x is None else max_x)
    max_y = np.float32(rows if max_y is None else max_y)
    min_x, min_y = np.float32(min_x), np.float32(min_y)
    return GridParams(min_x, min_y, max_x, max_y, np.float32(GRID_COLS) / (max_x - min_x), np.float32(GRID_ROWS) / (max_y - min_y))


def _gp_array(gp):
    return np.array([gp.min_x, gp.min_y, gp.max_x, gp.max_y, gp.inv_w, gp.inv_h], np.float32)


def _pad_records(sets, cap, dtype):
    """list of structured arrays -> (P, cap) array + counts."""
    out = np.zeros((len(sets), cap), dtype)
    n = np.zeros(len(sets), np.int32)
    for i, s in enumerate(sets):
        out[i, :len(s)] = s
        n[i] = len(s)
    return out, n


class FrameSearch:
    """The grid ("windowed") searches of the tracking thread on flat arrays, batched over independent frames.
    One instance holds the device copies of a batch of frames (the `Frame` side of every call):
      points: kps_un [KP_DTYPE], desc [n,32]         -> Frame::AssignFeaturesToGrid        (Frame.cc:278-293)
      lines : keylines [KL_DTYPE], ldesc, linefn     -> Frame::AssignFeaturesToGridForLine (Frame.cc:295-320)
    and exposes ORBmatcher::SearchForInitialization / SearchByProjection and LSDmatcher::SearchByProjection."""

    def __init__(self, gp, scale_factors, frames, device=0, lib=None, cap=None):
        self.lib = load(lib)
        self.D = _Dev(self.lib, device)
        self.gp = gp
        self.sf = np.ascontiguousarray(scale_factors, np.float32)
        D, L = self.D, self.lib
        self.P = len(frames)
        s = C.c_void_p(D.stream())
        self.cap = max(1, max(len(f.get("kps", ())) for f in frames), cap or 0)   # cap: a common capacity for SearchBySim3
        kps, self.n = _pad_records([f.get("kps", np.zeros(0, KP_DTYPE)) for f in frames], self.cap, KP_DTYPE)
        desc, _ = _pad_sets([f.get("desc", np.zeros((0, 32), np.uint8)) for f in frames], self.cap, 32, np.uint8)
        self.d_kps, self.d_desc, self.d_n = D.put(kps), D.put(desc), D.put(self.n)
        self.d_cs = D.empty((self.P, GRID_CELLS + 1), np.int32)
        self.d_ci = D.empty((self.P, self.cap), np.int32)
        _check(L, L.plh_frame_assign_grid_batch_dev(_p(self.d_kps), _p(self.d_n), self.cap, self.P, C.byref(gp), _p(self.d_cs),
                                                    _p(self.d_ci), s), "plh_frame_assign_grid_batch_dev")
        self.lcap = max(1, max(len(f.get("keylines", ())) for f in frames))
        self.item_cap = self.lcap * GRID_COLS
        kl, self.nl = _pad_records([f.get("keylines", np.zeros(0, KL_DTYPE)) for f in frames], self.lcap, KL_DTYPE)
        ld, _ = _pad_sets([f.get("ldesc", np.zeros((0, 32), np.uint8)) for f in frames], self.lcap, 32, np.uint8)
        fn, _ = _pad_sets([f.get("linefn", np.zeros((0, 3), np.float64)) for f in frames], self.lcap, 3, np.float64)
        self.d_kl, self.d_ld, self.d_fn, self.d_nl = D.put(kl), D.put(ld), D.put(fn), D.put(self.nl)
        self.d_lcs = D.empty((self.P, GRID_CELLS + 1), np.int32)
        self.d_lci = D.empty((self.P, self.item_cap), np.int32)
        _check(L, L.plh_frame_assign_grid_lines_batch_dev(_p(self.d_kl), _p(self.d_nl), self.lcap, self.P, C.byref(gp),
                                                          _p(self.d_lcs), _p(self.d_lci), self.item_cap, s),
               "plh_frame_assign_grid_lines_batch_dev")

    def grids(self):
        """(cell_start[P, 3073], cell_items[P, cap]) of the point grid and of the line grid."""
        D = self.D
        return (D.get(self.d_cs), D.get(self.d_ci)), (D.get(self.d_lcs), D.get(self.d_lci))

    def _queries(self, qs, fields):
        qcap = max(1, max(len(q["valid"]) for q in qs))
        out = []
        for name, width, dt in fields:
            a, nq = _pad_sets([q[name] for q in qs], qcap, width, dt)
            out.append(self.D.put(a))
        return qcap, self.D.put(nq), out

    def SearchForInitialization(self, f1s, prev_matched, windowSize=100, nnratio=0.9, checkOri=True):
        """ORBmatcher(nnratio, checkOri).SearchForInitialization(F1, F2 = this frame, vbPrevMatched, vnMatches12,
        windowSize).  f1s: list of dict(kps, desc); prev_matched: list of [n1, 2] float32 (returned updated).
        Returns (matches12[P, cap], nmatches[P], prev_matched[P, cap, 2])."""
        D, L = self.D, self.lib
        cap = max(self.cap, max(len(f["kps"]) for f in f1s))
        assert cap == self.cap, "F1 may not have more keypoints than the planned capacity"
        k1, n1 = _pad_records([f["kps"] for f in f1s], cap, KP_DTYPE)
        d1, _ = _pad_sets([f["desc"] for f in f1s], cap, 32, np.uint8)
        pm, _ = _pad_sets(prev_matched, cap, 2, np.float32)
        dk1, dd1, dn1, dpm = D.put(k1), D.put(d1), D.put(n1), D.put(pm)
        dm, dc = D.empty((self.P, cap), np.int32), D.empty((self.P,), np.int32)
        _check(L, L.plh_orb_search_for_initialization_batch_dev(
            _p(dk1), _p(dd1), _p(dn1), _p(self.d_kps), _p(self.d_desc), _p(self.d_n), cap, self.P, C.byref(self.gp),
            _p(self.d_cs), _p(self.d_ci), _p(dpm), int(windowSize), float(nnratio), int(checkOri), _p(dm), _p(dc),
            C.c_void_p(D.stream())), "plh_orb_search_for_initialization_batch_dev")
        return D.get(dm), D.get(dc), D.get(dpm)

    def SearchByProjectionMapPoints(self, qs, occupied, th=1.0, nnratio=0.8):
        """ORBmatcher(nnratio).SearchByProjection(F, vpMapPoints, th).  qs: per frame dict(valid, xy, level, viewcos,
        desc, hasobs); occupied: per frame u8[n].  Returns (assigned[P, cap], nmatches[P], occupied[P, cap])."""
        D, L = self.D, self.lib
        qcap, dnq, (qv, qxy, ql, qc, qd, qh) = self._queries(qs, [("valid", 0, np.uint8), ("xy", 2, np.float32),
                                                                   ("level", 0, np.int32), ("viewcos", 0, np.float32),
                                                                   ("desc", 32, np.uint8), ("hasobs", 0, np.uint8)])
        occ, _ = _pad_sets(occupied, self.cap, 0, np.uint8)
        docc = D.put(occ)
        da, dc = D.empty((self.P, self.cap), np.int32), D.empty((self.P,), np.int32)
        _check(L, L.plh_orb_search_by_projection_mp_batch_dev(
            _p(self.d_kps), _p(self.d_desc), _p(self.d_n), self.cap, self.P, C.byref(self.gp), _p(self.d_cs), _p(self.d_ci),
            _p(self.sf), len(self.sf), _p(docc), _p(dnq), qcap, _p(qv), _p(qxy), _p(ql), _p(qc), _p(qd), _p(qh), float(th),
            float(nnratio), _p(da), _p(dc), C.c_void_p(D.stream())), "plh_orb_search_by_projection_mp_batch_dev")
        return D.get(da), D.get(dc), D.get(docc)

    def SearchByProjectionLastFrame(self, qs, occupied, th=15.0, mode=0, checkOri=True):
        """ORBmatcher(0.9, checkOri).SearchByProjection(CurrentFrame = this frame, LastFrame, th, bMono).
        qs: per frame dict(valid, uv, octave, angle, desc, hasobs)."""
        D, L = self.D, self.lib
        qcap, dnq, (qv, quv, qo, qa, qd, qh) = self._queries(qs, [("valid", 0, np.uint8), ("uv", 2, np.float32),
                                                                   ("octave", 0, np.int32), ("angle", 0, np.float32),
                                                                   ("desc", 32, np.uint8), ("hasobs", 0, np.uint8)])
        occ, _ = _pad_sets(occupied, self.cap, 0, np.uint8)
        docc = D.put(occ)
        da, dc = D.empty((self.P, self.cap), np.int32), D.empty((self.P,), np.int32)
        _check(L, L.plh_orb_search_by_projection_frame_batch_dev(
            _p(self.d_kps), _p(self.d_desc), _p(self.d_n), self.cap, self.P, C.byref(self.gp), _p(self.d_cs), _p(self.d_ci),
            _p(self.sf), len(self.sf), _p(docc), _p(dnq), qcap, _p(qv), _p(quv), _p(qo), _p(qa), _p(qd), _p(qh), float(th),
            int(mode), int(checkOri), _p(da), _p(dc), C.c_void_p(D.stream())), "plh_orb_search_by_projection_frame_batch_dev")
        return D.get(da), D.get(dc), D.get(docc)

    def SearchByProjectionKeyFrame(self, qs, occupied, th=10.0, ORBdist=100, checkOri=True):
        """ORBmatcher(0.9, checkOri).SearchByProjection(CurrentFrame = this frame, pKF, sAlreadyFound, th, ORBdist)
        (relocalisation).  qs: per frame dict(valid, uv, level, angle, desc, hasobs)."""
        D, L = self.D, self.lib
        qcap, dnq, (qv, quv, ql, qa, qd, qh) = self._queries(qs, [("valid", 0, np.uint8), ("uv", 2, np.float32),
                                                                   ("level", 0, np.int32), ("angle", 0, np.float32),
                                                                   ("desc", 32, np.uint8), ("hasobs", 0, np.uint8)])
        occ, _ = _pad_sets(occupied, self.cap, 0, np.uint8)
        docc = D.put(occ)
        da, dc = D.empty((self.P, self.cap), np.int32), D.empty((self.P,), np.int32)
        _check(L, L.plh_orb_search_by_projection_kf_batch_dev(
            _p(self.d_kps), _p(self.d_desc), _p(self.d_n), self.cap, self.P, C.byref(self.gp), _p(self.d_cs), _p(self.d_ci),
            _p(self.sf), len(self.sf), _p(docc), _p(dnq), qcap, _p(qv), _p(quv), _p(ql), _p(qa), _p(qd), _p(qh), float(th),
            int(ORBdist), int(checkOri), _p(da), _p(dc), C.c_void_p(D.stream())), "plh_orb_search_by_projection_kf_batch_dev")
        return D.get(da), D.get(dc), D.get(docc)

    def FuseSearch(self, qs, inv_level_sigma2, th=3.0, TH_LOW=50):
        """The search inside ORBmatcher::Fuse(pKF = this frame, vpMapPoints, th).  qs: per frame dict(valid, uv, level, desc).
        Returns (best_idx[P, qcap], nfound[P])."""
        D, L = self.D, self.lib
        qcap, dnq, (qv, quv, ql, qd) = self._queries(qs, [("valid", 0, np.uint8), ("uv", 2, np.float32), ("level", 0, np.int32),
                                                         ("desc", 32, np.uint8)])
        is2 = np.ascontiguousarray(inv_level_sigma2, np.float32)
        db, dc = D.empty((self.P, qcap), np.int32), D.empty((self.P,), np.int32)
        _check(L, L.plh_orb_fuse_search_batch_dev(
            _p(self.d_kps), _p(self.d_desc), _p(self.d_n), self.cap, self.P, C.byref(self.gp), _p(self.d_cs), _p(self.d_ci),
            _p(self.sf), _p(is2), len(self.sf), _p(dnq), qcap, _p(qv), _p(quv), _p(ql), _p(qd), float(th), int(TH_LOW), _p(db),
            _p(dc), C.c_void_p(D.stream())), "plh_orb_fuse_search_batch_dev")
        return D.get(db), D.get(dc)

    def SearchByProjectionSim3(self, qs, occupied, th=10, TH_LOW=50):
        """ORBmatcher.SearchByProjection(pKF = this frame, Scw, vpPoints, vpMatched, th) (loop closing).
        qs: per frame dict(valid, uv, level, desc, hasobs)."""
        D, L = self.D, self.lib
        qcap, dnq, (qv, quv, ql, qd, qh) = self._queries(qs, [("valid", 0, np.uint8), ("uv", 2, np.float32), ("level", 0, np.int32),
                                                             ("desc", 32, np.uint8), ("hasobs", 0, np.uint8)])
        occ, _ = _pad_sets(occupied, self.cap, 0, np.uint8)
        docc = D.put(occ)
        da, dc = D.empty((self.P, self.cap), np.int32), D.empty((self.P,), np.int32)
        _check(L, L.plh_orb_search_by_projection_sim3_batch_dev(
            _p(self.d_kps), _p(self.d_desc), _p(self.d_n), self.cap, self.P, C.byref(self.gp), _p(self.d_cs), _p(self.d_ci),
            _p(self.sf), len(self.sf), _p(docc), _p(dnq), qcap, _p(qv), _p(quv), _p(ql), _p(qd), _p(qh), float(th), int(TH_LOW),
            _p(da), _p(dc), C.c_void_p(D.stream())), "plh_orb_search_by_projection_sim3_batch_dev")
        return D.get(da), D.get(dc), D.get(docc)

    def SearchBySim3(self, other, q12, q21, th=7.5, TH_HIGH=100):
        """ORBmatcher.SearchBySim3(pKF1 = this batch, pKF2 = `other`, vpMatches12, s12, R12, t12, th) (loop closing).
        q12: per pair dict(valid, uv, level, desc), one query per keypoint slot of KeyFrame 1 (its map point transformed into
        KeyFrame 2); q21 the reverse.  Returns (match12[P, cap], nfound[P], vnMatch1[P, cap], vnMatch2[P, cap])."""
        D, L = self.D, self.lib
        assert other.cap == self.cap and other.P == self.P, "both KeyFrame batches need the same capacity (FrameSearch(cap=...))"
        fields = [("valid", 0, np.uint8), ("uv", 2, np.float32), ("level", 0, np.int32), ("desc", 32, np.uint8)]
        dq = []
        for qs, fs in ((q12, self), (q21, other)):
            for b, q in enumerate(qs):
                assert len(q["valid"]) == fs.n[b], "one query per keypoint slot"
            dq.append([D.put(_pad_sets([q[name] for q in qs], self.cap, width, dt)[0]) for name, width, dt in fields])
        dm1, dm2, dm12 = (D.empty((self.P, self.cap), np.int32) for _ in range(3))
        dc = D.empty((self.P,), np.int32)
        _check(L, L.plh_orb_search_by_sim3_batch_dev(
            _p(self.d_kps), _p(self.d_desc), _p(self.d_n), _p(self.d_cs), _p(self.d_ci), _p(other.d_kps), _p(other.d_desc),
            _p(other.d_n), _p(other.d_cs), _p(other.d_ci), self.cap, self.P, C.byref(self.gp), _p(self.sf), len(self.sf),
            _p(dq[0][0]), _p(dq[0][1]), _p(dq[0][2]), _p(dq[0][3]), _p(dq[1][0]), _p(dq[1][1]), _p(dq[1][2]), _p(dq[1][3]),
            float(th), int(TH_HIGH), _p(dm1), _p(dm2), _p(dm12), _p(dc), C.c_void_p(D.stream())),
            "plh_orb_search_by_sim3_batch_dev")
        return D.get(dm12), D.get(dc), D.get(dm1), D.get(dm2)

    def LineFuseSearch(self, qs, scale_factors_line, cand_descs=None, th=3.0, cos_th=0.998, TH_LOW=50):
        """The search inside LSDmatcher::Fuse(pKF = this frame, vpMapLines, th).  qs: per frame dict(valid, seg, level, desc);
        cand_descs: per frame the rows compared against (default: the frame's LBD descriptors)."""
        D, L = self.D, self.lib
        qcap, dnq, (qv, qs_, ql, qd) = self._queries(qs, [("valid", 0, np.uint8), ("seg", 4, np.float32), ("level", 0, np.int32),
                                                         ("desc", 32, np.uint8)])
        dcand = self.d_ld
        if cand_descs is not None:
            cd, _ = _pad_sets(cand_descs, self.lcap, 32, np.uint8)
            dcand = D.put(cd)
        sfl = np.ascontiguousarray(scale_factors_line, np.float32)
        db, dc = D.empty((self.P, qcap), np.int32), D.empty((self.P,), np.int32)
        _check(L, L.plh_line_fuse_search_batch_dev(_p(self.d_kl), _p(dcand), _p(self.d_nl), self.lcap, self.P, _p(sfl), len(sfl), _p(dnq),
                                                   qcap, _p(qv), _p(qs_), _p(ql), _p(qd), float(th), float(cos_th), int(TH_LOW),
                                                   _p(db), _p(dc), C.c_void_p(D.stream())), "plh_line_fuse_search_batch_dev")
        return D.get(db), D.get(dc)

    def LineSearchByProjectionLastFrame(self, qs, occupied, th=8.0):
        """LSDmatcher.SearchByProjection(CurrentFrame = this frame, LastFrame, th).
        qs: per frame dict(valid, seg[nq,4], length, desc, hasobs)."""
        D, L = self.D, self.lib
        qcap, dnq, (qv, qs_, ql, qd, qh) = self._queries(qs, [("valid", 0, np.uint8), ("seg", 4, np.float32),
                                                              ("length", 0, np.float32), ("desc", 32, np.uint8),
                                                              ("hasobs", 0, np.uint8)])
        occ, _ = _pad_sets(occupied, self.lcap, 0, np.uint8)
        docc = D.put(occ)
        da, dc = D.empty((self.P, self.lcap), np.int32), D.empty((self.P,), np.int32)
        _check(L, L.plh_line_search_by_projection_frame_batch_dev(
            _p(self.d_kl), _p(self.d_ld), _p(self.d_fn), _p(self.d_nl), self.lcap, self.P, C.byref(self.gp), _p(self.d_lcs),
            _p(self.d_lci), self.item_cap, _p(docc), _p(dnq), qcap, _p(qv), _p(qs_), _p(ql), _p(qd), _p(qh), float(th),
            _p(da), _p(dc), C.c_void_p(D.stream())), "plh_line_search_by_projection_frame_batch_dev")
        return D.get(da), D.get(dc), D.get(docc)

    def LineSearchByProjectionMapLines(self, qs, occupied, th=1.0, nnratio=0.7):
        """LSDmatcher(nnratio).SearchByProjection(F = this frame, vpMapLines, th).
        qs: per frame dict(valid, seg[nq,4], viewcos, desc, hasobs)."""
        D, L = self.D, self.lib
        qcap, dnq, (qv, qs_, qc, qd, qh) = self._queries(qs, [("valid", 0, np.uint8), ("seg", 4, np.float32),
                                                              ("viewcos", 0, np.float32), ("desc", 32, np.uint8),
                                                              ("hasobs", 0, np.uint8)])
        occ, _ = _pad_sets(occupied, self.lcap, 0, np.uint8)
        docc = D.put(occ)
        da, dc = D.empty((self.P, self.lcap), np.int32), D.empty((self.P,), np.int32)
        _check(L, L.plh_line_search_by_projection_ml_batch_dev(
            _p(self.d_kl), _p(self.d_ld), _p(self.d_fn), _p(self.d_nl), self.lcap, self.P, C.byref(self.gp), _p(self.d_lcs),
            _p(self.d_lci), self.item_cap, _p(docc), _p(dnq), qcap, _p(qv), _p(qs_), _p(qc), _p(qd), _p(qh), float(th),
            float(nnratio), _p(da), _p(dc), C.c_void_p(D.stream())), "plh_line_search_by_projection_ml_batch_dev")
        return D.get(da), D.get(dc), D.get(docc)


def undistort_keypoints(kps_list, K, D, device=0, lib=None):
    """Frame::UndistortKeyPoints for a batch of frames: list of KP_DTYPE arrays -> list of undistorted copies."""
    L = load(lib)
    Dv = _Dev(L, device)
    cap = max(1, max(len(k) for k in kps_list))
    a, n = _pad_records(kps_list, cap, KP_DTYPE)
    da, dn = Dv.put(a), Dv.put(n)
    Kf = np.ascontiguousarray(K, np.float32)
    Df = np.ascontiguousarray(D if D is not None else np.zeros(5), np.float32)
    _check(L, L.plh_undistort_keypoints_batch_dev(_p(da), _p(dn), cap, len(kps_list), _p(Kf), _p(Df), _p(da),
                                                  C.c_void_p(Dv.stream())), "plh_undistort_keypoints_batch_dev")
    out = np.ascontiguousarray(Dv.get(da)).view(np.uint8).reshape(len(kps_list), cap, 28).copy().view(KP_DTYPE).reshape(len(kps_list), cap)
    return [out[i, :len(k)].copy() for i, k in enumerate(kps_list)]


VIEW_DTYPE = np.dtype([("Rcw", np.float32, 9), ("tcw", np.float32, 3), ("Ow", np.float32, 3), ("fx", np.float32), ("fy", np.float32),
                       ("cx", np.float32), ("cy", np.float32), ("min_x", np.float32), ("min_y", np.float32), ("max_x", np.float32),
                       ("max_y", np.float32), ("log_scale_factor", np.float32), ("n_scale_levels", np.int32)])   # == plh_frame_view


def project_points(views, positions, form, device=0, lib=None):
    """The inline projection of the pose-driven searches (form 0 / 1 / 2, see include/plslam_hip.h) for a batch of frames:
    positions = list of [n, 3] float32 world points.  Returns per frame (front[n] u8, uv[n, 2] float32)."""
    L = load(lib)
    Dv = _Dev(L, device)
    P = len(positions)
    qcap = max(1, max(len(x) for x in positions))
    pos, nq = _pad_sets([np.asarray(x, np.float32).reshape(-1, 3) for x in positions], qcap, 3, np.float32)
    dv = Dv.put(np.ascontiguousarray(views, VIEW_DTYPE).view(np.uint8).reshape(P, VIEW_DTYPE.itemsize))
    dp, dnq = Dv.put(pos), Dv.put(nq)
    of, ou = Dv.empty((P, qcap), np.uint8), Dv.empty((P, qcap, 2), np.float32)
    _check(L, L.plh_frame_project_points_batch_dev(_p(dv), P, _p(dnq), qcap, _p(dp), int(form), _p(of), _p(ou), C.c_void_p(Dv.stream())),
           "plh_frame_project_points_batch_dev")
    f, u = Dv.get(of), Dv.get(ou)
    return [(f[b, :len(x)].copy(), u[b, :len(x)].copy()) for b, x in enumerate(positions)]


GATES_DTYPE = np.dtype([("view", VIEW_DTYPE), ("R2", np.float32, 9), ("t2", np.float32, 3), ("flags", np.int32)])   # == plh_point_gates
GATE_Z, GATE_INVZ_DOUBLE, GATE_UV_NORMALISED, GATE_KEYFRAME_BOUNDS, GATE_DIST_OF_TARGET, GATE_NORMAL, GATE_SECOND = 1, 2, 4, 8, 16, 32, 64


def map_point_gates(view, flags, pos, normal, min_dist_inv, max_dist_inv, max_dist=None, pre=None, R2=None, t2=None, device=0, lib=None):
    """plh_map_point_gates (host buffers): the gates between the pose transform and the window lookup of the back end's pose-driven
    searches, for all map points of one call.  view: one VIEW_DTYPE record.  Returns (valid u8[n], uv [n, 2], dist [n], level [n])."""
    L = load(lib)
    n = len(pos)
    g = np.zeros(1, GATES_DTYPE)
    g["view"][0] = np.asarray(view, VIEW_DTYPE).reshape(())
    g["flags"][0] = int(flags)
    if R2 is not None:
        g["R2"][0] = np.asarray(R2, np.float32).reshape(9)
        g["t2"][0] = np.asarray(t2, np.float32).reshape(3)
    m = max(n, 1)
    valid = np.ones(m, np.uint8) if pre is None else np.ascontiguousarray(pre, np.uint8).copy()
    uv, dist, level = np.zeros((m, 2), np.float32), np.zeros(m, np.float32), np.zeros(m, np.int32)
    f32 = lambda a, w: np.ascontiguousarray(np.asarray(a, np.float32).reshape(-1, w) if w > 1 else np.asarray(a, np.float32).reshape(-1))
    a_pos, a_min, a_max = f32(pos, 3), f32(min_dist_inv, 1), f32(max_dist_inv, 1)
    a_nrm = f32(normal, 3) if normal is not None else None
    a_raw = f32(max_dist, 1) if max_dist is not None else None
    L.plh_map_point_gates.argtypes = [_V, _I, _V, _V, _V, _V, _V, _V, _V, _V, _V, _I]
    L.plh_map_point_gates.restype = _I
    _check(L, L.plh_map_point_gates(_p(g), n, _p(a_pos), _p(a_nrm) if a_nrm is not None else None, _p(a_min), _p(a_max),
                                    _p(a_raw) if a_raw is not None else None, _p(valid), _p(uv), _p(dist), _p(level), int(device)),
           "plh_map_point_gates")
    return valid[:n], uv[:n], dist[:n], level[:n]


def is_in_frustum(views, elems, viewing_cos_limit, lines=False, device=0, lib=None):
    """Frame::isInFrustum for the local map of a batch of frames.  views: VIEW_DTYPE[P]; elems: per frame
    dict(pos [n,3] (points) or [n,6] (lines), normal [n,3], min_dist [n], max_dist [n]).  Returns per frame
    dict(valid, uv | seg, level, viewcos) -- the query arrays of SearchByProjection(F, MapPoints / MapLines)."""
    L = load(lib)
    Dv = _Dev(L, device)
    P = len(elems)
    w = 6 if lines else 3
    qcap = max(1, max(len(e["min_dist"]) for e in elems))
    pos, nq = _pad_sets([np.asarray(e["pos"], np.float32).reshape(-1, w) for e in elems], qcap, w, np.float32)
    nrm, _ = _pad_sets([np.asarray(e["normal"], np.float32).reshape(-1, 3) for e in elems], qcap, 3, np.float32)
    mind, _ = _pad_sets([e["min_dist"] for e in elems], qcap, 0, np.float32)
    maxd, _ = _pad_sets([e["max_dist"] for e in elems], qcap, 0, np.float32)
    dv = Dv.put(np.ascontiguousarray(views, VIEW_DTYPE).view(np.uint8).reshape(P, VIEW_DTYPE.itemsize))
    dp, dn, dmi, dma, dnq = Dv.put(pos), Dv.put(nrm), Dv.put(mind), Dv.put(maxd), Dv.put(nq)
    pw = 4 if lines else 2
    ov, op, ol, oc = Dv.empty((P, qcap), np.uint8), Dv.empty((P, qcap, pw), np.float32), Dv.empty((P, qcap), np.int32), \
        Dv.empty((P, qcap), np.float32)
    fn = L.plh_frame_is_in_frustum_lines_batch_dev if lines else L.plh_frame_is_in_frustum_points_batch_dev
    _check(L, fn(_p(dv), P, _p(dnq), qcap, _p(dp), _p(dn), _p(dmi), _p(dma), float(viewing_cos_limit), _p(ov), _p(op), _p(ol), _p(oc),
                 C.c_void_p(Dv.stream())), "plh_frame_is_in_frustum_%s_batch_dev" % ("lines" if lines else "points"))
    v, pr, lv, vc = Dv.get(ov), Dv.get(op), Dv.get(ol), Dv.get(oc)
    out = []
    for b, e in enumerate(elems):
        n = len(e["min_dist"])
        out.append({"valid": v[b, :n].copy(), ("seg" if lines else "uv"): pr[b, :n].copy(), "level": lv[b, :n].copy(),
                    "viewcos": vc[b, :n].copy()})
    return out


def distinctive_descriptors(sets, device=0, lib=None):
    """MapPoint / MapLine::ComputeDistinctiveDescriptors for many map elements: list of [n_s, 32] u8 -> best row per set."""
    L = load(lib)
    Dv = _Dev(L, device)
    off = np.zeros(len(sets) + 1, np.int32)
    off[1:] = np.cumsum([len(s) for s in sets])
    allrows = np.concatenate([np.asarray(s, np.uint8).reshape(-1, 32) for s in sets] + [np.zeros((1, 32), np.uint8)])
    dd, do = Dv.put(allrows), Dv.put(off)
    db = Dv.empty((len(sets),), np.int32)
    _check(L, L.plh_distinctive_descriptor_batch_dev(_p(dd), _p(do), len(sets), _p(db), C.c_void_p(Dv.stream())),
           "plh_distinctive_descriptor_batch_dev")
    return Dv.get(db)


class LSDmatcher:
    """ORB_SLAM2::LSDmatcher(nnratio=0.7, checkOri=true) -- the brute-force descriptor searches
    (reference include/LSDmatcher.h:22-76, src/LSDmatcher.cpp:12-14).  Frames are given by their LBD
    descriptor matrices (mLdesc)."""
    TH_HIGH = 80
    TH_LOW = 50

    def __init__(self, nnratio=0.7, checkOri=True, device=0, lib=None):
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)
        self.lib = load(lib)
        self.D = _Dev(self.lib, device)

    def SearchDoubleBatch(self, descs1, descs2):
        """SearchDouble(Frame&, Frame&, vector<int>&) for P independent frame pairs.
        Returns (matches12[P, cap] (-1 = unmatched), nmatches[P])."""
        P = len(descs1)
        assert P == len(descs2) and P > 0
        cap = max(1, max(len(d) for d in list(descs1) + list(descs2)))
        a, n1 = _pad_sets(descs1, cap, 32, np.uint8)
        b, n2 = _pad_sets(descs2, cap, 32, np.uint8)
        D, L = self.D, self.lib
        da, db, dn1, dn2 = D.put(a), D.put(b), D.put(n1), D.put(n2)
        dm, dc = D.empty((P, cap), np.int32), D.empty((P,), np.int32)
        wsb = L.plh_line_search_double_workspace(cap, P)
        ws = D.empty((wsb,), np.uint8)
        _check(L, L.plh_line_search_double_batch_dev(_p(da), _p(dn1), _p(db), _p(dn2), cap, P, float(self.TH_LOW),
                                                     self.mfNNratio, _p(dm), _p(dc), _p(ws), wsb, C.c_void_p(D.stream())),
               "plh_line_search_double_batch_dev")
        return D.get(dm), D.get(dc)

    def SearchDouble(self, ldesc1, ldesc2):
        """Returns (nmatches, LineMatches[NL1]) like the reference's out-parameter version (LSDmatcher.cpp:427-460)."""
        ldesc1 = np.asarray(ldesc1, np.uint8).reshape(-1, 32)
        ldesc2 = np.asarray(ldesc2, np.uint8).reshape(-1, 32)
        if len(ldesc1) == 0 or len(ldesc2) == 0:
            return 0, np.full(len(ldesc1), -1, np.int32)
        m, c = self.SearchDoubleBatch([ldesc1], [ldesc2])
        return int(c[0]), m[0, :len(ldesc1)].copy()

    def FrameBFMatch(self, ldesc1, ldesc2, TH=None, nnratio=None):
        """LSDmatcher::FrameBFMatch (LSDmatcher.cpp:462-486): LineMatches[NL1]."""
        ldesc1 = np.asarray(ldesc1, np.uint8).reshape(-1, 32)
        ldesc2 = np.asarray(ldesc2, np.uint8).reshape(-1, 32)
        n1, n2 = len(ldesc1), len(ldesc2)
        if n1 == 0:
            return np.zeros(0, np.int32)
        D, L = self.D, self.lib
        cap = max(n1, n2, 1)
        a, na = _pad_sets([ldesc1], cap, 32, np.uint8)
        b, nb = _pad_sets([ldesc2], cap, 32, np.uint8)
        da, db, dna, dnb = D.put(a), D.put(b), D.put(na), D.put(nb)
        di, dd = D.empty((1, cap, 2), np.int32), D.empty((1, cap, 2), np.int32)
        dm = D.empty((1, cap), np.int32)
        s = C.c_void_p(D.stream())
        _check(L, L.plh_hamming_knn2_batch_dev(_p(da), _p(dna), cap, _p(db), _p(dnb), cap, 1, _p(di), _p(dd), s), "knn2")
        _check(L, L.plh_line_bfmatch_batch_dev(_p(di), _p(dd), _p(dna), _p(dnb), cap, 1,
                                               float(self.TH_LOW if TH is None else TH),
                                               float(self.mfNNratio if nnratio is None else nnratio), _p(dm), s), "bfmatch")
        return D.get(dm)[0, :n1].copy()

    def SerachForInitialize(self, ldesc1, ldesc2):
        """LSDmatcher::SerachForInitialize (sic; LSDmatcher.cpp:340-373, call site Tracking.cc:710 commented out): the nearest neighbour of
        every line of the initial frame, kept where the gap to the second nearest exceeds half the MAD of the gaps -- FrameBFMatch without
        its distance and ratio tests.  Returns (nmatches, LineMatches[NL1])."""
        ldesc1 = np.asarray(ldesc1, np.uint8).reshape(-1, 32)
        ldesc2 = np.asarray(ldesc2, np.uint8).reshape(-1, 32)
        if len(ldesc1) == 0 or len(ldesc2) == 0:
            return 0, np.full(len(ldesc1), -1, np.int32)
        m = self.FrameBFMatch(ldesc1, ldesc2, TH=float("inf"), nnratio=float("inf"))
        return int((m >= 0).sum()), m


    @staticmethod
    def _line_args(ldesc, seg, func):
        ldesc = np.ascontiguousarray(np.asarray(ldesc, np.uint8).reshape(-1, 32))
        seg = np.ascontiguousarray(np.asarray(seg, np.float32).reshape(-1, 4))
        func = np.ascontiguousarray(np.asarray(func, np.float64).reshape(-1, 3))
        assert len(seg) == len(ldesc) and len(func) == len(ldesc)
        return ldesc, seg, func

    def FrameBFMatchNew(self, ldesc1, ldesc2, seg1, seg2, func2, F, TH=None):
        """LSDmatcher::FrameBFMatchNew (LSDmatcher.cpp:488-548): LineMatches[NL1].  seg = (startPointX, startPointY, endPointX,
        endPointY) per KeyLine, func2 = mvKeyLineFunctions of set 2, F the 3 x 3 fundamental matrix (CV_32F)."""
        ldesc1, seg1, _ = self._line_args(ldesc1, seg1, np.zeros((len(np.asarray(seg1).reshape(-1, 4)), 3)))
        ldesc2, seg2, func2 = self._line_args(ldesc2, seg2, func2)
        F = np.ascontiguousarray(np.asarray(F, np.float32).reshape(9))
        m = np.full(max(len(ldesc1), 1), -1, np.int32)
        L = self.lib
        _check(L, L.plh_line_frame_bfmatch_new(_p(ldesc1), len(ldesc1), _p(ldesc2), len(ldesc2), _p(seg1), _p(seg2), _p(func2), _p(F),
                                               float(self.TH_LOW if TH is None else TH), self.mfNNratio, _p(m), self.D.device),
               "plh_line_frame_bfmatch_new")
        return m[:len(ldesc1)].copy()

    def SearchForTriangulationNew(self, ldesc1, ldesc2, seg1, seg2, func1, func2, F21, F12, has_ml1, has_ml2, isDouble=False):
        """LSDmatcher::SearchForTriangulationNew (LSDmatcher.cpp:780-832): (nmatches, vMatchedPairs[NL1]).  F21 = ComputeF12(pKF2, pKF1),
        F12 = ComputeF12(pKF1, pKF2); has_ml = the line already carries a MapLine."""
        ldesc1, seg1, func1 = self._line_args(ldesc1, seg1, func1)
        ldesc2, seg2, func2 = self._line_args(ldesc2, seg2, func2)
        F21 = np.ascontiguousarray(np.asarray(F21, np.float32).reshape(9))
        F12 = np.ascontiguousarray(np.asarray(F12, np.float32).reshape(9))
        ml1 = np.ascontiguousarray(np.asarray(has_ml1, np.uint8).reshape(-1))
        ml2 = np.ascontiguousarray(np.asarray(has_ml2, np.uint8).reshape(-1))
        assert len(ml1) == len(ldesc1) and len(ml2) == len(ldesc2)
        m = np.full(max(len(ldesc1), 1), -1, np.int32)
        c = C.c_int(0)
        L = self.lib
        _check(L, L.plh_line_search_for_triangulation_new(_p(ldesc1), len(ldesc1), _p(ldesc2), len(ldesc2), _p(seg1), _p(seg2), _p(func1),
                                                          _p(func2), _p(F21), _p(F12), _p(ml1), _p(ml2), float(self.TH_LOW),
                                                          self.mfNNratio, 1 if isDouble else 0, _p(m), C.byref(c), self.D.device),
               "plh_line_search_for_triangulation_new")
        return int(c.value), m[:len(ldesc1)].copy()


class ORBmatcher:
    """ORB_SLAM2::ORBmatcher(nnratio=0.6, checkOri=true) -- Hamming searches on flat arrays
    (reference include/ORBmatcher.h:37-102, src/ORBmatcher.cc:37-39)."""
    TH_HIGH = 100
    TH_LOW = 50
    HISTO_LENGTH = 30

    def __init__(self, nnratio=0.6, checkOri=True, device=0, lib=None):
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)
        self.lib = load(lib)
        self.D = _Dev(self.lib, device)

    @staticmethod
    def DescriptorDistance(a, b):
        return descriptor_distance(a, b)

    def SearchByBoWBatch(self, kf_sets, f_sets):
        """SearchByBoW(KeyFrame*, Frame&, ...) for P pairs.  Each kf set = dict(desc[n,32], angle[n], node[n], valid[n]);
        each frame set = dict(desc, angle, node).  Returns (matches21[P, cap], nmatches[P])."""
        P = len(kf_sets)
        cap = max(1, max(len(s["desc"]) for s in list(kf_sets) + list(f_sets)))
        d1, n1 = _pad_sets([s["desc"] for s in kf_sets], cap, 32, np.uint8)
        a1, _ = _pad_sets([s["angle"] for s in kf_sets], cap, 0, np.float32)
        k1, _ = _pad_sets([s["node"] for s in kf_sets], cap, 0, np.int32)
        v1, _ = _pad_sets([s["valid"] for s in kf_sets], cap, 0, np.uint8)
        d2, n2 = _pad_sets([s["desc"] for s in f_sets], cap, 32, np.uint8)
        a2, _ = _pad_sets([s["angle"] for s in f_sets], cap, 0, np.float32)
        k2, _ = _pad_sets([s["node"] for s in f_sets], cap, 0, np.int32)
        D, L = self.D, self.lib
        bufs = [D.put(x) for x in (d1, a1, k1, v1, n1, d2, a2, k2, n2)]
        dm, dc = D.empty((P, cap), np.int32), D.empty((P,), np.int32)
        _check(L, L.plh_orb_search_by_bow_batch_dev(*[_p(x) for x in bufs], cap, P, self.TH_LOW, self.mfNNratio,
                                                    int(self.mbCheckOrientation), _p(dm), _p(dc), C.c_void_p(D.stream())),
               "plh_orb_search_by_bow_batch_dev")
        return D.get(dm), D.get(dc)

    def SearchByBoWKeyFramesBatch(self, kf1_sets, kf2_sets):
        """SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vpMatches12) (ORBmatcher.cc:574-709) for P pairs.  Each set =
        dict(desc, kps[KP_DTYPE], node, valid).  Returns (matches12[P, cap], nmatches[P])."""
        P = len(kf1_sets)
        cap = max(1, max(len(s["desc"]) for s in list(kf1_sets) + list(kf2_sets)))
        D, L = self.D, self.lib
        bufs = []
        for sets in (kf1_sets, kf2_sets):
            d, n = _pad_sets([s["desc"] for s in sets], cap, 32, np.uint8)
            k, _ = _pad_records([s["kps"] for s in sets], cap, KP_DTYPE)
            nd, _ = _pad_sets([s["node"] for s in sets], cap, 0, np.int32)
            v, _ = _pad_sets([s["valid"] for s in sets], cap, 0, np.uint8)
            bufs += [D.put(d), D.put(k), D.put(nd), D.put(v), D.put(n)]
        dm, dc = D.empty((P, cap), np.int32), D.empty((P,), np.int32)
        _check(L, L.plh_orb_search_by_bow_kfkf_batch_dev(*[_p(x) for x in bufs], cap, P, self.TH_LOW, self.mfNNratio,
                                                         int(self.mbCheckOrientation), _p(dm), _p(dc), C.c_void_p(D.stream())),
               "plh_orb_search_by_bow_kfkf_batch_dev")
        return D.get(dm), D.get(dc)

    def SearchForTriangulationBatch(self, kf1_sets, kf2_sets, F12, epipole, scale_factors2, level_sigma2_2):
        """SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, false) (ORBmatcher.cc:720-912) for P pairs sharing one
        geometry.  Each set = dict(desc, kps[KP_DTYPE], node, has_mp).  Returns (matches12[P, cap], nmatches[P])."""
        P = len(kf1_sets)
        cap = max(1, max(len(s["desc"]) for s in list(kf1_sets) + list(kf2_sets)))
        D, L = self.D, self.lib
        bufs = []
        for sets in (kf1_sets, kf2_sets):
            d, n = _pad_sets([s["desc"] for s in sets], cap, 32, np.uint8)
            k, _ = _pad_records([s["kps"] for s in sets], cap, KP_DTYPE)
            nd, _ = _pad_sets([s["node"] for s in sets], cap, 0, np.int32)
            v, _ = _pad_sets([s["has_mp"] for s in sets], cap, 0, np.uint8)
            bufs += [D.put(k), D.put(d), D.put(nd), D.put(v), D.put(n)]
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        sf = np.ascontiguousarray(scale_factors2, np.float32)
        s2 = np.ascontiguousarray(level_sigma2_2, np.float32)
        dm, dc = D.empty((P, cap), np.int32), D.empty((P,), np.int32)
        _check(L, L.plh_orb_search_for_triangulation_batch_dev(*[_p(x) for x in bufs], cap, P, _p(F), float(epipole[0]),
                                                               float(epipole[1]), _p(sf), _p(s2), len(sf), self.TH_LOW,
                                                               int(self.mbCheckOrientation), _p(dm), _p(dc),
                                                               C.c_void_p(D.stream())),
               "plh_orb_search_for_triangulation_batch_dev")
        return D.get(dm), D.get(dc)

    def SearchByBoW(self, kf, frame):
        """Returns (nmatches, matches21[N_F]): for each Frame feature the matched KeyFrame feature (the reference
        stores that feature's MapPoint*) or -1."""
        m, c = self.SearchByBoWBatch([kf], [frame])
        return int(c[0]), m[0, :len(frame["desc"])].copy()


def bow_transform(descs, vocab, levelsup=4, device=0, lib=None):
    """Frame::ComputeBoW's per-feature part: DBoW2 transform of every descriptor set in `descs`
    (list of [n,32] u8).  Returns (nid[P,cap], word[P,cap]) with -1 for unused rows / stopped words."""
    L = load(lib)
    D = _Dev(L, device)
    P = len(descs)
    cap = max(1, max(len(d) for d in descs))
    a, n = _pad_sets(descs, cap, 32, np.uint8)
    da, dn = D.put(a), D.put(n)
    nd, cs, cc, wi, wt = vocab.device_arrays(D)
    dnid, dword = D.empty((P, cap), np.int32), D.empty((P, cap), np.int32)
    _check(L, L.plh_bow_transform_batch_dev(_p(da), _p(dn), cap, P, _p(nd), _p(cs), _p(cc), _p(wi), _p(wt), vocab.L, levelsup,
                                            _p(dnid), _p(dword), C.c_void_p(D.stream())), "plh_bow_transform_batch_dev")
    return D.get(dnid), D.get(dword)


class GatherBlock(C.Structure):
    _fields_ = [("send", C.c_void_p), ("recv", C.c_void_p), ("bytes", C.c_size_t)]


class Comm:
    """One communicator per process / GPU for collecting the records of a sharded frame batch (plh_comm_*, RCCL inside)."""

    def __init__(self, unique_id, rank, world, device=0, lib=None):
        self.lib = load(lib)
        self.rank, self.world = rank, world
        uid = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        h = C.c_void_p()
        _check(self.lib, self.lib.plh_comm_create(uid, rank, world, device, C.byref(h)), "plh_comm_create")
        self.h = h

    @staticmethod
    def unique_id(lib=None):
        L = load(lib)
        uid = (C.c_uint8 * 128)()
        _check(L, L.plh_comm_unique_id(uid), "plh_comm_unique_id")
        return bytes(uid)

    def rccl_version(self):
        v = C.c_int(0)
        _check(self.lib, self.lib.plh_comm_info(self.h, None, None, C.byref(v)), "plh_comm_info")
        return v.value

    def gather(self, pairs, root=-1, stream=0):
        """pairs: [(send tensor / array, recv tensor / array or None)]; one grouped launch on `stream`."""
        blocks = (GatherBlock * len(pairs))()
        for i, (snd, rcv) in enumerate(pairs):
            nbytes = snd.numel() * snd.element_size() if hasattr(snd, "numel") else snd.nbytes
            blocks[i] = GatherBlock(_p(snd), _p(rcv) if rcv is not None else None, nbytes)
        _check(self.lib, self.lib.plh_gather_records(self.h, blocks, len(pairs), root, C.c_void_p(stream)), "plh_gather_records")

    def close(self):
        if getattr(self, "h", None):
            self.lib.plh_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VocabInfo(C.Structure):
    _fields_ = [("k", C.c_int32), ("L", C.c_int32), ("scoring", C.c_int32), ("weighting", C.c_int32), ("n_nodes", C.c_int32),
                ("n_words", C.c_int32), ("identity_ids", C.c_int32)]


class ORBVocabulary:
    """ORB_SLAM2::ORBVocabulary (DBoW2 TemplatedVocabulary<FORB::TDescriptor, FORB>, include/ORBVocabulary.h:30-31) held by the
    C library: loadFromTextFile / loadFromBinaryFile / saveToBinaryFile are the reference's file formats
    (TemplatedVocabulary.h:1350-1536), `transform` is Frame::ComputeBoW's call (Frame.cc:906-913) on a batch of
    descriptor sets and returns both the FeatureVector node per feature and the BowVector."""

    def __init__(self, device=0, lib=None):
        self.lib = load(lib)
        self.device = device
        self.h = None

    def _set(self, h):
        self.close()
        self.h = h
        self.info = VocabInfo()
        _check(self.lib, self.lib.plh_vocab_get_info(self.h, C.byref(self.info)), "plh_vocab_get_info")
        return True

    def loadFromTextFile(self, path):
        h = C.c_void_p()
        _check(self.lib, self.lib.plh_vocab_load_text(os.fsencode(path), self.device, C.byref(h)), "plh_vocab_load_text")
        return self._set(h)

    def loadFromBinaryFile(self, path):
        h = C.c_void_p()
        _check(self.lib, self.lib.plh_vocab_load_binary(os.fsencode(path), self.device, C.byref(h)), "plh_vocab_load_binary")
        return self._set(h)

    def create(self, k, L, parent, is_leaf, node_desc, weight, scoring=0, weighting=0):
        parent = np.ascontiguousarray(parent, np.int32)
        is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        node_desc = np.ascontiguousarray(node_desc, np.uint8)
        weight = np.ascontiguousarray(weight, np.float64)
        h = C.c_void_p()
        _check(self.lib, self.lib.plh_vocab_create(k, L, scoring, weighting, len(parent), _p(parent), _p(is_leaf), _p(node_desc),
                                                   _p(weight), self.device, C.byref(h)), "plh_vocab_create")
        return self._set(h)

    def saveToBinaryFile(self, path):
        _check(self.lib, self.lib.plh_vocab_save_binary(self.h, os.fsencode(path)), "plh_vocab_save_binary")

    def size(self):
        return int(self.info.n_words)

    def arrays(self):
        """Host copies of the flat tree: dict(node_desc, child_start, child_count, word_id, weight (f64), node_id)."""
        n = int(self.info.n_nodes)
        a = dict(node_desc=np.zeros((n, 32), np.uint8), child_start=np.zeros(n, np.int32), child_count=np.zeros(n, np.int32),
                 word_id=np.zeros(n, np.int32), weight=np.zeros(n, np.float64), node_id=np.zeros(n, np.int32))
        _check(self.lib, self.lib.plh_vocab_read(self.h, _p(a["node_desc"]), _p(a["child_start"]), _p(a["child_count"]),
                                                 _p(a["word_id"]), _p(a["weight"]), _p(a["node_id"])), "plh_vocab_read")
        return a

    def transform(self, descs, levelsup=4):
        """descs: list of [n,32] u8.  Returns (nid[P,cap], word[P,cap], bow): bow[p] = (word ids, values f64) in map order."""
        L, D = self.lib, _Dev(self.lib, self.device)
        P = len(descs)
        cap = max(1, max(len(d) for d in descs))
        a, n = _pad_sets(descs, cap, 32, np.uint8)
        da, dn = D.put(a), D.put(n)
        dnid, dword, dbw = D.empty((P, cap), np.int32), D.empty((P, cap), np.int32), D.empty((P, cap), np.int32)
        dbv, dbn = D.empty((P, cap), np.float64), D.empty((P,), np.int32)
        _check(L, L.plh_vocab_transform_batch_dev(self.h, _p(da), _p(dn), cap, P, levelsup, _p(dnid), _p(dword), _p(dbw), _p(dbv),
                                                  _p(dbn), C.c_void_p(D.stream())), "plh_vocab_transform_batch_dev")
        bw, bv, bn = D.get(dbw), D.get(dbv), D.get(dbn)
        return D.get(dnid), D.get(dword), [(bw[i, :bn[i]].copy(), bv[i, :bn[i]].copy()) for i in range(P)]

    def close(self):
        if getattr(self, "h", None):
            self.lib.plh_vocab_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LINEextractor:
    """ORB_SLAM2::LINEextractor(numOctaves, scale, nLSDFeature, min_line_length) on the GPU
    (reference include/LineExtractor.h:20-62).  `__call__(image, mask)` is operator()
    (src/LineExtractor.cpp:26-93): returns (keylines[KL_DTYPE], descriptors[n,32] u8, lineVec2d[n,3] f64)."""

    def __init__(self, numOctaves, scale, nLSDFeature, min_line_length, rows=480, cols=640, max_batch=1, device=0,
                 lib=None, K=None, D=None):
        self.lib = load(lib)
        self.params = LineParams(numOctaves, scale, nLSDFeature, min_line_length)
        self.rows, self.cols, self.max_batch, self.device = rows, cols, max_batch, device
        h = C.c_void_p()
        _check(self.lib, self.lib.plh_line_create(C.byref(self.params), device, rows, cols, max_batch, C.byref(h)),
               "plh_line_create")
        self.h = h
        self.capacity = self.lib.plh_line_capacity(self.h)
        # the reference's scale tables (LineExtractor.cpp:7-23); numOctaves is 1 on this path
        self.mvScaleFactor = np.ones(numOctaves, np.float32)
        for i in range(1, numOctaves):
            self.mvScaleFactor[i] = np.float32(self.mvScaleFactor[i - 1] * np.float32(scale))
        if K is not None:
            self.set_undistort(K, D)

    def close(self):
        if getattr(self, "h", None):
            self.lib.plh_line_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def GetLevels(self):
        return int(self.params.num_octaves)

    def GetScaleFactor(self):
        return float(self.params.scale)

    def GetScaleFactors(self):
        return self.mvScaleFactor.copy()

    def GetInverseScaleFactors(self):
        return (np.float32(1.0) / self.mvScaleFactor).astype(np.float32)

    def GetScaleSigmaSquares(self):
        return (self.mvScaleFactor * self.mvScaleFactor).astype(np.float32)

    def GetInverseScaleSigmaSquares(self):
        return (np.float32(1.0) / (self.mvScaleFactor * self.mvScaleFactor)).astype(np.float32)

    def set_undistort(self, K, D):
        """Frame.cc:220-222: undistort the grey image in front of LSD (maps built once, not per frame)."""
        K = np.ascontiguousarray(K, np.float32)
        D = np.ascontiguousarray(D if D is not None else np.zeros(5), np.float32)
        _check(self.lib, self.lib.plh_line_set_undistort(self.h, _p(K), _p(D)), "plh_line_set_undistort")

    def __call__(self, image, mask=None):
        image = np.asarray(image)
        if image.size == 0:
            return np.zeros(0, KL_DTYPE), np.zeros((0, 32), np.uint8), np.zeros((0, 3), np.float64)
        assert image.dtype == np.uint8 and image.ndim == 2, "image.type() == CV_8UC1"
        image = np.ascontiguousarray(image)
        if mask is not None and np.asarray(mask).size:
            mask = np.ascontiguousarray(mask, np.uint8)
            if mask.shape != image.shape:
                raise PlhError("Mask error while detecting lines: please check its dimensions and that data type is CV_8UC1")
        else:
            mask = None
        cap = self.capacity
        kl = np.zeros(cap, KL_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        fn = np.zeros((cap, 3), np.float64)
        n = C.c_int(0)
        _check(self.lib, self.lib.plh_line_extract(self.h, _p(image), image.shape[0], image.shape[1], image.strides[0], _p(mask),
                                                   _p(kl), _p(desc), _p(fn), cap, C.byref(n)), "plh_line_extract")
        return kl[:n.value].copy(), desc[:n.value].copy(), fn[:n.value].copy()

    def extract_batch_dev(self, d_imgs, batch, frame_stride, d_keylines, d_desc, d_linefn, d_n, stream=0, d_mask=None):
        _check(self.lib, self.lib.plh_line_extract_batch_dev(self.h, _p(d_imgs), batch, frame_stride, _p(d_mask), _p(d_keylines),
                                                             _p(d_desc), _p(d_linefn), _p(d_n), C.c_void_p(stream)),
               "plh_line_extract_batch_dev")

    def set_refine(self, level):
        """cv::LineSegmentDetector's refine level: 0 = LSD_REFINE_STD (default), 1 = LSD_REFINE_ADV (NFA-validated rectangles)."""
        _check(self.lib, self.lib.plh_line_set_refine(self.h, int(level)), "plh_line_set_refine")

    def set_screen(self, on):
        """Density screen of region growing (default on); off = the exact rectangle for every decision.  Same segments."""
        _check(self.lib, self.lib.plh_line_set_screen(self.h, int(bool(on))), "plh_line_set_screen")

    def set_grow_tuning(self, run_ahead=0, drain_gap=0):
        """Schedule of the several-wavefronts-per-frame region growing (<= 0: defaults 448 / 8).  Same segments."""
        _check(self.lib, self.lib.plh_line_set_grow_tuning(self.h, int(run_ahead), int(drain_gap)), "plh_line_set_grow_tuning")

    def set_grow_waves(self, waves):
        """Wavefronts per frame of LSD's region growing: -1 automatic (by batch size), 0 one, 2..16 that many; same segments."""
        _check(self.lib, self.lib.plh_line_set_grow_waves(self.h, int(waves)), "plh_line_set_grow_waves")

    def status(self):
        """Capacity flags of the most recent extract call (0 = nothing was truncated); waits for that call."""
        f = C.c_int(0)
        _check(self.lib, self.lib.plh_line_status(self.h, C.byref(f)), "plh_line_status")
        return f.value

    def read_segments(self, b=0, cap=20000):
        out = np.zeros((cap, 4), np.float32)
        n = C.c_int(0)
        _check(self.lib, self.lib.plh_line_read_segments(self.h, b, _p(out), cap, C.byref(n)), "plh_line_read_segments")
        return out[:min(n.value, cap)].copy()
