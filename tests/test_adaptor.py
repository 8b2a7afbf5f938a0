"""The C++ adaptor headers (reference class signatures over the C ABI) must at least parse and type-check.
OpenCV / Eigen are absent from the image, so they are checked against tests/cv_stub (declarations only)."""
import os
import subprocess

import pytest

import _util

HDRS = ["ORBextractor.h", "LineExtractor.h", "HipMatchers.h"]


@pytest.mark.parametrize("hdr", HDRS)
def test_adaptor_header_typechecks(hdr, tmp_path):
    src = tmp_path / "t.cc"
    src.write_text('#include "%s"\n'
                   'template <class T> void use(T*) {}\n'
                   'void f() { use((ORB_SLAM2::ORBextractor*)0); }\n' if hdr == "ORBextractor.h" else '#include "%s"\n' % hdr)
    if hdr == "ORBextractor.h":
        src.write_text('#include "ORBextractor.h"\n'
                       'void f(cv::Mat& im, std::vector<cv::KeyPoint>& k, cv::Mat& d) {\n'
                       '  ORB_SLAM2::ORBextractor e(1000, 1.2f, 8, 20, 7);\n'
                       '  e(im, cv::Mat(), k, d);            // the call Frame::ExtractORB makes (Frame.cc:325)\n'
                       '  std::vector<float> s = e.GetScaleFactors(); (void)s; (void)e.GetLevels();\n'
                       '}\n')
    elif hdr == "LineExtractor.h":
        src.write_text('#include "LineExtractor.h"\n'
                       'void f(cv::Mat& im, cv::Mat& mask, std::vector<cv::line_descriptor::KeyLine>& k, cv::Mat& d,\n'
                       '       std::vector<Eigen::Vector3d>& fn) {\n'
                       '  ORB_SLAM2::LINEextractor e(1, 1.2f, 200, 0.0);\n'
                       '  e(im, mask, k, d, fn);             // the call Frame::ExtractLSD makes (Frame.cc:333)\n'
                       '}\n')
    else:
        src.write_text('#include "HipMatchers.h"\n'
                       'int f(cv::Mat& a, cv::Mat& b, std::vector<int>& m) { return ORB_SLAM2::hip::SearchDouble(a, b, m, 0.7f); }\n'
                       'int g(std::vector<cv::KeyPoint>& k1, std::vector<cv::KeyPoint>& k2, cv::Mat& a, cv::Mat& b,\n'
                       '      std::vector<cv::Point2f>& prev, std::vector<int>& m) {\n'
                       '  plh_grid_params gp = ORB_SLAM2::hip::GridParams(0, 0, 640, 480, 0.1f, 0.1f);\n'
                       '  return ORB_SLAM2::hip::SearchForInitialization(k1, a, k2, b, gp, prev, m, 100, 0.9f, true); }\n'
                       'int h(std::vector<cv::KeyPoint>& k, cv::Mat& d, std::vector<float>& sf, std::vector<uchar>& occ,\n'
                       '      ORB_SLAM2::hip::ProjQueries& q, std::vector<int>& asg, std::vector<cv::line_descriptor::KeyLine>& kl,\n'
                       '      std::vector<Eigen::Vector3d>& fn) {\n'
                       '  plh_grid_params gp = ORB_SLAM2::hip::GridParams(0, 0, 640, 480, 0.1f, 0.1f);\n'
                       '  return ORB_SLAM2::hip::SearchByProjection(k, d, gp, sf, occ, q, 1.f, 0.8f, asg) +\n'
                       '         ORB_SLAM2::hip::SearchByProjectionLastFrame(k, d, gp, sf, occ, q, 15.f, 0, true, asg) +\n'
                       '         ORB_SLAM2::hip::LineSearchByProjection(kl, d, fn, gp, occ, q, 8.f, 0.7f, true, asg); }\n')
    inc = [os.path.join(_util.ROOT, "pl-slam_amd", "adaptor"), os.path.join(_util.ROOT, "include"),
           os.path.join(_util.ROOT, "tests", "cv_stub")]
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-DPLH_LSD_REFINE_DEFAULT=1"] + ["-I" + i for i in inc] + [str(src)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    if hdr == "LineExtractor.h":   # the drop-in does not pick cv::LineSegmentDetector's refine level silently: the build must
        cmd.remove("-DPLH_LSD_REFINE_DEFAULT=1")
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode != 0 and "PLH_LSD_REFINE_DEFAULT" in r.stderr


def test_library_exports_every_declared_symbol(plslam):
    """CPU: the C-ABI library loads and exports every entry point include/plslam_hip.h declares (no compute calls)."""
    import ctypes
    lib_path = os.path.join(_util.ROOT, "pl-slam_amd", "libplslam_hip.so")
    if not os.path.exists(lib_path):
        import sys
        sys.path.insert(0, _util.ROOT)
        import __graft_entry__ as g
        g.build_hip()
    lib = ctypes.CDLL(lib_path)
    syms = plslam.exported_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert b"gfx950" in ctypes.cast(ctypes.CDLL(lib_path).plh_version, ctypes.CFUNCTYPE(ctypes.c_char_p))()


def test_every_c_entry_point_is_declared(plslam, emu_lib):
    """Every extern "C" plh_* function of the sources (the emulation build exports them all) is declared in
    include/plslam_hip.h -- an undeclared one would be hidden (-fvisibility=hidden) in the product library."""
    out = subprocess.run(["nm", "-D", "--defined-only", emu_lib], capture_output=True, text=True, check=True).stdout
    defined = {l.split()[-1] for l in out.splitlines() if l.split()[-1].startswith("plh_") and " T " in l}
    declared = set(plslam.exported_symbols())
    internal = {"plh_debug_grow_prof"}
    assert not (defined - declared - internal), sorted(defined - declared - internal)
