"""The layout of the level-line record plane (pl-slam_amd/csrc/line_plan.h, lsd_rec_index): 4 x 4-pixel blocks of 64 bytes.
Host logic, no GPU: the header's own function is compiled with g++ and checked for what the kernels rely on --
a bijection onto the plane, whole blocks per 64-byte sector, four pixels of a row (x a multiple of 4) contiguous, and the
null pixel (sw - 1, 0) inside the plane."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cstdio>
#include <vector>
#include "line_plan.h"
int main() {
  const int shapes[][3] = {{512, 512, 384}, {1024, 993, 301}, {64, 1, 1}, {128, 100, 7}, {256, 256, 96}, {192, 161, 123}};
  for (const auto& sp : shapes) {
    const int spitch = sp[0], sw = sp[1], sh = sp[2], rows = plh::lsd_rec_rows(sh);
    if (rows % 4 || rows < sh || rows >= sh + 4) { std::printf("rows %d for %d\n", rows, sh); return 1; }
    std::vector<unsigned char> seen((size_t)spitch * rows, 0);
    for (int y = 0; y < rows; y++)
      for (int x = 0; x < spitch; x++) {
        const uint32_t i = plh::lsd_rec_index((uint32_t)x, (uint32_t)y, (uint32_t)spitch);
        if (i >= (uint32_t)(spitch * rows) || seen[i]) { std::printf("not a bijection at %d %d (%d)\n", x, y, spitch); return 2; }
        seen[i] = 1;
        // a block (x / 4, y / 4) is 16 consecutive words starting at a multiple of 16: one 64-byte sector
        const uint32_t b0 = plh::lsd_rec_index((uint32_t)(x & ~3), (uint32_t)(y & ~3), (uint32_t)spitch);
        if (b0 % 16 || i < b0 || i >= b0 + 16) { std::printf("block broken at %d %d\n", x, y); return 3; }
        // four pixels of a row are contiguous (16-byte loads / stores of k_lsd_grad, k_lsd_bin_hist)
        if (x % 4 == 0 && plh::lsd_rec_index((uint32_t)x + 3, (uint32_t)y, (uint32_t)spitch) != i + 3) { std::printf("row run broken\n"); return 4; }
        // two blocks side by side are one 128-byte line
        if (x % 8 == 0 && y % 4 == 0 && (i % 32 || plh::lsd_rec_index((uint32_t)x + 4, (uint32_t)y, (uint32_t)spitch) != i + 16)) { std::printf("line broken\n"); return 5; }
      }
    if (plh::lsd_rec_index((uint32_t)(sw - 1), 0u, (uint32_t)spitch) >= (uint32_t)(spitch * rows)) return 6;
  }
  std::printf("ok\n");
  return 0;
}
"""


def test_record_plane_layout(tmp_path):
    src = tmp_path / "rec.cc"
    src.write_text(SRC)
    exe = tmp_path / "rec"
    inc = os.path.join(ROOT, "pl-slam_amd", "csrc")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-I", inc, "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.returncode, r.stdout)
