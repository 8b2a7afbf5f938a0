"""Host logic of the XCD-aware block order (pl-slam_amd/csrc/plh_xcd.h, plh_common.h: plh_xcd_make / plh_xcd_grid on the host,
plh_udiv_magic / plh_xcd_decode[_tiles] in the kernels), restated in numpy: the division by a launch constant through floor(2^32 / d)
is exact after one fix-up for every 32-bit dividend, and the decode visits every (block of a frame, frame) pair exactly once -- ragged
last groups of eight, batches below the threshold and one-block frames included.  (The kernels themselves are held to the oracle with
batches on both sides of the threshold: tests/test_line.py::test_emu_batch_of_ten_xcd_block_order on the emulator,
tests/test_e2e_gpu.py::test_ragged_group_of_eight_xcd_block_order and the 256 .. 1024-frame tests on the GPU.)"""
import re
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _min_batch():
    src = open(os.path.join(ROOT, "pl-slam_amd", "csrc", "plh_xcd.h")).read()
    return [int(v) for v in re.findall(r"constexpr int PLH_XCD_MIN_BATCH = (\d+);", src)]


def magic(d):
    return 0xFFFFFFFF if d <= 1 else (1 << 32) // d


def udiv_magic(q, d, m):
    """plh_udiv_magic: the estimate (q m) >> 32 and ONE fix-up"""
    q = np.asarray(q, np.uint64)
    b = (q * np.uint64(m)) >> np.uint64(32)
    x = q - b * np.uint64(d)
    fix = x >= np.uint64(d)
    return np.where(fix, b + np.uint64(1), b), np.where(fix, x - np.uint64(d), x)


def test_thresholds_in_the_header():
    assert sorted(_min_batch()) == [8, 64]   # emulator build / product build


@pytest.mark.parametrize("d", [1, 2, 3, 7, 12, 201, 480, 1307, 4095, 4096, 65535, 1 << 20, (1 << 31) - 1])
def test_magic_division_is_exact(d):
    rng = np.random.default_rng(d)
    q = np.concatenate([rng.integers(0, 1 << 32, 200000, dtype=np.uint64), np.arange(0, 5000, dtype=np.uint64),
                        np.uint64((1 << 32) - 1) - np.arange(0, 5000, dtype=np.uint64),
                        (np.arange(1, 3000, dtype=np.uint64) * np.uint64(d))[np.arange(1, 3000, dtype=np.uint64) * np.uint64(d) < (1 << 32)] - np.uint64(1)])
    b, x = udiv_magic(q, d, magic(d))
    assert (b == q // np.uint64(d)).all() and (x == q % np.uint64(d)).all()


def decode(L, per_frame, nx, batch, min_batch):
    if batch < min_batch:
        b, x = udiv_magic(L, per_frame, magic(per_frame))
    else:
        g, x = udiv_magic(L >> 3, per_frame, magic(per_frame))
        b = g * np.uint64(8) + (np.asarray(L, np.uint64) & np.uint64(7))
    by, bx = udiv_magic(x, nx, magic(nx))
    return b.astype(np.int64), x.astype(np.int64), bx.astype(np.int64), by.astype(np.int64)


@pytest.mark.parametrize("min_batch", [8, 64])
@pytest.mark.parametrize("nx,ny,batch", [(1, 1, 1), (5, 3, 1), (5, 3, 7), (5, 3, 8), (5, 3, 9), (10, 30, 63), (10, 30, 64), (10, 30, 67),
                                        (1, 480, 80), (201, 1, 1536), (1, 1, 65), (7, 1, 6144)])
def test_every_block_of_every_frame_exactly_once(nx, ny, batch, min_batch):
    per = nx * ny
    grid = per * batch if batch < min_batch else per * ((batch + 7) // 8) * 8   # plh_xcd_grid
    L = np.arange(grid, dtype=np.uint64)
    b, x, bx, by = decode(L, per, nx, batch, min_batch)
    live = b < batch
    assert live.sum() == per * batch
    seen = np.zeros((batch, per), np.int32)
    np.add.at(seen, (b[live], x[live]), 1)
    assert (seen == 1).all()
    assert (bx[live] < nx).all() and (by[live] < ny).all() and (by[live] * nx + bx[live] == x[live]).all()
    if batch >= min_batch:   # block L runs on XCD L % 8: a frame's blocks are all on one
        xcd = (L & np.uint64(7)).astype(np.int64)
        assert (xcd[live] == b[live] % 8).all()
