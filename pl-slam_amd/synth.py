"""Deterministic synthetic frames for parity tests and bench.py (sets S1-S4).

No dataset images exist in the reference tree or in this image, so every frame is generated:
mid-grey background, filled rectangles (axis-aligned and rotated), line strokes, additive
uniform noise, one 3x3 box blur.  PRNG = SplitMix64 (vectorised, so results are identical
everywhere numpy runs).  The frames carry enough corners / straight edges that the 1000-ORB /
200-line caps bind, as a real TUM / KITTI frame would.
"""
import numpy as np


class SplitMix64:
    def __init__(self, seed):
        self.state = np.uint64(seed)

    def u64(self, n):
        if n == 0:
            return np.zeros(0, np.uint64)
        with np.errstate(over="ignore"):
            idx = np.arange(1, n + 1, dtype=np.uint64)
            s = self.state + idx * np.uint64(0x9E3779B97F4A7C15)
            self.state = s[-1]
            z = s
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        return z

    def uniform(self, n, lo=0.0, hi=1.0):
        return lo + (hi - lo) * ((self.u64(n) >> np.uint64(11)).astype(np.float64) / float(1 << 53))

    def randint(self, n, lo, hi):  # [lo, hi)
        return lo + (self.u64(n) % np.uint64(hi - lo)).astype(np.int64)


def _fill_quad(img, cx, cy, hw, hh, ang, val):
    """Fill a rotated rectangle (centre, half sizes, angle) with `val`."""
    h, w = img.shape
    c, s = np.cos(ang), np.sin(ang)
    rad = int(np.ceil(np.hypot(hw, hh))) + 1
    x0, x1 = max(0, int(cx) - rad), min(w, int(cx) + rad + 1)
    y0, y1 = max(0, int(cy) - rad), min(h, int(cy) + rad + 1)
    if x0 >= x1 or y0 >= y1:
        return
    yy, xx = np.mgrid[y0:y1, x0:x1]
    dx, dy = xx - cx, yy - cy
    u = dx * c + dy * s
    v = -dx * s + dy * c
    m = (np.abs(u) <= hw) & (np.abs(v) <= hh)
    img[y0:y1, x0:x1][m] = val


def make_frame(seed, rows=480, cols=640, n_rect=400, n_line=200):
    """One u8 frame (rows x cols).  Set S1 = seed 1; S2 = seeds 2..65; S4 = seeds 1000.. at 376x1241."""
    rng = SplitMix64(seed)
    img = np.full((rows, cols), 128.0, dtype=np.float64)
    cx = rng.uniform(n_rect, 0, cols)
    cy = rng.uniform(n_rect, 0, rows)
    sw = rng.uniform(n_rect, 8, 120)
    sh = rng.uniform(n_rect, 8, 120)
    rot = rng.uniform(n_rect, 0, np.pi)
    axis = rng.randint(n_rect, 0, 2)
    val = rng.randint(n_rect, 0, 256)
    for i in range(n_rect):
        _fill_quad(img, cx[i], cy[i], sw[i] / 2, sh[i] / 2, 0.0 if axis[i] else rot[i], float(val[i]))
    lx = rng.uniform(n_line, 0, cols)
    ly = rng.uniform(n_line, 0, rows)
    ll = rng.uniform(n_line, 20, 200)
    la = rng.uniform(n_line, 0, np.pi)
    lw = rng.randint(n_line, 1, 4)
    lv = rng.randint(n_line, 0, 256)
    for i in range(n_line):
        _fill_quad(img, lx[i], ly[i], ll[i] / 2, lw[i] / 2.0, la[i], float(lv[i]))
    noise = rng.randint(rows * cols, -6, 7).reshape(rows, cols)
    img = np.clip(img + noise, 0, 255)
    # 3x3 box blur (edge-replicated), integer rounding
    p = np.pad(img, 1, mode="edge")
    acc = sum(p[dy:dy + rows, dx:dx + cols] for dy in range(3) for dx in range(3))
    return np.floor(acc / 9.0 + 0.5).astype(np.uint8)


def make_frames(seed0, count, rows=480, cols=640, unique=None, first=0, n=None):
    """`count` frames.  With `unique` < count only that many are rasterised; the rest are cheap,
    distinct variants (cyclic shift of rows + small brightness offset) so no two frames are equal.
    first / n: only frames [first, first + n) of the `count` are made (a rank's shard of one job: same frames as the whole)."""
    unique = count if unique is None else max(1, min(unique, count))
    reps = -(-count // unique)
    n = count - first if n is None else n
    need = sorted({i // reps for i in range(first, first + n)})
    made = {u: make_frame(seed0 + u, rows, cols) for u in need}
    base = [made.get(u) for u in range(unique)]
    out = np.empty((n, rows, cols), dtype=np.uint8)
    for j in range(n):
        i = first + j
        # consecutive indices are consecutive "camera poses" of the same scene (3-row shift + exposure change),
        # so frame-to-frame matching has real correspondences
        b = base[i // reps]
        k = i % reps
        if k == 0:
            out[j] = b
        else:
            out[j] = np.clip(np.roll(b, (3 * k) % rows, axis=0).astype(np.int16) + (k % 7) - 3, 0, 255).astype(np.uint8)
    return out


def make_descriptor_sets(seed, n, flip_p=0.08):
    """Set S3: A = n random 256-bit rows; B = A with each bit flipped w.p. flip_p, rows permuted.
    Returns (A, B, perm) with B[i] derived from A[perm[i]]."""
    rng = SplitMix64(seed)
    a = rng.u64(n * 4).view(np.uint8).reshape(n, 32).copy()
    flips = (rng.uniform(n * 256) < flip_p).reshape(n, 256)
    fb = np.packbits(flips, axis=1, bitorder="little")
    key = rng.u64(n)
    perm = np.argsort(key, kind="stable")
    b = (a ^ fb)[perm]
    return a, np.ascontiguousarray(b), perm
