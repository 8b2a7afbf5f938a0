"""Real frames for the batch front end: raw 8-bit planes (.bin / .raw / .gray: rows x cols bytes per frame, any number of frames
per file) and binary PGM (P5, maxval <= 255) -- no imaging dependency.  A directory is read in sorted order.  The reference's
drivers read a dataset's image list with cv::imread (Examples/Monocular/mono_tum.cc:60-75, mono_kitti.cc:60-75); what reaches
Frame::Frame is a CV_8UC1 plane of the camera's size (Tracking.cc:237-256 converts colour to grey), which is what this returns:
convert PNG sequences once with any tool (`convert rgb/*.png -colorspace Gray frame_%05d.pgm`)."""
import os

import numpy as np

RAW_EXT = (".bin", ".raw", ".gray")


def read_pgm(path):
    """One P5 PGM -> uint8 [rows, cols]."""
    data = open(path, "rb").read()
    if data[:2] != b"P5":
        raise ValueError("%s: not a binary PGM (P5)" % path)
    tok, pos = [], 2
    while len(tok) < 3:                      # width, height, maxval; '#' comments run to the end of the line
        while pos < len(data) and data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            while pos < len(data) and data[pos:pos + 1] != b"\n":
                pos += 1
            continue
        end = pos
        while end < len(data) and not data[end:end + 1].isspace():
            end += 1
        tok.append(int(data[pos:end]))
        pos = end
    pos += 1                                 # the single whitespace byte behind maxval
    cols, rows, maxval = tok
    if not 0 < maxval <= 255:
        raise ValueError("%s: maxval %d (8-bit PGM expected)" % (path, maxval))
    if len(data) - pos < rows * cols:
        raise ValueError("%s: truncated (%d of %d bytes)" % (path, len(data) - pos, rows * cols))
    return np.frombuffer(data, np.uint8, rows * cols, pos).reshape(rows, cols).copy()


def write_pgm(path, img):
    img = np.ascontiguousarray(img, np.uint8)
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


def load_frames(path, rows, cols):
    """uint8 [n, rows, cols] from a raw file, a PGM, or a directory of them (sorted).  Every frame must be rows x cols."""
    files = [path]
    if os.path.isdir(path):
        files = sorted(os.path.join(path, f) for f in os.listdir(path) if f.lower().endswith(RAW_EXT + (".pgm",)))
        if not files:
            raise ValueError("%s: no .pgm / .bin frames" % path)
    out = []
    for f in files:
        if f.lower().endswith(".pgm"):
            img = read_pgm(f)
            if img.shape != (rows, cols):
                raise ValueError("%s is %dx%d, the plan is %dx%d (pass --rows / --cols)" % (f, img.shape[1], img.shape[0], cols, rows))
            out.append(img[None])
        else:
            raw = np.fromfile(f, np.uint8)
            if raw.size == 0 or raw.size % (rows * cols):
                raise ValueError("%s: %d bytes is not a whole number of %dx%d frames" % (f, raw.size, cols, rows))
            out.append(raw.reshape(-1, rows, cols))
    return np.ascontiguousarray(np.concatenate(out, axis=0))


def tile_frames(frames, count):
    """`count` frames by cycling through the given ones (a resident batch larger than the sequence)."""
    idx = np.arange(count) % len(frames)
    return np.ascontiguousarray(frames[idx])
