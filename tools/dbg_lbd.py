"""GPU debug: line descriptors vs the oracle, count differing rows / bits."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import _util
plslam, oracle, synth = _util.plslam(), _util.oracle(), _util.synth()
for seed, rows, cols in ((1, 480, 640), (3, 240, 320)):
    img = synth.make_frame(seed, rows, cols)
    rk, rd, rf = oracle.line_extract(img, 200, 0.0)
    ex = plslam.LINEextractor(1, 1.2, 200, 0.0, rows=rows, cols=cols, max_batch=1)
    kl, desc, fn = ex(img)
    ex.close()
    n = min(len(kl), len(rk))
    same_kl = all((kl[f][:n] == rk[f][:n]).all() for f in kl.dtype.names)
    diff = np.unpackbits(desc[:n] ^ rd[:n], axis=1)
    rowsbad = (diff.sum(1) > 0).sum()
    print("seed", seed, "n", len(kl), len(rk), "keylines equal", same_kl, "rows differing", rowsbad, "bits", diff.sum(),
          "first bad rows", np.nonzero(diff.sum(1))[0][:10], "bits per bad row", diff.sum(1)[diff.sum(1) > 0][:10])
    bad = np.nonzero(diff.sum(1))[0]
    if len(bad):
        i = bad[0]
        print(" row", i, "numOfPixels", kl["numOfPixels"][i], "angle", kl["angle"][i], "len", kl["lineLength"][i])
        print(" gpu", desc[i]); print(" ref", rd[i])
