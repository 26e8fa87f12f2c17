"""Pin the host-side coarse edit (re_edit_2d = getRotationMatrix2D + warpAffine, nearest mask resize) on the ONE input -> output
set the reference tree holds: /root/reference/Examples/Editing/2D/tower/{source, source_mask, target_mask, coarse_result}.png, the coarse 2-D edit
of the reference's own demo (/root/reference/src/utils/vis_utils.py:210-274).  Build container only (reads /root/reference).

1. recover the edit parameters (dx, dy, rz, sx, sy) that map source_mask (640 x 640, resized to the image's 512 x 512 the way read_and_resize_mask
   does: cv2.INTER_NEAREST) onto target_mask: bounding boxes give the start, a local search over translation / rotation / scale confirms that no
   other parameter set reproduces the mask exactly;
2. check that this repository's re_edit_2d reproduces target_mask EXACTLY and coarse_result inside it;
3. write tests/golden/g12_tower_coarse_edit.npz: the two masks bit-packed at full size, the parameters, and a 114-column crop of source /
   coarse_result that holds a 64-column band of the object before and after the move (data only: inputs and expected outputs).
"""
import itertools
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from src.utils import vis_utils as V  # noqa: E402

SRC = "/root/reference/Examples/Editing/2D/tower"


def main():
    src = np.asarray(Image.open(f"{SRC}/source.png").convert("RGB"))
    sm640 = np.asarray(Image.open(f"{SRC}/source_mask.png"))
    tm = np.asarray(Image.open(f"{SRC}/target_mask.png"))
    co = np.asarray(Image.open(f"{SRC}/coarse_result.png").convert("RGB"))
    assert src.shape == (512, 512, 3) and sm640.shape == (640, 640) and tm.shape == (512, 512) and co.shape == src.shape
    sm = (V._resize_nearest(sm640, (512, 512)) > 0).astype(np.uint8)

    def bbox(m):
        ys, xs = np.where(m > 0)
        return xs.min(), xs.max(), ys.min(), ys.max()
    bs, bt = bbox(sm), bbox(tm)
    print("source mask bbox (x0, x1, y0, y1):", bs, " target:", bt)
    dx0 = int(bt[1] - bs[1])                      # the right / bottom edges are inside the frame in both
    dy0 = int(bt[3] - bs[3])
    best = []
    zero = np.zeros_like(src)
    for dx, dy, rz, s in itertools.product(range(dx0 - 2, dx0 + 3), range(dy0 - 2, dy0 + 3), (-1.0, -0.5, 0.0, 0.5, 1.0), (0.98, 1.0, 1.02)):
        _, tmask, _ = V.re_edit_2d(zero, sm, (dx, dy, rz, s, s), zero)
        best.append(((tmask != tm).sum(), dx, dy, rz, s))
    best.sort()
    print("best candidates (mismatching pixels, dx, dy, rz, s):", best[:4])
    assert best[0][0] == 0 and best[1][0] > 0, "the parameter set must be unique"
    _, dx, dy, rz, s = best[0]
    param = (float(dx), float(dy), float(rz), float(s), float(s))
    final, tmask, _ = V.re_edit_2d(src, sm, param, src)
    inside = tm > 0
    dev = np.abs(final.astype(int) - co.astype(int)).max(axis=2)
    print("edit_param", param, " target mask exact:", np.array_equal(tmask, tm), " coarse inside target: max |diff|", dev[inside].max())
    for name, m in (("PIL nearest", np.asarray(Image.fromarray(sm640).resize((512, 512), Image.NEAREST))),):
        _, t2, _ = V.re_edit_2d(zero, (m > 0).astype(np.uint8), param, zero)
        print(f"(with the source mask resized by {name} instead of cv2's floor rule: {(t2 != tm).sum()} mismatching pixels)")
    x0, x1 = 250, 364                              # holds the object band [300, 364) and where it lands, [250, 314)
    out = os.path.join(ROOT, "tests", "golden", "g12_tower_coarse_edit.npz")
    np.savez_compressed(out, source_mask_640=np.packbits(sm640 > 0), target_mask_512=np.packbits(tm > 0), edit_param=np.array(param),
                        crop=np.array([x0, x1]), source_crop=src[:, x0:x1], coarse_crop=co[:, x0:x1])
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
