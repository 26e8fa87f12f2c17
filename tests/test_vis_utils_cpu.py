"""Host pre/post-processing (src/utils/vis_utils.py, written without OpenCV) and the GeoBench harness's host logic.
cv2 cannot be imported here, so the interpolating functions are checked against independent restatements / properties, not
against cv2 output (parity unpinned, see the module header); index-exact functions are checked exactly."""
import os
import subprocess
import sys

import numpy as np
from PIL import Image
from scipy import ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from src.utils import vis_utils as V  # noqa: E402


def test_nearest_resize_and_mask_reader(tmp_path):
    rng = np.random.default_rng(0)
    m = (rng.random((37, 53)) > 0.5).astype(np.uint8) * 255
    p = tmp_path / "m.png"
    Image.fromarray(m).save(p)
    out = V.read_and_resize_mask(str(p), dsize=(64, 48))            # dsize = (w, h)
    assert out.shape == (48, 64, 3) and out.dtype == np.uint8 and set(np.unique(out)) <= {0, 1}
    for y in (0, 17, 47):
        for x in (0, 31, 63):
            assert out[y, x, 0] == (m[min(int(y * 37 / 48), 36), min(int(x * 53 / 64), 52)] > 0)
    same = V.read_and_resize_mask(str(p), dsize=(53, 37))
    assert np.array_equal(same[:, :, 0], (m > 0).astype(np.uint8))


def test_dilate_matches_bruteforce_and_constrain_area_wraps():
    rng = np.random.default_rng(1)
    m = (rng.random((40, 40)) > 0.97).astype(np.uint8)
    for k in (3, 6, 15):
        ref = np.zeros_like(m)
        a = k // 2
        for y in range(40):
            for x in range(40):
                ref[y, x] = m[max(0, y - a):min(40, y - a + k), max(0, x - a):min(40, x - a + k)].max()
        assert np.array_equal(V.dilate_mask(m, k), ref), k
    inst = [np.zeros((8, 8, 3), np.uint8) for _ in range(2)]
    inst[0][1:4, 1:4] = 255
    inst[1][5:7, 5:7] = 255
    ori = np.zeros((8, 8, 3), np.uint8)
    ori[0:3, 0:3] = 255                                   # sticks out of the union at row/col 0
    fa = V.get_constrain_areas(mask_list=inst, ori_mask=ori)
    assert fa.dtype == np.uint8 and fa[0, 0, 0] == 255 and fa[2, 2, 0] == 0 and fa[3, 3, 0] == 1 and fa[5, 5, 0] == 1
    assert set(np.unique(ori)) == {0, 1}                   # binarised in place, like the reference


def test_lanczos_resize_properties(tmp_path):
    const = np.full((33, 47, 3), 131, np.uint8)
    assert np.array_equal(V._resize_lanczos4(const, (64, 64)), np.full((64, 64, 3), 131, np.uint8))
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (40, 40, 3), dtype=np.uint8)
    assert np.array_equal(V._resize_lanczos4(img, (40, 40)), img)
    ramp = np.tile(np.linspace(20, 220, 50)[None, :, None], (50, 1, 3)).astype(np.uint8)
    up = V._resize_lanczos4(ramp, (100, 100)).astype(int)
    assert np.abs(np.diff(up[50, 10:90, 0])).max() <= 4 and abs(int(up[50, 50, 0]) - 120) <= 3
    p = tmp_path / "i.png"
    Image.fromarray(img).save(p)
    out = V.read_and_resize_img(str(p), dsize=(40, 40))
    assert np.array_equal(out, img)                        # RGB order survives the BGR round trip


def test_affine_edit_against_scipy():
    rng = np.random.default_rng(3)
    img = ndimage.gaussian_filter(rng.random((64, 64, 3)) * 255, (2, 2, 0)).astype(np.uint8)
    mask = np.zeros((64, 64), np.uint8)
    mask[20:36, 12:30] = 1
    bg = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    # identity edit: object stays, everything outside comes from the background
    fin, tm, hole = V.re_edit_2d(img, mask, (0, 0, 0, 1, 1), bg)
    assert np.array_equal(tm, mask * 255) and np.array_equal(fin[mask > 0], img[mask > 0]) and np.array_equal(fin[mask == 0], bg[mask == 0])
    assert np.array_equal(hole[mask == 0], img[mask == 0])
    # pure integer translation: exact
    fin, tm, _ = V.re_edit_2d(img, mask, (7, -5, 0, 1, 1), bg)
    assert np.array_equal(tm[15:31, 19:37], np.full((16, 18), 255, np.uint8)) and tm.sum() == 255 * 16 * 18
    assert np.array_equal(fin[15:31, 19:37], img[20:36, 12:30])
    # rotation + anisotropic scale, 9-parameter GeoBench form: compare the warp with scipy's affine_transform
    dx, dy, rz, sx, sy = 3.0, 2.0, 25.0, 1.2, 0.8
    M = V._affine_for_edit(mask, dx, dy, rz, sx, sy)
    A = np.linalg.inv(np.vstack([M, [0, 0, 1]]))          # dst -> src in (x, y)
    mat = np.array([[A[1, 1], A[1, 0]], [A[0, 1], A[0, 0]]])
    off = np.array([A[1, 2], A[0, 2]])
    ref = np.stack([ndimage.affine_transform(img[:, :, c].astype(np.float64), mat, offset=off, order=1, mode="constant", cval=0) for c in range(3)], -1)
    got = V._warp_affine(img, M, (64, 64)).astype(np.float64)
    inner = ndimage.binary_erosion(ndimage.affine_transform(np.ones((64, 64)), mat, offset=off, order=0, mode="constant", cval=0) > 0, iterations=2)
    assert np.abs(got - ref)[inner].max() <= 4.0          # 1/32-pixel coordinate quantisation on a smooth image
    fin9, tm9, _ = V.re_edit_2d(img, mask, (dx, dy, 0, 0, 0, rz, sx, sy, 1), bg)
    fin5, tm5, _ = V.re_edit_2d(img, mask, (dx, dy, rz, sx, sy), bg)
    assert np.array_equal(fin9, fin5) and np.array_equal(tm9, tm5) and 0 < (tm9 > 0).sum() < 64 * 64
    R = V._rotation_matrix_2d((10.0, 20.0), 90.0, 1.0)     # cv2.getRotationMatrix2D((10,20), 90, 1) = [[0,1,-10],[-1,0,30]]
    assert np.allclose(R, [[0, 1, -10], [-1, 0, 30]], atol=1e-12)


WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["FF_ROOT"])
from freefine_amd import geobench
class FakeModel:                      # host logic only: "edits" return the coarse input, tagged with the batch size
    def FreeFine_generation(self, ori_img, ori_mask, coarse_input, target_mask, guidance_text, guidance_scale, eta, **kw):
        assert kw["use_auto_draw"] and kw["reduce_inp_artifacts"] and kw["start_step"] == 35 and kw["seed"] == 42
        return coarse_input
    def FreeFine_generation_batch(self, cases, guidance_scale, eta, **kw):
        assert 1 < len(cases) <= 2 and kw["seeds"] == 42
        return [c["coarse_input"] for c in cases]
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
res = geobench.run(FakeModel(), os.environ["FF_DATA"], batch=2, rank=rank, world=world, verbose=False)
if rank == 0:
    print("GEO_OK", len(res))
dist.destroy_process_group()
'''


def test_geobench_harness_world2_gloo(tmp_path):
    from freefine_amd import geobench
    root = str(tmp_path / "data")
    data = geobench.make_synthetic_dataset(root, n_images=3, edits_per_image=2, size=64)
    n = sum(len(e) for d in data.values() for e in d["instances"].values())
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, FF_ROOT=ROOT, FF_DATA=root)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29531", str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert f"GEO_OK {n}" in out.stdout
    res = V.load_json(os.path.join(root, "generated_results_freefine_2d.json"))
    got = [(da, ins, ed) for da, d in res.items() for ins, e in d["instances"].items() for ed in e]
    assert len(got) == n
    for da, d in res.items():
        for ins, e in d["instances"].items():
            for ed, item in e.items():
                img = np.asarray(Image.open(item["gen_img_path"]))
                assert img.shape == (512, 512, 3) and item["edit_param"] == data[da]["instances"][ins][ed]["edit_param"]
    # second run: everything exists already -> nothing to do, JSON rebuilt from the existing results
    cl = geobench.CaseList(V.load_json(os.path.join(root, "annotations_2d.json")), os.path.join(root, geobench.GEN_SUBDIR))
    assert len(cl) == 0 and len(cl.existing_results) == n


def test_monocular_depth_wrapper_around_a_depth_network():
    """geobench.monocular_depth (the reference's get_monocular_depth_anything: resize to a multiple of 14 with the shorter side at 518, ImageNet
    normalisation, network, bilinear back, max - d, push-back) around a stand-in network that returns the normalised red channel"""
    import numpy as np
    import torch
    from freefine_amd import geobench

    seen = {}

    class Net:
        device = "cpu"

        def __call__(self, x):
            seen["shape"] = tuple(x.shape)
            return x[:, 0] * 0.229 + 0.485                      # undo the normalisation of channel 0: the image's red channel in [0, 1]

    img = np.zeros((60, 90, 3), np.uint8)
    img[:, :, 0] = np.linspace(0, 255, 90).astype(np.uint8)[None, :]     # red ramps left -> right: "disparity" grows to the right
    d = geobench.monocular_depth(img, Net(), translate_factor=0.1)
    assert seen["shape"][0] == 1 and seen["shape"][1] == 3 and seen["shape"][2] % 14 == 0 and seen["shape"][3] % 14 == 0
    assert min(seen["shape"][2:]) in (518 // 14 * 14, 518) and d.shape == (60, 90) and d.dtype == np.float32
    assert (np.diff(d.mean(0)) <= 1e-4).all()                   # depth = max - disparity: falls to the right
    assert d.min() > 0 and abs(d.min() - 0.1 * (d.max() - d.min()) / 1.0) < 0.02      # pushed back by translate_factor * max(depth before the push)


def test_images_to_u8_matches_the_reference_expression():
    """pipeline.images_to_u8 (multiply and truncating cast on the tensor's device, one contiguous copy) returns the bytes of the reference's
    `(x.permute(1, 2, 0).cpu().numpy() * 255).astype(np.uint8)` (/root/reference/src/demo/model.py:1046-1049) -- incl. the exact ends of the range"""
    import numpy as np
    import torch
    from freefine_amd.pipeline import images_to_u8
    g = torch.Generator().manual_seed(3)
    x = torch.rand(3, 3, 16, 24, generator=g)
    x[0, 0, 0, :6] = torch.tensor([0.0, 1.0, 0.999999, 1 / 255, 254.9999 / 255, 0.5])
    got = images_to_u8(x)
    assert len(got) == 3 and got[0].shape == (16, 24, 3) and got[0].dtype == np.uint8
    for k in range(3):
        ref = (x[k].permute(1, 2, 0).numpy() * 255).astype(np.uint8)
        assert (got[k] == ref).all()


def test_re_edit_2d_reproduces_the_references_tower_example():
    """N2 pinned on the one input -> output set the reference tree holds (/root/reference/Examples/Editing/2D/tower: source, source_mask, target_mask,
    coarse_result of the reference's own coarse 2-D edit; fixture written by tools/pin_n2_tower.py, which recovers edit_param = (-50, -50, 0, 1, 1)
    as the UNIQUE parameter set that maps one mask onto the other).  Pins: the cv2.INTER_NEAREST rule of read_and_resize_mask on the 640 x 640
    source mask (PIL's nearest rule misses by 437 pixels), getRotationMatrix2D / the translation conventions, the nearest mask warp, the
    paste-over-background composition -- target mask EXACT over the full frame, coarse image EXACT inside it (an integer translation: the bilinear
    weights degenerate, so rotation / scaling stay parity-unpinned and say so in src/utils/vis_utils.py)."""
    from src.utils import vis_utils as V
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g12_tower_coarse_edit.npz"))
    sm640 = np.unpackbits(g["source_mask_640"]).reshape(640, 640).astype(np.uint8) * 255
    tm = np.unpackbits(g["target_mask_512"]).reshape(512, 512).astype(np.uint8) * 255
    param = tuple(g["edit_param"].tolist())
    assert param == (-50.0, -50.0, 0.0, 1.0, 1.0)
    sm = V._resize_nearest(sm640, (512, 512))
    sm[sm > 0] = 1                                                           # read_and_resize_mask's binarisation
    zero = np.zeros((512, 512, 3), np.uint8)
    _, tmask, _ = V.re_edit_2d(zero, sm, param, zero)
    assert tmask.dtype == np.uint8 and np.array_equal(tmask, tm)             # exact, full frame (89706 object pixels, clipped at the top edge)
    x0, x1 = (int(v) for v in g["crop"])
    src_c, co_c = g["source_crop"], g["coarse_crop"]
    bg = np.full_like(src_c, 7)                                              # the inpainted background is not part of the fixture: a sentinel
    final, tmask_c, hole = V.re_edit_2d(src_c, sm[:, x0:x1], param, bg)
    band = np.zeros((512, x1 - x0), bool)
    band[:, :x1 - x0 - 50] = True                                            # columns whose source pixel (x + 50) lies inside the crop
    inside = (tm[:, x0:x1] > 0) & band
    assert inside.sum() > 25000 and np.array_equal(tmask_c[band] > 0, tm[:, x0:x1][band] > 0)
    assert np.array_equal(final[inside], co_c[inside])                       # the moved object, bit for bit
    assert (final[~(tmask_c > 0)] == 7).all()                                # everything else is the background handed in
    assert (hole[(sm[:, x0:x1] > 0) & ~(tmask_c > 0)] == 0).all()            # the vacated region of the hole image
