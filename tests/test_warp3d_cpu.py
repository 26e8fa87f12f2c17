"""CPU: oracle/warp3d.py (numpy restatement of geo_utils.py:427-528; PARITY UNPINNED, pytorch3d absent) against invariants of the domain."""
import numpy as np

from oracle import warp3d as OW

IDENT = [0, 0, 0, 0, 0, 0, 1, 1, 1]


def _case(h=24, w=24, seed=0, const_depth=None):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    depth = np.full((h, w), const_depth, np.float32) if const_depth else (2.0 + rng.random((h, w))).astype(np.float32)
    mask = np.zeros((h, w), np.uint8)
    mask[6:16, 5:14] = 1
    f = 0.5 * w / np.tan(np.deg2rad(30.0))          # the focal length at which the 60-degree camera reproduces the source pixel grid
    return img, depth, mask, f


def _union4(mask):
    """the reference lifts pixel i at its CORNER (i - W/2, geo_utils.py:439) while the rasterizer tests pixel CENTRES: an untransformed point
    sits half a pixel up-left of its source pixel's centre, at distance sqrt(0.5) pixels from four centres"""
    m = mask > 0
    out = m.copy()
    out[:, :-1] |= m[:, 1:]
    out[:-1, :] |= m[1:, :]
    out[:-1, :-1] |= m[1:, 1:]
    return out


def test_identity_transform_covers_the_four_pixel_centres_around_every_lifted_corner():
    img, depth, mask, f = _case()
    img[...] = (200, 120, 40)
    r_px = 0.8
    out, ref_mask, cov = OW.point_cloud_warp(img, depth, IDENT, f, f, mask, splatting_radius=r_px * 2 / 24, splatting_points_per_pixel=5)
    assert np.array_equal(cov > 0, _union4(mask))
    # an interior pixel sees four points at d^2 = 0.5 px^2: weight w = 1 - 0.5 / 0.64 each, composited front to back
    w = 1 - 0.5 / r_px ** 2
    want = np.array([200, 120, 40]) * (1 - (1 - w) ** 4)
    assert np.abs(out[10, 9].astype(float) - want).max() <= 1.0
    assert (out[~_union4(mask)] == 0).all() and (ref_mask == 255).all()       # black background; the reference's mask test is always true at K = 5


def test_pure_translation_at_constant_depth_shifts_the_object():
    img, depth, mask, f = _case(const_depth=2.0)
    r = 0.8 * 2 / 24
    base, _, cov0 = OW.point_cloud_warp(img, depth, IDENT, f, f, mask, splatting_radius=r)
    # the object spans 9 columns: extent in x = 8 pixels * z / f; a relative translation of -3/8 of it moves it 3 pixels to the RIGHT
    # (the world's +x points left after the sign flip of geo_utils.py:455)
    out, _, cov = OW.point_cloud_warp(img, depth, [-3 / 8, 0, 0, 0, 0, 0, 1, 1, 1], f, f, mask, splatting_radius=r)
    assert np.array_equal(cov, np.roll(cov0, 3, axis=1))
    assert np.abs(out.astype(int) - np.roll(base, 3, axis=1).astype(int)).max() <= 1


def test_nearest_points_win_and_weights_follow_the_distance():
    # two points on the same pixel: the nearer one (smaller z) is composited first with weight 1, so the farther one contributes nothing
    proj = np.array([[0.0, 0.0, 2.0], [0.0, 0.0, 1.0]], np.float32)
    xf = OW.pix_to_ndc(np.arange(8), 8, 8)
    proj[:, 0] = proj[:, 1] = xf[3]
    rgb = np.array([[200, 0, 0], [0, 100, 0]], np.float32)
    image, idx, _ = OW.splat(proj, rgb, 8, 8, 0.05, 2)
    r, c = 8 - 1 - 3, 8 - 1 - 3
    assert list(idx[r, c]) == [1, 0] and np.allclose(image[r, c], [0, 100, 0])
    # a point behind the camera is never drawn
    image, idx, _ = OW.splat(np.array([[xf[3], xf[3], -1.0]], np.float32), rgb[:1], 8, 8, 0.05, 2)
    assert (idx == -1).all() and (image == 0).all()


def test_rotation_matrix_is_pytorch3d_xyz_convention():
    R = OW.euler_xyz_matrix(90, 0, 0)
    assert np.allclose(np.array([0, 1, 0], np.float32) @ R, [0, 0, -1], atol=1e-6)     # row vector: y -> -z under Rx(90 deg)
    R = OW.euler_xyz_matrix(10, 20, 30)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-6)
