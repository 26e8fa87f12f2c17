"""GPU: the point-cloud warp on the HIP kernels (freefine_amd/warp3d.py -> csrc/splat.h through the C ABI) against the numpy restatement
oracle/warp3d.py of geo_utils.py:427-528 (PARITY UNPINNED: pytorch3d absent -- both restate its published semantics) and against the
domain invariants the CPU tests hold the oracle to."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(h, w, seed):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    depth = (2.0 + rng.random((h, w))).astype(np.float32)
    mask = np.zeros((h, w), np.uint8)
    mask[h // 4:3 * h // 4, w // 5:3 * w // 5] = 255
    return img, depth, mask, 0.5 * min(h, w) / np.tan(np.deg2rad(30.0))


@pytest.mark.parametrize("h,w,K,r_px,tf", [
    (48, 48, 5, 0.8, [0, 0, 0, 0, 0, 0, 1, 1, 1]),
    (64, 64, 5, 2.5, [0.2, -0.1, 0.05, 10, 25, -15, 1.0, 1.0, 1.0]),
    (40, 72, 12, 3.0, [-0.3, 0.2, 0.0, 0, 40, 0, 1.2, 0.8, 1.0]),          # non-square image (the longer side's NDC range grows), K on the 16-slot kernel
    (72, 40, 30, 2.0, [0, 0, -0.2, 5, 0, 60, 1.0, 1.0, 1.0]),             # K = 30: the one value at which the reference's own mask test is meaningful
    (96, 96, 8, 6.0, [0.4, 0.4, 0.3, -20, 10, 5, 0.7, 0.7, 0.7]),         # discs spanning several 16 x 16 tiles
])
def test_point_cloud_warp_matches_the_numpy_restatement(gpu, h, w, K, r_px, tf):
    from freefine_amd import warp3d
    from oracle import warp3d as OW
    img, depth, mask, f = _case(h, w, h + w + K)
    r = r_px * 2.0 / min(h, w)
    out, ref_mask, cov = warp3d.point_cloud_warp(img, depth, tf, f, f, mask, True, r, K, device=gpu, return_covered=True)
    o_out, o_ref_mask, o_cov = OW.point_cloud_warp(img, depth, tf, f, f, mask, True, r, K)
    n = h * w
    # a point whose distance to a pixel centre equals the radius to within rounding may flip between the two float evaluations
    assert (cov != o_cov).sum() <= max(2, n // 500), (cov != o_cov).sum()
    same = cov == o_cov
    bad = (np.abs(out.astype(int) - o_out.astype(int)).max(-1) > 1) & same
    assert bad.sum() <= max(2, n // 200), bad.sum()
    assert (ref_mask != o_ref_mask).sum() <= max(2, n // 500)
    assert o_cov.sum() > 0 and (out[cov == 0] == 0).all()


def test_point_cloud_warp_invariants_and_repeatability(gpu):
    """identity: every lifted corner covers the four pixel centres around it; translation at constant depth = shift; bit-repeatable (the
    tile lists are filled through atomics in any order, the top-K is ordered by (depth, point id))"""
    from freefine_amd import warp3d
    h = w = 64
    img, depth, mask, f = _case(h, w, 1)
    depth[...] = 2.0
    r = 0.8 * 2 / w
    ident = [0, 0, 0, 0, 0, 0, 1, 1, 1]
    base, _, cov0 = warp3d.point_cloud_warp(img, depth, ident, f, f, mask, True, r, 5, device=gpu, return_covered=True)
    m = mask > 0
    u = m.copy()
    u[:, :-1] |= m[:, 1:]
    u[:-1, :] |= m[1:, :]
    u[:-1, :-1] |= m[1:, 1:]
    assert np.array_equal(cov0 > 0, u)
    ext = np.ptp(np.nonzero(m.any(0))[0])                      # object extent in pixels along x
    out, _, cov = warp3d.point_cloud_warp(img, depth, [-5 / ext, 0, 0, 0, 0, 0, 1, 1, 1], f, f, mask, True, r, 5, device=gpu, return_covered=True)
    assert np.array_equal(cov, np.roll(cov0, 5, axis=1))
    assert np.abs(out.astype(int) - np.roll(base, 5, axis=1).astype(int)).max() <= 1
    # translation given in image pixels (the GeoBench edit_param convention of the 3d_rgb variant): +7 px right, +3 px down at constant depth
    outp, _, covp = warp3d.point_cloud_warp(img, depth, [7, 3, 0, 0, 0, 0, 1, 1, 1], f, f, mask, True, r, 5, device=gpu, return_covered=True, pixel_translation=True)
    assert np.array_equal(covp, np.roll(cov0, (3, 7), axis=(0, 1)))
    assert np.abs(outp.astype(int) - np.roll(base, (3, 7), axis=(0, 1)).astype(int)).max() <= 1
    big = [0.1, 0.1, 0, 15, -30, 20, 1.1, 0.9, 1.0]
    a = warp3d.point_cloud_warp(img, 2.0 + 0 * depth, big, f, f, mask, True, 4.0 * 2 / w, 15, device=gpu)     # constant depth: every depth ties
    b = warp3d.point_cloud_warp(img, 2.0 + 0 * depth, big, f, f, mask, True, 4.0 * 2 / w, 15, device=gpu)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # empty mask: black image, nothing covered
    e = warp3d.point_cloud_warp(img, depth, ident, f, f, np.zeros_like(mask), True, r, 5, device=gpu, return_covered=True)
    assert (e[0] == 0).all() and (e[2] == 0).all()
