"""ORACLE (test infrastructure, never imported by the product path).  PARITY UNPINNED: pytorch3d, whose renderer the reference calls, is not
in the build image, so no golden vectors could be generated; the semantics follow pytorch3d's published rasterize_points (naive kernel) and
alpha_composite, and the tests check invariants of the domain besides HIP == this restatement.

numpy restatement of IntegratedP3DTransRasterBlendingFull (/root/reference/src/utils/geo_utils.py:427-528) and its helpers
(get_transformation :343-378, refine_transforms :399-413, transform_point_cloud :416-425): lift the masked pixels through the depth map,
move the cloud about its centre (translate -> rotate XYZ -> scale, row vectors), project with a 60-degree FoV perspective camera that sits
where the original camera was, splat every point as a disc of `radius` NDC units keeping the K nearest per pixel, alpha-composite with
weights 1 - d^2 / r^2 over a black background.  Brute force over pixels x points: small cases only."""
import numpy as np


def euler_xyz_matrix(rx, ry, rz):
    """pytorch3d.transforms.euler_angles_to_matrix(convention="XYZ") of angles in DEGREES (get_transformation :365-368): Rx @ Ry @ Rz."""
    a, b, c = (np.deg2rad(np.float32(v)).astype(np.float32) for v in (rx, ry, rz))
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]], np.float32)
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]], np.float32)
    Rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]], np.float32)
    return (Rx @ Ry @ Rz).astype(np.float32)


def lift(depth, mask, fx, fy, object_only=True):
    h, w = depth.shape
    i, j = np.meshgrid(np.arange(w), np.arange(h), indexing="xy")
    z = depth.astype(np.float32)
    x = (i.astype(np.float32) - np.float32(w) * np.float32(0.5)) * z / np.float32(fx)
    y = (j.astype(np.float32) - np.float32(h) * np.float32(0.5)) * z / np.float32(fy)
    pts = np.stack((-x, -y, z), -1).reshape(-1, 3)                   # open-cv world -> pytorch3d world (:455)
    keep = (mask.reshape(-1) > 0) if object_only else np.ones(h * w, bool)
    return pts[keep], np.nonzero(keep)[0]


def absolute_translation(pts_centered, transforms):
    """refine_transforms (:399-413): a relative translation d becomes d * (extent of the cloud along that axis); 0 stays 0"""
    out = []
    for a in range(3):
        d = float(transforms[a])
        out.append(0.0 if d == 0 else float(pts_centered[:, a].max() - pts_centered[:, a].min()) * d)
    return out


def project(pts, transforms, fov_deg=60.0):
    """-> (ndc_x, ndc_y, z_view) per point"""
    c = pts.mean(0, dtype=np.float32)
    p = pts - c
    t = np.array(absolute_translation(p, transforms), np.float32)
    R = euler_xyz_matrix(*transforms[3:6])
    s = np.array(transforms[6:9], np.float32)
    v = ((p + t) @ R) * s + c                                        # Translate . Rotate . Scale, then the camera's T = centre (:478-481)
    inv_tan = np.float32(1.0 / np.tan(np.deg2rad(fov_deg) / 2))
    return np.stack((v[:, 0] * inv_tan / v[:, 2], v[:, 1] * inv_tan / v[:, 2], v[:, 2]), -1).astype(np.float32), c, t, R, s


def pix_to_ndc(i, S1, S2):
    rng = np.float32(2.0)
    if S1 > S2:
        rng = np.float32(S1) / np.float32(S2) * rng
    off = rng * np.float32(0.5)
    return -off + (rng * np.asarray(i, np.float32) + off) / np.float32(S1)


def splat(proj, rgb, h, w, radius, K):
    """-> image [h, w, 3] float32, idx [h, w, K] (-1 = empty), dist2 [h, w, K]"""
    r2 = np.float32(radius) * np.float32(radius)
    xf = pix_to_ndc(w - 1 - np.arange(w), w, h)
    yf = pix_to_ndc(h - 1 - np.arange(h), h, w)
    image = np.zeros((h, w, 3), np.float32)
    idx = -np.ones((h, w, K), np.int64)
    dist = np.zeros((h, w, K), np.float32)
    ok = proj[:, 2] >= 0
    order = np.lexsort((np.arange(len(proj)), proj[:, 2]))           # by depth, ties by point index
    order = order[ok[order]]
    px, py = proj[order, 0], proj[order, 1]
    for r in range(h):
        dy2 = (yf[r] - py) ** 2
        for cc in range(w):
            d2 = (xf[cc] - px) ** 2 + dy2
            hit = np.nonzero(d2 < r2)[0][:K]
            cum = np.float32(1.0)
            for k, q in enumerate(hit):
                wgt = np.float32(1.0) - d2[q] / r2
                image[r, cc] += cum * wgt * rgb[order[q]]
                cum = cum * (np.float32(1.0) - wgt)
                idx[r, cc, k] = order[q]
                dist[r, cc, k] = d2[q]
    return image, idx, dist


def point_cloud_warp(img, depth, transforms, fx, fy, mask, object_only=True, splatting_radius=0.1, splatting_points_per_pixel=5):
    """IntegratedP3DTransRasterBlendingFull(..., return_mask=True) -> (uint8 image, uint8 mask by the reference's own test :517, covered)"""
    h, w = depth.shape
    pts, keep = lift(depth, mask, fx, fy, object_only)
    rgb = img.reshape(-1, 3).astype(np.float32)[keep]
    proj, *_ = project(pts, transforms)
    image, idx, _ = splat(proj, rgb, h, w, splatting_radius, splatting_points_per_pixel)
    ref_mask = (idx.sum(-1) != -30).astype(np.uint8) * 255             # the reference's literal test: true everywhere unless K = 30
    covered = (idx[..., 0] >= 0).astype(np.uint8) * 255
    return image.astype(np.uint8), ref_mask, covered
