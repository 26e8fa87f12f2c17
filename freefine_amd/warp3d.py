"""The point-cloud warp of the 3-D coarse edit on the HIP kernels of csrc/splat.h: depth-lifted object pixels -> rigid
transform about the cloud's centre -> FoV-perspective projection -> disc splat with the K nearest points per pixel, alpha-composited.
Restates IntegratedP3DTransRasterBlendingFull (/root/reference/src/utils/geo_utils.py:427-528; helpers :343-425), whose renderer is
pytorch3d's PointsRasterizer + AlphaCompositor.  PARITY UNPINNED: pytorch3d is not in the build image, so the result is checked against
oracle/warp3d.py (a numpy restatement of the same published semantics) and against domain invariants, not against the reference's output.

Host side (plumbing): index list of the masked pixels, the centre / extents of the cloud (three reductions over [n, 3]), the 3 x 3 rotation,
the exclusive scan of the tile counts.  Device side: lift, transform + project, tile binning, top-K splat + compositing."""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L


def _p(t):
    return C.c_void_p(t.data_ptr())


def _stream(dev=None):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _device(device):
    """None -> the CURRENT device (rank r of a sharded run works on cuda:LOCAL_RANK; "cuda:0" was the default until round 4)"""
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


def euler_xyz_matrix(rx, ry, rz):
    """pytorch3d's euler_angles_to_matrix(convention="XYZ") of angles given in degrees (get_transformation, geo_utils.py:365-368)"""
    a, b, c = (torch.deg2rad(torch.tensor(float(v), dtype=torch.float32)) for v in (rx, ry, rz))
    one, zero = torch.tensor(1.0), torch.tensor(0.0)
    Rx = torch.stack([one, zero, zero, zero, torch.cos(a), -torch.sin(a), zero, torch.sin(a), torch.cos(a)]).reshape(3, 3)
    Ry = torch.stack([torch.cos(b), zero, torch.sin(b), zero, one, zero, -torch.sin(b), zero, torch.cos(b)]).reshape(3, 3)
    Rz = torch.stack([torch.cos(c), -torch.sin(c), zero, torch.sin(c), torch.cos(c), zero, zero, zero, one]).reshape(3, 3)
    return Rx @ Ry @ Rz


def point_cloud_warp(img, depth, transforms, focal_length_x, focal_length_y, mask, object_only=True, splatting_radius=0.1,
                     splatting_points_per_pixel=5, device=None, fov_deg=60.0, return_covered=False, pixel_translation=False):
    """img uint8 [H, W, 3], depth float [H, W], mask [H, W] (> 0 = object), transforms = [tx, ty, tz (relative to the cloud's extent),
    rx, ry, rz (degrees), sx, sy, sz] -> (rendered uint8 [H, W, 3], mask uint8 [H, W] by the reference's own test `sum of the K ids != -30`
    -- 255 everywhere unless K = 30, geo_utils.py:517) and, with return_covered, the pixels any point reached (x 255).
    pixel_translation (extension, off = the reference's semantics): tx, ty, tz are IMAGE PIXELS -- the cloud moves by t * mean depth / focal
    length, i.e. by t pixels to the right / down at its mean depth (tz: away from the camera) -- instead of fractions of the cloud's extent."""
    lib = L.load()
    dev = _device(device)
    H, W = depth.shape
    K = int(splatting_points_per_pixel)
    d = torch.as_tensor(np.ascontiguousarray(depth), dtype=torch.float32).to(dev).contiguous()
    m = torch.as_tensor(np.ascontiguousarray(mask)).to(dev).reshape(-1)
    keep = (m > 0) if object_only else torch.ones_like(m, dtype=torch.bool)
    idx = torch.nonzero(keep).flatten().to(torch.int32).contiguous()
    n = int(idx.numel())
    assert K * max(n, 1) < 2 ** 31, "the id sum of a pixel's K points is kept in int32 (the reference sums in int64): K * n must stay below 2^31"
    image = torch.zeros(H, W, 3, dtype=torch.float32, device=dev)
    idx_sum = torch.full((H, W), -K, dtype=torch.int32, device=dev)
    covered = torch.zeros(H, W, dtype=torch.uint8, device=dev)
    if n > 0:
        rgb = torch.as_tensor(np.ascontiguousarray(img)).to(dev).reshape(-1, 3).float().index_select(0, idx.long()).contiguous()
        pts = torch.empty(n, 4, dtype=torch.float32, device=dev)
        L.check(lib.ffn_splat_lift(_stream(dev), _p(d), _p(idx), _p(pts), n, W, H, float(focal_length_x), float(focal_length_y)), "splat_lift")
        xyz = pts[:, :3]
        # centre and extents (translation invariant: centred or not) in ONE host round trip
        c_h, ext = torch.cat([xyz.mean(0), xyz.max(0).values - xyz.min(0).values]).cpu().split(3)
        x = L.SplatXform()
        for a in range(3):
            x.center[a] = float(c_h[a])
            t = float(transforms[a])
            if pixel_translation:                                          # world +x / +y point left / up after the flip of geo_utils.py:455
                x.translate[a] = (-1.0 if a < 2 else 1.0) * t * float(c_h[2]) / float((focal_length_x, focal_length_y, focal_length_x)[a])
            else:
                x.translate[a] = 0.0 if t == 0 else float(ext[a]) * t      # refine_transforms (geo_utils.py:399-413)
            x.scale[a] = float(transforms[6 + a])
        R = euler_xyz_matrix(*transforms[3:6])
        for a in range(9):
            x.rotate[a] = float(R.reshape(-1)[a])
        x.tan_half_fov = math.tan(math.radians(fov_deg) / 2)
        proj = torch.empty(n, 4, dtype=torch.float32, device=dev)
        L.check(lib.ffn_splat_project(_stream(dev), _p(pts), _p(proj), n, C.byref(x)), "splat_project")
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        counts = torch.zeros(tiles, dtype=torch.int32, device=dev)
        r = float(splatting_radius)
        L.check(lib.ffn_splat_bin(_stream(dev), 0, _p(proj), n, r, W, H, _p(counts), None, None), "splat_bin(count)")
        offs = torch.zeros(tiles + 1, dtype=torch.int32, device=dev)
        offs[1:] = torch.cumsum(counts, 0)
        total = int(offs[-1].item())
        lst = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        counts.zero_()
        L.check(lib.ffn_splat_bin(_stream(dev), 1, _p(proj), n, r, W, H, _p(counts), _p(offs), _p(lst)), "splat_bin(fill)")
        L.check(lib.ffn_splat_render(_stream(dev), _p(proj), _p(rgb), _p(offs), _p(lst), r, K, W, H, _p(image), _p(idx_sum), _p(covered)), "splat_render")
    out = image.cpu().numpy().astype(np.uint8)                              # `.astype(np.uint8)` of the float image (geo_utils.py:516)
    ref_mask = ((idx_sum != -30).to(torch.uint8) * 255).cpu().numpy()
    if return_covered:
        return out, ref_mask, (covered * 255).cpu().numpy()
    return out, ref_mask


def coarse_edit_3d(ori_img, ori_mask, depth, transforms, background, focal_length=550.0, splatting_radius=None, points_per_pixel=5,
                   device=None, pixel_translation=True):
    """A coarse 3-D edit built from the RGB image and a transform instead of being read from disk (freefine_batch_infer_3d_depth.py:121 reads
    `coarse3d_depth_anything/...png`, rendered by the GeoDiffuser warp of get_3d_transform_correspondence.py:217-251, which is not in the
    reference tree): the object's pixels, lifted through `depth`, moved by `transforms` (translation in image pixels by default -- this
    module's own convention, an extension; see geobench.load_case_3d_rgb) and splatted over `background` (the inpainted scene).  NOT a
    restatement of the dataset's generator.  Returns (coarse uint8 [H, W, 3], target_mask uint8 {0, 255}).
    Default radius: 1.5 pixels in NDC units (holes between neighbouring source pixels close under moderate rotations / scalings)."""
    H, W = ori_mask.shape[:2]
    m2 = ori_mask if ori_mask.ndim == 2 else ori_mask[:, :, 0]
    r = splatting_radius if splatting_radius is not None else 1.5 * 2.0 / min(H, W)
    img, _, cov = point_cloud_warp(ori_img, depth, transforms, focal_length, focal_length, m2, True, r, points_per_pixel, device, return_covered=True,
                                   pixel_translation=pixel_translation)
    tgt = cov > 0
    coarse = np.where(tgt[:, :, None], img, background).astype(np.uint8)
    return coarse, tgt.astype(np.uint8) * 255
