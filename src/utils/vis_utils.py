"""Host-side pre/post-processing of the GeoBench drivers: drop-in for the reference's
`src/utils/vis_utils.py` import path (freefine_batch_infer_2d.py:10), written without OpenCV (absent here and on the GPU
box) on PIL / numpy / scipy.

Each function restates the documented behaviour of the cv2 call the reference makes (cited /root/reference/src/utils/vis_utils.py):
nearest / Lanczos-4 resize, ones-kernel dilation, getRotationMatrix2D + warpAffine (bilinear for images, nearest for masks,
constant-0 border), BGR<->RGB PNG I/O.  PINNED ON ONE REFERENCE VECTOR since round 6: the reference tree holds one input -> output set of
its own coarse edit (Examples/Editing/2D/tower), and re_edit_2d + the nearest mask resize reproduce its target mask exactly over the full
frame and its coarse image exactly inside it (tests/golden/g12_tower_coarse_edit.npz, tools/pin_n2_tower.py) -- that example is an integer
translation, so the INTERPOLATING paths (bilinear weights under rotation / scaling, Lanczos-4) remain PARITY UNPINNED: cv2 cannot be
imported in this environment, and its 8-bit paths use fixed-point coefficient tables, against which these float restatements may differ
by 1 LSB.  Index-exact functions (nearest resize, dilation, mask arithmetic incl. the uint8 wrap of
get_constrain_areas, JSON / path helpers) are covered by tests/test_vis_utils_cpu.py.
"""
import json
import os

import numpy as np
from PIL import Image
from scipy import ndimage


# ---------------------------------------------------------------------------------------------------------------------
# viewers (vis_utils.py:10-92): no display on a batch node -- accepted and ignored
# ---------------------------------------------------------------------------------------------------------------------
def temp_view_img(image, title=None):
    return None


def visualize_rgb_image(image, title=None):
    return None


def temp_view(mask, title="Mask", name=None):
    return None


# ---------------------------------------------------------------------------------------------------------------------
# PNG / JSON I/O (vis_utils.py:94-179)
# ---------------------------------------------------------------------------------------------------------------------
def _imread_bgr(path):
    """cv2.imread(path): uint8 [H,W,3] in B,G,R order (grey / palette / alpha inputs are converted to 3 channels); None if unreadable."""
    try:
        with Image.open(path) as im:
            rgb = np.asarray(im.convert("RGB"))
    except (FileNotFoundError, OSError):
        return None
    return np.ascontiguousarray(rgb[:, :, ::-1])


def _imwrite(path, arr):
    arr = np.asarray(arr)
    Image.fromarray(arr if arr.ndim == 2 else np.ascontiguousarray(arr[:, :, ::-1])).save(path)   # cv2.imwrite takes BGR


def _sub(dst_dir, da_name, ins_name):
    d = os.path.join(dst_dir, str(da_name), str(ins_name))
    os.makedirs(d, exist_ok=True)
    return d


def replace_mask(mask, src_mask_path):
    _imwrite(src_mask_path, mask.astype(np.uint8) * 255)
    return src_mask_path


def save_mask(mask, dst_dir, da_name, ins_name, sample_id):
    path = os.path.join(_sub(dst_dir, da_name, ins_name), f"{sample_id}.png")
    _imwrite(path, mask.astype(np.uint8) * 255)
    return path


def save_img(img, dst_dir, da_name, ins_name, sample_id):
    """img: uint8 RGB [H,W,3] (the reference converts RGB->BGR for cv2.imwrite, i.e. the file holds the RGB image)."""
    path = os.path.join(_sub(dst_dir, da_name, ins_name), f"{sample_id}.png")
    Image.fromarray(np.asarray(img, dtype=np.uint8)).save(path)
    return path


def save_masks(masks, dst_dir, da_name):
    d = os.path.join(dst_dir, str(da_name))
    os.makedirs(d, exist_ok=True)
    paths = []
    for idx, mask in enumerate(masks):
        p = os.path.join(d, f"mask_{idx + 1}.png")
        _imwrite(p, mask)
        paths.append(p)
    return paths


def save_json(data_dict, file_path):
    with open(file_path, "w", encoding="utf-8") as f:
        json.dump(data_dict, f, ensure_ascii=False, indent=4)


def load_json(file_path):
    """None (with a message) on a missing or malformed file, like the reference (vis_utils.py:162-179)."""
    try:
        with open(file_path, "r", encoding="utf-8") as f:
            return json.load(f)
    except FileNotFoundError:
        print(f"file not found: {file_path}")
    except json.JSONDecodeError:
        print(f"malformed JSON: {file_path}")
    return None


# ---------------------------------------------------------------------------------------------------------------------
# masks (vis_utils.py:183-208, 340-348)
# ---------------------------------------------------------------------------------------------------------------------
def get_constrain_areas(mask_list_path=None, mask_list=None, ori_mask=None):
    """union of the instance masks minus the edited object's mask, IN uint8: where ori_mask covers pixels outside the union the
    subtraction wraps to 255; ori_mask is binarised in place, as in the reference."""
    if mask_list is None:
        mask_list = [_imread_bgr(p) for p in mask_list_path]
    constrain = np.zeros_like(mask_list[0])
    for m in mask_list:
        constrain += m
    constrain[constrain > 0] = 1
    ori_mask[ori_mask > 0] = 1
    return constrain - ori_mask


def prepare_mask_pool(instances):
    pool = []
    for _, ins in instances.items():
        if len(ins) == 0:
            continue
        pool.append(ins[next(iter(ins))]["ori_mask_path"])
    return pool


def dilate_mask(mask, dilate_factor=15):
    """cv2.dilate(mask, ones(k,k)): anchor k//2, border treated as -inf (constant 0 for masks)."""
    mask = mask.astype(np.uint8)
    k = int(dilate_factor)
    size = (k, k) + (1,) * (mask.ndim - 2)
    return ndimage.maximum_filter(mask, size=size, mode="constant", cval=0)


# ---------------------------------------------------------------------------------------------------------------------
# resize (vis_utils.py:349-374)
# ---------------------------------------------------------------------------------------------------------------------
def _resize_nearest(img, dsize):
    """cv2.resize(..., INTER_NEAREST): src index = min(floor(dst * src/dst_size), src-1) on each axis (no half-pixel centre)."""
    w, h = dsize
    H, W = img.shape[:2]
    ys = np.minimum((np.arange(h) * (H / h)).astype(np.int64), H - 1)
    xs = np.minimum((np.arange(w) * (W / w)).astype(np.int64), W - 1)
    return np.ascontiguousarray(img[ys][:, xs])


def _lanczos4_matrix(n_dst, n_src):
    """[n_dst, n_src] resampling matrix of cv2's INTER_LANCZOS4: 8 taps around floor(x), x = (d+0.5)*scale-0.5, kernel
    sinc(t)sinc(t/4), weights normalised to 1, indices clamped to the border (replicate), no anti-aliasing when shrinking."""
    scale = n_src / n_dst
    x = (np.arange(n_dst) + 0.5) * scale - 0.5
    x0 = np.floor(x).astype(np.int64)
    fx = x - x0
    M = np.zeros((n_dst, n_src), dtype=np.float64)
    taps = np.arange(-3, 5)
    t = fx[:, None] - taps[None, :]
    wgt = np.sinc(t) * np.sinc(t / 4.0)
    wgt /= wgt.sum(axis=1, keepdims=True)
    idx = np.clip(x0[:, None] + taps[None, :], 0, n_src - 1)
    for j in range(8):
        np.add.at(M, (np.arange(n_dst), idx[:, j]), wgt[:, j])
    return M


def _resize_lanczos4(img, dsize):
    w, h = dsize
    H, W = img.shape[:2]
    if (H, W) == (h, w):
        return img.copy()
    My, Mx = _lanczos4_matrix(h, H), _lanczos4_matrix(w, W)
    out = np.einsum("yh,hwc->ywc", My, img.astype(np.float64))
    out = np.einsum("xw,ywc->yxc", Mx, out)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def read_and_resize_img(ori_img_path, dsize=(512, 512)):
    """uint8 RGB [h,w,3]"""
    bgr = _imread_bgr(ori_img_path)
    return _resize_lanczos4(np.ascontiguousarray(bgr[:, :, ::-1]), dsize)


def read_and_resize_mask(ori_mask_path, dsize=(512, 512)):
    """3-channel {0,1} uint8 mask"""
    m = _resize_nearest(_imread_bgr(ori_mask_path), dsize)
    m[m > 0] = 1
    return m


def read_and_resize_mask_with_dilation(ori_mask_path, dsize=(512, 512), dilation_factor=None, forbit_area=None):
    m = read_and_resize_mask(ori_mask_path, dsize)
    dil = m
    if dilation_factor is not None:
        dil = dilate_mask(m, dilation_factor)
    if forbit_area is not None:
        dil = np.where(forbit_area, 0, dil)
    return dil


# ---------------------------------------------------------------------------------------------------------------------
# coarse edit: 2-D affine re-placement of the object (vis_utils.py:210-339; driver copy freefine_batch_infer_2d.py:26-88)
# ---------------------------------------------------------------------------------------------------------------------
def _rotation_matrix_2d(center, angle_deg, scale):
    """cv2.getRotationMatrix2D: positive angle = counter-clockwise (origin top-left)."""
    a = np.deg2rad(angle_deg)
    al, be = scale * np.cos(a), scale * np.sin(a)
    cx, cy = center
    return np.array([[al, be, (1 - al) * cx - be * cy], [-be, al, be * cx + (1 - al) * cy]], dtype=np.float64)


def _warp_affine(src, M, dsize, nearest=False):
    """cv2.warpAffine(src, M, dsize) with M the FORWARD map (dst = M src): every destination pixel samples the source at
    M^-1 (x, y); bilinear (source coordinates quantised to 1/32 pixel like cv2's INTER_BITS=5 table) or nearest
    (round half up); outside the source -> 0 (BORDER_CONSTANT)."""
    w, h = dsize
    A = np.vstack([M, [0, 0, 1]])
    Ai = np.linalg.inv(A)
    ys, xs = np.mgrid[0:h, 0:w]
    sx = Ai[0, 0] * xs + Ai[0, 1] * ys + Ai[0, 2]
    sy = Ai[1, 0] * xs + Ai[1, 1] * ys + Ai[1, 2]
    H, W = src.shape[:2]
    s = src if src.ndim == 3 else src[:, :, None]

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = s[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.float64)
        return v * ok[:, :, None]

    if nearest:
        out = tap(np.floor(sy + 0.5).astype(np.int64), np.floor(sx + 0.5).astype(np.int64))
    else:
        qx, qy = np.floor(sx * 32 + 0.5) / 32.0, np.floor(sy * 32 + 0.5) / 32.0
        x0, y0 = np.floor(qx).astype(np.int64), np.floor(qy).astype(np.int64)
        fx, fy = (qx - x0)[:, :, None], (qy - y0)[:, :, None]
        out = (tap(y0, x0) * (1 - fx) + tap(y0, x0 + 1) * fx) * (1 - fy) + (tap(y0 + 1, x0) * (1 - fx) + tap(y0 + 1, x0 + 1) * fx) * fy
        out = np.rint(out)
    out = np.clip(out, 0, 255).astype(src.dtype)
    return out if src.ndim == 3 else out[:, :, 0]


def _affine_for_edit(src_mask, dx, dy, rz, sx, sy):
    ys, xs = np.where(src_mask)
    top, bottom, left, right = ys.min(), ys.max(), xs.min(), xs.max()
    cx, cy = (right + left) / 2, (top + bottom) / 2
    M = _rotation_matrix_2d((cx, cy), -rz, 1)
    M[0, 2] += dx + (1 - sx) * cx           # scaling about the mask centre, decoupled per axis
    M[1, 2] += dy + (1 - sy) * cy
    M[0, 0] *= sx
    M[1, 1] *= sy
    return M


def re_edit_2d(src_img, src_mask, edit_param, inp_cur):
    """move / rotate / scale the masked object of src_img and paste it over the inpainted background inp_cur.
    edit_param: (dx, dy, rz, sx, sy) (vis_utils.py:213) or the 9-tuple (dx, dy, dz, rx, ry, rz, sx, sy, sz) of the GeoBench-2D
    annotations (freefine_batch_infer_2d.py:29).  Returns (coarse image, target mask uint8 {0,255}, image with hole + object)."""
    if src_mask.ndim == 3:
        src_mask = src_mask[:, :, 0]
    if len(edit_param) == 9:
        dx, dy, _, _, _, rz, sx, sy, _ = edit_param
    else:
        dx, dy, rz, sx, sy = edit_param
    h, w = src_mask.shape[:2]
    M = _affine_for_edit(src_mask, dx, dy, rz, sx, sy)
    timg = _warp_affine(src_img, M, (w, h))
    tmask = _warp_affine(src_mask.astype(np.uint8), M, (w, h), nearest=True).astype(bool)
    hole = np.where(src_mask.astype(bool)[:, :, None], 0, src_img)
    trans_hole = np.where(tmask[:, :, None], timg, hole)
    final = np.where(tmask[:, :, None], timg, inp_cur)
    return final, tmask.astype(np.uint8) * 255, trans_hole


def re_edit_3d(src_img, src_mask, edit_param, inp_cur, ori_img_a, ori_mask_a):
    """as re_edit_2d, the hole image built from another view's image / mask (vis_utils.py:275-339)"""
    if src_mask.ndim == 3:
        src_mask = src_mask[:, :, 0]
    dx, dy, rz, sx, sy = edit_param
    h, w = src_mask.shape[:2]
    M = _affine_for_edit(src_mask, dx, dy, rz, sx, sy)
    timg = _warp_affine(src_img, M, (w, h))
    tmask = _warp_affine(src_mask.astype(np.uint8), M, (w, h), nearest=True).astype(bool)
    hole = np.where(ori_mask_a, 0, ori_img_a)
    trans_hole = np.where(tmask[:, :, None], timg, hole)
    final = np.where(tmask[:, :, None], timg, inp_cur)
    return final, tmask.astype(np.uint8) * 255, trans_hole
