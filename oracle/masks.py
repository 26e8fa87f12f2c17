"""ORACLE (test infrastructure, never imported by the product path).

CPU restatement of FreeFine's mask preparation (/root/reference/src/demo/model.py):
  dilate_mask :927-934 (cv2.dilate with a k x k ones kernel == max filter, window [-k//2, k-k//2-1], zero border),
  prepare_tensor_mask :1622-1639, prepare_various_mask :1431-1512, prepare_composition_masks :1514-1609,
  prepare_mask_bggen :1610-1620.
All masks stay in the dtype of the ndarray they came from (uint8 in the reference's drivers), so `a - b` and `1 - a`
wrap exactly as in the reference.
"""
from copy import deepcopy

import numpy as np
import torch
import torch.nn.functional as F
from scipy import ndimage


def dilate_mask(mask, k=15):
    mask = mask.astype(np.uint8)
    return ndimage.maximum_filter(mask, size=(k, k), mode="constant", cval=0)


def _nearest(t, size):
    return F.interpolate(t[None, None], size, mode="nearest")[0, 0]


def prepare_tensor_mask(mask, sup_res_w, sup_res_h):
    if mask.ndim == 3:
        mask = mask[:, :, 0]
    t = _nearest(torch.tensor(mask), (sup_res_h, sup_res_w))
    t[t > 0.0] = 1.0
    return t


def prepare_various_mask(shifted_mask, ori_mask, draw_mask, sup_res_w, sup_res_h, latent_hw, use_auto_draw=False,
                         cons_area=None, reduce_inp_artifacts=False):
    """returns fg_mask, shifted_mask_tensor, ori_mask_tensor, completion_mask_cfg [h,w], local_var_reg [h,w]"""
    ptm = lambda m: prepare_tensor_mask(m, sup_res_w, sup_res_h)
    if not use_auto_draw:
        if not reduce_inp_artifacts:
            shifted, ori = ptm(shifted_mask), ptm(ori_mask)
            flexible = ptm(draw_mask) * (1 - shifted)
            fg = flexible + shifted
            fg[fg > 0] = 1.0
            complete, local_var = flexible, flexible
        else:
            assert cons_area is not None
            dil = ptm(dilate_mask(ori_mask, 30))
            cons = ptm(cons_area)
            shifted, ori = ptm(shifted_mask), ptm(ori_mask)
            flexible = ptm(draw_mask) * (1 - shifted)
            fg = flexible + shifted
            fg[fg > 0] = 1.0
            complete = flexible
            local_var = (1 - cons) * (1 - shifted) * dil + flexible
            local_var[local_var > 0] = 1
    else:
        assert cons_area is not None
        if not reduce_inp_artifacts:
            dil_tgt = ptm(dilate_mask(shifted_mask, 15))
            shifted, ori, cons = ptm(shifted_mask), ptm(ori_mask), ptm(cons_area)
            fg = shifted
            cons = cons - ori
            complete = (1 - cons) * (1 - shifted) * dil_tgt
            local_var = complete
        else:
            dil_tgt_np = dilate_mask(shifted_mask, 15)
            dil_ori_np = dilate_mask(ori_mask, 30)
            dil, dil_tgt = ptm(dil_ori_np), ptm(dil_tgt_np)
            shifted, ori, cons = ptm(shifted_mask), ptm(ori_mask), ptm(cons_area)
            fg = shifted
            cons = cons - ori
            complete = dil + dil_tgt
            complete[complete > 0] = 1
            complete *= (1 - cons) * (1 - shifted)
            local_var = complete
    complete = _nearest(complete, latent_hw)
    local_var = _nearest(local_var, latent_hw)
    return fg, shifted, ori, complete, local_var


def prepare_mask_bggen(mask, sup_res_w, sup_res_h, latent_hw):
    t = prepare_tensor_mask(mask, sup_res_w, sup_res_h)
    return t, _nearest(t, latent_hw)


def prepare_composition_masks(ori_mask_lists, tgt_mask_lists, sup_res_w, sup_res_h, latent_hw, dil_completion=False,
                              dil_factor=15, draw_mask=None, appearance_transfer=False):
    ptm = lambda m: prepare_tensor_mask(m, sup_res_w, sup_res_h)
    ori = [ptm(m) for m in ori_mask_lists]
    tgt = []
    if appearance_transfer:
        lp = torch.zeros_like(ori[0])
        for sm in tgt_mask_lists:
            d = ptm(dilate_mask(sm, dil_factor))
            tgt.append(d)
            lp += d
        lp[lp > 0] = 1
        tgt.append(1 - lp)
        lp = _nearest(lp, latent_hw)
        return torch.stack(tgt), torch.stack(ori), lp, deepcopy(lp)
    lp, fg = torch.zeros_like(ori[0]), torch.zeros_like(ori[0])
    if draw_mask is None:
        for sm in tgt_mask_lists:
            d, s = ptm(dilate_mask(sm, dil_factor)), ptm(sm)
            tgt.append(d if dil_completion else s)
            fg += s
            lp += d
        fg[fg > 0] = 1
        lp[lp > 0] = 1
        tgt.append(1 - fg if dil_completion else 1 - lp)
        lp = _nearest(lp * (1 - fg), latent_hw)
        cfg = deepcopy(lp) if dil_completion else torch.zeros_like(lp)
        return torch.stack(tgt), torch.stack(ori), lp, cfg
    for i, sm in enumerate(tgt_mask_lists):
        s = ptm(sm)
        d = ptm(draw_mask[i]) + s
        d[d > 0] = 1
        tgt.append(d)
        fg += s
        lp += d
    fg[fg > 0] = 1
    lp[lp > 0] = 1
    tgt.append(1 - lp)
    lp = _nearest(lp * (1 - fg), latent_hw)
    return torch.stack(tgt), torch.stack(ori), lp, lp
