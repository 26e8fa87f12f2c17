"""CPU: the oracle's loops (oracle/pipeline.py) against trajectories produced by the reference's own loops + hooks
(tests/golden/g5_loops.npz, made by tools/gen_golden.py).  Also pins the (step, attention-call) -> branch /
context_guidance table captured from the reference's controller."""
import os

import numpy as np
import torch

from golden_cases import BG_CASES, BRANCH_CODE, CMP_CASES, compose_masks, edit_cases, mask_inputs, oracle_pipe, synth_images
from oracle import masks as OM

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
torch.set_grad_enabled(False)
TOL = 2e-4  # fp32 reassociation through ~10 tiny-UNet steps; reference-vs-oracle deviations measured at <= 5e-6


def traj_dev(traj, ref):
    worst = 0.0
    for a, b in zip(traj, ref):
        a = a if a.ndim == b.ndim else a[0]
        b = torch.from_numpy(np.asarray(b))
        fa, fb = torch.isfinite(a), torch.isfinite(b)
        assert torch.equal(fa, fb)
        worst = max(worst, ((a - b)[fa].abs().max() / (1.0 + b[fa].abs().max())).item())
    return worst


def test_g5_edit_loops_and_branch_table():
    g = np.load(os.path.join(GOLD, "g5_loops.npz"))
    ori_img, coarse, _ = synth_images()
    ori, tgt, *_ = mask_inputs()
    for name, unet_name, kw in edit_cases():
        kw = dict(kw)
        text, gs, eta = kw.pop("guidance_text"), kw.pop("guidance_scale"), kw.pop("eta")
        op = oracle_pipe(unet_name)
        img_e, img_r, traj = op.freefine_generation(ori_img, ori, coarse, tgt, text, gs, eta, seed=42, **kw)
        ref = g[f"{name}_traj"]
        assert len(traj) == len(ref) == kw["num_step"] - kw["start_step"] + 1
        assert traj_dev(traj, ref) < TOL, name
        assert np.abs(img_e[::4, ::4].astype(int) - g[f"{name}_img"].astype(int)).max() <= 1, name
        assert np.abs(img_r[::4, ::4].astype(int) - g[f"{name}_ref_img"].astype(int)).max() <= 1, name


def test_g2_branch_table_matches_reference():
    g = np.load(os.path.join(GOLD, "g5_loops.npz"))
    ori_img, coarse, _ = synth_images()
    ori, tgt, *_ = mask_inputs()
    for name, unet_name, kw in edit_cases():
        if name not in ("edit_tca_draw", "edit_mmsa_es", "edit_sdsa"):
            continue
        kw = dict(kw)
        text, gs, eta = kw.pop("guidance_text"), kw.pop("guidance_scale"), kw.pop("eta")
        op = oracle_pipe(unet_name)
        cgs = []
        orig = op.modulator.attend

        def attend(*a, _o=orig, _m=op.modulator, **k):
            cgs.append(_m.context_guidance)
            return _o(*a, **k)
        op.modulator.attend = attend
        op.freefine_generation(ori_img, ori, coarse, tgt, text, gs, eta, seed=42, **kw)
        rtrace = g[f"{name}_trace"]                       # rows: (cur_step, cur_att_layer, code, context_guidance)
        otrace = op.modulator.trace
        assert len(rtrace) == len(otrace) == 2 * (kw["num_step"] - kw["start_step"]) * 32
        for r, (s, blk, is_cross, place, br), cg in zip(rtrace, otrace, cgs):
            assert int(r[0]) == s and int(r[1]) // 2 == blk and int(r[2]) == BRANCH_CODE[br], (name, r, s, blk, br)
            if int(r[2]) == 1 and br == "tca:tca":
                assert abs(r[3] - cg) < 1e-12


def test_g5_background_generation():
    g = np.load(os.path.join(GOLD, "g5_loops.npz"))
    ori_img, _, _ = synth_images()
    ori, *_ = mask_inputs()
    hole = OM.dilate_mask(ori // 255, 30)
    for name, kw in BG_CASES:
        op = oracle_pipe("tiny")
        img, traj = op.freefine_background_generation(ori_img, hole, "empty scene", 3.5, 1.0, seed=7, **kw)
        assert traj_dev(traj, g[f"{name}_traj"]) < TOL, name
        assert np.abs(img[::4, ::4].astype(int) - g[f"{name}_img"].astype(int)).max() <= 1, name


def test_g5_composition():
    g = np.load(os.path.join(GOLD, "g5_loops.npz"))
    ori_img, coarse, img2 = synth_images()
    oris, tgts = compose_masks()
    for name, kw in CMP_CASES:
        op = oracle_pipe("tiny")
        img, traj = op.freefine_compose([ori_img, img2], oris, tgts, coarse, ["a cup", "a dog"], 7.5, 1.0, end_step=8, num_step=10,
                                        start_step=6, seed=11, dil_factor=9, end_scale=0.5, **kw)
        assert traj_dev(traj, g[f"{name}_traj"]) < TOL, name
        assert np.abs(img[::4, ::4].astype(int) - g[f"{name}_img"].astype(int)).max() <= 1, name


def _run_oracle_n50(name, hook, unet_name, planted, kw):
    from golden_cases import tiny_state
    ori_img, coarse, img2 = synth_images()
    ori, tgt, *_ = mask_inputs()
    kw = dict(kw)
    op = oracle_pipe(unet_name)
    op.unet.load_state_dict(tiny_state(unet_name, 0, planted))
    if hook == "edit":
        text, gs, eta = kw.pop("guidance_text"), kw.pop("guidance_scale"), kw.pop("eta")
        img, _, traj = op.freefine_generation(ori_img, ori, coarse, tgt, text, gs, eta, seed=42, **kw)
    elif hook == "bggen":
        img, traj = op.freefine_background_generation(ori_img, OM.dilate_mask(ori // 255, 30), "empty scene", 7.5, 1.0, seed=7, **kw)
    else:
        oris, tgts = compose_masks()
        img, traj = op.freefine_compose([ori_img, img2], oris, tgts, coarse, ["a cup", "a dog"], 7.5, 1.0, end_step=50, num_step=50,
                                        start_step=15, seed=11, dil_factor=9, end_scale=0.5, **kw)
    return img, [t if t.ndim == 3 or hook == "edit" else t[0] for t in traj]


def test_g9_metric_schedules_n50():
    """the oracle's loops on the METRIC's schedules (N = 50; start_step 0 / 35 / 15 for the edit, 1 for background generation, 15 for the
    composition -- the reference's call sites) against the trajectories the REFERENCE produced (tools/gen_golden.py run_g9):
    ABSOLUTE latent L-inf at every step (|latent| <= 13 on these trajectories)."""
    from golden_cases import n50_cases
    g = np.load(os.path.join(GOLD, "g9_n50_loops.npz"))
    for name, hook, unet_name, planted, kw in n50_cases():
        img, traj = _run_oracle_n50(name, hook, unet_name, planted, kw)
        ref = g[f"{name}_traj"]
        assert len(traj) == len(ref)
        dev = max((a.float() - torch.from_numpy(b)).abs().max().item() for a, b in zip(traj, ref))
        assert dev < 2e-4, (name, dev)
        assert np.abs(img[::4, ::4].astype(int) - g[f"{name}_img"].astype(int)).max() <= 1, name
