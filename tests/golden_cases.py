"""Seeded inputs and case tables shared by tools/gen_golden.py (reference side, build container) and the tests
(oracle / HIP side).  Data recipes only."""
import numpy as np
import torch


def rng_tensor(seed, shape, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def rect_mask(h, w, r0, r1, c0, c1, value=1, dtype=np.uint8):
    m = np.zeros((h, w), dtype=dtype)
    m[r0:r1, c0:c1] = value
    return m


def mask_inputs(H=128, W=128):
    ori = rect_mask(H, W, 50, 76, 24, 50, 255)
    tgt = rect_mask(H, W, 50, 76, 40, 66, 255)
    draw = rect_mask(H, W, 46, 80, 36, 72, 1)
    cons_sup = (np.maximum(ori, tgt) // 255).astype(np.uint8)  # cons_area >= ori: no uint8 wrap
    cons_tgt = (tgt // 255).astype(np.uint8)                   # GeoBench-2D call site: wraps where ori & ~cons
    return ori, tgt, draw, cons_sup, cons_tgt


def synth_images(H=128, W=128):
    ori_img = np.random.default_rng(0).integers(0, 256, (H, W, 3), dtype=np.uint8)
    coarse = np.random.default_rng(1).integers(0, 256, (H, W, 3), dtype=np.uint8)
    img2 = np.random.default_rng(2).integers(0, 256, (H, W, 3), dtype=np.uint8)
    return ori_img, coarse, img2


def edit_cases():
    ori, tgt, draw, cons_sup, cons_tgt = mask_inputs()
    base = dict(draw_mask=draw, use_auto_draw=False, cons_area=None, reduce_inp_artifacts=False, end_step=8, num_step=10,
                end_scale=0.5, guidance_text="a cup", guidance_scale=7.5, eta=1.0)
    return [
        ("edit_tca_draw", "tiny", dict(base, method_type="tca", start_step=6)),
        ("edit_tca_auto", "tiny-conv", dict(method_type="tca", draw_mask=None, use_auto_draw=True, cons_area=cons_sup, reduce_inp_artifacts=True,
                                            end_step=10, num_step=10, start_step=7, end_scale=0.0, guidance_text="", guidance_scale=7.5, eta=1.0)),
        ("edit_mmsa", "tiny", dict(base, method_type="mmsa", start_step=7, guidance_scale=4.0, eta=0.0)),
        ("edit_mmsa_es", "tiny", dict(base, method_type="mmsa_es", start_step=6)),
        ("edit_ssa", "tiny", dict(base, method_type="ssa", start_step=8)),
        ("edit_sdsa", "tiny", dict(base, method_type="sdsa", start_step=8)),
        ("edit_tca_wrap", "tiny", dict(method_type="tca", draw_mask=None, use_auto_draw=True, cons_area=cons_tgt, reduce_inp_artifacts=True,
                                       end_step=10, num_step=10, start_step=7, end_scale=0.0, guidance_text="", guidance_scale=7.5, eta=1.0)),
    ]


BG_CASES = (("bg_tca", dict(method_type="tca", end_step=6, num_step=10, start_step=1, end_scale=0.5)),
            ("bg_mmsa", dict(method_type="mmsa", end_step=6, num_step=10, start_step=5, end_scale=0.5)))
CMP_CASES = (("cmp_tca", dict(method_type="tca", appearance_transfer=False, dil_completion=True)),
             ("cmp_app", dict(method_type="tca", appearance_transfer=True, dil_completion=False)))


def compose_masks():
    ori, tgt, *_ = mask_inputs()
    ori2, tgt2 = rect_mask(128, 128, 10, 40, 70, 110, 255), rect_mask(128, 128, 84, 118, 60, 100, 255)
    return [ori, ori2], [tgt, tgt2]


def oracle_pipe(unet_name="tiny", seed=0):
    from freefine_amd.text import ByteTokenizer, SyntheticTextEncoder, make_text_embed
    from oracle import sd_unet, sd_vae
    from oracle.pipeline import OraclePipeline
    cfg = sd_unet.unet_config(unet_name)
    unet = sd_unet.init_unet(cfg, seed=seed)
    vae = sd_vae.init_vae(sd_vae.vae_config("tiny"), seed=seed + 1)
    return OraclePipeline(unet, vae, make_text_embed(ByteTokenizer(), SyntheticTextEncoder(cfg.cross_attention_dim)))


BRANCH_CODE = {"plain": 0, "tca:tca": 1, "tca:mmsa": 1, "cross_local": 2, "ssa": 3, "sdsa": 3}


# ----------------------------------------------------------------------------------------------------------------
# The metric's own schedules (N = 50): SURVEY 8(a) call-site table.  G9 = the REFERENCE's loops on the tiny topology
# (tools/gen_golden.py run_g9), G10 = OraclePipeline at FULL size (tools/gen_fullsize_traj.py; the oracle is pinned by G1-G9).
# "planted": freefine_amd.weights.plant_denoiser_path on the seeded state, so that a schedule that starts from pure noise
# (start_step 0 / 1) is a denoising trajectory with O(1) latents instead of a 14.6x amplification of the DDPM noise.
# ----------------------------------------------------------------------------------------------------------------
def n50_cases():
    ori, tgt, draw, cons_sup, cons_tgt = mask_inputs()
    return [
        # (name, hook, unet, planted gain, kwargs)
        ("n50_edit_s0", "edit", "tiny", 3.0, dict(method_type="tca", draw_mask=draw, use_auto_draw=False, cons_area=None, reduce_inp_artifacts=False,
                                                  end_step=50, num_step=50, start_step=0, end_scale=0.0, guidance_text="a cup", guidance_scale=7.5, eta=1.0)),
        # freefine_batch_infer_2d.py:212-230 (non-wrapping variant: cons_area >= ori_mask)
        ("n50_edit_s35", "edit", "tiny", 0.0, dict(method_type="tca", draw_mask=None, use_auto_draw=True, cons_area=cons_sup, reduce_inp_artifacts=True,
                                                   end_step=50, num_step=50, start_step=35, end_scale=0.0, guidance_text="", guidance_scale=7.5, eta=1.0)),
        # freefine_batch_infer_3d_depth.py:144-162 (non-wrapping variant: cons_area >= ori_mask)
        ("n50_edit_s15", "edit", "tiny", 0.0, dict(method_type="tca", draw_mask=draw, use_auto_draw=False, cons_area=cons_sup, reduce_inp_artifacts=True,
                                                   end_step=50, num_step=50, start_step=15, end_scale=0.0, guidance_text="a cup", guidance_scale=7.5, eta=1.0)),
        # freefine_batch_infer_bggen_2d.py:149,166-180
        ("n50_bg_s1", "bggen", "tiny", 3.0, dict(method_type="tca", end_step=35, num_step=50, start_step=1, end_scale=0.5)),
        # Appearance_transfer.ipynb cell 5 / SURVEY 8d C4
        ("n50_cmp_s15", "compose", "tiny", 0.0, dict(method_type="tca", appearance_transfer=True, dil_completion=False)),
    ]


def tiny_state(unet_name, seed, planted):
    """the seeded tiny-topology UNet state of the goldens (+ the planted denoiser path when planted > 0)"""
    from oracle import sd_unet
    st = sd_unet.init_unet(sd_unet.unet_config(unet_name), seed=seed).state_dict()
    if planted > 0:
        from freefine_amd.config import UNetConfig
        from freefine_amd.weights import plant_denoiser_path
        st = plant_denoiser_path(st, UNetConfig.preset(unet_name), planted)
    return st


def fullsize_inputs():
    H = 512
    ori_img, coarse, _ = synth_images(H, H)
    ori, tgt, draw = rect_mask(H, H, 200, 304, 96, 200, 255), rect_mask(H, H, 200, 304, 160, 264, 255), rect_mask(H, H, 184, 320, 144, 288, 1)
    cons_sup = (np.maximum(ori, tgt) // 255).astype(np.uint8)
    return ori_img, coarse, ori, tgt, draw, cons_sup


def fullsize_cases():
    ori_img, coarse, ori, tgt, draw, cons_sup = fullsize_inputs()
    return {
        # the GeoBench-2D call site at full size: SD-2.1-base topology, 512^2, N = 50, start_step = 35 (15 + 15 forwards), default-init weights
        "fs_edit_s35": (0.0, dict(method_type="tca", draw_mask=None, use_auto_draw=True, cons_area=cons_sup, reduce_inp_artifacts=True,
                                  end_step=50, num_step=50, start_step=35, end_scale=0.0, guidance_text="", guidance_scale=7.5, eta=1.0)),
        # the metric's schedule at full size: N = 50, start_step = 0 (50 + 50 forwards), planted denoiser path
        "fs_edit_s0": (3.0, dict(method_type="tca", draw_mask=draw, use_auto_draw=False, cons_area=None, reduce_inp_artifacts=False,
                                 end_step=50, num_step=50, start_step=0, end_scale=0.0, guidance_text="a photo of a cup", guidance_scale=7.5, eta=1.0)),
        # BASELINE.json configs[0], "SD-2.1 single 512x512 object-reposition edit, 20 DDIM steps, CPU path": the CPU run itself (20 + 20 forwards)
        "fs_edit_n20": (3.0, dict(method_type="tca", draw_mask=draw, use_auto_draw=False, cons_area=None, reduce_inp_artifacts=False,
                                  end_step=20, num_step=20, start_step=0, end_scale=0.0, guidance_text="a photo of a cup", guidance_scale=7.5, eta=1.0)),
    }


def fullsize_hook_cases():
    """the other two hooks at full size on the metric's N = 50 schedules (name -> (hook, planted gain, kwargs)):
    background generation at start_step 1 (freefine_batch_infer_bggen_2d.py:149,166-180: 49 + 49 forwards) and the
    composition with R = 2 references at start_step 15 (Appearance_transfer.ipynb cell 5 / SURVEY 8d C4: 35 + 35 forwards)."""
    return {
        "fs_bg_s1": ("bggen", 3.0, dict(method_type="tca", end_step=35, num_step=50, start_step=1, end_scale=0.5)),
        "fs_cmp_s15": ("compose", 0.0, dict(method_type="tca", appearance_transfer=True, dil_completion=False, end_step=50, num_step=50,
                                            start_step=15, end_scale=0.5, dil_factor=9)),
    }


def fullsize_hook_inputs():
    """inputs of fullsize_hook_cases: the 512^2 images of fullsize_inputs + a third image, two (source, target) mask pairs"""
    H = 512
    ori_img, coarse, img2 = synth_images(H, H)
    ori, tgt = rect_mask(H, H, 200, 304, 96, 200, 255), rect_mask(H, H, 200, 304, 160, 264, 255)
    ori2, tgt2 = rect_mask(H, H, 40, 160, 280, 440, 255), rect_mask(H, H, 336, 472, 240, 400, 255)
    return ori_img, coarse, img2, [ori, ori2], [tgt, tgt2]
