"""CPU: pin the oracle (oracle/) against golden vectors produced by the REFERENCE's own code (tools/gen_golden.py,
run in the build container with /root/reference imported read-only).  Nothing here touches /root/reference."""
import os

import numpy as np
import pytest
import torch

from oracle import attention_modulation as OA
from oracle import masks as OM
from oracle import scheduler as OS

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
torch.set_grad_enabled(False)


def rng_tensor(seed, shape, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def rect_mask(h, w, r0, r1, c0, c1, value=1, dtype=np.uint8):
    m = np.zeros((h, w), dtype=dtype)
    m[r0:r1, c0:c1] = value
    return m


def g1_masks(kind):
    H = W = 128
    ms = [rect_mask(H, W, 24, 72, 16, 64), rect_mask(H, W, 40, 100, 56, 120), rect_mask(H, W, 80, 120, 8, 48), rect_mask(H, W, 8, 40, 72, 104)]
    f = (lambda m: torch.tensor(m.astype(np.float32))) if kind == "float" else (lambda m: torch.tensor(m))
    src, tgt, src2, tgt2 = (f(m) for m in ms)
    return src, tgt, src2, tgt2


def g1_inputs(ci, heads, d, S):
    C = heads * d
    q4, k4, v4 = (rng_tensor(100 * ci + j, (4, S, C)) for j in range(3))
    kt, vt = rng_tensor(100 * ci + 3, (4, 77, C)), rng_tensor(100 * ci + 4, (4, 77, C))
    kc, vc = rng_tensor(100 * ci + 5, (6, 77, C)), rng_tensor(100 * ci + 6, (6, 77, C))
    return q4, k4, v4, kt, vt, kc, vc


def g1_oracle_outputs(ci, heads, d, S, kind, cg):
    src, tgt, src2, tgt2 = g1_masks(kind)
    q4, k4, v4, kt, vt, kc, vc = g1_inputs(ci, heads, d, S)
    scale = d ** -0.5
    bgm = 1 - torch.maximum(tgt, tgt2)
    return {
        "edit_tca": OA.tca_edit(q4, k4, v4, heads, scale, tgt, src, "tca", cg),
        "edit_mmsa": OA.tca_edit(q4, k4, v4, heads, scale, tgt, src, "mmsa", cg),
        "bg_tca": OA.tca_bg(q4, k4, v4, heads, scale, tgt, "tca", cg),
        "bg_mmsa": OA.tca_bg(q4, k4, v4, heads, scale, tgt, "mmsa", cg),
        "compose_tca": OA.tca_compose(q4, k4, v4, heads, scale, torch.stack([src, src2]), torch.stack([tgt, tgt2]), "tca", cg),
        "compose_mmsa": OA.tca_compose(q4, k4, v4, heads, scale, torch.stack([src, src2]), torch.stack([tgt, tgt2]), "mmsa", cg),
        "ssa": OA.shared_kv_attention(q4, k4, v4, heads, scale, None),
        "sdsa": OA.shared_kv_attention(q4, k4, v4, heads, scale, src),
        "cross_local": OA.cross_local(q4, kt, vt, heads, scale, tgt),
        "cross_compose": OA.cross_local_compose(q4, kc, vc, heads, scale, torch.stack([tgt, tgt2, bgm]), 3),
        "plain": OA.plain_attention(q4, k4, v4, heads, scale),
    }


def test_g1_attention_modulation():
    g = np.load(os.path.join(GOLD, "g1_attention.npz"))
    ncase = len([k for k in g.files if k.endswith("_meta")])
    assert ncase == 10
    nup = 0
    for ci in range(ncase):
        heads, d, S, is_float, upcast = (int(v) for v in g[f"c{ci}_meta"])
        nup += upcast       # cases 8, 9: the reference ran with upcast_attention = upcast_softmax = True (fp32 model: same maths, other code path)
        cg = float(g[f"c{ci}_cg"][0])
        outs = g1_oracle_outputs(ci, heads, d, S, "float" if is_float else "uint8", cg)
        for name, o in outs.items():
            sub = torch.from_numpy(g[f"c{ci}_{name}_sub"])
            assert (o[:, ::5, ::3] - sub).abs().max().item() < 5e-6, (ci, name)
            s1, s2 = g[f"c{ci}_{name}_sum"]
            assert abs(o.double().sum().item() - s1) < 1e-3 * (1 + abs(s1)), (ci, name)
            assert abs(o.double().pow(2).sum().item() - s2) < 1e-4 * (1 + abs(s2)), (ci, name)
    assert nup == 2


def test_g3_scheduler_steps():
    g = np.load(os.path.join(GOLD, "g3_scheduler.npz"))
    sched = OS.DDIMSchedule()
    assert np.array_equal(sched.alphas_cumprod.numpy(), g["alphas_cumprod"])
    mask_of = {"f01": lambda: torch.tensor(rect_mask(16, 16, 3, 9, 4, 12).astype(np.float32)),
               "u01": lambda: torch.tensor(rect_mask(16, 16, 3, 9, 4, 12)),
               "u2": lambda: torch.tensor(rect_mask(16, 16, 3, 9, 4, 12, value=2))}
    for N, ts in ((50, (981, 501, 21, 1)), (20, (951, 501, 1))):
        sched.set_timesteps(N)
        assert np.array_equal(sched.timesteps.numpy(), g[f"timesteps_{N}"])
        for ti, t in enumerate(ts):
            eps, x = rng_tensor(7 + ti, (2, 4, 16, 16)), rng_tensor(17 + ti, (2, 4, 16, 16))
            xn, _ = OS.inv_step(sched, eps, t, x)
            assert np.array_equal(xn.numpy(), g[f"inv_{N}_{t}"])
            for eta in (0.0, 1.0):
                for mk in ("f01", "u01", "u2"):
                    torch.manual_seed(5)
                    noise = torch.randn(eps.shape) if eta > 0 else None
                    xp, _ = OS.ctrl_step(sched, eps, t, x, mask_of[mk](), eta, noise)
                    assert np.array_equal(xp.numpy(), g[f"ctrl_{N}_{t}_{eta}_{mk}"]), (N, t, eta, mk)
            torch.manual_seed(6)
            xp1, _ = OS.ctrl_step(sched, eps[:1], t, x[:1], mask_of["u01"](), 1.0, torch.randn(eps[:1].shape))
            assert np.array_equal(xp1.numpy(), g[f"ctrl1_{N}_{t}"])
    for k in [k for k in g.files if k.startswith("lp_")]:
        i, s0, e, n, es = k[3:].split("_")
        assert abs(OS.linear_param(int(i), int(s0), int(e), int(n), float(es)) - g[k][0]) < 1e-12
    with pytest.raises(ValueError):
        OS.linear_param(3, 5, 8, 10)
    # the uint8 value-2 mask makes (1 - mask) wrap to 255: the golden holds the reference's own wrap
    assert np.abs(g["ctrl_50_981_1.0_u2"]).max() > 10 * np.abs(g["ctrl_50_981_1.0_u01"]).max()


def mask_inputs(H=128, W=128):
    ori = rect_mask(H, W, 50, 76, 24, 50, 255)
    tgt = rect_mask(H, W, 50, 76, 40, 66, 255)
    draw = rect_mask(H, W, 46, 80, 36, 72, 1)
    cons_sup = (np.maximum(ori, tgt) // 255).astype(np.uint8)
    cons_tgt = (tgt // 255).astype(np.uint8)
    return ori, tgt, draw, cons_sup, cons_tgt


def test_g4_mask_preparation():
    g = np.load(os.path.join(GOLD, "g4_masks.npz"))
    ori, tgt, draw, cons_sup, cons_tgt = mask_inputs()
    combos = [("draw", dict(use_auto_draw=False, reduce_inp_artifacts=False, cons_area=None), draw),
              ("draw_red", dict(use_auto_draw=False, reduce_inp_artifacts=True, cons_area=cons_sup), draw),
              ("auto", dict(use_auto_draw=True, reduce_inp_artifacts=False, cons_area=cons_sup), None),
              ("auto_red", dict(use_auto_draw=True, reduce_inp_artifacts=True, cons_area=cons_sup), None),
              ("auto_red_wrap", dict(use_auto_draw=True, reduce_inp_artifacts=True, cons_area=cons_tgt), None)]
    for name, kw, dm in combos:
        o = OM.prepare_various_mask(tgt.copy(), ori.copy(), None if dm is None else dm.copy(), 128, 128, (16, 16), **kw)
        for j, t in enumerate(o):
            ref = g[f"{name}_{j}"]
            assert t.numpy().dtype == ref.dtype == np.uint8 and np.array_equal(t.numpy(), ref), (name, j)
    # the GeoBench-2D call site (cons_area = target mask) wraps: values outside {0,1} appear in the 64x64-level masks
    assert set(np.unique(g["auto_red_wrap_4"])) - {0, 1}
    o = OM.prepare_mask_bggen(OM.dilate_mask(ori // 255, 30), 128, 128, (16, 16))
    for j, t in enumerate(o):
        assert np.array_equal(t.numpy(), g[f"bggen_{j}"])
    ori2, tgt2 = rect_mask(128, 128, 10, 40, 70, 110, 255), rect_mask(128, 128, 84, 118, 60, 100, 255)
    for name, kw in (("cmp", dict()), ("cmp_dil", dict(dil_completion=True)), ("cmp_app", dict(appearance_transfer=True, dil_factor=9)),
                     ("cmp_draw", dict(draw_mask=[draw, rect_mask(128, 128, 80, 124, 56, 108, 1)]))):
        o = OM.prepare_composition_masks([ori, ori2], [tgt, tgt2], 128, 128, (16, 16), **kw)
        for j, t in enumerate(o):
            assert np.array_equal(t.numpy(), g[f"{name}_{j}"]), (name, j)


def test_g6_unet_vs_intree_ldm():
    from oracle import sd_unet
    g = np.load(os.path.join(GOLD, "g6_ldm_unet.npz"))
    cfg = sd_unet.unet_config("tiny-conv")
    cfg.norm_num_groups, cfg.heads = 32, (4, 4, 4, 4)
    net = sd_unet.init_unet(cfg, seed=3)
    x, ctx = rng_tensor(31, (2, 4, 16, 16)), rng_tensor(32, (2, 77, cfg.cross_attention_dim))
    y = net(x, torch.tensor(int(g["t"][0])), ctx)
    assert (y - torch.from_numpy(g["y"])).abs().max().item() < 2e-5


def test_g6b_unet_linear_proj_and_per_level_heads_vs_intree_sgm():
    """SD-2.1's deltas from SD-1.x -- LINEAR proj_in / proj_out and head counts that follow a constant head width -- pinned against
    the in-tree Stability sgm UNetModel (generative-models/sgm/modules/diffusionmodules/openaimodel.py, num_head_channels=16,
    use_linear_in_transformer=True)."""
    from oracle import sd_unet
    g = np.load(os.path.join(GOLD, "g6b_sgm_unet.npz"))
    cfg = sd_unet.unet_config("tiny")
    cfg.norm_num_groups, cfg.heads = 32, tuple(int(h) for h in g["heads"])
    assert cfg.use_linear_projection and cfg.heads == (2, 4, 8, 8)
    net = sd_unet.init_unet(cfg, seed=5)
    x, ctx = rng_tensor(41, (2, 4, 16, 16)), rng_tensor(42, (2, 77, cfg.cross_attention_dim))
    y = net(x, torch.tensor(int(g["t"][0])), ctx)
    assert (y - torch.from_numpy(g["y"])).abs().max().item() < 2e-5


def test_g7_vae_vs_intree_ldm_encoder_decoder():
    """the VAE restatement against the in-tree CompVis Encoder / Decoder (evaluation/MotionGuidance/ldm/modules/diffusionmodules/
    model.py:368, 462): encoder moments (before quant_conv) and decoder output on seeded inputs, ragged 64x48 image."""
    from types import SimpleNamespace
    from oracle import sd_vae
    g = np.load(os.path.join(GOLD, "g7_ldm_vae.npz"))
    cfg = SimpleNamespace(name="g7", in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(32, 64, 128, 128),
                          layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)
    vae = sd_vae.init_vae(cfg, seed=9)
    x, z = rng_tensor(51, (2, 3, 64, 48)), rng_tensor(52, (2, 4, 8, 6))
    assert (vae.encoder(x) - torch.from_numpy(g["moments"])).abs().max().item() < 2e-5
    assert (vae.decoder(z) - torch.from_numpy(g["dec"])).abs().max().item() < 5e-5


def test_g8_depth_network_vs_reference_dinov2_and_dpt_head():
    """oracle/dpt.py (DINOv2 ViT + DPT head + the forward of DPT_DINOv2) against outputs of the reference's own modules
    (torchhub/facebookresearch_dinov2_main/vision_transformer.py, depth_anything/dpt.py:22-167) on seeded weights: a square input at the
    positional embedding's own grid, non-square inputs (bicubic interpolation of the positional embedding), two encoder sizes."""
    from oracle import dpt as OD
    g = np.load(os.path.join(GOLD, "g8_dpt.npz"))
    for name, img_size, sizes in (("tiny", 70, ((70, 70), (56, 98))), ("mini", 518, ((42, 70),))):
        cfg = OD.dpt_config(name)
        cfg.img_size = img_size
        st = OD.dpt_synthetic_state(cfg, seed=3 + len(name))
        for (H, W) in sizes:
            x = rng_tensor(80 + H + W, (2, 3, H, W))
            feats = OD.vit_features(cfg, st, x, 4)
            d = OD.depth_forward(cfg, st, x)
            key = f"{name}_{H}x{W}"
            assert (feats[3][0] - torch.from_numpy(g[key + "_feat3"])).abs().max().item() < 2e-5, key
            assert d.shape == (2, H, W) and (d - torch.from_numpy(g[key + "_depth"])).abs().max().item() < 2e-5, key
