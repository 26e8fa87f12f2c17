"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own code (imported read-only from
/root/reference, see tools/ref_harness.py) on seeded inputs.  Run in the build container only:

    python tools/gen_golden.py            # writes tests/golden/*.npz and prints oracle-vs-reference deviations

The fixtures hold inputs recipes (seeds) and expected outputs -- data, no reference source.  tests/test_oracle_golden.py
checks the oracle (oracle/) against them on CPU; the GPU parity tests then compare the HIP path with the oracle.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import ref_harness as RH  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)


def rng_tensor(seed, shape, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def rect_mask(h, w, r0, r1, c0, c1, value=1, dtype=np.uint8):
    m = np.zeros((h, w), dtype=dtype)
    m[r0:r1, c0:c1] = value
    return m


# --------------------------------------------------------------------------------------------------------------
# G1: attention-modulation known-answer tests, straight from Attention_Modulator
# --------------------------------------------------------------------------------------------------------------
def g1_cases():
    cases = []
    for (heads, d) in ((8, 8), (5, 16)):
        for S in (64, 256):
            for mask_kind in ("uint8", "float"):
                cases.append(dict(heads=heads, d=d, S=S, mask_kind=mask_kind, upcast=False))
    # SD-2.1's unet config sets upcast_attention: the fp32 casts of get_attention_scores (attention.py:776-778, 798-799) taken in an
    # fp32 model (the only precision the masked branches work in with that flag)
    cases.append(dict(heads=5, d=16, S=64, mask_kind="uint8", upcast=True))
    cases.append(dict(heads=8, d=8, S=256, mask_kind="float", upcast=True))
    return cases


def g1_masks(kind, seed):
    H = W = 128
    src = rect_mask(H, W, 24, 72, 16, 64)
    tgt = rect_mask(H, W, 40, 100, 56, 120)
    tgt2 = rect_mask(H, W, 8, 40, 72, 104)
    src2 = rect_mask(H, W, 80, 120, 8, 48)
    if kind == "float":
        f = lambda m: torch.tensor(m.astype(np.float32))
    else:
        f = lambda m: torch.tensor(m)
    return f(src), f(tgt), f(src2), f(tgt2)


def run_g1(A):
    out = {}
    devs = []
    from oracle import attention_modulation as OA
    for ci, c in enumerate(g1_cases()):
        heads, d, S = c["heads"], c["d"], c["S"]
        C = heads * d
        src, tgt, src2, tgt2 = g1_masks(c["mask_kind"], ci)
        q4, k4, v4 = (rng_tensor(100 * ci + j, (4, S, C)) for j in range(3))
        kt, vt = rng_tensor(100 * ci + 3, (4, 77, C)), rng_tensor(100 * ci + 4, (4, 77, C))
        scale = d ** -0.5
        cg = 0.3 + 0.05 * ci

        def fresh(method=None):
            m = A.Attention_Modulator(start_layer=10)
            m.heads, m.scale, m.upcast_attention, m.upcast_softmax = heads, scale, c["upcast"], c["upcast"]
            m.num_att_layers = 32
            m.cur_att_layer = 20  # block 10, inside layer_idx
            m.method, m.context_guidance = method, cg
            m.fg_retain_mask, m.fg_retain_mask_st2, m.fg_ref_mask, m.local_edit_region = tgt.clone(), tgt.clone(), src.clone(), tgt.clone()
            return m

        res = {}
        for method in ("tca", "mmsa"):
            res[f"edit_{method}"] = fresh(method).Temporal_contextal_attention(q4.clone(), k4.clone(), v4.clone(), False, "up")
            res[f"bg_{method}"] = fresh(method).Temporal_contextal_attention_bg(q4.clone(), k4.clone(), v4.clone(), False, "up")
            m = fresh(method)
            m.src_masks, m.tgt_masks = torch.stack([src, src2]), torch.stack([tgt, tgt2, 1 - torch.maximum(tgt, tgt2)])
            res[f"compose_{method}"] = m.Temporal_contextal_attention_compose(q4.clone(), k4.clone(), v4.clone(), False, "up")
        res["ssa"] = fresh("ssa").style_align_share_attention(q4.clone(), k4.clone(), v4.clone(), False, "up")
        res["sdsa"] = fresh("sdsa").style_align_share_attention(q4.clone(), k4.clone(), v4.clone(), False, "up")
        res["cross_local"] = fresh().modulate_local_cross_attn(q4.clone(), kt.clone(), vt.clone(), True, "up")
        m = fresh()
        m.tgt_masks = torch.stack([tgt, tgt2, 1 - torch.maximum(tgt, tgt2)])
        m.prompt_length = 3
        kc, vc = rng_tensor(100 * ci + 5, (3 + 3, 77, C)), rng_tensor(100 * ci + 6, (3 + 3, 77, C))
        res["cross_compose"] = m.modulate_local_cross_attn_compose(q4.clone(), kc.clone(), vc.clone(), True, "up")
        # plain branch of get_attention_scores + bmm
        m = fresh()
        probs = m.get_attention_scores(m.head_to_batch_dim(q4), m.head_to_batch_dim(k4), None)
        res["plain"] = m.batch_to_head_dim(torch.bmm(probs, m.head_to_batch_dim(v4)))

        # oracle deviations
        orc = {
            "edit_tca": OA.tca_edit(q4, k4, v4, heads, scale, tgt, src, "tca", cg),
            "edit_mmsa": OA.tca_edit(q4, k4, v4, heads, scale, tgt, src, "mmsa", cg),
            "bg_tca": OA.tca_bg(q4, k4, v4, heads, scale, tgt, "tca", cg),
            "bg_mmsa": OA.tca_bg(q4, k4, v4, heads, scale, tgt, "mmsa", cg),
            "compose_tca": OA.tca_compose(q4, k4, v4, heads, scale, torch.stack([src, src2]), torch.stack([tgt, tgt2]), "tca", cg),
            "compose_mmsa": OA.tca_compose(q4, k4, v4, heads, scale, torch.stack([src, src2]), torch.stack([tgt, tgt2]), "mmsa", cg),
            "ssa": OA.shared_kv_attention(q4, k4, v4, heads, scale, None),
            "sdsa": OA.shared_kv_attention(q4, k4, v4, heads, scale, src),
            "cross_local": OA.cross_local(q4, kt, vt, heads, scale, tgt),
            "cross_compose": OA.cross_local_compose(q4, kc, vc, heads, scale, torch.stack([tgt, tgt2, 1 - torch.maximum(tgt, tgt2)]), 3),
            "plain": OA.plain_attention(q4, k4, v4, heads, scale),
        }
        for name, r in res.items():
            r = r.float()
            devs.append((ci, name, (orc[name] - r).abs().max().item()))
            out[f"c{ci}_{name}_sub"] = r[:, ::5, ::3].numpy().copy()
            out[f"c{ci}_{name}_sum"] = np.array([r.double().sum().item(), r.double().pow(2).sum().item()])
        out[f"c{ci}_meta"] = np.array([heads, d, S, int(c["mask_kind"] == "float"), int(c["upcast"])])
        out[f"c{ci}_cg"] = np.array([cg])
    np.savez_compressed(os.path.join(GOLD, "g1_attention.npz"), **out)
    worst = max(devs, key=lambda t: t[2])
    print(f"[G1] {len(devs)} outputs, worst oracle-vs-reference deviation {worst}")
    return devs


# --------------------------------------------------------------------------------------------------------------
# G3: scheduler steps;  G4: mask preparation
# --------------------------------------------------------------------------------------------------------------
def make_ref_pipe(A, Mo, hook="edit", unet_name="tiny", seed=0):
    from oracle import scheduler as OS
    from oracle import sd_unet, sd_vae
    from freefine_amd.text import ByteTokenizer, SyntheticTextEncoder
    cfg = sd_unet.unet_config(unet_name)
    unet = sd_unet.init_unet(cfg, seed=seed)
    vae = sd_vae.init_vae(sd_vae.vae_config("tiny"), seed=seed + 1)
    tok, enc = ByteTokenizer(), SyntheticTextEncoder(cfg.cross_attention_dim)
    sched = OS.DDIMSchedule()
    return RH.build_reference_pipeline(A, Mo, unet, vae, tok, enc, sched, hook=hook)


def run_g3(A, Mo):
    from oracle import scheduler as OS
    p = make_ref_pipe(A, Mo)
    out, worst = {}, 0.0
    sched = OS.DDIMSchedule()
    for N in (50, 20):
        p.scheduler.set_timesteps(N)
        sched.set_timesteps(N)
        out[f"timesteps_{N}"] = p.scheduler.timesteps.numpy()
        for ti, t in enumerate((981, 501, 21, 1) if N == 50 else (951, 501, 1)):
            eps, x = rng_tensor(7 + ti, (2, 4, 16, 16)), rng_tensor(17 + ti, (2, 4, 16, 16))
            xn, p0 = p.inv_step(eps, t, x)
            out[f"inv_{N}_{t}"] = xn.numpy()
            o_xn, _ = OS.inv_step(sched, eps, t, x)
            worst = max(worst, (o_xn - xn).abs().max().item())
            for eta in (0.0, 1.0):
                for mk, mask in (("f01", torch.tensor(rect_mask(16, 16, 3, 9, 4, 12).astype(np.float32))),
                                 ("u01", torch.tensor(rect_mask(16, 16, 3, 9, 4, 12))),
                                 ("u2", torch.tensor(rect_mask(16, 16, 3, 9, 4, 12, value=2)))):
                    torch.manual_seed(5)
                    xp, _ = p.ctrl_step(eps, t, x, mask, eta=eta)
                    out[f"ctrl_{N}_{t}_{eta}_{mk}"] = xp.numpy()
                    torch.manual_seed(5)
                    noise = torch.randn(eps.shape) if eta > 0 else None
                    o_xp, _ = OS.ctrl_step(sched, eps, t, x, mask, eta, noise)
                    worst = max(worst, (o_xp - xp).abs().max().item())
            # compose form: one row, [h,w] mask
            torch.manual_seed(6)
            xp1, _ = p.ctrl_step(eps[:1], t, x[:1], torch.tensor(rect_mask(16, 16, 3, 9, 4, 12)), eta=1.0)
            out[f"ctrl1_{N}_{t}"] = xp1.numpy()
            torch.manual_seed(6)
            o_xp1, _ = OS.ctrl_step(sched, eps[:1], t, x[:1], torch.tensor(rect_mask(16, 16, 3, 9, 4, 12)), 1.0, torch.randn(eps[:1].shape))
            worst = max(worst, (o_xp1 - xp1).abs().max().item())
    for (i, s0, e, n, es) in ((35, 35, 50, 50, 0.0), (42, 35, 50, 50, 0.0), (50, 35, 50, 50, 0.0), (1, 1, 35, 50, 0.5), (20, 1, 35, 50, 0.5),
                              (40, 1, 35, 50, 0.5), (30, 15, 50, 50, 0.5)):
        v = p.linear_param(i, s0, e, n, end_scale=es)
        out[f"lp_{i}_{s0}_{e}_{n}_{es}"] = np.array([v])
        worst = max(worst, abs(OS.linear_param(i, s0, e, n, es) - v))
    out["alphas_cumprod"] = p.scheduler.alphas_cumprod.numpy()
    np.savez_compressed(os.path.join(GOLD, "g3_scheduler.npz"), **out)
    print(f"[G3] worst oracle-vs-reference deviation {worst:.3e}")


def mask_inputs(H=128, W=128):
    ori = rect_mask(H, W, 50, 76, 24, 50, 255)
    tgt = rect_mask(H, W, 50, 76, 40, 66, 255)
    draw = rect_mask(H, W, 46, 80, 36, 72, 1)
    cons_sup = np.maximum(ori, tgt) // 255  # cons_area >= ori: no wrap
    cons_tgt = tgt // 255                   # GeoBench-2D call site: cons_area = target mask -> uint8 wrap where ori & ~cons
    return ori, tgt, draw, cons_sup.astype(np.uint8), cons_tgt.astype(np.uint8)


def run_g4(A, Mo):
    from oracle import masks as OM
    p = make_ref_pipe(A, Mo)
    ori, tgt, draw, cons_sup, cons_tgt = mask_inputs()
    init_code = torch.zeros(2, 4, 16, 16)
    out, worst = {}, 0.0
    combos = [("draw", dict(use_auto_draw=False, reduce_inp_artifacts=False, cons_area=None), draw),
              ("draw_red", dict(use_auto_draw=False, reduce_inp_artifacts=True, cons_area=cons_sup), draw),
              ("auto", dict(use_auto_draw=True, reduce_inp_artifacts=False, cons_area=cons_sup), None),
              ("auto_red", dict(use_auto_draw=True, reduce_inp_artifacts=True, cons_area=cons_sup), None),
              ("auto_red_wrap", dict(use_auto_draw=True, reduce_inp_artifacts=True, cons_area=cons_tgt), None)]
    for name, kw, dm in combos:
        r = p.prepare_various_mask(tgt.copy(), ori.copy(), None if dm is None else dm.copy(), 128, 128, init_code, verbose=True, **kw)
        o = OM.prepare_various_mask(tgt.copy(), ori.copy(), None if dm is None else dm.copy(), 128, 128, (16, 16), **kw)
        for j, (a, b) in enumerate(zip(r, o)):
            assert a.dtype == b.dtype, (name, j, a.dtype, b.dtype)
            worst = max(worst, (a.float() - b.float()).abs().max().item())
            out[f"{name}_{j}"] = a.numpy()
    r = p.prepare_mask_bggen(p.dilate_mask(ori // 255, 30), 128, 128, init_code)
    o = OM.prepare_mask_bggen(OM.dilate_mask(ori // 255, 30), 128, 128, (16, 16))
    for j, (a, b) in enumerate(zip(r, o)):
        worst = max(worst, (a.float() - b.float()).abs().max().item())
        out[f"bggen_{j}"] = a.numpy()
    ori2, tgt2 = rect_mask(128, 128, 10, 40, 70, 110, 255), rect_mask(128, 128, 84, 118, 60, 100, 255)
    for name, kw in (("cmp", dict()), ("cmp_dil", dict(dil_completion=True)), ("cmp_app", dict(appearance_transfer=True, dil_factor=9)),
                     ("cmp_draw", dict(draw_mask=[draw, rect_mask(128, 128, 80, 124, 56, 108, 1)]))):
        r = p.prepare_composition_masks([ori, ori2], [tgt, tgt2], 128, 128, init_code, **kw)
        o = OM.prepare_composition_masks([ori, ori2], [tgt, tgt2], 128, 128, (16, 16), **kw)
        for j, (a, b) in enumerate(zip(r, o)):
            assert a.dtype == b.dtype and a.shape == b.shape, (name, j)
            worst = max(worst, (a.float() - b.float()).abs().max().item())
            out[f"{name}_{j}"] = a.numpy()
    np.savez_compressed(os.path.join(GOLD, "g4_masks.npz"), **out)
    print(f"[G4] worst oracle-vs-reference deviation {worst:.3e}")


# --------------------------------------------------------------------------------------------------------------
# G5: loop-level trajectories with the tiny deterministic UNet/VAE;  G2: (step, block) -> branch table
# --------------------------------------------------------------------------------------------------------------
from tests.golden_cases import BG_CASES, CMP_CASES, compose_masks, edit_cases, oracle_pipe, synth_images  # noqa: E402


def traj_arrays(prefix, traj, img):
    d = {f"{prefix}_traj": np.stack([t.numpy() if t.ndim == 4 else t[None].numpy() for t in traj]) if traj[0].ndim == traj[-1].ndim
         else np.stack([(t if t.ndim == 3 else t[0]).numpy() for t in traj])}
    d[f"{prefix}_img"] = np.asarray(img)[::4, ::4].copy()
    return d


def run_g5(A, Mo):
    ori_img, coarse, img2 = synth_images()
    ori, tgt, draw, cons_sup, cons_tgt = mask_inputs()
    out = {}
    report = []
    for name, unet_name, kw in edit_cases():
        p = make_ref_pipe(A, Mo, "edit", unet_name)
        trace = []
        _instrument(p.controller, trace)
        kw_ref = dict(kw)
        text, gs, eta = kw_ref.pop("guidance_text"), kw_ref.pop("guidance_scale"), kw_ref.pop("eta")
        cap = {}
        orig_fs = p.forward_sampling

        def fs(*a, **k):
            k["return_intermediates"] = True
            im, lst = orig_fs(*a, **k)
            cap["traj"] = lst
            return im, None
        p.forward_sampling = fs
        img_e, img_r = p.FreeFine_generation(ori_img, ori, coarse, tgt, text, gs, eta, verbose=True, return_ori=True, seed=42, **kw_ref)
        traj = cap["traj"]
        op = oracle_pipe(unet_name)
        o_e, o_r, o_traj = op.freefine_generation(ori_img, ori, coarse, tgt, text, gs, eta, seed=42, **kw_ref)
        dev = max(_nan_aware_dev(a, b) for a, b in zip(traj, o_traj))
        report.append((name, dev, int(np.abs(img_e.astype(int) - o_e.astype(int)).max())))
        out[f"{name}_traj"] = torch.stack(traj).numpy()
        out[f"{name}_img"] = img_e[::4, ::4].copy()
        out[f"{name}_ref_img"] = img_r[::4, ::4].copy()
        out[f"{name}_trace"] = np.array(_encode_trace(trace), dtype=np.float64)
    # --- background generation -----------------------------------------------------------------------------
    for name, kw in BG_CASES:
        p = make_ref_pipe(A, Mo, "bggen", "tiny")
        hole = p.dilate_mask(ori // 255, 30)
        cap = {}
        orig = p.forward_sampling_background_gen

        def fsb(*a, **k):
            k["return_intermediates"] = True
            im, lst = orig(*a, **k)
            cap["traj"] = lst
            return im, None
        p.forward_sampling_background_gen = fsb
        img = p.FreeFine_background_generation(ori_img, hole, "empty scene", 3.5, 1.0, verbose=True, seed=7, **kw)
        op = oracle_pipe("tiny")
        o_img, o_traj = op.freefine_background_generation(ori_img, hole, "empty scene", 3.5, 1.0, seed=7, **kw)
        traj = [t if t.ndim == 3 else t[0] for t in cap["traj"]]
        o_traj = [t if t.ndim == 3 else t[0] for t in o_traj]
        dev = max(_nan_aware_dev(a, b) for a, b in zip(traj, o_traj))
        report.append((name, dev, int(np.abs(img.astype(int) - o_img.astype(int)).max())))
        out[f"{name}_traj"] = torch.stack(traj).numpy()
        out[f"{name}_img"] = img[::4, ::4].copy()
    # --- composition (R=2), driven one level below the broken public wrapper ---------------------
    (ori, ori2), (tgt, tgt2) = compose_masks()
    for name, kw in CMP_CASES:
        p = make_ref_pipe(A, Mo, "compose", "tiny")
        Mo.seed_everything(11)
        lst = p.DDIM_inversion_func_compose(img=coarse, compose_imgs=[ori_img, img2], prompt="", num_step=10, start_step=6, verbose=True)
        cap = {}
        orig = p.forward_sampling_compose

        def fsc(*a, **k):
            k["return_intermediates"] = True
            im, l2 = orig(*a, **k)
            cap["traj"] = l2
            return im, None
        p.forward_sampling_compose = fsc
        img, _ = p.Details_Preserving_regeneration_compose(coarse, lst, ["a cup", "a dog"], [ori, ori2], [tgt, tgt2], None, num_steps=10,
                                                           start_step=6, end_step=8, eta=1.0, guidance_scale=7.5, verbose=True,
                                                           dil_factor=9, end_scale=0.5, **kw)
        op = oracle_pipe("tiny")
        o_img, o_traj = op.freefine_compose([ori_img, img2], [ori, ori2], [tgt, tgt2], coarse, ["a cup", "a dog"], 7.5, 1.0, end_step=8, num_step=10,
                                            start_step=6, seed=11, dil_factor=9, end_scale=0.5, **kw)
        traj = [t if t.ndim == 3 else t[0] for t in cap["traj"]]
        o_traj = [t if t.ndim == 3 else t[0] for t in o_traj]
        dev = max(_nan_aware_dev(a, b) for a, b in zip(traj, o_traj))
        report.append((name, dev, int(np.abs(img.astype(int) - o_img.astype(int)).max())))
        out[f"{name}_traj"] = torch.stack(traj).numpy()
        out[f"{name}_img"] = img[::4, ::4].copy()
    np.savez_compressed(os.path.join(GOLD, "g5_loops.npz"), **out)
    for r in report:
        print(f"[G5] {r[0]:16s} latent-trajectory deviation {r[1]:.3e}   uint8 image max diff {r[2]}")


def _nan_aware_dev(a, b):
    a, b = a.float(), b.float()
    fa, fb = torch.isfinite(a), torch.isfinite(b)
    if not torch.equal(fa, fb):
        return float("inf")
    if fa.sum() == 0:
        return 0.0
    return ((a - b)[fa].abs().max() / (1.0 + b[fa].abs().max())).item()


BRANCH_CODE = {"plain": 0, "tca": 1, "cross_local": 2, "style": 3}


def _instrument(controller, trace):
    """log (cur_step, cur_att_layer, branch code, context_guidance) for every attention call of the reference."""
    def wrap(name, code):
        orig = getattr(controller, name)

        def f(*a, **k):
            in_layer = (controller.cur_att_layer // 2) in controller.layer_idx
            c = code if (code != 1 or in_layer) else 0
            cg = controller.context_guidance if (c == 1 and controller.method == "tca") else -1.0
            trace.append((controller.cur_step, controller.cur_att_layer, c, -1.0 if cg is None else float(cg)))
            return orig(*a, **k)
        setattr(controller, name, f)
    wrap("Temporal_contextal_attention", 1)
    wrap("modulate_local_cross_attn", 2)
    wrap("style_align_share_attention", 3)
    orig_call = controller.__class__.__call__

    def call(self, attn, is_cross, place):
        trace.append((self.cur_step, self.cur_att_layer, 0, -1.0))
        return orig_call(self, attn, is_cross, place)
    controller.__class__ = type("InstrumentedModulator", (controller.__class__,), {"__call__": call})


def _encode_trace(trace):
    return [list(t) for t in trace]


# --------------------------------------------------------------------------------------------------------------
# G6: UNet arithmetic against the in-tree CompVis UNet (key-mapped seeded weights)
# --------------------------------------------------------------------------------------------------------------
def ldm_key_map(cfg):
    """diffusers-layout name prefix -> ldm UNetModel name prefix."""
    m = {"time_embedding.linear_1": "time_embed.0", "time_embedding.linear_2": "time_embed.2", "conv_in": "input_blocks.0.0",
         "conv_norm_out": "out.0", "conv_out": "out.2", "mid_block.resnets.0": "middle_block.0", "mid_block.attentions.0": "middle_block.1",
         "mid_block.resnets.1": "middle_block.2"}
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block):
            m[f"down_blocks.{i}.resnets.{j}"] = f"input_blocks.{3 * i + j + 1}.0"
            if cfg.down_has_attn[i]:
                m[f"down_blocks.{i}.attentions.{j}"] = f"input_blocks.{3 * i + j + 1}.1"
        if i < n - 1:
            m[f"down_blocks.{i}.downsamplers.0.conv"] = f"input_blocks.{3 * (i + 1)}.0.op"
    rev_attn = list(reversed(cfg.down_has_attn))
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            m[f"up_blocks.{i}.resnets.{j}"] = f"output_blocks.{3 * i + j}.0"
            if rev_attn[i]:
                m[f"up_blocks.{i}.attentions.{j}"] = f"output_blocks.{3 * i + j}.1"
        if i < n - 1:
            m[f"up_blocks.{i}.upsamplers.0.conv"] = f"output_blocks.{3 * i + 2}.{2 if rev_attn[i] else 1}.conv"
    return m


RES_SUB = {"norm1": "in_layers.0", "conv1": "in_layers.2", "time_emb_proj": "emb_layers.1", "norm2": "out_layers.0", "conv2": "out_layers.3",
           "conv_shortcut": "skip_connection"}


def to_ldm_state(cfg, sd):
    km = ldm_key_map(cfg)
    out = {}
    for k, v in sd.items():
        pref = max((p for p in km if k.startswith(p + ".")), key=len)
        rest = k[len(pref) + 1:]
        tgt = km[pref]
        if ".resnets." in pref:
            sub, leaf = rest.split(".", 1)
            rest = RES_SUB[sub] + "." + leaf
            if sub == "conv_shortcut":
                pass
        out[f"{tgt}.{rest}"] = v
    return out


def run_g6():
    from types import SimpleNamespace
    import types
    om = types.ModuleType("omegaconf")
    oml = types.ModuleType("omegaconf.listconfig")
    oml.ListConfig = type("ListConfig", (list,), {})
    om.listconfig = oml
    sys.modules["omegaconf"], sys.modules["omegaconf.listconfig"] = om, oml
    sys.path.insert(0, os.path.join(RH.REF, "evaluation", "MotionGuidance"))
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    from oracle import sd_unet
    cfg = sd_unet.unet_config("tiny-conv")
    cfg.norm_num_groups = 32
    cfg.heads = (4, 4, 4, 4)
    net = sd_unet.init_unet(cfg, seed=3)
    ldm = UNetModel(image_size=32, in_channels=4, model_channels=32, out_channels=4, num_res_blocks=2, attention_resolutions=[4, 2, 1],
                    channel_mult=[1, 2, 4, 4], num_heads=4, use_spatial_transformer=True, transformer_depth=1, context_dim=cfg.cross_attention_dim,
                    use_checkpoint=False, legacy=False).eval()
    sd = to_ldm_state(cfg, net.state_dict())
    missing, unexpected = ldm.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing[:5], unexpected[:5])
    x = rng_tensor(31, (2, 4, 16, 16))
    ctx = rng_tensor(32, (2, 77, cfg.cross_attention_dim))
    t = torch.tensor([481, 481])
    with torch.no_grad():
        y_ldm = ldm(x, t, context=ctx)
        y = net(x, torch.tensor(481), ctx)
    dev = (y - y_ldm).abs().max().item()
    np.savez_compressed(os.path.join(GOLD, "g6_ldm_unet.npz"), y=y_ldm.numpy(), t=np.array([481]))
    print(f"[G6] oracle UNet vs in-tree ldm UNetModel: max abs diff {dev:.3e} (|y|max {y_ldm.abs().max():.3f})")


def _bare_package(name, path):
    """register `name` as a package WITHOUT running its __init__ (sgm/__init__.py pulls pytorch_lightning / open_clip for model
    classes this script never touches); submodules then import from `path` as usual"""
    import types
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m


def run_g6b():
    """SD-2.1's two structural deltas from SD-1.x, pinned against the in-tree Stability `sgm` UNetModel
    (generative-models/sgm/modules/diffusionmodules/openaimodel.py:504 ff.): per-level head COUNTS from a constant head width
    (num_head_channels) and a LINEAR proj_in / proj_out in the transformer (use_linear_in_transformer)."""
    R = os.path.join(RH.REF, "generative-models", "sgm")
    _bare_package("sgm", R)
    _bare_package("sgm.modules", os.path.join(R, "modules"))
    _bare_package("sgm.modules.diffusionmodules", os.path.join(R, "modules", "diffusionmodules"))
    from sgm.modules.diffusionmodules.openaimodel import UNetModel
    from oracle import sd_unet
    cfg = sd_unet.unet_config("tiny")             # use_linear_projection=True
    cfg.norm_num_groups = 32
    cfg.heads = (2, 4, 8, 8)                      # 32, 64, 128, 128 channels / 16 per head
    assert cfg.use_linear_projection
    net = sd_unet.init_unet(cfg, seed=5)
    ref = UNetModel(in_channels=4, model_channels=32, out_channels=4, num_res_blocks=2, attention_resolutions=[4, 2, 1],
                    channel_mult=[1, 2, 4, 4], num_head_channels=16, transformer_depth=1, context_dim=cfg.cross_attention_dim,
                    use_checkpoint=False, use_linear_in_transformer=True, spatial_transformer_attn_type="softmax").eval()
    sd = to_ldm_state(cfg, net.state_dict())
    missing, unexpected = ref.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing[:5], unexpected[:5])
    x = rng_tensor(41, (2, 4, 16, 16))
    ctx = rng_tensor(42, (2, 77, cfg.cross_attention_dim))
    with torch.no_grad():
        y_ref = ref(x, torch.tensor([301, 301]), context=ctx)
        y = net(x, torch.tensor(301), ctx)
    dev = (y - y_ref).abs().max().item()
    np.savez_compressed(os.path.join(GOLD, "g6b_sgm_unet.npz"), y=y_ref.numpy(), t=np.array([301]), heads=np.array(cfg.heads))
    print(f"[G6b] oracle UNet (linear proj_in, per-level heads {cfg.heads}) vs in-tree sgm UNetModel: max abs diff {dev:.3e} "
          f"(|y|max {y_ref.abs().max():.3f})")


def vae_ldm_state(cfg, sd):
    """oracle (diffusers-layout) AutoencoderKL names -> ldm Encoder / Decoder names (two dicts); quant convs stay outside (they live in
    ldm's AutoencoderKL wrapper, which needs pytorch_lightning)."""
    n = len(cfg.block_out_channels)
    enc, dec = {}, {}
    attn = {"group_norm": "norm", "to_q": "q", "to_k": "k", "to_v": "v", "to_out.0": "proj_out"}
    for k, v in sd.items():
        side, rest = k.split(".", 1)
        if side not in ("encoder", "decoder"):
            continue
        dst = enc if side == "encoder" else dec
        parts = rest.split(".")
        if parts[0] in ("down_blocks", "up_blocks"):
            lvl = int(parts[1]) if side == "encoder" else n - 1 - int(parts[1])
            top = "down" if side == "encoder" else "up"
            if parts[2] == "resnets":
                leaf = ".".join(parts[4:]).replace("conv_shortcut", "nin_shortcut")
                name = f"{top}.{lvl}.block.{parts[3]}.{leaf}"
            else:
                name = f"{top}.{lvl}.{'downsample' if side == 'encoder' else 'upsample'}.conv.{parts[-1]}"
        elif parts[0] == "mid_block":
            if parts[1] == "resnets":
                name = f"mid.block_{int(parts[2]) + 1}." + ".".join(parts[3:]).replace("conv_shortcut", "nin_shortcut")
            else:
                sub = ".".join(parts[3:-1])
                name = f"mid.attn_1.{attn[sub]}.{parts[-1]}"
                if sub != "group_norm" and parts[-1] == "weight":
                    v = v[:, :, None, None]            # ldm's AttnBlock uses 1x1 convolutions
        elif parts[0] == "conv_norm_out":
            name = "norm_out." + parts[1]
        else:
            name = rest
        dst[name] = v
    return enc, dec


def run_g7():
    """the VAE restatement (oracle/sd_vae.py, diffusers-0.18 AutoencoderKL layout) against the in-tree CompVis Encoder / Decoder
    (evaluation/MotionGuidance/ldm/modules/diffusionmodules/model.py:368, 462) on key-mapped seeded weights."""
    from types import SimpleNamespace
    import types
    if "omegaconf" not in sys.modules:
        om, oml = types.ModuleType("omegaconf"), types.ModuleType("omegaconf.listconfig")
        oml.ListConfig = type("ListConfig", (list,), {})
        om.listconfig = oml
        sys.modules["omegaconf"], sys.modules["omegaconf.listconfig"] = om, oml
    mg = os.path.join(RH.REF, "evaluation", "MotionGuidance")
    if mg not in sys.path:
        sys.path.insert(0, mg)
    from ldm.modules.diffusionmodules.model import Decoder, Encoder
    from oracle import sd_vae
    cfg = SimpleNamespace(name="g7", in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(32, 64, 128, 128),
                          layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)
    vae = sd_vae.init_vae(cfg, seed=9)
    kw = dict(ch=32, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, attn_resolutions=[], dropout=0.0, in_channels=3, resolution=64, z_channels=4)
    enc, dec = Encoder(double_z=True, **kw).eval(), Decoder(**kw).eval()
    se, sdv = vae_ldm_state(cfg, vae.state_dict())
    for mod, st in ((enc, se), (dec, sdv)):
        missing, unexpected = mod.load_state_dict(st, strict=False)
        assert not missing and not unexpected, (missing[:5], unexpected[:5])
    x = rng_tensor(51, (2, 3, 64, 48))
    z = rng_tensor(52, (2, 4, 8, 6))
    with torch.no_grad():
        m_ref, d_ref = enc(x), dec(z)
        m_or, d_or = vae.encoder(x), vae.decoder(z)
    de, dd = (m_or - m_ref).abs().max().item(), (d_or - d_ref).abs().max().item()
    np.savez_compressed(os.path.join(GOLD, "g7_ldm_vae.npz"), moments=m_ref.numpy(), dec=d_ref.numpy())
    print(f"[G7] oracle VAE vs in-tree ldm Encoder / Decoder: encoder max abs diff {de:.3e} (|y|max {m_ref.abs().max():.3f}), "
          f"decoder {dd:.3e} (|y|max {d_ref.abs().max():.3f})")


# --------------------------------------------------------------------------------------------------------------
# G8: the depth network of the 3D front end -- the reference's own DinoVisionTransformer + DPTHead / DPT_DINOv2.forward
# --------------------------------------------------------------------------------------------------------------
def run_g8():
    """oracle/dpt.py against depth_anything/dpt.py (DPTHead, the forward of DPT_DINOv2: dpt.py:155-167) on the in-tree DINOv2
    (torchhub/facebookresearch_dinov2_main/vision_transformer.py), seeded weights, two small encoders, a square input at the
    pos-embed's own grid (no interpolation) and non-square inputs (bicubic pos-embed interpolation, offset 0.1)."""
    hub = os.path.join(RH.REF, "torchhub", "facebookresearch_dinov2_main")
    for q in (hub, RH.REF):
        if q not in sys.path:
            sys.path.insert(0, q)
    import vision_transformer as vits
    from depth_anything.dpt import DPTHead
    import torch.nn.functional as F
    from oracle import dpt as OD
    out = {}
    for name, img_size, sizes in (("tiny", 70, ((70, 70), (56, 98))), ("mini", 518, ((42, 70),))):
        cfg = OD.dpt_config(name)
        cfg.img_size = img_size
        st = OD.dpt_synthetic_state(cfg, seed=3 + len(name))
        vit = vits.DinoVisionTransformer(img_size=img_size, patch_size=14, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
                                         mlp_ratio=cfg.mlp_ratio, init_values=1.0, ffn_layer="mlp", block_chunks=0, num_register_tokens=0,
                                         interpolate_antialias=False, interpolate_offset=0.1).eval()
        head = DPTHead(1, cfg.embed_dim, cfg.features, False, out_channels=list(cfg.out_channels), use_clstoken=False).eval()
        missing, unexpected = vit.load_state_dict({k[len("pretrained."):]: v for k, v in st.items() if k.startswith("pretrained.")}, strict=True)
        assert not missing and not unexpected
        missing, unexpected = head.load_state_dict({k[len("depth_head."):]: v for k, v in st.items() if k.startswith("depth_head.")}, strict=True)
        assert not missing and not unexpected
        for (H, W) in sizes:
            x = rng_tensor(80 + H + W, (2, 3, H, W))
            with torch.no_grad():                               # DPT_DINOv2.forward, dpt.py:155-167
                feats = vit.get_intermediate_layers(x, 4, return_class_token=True)
                d = head(feats, H // 14, W // 14)
                d = F.relu(F.interpolate(d, size=(H, W), mode="bilinear", align_corners=True)).squeeze(1)
                o_feats = OD.vit_features(cfg, st, x, 4)
                o_d = OD.depth_forward(cfg, st, x)
            df = max((a[0] - b[0]).abs().max().item() for a, b in zip(feats, o_feats))
            dd = (d - o_d).abs().max().item()
            key = f"{name}_{H}x{W}"
            out[key + "_feat3"] = feats[3][0].numpy()
            out[key + "_depth"] = d.numpy()
            print(f"[G8] {key}: oracle vs reference  ViT features max abs diff {df:.3e} (|y|max {feats[3][0].abs().max():.3f}), "
                  f"depth {dd:.3e} (|y|max {d.abs().max():.3f}, {float((d > 0).float().mean()):.2f} of the pixels positive)")
    np.savez_compressed(os.path.join(GOLD, "g8_dpt.npz"), **out)


# --------------------------------------------------------------------------------------------------------------
# G9: the reference's loops on the METRIC's schedules (N = 50; call sites: SURVEY 8a table): tiny topology, tests/golden_cases.n50_cases
# --------------------------------------------------------------------------------------------------------------
def run_g9(A, Mo):
    from tests.golden_cases import n50_cases, tiny_state
    ori_img, coarse, img2 = synth_images()
    ori, tgt, draw, cons_sup, cons_tgt = mask_inputs()
    out, report = {}, []

    def capture(p, attr, cap):
        orig = getattr(p, attr)

        def f(*a, **k):
            k["return_intermediates"] = True
            im, lst = orig(*a, **k)
            cap["traj"] = lst
            return im, None
        setattr(p, attr, f)

    for name, hook, unet_name, planted, kw in n50_cases():
        kw = dict(kw)
        p = make_ref_pipe(A, Mo, hook, unet_name)
        st = tiny_state(unet_name, 0, planted)
        p.unet.load_state_dict(st)
        op = oracle_pipe(unet_name)
        op.unet.load_state_dict(st)
        cap = {}
        if hook == "edit":
            text, gs, eta = kw.pop("guidance_text"), kw.pop("guidance_scale"), kw.pop("eta")
            capture(p, "forward_sampling", cap)
            img, _ = p.FreeFine_generation(ori_img, ori, coarse, tgt, text, gs, eta, verbose=True, return_ori=True, seed=42, **kw)
            o_img, _, o_traj = op.freefine_generation(ori_img, ori, coarse, tgt, text, gs, eta, seed=42, **kw)
            traj = cap["traj"]
        elif hook == "bggen":
            hole = p.dilate_mask(ori // 255, 30)
            capture(p, "forward_sampling_background_gen", cap)
            img = p.FreeFine_background_generation(ori_img, hole, "empty scene", 7.5, 1.0, verbose=True, seed=7, **kw)
            o_img, o_traj = op.freefine_background_generation(ori_img, hole, "empty scene", 7.5, 1.0, seed=7, **kw)
            traj = [t if t.ndim == 3 else t[0] for t in cap["traj"]]
            o_traj = [t if t.ndim == 3 else t[0] for t in o_traj]
        else:
            (o1, o2), (t1, t2) = compose_masks()
            Mo.seed_everything(11)
            lst = p.DDIM_inversion_func_compose(img=coarse, compose_imgs=[ori_img, img2], prompt="", num_step=50, start_step=15, verbose=True)
            capture(p, "forward_sampling_compose", cap)
            img, _ = p.Details_Preserving_regeneration_compose(coarse, lst, ["a cup", "a dog"], [o1, o2], [t1, t2], None, num_steps=50,
                                                               start_step=15, end_step=50, eta=1.0, guidance_scale=7.5, verbose=True,
                                                               dil_factor=9, end_scale=0.5, **kw)
            o_img, o_traj = op.freefine_compose([ori_img, img2], [o1, o2], [t1, t2], coarse, ["a cup", "a dog"], 7.5, 1.0, end_step=50, num_step=50,
                                                start_step=15, seed=11, dil_factor=9, end_scale=0.5, **kw)
            traj = [t if t.ndim == 3 else t[0] for t in cap["traj"]]
            o_traj = [t if t.ndim == 3 else t[0] for t in o_traj]
        dev = max((a.float() - b.float()).abs().max().item() for a, b in zip(traj, o_traj))
        amax = max(a.abs().max().item() for a in traj)
        report.append((name, len(traj) - 1, dev, amax, int(np.abs(img.astype(int) - o_img.astype(int)).max())))
        out[f"{name}_traj"] = torch.stack(traj).numpy()
        out[f"{name}_img"] = img[::4, ::4].copy()
    np.savez_compressed(os.path.join(GOLD, "g9_n50_loops.npz"), **out)
    for r in report:
        print(f"[G9] {r[0]:14s} {r[1]:2d} guided steps: oracle vs reference ABSOLUTE latent deviation {r[2]:.3e} (|latent| max {r[3]:.2f}), uint8 image max diff {r[4]}")


# --------------------------------------------------------------------------------------------------------------
# G11: the model-free arithmetic of the metric suite (evaluation/metrics): warp error, Frechet distance, polynomial MMD^2
# --------------------------------------------------------------------------------------------------------------
def run_g11():
    import tempfile
    import types
    from PIL import Image
    mroot = os.path.join(RH.REF, "evaluation", "metrics")
    if mroot not in sys.path:
        sys.path.insert(0, mroot)
    for name in ("torchvision", "torchvision.transforms", "pytorch_fid", "pytorch_fid.inception"):      # imported by fid_score.py, unused by the function taken from it
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["pytorch_fid.inception"].InceptionV3 = type("InceptionV3", (), {"BLOCK_INDEX_BY_DIM": {2048: 3}})
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    from FID.fid_score import calculate_frechet_distance
    from FID.mmd import compute_mmd, compute_polynomial_mmd
    import wrap_error as WE
    out = {}
    rng = np.random.default_rng(11)
    f1, f2 = rng.standard_normal((300, 24)), rng.standard_normal((260, 24)) * 1.3 + 0.2
    out["feat_a"], out["feat_b"] = f1, f2
    mu1, s1, mu2, s2 = f1.mean(0), np.cov(f1, rowvar=False), f2.mean(0), np.cov(f2, rowvar=False)
    out["frechet"] = np.array([calculate_frechet_distance(mu1, s1, mu2, s2), calculate_frechet_distance(mu1, s1, mu1, s1)])
    out["mmd2"] = np.array([compute_polynomial_mmd(f1[:200], f2[:200]), compute_polynomial_mmd(f1[:64], f1[64:128])])
    np.random.seed(5)
    out["kd"] = compute_mmd(f1, f2, n_subsets=7, subset_size=100)
    with tempfile.TemporaryDirectory() as td:
        data, k = {}, 0
        for d in range(2):
            inst = {}
            for e in range(2):
                paths = {}
                for nm, arr in (("coarse_input_path", rng.integers(0, 256, (40, 48, 3), dtype=np.uint8)), ("gen", rng.integers(0, 256, (40, 48, 3), dtype=np.uint8)),
                                ("tgt_mask_path", (rng.random((40, 48)) > 0.6).astype(np.uint8) * 255)):
                    fp = os.path.join(td, f"{k}_{nm}.png")
                    Image.fromarray(arr).save(fp)
                    paths[nm] = fp
                    out[f"we_{k}_{nm}"] = arr
                inst[str(e)] = paths
                k += 1
            data[str(d)] = {"instances": {"0": inst}}
        out["we"] = np.array([WE.calculate_we(data, "gen")])
    np.savez_compressed(os.path.join(GOLD, "g11_metrics.npz"), **out)
    print(f"[G11] frechet {out['frechet']}, mmd2 {out['mmd2']}, kd mean {out['kd'].mean():.6f}, warp error {out['we'][0]:.6f}")


if __name__ == "__main__":
    only = sys.argv[1:] or ["g1", "g3", "g4", "g5", "g6", "g6b", "g7", "g8", "g9", "g11"]
    torch.set_grad_enabled(False)
    A, Mo = RH.import_reference()
    if "g1" in only:
        run_g1(A)
    if "g3" in only:
        run_g3(A, Mo)
    if "g4" in only:
        run_g4(A, Mo)
    if "g5" in only:
        run_g5(A, Mo)
    if "g6" in only:
        run_g6()
    if "g6b" in only:
        run_g6b()
    if "g7" in only:
        run_g7()
    if "g8" in only:
        run_g8()
    if "g9" in only:
        run_g9(A, Mo)
    if "g11" in only:
        run_g11()
