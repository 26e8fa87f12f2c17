"""GPU parity of the whole hot path: FreeFinePipeline (HIP engine, fp32 parity mode) against
  (a) the golden trajectories the REFERENCE produced (tests/golden/g5_loops.npz), and
  (b) the CPU oracle run in-process on identical seeded weights / inputs,
for the edit, background-generation and composition loops.  Tolerance: latent L-inf <= 1e-3 (north-star tolerance);
measured deviations are ~1e-5.  A bf16 fast-mode run is bounded loosely and its deviation printed."""
import os
import sys

import numpy as np
import pytest
import torch

from golden_cases import BG_CASES, CMP_CASES, compose_masks, edit_cases, mask_inputs, synth_images

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
torch.set_grad_enabled(False)
TOL = 1e-3


def make_pipe(gpu, unet_name, hook, dtype=torch.float32, graph=False, seed=0, ustate=None, x3=False):
    from freefine_amd.attention import (Attention_Modulator, register_attention_control, register_attention_control_4bggen,
                                        register_attention_control_compose)
    from freefine_amd.config import UNetConfig, VAEConfig
    from freefine_amd.pipeline import FreeFinePipeline
    from freefine_amd.scheduler import DDIMScheduler
    from freefine_amd.text import ByteTokenizer, SyntheticTextEncoder
    from oracle import sd_unet, sd_vae
    ocfg = sd_unet.unet_config(unet_name)
    ust = ustate if ustate is not None else sd_unet.init_unet(ocfg, seed=seed).state_dict()
    vst = sd_vae.init_vae(sd_vae.vae_config("tiny"), seed=seed + 1).state_dict()
    model = FreeFinePipeline.from_state(UNetConfig.preset(unet_name), ust, VAEConfig.preset("tiny"), vst, ByteTokenizer(),
                                        SyntheticTextEncoder(ocfg.cross_attention_dim), None, dtype, gpu, x3=x3)
    model.scheduler = DDIMScheduler.from_config(model.scheduler.config)
    controller = Attention_Modulator(start_layer=10)
    model.controller = controller
    {"edit": register_attention_control, "bggen": register_attention_control_4bggen,
     "compose": register_attention_control_compose}[hook](model, controller)
    model.modify_unet_forward()
    model.unet.use_graph = graph
    return model


def traj_dev(traj, ref, absolute=True):
    """latent L-inf over a trajectory: ABSOLUTE (the north-star tolerance is stated on the latent values themselves); only the
    uint8-wrap golden, whose latents grow to 1e3 ... inf by construction of the reference's mask arithmetic, is compared relative
    to the latent scale."""
    worst = 0.0
    assert len(traj) == len(ref)
    for a, b in zip(traj, ref):
        a = a.detach().float().cpu()
        b = torch.from_numpy(np.asarray(b))
        a = a if a.ndim == b.ndim else a[0]
        fa, fb = torch.isfinite(a), torch.isfinite(b)
        assert torch.equal(fa, fb)
        d = (a - b)[fa].abs().max().item()
        worst = max(worst, d if absolute else d / max(1.0, b[fa].abs().max().item()))
    return worst


@pytest.mark.parametrize("graph", [False, True])
def test_edit_loops_vs_reference_golden(gpu, graph):
    g = np.load(os.path.join(GOLD, "g5_loops.npz"))
    ori_img, coarse, _ = synth_images()
    ori, tgt, *_ = mask_inputs()
    for name, unet_name, kw in edit_cases():
        if graph and name not in ("edit_tca_draw", "edit_mmsa_es"):
            continue
        kw = dict(kw)
        text, gs, eta = kw.pop("guidance_text"), kw.pop("guidance_scale"), kw.pop("eta")
        model = make_pipe(gpu, unet_name, "edit", graph=graph)
        model.dedup_rows = not graph          # cover both: exact row de-duplication of the CFG batch on (eager) / off (graph)
        img_e, img_r = model.FreeFine_generation(ori_img, ori, coarse, tgt, text, gs, eta, verbose=True, return_ori=True, seed=42,
                                                 return_intermediates=True, **kw)
        dev = traj_dev(model.last_intermediates, g[f"{name}_traj"], absolute=name != "edit_tca_wrap")
        print(f"{name} graph={graph}: latent L-inf vs reference golden {dev:.2e}")
        assert dev < TOL, name
        assert np.abs(img_e[::4, ::4].astype(int) - g[f"{name}_img"].astype(int)).max() <= 1, name
        assert np.abs(img_r[::4, ::4].astype(int) - g[f"{name}_ref_img"].astype(int)).max() <= 1, name


def _batch_cases():
    ori_img, coarse, img2 = synth_images()
    ori, tgt, draw, *_ = mask_inputs()
    from golden_cases import rect_mask
    ori_b, tgt_b, draw_b = rect_mask(128, 128, 20, 60, 70, 110, 255), rect_mask(128, 128, 30, 70, 50, 90, 255), rect_mask(128, 128, 24, 76, 44, 98, 1)
    return [dict(ori_img=ori_img, ori_mask=ori, coarse_input=coarse, target_mask=tgt, guidance_text="a cup", draw_mask=draw),
            dict(ori_img=img2, ori_mask=ori_b, coarse_input=ori_img, target_mask=tgt_b, guidance_text="a dog on grass", draw_mask=draw_b),
            dict(ori_img=coarse, ori_mask=tgt, coarse_input=img2, target_mask=ori_b, guidance_text="", draw_mask=draw_b)]


@pytest.mark.parametrize("graph", [False, True])
def test_image_batched_edits_match_single_edits(gpu, graph):
    """FreeFine_generation_batch: K edits in one image-major batch == the K single-image edits, and image 0
    (the reference's golden edit_tca_draw inputs) still matches the REFERENCE's golden trajectory.  Two batches: images that
    agree on the CFG row de-duplication (3 physical rows each) and a mix with an empty prompt (falls back to 4 rows each)."""
    g = np.load(os.path.join(GOLD, "g5_loops.npz"))
    name, unet_name, kw = edit_cases()[0]
    kw = {k: v for k, v in kw.items() if k in ("end_step", "num_step", "start_step", "method_type", "end_scale")}
    cases, seeds = _batch_cases(), [42, 7, 1234]
    single = []
    model = make_pipe(gpu, unet_name, "edit", graph=graph)
    for c, sd in zip(cases, seeds):
        img = model.FreeFine_generation(c["ori_img"], c["ori_mask"], c["coarse_input"], c["target_mask"], c["guidance_text"], 7.5, 1.0,
                                        draw_mask=c["draw_mask"], seed=sd, return_intermediates=True, **kw)
        single.append((img, [t.clone() for t in model.last_intermediates]))
    assert traj_dev(single[0][1], g[f"{name}_traj"]) < TOL
    for sel in ([0, 1], [0, 1, 2], [1]):
        for rep in range(2 if graph else 1):                       # second pass replays the captured batched graphs
            imgs = model.FreeFine_generation_batch([cases[i] for i in sel], 7.5, 1.0, seeds=[seeds[i] for i in sel],
                                                   return_intermediates=True, **kw)
            for j, i in enumerate(sel):
                dev = traj_dev(model.last_intermediates[j], [t.cpu().numpy() for t in single[i][1]])
                print(f"batch {sel} graph={graph} rep={rep} image {i}: latent L-inf vs single-image edit {dev:.2e}")
                assert dev < 1e-4, (sel, i)
                assert np.abs(imgs[j].astype(int) - single[i][0].astype(int)).max() <= 1
            if sel[0] == 0:
                assert traj_dev(model.last_intermediates[0], g[f"{name}_traj"]) < TOL
    # the single-image path is intact after batching (controller restored)
    c = cases[1]
    img = model.FreeFine_generation(c["ori_img"], c["ori_mask"], c["coarse_input"], c["target_mask"], c["guidance_text"], 7.5, 1.0,
                                    draw_mask=c["draw_mask"], seed=seeds[1], return_intermediates=True, **kw)
    assert traj_dev(model.last_intermediates, [t.cpu().numpy() for t in single[1][1]]) < 1e-5


def test_bggen_and_compose_vs_reference_golden(gpu):
    g = np.load(os.path.join(GOLD, "g5_loops.npz"))
    ori_img, coarse, img2 = synth_images()
    ori, *_ = mask_inputs()
    for name, kw in BG_CASES:
        model = make_pipe(gpu, "tiny", "bggen")
        hole = model.dilate_mask(ori // 255, 30)
        img = model.FreeFine_background_generation(ori_img, hole, "empty scene", 3.5, 1.0, verbose=True, seed=7,
                                                   return_intermediates=True, **kw)
        dev = traj_dev(model.last_intermediates, g[f"{name}_traj"])
        print(f"{name}: latent L-inf vs reference golden {dev:.2e}")
        assert dev < TOL, name
        assert np.abs(img[::4, ::4].astype(int) - g[f"{name}_img"].astype(int)).max() <= 1, name
    oris, tgts = compose_masks()
    for name, kw in CMP_CASES:
        model = make_pipe(gpu, "tiny", "compose")
        img = model.FreeFine_cross_image_composition([ori_img, img2], oris, tgts, coarse, ["a cup", "a dog"], 7.5, 1.0, end_step=8,
                                                     num_step=10, start_step=6, verbose=True, seed=11, dil_factor=9, end_scale=0.5,
                                                     return_intermediates=True, **kw)
        dev = traj_dev(model.last_intermediates, g[f"{name}_traj"])
        print(f"{name}: latent L-inf vs reference golden {dev:.2e}")
        assert dev < TOL, name
        assert np.abs(img[::4, ::4].astype(int) - g[f"{name}_img"].astype(int)).max() <= 1, name


def test_vae_bracket_vs_oracle(gpu):
    from freefine_amd.config import VAEConfig
    from freefine_amd.vae import HipVAE
    from oracle import sd_vae
    ov = sd_vae.init_vae(sd_vae.vae_config("tiny"), seed=1)
    for dtype, x3, tol in ((torch.float32, False, 1e-4), (torch.float32, True, 1e-4), (torch.bfloat16, False, 5e-2)):
        hv = HipVAE(VAEConfig.preset("tiny"), ov.state_dict(), dtype=dtype, device=gpu, x3=x3)
        img = np.random.default_rng(5).integers(0, 256, (2, 64, 96, 3), dtype=np.uint8)
        x = (torch.from_numpy(img).float() / 127.5 - 1).permute(0, 3, 1, 2)
        ref = ov.encode_mean(x) * 0.18215
        out = hv.encode_mean_scaled(img_u8=torch.from_numpy(img)).cpu()
        assert (out - ref).abs().max() / ref.abs().max() < tol
        out2 = hv.encode_mean_scaled(x_nchw=x).cpu()
        assert (out2 - ref).abs().max() / ref.abs().max() < tol
        dref = (ov.decode(ref / 0.18215) / 2 + 0.5).clamp(0, 1)
        dout = hv.decode_image(ref.to(gpu)).cpu()
        assert dout.shape == dref.shape
        assert (dout - dref).abs().max() < tol * 2


def test_bf16_fast_mode_deviation_reported(gpu):
    """fast mode (bf16 MFMA operands / bf16 activations) is NOT expected to meet 1e-3; bound it and print it."""
    g = np.load(os.path.join(GOLD, "g5_loops.npz"))
    ori_img, coarse, _ = synth_images()
    ori, tgt, *_ = mask_inputs()
    name, unet_name, kw = edit_cases()[0]
    kw = dict(kw)
    text, gs, eta = kw.pop("guidance_text"), kw.pop("guidance_scale"), kw.pop("eta")
    model = make_pipe(gpu, unet_name, "edit", dtype=torch.bfloat16, graph=True)
    model.FreeFine_generation(ori_img, ori, coarse, tgt, text, gs, eta, verbose=True, seed=42, return_intermediates=True, **kw)
    dev = traj_dev(model.last_intermediates, g[f"{name}_traj"])
    print(f"bf16 fast mode: latent L-inf vs reference golden {dev:.3e}")
    assert dev < 0.5


@pytest.mark.parametrize("x3", [False, True], ids=["f32", "bf16x3"])
def test_geobench_harness_on_gpu(gpu, tmp_path, x3):
    """the GeoBench-2D harness (freefine_amd/geobench.py: case list, host pre-processing without cv2, batches of cases through
    FreeFine_generation_batch, PNG + JSON results) end to end on a synthetic GeoBenchMeta tree; one case re-run directly.  In the f32 parity
    mode and in the split-bf16 mode the drivers default to (evaluation/FreeFine/freefine_batch_infer_2d.py --dtype bf16x3 = bench.py's mode)."""
    from PIL import Image
    from freefine_amd import geobench
    root = str(tmp_path / "geo")
    geobench.make_synthetic_dataset(root, n_images=2, edits_per_image=2, size=128, seed=3, with_backgrounds=False)
    # stage 1: object removal writes the background images stage 2 pastes the moved object onto
    bg_model = make_pipe(gpu, "tiny", "bggen", graph=True, x3=x3)
    bgs = geobench.run_bggen(bg_model, root, blending=True, params=dict(num_step=10, start_step=1, end_step=6), dsize=(128, 128), seed=7,
                             verbose=False)
    assert len(bgs) == 2 and all(os.path.exists(b["inp_img_path"]) for b in bgs)
    assert geobench.run_bggen(bg_model, root, params=dict(num_step=10, start_step=1, end_step=6), dsize=(128, 128), seed=7, verbose=False) == []
    model = make_pipe(gpu, "tiny", "edit", graph=True, x3=x3)
    params = dict(num_step=10, start_step=7, end_step=10)
    res = geobench.run(model, root, batch=3, params=params, dsize=(128, 128), verbose=False)      # 4 cases: a batch of 3, then a single one
    assert len(res) == 4 and os.path.exists(os.path.join(root, "generated_results_freefine_2d.json"))
    case = res[1]
    inputs = geobench.load_case(case, root, (128, 128))
    direct = model.FreeFine_generation(inputs["ori_img"], inputs["ori_mask"], inputs["coarse_input"], inputs["target_mask"], "", 7.5, 1.0,
                                       end_step=10, num_step=10, start_step=7, seed=42, end_scale=0.0, draw_mask=inputs["draw_mask"],
                                       use_auto_draw=True, reduce_inp_artifacts=True, cons_area=inputs["cons_area"])
    saved = np.asarray(Image.open(case["gen_img_path"]))
    assert saved.shape == (128, 128, 3) and np.isfinite(direct.astype(float)).all()
    assert np.abs(saved.astype(int) - direct.astype(int)).max() <= 1
    # GeoBench-3D variant: coarse edits / target / draw masks read from disk, text = object label, start_step 15
    root3 = str(tmp_path / "geo3d")
    geobench.make_synthetic_dataset(root3, n_images=1, edits_per_image=3, size=128, seed=5, with_3d=True)
    res3 = geobench.run(model, root3, batch=2, params=dict(num_step=10, start_step=3, end_step=10), dsize=(128, 128), verbose=False,
                        variant="3d_depth")
    assert len(res3) == 3 and os.path.exists(os.path.join(root3, "generated_results_freefine_depth.json"))
    assert all(np.asarray(Image.open(r["gen_img_path"])).shape == (128, 128, 3) for r in res3)
    # GeoBench-3D from the RGB image + transform (BASELINE configs[4]'s front end): DepthAnything depth -> point-cloud warp -> guided edit
    from freefine_amd import depth as FDp
    dcfg = FDp.depth_config("tiny")
    dmodel = FDp.HipDepthAnything(dcfg, FDp.synthetic_state(dcfg, seed=3), dtype=torch.float32, device=gpu)
    root4 = str(tmp_path / "geo3d_rgb")
    geobench.make_synthetic_dataset(root4, n_images=2, edits_per_image=4, size=128, seed=6, with_3d=True, with_backgrounds=True)
    case = geobench.load_json(os.path.join(root4, "annotations.json"))["0000"]["instances"]["0"]["0"]
    inp = geobench.load_case_3d_rgb(dict(case, da_n="0000", ins_id="0", edit_ins="0"), root4, (128, 128), depth_model=dmodel)
    assert inp["coarse_input"].shape == (128, 128, 3) and inp["target_mask"].max() == 255 and 0 < (inp["target_mask"] > 0).mean() < 0.5
    # 8 cases in batches of 2 on a FRESH pipeline (no forward graph of this schedule captured yet) at start_step 4: the prefetch thread (host
    # only: file reads) runs ahead of the consumer while it captures graphs; depth network + warp run on the consumer thread
    model4 = make_pipe(gpu, "tiny", "edit", graph=True, x3=x3)
    res4 = geobench.run(model4, root4, batch=2, params=dict(num_step=10, start_step=4, end_step=10), dsize=(128, 128), verbose=False,
                        variant="3d_rgb", depth_model=dmodel)
    assert len(res4) == 8 and os.path.exists(os.path.join(root4, "generated_results_freefine_depth_rgb.json"))
    assert all(np.isfinite(np.asarray(Image.open(r["gen_img_path"])).astype(float)).all() for r in res4)
    raw = geobench.read_case_3d_rgb(dict(case, da_n="0000", ins_id="0", edit_ins="0"), root4, (128, 128))
    assert raw["_raw_3d_rgb"] and all(isinstance(raw[k], np.ndarray) for k in ("ori_img", "ori_mask", "bg"))      # host half: arrays only
    again = geobench.finish_case_3d_rgb(raw, dmodel)
    assert np.array_equal(again["coarse_input"], inp["coarse_input"]) and np.array_equal(again["target_mask"], inp["target_mask"])


def test_image_batched_background_generation_matches_single(gpu):
    """FreeFine_background_generation_batch == the single-image calls (and image 0 of the golden bg_tca configuration still
    matches the REFERENCE's golden trajectory)."""
    g = np.load(os.path.join(GOLD, "g5_loops.npz"))
    ori_img, coarse, img2 = synth_images()
    ori, tgt, *_ = mask_inputs()
    name, kw = BG_CASES[0]
    model = make_pipe(gpu, "tiny", "bggen", graph=True)
    holes = [model.dilate_mask(ori // 255, 30), model.dilate_mask(tgt // 255, 12)]
    imgs_in, texts, seeds = [ori_img, img2], ["empty scene", "a quiet street"], [7, 99]
    single = []
    for im, hole, txt, sd in zip(imgs_in, holes, texts, seeds):
        out = model.FreeFine_background_generation(im, hole, txt, 3.5, 1.0, verbose=True, seed=sd, return_intermediates=True, **kw)
        single.append((out, [t.clone() for t in model.last_intermediates]))
    assert traj_dev(single[0][1], g[f"{name}_traj"]) < TOL
    cases = [dict(ori_img=im, ori_mask=hole, guidance_text=txt) for im, hole, txt in zip(imgs_in, holes, texts)]
    for rep in range(2):
        outs = model.FreeFine_background_generation_batch(cases, 3.5, 1.0, seeds=seeds, return_intermediates=True, **kw)
        for k in range(2):
            dev = traj_dev(model.last_intermediates[k], [t.cpu().numpy() for t in single[k][1]])
            print(f"bg batch rep={rep} image {k}: latent L-inf vs single-image call {dev:.2e}")
            assert dev < 1e-4
            assert np.abs(outs[k].astype(int) - single[k][0].astype(int)).max() <= 1
    out = model.FreeFine_background_generation(imgs_in[1], holes[1], texts[1], 3.5, 1.0, verbose=True, seed=seeds[1], return_intermediates=True, **kw)
    assert traj_dev(model.last_intermediates, [t.cpu().numpy() for t in single[1][1]]) < 1e-5


def test_full_size_edit_loop_vs_oracle(gpu):
    """BASELINE config 2's topology end to end at full size: SD-2.1-base UNet (865.9 M parameters), 512x512 images -> 64x64 latents,
    a shortened schedule (N = 10, start_step = 8: 2 inversion forwards of 2 rows + 2 guided forwards of 4 rows with TCA in blocks
    10-15, local cross-attention, masked CFG 7.5, eta = 1 with CPU-generator noise), fp32 parity mode against OraclePipeline on the
    same seeded weights, images, masks and seeds.  Gate: ABSOLUTE latent L-inf <= 1e-3 at every recorded step (north star).
    The VAE is the tiny topology (the SD VAE has its own full-size test): it only brackets the loop."""
    from golden_cases import rect_mask
    from freefine_amd.text import ByteTokenizer, SyntheticTextEncoder, make_text_embed
    from oracle import sd_unet, sd_vae
    from oracle.pipeline import OraclePipeline
    torch.set_num_threads(max(8, min(32, torch.get_num_threads())))
    H = 512
    ori_img, coarse, _ = synth_images(H, H)
    ori, tgt, draw = rect_mask(H, H, 200, 304, 96, 200, 255), rect_mask(H, H, 200, 304, 160, 264, 255), rect_mask(H, H, 184, 320, 144, 288, 1)
    kw = dict(end_step=10, num_step=10, start_step=8, method_type="tca", end_scale=0.5, draw_mask=draw)
    cfg = sd_unet.unet_config("sd21-base")
    ounet = sd_unet.init_unet(cfg, seed=0)
    opipe = OraclePipeline(ounet, sd_vae.init_vae(sd_vae.vae_config("tiny"), seed=1),
                           make_text_embed(ByteTokenizer(), SyntheticTextEncoder(cfg.cross_attention_dim)))
    o_img, _, o_traj = opipe.freefine_generation(ori_img, ori, coarse, tgt, "a photo of a cup", 7.5, 1.0, seed=42, **kw)
    for graph, x3 in ((False, False), (True, False), (True, True)):
        model = make_pipe(gpu, "sd21-base", "edit", graph=graph, ustate=ounet.state_dict(), x3=x3)
        img = model.FreeFine_generation(ori_img, ori, coarse, tgt, "a photo of a cup", 7.5, 1.0, verbose=True, seed=42,
                                        return_intermediates=True, **kw)
        dev = traj_dev(model.last_intermediates, [t.numpy() for t in o_traj])
        print(f"full-size 2+2-step edit, {'split-bf16' if x3 else 'fp32'}, graph={graph}: absolute latent L-inf vs oracle {dev:.2e}; image max |diff| "
              f"{np.abs(img.astype(int) - o_img.astype(int)).max()}")
        assert dev < TOL
        assert np.abs(img.astype(int) - o_img.astype(int)).max() <= 1
        del model
        torch.cuda.empty_cache()


def test_split_bf16_mode_meets_parity_on_whole_loops(gpu):
    """the split-bf16 mode (FFN_BF16X3 GEMMs, fp32 activations) through WHOLE loops against the REFERENCE's golden trajectories:
    the same 1e-3 absolute latent gate as the fp32 parity mode (edit tca / mmsa, background generation, composition)."""
    g = np.load(os.path.join(GOLD, "g5_loops.npz"))
    ori_img, coarse, img2 = synth_images()
    ori, tgt, *_ = mask_inputs()
    for name, unet_name, kw in edit_cases():
        if name not in ("edit_tca_draw", "edit_tca_auto", "edit_mmsa"):
            continue
        kw = dict(kw)
        text, gs, eta = kw.pop("guidance_text"), kw.pop("guidance_scale"), kw.pop("eta")
        model = make_pipe(gpu, unet_name, "edit", graph=True, x3=True)
        img = model.FreeFine_generation(ori_img, ori, coarse, tgt, text, gs, eta, verbose=True, seed=42, return_intermediates=True, **kw)
        dev = traj_dev(model.last_intermediates, g[f"{name}_traj"])
        print(f"split-bf16 {name}: latent L-inf vs reference golden {dev:.2e}")
        assert dev < TOL, name
        assert np.abs(img[::4, ::4].astype(int) - g[f"{name}_img"].astype(int)).max() <= 1, name
    name, kw = BG_CASES[0]
    model = make_pipe(gpu, "tiny", "bggen", x3=True)
    model.FreeFine_background_generation(ori_img, model.dilate_mask(ori // 255, 30), "empty scene", 3.5, 1.0, verbose=True, seed=7,
                                         return_intermediates=True, **kw)
    dev = traj_dev(model.last_intermediates, g[f"{name}_traj"])
    print(f"split-bf16 {name}: latent L-inf vs reference golden {dev:.2e}")
    assert dev < TOL
    oris, tgts = compose_masks()
    name, kw = CMP_CASES[0]
    model = make_pipe(gpu, "tiny", "compose", x3=True)
    model.FreeFine_cross_image_composition([ori_img, img2], oris, tgts, coarse, ["a cup", "a dog"], 7.5, 1.0, end_step=8, num_step=10,
                                           start_step=6, verbose=True, seed=11, dil_factor=9, end_scale=0.5, return_intermediates=True, **kw)
    dev = traj_dev(model.last_intermediates, g[f"{name}_traj"])
    print(f"split-bf16 {name}: latent L-inf vs reference golden {dev:.2e}")
    assert dev < TOL


def test_vae_full_size_vs_oracle(gpu):
    """the SD VAE topology (128-256-512-512, 83.7 M parameters) at 512x512: encode (latent mean x 0.18215) and decode (-> [0,1] image)
    on HIP against the oracle VAE, itself pinned to the in-tree CompVis Encoder / Decoder by tests/golden/g7_ldm_vae.npz.
    fp32 parity mode <= 2e-4 of the output scale; the bf16 fast mode's deviation is printed and bounded."""
    from freefine_amd.config import VAEConfig
    from freefine_amd.vae import HipVAE
    from oracle import sd_vae
    torch.set_num_threads(max(8, min(32, torch.get_num_threads())))
    ov = sd_vae.init_vae(sd_vae.vae_config("sd"), seed=1)
    img = np.random.default_rng(5).integers(0, 256, (1, 512, 512, 3), dtype=np.uint8)
    x = (torch.from_numpy(img).float() / 127.5 - 1).permute(0, 3, 1, 2)
    ref = ov.encode_mean(x) * 0.18215
    dref = (ov.decode(ref / 0.18215) / 2 + 0.5).clamp(0, 1)
    for dtype, x3, tol in ((torch.float32, False, 2e-4), (torch.float32, True, 2e-4), (torch.bfloat16, False, 5e-2)):
        hv = HipVAE(VAEConfig.preset("sd"), ov.state_dict(), dtype=dtype, device=gpu, x3=x3)
        out = hv.encode_mean_scaled(img_u8=torch.from_numpy(img)).cpu()
        e_enc = ((out - ref).abs().max() / ref.abs().max()).item()
        dout = hv.decode_image(ref.to(gpu)).cpu()
        e_dec = (dout - dref).abs().max().item()
        print(f"SD VAE @512^2 {dtype}{' split-bf16' if x3 else ''}: encode max |diff| / max |ref| = {e_enc:.2e}, decode max |diff| (image in [0,1]) = {e_dec:.2e}")
        assert out.shape == ref.shape == (1, 4, 64, 64) and dout.shape == dref.shape == (1, 3, 512, 512)
        assert e_enc < tol and e_dec < 2 * tol
        del hv
        torch.cuda.empty_cache()


@pytest.mark.parametrize("graph", [False, True])
def test_reference_stream_reuse_is_exact(gpu, graph):
    """Reference-stream reuse (the guided loop's reference row re-enters the UNet at the first TCA block from the state the inversion
    recorded for the same latent / timestep / prompt) against the plain recomputation and against the REFERENCE's golden trajectory:
    single-image edits (3 physical rows with a prompt, 2 with an empty one, 4 without row de-duplication) and the image-batched path.
    fp32: the two ways agree to summation-order level; the forwards counted in `reuse_replays` prove the reuse path ran."""
    g = np.load(os.path.join(GOLD, "g5_loops.npz"))
    ori_img, coarse, _ = synth_images()
    ori, tgt, *_ = mask_inputs()
    for name, unet_name, kw in edit_cases():
        if name not in ("edit_tca_draw", "edit_tca_auto", "edit_mmsa_es"):
            continue
        kw = dict(kw)
        text, gs, eta = kw.pop("guidance_text"), kw.pop("guidance_scale"), kw.pop("eta")
        n_guided = kw["num_step"] - kw["start_step"]
        trajs = {}
        for reuse, dedup in ((True, True), (False, True), (True, False)):
            model = make_pipe(gpu, unet_name, "edit", graph=graph)
            model.reuse_ref_stream, model.dedup_rows = reuse, dedup
            model.FreeFine_generation(ori_img, ori, coarse, tgt, text, gs, eta, verbose=True, seed=42, return_intermediates=True, **kw)
            assert model.unet.reuse_replays == (n_guided if reuse else 0), (name, reuse, dedup, model.unet.reuse_replays)
            trajs[(reuse, dedup)] = [t.clone() for t in model.last_intermediates]
            dev = traj_dev(model.last_intermediates, g[f"{name}_traj"])
            print(f"{name} graph={graph} reuse={reuse} dedup={dedup}: latent L-inf vs reference golden {dev:.2e}")
            assert dev < TOL
        assert traj_dev(trajs[(True, True)], [t.cpu().numpy() for t in trajs[(False, True)]]) < 1e-4
    # image-batched edits: reuse on vs off
    name, unet_name, kw = edit_cases()[0]
    kw = {k: v for k, v in kw.items() if k in ("end_step", "num_step", "start_step", "method_type", "end_scale")}
    cases, seeds = _batch_cases(), [42, 7, 1234]
    outs = {}
    for reuse in (True, False):
        model = make_pipe(gpu, unet_name, "edit", graph=graph)
        model.reuse_ref_stream = reuse
        for sel in ([0, 1], [0, 1, 2]):
            model.FreeFine_generation_batch([cases[i] for i in sel], 7.5, 1.0, seeds=[seeds[i] for i in sel], return_intermediates=True, **kw)
            outs[(reuse, tuple(sel))] = [[t.clone() for t in tr] for tr in model.last_intermediates]
        assert (model.unet.reuse_replays > 0) == reuse
    for sel in ((0, 1), (0, 1, 2)):
        for j in range(len(sel)):
            assert traj_dev(outs[(True, sel)][j], [t.cpu().numpy() for t in outs[(False, sel)][j]]) < 1e-4, (sel, j)
    assert traj_dev(outs[(True, (0, 1))][0], g[f"{name}_traj"]) < TOL


def test_edit_at_768_px_vs_oracle(gpu):
    """BASELINE.json configs[4]'s resolution (768 x 768 images, 96 x 96 latents, S = 9216 / 2304 / 576 / 144 in the attention layers --
    the reference itself hard-codes 512, model.py:1347): one FreeFine_generation edit on the tiny topology, 1 + 1 steps with TCA, fp32
    parity mode and the split-bf16 mode against OraclePipeline; bf16 fast mode bounded.  Also: an original image LARGER than the coarse
    input is thumbnailed to the coarse input's size, not to 512."""
    from golden_cases import rect_mask
    torch.set_num_threads(max(8, min(32, torch.get_num_threads())))
    H, k = 768, 6
    rng = np.random.default_rng(0)
    ori_img, coarse = rng.integers(0, 256, (H, H, 3), dtype=np.uint8), rng.integers(0, 256, (H, H, 3), dtype=np.uint8)
    ori, tgt, draw = rect_mask(H, H, 50 * k, 76 * k, 24 * k, 50 * k, 255), rect_mask(H, H, 50 * k, 76 * k, 40 * k, 66 * k, 255), rect_mask(H, H, 46 * k, 80 * k, 36 * k, 72 * k, 1)
    kw = dict(end_step=10, num_step=10, start_step=9, method_type="tca", end_scale=0.5, draw_mask=draw)
    o_img, _, o_traj = oracle_pipe("tiny").freefine_generation(ori_img, ori, coarse, tgt, "a cup", 7.5, 1.0, seed=42, **kw)
    for dtype, x3, tol in ((torch.float32, False, TOL), (torch.float32, True, TOL), (torch.bfloat16, False, 0.5)):
        model = make_pipe(gpu, "tiny", "edit", dtype=dtype, graph=True, x3=x3)
        img = model.FreeFine_generation(ori_img, ori, coarse, tgt, "a cup", 7.5, 1.0, verbose=True, seed=42, return_intermediates=True, **kw)
        dev = traj_dev(model.last_intermediates, [t.numpy() for t in o_traj])
        print(f"768 x 768 edit, {dtype}{' split-bf16' if x3 else ''}: latent L-inf vs oracle {dev:.2e}")
        assert img.shape == (H, H, 3) and dev < tol
        if dtype == torch.float32:
            assert np.abs(img.astype(int) - o_img.astype(int)).max() <= 1
    assert model._work_size(coarse) == [768, 768] and model.resize_img(np.zeros((1024, 1024, 3), np.uint8), size=model._work_size(coarse)).shape == (768, 768, 3)
    assert model._work_size(np.zeros((256, 256, 3), np.uint8)) == [512, 512]


def test_full_size_edit_at_768_px_modes_agree(gpu):
    """BASELINE configs[4] at ITS size: the SD-2.1-base UNet (865.9 M parameters) on a 768 x 768 edit (96 x 96 latents; TCA in blocks 10-15 at
    S = 9216 / 2304, local cross-attention, masked CFG, eta = 1), N = 10 / start_step 7 (3 + 3 forwards), through the hipGraphs.  The CPU oracle
    cannot run this (its modulated attention materialises [4 h, S, S] scores: 20 GB of host memory per call), so the gate is CROSS-MODE: the
    split-bf16 mode against the fp32 mode -- itself pinned to the oracle per forward at this size (test_unet_full_size_768px_latent_96x96), per
    kernel at S = 9216 (test_attention_tca_production_shapes) and end to end at 64 x 64 (G10) -- at <= 1e-3 ABSOLUTE latent L-inf."""
    from golden_cases import rect_mask
    from oracle import sd_unet
    H, k = 768, 6
    rng = np.random.default_rng(0)
    ori_img, coarse = rng.integers(0, 256, (H, H, 3), dtype=np.uint8), rng.integers(0, 256, (H, H, 3), dtype=np.uint8)
    ori, tgt, draw = rect_mask(H, H, 50 * k, 76 * k, 24 * k, 50 * k, 255), rect_mask(H, H, 50 * k, 76 * k, 40 * k, 66 * k, 255), rect_mask(H, H, 46 * k, 80 * k, 36 * k, 72 * k, 1)
    kw = dict(end_step=10, num_step=10, start_step=7, method_type="tca", end_scale=0.0, draw_mask=draw)
    ust = sd_unet.init_unet(sd_unet.unet_config("sd21-base"), seed=0).state_dict()
    trajs = {}
    for name, dtype, x3 in (("fp32", torch.float32, False), ("split-bf16", torch.float32, True), ("bf16", torch.bfloat16, False)):
        model = make_pipe(gpu, "sd21-base", "edit", dtype=dtype, graph=True, ustate=ust, x3=x3)
        img = model.FreeFine_generation(ori_img, ori, coarse, tgt, "a photo of a cup", 7.5, 1.0, verbose=True, seed=42, return_intermediates=True, **kw)
        assert img.shape == (H, H, 3) and model.last_intermediates[0].shape[-2:] == (96, 96)
        trajs[name] = torch.stack([t.detach().float().cpu() for t in model.last_intermediates])
        del model
        torch.cuda.empty_cache()
    assert torch.isfinite(trajs["fp32"]).all()
    d3 = (trajs["split-bf16"] - trajs["fp32"]).abs().max().item()
    d16 = (trajs["bf16"] - trajs["fp32"]).abs().max().item()
    print(f"full-size 768 x 768 edit (3 + 3 forwards, |latent| max {trajs['fp32'].abs().max():.2f}): split-bf16 vs fp32 {d3:.2e}, bf16 vs fp32 {d16:.2e}")
    assert d3 < TOL


from golden_cases import oracle_pipe  # noqa: E402


@pytest.mark.parametrize("graph", [False, True])
def test_image_batched_composition_matches_single(gpu, graph):
    """FreeFine_cross_image_composition_batch (BASELINE configs[3], R = 2 references, P = 3 prompts incl. the trailing ""): K compositions
    in one image-major batch == the K single calls, and image 0 (the golden cmp_tca inputs) still matches the REFERENCE's golden
    trajectory.  The text batch has R + 1 + P rows per image against R + 2 latent rows: the cross-attention K / V rows of image i are
    offset by its TEXT block."""
    g = np.load(os.path.join(GOLD, "g5_loops.npz"))
    ori_img, coarse, img2 = synth_images()
    oris, tgts = compose_masks()
    name, kw = CMP_CASES[0]
    common = dict(end_step=8, num_step=10, start_step=6, dil_factor=9, end_scale=0.5, **kw)
    cases = [dict(img_lists=[ori_img, img2], ori_mask_lists=oris, tgt_mask_lists=tgts, coarse_input=coarse, guidance_text_list=["a cup", "a dog"]),
             dict(img_lists=[img2, coarse], ori_mask_lists=oris[::-1], tgt_mask_lists=tgts[::-1], coarse_input=ori_img, guidance_text_list=["a tree", "grass"]),
             dict(img_lists=[coarse, ori_img], ori_mask_lists=oris, tgt_mask_lists=tgts[::-1], coarse_input=img2, guidance_text_list=["", "a red car"])]
    seeds = [11, 5, 77]
    model = make_pipe(gpu, "tiny", "compose", graph=graph)
    single = []
    for c, sd in zip(cases, seeds):
        img = model.FreeFine_cross_image_composition(c["img_lists"], c["ori_mask_lists"], c["tgt_mask_lists"], c["coarse_input"], list(c["guidance_text_list"]),
                                                     7.5, 1.0, verbose=True, seed=sd, return_intermediates=True, **common)
        single.append((img, [t.clone() for t in model.last_intermediates]))
    assert traj_dev(single[0][1], g[f"{name}_traj"]) < TOL
    for sel in ([0, 1, 2], [0, 1]):
        for rep in range(2 if graph else 1):
            imgs = model.FreeFine_cross_image_composition_batch([cases[i] for i in sel], 7.5, 1.0, seeds=[seeds[i] for i in sel], return_intermediates=True,
                                                                **common)
            for j, i in enumerate(sel):
                dev = traj_dev(model.last_intermediates[j], [t.cpu().numpy() if t.ndim == 3 else t[0].cpu().numpy() for t in single[i][1]])
                print(f"compose batch {sel} graph={graph} rep={rep} image {i}: latent L-inf vs single call {dev:.2e}")
                assert dev < 1e-4, (sel, i)
                assert np.abs(imgs[j].astype(int) - single[i][0].astype(int)).max() <= 1
    assert traj_dev(model.last_intermediates[0], g[f"{name}_traj"]) < TOL


@pytest.mark.parametrize("graph", [False, True])
def test_composition_stored_reference_kv_is_exact(gpu, graph):
    """Composition hook (BASELINE configs[3] "shared K/V cache"): the R reference rows of the guided loop are the very (latent, timestep, "")
    rows the inversion evaluated and the hook never modulates them, so the loop runs the two edit rows only and reads the references' K / V
    of the modulated blocks from the inversion's record.  Against the plain recomputation (reuse off), against the REFERENCE's golden
    trajectory, single and image-batched, both methods; `reuse_replays` proves the stored-K/V path ran at every guided step."""
    g = np.load(os.path.join(GOLD, "g5_loops.npz"))
    ori_img, coarse, img2 = synth_images()
    oris, tgts = compose_masks()
    n_guided = 10 - 6
    as_np = lambda tr: [t.cpu().numpy() if t.ndim == 3 else t[0].cpu().numpy() for t in tr]
    for name, kw in CMP_CASES:
        trajs = {}
        for reuse in (True, False):
            model = make_pipe(gpu, "tiny", "compose", graph=graph)
            model.reuse_ref_stream = reuse
            model.FreeFine_cross_image_composition([ori_img, img2], oris, tgts, coarse, ["a cup", "a dog"], 7.5, 1.0, end_step=8, num_step=10,
                                                   start_step=6, verbose=True, seed=11, dil_factor=9, end_scale=0.5, return_intermediates=True, **kw)
            assert model.unet.reuse_replays == (n_guided if reuse else 0), (name, reuse, model.unet.reuse_replays)
            trajs[reuse] = [t.clone() for t in model.last_intermediates]
            dev = traj_dev(model.last_intermediates, g[f"{name}_traj"])
            print(f"{name} graph={graph} stored K/V={reuse}: latent L-inf vs reference golden {dev:.2e}")
            assert dev < TOL, (name, reuse)
        assert traj_dev(trajs[True], as_np(trajs[False])) < 1e-4, name
    name, kw = CMP_CASES[0]
    common = dict(end_step=8, num_step=10, start_step=6, dil_factor=9, end_scale=0.5, **kw)
    cases = [dict(img_lists=[ori_img, img2], ori_mask_lists=oris, tgt_mask_lists=tgts, coarse_input=coarse, guidance_text_list=["a cup", "a dog"]),
             dict(img_lists=[img2, coarse], ori_mask_lists=oris[::-1], tgt_mask_lists=tgts[::-1], coarse_input=ori_img, guidance_text_list=["a tree", "grass"])]
    outs = {}
    for reuse in (True, False):
        model = make_pipe(gpu, "tiny", "compose", graph=graph)
        model.reuse_ref_stream = reuse
        model.FreeFine_cross_image_composition_batch(cases, 7.5, 1.0, seeds=[11, 5], return_intermediates=True, **common)
        assert model.unet.reuse_replays == (n_guided if reuse else 0)
        outs[reuse] = [[t.clone() for t in tr] for tr in model.last_intermediates]
    for j in range(2):
        assert traj_dev(outs[True][j], as_np(outs[False][j])) < 1e-4, j
    assert traj_dev(outs[True][0], g[f"{name}_traj"]) < TOL


@pytest.mark.parametrize("x3", [False, True], ids=["f32", "bf16x3"])
def test_metric_schedules_n50_vs_reference_golden(gpu, x3):
    """The METRIC's own schedules (N = 50) against trajectories the REFERENCE produced (tests/golden/g9_n50_loops.npz, tools/gen_golden.py
    run_g9): edit at start_step 0 (50 + 50 forwards), 35 (GeoBench-2D, freefine_batch_infer_2d.py:212-230), 15 (GeoBench-3D,
    freefine_batch_infer_3d_depth.py:144-162), background generation at start_step 1 (freefine_batch_infer_bggen_2d.py:166-180) and the
    composition at start_step 15.  Gate: ABSOLUTE latent L-inf <= 1e-3 at EVERY step, in the fp32 parity mode and in the split-bf16 mode
    (the mode bench.py's headline is timed in), through the captured hipGraphs."""
    from golden_cases import n50_cases, tiny_state
    g = np.load(os.path.join(GOLD, "g9_n50_loops.npz"))
    ori_img, coarse, img2 = synth_images()
    ori, tgt, *_ = mask_inputs()
    for name, hook, unet_name, planted, kw in n50_cases():
        kw = dict(kw)
        model = make_pipe(gpu, unet_name, hook, graph=True, ustate=tiny_state(unet_name, 0, planted), x3=x3)
        if hook == "edit":
            text, gs, eta = kw.pop("guidance_text"), kw.pop("guidance_scale"), kw.pop("eta")
            img = model.FreeFine_generation(ori_img, ori, coarse, tgt, text, gs, eta, verbose=True, seed=42, return_intermediates=True, **kw)
        elif hook == "bggen":
            img = model.FreeFine_background_generation(ori_img, model.dilate_mask(ori // 255, 30), "empty scene", 7.5, 1.0, verbose=True, seed=7,
                                                       return_intermediates=True, **kw)
        else:
            oris, tgts = compose_masks()
            img = model.FreeFine_cross_image_composition([ori_img, img2], oris, tgts, coarse, ["a cup", "a dog"], 7.5, 1.0, end_step=50,
                                                         num_step=50, start_step=15, verbose=True, seed=11, dil_factor=9, end_scale=0.5,
                                                         return_intermediates=True, **kw)
        ref = g[f"{name}_traj"]
        dev = traj_dev(model.last_intermediates, ref)
        print(f"{'split-bf16' if x3 else 'fp32'} {name} ({len(ref) - 1} guided steps, |latent| max {np.abs(ref).max():.1f}): "
              f"ABSOLUTE latent L-inf vs reference golden {dev:.2e}")
        assert dev < TOL, name
        assert np.abs(img[::4, ::4].astype(int) - g[f"{name}_img"].astype(int)).max() <= 1, name
        if name in ("n50_edit_s0", "n50_edit_s35"):
            # the path bench.py times: the same edit as image 0 of an IMAGE-BATCHED call (FreeFine_generation_batch, other images beside it)
            cases = _batch_cases()
            c0 = dict(ori_img=ori_img, ori_mask=ori, coarse_input=coarse, target_mask=tgt, guidance_text=text, draw_mask=kw.get("draw_mask"))
            for k_ in ("use_auto_draw", "cons_area", "reduce_inp_artifacts"):
                c0[k_] = kw[k_]
                for c_ in cases:
                    c_[k_] = kw[k_] if k_ != "cons_area" or kw[k_] is None else np.ones_like(kw[k_])
            bkw = {k_: kw[k_] for k_ in ("end_step", "num_step", "start_step", "method_type", "end_scale")}
            model.FreeFine_generation_batch([c0, cases[1], cases[2]], gs, eta, seeds=[42, 7, 1234], return_intermediates=True, **bkw)
            devb = traj_dev(model.last_intermediates[0], ref)
            print(f"{'split-bf16' if x3 else 'fp32'} {name}, image 0 of a batch of 3: ABSOLUTE latent L-inf vs reference golden {devb:.2e}")
            assert devb < TOL, name


@pytest.mark.parametrize("case", ["fs_edit_s35", "fs_edit_s0", "fs_edit_n20"])
def test_full_size_n50_schedules_vs_oracle_fixture(gpu, case):
    """BASELINE's metric configuration end to end at FULL size on its own schedule: SD-2.1-base UNet (865.9 M parameters), 512x512 images,
    N = 50 with start_step = 35 (the GeoBench-2D call site: 15 inversion + 15 guided forwards) and start_step = 0 (the metric's "50-step":
    50 + 50 forwards; planted denoiser path, freefine_amd.weights.plant_denoiser_path, so that the trajectory is a denoising one) and BASELINE
    configs[0]'s 20-step plumbing run (N = 20, start_step = 0: the CPU run of that configuration IS the fixture), TCA in
    blocks 10-15, masked CFG 7.5, eta = 1 -- against the trajectory OraclePipeline produced in the build container
    (tests/golden/g10_fullsize_*.npz, tools/gen_fullsize_traj.py).  Gate: ABSOLUTE latent L-inf <= 1e-3 at every step of the edited row, in
    fp32 parity mode and in the split-bf16 mode (bench.py's headline mode); the bf16 fast mode is measured and printed, not gated."""
    from golden_cases import fullsize_cases, fullsize_inputs
    from freefine_amd.config import UNetConfig
    from freefine_amd.weights import plant_denoiser_path
    from oracle import sd_unet
    g = np.load(os.path.join(GOLD, f"g10_fullsize_{case}.npz"))
    ori_img, coarse, ori, tgt, draw, cons_sup = fullsize_inputs()
    planted, kw = fullsize_cases()[case]
    kw = dict(kw)
    text, gs, eta = kw.pop("guidance_text"), kw.pop("guidance_scale"), kw.pop("eta")
    ust = sd_unet.init_unet(sd_unet.unet_config("sd21-base"), seed=0).state_dict()
    if planted > 0:
        ust = plant_denoiser_path(ust, UNetConfig.preset("sd21-base"), planted)
    ref_e, ref_r = g["traj_edit"], g["traj_ref"]
    for mode, dtype, x3 in (("fp32", torch.float32, False), ("split-bf16", torch.float32, True), ("bf16", torch.bfloat16, False)):
        model = make_pipe(gpu, "sd21-base", "edit", dtype=dtype, graph=True, ustate=ust, x3=x3)
        img = model.FreeFine_generation(ori_img, ori, coarse, tgt, text, gs, eta, verbose=True, seed=42, return_intermediates=True, **kw)
        traj = torch.stack([t.detach().float().cpu() for t in model.last_intermediates])
        assert traj.shape[0] == ref_e.shape[0] == kw["num_step"] - kw["start_step"] + 1
        d_e = (traj[:, 0] - torch.from_numpy(ref_e)).abs().flatten(1).max(1).values
        d_r = (traj[::5, 1] - torch.from_numpy(ref_r)).abs().flatten(1).max(1).values
        print(f"full-size {case}, {mode}: ABSOLUTE latent L-inf vs oracle fixture over {traj.shape[0] - 1} guided steps: edited row max {d_e.max():.2e} "
              f"(final {d_e[-1]:.2e}), reference row {d_r.max():.2e}; |latent| max {np.abs(ref_e).max():.2f}; image max |diff| "
              f"{np.abs(img[::4, ::4].astype(int) - g['img'].astype(int)).max()}")
        if mode != "bf16":
            assert d_e.max() < TOL and d_r.max() < TOL, (case, mode)
            assert np.abs(img[::4, ::4].astype(int) - g["img"].astype(int)).max() <= 1
        del model
        torch.cuda.empty_cache()


def test_full_size_bench_layout_24_edits_per_batch_vs_oracle_fixture(gpu):
    """bench.py's DEFAULT layout itself against the oracle: 24 independent edits in one image-major UNet batch (48 rows in the inversion pass, 72 physical
    rows in the guided pass: row de-duplication, reference-stream reuse, hipGraphs, attention in row ranges of 16), split-bf16 mode, full-size SD-2.1
    topology, the metric's schedule (N = 50, start_step = 0: 50 + 50 forwards, planted denoiser path).  Image 0 of the batch is the fs_edit_s0 fixture's
    case (tests/golden/g10_fullsize_fs_edit_s0.npz, generated by OraclePipeline in the build container); the other 23 have their own images, masks and
    seeds.  Gate: ABSOLUTE latent L-inf <= 1e-3 at every step for image 0 (edited and reference row), every other image finite with its own result."""
    from golden_cases import fullsize_cases, fullsize_inputs
    from freefine_amd.config import UNetConfig
    from freefine_amd.weights import plant_denoiser_path
    from oracle import sd_unet
    g = np.load(os.path.join(GOLD, "g10_fullsize_fs_edit_s0.npz"))
    ori_img, coarse, ori, tgt, draw, cons_sup = fullsize_inputs()
    planted, kw = fullsize_cases()["fs_edit_s0"]
    kw = dict(kw)
    text, gs, eta = kw.pop("guidance_text"), kw.pop("guidance_scale"), kw.pop("eta")
    kw = {k: kw[k] for k in ("end_step", "num_step", "start_step", "method_type", "end_scale")}
    ust = plant_denoiser_path(sd_unet.init_unet(sd_unet.unet_config("sd21-base"), seed=0).state_dict(), UNetConfig.preset("sd21-base"), planted)
    K = 24
    cases, seeds = [], []
    for i in range(K):
        dy, dx = 8 * (i % 5), 8 * (i // 5)                         # image i: the fixture's inputs moved by (dy, dx) (images rolled, masks shifted: they stay inside)
        sh = lambda a: np.roll(np.roll(a, dy, 0), dx, 1)
        cases.append(dict(ori_img=sh(ori_img), ori_mask=sh(ori), coarse_input=sh(coarse), target_mask=sh(tgt), guidance_text=text if i % 3 else text + " " * (i > 0),
                          draw_mask=sh(draw)))
        seeds.append(42 + 17 * i)
    model = make_pipe(gpu, "sd21-base", "edit", graph=True, ustate=ust, x3=True)
    imgs = model.FreeFine_generation_batch(cases, gs, eta, seeds=seeds, return_intermediates=True, verbose=False, **kw)
    assert len(imgs) == K and all(im.shape == (512, 512, 3) and im.dtype == np.uint8 for im in imgs)
    traj = torch.stack([t.detach().float().cpu() for t in model.last_intermediates[0]])
    ref_e, ref_r = g["traj_edit"], g["traj_ref"]
    assert traj.shape[0] == ref_e.shape[0] == kw["num_step"] - kw["start_step"] + 1
    d_e = (traj[:, 0] - torch.from_numpy(ref_e)).abs().flatten(1).max(1).values
    d_r = (traj[::5, 1] - torch.from_numpy(ref_r)).abs().flatten(1).max(1).values
    print(f"24 edits per UNet batch, split-bf16, N=50 start_step=0: image 0 ABSOLUTE latent L-inf vs oracle fixture: edited row max {d_e.max():.2e} (final {d_e[-1]:.2e}), "
          f"reference row {d_r.max():.2e}; image max |diff| {np.abs(imgs[0][::4, ::4].astype(int) - g['img'].astype(int)).max()}")
    assert d_e.max() < TOL and d_r.max() < TOL
    assert np.abs(imgs[0][::4, ::4].astype(int) - g["img"].astype(int)).max() <= 1
    finals = torch.stack([model.last_intermediates[j][-1].detach().float().cpu() for j in range(K)])
    assert torch.isfinite(finals).all()
    for j in range(1, K):                                          # every image got its own result (no row of image 0 leaked into another slot)
        assert (finals[j] - finals[0]).abs().max() > 1e-3, j
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("case", ["fs_bg_s1", "fs_cmp_s15"])
def test_full_size_n50_other_hooks_vs_oracle_fixture(gpu, case):
    """The other two hooks at FULL size on the metric's N = 50 schedules (round 5; G10 covered the edit hook only): background generation at
    start_step 1 (freefine_batch_infer_bggen_2d.py:149,166-180: 49 + 49 forwards, planted denoiser path) and the cross-image composition with
    R = 2 references at start_step 15 (Appearance_transfer.ipynb / SURVEY 8d C4: 35 + 35 forwards of 3 / 4 rows, stored reference K / V on) --
    SD-2.1-base topology, 512^2, against the trajectory OraclePipeline produced in the build container (tests/golden/g10_fullsize_<case>.npz,
    tools/gen_fullsize_traj.py).  Gate: ABSOLUTE latent L-inf <= 1e-3 at every step of the edited row, fp32 parity mode and split-bf16 mode,
    through the captured hipGraphs; bf16 printed.  Reference loops: /root/reference/src/demo/model.py:656-812, 301-435."""
    from golden_cases import fullsize_hook_cases, fullsize_hook_inputs
    from freefine_amd.config import UNetConfig
    from freefine_amd.weights import plant_denoiser_path
    from oracle import sd_unet
    g = np.load(os.path.join(GOLD, f"g10_fullsize_{case}.npz"))
    img0, coarse0, img2, oris, tgts = fullsize_hook_inputs()
    hook, planted, kw = fullsize_hook_cases()[case]
    ust = sd_unet.init_unet(sd_unet.unet_config("sd21-base"), seed=0).state_dict()
    if planted > 0:
        ust = plant_denoiser_path(ust, UNetConfig.preset("sd21-base"), planted)
    ref = g["traj_edit"]
    for mode, dtype, x3 in (("fp32", torch.float32, False), ("split-bf16", torch.float32, True), ("bf16", torch.bfloat16, False)):
        model = make_pipe(gpu, "sd21-base", hook, dtype=dtype, graph=True, ustate=ust, x3=x3)
        if hook == "bggen":
            img = model.FreeFine_background_generation(img0, model.dilate_mask(oris[0] // 255, 30), "empty scene", 7.5, 1.0, verbose=True, seed=7,
                                                       return_intermediates=True, **kw)
        else:
            img = model.FreeFine_cross_image_composition([img0, img2], oris, tgts, coarse0, ["a cup", "a dog"], 7.5, 1.0, verbose=True, seed=11,
                                                         return_intermediates=True, **kw)
        traj = torch.stack([(t if t.ndim == 3 else t[0]).detach().float().cpu() for t in model.last_intermediates])
        assert traj.shape[0] == ref.shape[0] == kw["num_step"] - kw["start_step"] + 1
        d = (traj - torch.from_numpy(ref)).abs().flatten(1).max(1).values
        print(f"full-size {case} ({hook}), {mode}: ABSOLUTE latent L-inf vs oracle fixture over {traj.shape[0] - 1} guided steps: max {d.max():.2e} "
              f"(final {d[-1]:.2e}); |latent| max {np.abs(ref).max():.2f}; image max |diff| {np.abs(img[::4, ::4].astype(int) - g['img'].astype(int)).max()}")
        if mode != "bf16":
            assert d.max() < TOL, (case, mode)
            assert np.abs(img[::4, ::4].astype(int) - g["img"].astype(int)).max() <= 1
        del model
        torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------------------------------
# round 6: the checkpoint-folder branch of from_pretrained (tools/make_synthetic_checkpoint.py writes the HF layout offline)
# ---------------------------------------------------------------------------------------------------------------------
def _hook_edit(model):
    from freefine_amd.attention import Attention_Modulator, register_attention_control
    from freefine_amd.scheduler import DDIMScheduler
    model.scheduler = DDIMScheduler.from_config(model.scheduler.config)
    model.controller = Attention_Modulator(start_layer=10)
    register_attention_control(model, model.controller)
    model.modify_unet_forward()
    return model


@pytest.mark.parametrize("fdtype", ["fp32", "fp16"])
def test_from_pretrained_folder_equals_from_state_bit_for_bit(gpu, tmp_path, fdtype):
    """FreeFinePipeline.from_pretrained(<HF-layout folder>) -- the first line of every reference call site (freefine_batch_infer_2d.py:148-157) --
    against from_state on the tensors the folder was written from (fp16 file: on their fp16 roundings): one small edit through both pipelines,
    same transformers tokenizer / text encoder objects' weights, latent trajectory and image BIT FOR BIT equal.  Covers config.json parsing,
    safetensors loading + up-cast, legacy VAE attention names, scheduler_config.json and the CLIP wiring on the device."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_synthetic_checkpoint as M
    from transformers import CLIPTextModel, CLIPTokenizer
    from freefine_amd.pipeline import FreeFinePipeline
    d = str(tmp_path / "sd_tiny")
    ucfg, vcfg, ust, vst = M.write(d, "tiny", "tiny", fdtype, seed=5)
    a = _hook_edit(FreeFinePipeline.from_pretrained(d, torch_dtype=torch.float32, device=gpu).to(gpu))
    if fdtype == "fp16":
        ust, vst = ({k: v.to(torch.float16).float() for k, v in st.items()} for st in (ust, vst))
    tok = CLIPTokenizer.from_pretrained(os.path.join(d, "tokenizer"))
    enc = CLIPTextModel.from_pretrained(os.path.join(d, "text_encoder")).eval()
    b = _hook_edit(FreeFinePipeline.from_state(ucfg, ust, vcfg, vst, tok, enc, None, torch.float32, gpu))
    assert a.scheduler.config.steps_offset == 1 and a.scheduler.config.set_alpha_to_one is False
    ori_img, coarse, _ = synth_images()
    ori, tgt, *_ = mask_inputs()
    kw = dict(edit_cases()[0][2])                        # the first golden edit configuration (its masks / schedule), with a real prompt
    kw.pop("guidance_text")
    gs, eta = kw.pop("guidance_scale"), kw.pop("eta")
    kw.update(seed=42, return_intermediates=True, verbose=False)
    ia = a.FreeFine_generation(ori_img, ori, coarse, tgt, "a photo of a cup", gs, eta, **kw)
    ib = b.FreeFine_generation(ori_img, ori, coarse, tgt, "a photo of a cup", gs, eta, **kw)
    assert len(a.last_intermediates) == len(b.last_intermediates) > 0
    for x, y in zip(a.last_intermediates, b.last_intermediates):
        assert torch.equal(x, y)
    assert np.array_equal(ia, ib) and np.isfinite(ia.astype(float)).all()
    assert a.text_encoder_calls > 0 and isinstance(a.text_encoder, torch.nn.Module)


def test_geobench_driver_runs_from_a_checkpoint_folder(gpu, tmp_path):
    """`python evaluation/FreeFine/freefine_batch_infer_2d.py --base-dir <GeoBenchMeta> --model <SD folder>`: the reference's own command line with a
    folder (synthetic weights, tiny topology) on the synthetic GeoBench tree -- the driver process builds its pipeline through from_pretrained."""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_synthetic_checkpoint as M
    from freefine_amd import geobench
    d = str(tmp_path / "sd_tiny")
    M.write(d, "tiny", "tiny", "fp32", seed=1)
    root = str(tmp_path / "geo")
    geobench.make_synthetic_dataset(root, n_images=1, edits_per_image=2, size=128, seed=3, with_backgrounds=True)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "evaluation", "FreeFine", "freefine_batch_infer_2d.py"), "--base-dir", root, "--model", d,
                          "--dtype", "f32", "--batch", "2"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert os.path.exists(os.path.join(root, "generated_results_freefine_2d.json"))
    pngs = [f for _, _, fs in os.walk(os.path.join(root, geobench.GEN_SUBDIR)) for f in fs if f.endswith(".png")]
    assert len(pngs) >= 2, pngs
