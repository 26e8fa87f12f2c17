"""Generate tests/golden/g10_fullsize_<case>.npz: latent trajectories of FULL-SIZE edits (SD-2.1-base topology, 865.9 M parameters, 512x512
images -> 64x64 latents) on the metric's own schedules (N = 50; tests/golden_cases.py fullsize_cases) from the CPU oracle (OraclePipeline,
itself pinned to the reference by G1-G9).  Build container only (7 / 25 minutes on 8 host threads):

    python tools/gen_fullsize_traj.py fs_edit_s35 fs_edit_s0

The fixture holds the trajectory (fp32) -- every step of the edited row, every 5th step of the reference row -- and the sub-sampled image.
The GPU test (tests/test_pipeline_gpu.py::test_full_size_n50_schedules_vs_oracle_fixture) rebuilds weights and inputs from the same seeds."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_cases import fullsize_cases, fullsize_hook_cases, fullsize_hook_inputs, fullsize_inputs  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def fullsize_oracle(planted):
    from freefine_amd.config import UNetConfig
    from freefine_amd.text import ByteTokenizer, SyntheticTextEncoder, make_text_embed
    from freefine_amd.weights import plant_denoiser_path
    from oracle import sd_unet, sd_vae
    from oracle.pipeline import OraclePipeline
    cfg = sd_unet.unet_config("sd21-base")
    unet = sd_unet.init_unet(cfg, seed=0)
    if planted > 0:
        unet.load_state_dict(plant_denoiser_path(unet.state_dict(), UNetConfig.preset("sd21-base"), planted))
    return OraclePipeline(unet, sd_vae.init_vae(sd_vae.vae_config("tiny"), seed=1),
                          make_text_embed(ByteTokenizer(), SyntheticTextEncoder(cfg.cross_attention_dim)))


def main():
    torch.set_num_threads(int(os.environ.get("FFN_THREADS", "8")))
    ori_img, coarse, ori, tgt, draw, cons_sup = fullsize_inputs()
    cases = fullsize_cases()
    hook_cases = fullsize_hook_cases()
    for name in sys.argv[1:]:
        if name in hook_cases:                     # background generation / composition: the other two hooks (round 5)
            hook, planted, kw = hook_cases[name]
            img0, coarse0, img2, oris, tgts = fullsize_hook_inputs()
            op = fullsize_oracle(planted)
            t0 = time.time()
            if hook == "bggen":
                from oracle.masks import dilate_mask
                dil = dilate_mask(oris[0] // 255, 30)           # freefine_batch_infer_bggen_2d.py:149
                img_e, traj = op.freefine_background_generation(img0, dil, "empty scene", 7.5, 1.0, seed=7, **kw)
            else:
                img_e, traj = op.freefine_compose([img0, img2], oris, tgts, coarse0, ["a cup", "a dog"], 7.5, 1.0, seed=11, **kw)
            traj = torch.stack([(t if t.ndim == 3 else t[0]).float() for t in traj])      # [n + 1, 4, 64, 64]: the edited row
            print(f"{name}: {time.time() - t0:.0f} s, {traj.shape[0]} latents, |latent| max per step "
                  f"{[round(v, 2) for v in traj.abs().flatten(1).max(1).values.tolist()][::5]}", flush=True)
            np.savez_compressed(os.path.join(GOLD, f"g10_fullsize_{name}.npz"), traj_edit=traj.numpy(), img=img_e[::4, ::4].copy())
            continue
        planted, kw = cases[name]
        kw = dict(kw)
        text, gs, eta = kw.pop("guidance_text"), kw.pop("guidance_scale"), kw.pop("eta")
        op = fullsize_oracle(planted)
        t0 = time.time()
        img_e, img_r, traj = op.freefine_generation(ori_img, ori, coarse, tgt, text, gs, eta, seed=42, **kw)
        traj = torch.stack([t.float() for t in traj])                       # [n + 1, 2, 4, 64, 64]
        print(f"{name}: {time.time() - t0:.0f} s, {traj.shape[0]} latents, |latent| max per step "
              f"{[round(v, 2) for v in traj.abs().flatten(1).max(1).values.tolist()][::5]}", flush=True)
        np.savez_compressed(os.path.join(GOLD, f"g10_fullsize_{name}.npz"), traj_edit=traj[:, 0].numpy(), traj_ref=traj[::5, 1].numpy(),
                            img=img_e[::4, ::4].copy(), ref_img=img_r[::4, ::4].copy())


if __name__ == "__main__":
    main()
