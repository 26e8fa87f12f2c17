"""Timings of the other task-level entry points at full size (GPU box): background generation (GeoBench bg-gen schedule) and
cross-image composition with R=2 references.  python tools/bench_other.py [--dtype bf16x3|bf16|f32]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from freefine_amd.attention import Attention_Modulator, register_attention_control_4bggen, register_attention_control_compose  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16x3", choices=["bf16", "f32", "bf16x3"])
cli = ap.parse_args()
args = argparse.Namespace(model="sd21-base", vae="sd", dtype=cli.dtype, no_graph=False, no_dedup=False, num_step=50, start_step=0, batch=1, planted=3.0,
                          text="clip", fp8_conv=False)
print(f"mode {cli.dtype}; planted-denoiser synthetic weights; real-size CLIP-shaped text encoder (prompt cache on)")
dev = torch.device("cuda:0")
model = bench.build_model(args, dev, 0, 1)
ori_img, ori_mask, coarse, tgt_mask, draw = bench.synth_inputs(0)
img2 = np.random.default_rng(7).integers(0, 256, (512, 512, 3), dtype=np.uint8)


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / reps


c = Attention_Modulator(start_layer=10)
model.controller = c
register_attention_control_4bggen(model, c)
hole = model.dilate_mask(ori_mask, 30)
t = timed(lambda: model.FreeFine_background_generation(ori_img, hole, "empty scene", 7.5, 1.0, end_step=35, num_step=50, start_step=1,
                                                       end_scale=0.5, seed=1, verbose=False))
print(f"background generation (N=50, S0=1, n=49), one image: {t * 1e3:.0f} ms")
cases = [dict(ori_img=ori_img, ori_mask=hole, guidance_text="empty scene")] * 8
t = timed(lambda: model.FreeFine_background_generation_batch(cases, 7.5, 1.0, end_step=35, num_step=50, start_step=1, end_scale=0.5, seeds=1,
                                                             verbose=False), reps=1)
print(f"background generation, 8 images per batch: {t / 8 * 1e3:.0f} ms per image")
c = Attention_Modulator(start_layer=10)
model.controller = c
register_attention_control_compose(model, c)
m2 = np.zeros((512, 512), np.uint8); m2[60:160, 300:420] = 255
t2 = np.zeros((512, 512), np.uint8); t2[320:420, 280:400] = 255
t = timed(lambda: model.FreeFine_cross_image_composition([ori_img, img2], [ori_mask * 255, m2], [tgt_mask, t2], coarse, ["a cup", "a dog"], 7.5, 1.0,
                                                         end_step=50, num_step=50, start_step=15, seed=3, dil_factor=15, end_scale=0.5, verbose=False))
compose = lambda: model.FreeFine_cross_image_composition([ori_img, img2], [ori_mask * 255, m2], [tgt_mask, t2], coarse, ["a cup", "a dog"], 7.5, 1.0,
                                                         end_step=50, num_step=50, start_step=15, seed=3, dil_factor=15, end_scale=0.5, verbose=False)
print(f"cross-image composition R=2 (N=50, S0=15, n=35), stored reference K/V ({model.unet.reuse_replays} replayed forwards so far): {t * 1e3:.0f} ms")
model.reuse_ref_stream = False
print(f"cross-image composition R=2, reference rows recomputed like the reference does: {timed(compose) * 1e3:.0f} ms")
model.reuse_ref_stream = True
ccases = [dict(img_lists=[ori_img, img2], ori_mask_lists=[ori_mask * 255, m2], tgt_mask_lists=[tgt_mask, t2], coarse_input=coarse, guidance_text_list=["a cup", "a dog"])] * 8
for on in (True, False):
    model.reuse_ref_stream = on
    t = timed(lambda: model.FreeFine_cross_image_composition_batch(ccases, 7.5, 1.0, end_step=50, num_step=50, start_step=15, seeds=3, dil_factor=15,
                                                                   end_scale=0.5, verbose=False), reps=1)
    print(f"cross-image composition R=2, 8 images per batch, stored reference K/V {'on' if on else 'off'}: {t / 8 * 1e3:.0f} ms per image")
