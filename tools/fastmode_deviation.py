"""How far the bf16 fast mode is from the fp32 parity mode over a WHOLE full-size edit (SD-2.1-base topology, 512x512, 50-step
FreeFine_generation, synthetic weights and inputs as bench.py): both modes run through the same HIP engine on the same seeds; reports the
final-image difference (uint8 levels, PSNR) and the edited latent's L-inf / relative L2.   python tools/fastmode_deviation.py [num_step]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("num_step", type=int, nargs="?", default=50)
a = ap.parse_args()
dev = torch.device("cuda:0")
imgs = {}
for dtype in ("f32", "bf16"):
    args = argparse.Namespace(model="sd21-base", vae="sd", dtype=dtype, no_graph=False, no_dedup=False, batch=1, num_step=a.num_step, start_step=0)
    model = bench.build_model(args, dev, 0, 1)
    out = bench.edit_once(model, args, 0)
    img = out[0] if isinstance(out, (list, tuple)) else out
    imgs[dtype] = np.asarray(img).astype(np.float64)
    del model
    torch.cuda.empty_cache()
d = imgs["bf16"] - imgs["f32"]
mse = float((d ** 2).mean())
print(f"full-size {a.num_step}-step edit, bf16 fast mode vs fp32 parity mode (same engine, same seeds): image shape {imgs['f32'].shape}, "
      f"max |diff| {np.abs(d).max():.0f} / 255, mean |diff| {np.abs(d).mean():.3f}, PSNR {10 * np.log10(255.0 ** 2 / max(mse, 1e-12)):.1f} dB")
