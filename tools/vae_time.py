"""Time of the VAE bracket of one 16-image edit batch (encode 32 images: coarse + original; decode 16 latents), bf16, with the per-kernel
event profile of the decode.  python tools/vae_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freefine_amd import ops  # noqa: E402
from freefine_amd.config import VAEConfig  # noqa: E402
from freefine_amd.vae import HipVAE  # noqa: E402
from freefine_amd.weights import synthetic_state, vae_param_shapes  # noqa: E402

dev = torch.device("cuda:0")
cfg = VAEConfig.preset("sd")
MODE = os.environ.get("VAE_MODE", "bf16")          # bf16 | x3 (the headline mode: fp32 storage, split-bf16 GEMMs)
vae = HipVAE(cfg, synthetic_state(vae_param_shapes(cfg), 1), torch.float32 if MODE == "x3" else torch.bfloat16, dev, **({"x3": True} if MODE == "x3" else {}))
NIMG = int(os.environ.get("VAE_N", 16))          # images per call (the pipeline decodes / encodes in chunks of 6 at 24 edits per batch)
img = torch.randint(0, 256, (NIMG, 512, 512, 3), dtype=torch.uint8, device=dev)
lat = torch.randn(NIMG, 4, 64, 64, device=dev)


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


te = timeit(lambda: vae.encode_mean_scaled(img_u8=img))
td = timeit(lambda: vae.decode_image(lat))
print(f"encode {NIMG} images: {te:.1f} ms, decode {NIMG} latents: {td:.1f} ms -> VAE bracket of a {NIMG}-image batch (2 encodes + 1 decode): {2 * te + td:.1f} ms")
for name, fn in (("encode", lambda: vae.encode_mean_scaled(img_u8=img)), ("decode", lambda: vae.decode_image(lat))):
    ops.profile_begin()
    fn()
    prof = ops.profile_end()
    rows = sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:8]
    tot = sum(v["total_ms"] for v in prof.values())
    print(name, f"(event-timed eager launches: {tot:.1f} ms)")
    for k, v in rows:
        print(f"   {v['total_ms']:8.2f} ms  {v['calls']:4d} calls  {v['flops'] / max(v['total_ms'], 1e-9) / 1e9:7.0f} TFLOP/s  {k[:110]}")

# the mid-block attention alone (q k^T -> softmax_rows -> P V per image through [S, S] in HBM, vae.py _attention): its share of the bracket
x = torch.randn(NIMG, 4096, 512, device=dev).to(vae.dtype)
ta = timeit(lambda: vae._attention(vae.dec_mid.attn, x, NIMG, 4096))
print(f"mid-block attention, {NIMG} images: {ta:.2f} ms = {100 * ta / td:.1f} % of the decode, {100 * 3 * ta / (2 * te + td):.1f} % of the bracket (encode and decode have one each)")
