"""Quick UNet-forward timing on the GPU box: python tools/bench_unet.py [--dtype bf16|f32] [--B 4] [--graph] [--tca]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freefine_amd.config import UNetConfig  # noqa: E402
from freefine_amd.unet import HipUNet  # noqa: E402
from freefine_amd.weights import synthetic_state, unet_param_shapes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--model", default="sd21-base")
ap.add_argument("--B", type=int, default=4)
ap.add_argument("--hw", type=int, default=64)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--graph", action="store_true")
ap.add_argument("--tca", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = UNetConfig.preset(a.model)
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
t0 = time.time()
net = HipUNet(cfg, synthetic_state(unet_param_shapes(cfg), 0), dtype=dt, device=dev)
print(f"packed in {time.time() - t0:.1f}s", flush=True)
if a.tca:
    from types import SimpleNamespace
    from freefine_amd.attention import Attention_Modulator, register_attention_control
    c = Attention_Modulator(start_layer=10)
    register_attention_control(SimpleNamespace(unet=net), c)
    m1 = torch.zeros(512, 512, dtype=torch.uint8); m1[200:300, 100:200] = 1
    m2 = torch.zeros(512, 512, dtype=torch.uint8); m2[200:300, 160:260] = 1
    c.layer_idx, c.local_edit, c.context_guidance, c.use_tca, c.method = list(range(10, 16)), True, 0.5, True, "tca"
    c.fg_retain_mask = c.fg_retain_mask_st2 = c.local_edit_region = m2
    c.fg_ref_mask = m1
g = torch.Generator().manual_seed(0)
x = torch.randn(a.B, 4, a.hw, a.hw, generator=g).to(dev)
enc = torch.randn(a.B, 77, cfg.cross_attention_dim, generator=g).to(dev)
net.use_graph = a.graph
for _ in range(3):
    y = net(x, 481, enc)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(a.iters):
    y = net(x, 481, enc)
torch.cuda.synchronize()
ms = (time.time() - t0) / a.iters * 1e3
flops = a.B * 0.804e12 * (a.hw / 64) ** 2
print(f"model={a.model} dtype={a.dtype} B={a.B} hw={a.hw} graph={a.graph} tca={a.tca}: {ms:.2f} ms/forward  ~{flops / ms / 1e9:.1f} TFLOP/s  finite={torch.isfinite(y).all().item()}")
