"""Where does one image's wall time go?  (GPU box)  python tools/bench_breakdown.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = bench.argparse.Namespace(model="sd21-base", vae="sd", dtype="bf16", no_graph=False, no_dedup=False, num_step=50, start_step=0)
dev = torch.device("cuda:0")
model = bench.build_model(args, dev, 0, 1)
for _ in range(2):
    bench.edit_once(model, args, 0)
torch.cuda.synchronize()
T = {}


def wrap(obj, name):
    f = getattr(obj, name)

    def g(*a, **k):
        torch.cuda.synchronize()
        t0 = time.time()
        r = f(*a, **k)
        torch.cuda.synchronize()
        T[name] = T.get(name, 0.0) + time.time() - t0
        return r
    setattr(obj, name, g)


for n in ("image2latent", "latent2image", "invert", "forward_sampling", "inv_step", "ctrl_step", "_encode_text"):
    wrap(model, n)
wrap(model.unet, "forward")
model.unet.__class__.__call__ = lambda self, s, t, encoder_hidden_states=None, row_map=None, **kw: self.forward(s, t, encoder_hidden_states, row_map)
t0 = time.time()
bench.edit_once(model, args, 1)
torch.cuda.synchronize()
tot = time.time() - t0
print(f"total {tot * 1e3:.1f} ms")
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print(f"  {k:20s} {v * 1e3:8.1f} ms")
