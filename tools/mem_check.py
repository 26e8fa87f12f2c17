"""peak device memory of the default bench configuration (GPU box): python tools/mem_check.py [--batch 16] [--concurrent 2]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=24)
ap.add_argument("--dtype", default="bf16x3")
ap.add_argument("--concurrent", type=int, default=2)
a = ap.parse_args()
args = argparse.Namespace(model="sd21-base", vae="sd", dtype=a.dtype, planted=3.0, text="clip", fp8_conv=False, no_graph=False, no_dedup=False, num_step=50, start_step=0, batch=a.batch)
dev = torch.device("cuda:0")
model = bench.build_model(args, dev, 0, 1)
models = [model] + [bench.add_sibling(model) for _ in range(a.concurrent - 1)]
streams = [torch.cuda.Stream(device=dev) for _ in models]
for j, (m, st) in enumerate(zip(models, streams)):
    with torch.cuda.stream(st):
        bench.edit_once(m, args, j)
        bench.edit_once(m, args, 10 + j)
    st.synchronize()
free, total = torch.cuda.mem_get_info()
print(f"dtype={a.dtype} batch={a.batch} concurrent={a.concurrent}: max_allocated {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, reserved "
      f"{torch.cuda.memory_reserved() / 2**30:.1f} GiB, device used {(total - free) / 2**30:.1f} of {total / 2**30:.1f} GiB")
