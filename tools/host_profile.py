"""cProfile of the host side of one image (GPU box)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = bench.argparse.Namespace(model="sd21-base", vae="sd", dtype=os.environ.get("HP_DTYPE", "bf16x3"), no_graph=False, no_dedup=False, num_step=50, start_step=0,
                                batch=int(os.environ.get("HP_BATCH", "1")), planted=3.0, text="table", fp8_conv=False)
model = bench.build_model(args, torch.device("cuda:0"), 0, 1)
for _ in range(2):
    bench.edit_once(model, args, 0)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
bench.edit_once(model, args, 1)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(30)
