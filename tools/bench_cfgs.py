"""Time every valid bf16 igemm configuration on a few dense shapes (GPU box): python tools/bench_cfgs.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freefine_amd import _lib as L  # noqa: E402
from freefine_amd import ops  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
dt = torch.bfloat16
g = torch.Generator().manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dt).to(dev)
NAMES = ["64x64", "128x64", "128x128/8w", "128x128/16w", "256x128", "256x256", "128x320", "128x160", "192x320", "H128x320", "H256x128", "H256x256", "H128x128", "PP256x320", "PP256x256", "PP192x320", "PP192x256"]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(iters):
            fn()
    best = 1e30
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); gr.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters * 1e3)
    return best


shapes = [(98304, 320, 320, False), (98304, 640, 320, False), (98304, 2560, 320, True), (98304, 320, 1280, False), (24576, 640, 640, False),
          (24576, 5120, 640, True), (98304, 320, -320, False), (98304, 320, -640, False)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.rstrip("g").split(",")[:3]) + (a.endswith("g"),) for a in sys.argv[1:]]
for (M, N, K, geglu) in shapes:
    if K < 0:       # conv: K = -Cin, M = B*64*64
        cin, B = -K, M // 4096
        xc = rnd(B, 4096, cin)
        wc = ops.pack_conv3x3(rnd(N, cin, 3, 3, scale=(9 * cin) ** -0.5), dt)
        bc = torch.zeros(N, device=dev)
        row = []
        for cfg in range(lib.ffn_igemm_num_configs()):
            lib.ffn_igemm_force_config(cfg)
            row.append(f"{NAMES[cfg]}={timeit(lambda: ops.conv3x3(xc, wc, bc, B, 64, 64, cin)):.0f}")
        lib.ffn_igemm_force_config(-1)
        print(f"conv M={M} N={N} Cin={cin}: " + "  ".join(row), flush=True)
        continue
    x = rnd(M, K)
    w = ops.pack_linear(rnd(N, K, scale=K ** -0.5), dt)
    out = torch.empty(M, N // 2 if geglu else N, dtype=dt, device=dev)
    row = []
    for cfg in range(lib.ffn_igemm_num_configs()):
        lib.ffn_igemm_force_config(cfg)
        buf = L.C.create_string_buffer(200) if hasattr(L, "C") else None
        us = timeit(lambda: ops.linear(x, w, None, K=K, geglu=geglu, out=out))
        row.append(f"{NAMES[cfg]}={us:.0f}")
    lib.ffn_igemm_force_config(-1)
    print(f"M={M} N={N} K={K} geglu={geglu}: " + "  ".join(row), flush=True)
