"""A/B timing of the split-bf16 attention kernels: python tools/scratch/x3p_ab.py   (FFN_ATTN_PP=0 selects attn_x3_kernel)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from freefine_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
for (B, S, C, heads, passes) in ((8, 4096, 320, 5, 1), (8, 4096, 320, 5, 2), (16, 4096, 320, 5, 2), (16, 1024, 640, 10, 2), (16, 256, 1280, 20, 1)):
    q, k, vt = rnd(B, S, C), rnd(B, S, C), rnd(B, C, S)
    km = (torch.rand(S, generator=g) > 0.7).to(torch.uint8).to(dev)
    qs = (torch.rand(S, generator=g) > 0.5).to(torch.uint8).to(dev)
    cg = torch.tensor([0.5], device=dev)
    P = None if passes == 1 else [[ops.AttnEntrySpec(b, b | 1, 0.0, 1.0, kmask=km, qsel=qs, flags=1) for b in range(B)],
                                  [ops.AttnEntrySpec(b, b, 1.0, -1.0) for b in range(B)]]
    fn = lambda: ops.attention(q, k, vt, heads, (C // heads) ** -0.5, P, w_dev=cg, x3=True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    fl = 4.0 * passes * B * S * S * C
    print(f"B={B} S={S} C={C} h={heads} passes={passes}: {us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s effective  (PP={os.environ.get('FFN_ATTN_PP', '1')})")
