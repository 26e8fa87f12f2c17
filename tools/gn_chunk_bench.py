"""GroupNorm (three-launch form, pair output) over whole batches vs row chunks sized for the Infinity Cache -- GPU box.
    python tools/gn_chunk_bench.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freefine_amd import ops

dev = torch.device("cuda:0")
shapes = [(72, 4096, 320), (72, 4096, 640), (72, 4096, 960), (72, 1024, 640), (72, 1024, 1280), (72, 1024, 1920), (48, 4096, 320), (48, 1024, 640)]
for B, HW, C in shapes:
    x = torch.randn(B, HW, C, device=dev)
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    line = f"[{B}, {HW}, {C}] {x.numel() * 4 / 2**20:7.0f} MiB:"
    ref = None
    for mb in (0, 24, 48, 96, 160):
        ops._GN_CHUNK_MB = float(mb)
        y = ops.groupnorm(x, g, b, 32, 1e-5, silu=True, pair=True)
        if ref is None:
            ref = y.clone()
        assert torch.equal(y, ref)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # a 512 MiB write between repetitions empties the cache of this tensor (the UNet never normalises the same tensor twice in a row)
        junk = torch.empty(128 * 2**20, device=dev)
        ts = []
        for _ in range(5):
            junk.fill_(1.0)
            e0.record()
            ops.groupnorm(x, g, b, 32, 1e-5, silu=True, pair=True)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        t = sorted(ts)[len(ts) // 2]
        line += f"  chunk {mb:3d} MiB {t:7.1f} us ({2 * x.numel() * 4 / t * 1e-6:4.2f} TB/s alg)"
    print(line, flush=True)
