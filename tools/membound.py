"""Achieved HBM bandwidth of the memory-bound kernels of the UNet at the three attention levels (48 rows = 16 images x 3 physical rows,
bf16): µs per launch and algorithmic GB/s (bytes every launch must move at least once).  python tools/membound.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freefine_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
g = torch.Generator().manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dt).to(dev)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def report(name, us, nbytes):
    print(f"  {name:44s} {us:8.1f} us  {nbytes / us * 1e-3:7.0f} GB/s  ({nbytes / 1e6:.0f} MB)")


for hw, C in ((64, 320), (32, 640), (16, 1280)):
    S, heads = hw * hw, C // 64
    M = B * S
    print(f"level {hw}x{hw} C={C}: M = {M}")
    x = rnd(B, S, C)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    o = torch.empty_like(x)
    report("layernorm", timeit(lambda: ops.layernorm(x, gam, bet, out=o)), 4 * M * C)
    ws = ops.gn_workspace(B, S, C, dev)
    report("groupnorm + SiLU (stats + apply)", timeit(lambda: ops.groupnorm(x, gam, bet, 32, 1e-5, silu=True, out=o, ws=ws)), 6 * M * C)
    x2 = rnd(B, S, 2 * C)
    gam2, bet2 = torch.ones(2 * C, device=dev), torch.zeros(2 * C, device=dev)
    o2 = torch.empty_like(x2)
    ws2 = ops.gn_workspace(B, S, 2 * C, dev)
    report("groupnorm + SiLU on the 2C concat", timeit(lambda: ops.groupnorm(x2, gam2, bet2, 32, 1e-5, silu=True, out=o2, ws=ws2)), 6 * M * 2 * C)
    report("concat C|C", timeit(lambda: ops.concat(x, o, out=o2)), 8 * M * C)
    w = ops.pack_linear(rnd(C, C, scale=C ** -0.5), dt)
    bias = torch.zeros(C, device=dev)
    report("linear C->C", timeit(lambda: ops.linear(x, w, None, out=o)), 4 * M * C)
    r = rnd(B, S, C)
    report("linear C->C + bias + residual", timeit(lambda: ops.linear(x, w, bias, out=o, residual=r)), 6 * M * C)
    vt = torch.empty(B, C, S, dtype=dt, device=dev)
    report("linear C->C transposed (V^T)", timeit(lambda: ops.linear(x, w, None, out=vt, transposed_ld=S, rows_per_batch=S)), 4 * M * C)
    k = rnd(B, 77, C)
    vtx = rnd(B, C, 80)
    report("cross-attention Sk=77", timeit(lambda: ops.attention(x, k, vtx, heads, 0.125, Sk=77, out=o)), 4 * M * C)
    fw = torch.rand(S, generator=g).to(dev)
    ofw = 1.0 - fw
    def local_passes():                       # the guided pass's local cross-attention over images of 3 physical rows [u_e, ref, c_e]
        p0, p1 = [], []
        for r0 in range(0, B - 2, 3):
            p0 += [ops.AttnEntrySpec(r0, r0), ops.AttnEntrySpec(r0 + 1, r0 + 1), ops.AttnEntrySpec(r0 + 2, r0 + 2, wq=fw)]
            p1 += [None, None, ops.AttnEntrySpec(r0, r0, wq=ofw)]
        return [p0, p1]
    lp = local_passes()
    nb = len(lp[0])
    report("cross-attention Sk=77, local edit (2 passes, wq)", timeit(lambda: ops.attention(x, k, vtx, heads, 0.125, lp, Sk=77, out=o[:nb])), 4 * nb * S * C + 2 * (nb // 3) * S * C)
    w4 = ops.pack_linear(rnd(C, 4 * C, scale=(4 * C) ** -0.5), dt)
    h = rnd(B, S, 4 * C)
    report("linear 4C->C + bias + residual (FF out)", timeit(lambda: ops.linear(h, w4, bias, out=o, residual=r)), 2 * M * 4 * C + 4 * M * C)
    wg, bg = ops.pack_geglu(rnd(8 * C, C, scale=C ** -0.5), torch.zeros(8 * C, device=dev), dt)
    report("GEGLU C->8C->4C", timeit(lambda: ops.linear(x, wg, bg, K=C, geglu=True, out=h)), 2 * M * C + 2 * M * 4 * C)
