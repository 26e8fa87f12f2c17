"""Per-shape timing of the hot kernels at the UNet's real shapes (SD-2.1-base, 64x64 latent).  GPU box only.
    python tools/bench_kernels.py [--dtype bf16] [--B 4] [--only conv|gemm|attn|norm]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freefine_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--B", type=int, default=4)
ap.add_argument("--only", default="")
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
dev = torch.device("cuda:0")
B = a.B
g = torch.Generator().manual_seed(0)


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(dt).to(dev)


def timeit(fn):
    """device time per call: the calls are captured into a hipGraph so host launch overhead (~15 us through ctypes) is excluded"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g_ = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_):
        for _ in range(a.iters):
            fn()
    g_.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g_.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / a.iters * 1e3)
    return best  # us


def report(name, us, flops, nbytes):
    print(f"{name:46s} {us:9.1f} us  {flops / us / 1e6:8.1f} TFLOP/s  {nbytes / us / 1e3:8.1f} GB/s", flush=True)


if a.only in ("", "conv"):
    for (hw, cin, cout, stride, up) in [(64, 320, 320, 1, 0), (64, 960, 320, 1, 0), (64, 640, 320, 1, 0), (32, 640, 640, 1, 0), (32, 1920, 640, 1, 0),
                                        (32, 1280, 1280, 1, 1), (16, 1280, 1280, 1, 0), (16, 2560, 1280, 1, 0), (8, 1280, 1280, 1, 0), (8, 2560, 1280, 1, 0),
                                        (64, 320, 320, 2, 0), (64, 320, 4, 1, 0)]:
        hin = hw // 2 if up else hw
        x = rnd(B, hin * hin, cin)
        w = ops.pack_conv3x3(rnd(cout, cin, 3, 3, scale=(9 * cin) ** -0.5), dt)
        b = torch.zeros(cout, device=dev)
        ho = hw // stride
        us = timeit(lambda: ops.conv3x3(x, w, b, B, hin, hin, cin, stride=stride, upsample=bool(up)))
        M = B * ho * ho
        report(f"conv3x3 {hw}x{hw} {cin}->{cout} s{stride} up{up} (M={M})", us, 2.0 * M * cout * 9 * cin, 2 * (x.numel() + w.numel() + M * cout))

if a.only in ("", "gemm"):
    for (hw, K, N, kind) in [(64, 320, 640, "qk"), (64, 320, 320, "res"), (64, 320, 2560, "geglu"), (64, 1280, 320, "res"), (32, 640, 1280, "qk"),
                             (32, 640, 5120, "geglu"), (32, 2560, 640, "res"), (16, 1280, 2560, "qk"), (16, 1280, 10240, "geglu"), (16, 5120, 1280, "res"),
                             (8, 1280, 10240, "geglu"), (8, 5120, 1280, "res"), (64, 320, 320, "vt")]:
        M = B * hw * hw
        x = rnd(B, hw * hw, K)
        w = rnd(N, K, scale=K ** -0.5)
        if kind == "geglu":
            wp, bp = ops.pack_geglu(w, torch.zeros(N, device=dev), dt)
            fn = lambda: ops.linear(x, wp, bp, geglu=True)
        elif kind == "res":
            wp = ops.pack_linear(w, dt)
            r = rnd(B, hw * hw, N)
            bb = torch.zeros(N, device=dev)
            fn = lambda: ops.linear(x, wp, bb, residual=r)
        elif kind == "vt":
            wp = ops.pack_linear(w, dt)
            fn = lambda: ops.linear(x, wp, None, rows_per_batch=hw * hw, transposed_ld=hw * hw)
        else:
            wp = ops.pack_linear(w, dt)
            fn = lambda: ops.linear(x, wp, None)
        us = timeit(fn)
        report(f"gemm {kind:5s} M={M} K={K} N={N}", us, 2.0 * M * N * K, 2 * (M * K + N * K + M * N))

if a.only in ("", "attn"):
    for (hw, C, heads, Sk, passes) in [(64, 320, 5, 4096, 1), (64, 320, 5, 4096, 2), (32, 640, 10, 1024, 1), (32, 640, 10, 1024, 2), (16, 1280, 20, 256, 1),
                                       (8, 1280, 20, 64, 1), (64, 320, 5, 77, 1), (32, 640, 10, 77, 1), (16, 1280, 20, 77, 1)]:
        S = hw * hw
        q, k = rnd(B, S, C), rnd(B, Sk, C)
        vt = rnd(B, C, (Sk + 7) // 8 * 8)
        D = C // heads
        km = (torch.rand(Sk, generator=g) > 0.7).to(torch.uint8).to(dev)
        qs = (torch.rand(S, generator=g) > 0.5).to(torch.uint8).to(dev)
        cg = torch.tensor([0.5], device=dev)
        if passes == 2:
            P = [[ops.AttnEntrySpec(b, [1, 1, 3, 3][b % 4] % B, 0.0, 1.0, kmask=km, qsel=qs, flags=1) for b in range(B)],
                 [ops.AttnEntrySpec(b, b, 1.0, -1.0) for b in range(B)]]
        else:
            P = None
        us = timeit(lambda: ops.attention(q, k, vt, heads, D ** -0.5, P, Sk=Sk, w_dev=cg))
        report(f"attn S={S} Sk={Sk} C={C} h={heads} passes={passes}", us, 4.0 * passes * B * S * Sk * C, 2 * B * (2 * S * C + 2 * passes * Sk * C))

if a.only in ("", "norm"):
    for (hw, C) in [(64, 320), (64, 960), (32, 640), (32, 1920), (16, 1280), (16, 2560), (8, 1280), (8, 2560)]:
        x = rnd(B, hw * hw, C)
        gm, bt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        us = timeit(lambda: ops.groupnorm(x, gm, bt, 32, 1e-5, silu=True))
        report(f"groupnorm+silu {hw}x{hw} C={C}", us, 0, 2 * 3 * x.numel())
    for (hw, C) in [(64, 320), (32, 640), (16, 1280)]:
        x = rnd(B, hw * hw, C)
        gm, bt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        us = timeit(lambda: ops.layernorm(x, gm, bt))
        report(f"layernorm {hw}x{hw} C={C}", us, 0, 2 * 2 * x.numel())
