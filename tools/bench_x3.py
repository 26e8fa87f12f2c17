"""Per-shape timing of the split-bf16 (FFN_BF16X3) GEMMs / convolutions at the UNet's shapes (SD-2.1-base, 64x64 latent, image-batched).
GPU box only.   python tools/bench_x3.py [--rows 48]     (FREEFINE_HIP_LIB=<other .so> for A/B builds of the same ABI)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freefine_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=48, help="UNet batch rows (16 images x 3 physical rows)")
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
B = a.rows
g = torch.Generator().manual_seed(0)


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(dev)


def timeit(fn):
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
    return best


def report(name, us, flops):
    print(f"{name:52s} {us:9.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  {flops / us / 1e6 / 833.3:5.2f} of the 833 TFLOP/s split-bf16 ceiling", flush=True)


print("lib:", os.environ.get("FREEFINE_HIP_LIB", "default"))
for (hw, cin, cout) in [(64, 320, 320), (64, 640, 320), (32, 640, 640), (32, 1280, 640), (16, 1280, 1280), (16, 2560, 1280), (8, 1280, 1280)]:
    x = ops.split_pair(rnd(B, hw * hw, cin), cin)
    w = ops.pack_conv3x3(rnd(cout, cin, 3, 3, scale=(9 * cin) ** -0.5), torch.float32, None, True)
    b = torch.zeros(cout, device=dev)
    r = rnd(B, hw * hw, cout)
    for res in (None, r):
        us = timeit(lambda: ops.conv3x3(x, w, b, B, hw, hw, cin, residual=res))
        report(f"x3 conv3x3 {hw}x{hw} {cin}->{cout}{' +res' if res is not None else ''} (M={B * hw * hw})", us, 2.0 * B * hw * hw * cout * 9 * cin)
for (hw, K, N, kind) in [(64, 320, 640, "qk"), (64, 320, 320, "res"), (64, 320, 2560, "geglu"), (64, 1280, 320, "res"), (32, 640, 5120, "geglu"),
                         (32, 2560, 640, "res"), (16, 1280, 10240, "geglu"), (16, 5120, 1280, "res")]:
    M = B * hw * hw
    x = ops.split_pair(rnd(B, hw * hw, K), K)
    w = rnd(N, K, scale=K ** -0.5)
    if kind == "geglu":
        wp, bp = ops.pack_geglu(w, torch.zeros(N, device=dev), torch.float32, True)
        fn = lambda: ops.linear(x, wp, bp, K=K, geglu=True, out_pair=True)       # the pipeline's form: pair rows for ff.net.2 (without out_pair the launch
                                                                                 # takes the generic two-stage tile: the 0.25-0.32 of the round-5 table)
    elif kind == "res":
        wp = ops.pack_linear(w, torch.float32, True)
        r = rnd(B, hw * hw, N)
        bb = torch.zeros(N, device=dev)
        fn = lambda: ops.linear(x, wp, bb, K=K, residual=r)
    else:
        wp = ops.pack_linear(w, torch.float32, True)
        fn = lambda: ops.linear(x, wp, None, K=K)
    us = timeit(fn)
    report(f"x3 gemm {kind:5s} M={M} K={K} N={N}", us, 2.0 * M * N * K)
    if kind == "qk":        # round 6: the k half written as the attention kernels' pre-split image, and the V^T projection beside it
        us = timeit(lambda: ops.linear(x, wp, None, K=K, kv64_from=N // 2))
        report(f"x3 gemm qk    M={M} K={K} N={N} k half as [hi|lo] image", us, 2.0 * M * N * K)
        wv = ops.pack_linear(rnd(K, K, scale=K ** -0.5), torch.float32, True)
        for img in (None, 0):
            us = timeit(lambda: ops.linear(x, wv, None, K=K, rows_per_batch=hw * hw, transposed_ld=hw * hw, kv64_from=img))
            report(f"x3 gemm v^T   M={M} K={K} N={K}{' [hi|lo] image' if img is not None else ''}", us, 2.0 * M * K * K)
    if kind == "res" and N * 4 == K:      # ff.net.2: round 6 writes the residual sum as pair rows for proj_out
        us = timeit(lambda: ops.linear(x, wp, bb, K=K, residual=r, out_pair=True))
        report(f"x3 gemm res   M={M} K={K} N={N} -> pair rows", us, 2.0 * M * N * K)
