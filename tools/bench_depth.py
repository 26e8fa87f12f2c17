"""Throughput of the depth network on one GPU, with the oracle timed on the host cores beside it:
    python tools/bench_depth.py [--encoder vitl] [--size 518] [--batch 4] [--steps 10] [--dtype bf16|f32] [--no-cpu-baseline] [--kernel-table FILE]
Prints one JSON line (depth maps / s; whole-network algorithmic FLOPs against the dense MFMA peak of the dtype)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def flops(cfg, H, W):
    """algorithmic FLOPs of one image: ViT blocks (qkv, attention, proj, mlp), patch embedding, DPT head convolutions"""
    C, S = cfg.embed_dim, 1 + (H // 14) * (W // 14)
    n = (H // 14) * (W // 14)
    f = 2.0 * n * 588 * C
    f += cfg.depth * (2.0 * S * C * 3 * C + 4.0 * S * S * C + 2.0 * S * C * C + 4.0 * S * C * cfg.mlp_ratio * C)
    oc, ft = cfg.out_channels, cfg.features
    ph, pw = H // 14, W // 14
    sizes = [(4 * ph, 4 * pw), (2 * ph, 2 * pw), (ph, pw), ((ph - 1) // 2 + 1, (pw - 1) // 2 + 1)]
    for i in range(4):
        f += 2.0 * n * C * oc[i]
    f += 2.0 * n * oc[0] * 16 * oc[0] + 2.0 * n * oc[1] * 4 * oc[1] + 2.0 * sizes[3][0] * sizes[3][1] * 9 * oc[3] * oc[3]
    for i in range(4):
        h, w = sizes[i]
        f += 2.0 * h * w * 9 * oc[i] * ft                                        # layer_rn
        f += (2 if i == 3 else 4) * 2.0 * h * w * 9 * ft * ft                    # ResidualConvUnits (refinenet4 runs one)
    outs = [sizes[2], sizes[1], sizes[0], (2 * sizes[0][0], 2 * sizes[0][1])]
    for h, w in outs:
        f += 2.0 * h * w * ft * ft                                               # out_conv at the resized resolution
    h1, w1 = outs[3]
    f += 2.0 * h1 * w1 * 9 * ft * (ft // 2) + 2.0 * H * W * 9 * (ft // 2) * 32 + 2.0 * H * W * 32
    return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--encoder", default="vitl")
    ap.add_argument("--size", type=int, default=518)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-table", default="", help="write the per-kernel event table (HIP events around every launch of one forward) to this file")
    a = ap.parse_args()
    from freefine_amd.depth import HipDepthAnything, depth_config, synthetic_state
    dev = torch.device("cuda:0")
    cfg = depth_config(a.encoder)
    st = synthetic_state(cfg, seed=1)
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    net = HipDepthAnything(cfg, st, dtype=dt, device=dev)
    x = torch.randn(a.batch, 3, a.size, a.size, generator=torch.Generator().manual_seed(0)).to(dev)
    for _ in range(2):
        net(x)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.steps):
        d = net(x)
    torch.cuda.synchronize()
    dt_s = (time.time() - t0) / a.steps
    fl = flops(cfg, a.size, a.size)
    peak = 2500.0 if a.dtype == "bf16" else 157.3
    line = {"metric": f"depth maps/sec/GPU @{a.size}px DepthAnything-{a.encoder}", "value": round(a.batch / dt_s, 2), "unit": "images/s",
            "ms_per_batch": round(dt_s * 1e3, 2), "batch": a.batch, "dtype": a.dtype, "data": "synthetic (seeded random weights)",
            "algorithmic_gflop_per_image": round(fl / 1e9, 1), "tflops": round(fl * a.batch / dt_s / 1e12, 1),
            "frac_of_mfma_peak": round(fl * a.batch / dt_s / 1e12 / peak, 4)}
    if a.kernel_table:
        from freefine_amd import ops
        ops.profile_begin()
        net(x)
        rec = ops.profile_end()
        with open(a.kernel_table, "w") as f:
            f.write("kernel\tcalls\ttotal_ms\talgorithmic_TFLOP/s\talgorithmic_GB/s\n")
            for k, v in sorted(rec.items(), key=lambda kv: -kv[1]["total_ms"]):
                ms = max(v["total_ms"], 1e-9)
                f.write(f"{k}\t{v['calls']}\t{v['total_ms']:.3f}\t{v['flops'] / ms / 1e9:.1f}\t{v['bytes'] / ms / 1e6:.1f}\n")
    if not a.no_cpu_baseline:
        from oracle import dpt as OD                  # the CPU baseline leg only: the oracle is never on the GPU path
        torch.set_num_threads(max(8, min(32, torch.get_num_threads())))
        xc = x[:1].cpu()
        t0 = time.time()
        OD.depth_forward(OD.dpt_config(a.encoder), st, xc)
        tc = time.time() - t0
        line["cpu_baseline"] = {"value": round(1.0 / tc, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                                "sample": f"one {a.size} x {a.size} image through oracle/dpt.py (torch fp32), {tc:.1f} s"}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
