"""Run ONE kernel shape repeatedly (for rocprofv3 --pmc passes).  python tools/one_kernel.py conv 64 320 320 | gemm M K N | geglu M K N | attn S C heads passes | xattn S C heads passes
ONE_B = batch rows (conv / attn); ONE_MODE = bf16 (default) | x3 (split-bf16: fp32 activations, FFN_BF16X3 weights / attn_x3_kernel); ONE_SPLITK = forced K slices (conv)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freefine_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
X3 = os.environ.get("ONE_MODE", "bf16") == "x3"
dt = torch.float32 if X3 else torch.bfloat16
g = torch.Generator().manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dt).to(dev)
kind = sys.argv[1]
B = int(os.environ.get("ONE_B", 4))
if kind == "conv":
    hw, cin, cout = map(int, sys.argv[2:5])
    x = rnd(B, hw * hw, cin)
    w = ops.pack_conv3x3(rnd(cout, cin, 3, 3, scale=(9 * cin) ** -0.5), dt, x3=X3)
    b = torch.zeros(cout, device=dev)
    xin = ops.split_pair(x, cin) if X3 else x            # (the UNet's norms hand the convolutions pair rows)
    SK = int(os.environ.get("ONE_SPLITK", 0))              # 0 = the library's choice (tuner), k = force k K-slices
    fn = lambda: ops.conv3x3(xin, w, b, B, hw, hw, cin, splitk=SK)
elif kind == "geglu":
    M, K, N = map(int, sys.argv[2:5])       # N = packed hidden | gate columns
    x = rnd(M, K)
    wp, bp = ops.pack_geglu(rnd(N, K, scale=K ** -0.5), torch.zeros(N, device=dev), dt, x3=X3)
    out = None if X3 else torch.empty(M, N // 2, dtype=dt, device=dev)
    xin = ops.split_pair(x, K) if X3 else x
    fn = lambda: ops.linear(xin, wp, bp, K=K, geglu=True, out=out, out_pair=X3)
elif kind == "gemm":
    M, K, N = map(int, sys.argv[2:5])
    x = rnd(M, K)
    w = ops.pack_linear(rnd(N, K, scale=K ** -0.5), dt, x3=X3)
    xin = ops.split_pair(x, K) if X3 else x
    fn = lambda: ops.linear(xin, w, None, K=K)
elif kind == "xattn":                       # the text cross-attention: 77 keys, V^T rows padded to 80; passes = 2: the two-pass local form with per-query weights
    S, C, heads, passes = map(int, sys.argv[2:6])
    q, k, vt = rnd(B, S, C), rnd(B, 77, C), rnd(B, C, 80)
    wq = torch.rand(S, generator=g).to(dev)
    P = None if passes == 1 else [[ops.AttnEntrySpec(b, b, 1.0, 0.0, wq=wq) for b in range(B)], [ops.AttnEntrySpec(b, b ^ 1, 1.0, 0.0, wq=wq) for b in range(B)]]
    fn = lambda: ops.attention(q, k, vt, heads, (C // heads) ** -0.5, P, Sk=77, x3=X3, out_pair=X3)
else:
    S, C, heads, passes = map(int, sys.argv[2:6])
    q, k, vt = rnd(B, S, C), rnd(B, S, C), rnd(B, C, S)
    km = (torch.rand(S, generator=g) > 0.7).to(torch.uint8).to(dev)
    qs = (torch.rand(S, generator=g) > 0.5).to(torch.uint8).to(dev)
    cg = torch.tensor([0.5], device=dev)
    P = None if passes == 1 else [[ops.AttnEntrySpec(b, b | 1, 0.0, 1.0, kmask=km, qsel=qs, flags=1) for b in range(B)],
                                  [ops.AttnEntrySpec(b, b, 1.0, -1.0) for b in range(B)]]
    fn = lambda: ops.attention(q, k, vt, heads, (C // heads) ** -0.5, P, w_dev=cg, x3=X3)
for _ in range(10):
    fn()
torch.cuda.synchronize()
if os.environ.get("ONE_TIME"):              # event-timed average over ONE_TIME launches (not under the profiler)
    n = int(os.environ["ONE_TIME"])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{' '.join(sys.argv[1:])} B={B} mode={'x3' if X3 else 'bf16'}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per call", flush=True)
