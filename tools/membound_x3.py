"""Achieved HBM bandwidth of the memory-bound kernels of the split-bf16 (headline) mode at the bench's row counts, each launch timed on cold caches
(a 512 MiB fill between repetitions): us per launch and GB/s of the bytes the launch ACTUALLY moves (reads + writes).   python tools/membound_x3.py [rows]"""
import ctypes as CT
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freefine_amd import _lib as L
from freefine_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 72
lib = L.load()
junk = torch.empty(128 * 2**20, device=dev)
st = lambda: CT.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, reps=7):
    fn()
    ts = []
    for _ in range(reps):
        junk.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


def report(name, us, nbytes):
    print(f"  {name:52s} {us:8.1f} us  {nbytes / us * 1e-3:7.0f} GB/s  ({nbytes / 1e6:.0f} MB moved)", flush=True)


for hw, C in ((64, 320), (64, 640), (32, 640), (32, 1280), (16, 1280)):
    S = hw * hw
    M = B * S
    print(f"{hw}x{hw} C={C}, {B} rows: M = {M}")
    x = torch.randn(B, S, C, device=dev)
    gam, bet = torch.randn(C, device=dev), torch.randn(C, device=dev)
    nb = x.numel() * 4
    if not lib.ffn_gn_fused(B, S, C, 32):
        part, sc, sh = ops.gn_workspace(B, S, C, dev)
        y = torch.empty(B, S, 2 * C, dtype=torch.bfloat16, device=dev)
        report("gn_partial + gn_finalize", timeit(lambda: L.check(lib.ffn_gn_stats(st(), L.FFN_F32, x.data_ptr(), gam.data_ptr(), bet.data_ptr(), B, S, C, 32, 1e-5,
                                                                                   part.data_ptr(), sc.data_ptr(), sh.data_ptr()))), nb)
        report("gn_apply (SiLU, pair out)", timeit(lambda: L.check(lib.ffn_gn_apply(st(), L.FFN_F32, x.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr(), B, S, C,
                                                                                    L.NORM_SILU | L.NORM_OUT_PAIR))), 2 * nb)
    report("groupnorm + SiLU -> pair (all launches)", timeit(lambda: ops.groupnorm(x, gam, bet, 32, 1e-5, silu=True, pair=True)), 3 * nb)
    report("layernorm -> pair", timeit(lambda: ops.layernorm(x, gam, bet, pair=True)), 2 * nb)
    report("split_pair", timeit(lambda: ops.split_pair(x, C)), 2 * nb)
    if C % 64 == 0 and S % 64 == 0:
        heads = C // 64
        kp = torch.empty(B, S, 2 * C, dtype=torch.bfloat16, device=dev)
        vp = torch.empty(B, C, 2 * S, dtype=torch.bfloat16, device=dev)
        vt = torch.randn(B, C, S, device=dev)
        report("attn_presplit (K and V^T)", timeit(lambda: L.check(lib.ffn_attn_presplit(st(), x.data_ptr(), vt.data_ptr(), kp.data_ptr(), vp.data_ptr(), B, S, heads, C, S))), 4 * nb)
    x2 = torch.randn(B, S, C, device=dev)
    o2 = torch.empty(B, S, 2 * C, device=dev)
    report("concat C|C (both halves copied)", timeit(lambda: ops.concat(x, x2, out=o2)), 4 * nb)
