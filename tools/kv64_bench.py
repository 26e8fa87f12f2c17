"""Round 6: cost of writing the pre-split K / V^T images in the projection epilogues (FFN_IG_OUT_KV64) against the fp32 projections + ffn_attn_presplit.
    python tools/kv64_bench.py [rows]      (rows of the batched UNet forward: 72 = guided pass of 24 images, 48 = inversion)"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freefine_amd import _lib, ops


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 72
    dev = torch.device("cuda:0")
    lib = _lib.load()
    for S, heads in ((4096, 5), (1024, 10), (256, 20)):
        C = heads * 64
        g = torch.Generator().manual_seed(S)
        y = torch.randn(rows, S, C, generator=g).to(dev)
        ya = ops.layernorm(y, torch.ones(C, device=dev), torch.zeros(C, device=dev), pair=True)
        wqk = ops.pack_linear((torch.randn(2 * C, C, generator=g) * C ** -0.5).to(dev), torch.float32, x3=True)
        wv = ops.pack_linear((torch.randn(C, C, generator=g) * C ** -0.5).to(dev), torch.float32, x3=True)
        qk = torch.empty(rows, S, 2 * C, device=dev)
        vt = torch.empty(rows, C, S, device=dev)
        kp = torch.empty(rows, S, 2 * C, dtype=torch.bfloat16, device=dev)
        vp = torch.empty(rows, C, 2 * S, dtype=torch.bfloat16, device=dev)
        t = {}
        t["qk fp32"] = timeit(lambda: ops.linear(ya, wqk, None, K=C, out=qk))
        t["qk kv64"] = timeit(lambda: ops.linear(ya, wqk, None, K=C, out=qk, kv64_from=C))
        t["vt fp32"] = timeit(lambda: ops.linear(ya, wv, None, K=C, rows_per_batch=S, transposed_ld=S, out=vt))
        t["vt kv64"] = timeit(lambda: ops.linear(ya, wv, None, K=C, rows_per_batch=S, transposed_ld=S, out=vt, kv64_from=0))
        t["presplit"] = timeit(lambda: lib.ffn_attn_presplit(ops._stream(), qk[..., C:].data_ptr(), vt.data_ptr(), kp.data_ptr(), vp.data_ptr(), rows, S, heads, 2 * C, S))
        print(f"rows={rows} S={S} C={C}: " + "  ".join(f"{k} {v:.1f} us" for k, v in t.items()) +
              f"  | fp32+presplit {t['qk fp32'] + t['vt fp32'] + t['presplit']:.1f}  kv64 {t['qk kv64'] + t['vt kv64']:.1f}")


if __name__ == "__main__":
    main()
