"""Per-kernel numerics: every HIP kernel (called through the C ABI) against a plain torch fp32/fp64 statement of the
same op on identical seeded inputs.  Tolerances: fp32 mode <= 2e-5 of the output scale (exact-fp32 MFMA, different
summation order only); bf16 mode compares against the same op evaluated on bf16-rounded inputs, <= 1.5e-2."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def pair_value(p, C):
    """hi + lo of split-bf16 pair rows [..., 2C] (include/freefine_hip.h FFN_BF16X3: 128-byte blocks [hi(32) | lo(32)] when C % 32 == 0, else planes)"""
    if C % 32 == 0:
        b = p.double().reshape(*p.shape[:-1], C // 32, 2, 32)
        return (b[..., 0, :] + b[..., 1, :]).reshape(*p.shape[:-1], C)
    return p[..., :C].double() + p[..., C:].double()


def tol(dtype):
    return 2e-5 if dtype == torch.float32 else 1.5e-2


def rnd(shape, dtype, dev, g, scale=1.0):
    x = (torch.randn(*shape, generator=g) * scale).to(dtype)
    return x.to(dev)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(256, 320, 320), (4096, 1280, 320), (77 * 4, 640, 1024), (4, 1280, 320),
                                    (1000, 4, 320), (16384, 320, 1280), (130, 200, 72)])
def test_linear(gpu, dtype, M, N, K):
    from freefine_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    K = K // ops.epc(dtype) * ops.epc(dtype)
    x = rnd((M, K), dtype, gpu, g)
    w = rnd((N, K), dtype, gpu, g, K ** -0.5)
    b = torch.randn(N, generator=g).to(gpu)
    res = rnd((M, N), dtype, gpu, g)
    wp = ops.pack_linear(w, dtype)
    ref = x.double() @ w.double().t() + b.double()
    out = ops.linear(x, wp, b, K=K)
    assert relerr(out, ref) < tol(dtype)
    out = ops.linear(x, wp, b, K=K, residual=res)
    assert relerr(out, ref + res.double()) < tol(dtype)
    out = ops.linear(x, wp, b, K=K, silu=True)
    assert relerr(out, F.silu(ref)) < tol(dtype)
    out = ops.linear(x, wp, None, K=K, out_f32=True)
    assert out.dtype == torch.float32
    assert relerr(out, ref - b.double()) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_rowbias_and_geglu(gpu, dtype):
    from freefine_amd import ops
    g = torch.Generator().manual_seed(3)
    B, S, K, Fh = 2, 192, 320, 1280
    x = rnd((B, S, K), dtype, gpu, g)
    w = rnd((2 * Fh, K), dtype, gpu, g, K ** -0.5)
    b = torch.randn(2 * Fh, generator=g).to(gpu)
    wp, bp = ops.pack_geglu(w, b, dtype)
    out = ops.linear(x, wp, bp, geglu=True)
    y = x.double() @ w.double().t() + b.double()
    ref = y[..., :Fh] * F.gelu(y[..., Fh:])
    assert out.shape == (B, S, Fh)
    assert relerr(out, ref) < tol(dtype)
    # row bias (time-embedding projection added per batch row)
    w2 = rnd((640, K), dtype, gpu, g, K ** -0.5)
    rb = torch.randn(B, 640, generator=g).to(gpu)
    out = ops.linear(x, ops.pack_linear(w2, dtype), None, rowbias=rb, rows_per_batch=S)
    ref = x.double() @ w2.double().t() + rb.double()[:, None, :]
    assert relerr(out, ref) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("S", [256, 77, 400])
def test_linear_transposed(gpu, dtype, S):
    """S = 256 / 400 (bf16): the ping-pong kernel's transposed-output form (400: ragged last tile, 5 images, bias); 77: 2-stage kernel"""
    from freefine_amd import ops
    g = torch.Generator().manual_seed(5)
    B, K, N = (5, 320, 640) if S == 400 else (3, 320, 320)
    x = rnd((B, S, K), dtype, gpu, g)
    w = rnd((N, K), dtype, gpu, g, K ** -0.5)
    b = rnd((N,), torch.float32, gpu, g) if S == 400 else None
    ld = (S + 7) // 8 * 8
    out = ops.linear(x, ops.pack_linear(w, dtype), b, rows_per_batch=S, transposed_ld=ld)
    ref = (x.double() @ w.double().t() + (b.double() if b is not None else 0.0)).transpose(1, 2)
    assert out.shape == (B, N, ld)
    assert relerr(out[:, :, :S], ref) < tol(dtype)
    assert (out[:, :, S:] == 0).all()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [
    dict(B=2, H=16, W=16, Cin=64, Cout=96, stride=1, pad=1, up=False),
    dict(B=2, H=16, W=16, Cin=64, Cout=64, stride=2, pad=1, up=False),
    dict(B=1, H=8, W=8, Cin=128, Cout=64, stride=1, pad=1, up=True),
    dict(B=4, H=8, W=8, Cin=320, Cout=4, stride=1, pad=1, up=False),
    dict(B=2, H=12, W=20, Cin=8, Cout=320, stride=1, pad=1, up=False),
    dict(B=1, H=16, W=16, Cin=32, Cout=32, stride=2, pad=0, up=False),  # VAE encoder: pad (0,1,0,1) then stride 2
])
def test_conv3x3(gpu, dtype, cfg):
    from freefine_amd import ops
    g = torch.Generator().manual_seed(11)
    B, H, W, Cin, Cout = cfg["B"], cfg["H"], cfg["W"], cfg["Cin"], cfg["Cout"]
    x = rnd((B, Cin, H, W), dtype, gpu, g)
    w = rnd((Cout, Cin, 3, 3), dtype, gpu, g, (9 * Cin) ** -0.5)
    b = torch.randn(Cout, generator=g).to(gpu)
    xin = x.double()
    if cfg["up"]:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    if cfg["pad"] == 0:
        xin = F.pad(xin, (0, 1, 0, 1))
        ref = F.conv2d(xin, w.double(), b.double(), stride=cfg["stride"], padding=0)
    else:
        ref = F.conv2d(xin, w.double(), b.double(), stride=cfg["stride"], padding=1)
    Ho, Wo = ref.shape[-2:]
    x_nhwc = x.permute(0, 2, 3, 1).reshape(B, H * W, Cin).contiguous()
    out = ops.conv3x3(x_nhwc, ops.pack_conv3x3(w, dtype), b, B, H, W, Cin, stride=cfg["stride"], pad=cfg["pad"],
                      upsample=cfg["up"], Hout=Ho, Wout=Wo)
    ref_nhwc = ref.permute(0, 2, 3, 1).reshape(B, Ho * Wo, Cout)
    assert relerr(out, ref_nhwc) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv3x3_epilogue(gpu, dtype):
    from freefine_amd import ops
    g = torch.Generator().manual_seed(12)
    B, H, W, Cin, Cout = 2, 16, 16, 64, 128
    x = rnd((B, H * W, Cin), dtype, gpu, g)
    w = rnd((Cout, Cin, 3, 3), dtype, gpu, g, (9 * Cin) ** -0.5)
    b = torch.randn(Cout, generator=g).to(gpu)
    rb = torch.randn(B, Cout, generator=g).to(gpu)
    res = rnd((B, H * W, Cout), dtype, gpu, g)
    ref = F.conv2d(x.double().reshape(B, H, W, Cin).permute(0, 3, 1, 2), w.double(), b.double(), padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(B, H * W, Cout) + rb.double()[:, None] + res.double()
    out = ops.conv3x3(x, ops.pack_conv3x3(w, dtype), b, B, H, W, Cin, rowbias=rb, residual=res)
    assert relerr(out, ref) < tol(dtype)


def ref_attention(q, k, v, heads, scale, allowed=None):
    """q [S,C], k,v [Sk,C] double; allowed [heads,S,Sk] bool or None.  Empty rows -> uniform (reference behaviour)."""
    S, Cc = q.shape
    d = Cc // heads
    out = torch.zeros(S, Cc, dtype=torch.float64)
    for h in range(heads):
        sl = slice(h * d, (h + 1) * d)
        s = scale * q[:, sl] @ k[:, sl].t()
        if allowed is not None:
            a = allowed[h]
            s = torch.where(a, s, torch.full_like(s, -1e300))
            empty = ~a.any(dim=1)
            s[empty] = 0.0
        out[:, sl] = torch.softmax(s, dim=-1) @ v[:, sl]
    return out


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("S,Sk,heads,D", [(256, 256, 5, 64), (64, 77, 8, 40), (200, 130, 2, 80), (128, 64, 2, 160)])
def test_attention_plain(gpu, dtype, S, Sk, heads, D):
    from freefine_amd import ops
    g = torch.Generator().manual_seed(S + D)
    B, Cc = 2, heads * D
    q = rnd((B, S, Cc), dtype, gpu, g)
    k = rnd((B, Sk, Cc), dtype, gpu, g)
    v = rnd((B, Sk, Cc), dtype, gpu, g)
    vt = ops.transpose(v, ld_dst=(Sk + 7) // 8 * 8)
    scale = D ** -0.5
    out = ops.attention(q, k, vt, heads, scale, Sk=Sk)
    for b in range(B):
        ref = ref_attention(q[b].double().cpu(), k[b].double().cpu(), v[b].double().cpu(), heads, scale)
        assert relerr(out[b], ref) < tol(dtype)


@pytest.mark.parametrize("S,Sk,heads,ldvt", [(4096, 77, 5, 80), (1024, 77, 10, 80), (100, 77, 2, 80), (256, 20, 3, 24), (72, 96, 1, 96), (40, 81, 2, 88)])
def test_cross_attention_short_keys(gpu, S, Sk, heads, ldvt):
    """xattn_kernel (bf16, d = 64, Sk <= 96, one unmasked pass): K / V^T held in registers per wave, query blocks of 32 streamed.
    Row maps (q_row / kv_row, as in row-deduplicated CFG batches), a device-scalar weight, ragged S, Sk below / at / above the fragment
    counts the kernel is instantiated for (2, 5, 6 fragments of 16 keys), V^T padding up to ldvt filled with huge finite values."""
    import ctypes
    from freefine_amd import _lib, ops
    g = torch.Generator().manual_seed(S + Sk)
    dtype, D = torch.bfloat16, 64
    Bq, Bk, Cc = 3, 2, heads * D
    q = rnd((Bq, S, Cc), dtype, gpu, g)
    k = rnd((Bk, Sk, Cc), dtype, gpu, g)
    v = rnd((Bk, Sk, Cc), dtype, gpu, g)
    vt = torch.full((Bk, Cc, ldvt), 3.0e38, dtype=dtype, device=gpu)
    vt[:, :, :Sk] = v.transpose(1, 2)
    scale = D ** -0.5
    cg = torch.tensor([0.25], device=gpu)
    rows = [ops.AttnEntrySpec(2, 1, 1.0, 0.0), ops.AttnEntrySpec(0, 0, 0.5, 2.0), ops.AttnEntrySpec(1, 1, 1.0, 0.0), ops.AttnEntrySpec(2, 0, 1.0, 0.0)]
    out = ops.attention(q, k, vt, heads, scale, [rows], Sk=Sk, w_dev=cg)
    d = _lib.AttnDesc()
    d.Bo, d.S, d.Sk, d.heads, d.D, d.npass, d.ldo, d.w_dev = 4, S, Sk, heads, D, 1, Cc, cg.data_ptr()
    for b, sp in enumerate(rows):
        d.e[b].q_row, d.e[b].kv_row, d.e[b].w_const, d.e[b].w_slope = sp.q_row, sp.kv_row, sp.w_const, sp.w_slope
    name = ctypes.create_string_buffer(160)
    _lib.load().ffn_attn_kernel_name(1, ctypes.byref(d), name, 160)
    assert b"xattn_kernel" in name.value, name.value
    for b, sp in enumerate(rows):
        wgt = sp.w_const + sp.w_slope * 0.25
        ref = wgt * ref_attention(q[sp.q_row].double().cpu(), k[sp.kv_row].double().cpu(), v[sp.kv_row].double().cpu(), heads, scale)
        assert relerr(out[b], ref) < tol(dtype), (b, relerr(out[b], ref))


@pytest.mark.parametrize("S,Sk,heads", [(4096, 77, 5), (1024, 77, 10), (100, 77, 2), (64, 30, 1)])
def test_cross_attention_short_keys_multi_pass(gpu, S, Sk, heads):
    """xattn_mp_kernel: the guided pass's local cross-attention (two passes, per-query blend weights, skipped entries, a row that no
    pass contributes to) against the fp64 statement; K / V^T fragment images of every pass in LDS."""
    import ctypes
    from freefine_amd import _lib, ops
    g = torch.Generator().manual_seed(7 * S + Sk)
    dtype, D = torch.bfloat16, 64
    B, Cc = 4, heads * D
    q = rnd((B, S, Cc), dtype, gpu, g)
    k = rnd((B, Sk, Cc), dtype, gpu, g)
    v = rnd((B, Sk, Cc), dtype, gpu, g)
    vt = ops.transpose(v, ld_dst=(Sk + 7) // 8 * 8)
    scale = D ** -0.5
    f = torch.rand(S, generator=g).to(gpu)
    omf = 1.0 - f
    p0 = [ops.AttnEntrySpec(0, 0), ops.AttnEntrySpec(1, 1), ops.AttnEntrySpec(2, 2, wq=f), None, ops.AttnEntrySpec(1, 3, 0.5, 0.0)]
    p1 = [None, None, ops.AttnEntrySpec(0, 0, wq=omf), None, ops.AttnEntrySpec(3, 2, 0.25, 0.0, wq=f)]
    out = ops.attention(q, k, vt, heads, scale, [p0, p1], Sk=Sk)
    d = _lib.AttnDesc()
    d.Bo, d.S, d.Sk, d.heads, d.D, d.npass, d.ldo = 5, S, Sk, heads, D, 2, Cc
    for pi, rows in enumerate((p0, p1)):
        for b, sp in enumerate(rows):
            e = d.e[pi * _lib.ATT_MAXB + b]
            if sp is not None:
                e.q_row, e.kv_row, e.w_const, e.w_slope = sp.q_row, sp.kv_row, sp.w_const, sp.w_slope
                e.wq = 0 if sp.wq is None else sp.wq.data_ptr()
    name = ctypes.create_string_buffer(160)
    _lib.load().ffn_attn_kernel_name(1, ctypes.byref(d), name, 160)
    assert b"xattn_mp_kernel" in name.value, name.value
    A = lambda qi, ki: ref_attention(q[qi].double().cpu(), k[ki].double().cpu(), v[ki].double().cpu(), heads, scale)
    fd, od = f.double().cpu()[:, None], omf.double().cpu()[:, None]
    refs = [A(0, 0), A(1, 1), fd * A(2, 2) + od * A(0, 0), torch.zeros(S, Cc, dtype=torch.float64), 0.5 * A(1, 3) + 0.25 * fd * A(3, 2)]
    scale_ref = max(r.abs().max().item() for r in refs)
    for b, ref in enumerate(refs):
        err = (out[b].double().cpu() - ref).abs().max().item() / scale_ref
        assert err < tol(dtype), (b, err)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("S,heads,D", [(192, 5, 64), (200, 8, 40), (144, 4, 80), (136, 2, 160)])
def test_attention_tca_edit(gpu, dtype, S, heads, D):
    """TCA edit branch: K/V from the reference rows, per-key source mask, per-query selector, tiled-head rule,
    context-guidance blend with the self pass (weights from a device scalar); SD-2.1 (d=64) and SD-1.5 (d=40/80/160) head sizes,
    ragged S (mask-on-MFMA tile with out-of-range keys)."""
    from freefine_amd import ops
    from freefine_amd._lib import ATT_HEAD_RULE
    g = torch.Generator().manual_seed(21)
    B = 4
    Cc = heads * D
    q = rnd((B, S, Cc), dtype, gpu, g)
    k = rnd((B, S, Cc), dtype, gpu, g)
    v = rnd((B, S, Cc), dtype, gpu, g)
    vt = ops.transpose(v)
    src = (torch.rand(S, generator=g) > 0.6).to(torch.uint8)
    tgt = (torch.rand(S, generator=g) > 0.5).to(torch.uint8)
    cg = 0.35
    cg_dev = torch.tensor([cg], dtype=torch.float32, device=gpu)
    scale = D ** -0.5
    ref_rows = [1, 1, 3, 3]
    p_ref = [ops.AttnEntrySpec(b, ref_rows[b], 0.0, 1.0, kmask=src.to(gpu), qsel=tgt.to(gpu), flags=ATT_HEAD_RULE) for b in range(B)]
    p_self = [ops.AttnEntrySpec(b, b, 1.0, -1.0) for b in range(B)]
    out = ops.attention(q, k, vt, heads, scale, [p_ref, p_self], w_dev=cg_dev)
    qc, kc, vc = q.double().cpu(), k.double().cpu(), v.double().cpu()
    for b in range(B):
        allowed = torch.ones(heads, S, S, dtype=torch.bool)
        for h in range(heads):
            if (b * heads + h) % 2 == 0:
                allowed[h] = (src[None, :] != 0) == (tgt[:, None] != 0)
        r = ref_attention(qc[b], kc[ref_rows[b]], vc[ref_rows[b]], heads, scale, allowed)
        s = ref_attention(qc[b], kc[b], vc[b], heads, scale)
        assert relerr(out[b], cg * r + (1 - cg) * s) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_image_batched_rows(gpu, dtype):
    """image-batched launch: 5 images x 4 rows = 20 output rows (> FFN_ATT_MAXB -> issued as row ranges); every image uses its
    own masks, reference rows offset by the image base, and the tiled-head rule pinned to the row index INSIDE the image."""
    from freefine_amd import ops
    from freefine_amd._lib import ATT_HEAD_RULE, ATT_MAXB
    g = torch.Generator().manual_seed(77)
    K, Bp, S, heads, D = 5, 4, 96, 3, 64
    assert K * Bp > ATT_MAXB
    Cc = heads * D
    q = rnd((K * Bp, S, Cc), dtype, gpu, g)
    k = rnd((K * Bp, S, Cc), dtype, gpu, g)
    v = rnd((K * Bp, S, Cc), dtype, gpu, g)
    vt = ops.transpose(v)
    cg = 0.6
    cg_dev = torch.tensor([cg], dtype=torch.float32, device=gpu)
    scale = D ** -0.5
    ref_rows = [1, 1, 3, 3]
    srcs = [(torch.rand(S, generator=g) > 0.5).to(torch.uint8) for _ in range(K)]
    tgts = [(torch.rand(S, generator=g) > 0.5).to(torch.uint8) for _ in range(K)]
    p_ref, p_self = [], []
    for i in range(K):
        sg, tg = srcs[i].to(gpu), tgts[i].to(gpu)
        for b in range(Bp):
            p_ref.append(ops.AttnEntrySpec(b, ref_rows[b], 0.0, 1.0, kmask=sg, qsel=tg, flags=ATT_HEAD_RULE).shifted(i * Bp, b))
            p_self.append(ops.AttnEntrySpec(b, b, 1.0, -1.0).shifted(i * Bp, b))
    out = ops.attention(q, k, vt, heads, scale, [p_ref, p_self], w_dev=cg_dev)
    qc, kc, vc = q.double().cpu(), k.double().cpu(), v.double().cpu()
    for i in range(K):
        for b in range(Bp):
            allowed = torch.ones(heads, S, S, dtype=torch.bool)
            for h in range(heads):
                if (b * heads + h) % 2 == 0:
                    allowed[h] = (srcs[i][None, :] != 0) == (tgts[i][:, None] != 0)
            r = ref_attention(qc[i * Bp + b], kc[i * Bp + ref_rows[b]], vc[i * Bp + ref_rows[b]], heads, scale, allowed)
            s_ = ref_attention(qc[i * Bp + b], kc[i * Bp + b], vc[i * Bp + b], heads, scale)
            assert relerr(out[i * Bp + b], cg * r + (1 - cg) * s_) < tol(dtype), (i, b)
    # plain attention over more rows than one launch takes
    out = ops.attention(q, k, vt, heads, scale)
    for b in (0, ATT_MAXB - 1, ATT_MAXB, K * Bp - 1):
        assert relerr(out[b], ref_attention(qc[b], kc[b], vc[b], heads, scale)) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_uniform_and_wq(gpu, dtype):
    """empty allowed set -> uniform softmax over all keys; per-query weights; skipped entries; q_row remap."""
    from freefine_amd import ops
    from freefine_amd._lib import ATT_UNIFORM_SEL1
    g = torch.Generator().manual_seed(22)
    B, S, Sk, heads, D = 3, 100, 77, 8, 40
    Cc = heads * D
    q = rnd((B, S, Cc), dtype, gpu, g)
    k = rnd((B, Sk, Cc), dtype, gpu, g)
    v = rnd((B, Sk, Cc), dtype, gpu, g)
    vt = ops.transpose(v, ld_dst=80)
    scale = D ** -0.5
    km = torch.zeros(Sk, dtype=torch.uint8, device=gpu)  # no key has mask==1 -> sel=1 queries see a uniform softmax
    wq = torch.rand(S, generator=g).to(gpu)
    p0 = [ops.AttnEntrySpec(0, 0, 1.0, 0.0, kmask=km, flags=ATT_UNIFORM_SEL1), ops.AttnEntrySpec(1, 1), ops.AttnEntrySpec(2, 2, wq=wq)]
    p1 = [None, None, ops.AttnEntrySpec(0, 1, 0.5)]
    out = ops.attention(q, k, vt, heads, scale, [p0, p1], Sk=Sk)
    qc, kc, vc = q.double().cpu(), k.double().cpu(), v.double().cpu()
    ref0 = vc[0].mean(dim=0, keepdim=True).expand(S, Cc)
    assert relerr(out[0], ref0) < tol(dtype)
    assert relerr(out[1], ref_attention(qc[1], kc[1], vc[1], heads, scale)) < tol(dtype)
    ref2 = wq.double().cpu()[:, None] * ref_attention(qc[2], kc[2], vc[2], heads, scale) + 0.5 * ref_attention(qc[0], kc[1], vc[1], heads, scale)
    assert relerr(out[2], ref2) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,HW,Cc", [(2, 256, 320), (1, 64, 1280), (2, 1024, 960), (1, 4096, 128), (2, 64, 2560)])
def test_groupnorm(gpu, dtype, B, HW, Cc):
    from freefine_amd import ops
    g = torch.Generator().manual_seed(HW + Cc)
    x = rnd((B, HW, Cc), dtype, gpu, g, 2.0) + 0.5
    gamma = torch.randn(Cc, generator=g).to(gpu)
    beta = torch.randn(Cc, generator=g).to(gpu)
    for silu in (False, True):
        out = ops.groupnorm(x, gamma, beta, 32, 1e-5, silu=silu)
        ref = F.group_norm(x.double().transpose(1, 2), 32, gamma.double(), beta.double(), 1e-5).transpose(1, 2)
        if silu:
            ref = F.silu(ref)
        assert relerr(out, ref) < (1e-5 if dtype == torch.float32 else 1.5e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,Cc", [(300, 320), (77, 1024), (64, 1280), (10, 640)])
def test_layernorm(gpu, dtype, M, Cc):
    from freefine_amd import ops
    g = torch.Generator().manual_seed(M + Cc)
    x = rnd((M, Cc), dtype, gpu, g, 3.0) + 1.0
    gamma = torch.randn(Cc, generator=g).to(gpu)
    beta = torch.randn(Cc, generator=g).to(gpu)
    out = ops.layernorm(x, gamma, beta, 1e-5)
    ref = F.layer_norm(x.double(), (Cc,), gamma.double(), beta.double(), 1e-5)
    assert relerr(out, ref) < (1e-5 if dtype == torch.float32 else 1.5e-2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_softmax_concat_transpose_cast(gpu, dtype):
    from freefine_amd import ops
    g = torch.Generator().manual_seed(1)
    x = rnd((50, 1000), dtype, gpu, g, 3.0)
    assert relerr(ops.softmax_rows(x, 0.7), torch.softmax(x.double() * 0.7, -1)) < tol(dtype)
    a, b = rnd((3, 70, 64), dtype, gpu, g), rnd((3, 70, 128), dtype, gpu, g)
    assert torch.equal(ops.concat(a, b), torch.cat([a, b], -1))
    t = ops.transpose(a, ld_dst=72)
    assert torch.equal(t[:, :, :70], a.transpose(1, 2))
    other = torch.bfloat16 if dtype == torch.float32 else torch.float32
    assert torch.equal(ops.cast(a, other), a.to(other))


def test_scheduler_elementwise(gpu):
    from freefine_amd import ops
    g = torch.Generator().manual_seed(9)
    eu, ec = torch.randn(2, 4, 64, 64, generator=g).to(gpu), torch.randn(2, 4, 64, 64, generator=g).to(gpu)
    mask = (torch.rand(64, 64, generator=g) > 0.5).float().to(gpu)
    out = ops.cfg_masked(eu, ec, mask.reshape(-1), 7.5)
    ref = eu + 7.5 * (ec - eu) * mask
    assert relerr(out, ref) < 1e-6
    out = ops.cfg_masked(eu, ec, None, 7.5)
    assert relerr(out, eu + 7.5 * (ec - eu)) < 1e-6
    x = torch.randn(2, 4, 64, 64, generator=g).to(gpu)
    xn, p0 = ops.ddim_inv_step(eu, x, 0.3, 0.95, 0.9, 0.43, want_pred_x0=True)
    rp0 = (x - 0.3 * eu) / 0.95
    assert relerr(p0, rp0) < 1e-6 and relerr(xn, 0.9 * rp0 + 0.43 * eu) < 1e-6
    noise = torch.randn(2, 4, 64, 64, generator=g).to(gpu)
    m = mask.reshape(-1)
    om = (1 - mask).reshape(-1) * 255.0  # stands in for the uint8 wrap values: the kernel must use om as given
    xp, _ = ops.ddim_ctrl_step(eu, x, noise, m, om, 0.3, 0.95, 0.97, 0.24, [0.2, 0.24], [0.13, 0.0], [1, 0])
    rp0 = (x - 0.3 * eu) / 0.95
    mm = torch.stack([mask, torch.ones_like(mask)])[:, None]
    oo = torch.stack([(1 - mask) * 255.0, torch.zeros_like(mask)])[:, None]
    cd = torch.tensor([0.2, 0.24], device=gpu)[:, None, None, None]
    sd = torch.tensor([0.13, 0.0], device=gpu)[:, None, None, None]
    ref = 0.97 * rp0 + (0.24 * eu * oo + cd * eu * mm) + sd * noise * mm
    assert relerr(xp, ref) < 1e-6


@pytest.mark.parametrize("dtype", DTYPES)
def test_pack_unpack_temb(gpu, dtype):
    from freefine_amd import ops
    g = torch.Generator().manual_seed(2)
    lat = torch.randn(2, 4, 16, 16, generator=g).to(gpu)
    p = ops.pack_nchw(lat, [0, 1, 0, 1], 8, dtype)
    ref = torch.cat([lat, lat]).permute(0, 2, 3, 1).reshape(4, 256, 4)
    assert relerr(p[..., :4], ref) < (1e-7 if dtype == torch.float32 else 5e-3)
    assert (p[..., 4:] == 0).all()
    e = torch.randn(3, 256, 4, generator=g).to(gpu)
    assert torch.equal(ops.nhwc_to_nchw_f32(e, 4, 16, 16), e.permute(0, 2, 1).reshape(3, 4, 16, 16))
    freq = ops.timestep_freqs(320, gpu)
    t = torch.tensor([981.0], device=gpu)
    emb = ops.timestep_embed(t, freq, 2, dtype)
    arg = 981.0 * freq.double()
    ref = torch.cat([arg.cos(), arg.sin()])[None].expand(2, -1)
    assert relerr(emb, ref) < (2e-4 if dtype == torch.float32 else 1e-2)  # fp32 argument t*freq ~ 1e3 -> 6e-5 abs
    img = torch.randint(0, 256, (1, 8, 8, 3), generator=g, dtype=torch.uint8).to(gpu)
    x = ops.image_to_nhwc(img, 8, dtype)
    assert relerr(x[..., :3], img.reshape(1, 64, 3).float() / 127.5 - 1) < (1e-6 if dtype == torch.float32 else 5e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("splitk", [0, 1, 3, 7])
def test_splitk_conv_and_linear(gpu, dtype, splitk):
    """split-K (fp32 partial slabs + reduce kernel with the full epilogue) must agree with the unsplit result."""
    from freefine_amd import ops
    g = torch.Generator().manual_seed(31)
    B, H, W, Cin, Cout = 2, 8, 8, 256, 192
    x = rnd((B, H * W, Cin), dtype, gpu, g)
    w = rnd((Cout, Cin, 3, 3), dtype, gpu, g, (9 * Cin) ** -0.5)
    b = torch.randn(Cout, generator=g).to(gpu)
    rb = torch.randn(B, Cout, generator=g).to(gpu)
    res = rnd((B, H * W, Cout), dtype, gpu, g)
    ref = F.conv2d(x.double().reshape(B, H, W, Cin).permute(0, 3, 1, 2), w.double(), b.double(), padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(B, H * W, Cout) + rb.double()[:, None] + res.double()
    out = ops.conv3x3(x, ops.pack_conv3x3(w, dtype), b, B, H, W, Cin, rowbias=rb, residual=res, splitk=splitk)
    assert relerr(out, ref) < tol(dtype)
    M, K, N = 200, 2048, 320
    a = rnd((M, K), dtype, gpu, g)
    wl = rnd((N, K), dtype, gpu, g, K ** -0.5)
    out = ops.linear(a, ops.pack_linear(wl, dtype), b[:N] if N <= Cout else None, silu=True, splitk=splitk)
    assert relerr(out, F.silu(a.double() @ wl.double().t())) < tol(dtype)
    out = ops.linear(a, ops.pack_linear(wl, dtype), None, out_f32=True, splitk=splitk)
    assert out.dtype == torch.float32 and relerr(out, a.double() @ wl.double().t()) < tol(dtype)


def test_kernels_are_deterministic(gpu):
    """Race detector for the software-pipelined kernels (counted waits, LDS-DMA in flight across barriers, epilogue stores named in the
    waits): their arithmetic order is fixed, so repeated launches on the same operands must agree bit for bit -- on shapes with many
    tiles per workgroup, every epilogue option, both attention kernels and the cross-attention kernels."""
    from freefine_amd import ops
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(11)
    def same(fn, n=6):
        ref = fn().clone()
        torch.cuda.synchronize()
        for _ in range(n):
            assert torch.equal(fn(), ref)
    B, H, Cin, Cout = 24, 64, 320, 320
    x, w = rnd((B, H * H, Cin), dt, gpu, g), ops.pack_conv3x3(rnd((Cout, Cin, 3, 3), dt, gpu, g, (9 * Cin) ** -0.5), dt)
    b, rb, r = rnd((Cout,), torch.float32, gpu, g), rnd((B, Cout), torch.float32, gpu, g), rnd((B, H * H, Cout), dt, gpu, g)
    same(lambda: ops.conv3x3(x, w, b, B, H, H, Cin, rowbias=rb))
    same(lambda: ops.conv3x3(x, w, b, B, H, H, Cin, residual=r))
    xm = x.reshape(B * H * H, Cin)
    wl, w4 = ops.pack_linear(rnd((Cout, Cin), dt, gpu, g, Cin ** -0.5), dt), ops.pack_linear(rnd((Cout, 4 * Cin), dt, gpu, g, (4 * Cin) ** -0.5), dt)
    same(lambda: ops.linear(xm, wl, b))
    same(lambda: ops.linear(xm, wl, b, residual=r.reshape(-1, Cout)))
    wg, bg = ops.pack_geglu(rnd((8 * Cin, Cin), dt, gpu, g, Cin ** -0.5), rnd((8 * Cin,), torch.float32, gpu, g), dt)
    hgl = ops.linear(xm, wg, bg, K=Cin, geglu=True)
    same(lambda: ops.linear(xm, wg, bg, K=Cin, geglu=True), 3)
    same(lambda: ops.linear(hgl, w4, b, residual=r.reshape(-1, Cout)), 3)
    same(lambda: ops.linear(x, wl, None, rows_per_batch=H * H, transposed_ld=H * H), 3)
    S, heads = 4096, 5
    q, k, vt = rnd((4, S, Cin), dt, gpu, g), rnd((4, S, Cin), dt, gpu, g), rnd((4, Cin, S), dt, gpu, g)
    km = (torch.rand(S, generator=g) > 0.6).to(torch.uint8).to(gpu)
    qs = (torch.rand(S, generator=g) > 0.5).to(torch.uint8).to(gpu)
    cg = torch.tensor([0.4], device=gpu)
    tca = [[ops.AttnEntrySpec(i, i | 1, 0.0, 1.0, kmask=km, qsel=qs, flags=1) for i in range(4)], [ops.AttnEntrySpec(i, i, 1.0, -1.0) for i in range(4)]]
    same(lambda: ops.attention(q, k, vt, heads, 0.125), 3)
    same(lambda: ops.attention(q, k, vt, heads, 0.125, tca, w_dev=cg), 3)
    kt, vtt = rnd((4, 77, Cin), dt, gpu, g), rnd((4, Cin, 80), dt, gpu, g)
    fw = torch.rand(S, generator=g).to(gpu)
    loc = [[ops.AttnEntrySpec(0, 0), ops.AttnEntrySpec(1, 1), ops.AttnEntrySpec(2, 2, wq=fw), ops.AttnEntrySpec(1, 1)], [None, None, ops.AttnEntrySpec(0, 0, wq=1.0 - fw), None]]
    same(lambda: ops.attention(q, kt, vtt, heads, 0.125, Sk=77), 3)
    same(lambda: ops.attention(q, kt, vtt, heads, 0.125, loc, Sk=77), 3)


def test_every_bf16_igemm_configuration(gpu):
    """ffn_igemm picks (tile, K-split) per shape by timing; force EVERY bf16 configuration in turn (incl. the 256x256 / 128x320
    tiles, the persistent tile walk and the weight-stationary kernels) on shapes where it is valid and check the result."""
    from freefine_amd import _lib as L
    from freefine_amd import ops
    lib = L.load()
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    try:
        for cfg in range(lib.ffn_igemm_num_configs()):
            lib.ffn_igemm_force_config(cfg)
            # dense, K = 320 (weight panel fits the LDS), bias + residual; ragged M / N
            for (M, N, K) in [(16384 + 72, 320, 320), (20000, 640, 640), (300, 328, 136)]:
                x, w = rnd((M, K), dt, gpu, g), rnd((N, K), dt, gpu, g, K ** -0.5)
                b, r = rnd((N,), torch.float32, gpu, g), rnd((M, N), dt, gpu, g)
                out = ops.linear(x, ops.pack_linear(w, dt), b, K=K, residual=r)
                ref = x.double().cpu() @ w.double().cpu().t() + b.double().cpu() + r.double().cpu()
                assert relerr(out, ref) < tol(dt), (cfg, M, N, K)
            # GEGLU epilogue
            M, K, F = 16384, 320, 640
            x, w, b = rnd((M, K), dt, gpu, g), rnd((2 * F, K), dt, gpu, g, K ** -0.5), rnd((2 * F,), torch.float32, gpu, g)
            wp, bp = ops.pack_geglu(w, b, dt)
            out = ops.linear(x, wp, bp, K=K, geglu=True)
            y = x.double().cpu() @ w.double().cpu().t() + b.double().cpu()
            ref = y[:, :F] * torch.nn.functional.gelu(y[:, F:])
            assert relerr(out, ref) < tol(dt), (cfg, "geglu")
            # 3x3 conv with time-embedding row bias and residual
            B, H, Cin, Cout = 3, 24, 64, 320
            x, w = rnd((B, H * H, Cin), dt, gpu, g), rnd((Cout, Cin, 3, 3), dt, gpu, g, (9 * Cin) ** -0.5)
            b, rb, r = rnd((Cout,), torch.float32, gpu, g), rnd((B, Cout), torch.float32, gpu, g), rnd((B, H * H, Cout), dt, gpu, g)
            out = ops.conv3x3(x, ops.pack_conv3x3(w, dt), b, B, H, H, Cin, rowbias=rb, residual=r)
            xr = x.double().cpu().reshape(B, H, H, Cin).permute(0, 3, 1, 2)
            ref = torch.nn.functional.conv2d(xr, w.double().cpu(), b.double().cpu(), padding=1) + rb.double().cpu()[:, :, None, None]
            ref = ref.permute(0, 2, 3, 1).reshape(B, H * H, Cout) + r.double().cpu()
            assert relerr(out, ref) < tol(dt), (cfg, "conv")
            # shapes the ping-pong kernel accepts (256-row tiles, N % 320 == 0 or N % 256 == 0, rows per image % 128 == 0), with the
            # time-embedding row bias arriving through its LDS column vectors and the residual through its accumulator start values
            for (B, H, W, Cin, Cout) in [(3, 32, 32, 128, 320), (2, 16, 48, 64, 512), (5, 16, 16, 192, 640)]:
                x, w = rnd((B, H * W, Cin), dt, gpu, g), rnd((Cout, Cin, 3, 3), dt, gpu, g, (9 * Cin) ** -0.5)
                b, rb, r = rnd((Cout,), torch.float32, gpu, g), rnd((B, Cout), torch.float32, gpu, g), rnd((B, H * W, Cout), dt, gpu, g)
                out = ops.conv3x3(x, ops.pack_conv3x3(w, dt), b, B, H, W, Cin, rowbias=rb, residual=r)
                xr = x.double().cpu().reshape(B, H, W, Cin).permute(0, 3, 1, 2)
                ref = torch.nn.functional.conv2d(xr, w.double().cpu(), b.double().cpu(), padding=1) + rb.double().cpu()[:, :, None, None]
                ref = ref.permute(0, 2, 3, 1).reshape(B, H * W, Cout) + r.double().cpu()
                assert relerr(out, ref) < tol(dt), (cfg, "conv-pp-shapes", H, W, Cout)
            # caller-forced split-K (the ping-pong configurations then run their split form: fp32 slabs + the reduce kernel's epilogue)
            B, H, Cin, Cout = 3, 16, 256, 320
            x, w = rnd((B, H * H, Cin), dt, gpu, g), rnd((Cout, Cin, 3, 3), dt, gpu, g, (9 * Cin) ** -0.5)
            b, rb, r = rnd((Cout,), torch.float32, gpu, g), rnd((B, Cout), torch.float32, gpu, g), rnd((B, H * H, Cout), dt, gpu, g)
            out = ops.conv3x3(x, ops.pack_conv3x3(w, dt), b, B, H, H, Cin, rowbias=rb, residual=r, splitk=12)
            xr = x.double().cpu().reshape(B, H, H, Cin).permute(0, 3, 1, 2)
            ref = torch.nn.functional.conv2d(xr, w.double().cpu(), b.double().cpu(), padding=1) + rb.double().cpu()[:, :, None, None]
            ref = ref.permute(0, 2, 3, 1).reshape(B, H * H, Cout) + r.double().cpu()
            assert relerr(out, ref) < tol(dt), (cfg, "conv-splitk")
            M, N, K = 600, 512, 4096
            x, w, b = rnd((M, K), dt, gpu, g), rnd((N, K), dt, gpu, g, K ** -0.5), rnd((N,), torch.float32, gpu, g)
            out = ops.linear(x, ops.pack_linear(w, dt), b, K=K, silu=True, splitk=8)
            assert relerr(out, torch.nn.functional.silu(x.double().cpu() @ w.double().cpu().t() + b.double().cpu())) < tol(dt), (cfg, "dense-splitk")
            # ping-pong kernel: fused nearest-2x upsample, stride 2 (pad 1, and the VAE encoder's pad-right/bottom-only form)
            for (B, H, W, Cin, Cout, stride, pad, up) in [(2, 16, 16, 128, 320, 1, 1, True), (2, 32, 32, 64, 320, 2, 1, False), (3, 32, 32, 64, 256, 2, 0, False)]:
                x, w = rnd((B, Cin, H, W), dt, gpu, g), rnd((Cout, Cin, 3, 3), dt, gpu, g, (9 * Cin) ** -0.5)
                b = rnd((Cout,), torch.float32, gpu, g)
                xin = x.double().cpu()
                if up:
                    xin = torch.nn.functional.interpolate(xin, scale_factor=2.0, mode="nearest")
                if pad == 0:
                    xin = torch.nn.functional.pad(xin, (0, 1, 0, 1))
                ref = torch.nn.functional.conv2d(xin, w.double().cpu(), b.double().cpu(), stride=stride, padding=pad)
                Ho, Wo = ref.shape[-2:]
                out = ops.conv3x3(x.permute(0, 2, 3, 1).reshape(B, H * W, Cin).contiguous(), ops.pack_conv3x3(w, dt), b, B, H, W, Cin, stride=stride,
                                  pad=pad, upsample=up, Hout=Ho, Wout=Wo)
                assert relerr(out, ref.permute(0, 2, 3, 1).reshape(B, Ho * Wo, Cout)) < tol(dt), (cfg, "conv-pp-variants", stride, pad, up)
            # shapes the halo conv kernel accepts: tiles of whole image rows (32x32, 16x16) and pieces of one wide row (4x256)
            for (B, H, W, Cin, Cout) in [(2, 32, 32, 128, 320), (3, 16, 16, 192, 256), (1, 4, 256, 64, 128)]:
                x, w = rnd((B, H * W, Cin), dt, gpu, g), rnd((Cout, Cin, 3, 3), dt, gpu, g, (9 * Cin) ** -0.5)
                b, r = rnd((Cout,), torch.float32, gpu, g), rnd((B, H * W, Cout), dt, gpu, g)
                out = ops.conv3x3(x, ops.pack_conv3x3(w, dt), b, B, H, W, Cin, residual=r)
                xr = x.double().cpu().reshape(B, H, W, Cin).permute(0, 3, 1, 2)
                ref = torch.nn.functional.conv2d(xr, w.double().cpu(), b.double().cpu(), padding=1)
                ref = ref.permute(0, 2, 3, 1).reshape(B, H * W, Cout) + r.double().cpu()
                assert relerr(out, ref) < tol(dt), (cfg, "conv-halo-shapes", H, W)
    finally:
        lib.ffn_igemm_force_config(-1)


def _ref_attention_gpu(q, k, v, heads, scale, allowed=None):
    """ref_attention evaluated in fp64 on the GPU by plain torch (the production shapes: S = 4096 x 4096 scores per head)."""
    S, Cc = q.shape
    d = Cc // heads
    out = torch.zeros(S, Cc, dtype=torch.float64, device=q.device)
    for h in range(heads):
        sl = slice(h * d, (h + 1) * d)
        s = scale * q[:, sl] @ k[:, sl].t()
        if allowed is not None and allowed[h] is not None:
            a = allowed[h]
            s = torch.where(a, s, torch.full_like(s, -1e300))
            s[~a.any(dim=1)] = 0.0
        out[:, sl] = torch.softmax(s, dim=-1) @ v[:, sl]
    return out


def _production_masks(S, kind, g):
    """per-key source mask / per-query target mask at a production sequence length: 'rect' = rectangles on the sqrt(S) grid (whole
    64-key tiles masked or unmasked, the shape GeoBench masks have), 'rand' = random bytes (every tile mixed)."""
    n = int(math.isqrt(S))
    if kind == "rect":
        src, tgt = torch.zeros(n, n, dtype=torch.uint8), torch.zeros(n, n, dtype=torch.uint8)
        src[n * 25 // 64:n * 37 // 64, n * 12 // 64:n * 25 // 64] = 1
        tgt[n * 25 // 64:n * 37 // 64, n * 20 // 64:n * 33 // 64] = 1
        return src.flatten(), tgt.flatten()
    return (torch.rand(S, generator=g) > 0.6).to(torch.uint8), (torch.rand(S, generator=g) > 0.5).to(torch.uint8)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hook", ["edit", "bggen"])
@pytest.mark.parametrize("kind", ["rect", "rand"])
@pytest.mark.parametrize("S,heads", [(4096, 5), (1024, 10), (9216, 5)])
def test_attention_tca_production_shapes(gpu, dtype, hook, kind, S, heads):
    """The TCA pass tables of the guided loop at the shapes BASELINE config 2 runs them (SD-2.1: S = 4096 / h = 5 and S = 1024 / h = 10,
    d = 64, B = 4 rows [u_e, u_r, c_e, c_r]) against the fp64 statement of /root/reference/src/utils/attention.py:1043-1091 (edit) and
    :1284-1324 (bg): reference-row K/V, per-key mask, per-query selector (edit), tiled-head rule on odd / even j = b * heads + head,
    context-guidance blend from a device scalar.  In bf16 this is attn_pp_kernel<true> with its 4-slot LDS rings wrapping 16 / 4 times
    per query block; in f32 the exact-fp32 attn_kernel."""
    import ctypes
    from freefine_amd import _lib, ops
    from freefine_amd._lib import ATT_HEAD_RULE
    g = torch.Generator().manual_seed(S + heads)
    B, D = 4, 64
    Cc = heads * D
    q = rnd((B, S, Cc), dtype, gpu, g)
    k = rnd((B, S, Cc), dtype, gpu, g)
    v = rnd((B, S, Cc), dtype, gpu, g)
    vt = ops.transpose(v)
    src, tgt = _production_masks(S, kind, g)
    cg = 0.35
    cg_dev = torch.tensor([cg], dtype=torch.float32, device=gpu)
    scale = D ** -0.5
    ref_rows = [1, 1, 3, 3]
    if hook == "edit":
        kmask, qsel = src.to(gpu), tgt.to(gpu)
        p_ref = [ops.AttnEntrySpec(b, ref_rows[b], 0.0, 1.0, kmask=kmask, qsel=qsel, flags=ATT_HEAD_RULE) for b in range(B)]
    else:       # keys allowed OUTSIDE the hole, no query-side blend
        kmask = (1 - src).to(gpu)
        p_ref = [ops.AttnEntrySpec(b, ref_rows[b], 0.0, 1.0, kmask=kmask, flags=ATT_HEAD_RULE) for b in range(B)]
    p_self = [ops.AttnEntrySpec(b, b, 1.0, -1.0) for b in range(B)]
    out = ops.attention(q, k, vt, heads, scale, [p_ref, p_self], w_dev=cg_dev)
    if dtype == torch.bfloat16:          # the kernel the bench runs for these launches
        d = _lib.AttnDesc()
        d.Bo, d.S, d.Sk, d.heads, d.D, d.npass, d.ldo, d.w_dev = B, S, S, heads, D, 2, Cc, cg_dev.data_ptr()
        for b, sp in enumerate(p_ref):
            e = d.e[b]
            e.q_row, e.kv_row, e.w_const, e.w_slope, e.kmask, e.flags = sp.q_row, sp.kv_row, sp.w_const, sp.w_slope, sp.kmask.data_ptr(), sp.flags
        for b, sp in enumerate(p_self):
            e = d.e[_lib.ATT_MAXB + b]
            e.q_row, e.kv_row, e.w_const, e.w_slope = sp.q_row, sp.kv_row, sp.w_const, sp.w_slope
        name = ctypes.create_string_buffer(160)
        _lib.load().ffn_attn_kernel_name(1, ctypes.byref(d), name, 160)
        assert b"attn_pp_kernel<true>" in name.value, name.value
    qd, kd, vd = q.double(), k.double(), v.double()
    sg, tg = src.to(gpu), tgt.to(gpu)
    if hook == "edit":
        a_even = (sg[None, :] != 0) == (tg[:, None] != 0)
    else:
        a_even = (sg[None, :] == 0).expand(S, S)
    worst = 0.0
    for b in range(B):
        allowed = [a_even if (b * heads + h) % 2 == 0 else None for h in range(heads)]
        r = _ref_attention_gpu(qd[b], kd[ref_rows[b]], vd[ref_rows[b]], heads, scale, allowed)
        s_ = _ref_attention_gpu(qd[b], kd[b], vd[b], heads, scale)
        worst = max(worst, relerr(out[b], cg * r + (1 - cg) * s_))
    print(f"TCA {hook} {kind} S={S} h={heads} {dtype}: max |diff| / max |ref| = {worst:.2e}")
    assert worst < tol(dtype)


def test_ragged_m_tail_rows_are_not_written(gpu):
    """The ping-pong kernel's bf16 / GEGLU / residual epilogues leave rows >= M of a tail tile to the buffer descriptor's range check
    (igemm_p8.h store_rows).  Canary: `out` (and the residual) are the first M rows of LARGER buffers filled with a sentinel; after the
    launch every row >= M must still hold it -- for each ping-pong configuration forced in turn, M not a multiple of the tile height."""
    from freefine_amd import _lib as L
    from freefine_amd import ops
    lib = L.load()
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(17)
    M, K, guard = 256 * 9 + 72, 320, 512
    names = []
    try:
        for cfg in range(lib.ffn_igemm_num_configs()):
            lib.ffn_igemm_force_config(cfg)
            for (N, geglu, with_res) in [(320, False, True), (640, False, False), (1280, True, False), (512, False, True)]:
                x = rnd((M, K), dt, gpu, g)
                w = rnd((N, K), dt, gpu, g, K ** -0.5)
                b = rnd((N,), torch.float32, gpu, g)
                n_out = N // 2 if geglu else N
                big = torch.full((M + guard, n_out), 7.0, dtype=dt, device=gpu)
                out = big[:M]
                rbig = rnd((M + guard, n_out), dt, gpu, g)
                if geglu:
                    wp, bp = ops.pack_geglu(w, b, dt)
                    ops.linear(x, wp, bp, K=K, geglu=True, out=out)
                    y = x.double() @ w.double().t() + b.double()
                    ref = y[:, :n_out] * F.gelu(y[:, n_out:])
                else:
                    ops.linear(x, ops.pack_linear(w, dt), b, K=K, out=out, residual=rbig[:M] if with_res else None)
                    ref = x.double() @ w.double().t() + b.double() + (rbig[:M].double() if with_res else 0.0)
                assert relerr(out, ref) < tol(dt), (cfg, N)
                assert (big[M:] == 7.0).all(), (cfg, N, geglu, with_res)
    finally:
        lib.ffn_igemm_force_config(-1)


# ---------------------------------------------------------------------------------------------------------------------
# FFN_BF16X3 ("split-bf16"): fp32 activations carried as hi + lo bf16, a product = hi*hi + hi*lo + lo*hi on the bf16 MFMA with
# fp32 accumulation.  Expected deviation from the fp64 statement: ~2^-17 per product term, averaged over K -> a few 1e-6 of the
# output scale; the gate is 2e-5 (the f32 gate), i.e. two orders below what 1e-3 latent parity needs and three below bf16 (1.5e-2).
# ---------------------------------------------------------------------------------------------------------------------
X3_TOL = 2e-5
# split-bf16 SELF attention since round 6 (attn_x3w_kernel: 32x32x16 MFMAs, 16-key accumulation steps -- sqrt 2 more fp32 accumulator roundings per
# key than attn_x3p_kernel's 32-key steps: measured 2.1e-5 at S = 4096 where attn_x3p measured 1.5e-5, 1.2e-5 vs 1.4e-5 at S = 1024)
X3_ATT_TOL = 3e-5


def test_x3_split_pair(gpu):
    from freefine_amd import ops
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(37, 328, generator=g) * torch.logspace(-3, 3, 328)).to(gpu)
    p = ops.split_pair(x, 320)
    assert p.shape == (37, 640) and p.dtype == torch.bfloat16
    # C % 32 == 0: the BLOCKED pair form -- 128-byte blocks [hi(32) | lo(32)] per 32 columns (include/freefine_hip.h FFN_BF16X3)
    blk = p.view(37, 10, 2, 32)
    hi, lo = blk[:, :, 0].reshape(37, 320).float(), blk[:, :, 1].reshape(37, 320).float()
    assert torch.equal(hi, x[:, :320].to(torch.bfloat16).float())
    assert torch.equal(lo, (x[:, :320] - hi).to(torch.bfloat16).float())
    assert ((hi + lo - x[:, :320]).abs() <= x[:, :320].abs() * 2.0 ** -16).all()
    # any other width: one block of C columns = the planes [hi(C) | lo(C)]
    p = ops.split_pair(x, 72)
    assert p.shape == (37, 144)
    hi, lo = p[:, :72].float(), p[:, 72:].float()
    assert torch.equal(hi, x[:, :72].to(torch.bfloat16).float()) and torch.equal(lo, (x[:, :72] - hi).to(torch.bfloat16).float())


@pytest.mark.parametrize("M,N,K", [(256, 320, 320), (4096, 1280, 320), (77 * 4, 640, 1024), (4, 1280, 320), (1000, 4, 320), (16384, 320, 1280),
                                    (130, 200, 72), (196608 // 8, 640, 640), (2500, 1280, 5120),
                                    # K % 64 == 32: the blocked pair layout with an ODD number of 32-deep stages on the ping-pong core
                                    (1024, 320, 96), (4096, 320, 160), (3000, 640, 224), (512, 256, 32)])
def test_x3_linear(gpu, M, N, K):
    from freefine_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    dt = torch.float32
    x = rnd((M, K), dt, gpu, g)
    w = rnd((N, K), dt, gpu, g, K ** -0.5)
    b = torch.randn(N, generator=g).to(gpu)
    res = rnd((M, N), dt, gpu, g)
    wp = ops.pack_linear(w, dt, x3=True)
    assert ops.is_x3(wp) and wp.shape == (N, (2 if K % 32 == 0 else 3) * K) and wp.dtype == torch.bfloat16
    ref = x.double() @ w.double().t() + b.double()
    out = ops.linear(x, wp, b, K=K)
    assert out.dtype == torch.float32
    e0 = relerr(out, ref)
    e1 = relerr(ops.linear(x, wp, b, K=K, residual=res), ref + res.double())
    e2 = relerr(ops.linear(x, wp, b, K=K, silu=True), F.silu(ref))
    print(f"x3 linear M={M} N={N} K={K}: {e0:.2e} / residual {e1:.2e} / silu {e2:.2e}")
    assert max(e0, e1, e2) < X3_TOL


def test_x3_geglu_rowbias_transposed_splitk(gpu):
    from freefine_amd import ops
    g = torch.Generator().manual_seed(3)
    dt = torch.float32
    B, S, K, Fh = 2, 1024, 320, 1280
    x = rnd((B, S, K), dt, gpu, g)
    w = rnd((2 * Fh, K), dt, gpu, g, K ** -0.5)
    b = torch.randn(2 * Fh, generator=g).to(gpu)
    wp, bp = ops.pack_geglu(w, b, dt, x3=True)
    out = ops.linear(x, wp, bp, geglu=True)
    y = x.double() @ w.double().t() + b.double()
    ref = y[..., :Fh] * F.gelu(y[..., Fh:])
    assert out.shape == (B, S, Fh) and out.dtype == torch.float32
    print(f"x3 geglu: {relerr(out, ref):.2e}")
    assert relerr(out, ref) < X3_TOL
    w2 = rnd((640, K), dt, gpu, g, K ** -0.5)
    rb = torch.randn(B, 640, generator=g).to(gpu)
    out = ops.linear(x, ops.pack_linear(w2, dt, x3=True), None, rowbias=rb, rows_per_batch=S)
    assert relerr(out, x.double() @ w2.double().t() + rb.double()[:, None, :]) < X3_TOL
    for S2 in (256, 77, 400):        # V^T: fp32 transposed output
        x2 = rnd((3, S2, K), dt, gpu, g)
        ld = (S2 + 7) // 8 * 8
        out = ops.linear(x2, ops.pack_linear(w2, dt, x3=True), None, rows_per_batch=S2, transposed_ld=ld)
        assert out.shape == (3, 640, ld) and relerr(out[:, :, :S2], (x2.double() @ w2.double().t()).transpose(1, 2)) < X3_TOL
    Mk, Kk, Nk = 600, 4096, 512
    a = rnd((Mk, Kk), dt, gpu, g)
    wl = rnd((Nk, Kk), dt, gpu, g, Kk ** -0.5)
    for sk in (0, 1, 4, 8):
        out = ops.linear(a, ops.pack_linear(wl, dt, x3=True), b[:Nk], splitk=sk, residual=a[:, :Nk].contiguous())
        assert relerr(out, a.double() @ wl.double().t() + b[:Nk].double() + a[:, :Nk].double()) < X3_TOL, sk


@pytest.mark.parametrize("cfg", [
    dict(B=2, H=16, W=16, Cin=64, Cout=96, stride=1, pad=1, up=False),
    dict(B=2, H=16, W=16, Cin=64, Cout=64, stride=2, pad=1, up=False),
    dict(B=1, H=8, W=8, Cin=128, Cout=64, stride=1, pad=1, up=True),
    dict(B=4, H=8, W=8, Cin=320, Cout=4, stride=1, pad=1, up=False),
    dict(B=2, H=12, W=20, Cin=8, Cout=320, stride=1, pad=1, up=False),
    dict(B=3, H=32, W=32, Cin=128, Cout=320, stride=1, pad=1, up=False),       # ping-pong tile shapes from here on
    dict(B=2, H=16, W=16, Cin=128, Cout=320, stride=1, pad=1, up=True),
    dict(B=2, H=32, W=32, Cin=64, Cout=320, stride=2, pad=1, up=False),
    dict(B=5, H=16, W=16, Cin=192, Cout=640, stride=1, pad=1, up=False),
    dict(B=4, H=64, W=64, Cin=320, Cout=320, stride=1, pad=1, up=False),
    dict(B=3, H=8, W=8, Cin=2560, Cout=1280, stride=1, pad=1, up=False),       # split-K at the coarse level, Cin = concat width
    # Cin % 64 == 32: one 32-channel block per tap on the chunk-major walk, odd stage counts, enough rows for a ping-pong tile
    dict(B=4, H=32, W=32, Cin=32, Cout=320, stride=1, pad=1, up=False),
    dict(B=4, H=32, W=32, Cin=96, Cout=320, stride=1, pad=1, up=False),
    dict(B=6, H=16, W=16, Cin=160, Cout=320, stride=1, pad=1, up=False),
    dict(B=4, H=32, W=32, Cin=96, Cout=256, stride=2, pad=1, up=False),
])
def test_x3_conv3x3(gpu, cfg):
    from freefine_amd import ops
    g = torch.Generator().manual_seed(11)
    dt = torch.float32
    B, H, W, Cin, Cout = cfg["B"], cfg["H"], cfg["W"], cfg["Cin"], cfg["Cout"]
    x = rnd((B, Cin, H, W), dt, gpu, g)
    w = rnd((Cout, Cin, 3, 3), dt, gpu, g, (9 * Cin) ** -0.5)
    b = torch.randn(Cout, generator=g).to(gpu)
    xin = x.double()
    if cfg["up"]:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, w.double(), b.double(), stride=cfg["stride"], padding=1)
    Ho, Wo = ref.shape[-2:]
    rb = torch.randn(B, Cout, generator=g).to(gpu)
    res = rnd((B, Ho * Wo, Cout), dt, gpu, g)
    x_nhwc = x.permute(0, 2, 3, 1).reshape(B, H * W, Cin).contiguous()
    wp = ops.pack_conv3x3(w, dt, x3=True)
    assert wp.shape == (Cout, (18 if Cin % 32 == 0 else 27) * Cin)
    ref_nhwc = ref.permute(0, 2, 3, 1).reshape(B, Ho * Wo, Cout)
    out = ops.conv3x3(x_nhwc, wp, b, B, H, W, Cin, stride=cfg["stride"], pad=1, upsample=cfg["up"], Hout=Ho, Wout=Wo)
    e0 = relerr(out, ref_nhwc)
    out = ops.conv3x3(x_nhwc, wp, b, B, H, W, Cin, stride=cfg["stride"], pad=1, upsample=cfg["up"], Hout=Ho, Wout=Wo, rowbias=rb, residual=res)
    e1 = relerr(out, ref_nhwc + rb.double()[:, None] + res.double())
    print(f"x3 conv {cfg}: {e0:.2e} / rowbias+residual {e1:.2e}")
    assert max(e0, e1) < X3_TOL


def test_x3_every_configuration_and_determinism(gpu):
    """every split-bf16 configuration forced in turn (generic 64x64 / 128x64 / 128x128 tiles, the four ping-pong tiles, unsplit and
    split-K, and the 256 x 128 ping-pong tile of the 128-channel convolutions), ragged M, GEGLU, residual; bit-repeatability of the pipelined kernel in this mode."""
    from freefine_amd import _lib as L
    from freefine_amd import ops
    lib = L.load()
    dt = torch.float32
    g = torch.Generator().manual_seed(5)
    try:
        for cfg in range(lib.ffn_igemm_num_configs()):
            lib.ffn_igemm_force_config(cfg)
            for (M, N, K) in [(16384 + 72, 320, 320), (6000, 640, 640), (300, 328, 136)]:
                x, w = rnd((M, K), dt, gpu, g), rnd((N, K), dt, gpu, g, K ** -0.5)
                b, r = rnd((N,), dt, gpu, g), rnd((M, N), dt, gpu, g)
                big = torch.full((M + 300, N), 7.0, dtype=dt, device=gpu)
                out = ops.linear(x, ops.pack_linear(w, dt, x3=True), b, K=K, residual=r, out=big[:M])
                assert relerr(out, x.double() @ w.double().t() + b.double() + r.double()) < X3_TOL, (cfg, M, N, K)
                assert (big[M:] == 7.0).all(), (cfg, M, N, K)
            M, K, Fh = 16384 + 40, 320, 640
            x, w, b = rnd((M, K), dt, gpu, g), rnd((2 * Fh, K), dt, gpu, g, K ** -0.5), rnd((2 * Fh,), dt, gpu, g)
            wp, bp = ops.pack_geglu(w, b, dt, x3=True)
            y = x.double() @ w.double().t() + b.double()
            assert relerr(ops.linear(x, wp, bp, K=K, geglu=True), y[:, :Fh] * F.gelu(y[:, Fh:])) < X3_TOL, (cfg, "geglu")
            B, H, Cin, Cout = 3, 16, 256, 320
            x, w = rnd((B, H * H, Cin), dt, gpu, g), rnd((Cout, Cin, 3, 3), dt, gpu, g, (9 * Cin) ** -0.5)
            b, rb, r = rnd((Cout,), dt, gpu, g), rnd((B, Cout), dt, gpu, g), rnd((B, H * H, Cout), dt, gpu, g)
            xr = x.double().reshape(B, H, H, Cin).permute(0, 3, 1, 2)
            ref = F.conv2d(xr, w.double(), b.double(), padding=1) + rb.double()[:, :, None, None]
            ref = ref.permute(0, 2, 3, 1).reshape(B, H * H, Cout) + r.double()
            for sk in (0, 12):
                out = ops.conv3x3(x, ops.pack_conv3x3(w, dt, x3=True), b, B, H, H, Cin, rowbias=rb, residual=r, splitk=sk)
                assert relerr(out, ref) < X3_TOL, (cfg, "conv", sk)
            # Cout = 128 (the VAE's 128-channel layers): round 6's 256 x 128 ping-pong tile where forced, the generic tiles otherwise; with and without residual
            for Cin3 in (128, 256):
                B3, H3, Cout3 = 3, 32, 128
                x3_, w3 = rnd((B3, H3 * H3, Cin3), dt, gpu, g), rnd((Cout3, Cin3, 3, 3), dt, gpu, g, (9 * Cin3) ** -0.5)
                b3, r3 = rnd((Cout3,), dt, gpu, g), rnd((B3, H3 * H3, Cout3), dt, gpu, g)
                ref3 = F.conv2d(x3_.double().reshape(B3, H3, H3, Cin3).permute(0, 3, 1, 2), w3.double(), b3.double(), padding=1).permute(0, 2, 3, 1).reshape(B3, H3 * H3, Cout3)
                wp3 = ops.pack_conv3x3(w3, dt, x3=True)
                assert relerr(ops.conv3x3(x3_, wp3, b3, B3, H3, H3, Cin3), ref3) < X3_TOL, (cfg, "conv N=128", Cin3)
                assert relerr(ops.conv3x3(x3_, wp3, b3, B3, H3, H3, Cin3, residual=r3), ref3 + r3.double()) < X3_TOL, (cfg, "conv N=128 + res", Cin3)
            # Cin = 96: 27 stages of 32 (odd), forced split-K on the chunk-major walk
            Cin2 = 96
            x2, w2 = rnd((B, H * H, Cin2), dt, gpu, g), rnd((Cout, Cin2, 3, 3), dt, gpu, g, (9 * Cin2) ** -0.5)
            ref2 = F.conv2d(x2.double().reshape(B, H, H, Cin2).permute(0, 3, 1, 2), w2.double(), b.double(), padding=1).permute(0, 2, 3, 1).reshape(B, H * H, Cout)
            for sk in (0, 3, 9):
                out = ops.conv3x3(x2, ops.pack_conv3x3(w2, dt, x3=True), b, B, H, H, Cin2, splitk=sk)
                assert relerr(out, ref2) < X3_TOL, (cfg, "conv Cin 96", sk)
    finally:
        lib.ffn_igemm_force_config(-1)
    B, H, Cin, Cout = 12, 64, 320, 320
    x, w = rnd((B, H * H, Cin), dt, gpu, g), ops.pack_conv3x3(rnd((Cout, Cin, 3, 3), dt, gpu, g, (9 * Cin) ** -0.5), dt, x3=True)
    b, r = rnd((Cout,), dt, gpu, g), rnd((B, H * H, Cout), dt, gpu, g)
    ref = ops.conv3x3(x, w, b, B, H, H, Cin, residual=r).clone()
    for _ in range(4):
        assert torch.equal(ops.conv3x3(x, w, b, B, H, H, Cin, residual=r), ref)


@pytest.mark.parametrize("S,Sk,heads,D", [(256, 256, 5, 64), (64, 77, 8, 40), (200, 130, 4, 16), (100, 77, 2, 8), (128, 64, 2, 160)])
def test_x3_attention_plain_and_cross(gpu, S, Sk, heads, D):
    """attn_x3_kernel (fp32 operands, split-bf16 products): plain self / cross attention, ragged S and Sk, head sizes below 64 (zero
    padded) and one above (falls back to the exact fp32 kernel), row remaps with a device-scalar weight."""
    from freefine_amd import ops
    g = torch.Generator().manual_seed(S + D)
    dt = torch.float32
    B, Cc = 3, heads * D
    q, k, v = rnd((B, S, Cc), dt, gpu, g), rnd((2, Sk, Cc), dt, gpu, g), rnd((2, Sk, Cc), dt, gpu, g)
    vt = ops.transpose(v, ld_dst=(Sk + 7) // 8 * 8)
    scale = D ** -0.5
    cg = torch.tensor([0.25], device=gpu)
    rows = [ops.AttnEntrySpec(2, 1, 1.0, 0.0), ops.AttnEntrySpec(0, 0, 0.5, 2.0), ops.AttnEntrySpec(1, 1, 1.0, 0.0), ops.AttnEntrySpec(2, 0, 1.0, 0.0)]
    out = ops.attention(q, k, vt, heads, scale, [rows], Sk=Sk, w_dev=cg, x3=True)
    assert out.dtype == torch.float32
    worst = 0.0
    for b, sp in enumerate(rows):
        ref = (sp.w_const + sp.w_slope * 0.25) * ref_attention(q[sp.q_row].double().cpu(), k[sp.kv_row].double().cpu(), v[sp.kv_row].double().cpu(), heads, scale)
        worst = max(worst, relerr(out[b], ref))
    print(f"x3 attention S={S} Sk={Sk} h={heads} d={D}: {worst:.2e}")
    assert worst < X3_TOL


@pytest.mark.parametrize("hook", ["edit", "bggen"])
@pytest.mark.parametrize("S,heads", [(4096, 5), (1024, 10), (200, 3)])
def test_x3_attention_tca(gpu, hook, S, heads):
    """the TCA pass tables (reference rows, key mask, query selector, tiled-head rule, context-guidance blend) through attn_x3_kernel
    at the production shapes and a ragged one, against the fp64 statement"""
    from freefine_amd import ops
    from freefine_amd._lib import ATT_HEAD_RULE
    g = torch.Generator().manual_seed(S + heads)
    dt, B, D = torch.float32, 4, 64
    Cc = heads * D
    q, k, v = rnd((B, S, Cc), dt, gpu, g), rnd((B, S, Cc), dt, gpu, g), rnd((B, S, Cc), dt, gpu, g)
    vt = ops.transpose(v, ld_dst=(S + 7) // 8 * 8)
    src, tgt = (torch.rand(S, generator=g) > 0.6).to(torch.uint8), (torch.rand(S, generator=g) > 0.5).to(torch.uint8)
    cg = 0.35
    cg_dev = torch.tensor([cg], dtype=torch.float32, device=gpu)
    scale = D ** -0.5
    ref_rows = [1, 1, 3, 3]
    if hook == "edit":
        p_ref = [ops.AttnEntrySpec(b, ref_rows[b], 0.0, 1.0, kmask=src.to(gpu), qsel=tgt.to(gpu), flags=ATT_HEAD_RULE) for b in range(B)]
    else:
        p_ref = [ops.AttnEntrySpec(b, ref_rows[b], 0.0, 1.0, kmask=(1 - src).to(gpu), flags=ATT_HEAD_RULE) for b in range(B)]
    p_self = [ops.AttnEntrySpec(b, b, 1.0, -1.0) for b in range(B)]
    out = ops.attention(q, k, vt, heads, scale, [p_ref, p_self], w_dev=cg_dev, Sk=S, x3=True)
    qd, kd, vd = q.double(), k.double(), v.double()
    sg, tg = src.to(gpu), tgt.to(gpu)
    a_even = ((sg[None, :] != 0) == (tg[:, None] != 0)) if hook == "edit" else (sg[None, :] == 0).expand(S, S)
    worst = 0.0
    for b in range(B):
        allowed = [a_even if (b * heads + h) % 2 == 0 else None for h in range(heads)]
        r = _ref_attention_gpu(qd[b], kd[ref_rows[b]], vd[ref_rows[b]], heads, scale, allowed)
        s_ = _ref_attention_gpu(qd[b], kd[b], vd[b], heads, scale)
        worst = max(worst, relerr(out[b], cg * r + (1 - cg) * s_))
    print(f"x3 TCA {hook} S={S} h={heads}: {worst:.2e}")
    assert worst < X3_ATT_TOL


def test_x3_attention_uniform_wq_and_determinism(gpu):
    from freefine_amd import ops
    from freefine_amd._lib import ATT_UNIFORM_SEL1
    g = torch.Generator().manual_seed(22)
    dt = torch.float32
    B, S, Sk, heads, D = 3, 100, 77, 8, 40
    Cc = heads * D
    q, k, v = rnd((B, S, Cc), dt, gpu, g), rnd((B, Sk, Cc), dt, gpu, g), rnd((B, Sk, Cc), dt, gpu, g)
    vt = ops.transpose(v, ld_dst=80)
    scale = D ** -0.5
    km = torch.zeros(Sk, dtype=torch.uint8, device=gpu)
    wq = torch.rand(S, generator=g).to(gpu)
    p0 = [ops.AttnEntrySpec(0, 0, 1.0, 0.0, kmask=km, flags=ATT_UNIFORM_SEL1), ops.AttnEntrySpec(1, 1), ops.AttnEntrySpec(2, 2, wq=wq)]
    p1 = [None, None, ops.AttnEntrySpec(0, 1, 0.5)]
    out = ops.attention(q, k, vt, heads, scale, [p0, p1], Sk=Sk, x3=True)
    qc, kc, vc = q.double().cpu(), k.double().cpu(), v.double().cpu()
    assert relerr(out[0], vc[0].mean(dim=0, keepdim=True).expand(S, Cc)) < X3_TOL
    assert relerr(out[1], ref_attention(qc[1], kc[1], vc[1], heads, scale)) < X3_TOL
    ref2 = wq.double().cpu()[:, None] * ref_attention(qc[2], kc[2], vc[2], heads, scale) + 0.5 * ref_attention(qc[0], kc[1], vc[1], heads, scale)
    assert relerr(out[2], ref2) < X3_TOL
    for _ in range(3):
        assert torch.equal(ops.attention(q, k, vt, heads, scale, [p0, p1], Sk=Sk, x3=True), out)


@pytest.mark.parametrize("S,Sk,heads", [(128, 64, 2), (256, 192, 5), (384, 1024, 3), (1000, 320, 2)])
def test_x3_attention_pingpong_schedule(gpu, S, Sk, heads):
    """attn_x3p_kernel (d = 64, Sk % 64 == 0, S >= 128): one key tile, an odd tile count, a ragged last query block, K / V rows that differ
    from the Q rows, two passes with a key mask + query selector + per-query weight and a skipped entry, fp32 and pair-form output,
    bit-for-bit repeatability (the double-buffered LDS images are rewritten one phase before they are read)."""
    from freefine_amd import ops
    from freefine_amd import _lib as L
    g = torch.Generator().manual_seed(S + Sk)
    dt, D, B = torch.float32, 64, 3
    Cc = heads * D
    q, k, v = rnd((B, S, Cc), dt, gpu, g), rnd((2, Sk, Cc), dt, gpu, g), rnd((2, Sk, Cc), dt, gpu, g)
    vt = ops.transpose(v, ld_dst=Sk)
    scale = D ** -0.5
    km = (torch.rand(Sk, generator=g) > 0.5).to(torch.uint8).to(gpu)
    qs = (torch.rand(S, generator=g) > 0.4).to(torch.uint8).to(gpu)
    wq = torch.rand(S, generator=g).to(gpu)
    cg = torch.tensor([0.3], device=gpu)
    p0 = [ops.AttnEntrySpec(0, 1, 0.0, 1.0, kmask=km, qsel=qs), ops.AttnEntrySpec(1, 0, 1.0, 0.0), ops.AttnEntrySpec(2, 1, 1.0, 0.0, wq=wq)]
    p1 = [ops.AttnEntrySpec(0, 0, 1.0, -1.0), None, ops.AttnEntrySpec(1, 0, 0.5, 0.0)]
    out = ops.attention(q, k, vt, heads, scale, [p0, p1], Sk=Sk, w_dev=cg, x3=True)
    qc, kc, vc = q.double(), k.double(), v.double()
    allowed = (km[None, :] != 0) == (qs[:, None] != 0)
    ref0 = 0.3 * _ref_attention_gpu(qc[0], kc[1], vc[1], heads, scale, [allowed] * heads) + 0.7 * _ref_attention_gpu(qc[0], kc[0], vc[0], heads, scale)
    ref1 = _ref_attention_gpu(qc[1], kc[0], vc[0], heads, scale)
    ref2 = wq.double()[:, None] * _ref_attention_gpu(qc[2], kc[1], vc[1], heads, scale) + 0.5 * _ref_attention_gpu(qc[1], kc[0], vc[0], heads, scale)
    errs = [relerr(out[0], ref0), relerr(out[1], ref1), relerr(out[2], ref2)]
    print(f"x3 ping-pong attention S={S} Sk={Sk} h={heads}: {max(errs):.2e}")
    assert max(errs) < X3_ATT_TOL
    pair = ops.attention(q, k, vt, heads, scale, [p0, p1], Sk=Sk, w_dev=cg, x3=True, out_pair=True)
    assert pair.dtype == torch.bfloat16 and pair.shape == (B, S, 2 * Cc)
    assert relerr(pair_value(pair, Cc), out.double()) < 2e-5
    for _ in range(3):
        assert torch.equal(ops.attention(q, k, vt, heads, scale, [p0, p1], Sk=Sk, w_dev=cg, x3=True), out)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(3, 8, 8, 1280, 1280), (5, 16, 16, 1280, 1280), (4, 32, 32, 640, 640), (2, 12, 20, 64, 256), (1, 16, 12, 128, 320)])
def test_subpixel_upsample_conv(gpu, B, H, W, Cin, Cout):
    """`nearest-2x upsample -> 3x3 conv` evaluated at low resolution as four 2x2 convolutions (conv = 2 of ffn_igemm: the ping-pong kernel
    with two taps per window row and a per-class window origin) + pixel shuffle, at the three upsampler shapes of the SD UNet and two
    ragged ones: against the fp64 convolution of the upsampled input with the bf16-rounded ORIGINAL weights (the summed taps are rounded
    once, so the tolerance is bf16's), against the fused-gather kernel it replaces, written into a column view of a wider buffer."""
    from freefine_amd import ops
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(B + H + Cin)
    dt = torch.bfloat16
    x = rnd((B, H * W, Cin), dt, gpu, g)
    w = rnd((Cout, Cin, 3, 3), torch.float32, gpu, g, (9 * Cin) ** -0.5)
    b = rnd((Cout,), torch.float32, gpu, g)
    assert ops.up2x_eligible(Cin, Cout, B * H * W)
    w4 = ops.pack_conv3x3_up2x(w, dt)
    big = torch.full((B, 4 * H * W, Cout + 64), 7.0, dtype=dt, device=gpu)
    out = ops.conv3x3_up2x(x, w4, b, B, H, W, Cin, out=big[..., :Cout])
    xr = F.interpolate(x.double().view(B, H, W, Cin).permute(0, 3, 1, 2), scale_factor=2, mode="nearest")
    ref = F.conv2d(xr, w.double(), b.double(), padding=1).permute(0, 2, 3, 1).reshape(B, 4 * H * W, Cout)
    old = ops.conv3x3(x, ops.pack_conv3x3(w, dt), b, B, H, W, Cin, upsample=True)
    e_new, e_old = relerr(out, ref), relerr(old, ref)
    print(f"sub-pixel upsample conv {B}x{H}x{W} {Cin}->{Cout}: {e_new:.2e} (fused-gather 3x3 kernel: {e_old:.2e})")
    assert e_new < tol(dt) and (big[..., Cout:] == 7.0).all()
    for _ in range(2):
        assert torch.equal(ops.conv3x3_up2x(x, w4, b, B, H, W, Cin), out.contiguous())
    # split-bf16 mode: fp32 activations, [hi | lo | hi] weights of the summed taps, fp32 result at fp32-level accuracy
    xf = rnd((B, H * W, Cin), torch.float32, gpu, g)
    o3 = ops.conv3x3_up2x(xf, ops.pack_conv3x3_up2x(w, torch.float32, x3=True), b, B, H, W, Cin)
    xr = F.interpolate(xf.double().view(B, H, W, Cin).permute(0, 3, 1, 2), scale_factor=2, mode="nearest")
    ref3 = F.conv2d(xr, w.double(), b.double(), padding=1).permute(0, 2, 3, 1).reshape(B, 4 * H * W, Cout)
    assert o3.dtype == torch.float32 and relerr(o3, ref3) < X3_TOL, relerr(o3, ref3)


# ---------------------------------------------------------------------------------------------------------------------
# FFN_FP8: e4m3 operands for the 3x3 convolutions of the bf16 fast mode.  The GEMM is exact on its quantised operands up to fp32
# accumulation and the bf16 output rounding, so the kernel tests quantise on the host and compare against the fp64 convolution of the
# DEQUANTISED operands at the bf16 tolerance; what the quantisation costs is measured at the UNet level, not here.
# ---------------------------------------------------------------------------------------------------------------------
def _quant_act_f8(x, Cp):
    """[B, HW, C] float -> (uint8 [B, HW, Cp] e4m3 of x * F8_ACT_SCALE, dequantised double [B, HW, C])"""
    from freefine_amd import ops
    B, HW, C = x.shape
    q = torch.zeros(B, HW, Cp, dtype=torch.float32, device=x.device)
    q[..., :C] = (x.float() * ops.F8_ACT_SCALE).clamp(-448, 448)
    q8 = q.to(torch.float8_e4m3fn)
    out = q8.view(torch.uint8).contiguous()
    out._ffn_f8_act = C
    return out, q8.float()[..., :C].double() / ops.F8_ACT_SCALE


@pytest.mark.parametrize("cfg", [
    dict(B=3, H=32, W=32, Cin=128, Cout=320),          # ping-pong tile shapes
    dict(B=4, H=64, W=64, Cin=320, Cout=320),          # Cin padded 320 -> 384
    dict(B=5, H=16, W=16, Cin=640, Cout=640),
    dict(B=3, H=8, W=8, Cin=2560, Cout=1280),          # split-K at the coarse level
    dict(B=1, H=8, W=8, Cin=64, Cout=96),              # generic tiles (M < 192, N not a tile multiple), Cin padded 64 -> 128
    dict(B=2, H=12, W=20, Cin=32, Cout=4),
])
def test_fp8_conv3x3(gpu, cfg):
    from freefine_amd import ops
    g = torch.Generator().manual_seed(11)
    B, H, W, Cin, Cout = cfg["B"], cfg["H"], cfg["W"], cfg["Cin"], cfg["Cout"]
    x = rnd((B, H * W, Cin), torch.float32, gpu, g).abs_() * 0.7          # SiLU-like range, mostly positive
    w = rnd((Cout, Cin, 3, 3), torch.float32, gpu, g, (9 * Cin) ** -0.5)
    b = torch.randn(Cout, generator=g).to(gpu)
    rb = torch.randn(B, Cout, generator=g).to(gpu)
    res = rnd((B, H * W, Cout), torch.bfloat16, gpu, g)
    wp = ops.pack_conv3x3_f8(w)
    Cp, alpha = wp._ffn_f8
    assert Cp % 128 == 0 and wp.shape == (Cout, 9 * Cp)
    wdq = wp.view(torch.float8_e4m3fn).float().reshape(Cout, 3, 3, Cp)[..., :Cin].permute(0, 3, 1, 2).double() * (alpha * ops.F8_ACT_SCALE)
    x8, xdq = _quant_act_f8(x, Cp)
    ref = F.conv2d(xdq.reshape(B, H, W, Cin).permute(0, 3, 1, 2), wdq, b.double(), padding=1).permute(0, 2, 3, 1).reshape(B, H * W, Cout)
    out = ops.conv3x3(x8, wp, b, B, H, W, Cin)
    assert out.dtype == torch.bfloat16
    e0 = relerr(out, ref)
    out = ops.conv3x3(x8, wp, b, B, H, W, Cin, rowbias=rb, residual=res)
    e1 = relerr(out, ref + rb.double()[:, None] + res.double())
    # and what the quantisation itself costs on this input, for the record
    true = F.conv2d(x.double().reshape(B, H, W, Cin).permute(0, 3, 1, 2), w.double(), b.double(), padding=1).permute(0, 2, 3, 1).reshape(B, H * W, Cout)
    print(f"fp8 conv {cfg}: kernel vs dequantised operands {e0:.2e} / {e1:.2e}; quantisation error of the result {relerr(ref, true):.2e}")
    assert max(e0, e1) < tol(torch.bfloat16)


def test_fp8_groupnorm_and_every_configuration(gpu):
    from freefine_amd import _lib as L
    from freefine_amd import ops
    lib = L.load()
    g = torch.Generator().manual_seed(3)
    # GroupNorm + SiLU written as e4m3 with channel padding
    for (B, HW, C, Cp) in [(2, 1024, 320, 384), (1, 256, 640, 640), (3, 64, 64, 128)]:
        x = rnd((B, HW, C), torch.bfloat16, gpu, g, 2.0) + 0.5
        gamma, beta = torch.randn(C, generator=g).to(gpu), torch.randn(C, generator=g).to(gpu)
        y8 = ops.groupnorm_f8(x, gamma, beta, 32, 1e-5, Cp, silu=True)
        assert y8.shape == (B, HW, Cp) and y8.dtype == torch.uint8 and (y8[..., C:] == 0).all()
        ref = F.silu(F.group_norm(x.double().transpose(1, 2), 32, gamma.double(), beta.double(), 1e-5).transpose(1, 2)) * ops.F8_ACT_SCALE
        deq = y8.view(torch.float8_e4m3fn).float()[..., :C].double()
        err = (deq - ref.clamp(-448, 448)).abs()
        assert (err <= ref.abs() * 2.0 ** -4 + 2.0 ** -9 + 2e-2).all(), err.max().item()      # e4m3: 3 mantissa bits, subnormal step 2^-9; bf16 input
    # every configuration forced in turn on an fp8 conv
    try:
        for cfg in range(lib.ffn_igemm_num_configs()):
            lib.ffn_igemm_force_config(cfg)
            for (B, H, Cin, Cout, sk) in [(3, 32, 128, 320, 0), (3, 16, 256, 320, 0), (3, 16, 256, 320, 6), (2, 24, 64, 96, 0)]:
                x = rnd((B, H * H, Cin), torch.float32, gpu, g).abs_() * 0.7
                w = rnd((Cout, Cin, 3, 3), torch.float32, gpu, g, (9 * Cin) ** -0.5)
                b, rb, r = rnd((Cout,), torch.float32, gpu, g), rnd((B, Cout), torch.float32, gpu, g), rnd((B, H * H, Cout), torch.bfloat16, gpu, g)
                wp = ops.pack_conv3x3_f8(w)
                Cp, alpha = wp._ffn_f8
                wdq = wp.view(torch.float8_e4m3fn).float().reshape(Cout, 3, 3, Cp)[..., :Cin].permute(0, 3, 1, 2).double() * (alpha * ops.F8_ACT_SCALE)
                x8, xdq = _quant_act_f8(x, Cp)
                ref = F.conv2d(xdq.reshape(B, H, H, Cin).permute(0, 3, 1, 2), wdq, b.double(), padding=1).permute(0, 2, 3, 1).reshape(B, H * H, Cout)
                out = ops.conv3x3(x8, wp, b, B, H, H, Cin, rowbias=rb, residual=r, splitk=sk)
                assert relerr(out, ref + rb.double()[:, None] + r.double()) < tol(torch.bfloat16), (cfg, H, Cin, sk)
    finally:
        lib.ffn_igemm_force_config(-1)


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "bf16"])
def test_ffn_attn_vs_reference_g1_fixture(gpu, mode):
    """ffn_attn, driven by Attention_Modulator's own pass tables, against the numbers the REFERENCE's Attention_Modulator produced
    (tests/golden/g1_attention.npz: Temporal_contextal_attention / _bg / _compose, modulate_local_cross_attn / _compose and the plain branch of
    /root/reference/src/utils/attention.py at S = 64 / 256, 8 x 8 and 5 x 16 heads, uint8 and float masks, with and without upcast): the
    kernel tied to the reference's numbers directly, not through an in-test restatement."""
    from test_oracle_golden import g1_inputs, g1_masks
    from freefine_amd import ops
    from freefine_amd.attention import Attention_Modulator
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g1_attention.npz"))
    dtype = torch.bfloat16 if mode == "bf16" else torch.float32
    x3 = mode == "bf16x3"
    limit = {"f32": 2e-5, "bf16x3": 1e-4, "bf16": 4e-2}[mode]
    ncase = len([k for k in g.files if k.endswith("_meta")])
    worst = {}
    for ci in range(ncase):
        heads, d, S, is_float, upcast = (int(v) for v in g[f"c{ci}_meta"])
        cg = float(g[f"c{ci}_cg"][0])
        src, tgt, src2, tgt2 = g1_masks("float" if is_float else "uint8")
        q4, k4, v4, kt, vt_, kc, vc = g1_inputs(ci, heads, d, S)
        scale = d ** -0.5
        cg_dev = torch.tensor([cg], dtype=torch.float32, device=gpu)
        dev = lambda t: t.to(gpu, dtype).contiguous()
        q, k, vT = dev(q4), dev(k4), ops.transpose(dev(v4))

        def run(hook, method, is_cross, kk=None, vv=None, setup=None):
            m = Attention_Modulator(start_layer=10)
            m.num_att_layers, m.cur_att_layer = 32, 20 + (1 if is_cross else 0)              # block 10: inside layer_idx
            m.method, m.context_guidance = method, cg
            m.use_tca, m.local_edit = (not is_cross), is_cross
            m.fg_retain_mask, m.fg_retain_mask_st2, m.fg_ref_mask, m.local_edit_region = tgt.clone(), tgt.clone(), src.clone(), tgt.clone()
            if setup:
                setup(m)
            kq = k if kk is None else dev(kk)
            Sk = kq.shape[1]
            vq = vT if vv is None else ops.transpose(dev(vv), ld_dst=(Sk + 7) // 8 * 8)          # V^T rows padded to whole 16-byte chunks (77 -> 80)
            plan = m.plan(hook, is_cross, "up", 4, S, heads, gpu)
            return ops.attention(q, kq, vq, heads, scale, plan["passes"], Sk=Sk, w_dev=cg_dev if plan["needs_cg"] else None, x3=x3)

        def compose(m):
            m.src_masks, m.tgt_masks = torch.stack([src, src2]), torch.stack([tgt, tgt2, 1 - torch.maximum(tgt, tgt2)])
            m.prompt_length = 3
        outs = {}
        for method in ("tca", "mmsa"):
            outs[f"edit_{method}"] = run("edit", method, False)
            outs[f"bg_{method}"] = run("bggen", method, False)
            outs[f"compose_{method}"] = run("compose", method, False, setup=compose)
        outs["cross_local"] = run("edit", None, True, kt, vt_)
        outs["cross_compose"] = run("compose", None, True, kc, vc, setup=compose)
        outs["plain"] = ops.attention(q, k, vT, heads, scale, None, x3=x3)
        for name, o in outs.items():
            o = o.float().cpu()
            sub = torch.from_numpy(g[f"c{ci}_{name}_sub"])
            e = (o[:, ::5, ::3] - sub).abs().max().item() / max(1.0, sub.abs().max().item())
            worst[name] = max(worst.get(name, 0.0), e)
            assert e < limit, (ci, name, e)
            if mode != "bf16":
                s1, _ = g[f"c{ci}_{name}_sum"]
                assert abs(o.double().sum().item() - s1) < 2e-3 * (1 + abs(s1)), (ci, name)
    print(f"ffn_attn vs the reference's G1 outputs, {mode}: " + ", ".join(f"{k} {v:.1e}" for k, v in worst.items()))


@pytest.mark.parametrize("S,heads,passes", [(4096, 5, 2), (1024, 10, 1), (256, 20, 2), (320, 5, 2)])
def test_x3_attention_presplit_kv_is_bit_identical(gpu, monkeypatch, S, heads, passes):
    """Split-bf16 self attention with K / V^T pre-split ONCE per call (ffn_attn_presplit: bf16 hi / lo images staged by LDS-DMA).
    Round 5: attn_x3p_kernel<., PAIRKV = true> against the same kernel splitting the fp32 tiles inside its key loop -- same split values, MFMA order
    and softmax, so the results agree BIT FOR BIT (FFN_ATTN_X3W=0 selects that kernel).  Round 6: attn_x3w_kernel (one wave per SIMD, 32x32x16 MFMAs,
    the default for pre-split launches) reads the same images and agrees with it to fp32 summation order -- masked two-pass TCA tables (key mask,
    query selector, tiled-head rule, device-scalar blend) and the plain pass, rings wrapping 64 / 16 / 4 / 5 times; fp64 check on the plain pass;
    bit-for-bit repeatability of the new kernel."""
    import ctypes
    from freefine_amd import _lib, ops
    from freefine_amd._lib import ATT_HEAD_RULE
    g = torch.Generator().manual_seed(S * 3 + heads)
    B, D = 4, 64
    Cc = heads * D
    q, k, v = (rnd((B, S, Cc), torch.float32, gpu, g) for _ in range(3))
    vt = ops.transpose(v)
    scale = D ** -0.5
    cg_dev = torch.tensor([0.35], dtype=torch.float32, device=gpu)
    if passes == 2:
        km = (torch.rand(S, generator=g) > 0.7).to(torch.uint8).to(gpu)
        qs = (torch.rand(S, generator=g) > 0.5).to(torch.uint8).to(gpu)
        P = [[ops.AttnEntrySpec(b, b | 1, 0.0, 1.0, kmask=km, qsel=qs, flags=ATT_HEAD_RULE) for b in range(B)], [ops.AttnEntrySpec(b, b, 1.0, -1.0) for b in range(B)]]
    else:
        P = None
    d = _lib.AttnDesc()
    d.Bo, d.S, d.Sk, d.heads, d.D, d.npass, d.ldo, d.kv_pair = B, S, S, heads, D, 1, Cc, 1
    for b in range(B):
        d.e[b].q_row, d.e[b].kv_row, d.e[b].w_const = b, b, 1.0
    name = ctypes.create_string_buffer(160)
    monkeypatch.setenv("FFN_ATTN_X3W", "0")
    _lib.load().ffn_attn_kernel_name(_lib.FFN_BF16X3, ctypes.byref(d), name, 160)
    assert b"attn_x3p_kernel<false, true>" in name.value, name.value
    outs = {}
    for pre in (True, False):
        monkeypatch.setattr(ops, "_ATTN_PRESPLIT", pre)
        outs[pre] = ops.attention(q, k, vt, heads, scale, P, w_dev=cg_dev, x3=True)
    assert torch.equal(outs[True], outs[False])
    monkeypatch.setenv("FFN_ATTN_X3W", "1")
    monkeypatch.setattr(ops, "_ATTN_PRESPLIT", True)
    _lib.load().ffn_attn_kernel_name(_lib.FFN_BF16X3, ctypes.byref(d), name, 160)
    assert b"attn_x3w_kernel<false>" in name.value, name.value
    ow = ops.attention(q, k, vt, heads, scale, P, w_dev=cg_dev, x3=True)
    e = relerr(ow, outs[True].double())
    print(f"attn_x3w vs attn_x3p (pre-split images) S={S} h={heads} passes={passes}: {e:.2e}")
    assert e < 1e-5
    for _ in range(3):
        assert torch.equal(ops.attention(q, k, vt, heads, scale, P, w_dev=cg_dev, x3=True), ow)
    if passes == 1:
        ref = torch.stack([_ref_attention_gpu(q[b].double(), k[b].double(), v[b].double(), heads, scale) for b in range(B)])
        assert relerr(outs[True], ref) < X3_TOL and relerr(ow, ref) < X3_TOL


@pytest.mark.parametrize("tuned", [False, True])
@pytest.mark.parametrize("B,S,heads,pairA", [(8, 4096, 5, True), (3, 4096, 5, True), (12, 1024, 10, True), (3, 1024, 10, False), (24, 256, 20, True), (3, 128, 5, True)])
def test_x3_kv64_projections_write_the_presplit_images(gpu, B, S, heads, pairA, tuned):
    """FFN_IG_OUT_KV64 (round 6): the self-attention K projection (the k half of the fused q|k GEMM, row-major) and the V^T projection (transposed
    output) write the attention kernels' pre-split [hi(64) | lo(64)] images in the bytes of the fp32 values they replace.
    tuned = False (ffn_igemm_tune_enable(0): the heuristic tile, the same kernel with and without the flag -- ping-pong tiles for the large batches,
    the two-stage tile for the small ones): against ffn_attn_presplit of the plain fp32 projections the images agree BIT FOR BIT, the q half is
    untouched, and attention over the strided images (kv_images=True) is bit-equal to the presplit path.
    tuned = True (the timing-based tile choice may differ between the two launches, i.e. fp32 summation order): the images decode (hi + lo) to the fp32
    projections within the pair form's 2^-16 and attention agrees to the split-bf16 tolerance.  All UNet levels' widths (C = 320 / 640 / 1280)."""
    from freefine_amd import _lib, ops
    g = torch.Generator().manual_seed(S + heads)
    D = 64
    C = heads * D
    lib = _lib.load()
    y = rnd((B, S, C), torch.float32, gpu, g)
    wqk = ops.pack_linear(rnd((2 * C, C), torch.float32, gpu, g, C ** -0.5), torch.float32, x3=True)
    wv = ops.pack_linear(rnd((C, C), torch.float32, gpu, g, C ** -0.5), torch.float32, x3=True)
    ya = ops.layernorm(y, torch.ones(C, device=gpu), torch.zeros(C, device=gpu), pair=True) if pairA else y
    prev = lib.ffn_igemm_tune_enable(1 if tuned else 0)
    try:
        qk = ops.linear(ya, wqk, None, K=C, splitk=1)             # (the image-writing launches never take the split-K form)
        vt = ops.linear(ya, wv, None, K=C, rows_per_batch=S, transposed_ld=S)
        qk_i = ops.linear(ya, wqk, None, K=C, kv64_from=C)
        vt_i = ops.linear(ya, wv, None, K=C, rows_per_batch=S, transposed_ld=S, kv64_from=0)
    finally:
        lib.ffn_igemm_tune_enable(prev)
    assert qk_i.shape == qk.shape and vt_i.shape == vt.shape and qk_i.dtype == torch.float32
    kp = torch.empty(B, S, 2 * C, dtype=torch.bfloat16, device=gpu)
    vp = torch.empty(B, C, 2 * S, dtype=torch.bfloat16, device=gpu)
    _lib.check(lib.ffn_attn_presplit(ops._stream(), qk[..., C:].data_ptr(), vt.data_ptr(), kp.data_ptr(), vp.data_ptr(), B, S, heads, 2 * C, S), "ffn_attn_presplit")
    ki = qk_i.view(torch.bfloat16)[..., 2 * C:]                  # [B, S, 2C] bf16: per head [hi(64) | lo(64)]
    vi = vt_i.view(torch.bfloat16)                               # [B, C, 2S] bf16: per 64 keys [hi(64) | lo(64)]
    if not tuned:
        assert torch.equal(qk_i[..., :C], qk[..., :C])           # the q half: plain fp32
        assert torch.equal(ki.view(torch.int16), kp.view(torch.int16))
        assert torch.equal(vi.view(torch.int16), vp.view(torch.int16))
    else:
        assert relerr(qk_i[..., :C], qk[..., :C].double()) < 2e-6
        kd = ki.reshape(B, S, heads, 2, 64).float().sum(3).reshape(B, S, C)
        vd = vi.reshape(B, C, S // 64, 2, 64).float().sum(3).reshape(B, C, S)
        assert relerr(kd, qk[..., C:].double()) < 2e-5 and relerr(vd, vt.double()) < 2e-5
        assert (kd - qk[..., C:]).abs().max() < 1e-4 and (vd - vt).abs().max() < 1e-4           # no misplaced element
    # attention over the images where they lie (K image rows 2C fp32 apart, inside the q|k buffer)
    km = (torch.rand(S, generator=g) > 0.6).to(torch.uint8).to(gpu)
    Ba = min(B, 4)
    P = [[ops.AttnEntrySpec(b, (b + 1) % Ba, 0.0, 1.0, kmask=km) for b in range(Ba)], [ops.AttnEntrySpec(b, b, 1.0, -1.0) for b in range(Ba)]]
    cg = torch.tensor([0.4], dtype=torch.float32, device=gpu)
    scale = D ** -0.5
    for passes in ([[ops.AttnEntrySpec(b, b) for b in range(Ba)]], P):
        assert ops.kv_images_ok(D, S, S, passes)
        ref = ops.attention(qk, qk[..., C:], vt, heads, scale, passes, Sk=S, C=C, w_dev=cg, x3=True)
        img = ops.attention(qk_i, qk_i[..., C:], vt_i, heads, scale, passes, Sk=S, C=C, w_dev=cg, x3=True, kv_images=True)
        if not tuned:
            assert torch.equal(ref, img)
        else:
            assert relerr(img, ref.double()) < X3_ATT_TOL
    # a plan with a degenerate uniform-softmax entry runs on the fp32-operand kernel: not eligible
    Pu = [[ops.AttnEntrySpec(b, b, 1.0, 0.0, kmask=km, flags=_lib.ATT_UNIFORM_SEL1) for b in range(Ba)]]
    assert not ops.kv_images_ok(D, S, S, Pu)
    assert not ops.kv_images_ok(D, S, S, None, 2 ** 31) and not ops.kv_images_ok(40, S, S) and not ops.kv_images_ok(D, 64, 64)


@pytest.mark.parametrize("tuned", [False, True])
@pytest.mark.parametrize("M,C", [(8 * 4096, 320), (384, 320), (12 * 1024, 640), (24 * 256, 1280), (6 * 64, 1280)])
def test_x3_linear_pair_output_with_residual(gpu, M, C, tuned):
    """Round 6: the feed-forward's second GEMM (K = 4C, + bias + fp32 residual) writes the block's last residual sum directly as the pair rows the proj_out
    GEMM reads (FFN_IG_OUT_PAIR beside a residual: ping-pong tile, two-stage tile and the split-K reduce all add the residual in fp32 BEFORE the split).
    Same kernel with and without the flag (tuner off): bit-equal to ffn_split_pair of the fp32 result.  Tuner on (tile / split-K choice may differ):
    hi + lo equals the fp32 result to the pair form's resolution."""
    from freefine_amd import _lib, ops
    g = torch.Generator().manual_seed(M + C)
    lib = _lib.load()
    y = rnd((M, 4 * C), torch.float32, gpu, g)
    res = rnd((M, C), torch.float32, gpu, g)
    w = ops.pack_linear(rnd((C, 4 * C), torch.float32, gpu, g, (4 * C) ** -0.5), torch.float32, x3=True)
    b = rnd((C,), torch.float32, gpu, g)
    ya = ops.split_pair(y, 4 * C)
    prev = lib.ffn_igemm_tune_enable(1 if tuned else 0)
    try:
        f = ops.linear(ya, w, b, K=4 * C, residual=res)
        p = ops.linear(ya, w, b, K=4 * C, residual=res, out_pair=True)
    finally:
        lib.ffn_igemm_tune_enable(prev)
    assert p.dtype == torch.bfloat16 and tuple(p.shape) == (M, 2 * C) and ops.pair_width(p) == C
    if not tuned:
        assert torch.equal(p.view(torch.int16), ops.split_pair(f, C).view(torch.int16))
    else:
        assert relerr(pair_value(p, C), f.double()) < 1e-5
        assert (pair_value(p, C) - f).abs().max() < 1e-4
    # and it is what the next GEMM reads: proj_out on the pair rows == proj_out on the fp32 sum
    w2 = ops.pack_linear(rnd((C, C), torch.float32, gpu, g, C ** -0.5), torch.float32, x3=True)
    assert relerr(ops.linear(p, w2, None, K=C), ops.linear(f, w2, None, K=C).double()) < 2e-5


@pytest.mark.parametrize("B,HW,C,silu", [(2, 4096, 640, True), (3, 1024, 1920, True), (2, 4096, 320, False), (2, 64, 1280, True), (1, 256, 2560, True)])
def test_groupnorm_pair_raw_writes_both_operands(gpu, B, HW, C, silu):
    """ffn_groupnorm_pair_raw (round 6): norm1 of a ResBlock with a 1x1 shortcut writes, from one apply pass, the pair rows of SiLU(GroupNorm(x)) (conv1's operand) and
    the pair rows of x itself (the shortcut GEMM's operand).  Both BIT FOR BIT what the separate launches give (ffn_groupnorm(..., FFN_NORM_OUT_PAIR) and
    ffn_split_pair); small tensors (the fused one-launch GroupNorm's shapes) keep the separate launches behind the same call."""
    from freefine_amd import ops
    g = torch.Generator().manual_seed(B * HW + C)
    x = rnd((B, HW, C), torch.float32, gpu, g)
    ga, be = torch.randn(C, generator=g).to(gpu), torch.randn(C, generator=g).to(gpu)
    y, yr = ops.groupnorm_pair_raw(x, ga, be, 32, 1e-5, silu=silu)
    assert ops.pair_width(y) == C and ops.pair_width(yr) == C and y.dtype == torch.bfloat16 and tuple(yr.shape) == (B, HW, 2 * C)
    assert torch.equal(y.view(torch.int16), ops.groupnorm(x, ga, be, 32, 1e-5, silu=silu, pair=True).view(torch.int16))
    assert torch.equal(yr.view(torch.int16), ops.split_pair(x, C).view(torch.int16))
    # and the shortcut GEMM reads it: same result as from the fp32 tensor
    w = ops.pack_linear(rnd((320, C), torch.float32, gpu, g, C ** -0.5), torch.float32, x3=True)
    assert torch.equal(ops.linear(yr, w, None, K=C, splitk=1), ops.linear(x, w, None, K=C, splitk=1))


def test_x3_kv64_rejects_what_it_cannot_write(gpu):
    """the C ABI refuses FFN_IG_OUT_KV64 outside its contract instead of writing a wrong image: not split-bf16, an epilogue with bias-free extras,
    widths that are not whole 64-blocks"""
    from freefine_amd import ops
    g = torch.Generator().manual_seed(2)
    C, S = 320, 128
    y = rnd((1, S, C), torch.float32, gpu, g)
    w = ops.pack_linear(rnd((2 * C, C), torch.float32, gpu, g, C ** -0.5), torch.float32, x3=True)
    with pytest.raises(RuntimeError):
        ops.linear(y, w, None, K=C, kv64_from=C + 32)                                       # not a 64-column boundary
    with pytest.raises(RuntimeError):
        ops.linear(y, w, None, K=C, rows_per_batch=S, transposed_ld=S + 8, kv64_from=0)     # row stride not whole blocks
    wf = ops.pack_linear(rnd((2 * C, C), torch.float32, gpu, g, C ** -0.5), torch.float32)
    with pytest.raises((RuntimeError, AssertionError)):
        ops.linear(y, wf, None, K=C, kv64_from=C)                                           # fp32 arithmetic: no images


def test_x3_pair_producers_write_the_blocked_layout(gpu):
    """every producer of split-bf16 pair rows writes the layout ffn_split_pair writes (128-byte blocks [hi(32) | lo(32)] for C % 32 == 0):
    GroupNorm (three-launch and fused forms), LayerNorm, the GEGLU projection's pair output and the attention output -- each against the split
    of its own fp32 result."""
    from freefine_amd import ops
    g = torch.Generator().manual_seed(5)
    for B, HW, C in ((2, 4096, 320), (3, 64, 1280)):
        x = rnd((B, HW, C), torch.float32, gpu, g)
        ga, be = torch.randn(C, generator=g).to(gpu), torch.randn(C, generator=g).to(gpu)
        for silu in (False, True):
            f = ops.groupnorm(x, ga, be, 32, 1e-5, silu=silu)
            p = ops.groupnorm(x, ga, be, 32, 1e-5, silu=silu, pair=True)
            assert ops.pair_width(p) == C
            from freefine_amd import _lib as L_
            if silu and not L_.load().ffn_gn_fused(B, HW, C, 32):
                # round 6: the 8-wide pair apply evaluates SiLU as x * rcp(1 + exp2(-x log2 e)) (~2e-7 relative) where the fp32 form calls libm:
                # same layout, values equal to the pair form's own 2^-17 resolution
                assert relerr(pair_value(p, C), f) < 8e-6
            else:
                assert torch.equal(p, ops.split_pair(f, C))
        f, p = ops.layernorm(x, ga, be), ops.layernorm(x, ga, be, pair=True)
        assert ops.pair_width(p) == C and torch.equal(p, ops.split_pair(f, C))
    M, K, Fh = 4096, 320, 1280
    x = rnd((M, K), torch.float32, gpu, g)
    w = rnd((2 * Fh, K), torch.float32, gpu, g, K ** -0.5)
    wp, bp = ops.pack_geglu(w, torch.randn(2 * Fh, generator=g).to(gpu), torch.float32, x3=True)
    f = ops.linear(x, wp, bp, geglu=True)
    p = ops.linear(x, wp, bp, geglu=True, out_pair=True)
    assert ops.pair_width(p) == Fh and relerr(pair_value(p, Fh), f.double()) < X3_TOL      # (two kernels, two erf forms: the layout is what is checked)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,Cin", [(3, 64, 64, 320), (2, 16, 24, 32), (1, 5, 7, 128)])
def test_conv3x3_n4_direct(gpu, dtype, B, H, W, Cin):
    """the UNet's conv_out (3x3, four output channels) as a direct fp32 convolution on the vector ALU (ffn_conv3x3_n4, round 5) against
    torch's conv2d in fp64: exact fp32 products in every mode -- 1e-6 of the output scale on fp32 activations, the bf16 INPUT rounding only on bf16."""
    from freefine_amd import ops
    g = torch.Generator().manual_seed(B * 100 + Cin)
    x = rnd((B, Cin, H, W), torch.float32, gpu, g)
    w = rnd((4, Cin, 3, 3), torch.float32, gpu, g, (9 * Cin) ** -0.5)
    b = torch.randn(4, generator=g).to(gpu)
    xin = x.to(dtype)
    ref = F.conv2d(xin.double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1).reshape(B, H * W, 4)
    out = ops.conv3x3_n4(xin.permute(0, 2, 3, 1).reshape(B, H * W, Cin).contiguous(), ops.pack_conv3x3_n4(w), b, B, H, W, Cin)
    assert out.dtype == torch.float32 and out.shape == (B, H * W, 4)
    err = relerr(out, ref)
    print(f"conv3x3_n4 {dtype} B={B} {H}x{W} Cin={Cin}: {err:.2e}")
    assert err < 2e-6
