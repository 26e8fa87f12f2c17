"""Typed wrappers over the C ABI for torch device tensors (device memory + current stream only; no torch math).

Activations are [B, HW, C] channel-contiguous tensors of dtype float32 ("parity mode") or bfloat16 ("fast mode").
"""
import ctypes as CT
import math

import os

import torch

from . import _lib as L


def _dt(t):
    if t.dtype == torch.float32:
        return L.FFN_F32
    if t.dtype == torch.bfloat16:
        return L.FFN_BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def epc(dtype):
    return 4 if dtype == torch.float32 else 8


def kstage(dtype):
    return 8 * epc(dtype)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


# ---------------------------------------------------------------------------------------------------------------
# optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg)
# ---------------------------------------------------------------------------------------------------------------
_PROF = None


def profile_begin():
    global _PROF
    _PROF = []


def profile_end():
    """-> {kernel name: dict(calls, total_ms, flops, bytes)}; kernel names spelled like rocprofv3's kernel trace."""
    global _PROF
    rec, _PROF = _PROF, None
    torch.cuda.synchronize()
    out = {}
    for name, flops, nbytes, s, e in rec:
        d = out.setdefault(name, dict(calls=0, total_ms=0.0, flops=0.0, bytes=0.0))
        d["calls"] += 1
        d["total_ms"] += s.elapsed_time(e)
        d["flops"] += flops
        d["bytes"] += nbytes
    return out


def _timed(name, flops, nbytes, fn):
    if _PROF is None:
        return fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    rc = fn()
    e.record()
    _PROF.append((name, flops, nbytes, s, e))
    return rc


def _tname(t):
    return "float" if t.dtype == torch.float32 else "bf16"


def _igemm_name(lib, dcode, d):
    buf = CT.create_string_buffer(160)
    lib.ffn_igemm_kernel_name(dcode, CT.byref(d), buf, 160)
    return buf.value.decode()


# ---------------------------------------------------------------------------------------------------------------
# weight packing (host side, once at load time)
# ---------------------------------------------------------------------------------------------------------------
_GN_CHUNK_MB = float(os.environ.get("FFN_GN_CHUNK_MB", "0"))           # GroupNorm row chunks (MiB of input per chunk; 0 = whole batch at once)
_GN_RAW = os.environ.get("FFN_GN_RAW", "1") != "0"                     # ResBlock norm1 also writes the pair rows of its input for the 1x1 shortcut GEMM (0: separate ffn_split_pair pass)
_KV64 = os.environ.get("FFN_KV64", "1") != "0"                         # self-attention K / V^T projections write the pre-split images themselves (0: fp32 + ffn_attn_presplit)
_ATTN_PRESPLIT = os.environ.get("FFN_ATTN_PRESPLIT", "1") != "0"      # split-bf16 self attention: pre-split K / V^T once per call (0: split inside the kernel's key loop)


def _mark_x3(t, order=1):
    t._ffn_x3 = order         # linear() / conv3x3() recognise a split-bf16 weight by this tag and run the FFN_BF16X3 path
    return t                  # (1 = planes [hi | lo | hi] over the whole K / tap, 2 = blocked: 128-byte blocks [hi(32) | lo(32)] per 32 elements of K)


def is_x3(w):
    return bool(getattr(w, "_ffn_x3", 0))


def split_hi_lo(w):
    """fp32 -> (hi, lo) bf16 with hi = bf16(w) (RNE), lo = bf16(w - hi): the operand form of the FFN_BF16X3 GEMMs"""
    w = w.float()
    hi = w.to(torch.bfloat16)
    return hi, (w - hi.float()).to(torch.bfloat16)


def pack_linear(w, dtype, x3=False):
    """[N, K] -> [N, Kpad] zero padded, contiguous, dtype.  x3 (split-bf16, include/freefine_hip.h FFN_BF16X3): K % 32 == 0 -> the BLOCKED pair
    form of every row, bf16 [N, 2K] = 128-byte blocks [W_hi(32) | W_lo(32)] (layout 2: what the ping-pong tile's split-bf16 core streams);
    otherwise the planes bf16 [N, 3K] = [W_hi | W_lo | W_hi] (layout 1), the virtual contraction run against the activation's [A_hi | A_hi | A_lo]."""
    n, k = w.shape
    if x3:
        assert k % 8 == 0, f"split-bf16 weights need K % 8 == 0 (K={k})"
        hi, lo = split_hi_lo(w)
        if k % 32 == 0:
            return _mark_x3(torch.stack([hi.reshape(n, k // 32, 32), lo.reshape(n, k // 32, 32)], dim=2).reshape(n, 2 * k).contiguous(), 2)
        return _mark_x3(torch.cat([hi, lo, hi], dim=1).contiguous(), 1)
    ks = kstage(dtype)
    kpad = (k + ks - 1) // ks * ks
    out = torch.zeros(n, kpad, dtype=dtype, device=w.device)
    out[:, :k] = w.to(dtype)
    return out


def pack_conv3x3(w, dtype, cin_pad=None, x3=False):
    """[Cout, Cin, 3, 3] -> [Cout, Kpad], k = (ky*3+kx)*Cin_p + ci  (x3: per tap the blocked pair form of its Cin_p columns, Cin_p % 32 == 0;
    else k' = ((ky*3+kx)*3 + seg)*Cin_p + ci, seg = hi, lo, hi)."""
    cout, cin = w.shape[:2]
    cp = cin_pad or cin
    w2 = torch.zeros(cout, 3, 3, cp, dtype=torch.float32, device=w.device)
    w2[..., :cin] = w.permute(0, 2, 3, 1).float()
    if x3:
        assert cp % 8 == 0, f"split-bf16 conv weights need Cin % 8 == 0 (Cin={cp})"
        hi, lo = split_hi_lo(w2.reshape(cout, 9, cp))
        if cp % 32 == 0:
            return _mark_x3(torch.stack([hi.reshape(cout, 9, cp // 32, 32), lo.reshape(cout, 9, cp // 32, 32)], dim=3).reshape(cout, 18 * cp).contiguous(), 2)
        return _mark_x3(torch.cat([hi, lo, hi], dim=2).reshape(cout, 27 * cp).contiguous(), 1)
    return pack_linear(w2.reshape(cout, 9 * cp), dtype)


def pack_conv3x3_n4(w):
    """[4, Cin, 3, 3] -> fp32 [4, 9 * Cin] in (ky, kx, ci) order: the weights of ffn_conv3x3_n4 (the UNet's conv_out on the vector ALU)"""
    assert w.shape[0] == 4 and w.shape[2:] == (3, 3)
    return w.float().permute(0, 2, 3, 1).reshape(4, -1).contiguous()


def conv3x3_n4_eligible(cout, cin):
    return cout == 4 and cin % 16 == 0 and cin <= 448 and os.environ.get("FFN_CONV_N4", "1") != "0"


def conv3x3_n4(x, w4, bias, B, H, W, Cin):
    """3x3 / stride 1 / pad 1 convolution with four output channels as a direct fp32 convolution (ffn_conv3x3_n4): x [B, H*W, Cin] fp32 or
    bf16, w4 from pack_conv3x3_n4; returns fp32 [B, H*W, 4]"""
    lib = L.load()
    assert x.is_contiguous() and x.shape[-1] == Cin and x.dtype in (torch.float32, torch.bfloat16)
    out = torch.empty(B, H * W, 4, dtype=torch.float32, device=x.device)
    L.check(_timed("conv3x3_n4_kernel", 2.0 * B * H * W * 4 * 9 * Cin, x.element_size() * x.numel() + 16.0 * B * H * W,
                   lambda: lib.ffn_conv3x3_n4(_stream(), _dt(x), x.data_ptr(), w4.data_ptr(), _p(bias), out.data_ptr(), B, H, W, Cin)), "ffn_conv3x3_n4")
    return out


def up2x_eligible(cin, cout, rows):
    """does the sub-pixel form of `nearest-2x upsample -> 3x3 conv` apply?  (bf16 ping-pong tiles: whole 64-channel chunks, N a tile multiple,
    at least one 192-row tile of LOW-resolution pixels)"""
    return cin % 64 == 0 and (cout % 256 == 0 or cout % 320 == 0) and rows >= 192


def pack_conv3x3_up2x(w, dtype, x3=False):
    """[Cout, Cin, 3, 3] of a 3x3 conv that follows a nearest-2x upsample -> [4 Cout, 4 Cin] weights of its four sub-pixel classes.
    Output pixel (2y + a, 2x + b) sees the upsampled rows 2y + a - 1 .. 2y + a + 1 = low-resolution rows {y - 1, y, y} (a = 0) or {y, y, y + 1}
    (a = 1): a 2x2 window starting at (y + a - 1, x + b - 1) whose taps are sums of the 3x3 taps that coincide -- 4/9 of the multiplies.
    Rows [c Cout, (c + 1) Cout) hold class c = 2a + b, k = (dy*2 + dx) Cin + ci.  The sums are taken in fp32 and rounded once.
    x3 (split-bf16 mode, Cin % 32 == 0): the blocked pair form of pack_conv3x3 over the four taps."""
    assert dtype == torch.bfloat16 or x3
    cout, cin = w.shape[:2]
    wf = w.float()
    groups = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}           # parity -> 3x3 tap indices feeding window position 0 / 1
    out = torch.empty(4, cout, 2, 2, cin, dtype=torch.float32, device=w.device)
    for a in (0, 1):
        for b in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    acc = torch.zeros(cout, cin, dtype=torch.float32, device=w.device)
                    for ky in groups[a][dy]:
                        for kx in groups[b][dx]:
                            acc += wf[:, :, ky, kx]
                    out[2 * a + b, :, dy, dx] = acc
    if x3:
        assert cin % 32 == 0
        hi, lo = split_hi_lo(out.reshape(4 * cout, 4, cin))
        return _mark_x3(torch.stack([hi.reshape(4 * cout, 4, cin // 32, 32), lo.reshape(4 * cout, 4, cin // 32, 32)], dim=3).reshape(4 * cout, 8 * cin).contiguous(), 2)
    return pack_linear(out.reshape(4 * cout, 4 * cin), dtype)


def conv3x3_up2x(x, w4, bias, B, Hin, Win, Cin, *, out=None):
    """nearest-2x upsample followed by the 3x3 conv (pad 1) packed by pack_conv3x3_up2x, evaluated at LOW resolution: four 2x2 convolutions
    (conv = 2 of ffn_igemm, one per output-pixel parity class) into one [B, Hin*Win, 4 Cout] buffer + a pixel shuffle (layout plumbing).
    x: [B, Hin*Win, Cin] bf16 (split-bf16 weights: fp32, or its pair rows); returns [B, 4*Hin*Win, Cout] (written into `out`, which may be a
    column view of a wider buffer)."""
    lib = L.load()
    cout = w4.shape[0] // 4
    x3 = is_x3(w4)
    # the ping-pong kernel addresses its operands through 31-bit buffer resources: batches whose [B, HW, 4 Cout] result (or input) would pass
    # 1 GiB go in slices of whole images (the VAE decoder at 256 x 256 x 512 channels)
    per_img = Hin * Win * max(4 * cout * (4 if x3 else 2), Cin * (4 if x3 else 2))
    nb = max(1, ((1 << 30) - 1) // per_img)
    if B > nb:
        if out is None:
            out = torch.empty(B, 4 * Hin * Win, cout, dtype=torch.float32 if x3 else x.dtype, device=x.device)
        for b0 in range(0, B, nb):
            b1 = min(B, b0 + nb)
            conv3x3_up2x(x[b0:b1], w4, bias, b1 - b0, Hin, Win, Cin, out=out[b0:b1])
        return out
    if x3:                                          # split-bf16 mode: fp32 activations as pair rows, fp32 result
        xa = x if pair_width(x) is not None else split_pair(x, Cin)
        assert pair_width(xa) == Cin
        odt, dcode = torch.float32, L.FFN_BF16X3
    else:
        assert x.dtype == torch.bfloat16 and x.is_contiguous()
        xa, odt, dcode = x, x.dtype, L.FFN_BF16
    assert w4.shape[1] >= (8 if x3 else 4) * Cin
    y = torch.empty(B, Hin * Win, 4 * cout, dtype=odt, device=x.device)
    for c in range(4):
        a, b = c >> 1, c & 1
        d = L.IgemmDesc()
        d.A, d.W = xa.data_ptr(), w4.data_ptr() + c * cout * w4.stride(0) * w4.element_size()
        d.bias, d.rowbias, d.residual = _p(bias), None, None
        d.M, d.N, d.K, d.Kpad = B * Hin * Win, cout, 4 * Cin, w4.stride(0)
        d.lda, d.a_lo, d.x3 = (2 * Cin, 32, 2) if x3 else (Cin, 0, 0)
        d.rows_per_batch, d.ldrb = Hin * Win, 0
        d.out, d.ldo, d.ldr = y.data_ptr() + c * cout * y.element_size(), 4 * cout, 0
        d.Hin, d.Win, d.Cin, d.Hout, d.Wout = Hin, Win, Cin, Hin, Win
        d.stride, d.pad, d.upsample = 1, 2 * (1 - a) + (1 - b), 0
        d.flags, d.alpha, d.conv = 0, 1.0, 2
        d.splitk, d.ws, d.ws_bytes = 0, _workspace(x.device).data_ptr(), WS_BYTES
        if _PROF is None:
            L.check(lib.ffn_igemm(_stream(), dcode, CT.byref(d)), "ffn_igemm(conv 2x2)")
        else:
            L.check(_timed(_igemm_name(lib, dcode, d), 2.0 * d.M * cout * 4 * Cin, y.element_size() * (x.numel() + cout * 4 * Cin + d.M * cout),
                           lambda: lib.ffn_igemm(_stream(), dcode, CT.byref(d))), "ffn_igemm(conv 2x2)")
    if out is None:
        out = torch.empty(B, 4 * Hin * Win, cout, dtype=odt, device=x.device)
    # pixel shuffle: class (a, b) of low-resolution pixel (y, x) is output pixel (2y + a, 2x + b)
    out.view(B, Hin, 2, Win, 2, cout).copy_(y.view(B, Hin, Win, 2, 2, cout).permute(0, 1, 3, 2, 4, 5))
    return out


F8_ACT_SCALE = 16.0         # power-of-two scale of fp8 activations: SiLU(GroupNorm(x)) * 16 saturates at 28, far above its range


def pack_conv3x3_f8(w, cin_pad=None):
    """[Cout, Cin, 3, 3] -> e4m3 bytes [Cout, 9 * Cp] (uint8 storage), k = (ky*3+kx)*Cp + ci, Cp = Cin padded to a multiple of 128 (whole
    K tiles of the fp8 ping-pong conv), scaled by a per-tensor power of two so that max |w| lands in [224, 448).  The tensor carries
    `_ffn_f8 = (Cp, alpha)`, alpha = 1 / (weight scale * F8_ACT_SCALE) = what the GEMM epilogue multiplies by (FFN_FP8, freefine_hip.h)."""
    cout, cin = w.shape[:2]
    cp = cin_pad or (cin + 127) // 128 * 128
    w2 = torch.zeros(cout, 3, 3, cp, dtype=torch.float32, device=w.device)
    w2[..., :cin] = w.permute(0, 2, 3, 1).float()
    amax = float(w2.abs().max())
    k = math.floor(math.log2(448.0 / amax)) if amax > 0 else 0
    ws = 2.0 ** k
    q = (w2 * ws).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).reshape(cout, 9 * cp).contiguous().view(torch.uint8)
    q._ffn_f8 = (cp, 1.0 / (ws * F8_ACT_SCALE))
    return q


def is_f8(w):
    return getattr(w, "_ffn_f8", None) is not None


def groupnorm_f8(x, gamma, beta, G, eps, Cp, silu=False, ws=None):
    """bf16 x [B, HW, C] -> e4m3 bytes [B, HW, Cp] = act(GroupNorm(x)) * F8_ACT_SCALE, channels >= C zero: the A operand of an fp8 conv"""
    lib = L.load()
    B, HW, Cc = x.shape
    assert x.dtype == torch.bfloat16
    out = torch.empty(B, HW, Cp, dtype=torch.uint8, device=x.device)
    out._ffn_f8_act = Cc
    partial, scale, shift = ws if ws is not None else gn_workspace(B, HW, Cc, x.device)
    call = lambda: lib.ffn_groupnorm_f8(_stream(), x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), B, HW, Cc, Cp, G, eps,
                                        L.NORM_SILU if silu else 0, F8_ACT_SCALE, partial.data_ptr(), scale.data_ptr(), shift.data_ptr())
    L.check(_timed("gn_partial+gn_finalize+gn_apply_f8", 0.0, x.numel() * 2.0 + out.numel(), call), "ffn_groupnorm_f8")
    return out


def pack_geglu(w, b, dtype, x3=False):
    """GEGLU proj [2F, K] (+bias [2F]): interleave 16-row blocks hidden/gate so one lane holds both halves."""
    f2, k = w.shape
    f = f2 // 2
    assert f % 16 == 0
    idx = torch.arange(f, device=w.device).reshape(f // 16, 16)
    order = torch.stack([idx, idx + f], dim=1).reshape(-1)  # [h0..h15, g0..g15, h16.., ...]
    wp = pack_linear(w[order], dtype, x3)
    bp = None if b is None else b[order].float().contiguous()
    return wp, bp


def _mark_pair(t, C):
    t._ffn_pair = C            # bf16 [..., 2C]: the pair form of rows of C fp32 values (C % 32 == 0: 128-byte blocks [hi(32) | lo(32)]; else planes): the A operand of an FFN_BF16X3 GEMM
    return t


def pair_width(x):
    """C if x is a split-bf16 pair tensor [..., 2C], else None"""
    return getattr(x, "_ffn_pair", None)


def split_pair(x, K=None):
    """fp32 [..., ld] (first K columns) -> bf16 PAIR rows [..., 2K] (ffn_split_pair; layout: include/freefine_hip.h FFN_BF16X3): the A operand of an FFN_BF16X3 GEMM"""
    lib = L.load()
    assert x.dtype == torch.float32 and x.stride(-1) == 1
    K = K if K is not None else x.shape[-1]
    rows = x.numel() // x.shape[-1]
    ld = x.stride(-2) if x.ndim > 1 else x.shape[-1]
    out = torch.empty(*x.shape[:-1], 2 * K, dtype=torch.bfloat16, device=x.device)
    L.check(_timed("split_pair_kernel", 0.0, 8.0 * rows * K, lambda: lib.ffn_split_pair(_stream(), x.data_ptr(), out.data_ptr(), rows, K, ld)),
            "ffn_split_pair")
    return _mark_pair(out, K)


# ---------------------------------------------------------------------------------------------------------------
# igemm
# ---------------------------------------------------------------------------------------------------------------
_WS = {}
WS_BYTES = 96 << 20


def _workspace(device):
    """fp32 scratch for split-K partial slabs, one per (device, stream): reuse is ordered by the stream"""
    key = (device, torch.cuda.current_stream().cuda_stream)     # one buffer per stream: launches on different streams may overlap
    ws = _WS.get(key)
    if ws is None:
        ws = _WS[key] = torch.empty(WS_BYTES // 4, dtype=torch.float32, device=device)
    return ws

def linear(x, w, bias=None, *, K=None, out=None, residual=None, rowbias=None, rows_per_batch=None, silu=False,
           geglu=False, out_f32=False, transposed_ld=None, alpha=1.0, splitk=0, out_pair=False, gelu=False, relu=False, kv64_from=None):
    """out = x @ w[:, :K]^T (+bias ...).  x: [..., K] contiguous rows (M = prod of leading dims).
    kv64_from (split-bf16 only): the projection writes the attention kernels' pre-split K / V^T images itself (FFN_IG_OUT_KV64): row-major output --
    columns >= kv64_from of every row as [hi(64) | lo(64)] blocks per head in the bytes of their fp32 values; transposed output (any value, use 0) --
    every run of 64 positions of a row as one such block.  The caller hands such buffers to ops.attention(kv_images=True)."""
    lib = L.load()
    K = K if K is not None else x.shape[-1]
    M = x.numel() // x.shape[-1]
    N = w.shape[0]
    d = L.IgemmDesc()
    x3 = is_x3(w)
    if x3 and pair_width(x) is not None:            # the producer (norm / GEGLU / attention) already wrote the [hi | lo] pair rows
        K = pair_width(x)
        M = x.numel() // x.shape[-1]
        xa = x
    else:
        xa = split_pair(x, K) if x3 else x          # split-bf16: fp32 activations -> [hi | lo] bf16 rows; results / residual stay fp32
    dcode = L.FFN_BF16X3 if x3 else _dt(x)
    d.A, d.W = xa.data_ptr(), w.data_ptr()
    d.bias, d.rowbias, d.residual = _p(bias), _p(rowbias), _p(residual)
    d.M, d.N, d.K, d.Kpad = M, N, K, w.stride(0)    # Kpad = row stride of W (an activation view can serve as W)
    d.lda = xa.stride(-2) if xa.ndim > 1 else xa.shape[-1]
    d.x3 = int(w._ffn_x3) if x3 else 0
    d.a_lo = (32 if d.x3 == 2 else K) if x3 else 0          # blocked pair rows (K % 32 == 0) / planes
    odt = torch.float32 if x3 else x.dtype          # element type of out / residual
    d.rows_per_batch = rows_per_batch or M
    d.ldrb = rowbias.stride(0) if rowbias is not None else 0
    flags = 0
    n_out = N
    if silu:
        flags |= L.IG_OUT_SILU
    if gelu:
        flags |= L.IG_OUT_GELU
    if relu:
        flags |= L.IG_OUT_RELU
    if geglu:
        flags |= L.IG_GEGLU
        n_out = N // 2
    if out_f32:
        flags |= L.IG_OUT_F32
    if transposed_ld is not None:
        flags |= L.IG_OUT_TRANSPOSED
        nb = M // d.rows_per_batch
        if out is None:
            # columns past the rows of a batch are padding the attention kernels may multiply by P = 0: they must be finite, so a
            # padded buffer is zero-filled; without padding (every UNet level: S % 8 == 0) the GEMM writes every element
            alloc = torch.zeros if transposed_ld > d.rows_per_batch else torch.empty
            out = alloc(nb, N, transposed_ld, dtype=odt, device=x.device)
        d.ldo = transposed_ld
    elif out_pair:                                  # split-bf16 only: the result in the pair form the next GEMM reads
        assert x3 and out is None                   # (a residual is added in fp32 before the split)
        flags |= L.IG_OUT_PAIR
        out = _mark_pair(torch.empty(*x.shape[:-1], 2 * n_out, dtype=torch.bfloat16, device=x.device), n_out)
        d.ldo = 2 * n_out
    else:
        if out is None:
            out = torch.empty(*x.shape[:-1], n_out, dtype=torch.float32 if out_f32 else odt, device=x.device)
        d.ldo = out.stride(-2) if out.ndim > 1 else n_out          # a column view of a wider buffer (cat_dst) keeps the buffer's row stride
    d.ldr = residual.shape[-1] if residual is not None else 0
    d.out = out.data_ptr()
    if kv64_from is not None:
        assert x3 and residual is None and rowbias is None and not (silu or gelu or relu or geglu or out_pair)
        flags |= L.IG_OUT_KV64
        d.kv64_from = int(kv64_from)
        splitk = 1
    d.flags, d.alpha, d.conv = flags, alpha, 0
    d.splitk, d.ws, d.ws_bytes = splitk, _workspace(x.device).data_ptr(), WS_BYTES
    if _PROF is None:
        L.check(lib.ffn_igemm(_stream(), dcode, CT.byref(d)), "ffn_igemm")
    else:
        esz = x.element_size()
        L.check(_timed(_igemm_name(lib, dcode, d), 2.0 * M * N * K, esz * (M * K + N * K + M * n_out),
                       lambda: lib.ffn_igemm(_stream(), dcode, CT.byref(d))), "ffn_igemm")
    return out


def conv3x3(x, w, bias, B, Hin, Win, Cin, *, stride=1, pad=1, upsample=False, out=None, residual=None, rowbias=None,
            rowbias_ld=None, out_f32=False, Hout=None, Wout=None, splitk=0, relu=False):
    """x: [B, Hin*Win, Cin] NHWC; w: packed [Cout, Kpad]; returns [B, Hout*Wout, Cout]."""
    lib = L.load()
    He, We = (Hin * 2, Win * 2) if upsample else (Hin, Win)
    if Hout is None:
        Hout = (He + 2 * pad - 3) // stride + 1
        Wout = (We + 2 * pad - 3) // stride + 1
    N = w.shape[0]
    d = L.IgemmDesc()
    x3 = is_x3(w)
    f8 = is_f8(w)
    if f8:                                          # fp8 convolution: x is the e4m3 tensor groupnorm_f8 wrote, [B, HW, Cp]
        assert x.dtype == torch.uint8 and x.shape[-1] == w._ffn_f8[0] and out_f32 is False
        Cin = x.shape[-1]
        xa = x
    elif x3 and pair_width(x) is not None:
        assert pair_width(x) == Cin
        xa = x
    else:
        xa = split_pair(x, Cin) if x3 else x        # split-bf16: every pixel becomes [hi(Cin) | lo(Cin)]
    dcode = L.FFN_FP8 if f8 else (L.FFN_BF16X3 if x3 else _dt(x))
    odt = torch.float32 if x3 else (torch.bfloat16 if f8 else x.dtype)
    d.A, d.W = xa.data_ptr(), w.data_ptr()
    d.bias, d.rowbias, d.residual = _p(bias), _p(rowbias), _p(residual)
    d.M, d.N, d.K, d.Kpad = B * Hout * Wout, N, 9 * Cin, w.shape[1]
    d.lda = 2 * Cin if x3 else Cin
    d.x3 = int(w._ffn_x3) if x3 else 0
    d.a_lo = (32 if d.x3 == 2 else Cin) if x3 else 0
    d.rows_per_batch = Hout * Wout
    d.ldrb = (rowbias_ld or rowbias.stride(0)) if rowbias is not None else 0
    if out is None:
        out = torch.empty(B, Hout * Wout, N, dtype=torch.float32 if out_f32 else odt, device=x.device)
    d.out, d.ldo = out.data_ptr(), out.stride(-2)                   # (a column view of a wider buffer keeps the buffer's row stride)
    d.ldr = residual.shape[-1] if residual is not None else 0
    d.Hin, d.Win, d.Cin, d.Hout, d.Wout = Hin, Win, Cin, Hout, Wout
    d.stride, d.pad, d.upsample = stride, pad, 1 if upsample else 0
    d.flags, d.alpha, d.conv = (L.IG_OUT_F32 if out_f32 else 0) | (L.IG_OUT_RELU if relu else 0), (w._ffn_f8[1] if f8 else 1.0), 1
    d.splitk, d.ws, d.ws_bytes = splitk, _workspace(x.device).data_ptr(), WS_BYTES
    if _PROF is None:
        L.check(lib.ffn_igemm(_stream(), dcode, CT.byref(d)), "ffn_igemm(conv)")
    else:
        esz = x.element_size()
        k_real = 9 * getattr(x, "_ffn_f8_act", Cin)       # fp8: the padded channels are zeros, not work
        L.check(_timed(_igemm_name(lib, dcode, d), 2.0 * d.M * N * k_real, esz * (x.numel() + N * d.K + d.M * N),
                       lambda: lib.ffn_igemm(_stream(), dcode, CT.byref(d))), "ffn_igemm(conv)")
    return out


# ---------------------------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------------------------
class AttnEntrySpec:
    """One (pass, output-row) term of ffn_attn (see include/freefine_hip.h)."""
    __slots__ = ("q_row", "kv_row", "w_const", "w_slope", "wq", "kmask", "qsel", "flags", "hr_row")

    def __init__(self, q_row, kv_row, w_const=1.0, w_slope=0.0, wq=None, kmask=None, qsel=None, flags=0, hr_row=None):
        self.q_row, self.kv_row, self.w_const, self.w_slope = q_row, kv_row, w_const, w_slope
        self.wq, self.kmask, self.qsel, self.flags, self.hr_row = wq, kmask, qsel, flags, hr_row

    def remap(self, row_map, logical_row):
        """the same term for a row-deduplicated batch: Q / KV rows through logical->physical, head rule pinned to the logical row"""
        return AttnEntrySpec(row_map[self.q_row], row_map[self.kv_row], self.w_const, self.w_slope, self.wq, self.kmask, self.qsel,
                             self.flags, logical_row if self.hr_row is None else self.hr_row)

    def pinned(self, logical_row):
        """the same term with the tiled-head rule pinned to `logical_row` (unless it already is pinned)"""
        return self if self.hr_row is not None else AttnEntrySpec(self.q_row, self.kv_row, self.w_const, self.w_slope, self.wq, self.kmask, self.qsel,
                                                                  self.flags, logical_row)

    def renumber(self, ren, kv_ren=None):
        """the same term inside a launch that holds only a SUBSET of the physical rows (`ren`: physical row -> row of that launch;
        `kv_ren`: the same for the K / V row where the launch's K / V tensors are not row-aligned with its Q tensor)"""
        kv_ren = ren if kv_ren is None else kv_ren
        if self.q_row not in ren or self.kv_row not in kv_ren:
            raise ValueError(f"attention term (q row {self.q_row}, kv row {self.kv_row}) reaches outside the rows of this phase {sorted(ren)} / {sorted(kv_ren)}")
        return AttnEntrySpec(ren[self.q_row], kv_ren[self.kv_row], self.w_const, self.w_slope, self.wq, self.kmask, self.qsel, self.flags, self.hr_row)

    def shifted(self, base, logical_row, kv_base=None):
        """the same term inside an image-batched launch: this image's physical rows start at `base` (its K / V rows at `kv_base`
        where they differ: cross attention of the composition hook, whose text batch has more rows than the latent batch); the
        tiled-head rule (attention.py:859 vs 761) keeps using the row index the image would have had in its own batch"""
        return AttnEntrySpec(self.q_row + base, self.kv_row + (base if kv_base is None else kv_base), self.w_const, self.w_slope, self.wq,
                             self.kmask, self.qsel, self.flags, logical_row if self.hr_row is None else self.hr_row)


def kv_images_ok(Dh, S, Sk, passes=None, nbytes=0):
    """split-bf16 self attention whose K / V^T projections may write the attention kernel's pre-split images themselves (linear(kv64_from=...),
    attention(kv_images=True)): the shapes attn_x3w_kernel / attn_x3p_kernel take, a plan without degenerate uniform-softmax entries (those run on
    attn_x3_kernel, which reads fp32 K / V^T), and buffers the kernels' 32-bit byte offsets reach (nbytes = the larger of the K and V^T buffers)"""
    if not (_ATTN_PRESPLIT and _KV64 and Dh == 64 and Sk % 64 == 0 and S >= 128 and nbytes < 2 ** 31 - 65536):
        return False
    for rows in passes or ():
        for sp in rows:
            if sp is not None and (sp.w_const != 0.0 or sp.w_slope != 0.0) and sp.kmask is not None and (sp.flags & (L.ATT_UNIFORM_SEL1 | L.ATT_UNIFORM_SEL0)):
                return False
    return True


_ROWSPLIT = os.environ.get("FFN_ATT_ROWSPLIT", "1") != "0"
_CUS = {}


def attn_row_split(Bo, wg_per_row, maxb, cus):
    """output rows per launch of a self-attention call on the one-workgroup-per-CU kernels (attn_x3w / attn_x3p / attn_pp: a (row, head, 256-query block)
    workgroup each, `cus` of them resident at a time): the uniform chunk n <= maxb that minimises the number of resident ROUNDS
    sum_i ceil(n_i * wg_per_row / cus) over the call's launches (launches of one stream do not overlap: every launch pays its own partly filled last
    round); ties go to the larger chunk (fewer launches).  72 rows: S = 4096, 5 heads (80 workgroups per row) 16 + 16 + 16 + 16 + 8 = 23 rounds (as
    before); S = 1024, 10 heads (40 per row) 6 x 12 = 12 rounds instead of 14; S = 256, 20 heads (20 per row) 6 x 12 = 6 rounds instead of 9."""
    best_n, best = min(maxb, Bo), None
    for n in range(min(maxb, Bo), 0, -1):
        full, rest = divmod(Bo, n)
        rounds = full * -(-n * wg_per_row // cus) + (-(-rest * wg_per_row // cus) if rest else 0)
        if best is None or rounds < best:
            best_n, best = n, rounds
    return best_n


def attention(q, k, vt, heads, scale, passes=None, *, Sk=None, out=None, w_dev=None, Bo=None, C=None, x3=False, out_pair=False, kv_images=False):
    """q: [Bq,S,C]; k: [Bk,Sk,C]; vt: [Bk,C,ldvt] (V transposed).  passes: list (per pass) of lists (per output
    row) of AttnEntrySpec or None (= skipped).  passes=None -> plain attention, row b uses its own K/V.
    More than FFN_ATT_MAXB output rows are issued as several launches over row ranges (entries name absolute Q/KV rows).
    kv_images (split-bf16 self attention): k / vt already ARE the pre-split images (written by their projections with kv64_from: fp32-typed tensors of
    the shapes above whose bytes hold [hi(64) | lo(64)] blocks); the launches run with kv_pair = 1 and no ffn_attn_presplit pass."""
    lib = L.load()
    Bq, S, _ = q.shape
    Cq = C if C is not None else q.shape[2]          # q / k may be column views of a wider [B,S,ld] buffer
    Sk = Sk if Sk is not None else k.shape[1]
    Dh = Cq // heads
    if passes is None:
        passes = [[AttnEntrySpec(b, b) for b in range(Bq)]]
    Bo = Bo if Bo is not None else len(passes[0])
    out_pair = bool(out_pair and x3 and q.dtype == torch.float32 and Dh <= 64 and out is None and Cq % 8 == 0)
    if out_pair:                                     # the result as pair rows for the to_out projection's split-bf16 GEMM
        out = _mark_pair(torch.empty(Bo, S, 2 * Cq, dtype=torch.bfloat16, device=q.device), Cq)
    if out is None:
        out = torch.empty(Bo, S, Cq, dtype=q.dtype, device=q.device)
    for rows in passes:
        assert len(rows) == Bo
    esz = q.element_size()
    dcode = L.FFN_BF16X3 if (x3 and q.dtype == torch.float32) else _dt(q)      # split-bf16 arithmetic on fp32 operands
    descs = []
    chunk = L.ATT_MAXB
    if _ROWSPLIT and Bo > 1 and Dh == 64 and S >= 128 and Sk % 64 == 0 and q.is_cuda:      # the 256-queries-per-workgroup, one-workgroup-per-CU kernels
        di = q.device.index if q.device.index is not None else torch.cuda.current_device()
        if di not in _CUS:
            _CUS[di] = torch.cuda.get_device_properties(di).multi_processor_count
        chunk = attn_row_split(Bo, heads * -(-S // 256), L.ATT_MAXB, _CUS[di])
    for b0 in range(0, Bo, chunk):
        nb = min(chunk, Bo - b0)
        d = L.AttnDesc()
        d.q, d.k, d.vt, d.w_dev = q.data_ptr(), k.data_ptr(), vt.data_ptr(), _p(w_dev)
        d.out = out.data_ptr() + b0 * S * Cq * esz       # (pair rows: 2 Cq bf16 = Cq fp32 worth of bytes)
        d.Bo, d.S, d.Sk, d.heads, d.D = nb, S, Sk, heads, Dh
        d.ldq, d.ldk, d.ldvt, d.ldo = q.stride(1), k.stride(1), vt.stride(1), (2 * Cq if out_pair else Cq)
        d.out_pair = 1 if out_pair else 0
        d.scale, d.npass = scale, len(passes)
        d.kv_pair = 1 if kv_images else 0
        for p, rows in enumerate(passes):
            for b in range(nb):
                sp = rows[b0 + b]
                e = d.e[p * L.ATT_MAXB + b]
                if sp is None:
                    e.w_const = e.w_slope = 0.0
                    continue
                e.q_row, e.kv_row, e.w_const, e.w_slope = sp.q_row, sp.kv_row, sp.w_const, sp.w_slope
                e.wq, e.kmask, e.qsel, e.flags = _p(sp.wq), _p(sp.kmask), _p(sp.qsel), sp.flags
                # the tiled-head rule defaults to the OUTPUT row index, which a row-range launch shifts by b0: pin it
                hr = sp.hr_row if sp.hr_row is not None else (b0 + b if b0 else None)
                e.hr_row = 0 if hr is None else hr + 1
        descs.append((b0, nb, d))
    # split-bf16 self attention on the ping-pong kernel: K / V^T are split ONCE per call into the bf16 images the kernel stages by LDS-DMA
    # (ffn_attn_presplit; the kernel's 16 query-block workgroups per (row, head) otherwise each split the whole K / V^T in their key loops)
    # (the images are addressed with 32-bit byte offsets: K / V^T beyond 2 GiB keep the in-kernel split instead of failing in ffn_attn)
    if kv_images:
        assert dcode == L.FFN_BF16X3 and kv_images_ok(Dh, S, Sk)
    elif dcode == L.FFN_BF16X3 and _ATTN_PRESPLIT and Dh == 64 and Sk % 64 == 0 and S >= 128 and k.stride(2) == 1 and vt.stride(2) == 1 \
            and k.stride(0) == Sk * k.stride(1) and vt.stride(0) == heads * Dh * vt.stride(1) and k.shape[0] * Sk * heads * 256 < 2 ** 31 - 65536:
        nbuf = CT.create_string_buffer(160)
        names = []
        for _, _, d in descs:
            lib.ffn_attn_kernel_name(dcode, CT.byref(d), nbuf, 160)
            names.append(nbuf.value.decode())
        if all("attn_x3p_kernel" in n_ for n_ in names):
            Bk = k.shape[0]
            kp = torch.empty(Bk, Sk, 2 * heads * Dh, dtype=torch.bfloat16, device=q.device)
            vp = torch.empty(Bk, heads * Dh, 2 * Sk, dtype=torch.bfloat16, device=q.device)
            L.check(_timed("attn_presplit_kernel", 0.0, 16.0 * Bk * Sk * heads * Dh,
                           lambda: lib.ffn_attn_presplit(_stream(), k.data_ptr(), vt.data_ptr(), kp.data_ptr(), vp.data_ptr(), Bk, Sk, heads, k.stride(1), vt.stride(1))),
                    "ffn_attn_presplit")
            for _, _, d in descs:
                d.k, d.vt, d.kv_pair = kp.data_ptr(), vp.data_ptr(), 1
                d.ldk, d.ldvt = heads * Dh, Sk              # compact images: the strides of the fp32 tensors they replace
    for b0, nb, d in descs:
        if _PROF is None:
            L.check(lib.ffn_attn(_stream(), dcode, CT.byref(d)), "ffn_attn")
        else:
            nterms = sum(1 for rows in passes for sp in rows[b0:b0 + nb] if sp is not None and (sp.w_const != 0.0 or sp.w_slope != 0.0))
            nbuf = CT.create_string_buffer(160)
            lib.ffn_attn_kernel_name(dcode, CT.byref(d), nbuf, 160)
            L.check(_timed(nbuf.value.decode(), 4.0 * nterms * S * Sk * Cq,
                           esz * nterms * (S * Cq + 2 * Sk * Cq) + esz * nb * S * Cq,
                           lambda: lib.ffn_attn(_stream(), dcode, CT.byref(d))), "ffn_attn")
    return out


# ---------------------------------------------------------------------------------------------------------------
# norms
# ---------------------------------------------------------------------------------------------------------------
def gn_workspace(B, HW, C, device):
    lib = L.load()
    nchunk = lib.ffn_gn_nchunk(HW)
    return (torch.empty(B * nchunk * 2 * C, dtype=torch.float32, device=device),
            torch.empty(B, C, dtype=torch.float32, device=device), torch.empty(B, C, dtype=torch.float32, device=device))


def groupnorm(x, gamma, beta, G, eps, silu=False, out=None, ws=None, pair=False):
    """x: [B, HW, C].  One fused launch for slices <= 131072 elements per (batch, group), else stats + apply.
    pair (fp32 x): the result as the bf16 pair rows [B, HW, 2C] an FFN_BF16X3 GEMM reads."""
    lib = L.load()
    B, HW, Cc = x.shape
    if pair:
        assert x.dtype == torch.float32 and out is None
        out = _mark_pair(torch.empty(B, HW, 2 * Cc, dtype=torch.bfloat16, device=x.device), Cc)
    if out is None:
        out = torch.empty_like(x)
    if lib.ffn_gn_fused(B, HW, Cc, G):
        partial = scale = shift = None
    else:
        partial, scale, shift = ws if ws is not None else gn_workspace(B, HW, Cc, x.device)
    fl = (L.NORM_SILU if silu else 0) | (L.NORM_OUT_PAIR if pair else 0)
    # Row chunks (three-launch form only): statistics and apply of a chunk run back to back, so that the apply pass finds the chunk's input in the
    # 256 MiB Infinity Cache instead of fetching it from HBM a second time (the three-launch form reads x twice: 12 bytes per element moved for 8).
    nb = B
    if partial is not None and _GN_CHUNK_MB > 0 and x.is_contiguous() and out.is_contiguous():
        row_bytes = HW * Cc * x.element_size()
        nb = max(1, min(B, int(_GN_CHUNK_MB * 2 ** 20) // row_bytes))
        nb = -(-B // -(-B // nb))                    # equal chunks
    def call():
        rc = 0
        for b0 in range(0, B, nb):
            n = min(nb, B - b0)
            rc = lib.ffn_groupnorm(_stream(), _dt(x), x.data_ptr() + b0 * HW * Cc * x.element_size(), out.data_ptr() + b0 * HW * out.shape[-1] * out.element_size(),
                                   gamma.data_ptr(), beta.data_ptr(), n, HW, Cc, G, eps, fl, _p(partial), _p(scale), _p(shift))
            if rc:
                break
        return rc
    if _PROF is None:
        L.check(call(), "ffn_groupnorm")
    else:       # algorithmic bytes: one read + one write (the three-launch form reads x twice: that shows as a lower GB/s)
        name = f"gn_fused_kernel<{_tname(x)}>" if partial is None else f"gn_partial+gn_finalize+gn_apply<{_tname(x)}>"
        L.check(_timed(name, 0.0, 2.0 * x.numel() * x.element_size(), call), "ffn_groupnorm")
    return out


def groupnorm_pair_raw(x, gamma, beta, G, eps, silu=False, ws=None):
    """split-bf16 mode, ResBlock with a 1x1 shortcut: (pair rows of act(GroupNorm(x)), pair rows of x itself) from ONE apply pass over x (ffn_groupnorm_pair_raw) --
    the second is what split_pair(x) would give, bit for bit.  Shapes the one-launch fused GroupNorm takes (small tensors) keep that kernel + a split_pair pass."""
    lib = L.load()
    B, HW, Cc = x.shape
    if not _GN_RAW or Cc % 8 != 0 or lib.ffn_gn_fused(B, HW, Cc, G) or not x.is_contiguous():
        return groupnorm(x, gamma, beta, G, eps, silu=silu, pair=True, ws=ws), split_pair(x, Cc)
    assert x.dtype == torch.float32
    y = _mark_pair(torch.empty(B, HW, 2 * Cc, dtype=torch.bfloat16, device=x.device), Cc)
    yr = _mark_pair(torch.empty(B, HW, 2 * Cc, dtype=torch.bfloat16, device=x.device), Cc)
    partial, scale, shift = ws if ws is not None else gn_workspace(B, HW, Cc, x.device)
    call = lambda: lib.ffn_groupnorm_pair_raw(_stream(), x.data_ptr(), y.data_ptr(), yr.data_ptr(), gamma.data_ptr(), beta.data_ptr(), B, HW, Cc, G, eps,
                                              L.NORM_SILU if silu else 0, _p(partial), _p(scale), _p(shift))
    if _PROF is None:
        L.check(call(), "ffn_groupnorm_pair_raw")
    else:       # algorithmic bytes: one read + two writes
        L.check(_timed("gn_partial+gn_finalize+gn_apply<float> (+ raw pair)", 0.0, 3.0 * x.numel() * x.element_size(), call), "ffn_groupnorm_pair_raw")
    return y, yr


def layernorm(x, gamma, beta, eps=1e-5, out=None, pair=False):
    lib = L.load()
    Cc = x.shape[-1]
    M = x.numel() // Cc
    if pair:
        assert x.dtype == torch.float32 and out is None
        out = _mark_pair(torch.empty(*x.shape[:-1], 2 * Cc, dtype=torch.bfloat16, device=x.device), Cc)
        call = lambda: lib.ffn_layernorm_pair(_stream(), x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), M, Cc, eps)
    else:
        if out is None:
            out = torch.empty_like(x)
        call = lambda: lib.ffn_layernorm(_stream(), _dt(x), x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), M, Cc, eps)
    if _PROF is None:
        L.check(call(), "ffn_layernorm")
    else:
        L.check(_timed(f"layernorm_kernel<{_tname(x)}>", 0.0, 2.0 * x.numel() * x.element_size(), call), "ffn_layernorm")
    return out


def softmax_rows(x, scale=1.0, out=None):
    lib = L.load()
    N = x.shape[-1]
    M = x.numel() // N
    if out is None:
        out = torch.empty_like(x)
    L.check(_timed(f"softmax_rows_kernel<{_tname(x)}>", 0.0, 2.0 * x.numel() * x.element_size(),
                   lambda: lib.ffn_softmax_rows(_stream(), _dt(x), x.data_ptr(), out.data_ptr(), M, N, scale)), "ffn_softmax_rows")
    return out


# ---------------------------------------------------------------------------------------------------------------
# scheduler / guidance (fp32 NCHW, like the reference)
# ---------------------------------------------------------------------------------------------------------------
def cfg_masked(eps_u, eps_c, mask_f, cfg, out=None):
    lib = L.load()
    if out is None:
        out = torch.empty_like(eps_u)
    HW = eps_u.shape[-1] * eps_u.shape[-2]
    L.check(_timed("cfg_masked_kernel", 0.0, 12.0 * eps_u.numel(),
                   lambda: lib.ffn_cfg_masked(_stream(), eps_u.data_ptr(), eps_c.data_ptr(), _p(mask_f), float(cfg), out.data_ptr(),
                                              eps_u.numel(), HW)), "ffn_cfg_masked")
    return out


def ddim_inv_step(eps, x, c_bt, c_at, c_an, c_bn, want_pred_x0=False):
    lib = L.load()
    x_next = torch.empty_like(x)
    p0 = torch.empty_like(x) if want_pred_x0 else None
    L.check(_timed("ddim_inv_step_kernel", 0.0, (12.0 + (4.0 if want_pred_x0 else 0.0)) * x.numel(),
                   lambda: lib.ffn_ddim_inv_step(_stream(), eps.data_ptr(), x.data_ptr(), c_bt, c_at, c_an, c_bn, x_next.data_ptr(), _p(p0),
                                                 x.numel())), "ffn_ddim_inv_step")
    return x_next, p0


def ddim_ctrl_step(eps, x, noise, m_f, om_f, c_bt, c_at, c_ap, c_dir, c_dirm, stdv, row_masked, want_pred_x0=False):
    lib = L.load()
    rows = x.shape[0]
    d = L.CtrlStepDesc()
    x_prev = torch.empty_like(x)
    p0 = torch.empty_like(x) if want_pred_x0 else None
    d.eps, d.x, d.noise, d.m, d.om = eps.data_ptr(), x.data_ptr(), _p(noise), m_f.data_ptr(), om_f.data_ptr()
    d.x_prev, d.pred_x0 = x_prev.data_ptr(), _p(p0)
    d.c_bt, d.c_at, d.c_ap, d.c_dir = c_bt, c_at, c_ap, c_dir
    for b in range(rows):
        d.c_dirm[b], d.stdv[b], d.row_masked[b] = c_dirm[b], stdv[b], int(row_masked[b])
    d.rows, d.CHW, d.HW = rows, x[0].numel(), x.shape[-1] * x.shape[-2]
    L.check(_timed("ddim_ctrl_step_kernel", 0.0, 20.0 * x.numel(), lambda: lib.ffn_ddim_ctrl_step(_stream(), CT.byref(d))), "ffn_ddim_ctrl_step")
    return x_prev, p0


# ---------------------------------------------------------------------------------------------------------------
# layout / misc
# ---------------------------------------------------------------------------------------------------------------
def pack_nchw(src, src_rows, CP, dtype, out=None):
    """fp32 NCHW [Bs,Cl,H,W] -> dtype NHWC [len(src_rows), HW, CP]."""
    lib = L.load()
    Bs, Cl, H, W = src.shape
    B = len(src_rows)
    if out is None:
        out = torch.empty(B, H * W, CP, dtype=dtype, device=src.device)
    for b0 in range(0, B, 16):                     # the descriptor names up to 16 source rows per launch
        nb = min(16, B - b0)
        d = L.PackDesc()
        d.src, d.dst = src.data_ptr(), out.data_ptr() + b0 * H * W * CP * out.element_size()
        for i in range(nb):
            d.src_row[i] = src_rows[b0 + i]
        d.B, d.Cl, d.CP, d.HW = nb, Cl, CP, H * W
        L.check(_timed(f"pack_nchw_kernel<{_tname(out)}>", 0.0, nb * H * W * (4.0 * Cl + CP * out.element_size()),
                       lambda: lib.ffn_pack_nchw(_stream(), _dt(out), CT.byref(d))), "ffn_pack_nchw")
    return out


def nhwc_to_nchw_f32(src, C_, H, W, out=None):
    lib = L.load()
    B, HW, ld = src.shape
    if out is None:
        out = torch.empty(B, C_, H, W, dtype=torch.float32, device=src.device)
    L.check(_timed("nhwc_to_nchw_f32_kernel", 0.0, 8.0 * B * HW * C_,
                   lambda: lib.ffn_nhwc_to_nchw_f32(_stream(), src.data_ptr(), out.data_ptr(), B, HW, C_, ld)), "ffn_nhwc_to_nchw_f32")
    return out


def concat(a, b, out=None):
    """[a | b] along the channel axis.  If `a` is the left column view of a cat_dst() buffer (its producer already wrote it in place),
    only b is copied and that buffer is returned."""
    lib = L.load()
    C1, C2 = a.shape[-1], b.shape[-1]
    rows = a.numel() // C1
    base = getattr(a, "_ffn_cat_base", None)
    if out is None and base is not None and base.shape[-1] == C1 + C2:
        L.check(_timed(f"concat_kernel<{_tname(b)}>", 0.0, 2.0 * rows * C2 * b.element_size(),
                       lambda: lib.ffn_concat(_stream(), _dt(b), None, b.data_ptr(), base.data_ptr(), rows, C1, C2)), "ffn_concat")
        return base
    if base is not None:
        a = a.contiguous()
    if out is None:
        out = torch.empty(*a.shape[:-1], C1 + C2, dtype=a.dtype, device=a.device)
    L.check(_timed(f"concat_kernel<{_tname(a)}>", 0.0, 2.0 * rows * (C1 + C2) * a.element_size(),
                   lambda: lib.ffn_concat(_stream(), _dt(a), a.data_ptr(), b.data_ptr(), out.data_ptr(), rows, C1, C2)), "ffn_concat")
    return out


def cat_dst(shape_lead, C1, C2, dtype, device):
    """Destination for a producer whose output will be concatenated with C2 more channels: a [*, C1 + C2] buffer and its left
    [*, C1] column view (pass the view as `out=` to linear / conv3x3; concat(view, skip) then only copies the skip)."""
    base = torch.empty(*shape_lead, C1 + C2, dtype=dtype, device=device)
    view = base[..., :C1]
    view._ffn_cat_base = base
    return view


def timestep_freqs(dim, device, max_period=10000.0, shift=0.0):
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32) / (half - shift)
    return torch.exp(exponent).to(device)


def timestep_embed(t_dev, freq, B, dtype, flip=True, out=None):
    lib = L.load()
    half = freq.numel()
    if out is None:
        out = torch.empty(B, 2 * half, dtype=dtype, device=freq.device)
    L.check(_timed(f"timestep_embed_kernel<{_tname(out)}>", 0.0, float(out.numel() * out.element_size()),
                   lambda: lib.ffn_timestep_embed(_stream(), _dt(out), t_dev.data_ptr(), freq.data_ptr(), out.data_ptr(), B, half,
                                                  1 if flip else 0)), "ffn_timestep_embed")
    return out


def transpose(src, ld_dst=None, out=None):
    """[B,R,C] -> [B,C,ld_dst] (first R columns written)."""
    lib = L.load()
    B, R, Cc = src.shape
    ld_dst = ld_dst or R
    if out is None:
        out = torch.zeros(B, Cc, ld_dst, dtype=src.dtype, device=src.device)
    L.check(_timed(f"transpose_kernel<{_tname(src)}>", 0.0, 2.0 * src.numel() * src.element_size(),
                   lambda: lib.ffn_transpose(_stream(), _dt(src), src.data_ptr(), out.data_ptr(), B, R, Cc, Cc, ld_dst)), "ffn_transpose")
    return out


def cast(src, dtype, out=None):
    lib = L.load()
    if out is None:
        out = torch.empty(src.shape, dtype=dtype, device=src.device)
    L.check(_timed(f"cast_kernel<{_tname(src)}, {_tname(out)}>", 0.0, float(src.numel() * (src.element_size() + out.element_size())),
                   lambda: lib.ffn_cast(_stream(), _dt(src), _dt(out), src.data_ptr(), out.data_ptr(), src.numel())), "ffn_cast")
    return out


def relu(x, out=None):
    """max(x, 0) (fp32 / bf16, numel % 4 == 0)"""
    lib = L.load()
    out = torch.empty_like(x) if out is None else out
    L.check(_timed("eltwise_kernel<relu>", 0.0, 2.0 * x.numel() * x.element_size(),
                   lambda: lib.ffn_eltwise(_stream(), _dt(x), L.ELT_RELU, x.data_ptr(), None, out.data_ptr(), x.numel())), "ffn_eltwise")
    return out


def add(a, b, out=None):
    """a + b (same shape and dtype, contiguous)"""
    lib = L.load()
    assert a.shape == b.shape and a.dtype == b.dtype and a.is_contiguous() and b.is_contiguous()
    out = torch.empty_like(a) if out is None else out
    L.check(_timed("eltwise_kernel<add>", 0.0, 3.0 * a.numel() * a.element_size(),
                   lambda: lib.ffn_eltwise(_stream(), _dt(a), L.ELT_ADD, a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel())), "ffn_eltwise")
    return out


def resize_bilinear(x, B, Hin, Win, Hout, Wout, relu=False):
    """x: [B, Hin*Win, C] NHWC -> [B, Hout*Wout, C]; torch's bilinear interpolation with align_corners=True"""
    lib = L.load()
    C_ = x.shape[-1]
    assert x.is_contiguous() and x.numel() == B * Hin * Win * C_
    out = torch.empty(B, Hout * Wout, C_, dtype=x.dtype, device=x.device)
    L.check(_timed("resize_bilinear_kernel", 0.0, (x.numel() + out.numel()) * x.element_size(),
                   lambda: lib.ffn_resize_bilinear(_stream(), _dt(x), x.data_ptr(), out.data_ptr(), B, Hin, Win, Hout, Wout, C_, 1 if relu else 0)),
            "ffn_resize_bilinear")
    return out


def image_to_nhwc(img_u8, CP, dtype, out=None):
    """uint8 [B,H,W,3] -> dtype [B,HW,CP] in [-1,1]."""
    lib = L.load()
    B, H, W, _ = img_u8.shape
    if out is None:
        out = torch.empty(B, H * W, CP, dtype=dtype, device=img_u8.device)
    L.check(_timed(f"image_to_nhwc_kernel<{_tname(out)}>", 0.0, B * H * W * (3.0 + CP * out.element_size()),
                   lambda: lib.ffn_image_to_nhwc(_stream(), _dt(out), img_u8.data_ptr(), out.data_ptr(), B * H * W, CP)), "ffn_image_to_nhwc")
    return out


def nhwc_to_image(src, H, W, out=None):
    lib = L.load()
    B, HW, ld = src.shape
    if out is None:
        out = torch.empty(B, 3, H, W, dtype=torch.float32, device=src.device)
    L.check(_timed(f"nhwc_to_image_kernel<{_tname(src)}>", 0.0, B * HW * 3.0 * (src.element_size() + 4),
                   lambda: lib.ffn_nhwc_to_image(_stream(), _dt(src), src.data_ptr(), out.data_ptr(), B, HW, ld)), "ffn_nhwc_to_image")
    return out


# ---------------------------------------------------------------------------------------------------------------
# the igemm tuner's table as data (persist across processes / broadcast across ranks)
# ---------------------------------------------------------------------------------------------------------------
def tune_table_export():
    """int32 tensor [n, entry_ints] of the bf16 igemm configurations tuned so far in this process."""
    lib = L.load()
    w = lib.ffn_igemm_tune_entry_ints()
    n = lib.ffn_igemm_tune_export(None, 0)
    buf = (CT.c_int * max(1, n * w))()
    n = min(n, lib.ffn_igemm_tune_export(buf, n))
    return torch.tensor(list(buf[:n * w]), dtype=torch.int32).reshape(n, w)


def tune_table_import(table):
    """merge a table produced by tune_table_export (another process, another rank); returns the number of entries taken."""
    lib = L.load()
    table = table.to(torch.int32).contiguous().cpu()
    w = lib.ffn_igemm_tune_entry_ints()
    if table.numel() == 0 or table.shape[1] != w:
        return 0
    buf = (CT.c_int * table.numel())(*table.flatten().tolist())
    return lib.ffn_igemm_tune_import(buf, table.shape[0])


def tune_table_clear():
    return L.load().ffn_igemm_tune_clear()


def tune_enable(on):
    """timing-based tuning of unseen bf16 igemm shapes on / off for this process (off: the deterministic rule); returns the previous setting"""
    return bool(L.load().ffn_igemm_tune_enable(1 if on else 0))


def tune_table_save(path):
    torch.save(tune_table_export(), path)


def tune_table_load(path):
    """a table written by tune_table_save.  weights_only: the file is data (a 2-D int32 tensor), never code; anything else is ignored."""
    if not os.path.exists(path):
        return 0
    table = torch.load(path, weights_only=True, map_location="cpu")
    if not torch.is_tensor(table) or table.ndim != 2 or table.dtype not in (torch.int32, torch.int64):
        return 0
    return tune_table_import(table)
