"""DepthAnything (DINOv2 ViT encoder + DPT head) on the HIP kernels: the depth network of the 3D coarse-edit front end
(/root/reference/depth_anything/dpt.py:143-172 DPT_DINOv2.forward, :22-137 DPTHead, blocks.py:38-153,
torchhub/facebookresearch_dinov2_main/vision_transformer.py:178-317).

Same design as unet.py / vae.py: activations are token / NHWC rows, every Linear, 1x1 conv, 3x3 conv, patch embedding and
transposed conv is an `ffn_igemm` call, LayerNorm and attention are the UNet's kernels (`ffn_layernorm`, `ffn_attn` with head
dim 64 and the ragged S = 1 + (H/14)(W/14)); what the UNet did not need is in the C ABI as `FFN_IG_OUT_GELU` / `FFN_IG_OUT_RELU`
epilogues, `ffn_eltwise` (the activation in front of a ResidualConvUnit, the fusion blocks' skip add) and `ffn_resize_bilinear`
(align_corners = True).  Folded at pack time: LayerScale into attn.proj / mlp.fc2 (rows scaled by gamma), the patch embedding
into a [C, 3*14*14] GEMM weight over an im2col view of the image, ConvTranspose2d(k = s) into a GEMM with k*k*C_out columns followed by
a pixel shuffle (tensor plumbing).  torch is used for layout plumbing only (im2col view, concatenating the class token, the pixel
shuffle, the one-off bicubic interpolation of the positional embedding per input size).

dtype float32 = parity mode (exact-fp32 MFMA), bfloat16 = fast mode.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from . import ops


def depth_config(name="vitl"):
    """encoder = dinov2_{vits,vitb,vitl}14 (hubconf.py: img_size 518, patch 14, LayerScale, mlp ffn, interpolate_offset 0.1); DPT head sizes of
    the released Depth-Anything checkpoints (dpt.py:144: the vitl head)"""
    enc = dict(vits=(384, 12, 6), vitb=(768, 12, 12), vitl=(1024, 24, 16), tiny=(128, 4, 2), mini=(192, 5, 3))[name]
    head = dict(vits=(64, (48, 96, 192, 384)), vitb=(128, (96, 192, 384, 768)), vitl=(256, (256, 512, 1024, 1024)),
                tiny=(32, (16, 32, 64, 64)), mini=(48, (24, 48, 96, 96)))[name]
    return SimpleNamespace(name=name, embed_dim=enc[0], depth=enc[1], num_heads=enc[2], patch=14, img_size=518, mlp_ratio=4,
                           features=head[0], out_channels=head[1], interpolate_offset=0.1, ln_eps=1e-6)


def depth_param_shapes(cfg):
    """name -> shape of DPT_DINOv2(encoder, features, out_channels, use_bn=False, use_clstoken=False).state_dict() (dpt.py:143-153)"""
    C, hid = cfg.embed_dim, cfg.embed_dim * cfg.mlp_ratio
    n = (cfg.img_size // cfg.patch) ** 2
    sh = {"pretrained.cls_token": (1, 1, C), "pretrained.pos_embed": (1, n + 1, C), "pretrained.mask_token": (1, C),
          "pretrained.patch_embed.proj.weight": (C, 3, cfg.patch, cfg.patch), "pretrained.patch_embed.proj.bias": (C,),
          "pretrained.norm.weight": (C,), "pretrained.norm.bias": (C,)}
    for i in range(cfg.depth):
        p = f"pretrained.blocks.{i}."
        sh.update({p + "norm1.weight": (C,), p + "norm1.bias": (C,), p + "attn.qkv.weight": (3 * C, C), p + "attn.qkv.bias": (3 * C,),
                   p + "attn.proj.weight": (C, C), p + "attn.proj.bias": (C,), p + "ls1.gamma": (C,),
                   p + "norm2.weight": (C,), p + "norm2.bias": (C,), p + "mlp.fc1.weight": (hid, C), p + "mlp.fc1.bias": (hid,),
                   p + "mlp.fc2.weight": (C, hid), p + "mlp.fc2.bias": (C,), p + "ls2.gamma": (C,)})
    oc, f = cfg.out_channels, cfg.features
    h = "depth_head."
    for i in range(4):
        sh[h + f"projects.{i}.weight"], sh[h + f"projects.{i}.bias"] = (oc[i], C, 1, 1), (oc[i],)
        sh[h + f"scratch.layer{i + 1}_rn.weight"] = (f, oc[i], 3, 3)
    sh[h + "resize_layers.0.weight"], sh[h + "resize_layers.0.bias"] = (oc[0], oc[0], 4, 4), (oc[0],)
    sh[h + "resize_layers.1.weight"], sh[h + "resize_layers.1.bias"] = (oc[1], oc[1], 2, 2), (oc[1],)
    sh[h + "resize_layers.3.weight"], sh[h + "resize_layers.3.bias"] = (oc[3], oc[3], 3, 3), (oc[3],)
    for i in range(1, 5):
        r = h + f"scratch.refinenet{i}."
        sh[r + "out_conv.weight"], sh[r + "out_conv.bias"] = (f, f, 1, 1), (f,)
        for u in ("resConfUnit1", "resConfUnit2"):
            for c in ("conv1", "conv2"):
                sh[r + f"{u}.{c}.weight"], sh[r + f"{u}.{c}.bias"] = (f, f, 3, 3), (f,)
    sh[h + "scratch.output_conv1.weight"], sh[h + "scratch.output_conv1.bias"] = (f // 2, f, 3, 3), (f // 2,)
    sh[h + "scratch.output_conv2.0.weight"], sh[h + "scratch.output_conv2.0.bias"] = (32, f // 2, 3, 3), (32,)
    sh[h + "scratch.output_conv2.2.weight"], sh[h + "scratch.output_conv2.2.bias"] = (1, 32, 1, 1), (1,)
    return sh


def synthetic_state(cfg, seed=0):
    """seeded random weights of a plausible scale for benchmarks without a checkpoint (`tools/bench_depth.py`; there is no network)"""
    g = torch.Generator().manual_seed(seed)
    st = {}
    for k, shp in depth_param_shapes(cfg).items():
        if k.endswith("norm.weight") or k.endswith("norm1.weight") or k.endswith("norm2.weight"):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".gamma"):
            t = 0.5 + 0.25 * torch.rand(shp, generator=g)
        elif k.endswith(".bias"):
            t = 0.05 * torch.randn(shp, generator=g)
        elif k.endswith("pos_embed") or k.endswith("cls_token") or k.endswith("mask_token"):
            t = 0.2 * torch.randn(shp, generator=g)
        else:
            fan_in = shp[0] if ("resize_layers.0" in k or "resize_layers.1" in k) else math.prod(shp[1:])
            t = torch.randn(shp, generator=g) / math.sqrt(fan_in)
        st[k] = t.float()
    return st


class _O:
    pass


class HipDepthAnything:
    def __init__(self, cfg, state, dtype=torch.bfloat16, device="cuda:0"):
        """state: DPT_DINOv2.state_dict() (`pretrained.*`, `depth_head.*`; use_bn = False, use_clstoken = False)"""
        assert dtype in (torch.float32, torch.bfloat16)
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        C = cfg.embed_dim
        assert C % cfg.num_heads == 0 and C % 8 == 0 and cfg.features % 16 == 0 and all(c % 8 == 0 for c in cfg.out_channels), \
            "channel counts must be whole 16-byte chunks (features / 2 feeds a 3x3 conv)"
        self._pos = {}
        self._pack({k: v.detach().to(self.device, torch.float32) for k, v in state.items()})

    # ------------------------------------------------------------------------------------------------------------
    # weights
    # ------------------------------------------------------------------------------------------------------------
    def _lin(self, w, b=None, scale=None, n_pad=None):
        w = w.reshape(w.shape[0], -1)
        if scale is not None:                                 # LayerScale folded: gamma * (W x + b)
            w = w * scale[:, None]
            b = None if b is None else b * scale
        if n_pad and n_pad > w.shape[0]:
            w = torch.cat([w, torch.zeros(n_pad - w.shape[0], w.shape[1], device=w.device)], 0)
            b = None if b is None else torch.cat([b, torch.zeros(n_pad - b.shape[0], device=b.device)], 0)
        return ops.pack_linear(w.contiguous(), self.dtype), (None if b is None else b.float().contiguous()), w.shape[1]

    def _conv(self, st, name, bias=True):
        w = st[name + ".weight"]
        return ops.pack_conv3x3(w, self.dtype), (st[name + ".bias"].contiguous() if bias else None), w.shape[1]

    def _deconv(self, st, name, k):
        """ConvTranspose2d(kernel = stride = k): out[4y+dy, 4x+dx, co] = sum_ci in[y, x, ci] w[ci, co, dy, dx] + b[co] -> GEMM columns (dy, dx, co)"""
        w, b = st[name + ".weight"], st[name + ".bias"]       # [in, out, k, k]
        cin, cout = w.shape[0], w.shape[1]
        wg = w.permute(2, 3, 1, 0).reshape(k * k * cout, cin).contiguous()
        return ops.pack_linear(wg, self.dtype), b.repeat(k * k).contiguous(), cin, cout, k

    def _pack(self, st):
        cfg = self.cfg
        C = cfg.embed_dim
        p = "pretrained."
        e = 8
        K = 3 * cfg.patch * cfg.patch
        self.kpe = (K + e - 1) // e * e                        # patch-embedding contraction length, padded to whole chunks
        wpe = torch.zeros(C, self.kpe, device=self.device)
        wpe[:, :K] = st[p + "patch_embed.proj.weight"].reshape(C, K)
        self.pe = (ops.pack_linear(wpe, self.dtype), st[p + "patch_embed.proj.bias"].contiguous())
        self.pos_embed, self.cls_token = st[p + "pos_embed"], st[p + "cls_token"]
        self.blocks = []
        for i in range(cfg.depth):
            b, q = _O(), f"{p}blocks.{i}."
            b.n1 = (st[q + "norm1.weight"].contiguous(), st[q + "norm1.bias"].contiguous())
            b.n2 = (st[q + "norm2.weight"].contiguous(), st[q + "norm2.bias"].contiguous())
            wqkv, bqkv = st[q + "attn.qkv.weight"], st[q + "attn.qkv.bias"]
            b.qk = self._lin(wqkv[:2 * C], bqkv[:2 * C])        # q | k in one GEMM, V^T from its own (transposed-output) GEMM
            b.v = self._lin(wqkv[2 * C:], bqkv[2 * C:])
            b.proj = self._lin(st[q + "attn.proj.weight"], st[q + "attn.proj.bias"], scale=st[q + "ls1.gamma"])
            b.fc1 = self._lin(st[q + "mlp.fc1.weight"], st[q + "mlp.fc1.bias"])
            b.fc2 = self._lin(st[q + "mlp.fc2.weight"], st[q + "mlp.fc2.bias"], scale=st[q + "ls2.gamma"])
            self.blocks.append(b)
        self.norm = (st[p + "norm.weight"].contiguous(), st[p + "norm.bias"].contiguous())
        h = "depth_head."
        self.projects = [self._lin(st[h + f"projects.{i}.weight"], st[h + f"projects.{i}.bias"]) for i in range(4)]
        self.up0 = self._deconv(st, h + "resize_layers.0", 4)
        self.up1 = self._deconv(st, h + "resize_layers.1", 2)
        self.down3 = self._conv(st, h + "resize_layers.3")
        self.rn = [self._conv(st, h + f"scratch.layer{i + 1}_rn", bias=False) for i in range(4)]
        self.refine = []
        for i in range(1, 5):
            r, q = _O(), h + f"scratch.refinenet{i}."
            r.u1 = (self._conv(st, q + "resConfUnit1.conv1"), self._conv(st, q + "resConfUnit1.conv2"))
            r.u2 = (self._conv(st, q + "resConfUnit2.conv1"), self._conv(st, q + "resConfUnit2.conv2"))
            r.out = self._lin(st[q + "out_conv.weight"], st[q + "out_conv.bias"])
            self.refine.append(r)
        self.oc1 = self._conv(st, h + "scratch.output_conv1")
        self.oc2 = self._conv(st, h + "scratch.output_conv2.0")
        self.oc3 = self._lin(st[h + "scratch.output_conv2.2.weight"], st[h + "scratch.output_conv2.2.bias"], n_pad=4)

    # ------------------------------------------------------------------------------------------------------------
    # encoder
    # ------------------------------------------------------------------------------------------------------------
    def _pos_tokens(self, H, W):
        """(class row = cls_token + pos[0] [1, C], positional embedding of the H/14 x W/14 patches [N, C]) in the activation dtype --
        vision_transformer.py:178-209 (bicubic, antialias off, offset 0.1; evaluated once per input size)"""
        key = (H, W)
        hit = self._pos.get(key)
        if hit is not None:
            return hit
        cfg = self.cfg
        pe = self.pos_embed.float()
        N = pe.shape[1] - 1
        npatch = (H // cfg.patch) * (W // cfg.patch)
        patch_pos = pe[:, 1:]
        if not (npatch == N and H == W):
            dim = pe.shape[-1]
            w0, h0 = H // cfg.patch + cfg.interpolate_offset, W // cfg.patch + cfg.interpolate_offset
            sq = math.sqrt(N)
            patch_pos = F.interpolate(patch_pos.reshape(1, int(sq), int(sq), dim).permute(0, 3, 1, 2), scale_factor=(float(w0) / sq, float(h0) / sq),
                                      mode="bicubic", antialias=False)
            assert int(w0) == patch_pos.shape[-2] and int(h0) == patch_pos.shape[-1]
            patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, -1, dim)
        cls_row = (self.cls_token.float()[0] + pe[:, 0]).to(self.dtype).contiguous()
        hit = self._pos[key] = (cls_row, patch_pos[0].to(self.dtype).contiguous())
        return hit

    @staticmethod
    def _im2col(x, ps):
        """[B, 3, H, W] -> [B * (H/ps) * (W/ps), 3 * ps * ps]: one row per patch (row-major over the patch grid), columns ordered (channel, ky, kx)
        like patch_embed.proj.weight.reshape(C, -1) (patch_embed.py:75: Conv2d(kernel = stride = patch))"""
        B, Cc, H, W = x.shape
        ph, pw = H // ps, W // ps
        return x.reshape(B, Cc, ph, ps, pw, ps).permute(0, 2, 4, 1, 3, 5).reshape(B * ph * pw, Cc * ps * ps)

    @staticmethod
    def _pixel_shuffle(y, B, H, W, k, cout):
        """GEMM output [B, H*W, (dy, dx, co)] of a ConvTranspose2d(kernel = stride = k) -> NHWC rows [B, (H k)(W k), co]"""
        return y.view(B, H, W, k, k, cout).permute(0, 1, 3, 2, 4, 5).reshape(B, H * k * W * k, cout).contiguous()

    def _tokens(self, x):
        """patch embedding (a GEMM over the im2col view of the image, positional embedding added as the GEMM's residual) + class row"""
        cfg = self.cfg
        B, _, H, W = x.shape
        ps = cfg.patch
        ph, pw = H // ps, W // ps
        cols = self._im2col(x.to(self.device, torch.float32), ps)
        a = torch.zeros(B * ph * pw, self.kpe, dtype=self.dtype, device=self.device)
        a[:, :cols.shape[1]] = cols.to(self.dtype)
        cls_row, pos = self._pos_tokens(H, W)
        res = pos.unsqueeze(0).expand(B, -1, -1).reshape(B * ph * pw, -1).contiguous()
        t = ops.linear(a, self.pe[0], self.pe[1], K=self.kpe, residual=res)
        return torch.cat([cls_row.unsqueeze(0).expand(B, -1, -1), t.view(B, ph * pw, -1)], dim=1).contiguous(), ph, pw

    def _block(self, b, t, B, S):
        cfg = self.cfg
        C, nh = cfg.embed_dim, cfg.num_heads
        y = ops.layernorm(t, *b.n1, eps=cfg.ln_eps)
        qk = ops.linear(y, b.qk[0], b.qk[1], K=C)                                   # [B, S, 2C]: q | k
        vt = ops.linear(y, b.v[0], b.v[1], K=C, rows_per_batch=S, transposed_ld=(S + 7) // 8 * 8)      # V^T [B, C, S']
        a = ops.attention(qk, qk[..., C:], vt, nh, (C // nh) ** -0.5, None, Sk=S, C=C)
        t = ops.linear(a, b.proj[0], b.proj[1], K=C, residual=t)                     # x + ls1 * proj(attn)
        y = ops.layernorm(t, *b.n2, eps=cfg.ln_eps)
        y = ops.linear(y, b.fc1[0], b.fc1[1], K=C, gelu=True)
        return ops.linear(y, b.fc2[0], b.fc2[1], K=C * cfg.mlp_ratio, residual=t)    # x + ls2 * fc2(gelu(fc1))

    def features(self, x):
        """get_intermediate_layers(x, 4, return_class_token=True, norm=True): patch tokens [B, N, C] of the last four blocks"""
        cfg = self.cfg
        t, ph, pw = self._tokens(x)
        B, S, _ = t.shape
        outs = []
        for i, b in enumerate(self.blocks):
            t = self._block(b, t, B, S)
            if i >= cfg.depth - 4:
                outs.append(t)
        outs = [ops.layernorm(o, *self.norm, eps=cfg.ln_eps)[:, 1:].contiguous() for o in outs]
        return outs, ph, pw

    # ------------------------------------------------------------------------------------------------------------
    # DPT head (NHWC rows)
    # ------------------------------------------------------------------------------------------------------------
    def _up(self, x, B, H, W, dc):
        w, b, cin, cout, k = dc
        y = ops.linear(x, w, b, K=cin)                                               # [B, H*W, k*k*cout]
        return self._pixel_shuffle(y, B, H, W, k, cout), H * k, W * k

    def _rcu(self, u, x, B, H, W, extra=None):
        """ResidualConvUnit (blocks.py:68-83): conv2(relu(conv1(relu(x)))) + x (+ extra: the fusion block's other input)"""
        C = x.shape[-1]
        y = ops.conv3x3(ops.relu(x), u[0][0], u[0][1], B, H, W, C, relu=True)
        return ops.conv3x3(y, u[1][0], u[1][1], B, H, W, C, residual=x if extra is None else ops.add(x, extra))

    def _fusion(self, r, xs, B, H, W, size):
        """FeatureFusionBlock (blocks.py:128-153), xs = [path (, skip level)], resized to `size` (align_corners = True), out_conv"""
        out = xs[0]
        if len(xs) == 2:
            out = self._rcu(r.u1, xs[1], B, H, W, extra=out)
        out = self._rcu(r.u2, out, B, H, W)
        out = ops.resize_bilinear(out, B, H, W, size[0], size[1])
        return ops.linear(out, r.out[0], r.out[1], K=out.shape[-1])

    def head(self, feats, ph, pw):
        cfg = self.cfg
        B = feats[0].shape[0]
        C = cfg.embed_dim
        lv = []
        for i, x in enumerate(feats):
            x = ops.linear(x, self.projects[i][0], self.projects[i][1], K=C)
            if i == 0:
                x, h, w = self._up(x, B, ph, pw, self.up0)
            elif i == 1:
                x, h, w = self._up(x, B, ph, pw, self.up1)
            elif i == 2:
                h, w = ph, pw
            else:
                x = ops.conv3x3(x, self.down3[0], self.down3[1], B, ph, pw, x.shape[-1], stride=2)
                h, w = (ph - 1) // 2 + 1, (pw - 1) // 2 + 1
            lv.append((ops.conv3x3(x, self.rn[i][0], None, B, h, w, x.shape[-1]), h, w))
        (l1, h1, w1), (l2, h2, w2), (l3, h3, w3), (l4, h4, w4) = lv
        p4 = self._fusion(self.refine[3], [l4], B, h4, w4, (h3, w3))
        p3 = self._fusion(self.refine[2], [p4, l3], B, h3, w3, (h2, w2))
        p2 = self._fusion(self.refine[1], [p3, l2], B, h2, w2, (h1, w1))
        p1 = self._fusion(self.refine[0], [p2, l1], B, h1, w1, (2 * h1, 2 * w1))
        H1, W1 = 2 * h1, 2 * w1
        out = ops.conv3x3(p1, self.oc1[0], self.oc1[1], B, H1, W1, cfg.features)
        Ho, Wo = ph * 14, pw * 14
        out = ops.resize_bilinear(out, B, H1, W1, Ho, Wo)
        out = ops.conv3x3(out, self.oc2[0], self.oc2[1], B, Ho, Wo, cfg.features // 2, relu=True)
        out = ops.linear(out, self.oc3[0], self.oc3[1], K=32, relu=True, out_f32=True)       # [B, Ho*Wo, 4]: channel 0 is the depth
        return out, Ho, Wo

    @torch.no_grad()
    def forward(self, x):
        """x [B, 3, H, W] (normalised image, H and W multiples of 14) -> depth [B, H, W] fp32 (DPT_DINOv2.forward, dpt.py:155-167)"""
        B, _, H, W = x.shape
        assert H % 14 == 0 and W % 14 == 0, "DINOv2 patch embedding: image sides must be multiples of 14"
        feats, ph, pw = self.features(x)
        d, Ho, Wo = self.head(feats, ph, pw)
        if (Ho, Wo) != (H, W):                                 # (never with sides that are multiples of 14; kept for the reference's call shape)
            d = ops.resize_bilinear(d, B, Ho, Wo, H, W, relu=True)
        return d[..., 0].reshape(B, H, W).float()              # the final F.relu is the last GEMM's epilogue

    __call__ = forward
