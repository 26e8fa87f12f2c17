"""ORACLE (test infrastructure, never imported by the product path).

CPU restatement (plain torch, fp32 or fp64) of the depth network of the 3D coarse-edit front end:
DepthAnything = DINOv2 ViT encoder + DPT head,
    /root/reference/depth_anything/dpt.py:143-172      DPT_DINOv2.forward: last-4-block features -> DPTHead -> bilinear -> relu
    /root/reference/depth_anything/dpt.py:22-137       DPTHead
    /root/reference/depth_anything/blocks.py:38-153    ResidualConvUnit, FeatureFusionBlock (align_corners = True)
    /root/reference/torchhub/facebookresearch_dinov2_main/vision_transformer.py:178-317   pos-embed interpolation, tokens,
                                                       get_intermediate_layers(n = 4, return_class_token = True, norm = True)
    .../dinov2/layers/{patch_embed,block,attention,mlp,layer_scale}.py
Parameter names are those of DPT_DINOv2.state_dict(): `pretrained.*` (the ViT) and `depth_head.*`.
Pinned by tests/golden/g8_dpt.npz (tools/gen_golden.py run_g8: the reference's own classes, imported, seeded weights).
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F


def dpt_config(name="vitl"):
    """encoder = dinov2_{vits,vitb,vitl}14 (hubconf.py: img_size 518, patch 14, init_values 1.0, mlp ffn, interpolate_offset 0.1);
    head sizes as the released Depth-Anything checkpoints (dpt.py:144 defaults are the vitl head)."""
    enc = dict(vits=(384, 12, 6), vitb=(768, 12, 12), vitl=(1024, 24, 16), tiny=(128, 4, 2), mini=(192, 5, 3))[name]
    head = dict(vits=(64, (48, 96, 192, 384)), vitb=(128, (96, 192, 384, 768)), vitl=(256, (256, 512, 1024, 1024)),
                tiny=(32, (16, 32, 64, 64)), mini=(48, (24, 48, 96, 96)))[name]
    return SimpleNamespace(name=name, embed_dim=enc[0], depth=enc[1], num_heads=enc[2], patch=14, img_size=518, mlp_ratio=4,
                           features=head[0], out_channels=head[1], interpolate_offset=0.1, ln_eps=1e-6)


def dpt_param_shapes(cfg):
    """name -> shape of DPT_DINOv2(encoder, features, out_channels, use_bn=False, use_clstoken=False).state_dict()"""
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
    sh[h + "resize_layers.0.weight"], sh[h + "resize_layers.0.bias"] = (oc[0], oc[0], 4, 4), (oc[0],)      # ConvTranspose2d: [in, out, kh, kw]
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


def dpt_synthetic_state(cfg, seed=0):
    """seeded weights of a plausible scale (fan-in scaled matrices, norm gains near 1, LayerScale gains O(1) so that every block
    matters), identical wherever this function runs: the golden generator loads them into the reference's modules"""
    g = torch.Generator().manual_seed(seed)
    st = {}
    for k, shp in dpt_param_shapes(cfg).items():
        if k.endswith("norm.weight") or k.endswith("norm1.weight") or k.endswith("norm2.weight"):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".gamma"):
            t = 0.5 + 0.25 * torch.rand(shp, generator=g)
        elif k.endswith(".bias"):
            t = 0.05 * torch.randn(shp, generator=g)
        elif k.endswith("pos_embed") or k.endswith("cls_token") or k.endswith("mask_token"):
            t = 0.2 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for d in (shp[1:] if "resize_layers.0" not in k and "resize_layers.1" not in k else (shp[0],)):
                fan_in *= d
            t = torch.randn(shp, generator=g) * (1.0 / math.sqrt(fan_in))
        st[k] = t.float()
    return st


# ---------------------------------------------------------------------------------------------------------------
# DINOv2 ViT (vision_transformer.py)
# ---------------------------------------------------------------------------------------------------------------
def interpolate_pos_encoding(cfg, pos_embed, npatch, w, h):
    """vision_transformer.py:178-209 (w, h are x.shape[2], x.shape[3] of the image tensor, in that order)"""
    N = pos_embed.shape[1] - 1
    if npatch == N and w == h:
        return pos_embed
    pe = pos_embed.float()
    class_pos, patch_pos = pe[:, 0], pe[:, 1:]
    dim = pe.shape[-1]
    w0, h0 = w // cfg.patch + cfg.interpolate_offset, h // cfg.patch + cfg.interpolate_offset
    sqrt_n = math.sqrt(N)
    sx, sy = float(w0) / sqrt_n, float(h0) / sqrt_n
    patch_pos = F.interpolate(patch_pos.reshape(1, int(sqrt_n), int(sqrt_n), dim).permute(0, 3, 1, 2), scale_factor=(sx, sy),
                              mode="bicubic", antialias=False)
    assert int(w0) == patch_pos.shape[-2] and int(h0) == patch_pos.shape[-1]
    patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((class_pos.unsqueeze(0), patch_pos), dim=1).to(pos_embed.dtype)


def vit_features(cfg, st, x, n_last=4):
    """get_intermediate_layers(x, 4, return_class_token=True): [(patch tokens [B, N, C], class token [B, C])] of the last 4 blocks,
    final LayerNorm applied to each (vision_transformer.py:262-317)"""
    p = "pretrained."
    B, _, w, h = x.shape
    dt = x.dtype
    W = lambda k: st[p + k].to(dt)
    t = F.conv2d(x, W("patch_embed.proj.weight"), W("patch_embed.proj.bias"), stride=cfg.patch)          # patch_embed.py:75
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat((W("cls_token").expand(B, -1, -1), t), dim=1)
    t = t + interpolate_pos_encoding(cfg, W("pos_embed"), t.shape[1] - 1, w, h)
    C, nh = cfg.embed_dim, cfg.num_heads
    outs = []
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        y = F.layer_norm(t, (C,), W(b + "norm1.weight"), W(b + "norm1.bias"), cfg.ln_eps)
        qkv = F.linear(y, W(b + "attn.qkv.weight"), W(b + "attn.qkv.bias")).reshape(B, -1, 3, nh, C // nh).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * (C // nh) ** -0.5, qkv[1], qkv[2]                                               # attention.py:55-58
        a = (q @ k.transpose(-2, -1)).softmax(dim=-1)
        y = (a @ v).transpose(1, 2).reshape(B, -1, C)
        y = F.linear(y, W(b + "attn.proj.weight"), W(b + "attn.proj.bias"))
        t = t + W(b + "ls1.gamma") * y                                                                     # block.py:86-87, 108
        y = F.layer_norm(t, (C,), W(b + "norm2.weight"), W(b + "norm2.bias"), cfg.ln_eps)
        y = F.linear(F.gelu(F.linear(y, W(b + "mlp.fc1.weight"), W(b + "mlp.fc1.bias"))), W(b + "mlp.fc2.weight"), W(b + "mlp.fc2.bias"))
        t = t + W(b + "ls2.gamma") * y
        if i >= cfg.depth - n_last:
            outs.append(t)
    outs = [F.layer_norm(o, (C,), W("norm.weight"), W("norm.bias"), cfg.ln_eps) for o in outs]
    return [(o[:, 1:], o[:, 0]) for o in outs]


# ---------------------------------------------------------------------------------------------------------------
# DPT head (dpt.py:22-137, blocks.py)
# ---------------------------------------------------------------------------------------------------------------
def _rcu(st, p, x):
    """ResidualConvUnit (blocks.py:68-83), bn = False"""
    dt = x.dtype
    out = F.conv2d(F.relu(x), st[p + "conv1.weight"].to(dt), st[p + "conv1.bias"].to(dt), padding=1)
    out = F.conv2d(F.relu(out), st[p + "conv2.weight"].to(dt), st[p + "conv2.bias"].to(dt), padding=1)
    return out + x


def _fusion(st, p, xs, size=None):
    """FeatureFusionBlock (blocks.py:128-153): size given, else scale_factor 2; align_corners = True"""
    dt = xs[0].dtype
    out = xs[0]
    if len(xs) == 2:
        out = out + _rcu(st, p + "resConfUnit1.", xs[1])
    out = _rcu(st, p + "resConfUnit2.", out)
    mod = dict(scale_factor=2) if size is None else dict(size=size)
    out = F.interpolate(out, **mod, mode="bilinear", align_corners=True)
    return F.conv2d(out, st[p + "out_conv.weight"].to(dt), st[p + "out_conv.bias"].to(dt))


def dpt_head(cfg, st, feats, ph, pw):
    h = "depth_head."
    dt = feats[0][0].dtype
    W = lambda k: st[h + k].to(dt)
    layers = []
    for i, (x, _cls) in enumerate(feats):                                                                  # use_clstoken = False: x[0]
        B = x.shape[0]
        x = x.permute(0, 2, 1).reshape(B, x.shape[-1], ph, pw)
        x = F.conv2d(x, W(f"projects.{i}.weight"), W(f"projects.{i}.bias"))
        if i == 0:
            x = F.conv_transpose2d(x, W("resize_layers.0.weight"), W("resize_layers.0.bias"), stride=4)
        elif i == 1:
            x = F.conv_transpose2d(x, W("resize_layers.1.weight"), W("resize_layers.1.bias"), stride=2)
        elif i == 3:
            x = F.conv2d(x, W("resize_layers.3.weight"), W("resize_layers.3.bias"), stride=2, padding=1)
        layers.append(x)
    rn = [F.conv2d(layers[i], W(f"scratch.layer{i + 1}_rn.weight"), None, padding=1) for i in range(4)]
    s = h + "scratch."
    path4 = _fusion(st, s + "refinenet4.", [rn[3]], size=rn[2].shape[2:])
    path3 = _fusion(st, s + "refinenet3.", [path4, rn[2]], size=rn[1].shape[2:])
    path2 = _fusion(st, s + "refinenet2.", [path3, rn[1]], size=rn[0].shape[2:])
    path1 = _fusion(st, s + "refinenet1.", [path2, rn[0]])
    out = F.conv2d(path1, W("scratch.output_conv1.weight"), W("scratch.output_conv1.bias"), padding=1)
    out = F.interpolate(out, (int(ph * 14), int(pw * 14)), mode="bilinear", align_corners=True)
    out = F.relu(F.conv2d(out, W("scratch.output_conv2.0.weight"), W("scratch.output_conv2.0.bias"), padding=1))
    return F.relu(F.conv2d(out, W("scratch.output_conv2.2.weight"), W("scratch.output_conv2.2.bias")))


@torch.no_grad()
def depth_forward(cfg, st, x):
    """DPT_DINOv2.forward (dpt.py:155-167): x [B, 3, H, W] (H, W multiples of 14) -> depth [B, H, W]"""
    H, W = x.shape[-2:]
    feats = vit_features(cfg, st, x, 4)
    ph, pw = H // 14, W // 14
    depth = dpt_head(cfg, st, feats, ph, pw)
    depth = F.interpolate(depth, size=(H, W), mode="bilinear", align_corners=True)
    return F.relu(depth).squeeze(1)
