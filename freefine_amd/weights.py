"""State dicts in diffusers' layout: shape tables, seeded synthetic initialisation (no checkpoints exist in this
environment) and a safetensors loader for real HF folders (unet/diffusion_pytorch_model.safetensors etc.)."""
import json
import math
import os

import numpy as np
import torch

from .config import UNetConfig, VAEConfig


def _lin(d, name, cout, cin, bias=True):
    d[name + ".weight"] = (cout, cin)
    if bias:
        d[name + ".bias"] = (cout,)


def _conv(d, name, cout, cin, k):
    d[name + ".weight"] = (cout, cin, k, k)
    d[name + ".bias"] = (cout,)


def _norm(d, name, c):
    d[name + ".weight"] = (c,)
    d[name + ".bias"] = (c,)


def _resnet(d, p, cin, cout, temb):
    _norm(d, p + ".norm1", cin)
    _conv(d, p + ".conv1", cout, cin, 3)
    if temb:
        _lin(d, p + ".time_emb_proj", cout, temb)
    _norm(d, p + ".norm2", cout)
    _conv(d, p + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(d, p + ".conv_shortcut", cout, cin, 1)


def _transformer(d, p, c, cross, linear):
    _norm(d, p + ".norm", c)
    if linear:
        _lin(d, p + ".proj_in", c, c)
        _lin(d, p + ".proj_out", c, c)
    else:
        _conv(d, p + ".proj_in", c, c, 1)
        _conv(d, p + ".proj_out", c, c, 1)
    b = p + ".transformer_blocks.0"
    for i, kv in ((1, c), (2, cross)):
        _norm(d, f"{b}.norm{i}", c)
        _lin(d, f"{b}.attn{i}.to_q", c, c, bias=False)
        _lin(d, f"{b}.attn{i}.to_k", c, kv, bias=False)
        _lin(d, f"{b}.attn{i}.to_v", c, kv, bias=False)
        _lin(d, f"{b}.attn{i}.to_out.0", c, c)
    _norm(d, b + ".norm3", c)
    _lin(d, b + ".ff.net.0.proj", 8 * c, c)
    _lin(d, b + ".ff.net.2", c, 4 * c)


def unet_param_shapes(cfg: UNetConfig):
    d = {}
    ch = cfg.block_out_channels
    temb = ch[0] * 4
    n = len(ch)
    _conv(d, "conv_in", ch[0], cfg.in_channels, 3)
    _lin(d, "time_embedding.linear_1", temb, ch[0])
    _lin(d, "time_embedding.linear_2", temb, temb)
    cout = ch[0]
    for i in range(n):
        cin, cout = cout, ch[i]
        for j in range(cfg.layers_per_block):
            _resnet(d, f"down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, temb)
            if cfg.down_has_attn[i]:
                _transformer(d, f"down_blocks.{i}.attentions.{j}", cout, cfg.cross_attention_dim, cfg.use_linear_projection)
        if i < n - 1:
            _conv(d, f"down_blocks.{i}.downsamplers.0.conv", cout, cout, 3)
    _resnet(d, "mid_block.resnets.0", ch[-1], ch[-1], temb)
    _transformer(d, "mid_block.attentions.0", ch[-1], cfg.cross_attention_dim, cfg.use_linear_projection)
    _resnet(d, "mid_block.resnets.1", ch[-1], ch[-1], temb)
    rev = list(reversed(ch))
    rev_attn = list(reversed(cfg.down_has_attn))
    cout = rev[0]
    for i in range(n):
        cprev, cout = cout, rev[i]
        cin = rev[min(i + 1, n - 1)]
        for j in range(cfg.layers_per_block + 1):
            skip = cin if j == cfg.layers_per_block else cout
            rin = cprev if j == 0 else cout
            _resnet(d, f"up_blocks.{i}.resnets.{j}", rin + skip, cout, temb)
            if rev_attn[i]:
                _transformer(d, f"up_blocks.{i}.attentions.{j}", cout, cfg.cross_attention_dim, cfg.use_linear_projection)
        if i < n - 1:
            _conv(d, f"up_blocks.{i}.upsamplers.0.conv", cout, cout, 3)
    _norm(d, "conv_norm_out", ch[0])
    _conv(d, "conv_out", cfg.out_channels, ch[0], 3)
    return d


def _vae_attn(d, p, c):
    _norm(d, p + ".group_norm", c)
    for nme in ("to_q", "to_k", "to_v", "to_out.0"):
        _lin(d, f"{p}.{nme}", c, c)


def vae_param_shapes(cfg: VAEConfig):
    d = {}
    ch = cfg.block_out_channels
    n = len(ch)
    _conv(d, "encoder.conv_in", ch[0], cfg.in_channels, 3)
    cout = ch[0]
    for i, c in enumerate(ch):
        cin, cout = cout, c
        for j in range(cfg.layers_per_block):
            _resnet(d, f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, 0)
        if i < n - 1:
            _conv(d, f"encoder.down_blocks.{i}.downsamplers.0.conv", cout, cout, 3)
    for side, c in (("encoder", ch[-1]), ("decoder", ch[-1])):
        _resnet(d, f"{side}.mid_block.resnets.0", c, c, 0)
        _vae_attn(d, f"{side}.mid_block.attentions.0", c)
        _resnet(d, f"{side}.mid_block.resnets.1", c, c, 0)
    _norm(d, "encoder.conv_norm_out", ch[-1])
    _conv(d, "encoder.conv_out", 2 * cfg.latent_channels, ch[-1], 3)
    _conv(d, "quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    _conv(d, "post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    rev = list(reversed(ch))
    _conv(d, "decoder.conv_in", rev[0], cfg.latent_channels, 3)
    cout = rev[0]
    for i, c in enumerate(rev):
        cin, cout = cout, c
        for j in range(cfg.layers_per_block + 1):
            _resnet(d, f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, 0)
        if i < n - 1:
            _conv(d, f"decoder.up_blocks.{i}.upsamplers.0.conv", cout, cout, 3)
    _norm(d, "decoder.conv_norm_out", ch[0])
    _conv(d, "decoder.conv_out", cfg.out_channels, ch[0], 3)
    return d


def synthetic_state(shapes, seed=0):
    """torch-default-like init (U(-1/sqrt(fan_in), +)) for weights and biases, norms ~ (1, 0) with a small seeded
    perturbation.  numpy Generator -> identical on every host."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shp in shapes.items():
        is_norm = ("norm" in name.split(".")[-2]) or name.split(".")[-2] in ("norm",)
        if is_norm:
            base = 1.0 if name.endswith("weight") else 0.0
            a = base + 0.1 * rng.standard_normal(shp)
        else:
            wshape = shapes[name[: -len("bias")] + "weight"] if name.endswith("bias") else shp
            fan_in = int(np.prod(wshape[1:]))
            bound = 1.0 / math.sqrt(fan_in)
            a = rng.uniform(-bound, bound, shp)
        out[name] = torch.from_numpy(a.astype(np.float32))
    return out


# the VAE mid-block attention of the SD-1.x / 2.x checkpoints on the hub still carries the pre-0.14 diffusers parameter names (1x1
# convolutions stored 4-D); diffusers renames them at load time (AttentionBlock -> Attention deprecation path), so do we
_LEGACY_VAE_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def normalize_state_dict(state):
    """legacy -> current diffusers key names (VAE attention), 1x1-conv weights of Linear-shaped layers flattened to 2-D."""
    out = {}
    for k, v in state.items():
        parts = k.split(".")
        if len(parts) >= 2 and ".attentions." in k and parts[-2] in _LEGACY_VAE_ATTN:
            parts[-2] = _LEGACY_VAE_ATTN[parts[-2]]
            k = ".".join(parts)
            if v.ndim == 4 and v.shape[2] == v.shape[3] == 1:
                v = v.reshape(v.shape[0], v.shape[1])
        out[k] = v
    return out


def validate_state_dict(state, shapes, what):
    """fail with a readable message (not a KeyError deep inside the packer) when a checkpoint does not match the topology."""
    missing = sorted(set(shapes) - set(state))
    if missing:
        raise ValueError(f"{what}: {len(missing)} parameter(s) missing from the checkpoint, e.g. {missing[:4]} "
                         f"(have {len(state)} tensors; unexpected: {sorted(set(state) - set(shapes))[:4]})")
    for k, shp in shapes.items():
        got = tuple(state[k].shape)
        if got != tuple(shp) and not (len(got) == 4 and got[2:] == (1, 1) and got[:2] == tuple(shp)):
            raise ValueError(f"{what}: parameter {k} has shape {got}, expected {tuple(shp)}")


def load_safetensors_dir(path, sub):
    """path/sub/{config.json, diffusion_pytorch_model.safetensors}"""
    from safetensors.torch import load_file
    folder = os.path.join(path, sub)
    with open(os.path.join(folder, "config.json")) as f:
        cfg = json.load(f)
    for fn in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors", "model.safetensors"):
        fp = os.path.join(folder, fn)
        if os.path.exists(fp):
            return cfg, normalize_state_dict({k: v.float() for k, v in load_file(fp).items()})
    raise FileNotFoundError(f"no safetensors under {folder}")


def plant_denoiser_path(state, cfg: UNetConfig, gain=3.0):
    """Make a seeded RANDOM UNet state behave like a (crude) noise predictor, so that a full 50-step DDIM edit with it is a denoising
    trajectory (latents stay O(1)) instead of a 15x amplification of whatever eps it emits (sqrt(abar_0 / abar_981) = 14.6: with eps
    unrelated to the noise in x nothing removes what the DDPM step injects).  Three tensors get a linear path ADDED to their random
    values -- nothing is scaled down, every other weight keeps its default-init value and still feeds eps through the final GroupNorm:
      conv_in centre tap          += E                (E: [C0, 4], columns = +-1 sign patterns, mutually orthogonal)
      last up-ResBlock shortcut   += gain * I on the columns of conv_in's skip tensor (the linear skip diffusers' UNet already has)
      conv_out centre tap         += E^T / (C0 / 4)   (reads the planted component back out of silu(GroupNorm(h)): the odd part of silu is z / 2,
                                                       GroupNorm divides E x by 2 rms(x))
    so eps ~= x / rms(x) + (what the random network adds): the optimal prediction for unit-variance noise around small data.  Bench / parity
    fixtures only; a real checkpoint needs none of this."""
    C0 = cfg.block_out_channels[0]
    assert C0 % 16 == 0 and cfg.in_channels == 4 and cfg.out_channels == 4
    c = np.arange(C0)
    E = np.stack([1.0 - 2.0 * ((c >> k) & 1) for k in range(4)], axis=1).astype(np.float32)       # [C0, 4]
    out = dict(state)
    w = out["conv_in.weight"].clone()
    w[:, :, 1, 1] += torch.from_numpy(E)
    out["conv_in.weight"] = w
    n = len(cfg.block_out_channels)
    key = f"up_blocks.{n - 1}.resnets.{cfg.layers_per_block}.conv_shortcut.weight"
    w = out[key].clone()                                                                           # [C0, 2 C0, 1, 1]: columns [hidden | skip]
    idx = torch.arange(C0)
    w[idx, C0 + idx, 0, 0] += gain
    out[key] = w
    w = out["conv_out.weight"].clone()
    w[:, :, 1, 1] += torch.from_numpy(E.T) / (C0 / 4.0)
    out["conv_out.weight"] = w
    return out
