"""ORACLE (test infrastructure, never imported by the product path).

CPU restatement (plain torch fp32) of diffusers-0.18 AutoencoderKL as used by the reference's image2latent /
latent2image (/root/reference/src/demo/model.py:223-280): encode -> latent_dist.mean, decode -> sample.
Parameter names follow diffusers' state-dict layout (0.18 spelling of the mid-block attention: to_q/to_k/to_v/
to_out.0 with group_norm).  In-tree structural cross-check: evaluation/MotionGuidance/ldm/modules/diffusionmodules/
model.py (Encoder/Decoder).
"""
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


def vae_config(name="sd"):
    if name == "sd":
        return SimpleNamespace(name=name, in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                               layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)
    if name == "tiny":
        return SimpleNamespace(name=name, in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(16, 32, 32, 32),
                               layers_per_block=1, norm_num_groups=8, scaling_factor=0.18215)
    raise ValueError(name)


class VaeResnet(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class VaeAttention(nn.Module):
    """single-head spatial self-attention with GroupNorm and residual (diffusers Attention(heads=1, bias=True,
    residual_connection=True, norm_num_groups=32))."""

    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def forward(self, x):
        b, c, h, w = x.shape
        y = self.group_norm(x).view(b, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(y), self.to_k(y), self.to_v(y)
        p = (c ** -0.5 * (q @ k.transpose(1, 2))).softmax(dim=-1)
        y = self.to_out[0](p @ v)
        return x + y.transpose(1, 2).reshape(b, c, h, w)


class VaeMid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnet(c, c, groups), VaeResnet(c, c, groups)])
        self.attentions = nn.ModuleList([VaeAttention(c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class DownEncoderBlock(nn.Module):
    def __init__(self, cin, cout, n, groups, down):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnet(cin if i == 0 else cout, cout, groups) for i in range(n)])
        self.downsamplers = nn.ModuleList([nn.Module()]) if down else None
        if down:
            self.downsamplers[0].conv = nn.Conv2d(cout, cout, 3, stride=2, padding=0)

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0].conv(F.pad(x, (0, 1, 0, 1)))
        return x


class UpDecoderBlock(nn.Module):
    def __init__(self, cin, cout, n, groups, up):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnet(cin if i == 0 else cout, cout, groups) for i in range(n)])
        self.upsamplers = nn.ModuleList([nn.Module()]) if up else None
        if up:
            self.upsamplers[0].conv = nn.Conv2d(cout, cout, 3, padding=1)

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0].conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))
        return x


class Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        ch, g = cfg.block_out_channels, cfg.norm_num_groups
        self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        cout = ch[0]
        for i, c in enumerate(ch):
            cin, cout = cout, c
            self.down_blocks.append(DownEncoderBlock(cin, cout, cfg.layers_per_block, g, i < len(ch) - 1))
        self.mid_block = VaeMid(ch[-1], g)
        self.conv_norm_out = nn.GroupNorm(g, ch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[-1], 2 * cfg.latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        ch, g = cfg.block_out_channels, cfg.norm_num_groups
        rev = list(reversed(ch))
        self.conv_in = nn.Conv2d(cfg.latent_channels, rev[0], 3, padding=1)
        self.mid_block = VaeMid(rev[0], g)
        self.up_blocks = nn.ModuleList()
        cout = rev[0]
        for i, c in enumerate(rev):
            cin, cout = cout, c
            self.up_blocks.append(UpDecoderBlock(cin, cout, cfg.layers_per_block + 1, g, i < len(ch) - 1))
        self.conv_norm_out = nn.GroupNorm(g, ch[0], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[0], cfg.out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKL(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.encoder, self.decoder = Encoder(cfg), Decoder(cfg)
        self.quant_conv = nn.Conv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(cfg.latent_channels, cfg.latent_channels, 1)

    @property
    def dtype(self):
        return self.quant_conv.weight.dtype

    def encode_mean(self, x):
        moments = self.quant_conv(self.encoder(x))
        return moments[:, : self.cfg.latent_channels]

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))


def init_vae(cfg, seed=1, perturb_norms=True):
    st = torch.random.get_rng_state()
    torch.manual_seed(seed)
    net = AutoencoderKL(cfg)
    if perturb_norms:
        with torch.no_grad():
            for m in net.modules():
                if isinstance(m, nn.GroupNorm):
                    m.weight.add_(0.1 * torch.randn_like(m.weight))
                    m.bias.add_(0.1 * torch.randn_like(m.bias))
    torch.random.set_rng_state(st)
    return net.eval()
