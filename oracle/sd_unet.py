"""ORACLE (test infrastructure, never imported by the product path).

CPU restatement, in plain torch fp32, of the Stable-Diffusion UNet2DConditionModel arithmetic that the reference
reaches through un-vendored diffusers==0.18.0 (/root/reference/requirements.txt:5).  The reference only patches
`Attention.forward` and `unet.forward` (src/utils/attention.py:11-225, 226-564); the block arithmetic below restates
the published diffusers-0.18 modules (ResnetBlock2D, Transformer2DModel, BasicTransformerBlock, Attention, GEGLU,
Downsample2D, Upsample2D, Timesteps, TimestepEmbedding).  Module and parameter names follow diffusers' state-dict
layout so HF safetensors load unchanged, and so that the reference's own hook registrars (which match modules by the
class name `Attention` and walk `unet.named_children()`, attention.py:433-452) can drive these modules when golden
vectors are generated (tools/gen_golden.py).

Parity pin: this restatement is checked against the importable in-tree CompVis UNet
(evaluation/MotionGuidance/ldm/modules/diffusionmodules/openaimodel.py:413) with key-mapped random weights
(tests/golden/g6_ldm_unet.npz).
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


def unet_config(name="sd21-base"):
    """Configs as plain namespaces (duck-typed: freefine_amd.config.UNetConfig has the same attributes)."""
    base = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                down_has_attn=(True, True, True, False), norm_num_groups=32, norm_eps=1e-5, flip_sin_to_cos=True,
                freq_shift=0.0, sample_size=64)
    if name == "sd15":
        base.update(cross_attention_dim=768, heads=(8, 8, 8, 8), use_linear_projection=False, upcast_attention=False)
    elif name == "sd21-base":
        base.update(cross_attention_dim=1024, heads=(5, 10, 20, 20), use_linear_projection=True, upcast_attention=False)
    elif name == "tiny":  # same topology, small widths: for loop-level fixtures and CPU tests
        base.update(block_out_channels=(32, 64, 128, 128), cross_attention_dim=64, heads=(2, 4, 4, 4),
                    use_linear_projection=True, upcast_attention=False, norm_num_groups=8, sample_size=16)
    elif name == "tiny-conv":  # SD-1.5 style: conv proj_in/out, constant head count
        base.update(block_out_channels=(32, 64, 128, 128), cross_attention_dim=48, heads=(4, 4, 4, 4),
                    use_linear_projection=False, upcast_attention=False, norm_num_groups=8, sample_size=16)
    else:
        raise ValueError(name)
    return SimpleNamespace(name=name, **base)


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels, self.flip, self.shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(self, timesteps):
        half = self.num_channels // 2
        exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / (half - self.shift)
        emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
        if self.flip:
            emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
        return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels else None
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Attention(nn.Module):
    """Attribute surface the reference hooks rely on (src/utils/attention.py:228-302): heads, scale, upcast_*,
    spatial_norm, group_norm, norm_cross, residual_connection, rescale_output_factor, to_q/k/v, to_out,
    prepare_attention_mask, head_to_batch_dim, batch_to_head_dim, get_attention_scores."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, bias=False, upcast_attention=False,
                 norm_num_groups=None, eps=1e-5, residual_connection=False):
        super().__init__()
        inner = heads * dim_head
        cross = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads, self.scale = heads, dim_head ** -0.5
        self.upcast_attention, self.upcast_softmax = upcast_attention, False
        self.spatial_norm, self.norm_cross = None, None
        self.group_norm = nn.GroupNorm(norm_num_groups, query_dim, eps=eps) if norm_num_groups else None
        self.residual_connection, self.rescale_output_factor = residual_connection, 1.0
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(cross, inner, bias=bias)
        self.to_v = nn.Linear(cross, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])
        self.modulator = None  # oracle-side hook (oracle/attention_modulation.py); None -> plain attention
        self.place, self.is_self = None, cross_attention_dim is None

    def prepare_attention_mask(self, attention_mask, target_length, batch_size):
        assert attention_mask is None
        return None

    def head_to_batch_dim(self, t):
        b, s, d = t.shape
        return t.reshape(b, s, self.heads, d // self.heads).permute(0, 2, 1, 3).reshape(b * self.heads, s, d // self.heads)

    def batch_to_head_dim(self, t):
        bh, s, d = t.shape
        return t.reshape(bh // self.heads, self.heads, s, d).permute(0, 2, 1, 3).reshape(bh // self.heads, s, d * self.heads)

    def get_attention_scores(self, query, key, attention_mask=None):
        scores = self.scale * torch.bmm(query, key.transpose(-1, -2))
        if attention_mask is not None:
            scores = scores + attention_mask
        return scores.softmax(dim=-1)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        residual = hidden_states
        ndim = hidden_states.ndim
        if ndim == 4:
            b, c, hh, ww = hidden_states.shape
            hidden_states = hidden_states.view(b, c, hh * ww).transpose(1, 2)
        if self.group_norm is not None:
            hidden_states = self.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
        is_cross = encoder_hidden_states is not None
        ctx = encoder_hidden_states if is_cross else hidden_states
        q, k, v = self.to_q(hidden_states), self.to_k(ctx), self.to_v(ctx)
        if self.modulator is not None:
            out = self.modulator.attend(q, k, v, self.heads, self.scale, is_cross, self.place)
        else:
            from .attention_modulation import plain_attention
            out = plain_attention(q, k, v, self.heads, self.scale)
        out = self.to_out[0](out)
        if ndim == 4:
            out = out.transpose(-1, -2).reshape(b, c, hh, ww)
        if self.residual_connection:
            out = out + residual
        return out / self.rescale_output_factor


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        x, gate = self.proj(x).chunk(2, dim=-1)
        return x * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_attention_dim, upcast_attention):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads, dim_head, upcast_attention=upcast_attention)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, cross_attention_dim, heads, dim_head, upcast_attention=upcast_attention)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, encoder_hidden_states=None, **kw):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), encoder_hidden_states=encoder_hidden_states) + x
        return self.ff(self.norm3(x)) + x


class Transformer2DModel(nn.Module):
    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, groups, use_linear_projection, upcast_attention):
        super().__init__()
        inner = heads * dim_head
        self.use_linear_projection = use_linear_projection
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner) if use_linear_projection else nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim, upcast_attention)])
        self.proj_out = nn.Linear(inner, in_channels) if use_linear_projection else nn.Conv2d(inner, in_channels, 1)

    def forward(self, x, encoder_hidden_states=None, **kw):
        b, c, h, w = x.shape
        residual = x
        x = self.norm(x)
        if not self.use_linear_projection:
            x = self.proj_in(x).permute(0, 2, 3, 1).reshape(b, h * w, -1)
        else:
            x = self.proj_in(x.permute(0, 2, 3, 1).reshape(b, h * w, c))
        for blk in self.transformer_blocks:
            x = blk(x, encoder_hidden_states=encoder_hidden_states)
        if not self.use_linear_projection:
            x = self.proj_out(x.reshape(b, h, w, -1).permute(0, 3, 1, 2))
        else:
            x = self.proj_out(x).reshape(b, h, w, -1).permute(0, 3, 1, 2)
        return x + residual


class Downsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x, output_size=None):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    """CrossAttnDownBlock2D (has_cross_attention=True) or DownBlock2D."""

    def __init__(self, cin, cout, temb, n_layers, groups, eps, attn, add_downsample):
        super().__init__()
        self.has_cross_attention = attn is not None
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb, groups, eps) for i in range(n_layers)])
        if attn is not None:
            self.attentions = nn.ModuleList([Transformer2DModel(attn["heads"], cout // attn["heads"], cout, attn["cross"], groups,
                                                                attn["linear"], attn["upcast"]) for _ in range(n_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_downsample else None

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, **kw):
        outs = ()
        for i, res in enumerate(self.resnets):
            hidden_states = res(hidden_states, temb)
            if self.has_cross_attention:
                hidden_states = self.attentions[i](hidden_states, encoder_hidden_states=encoder_hidden_states)
            outs += (hidden_states,)
        if self.downsamplers is not None:
            hidden_states = self.downsamplers[0](hidden_states)
            outs += (hidden_states,)
        return hidden_states, outs


class MidBlock(nn.Module):
    def __init__(self, c, temb, groups, eps, attn):
        super().__init__()
        self.has_cross_attention = True
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb, groups, eps), ResnetBlock2D(c, c, temb, groups, eps)])
        self.attentions = nn.ModuleList([Transformer2DModel(attn["heads"], c // attn["heads"], c, attn["cross"], groups,
                                                            attn["linear"], attn["upcast"])])

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, **kw):
        hidden_states = self.resnets[0](hidden_states, temb)
        hidden_states = self.attentions[0](hidden_states, encoder_hidden_states=encoder_hidden_states)
        return self.resnets[1](hidden_states, temb)


class UpBlock(nn.Module):
    def __init__(self, cin, cout, cprev, temb, n_layers, groups, eps, attn, add_upsample):
        super().__init__()
        self.has_cross_attention = attn is not None
        res = []
        for i in range(n_layers):
            skip = cin if i == n_layers - 1 else cout
            rin = cprev if i == 0 else cout
            res.append(ResnetBlock2D(rin + skip, cout, temb, groups, eps))
        self.resnets = nn.ModuleList(res)
        if attn is not None:
            self.attentions = nn.ModuleList([Transformer2DModel(attn["heads"], cout // attn["heads"], cout, attn["cross"], groups,
                                                                attn["linear"], attn["upcast"]) for _ in range(n_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, encoder_hidden_states=None, upsample_size=None, **kw):
        for i, res in enumerate(self.resnets):
            skip = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = res(torch.cat([hidden_states, skip], dim=1), temb)
            if self.has_cross_attention:
                hidden_states = self.attentions[i](hidden_states, encoder_hidden_states=encoder_hidden_states)
        if self.upsamplers is not None:
            hidden_states = self.upsamplers[0](hidden_states, upsample_size)
        return hidden_states


class UNet2DConditionModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.config = SimpleNamespace(center_input_sample=False, class_embed_type=None, addition_embed_type=None,
                                      class_embeddings_concat=False, in_channels=cfg.in_channels, sample_size=cfg.sample_size)
        ch = cfg.block_out_channels
        temb = ch[0] * 4
        g, eps = cfg.norm_num_groups, cfg.norm_eps
        self.in_channels = cfg.in_channels
        self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, padding=1)
        self.time_proj = Timesteps(ch[0], cfg.flip_sin_to_cos, cfg.freq_shift)
        self.time_embedding = TimestepEmbedding(ch[0], temb)
        self.class_embedding, self.time_embed_act, self.encoder_hid_proj = None, None, None

        def attn_cfg(i):
            return dict(heads=cfg.heads[i], cross=cfg.cross_attention_dim, linear=cfg.use_linear_projection, upcast=cfg.upcast_attention)

        n = len(ch)
        self.down_blocks = nn.ModuleList()
        cout = ch[0]
        for i in range(n):
            cin, cout = cout, ch[i]
            self.down_blocks.append(DownBlock(cin, cout, temb, cfg.layers_per_block, g, eps,
                                              attn_cfg(i) if cfg.down_has_attn[i] else None, add_downsample=i < n - 1))
        self.mid_block = MidBlock(ch[-1], temb, g, eps, attn_cfg(n - 1))
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(ch))
        rev_attn = list(reversed(cfg.down_has_attn))
        rev_heads = list(reversed(range(n)))
        cout = rev[0]
        self.num_upsamplers = 0
        for i in range(n):
            cprev, cout = cout, rev[i]
            cin = rev[min(i + 1, n - 1)]
            add_up = i < n - 1
            self.num_upsamplers += int(add_up)
            self.up_blocks.append(UpBlock(cin, cout, cprev, temb, cfg.layers_per_block + 1, g, eps,
                                          attn_cfg(rev_heads[i]) if rev_attn[i] else None, add_up))
        self.conv_norm_out = nn.GroupNorm(g, ch[0], eps=eps)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(ch[0], cfg.out_channels, 3, padding=1)
        for place, mod in (("down", self.down_blocks), ("mid", self.mid_block), ("up", self.up_blocks)):
            for m in mod.modules():
                if isinstance(m, Attention):
                    m.place = place

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def attention_modules(self):
        """Attention modules in execution order (down -> mid -> up), as the reference's counter sees them."""
        out = []
        for mod in (self.down_blocks, self.mid_block, self.up_blocks):
            out += [m for m in mod.modules() if isinstance(m, Attention)]
        return out

    def set_modulator(self, modulator):
        for m in self.attention_modules():
            m.modulator = modulator

    def forward(self, sample, timestep, encoder_hidden_states):
        """Restatement of diffusers-0.18 UNet2DConditionModel.forward as used by the reference (bare tensor out,
        src/utils/attention.py:214-223)."""
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.int64, device=sample.device)
        elif t.ndim == 0:
            t = t[None].to(sample.device)
        t = t.expand(sample.shape[0])
        emb = self.time_embedding(self.time_proj(t).to(sample.dtype))
        h = self.conv_in(sample)
        skips = (h,)
        for blk in self.down_blocks:
            h, outs = blk(hidden_states=h, temb=emb, encoder_hidden_states=encoder_hidden_states)
            skips += outs
        h = self.mid_block(h, emb, encoder_hidden_states=encoder_hidden_states)
        for blk in self.up_blocks:
            nres = len(blk.resnets)
            res, skips = skips[-nres:], skips[:-nres]
            h = blk(hidden_states=h, temb=emb, res_hidden_states_tuple=res, encoder_hidden_states=encoder_hidden_states)
        return self.conv_out(self.conv_act(self.conv_norm_out(h)))


def init_unet(cfg, seed=0, perturb_norms=True):
    """Deterministic random-init UNet (torch default inits from a seeded generator; norm affine params perturbed so
    parity tests notice a dropped gamma/beta)."""
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    net = UNet2DConditionModel(cfg)
    if perturb_norms:
        with torch.no_grad():
            for m in net.modules():
                if isinstance(m, (nn.GroupNorm, nn.LayerNorm)):
                    m.weight.add_(0.1 * torch.randn_like(m.weight))
                    m.bias.add_(0.1 * torch.randn_like(m.bias))
    torch.random.set_rng_state(gen_state)
    return net.eval()
