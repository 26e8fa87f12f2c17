"""SD AutoencoderKL encode (-> latent mean) / decode on the HIP kernels: the VAE bracket of the hot path
(/root/reference/src/demo/model.py:223-280, image2latent / latent2image).

Same design as unet.py: NHWC activations, 3x3 convs as implicit GEMM (the encoder's asymmetric (0,1,0,1) padding is the
kernel's pad=0 / stride=2 geometry, the decoder's nearest-2x upsample is fused into the next conv's gather), GroupNorm
+SiLU kernels, and the single-head mid-block attention (head dim 512, beyond the fused kernel's 160) as
GEMM -> row softmax -> GEMM with K / V^T used directly as the GEMMs' W operands.  The 0.18215 latent scale is folded into
quant_conv / post_quant_conv at pack time.
"""
import os

import torch

from . import ops
from .config import VAEConfig


class _O:
    pass


class HipVAE:
    def __init__(self, cfg: VAEConfig, state, dtype=torch.bfloat16, device="cuda:0", x3=False):
        """x3 (with dtype float32): split-bf16 mode -- fp32 activations, every conv / Linear an FFN_BF16X3 GEMM, norms write the pair
        form; the mid-block attention's two activation-by-activation products stay on the exact-fp32 GEMM (their "weights" are
        activations, which have no packed [hi | lo | hi] form)."""
        assert not x3 or dtype == torch.float32
        self.x3 = bool(x3)
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        self.G = cfg.norm_num_groups
        self._pack(state)

    def _f32(self, t):
        return t.detach().float().contiguous().to(self.device)

    def _conv(self, st, name, cin_pad=None, cout_pad=None):
        w, b = st[name + ".weight"].float().to(self.device), st[name + ".bias"].float().to(self.device)
        if cout_pad and cout_pad > w.shape[0]:
            w = torch.cat([w, torch.zeros(cout_pad - w.shape[0], *w.shape[1:], device=self.device)], 0)
            b = torch.cat([b, torch.zeros(cout_pad - b.shape[0], device=self.device)], 0)
        return ops.pack_conv3x3(w, self.dtype, cin_pad, self.x3), b.contiguous(), (cin_pad or w.shape[1])

    def _lin(self, st, name):
        w, b = st[name + ".weight"].float().to(self.device), st[name + ".bias"].float().to(self.device)
        if w.ndim == 4:
            w = w.reshape(w.shape[0], w.shape[1])
        return ops.pack_linear(w, self.dtype, self.x3), b.contiguous(), w.shape[1]

    def _res(self, st, p):
        r = _O()
        r.n1 = (self._f32(st[p + ".norm1.weight"]), self._f32(st[p + ".norm1.bias"]))
        r.c1 = self._conv(st, p + ".conv1")
        r.n2 = (self._f32(st[p + ".norm2.weight"]), self._f32(st[p + ".norm2.bias"]))
        r.c2 = self._conv(st, p + ".conv2")
        r.cout = st[p + ".conv1.weight"].shape[0]
        r.sc = self._lin(st, p + ".conv_shortcut") if (p + ".conv_shortcut.weight") in st else None
        return r

    def _attn(self, st, p):
        a = _O()
        a.gn = (self._f32(st[p + ".group_norm.weight"]), self._f32(st[p + ".group_norm.bias"]))
        a.q, a.k, a.v, a.o = (self._lin(st, f"{p}.{n}") for n in ("to_q", "to_k", "to_v", "to_out.0"))
        return a

    def _mid(self, st, p):
        m = _O()
        m.r0, m.attn, m.r1 = self._res(st, p + ".resnets.0"), self._attn(st, p + ".attentions.0"), self._res(st, p + ".resnets.1")
        return m

    def _pack(self, st):
        cfg = self.cfg
        e = 8 if self.x3 else ops.epc(self.dtype)
        s = cfg.scaling_factor
        n = len(cfg.block_out_channels)
        self.img_cp = (cfg.in_channels + e - 1) // e * e
        self.enc_in = self._conv(st, "encoder.conv_in", self.img_cp)
        self.enc_down = []
        for i in range(n):
            b = _O()
            b.res = [self._res(st, f"encoder.down_blocks.{i}.resnets.{j}") for j in range(cfg.layers_per_block)]
            b.down = self._conv(st, f"encoder.down_blocks.{i}.downsamplers.0.conv") if i < n - 1 else None
            self.enc_down.append(b)
        self.enc_mid = self._mid(st, "encoder.mid_block")
        self.enc_norm = (self._f32(st["encoder.conv_norm_out.weight"]), self._f32(st["encoder.conv_norm_out.bias"]))
        self.enc_out = self._conv(st, "encoder.conv_out")                       # -> 2*latent channels
        lc = cfg.latent_channels
        self.lat_cp = (2 * lc + e - 1) // e * e
        qw, qb = st["quant_conv.weight"].float().reshape(2 * lc, 2 * lc), st["quant_conv.bias"].float()
        # only the mean rows are needed (latent_dist.mean, model.py:267); scale folded in; N padded to a multiple of 4
        self.quant = (ops.pack_linear((qw[:lc] * s).to(self.device), self.dtype, self.x3), (qb[:lc] * s).to(self.device).contiguous(), 2 * lc)
        self.zin_cp = (lc + e - 1) // e * e
        pw = torch.zeros(self.zin_cp, self.zin_cp)
        pw[:lc, :lc] = st["post_quant_conv.weight"].float().reshape(lc, lc) / s          # decode(z / 0.18215): 1/s folded in
        pb = torch.zeros(self.zin_cp)
        pb[:lc] = st["post_quant_conv.bias"].float()
        self.post_quant = (ops.pack_linear(pw.to(self.device), self.dtype, self.x3), pb.to(self.device).contiguous(), self.zin_cp)
        self.dec_in = self._conv(st, "decoder.conv_in", self.zin_cp)
        self.dec_mid = self._mid(st, "decoder.mid_block")
        self.dec_up = []
        for i in range(n):
            b = _O()
            b.res = [self._res(st, f"decoder.up_blocks.{i}.resnets.{j}") for j in range(cfg.layers_per_block + 1)]
            b.up = self._conv(st, f"decoder.up_blocks.{i}.upsamplers.0.conv") if i < n - 1 else None
            b.up2 = None                                    # sub-pixel form of `nearest-2x -> 3x3 conv` (4/9 of the FLOPs, ops.pack_conv3x3_up2x)
            if b.up is not None and (self.dtype == torch.bfloat16 or self.x3) and os.environ.get("FFN_UP2X", "1") != "0":      # same switch as HipUNet
                wu = st[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"].float().to(self.device)
                if ops.up2x_eligible(wu.shape[1], wu.shape[0], 192):
                    b.up2 = ops.pack_conv3x3_up2x(wu, self.dtype, x3=self.x3)
            self.dec_up.append(b)
        self.dec_norm = (self._f32(st["decoder.conv_norm_out.weight"]), self._f32(st["decoder.conv_norm_out.bias"]))
        self.dec_out = self._conv(st, "decoder.conv_out", cout_pad=4)

    # ------------------------------------------------------------------------------------------------------------
    def _gn(self, x, gb, silu):
        return ops.groupnorm(x, gb[0], gb[1], self.G, 1e-6, silu=silu, pair=self.x3)

    def _resblock(self, r, x, B, H, W):
        cin = x.shape[-1]
        h = self._gn(x, r.n1, True)
        h = ops.conv3x3(h, r.c1[0], r.c1[1], B, H, W, cin)
        h = self._gn(h, r.n2, True)
        if r.sc is not None:
            x = ops.linear(x, r.sc[0], r.sc[1], K=cin)
        return ops.conv3x3(h, r.c2[0], r.c2[1], B, H, W, r.cout, residual=x)

    def _attention(self, a, x, B, S):
        C = x.shape[-1]
        y = self._gn(x, a.gn, False)
        q = ops.linear(y, a.q[0], a.q[1], K=C)
        k = ops.linear(y, a.k[0], a.k[1], K=C)
        ld = (S + 7) // 8 * 8
        vt = ops.linear(y, a.v[0], a.v[1], K=C, rows_per_batch=S, transposed_ld=ld)      # [B,C,ld]
        o = torch.empty(B, S, C, dtype=self.dtype, device=self.device)
        for b in range(B):
            scores = ops.linear(q[b], k[b], None, K=C)                                   # [S,S] = q k^T   (k as the W operand)
            p = ops.softmax_rows(scores, C ** -0.5)
            ops.linear(p, vt[b], None, K=S, out=o[b])                                    # P V           (V^T as the W operand)
        return ops.linear(o, a.o[0], a.o[1], K=C, residual=x)

    def _midblock(self, m, x, B, H, W):
        x = self._resblock(m.r0, x, B, H, W)
        x = self._attention(m.attn, x, B, H * W)
        return self._resblock(m.r1, x, B, H, W)

    @torch.no_grad()
    def encode_mean_scaled(self, img_u8=None, x_nchw=None):
        """uint8 [B,H,W,3] (or float NCHW in [-1,1]) -> latent_dist.mean * 0.18215 as fp32 NCHW [B,4,H/8,W/8]."""
        if img_u8 is not None:
            B, H, W, _ = img_u8.shape
            x = ops.image_to_nhwc(img_u8.to(self.device).contiguous(), self.img_cp, self.dtype)
        else:
            B, _, H, W = x_nchw.shape
            x = ops.pack_nchw(x_nchw.to(self.device, torch.float32).contiguous(), list(range(B)), self.img_cp, self.dtype)
        x = ops.conv3x3(x, self.enc_in[0], self.enc_in[1], B, H, W, self.img_cp)
        for blk in self.enc_down:
            for r in blk.res:
                x = self._resblock(r, x, B, H, W)
            if blk.down is not None:
                C = x.shape[-1]
                x = ops.conv3x3(x, blk.down[0], blk.down[1], B, H, W, C, stride=2, pad=0, Hout=H // 2, Wout=W // 2)
                H, W = H // 2, W // 2
        x = self._midblock(self.enc_mid, x, B, H, W)
        C = x.shape[-1]
        x = self._gn(x, self.enc_norm, True)
        x = ops.conv3x3(x, self.enc_out[0], self.enc_out[1], B, H, W, C)
        z = ops.linear(x, self.quant[0], self.quant[1], K=self.quant[2], out_f32=True)     # [B,HW,4] fp32, already * 0.18215
        return ops.nhwc_to_nchw_f32(z, self.cfg.latent_channels, H, W)

    @torch.no_grad()
    def decode_image(self, latents):
        """latents fp32 NCHW [B,4,h,w] (scaled) -> clamp(decode(z/0.18215)/2+0.5, 0, 1) as fp32 NCHW [B,3,8h,8w]."""
        latents = latents.to(self.device, torch.float32).contiguous()
        B, _, H, W = latents.shape
        z = ops.pack_nchw(latents, list(range(B)), self.zin_cp, self.dtype)
        x = ops.linear(z, self.post_quant[0], self.post_quant[1], K=self.zin_cp)
        x = ops.conv3x3(x, self.dec_in[0], self.dec_in[1], B, H, W, self.zin_cp)
        x = self._midblock(self.dec_mid, x, B, H, W)
        for blk in self.dec_up:
            for r in blk.res:
                x = self._resblock(r, x, B, H, W)
            if blk.up is not None:
                C = x.shape[-1]
                if blk.up2 is not None and ops.up2x_eligible(C, C, B * H * W):
                    x = ops.conv3x3_up2x(x, blk.up2, blk.up[1], B, H, W, C)
                else:
                    x = ops.conv3x3(x, blk.up[0], blk.up[1], B, H, W, C, upsample=True)
                H, W = 2 * H, 2 * W
        C = x.shape[-1]
        x = self._gn(x, self.dec_norm, True)
        x = ops.conv3x3(x, self.dec_out[0], self.dec_out[1], B, H, W, C)                   # [B,HW,4] (3 real channels)
        return ops.nhwc_to_image(x, H, W)
