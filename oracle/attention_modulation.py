"""ORACLE (test infrastructure, never imported by the product path).

CPU restatement of FreeFine's attention modulation (/root/reference/src/utils/attention.py:640-1443) as PURE functions
of (q, k, v, masks, method, context_guidance) plus a small dispatcher object that reproduces the reference's
call-counting protocol (`cur_att_layer`, attention.py:674-680, 1051-1058, 1086-1090).

Math (verified against the imported reference to <= 4e-7, fixtures tests/golden/g1_*.npz):
  A(q,K,V,m) = softmax(scale * q K^T + mu(m)) V, mu = 0 where m == 1, finfo.min where m == 0 (additive, so an
  all-masked row degenerates to a uniform softmax);  rows of a head-batched tensor are j = b*heads + head and the
  reference tiles its masks with .repeat(heads,1,1) (attention.py:859) so they apply iff j is even.
"""
import math

import torch
import torch.nn.functional as F


def heads_split(t, heads):  # [B,S,C] -> [B,h,S,d]     (head_to_batch_dim, attention.py:758-767, kept 4-D)
    b, s, c = t.shape
    return t.reshape(b, s, heads, c // heads).permute(0, 2, 1, 3)


def heads_merge(t):  # [B,h,S,d] -> [B,S,C]            (batch_to_head_dim, attention.py:768-773)
    b, h, s, d = t.shape
    return t.permute(0, 2, 1, 3).reshape(b, s, h * d)


def attn_core(q, k, v, scale, bias=None):
    """q [..,S,d], k,v [..,Sk,d], bias broadcastable to [..,S,Sk] (additive, attention.py:789-795)."""
    s = scale * (q @ k.transpose(-1, -2))
    if bias is not None:
        s = s + bias
    return s.softmax(dim=-1) @ v


def plain_attention(q, k, v, heads, scale):
    return heads_merge(attn_core(heads_split(q, heads), heads_split(k, heads), heads_split(v, heads), scale))


def downsample_mask(mask, seq):
    """process_mask_before_attention (attention.py:841-855): normalise >1 masks by their max IN THEIR OWN DTYPE,
    nearest-resize a H x W mask to the sqrt(seq) grid; returns the flattened mask (dtype preserved)."""
    if mask.max() > 1:
        mask = (mask / mask.max()).to(mask.dtype)
    h, w = mask.shape
    d_ratio = 2 ** int(math.log2((h * w // seq) ** 0.5) + 0.5)
    # get_down_h_w (attention.py:713-733): latent = //8, then ceil-halving until the ratio is met
    r = d_ratio // 8
    nh, nw = h // 8, w // 8
    while r != 1:
        r //= 2
        nh, nw = (nh + 1) // 2, (nw + 1) // 2
    assert nh * nw == seq, f"{nh}*{nw} != {seq}"
    m = F.interpolate(mask[None, None], size=(nh, nw), mode="nearest")[0, 0]
    return m.flatten()


def key_bias(mask_vec, dtype):
    """post_process_attn_mask (attention.py:856-858): values == 0 -> finfo.min, == 1 -> 0, anything else stays."""
    b = mask_vec.to(dtype).clone()
    zero, one = b == 0, b == 1
    b[zero] = torch.finfo(dtype).min
    b[one] = 0
    return b


def even_j(b_rows, heads):
    """[B,h] bool: the tiled-head rule -- masks built by cat(...).repeat(heads,1,1) hit row j iff j % 4 in {0,2}."""
    j = torch.arange(b_rows * heads).reshape(b_rows, heads)
    return (j % 2) == 0


def tca_edit(q, k, v, heads, scale, fg_retain_mask, fg_ref_mask, method, context_guidance):
    """Temporal_contextal_attention (attention.py:1043-1091), batch rows [u_e, u_r, c_e, c_r]."""
    B, S, _ = q.shape
    assert B == 4
    qh, kh, vh = heads_split(q, heads), heads_split(k, heads), heads_split(v, heads)
    ref_rows = [1, 1, 3, 3]                      # cross_manner_attention_modulate (attention.py:1033-1035)
    kr, vr = kh[ref_rows], vh[ref_rows]
    src = downsample_mask(fg_ref_mask, S)        # per key
    tgt = downsample_mask(fg_retain_mask, S)     # per query
    dt = q.dtype
    ones = torch.ones(S, dtype=dt)
    bias_fg = key_bias(ones * src, dt)           # FG_mask = ones * ref_mask (attention.py:872)
    bias_bg = key_bias(ones * (1 - src), dt)     # 1 - ref_mask in the MASK's dtype (uint8 wraps), attention.py:873
    ev = even_j(B, heads)[:, :, None, None]
    zero = torch.zeros(S, dtype=dt)
    b_fg = torch.where(ev, bias_fg[None, None, None, :], zero[None, None, None, :])
    b_bg = torch.where(ev, bias_bg[None, None, None, :], zero[None, None, None, :])
    out_fg = attn_core(qh, kr, vr, scale, b_fg)
    out_bg = attn_core(qh, kr, vr, scale, b_bg)
    T = torch.where(ev[:, :, :, 0], tgt[None, None, :].to(dt), torch.ones(1, 1, S, dtype=dt))  # final_mask_fg, attention.py:880-881
    if method == "mmsa":
        hidden = T[..., None] * out_fg + (1 - T)[..., None] * out_bg
    elif method == "tca":
        T = (T > 0).to(dt)                       # attention.py:1071
        ref_hidden = T[..., None] * out_fg + (1 - T)[..., None] * out_bg
        self_hidden = attn_core(qh, kh, vh, scale)
        hidden = ref_hidden * context_guidance + self_hidden * (1 - context_guidance)
    else:
        raise ValueError(method)
    return heads_merge(hidden)


def tca_bg(q, k, v, heads, scale, hole_mask, method, context_guidance):
    """Temporal_contextal_attention_bg (attention.py:1284-1324): keys restricted to OUTSIDE the hole."""
    B, S, _ = q.shape
    assert B == 4
    qh, kh, vh = heads_split(q, heads), heads_split(k, heads), heads_split(v, heads)
    ref_rows = [1, 1, 3, 3]
    kr, vr = kh[ref_rows], vh[ref_rows]
    hole = downsample_mask(hole_mask, S)
    dt = q.dtype
    bias_bg = key_bias(torch.ones(S, dtype=dt) * (1 - hole), dt)
    ev = even_j(B, heads)[:, :, None, None]
    b_bg = torch.where(ev, bias_bg[None, None, None, :], torch.zeros(1, 1, 1, S, dtype=dt))
    out_bg = attn_core(qh, kr, vr, scale, b_bg)
    if method == "mmsa":
        hidden = out_bg
    elif method == "tca":
        hidden = attn_core(qh, kh, vh, scale) * (1 - context_guidance) + out_bg * context_guidance
    else:
        raise ValueError(method)
    return heads_merge(hidden)


def tca_compose(q, k, v, heads, scale, src_masks, tgt_masks, method, context_guidance):
    """Temporal_contextal_attention_compose (attention.py:1092-1140): rows [e_u, r_1..r_R, e_c]; no head rule."""
    B, S, _ = q.shape
    R = B - 2
    qh, kh, vh = heads_split(q, heads), heads_split(k, heads), heads_split(v, heads)
    self_hidden = attn_core(qh, kh, vh, scale)
    dt = q.dtype
    new = {0: torch.zeros_like(self_hidden[0]), B - 1: torch.zeros_like(self_hidden[0])}
    for i in range(R):
        src = downsample_mask(src_masks[i], S)
        tgt = downsample_mask(tgt_masks[i], S)
        bias = key_bias(torch.ones(S, dtype=dt) * src, dt)[None, None, :]
        for row in (0, B - 1):
            new[row] = new[row] + tgt[None, :, None] * attn_core(qh[row], kh[1 + i], vh[1 + i], scale, bias)
    out = self_hidden.clone()
    for row in (0, B - 1):
        if method == "mmsa":
            out[row] = new[row]
        elif method == "tca":
            out[row] = new[row] * context_guidance + self_hidden[row] * (1 - context_guidance)
        else:
            raise ValueError(method)
    return heads_merge(out)


def shared_kv_attention(q, k, v, heads, scale, fg_ref_mask=None):
    """style_align_share_attention (attention.py:1142-1192): keys/values = concat(own, reference row) over 2S keys;
    SDSA masks the reference half to the source object (prepare_sdsa_mask, attention.py:940-951), even j only."""
    B, S, _ = q.shape
    ref_rows = [1] * (B // 2) + [B // 2 + 1] * (B // 2)
    k2, v2 = torch.cat([k, k[ref_rows]], dim=1), torch.cat([v, v[ref_rows]], dim=1)
    qh, kh, vh = heads_split(q, heads), heads_split(k2, heads), heads_split(v2, heads)
    bias = None
    if fg_ref_mask is not None:
        m = downsample_mask(fg_ref_mask, S)
        dt = q.dtype
        full = torch.cat([torch.ones_like(m), m])
        kb = key_bias(torch.ones(2 * S, dtype=dt) * full, dt)
        ev = even_j(B, heads)[:, :, None, None]
        bias = torch.where(ev, kb[None, None, None, :], torch.zeros(1, 1, 1, 2 * S, dtype=dt))
    return heads_merge(attn_core(qh, kh, vh, scale, bias))


def cross_local(q, k, v, heads, scale, local_edit_region):
    """modulate_local_cross_attn / _bg (attention.py:1360-1393, 1326-1357): rows [X0, X1, R*X2+(1-R)*X0, X1]."""
    B, S, _ = q.shape
    x = heads_merge(attn_core(heads_split(q, heads), heads_split(k, heads), heads_split(v, heads), scale))
    h, w = local_edit_region.shape
    region = downsample_mask_raw(local_edit_region, S)
    mod = region[:, None] * x[2] + (1 - region)[:, None] * x[0]
    return torch.stack([x[0], x[1], mod, x[1]], dim=0)


def downsample_mask_raw(mask, seq):
    """the cross-attention variant resizes WITHOUT the >1 normalisation (attention.py:1364-1371)."""
    h, w = mask.shape
    d_ratio = 2 ** int(math.log2((h * w // seq) ** 0.5) + 0.5)
    r = d_ratio // 8
    nh, nw = h // 8, w // 8
    while r != 1:
        r //= 2
        nh, nw = (nh + 1) // 2, (nw + 1) // 2
    assert nh * nw == seq
    return F.interpolate(mask[None, None], size=(nh, nw), mode="nearest")[0, 0].flatten()


def cross_local_compose(q, k, v, heads, scale, tgt_masks, prompt_length):
    """modulate_local_cross_attn_compose (attention.py:1394-1432): q rows [e_u, r_1..r_R, e_c] (B), text rows B-1+P
    ([""]*(B-1) then P prompts incl. the trailing ""); last output row = sum_i tgt_i * A(q_ec, K_prompt_i, V_prompt_i)."""
    B, S, _ = q.shape
    qh, kh, vh = heads_split(q, heads), heads_split(k, heads), heads_split(v, heads)
    nu = B - 1
    hu = attn_core(qh[:nu], kh[:nu], vh[:nu], scale)
    hc = torch.zeros_like(qh[nu])
    for i in range(prompt_length):
        region = downsample_mask_raw(tgt_masks[i], S)
        hc = hc + region[None, :, None] * attn_core(qh[nu], kh[nu + i], vh[nu + i], scale)
    return heads_merge(torch.cat([hu, hc[None]], dim=0))


class Modulator:
    """Oracle-side stand-in for Attention_Modulator + ca_forward dispatch (attention.py:388-404, 273-290, 502-516).

    hook in {'edit', 'bggen', 'compose'}.  Keeps the reference's counter: every attention call bumps
    `cur_att_layer`; `cur_att_layer // 2` is the transformer-block index tested against `layer_idx`."""

    def __init__(self, hook="edit", num_att_layers=32):
        self.hook, self.num_att_layers = hook, num_att_layers
        self.layer_idx = list(range(16))
        self.reset()
        self.fg_retain_mask = self.fg_retain_mask_st2 = self.fg_ref_mask = self.local_edit_region = None
        self.src_masks = self.tgt_masks = None
        self.prompt_length = None
        self.trace = []  # (step, block, is_cross, branch) for G2-style tables

    def reset(self):
        self.cur_step = self.cur_att_layer = 0
        self.use_tca = self.use_style_align = self.local_edit = False
        self.method = None
        self.context_guidance = None
        self.tca_scope = ["up"]
        self.style_align_scope = ["down", "mid", "up"]

    def _tick(self):
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1

    def attend(self, q, k, v, heads, scale, is_cross, place):
        block = self.cur_att_layer // 2
        branch = "plain"
        if (not is_cross) and self.hook == "edit" and self.use_style_align and place in self.style_align_scope:
            branch = self.method
            out = shared_kv_attention(q, k, v, heads, scale, self.fg_ref_mask if self.method == "sdsa" else None)
        elif (not is_cross) and self.use_tca and place in self.tca_scope:
            if block not in self.layer_idx:
                out = plain_attention(q, k, v, heads, scale)
            else:
                branch = "tca:" + self.method
                if self.hook == "edit":
                    out = tca_edit(q, k, v, heads, scale, self.fg_retain_mask, self.fg_ref_mask, self.method, self.context_guidance)
                elif self.hook == "bggen":
                    out = tca_bg(q, k, v, heads, scale, self.fg_retain_mask, self.method, self.context_guidance)
                else:
                    out = tca_compose(q, k, v, heads, scale, self.src_masks, self.tgt_masks, self.method, self.context_guidance)
        elif is_cross and self.local_edit:
            branch = "cross_local"
            if self.hook == "compose":
                out = cross_local_compose(q, k, v, heads, scale, self.tgt_masks, self.prompt_length)
            else:
                out = cross_local(q, k, v, heads, scale, self.local_edit_region)
        else:
            out = plain_attention(q, k, v, heads, scale)
        self.trace.append((self.cur_step, block, bool(is_cross), place, branch))
        self._tick()
        return out
