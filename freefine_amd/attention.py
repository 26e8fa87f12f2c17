"""Host-side mirror of the reference's attention-control operator API (/root/reference/src/utils/attention.py):
`Attention_Modulator`, `AttentionControl`, `AttentionStore`, `register_attention_control{,_4bggen,_compose}` keep their
names, attributes and call pattern, but instead of monkey-patching torch modules and materialising [4*heads,S,S]
masks they emit, per attention call, a PASS TABLE for the fused HIP kernel ffn_attn (include/freefine_hip.h):
byte vectors per key / per query, batch-row remaps and scalar weights.

Dispatch conditions mirror ca_forward.forward (attention.py:388-404 edit, 273-290 bggen, 502-516 compose); the
`cur_att_layer` counting protocol (attention.py:674-680, 1051-1058, 1086-1090) is kept so `cur_att_layer // 2` is the
transformer-block index tested against `layer_idx`.
"""
import abc
import math

import torch
import torch.nn.functional as F

from . import _lib as L
from .ops import AttnEntrySpec


class AttentionControl(abc.ABC):
    def step_callback(self, x_t):
        return x_t

    def between_steps(self):
        return

    @property
    def num_uncond_att_layers(self):
        return self.num_att_layers if self.LOW_RESOURCE else 0

    def reset(self):
        self.cur_step = 0
        self.cur_att_layer = 0

    def __init__(self):
        self.cur_step = 0
        self.num_att_layers = -1
        self.cur_att_layer = 0
        self.LOW_RESOURCE = False


class AttentionStore(AttentionControl):
    """kept for import compatibility (freefine_batch_infer_2d.py:8); the fused kernels never materialise probabilities,
    so there is nothing to store -- dead code on the reference's path as well."""

    def reset(self):
        super().reset()
        self.step_store, self.attention_store = {}, {}

    def __init__(self):
        super().__init__()
        self.step_store, self.attention_store = {}, {}


def _down_hw(h, w, seq):
    """get_down_h_w (attention.py:713-733) for d_ratio = 2**round(log2(sqrt(h*w/seq)))."""
    d_ratio = 2 ** int(math.log2((h * w // seq) ** 0.5) + 0.5)
    r = d_ratio // 8
    nh, nw = h // 8, w // 8
    while r != 1:
        r //= 2
        nh, nw = (nh + 1) // 2, (nw + 1) // 2
    assert nh * nw == seq, f"{nh} * {nw} != {seq}"
    return nh, nw


def _resize_flat(mask, seq, normalise):
    """process_mask_before_attention (attention.py:841-855) / the raw resize of the cross-attn path (:1364-1371);
    evaluated on the host in the mask's own dtype, exactly as torch would."""
    m = mask.detach().cpu()
    if normalise and m.max() > 1:
        m = (m / m.max()).to(m.dtype)
    nh, nw = _down_hw(m.shape[0], m.shape[1], seq)
    return F.interpolate(m[None, None], size=(nh, nw), mode="nearest")[0, 0].flatten()


class Attention_Modulator(AttentionControl):
    _MASK_ATTRS = ("fg_retain_mask", "fg_retain_mask_st2", "fg_ref_mask", "obj_mask", "local_edit_region", "src_masks", "tgt_masks")

    def __setattr__(self, name, value):
        if name in Attention_Modulator._MASK_ATTRS:      # any (re)assignment of a mask invalidates the cached device vectors
            object.__setattr__(self, "_mask_epoch", getattr(self, "_mask_epoch", 0) + 1)
        object.__setattr__(self, name, value)

    def __init__(self, start_layer=None):
        super().__init__()
        self.step_num = 0
        self.model_type = "Inverse"
        self.use_cfg = False
        self.use_style_align = False
        self.use_tca = False
        self.local_edit = False
        self.fg_retain_mask = None
        self.fg_retain_mask_st2 = None
        self.fg_ref_mask = None
        self.obj_mask = None
        self.local_edit_region = None
        self.layer_idx = list(range(start_layer, 16)) if start_layer is not None else list(range(16))
        self.down_sampling_shape = dict()
        self.method = None
        self.context_guidance = None
        self.tca_scope = ["up"]
        self.style_align_scope = ["down", "mid", "up"]
        self.src_masks = None
        self.tgt_masks = None
        self.prompt_length = None
        self._vec_cache = {}

    def reset(self):
        super().reset()
        self.step_num = 0
        self.model_type = "Inverse"
        self.use_cfg = False
        self.use_tca = False
        self.use_style_align = False
        self.local_edit = False
        self.down_sampling_shape = dict()
        self.style_align = False
        self.method = None
        self.context_guidance = None
        self.tca_scope = ["up"]
        self.style_align_scope = ["down", "mid", "up"]

    def __call__(self, attn, is_cross, place_in_unet):
        self._tick()
        return attn

    def forward(self, attn, is_cross, place_in_unet):
        return attn

    def _tick(self):
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers + self.num_uncond_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
            self.between_steps()

    # ---- mask vectors: STATIC device buffers per (role, sequence length), refreshed in place when the mask changes,
    #      so a captured hipGraph of the UNet forward stays valid from one edit to the next -------------------------
    def _vectors(self, role, mask, seq, device, kind):
        key = (role, seq, kind, str(device))
        tag = (self._mask_epoch, mask.data_ptr(), mask._version, tuple(mask.shape), mask.dtype)
        ent = self._vec_cache.get(key)
        if ent is not None and ent["tag"] == tag:
            return ent["out"]
        if kind == "key":      # FG keys: bytes (m==1); counts for the uniform-softmax degenerate case
            m = _resize_flat(mask, seq, True)
            vals = set(m.unique().tolist())
            if not vals <= {0, 1}:
                raise NotImplementedError(f"non-binary attention mask values {sorted(vals)}: the reference would add them as "
                                          "score biases (attention.py:856-858); only {0,1} masks are supported by the fused kernel")
            fg = (m == 1).to(torch.uint8)
            host = dict(fg=fg, bg=1 - fg, n1=int(fg.sum()), n0=int((1 - fg).sum()))
        elif kind == "query_norm":   # T = resized, normalised target mask per query (TCA): float + byte selector
            m = _resize_flat(mask, seq, True)
            host = dict(f=m.float(), sel=(m > 0).to(torch.uint8), binary=bool(set(m.unique().tolist()) <= {0, 1}))
        elif kind == "query_raw":    # R and (1 - R) for the local cross-attn blend, (1-R) in the mask's own dtype
            m = _resize_flat(mask, seq, False)
            host = dict(f=m.float(), omf=(1 - m).float())
        else:
            raise ValueError(kind)
        if ent is None:
            ent = dict(out={k: (v.to(device) if torch.is_tensor(v) else v) for k, v in host.items()})
            self._vec_cache[key] = ent
        else:
            for k, v in host.items():
                if torch.is_tensor(v):
                    ent["out"][k].copy_(v)
                else:
                    ent["out"][k] = v
        ent["tag"] = tag
        return ent["out"]

    @staticmethod
    def _kflags(v, sel1=True, sel0=False):
        f = 0
        if sel1 and v["n1"] == 0:
            f |= L.ATT_UNIFORM_SEL1
        if sel0 and v["n0"] == 0:
            f |= L.ATT_UNIFORM_SEL0
        return f

    # ---- the dispatcher -------------------------------------------------------------------------------------------
    def plan(self, hook, is_cross, place, B, S, heads, device):
        """Returns dict(kind=..., passes=..., needs_cg=bool) for this attention call and advances the counter.
        kind: 'passes' (general multi-pass kernel call on the layer's own q/k/v) or 'shared_kv' (SSA/SDSA: keys/values
        = concat(own, reference row), then one masked pass)."""
        block = self.cur_att_layer // 2
        out = None
        if hook == "edit" and (not is_cross) and self.use_style_align and place in self.style_align_scope:
            out = self._plan_shared_kv(B, S, device)
        elif (not is_cross) and self.use_tca and place in (self.tca_scope if hook != "bggen" else ["up"]):
            if block in self.layer_idx:
                out = {"edit": self._plan_tca_edit, "bggen": self._plan_tca_bg, "compose": self._plan_tca_compose}[hook](B, S, device)
        elif is_cross and self.local_edit:
            out = self._plan_cross_compose(B, S, device) if hook == "compose" else self._plan_cross_local(B, S, device)
        if out is None:
            out = dict(kind="passes", passes=None, needs_cg=False, branch="plain")
        self._tick()
        return out

    def _plan_tca_edit(self, B, S, device):
        assert B == 4, "edit hook expects rows [u_e, u_r, c_e, c_r]"
        kv = self._vectors("fg_ref", self.fg_ref_mask, S, device, "key")
        qv = self._vectors("fg_retain", self.fg_retain_mask, S, device, "query_norm")
        ref_rows = [1, 1, 3, 3]
        flags = L.ATT_HEAD_RULE | self._kflags(kv, True, True)
        if self.method == "tca":
            p_ref = [AttnEntrySpec(b, ref_rows[b], 0.0, 1.0, kmask=kv["fg"], qsel=qv["sel"], flags=flags) for b in range(B)]
            p_self = [AttnEntrySpec(b, b, 1.0, -1.0) for b in range(B)]
            return dict(kind="passes", passes=[p_ref, p_self], needs_cg=True, branch="tca:tca")
        if self.method == "mmsa":
            if not qv["binary"]:
                raise NotImplementedError("mmsa with a non-binary target mask")
            p_ref = [AttnEntrySpec(b, ref_rows[b], 1.0, 0.0, kmask=kv["fg"], qsel=qv["sel"], flags=flags) for b in range(B)]
            return dict(kind="passes", passes=[p_ref], needs_cg=False, branch="tca:mmsa")
        raise ValueError(self.method)

    def _plan_tca_bg(self, B, S, device):
        assert B == 4
        kv = self._vectors("fg_retain", self.fg_retain_mask, S, device, "key")   # the hole; keys allowed OUTSIDE it
        ref_rows = [1, 1, 3, 3]
        flags = L.ATT_HEAD_RULE | (L.ATT_UNIFORM_SEL1 if kv["n0"] == 0 else 0)
        if self.method == "tca":
            p_ref = [AttnEntrySpec(b, ref_rows[b], 0.0, 1.0, kmask=kv["bg"], flags=flags) for b in range(B)]
            p_self = [AttnEntrySpec(b, b, 1.0, -1.0) for b in range(B)]
            return dict(kind="passes", passes=[p_ref, p_self], needs_cg=True, branch="tca:tca")
        if self.method == "mmsa":
            p_ref = [AttnEntrySpec(b, ref_rows[b], 1.0, 0.0, kmask=kv["bg"], flags=flags) for b in range(B)]
            return dict(kind="passes", passes=[p_ref], needs_cg=False, branch="tca:mmsa")
        raise ValueError(self.method)

    def _plan_tca_compose(self, B, S, device):
        R = B - 2
        if R + 1 > L.ATT_MAXP:
            raise NotImplementedError(f"compose with {R} references needs {R + 1} passes (max {L.ATT_MAXP})")
        tca = self.method == "tca"
        edit_rows = (0, B - 1)
        p_self = [AttnEntrySpec(b, b, 1.0, -1.0 if tca else 0.0) if b in edit_rows else AttnEntrySpec(b, b) for b in range(B)]
        if not tca:
            p_self = [None if b in edit_rows else AttnEntrySpec(b, b) for b in range(B)]
        passes = [p_self]
        for i in range(R):
            kv = self._vectors(f"src{i}", self.src_masks[i], S, device, "key")
            qv = self._vectors(f"tgt{i}", self.tgt_masks[i], S, device, "query_norm")
            fl = self._kflags(kv, True, False)
            w = (0.0, 1.0) if tca else (1.0, 0.0)
            passes.append([AttnEntrySpec(b, 1 + i, w[0], w[1], wq=qv["f"], kmask=kv["fg"], flags=fl) if b in edit_rows else None
                           for b in range(B)])
        return dict(kind="passes", passes=passes, needs_cg=tca, branch="tca:" + self.method)

    def _plan_cross_local(self, B, S, device):
        assert B == 4
        rv = self._vectors("local_edit", self.local_edit_region, S, device, "query_raw")
        p0 = [AttnEntrySpec(0, 0), AttnEntrySpec(1, 1), AttnEntrySpec(2, 2, wq=rv["f"]), AttnEntrySpec(1, 1)]
        p1 = [None, None, AttnEntrySpec(0, 0, wq=rv["omf"]), None]
        return dict(kind="passes", passes=[p0, p1], needs_cg=False, branch="cross_local")

    def _plan_cross_compose(self, B, S, device):
        P = self.prompt_length
        if P > L.ATT_MAXP:
            raise NotImplementedError(f"compose with {P} prompts needs {P} passes (max {L.ATT_MAXP})")
        nu = B - 1
        passes = []
        for p in range(P):
            rv = self._vectors(f"tgt{p}", self.tgt_masks[p], S, device, "query_raw")
            rows = [AttnEntrySpec(b, b) if p == 0 else None for b in range(nu)]
            rows.append(AttnEntrySpec(nu, nu + p, wq=rv["f"]))
            passes.append(rows)
        return dict(kind="passes", passes=passes, needs_cg=False, branch="cross_local")

    def _plan_shared_kv(self, B, S, device):
        ref_rows = [1] * (B // 2) + [B // 2 + 1] * (B // 2)
        kmask, flags = None, 0
        if self.method == "sdsa":
            kv = self._vectors("fg_ref", self.fg_ref_mask, S, device, "key")
            buf = self._vec_cache.setdefault(("sdsa_cat", S, str(device)), torch.ones(2 * S, dtype=torch.uint8, device=device))
            buf[S:].copy_(kv["fg"])
            kmask = buf
            flags = L.ATT_HEAD_RULE
        passes = [[AttnEntrySpec(b, b, kmask=kmask, flags=flags) for b in range(B)]]
        return dict(kind="shared_kv", passes=passes, ref_rows=ref_rows, needs_cg=False, branch=self.method)


def _register(model, controller, hook):
    unet = model.unet
    unet.set_attention_control(hook, controller)
    controller.num_att_layers = unet.num_attention_calls


def register_attention_control(model, controller):
    _register(model, controller, "edit")


def register_attention_control_4bggen(model, controller):
    _register(model, controller, "bggen")


def register_attention_control_compose(model, controller):
    _register(model, controller, "compose")
