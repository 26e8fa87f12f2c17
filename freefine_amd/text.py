"""Text conditioning for the hot path.

The reference tokenises with CLIP and calls `self.text_encoder(ids)[0]` (/root/reference/src/demo/model.py:536-567, 842-848).
No CLIP weights exist in this environment, so synthetic runs use a deterministic byte-level stand-in with the same
interface pair (tokenizer(prompts, padding=..., max_length=77, return_tensors="pt").input_ids ; text_encoder(ids)[0]
-> [N,77,D]).  A real HF tokenizer/text-encoder pair can be plugged into FreeFinePipeline unchanged.
"""
from types import SimpleNamespace

import numpy as np
import torch


class ByteTokenizer:
    model_max_length = 77

    def __call__(self, prompts, padding="max_length", max_length=77, return_tensors="pt", **kw):
        if isinstance(prompts, str):
            prompts = [prompts]
        ids = np.zeros((len(prompts), max_length), dtype=np.int64)
        for i, p in enumerate(prompts):
            b = list(p.encode("utf-8"))[: max_length - 2]
            ids[i, 0] = 257                      # BOS
            ids[i, 1:1 + len(b)] = np.asarray(b, dtype=np.int64) + 1
            ids[i, 1 + len(b)] = 258             # EOS, then 0-padding
        return SimpleNamespace(input_ids=torch.from_numpy(ids))


class SyntheticTextEncoder:
    """ids [N,77] -> ([N,77,D],): seeded token table + position table, unit-RMS rows.  Host side, fp32, deterministic."""

    def __init__(self, dim, seed=1234, device="cpu"):
        rng = np.random.default_rng(seed)
        self.table = torch.from_numpy(rng.standard_normal((259, dim)).astype(np.float32))
        self.pos = torch.from_numpy((0.3 * rng.standard_normal((77, dim))).astype(np.float32))
        self.device = torch.device(device)
        self.dim = dim

    def to(self, device):
        self.device = torch.device(device)
        return self

    def __call__(self, input_ids):
        ids = input_ids.cpu()
        e = self.table[ids] + self.pos[None, : ids.shape[1]]
        e = e / e.pow(2).mean(dim=-1, keepdim=True).sqrt()
        return (e.to(self.device),)


def make_text_embed(tokenizer, text_encoder):
    """callable(list[str]) -> [N,77,D], the form the oracle consumes."""
    def f(prompts):
        return text_encoder(tokenizer(prompts, padding="max_length", max_length=77, return_tensors="pt").input_ids)[0]
    return f


def clip_shaped_text_encoder(dim, seed=1234, layers=None):
    """A REAL-SIZE text encoder for synthetic runs (bench.py): transformers' CLIPTextModel built from a config of the checkpoint's shape --
    SD-2.1's OpenCLIP ViT-H text tower (width 1024, 23 layers = the penultimate-layer output the checkpoint ships, 16 heads, MLP 4096, gelu),
    SD-1.x's CLIP ViT-L (768, 12 layers, 12 heads, 3072, quick_gelu) -- with seeded random weights (no checkpoints exist offline), so that the
    text side of an edit costs what it costs with a checkpoint.  torch module: FreeFinePipeline moves it to its device on first use."""
    from transformers import CLIPTextConfig, CLIPTextModel
    if dim == 1024:
        cfg = CLIPTextConfig(vocab_size=49408, hidden_size=1024, intermediate_size=4096, num_hidden_layers=layers or 23, num_attention_heads=16,
                             max_position_embeddings=77, hidden_act="gelu", projection_dim=1024)
    else:
        cfg = CLIPTextConfig(vocab_size=49408, hidden_size=dim, intermediate_size=4 * dim, num_hidden_layers=layers or 12,
                             num_attention_heads=max(1, dim // 64), max_position_embeddings=77, hidden_act="quick_gelu", projection_dim=dim)
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    enc = CLIPTextModel(cfg).eval()
    torch.random.set_rng_state(state)
    for q in enc.parameters():
        q.requires_grad_(False)
    return enc
