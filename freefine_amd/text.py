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
