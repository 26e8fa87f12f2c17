"""CPU: the text side of the pipeline (model.py:536-567): prompt-embedding cache semantics and the real-size CLIP-shaped stand-in."""
import torch

from freefine_amd.pipeline import FreeFinePipeline
from freefine_amd.scheduler import DDIMScheduler
from freefine_amd.text import ByteTokenizer, SyntheticTextEncoder, clip_shaped_text_encoder


def _pipe(enc):
    return FreeFinePipeline(None, None, ByteTokenizer(), enc, DDIMScheduler(), "cpu")


def test_text_cache_returns_the_encoder_rows_in_request_order():
    enc = SyntheticTextEncoder(64)
    p = _pipe(enc)
    want = enc(ByteTokenizer()(["a cup", "", "a dog", ""]).input_ids)[0]
    got = p._encode_text(["a cup", "", "a dog", ""])
    assert torch.equal(got, want) and p.text_encoder_calls == 1
    again = p._encode_text(["", "a dog"])                       # served from the cache: no encoder call
    assert torch.equal(again, want[[1, 2]]) and p.text_encoder_calls == 1
    p.text_cache_max = 2
    p._encode_text(["x", "y", "z"])
    assert len(p._text_cache) == 2 and "z" in p._text_cache and "a cup" not in p._text_cache
    p.text_cache = False
    assert torch.equal(p._encode_text("a cup"), want[:1]) and p.text_encoder_calls == 3
    assert torch.equal(p.get_text_embeddings(["a cup"]), want[:1]) and p.text_encoder_calls == 4


def test_text_cache_eviction_never_drops_a_prompt_of_the_same_call():
    """with a full cache a call that mixes an old prompt (a hit) with a new one must not evict the hit before its row is built"""
    enc = SyntheticTextEncoder(64)
    p = _pipe(enc)
    p.text_cache_max = 2
    tok = ByteTokenizer()
    p._encode_text(["a", "b"])
    got = p._encode_text(["a", "c"])                            # used to raise KeyError('a')
    assert torch.equal(got, enc(tok(["a", "c"]).input_ids)[0])
    assert list(p._text_cache) == ["c", "a"] and p.text_encoder_calls == 2      # "b" evicted; the hit re-inserted last (most recently used)
    got = p._encode_text(["d", "a", "a", "e"])                  # more distinct prompts than the cache holds: rows still complete and in order
    assert torch.equal(got, enc(tok(["d", "a", "a", "e"]).input_ids)[0]) and len(p._text_cache) == 2


def test_clip_shaped_text_encoder_is_a_transformers_clip_text_model_of_the_checkpoints_shape():
    enc = clip_shaped_text_encoder(1024, layers=2)              # 2 of the 23 layers: shape check only (the bench builds all 23)
    assert enc.config.hidden_size == 1024 and enc.config.num_attention_heads == 16 and enc.config.intermediate_size == 4096
    p = _pipe(enc)
    e = p._encode_text(["a photo of a cup", ""])
    assert e.shape == (2, 77, 1024) and e.dtype == torch.float32 and torch.isfinite(e).all()
    assert not torch.equal(e[0], e[1])
    e2 = clip_shaped_text_encoder(1024, layers=2)(ByteTokenizer()(["a photo of a cup"]).input_ids)[0]
    assert torch.equal(e2[0], e[0])                             # seeded: every rank builds the same encoder; one prompt per encoder call
    p2 = _pipe(enc)
    assert torch.equal(p2._encode_text(["", "x", "a photo of a cup"])[2], e[0]) and p2.text_encoder_calls == 3   # bit-identical whatever is beside it
