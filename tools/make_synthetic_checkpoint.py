"""Write a SYNTHETIC Stable-Diffusion checkpoint folder in the Hugging Face layout the reference loads with
`FreeFinePipeline.from_pretrained(path)` (/root/reference/evaluation/FreeFine/freefine_batch_infer_2d.py:148-157):

    <out>/unet/config.json + diffusion_pytorch_model.safetensors        diffusers UNet2DConditionModel field / parameter names
    <out>/vae/config.json + diffusion_pytorch_model.safetensors         AutoencoderKL; mid-block attention under the LEGACY names the hub checkpoints
                                                                         still carry (query / key / value / proj_attn, 1x1-conv weights stored 4-D)
    <out>/scheduler/scheduler_config.json                               SD's PNDM config (what `DDIMScheduler.from_config` is fed, model.py:123-127)
    <out>/tokenizer/, <out>/text_encoder/                               a byte-level CLIP tokenizer (no merges) + a small seeded CLIPTextModel, both
                                                                         written with transformers' save_pretrained

No network is needed and no real weights exist offline: the tensors are the seeded default-init state `synthetic:<preset>` generates
(freefine_amd.weights.synthetic_state), so a pipeline built from the folder must reproduce the `from_state` pipeline of the same seed bit for bit
(fp32 file) or its fp16-rounded weights (`--dtype fp16`, the format of the hub's fp16 shards).  tests/test_checkpoint_cpu.py and
tests/test_pipeline_gpu.py::test_from_pretrained_folder_* drive it.

    python tools/make_synthetic_checkpoint.py --out /tmp/sd_tiny --unet tiny --vae tiny [--dtype fp16] [--seed 0] [--prediction-type epsilon]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SD_SCHEDULER = {   # stabilityai/stable-diffusion-2-1-base scheduler/scheduler_config.json (PNDM), the constants SURVEY section 8a lists
    "_class_name": "PNDMScheduler", "_diffusers_version": "0.10.0.dev0", "beta_end": 0.012, "beta_schedule": "scaled_linear", "beta_start": 0.00085,
    "clip_sample": False, "num_train_timesteps": 1000, "prediction_type": "epsilon", "set_alpha_to_one": False, "skip_prk_steps": True,
    "steps_offset": 1, "trained_betas": None,
}
_LEGACY = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}


def byte_chars():
    """GPT-2 / CLIP byte -> printable character table (every byte gets a character the BPE vocabulary can hold)"""
    keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    out, n = {}, 0
    for b in range(256):
        if b in keep:
            out[b] = chr(b)
        else:
            out[b] = chr(256 + n)
            n += 1
    return out


def unet_config_json(c):
    n = len(c.block_out_channels)
    return {
        "_class_name": "UNet2DConditionModel", "_diffusers_version": "0.18.0", "act_fn": "silu", "attention_head_dim": list(c.heads),
        "block_out_channels": list(c.block_out_channels), "center_input_sample": False, "cross_attention_dim": c.cross_attention_dim,
        "down_block_types": ["CrossAttnDownBlock2D" if a else "DownBlock2D" for a in c.down_has_attn], "downsample_padding": 1,
        "dual_cross_attention": False, "flip_sin_to_cos": c.flip_sin_to_cos, "freq_shift": c.freq_shift, "in_channels": c.in_channels,
        "layers_per_block": c.layers_per_block, "mid_block_scale_factor": 1, "norm_eps": c.norm_eps, "norm_num_groups": c.norm_num_groups,
        "out_channels": c.out_channels, "sample_size": c.sample_size,
        "up_block_types": ["CrossAttnUpBlock2D" if a else "UpBlock2D" for a in reversed(c.down_has_attn)][:n],
        "use_linear_projection": c.use_linear_projection, "upcast_attention": c.upcast_attention,
    }


def vae_config_json(c):
    n = len(c.block_out_channels)
    return {
        "_class_name": "AutoencoderKL", "_diffusers_version": "0.18.0", "act_fn": "silu", "block_out_channels": list(c.block_out_channels),
        "down_block_types": ["DownEncoderBlock2D"] * n, "up_block_types": ["UpDecoderBlock2D"] * n, "in_channels": c.in_channels,
        "out_channels": c.out_channels, "latent_channels": c.latent_channels, "layers_per_block": c.layers_per_block,
        "norm_num_groups": c.norm_num_groups, "sample_size": 512, "scaling_factor": c.scaling_factor,
    }


def legacy_vae_names(state):
    """current diffusers names -> the pre-0.14 ones of the hub's SD VAE checkpoints (mid-block attention as 1x1 convolutions)"""
    out = {}
    for k, v in state.items():
        parts = k.split(".")
        if ".attentions." in k:
            for new, old in _LEGACY.items():
                nparts = new.split(".")
                if parts[-1 - len(nparts):-1] == nparts:
                    parts = parts[:-1 - len(nparts)] + [old] + parts[-1:]
                    if v.ndim == 2:
                        v = v.reshape(v.shape[0], v.shape[1], 1, 1)
                    break
        out[".".join(parts)] = v.contiguous()
    return out


def write(out, unet="tiny", vae="tiny", dtype="fp32", seed=0, prediction_type="epsilon", scheduler=True, text_width=None, text_layers=2):
    from safetensors.torch import save_file
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTokenizer
    from freefine_amd.config import UNetConfig, VAEConfig
    from freefine_amd.weights import synthetic_state, unet_param_shapes, vae_param_shapes
    ucfg, vcfg = UNetConfig.preset(unet), VAEConfig.preset(vae)
    tdt = torch.float16 if dtype == "fp16" else torch.float32
    ust = synthetic_state(unet_param_shapes(ucfg), seed)
    vst = synthetic_state(vae_param_shapes(vcfg), seed + 1)
    for sub, cfg, st in (("unet", unet_config_json(ucfg), ust), ("vae", vae_config_json(vcfg), legacy_vae_names(vst))):
        os.makedirs(os.path.join(out, sub), exist_ok=True)
        with open(os.path.join(out, sub, "config.json"), "w") as f:
            json.dump(cfg, f, indent=2)
        save_file({k: v.to(tdt).contiguous() for k, v in st.items()}, os.path.join(out, sub, "diffusion_pytorch_model.safetensors"))
    if scheduler:
        os.makedirs(os.path.join(out, "scheduler"), exist_ok=True)
        with open(os.path.join(out, "scheduler", "scheduler_config.json"), "w") as f:
            json.dump(dict(SD_SCHEDULER, prediction_type=prediction_type), f, indent=2)
    chars = [byte_chars()[i] for i in range(256)]
    vocab = {}
    for c in chars:
        vocab[c] = len(vocab)
    for c in chars:
        vocab[c + "</w>"] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    tok = CLIPTokenizer(vocab=vocab, merges=[], model_max_length=77)
    tok.save_pretrained(os.path.join(out, "tokenizer"))
    width = text_width or ucfg.cross_attention_dim
    tcfg = CLIPTextConfig(vocab_size=len(vocab), hidden_size=width, intermediate_size=2 * width, num_hidden_layers=text_layers,
                          num_attention_heads=max(1, width // 32), max_position_embeddings=77, bos_token_id=vocab["<|startoftext|>"],
                          eos_token_id=vocab["<|endoftext|>"], pad_token_id=vocab["<|endoftext|>"])
    torch.manual_seed(seed + 2)
    CLIPTextModel(tcfg).eval().save_pretrained(os.path.join(out, "text_encoder"))
    return ucfg, vcfg, ust, vst


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--unet", default="tiny")
    ap.add_argument("--vae", default="tiny")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "fp16"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--prediction-type", dest="prediction_type", default="epsilon")
    ap.add_argument("--no-scheduler", dest="scheduler", action="store_false")
    a = ap.parse_args()
    write(a.out, a.unet, a.vae, a.dtype, a.seed, a.prediction_type, a.scheduler)
    print("wrote", a.out, sorted(os.listdir(a.out)))
