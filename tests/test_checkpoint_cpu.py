"""The checkpoint-folder branch of FreeFinePipeline.from_pretrained (the first line of every reference call site,
/root/reference/evaluation/FreeFine/freefine_batch_infer_2d.py:148-157) on a SYNTHETIC Hugging Face layout folder written offline by
tools/make_synthetic_checkpoint.py: configs parsed from diffusers' config.json fields (SD-2.1's attention_head_dim list, use_linear_projection,
upcast_attention), fp32 and fp16 safetensors, the hub's legacy VAE attention names, the scheduler constants READ and asserted, a real
transformers CLIP tokenizer + text encoder saved with save_pretrained.  No GPU: FreeFinePipeline.components stops short of the HIP executors."""
import dataclasses
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def folders(tmp_path_factory):
    import make_synthetic_checkpoint as M
    out = {}
    for dtype in ("fp32", "fp16"):
        d = str(tmp_path_factory.mktemp(f"sd_tiny_{dtype}"))
        out[dtype] = (d,) + M.write(d, "tiny", "tiny", dtype, seed=3)
    return out


def test_folder_layout_is_the_hub_layout(folders):
    d = folders["fp32"][0]
    for sub, f in (("unet", "config.json"), ("unet", "diffusion_pytorch_model.safetensors"), ("vae", "config.json"),
                   ("vae", "diffusion_pytorch_model.safetensors"), ("scheduler", "scheduler_config.json"), ("tokenizer", "tokenizer_config.json"),
                   ("text_encoder", "config.json")):
        assert os.path.exists(os.path.join(d, sub, f)), (sub, f)
    ucfg = json.load(open(os.path.join(d, "unet", "config.json")))
    assert ucfg["attention_head_dim"] == [2, 4, 4, 4] and ucfg["use_linear_projection"] is True and "upcast_attention" in ucfg
    from safetensors.torch import load_file
    vae = load_file(os.path.join(d, "vae", "diffusion_pytorch_model.safetensors"))
    legacy = [k for k in vae if ".attentions.0.query." in k or ".attentions.0.proj_attn." in k]
    assert len(legacy) == 8 and all(".to_q." not in k for k in vae)                  # encoder + decoder mid block, weight + bias, two of the four names
    assert vae["encoder.mid_block.attentions.0.query.weight"].ndim == 4              # 1x1-conv storage of the old checkpoints


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_components_from_folder_equal_the_generating_state(folders, dtype):
    from freefine_amd.pipeline import FreeFinePipeline
    from freefine_amd.config import UNetConfig, VAEConfig
    from freefine_amd.weights import normalize_state_dict, unet_param_shapes, vae_param_shapes, validate_state_dict
    d, ucfg0, vcfg0, ust0, vst0 = folders[dtype]
    ucfg, ust, vcfg, vst, tok, enc, sched, sdt = FreeFinePipeline.components(d, torch.float32, "cpu")
    a, b = dataclasses.asdict(ucfg), dataclasses.asdict(ucfg0)
    a.pop("name"), b.pop("name")
    assert a == b, (a, b)                                                            # incl. heads from attention_head_dim, linear projections, norm groups
    assert ucfg.heads == (2, 4, 4, 4) and ucfg.use_linear_projection and ucfg.down_has_attn == (True, True, True, False)
    va, vb = dataclasses.asdict(vcfg), dataclasses.asdict(vcfg0)
    va.pop("name"), vb.pop("name")
    assert va == vb
    ust, vst = normalize_state_dict(ust), normalize_state_dict(vst)
    validate_state_dict(ust, unet_param_shapes(ucfg), "unet")
    validate_state_dict(vst, vae_param_shapes(vcfg), "vae")
    for got, ref in ((ust, ust0), (vst, vst0)):
        assert set(got) == set(ref)
        for k in ref:
            want = ref[k] if dtype == "fp32" else ref[k].to(torch.float16).float()   # fp16 shards are up-cast, nothing else
            assert got[k].dtype == torch.float32 and got[k].shape == ref[k].shape and torch.equal(got[k], want), k
    # the scheduler constants come from the file and are SD's: timesteps 981, 961, ..., 1 and alpha_bar_0 as the final alpha
    c = sched.config
    assert (c.num_train_timesteps, c.beta_start, c.beta_end, c.beta_schedule, c.steps_offset, c.set_alpha_to_one, c.prediction_type) == \
        (1000, 0.00085, 0.012, "scaled_linear", 1, False, "epsilon")
    sched.set_timesteps(50)
    assert sched.timesteps[:3].tolist() == [981, 961, 941] and int(sched.timesteps[-1]) == 1
    assert torch.equal(sched.final_alpha_cumprod, sched.alphas_cumprod[0])
    # tokenizer + text encoder through transformers: 77 tokens, BOS first, embeddings of the UNet's cross-attention width
    ids = tok(["a photo of a cup", ""], padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
    assert ids.shape == (2, 77) and ids[0, 0] == tok.bos_token_id and ids[1, 1] == tok.eos_token_id
    with torch.no_grad():
        emb = enc(ids)[0]
    assert emb.shape == (2, 77, ucfg.cross_attention_dim) and torch.isfinite(emb).all()


def test_scheduler_config_is_asserted_not_defaulted(folders, tmp_path):
    import shutil
    from freefine_amd.pipeline import FreeFinePipeline
    src = folders["fp32"][0]

    def variant(name, edit):
        d = str(tmp_path / name)
        shutil.copytree(src, d)
        sp = os.path.join(d, "scheduler", "scheduler_config.json")
        edit(sp)
        return d

    def rewrite(**kw):
        def f(sp):
            cfg = json.load(open(sp))
            for k, v in kw.items():
                if v is KeyError:
                    cfg.pop(k)
                else:
                    cfg[k] = v
            json.dump(cfg, open(sp, "w"))
        return f

    with pytest.raises(FileNotFoundError, match="scheduler_config.json"):
        FreeFinePipeline.components(variant("nofile", os.remove), torch.float32, "cpu")
    with pytest.raises(ValueError, match="epsilon"):                                 # the 768-v checkpoint of SD-2.1 must not load silently
        FreeFinePipeline.components(variant("vpred", rewrite(prediction_type="v_prediction")), torch.float32, "cpu")
    with pytest.raises(ValueError, match="steps_offset"):
        FreeFinePipeline.components(variant("nokey", rewrite(steps_offset=KeyError)), torch.float32, "cpu")
    with pytest.raises(ValueError, match="trained_betas"):
        FreeFinePipeline.components(variant("betas", rewrite(trained_betas=[0.1, 0.2])), torch.float32, "cpu")
    with pytest.raises(ValueError, match="timestep_spacing"):
        FreeFinePipeline.components(variant("spacing", rewrite(timestep_spacing="trailing")), torch.float32, "cpu")
    # an SD-1.5-style file without prediction_type is epsilon prediction (diffusers' default) and loads
    out = FreeFinePipeline.components(variant("noptype", rewrite(prediction_type=KeyError)), torch.float32, "cpu")
    assert out[6].config.prediction_type == "epsilon"


def test_truncated_checkpoint_fails_with_a_readable_message(folders, tmp_path):
    import shutil
    from safetensors.torch import load_file, save_file
    from freefine_amd.pipeline import FreeFinePipeline
    from freefine_amd.weights import normalize_state_dict, unet_param_shapes, validate_state_dict
    d = str(tmp_path / "cut")
    shutil.copytree(folders["fp32"][0], d)
    fp = os.path.join(d, "unet", "diffusion_pytorch_model.safetensors")
    st = load_file(fp)
    st.pop("conv_in.weight")
    save_file(st, fp)
    ucfg, ust, *_ = FreeFinePipeline.components(d, torch.float32, "cpu")
    with pytest.raises(ValueError, match="conv_in.weight"):
        validate_state_dict(normalize_state_dict(ust), unet_param_shapes(ucfg), "unet")
