"""Build-container only: import the read-only reference (/root/reference) so that golden vectors can be generated from
its OWN code (tools/gen_golden.py).  Nothing here travels to the GPU box as an executable dependency: tests read only
the .npz fixtures this produces.

`src/utils/attention.py` imports as-is.  `src/demo/model.py` needs four packages that are absent here (diffusers, cv2,
rembg, pytorch_lightning); they are stood in for by the minimal stubs below -- stubs of the
reference's DEPENDENCIES, not of the reference.  The UNet/VAE it drives are the oracle's diffusers-layout modules
(oracle/sd_unet.py, oracle/sd_vae.py), i.e. the reference's hook registrars and loops run unmodified over them.
"""
import os
import random
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def install_stubs():
    import matplotlib
    matplotlib.use("Agg")
    from scipy import ndimage

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class StableDiffusionPipeline:  # empty base, the harness sets the attributes by hand
        pass

    class DDIMScheduler:
        pass

    d = mod("diffusers", StableDiffusionPipeline=StableDiffusionPipeline, DDIMScheduler=DDIMScheduler)
    du = mod("diffusers.utils")
    dut = mod("diffusers.utils.torch_utils",
              randn_tensor=lambda shape, generator=None, device=None, dtype=None: torch.randn(shape, generator=generator, dtype=dtype).to(device))
    d.utils, du.torch_utils = du, dut

    def dilate(src, kernel, iterations=1):
        return ndimage.maximum_filter(src, size=kernel.shape, mode="constant", cval=0)

    def erode(src, kernel, iterations=1):
        return ndimage.minimum_filter(src, size=kernel.shape, mode="constant", cval=0)

    def cvtColor(img, code):
        return np.ascontiguousarray(img[..., ::-1])

    mod("cv2", dilate=dilate, erode=erode, cvtColor=cvtColor, COLOR_RGB2BGR=4, COLOR_BGR2RGB=4, COLOR_RGB2GRAY=7)
    mod("rembg", remove=lambda *a, **k: None)

    def seed_everything(seed):
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        return seed

    pl = mod("pytorch_lightning", seed_everything=seed_everything)
    plu = mod("pytorch_lightning.utilities", rank_zero_warn=lambda *a, **k: None)
    pl.utilities = plu


def import_reference():
    """returns (attention_module, model_module) of the reference."""
    install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import importlib
    # the repo root also has a `src` package (drop-in shims); make sure the REFERENCE's `src` wins here
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    sys.path = [REF] + [p for p in sys.path if os.path.abspath(p or ".") != ROOT and p != REF]
    A = importlib.import_module("src.utils.attention")
    Mo = importlib.import_module("src.demo.model")
    assert A.__file__.startswith(REF) and Mo.__file__.startswith(REF)
    sys.path.append(ROOT)
    return A, Mo


class VaeAdapter:
    """the two diffusers AutoencoderKL calls the reference makes (model.py:267, 272)."""

    def __init__(self, vae):
        self.vae = vae
        self.dtype = torch.float32

    def parameters(self):
        return self.vae.parameters()

    def encode(self, x):
        mean = self.vae.encode_mean(x)
        return {"latent_dist": types.SimpleNamespace(mean=mean)}

    def decode(self, z):
        return {"sample": self.vae.decode(z)}


def build_reference_pipeline(A, Mo, unet, vae, tokenizer, text_encoder, sched, hook="edit", start_layer=10):
    p = Mo.FreeFinePipeline()
    p.device = torch.device("cpu")
    p.unet, p.vae, p.tokenizer, p.text_encoder, p.scheduler = unet, VaeAdapter(vae), tokenizer, text_encoder, sched
    controller = A.Attention_Modulator(start_layer=start_layer)
    p.controller = controller
    {"edit": A.register_attention_control, "bggen": A.register_attention_control_4bggen,
     "compose": A.register_attention_control_compose}[hook](p, controller)
    p.modify_unet_forward()
    return p
