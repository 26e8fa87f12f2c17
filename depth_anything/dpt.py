"""Drop-in import path of the reference's depth network (/root/reference/depth_anything/dpt.py: DPT_DINOv2, DepthAnything) on the
MI355X engine -- `from depth_anything.dpt import DepthAnything` resolves here when this repository is first on sys.path.

    model = DepthAnything(dict(encoder="vitl", features=256, out_channels=[256, 512, 1024, 1024]))      # dpt.py:170-172
    model.load_state_dict(torch.load("depth_anything_vitl14.pth"))                                      # the reference's checkpoint layout
    depth = model(image)            # [B, 3, H, W], sides multiples of 14 -> [B, H, W] (DPT_DINOv2.forward, dpt.py:155-167)

The reference's `from_pretrained` downloads from the Hugging Face hub (PyTorchModelHubMixin); there is no network here, so weights
arrive as a state dict.  use_bn / use_clstoken are the released checkpoints' (False, False)."""
import torch

from freefine_amd.depth import HipDepthAnything, depth_config


class DPT_DINOv2:
    def __init__(self, encoder="vitl", features=256, out_channels=(256, 512, 1024, 1024), use_bn=False, use_clstoken=False, localhub=True,
                 torch_dtype=torch.float32, device="cuda:0"):
        assert encoder in ("vits", "vitb", "vitl") and not use_bn and not use_clstoken
        self.cfg = depth_config(encoder)
        self.cfg.features, self.cfg.out_channels = int(features), tuple(out_channels)
        self.dtype, self.device, self.engine = torch_dtype, device, None

    def load_state_dict(self, state, strict=True):
        self.engine = HipDepthAnything(self.cfg, state, dtype=self.dtype, device=self.device)
        return self

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def forward(self, x):
        if self.engine is None:
            raise RuntimeError("load_state_dict first: the engine packs its weights once")
        return self.engine(x)

    __call__ = forward


class DepthAnything(DPT_DINOv2):
    def __init__(self, config, **kw):
        super().__init__(**config, **kw)
