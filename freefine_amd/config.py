"""Model configurations (diffusers config.json field names) for the SD UNet / VAE the hot path runs on."""
from dataclasses import dataclass, field
from typing import Tuple


@dataclass
class UNetConfig:
    name: str = "sd21-base"
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    down_has_attn: Tuple[bool, ...] = (True, True, True, False)
    cross_attention_dim: int = 1024
    heads: Tuple[int, ...] = (5, 10, 20, 20)       # per resolution level (SD-2.x: attention_head_dim list; SD-1.5: 8 everywhere)
    use_linear_projection: bool = True
    upcast_attention: bool = False
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    flip_sin_to_cos: bool = True
    freq_shift: float = 0.0
    sample_size: int = 64

    @staticmethod
    def preset(name):
        if name == "sd21-base":
            return UNetConfig()
        if name == "sd15":
            return UNetConfig(name="sd15", cross_attention_dim=768, heads=(8, 8, 8, 8), use_linear_projection=False)
        if name == "tiny":
            return UNetConfig(name="tiny", block_out_channels=(32, 64, 128, 128), cross_attention_dim=64, heads=(2, 4, 4, 4),
                              norm_num_groups=8, sample_size=16)
        if name == "tiny-conv":
            return UNetConfig(name="tiny-conv", block_out_channels=(32, 64, 128, 128), cross_attention_dim=48, heads=(4, 4, 4, 4),
                              use_linear_projection=False, norm_num_groups=8, sample_size=16)
        raise ValueError(name)

    @staticmethod
    def from_diffusers(cfg: dict):
        """from a diffusers unet/config.json dict."""
        ahd = cfg.get("attention_head_dim", 8)
        n = len(cfg["block_out_channels"])
        heads = tuple(ahd) if isinstance(ahd, (list, tuple)) else (ahd,) * n
        return UNetConfig(name=cfg.get("_name_or_path", "custom"), in_channels=cfg["in_channels"], out_channels=cfg["out_channels"],
                          block_out_channels=tuple(cfg["block_out_channels"]), layers_per_block=cfg["layers_per_block"],
                          down_has_attn=tuple("CrossAttn" in t for t in cfg["down_block_types"]),
                          cross_attention_dim=cfg["cross_attention_dim"], heads=heads,
                          use_linear_projection=cfg.get("use_linear_projection", False), upcast_attention=cfg.get("upcast_attention", False),
                          norm_num_groups=cfg.get("norm_num_groups", 32), norm_eps=cfg.get("norm_eps", 1e-5),
                          flip_sin_to_cos=cfg.get("flip_sin_to_cos", True), freq_shift=float(cfg.get("freq_shift", 0)),
                          sample_size=cfg.get("sample_size", 64))


@dataclass
class VAEConfig:
    name: str = "sd"
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215

    @staticmethod
    def preset(name):
        if name == "sd":
            return VAEConfig()
        if name == "tiny":
            return VAEConfig(name="tiny", block_out_channels=(16, 32, 32, 32), layers_per_block=1, norm_num_groups=8)
        raise ValueError(name)
