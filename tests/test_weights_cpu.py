"""CPU: synthetic weights of the bench / N = 50 fixtures: the planted denoiser path (freefine_amd.weights.plant_denoiser_path)."""
import torch

from freefine_amd.config import UNetConfig
from freefine_amd.weights import plant_denoiser_path, synthetic_state, unet_param_shapes


def test_planted_path_touches_three_tensors_and_adds_without_scaling():
    cfg = UNetConfig.preset("tiny")
    st = synthetic_state(unet_param_shapes(cfg), 0)
    pl = plant_denoiser_path(st, cfg, 3.0)
    changed = sorted(k for k in st if not torch.equal(st[k], pl[k]))
    assert changed == ["conv_in.weight", "conv_out.weight", "up_blocks.3.resnets.2.conv_shortcut.weight"]
    C0 = cfg.block_out_channels[0]
    c = torch.arange(C0)
    E = torch.stack([1.0 - 2.0 * ((c >> k) & 1) for k in range(4)], dim=1)               # [C0, 4], +-1 sign patterns
    assert torch.equal(E.t() @ E, C0 * torch.eye(4))                                     # mutually orthogonal columns
    assert torch.allclose((pl["conv_in.weight"] - st["conv_in.weight"])[:, :, 1, 1], E, atol=1e-6)
    d = (pl["conv_in.weight"] - st["conv_in.weight"]).clone()
    d[:, :, 1, 1] = 0
    assert d.abs().max() == 0                                                            # only the centre tap
    sc = (pl[changed[2]] - st[changed[2]])[:, :, 0, 0]
    assert torch.allclose(sc[:, C0:], 3.0 * torch.eye(C0), atol=1e-6) and sc[:, :C0].abs().max() == 0            # gain * I on the skip columns only
    assert torch.allclose((pl["conv_out.weight"] - st["conv_out.weight"])[:, :, 1, 1], E.t() / (C0 / 4), atol=1e-6)
    assert st["conv_in.weight"].data_ptr() != pl["conv_in.weight"].data_ptr()            # the input state is not modified


def test_planted_unet_predicts_the_normalised_input():
    """eps ~= x / rms(x) + (random network): correlation >= 0.9 at any input scale, unit rms -- what makes a 50-step DDIM edit with these
    weights a denoising trajectory (tests/golden_cases.py n50_cases, bench.py)"""
    from oracle import sd_unet
    cfg = sd_unet.unet_config("tiny")
    net = sd_unet.init_unet(cfg, seed=0)
    base = net.state_dict()
    g = torch.Generator().manual_seed(0)
    x, e = torch.randn(2, 4, 16, 16, generator=g), torch.randn(2, 77, cfg.cross_attention_dim, generator=g)
    with torch.no_grad():
        c0 = _corr(net(x, torch.tensor(501), e), x)
        net.load_state_dict(plant_denoiser_path(base, UNetConfig.preset("tiny"), 3.0))
        for s in (1.0, 0.2, 3.0):
            eps = net(x * s, torch.tensor(501), e)
            assert _corr(eps, x) > 0.9 and 0.8 < eps.pow(2).mean().sqrt() < 1.2, s
    assert abs(c0) < 0.3                                                                   # the un-planted random network does not


def _corr(a, b):
    return ((a * b).mean() / (a.pow(2).mean().sqrt() * b.pow(2).mean().sqrt())).item()
