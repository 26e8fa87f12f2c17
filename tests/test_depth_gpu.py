"""GPU: the depth network of the 3D front end (freefine_amd/depth.py) against its oracle (oracle/dpt.py, pinned to the
reference's DINOv2 + DPT head by tests/golden/g8_dpt.npz), through the C ABI."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def rng_tensor(seed, shape, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def relerr(a, b):
    return ((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def test_new_epilogues_and_helpers(gpu):
    """FFN_IG_OUT_GELU / FFN_IG_OUT_RELU epilogues (Linear and 3x3 conv, with residual, through split-K too), ffn_eltwise, ffn_resize_bilinear
    (align_corners = True; up, down, identity, 1-pixel sources) against torch in fp64"""
    from freefine_amd import ops
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    for dt, tol in ((torch.float32, 2e-5), (torch.bfloat16, 2e-2)):
        x = torch.randn(3, 200, 192, generator=g).to(dt).to(gpu)
        w, b = (torch.randn(256, 192, generator=g) * 192 ** -0.5).to(gpu), torch.randn(256, generator=g).to(gpu)
        r = torch.randn(3, 200, 256, generator=g).to(dt).to(gpu)
        wp = ops.pack_linear(w, dt)
        wd = wp.double()[:, :192]
        ref = x.double() @ wd.t() + b.double()
        for sk in (0, 3):
            assert relerr(ops.linear(x, wp, b, K=192, gelu=True, splitk=sk), F.gelu(ref)) < tol
            assert relerr(ops.linear(x, wp, b, K=192, relu=True, residual=r, splitk=sk), F.relu(ref) + r.double()) < tol
        xc = torch.randn(2, 12 * 20, 32, generator=g).to(dt).to(gpu)
        wc = (torch.randn(48, 32, 3, 3, generator=g) * (9 * 32) ** -0.5).to(gpu)
        bc = torch.randn(48, generator=g).to(gpu)
        wq = wc.to(dt).double()
        refc = F.relu(F.conv2d(xc.double().view(2, 12, 20, 32).permute(0, 3, 1, 2), wq, bc.double(), padding=1)).permute(0, 2, 3, 1).reshape(2, 240, 48)
        assert relerr(ops.conv3x3(xc, ops.pack_conv3x3(wc, dt), bc, 2, 12, 20, 32, relu=True), refc) < tol
        assert torch.equal(ops.relu(x), torch.relu(x))
        assert relerr(ops.add(r, r.flip(0).contiguous()), r.double() + r.flip(0).double()) < (1e-6 if dt == torch.float32 else 1e-2)
        for (hi, wi, ho, wo) in ((5, 7, 20, 28), (37, 37, 74, 74), (40, 30, 13, 9), (6, 6, 6, 6), (1, 3, 4, 5), (8, 8, 1, 1)):
            t = torch.randn(2, hi * wi, 16, generator=g).to(dt).to(gpu)
            ref = F.interpolate(t.double().view(2, hi, wi, 16).permute(0, 3, 1, 2), size=(ho, wo), mode="bilinear", align_corners=True)
            ref = ref.permute(0, 2, 3, 1).reshape(2, ho * wo, 16)
            assert relerr(ops.resize_bilinear(t, 2, hi, wi, ho, wo), ref) < (1e-6 if dt == torch.float32 else 1e-2), (hi, wi, ho, wo)
            assert relerr(ops.resize_bilinear(t, 2, hi, wi, ho, wo, relu=True), F.relu(ref)) < (1e-6 if dt == torch.float32 else 1e-2)


def test_activation_epilogues_in_every_igemm_configuration(gpu):
    """GELU / ReLU through EVERY bf16 tile configuration forced in turn (the ping-pong tiles apply them in their plain epilogue when there
    is no residual, everything else in the generic kernel / the split-K reduce), ragged M, shapes of the DINOv2 MLP and of the DPT head"""
    from freefine_amd import _lib as L
    from freefine_amd import ops
    import torch.nn.functional as F
    lib = L.load()
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(9)
    try:
        for cfg in range(lib.ffn_igemm_num_configs()):
            lib.ffn_igemm_force_config(cfg)
            for (M, N, K) in ((2740 + 8, 1024, 256), (5480, 640, 320)):
                x = torch.randn(M, K, generator=g).to(dt).to(gpu)
                wp = ops.pack_linear((torch.randn(N, K, generator=g) * K ** -0.5).to(gpu), dt)
                b = torch.randn(N, generator=g).to(gpu)
                r = torch.randn(M, N, generator=g).to(dt).to(gpu)
                ref = x.double() @ wp.double()[:, :K].t() + b.double()
                big = torch.full((M + 300, N), 7.0, dtype=dt, device=gpu)
                assert relerr(ops.linear(x, wp, b, K=K, gelu=True, out=big[:M]), F.gelu(ref)) < 2e-2, (cfg, M, N, K, "gelu")
                assert (big[M:] == 7.0).all(), (cfg, "rows past M written")
                assert relerr(ops.linear(x, wp, b, K=K, relu=True), F.relu(ref)) < 2e-2, (cfg, M, N, K, "relu")
                assert relerr(ops.linear(x, wp, b, K=K, relu=True, residual=r), F.relu(ref) + r.double()) < 2e-2, (cfg, M, N, K, "relu+res")
            B, H, W, Cin, Cout = 2, 32, 40, 64, 256
            xc = torch.randn(B, H * W, Cin, generator=g).to(dt).to(gpu)
            wc = (torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5).to(gpu)
            bc = torch.randn(Cout, generator=g).to(gpu)
            refc = F.relu(F.conv2d(xc.double().view(B, H, W, Cin).permute(0, 3, 1, 2), wc.to(dt).double(), bc.double(), padding=1))
            refc = refc.permute(0, 2, 3, 1).reshape(B, H * W, Cout)
            for sk in (0, 3):
                assert relerr(ops.conv3x3(xc, ops.pack_conv3x3(wc, dt), bc, B, H, W, Cin, relu=True, splitk=sk), refc) < 2e-2, (cfg, "conv relu", sk)
    finally:
        lib.ffn_igemm_force_config(-1)


@pytest.mark.parametrize("name,img_size,sizes", [("tiny", 70, ((70, 70), (56, 98))), ("mini", 518, ((42, 70),))])
def test_depth_network_vs_oracle_and_reference_golden(gpu, name, img_size, sizes):
    """HipDepthAnything vs oracle/dpt.py AND vs the reference's own outputs (G8): ViT features of the last block and the depth map;
    fp32 parity mode at 1e-4 of the output scale, bf16 fast mode bounded"""
    from oracle import dpt as OD
    from freefine_amd.depth import HipDepthAnything, depth_config
    gold = np.load(os.path.join(GOLD, "g8_dpt.npz"))
    ocfg = OD.dpt_config(name)
    ocfg.img_size = img_size
    st = OD.dpt_synthetic_state(ocfg, seed=3 + len(name))
    cfg = depth_config(name)
    cfg.img_size = img_size
    for dt, tol in ((torch.float32, 1e-4), (torch.bfloat16, 6e-2)):
        net = HipDepthAnything(cfg, st, dtype=dt, device=gpu)
        for (H, W) in sizes:
            x = rng_tensor(80 + H + W, (2, 3, H, W))
            key = f"{name}_{H}x{W}"
            feats, ph, pw = net.features(x)
            d = net(x)
            ef = relerr(feats[3], torch.from_numpy(gold[key + "_feat3"]))
            ed = relerr(d, torch.from_numpy(gold[key + "_depth"]))
            eo = relerr(d, OD.depth_forward(ocfg, st, x))
            print(f"depth net {key} {dt}: ViT features vs reference {ef:.2e}, depth vs reference {ed:.2e}, vs oracle {eo:.2e}")
            assert d.shape == (2, H, W) and d.dtype == torch.float32 and (d >= 0).all()
            assert ef < tol and ed < tol and eo < tol, key


def test_depth_anything_vitl_518_vs_oracle(gpu):
    """the production configuration (dinov2_vitl14 + the vitl DPT head, 518 x 518: 1370 tokens, 24 blocks) on seeded weights against the
    oracle in fp32 (the oracle costs ~15 s on the host), and the drop-in class of depth_anything/dpt.py"""
    from oracle import dpt as OD
    from depth_anything.dpt import DepthAnything
    torch.set_num_threads(max(8, min(32, torch.get_num_threads())))
    cfg = OD.dpt_config("vitl")
    st = OD.dpt_synthetic_state(cfg, seed=1)
    x = rng_tensor(7, (1, 3, 518, 518))
    ref = OD.depth_forward(cfg, st, x)
    for dt, tol in ((torch.float32, 2e-4), (torch.bfloat16, 8e-2)):
        model = DepthAnything(dict(encoder="vitl", features=256, out_channels=[256, 512, 1024, 1024]), torch_dtype=dt, device=gpu)
        model.load_state_dict(st)
        d = model(x)
        e = relerr(d, ref)
        print(f"DepthAnything vitl 518 x 518 {dt}: depth vs oracle {e:.2e} (|depth| max {ref.abs().max():.3f})")
        assert d.shape == (1, 518, 518) and e < tol
        del model
        torch.cuda.empty_cache()
