"""CPU: host logic of the depth network (freefine_amd/depth.py) -- the weight foldings and layout plumbing around the GEMMs, checked with
plain torch matmuls in place of the kernels (HipDepthAnything only packs weights in its constructor: tensor plumbing, runs on the CPU device)."""
import torch
import torch.nn.functional as F

from freefine_amd.depth import HipDepthAnything, depth_config
from oracle import dpt as OD


def _net(name="tiny", img_size=70):
    cfg, ocfg = depth_config(name), OD.dpt_config(name)
    cfg.img_size = ocfg.img_size = img_size
    st = OD.dpt_synthetic_state(ocfg, seed=2)
    return HipDepthAnything(cfg, st, dtype=torch.float32, device="cpu"), ocfg, st


def test_patch_embedding_is_a_gemm_over_im2col_rows():
    net, ocfg, st = _net()
    x = torch.randn(2, 3, 56, 98, generator=torch.Generator().manual_seed(0))
    cols = net._im2col(x, 14)
    w = net.pe[0][:, :588]                                   # packed [C, 592] fp32, columns >= 588 zero
    assert net.kpe == 592 and torch.count_nonzero(net.pe[0][:, 588:]) == 0
    got = (cols @ w.t() + net.pe[1]).view(2, 4 * 7, -1)
    ref = F.conv2d(x, st["pretrained.patch_embed.proj.weight"], st["pretrained.patch_embed.proj.bias"], stride=14).flatten(2).transpose(1, 2)
    assert (got - ref).abs().max() < 1e-4


def test_positional_embedding_rows_match_the_reference_interpolation():
    net, ocfg, st = _net()
    for (H, W) in ((70, 70), (56, 98), (42, 28)):
        cls_row, pos = net._pos_tokens(H, W)
        ref = OD.interpolate_pos_encoding(ocfg, st["pretrained.pos_embed"], (H // 14) * (W // 14), H, W)
        assert (pos - ref[0, 1:]).abs().max() < 1e-6 and pos.shape == ((H // 14) * (W // 14), ocfg.embed_dim)
        assert (cls_row - (st["pretrained.cls_token"][0] + ref[:, 0])).abs().max() < 1e-6


def test_layerscale_is_folded_into_the_projection_rows():
    net, ocfg, st = _net()
    C = ocfg.embed_dim
    y = torch.randn(5, C, generator=torch.Generator().manual_seed(1))
    b = net.blocks[1]
    got = y @ b.proj[0][:, :C].t() + b.proj[1]
    ref = st["pretrained.blocks.1.ls1.gamma"] * F.linear(y, st["pretrained.blocks.1.attn.proj.weight"], st["pretrained.blocks.1.attn.proj.bias"])
    assert (got - ref).abs().max() < 1e-5
    h = torch.randn(5, 4 * C, generator=torch.Generator().manual_seed(2))
    got = h @ b.fc2[0][:, :4 * C].t() + b.fc2[1]
    ref = st["pretrained.blocks.1.ls2.gamma"] * F.linear(h, st["pretrained.blocks.1.mlp.fc2.weight"], st["pretrained.blocks.1.mlp.fc2.bias"])
    assert (got - ref).abs().max() < 1e-5
    # q | k and v split of the fused qkv projection
    qkv = F.linear(y, st["pretrained.blocks.1.attn.qkv.weight"], st["pretrained.blocks.1.attn.qkv.bias"])
    assert (y @ b.qk[0][:, :C].t() + b.qk[1] - qkv[:, :2 * C]).abs().max() < 1e-5
    assert (y @ b.v[0][:, :C].t() + b.v[1] - qkv[:, 2 * C:]).abs().max() < 1e-5


def test_transposed_convolutions_are_gemms_plus_a_pixel_shuffle():
    net, ocfg, st = _net()
    g = torch.Generator().manual_seed(3)
    for dc, name, k in ((net.up0, "depth_head.resize_layers.0", 4), (net.up1, "depth_head.resize_layers.1", 2)):
        w, b, cin, cout, kk = dc
        assert kk == k
        B, H, W = 2, 3, 5
        x = torch.randn(B, cin, H, W, generator=g)
        rows = x.permute(0, 2, 3, 1).reshape(B, H * W, cin)
        y = rows @ w[:, :cin].t() + b                         # [B, H*W, k*k*cout]
        got = net._pixel_shuffle(y.contiguous(), B, H, W, k, cout).view(B, H * k, W * k, cout).permute(0, 3, 1, 2)
        ref = F.conv_transpose2d(x, st[name + ".weight"], st[name + ".bias"], stride=k)
        assert got.shape == ref.shape and (got - ref).abs().max() < 1e-5


def test_depth_scalar_head_is_padded_to_four_output_columns():
    net, ocfg, st = _net()
    assert net.oc3[0].shape[0] == 4 and torch.count_nonzero(net.oc3[0][1:]) == 0 and torch.count_nonzero(net.oc3[1][1:]) == 0
