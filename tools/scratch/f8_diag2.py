import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from freefine_amd import ops, _lib as L
lib = L.load()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
def q(x, Cp):
    B, HW, C = x.shape
    t = torch.zeros(B, HW, Cp, device=dev); t[..., :C] = (x * 16).clamp(-448, 448)
    t8 = t.to(torch.float8_e4m3fn); o = t8.view(torch.uint8).contiguous(); o._ffn_f8_act = C
    return o, t8.float()[..., :C].double() / 16
for (B, H, Cin, Cout, cfg) in [(3, 32, 128, 320, 13), (3, 32, 256, 256, 14)]:
    x = (torch.randn(B, H * H, Cin, generator=g).abs() * 0.7).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5).to(dev)
    wp = ops.pack_conv3x3_f8(w); Cp, alpha = wp._ffn_f8
    wdq = wp.view(torch.float8_e4m3fn).float().reshape(Cout, 3, 3, Cp)[..., :Cin].permute(0, 3, 1, 2).double() * (alpha * 16)
    x8, xdq = q(x, Cp)
    ref = F.conv2d(xdq.reshape(B, H, H, Cin).permute(0, 3, 1, 2), wdq, None, padding=1).permute(0, 2, 3, 1).reshape(B * H * H, Cout)
    lib.ffn_igemm_force_config(cfg)
    for rep in range(3):
        out = ops.conv3x3(x8, wp, torch.zeros(Cout, device=dev), B, H, H, Cin).reshape(B * H * H, Cout)
        bad = torch.isnan(out) | ((out.double() - ref).abs() > 0.05 * ref.abs().max())
        idx = bad.nonzero()
        print(f"cfg {cfg} rep {rep}: bad {idx.shape[0]}; rows%256 {sorted(set((idx[:,0]%256).tolist()))[:40]} cols {sorted(set(idx[:,1].tolist()))[:40]} tiles {sorted(set((idx[:,0]//256).tolist()))}")
        vals = out[bad][:10]
        print("   values", vals.tolist(), "ref", ref[bad][:10].tolist())
    # split-K forced
    for sk in (3, 9):
        out = ops.conv3x3(x8, wp, torch.zeros(Cout, device=dev), B, H, H, Cin, splitk=sk).reshape(B * H * H, Cout)
        bad = torch.isnan(out) | ((out.double() - ref).abs() > 0.05 * ref.abs().max())
        print(f"cfg {cfg} splitk {sk}: bad {int(bad.sum())}")
lib.ffn_igemm_force_config(-1)
