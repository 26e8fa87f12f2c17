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
for (B, H, Cin, Cout) in [(3, 32, 128, 320), (4, 64, 320, 320), (3, 32, 256, 256), (8, 32, 128, 640)]:
    x = (torch.randn(B, H * H, Cin, generator=g).abs() * 0.7).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5).to(dev)
    b = torch.randn(Cout, generator=g).to(dev); rb = torch.randn(B, Cout, generator=g).to(dev)
    r = torch.randn(B, H * H, Cout, generator=g).to(torch.bfloat16).to(dev)
    wp = ops.pack_conv3x3_f8(w); Cp, alpha = wp._ffn_f8
    wdq = wp.view(torch.float8_e4m3fn).float().reshape(Cout, 3, 3, Cp)[..., :Cin].permute(0, 3, 1, 2).double() * (alpha * 16)
    x8, xdq = q(x, Cp)
    ref = F.conv2d(xdq.reshape(B, H, H, Cin).permute(0, 3, 1, 2), wdq, None, padding=1).permute(0, 2, 3, 1).reshape(B, H * H, Cout)
    for cfg in (13, 14, 15, 16, 2):
        lib.ffn_igemm_force_config(cfg)
        for name, kw, add in (("plain", {}, 0), ("bias", dict(bias=True), b.double()), ("rb", dict(rowbias=rb), rb.double()[:, None]), ("res", dict(residual=r), r.double())):
            bias = b if kw.pop("bias", False) else torch.zeros(Cout, device=dev)
            out = ops.conv3x3(x8, wp, bias, B, H, H, Cin, **kw)
            e = ((out.double() - (ref + add)).abs().max() / (ref + add).abs().max()).item()
            nan = int(torch.isnan(out).sum())
            print(f"B={B} H={H} Cin={Cin} Cout={Cout} cfg={cfg} {name}: err {e:.3e} nan {nan} alpha {alpha:.3e}")
lib.ffn_igemm_force_config(-1)
