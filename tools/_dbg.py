import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from advchain_amd import ops
from oracle import advchain_oracle as O
torch.manual_seed(0)
dev = "cuda"
dims = (40, 36, 48); N = 2; d = 3
for amp in (1.6, 3.6):
    low = torch.rand(N, d, 5, 5, 6) * 2 - 1
    up = F.interpolate(low, size=dims, mode="trilinear", align_corners=True); up = up / up.abs().max()
    sc = torch.tensor([2.0 * amp / (dims[d - 1 - a] - 1) for a in range(d)]).view(1, d, 1, 1, 1)
    phi = (O.identity_grid(N, dims) + up * sc).contiguous()
    for name, w in (("uniform", torch.rand(N, d, *dims)), ("heavy", torch.randn(N, d, *dims) ** 5)):
        p = phi.double().clone().requires_grad_(True)
        perm = (0, 2, 3, 4, 1)
        out = F.grid_sample(p, p.permute(*perm), padding_mode="border", align_corners=True)
        (out * w.double()).sum().backward()
        ref = p.grad
        ws = ops._scatter_workspace(N, dims, dev)
        H = int(amp + 1)
        for nm, halo in (("march", -H), ("window", 8)):
            g = ops.raw_compose_self_bwd(w.to(dev), phi.to(dev), ws, False, halo).cpu().double()
            e = (g - ref).abs()
            print("amp %.1f gout %-7s %-6s max err / max|ref| %.2e   rel L2 %.2e   median rel err %.2e" % (
                amp, name, nm, e.max() / ref.abs().max(), (e.pow(2).sum() / ref.pow(2).sum()).sqrt(), (e / ref.abs().clamp_min(1e-30)).median()))
