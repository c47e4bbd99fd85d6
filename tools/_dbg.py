import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from advchain_amd import ops
from oracle import advchain_oracle as O
torch.manual_seed(0)
dev = "cuda"
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/reps*1e6
for dims, N in (((128,128,64),4), ((160,160,80),4)):
    d=3
    for amp in (0.5, 0.9, 1.2, 1.6, 2.0, 4.0, 7.0):
        low = torch.rand(N, d, *[max(2, s // 16) for s in dims]) * 2 - 1     # smooth like the solver's fields
        up = F.interpolate(low, size=dims, mode="trilinear", align_corners=True); up = up / up.abs().max()
        sc = torch.tensor([2.0 * amp / (dims[d - 1 - a] - 1) for a in range(d)]).view(1, d, 1, 1, 1)
        q = (O.identity_grid(N, dims) + up * sc).contiguous().to(dev)
        t = bench(lambda: ops.raw_compose_self_fwd(q))
        print("dims %s amp %.1f self fwd %.1f us" % (dims, amp, t))
