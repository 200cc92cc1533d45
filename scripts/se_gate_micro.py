"""squeeze-excite gate kernel: time per call and a bit checksum of the gates (compare builds with CRESTE_HIP_LIB=...)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops
dev = torch.device("cuda")
def timeit(fn, n=50):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
torch.manual_seed(0)
for N in (8, 16, 1):
    for (H, W, C, Cse) in [(19, 38, 1152, 48), (19, 38, 672, 28), (38, 76, 672, 28), (38, 76, 480, 20), (38, 76, 240, 10), (76, 152, 240, 10), (76, 152, 144, 6), (152, 304, 96, 4), (5, 7, 36, 3)]:
        x = ops.Act(torch.randn(N, H, W, C, device=dev), C, 0)
        w1, b1 = torch.randn(Cse, C, device=dev) / C ** 0.5, torch.randn(Cse, device=dev)
        w2, b2 = torch.randn(C, Cse, device=dev) / Cse ** 0.5, torch.randn(C, device=dev)
        t = timeit(lambda: ops.se_gate(x, w1, b1, w2, b2))
        g = ops.se_gate(x, w1, b1, w2, b2)
        ref = torch.sigmoid(torch.nn.functional.silu(x.buf.double().mean((1, 2)) @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double())
        err = float((g.double() - ref).abs().max())
        print(f"N{N} {H}x{W} C{C} Cse{Cse}: {t:6.1f} us (partial sums + gate)  chk {int(g.view(torch.int32).to(torch.int64).sum().item())}  max err vs float64 {err:.1e}")
