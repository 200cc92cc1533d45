import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops
dev = torch.device("cuda")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
N, H, W, Cin, Cout = 16, 152, 304, 288, 96
w = torch.randn(Cout, Cin, 1, 1, device=dev) / Cin ** 0.5
pc = ops.pack_conv(w, None, None, 1, 0, 1, ops.PREC_F16X3)
for name, mk in (("randn", lambda: torch.randn(N, H, W, Cin, device=dev)), ("relu(randn)", lambda: torch.relu(torch.randn(N, H, W, Cin, device=dev))),
                 ("zeros", lambda: torch.zeros(N, H, W, Cin, device=dev)), ("randn*1e-3", lambda: torch.randn(N, H, W, Cin, device=dev) * 1e-3)):
    x = ops.Act(mk(), Cin, 0); x.amax = x.buf.abs().max().reshape(1).clamp_min(1e-30)
    out = ops.Act.empty(N, H, W, Cout, dev)
    print(f"{name:12s}: fresh output {timeit(lambda: ops.conv2d(x, pc))*1e3:7.1f} us, preallocated output {timeit(lambda: ops.conv2d(x, pc, out=out))*1e3:7.1f} us")
