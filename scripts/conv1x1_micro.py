import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops
dev = torch.device("cuda")
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (N, H, W, Cin, Cout) in [(16, 152, 304, 496, 256), (16, 152, 304, 288, 96), (16, 152, 304, 256, 128), (16, 152, 304, 128, 128), (16, 19, 38, 192, 1152), (16, 38, 76, 112, 672), (16, 38, 76, 80, 480), (16, 19, 38, 1152, 192), (16, 38, 76, 672, 112)]:
    x = ops.Act(torch.randn(N, H, W, Cin, device=dev), Cin, 0); x.amax = x.buf.abs().max().reshape(1)
    w = torch.randn(Cout, Cin, 1, 1, device=dev) / Cin ** 0.5
    r = []
    for act in (0, 1, 2):
        pc = ops.pack_conv(w, None, None, 1, 0, act, ops.PREC_F16X3)
        r.append(timeit(lambda: ops.conv2d(x, pc)) * 1e3)
    print(f"{Cin}->{Cout} @{H}x{W} x{N}: none {r[0]:.1f} us, relu {r[1]:.1f} us, swish {r[2]:.1f} us")
