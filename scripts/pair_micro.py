"""conv3x3 -> conv3x3 pairs of the step (Up.conv), fused output -> input transform on / off, batch 16, bf16x6.
usage: pair_micro.py [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops
dev = torch.device("cuda")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def timeit(fn, n=iters):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (Cin, C, H, W) in [(496, 496, 152, 304), (472, 472, 76, 152), (432, 432, 38, 76), (320, 256, 128, 128)]:
    N = 16
    x = ops.Act(torch.randn(N, H, W, Cin, device=dev), Cin, 0)
    w1 = torch.randn(C, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5
    w2 = torch.randn(C, C, 3, 3, device=dev) / (C * 9) ** 0.5
    pc1 = ops.pack_conv(w1, None, None, 1, 1, ops.ACT_RELU, ops.PREC_BF16X6, algo=ops.ALGO_WINOGRAD4)
    pc2 = ops.pack_conv(w2, None, None, 1, 1, ops.ACT_RELU, ops.PREC_BF16X6, algo=ops.ALGO_WINOGRAD4)
    out = ops.Act.empty(N, H, W, C, dev)
    r = {}
    for f in (True, False, True, False):
        ops.FUSE_CONV_PAIRS = f
        r.setdefault(f, []).append(timeit(lambda: ops.conv2d_pair(x, pc1, pc2, out=out)))
    print(f"{Cin}->{C}->{C} @{H}x{W} x{N}: fused {min(r[True]):.3f} ms, separate {min(r[False]):.3f} ms")
