"""training-mode BatchNorm forward / backward at the encoder's largest map: achieved HBM rate"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import train_ops as T, ops
dev = torch.device("cuda")
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (N, C, H, W) in [(8, 496, 152, 304), (8, 256, 256, 256), (8, 144, 152, 304), (8, 672, 38, 76)]:
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    op = T.BNT(bn, True)
    x = ops.Act(torch.randn(N, H, W, C, device=dev), C, 0)
    gy = ops.Act(torch.randn(N, H, W, C, device=dev), C, 0)
    by = x.buf.numel() * 4
    tf = timeit(lambda: op.fwd(x))
    op.fwd(x)
    tb = timeit(lambda: op.bwd(gy, None, {}))
    print(f"C={C} {H}x{W}x{N}: forward {tf*1e3:7.1f} us = {3*by/tf/1e9:.2f} TB/s over 2 reads + 1 write; backward {tb*1e3:7.1f} us = {5*by/tb/1e9:.2f} TB/s over 4 reads + 1 write")
