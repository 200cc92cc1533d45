"""The 1x1 convs of one batch-16 inference step (bf16x6), one line per layer: us per launch and the HBM rate of its
algorithmic bytes (one read of the input, one write of the output).  A/B of two builds on one box:
  python scripts/conv1x1_ab.py ; CRESTE_HIP_LIB=$PWD/creste_public_amd/lib/libcreste_hip_prev.so python scripts/conv1x1_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops
dev = torch.device("cuda")
prec = {"bf16x6": ops.PREC_BF16X6, "f16x3": ops.PREC_F16X3, "bf16": ops.PREC_BF16}[os.environ.get("PREC", "bf16x6")]


def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


LAYERS = [(152, 304, 496, 256), (152, 304, 128, 128), (152, 304, 288, 96), (152, 304, 256, 128), (19, 38, 1152, 192),
          (38, 76, 112, 672), (76, 152, 40, 240), (19, 38, 192, 1152), (256, 256, 128, 32), (256, 256, 128, 6), (256, 256, 128, 2),
          (304, 608, 32, 16), (38, 76, 80, 480), (38, 76, 672, 112), (152, 304, 144, 24), (38, 76, 480, 80), (152, 304, 96, 24),
          (19, 38, 1152, 320), (76, 152, 240, 40), (38, 76, 480, 112), (19, 38, 672, 192), (76, 152, 144, 40), (38, 76, 240, 80),
          (64, 128, 48, 1)]
tot = 0.0
for (H, W, Cin, Cout) in LAYERS:
    N = 16
    x = ops.Act(torch.randn(N, H, W, Cin, device=dev), Cin, 0); x.amax = x.buf.abs().max().reshape(1)
    w = torch.randn(Cout, Cin, 1, 1, device=dev) / Cin ** 0.5
    pc = ops.pack_conv(w, None, None, 1, 0, 1, prec)
    out = ops.Act.empty(N, H, W, (Cout + 3) // 4 * 4, dev) if Cout % 4 else None
    us = timeit(lambda: ops.conv2d(x, pc)) * 1e3
    by = 4.0 * N * H * W * (Cin + Cout)
    tot += us
    print(f"{Cin:5d}->{Cout:4d} @{H}x{W}: {us:8.1f} us  {by / us / 1e6:6.2f} TB/s  {2.0 * N * H * W * Cin * Cout / us / 1e6:7.1f} TF")
print(f"sum {tot:.1f} us")
