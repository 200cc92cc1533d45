import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops
dev = torch.device("cuda")
for (N, H, W, Cin, Cout) in [(16, 152, 304, 144, 24), (16, 304, 608, 32, 16), (16, 256, 256, 128, 32), (16, 152, 304, 288, 96)]:
    x = ops.Act(torch.randn(N, H, W, Cin, device=dev), Cin, 0); x.amax = x.buf.abs().max().reshape(1)
    w = torch.randn(Cout, Cin, 1, 1, device=dev) / Cin ** 0.5
    pc = ops.pack_conv(w, None, None, 1, 0, 0, ops.PREC_F16X3)
    out = ops.Act.empty(N, H, W, Cout, dev)
    for _ in range(3): ops.conv2d(x, pc, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.conv2d(x, pc, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    by = (x.buf.numel() + out.buf.numel()) * 4
    print(f"{Cin}->{Cout} @{H}x{W} x{N}: {ms*1e3:.1f} us, algorithmic {by/1e6:.0f} MB -> {by/ms/1e9:.2f} TB/s")
