"""1x1 convs of the small encoder maps: conv_patch_kernel<1,...> against conv1x1_deep_kernel (CRESTE_CONV1X1_DEEP=0 / 1 / 2, read once
per process): time per call with cold-ish inputs (a rotating set of input buffers) and a bit pattern checksum of the output.
usage: CRESTE_CONV1X1_DEEP=0|1|2 python scripts/conv1x1_deep_ab.py [prec]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops
dev = torch.device("cuda")
prec = {"bf16x6": ops.PREC_BF16X6, "bf16x3": ops.PREC_BF16X3, "bf16": ops.PREC_BF16}[sys.argv[1] if len(sys.argv) > 1 else "bf16x6"]
def timeit(fn, n=40):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
SHAPES = [(19, 38, 1152, 192, 1), (19, 38, 192, 1152, 0), (19, 38, 1152, 320, 1), (19, 38, 672, 192, 1), (38, 76, 112, 672, 0), (38, 76, 672, 112, 1),
          (38, 76, 80, 480, 0), (38, 76, 480, 80, 1), (38, 76, 480, 112, 1), (38, 76, 240, 80, 1), (76, 152, 40, 240, 0), (76, 152, 240, 40, 1),
          (76, 152, 144, 40, 1), (152, 304, 144, 24, 1), (152, 304, 96, 24, 1), (152, 304, 288, 96, 0), (152, 304, 128, 128, 0), (152, 304, 256, 128, 0), (152, 304, 496, 256, 0), (128, 128, 512, 256, 0), (256, 256, 128, 32, 0)]
torch.manual_seed(0)
for N in (8, 16):
    for (H, W, Cin, Cout, gated) in SHAPES:
        xs = [ops.Act(torch.randn(N, H, W, Cin, device=dev), Cin, 0) for _ in range(4)]
        w = torch.randn(Cout, Cin, 1, 1, device=dev) / Cin ** 0.5
        gate = torch.rand(N, Cin, device=dev) if gated else None
        pc = ops.pack_conv(w, None, None, 1, 0, 0, prec)
        i = [0]
        def run():
            i[0] += 1
            return ops.conv2d(xs[i[0] % 4], pc, a_scale=gate)
        t = timeit(run)
        y = ops.conv2d(xs[0], pc, a_scale=gate).buf
        torch.cuda.synchronize()
        chk = int(y.view(torch.int32).to(torch.int64).sum().item())
        print(f"N{N} {Cin:5d}->{Cout:5d} @{H}x{W} gate{gated}: {t:7.1f} us  chk {chk}")
