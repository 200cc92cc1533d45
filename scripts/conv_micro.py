"""Micro-benchmark of one conv shape through the C ABI (tuning aid; run on the GPU box).
usage: [STRIDE=2] conv_micro.py PREC Cin Cout K H W [N] [iters]   (H, W: input extent)"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops
prec = {"f32": 0, "bf16": 1, "bf16x3": 2, "bf16x6": 3, "f16x3": 4}[sys.argv[1]]
Cin, Cout, K, H, W = map(int, sys.argv[2:7])
N = int(sys.argv[7]) if len(sys.argv) > 7 else 16
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 10
torch.manual_seed(0)
x = ops.Act((torch.zeros if os.environ.get("ZERO") else torch.randn)(N, H, W, Cin, device="cuda"), Cin)
w = torch.randn(Cout, Cin, K, K, device="cuda") / (Cin * K * K) ** 0.5
S = int(os.environ.get("STRIDE", "1"))
prec = ops.conv_precision(prec, K, S, Cin)
pc = ops.pack_conv(w, None, None, S, K // 2, ops.ACT_RELU, prec)
Ho, Wo = (H + 2 * (K // 2) - K) // S + 1, (W + 2 * (K // 2) - K) // S + 1
out = ops.Act.empty(N, Ho, Wo, Cout, "cuda")
for _ in range(2):
    ops.conv2d(x, pc, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ops.conv2d(x, pc, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = 2.0 * N * Ho * Wo * Cout * Cin * K * K
print(f"{sys.argv[1]} (engine {prec}) {Cin}->{Cout} k{K}/{S} {H}x{W} N={N}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s (algorithmic)")
