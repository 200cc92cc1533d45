"""Winograd F(2x2,3x3) vs the direct 3x3 kernel on one layer shape (tuning aid; run on the GPU box).
usage: wino_micro.py PREC Cin Cout H W [N] [iters]"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops
prec = {"bf16": 1, "bf16x3": 2, "bf16x6": 3}[sys.argv[1]]
Cin, Cout, H, W = map(int, sys.argv[2:6])
N = int(sys.argv[6]) if len(sys.argv) > 6 else 16
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 10
torch.manual_seed(0)
x = ops.Act(torch.relu(torch.randn(N, H, W, Cin, device="cuda")), Cin)
w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5
out = ops.Act.empty(N, H, W, Cout, "cuda")
res = {}
for name, algo in (("direct", ops.ALGO_DIRECT), ("winograd", ops.ALGO_WINOGRAD), ("winograd4", ops.ALGO_WINOGRAD4)):
    pc = ops.pack_conv(w, None, None, 1, 1, ops.ACT_RELU, prec, algo=algo)
    for _ in range(2):
        ops.conv2d(x, pc, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv2d(x, pc, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * N * H * W * Cout * Cin * 9
    res[name] = out.buf.clone()
    print(f"{sys.argv[1]} {name:9s} {Cin}->{Cout} {H}x{W} N={N}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s (algorithmic)")
d = (res["direct"] - res["winograd"]).double()
print(f"   winograd vs direct: rel rms {float(d.pow(2).mean().sqrt() / res['direct'].double().pow(2).mean().sqrt()):.2e}")
d = (res["direct"] - res["winograd4"]).double()
print(f"   winograd4 vs direct: rel rms {float(d.pow(2).mean().sqrt() / res['direct'].double().pow(2).mean().sqrt()):.2e}")
