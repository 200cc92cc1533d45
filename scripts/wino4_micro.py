"""F(4x4,3x3) call on one layer shape, timed by HIP events (tuning aid; run on the GPU box, optionally under rocprofv3
--kernel-trace --stats for the per-kernel split).  usage: wino4_micro.py Cin Cout H W [N] [iters] [up]"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops
Cin, Cout, H, W = map(int, sys.argv[1:5])
N = int(sys.argv[5]) if len(sys.argv) > 5 else 16
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 10
torch.manual_seed(0)
x = ops.Act(torch.relu(torch.randn(N, H, W, Cin, device="cuda")), Cin)
w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5
out = ops.Act.empty(N, H, W, Cout, "cuda")
pc = ops.pack_conv(w, None, None, 1, 1, ops.ACT_RELU, ops.PREC_BF16X6, algo=ops.ALGO_WINOGRAD4)
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
print(f"order={os.environ.get('CRESTE_W4_ORDER')} f32v={os.environ.get('CRESTE_W4_F32V')} {Cin}->{Cout} {H}x{W} N={N}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s  checksum {float(out.buf.double().sum()):.6e}")
