"""F(4x4,3x3) conv over cat([skip, up2x(x1)]) formed inside the input transform (tuning aid; GPU box).
usage: wino_up_micro.py C1 C2 H1 W1 Cout [N] [iters]"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops
C1, C2, H1, W1, Cout = map(int, sys.argv[1:6])
N = int(sys.argv[6]) if len(sys.argv) > 6 else 16
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 10
torch.manual_seed(0)
x1 = ops.Act(torch.relu(torch.randn(N, H1, W1, C1, device="cuda")), C1)
skip = ops.Act(torch.relu(torch.randn(N, 2 * H1, 2 * W1, C2, device="cuda")), C2) if C2 else None
w = torch.randn(Cout, C1 + C2, 3, 3, device="cuda") / ((C1 + C2) * 9) ** 0.5
pc = ops.pack_conv(w, None, None, 1, 1, ops.ACT_RELU, ops.PREC_BF16X6, algo=ops.ALGO_WINOGRAD4)
out = ops.Act.empty(N, 2 * H1, 2 * W1, Cout, "cuda")
for fused in (True, False):
    ops.FUSE_UPSAMPLE = fused
    def run():
        ops.conv2d(ops.upsample_concat_lazy(x1, skip, 2 * H1, 2 * W1, 0.5, 0.5), pc, out=out)
    run(); run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record(); torch.cuda.synchronize()
    print(f"{C2}+up({C1}) -> {Cout} @{2 * H1}x{2 * W1} N={N} {'fused' if fused else 'materialised'}: {e0.elapsed_time(e1) / iters:.3f} ms")
