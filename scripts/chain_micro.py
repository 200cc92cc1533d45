"""the distillation head (256 -> 128 -> 128 -> 128 at 152 x 304, batch 16): one fused launch against three launches of the 1x1 engine"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops
dev = torch.device("cuda")
N, H, W = 16, 152, 304
x = ops.Act(torch.randn(N, H, W, 256, device=dev), 256, 0)
layers = []
for ci in (256, 128, 128):
    layers.append((torch.randn(128, ci, 1, 1, device=dev) / ci ** 0.5, torch.randn(128, device=dev) * 0.1, None))
pk = ops.pack_conv1x1_chain3(layers, ops.PREC_BF16X6)
pcs = [ops.pack_conv(w, b, None, 1, 0, ops.ACT_RELU, ops.PREC_BF16X6) for w, b, _ in layers]
out = ops.Act.empty(N, H, W, 128, dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def three():
    z = x
    for pc in pcs: z = ops.conv2d(z, pc)
print(f"fused chain   {t(lambda: ops.conv1x1_chain3(x, pk, out=out)):8.1f} us")
print(f"three launches {t(three):8.1f} us")
