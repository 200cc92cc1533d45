"""BEV splat micro-benchmark at the bench shape (GPU box): B=16, P=46208, F=96, 256x256 grid."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops
B, P, F, G = 16, 152 * 304, 96, 256
torch.manual_seed(0)
xyz = torch.empty(B, P, 3, device="cuda")
xyz[..., 0] = torch.rand(B, P, device="cuda") * 30 - 2       # forward 0..28 m (part out of range)
xyz[..., 1] = torch.rand(B, P, device="cuda") * 30 - 15
xyz[..., 2] = torch.rand(B, P, device="cuda") * 3 - 2
feats = ops.Act(torch.randn(B, 152, 304, F, device="cuda"), F)
for _ in range(3):
    out = ops.bev_splat(xyz, feats, (12.8, 12.8), (0.1, 0.1), G, G)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    out = ops.bev_splat(xyz, feats, (12.8, 12.8), (0.1, 0.1), G, G)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
alg = 4.0 * (F * P + 2 * P + F * G * G + G * G) * B
print(f"bev_splat B={B} P={P} F={F}: {ms * 1e3:.1f} us  algorithmic {alg / 1e6:.1f} MB -> {alg / ms / 1e9:.2f} TB/s = {alg / ms / 1e9 / 8 * 100:.1f}% of 8 TB/s")
