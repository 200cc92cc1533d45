"""BEV splat gather over an EMPTY plan (every cell a zero row: the pure store path of splat_gather8_kernel) against a synthetic frustum
plan, batch 16 (GPU box): how much of the gather's 145-153 us is the 403 MB of output stores alone?"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops
dev = torch.device("cuda")
B, P, F, G = 16, 46208, 96, 256
feats = ops.Act(torch.randn(B, 152, 304, F, device=dev), F, 0)
off, vox = (-12.8, -12.8), (0.1, 0.1)
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
far = torch.full((B, P, 3), 1.0e3, device=dev)                      # every point outside the grid
g = torch.Generator(device="cuda").manual_seed(1)
fr = torch.rand(B, P, 3, device=dev, generator=g)
fr[..., 0] = (fr[..., 0] ** 2) * 12.0                               # a wedge: dense near the sensor, sparse far away
fr[..., 1] = (fr[..., 1] - 0.5) * fr[..., 0] * 1.2
for name, xyz in (("empty plan", far), ("synthetic wedge", fr)):
    plan = ops.bev_splat_plan(xyz.contiguous(), off, vox, G, G)
    us = t(lambda: ops.bev_splat_gather(plan, feats, 1.0, "mean"))
    bev, dens = ops.bev_splat_gather(plan, feats, 1.0, "mean")
    print(f"{name}: gather {us:.1f} us; occupied cells {100 * float((dens > 0).float().mean()):.1f} %; 403 MB of output = {403e6 / us / 1e6:.2f} TB/s of stores")
x = torch.empty(B * G * G * (F + 1), device=dev)
print(f"torch fill of the same bytes: {t(lambda: x.zero_()):.1f} us")
