"""Value iteration + SVF micro-benchmark (GPU box). usage: vi_micro.py B H W"""
import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops, synth
B, H, W = map(int, sys.argv[1:4])
torch.manual_seed(0)
r = torch.rand(B, H, W, device="cuda")
for _ in range(2):
    v, q, pi, sw = ops.value_iteration(r, 0.99, 1e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    v, q, pi, sw = ops.value_iteration(r, 0.99, 1e-3)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
n = int(sw.item())
bytes_ = B * H * W * (12 * n + 72)
print(f"VI B={B} {H}x{W}: {ms:.3f} ms, {n} sweeps, {ms / n * 1e3:.2f} us/sweep, algorithmic {bytes_ / ms / 1e6:.1f} GB/s")
if H <= 256:
    expert = synth.make_experts(B, 50, 2 * H if H == 64 else H * 2, seed=3)[:, :, :2, 2].contiguous().cuda()
    fov = torch.ones(H, W, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        out = ops.expected_svf(pi, expert, fov, 50, 2.0, 0.005, True, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = ops.expected_svf(pi, expert, fov, 50, 2.0, 0.005, True, False)
    torch.cuda.synchronize()
    ms2 = (time.perf_counter() - t0) / 10 * 1e3
    print(f"SVF B={B} {H}x{W} T=50: {ms2:.3f} ms, algorithmic {B * H * W * (32 + 8 * 49 + 4) / ms2 / 1e6:.1f} GB/s")
