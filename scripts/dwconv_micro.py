"""depthwise conv + swish + SE sums at the bench sizes (blocks 4..15 of EfficientNet-B0 at 608x1216, batch 16)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops
dev = torch.device("cuda")
B = 16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
# (C, K, S, H, W, count per step)
CASES = [(96, 3, 2, 304, 608, 0), (144, 3, 1, 152, 304, 0), (240, 5, 1, 76, 152, 1), (240, 3, 2, 76, 152, 1), (480, 3, 1, 38, 76, 2), (480, 5, 1, 38, 76, 1), (672, 5, 1, 38, 76, 2),
         (672, 5, 2, 38, 76, 1), (1152, 5, 1, 19, 38, 3), (1152, 3, 1, 19, 38, 1)]
tot = 0.0
for (C, K, S, H, W, cnt) in CASES:
    x = ops.Act(torch.randn(B, H, W, C, device=dev), C, 0)
    w = torch.randn(K * K, C, device=dev) / K; b = torch.randn(C, device=dev)
    Cse = max(1, C // 24)
    w1, b1 = torch.randn(Cse, C, device=dev) / C ** 0.5, torch.randn(Cse, device=dev)
    w2, b2 = torch.randn(C, Cse, device=dev) / Cse ** 0.5, torch.randn(C, device=dev)
    pad = (K // 2, K // 2, K // 2, K // 2) if S == 1 else ((K - 2) // 2, (K - 2) - (K - 2) // 2 + 0, (K - 2) // 2, (K - 2) - (K - 2) // 2)
    ops.DW_TILE = False
    o0, g0 = ops.dwconv2d_se(x, w, b, K, S, pad, ops.ACT_SWISH, w1, b1, w2, b2)
    ms0 = timeit(lambda: ops.dwconv2d_se(x, w, b, K, S, pad, ops.ACT_SWISH, w1, b1, w2, b2))
    ops.DW_TILE = True
    o1, g1 = ops.dwconv2d_se(x, w, b, K, S, pad, ops.ACT_SWISH, w1, b1, w2, b2)
    dd, dg = (o1.buf - o0.buf).abs().max().item(), (g1 - g0).abs().max().item()
    ms = timeit(lambda: ops.dwconv2d_se(x, w, b, K, S, pad, ops.ACT_SWISH, w1, b1, w2, b2))
    Ho, Wo = (H + pad[0] + pad[1] - K) // S + 1, (W + pad[2] + pad[3] - K) // S + 1
    by = (B * H * W * C + B * Ho * Wo * C) * 4
    tot += ms * cnt
    print(f"C={C:5d} k{K} s{S} {H}x{W}: register-blocked {ms0*1e3:7.1f} us, tile {ms*1e3:7.1f} us  {by/ms/1e9:5.2f} TB/s  (x{cnt} per step)  max|diff| out {dd:.1e} gate {dg:.1e}")
print(f"per step: {tot:.3f} ms")
