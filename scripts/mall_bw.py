"""Streaming bandwidth vs working-set size: does a producer -> consumer pair whose tensors fit the 256 MB Infinity Cache
run faster than one that streams through HBM?  (ping-pong copy x -> y -> x of S bytes each; working set 2S)"""
import torch
for mb in (8, 16, 32, 48, 64, 96, 128, 192, 256, 512, 1024):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device="cuda"); y = torch.empty_like(x)
    for _ in range(3): y.copy_(x); x.copy_(y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(4, 2048 // mb)
    e0.record()
    for _ in range(reps): y.copy_(x); x.copy_(y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (2 * reps)
    print(f"S = {mb:5d} MB (working set {2*mb:5d} MB): copy {ms*1e3:8.1f} us -> {2*n*4/ms/1e9:6.2f} TB/s read+write")
