import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops
torch.manual_seed(0)
shape = (8, 160, 256)
r = torch.rand(*shape, device="cuda")
for ms in (8, 16, 24):
    os.environ["CRESTE_VI_SYNC"] = "1"
    v0, q0, p0, s0 = ops.value_iteration(r, 0.99, 1e-3, max_sweeps=ms)
    os.environ["CRESTE_VI_SYNC"] = "0"
    v1, q1, p1, s1 = ops.value_iteration(r, 0.99, 1e-3, max_sweeps=ms)
    torch.cuda.synchronize()
    d = (v0 - v1).abs()
    idx = torch.nonzero(d > 0)
    print("max_sweeps", ms, "differing", idx.shape[0])
    if idx.shape[0]:
        b, y, x = idx[:, 0], idx[:, 1], idx[:, 2]
        print("  images", sorted(set(b.tolist())), "tile rows", sorted(set((y // 32).tolist())), "tile cols", sorted(set((x // 32).tolist())))
        print("  y%32", sorted(set((y % 32).tolist())), "x%32", sorted(set((x % 32).tolist())))
        print("  first", idx[:12].tolist())
