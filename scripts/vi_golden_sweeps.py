"""sweep counts of the HIP solver against the reference-generated golden vectors (tests/golden/vi_a.npz, vi_b.npz) and the margin of
the convergence test around the last sweep (GPU box)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops
for name in ("vi_a.npz", "vi_b.npz"):
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", name))
    r = torch.from_numpy(g["r"])[:, 0].cuda().contiguous()
    v, q, pi, sw = ops.value_iteration(r, float(g["discount"]), float(g["threshold"]))
    # the reference's loop on the CPU in float64 and float32: delta of the last few sweeps
    import torch.nn.functional as F
    def loop(dtype):
        rr = torch.from_numpy(g["r"]).to(dtype); vv = torch.zeros_like(rr); w = torch.zeros(8, 1, 3, 3, dtype=dtype)
        left = [[1, 0], [0, 0], [0, 1], [2, 0], [0, 2], [2, 1], [2, 2], [1, 2]]; center = [[0, 0], [0, 1], [0, 2], [1, 0], [1, 2], [2, 0], [2, 1], [2, 2]]
        right = [[0, 1], [0, 2], [1, 2], [0, 0], [2, 2], [1, 0], [2, 0], [2, 1]]
        for i in range(8):
            w[i, 0, left[i][0], left[i][1]] = 0.1; w[i, 0, center[i][0], center[i][1]] = 0.8; w[i, 0, right[i][0], right[i][1]] = 0.1
        n, hist = 0, []
        while True:
            nv = F.conv2d(rr + vv * float(g["discount"]), w, padding=1).max(dim=1, keepdim=True)[0]
            d = float((nv - vv).abs().max()); vv = nv; n += 1; hist.append(d)
            if not d > float(g["threshold"]): return n, hist[-3:]
    print(name, "HIP sweeps", int(sw.item()), "golden", int(g["sweeps"]), "| CPU fp32 loop", loop(torch.float32), "| CPU fp64 loop", loop(torch.float64), "threshold", float(g["threshold"]))
