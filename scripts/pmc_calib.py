"""Known-byte-count launches for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this box (MI355X_MICROARCH.md, section
HBM: 'calibrate on a known byte count in your own access pattern'): run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc
WRITE_SIZE` (scripts/pmc_calib.sh), 4 launches each, in this order:
  fill   : 1 GiB written by torch's fill kernel (plain 16-byte stores)
  copy   : 1 GiB read + 1 GiB written by torch's copy kernel
  gather0: splat_gather8 over an EMPTY plan (every point outside the grid): 402.7 MB BEV + 4.2 MB densities written with the
           kernel's nontemporal 16-byte stores, next to nothing read
  gather1: splat_gather8 over the frustum distribution of scripts/splat_micro.py (284 MB of feature rows read, the same writes)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops, synth
dev = "cuda"
x = torch.empty(1 << 28, device=dev); y = torch.empty(1 << 28, device=dev)
for _ in range(4):
    x.fill_(1.0)
torch.cuda.synchronize()
for _ in range(4):
    y.copy_(x)
torch.cuda.synchronize()
B, Hs, Ws, F, G = 16, 152, 304, 96, 256
P = Hs * Ws
feats = ops.Act(torch.randn(B, Hs, Ws, F, device=dev), F)
far = torch.full((B, P, 3), 1000.0, device=dev)
for _ in range(4):
    ops.bev_splat(far, feats, (12.8, 12.8), (0.1, 0.1), G, G)
torch.cuda.synchronize()
torch.manual_seed(0)
depth = torch.rand(B, Hs, Ws, device=dev) * 25.1 + 0.3
p2p = synth.make_p2p(B, 608, 1216).to(dev)[:, 0]
dummy = torch.zeros(1, device=dev); one = torch.zeros(1, 1, device=dev)
bounds = torch.tensor([-3e38] * 3 + [3e38] * 3, device=dev)
xyz, _ = ops.pixel_geometry(depth, p2p.contiguous().float(), bounds, dummy, dummy, one, dummy, ops.Act.empty(B, Hs, Ws, 1, dev))
xyz = xyz.reshape(B, P, 3)
for _ in range(4):
    ops.bev_splat(xyz, feats, (12.8, 12.8), (0.1, 0.1), G, G)
torch.cuda.synchronize()
