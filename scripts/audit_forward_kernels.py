"""List every GPU kernel / memcpy / memset of ONE steady-state inference forward (after 2 warm-ups) that is not a
creste:: kernel -- what a Python-free replay of the C-ABI call sequence would miss.
    python scripts/audit_forward_kernels.py [precision] [B]"""
import os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, ROOT)
import creste_public_amd
from creste_public_amd import MaxEntIRL, maxent_irl_cfg, synth
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
H, W = 608, 1216
creste_public_amd.set_precision(prec)
torch.manual_seed(0)
m = MaxEntIRL(maxent_irl_cfg((H, W), solve_mdp=False))
synth.randomize_bn(m, seed=1)
m = m.cuda().eval()
rgbd, p2p = synth.make_frames(B, H, W, seed=3)
rgbd, p2p = rgbd.cuda(), p2p.cuda()
with torch.no_grad():
    for _ in range(2):
        m((rgbd, p2p))
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        out = m((rgbd, p2p))
        torch.cuda.synchronize()
names = {}
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        names[e.name] = names.get(e.name, 0) + 1
tot = sum(names.values())
other = {k: v for k, v in names.items() if "creste" not in k}
print(f"{prec} B={B}: {tot} device activities in one forward, {sum(other.values())} not creste:: kernels")
for k, v in sorted(other.items(), key=lambda kv: -kv[1]):
    print(f"  {v:4d}  {k[:150]}")
