"""aten ops (host side) and non-creste GPU kernels of ONE steady-state IRL training step (GPU box)."""
import os, sys, torch
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench, creste_public_amd
from creste_public_amd import LossManager, MaxEntIRL, maxent_irl_cfg, synth
creste_public_amd.set_precision("f16x3")
dev = torch.device("cuda", 0)
B = 8
cfg = maxent_irl_cfg((bench.IMG_H, bench.IMG_W), solve_mdp=True)
torch.manual_seed(0)
model = MaxEntIRL(cfg)
synth.randomize_bn(model, seed=1)
with torch.no_grad():
    model.traversability_head.r.postpool[0].norm.weight.mul_(0.01); model.traversability_head.r.postpool[0].norm.bias.mul_(0.01)
model = model.to(dev).train()
model.traversability_head.r.train_graphs = True
lm = LossManager(cfg).to(dev)
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-4)
rgbd, p2p = synth.make_frames(B, bench.IMG_H, bench.IMG_W, seed=1); rgbd, p2p = rgbd.to(dev), p2p.to(dev)
expert = synth.make_experts(B, 50, 256, seed=5).to(dev)
fov = torch.ones(B, 256, 256, dtype=torch.bool, device=dev)
rng = np.random.RandomState(0)
cf = [dict(trajectories=(np.array([[100.0, 128.0]]) + np.linspace(0, 1, 20)[None, :, None] * rng.uniform(-80, 80, size=(2, 1, 2))).astype(np.float32), rank=np.array([0, 1])) for _ in range(B)]
def step():
    opt.zero_grad()
    out = model((rgbd, p2p, expert))
    td = {f"outputs/{k}": v for k, v in out.items()}
    td.update({"inputs/traversability_label": expert, "inputs/fov_mask": fov, "inputs/counterfactuals_label": cf, "task": "irl"})
    ld, _ = lm(td)
    loss = sum(w * v for w, v in ld.values())
    loss.backward()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    step(); torch.cuda.synchronize()
ev = prof.events()
k = {}
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA and "creste" not in e.name:
        k[e.name[:100]] = k.get(e.name[:100], 0) + 1
print("non-creste device activities:", sum(k.values()))
for n, c in sorted(k.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {c:4d} {n}")
ops = {}
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::"):
        ops[e.name] = ops.get(e.name, 0) + 1
print("aten ops:", sum(ops.values()))
for n, c in sorted(ops.items(), key=lambda kv: -kv[1])[:30]:
    print(f"  {c:4d} {n}")
