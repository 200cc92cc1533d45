"""Time the pieces of the IRL train step (GPU box)."""
import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import bench
import creste_public_amd
from creste_public_amd import LossManager, MaxEntIRL, maxent_irl_cfg, synth
creste_public_amd.set_precision(os.environ.get("PREC", "f16x3"))
dev = torch.device("cuda", 0)
B = 8
cfg = maxent_irl_cfg((bench.IMG_H, bench.IMG_W), solve_mdp=True)
torch.manual_seed(0)
model = MaxEntIRL(cfg)
synth.randomize_bn(model, seed=1)
with torch.no_grad():
    model.backbone.depthcomp.depthcomp.vision_backbone.model.trunk._bn0.running_var.fill_(1e7)
    model.traversability_head.r.postpool[0].norm.weight.mul_(0.01); model.traversability_head.r.postpool[0].norm.bias.mul_(0.01)
model = model.to(dev).train()
model.traversability_head.r.train_graphs = bool(int(os.environ.get('GRAPHS', '1')))
lm = LossManager(cfg).to(dev)
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-4)
rgbd, p2p = synth.make_frames(B, bench.IMG_H, bench.IMG_W, seed=1); rgbd, p2p = rgbd.to(dev), p2p.to(dev)
expert = synth.make_experts(B, 50, 256, seed=5).to(dev)
fov = torch.ones(B, 256, 256, dtype=torch.bool, device=dev)
rng = np.random.RandomState(0)
cf = [dict(trajectories=(np.array([[100.0, 128.0]]) + np.linspace(0, 1, 20)[None, :, None] * rng.uniform(-80, 80, size=(2, 1, 2))).astype(np.float32), rank=np.array([0, 1])) for _ in range(B)]
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for it in range(5):
    t0 = T(); opt.zero_grad()
    r = model.backbone.forward_act(rgbd, p2p); outputs = model.backbone.pack_outputs(r, B); t1 = T()
    head = model.traversability_head
    view = head.input_view_act(r["preds_buf"]); t2 = T()
    S = expert[:, :, :2, 2].long() // 2
    o = head.forward_from_view(view, 256, 256, S, solve_mdp=True); outputs.update(o); t3 = T()
    with torch.no_grad():
        outputs.update(model.expected_state_visitation_frequency(outputs["policy"], expert)); t4 = T()
    td = {f"outputs/{k}": v for k, v in outputs.items()}
    td.update({"inputs/traversability_label": expert, "inputs/fov_mask": fov, "inputs/counterfactuals_label": cf, "task": "irl"})
    ld, _ = lm(td); loss = sum(w * v for w, v in ld.values()); t5 = T()
    loss.backward(); t6 = T()
    opt.step(); t7 = T()
    print(f"it{it}: backbone {1e3*(t1-t0):.1f} | pool {1e3*(t2-t1):.2f} | reward fwd + VI {1e3*(t3-t2):.1f} | svf {1e3*(t4-t3):.2f} | loss {1e3*(t5-t4):.1f} | backward {1e3*(t6-t5):.1f} | adam {1e3*(t7-t6):.2f} | total {1e3*(t7-t0):.1f}")
