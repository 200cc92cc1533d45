"""IRL train step with the next batch's frozen backbone enqueued BEFORE this batch's reward forward + value iteration (early) instead of
behind the value iteration (late = IRLTrainer's order): step time and the persistent solver's verdicts (GPU box).
usage: irl_early.py reference|mdp256|cf512 [steps]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench, creste_public_amd
from creste_public_amd import LossManager, MaxEntIRL, maxent_irl_cfg, synth, ops
variant = sys.argv[1] if len(sys.argv) > 1 else "reference"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
creste_public_amd.set_precision("bf16x6")
infer = bench.build_model(dev)
v = bench.IRL_VARIANTS[variant]
if v["prec"]:
    creste_public_amd.set_precision(v["prec"])
B, (GH, GW) = v["B"], v["bev"]
cfg = maxent_irl_cfg((bench.IMG_H, bench.IMG_W), solve_mdp=True, map_size=v["map_size"], map_ds=v["map_ds"], point_cloud_range=v["pcr"], voxel_size=v["voxel"])
model = MaxEntIRL(cfg)
sd = {k: t for k, t in infer.state_dict().items() if ".cam2map." not in k or "z_proj" in k or "vision_fusion" in k}
model.load_state_dict(sd, strict=False)
with torch.no_grad():
    model.traversability_head.r.postpool[0].norm.weight.mul_(0.01); model.traversability_head.r.postpool[0].norm.bias.mul_(0.01)
model = model.to(dev).train()
model.traversability_head.r.train_graphs = True
lm = LossManager(cfg).to(dev)
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-4)
rgbd, p2p = synth.make_frames(B, bench.IMG_H, bench.IMG_W, seed=4242); rgbd, p2p = rgbd.to(dev), p2p.to(dev)
expert = synth.make_experts(B, 50, (GH, GW), seed=5).to(dev)
fov = torch.ones(B, max(GH, 2 * v["map_size"][0]), max(GW, 2 * v["map_size"][1]), dtype=torch.bool, device=dev)
rng = np.random.RandomState(0)
c0 = np.array([[GH / 2 - 28.0, GW / 2.0]])
cf = [dict(trajectories=(c0 + np.linspace(0, 1, 20)[None, :, None] * rng.uniform(-0.3 * GW, 0.3 * GW, size=(2, 1, 2))).astype(np.float32), rank=np.array([0, 1])) for _ in range(B)]
hp = torch.cuda.Stream(priority=min(torch.cuda.Stream.priority_range()))
aborts = 0
def step(early):
    global aborts
    opt.zero_grad()
    inputs = (rgbd, p2p, expert)
    if model._prefetched is None:
        model.prefetch_backbone(inputs)
    pf = model._take_prefetched(inputs)
    if early:
        model.prefetch_backbone(inputs)
    out = model._forward_trainable(inputs, pf)
    if not early:
        model.prefetch_backbone(inputs)
    td = {f"outputs/{k}": t for k, t in out.items()}
    td.update({"inputs/traversability_label": expert, "inputs/fov_mask": fov, "inputs/counterfactuals_label": cf, "task": "irl"})
    ld, _ = lm(td)
    loss = sum(w * t for w, t in ld.values())
    loss.backward()
    for n in ops.vi_poll(wait=True):
        if n == ops.VI_ABORTED:
            aborts += 1
    opt.step()
    return loss.detach()
def run(early, use_hp=True):
    global aborts
    aborts = 0
    main = torch.cuda.current_stream()
    hp.wait_stream(main)
    with torch.cuda.stream(hp if use_hp else main):
        for _ in range(3): step(early)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): l = step(early)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / steps * 1e3
    main.wait_stream(hp)
    model._prefetched = None
    return ms, float(l), aborts
step(False); step(False); torch.cuda.synchronize(); model._prefetched = None
for rep in range(2):
    for use_hp in (False, True):
        for early in (False, True):
            ms, l, ab = run(early, use_hp)
            print(f"{variant} {'high-priority' if use_hp else 'default-prio '} stream, {'EARLY' if early else 'late '} prefetch: {ms:.2f} ms / step, aborted solves {ab} of {steps + 3}")
