"""Which python-level ops launch the non-creste:: kernels of one BEV-SSC training step (GPU box)?"""
import os, sys, collections, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import creste_public_amd
from creste_public_amd import harness, synth
from creste_public_amd.creste.models.terrainnet import TerrainNet
from creste_public_amd.creste.utils.loss_utils import LossManager
from test_train_terrain_gpu import _ssc_batch
B, H, W = 8, 608, 1216
creste_public_amd.set_precision("f16x3")
harness.seed_everything(0)
cfg = harness.ssc_cfg((H, W), class_weights=[0.5, 0.2, 0.1, 0.1, 0.05, 0.05])
model = TerrainNet(cfg).cuda()
synth.randomize_bn(model, seed=1); synth.peak_depth_head(model)
tr = harness.SSCTrainer(model, LossManager(cfg).cuda(), cfg)
batch = _ssc_batch(B, H, W)
for _ in range(2): tr.training_step(batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True) as prof:
    tr.training_step(batch); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
evs = prof.events()
for e in evs:
    if e.device_type == torch.autograd.DeviceType.CUDA and "creste" not in e.name and "mpc_" not in e.name and "max2_f32" not in e.name:
        a = agg[e.name[:90]]
        a[0] += 1; a[1] += e.device_time if hasattr(e, "device_time") else e.cuda_time
for k, (n, t, _) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{n:5d} {t/1e3:8.3f} ms  {k}")
# aten::copy_ calls with device time, by the innermost frames inside the package
ops = collections.defaultdict(lambda: [0, 0.0])
for e in evs:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name in ("aten::copy_", "aten::add_", "aten::add") :
        dt = getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
        if dt <= 0: continue
        st = [s_.split("/")[-1][:60] for s_ in (e.stack or []) if "creste_public_amd" in s_ or "harness" in s_]
        ops[(e.name, " <- ".join(st[:2]) if st else "?")][0] += 1
        ops[(e.name, " <- ".join(st[:2]) if st else "?")][1] += dt
for (name, where), (n, t) in sorted(ops.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{n:5d} {t/1e3:8.3f} ms  {name:12s} {where}")
