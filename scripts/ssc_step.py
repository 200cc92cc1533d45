"""Time the BEV-SSC training step (config 4 stage 2, per-GPU part) on the GPU box. usage: ssc_step.py [B] [precision]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import creste_public_amd
from creste_public_amd import harness, synth
from creste_public_amd.creste.models.terrainnet import TerrainNet
from creste_public_amd.creste.utils.loss_utils import LossManager
from test_train_terrain_gpu import _ssc_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
H, W = 608, 1216
creste_public_amd.set_precision(prec)
harness.seed_everything(0)
cfg = harness.ssc_cfg((H, W), class_weights=[0.5, 0.2, 0.1, 0.1, 0.05, 0.05])
model = TerrainNet(cfg).cuda()
synth.randomize_bn(model, seed=1)
synth.peak_depth_head(model)
tr = harness.SSCTrainer(model, LossManager(cfg).cuda(), cfg)
batch = _ssc_batch(B, H, W)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for it in range(4):
    t0 = T(); logs = tr.training_step(batch); t1 = T()
    print(f"it{it}: step {1e3 * (t1 - t0):.1f} ms  loss {float(logs['train/loss']):.4f}  "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
model.train(); tr.optimizer.zero_grad()
data = batch["joint"]
t0 = T(); out = model((data["image"], data["p2p"])); t1 = T()
td = {f"outputs/{k}": v for k, v in out.items()}; td.update({f"inputs/{k}": v for k, v in data.items()}); td["task"] = "joint"
ld, _ = tr.loss(td); loss = sum(w * v for w, v in ld.values()); t2 = T()
loss.backward(); t3 = T()
print(f"B={B} {W}x{H} {prec}: forward {1e3*(t1-t0):.1f} | losses {1e3*(t2-t1):.1f} | backward {1e3*(t3-t2):.1f} ms; "
      f"{B / (t3 - t0):.1f} frames/s fwd+bwd")
