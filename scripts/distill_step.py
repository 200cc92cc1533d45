"""Time the stage-1 distillation training step (config 4, per-GPU part) on the GPU box.
usage: distill_step.py [B] [precision] [H W]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import creste_public_amd
from creste_public_amd import harness, synth
from creste_public_amd.creste.models.distillation import DistillationBackbone
from creste_public_amd.creste.utils.loss_utils import LossManager
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (608, 1216)
creste_public_amd.set_precision(prec)
harness.seed_everything(0)
cfg = harness.distillation_cfg((H, W))
model = DistillationBackbone(cfg).cuda()
synth.randomize_bn(model, seed=1)
rgbd, _ = synth.make_frames(B, H, W, seed=2)
g = torch.Generator().manual_seed(3)
batch = {"image": rgbd.cuda(), "depth_label": (torch.rand(B, 1, H // 4, W // 4, generator=g) * 26000.0).cuda(),
         "fimg_label": torch.randn(B, 1, 128, H // 4, W // 4, generator=g).cuda()}
tr = harness.DistillTrainer(model, LossManager(cfg), cfg)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for it in range(4):
    t0 = T(); logs = tr.training_step(batch); t1 = T()
    print(f"it{it}: step {1e3 * (t1 - t0):.1f} ms  loss {float(logs['train/loss']):.4f}  "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
# forward / backward split
model.train(); tr.optimizer.zero_grad()
t0 = T(); out = model(batch["image"]); t1 = T()
td = {f"outputs/{k}": v for k, v in out.items()}; td.update({f"inputs/{k}": v for k, v in batch.items()}); td["task"] = None
ld, _ = tr.loss(td); loss = sum(w * v for w, v in ld.values()); t2 = T()
loss.backward(); t3 = T()
print(f"B={B} {W}x{H} {prec}: forward {1e3*(t1-t0):.1f} | losses {1e3*(t2-t1):.1f} | backward {1e3*(t3-t2):.1f} ms; "
      f"{B / (t3 - t0):.1f} frames/s fwd+bwd")
