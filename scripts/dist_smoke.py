"""Single-node distributed smoke (GPU box): run under torch.distributed.run with any --nproc-per-node that the box has
GPUs for; exercises RCCL init, the DistillTrainer's bucketed async all-reduce (GradArena) and the flat all-reduce of the
IRL / SSC trainers, and checks that every rank ends with identical parameters."""
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import creste_public_amd
from creste_public_amd import harness, synth
from creste_public_amd.creste.models.distillation import DistillationBackbone
from creste_public_amd.creste.utils.loss_utils import LossManager
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
creste_public_amd.set_precision("f16x3")
H, W, B = 128, 192, 2
harness.seed_everything(0)                                    # identical replicas
cfg = harness.distillation_cfg((H, W))
model = DistillationBackbone(cfg).cuda()
tr = harness.DistillTrainer(model, LossManager(cfg), cfg, bucket_mb=8)
rgbd, _ = synth.make_frames(B, H, W, seed=10 + rank)          # every rank its own frames
g = torch.Generator().manual_seed(20 + rank)
batch = {"image": rgbd.cuda(), "depth_label": (torch.rand(B, 1, H // 4, W // 4, generator=g) * 26000).cuda(),
         "fimg_label": torch.randn(B, 1, 128, H // 4, W // 4, generator=g).cuda()}
for _ in range(3):
    logs = tr.training_step(batch)
flat = torch.cat([p.detach().flatten() for p in model.parameters()])
ref = flat.clone()
dist.broadcast(ref, src=0)
same = bool(torch.equal(flat, ref))
ok = torch.tensor([int(same)], device="cuda")
dist.all_reduce(ok, op=dist.ReduceOp.MIN)
if rank == 0:
    print(f"world {world}: loss {float(logs['train/loss']):.4f}; parameters identical on every rank: {bool(ok.item())}; "
          f"arena buckets of 8 MB over {flat.numel()} parameters")
dist.destroy_process_group()
