"""the CF-IRL step at the 512x512 BEV grid (bench variant cf512): kernel-level view of one steady step"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench, creste_public_amd
from creste_public_amd import train_ops
train_ops.REWARD_FOLLOWS_F16X3 = os.environ.get('REWARD_F16X3') == '1'
dev = torch.device("cuda")
creste_public_amd.set_precision("f16x3")
model = bench.build_model(dev)
print(bench.irl_step_bench(model, dev, sys.argv[1] if len(sys.argv) > 1 else "cf512", steps=3))
