"""IRL train-step timing only (bench.py's irl_extras) -- tuning aid for the GPU box.  usage: irl_bench.py [precision]"""
import json, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench, creste_public_amd
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
creste_public_amd.set_precision(prec)
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
out = {k: bench.irl_step_bench(model, dev, k) for k in (sys.argv[2:] or list(bench.IRL_VARIANTS))}
print(json.dumps(out, indent=1))
