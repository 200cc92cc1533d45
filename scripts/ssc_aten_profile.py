import os, sys
sys.argv = ["ssc_step.py", "8", "bf16x6"]
import torch
from torch.profiler import profile, ProfilerActivity
src = open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "scripts/ssc_step.py")).read().split("model.train(); tr.optimizer.zero_grad()")[0]
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "scripts"))
__file__ = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "scripts/ssc_step.py")
exec(src)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    tr.training_step(batch); torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:14]:
    print(f"{e.key:40s} calls {e.count:4d}  device {e.device_time_total/1e3:8.3f} ms")
