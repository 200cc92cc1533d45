"""1-rank RCCL process group: the distillation step with / without the gradient collectives, per collective-issue strategy
(CRESTE_COLL_ISSUE = join | third | nojoin) and with the weight-gradient stream off (CRESTE_WGRAD_STREAM=0)."""
import os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import bench, creste_public_amd
from creste_public_amd import dist_utils, ops
creste_public_amd.set_precision("bf16x6")
device = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "distill"
objs = (bench._distill_setup if which == "distill" else bench._ssc_setup)(device, 8, seed=0)
step = objs[0]
step(); torch.cuda.synchronize()
r = dist_utils.measure_dp_step(step, 5, 8, device=device, warmup=1)
print(which, os.environ.get("CRESTE_COLL_ISSUE", "join"), "wgrad stream", os.environ.get("CRESTE_WGRAD_STREAM", "1"),
      {k: r[k] for k in ("step_ms", "step_ms_no_collective", "allreduce_exposed_ms", "collective_calls")}, "probes", ops._probe_log)
dist.destroy_process_group()
