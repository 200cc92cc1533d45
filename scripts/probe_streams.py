import os, sys, torch
sys.path.insert(0, "/root/repo")
import creste_public_amd
from creste_public_amd import ops
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
s = ops.concurrent_stream(dev, "parts")
print("found", s is not None, "probe log (us):", ops._probe_log, "GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))
