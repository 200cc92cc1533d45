"""Host time to ISSUE one inference step (return of model() without synchronising) vs its GPU time, one stream and pipelined,
with and without an initialised RCCL process group (its watchdog thread)."""
import os, sys, time, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench
if len(sys.argv) > 1 and sys.argv[1] == "pg":
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import creste_public_amd
from creste_public_amd import synth, ops
device = torch.device("cuda", 0)
creste_public_amd.set_precision("bf16x6")
model = bench.build_model(device)
B, H, W = int(os.environ.get("B", "16")), bench.IMG_H, bench.IMG_W
rgbd, p2p = synth.make_frames(B, H, W, seed=3)
rgbd, p2p = rgbd.to(device), p2p.to(device)
for parts in (0, 2):
    model.inference_parts = parts; model.inference_part_rows = 1
    with torch.no_grad():
        for _ in range(3):
            model((rgbd, p2p))
        torch.cuda.synchronize()
        issue, total = [], []
        for _ in range(8):
            t0 = time.perf_counter()
            model((rgbd, p2p))
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            issue.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
        t0 = time.perf_counter()
        for _ in range(10):
            model((rgbd, p2p))
        torch.cuda.synchronize()
        loop = (time.perf_counter() - t0) / 10 * 1e3
    print(f"{'pg' if len(sys.argv) > 1 else 'plain'} parts {parts}: host issue {sorted(issue)[len(issue) // 2]:.1f} ms, one step alone {sorted(total)[len(total) // 2]:.1f} ms, "
          f"10 back-to-back {loop:.2f} ms / step; probes (us) {ops._probe_log}")
