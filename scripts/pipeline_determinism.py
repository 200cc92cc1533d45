"""Are back-to-back pipelined steps (no host synchronisation in between) reproducible?  Per output key."""
import os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench
import creste_public_amd
from creste_public_amd import synth
from creste_public_amd.creste.utils.projection import lidar_depth_images
B, H, W = 16, bench.IMG_H, bench.IMG_W
device = torch.device("cuda", 0)
if os.environ.get("PG"):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=device)
creste_public_amd.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16x6")
model = bench.build_model(device)
gen = torch.Generator().manual_seed(1337)
rgbd = torch.zeros(B, 1, 4, H, W, device=device)
rgbd[:, 0, :3] = torch.rand(B, 3, H, W, generator=gen).to(device)
scan = synth.lidar_scan(B, gen).to(device)
l2c = synth.lidar2camrect(B, H, W).to(device)
p2p = synth.make_p2p(B, H, W).to(device)


def step():
    with torch.no_grad():
        lidar_depth_images(scan, l2c, H, W, out=rgbd[:, 0, 3], scale=1000.0, depth_priority="max")
        return model((rgbd, p2p))


for parts in (0, 2):
    model.inference_parts = parts
    step(); torch.cuda.synchronize()
    ref = {k: v.clone() for k, v in step().items()}
    torch.cuda.synchronize()
    KEYS = ("depth_preds_feats", "depth_preds_logits", "bev_features", "elevation_features", "traversability_preds")
    flags = []
    for _ in range(int(os.environ.get('STEPS', '6'))):          # no synchronisation between the steps, nothing kept but flags
        o = step()
        flags.append(torch.stack([(o[k] != ref[k]).reshape(B, -1).any(1) for k in KEYS]))     # [keys, B] bools, async
        del o
    torch.cuda.synchronize()
    bad = {}
    for i, f in enumerate(flags):
        f = f.cpu()
        for j, k in enumerate(KEYS):
            if f[j].any():
                bad.setdefault(k, []).append((i, torch.nonzero(f[j]).flatten().tolist()))
    print(f"parts {parts}: {'reproducible' if not bad else 'DIFFERS'}")
    for k, v in bad.items():
        print("   ", k, len(v), "steps differ;", v[:2])

if os.environ.get("HOSTFED"):
    # bench.py's host-fed loop: every batch copied from pinned host memory on a copy stream into one of two device buffers
    # while the previous batch computes
    model.inference_parts = 2
    ref = {k: v.clone() for k, v in step().items()}
    h_rgbd, h_scan = rgbd.cpu().pin_memory(), scan.cpu().pin_memory()
    bufs = [(torch.empty_like(rgbd), torch.empty_like(scan)) for _ in range(2)]
    cstream, cur = torch.cuda.Stream(device=device), torch.cuda.current_stream()
    ready = [torch.cuda.Event() for _ in range(2)]
    free = [torch.cuda.Event() for _ in range(2)]

    def step2(rg, sc):
        with torch.no_grad():
            lidar_depth_images(sc, l2c, H, W, out=rg[:, 0, 3], scale=1000.0, depth_priority="max")
            return model((rg, p2p))

    def enqueue_copy(k):
        with torch.cuda.stream(cstream):
            cstream.wait_event(free[k])
            bufs[k][0].copy_(h_rgbd, non_blocking=True)
            bufs[k][1].copy_(h_scan, non_blocking=True)
            ready[k].record(cstream)
    for k in range(2):
        free[k].record(cur)
    n = int(os.environ.get("STEPS", "6"))
    flags = []
    enqueue_copy(0)
    for i in range(n):
        k = i & 1
        if i + 1 < n:
            enqueue_copy(k ^ 1)
        cur.wait_event(ready[k])
        o = step2(*bufs[k])
        free[k].record(cur)
        flags.append(torch.stack([(o[kk] != ref[kk]).reshape(B, -1).any(1) for kk in KEYS]))
        del o
    torch.cuda.synchronize()
    bad = {}
    for i, f in enumerate(flags):
        f = f.cpu()
        for j, kk in enumerate(KEYS):
            if f[j].any():
                bad.setdefault(kk, []).append((i, torch.nonzero(f[j]).flatten().tolist()))
    print(f"host-fed, parts 2: {'reproducible' if not bad else 'DIFFERS'}")
    for kk, v in bad.items():
        print("   ", kk, len(v), "steps differ;", v[:4])
