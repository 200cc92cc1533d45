"""what would removing the 16 se_gate launches buy at most?  The gate kernels replaced by no-ops (WRONG results, timing only), batch 16
pipelined and one-stream steps (GPU box)"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench, creste_public_amd
from creste_public_amd import synth, _lib
from creste_public_amd.creste.utils.projection import lidar_depth_images
creste_public_amd.set_precision("bf16x6")
dev = torch.device("cuda")
model = bench.build_model(dev)
B, H, W = 16, bench.IMG_H, bench.IMG_W
gen = torch.Generator().manual_seed(1337)
rgbd = torch.zeros(B, 1, 4, H, W, device=dev); rgbd[:, 0, :3] = torch.rand(B, 3, H, W, generator=gen).to(dev)
scan = synth.lidar_scan(B, gen).to(dev); l2c = synth.lidar2camrect(B, H, W).to(dev); p2p = synth.make_p2p(B, H, W).to(dev)
def step():
    with torch.no_grad():
        lidar_depth_images(scan, l2c, H, W, out=rgbd[:, 0, 3], scale=1000.0, depth_priority="max")
        return model((rgbd, p2p))
def t(n=20):
    for _ in range(4): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
lib = _lib.load()
orig = (lib.creste_se_gate_partial_f32, lib.creste_se_gate_f32)
for rep in range(2):
    for parts in (2, 0):
        model.inference_parts = parts
        a = t()
        lib.creste_se_gate_partial_f32 = lambda *a: 0
        lib.creste_se_gate_f32 = lambda *a: 0
        b = t()
        lib.creste_se_gate_partial_f32, lib.creste_se_gate_f32 = orig
        print(f"parts={parts}: with se_gate {a:.2f} ms, without {b:.2f} ms")
