"""Is a half-batch forward bit-identical to the same rows of the whole-batch forward?  Per output key and per internal stage."""
import os, sys, time, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench
import creste_public_amd
from creste_public_amd import synth, ops
B, H, W = 16, bench.IMG_H, bench.IMG_W
device = torch.device("cuda", 0)
creste_public_amd.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16x6")
model = bench.build_model(device)
rgbd, p2p = synth.make_frames(B, H, W, seed=99)
rgbd, p2p = rgbd.to(device), p2p.to(device)
h = B // 2
with torch.no_grad():
    full = {k: v.clone() for k, v in model((rgbd, p2p)).items()}
    half = {k: v.clone() for k, v in model((rgbd[:h].contiguous(), p2p[:h].contiguous())).items()}
    rf = model.backbone.depthcomp.forward_act(ops.nchw_to_nhwc(rgbd.reshape(B, 4, H, W).contiguous()))
    rh = model.backbone.depthcomp.forward_act(ops.nchw_to_nhwc(rgbd[:h].reshape(h, 4, H, W).contiguous()))
for k in full:
    a, b = full[k][:h], half[k]
    if a.shape != b.shape:
        print(f"{k}: shapes {tuple(a.shape)} vs {tuple(b.shape)}"); continue
    same = torch.equal(a, b)
    d = (a.double() - b.double()).abs().max().item() if a.is_floating_point() else (a != b).sum().item()
    print(f"{k:40s} {tuple(full[k].shape)} {full[k].numel() * full[k].element_size() / 1e6:9.1f} MB  identical {same}  max|diff| {d:.3e}")
for k in rf:
    a, b = rf[k], rh[k]
    if hasattr(a, "buf"):
        a, b = a.buf[:h, ..., a.co:a.co + a.C], b.buf[..., b.co:b.co + b.C]
        print(f"stage {k:30s} identical {torch.equal(a, b)} max|diff| {(a - b).abs().max().item():.3e}")
    elif torch.is_tensor(a):
        print(f"stage {k:30s} identical {torch.equal(a[:h], b)}")
torch.cuda.synchronize()
outs = [full, full]
t0 = time.perf_counter()
for _ in range(10):
    c = {k: torch.cat([o[k] for o in outs]) for k in full}
torch.cuda.synchronize()
print(f"torch.cat of every output key: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
