"""B=1 inference latency, eager launches vs one hipGraph replay (GPU box)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench, creste_public_amd
from creste_public_amd import MaxEntIRL, maxent_irl_cfg, synth, ops
creste_public_amd.set_precision(os.environ.get("PREC", "f16x3"))
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(0)
model = MaxEntIRL(maxent_irl_cfg((bench.IMG_H, bench.IMG_W), solve_mdp=False))
synth.randomize_bn(model, seed=1)
model = model.to(dev).eval()
rgbd, p2p = synth.make_frames(B, bench.IMG_H, bench.IMG_W, seed=1)
rgbd, p2p = rgbd.to(dev), p2p.to(dev)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
with torch.no_grad():
    for _ in range(3):
        out = model((rgbd, p2p))
    t0 = T()
    for _ in range(20):
        out = model((rgbd, p2p))
    t1 = T()
    ref = out["traversability_preds"].clone()
    print(f"B={B} eager: {(t1 - t0) / 20 * 1e3:.2f} ms/frame-batch")
    ops._AmaxPool._blocks.clear()
    g = torch.cuda.CUDAGraph()
    s_rgbd, s_p2p = rgbd.clone(), p2p.clone()
    with torch.cuda.graph(g):
        gout = model((s_rgbd, s_p2p))
    for _ in range(3):
        g.replay()
    t0 = T()
    for _ in range(50):
        g.replay()
    t1 = T()
    print(f"B={B} hipGraph replay: {(t1 - t0) / 50 * 1e3:.2f} ms/frame-batch; max |diff| vs eager "
          f"{float((gout['traversability_preds'] - ref).abs().max()):.3e} (costmap max {float(ref.abs().max()):.3e})")
