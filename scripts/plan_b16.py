"""batch-16 inference: eager Python launches vs the C plan runtime (replayed launches / hipGraph)"""
import os, sys, time, torch, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench, creste_public_amd
from creste_public_amd import synth, deploy
creste_public_amd.set_precision("f16x3")
dev = torch.device("cuda")
model = bench.build_model(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rgbd, p2p = synth.make_frames(B, bench.IMG_H, bench.IMG_W, seed=1)
rgbd, p2p = rgbd.to(dev), p2p.to(dev)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
with torch.no_grad():
    for _ in range(3): model((rgbd, p2p))
    t0 = T()
    for _ in range(10): model((rgbd, p2p))
    t1 = T()
print(f"B={B} eager: {(t1 - t0) / 10 * 1e3:.2f} ms")
path = os.path.join(tempfile.mkdtemp(), "plan.bin")
info = deploy.export_plan(model, (rgbd, p2p), path)
for graph in (False, True):
    pm = deploy.PlanModel(path, graph=graph)
    for _ in range(3): pm.run((rgbd, p2p))
    t0 = T()
    for _ in range(10): pm.run((rgbd, p2p))
    t1 = T()
    print(f"B={B} plan runtime{' + hipGraph' if graph else ''}: {(t1 - t0) / 10 * 1e3:.2f} ms")
