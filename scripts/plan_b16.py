"""batch-16 inference (and batch 1): the Python path vs the C plan runtime replaying the SAME forward -- pipelined plan
(two streams inside the runtime), one-stream plan, each as replayed launches and as one hipGraph; outputs compared bit for bit.
usage: plan_b16.py [B] [precision]      (profiles/r05_plan_runtime.md)"""
import os, sys, time, torch, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench, creste_public_amd
from creste_public_amd import synth, deploy
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
creste_public_amd.set_precision(sys.argv[2] if len(sys.argv) > 2 else "bf16x6")
dev = torch.device("cuda")
model = bench.build_model(dev)
rgbd, p2p = synth.make_frames(B, bench.IMG_H, bench.IMG_W, seed=1)
rgbd, p2p = rgbd.to(dev), p2p.to(dev)
N = 20


def T():
    torch.cuda.synchronize(); return time.perf_counter()


def timed(fn):
    for _ in range(3): fn()
    t0 = T()
    for _ in range(N): fn()
    return (T() - t0) / N * 1e3


with torch.no_grad():
    parts = model._parts_for(B)
    ms = timed(lambda: model((rgbd, p2p)))
    ref = {k: v.clone() for k, v in model((rgbd, p2p)).items()}
    print(f"B={B} Python path ({parts} part{'s' if parts > 1 else ''}): {ms:.3f} ms")
    if parts > 1:
        model.inference_parts = 0
        ms1 = timed(lambda: model((rgbd, p2p)))
        ref1 = {k: v.clone() for k, v in model((rgbd, p2p)).items()}
        del model.inference_parts
        print(f"B={B} Python path (one stream): {ms1:.3f} ms")
d = tempfile.mkdtemp()
for pipelined in ((True, False) if parts > 1 else (False,)):
    path = os.path.join(d, f"plan_{int(pipelined)}.bin")
    info = deploy.export_plan(model, (rgbd, p2p), path, pipelined=pipelined)
    want = ref if (pipelined or parts == 1) else ref1
    for graph in (False, True):
        pm = deploy.PlanModel(path, graph=graph)
        ms = timed(lambda: pm.run((rgbd, p2p)))
        got = pm((rgbd, p2p))
        same = all(torch.equal(got[k], want[k].cpu().to(got[k].dtype)) for k in want)
        print(f"B={B} plan runtime, {pm.num_streams} stream(s){' + hipGraph' if graph else ''}: {ms:.3f} ms  "
              f"({info['calls']} launches, {info['events']} edges)  == Python path: {same}")
        pm.close()
