"""IRL training step: time until IRLTrainer.training_step() returns (host issue) against the steady-state step time.
usage: irl_host_issue.py reference|mdp256|cf512"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench, creste_public_amd
variant = sys.argv[1] if len(sys.argv) > 1 else "reference"
creste_public_amd.set_precision("bf16x6")
dev = torch.device("cuda", 0)
infer = bench.build_model(dev)
v = bench.IRL_VARIANTS[variant]
if v["prec"]: creste_public_amd.set_precision(v["prec"])
step, tr = bench._irl_setup(infer, dev, variant)
for _ in range(4): step()
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
med = lambda a: sorted(a)[len(a) // 2]
print(f"{variant}: host returns after {med(enq):.1f} ms, step done after {med(tot):.1f} ms (synchronised per step)")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"{variant}: 20 steps back to back: host done after {(t1 - t0) * 50:.1f} ms / step, device after {(t2 - t0) * 50:.1f} ms / step")
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{variant}: repeat {rep}: 10 steps back to back: host {(t1 - t0) * 100:.1f} ms / step, device {(t2 - t0) * 100:.1f} ms / step, vi_retries {tr.vi_retries}")
