"""N steps of harness.IRLTrainer.training_step for one of bench.py's IRL variants (GPU box; for rocprofv3 step tables).
usage: irl_step.py reference|mdp256|cf512 [prefetch 0|1] [steps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench, creste_public_amd
variant = sys.argv[1] if len(sys.argv) > 1 else "reference"
prefetch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda", 0)
creste_public_amd.set_precision("bf16x6")
infer = bench.build_model(dev)
v = bench.IRL_VARIANTS[variant]
if v["prec"]:
    creste_public_amd.set_precision(v["prec"])
step, tr = bench._irl_setup(infer, dev, variant)
def serial():
    # the same trainer step without the look-ahead batch (the reference's order)
    b = [c.cell_contents for c in step.__closure__ if isinstance(c.cell_contents, dict)][0]
    return tr.training_step(b, None)
fn = step if prefetch else serial
for _ in range(3):
    fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    fn()
torch.cuda.synchronize()
print(f"{variant} prefetch={prefetch}: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms / step (operands {creste_public_amd.get_precision()})")
