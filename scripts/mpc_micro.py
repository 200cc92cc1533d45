"""fused multi-positive contrastive loss (csrc/losses.hip) at SSC-batch size: forward + backward time."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd.loss_ops import MultiPosConFn
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
D = int(sys.argv[2]) if len(sys.argv) > 2 else 32
g = torch.Generator().manual_seed(0)
f = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=1).cuda().requires_grad_(True)
lab = torch.randint(0, 40, (N,), generator=g).cuda()
w = torch.rand(N, generator=g).cuda()
def run():
    a = f.detach().clone().requires_grad_(True)
    loss = MultiPosConFn.apply(f, a, lab, lab, w, 0, 0.07)
    loss.backward()
    return loss
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): l = run()
e1.record(); torch.cuda.synchronize()
print(f"N = M = {N}, D = {D}: forward + backward {e0.elapsed_time(e1) / 5:.3f} ms, loss {float(l):.6f}, |grad| {float(f.grad.abs().sum()):.6e}")
