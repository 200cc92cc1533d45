"""Full-size (608x1216, B=2) end-to-end noise table on the GPU box: for every float output key, rms distance to the
float64 oracle of (a) the fp32 CPU oracle, (b) the HIP path in f32 / bf16x6 / f16x3, plus voxel-flip counts.
    python scripts/fullsize_noise.py > gpurun_out/fullsize_noise.md"""
import copy, os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import creste_public_amd
from creste_public_amd import MaxEntIRL, synth
from creste_public_amd.config import maxent_irl_cfg
from oracle.irl import MaxEntIRL as OracleIRL
from test_model_gpu import _rms, calibrate_bn

H, W, B = 608, 1216, 2
torch.manual_seed(4321)
cfg = maxent_irl_cfg((H, W), solve_mdp=False)
oracle = OracleIRL(cfg)
rgbd, p2p = synth.make_frames(B, H, W, seed=77)
calibrate_bn(oracle, lambda: oracle((rgbd, p2p)))
with torch.no_grad():
    oracle.traversability_head.r.postpool[0].norm.weight.mul_(0.01)
    oracle.traversability_head.r.postpool[0].norm.bias.mul_(0.01)
    ref = oracle((rgbd, p2p))
    o64 = copy.deepcopy(oracle).double(); o64.fov_mask = oracle.fov_mask
    ref64 = o64((rgbd.double(), p2p.double()))
outs = {}
for mode in ("f32", "bf16x6", "f16x3"):
    creste_public_amd.set_precision(mode)
    m = MaxEntIRL(maxent_irl_cfg((H, W), solve_mdp=False))
    m.load_state_dict(oracle.state_dict(), strict=True)
    m = m.cuda().eval()
    with torch.no_grad():
        outs[mode] = {k: v.detach().cpu() for k, v in m((rgbd.cuda(), p2p.cuda())).items()}
    del m
cell64 = ref64["bev_coords"].floor()
def flips(c): return int((c.double().floor() != cell64).any(dim=-1).sum())
print(f"# full-size e2e noise, B={B} frames of {W}x{H}; {B * (H // 4) * (W // 4)} points; occupied BEV cells "
      f"{float((ref['bev_densities'] > 0).float().mean()):.3f}\n")
print(f"voxel-cell flips vs float64: cpu-fp32 {flips(ref['bev_coords'])}, " +
      ", ".join(f"hip-{k} {flips(v['bev_coords'])}" for k, v in outs.items()) + "\n")
def q999(err):
    x = err.abs().flatten()
    if x.numel() > 2_000_000:
        x = x[::max(1, x.numel() // 2_000_000)]
    return float(torch.kthvalue(x, max(1, int(0.999 * x.numel()))).values)
print("rms error vs float64:\n")
print("| key | rms(f64) | cpu fp32 | hip f32 | hip bf16x6 | hip f16x3 |\n|---|---|---|---|---|---|")
for k, t in ref64.items():
    if k.startswith("_") or not torch.is_tensor(t) or not t.is_floating_point():
        continue
    t = t.double()
    row = [f"{_rms(t):.3e}", f"{_rms(ref[k].double() - t):.3e}"] + [f"{_rms(outs[mo][k].double() - t):.3e}" for mo in outs]
    print(f"| {k} | " + " | ".join(row) + " |")
def quant(err, q):
    x = err.abs().flatten()
    if x.numel() > 2_000_000:
        x = x[::max(1, x.numel() // 2_000_000)]
    return float(torch.kthvalue(x, max(1, int(q * x.numel()))).values)
for q in (0.5, 0.9, 0.999):
    print(f"\n{q}-quantile of |error| vs float64\n\n| key | cpu fp32 | hip f32 | hip bf16x6 | hip f16x3 |\n|---|---|---|---|---|")
    for k, t in ref64.items():
        if k.startswith("_") or not torch.is_tensor(t) or not t.is_floating_point():
            continue
        t = t.double()
        row = [f"{quant(ref[k].double() - t, q):.3e}"] + [f"{quant(outs[mo][k].double() - t, q):.3e}" for mo in outs]
        print(f"| {k} | " + " | ".join(row) + " |")
