"""MBConv front half (expand 1x1 -> depthwise -> SE gate) of the early EfficientNet-B0 blocks at the bench size:
the fused kernel (csrc/mbconv.hip) against the two-kernel path -- equality and time.  usage: mbconv_micro.py [precision]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import creste_public_amd
from creste_public_amd import ops
from creste_public_amd.creste.models.blocks import effnet as E
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
creste_public_amd.set_precision(prec)
dev = torch.device("cuda")
B = 16
# (k, s, cin, cout, H, W) of blocks 1..5 at 608x1216 input
BLOCKS = [(3, 2, 16, 24, 304, 608), (3, 1, 24, 24, 152, 304), (5, 2, 24, 40, 152, 304), (5, 1, 40, 40, 76, 152), (3, 2, 40, 80, 76, 152)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
torch.manual_seed(0)
tot = [0.0, 0.0]
for (k, s, cin, cout, H, W) in BLOCKS:
    pad = E._same_pad(112, k, s); pad = (*pad, *pad)
    blk = E.MBConvBlock(k, s, 6, cin, cout, pad).to(dev).eval()
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    x = ops.Act(torch.randn(B, H, W, cin, device=dev), cin, 0)
    x.amax = x.buf.abs().max().reshape(1).clone()
    outs = {}
    with torch.no_grad():
        for fused in (False, True):
            E.FUSE_MBCONV = fused
            y = blk.forward_act(x)
            outs[fused] = y.buf.clone()
            t_all = timeit(lambda: blk.forward_act(x))
            p = blk._plan
            w, b = p["dw"].get()
            if fused:
                we, be = p["expand_fused"].get()
                t_front = timeit(lambda: ops.mbconv_expand_dw_se(x, we, be, w, b, k, s, pad, *p["se"].get()))
            else:
                t_front = timeit(lambda: ops.dwconv2d_se(p["expand"](x), w, b, k, s, pad, ops.ACT_SWISH, *p["se"].get()))
            tot[fused] += t_front
            print(f"  {'fused' if fused else 'split'}: block {t_all * 1e3:7.1f} us, front half (expand+dw+gate) {t_front * 1e3:7.1f} us")
    d = (outs[True] - outs[False]).abs().max().item(); sc = outs[False].abs().max().item()
    print(f"k{k} s{s} {cin}->{6*cin}->{cout} @{H}x{W}: max|fused - split| = {d:.3e} (max|out| {sc:.3e})")
print(f"{prec}: front halves of the five blocks: split {tot[0]:.3f} ms, fused {tot[1]:.3f} ms")
# stem + block 0 (fused stem / depthwise kernel vs conv kernel + depthwise kernel)
from creste_public_amd.creste.models.blocks.effnet import EfficientNetB0Trunk
trunk = EfficientNetB0Trunk(4, (608, 1216)).to(dev).eval()
for m in trunk.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
img = ops.nchw_to_nhwc(torch.rand(B, 4, 608, 1216, device=dev))
img.amax = img.buf.abs().max().reshape(1).clone()
trunk._blocks = trunk._blocks[:1]
res = {}
with torch.no_grad():
    for fused in (False, True):
        E.FUSE_MBCONV = fused
        res[fused] = list(trunk.extract_endpoints_act(img).values())[-1].buf.clone()
        print(f"stem + block 0 {'fused' if fused else 'split'}: {timeit(lambda: trunk.extract_endpoints_act(img)) * 1e3:7.1f} us")
print(f"max|fused - split| = {(res[True] - res[False]).abs().max().item():.3e} (max|out| {res[False].abs().max().item():.3e})")
