"""Stand-alone two-kernel reproducer attempt of the packed-fp32 corruption (profiles/r05_pk_hazard.md), through the C ABI, no model:
stream 0 runs the REAL fused MBConv kernel (mbconv_expand_dw_kernel<3,2,16,16,2>: EfficientNet block 1, 16 -> 96 channels, k3 s2) on
random data, stream 1 a REAL bf16x6 1x1 conv (conv_patch_kernel<1,3,1>: 256 -> 64) on unrelated buffers; the fused kernel's output
is compared bit for bit with the same launch run alone.  Run it twice: with the shipped library and with a variant whose mbconv.hip is
compiled with packed fp32 allowed (CRESTE_HIP_LIB=.../libcreste_hip_pk.so; scripts/micro/pk_pair.sh builds it on the box).
usage: pk_pair.py [rounds]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import creste_public_amd
from creste_public_amd import ops
R = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(7)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
N, H, W, Cin, Cexp = 8, 304, 608, 16, 96
x = ops.Act(rn(N, H, W, Cin), Cin, 0)
we, be = rn(Cin, Cexp) / 4, rn(Cexp) * 0.1
wt, bd = rn(9, Cexp) / 3, rn(Cexp) * 0.1
w1, b1, w2, b2 = rn(4, Cexp) / 10, rn(4) * 0.1, rn(Cexp, 4) / 2, rn(Cexp) * 0.1
fused = lambda: ops.mbconv_expand_dw_se(x, we, be, wt, bd, 3, 2, (0, 1, 0, 1), w1, b1, w2, b2)[0].buf
y = ops.Act(rn(N, 152, 304, 256), 256, 0)
pc = ops.pack_conv(rn(64, 256, 1, 1) / 16, None, None, 1, 0, 0, ops.PREC_BF16X6)
ref = fused().clone(); torch.cuda.synchronize()
assert torch.equal(fused(), ref)
s0 = torch.cuda.current_stream()
s1 = ops.concurrent_stream(dev, "parts") or torch.cuda.Stream()
wrong, lanes = 0, []
for r in range(R):
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        for _ in range(3): ops.conv2d(y, pc)             # ~3 x 60 us of MFMA waves next to ...
    outs = [fused() for _ in range(2)]                    # ... two launches of the fused kernel (~2 x 190 us at batch 8)
    torch.cuda.synchronize()
    for o in outs:
        if not torch.equal(o, ref):
            wrong += 1
            d = (o != ref).nonzero()
            lanes.append((int(d.shape[0]), d[0].tolist(), sorted(set((d[:, 3] % 4).tolist())), float((o - ref).abs().max())))
print(f"library {os.environ.get('CRESTE_HIP_LIB', 'shipped')}: fused kernel wrong in {wrong} of {2 * R} launches beside the 1x1 conv")
for n, first, comps, mx in lanes[:6]:
    print(f"   {n} wrong floats, first at [n, y, x, c] = {first}, float4 components hit {comps}, max |diff| {mx:.3g}")
