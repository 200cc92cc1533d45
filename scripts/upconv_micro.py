"""DeconvHead.up2 (Upsample -> conv3x3 256 -> 128 at 256 x 256, batch 16): phase form against the conv over the upsampled map, and
the ring-fix kernel alone (GPU box)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops, _lib
dev = torch.device("cuda")
N, H, W, Cin, Cout = 16, 128, 128, 256, 128
x = ops.Act(torch.randn(N, H, W, Cin, device=dev), Cin, 0)
w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5
pu = ops.pack_upconv2x(w, None, None, ops.ACT_RELU, ops.PREC_BF16X6)
pc = ops.pack_conv(w, None, None, 1, 1, ops.ACT_RELU, ops.PREC_BF16X6, algo=ops.ALGO_WINOGRAD4)
out = ops.Act.empty(N, 2 * H, 2 * W, Cout, dev)
lib = _lib.load()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"phase form        {t(lambda: ops.upconv2x(x, pu, out=out)):8.1f} us")
print(f"upsampled-map conv {t(lambda: ops.conv2d(ops.upsample_concat_lazy(x, None, 2 * H, 2 * W, 0.5, 0.5), pc, out=out)):8.1f} us")
ring = lambda: _lib.check(lib.creste_upconv2x_ring_fix_f32(x.ptr, x.cs, N, H, W, Cin, pu.w_ring.data_ptr(), Cout, 1, out.buf.data_ptr(), out.cs, out.co,
                                                           torch.cuda.current_stream().cuda_stream), "ring")
print(f"ring fix alone    {t(ring):8.1f} us")
