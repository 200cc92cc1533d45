"""Micro-benchmark of the conv weight-gradient kernels (GPU box). usage: wgrad_micro.py Cin Cout K H W N"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import _lib, ops
lib = _lib.load()
Cin, Cout, K, H, W, N = (int(v) for v in sys.argv[1:7]) if len(sys.argv) > 6 else (496, 496, 3, 152, 304, 8)
torch.manual_seed(0)
x = torch.randn(N, H, W, Cin, device="cuda"); gy = torch.randn(N, H, W, Cout, device="cuda") * 1e-3
gw = torch.empty(Cout, Cin, K, K, device="cuda")
work = torch.empty(lib.creste_conv_wgrad_strided_workspace_bytes(N, H, W, Cin, Cout, K), dtype=torch.uint8, device="cuda")
xa, ga = ops.absmax(ops.Act(x, Cin)), ops.absmax(ops.Act(gy, Cout))
s = torch.cuda.current_stream().cuda_stream
def f16(): _lib.check(lib.creste_conv_wgrad_f16x3(x.data_ptr(), Cin, gy.data_ptr(), Cout, gw.data_ptr(), xa.data_ptr(), ga.data_ptr(), N, H, W, H, W, Cin, Cout, K, 1, K // 2, K // 2, 0, work.data_ptr(), s), "f16")
def f32(): _lib.check(lib.creste_conv_wgrad_strided_f32(x.data_ptr(), Cin, gy.data_ptr(), Cout, gw.data_ptr(), N, H, W, H, W, Cin, Cout, K, 1, K // 2, K // 2, 0, work.data_ptr(), s), "f32")
fl = 2.0 * N * H * W * Cin * Cout * K * K
for name, fn in (("f16x3", f16), ("f32 tiled", f32)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"wgrad {name:10s} {Cin}->{Cout} k{K} {H}x{W} N={N}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s")
