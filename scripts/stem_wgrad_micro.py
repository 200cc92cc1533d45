"""Weight gradient of the 4-channel RGB-D stem (3x3/2, creste_conv_wgrad_strided_f32) at the distillation batch (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import _lib
lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
N, H, W, Cin, Cout, K = 8, 608, 1216, 4, 32, 3
Ho, Wo = H // 2, W // 2
x = torch.randn(N, H, W, Cin, device="cuda"); gy = torch.randn(N, Ho, Wo, Cout, device="cuda")
gw = torch.empty(Cout, Cin, K, K, device="cuda")
work = torch.empty(lib.creste_conv_wgrad_strided_workspace_bytes(N, Ho, Wo, Cin, Cout, K), dtype=torch.uint8, device="cuda")
def run(n):
    _lib.check(lib.creste_conv_wgrad_strided_f32(x.data_ptr(), Cin, gy.data_ptr(), Cout, gw.data_ptr(), n, H, W, Ho, Wo, Cin, Cout,
                                                 K, 2, 0, 0, 0, work.data_ptr(), s), "wgrad")
run(1); torch.cuda.synchronize()
ref = torch.nn.grad.conv2d_weight(torch.nn.functional.pad(x[:1].permute(0, 3, 1, 2).double(), (0, 1, 0, 1)), gw.shape,
                                  gy[:1].permute(0, 3, 1, 2).double(), stride=2)
err = float((gw.double() - ref).abs().max() / ref.abs().max())
run(N); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run(N)
e1.record(); torch.cuda.synchronize()
print(f"stem wgrad {Cin}->{Cout} k{K}/2 {H}x{W} N={N}: {e0.elapsed_time(e1) * 100:.0f} us  (N=1 rel err {err:.1e})")
