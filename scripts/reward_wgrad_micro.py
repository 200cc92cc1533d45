"""Reward-network weight gradients (creste_conv_wgrad_f32, csrc/train.hip) at the MDP grids of the IRL variants (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import _lib
lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
for (Cin, Cout, K, H, W, N) in [(40, 64, 5, 256, 256, 8), (64, 32, 3, 256, 256, 8), (32, 32, 3, 256, 256, 8),
                                 (32, 32, 3, 128, 128, 8), (40, 64, 5, 128, 256, 8), (40, 64, 5, 64, 128, 8),
                                 (64, 32, 3, 64, 128, 8), (32, 32, 1, 256, 256, 8), (32, 16, 1, 256, 256, 8), (48, 1, 1, 256, 256, 8),
                                 (32, 32, 1, 64, 64, 8)]:
    x = torch.randn(N, H, W, Cin, device="cuda"); gy = torch.randn(N, H, W, Cout, device="cuda")
    gw = torch.empty(Cout, Cin, K, K, device="cuda")
    work = torch.empty(lib.creste_conv_wgrad_workspace_bytes(N, H, W, Cin, Cout, K), dtype=torch.uint8, device="cuda")
    fn = lambda: _lib.check(lib.creste_conv_wgrad_f32(x.data_ptr(), Cin, gy.data_ptr(), Cout, gw.data_ptr(), N, H, W, Cin,
                                                      Cout, K, K // 2, 0, work.data_ptr(), s), "wgrad")
    fn(); torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), gw.shape, gy.permute(0, 3, 1, 2).double(), padding=K // 2) \
        if H * W <= 128 * 128 else None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * N * H * W * Cin * Cout * K * K
    err = "" if ref is None else f"  rel err {float((gw.double() - ref).abs().max() / ref.abs().max()):.1e}"
    print(f"wgrad f32 {Cin}->{Cout} k{K} {H}x{W} N={N}: {ms * 1e3:.0f} us  {fl / ms / 1e9:.1f} TFLOP/s{err}", flush=True)
