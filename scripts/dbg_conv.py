import torch, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch.nn.functional as F
from creste_public_amd import ops
torch.manual_seed(0)
for (N, Cin, H, W, Cout, K) in [(3, 64, 20, 36, 40, 5), (3, 64, 20, 36, 64, 5), (2, 64, 32, 48, 40, 5), (3, 64, 20, 36, 40, 3), (3, 32, 20, 36, 40, 5), (1, 64, 20, 36, 40, 5)]:
    x = torch.randn(N, Cin, H, W); w = torch.randn(Cout, Cin, K, K) / (Cin * K * K) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=K // 2)
    pc = ops.pack_conv(w.cuda(), None, None, 1, K // 2, 0, ops.PREC_F32)
    y = ops.conv2d(ops.nchw_to_nhwc(x.cuda()), pc).nchw().cpu().double()
    e = (y - ref).abs()
    print((N, Cin, H, W, Cout, K), "rel rms", float((y - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()), "max", float(e.max()),
          "bad px", int((e.amax(dim=1) > 1e-3).sum()), "of", N * H * W, "first bad", (e.amax(dim=1) > 1e-3).nonzero()[:3].tolist())
