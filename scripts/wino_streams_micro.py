"""Do two independent F(4x4,3x3) convs overlap when issued on two streams?  (the BEV heads are independent branches)
usage: wino_streams_micro.py Cin Cout H W [N] [nconv]"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops
Cin, Cout, H, W = map(int, sys.argv[1:5])
N = int(sys.argv[5]) if len(sys.argv) > 5 else 16
K = int(sys.argv[6]) if len(sys.argv) > 6 else 3
torch.manual_seed(0)
xs = [ops.Act(torch.relu(torch.randn(N, H, W, Cin, device="cuda")), Cin) for _ in range(K)]
ws = [torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5 for _ in range(K)]
pcs = [ops.pack_conv(w, None, None, 1, 1, ops.ACT_RELU, ops.PREC_BF16X6, algo=ops.ALGO_WINOGRAD4) for w in ws]
outs = [ops.Act.empty(N, H, W, Cout, "cuda") for _ in range(K)]
streams = [torch.cuda.Stream() for _ in range(K)]
def serial():
    for x, pc, o in zip(xs, pcs, outs):
        ops.conv2d(x, pc, out=o)
def forked():
    main = torch.cuda.current_stream()
    for s, x, pc, o in zip(streams, xs, pcs, outs):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            ops.conv2d(x, pc, out=o)
    for s in streams:
        main.wait_stream(s)
for name, fn in (("serial", serial), ("one stream per conv", forked)):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{K} x ({Cin}->{Cout} @{H}x{W} N={N}) {name}: {e0.elapsed_time(e1) / 10:.3f} ms")
