"""Can the bandwidth-bound transform kernels of one half-batch run under the matrix-bound GEMM of the other half?
A chain of L identical F(4x4,3x3) convs over a batch of N, (a) whole batch on one stream, (b) two half-batches on two
streams issued layer by layer (A(l), B(l), A(l+1), ...), with / without the library's GEMM chain (CRESTE_W4_CHAIN: the
GEMM kernels of all streams run in host issue order, so the halves fall into anti-phase) and with the persistent GEMM
on fewer workgroups per XCD (CRESTE_W4_GEMM_WGS: the rest of the CUs stay free for the other stream's transforms).
usage: wino_overlap_micro.py Cin Cout H W [N] [L]"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops
Cin, Cout, H, W = map(int, sys.argv[1:5])
N = int(sys.argv[5]) if len(sys.argv) > 5 else 16
L = int(sys.argv[6]) if len(sys.argv) > 6 else 4
assert Cin == Cout or L == 1
torch.manual_seed(0)
x = torch.relu(torch.randn(N, H, W, Cin, device="cuda"))
ws = [torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5 for _ in range(L)]
pcs = [ops.pack_conv(w, None, None, 1, 1, ops.ACT_RELU, ops.PREC_BF16X6, algo=ops.ALGO_WINOGRAD4) for w in ws]
h = N // 2
full = [ops.Act.empty(N, H, W, Cout, "cuda") for _ in range(L)]
halves = [[ops.Act(full[l].buf[i * h:(i + 1) * h], Cout) for l in range(L)] for i in range(2)]
xin = ops.Act(x, Cin)
xh = [ops.Act(x[i * h:(i + 1) * h], Cin) for i in range(2)]
s = [torch.cuda.Stream(), torch.cuda.Stream()]


def whole():
    a = xin
    for l in range(L):
        a = ops.conv2d(a, pcs[l], out=full[l])


def split():
    main = torch.cuda.current_stream()
    cur = list(xh)
    for st in s:
        st.wait_stream(main)
    for l in range(L):
        for i in range(2):
            with torch.cuda.stream(s[i]):
                cur[i] = ops.conv2d(cur[i], pcs[l], out=halves[i][l])
    for st in s:
        main.wait_stream(st)


def timeit(fn, reps=6):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


whole(); torch.cuda.synchronize()
ref = full[-1].buf.clone()
print(f"{L} x ({Cin}->{Cout} @{H}x{W} N={N})")
print(f"  whole batch, one stream: {timeit(whole):.3f} ms")
QUICK = os.environ.get("OVERLAP_QUICK") == "1"          # only the default variant (for a kernel trace)
for wgs in ((0,) if QUICK else (0, 28, 24, 20, 16)):
    for chain in ((0,) if QUICK else (0, 1)):
        os.environ["CRESTE_W4_GEMM_WGS"] = str(wgs)
        os.environ["CRESTE_W4_CHAIN"] = str(chain)
        t = timeit(split)
        same = torch.equal(full[-1].buf, ref)
        print(f"  halves on two streams, GEMM workgroups / XCD {wgs or 32}, chain {chain}: {t:.3f} ms  bit-identical {same}")
os.environ["CRESTE_W4_CHAIN"] = "0"
for wgs in (() if QUICK else (28, 24)):
    os.environ["CRESTE_W4_GEMM_WGS"] = str(wgs)
    print(f"  whole batch, one stream, GEMM workgroups / XCD {wgs}: {timeit(whole):.3f} ms")
