"""Does a bandwidth-bound kernel that FITS beside the persistent F(4x4) GEMM workgroup (<= 112 VGPRs, <= 40 KiB LDS) stream
at a useful rate on the same CUs, and what does the GEMM lose?  (profiles/r05_coresidency.md)
stream A: the GEMM kernel of a 496 -> 496 @152x304 batch-16 conv call, alone, R times;  stream B: a 256-thread copy kernel
(scripts/micro/stream_probe.hip: 8 x 16-byte loads per lane, then the stores; `lds` bytes of dynamic LDS)."""
import ctypes as C, os, subprocess, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from creste_public_amd import ops
so = os.path.join(ROOT, "scripts", "micro", "libstream_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(ROOT, "scripts", "micro", "stream_probe.hip"), "-o", so])
lib = C.CDLL(so)
lib.stream_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
Cin = Cout = int(os.environ.get("CH", "496")); H, W, N = 152, 304, 16
torch.manual_seed(0)
x = ops.Act(torch.relu(torch.randn(N, H, W, Cin, device="cuda")), Cin)
w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5
pc = ops.pack_conv(w, None, None, 1, 1, ops.ACT_RELU, ops.PREC_BF16X6, algo=ops.ALGO_WINOGRAD4)
out = ops.Act.empty(N, H, W, Cout, "cuda")
# the conv's workspace (V, M) must hold REAL data for the GEMM-only launches (all-zero operands draw less power and clock
# higher): ops.conv2d's workspace allocation is pinned to one buffer
_work, _real_empty = {}, torch.empty


def _empty(*a, **k):
    if k.get("dtype") == torch.uint8 and len(a) == 1 and isinstance(a[0], int) and a[0] > (1 << 28):
        if a[0] not in _work:
            _work[a[0]] = _real_empty(*a, **k)
        return _work[a[0]]
    return _real_empty(*a, **k)


torch.empty = _empty
ops.conv2d(x, pc, out=out); torch.cuda.synchronize()          # fills V / M of the (recycled) workspace block
GB = 1 << 30
src = torch.randn(GB // 4, device="cuda"); dst = torch.empty_like(src)
dev = torch.device("cuda", 0)
main = torch.cuda.current_stream()
side = ops.concurrent_stream(dev, "parts")
print("side stream probed:", side is not None, ops._probe_log)
side = side or torch.cuda.Stream()


def gemms(R):
    os.environ["CRESTE_W4_ONLY"] = os.environ.get("ONLY", "2")
    for _ in range(R):
        ops.conv2d(x, pc, out=out)
    os.environ["CRESTE_W4_ONLY"] = "7"


def copies(R, lds, unroll, stream):
    for _ in range(R):
        rc = lib.stream_probe(src.data_ptr(), dst.data_ptr(), GB, lds, unroll, stream.cuda_stream)
        assert rc == 0, rc


def ev():
    return torch.cuda.Event(enable_timing=True)


R = 8
gemms(2); torch.cuda.synchronize()
a0, a1 = ev(), ev()
a0.record(); gemms(R); a1.record(); torch.cuda.synchronize()
tA = a0.elapsed_time(a1) / R
print(f"GEMM alone: {tA:.3f} ms per launch (ONLY={os.environ.get('ONLY', '2')})")
for lds, unroll in ((0, 8), (36864, 8), (36864, 4)):
    copies(2, lds, unroll, side); torch.cuda.synchronize()
    b0, b1 = ev(), ev()
    b0.record(side); copies(20, lds, unroll, side); b1.record(side); torch.cuda.synchronize()
    tB = b0.elapsed_time(b1) / 20
    print(f"copy alone (lds {lds}, unroll {unroll}): {tB * 1e3:.1f} us per GiB -> {2 * GB / tB / 1e9:.2f} TB/s (read + write)")
    # both: copies for ~2x the GEMMs' time, an event after every copy
    nB = int(2.2 * R * tA / tB) + 2
    torch.cuda.synchronize()
    start = ev(); start.record(main); side.wait_event(start)
    marks = []
    a0, a1 = ev(), ev()
    a0.record(main); gemms(R); a1.record(main)
    for _ in range(nB):
        copies(1, lds, unroll, side)
        e = ev(); e.record(side); marks.append(e)
    torch.cuda.synchronize()
    tA2 = a0.elapsed_time(a1) / R
    endA = start.elapsed_time(a1)
    done = [start.elapsed_time(e) for e in marks]
    k = sum(1 for t in done if t <= endA)
    rate = 2 * GB * k / (done[k - 1] if k else 1) / 1e9 if k else 0.0
    after = (nB - k) * 2 * GB / max(done[-1] - (done[k - 1] if k else 0), 1e-6) / 1e9
    print(f"  beside each other: GEMM {tA2:.3f} ms per launch ({tA2 / tA:.3f} x), copy {rate:.2f} TB/s while the GEMMs ran "
          f"({k} of {nB} copies), {after:.2f} TB/s after")
