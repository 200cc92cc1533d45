"""Cycle stamps of the persistent value-iteration solver (library built with -DVI_TRACE; CRESTE_VI_DBG = ablation bits)."""
import sys, os, ctypes as C, torch, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops, _lib
B, H, W = map(int, sys.argv[1:4])
torch.manual_seed(0)
r = torch.rand(B, H, W, device="cuda")
for _ in range(2):
    v, q, pi, sw = ops.value_iteration(r, 0.99, 1e-3, max_sweeps=100000)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    v, q, pi, sw = ops.value_iteration(r, 0.99, 1e-3, max_sweeps=100000)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 3 * 1e3
lib = C.CDLL(os.environ["CRESTE_HIP_LIB"])
out = (C.c_longlong * 8)()
assert lib.creste_vi_trace_read(out) == 0
o = out[0:4]
print(f"dbg={os.environ.get('CRESTE_VI_DBG', '0'):>2s} B={B} {H}x{W}: {ms:.3f} ms sweeps {int(sw.item())}; wg0 chunks {o[0]}, per chunk: sweeps+store {o[1]} rendezvous {o[2]} halo {o[3]} cycles; wg100 {out[5]} / {out[6]} / {out[7]}")
