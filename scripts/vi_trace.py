"""Cycle stamps of the persistent value-iteration solvers (library built with -DVI_TRACE: scripts/vi_trace.sh).
CRESTE_VI_SYNC=1: the rendezvous kernel {chunks, sweeps + store, rendezvous, halo}; default: the barrier-free kernel
{sweeps, publish, halo polling, verdict} cycles per chunk, workgroups 0 and 100."""
import sys, os, ctypes as C, torch, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import ops, _lib
B, H, W = map(int, sys.argv[1:4])
torch.manual_seed(0)
r = torch.rand(B, H, W, device="cuda")
for _ in range(2):
    v, q, pi, sw = ops.value_iteration(r, 0.99, 1e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    v, q, pi, sw = ops.value_iteration(r, 0.99, 1e-3)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 3 * 1e3
lib = C.CDLL(os.environ["CRESTE_HIP_LIB"])
out = (C.c_longlong * 8)()
assert lib.creste_vi_trace_read(out) == 0
print(f"sync={os.environ.get('CRESTE_VI_SYNC', '0')} B={B} {H}x{W}: {ms:.3f} ms sweeps {int(sw.item())}; wg0 {list(out[0:4])} wg100 {list(out[4:8])}")
