"""Is the training step bound by the host?  Time until training_step() RETURNS (launches enqueued, nothing waited for except what the
step itself waits for) against the synchronised step time.  usage: train_host_issue.py distill|ssc [B] [precision]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench, creste_public_amd
which = sys.argv[1] if len(sys.argv) > 1 else "distill"
creste_public_amd.set_precision(sys.argv[3] if len(sys.argv) > 3 else "bf16x6")
dev = torch.device("cuda", 0)
step, _, tr = (bench._distill_setup if which == "distill" else bench._ssc_setup)(dev, int(sys.argv[2]) if len(sys.argv) > 2 else 8)
for _ in range(3): step()
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print(f"{which}: host returns after {sorted(enq)[len(enq) // 2]:.1f} ms, step done after {sorted(tot)[len(tot) // 2]:.1f} ms (medians of 8)")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
print(f"{which}: 10 steps back to back, one synchronisation: {(time.perf_counter() - t0) * 100:.1f} ms / step")
