import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench
from creste_public_amd import synth
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
rgbd, p2p = synth.make_frames(16, bench.IMG_H, bench.IMG_W, seed=1337)
rgbd, p2p = rgbd.to(dev), p2p.to(dev)
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad(): out = model((rgbd, p2p))
    t1 = time.perf_counter()          # host-side issue time
    torch.cuda.synchronize(); t2 = time.perf_counter()
    st = torch.cuda.memory_stats()
    print(f"step {i}: host issue {1e3*(t1-t0):7.1f} ms, total {1e3*(t2-t0):7.1f} ms, reserved {st['reserved_bytes.all.current']/2**30:.2f} GiB, "
          f"allocated peak {st['allocated_bytes.all.peak']/2**30:.2f} GiB, segments {st['segment.all.current']}, cudaMalloc calls {st['num_device_alloc']}, frees {st['num_device_free']}")
    del out
