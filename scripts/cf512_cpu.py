import os, sys, torch, cProfile, pstats, io
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench, creste_public_amd
dev = torch.device("cuda")
creste_public_amd.set_precision("f16x3")
model = bench.build_model(dev)
pr = cProfile.Profile()
pr.enable()
r = bench.irl_step_bench(model, dev, "cf512", steps=3)
pr.disable()
print(r)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
