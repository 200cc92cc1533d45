import torch
x = torch.randn(256*1024*1024, device="cuda")   # 1 GiB
y = torch.empty_like(x)
for name, fn, bytes_ in (("copy 1GiB", lambda: y.copy_(x), 2*x.numel()*4), ("fill 1GiB", lambda: y.fill_(1.0), x.numel()*4), ("sum 1GiB", lambda: x.sum(), x.numel()*4)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/10
    print(f"{name}: {ms:.3f} ms -> {bytes_/ms/1e9:.2f} TB/s")
