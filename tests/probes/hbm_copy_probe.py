import torch, time
x = torch.empty(1<<30, device="cuda", dtype=torch.float32)  # 4 GB
y = torch.empty_like(x)
x.normal_()
for name, fn in [("copy", lambda: y.copy_(x)), ("silu", lambda: torch.nn.functional.silu(x, inplace=False)), ("read(sum)", lambda: x.sum()), ("fill", lambda: y.fill_(1.0))]:
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/5
    nbytes = {"copy": 8, "silu": 8, "read(sum)": 4, "fill": 4}[name] * x.numel()
    print(f"{name:10s} {ms:7.3f} ms  {nbytes/ms/1e9:6.2f} TB/s")
