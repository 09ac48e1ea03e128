"""HBM rates of plain torch kernels at the read : write mixes of the pointwise layers (rows = 131072 x 320 bf16 = 84 MB)."""
import torch
dev = torch.device("cuda:0")
x = torch.randn(131072, 320, device=dev).to(torch.bfloat16)
y1 = torch.empty_like(x)
y3 = torch.empty(131072, 960, device=dev, dtype=torch.bfloat16)
r = torch.randn(131072, 320, device=dev).to(torch.bfloat16)


def timeit(f, reps=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


MB = x.numel() * 2 / 1e6
for name, f, mb in [("fill 252 MB (write only)", lambda: y3.fill_(1.0), 3 * MB),
                    ("copy 84 -> 84 MB", lambda: y1.copy_(x), 2 * MB),
                    ("add 84 + 84 -> 84 MB", lambda: torch.add(x, r, out=y1), 3 * MB),
                    ("cat 84 -> 252 MB", lambda: torch.cat((x, x, x), 1, out=y3), 4 * MB),
                    ("sum 84 MB (read only)", lambda: x.sum(), MB)]:
    t = timeit(f)
    print(f"{name}: {t:.1f} us, {mb / t:.2f} TB/s", flush=True)
