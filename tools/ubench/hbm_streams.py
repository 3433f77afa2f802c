import torch, time
x = torch.empty(1 << 28, dtype=torch.float32, device="cuda")   # 1 GiB
y = torch.empty_like(x)
def bw(f, bytes_, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return bytes_ * n / (time.perf_counter() - t0) / 1e12
print("fill_ (pure write) 1 GiB: %.2f TB/s" % bw(lambda: x.fill_(1.0), x.numel() * 4))
print("zero_              1 GiB: %.2f TB/s" % bw(lambda: x.zero_(), x.numel() * 4))
print("copy_ (read+write) 1 GiB: %.2f TB/s (bytes moved)" % bw(lambda: y.copy_(x), 2 * x.numel() * 4))
print("sum (pure read)    1 GiB: %.2f TB/s" % bw(lambda: x.sum(), x.numel() * 4))
