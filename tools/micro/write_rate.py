"""Write-only and read-only stream rates on this GPU (torch fill / sum over 628 MB): ceilings for the band product kernels."""
import torch
dev = torch.device('cuda:0')
x = torch.empty(399 * 1024 * 384, device=dev)
thrash = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
def t(fn, n=10):
    for _ in range(3): fn()
    ts = []
    for _ in range(n):
        thrash.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[n // 2]
nb = x.numel() * 4
a = t(lambda: x.fill_(1.0)); print("fill_  %.1f us  %.2f TB/s" % (a, nb / a / 1e6))
a = t(lambda: x.zero_()); print("zero_  %.1f us  %.2f TB/s" % (a, nb / a / 1e6))
y = torch.empty_like(x)
a = t(lambda: y.copy_(x)); print("copy_  %.1f us  %.2f TB/s (read + write)" % (a, 2 * nb / a / 1e6))
a = t(lambda: x.sum()); print("sum    %.1f us  %.2f TB/s" % (a, nb / a / 1e6))
