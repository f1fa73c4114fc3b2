import torch, time
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
T, N, H = 3, 3334, 128
a = torch.randn(T, N, H, device=dev); w = torch.randn(T, H, 3 * H, device=dev); b = torch.randn(T, 1, 3 * H, device=dev)
out = torch.empty(T, N, 3 * H, device=dev)
n2 = torch.randn(T * N, H, device=dev); w1 = torch.randn(T * H, H, device=dev); b1 = torch.randn(T * H, device=dev)
def t_baddbmm(): torch.baddbmm(b, a, w, out=out)
def t_bmm(): torch.bmm(a, w, out=out)
def t_addmm3():
    for t in range(T): torch.addmm(b[t, 0], a[t], w[t], out=out[t])
def t_addmm_first(): return torch.addmm(b1, n2, w1.t())
def t_mm_first(): return torch.mm(n2, w1.t())
for name, fn in [("baddbmm", t_baddbmm), ("bmm", t_bmm), ("addmm x3", t_addmm3), ("addmm first", t_addmm_first), ("mm first", t_mm_first)]:
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): fn()
    e.record(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn(); torch.cuda.synchronize()
    ks = [(ev.key[:60], round(ev.device_time_total, 1)) for ev in prof.key_averages() if ev.device_time_total > 0]
    print("%-12s %.1f us   %s" % (name, s.elapsed_time(e) * 20, ks))
