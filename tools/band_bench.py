"""Times of the band product kernels (csrc/band_product.hip) at the training bench's shape: 399 chunks x 1024 rows, width 384."""
import sys
sys.path.insert(0, '.')
import torch
from hermnet_amd import _lib
from hermnet_amd.ops import _stream
dev = torch.device('cuda:0')
nc, C, N = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (399, 1024, 384)))
g = torch.Generator().manual_seed(0)
A = torch.randn(nc, C, 32, generator=g).to(dev); B = torch.randn(nc, 32, N, generator=g).to(dev); b = torch.randn(nc, N, generator=g).to(dev)
g1 = torch.randn(nc, C, N, device=dev); g2 = torch.randn(nc, C, N, device=dev)
out = torch.empty(nc, C, N, device=dev); gA = torch.empty(nc, C, 32, device=dev); gB = torch.empty(nc, 32, N, device=dev); gb = torch.empty(nc, N, device=dev)
L, P = _lib.load(), _lib.ptr
cases = {
    "P  (product + bias)": (lambda: L.hermnet_band_product(P(A), P(B), P(b), nc, C, N, P(out), _stream()), nc * C * N * 4),
    "Q  (one addend)": (lambda: L.hermnet_band_product_grad_a(P(g1), None, P(B), nc, C, N, P(gA), _stream()), nc * C * N * 4),
    "Q  (two addends)": (lambda: L.hermnet_band_product_grad_a(P(g1), P(g2), P(B), nc, C, N, P(gA), _stream()), 2 * nc * C * N * 4),
    "S  (one addend)": (lambda: L.hermnet_band_product_grad_b(P(A), P(g1), None, nc, C, N, P(gB), P(gb), _stream()), nc * C * N * 4),
    "S  (two addends)": (lambda: L.hermnet_band_product_grad_b(P(A), P(g1), P(g2), nc, C, N, P(gB), P(gb), _stream()), 2 * nc * C * N * 4),
    "QS (one addend)": (lambda: L.hermnet_band_product_grads(P(A), P(B), P(g1), None, nc, C, N, P(gA), P(gB), P(gb), _stream()), nc * C * N * 4),
    "QS (two addends)": (lambda: L.hermnet_band_product_grads(P(A), P(B), P(g1), P(g2), nc, C, N, P(gA), P(gB), P(gb), _stream()), 2 * nc * C * N * 4),
}
thrash = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for name, (fn, nbytes) in cases.items():
    for _ in range(3): fn()
    ts = []
    for _ in range(10):
        thrash.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rc = fn(); e1.record(); torch.cuda.synchronize()
        assert rc == 0
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print("%-22s median %7.1f us  min %7.1f us   %5.2f TB/s of the [nc,C,N] side" % (name, ts[len(ts) // 2], ts[0], nbytes / ts[len(ts) // 2] / 1e6))
