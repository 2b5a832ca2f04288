"""Probe: time the fused node-wise layer kernels at Elec node count."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommend_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 255404
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = "cuda"
x = torch.randn(N, D, device=dev, requires_grad=True)
x2 = torch.randn(N, D, device=dev, requires_grad=True)
w = (torch.randn(D, D, device=dev) / 8).requires_grad_(True)
b = torch.randn(D, device=dev, requires_grad=True)
a = torch.tensor([0.25], device=dev, requires_grad=True)
g = torch.randn(N, D, device=dev)

def t(fn, reps=50):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps * 1e3

by = N * D * 4
print(f"fwd act        : {t(lambda: ops.dense(x, w, b, alpha=a)):.1f} us   (ideal {2*by/5e6:.1f} us at 5 TB/s)")
print(f"fwd act+drop   : {t(lambda: ops.dense(x, w, b, alpha=a, p=0.3)):.1f} us")
print(f"fwd dual       : {t(lambda: ops.dense(x, w, b, x2=x2, weight2=w, bias2=b)):.1f} us   (ideal {3*by/5e6:.1f})")
y = ops.dense(x, w, b, alpha=a, p=0.3)
r = torch.randn(N, D, device=dev)
print(f"fwd dual+res   : {t(lambda: ops.dense(x, w, b, x2=x2, weight2=w, bias2=b, residual=r)):.1f} us")
print(f"fwd+bwd act    : {t(lambda: torch.autograd.grad(ops.dense(x, w, b, alpha=a, p=0.3), [x, w, b, a], g)):.1f} us  (bwd ideal {3*by/5e6:.1f})")
print(f"fwd+bwd linear : {t(lambda: torch.autograd.grad(ops.dense(x, w, b), [x, w, b], g)):.1f} us")
print(f"copy N x D     : {t(lambda: g.clone()):.1f} us")
