"""Probe: the modality projection forward / backward at Elec item count (run under rocprofv3
--kernel-trace --stats for per-kernel times; prints wall times per call otherwise)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommend_amd import ops

I, U = 63001, 192403
dev = "cuda"
for F in ([int(os.environ["PROJ_F"])] if "PROJ_F" in os.environ else (4096, 384)):
    x = torch.randn(I, F, device=dev)
    w = (torch.randn(64, F, device=dev) / F ** 0.5).requires_grad_(True)
    b = torch.randn(64, device=dev, requires_grad=True)
    g = torch.randn(U + I, 64, device=dev)
    for _ in range(3):
        y = ops.proj(x, w, b, U)
        torch.autograd.grad(y, [w, b], g)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        y = ops.proj(x, w, b, U)
    e.record(); e.synchronize()
    tf = s.elapsed_time(e) / 20 * 1e3
    s.record()
    for _ in range(20):
        torch.autograd.grad(ops.proj(x, w, b, U), [w, b], g)
    e.record(); e.synchronize()
    tb = s.elapsed_time(e) / 20 * 1e3 - tf
    gb = I * F * 4 / 1e9
    print(f"F={F}: fwd {tf:.1f} us ({gb / tf * 1e3:.2f} TB/s of X), bwd {tb:.1f} us ({gb / tb * 1e3:.2f} TB/s)")
    y64 = (x.double() @ w.detach().double().T + b.detach().double())
    err = float((y[U:].double() - y64).abs().max() / y64.abs().max())
    print(f"   fwd max rel err vs fp64: {err:.2e}")
