import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommend_amd import ops
torch.manual_seed(0)
dev = "cuda"
for N in (255404, 70000, 53955, 262144):
    x = torch.randn(N, 64, device=dev); x2 = torch.randn(N, 64, device=dev); r = torch.randn(N, 64, device=dev)
    w = torch.randn(64, 64, device=dev) / 8; b = torch.randn(64, device=dev); a = torch.tensor([0.25], device=dev)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    for name, y, want in (
        ("plain", ops.dense(x, w, b), ref),
        ("act", ops.dense(x, w, b, alpha=a), torch.nn.functional.prelu(ref, a.double())),
        ("resid", ops.dense(x, w, b, alpha=a, residual=r), torch.nn.functional.prelu(ref, a.double()) + r.double()),
        ("dual", ops.dense(x, w, b, x2=x2, weight2=w, bias2=b), ref + torch.nn.functional.linear(x2.double(), w.double(), b.double())),
    ):
        bad = ~torch.isfinite(y)
        err = float((y.double() - want).abs().max() / want.abs().max()) if not bad.any() else float("nan")
        rows = torch.nonzero(bad.any(dim=1)).flatten()
        print(N, name, "err", err, "nonfinite rows", int(rows.numel()), rows[:6].tolist(), rows[-3:].tolist() if rows.numel() else "")
I, U, F = 63001, 192403, 4096
xi = torch.randn(I, F, device=dev); wp = torch.randn(64, F, device=dev) / 64; bp = torch.randn(64, device=dev)
y = ops.proj(xi, wp, bp, U)
want = xi.double() @ wp.double().T + bp.double()
print("proj err", float((y[U:].double() - want).abs().max() / want.abs().max()), "user rows ok", bool((y[:U] == bp).all()), "finite", bool(torch.isfinite(y).all()))
