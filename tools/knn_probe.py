"""Probe: item-item kNN similarity + top-K (rk_score_topk_wide_f32) at the Elec item count for the
text (384) and image (4096) feature widths, against the library GEMM + row top-K it replaced."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommend_amd import ops

I, K = int(os.environ.get("KNN_ITEMS", 63001)), 10
for F in (384, 4096):
    x = torch.nn.functional.normalize(torch.randn(I, F, device="cuda", generator=torch.Generator("cuda").manual_seed(F)))
    ops.score_topk_wide(x[:512], x, K)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    v, i = ops.score_topk_wide(x, x, K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"F={F}: fused {dt * 1e3:.1f} ms = {2.0 * I * I * F / dt / 1e12:.1f} TFLOP/s", flush=True)
    t0 = time.perf_counter()
    for b in range(0, I, 8192):
        s = x[b:b + 8192] @ x.T
        v2, i2 = ops.rows_topk(s, K)
    torch.cuda.synchronize()
    print(f"        library GEMM + rows_topk in blocks of 8192 rows: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
    # (random rows: a few near-ties may rank differently under the library's accumulation order)
    print("        rows of the last block whose ids differ:", int((i[b:] != i2).any(1).sum()), "of", i2.shape[0], flush=True)
