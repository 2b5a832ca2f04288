"""Sizing run for the large synthetic configuration (10M users x 2M items, 500M interactions,
emb_dim 128) on ONE GPU of eight: the row block of rank 0 under row sharding (1.5M user rows,
their ~40 training edges each into the 2M item rows, self loops), the all-gathered layer it reads
(12M x 128 fp32), the 128-wide node-wise layers on the block, and the rank's share of the
item-sharded evaluation (all users against 250k items, timed on 1M users)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommend_amd import ops

U, I, D, WORLD = 10_000_000, 2_000_000, 128, 8
N = U + I
block = -(-N // WORLD)
dev = torch.device("cuda", 0)
torch.manual_seed(0)

def t_ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps

t0 = time.time()
deg = torch.poisson(torch.full((block,), 39.0, device=dev)).to(torch.int64) + 1          # train edges per user row
p = 1.0 / torch.arange(1, I + 1, device=dev, dtype=torch.float64)
cdf = torch.cumsum(p / p.sum(), 0).float()
E = int(deg.sum())
cols = torch.searchsorted(cdf, torch.rand(E, device=dev)).clamp_(max=I - 1).to(torch.int32) + U
rows = torch.repeat_interleave(torch.arange(block, device=dev, dtype=torch.int32), deg)
# self loops, then CSR by row (edge order inside a row does not matter for sizing)
rows = torch.cat([rows, torch.arange(block, device=dev, dtype=torch.int32)])
cols = torch.cat([cols, torch.arange(block, device=dev, dtype=torch.int32)])
order = torch.argsort(rows.to(torch.int64), stable=True)
cols = cols[order]
rowptr = torch.zeros(block + 1, dtype=torch.int64, device=dev)
rowptr[1:] = torch.cumsum(deg + 1, 0)
nnz = int(rowptr[-1])
mat = ops.CsrMatrix(rowptr.cpu().numpy().astype(np.int32), cols.cpu().numpy(), np.arange(nnz, dtype=np.int32), dev, n_cols=N)
mat.set_values_from_coo(torch.rand(nnz, device=dev) / 40)
del rows, order, cdf, p
print(f"block of rank 0: {block} rows, {nnz} edges ({nnz * 12 / 1e9:.2f} GB as CSR), built in {time.time() - t0:.1f} s", flush=True)

h_full = torch.randn(WORLD * block, D, device=dev)
out = torch.empty(block, D, device=dev)
ms = t_ms(lambda: ops.spmm_raw(mat, h_full, out=out))
alg = nnz * 8 + (block + 1) * 4 + (nnz * 0 + block) * D * 4 + N * D * 4
print(f"local SpMM hop (D=128): {ms:.3f} ms; gathered layer {h_full.numel() * 4 / 1e9:.2f} GB; "
      f"{nnz * (8 + D * 4) / ms / 1e6:.0f} GB/s at the gather level", flush=True)
x = torch.randn(block, D, device=dev, requires_grad=True)
w = (torch.randn(D, D, device=dev) / 11).requires_grad_(True)
b = torch.zeros(D, device=dev, requires_grad=True)
al = torch.tensor([0.25], device=dev, requires_grad=True)
fwd = t_ms(lambda: ops.dense(x, w, b, alpha=al, p=0.3, seed=5))
y = ops.dense(x, w, b, alpha=al, p=0.3, seed=5)
g = torch.randn_like(y)
bwd = t_ms(lambda: torch.autograd.grad(y, (x, w, b, al), g, retain_graph=True))
print(f"128-wide layer on the block: forward {fwd:.3f} ms ({2 * block * D * 4 / fwd / 1e6:.0f} GB/s), backward {bwd:.3f} ms", flush=True)
del x, y, g, h_full, out
torch.cuda.empty_cache()
Us, Is = 1_000_000, I // WORLD
ue, ie = torch.randn(Us, D, device=dev) * 0.1, torch.randn(Is, D, device=dev) * 0.1
ev = t_ms(lambda: ops.score_topk(ue, ie, 20, mode="filter"), 3)
print(f"item-sharded evaluation share: {Us} users x {Is} items, top-20: {ev:.2f} ms -> {U / Us * ev / 1e3:.2f} s for all "
      f"{U} users ({2.0 * Us * Is * D / ev / 1e9:.0f} TFLOP/s)", flush=True)
print(f"peak device memory {torch.cuda.max_memory_allocated() / 1e9:.1f} GB", flush=True)
