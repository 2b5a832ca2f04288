"""Sizing run for the large synthetic configuration (10M users x 2M items, 500M interactions,
emb_dim 128) on ONE GPU of eight: the row block of rank 0 under row sharding (1.5M user rows,
their ~40 training edges each into the 2M item rows, self loops), the all-gathered layer it reads
(12M x 128 fp32), the 128-wide node-wise layers on the block, and the rank's share of the
item-sharded evaluation (all users against 250k items, timed on 1M users).

`rank_slice_hop(dev)` is the part bench.py reports as `roofline_spmm_scaleup_slice`: the local hop
of that block, checked against a float64 sum on a sample of rows, with the bytes it cannot avoid
moving: its index arrays, every DISTINCT row of the layer its edges name (once), its output."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommend_amd import ops

U, I, D, WORLD = 10_000_000, 2_000_000, 128, 8
N = U + I
BLOCK = -(-N // WORLD)


def t_ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps


def build_rank0_block(dev, seed=0):
    """CSR of rank 0's rows (users 0 .. BLOCK-1): Poisson(39) + 1 training edges per user into item
    nodes drawn with Zipf(1) popularity, plus the self loop.  -> (CsrMatrix with values, cols int32 tensor)"""
    torch.manual_seed(seed)
    deg = torch.poisson(torch.full((BLOCK,), 39.0, device=dev)).to(torch.int64) + 1          # train edges per user row
    p = 1.0 / torch.arange(1, I + 1, device=dev, dtype=torch.float64)
    cdf = torch.cumsum(p / p.sum(), 0).float()
    E = int(deg.sum())
    cols = torch.searchsorted(cdf, torch.rand(E, device=dev)).clamp_(max=I - 1).to(torch.int32) + U
    rows = torch.repeat_interleave(torch.arange(BLOCK, device=dev, dtype=torch.int32), deg)
    # self loops, then CSR by row (edge order inside a row does not matter for sizing)
    rows = torch.cat([rows, torch.arange(BLOCK, device=dev, dtype=torch.int32)])
    cols = torch.cat([cols, torch.arange(BLOCK, device=dev, dtype=torch.int32)])
    order = torch.argsort(rows.to(torch.int64), stable=True)
    cols = cols[order]
    rowptr = torch.zeros(BLOCK + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(deg + 1, 0)
    nnz = int(rowptr[-1])
    mat = ops.CsrMatrix(rowptr.cpu().numpy().astype(np.int32), cols.cpu().numpy(), np.arange(nnz, dtype=np.int32), dev, n_cols=N)
    mat.set_values_from_coo(torch.rand(nnz, device=dev) / 40)
    return mat, cols


def build_rank7_block(dev, seed=0):
    """CSR of the last rank's rows (item nodes 10.5M .. 12M = items 0.5M .. 2M, the cold three quarters
    of the Zipf(1) popularity curve): item i has round(400M * p_i) training edges (53 down to 13) from
    users drawn uniformly (every user has about the same ~40 edges), plus the self loop.  Nearly every
    gathered row is a different one: the block whose hop HBM bounds."""
    torch.manual_seed(seed)
    lo_item = (WORLD - 1) * BLOCK - U
    ranks = torch.arange(lo_item + 1, I + 1, device=dev, dtype=torch.float64)
    harmonic = float((1.0 / torch.arange(1, I + 1, device=dev, dtype=torch.float64)).sum())
    deg = torch.round(400e6 / harmonic / ranks).to(torch.int64)
    n_rows = int(deg.numel())
    E = int(deg.sum())
    cols = torch.randint(0, U, (E,), device=dev, dtype=torch.int32)
    rows = torch.repeat_interleave(torch.arange(n_rows, device=dev, dtype=torch.int32), deg)
    rows = torch.cat([rows, torch.arange(n_rows, device=dev, dtype=torch.int32)])
    cols = torch.cat([cols, torch.arange(n_rows, device=dev, dtype=torch.int32) + (WORLD - 1) * BLOCK])
    order = torch.argsort(rows.to(torch.int64), stable=True)
    cols = cols[order]
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(deg + 1, 0)
    nnz = int(rowptr[-1])
    mat = ops.CsrMatrix(rowptr.cpu().numpy().astype(np.int32), cols.cpu().numpy(), np.arange(nnz, dtype=np.int32), dev, n_cols=N)
    mat.set_values_from_coo(torch.rand(nnz, device=dev) / 40)
    return mat, cols


def rank_slice_hop(dev, rank=0, reps=5, check_rows=2000):
    """Times the hop of one rank's block (0: user rows gathering the Zipf-popular item rows, mostly
    from cache; 7: cold item rows gathering user rows, nearly all distinct) on a full-height layer and
    checks a sample of its rows against float64.  Returns a dict: ms, rows, nnz, distinct gathered
    rows, compulsory bytes, gather-level bytes, max relative error of the sample."""
    mat, cols = build_rank0_block(dev) if rank == 0 else build_rank7_block(dev)
    n_rows = mat.n_rows
    nnz = mat.nnz
    distinct = int(torch.unique(cols).numel())
    h_full = torch.randn(WORLD * BLOCK, D, device=dev)
    out = torch.empty(n_rows, D, device=dev)
    ms = t_ms(lambda: ops.spmm_raw(mat, h_full, out=out), reps)
    # float64 check on a sample of rows (the products and sums of the row, nothing else)
    g = torch.Generator(device="cpu").manual_seed(1)
    sample = torch.randint(0, n_rows, (check_rows,), generator=g).to(dev)
    lo, hi = mat.rowptr[sample.long()].long(), mat.rowptr[sample.long() + 1].long()
    lens = hi - lo
    seg = torch.repeat_interleave(torch.arange(check_rows, device=dev), lens)
    pos = torch.arange(int(lens.sum()), device=dev) - torch.repeat_interleave(torch.cumsum(lens, 0) - lens, lens) + \
        torch.repeat_interleave(lo, lens)
    contrib = mat.val[pos].double().unsqueeze(1) * h_full[mat.idx[pos].long()].double()
    want = torch.zeros(check_rows, D, dtype=torch.float64, device=dev).index_add_(0, seg, contrib)
    err = float((out[sample.long()].double() - want).abs().max() / want.abs().max())
    compulsory = nnz * 8 + (n_rows + 1) * 4 + distinct * D * 4 + n_rows * D * 4
    res = {"ms": ms, "rows": n_rows, "nnz": nnz, "distinct_rows_gathered": distinct, "compulsory_bytes": compulsory,
           "gather_level_bytes": nnz * (8 + D * 4) + n_rows * D * 4, "max_rel_err_vs_float64": err,
           "layer_bytes": WORLD * BLOCK * D * 4}
    del h_full, out, mat, cols
    torch.cuda.empty_cache()
    return res


def main():
    dev = torch.device("cuda", 0)
    only = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--rank=")]
    for rank in (only or [0, WORLD - 1]):
        t0 = time.time()
        r = rank_slice_hop(dev, rank)
        print(f"block of rank {rank}: {r['rows']} rows, {r['nnz']} edges ({r['nnz'] * 12 / 1e9:.2f} GB as CSR), "
              f"{r['distinct_rows_gathered']} distinct layer rows gathered, {time.time() - t0:.1f} s to build, time and check", flush=True)
        print(f"  local SpMM hop (D=128): {r['ms']:.3f} ms; gathered layer {r['layer_bytes'] / 1e9:.2f} GB; "
              f"{r['gather_level_bytes'] / r['ms'] / 1e6:.0f} GB/s at the gather level; compulsory {r['compulsory_bytes'] / 1e9:.2f} GB "
              f"= {r['compulsory_bytes'] / r['ms'] / 1e6:.0f} GB/s; max rel. error of {2000} sampled rows vs float64 "
              f"{r['max_rel_err_vs_float64']:.2e}", flush=True)
    if "--spmm-only" in sys.argv:
        return
    torch.manual_seed(0)
    x = torch.randn(BLOCK, D, device=dev, requires_grad=True)
    w = (torch.randn(D, D, device=dev) / 11).requires_grad_(True)
    b = torch.zeros(D, device=dev, requires_grad=True)
    al = torch.tensor([0.25], device=dev, requires_grad=True)
    fwd = t_ms(lambda: ops.dense(x, w, b, alpha=al, p=0.3, seed=5))
    y = ops.dense(x, w, b, alpha=al, p=0.3, seed=5)
    g = torch.randn_like(y)
    bwd = t_ms(lambda: torch.autograd.grad(y, (x, w, b, al), g, retain_graph=True))
    print(f"128-wide layer on the block: forward {fwd:.3f} ms ({2 * BLOCK * D * 4 / fwd / 1e6:.0f} GB/s), backward {bwd:.3f} ms", flush=True)
    del x, y, g
    torch.cuda.empty_cache()
    Us, Is = 1_000_000, I // WORLD
    ue, ie = torch.randn(Us, D, device=dev) * 0.1, torch.randn(Is, D, device=dev) * 0.1
    ev = t_ms(lambda: ops.score_topk(ue, ie, 20, mode="filter"), 3)
    print(f"item-sharded evaluation share: {Us} users x {Is} items, top-20: {ev:.2f} ms -> {U / Us * ev / 1e3:.2f} s for all "
          f"{U} users ({2.0 * Us * Is * D / ev / 1e9:.0f} TFLOP/s)", flush=True)
    print(f"peak device memory {torch.cuda.max_memory_allocated() / 1e9:.1f} GB", flush=True)


if __name__ == "__main__":
    main()
