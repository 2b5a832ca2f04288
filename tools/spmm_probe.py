"""Probe: run the SpMM hop on a synthetic graph a few times (for rocprofv3 passes)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommend_amd import ops, synth
from recommend_amd.graph_constructor import GraphConstructor

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="elec")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--dim", type=int, default=64)
a = ap.parse_args()
d = synth.make_named(a.shape)
tu, ti, tr = d.split(0)
gc = GraphConstructor()
gc.build_graph((tu, ti, tr), d.n_users, d.n_items)
gc.creat_feature_weight(None)
g = gc.graph
g.refresh_values("weight")
N = d.n_users + d.n_items
h = torch.randn(N, a.dim, device="cuda")
out = torch.empty_like(h)
for m in (g.fwd, g.bwd):
    ops.spmm_raw(m, h, out=out)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, m in (("fwd", g.fwd), ("bwd", g.bwd)):
    s.record()
    for _ in range(a.reps):
        ops.spmm_raw(m, h, out=out)
    e.record(); e.synchronize()
    nnz = g.num_edges()
    by = nnz * 8 + (N + 1) * 4 + 2 * N * a.dim * 4
    ms = s.elapsed_time(e) / a.reps
    print(f"{name}: {ms*1e3:.1f} us/hop  {by/ms/1e6:.0f} GB/s algorithmic  (long rows {m.n_long}, chunks {m.n_chunks})")
# ping-pong two hops (h -> out -> h), as the k-hop loop runs
s.record()
for _ in range(a.reps):
    ops.spmm_raw(g.fwd, h, out=out)
    ops.spmm_raw(g.fwd, out, out=h)
e.record(); e.synchronize()
print(f"pingpong: {s.elapsed_time(e)/a.reps/2*1e3:.1f} us/hop")
