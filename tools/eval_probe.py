"""Probe: fused score + mask + top-K at a named shape with random embeddings."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommend_amd import ops, synth
from recommend_amd.evaluator import pairs_to_csr

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="elec"); ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--k", type=int, default=20); ap.add_argument("--dim", type=int, default=64)
a = ap.parse_args()
d = synth.make_named(a.shape)
tu, ti, _ = d.split(0)
ptr, cols = pairs_to_csr(tu, ti, d.n_users)
dev = "cuda"
g = torch.Generator().manual_seed(0)
ue = (torch.randn(d.n_users, a.dim, generator=g) * 0.1).to(dev)
ie = (torch.randn(d.n_items, a.dim, generator=g) * 0.1).to(dev)
mp, mc = torch.from_numpy(ptr).to(dev), torch.from_numpy(cols).to(dev)
ops.score_topk(ue, ie, a.k, mp, mc); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(a.reps): ops.score_topk(ue, ie, a.k, mp, mc)
e.record(); e.synchronize()
ms = s.elapsed_time(e) / a.reps
print(f"score_topk {a.shape} K={a.k}: {ms:.3f} ms  {2.0*d.n_users*d.n_items*a.dim/ms/1e9:.1f} TFLOP/s")
