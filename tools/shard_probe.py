"""Probe: the per-rank top-K call of the item-sharded evaluation at Elec user count, for the
item-shard sizes of 1/2/4/8 ranks, exact kernel against the candidate filter."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from recommend_amd import ops, synth
from recommend_amd.evaluator import pairs_to_csr, restrict_csr

U, I, D = 192403, 63001, 64
d = synth.make_named("elec")
tu, ti, _ = d.split(0)
mp, mc = pairs_to_csr(tu, ti, U)
g = torch.Generator().manual_seed(0)
ue = (torch.randn(U, D, generator=g) * 0.1).cuda()
ie = (torch.randn(I, D, generator=g) * 0.1).cuda()

def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps

WORLDS = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
for world in WORLDS:
    hi = I // world
    p, c = restrict_csr(mp, mc, 0, hi)
    pt, ct = torch.from_numpy(p).cuda(), torch.from_numpy(c).cuda()
    shard = ie[:hi].contiguous()
    row = [f"world {world}: {hi:6d} items/rank"]
    for mode in ("exact", "filter", "auto"):
        row.append(f"{mode} {t(lambda: ops.score_topk(ue, shard, 20, pt, ct, mode=mode)):.3f} ms")
    print("  ".join(row))
