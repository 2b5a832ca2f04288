"""Probe: the per-rank top-K call of the USER-sharded evaluation at the Elec shape -- the block of
users one of 1/2/4/8 ranks ranks against the whole item table -- exact kernel, candidate filter and
what the library picks, plus the whole evaluate() of such a block (top-K + metrics + host copy)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from recommend_amd import ops, synth
from recommend_amd.evaluator import pairs_to_csr

U, I, D = 192403, 63001, 64
d = synth.make_named("elec")
tu, ti, _ = d.split(0)
mp, mc = pairs_to_csr(tu, ti, U)
g = torch.Generator().manual_seed(0)
ue = (torch.randn(U, D, generator=g) * 0.1).cuda()
ie = (torch.randn(I, D, generator=g) * 0.1).cuda()

def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps

WORLDS = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
base = None
for world in WORLDS:
    hi = U // world
    p = (mp[:hi + 1] - mp[0]).astype(np.int32)
    c = mc[:mp[hi]].astype(np.int32)
    pt, ct = torch.from_numpy(p).cuda(), torch.from_numpy(c).cuda()
    blk = ue[:hi].contiguous()
    row = [f"world {world}: {hi:6d} users/rank"]
    for mode in os.environ.get("PROBE_MODES", "exact,filter,auto").split(","):
        ms = t(lambda: ops.score_topk(blk, ie, 20, pt, ct, mode=mode))
        row.append(f"{mode} {ms:.3f} ms")
    if world == 1:
        base = ms
    if base is not None:
        row.append(f"ideal {base / world:.3f} ms")
    print("  ".join(row), flush=True)
