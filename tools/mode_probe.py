"""Probe: exact kernel against the candidate filter at the Baby / Sports / Clothing shapes (where the
automatic choice of rk_score_topk_f32 switches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommend_amd import ops, synth
from recommend_amd.evaluator import pairs_to_csr
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps
g = torch.Generator().manual_seed(0)
for name in ("baby", "sports", "clothing"):
    d = synth.make_named(name)
    tu, ti, _ = d.split(0)
    p, c = pairs_to_csr(tu, ti, d.n_users)
    pt, ct = torch.from_numpy(p).cuda(), torch.from_numpy(c).cuda()
    ue = (torch.randn(d.n_users, 64, generator=g) * 0.1).cuda()
    ie = (torch.randn(d.n_items, 64, generator=g) * 0.1).cuda()
    print(name, d.n_users, d.n_items, " ".join(f"{m} {t(lambda: ops.score_topk(ue, ie, 20, pt, ct, mode=m)):.3f} ms" for m in ("exact", "filter")))
