"""Probe: where the time of bench.py's timed evaluation loop goes besides the steps themselves --
host time per evaluate() call, GPU time between per-step events, and the wall clock around K steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommend_amd import synth
from recommend_amd.evaluator import RankingEvaluator

K_LIST, METRICS = [5, 10, 20], ["precision", "recall", "hit_rate", "ndcg", "map", "mrr"]
U, I, E = synth.SHAPES["elec"]
d = synth.make_named("elec")
tu, ti, _ = d.split(0)
vu, vi, _ = d.split(1)
g = torch.Generator().manual_seed(0)
ue = (torch.randn(U, 64, generator=g) * 0.1).cuda()
ie = (torch.randn(I, 64, generator=g) * 0.1).cuda()
ev = RankingEvaluator((vu, vi), (tu, ti), K_LIST, METRICS, n_users=U)
for _ in range(2):
    ev.evaluate(ue, ie)
for steps in (10, 10, 40):
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    host = []
    t0 = time.perf_counter()
    evs[0].record()
    for s in range(steps):
        h0 = time.perf_counter()
        ev.evaluate(ue, ie)
        host.append(time.perf_counter() - h0)
        evs[s + 1].record()
    t_launch = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    gpu = [evs[s].elapsed_time(evs[s + 1]) for s in range(steps)]
    print(f"steps {steps}: wall {1e3 * wall:.2f} ms = {1e3 * wall / steps:.4f} per step; all launched after {1e3 * t_launch:.2f} ms; "
          f"host per call {1e3 * min(host):.3f}..{1e3 * max(host):.3f} ms; gpu per step first {gpu[0]:.3f}, rest {min(gpu[1:]):.3f}..{max(gpu[1:]):.3f}, "
          f"sum {sum(gpu):.2f}", flush=True)
