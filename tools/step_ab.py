"""A/B timing of the replayed training step in one process on one box (boxes differ by several
per cent, so variants are only comparable within a run): the step as built, against the step
with `ops.split_rows` replaced by plain slicing (STEP_AB=split), alternating A B A B."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from recommend_amd import ops, synth  # noqa: E402
from recommend_amd.loss import gather_dot_bpr_loss  # noqa: E402
from recommend_amd.model import ModelFactory  # noqa: E402
from recommend_amd.optim import FusedClipAdam  # noqa: E402
from recommend_amd.sampler import NegativeSampler, TrainBatchLoader, build_user_csr  # noqa: E402
from recommend_amd.train_step import GraphedTrainStep  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "elec"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device("cuda:0")
U, I, E = synth.SHAPES[shape]
data = synth.make_named(shape)
tu, ti, tr = data.split(0)
g = torch.Generator(device="cpu").manual_seed(1234)
feats = {"image": torch.randn(I, 4096, generator=g), "text": torch.nn.functional.normalize(torch.randn(I, 384, generator=g))}
p_ptr, p_items = build_user_csr(data.users, data.items, U)

real_split = ops.split_rows


def plain_split(x, n):
    return x[:n], x[n:]


def measure(variant):
    ops.split_rows = plain_split if variant == "sliced" else real_split
    torch.manual_seed(42)
    model = ModelFactory.create_Model(bench.make_config(U, I, 64, str(dev), 0.3), {}, feats).to(dev)
    model.build_graph((tu, ti, tr))
    model.creat_feature_weight()
    sampler = NegativeSampler(p_ptr, p_items, I, seed=42)
    dp, di = sampler.device_csr(dev)
    loader = TrainBatchLoader(tu, ti, tr, bench.BATCH[shape], sampler, device=dev, resolve_on_host=False)
    opt = FusedClipAdam(model.parameters(), lr=1e-3)
    model.train()
    it = iter(loader)
    first = next(it)
    stepper = GraphedTrainStep(model, opt, gather_dot_bpr_loss, first, constants={"pos_ptr": dp, "pos_items": di},
                               max_grad_norm=1.0)
    for _ in range(5):
        stepper(next(it))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = stepper(next(it))
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    # the captured graph alone, replayed back to back on the batch it holds (no loader, no copies)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        stepper.graph.replay()
    torch.cuda.synchronize()
    ms_replay = 1e3 * (time.perf_counter() - t0) / steps
    # and the host side alone: batches drawn and copied, nothing launched
    t0 = time.perf_counter()
    for _ in range(steps):
        b = next(it)
    torch.cuda.synchronize()
    ms_loader = 1e3 * (time.perf_counter() - t0) / steps
    print(f"{variant:8s} {ms:.4f} ms/step  (graph replays alone {ms_replay:.4f}, loader alone {ms_loader:.4f})  loss {float(loss):.6f}", flush=True)
    return ms


res = {"built": [], "sliced": []}
for _ in range(2):
    for v in ("sliced", "built"):
        res[v].append(measure(v))
print({k: round(sum(v) / len(v), 4) for k, v in res.items()})
