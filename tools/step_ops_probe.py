"""Probe: which framework ops (not library kernels) one eager MMFCN training step launches -- the
copies, fills and elementwise kernels that show up between the library's kernels in a step trace."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from recommend_amd import synth  # noqa: E402
from recommend_amd.loss import gather_dot_bpr_loss  # noqa: E402
from recommend_amd.model import ModelFactory  # noqa: E402
from recommend_amd.optim import FusedClipAdam  # noqa: E402
from recommend_amd.sampler import NegativeSampler, TrainBatchLoader, build_user_csr  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "elec"
dev = torch.device("cuda:0")
U, I, E = synth.SHAPES[shape]
data = synth.make_named(shape)
tu, ti, tr = data.split(0)
g = torch.Generator(device="cpu").manual_seed(1234)
feats = {"image": torch.randn(I, 4096, generator=g), "text": torch.nn.functional.normalize(torch.randn(I, 384, generator=g))}
p_ptr, p_items = build_user_csr(data.users, data.items, U)
torch.manual_seed(42)
model = ModelFactory.create_Model(bench.make_config(U, I, bench.EMB_DIM[shape], str(dev), 0.3), {}, feats).to(dev)
model.build_graph((tu, ti, tr))
model.creat_feature_weight()
sampler = NegativeSampler(p_ptr, p_items, I, seed=42)
dp, di = sampler.device_csr(dev)
loader = TrainBatchLoader(tu, ti, tr, bench.BATCH[shape], sampler, device=dev, resolve_on_host=False)
opt = FusedClipAdam(model.parameters(), lr=1e-3, max_grad_norm=1.0)
model.train()
it = iter(loader)


def step():
    batch = next(it)
    batch["pos_ptr"], batch["pos_items"] = dp, di
    opt.zero_grad(set_to_none=True)
    loss = gather_dot_bpr_loss(model(batch), batch)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
import traceback  # noqa: E402
# the profiler's stacks are empty on this build: find the call sites by running the step once more
# with the interesting tensor methods wrapped
sites = {}


def wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        fr = [x for x in traceback.extract_stack()[:-1] if "recommend_amd" in x.filename or "bench.py" in x.filename]
        key = (name, " <- ".join(f"{os.path.basename(x.filename)}:{x.lineno}" for x in fr[-3:]))
        sites[key] = sites.get(key, 0) + 1
        return orig(*a, **k)
    setattr(owner, name, f)


for nm in ("copy_", "contiguous", "clone", "to", "fill_", "zero_"):
    wrap(torch.Tensor, nm)
for nm in ("zeros", "zeros_like", "ones_like", "empty_like", "cat", "stack"):
    wrap(torch, nm)
step()
torch.cuda.synchronize()
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total):
    if e.key.startswith("aten::") and e.device_time_total > 0:
        print(f"{e.key:28s} x{e.count:3d}  {e.device_time_total:8.1f} us")
print("-- copies, fills and their shapes, in launch order")
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::add_", "aten::ones_like") and e.device_time_total > 0:
        print(f"   {e.name:18s} {e.device_time_total:6.1f} us  shapes {e.input_shapes}")
for (name, where), n in sorted(sites.items()):
    print(f"{name:12s} x{n:3d}  {where}")
