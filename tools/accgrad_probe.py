"""Where does autograd's 'AccumulateGrad node's stream does not match' warning come from in the
captured training step?  Runs the phases one by one and reports the first that warns."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from recommend_amd import ops, synth
from recommend_amd.loss import gather_dot_bpr_loss
from recommend_amd.model import ModelFactory
from recommend_amd.optim import FusedClipAdam
import bench

dev = torch.device("cuda:0")
U, I, E = synth.SHAPES["baby"]
data = synth.make_named("baby")
tu, ti, tr = data.split(0)
g = torch.Generator().manual_seed(1)
feats = {"image": torch.randn(I, 512, generator=g), "text": torch.randn(I, 96, generator=g)}
cfg = bench.make_config(U, I, 64, "cuda:0", 0.3)
model = ModelFactory.create_Model(cfg, {}, feats).to(dev)
model.build_graph((tu, ti, tr)); model.creat_feature_weight()
opt = FusedClipAdam(model.parameters(), lr=1e-3, max_grad_norm=1.0)
batch = {"user_ids": torch.from_numpy(tu[:256]).to(dev), "item_ids": torch.from_numpy(ti[:256]).to(dev),
         "neg_items": torch.randint(0, I, (256, 1), device=dev)}
model.train()

def phase(name, fn):
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        fn()
        torch.cuda.synchronize()
    hits = [str(x.message)[:60] for x in w if "AccumulateGrad" in str(x.message)]
    print(f"{name}: {'WARNS' if hits else 'quiet'}", flush=True)

def body():
    loss = gather_dot_bpr_loss(model(batch), batch)
    loss.backward()
    opt.step()
    return loss.detach()

s = torch.cuda.Stream(device=dev)
s.wait_stream(torch.cuda.current_stream(dev))
def warm():
    with torch.cuda.stream(s):
        opt.zero_grad(set_to_none=True)
        body()
phase("warm-up 1 (side stream)", warm)
phase("warm-up 2 (side stream)", warm)
torch.cuda.current_stream(dev).wait_stream(s)
graph = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
def cap():
    with torch.cuda.graph(graph, stream=s):
        body()
phase("capture (same side stream)", cap)
phase("replay", graph.replay)
