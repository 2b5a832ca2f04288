"""FusedClipAdam (two launches: squared-norm partials, clip + Adam update) against
torch.nn.utils.clip_grad_norm_ + torch.optim.Adam over ten steps on the same gradients."""
import copy

import pytest
import torch

from recommend_amd.optim import FusedClipAdam

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("wd,max_norm", [(0.0, 1.0), (0.01, 0.3), (0.0, None)])
def test_fused_clip_adam_matches_torch_over_ten_steps(wd, max_norm):
    g = torch.Generator().manual_seed(3)
    shapes = [(64, 64), (64,), (1,), (64, 4096), (5000, 7), (128, 128), (3,)]
    init = [torch.randn(s, generator=g) * 0.1 for s in shapes]
    ours = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    ref = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    unused = torch.nn.Parameter(torch.zeros(10, device=DEV))         # never gets a gradient: stays out
    opt = FusedClipAdam(ours + [unused], lr=1e-3, weight_decay=wd, max_grad_norm=max_norm)
    topt = torch.optim.Adam(ref, lr=1e-3, weight_decay=wd)
    for step in range(10):
        scale = 10.0 if step % 3 == 0 else 0.05                      # norms above and below the limit
        grads = [torch.randn(s, generator=g) * scale for s in shapes]
        for p, q, gr in zip(ours, ref, grads):
            p.grad = gr.clone().to(DEV)
            q.grad = gr.clone().to(DEV)
        want_norm = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(q.grad) for q in ref]))
        if max_norm is not None:
            torch.nn.utils.clip_grad_norm_(ref, max_norm)
        topt.step()
        opt.step()
        assert abs(float(opt.grad_norm) - float(want_norm)) <= 1e-5 * float(want_norm)
        if step == 4:                                                # a scheduler changes the rate
            for o in (opt, topt):
                o.param_groups[0]["lr"] = 5e-4
    for p, q in zip(ours, ref):
        assert float((p - q).abs().max()) <= 1e-6 * max(1.0, float(q.abs().max())), p.shape
    assert float(unused.abs().max()) == 0.0
    assert int(opt.step_count) == 10


def _grads(shapes, g, scale):
    return [torch.randn(s, generator=g) * scale for s in shapes]


def test_checkpoint_round_trip_continues_like_torch_adam():
    """train/graph_trainer.py:279,315 checkpoint `optimizer.state_dict()`: after save -> load into a
    NEW optimizer (and into one that has already stepped) the next steps equal torch.optim.Adam's,
    i.e. the bias correction continues at the saved step count and the loaded moments are the
    ones updated.  The state dict has torch.optim.Adam's layout: torch's own checkpoint loads too."""
    g = torch.Generator().manual_seed(11)
    shapes = [(64, 64), (64,), (300, 5)]
    init = [torch.randn(s, generator=g) * 0.1 for s in shapes]
    ours = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    ref = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    opt = FusedClipAdam(ours, lr=1e-3, max_grad_norm=1.0)
    topt = torch.optim.Adam(ref, lr=1e-3)

    def both_step(o, params):
        grads = _grads(shapes, g, 0.5)
        for p, q, gr in zip(params, ref, grads):
            p.grad = gr.clone().to(DEV)
            q.grad = gr.clone().to(DEV)
        torch.nn.utils.clip_grad_norm_(ref, 1.0)
        topt.step()
        o.step()

    for _ in range(4):
        both_step(opt, ours)
    sd = copy.deepcopy(opt.state_dict())           # (as torch.save / torch.load hand it over: no shared tensors)
    assert all(float(st["step"]) == 4.0 for st in sd["state"].values())
    # (a) a fresh optimizer over fresh parameter objects, as a resumed run builds them
    ours2 = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    opt2 = FusedClipAdam(ours2, lr=1e-3, max_grad_norm=1.0)
    opt2.load_state_dict(sd)
    for _ in range(3):
        both_step(opt2, ours2)
    assert int(opt2.step_count) == 7
    for p, q in zip(ours2, ref):
        assert float((p - q).abs().max()) <= 1e-6 * max(1.0, float(q.abs().max()))
    # (b) loading into an optimizer that has stepped: its descriptor tables must not outlive the moments
    sd_t = copy.deepcopy(topt.state_dict())        # torch's own checkpoint, step 7
    for p, q in zip(ours, ref):
        p.data.copy_(q.data)
    where = [opt.state[p]["exp_avg"].data_ptr() for p in ours]
    opt.load_state_dict(sd_t)
    # the moments took the checkpoint's values in place: a step recorded in a HIP graph earlier holds these addresses
    assert where == [opt.state[p]["exp_avg"].data_ptr() for p in ours]
    for p, q in zip(ours, ref):
        assert torch.equal(opt.state[p]["exp_avg"], topt.state[q]["exp_avg"])
    for _ in range(2):
        both_step(opt, ours)
    assert int(opt.step_count) == 9
    for p, q in zip(ours, ref):
        assert float((p - q).abs().max()) <= 1e-6 * max(1.0, float(q.abs().max()))
    # and torch.optim.Adam accepts ours
    torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in ours], lr=1e-3).load_state_dict(opt.state_dict())


def test_row_sharded_parameters_clip_by_the_norm_over_all_ranks():
    """Two 'ranks' in one process: each holds the replicated tensors and its own rows of a sharded
    table; the all-reduce between the two launches is a function that adds the other rank's shard
    sum.  Both ranks must end with the parameters of torch.optim.Adam + clip_grad_norm_ over the
    whole table."""
    g = torch.Generator().manual_seed(5)
    rep_shapes, rows, D = [(64, 64), (64,)], 1000, 64
    rep0 = [torch.randn(s, generator=g) * 0.1 for s in rep_shapes]
    table0 = torch.randn(rows, D, generator=g) * 0.1
    ref = [torch.nn.Parameter(t.clone().to(DEV)) for t in rep0 + [table0]]
    topt = torch.optim.Adam(ref, lr=1e-3)
    cut = 437
    ranks = []
    for lo, hi in ((0, cut), (cut, rows)):
        ps = [torch.nn.Parameter(t.clone().to(DEV)) for t in rep0] + [torch.nn.Parameter(table0[lo:hi].clone().to(DEV))]
        ranks.append({"params": ps, "lo": lo, "hi": hi})
    pending = {}

    def make_all_reduce(me, other):
        def all_reduce(t, group):
            # rank `me` arrives with its shard sum; the other rank's sum was computed beforehand
            t.add_(pending[other])
        return all_reduce

    for r, (me, other) in enumerate(((0, 1), (1, 0))):
        ranks[r]["opt"] = FusedClipAdam(ranks[r]["params"], lr=1e-3, max_grad_norm=0.7,
                                        sharded_params=[ranks[r]["params"][-1]], all_reduce=make_all_reduce(me, other))
    for step in range(5):
        grads = _grads(rep_shapes + [(rows, D)], g, 2.0 if step % 2 == 0 else 0.01)
        for q, gr in zip(ref, grads):
            q.grad = gr.clone().to(DEV)
        want_norm = float(torch.linalg.vector_norm(torch.cat([q.grad.reshape(-1) for q in ref])))
        torch.nn.utils.clip_grad_norm_(ref, 0.7)
        topt.step()
        for ri, r in enumerate(ranks):
            for p, gr in zip(r["params"][:-1], grads[:-1]):
                p.grad = gr.clone().to(DEV)
            r["params"][-1].grad = grads[-1][r["lo"]:r["hi"]].clone().to(DEV)
            pending[ri] = (r["params"][-1].grad.double() ** 2).sum().float().reshape(1)
        for r in ranks:
            r["opt"].step()
            assert abs(float(r["opt"].grad_norm) - want_norm) <= 2e-5 * want_norm
    for r in ranks:
        for p, q in zip(r["params"][:-1], ref[:-1]):
            assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(q.abs().max()))
        assert float((r["params"][-1] - ref[-1][r["lo"]:r["hi"]]).abs().max()) <= 2e-6 * max(1.0, float(ref[-1].abs().max()))
