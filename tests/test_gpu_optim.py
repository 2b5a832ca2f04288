"""FusedClipAdam (two launches: squared-norm partials, clip + Adam update) against
torch.nn.utils.clip_grad_norm_ + torch.optim.Adam over ten steps on the same gradients."""
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
