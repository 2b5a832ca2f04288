"""GPU parity: fused node-wise layer (Linear + PReLU + dropout [+ residual], and
the two-input linear sum) forward and backward against a float64 torch
restatement of the same composition of nn modules.  Tolerance 1e-5 of scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from recommend_amd import fused, ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def make(n, seed, D=64):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, D, generator=g)
    w = torch.randn(D, D, generator=g) / 8
    b = torch.randn(D, generator=g)
    r = torch.randn(n, D, generator=g)
    go = torch.randn(n, D, generator=g)
    return x, w, b, r, go


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("n", [1, 63, 64, 130, 1000, 70000])
@pytest.mark.parametrize("act,res", [(True, False), (True, True), (False, False)])
def test_dense_forward_backward_no_dropout(n, act, res, D):
    x, w, b, r, go = make(n, n, D)
    alpha = torch.tensor([0.25])
    xd = x.to(DEV).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True)
    ad = alpha.to(DEV).requires_grad_(True) if act else None
    rd = r.to(DEV).requires_grad_(True) if res else None
    y = ops.dense(xd, wd, bd, alpha=ad, residual=rd)
    y.backward(go.to(DEV))
    x6, w6, b6 = (t.double().requires_grad_(True) for t in (x, w, b))
    a6 = alpha.double().requires_grad_(True)
    r6 = r.double().requires_grad_(True)
    y6 = F.linear(x6, w6, b6)
    if act:
        y6 = F.prelu(y6, a6)
    if res:
        y6 = y6 + r6
    y6.backward(go.double())
    assert rel(y, y6) <= 1e-5
    assert rel(xd.grad, x6.grad) <= 1e-5
    assert rel(wd.grad, w6.grad) <= 1e-5
    assert rel(bd.grad, b6.grad) <= 1e-5
    if act:
        assert rel(ad.grad, a6.grad) <= 1e-5
    if res:
        assert torch.equal(rd.grad.cpu(), go)


@pytest.mark.parametrize("D", [64, 128])
def test_dense_two_input_linear_sum(D):
    n = 5000
    x1, w1, b1, _, go = make(n, 1, D)
    x2, w2, b2, _, _ = make(n, 2, D)
    t = [v.to(DEV).requires_grad_(True) for v in (x1, w1, b1, x2, w2, b2)]
    y = ops.dense(t[0], t[1], t[2], x2=t[3], weight2=t[4], bias2=t[5])
    y.backward(go.to(DEV))
    d = [v.double().requires_grad_(True) for v in (x1, w1, b1, x2, w2, b2)]
    y6 = F.linear(d[0], d[1], d[2]) + F.linear(d[3], d[4], d[5])
    y6.backward(go.double())
    assert rel(y, y6) <= 1e-5
    for a, b in zip(t, d):
        assert rel(a.grad, b.grad) <= 1e-5


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("act", [True, False])
def test_dense_dropout_is_consistent_and_unbiased(act, D):
    """Dropout uses its own counter hash: check the keep rate, the 1/(1-p) scaling,
    that forward and backward use the same mask, and seed reproducibility."""
    n, p = 20000, 0.3
    x, w, b, _, go = make(n, 7, D)
    alpha = torch.tensor([0.25], device=DEV, requires_grad=True) if act else None
    xd = x.to(DEV).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True)
    y = ops.dense(xd, wd, bd, alpha=alpha, p=p, seed=1234)
    y2 = ops.dense(xd, wd, bd, alpha=alpha, p=p, seed=1234)
    y3 = ops.dense(xd, wd, bd, alpha=alpha, p=p, seed=1235)
    assert torch.equal(y, y2) and not torch.equal(y, y3)
    full = ops.dense(xd, wd, bd, alpha=alpha)
    keep = (y != 0) | (full == 0)
    rate = float(keep.float().mean())
    assert abs(rate - (1 - p)) < 0.005
    torch.testing.assert_close(y[keep], (full / (1 - p))[keep], rtol=1e-6, atol=1e-7)
    # rows/columns are not correlated: per-column keep rate is also ~1-p
    assert float((keep.float().mean(0) - (1 - p)).abs().max()) < 0.02
    # backward: gradient equals that of the same function with the mask made explicit
    y.backward(go.to(DEV))
    mask = keep.double().cpu() / (1 - p)
    x6, w6, b6 = (t.double().requires_grad_(True) for t in (x, w, b))
    z6 = F.linear(x6, w6, b6)
    if act:
        a6 = torch.tensor([0.25], dtype=torch.float64, requires_grad=True)
        z6 = F.prelu(z6, a6)
    (z6 * mask).backward(go.double())
    assert rel(xd.grad, x6.grad) <= 1e-5
    assert rel(wd.grad, w6.grad) <= 1e-5
    assert rel(bd.grad, b6.grad) <= 1e-5
    if act:
        assert rel(alpha.grad, a6.grad) <= 1e-5


def test_module_routing_matches_unfused_modules():
    """fused.linear_act / linear_sum give what the plain nn modules give (eval mode)."""
    torch.manual_seed(0)
    lin, lin2 = torch.nn.Linear(64, 64).to(DEV), torch.nn.Linear(64, 64).to(DEV)
    act, dr = torch.nn.PReLU().to(DEV), torch.nn.Dropout(0.5).eval()
    x, x2 = torch.randn(777, 64, device=DEV), torch.randn(777, 64, device=DEV)
    torch.testing.assert_close(fused.linear_act(lin, x, act, dr), dr(act(lin(x))), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(fused.linear_act(lin, x, None, dr, residual=x2), x2 + lin(x), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(fused.linear_sum(lin, x, lin2, x2), lin(x) + lin2(x2), rtol=1e-5, atol=1e-5)
    # shapes the kernels are not built for raise: there is no second implementation behind them
    from recommend_amd import _ffi
    lin3 = torch.nn.Linear(48, 64).to(DEV)
    x3 = torch.randn(10, 48, device=DEV)
    with pytest.raises(_ffi.RkError) as e:
        fused.linear_act(lin3, x3, act, dr)
    assert e.value.code == _ffi.RK_ERR_UNSUPPORTED
    with pytest.raises(_ffi.RkError) as e:
        fused.linear_act(lin.cpu(), x.cpu(), act.cpu(), dr)
    assert e.value.code == _ffi.RK_ERR_NODEV
    lin.to(DEV), act.to(DEV)
    with pytest.raises(_ffi.RkError):
        ops.modal_sum([torch.randn(8, 64, device=DEV) for _ in range(5)], torch.randn(5, device=DEV))


@pytest.mark.parametrize("M,n", [(2, 1000), (3, 255404), (1, 7), (4, 64)])
def test_modal_weighted_sum_vs_torch(M, n):
    """out = sum_m softmax(alphas)_m F_m (out_Layer.py:52-54): forward and all gradients against
    the reference formulation in torch (fp64 for the reductions)."""
    g = torch.Generator().manual_seed(M * 100 + n % 97)
    feats = [torch.randn(n, 64, generator=g) for _ in range(M)]
    alphas = torch.randn(M, generator=g)
    up = torch.randn(n, 64, generator=g)
    fd = [f.to(DEV).requires_grad_(True) for f in feats]
    ad = alphas.to(DEV).requires_grad_(True)
    assert ops.modal_sum_supported(fd, ad)
    out = ops.modal_sum(fd, ad)
    out.backward(up.to(DEV))
    fr = [f.double().requires_grad_(True) for f in feats]
    ar = alphas.double().requires_grad_(True)
    ref = torch.sum(torch.stack(fr, dim=0) * torch.softmax(ar, dim=0).unsqueeze(-1).unsqueeze(-1), dim=0)
    ref.backward(up.double())
    torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=1e-5, atol=1e-6)
    for a, b in zip(fd, fr):
        torch.testing.assert_close(a.grad.cpu().double(), b.grad, rtol=1e-5, atol=1e-6)
    scale = float(ar.grad.abs().max()) + 1e-12
    assert float((ad.grad.cpu().double() - ar.grad).abs().max()) <= 1e-4 * scale + 1e-6


def test_dense_products_keep_fp32_accuracy_over_the_exponent_range():
    """The kernels multiply on the bf16 matrix pipe with every fp32 operand split into three
    bf16 parts.  Componentwise error against float64, relative to sum_k |x_k w_k|, must stay at
    the level of an fp32 dot product (64 terms: a few 1e-7), for rows spread over 24 decades."""
    n = 4096
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, 64, generator=g) * 10.0 ** (torch.rand(n, 1, generator=g) * 24 - 12)
    w = torch.randn(64, 64, generator=g) * 10.0 ** (torch.rand(64, 1, generator=g) * 4 - 2)
    go = torch.randn(n, 64, generator=g)
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = ops.dense(xd, wd, None)
    y.backward(go.to(DEV))
    x6, w6, g6 = x.double(), w.double(), go.double()
    bound_y = x6.abs() @ w6.abs().T
    assert float(((y.detach().cpu().double() - x6 @ w6.T).abs() / bound_y).max()) <= 1e-6
    bound_dx = g6.abs() @ w6.abs()
    assert float(((xd.grad.cpu().double() - g6 @ w6).abs() / bound_dx).max()) <= 1e-6
    bound_dw = g6.abs().T @ x6.abs()
    assert float(((wd.grad.cpu().double() - g6.T @ x6).abs() / bound_dw).max()) <= 1e-6


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("n", [5000, 70001])
def test_deferred_weight_gradient_reductions_equal_the_immediate_ones(n, D):
    """ops.DenseReduceQueue: the closing reductions of several layer backward calls as one launch,
    among them one weight applied three times (its slabs are added in call order).  Every output
    bit for bit what the calls produce when each reduces at once."""
    g = torch.Generator().manual_seed(n)
    xs = [torch.randn(n, D, generator=g).to(DEV) for _ in range(5)]
    gs = [torch.randn(n, D, generator=g).to(DEV) for _ in range(5)]
    ws_ = [(torch.randn(D, D, generator=g) * 0.2).to(DEV) for _ in range(3)]
    b = (torch.randn(D, generator=g) * 0.1).to(DEV)
    al = torch.tensor([0.25], device=DEV)

    def run(queue):
        outs = []
        # one weight, three applications, gradients accumulated
        acc = None
        for t in range(3):
            dx, dw, db, da = ops.dense_bwd_raw(xs[t], ws_[0], b, al, 0.3, 77 + t, gs[t], True, True, acc, None, queue)
            acc = (dw, db, da)
            outs.append(dx)
        outs += list(acc)
        # a linear layer without bias gradient or input gradient, and one without activation
        dx, dw, db, da = ops.dense_bwd_raw(xs[3], ws_[1], None, None, 0.0, 0, gs[3], False, False, None, None, queue)
        assert dx is None and db is None and da is None
        outs.append(dw)
        dx, dw, db, da = ops.dense_bwd_raw(xs[4], ws_[2], b, None, 0.0, 0, gs[4], True, True, None, None, queue)
        outs += [dx, dw, db]
        if queue is not None:
            queue.flush()
        return outs

    want = run(None)
    q = ops.DenseReduceQueue(n, D, torch.device(DEV))
    assert q.usable
    got = run(q)
    assert not q.jobs
    for a, e in zip(got, want):
        assert torch.equal(a, e)
    # a width the deferred form is not built for reduces at once
    q256 = ops.DenseReduceQueue(1000, 256, torch.device(DEV))
    assert not q256.usable
