"""MGDCF / II_GCN / IU_GCN on the HIP SpMM against the oracle (parity unpinned: the reference's
arithmetic lives in DGL, which is not installed; the oracle follows model/mig/mgdcf.py and
model/sgrec/gcn.py as text and is cross-checked against a dense fp64 computation on the CPU)."""
import numpy as np
import pytest
import torch

from oracle import graph_oracle
from recommend_amd.propagation import MGDCF, II_GCN, IU_GCN

pytestmark = pytest.mark.gpu
DEV = "cuda"
RTOL = 1e-5


def _graph(U=300, I=200, E=2500, seed=0):
    rng = np.random.default_rng(seed)
    pairs = np.unique(np.stack([rng.integers(0, U, E), rng.integers(0, I, E)], 1), axis=0)
    g = MGDCF.build_sorted_homo_graph(pairs, U, I)
    MGDCF.norm_adj(g)
    src, dst, n = graph_oracle.mgdcf_sorted_homo_edges(pairs[:, 0], pairs[:, 1], U, I)
    np.testing.assert_array_equal(g.src, src)
    np.testing.assert_array_equal(g.dst, dst)
    w = graph_oracle.deg_normalize(src, dst, n)
    np.testing.assert_allclose(g.edata[MGDCF.CACHE_KEY].cpu().numpy(), w, rtol=1e-6)   # pow(-0.5): 1 ulp apart
    return g, src, dst, w, n


def _close(a, b):
    scale = np.abs(b).max() + 1e-30
    assert np.abs(a - b).max() <= RTOL * scale, np.abs(a - b).max() / scale


@pytest.mark.parametrize("k,alpha,beta", [(1, 0.1, 0.9), (3, 0.1, 0.9), (4, 0.3, 1.0)])
def test_mgdcf_eval_mode_vs_oracle(k, alpha, beta):
    g, src, dst, w, n = _graph()
    x = torch.randn(n, 64, generator=torch.Generator().manual_seed(1))
    layer = MGDCF(k, alpha, beta, 0.3, 0.3, 0.3).eval()
    assert abs(float(layer.gamma) - graph_oracle.mgdcf_gamma(alpha, beta, k)) < 1e-6
    _close(layer(g, x.to(DEV)).cpu().numpy(), graph_oracle.mgdcf_forward(src, dst, w, x.numpy(), k, alpha, beta))
    got = layer(g, x.to(DEV), return_all=True)
    want = graph_oracle.mgdcf_forward(src, dst, w, x.numpy(), k, alpha, beta, return_all=True)
    assert len(got) == k
    for a, b in zip(got, want):
        _close(a.cpu().numpy(), b)


def test_sgrec_layers_eval_mode_vs_oracle():
    g, src, dst, w, n = _graph(seed=3)
    x = torch.randn(n, 64, generator=torch.Generator().manual_seed(2))
    ii = II_GCN(3, 0.1, 0.9, 0.2, 0.2, 0.2, MGDCF.CACHE_KEY).eval()
    _close(ii(x.to(DEV), g).cpu().numpy(), graph_oracle.ii_gcn_forward(src, dst, w, x.numpy(), 3, 0.1, 0.9))
    iu = IU_GCN(0.2, 0.2, 0.2, MGDCF.CACHE_KEY).eval()
    _close(iu(x.to(DEV), g, 2).cpu().numpy(), graph_oracle.iu_gcn_forward(src, dst, w, x.numpy(), 2))


def test_backward_matches_dense_autograd():
    g, src, dst, w, n = _graph(U=60, I=40, E=300, seed=5)
    A = torch.zeros(n, n, dtype=torch.float64)
    A.index_put_((torch.from_numpy(dst).long(), torch.from_numpy(src).long()), torch.from_numpy(w).double(),
                 accumulate=True)
    x = torch.randn(n, 64, generator=torch.Generator().manual_seed(4))
    r = torch.randn(n, 64, generator=torch.Generator().manual_seed(5))
    xd = x.double().requires_grad_(True)
    h = xd
    for _ in range(3):
        h = 0.9 * (A @ h) + 0.1 * xd
    ((h / graph_oracle.mgdcf_gamma(0.1, 0.9, 3)) * r.double()).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    (MGDCF(3, 0.1, 0.9, 0.0, 0.0, 0.0)(g, xg) * r.to(DEV)).sum().backward()
    _close(xg.grad.cpu().numpy(), xd.grad.float().numpy())


def test_training_mode_dropouts_are_consistent_between_forward_and_backward():
    """With the generator state fixed the layer is linear in x (masks are data-independent):
    f(2x) = 2 f(x) for the same draw, and the autograd gradient equals the directional difference
    -- which holds only if backward used the very edge / x / z masks of that forward."""
    g, src, dst, w, n = _graph(seed=7)
    layer = MGDCF(2, 0.1, 0.9, 0.3, 0.3, 0.3).train()
    x = torch.randn(n, 64, device=DEV)
    d = torch.randn(n, 64, device=DEV)
    r = torch.randn(n, 64, device=DEV)

    def f(inp):
        torch.manual_seed(11)
        return layer(g, inp)
    y1, y2 = f(x), f(2 * x)
    assert torch.allclose(y2, 2 * y1, rtol=1e-5, atol=1e-6)
    assert not torch.equal(y1, layer.eval()(g, x)), "dropout had no effect in training mode"
    layer.train()
    xg = x.clone().requires_grad_(True)
    (f(xg) * r).sum().backward()
    want = ((f(x + d) - f(x - d)) * r).sum() / 2          # exact for a linear map
    got = (xg.grad * d).sum()
    assert abs(float(got) - float(want)) <= 1e-4 * (abs(float(want)) + 1.0)
    # edge dropout keeps the expectation: the mean over draws approaches the eval-mode output
    layer2 = MGDCF(1, 0.0, 1.0, 0.0, 0.5, 0.0).train()
    acc = torch.zeros_like(x)
    for s in range(64):
        torch.manual_seed(100 + s)
        acc += layer2(g, x)
    ref = layer2.eval()(g, x)
    assert float((acc / 64 - ref).abs().mean()) < 0.2 * float(ref.abs().mean())


def test_item_knn_graph_vs_dense_oracle():
    rng = np.random.default_rng(9)
    feat = rng.standard_normal((700, 384)).astype(np.float32)
    from recommend_amd.propagation import build_item_graph
    g = build_item_graph(torch.from_numpy(feat).to(DEV), 10, "ii_weight")
    src, dst, w = graph_oracle.build_item_graph(feat, 10)
    np.testing.assert_array_equal(g.src, src)
    np.testing.assert_array_equal(g.dst, dst)
    np.testing.assert_allclose(g.edata["ii_weight"].cpu().numpy(), w, rtol=2e-5)
    assert g.num_edges() == 700 * 10
    # rows sum to one; the graph feeds the item-item layer
    x = torch.randn(700, 64, device=DEV)
    out = II_GCN(2, 0.1, 0.9, 0.0, 0.0, 0.0, "ii_weight").eval()(x, g)
    _close(out.cpu().numpy(), graph_oracle.ii_gcn_forward(src, dst, w.astype(np.float32), x.cpu().numpy(), 2, 0.1, 0.9))


@pytest.mark.parametrize("n_a,n_b,F,K", [(700, 700, 384, 10), (300, 1000, 4096, 20), (65, 33, 64, 32),
                                         (1, 129, 128, 4), (1000, 257, 200, 5), (513, 4100, 192, 12)])
def test_score_topk_wide_vs_float64(n_a, n_b, F, K):
    """rk_score_topk_wide_f32 (the similarity + top-K of the item kNN graph, feature width as a loop
    inside the kernel) against a float64 score matrix: values to 1e-5 of |a||b|, ids equal wherever
    the float64 scores around the list are further apart than that."""
    from recommend_amd import ops
    rng = np.random.default_rng(n_a + n_b + F)
    A = rng.standard_normal((n_a, F)).astype(np.float32)
    B = rng.standard_normal((n_b, F)).astype(np.float32)
    vals, ids = ops.score_topk_wide(torch.from_numpy(A).to(DEV), torch.from_numpy(B).to(DEV), K)
    vals, ids = vals.cpu().numpy(), ids.cpu().numpy().astype(np.int64)
    S = A.astype(np.float64) @ B.astype(np.float64).T
    scale = np.linalg.norm(A, axis=1, keepdims=True) * np.linalg.norm(B, axis=1).max()
    order = np.argsort(-S, axis=1, kind="stable")
    want_ids = order[:, :K]
    want = np.take_along_axis(S, want_ids, 1)
    assert (ids >= 0).all() and (ids < n_b).all()
    got64 = np.take_along_axis(S, ids, 1)
    assert (np.abs(vals - got64) <= 1e-5 * scale).all()
    assert (np.diff(vals, axis=1) <= 0).all()
    assert (np.abs(got64 - want) <= 2e-5 * scale).all()           # the same scores, place by place
    # a place is clear when its float64 score is further than the tolerance from both neighbours
    # in the float64 ranking (the (K+1)-th included)
    ext = np.take_along_axis(S, order[:, :min(K + 1, n_b)], 1)
    gap = np.abs(np.diff(ext, axis=1)) > 4e-5 * scale
    gap = np.concatenate([np.ones((n_a, 1), bool), gap, np.ones((n_a, 1), bool)], axis=1)
    clear = gap[:, :K] & gap[:, 1:K + 1]
    assert clear.mean() > 0.5
    np.testing.assert_array_equal(ids[clear], want_ids[clear])


def test_score_topk_wide_ties_go_to_the_lower_id():
    """Small integer features make every product and sum exact in fp32, so equal scores are equal
    bits: the ranking must be numpy's stable descending order, duplicated rows included."""
    from recommend_amd import ops
    rng = np.random.default_rng(4)
    X = rng.integers(-3, 4, (600, 128)).astype(np.float32)
    X[100:200] = X[300:400]                       # duplicated rows: whole score columns tie
    vals, ids = ops.score_topk_wide(torch.from_numpy(X).to(DEV), torch.from_numpy(X).to(DEV), 15)
    S = X.astype(np.int64) @ X.astype(np.int64).T
    want = np.argsort(-S, axis=1, kind="stable")[:, :15]
    np.testing.assert_array_equal(ids.cpu().numpy(), want)
    np.testing.assert_array_equal(vals.cpu().numpy(), np.take_along_axis(S, want, 1).astype(np.float32))
    from recommend_amd import _ffi
    with pytest.raises(_ffi.RkError):
        ops.score_topk_wide(torch.from_numpy(X).to(DEV), torch.from_numpy(X).to(DEV), 33)


def test_sgrec_graph_methods_vs_oracle():
    """SGrecGraphOps (build_graph / norm_adj of model/sgrec/graph.py:88-135) feeding IU_GCN: edge order,
    degree normalisation and k hops against the oracle."""
    from recommend_amd.propagation import SGrecGraphOps

    class G(SGrecGraphOps):
        num_users, num_items, num_nodes, device, weight_cache = 120, 80, 200, DEV, "weight"

    rng = np.random.default_rng(2)
    inter = [(int(u), int(i), 1.0) for u, i in zip(rng.integers(0, 120, 900), rng.integers(0, 80, 900))]
    gm = G()
    g = gm.build_graph(inter)
    gm.creat_graph_weight()
    u = np.array([t[0] for t in inter]); it = np.array([t[1] for t in inter]) + 120
    loops = np.arange(200)
    src = np.concatenate([u, it, loops]); dst = np.concatenate([it, u, loops])
    np.testing.assert_array_equal(g.src, src)
    np.testing.assert_array_equal(g.dst, dst)
    w = graph_oracle.deg_normalize(src, dst, 200)
    np.testing.assert_allclose(g.edata["weight"].cpu().numpy(), w, rtol=1e-6)
    x = torch.randn(200, 64, device=DEV)
    out = IU_GCN(0.0, 0.0, 0.0, "weight").eval()(x, g, 2)
    _close(out.cpu().numpy(), graph_oracle.iu_gcn_forward(src, dst, w.astype(np.float32), x.cpu().numpy(), 2))
