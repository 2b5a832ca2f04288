"""GPU parity: graph normalisation and propagation kernels against the oracle,
through the C ABI (recommend_amd.ops -> librecommend_amd.so)."""
import numpy as np
import pytest
import torch

from oracle import graph_oracle
from recommend_amd import ops, synth
from recommend_amd.graph_constructor import GraphConstructor

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def random_graph(U, I, E, seed, hub=0, self_loops=True):
    rng = np.random.default_rng(seed)
    users, items = rng.integers(0, U, E), rng.integers(0, I, E)
    if hub:  # one very popular item and one very active user -> long rows
        users = np.concatenate([users, rng.integers(0, U, hub), np.zeros(hub, np.int64)])
        items = np.concatenate([items, np.zeros(hub, np.int64), rng.integers(0, I, hub)])
    ratings = rng.integers(1, 6, users.shape[0]).astype(np.float32)
    return users, items, ratings


def build(users, items, ratings, U, I, self_loops=True, normalize=True):
    gc = GraphConstructor(add_self_loops=self_loops, normalize_adj=normalize)
    gc.build_graph((users, items, ratings), U, I)
    gc.move_to_device(DEV)
    return gc


@pytest.mark.parametrize("U,I,E,hub", [(40, 30, 300, 0), (300, 200, 3000, 900), (1, 1, 1, 0)])
def test_sym_normalize_bit_exact(U, I, E, hub):
    users, items, ratings = random_graph(U, I, E, 1, hub)
    gc = build(users, items, ratings, U, I)
    gc.creat_feature_weight(None)
    src, dst, w = graph_oracle.build_coo_vectorized(users, items, ratings, U, I, True)
    want, in_w = graph_oracle.normalize(src, dst, w, U + I)
    got = gc.graph.edata["weight"].cpu().numpy()
    np.testing.assert_array_equal(got, want)
    stats = gc.get_graph_statistics()
    assert stats["num_edges"] == src.shape[0] and stats["num_nodes"] == U + I
    assert abs(stats["edge_weight_stats"]["max"] - float(want.max())) < 1e-7


def test_normalize_isolated_nodes_without_self_loops():
    # users 3,4 and item 2 have no edges: in_w clamps to 1e-12, no NaN/inf appears
    users, items, ratings = np.array([0, 1, 2]), np.array([0, 1, 0]), np.array([5, 3, 1], np.float32)
    gc = build(users, items, ratings, 5, 3, self_loops=False)
    gc.creat_feature_weight(None)
    src, dst, w = graph_oracle.build_coo_vectorized(users, items, ratings, 5, 3, False)
    want, _ = graph_oracle.normalize(src, dst, w, 8)
    np.testing.assert_array_equal(gc.graph.edata["weight"].cpu().numpy(), want)
    h = torch.randn(8, 64, device=DEV)
    out = gc.propagate(h).cpu().numpy()
    np.testing.assert_allclose(out, graph_oracle.spmm(src, dst, want, h.cpu().numpy()), rtol=1e-5, atol=1e-6)
    assert np.all(out[[3, 4, 7]] == 0)


@pytest.mark.parametrize("D", [16, 32, 64, 128, 256])
@pytest.mark.parametrize("hub", [0, 1500, 6000])
def test_spmm_forward_backward_vs_oracle(D, hub):
    U, I, E = 500, 300, 4000
    users, items, ratings = random_graph(U, I, E, D + hub, hub)
    gc = build(users, items, ratings, U, I)
    gc.creat_feature_weight(None)
    src, dst, w = graph_oracle.build_coo_vectorized(users, items, ratings, U, I, True)
    wn, _ = graph_oracle.normalize(src, dst, w, U + I)
    rng = np.random.default_rng(0)
    h = rng.standard_normal((U + I, D)).astype(np.float32)
    g = rng.standard_normal((U + I, D)).astype(np.float32)
    ht = torch.from_numpy(h).to(DEV).requires_grad_(True)
    out = gc.propagate(ht)
    out.backward(torch.from_numpy(g).to(DEV))
    ref = graph_oracle.spmm_f64(src, dst, wn, h)
    ref_t = graph_oracle.spmm_f64(dst, src, wn, g)          # transpose: swap the edge ends
    scale = np.abs(ref).max()
    assert np.abs(out.detach().cpu().numpy() - ref).max() <= 1e-5 * scale
    assert np.abs(ht.grad.cpu().numpy() - ref_t).max() <= 1e-5 * np.abs(ref_t).max()
    # same order of magnitude of agreement with the fp32 oracle
    np.testing.assert_allclose(out.detach().cpu().numpy(), graph_oracle.spmm(src, dst, wn, h), rtol=1e-4, atol=1e-5 * scale)
    if hub:
        assert gc.graph.fwd.n_long >= 1 and gc.graph.bwd.n_long >= 1


def test_spmm_repeated_launches_on_one_operand_with_changing_inputs():
    """Rows cut into several chunks hand their partial sums from workgroup to workgroup inside
    the launch, through a buffer and counters that live with the operand.  Launch after launch
    on the SAME operand with DIFFERENT inputs (each launch first reads the partial buffer's
    old contents into the caches through plain loads) must give that launch's result: the hub
    rows against float64, the whole output bit for bit against a second operand object that has
    never been used before."""
    U, I, E, hub = 500, 300, 4000, 9000
    users, items, ratings = random_graph(U, I, E, 77, hub)
    gc = build(users, items, ratings, U, I)
    gc.creat_feature_weight(None)
    gc.graph.refresh_values("weight")
    fwd = gc.graph.fwd
    assert fwd.n_multi >= 1
    src, dst, w = graph_oracle.build_coo_vectorized(users, items, ratings, U, I, True)
    wn, _ = graph_oracle.normalize(src, dst, w, U + I)
    rng = np.random.default_rng(5)
    for it in range(12):
        h = (rng.standard_normal((U + I, 64)) * (it + 1)).astype(np.float32)
        ht = torch.from_numpy(h).to(DEV)
        part, _ = fwd._partial.get(64, (None, None))
        if part is not None:
            float(part.sum())                      # plain loads of every partial line, old contents
        out = ops.spmm_raw(fwd, ht).cpu().numpy()
        ref = graph_oracle.spmm_f64(src, dst, wn, h)
        assert np.abs(out - ref).max() <= 1e-5 * np.abs(ref).max(), it
        fresh = ops.CsrMatrix(gc.graph.csr_host[0], gc.graph.csr_host[1], gc.graph.csr_host[2], DEV)
        fresh.set_values_from_coo(gc.graph.edata["weight"])
        np.testing.assert_array_equal(out, ops.spmm_raw(fresh, ht).cpu().numpy())
    # the counters are back at zero after every launch
    assert int(fwd._partial[64][1].abs().sum()) == 0


def test_aggrate_and_agg_func_vs_oracle():
    U, I, E, D = 200, 150, 2500, 64
    users, items, ratings = random_graph(U, I, E, 5, 400)
    gc = build(users, items, ratings, U, I)
    gc.creat_feature_weight(None)
    src, dst, w = graph_oracle.build_coo_vectorized(users, items, ratings, U, I, True)
    wn, _ = graph_oracle.normalize(src, dst, w, U + I)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((U + I, D)).astype(np.float32)
    xt = torch.from_numpy(x).to(DEV).requires_grad_(True)
    got = gc.aggrate(xt, 3, 0.15)
    want = graph_oracle.aggrate(src, dst, wn, x, 3, 0.15)
    np.testing.assert_allclose(got.detach().cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    # gradient of sum(out * G) w.r.t. x equals the transposed recursion applied to G
    G = rng.standard_normal((U + I, D)).astype(np.float32)
    got.backward(torch.from_numpy(G).to(DEV))
    # d/dx of h_k: h_k = a x + (1-a) A h_{k-1}, h_0 = x  =>  grad = sum over paths
    import scipy.sparse as sp
    A = sp.coo_matrix((wn.astype(np.float64), (dst, src)), shape=(U + I, U + I)).tocsr()
    a, acc, cur = 0.15, np.zeros_like(G, dtype=np.float64), G.astype(np.float64)
    for _ in range(3):
        acc += a * cur
        cur = (1 - a) * (A.T @ cur)
    acc += cur
    np.testing.assert_allclose(xt.grad.cpu().numpy(), acc, rtol=1e-4, atol=1e-5)

    lin = torch.nn.Linear(D, D).to(DEV)
    act = torch.nn.PReLU().to(DEV)
    got2 = gc.agg_func(torch.from_numpy(x).to(DEV), 2, lin, torch.nn.Dropout(0.0), act)
    Wn, bn, al = lin.weight.detach().cpu().numpy(), lin.bias.detach().cpu().numpy(), float(act.weight)
    def f(hh):
        y = hh @ Wn.T + bn
        return np.where(y > 0, y, al * y).astype(np.float32)
    want2 = graph_oracle.agg_func(src, dst, wn, x, 2, f)
    np.testing.assert_allclose(got2.detach().cpu().numpy(), want2, rtol=1e-4, atol=1e-5)


def test_deg_normalize_vs_oracle():
    U, I, E = 120, 80, 900
    users, items, ratings = random_graph(U, I, E, 9)
    gc = build(users, items, None, U, I)
    got = ops.deg_normalize(gc.graph.fwd).cpu().numpy()
    src, dst, _ = graph_oracle.build_coo_vectorized(users, items, np.ones(E, np.float32), U, I, True)
    np.testing.assert_allclose(got, graph_oracle.deg_normalize(src, dst, U + I), rtol=1e-6)


def test_baby_shape_properties_and_f64_reference():
    """Full Baby-shape graph: fp64 scipy reference, linearity, and row sums."""
    d = synth.make_named("baby")
    u, i, r = d.split(0)
    gc = build(u, i, r, d.n_users, d.n_items)
    gc.creat_feature_weight(None)
    N, D = d.n_users + d.n_items, 64
    src, dst, w = graph_oracle.build_coo_vectorized(u, i, r, d.n_users, d.n_items, True)
    wn, _ = graph_oracle.normalize(src, dst, w, N)
    np.testing.assert_array_equal(gc.graph.edata["weight"].cpu().numpy(), wn)
    # CSR contract on the device copy
    p, idx, eid = graph_oracle.csr_from_coo(dst, src, N)
    np.testing.assert_array_equal(gc.graph.fwd.rowptr.cpu().numpy(), p)
    np.testing.assert_array_equal(gc.graph.fwd.idx.cpu().numpy(), idx)
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(N, D, generator=g)
    y = torch.randn(N, D, generator=g)
    ax = gc.propagate(x.to(DEV))
    np.testing.assert_array_equal(gc.graph.fwd.val.cpu().numpy(), wn[eid])
    ay = gc.propagate(y.to(DEV))
    ref = graph_oracle.spmm_f64(src, dst, wn, x.numpy())
    assert np.abs(ax.cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max()
    axy = gc.propagate((2.0 * x - 0.5 * y).to(DEV))
    torch.testing.assert_close(axy, 2.0 * ax - 0.5 * ay, rtol=1e-4, atol=1e-5)
    ones = gc.propagate(torch.ones(N, D, device=DEV))[:, 0].cpu().numpy()
    rowsum = np.zeros(N)
    np.add.at(rowsum, dst, wn.astype(np.float64))
    np.testing.assert_allclose(ones, rowsum, rtol=1e-5)


def test_spmm_argument_errors():
    users, items, ratings = random_graph(10, 10, 30, 0)
    gc = build(users, items, ratings, 10, 10)
    from recommend_amd._ffi import RkError
    with pytest.raises(RkError, match="not built"):
        gc.propagate(torch.randn(20, 20, device=DEV))
    with pytest.raises(RkError):
        gc.propagate(torch.randn(20, 64))          # CPU tensor: no fallback


@pytest.mark.parametrize("kind", ["cosine", "dot"])
@pytest.mark.parametrize("F", [384, 4096, 30])
def test_feature_similarity_edge_weights_vs_float64(kind, F):
    """rk_edge_feature_sim_f32 (one wavefront per edge) against the float64 restatement of
    graph_constructor.py:137-156, on an edge list with both endpoints inside, one endpoint inside
    and no endpoint inside the node range that carries features; then the same through
    GraphConstructor.creat_feature_weight on a bipartite graph with item-side features."""
    import torch
    from recommend_amd import ops
    rng = np.random.default_rng(F)
    n_nodes, row_lo, n_rows, nnz = 500, 120, 300, 20000
    x = (rng.standard_normal((n_rows, F)) * (0.2 if kind == "dot" else 1.0)).astype(np.float32)
    x[7] = 0.0                                           # an all-zero feature row inside the range
    src = rng.integers(0, n_nodes, nnz).astype(np.int32)
    dst = rng.integers(0, n_nodes, nnz).astype(np.int32)
    dst[:200] = src[:200]                                # self loops: cosine exactly 1 up to rounding
    padded = np.zeros((n_nodes, F), np.float32)
    padded[row_lo:row_lo + n_rows] = x
    want = graph_oracle.edge_feature_weights(src, dst, padded, kind)
    got = ops.edge_feature_sim(torch.from_numpy(src).to(DEV), torch.from_numpy(dst).to(DEV),
                               torch.from_numpy(x).to(DEV), row_lo, kind).cpu().numpy()
    assert np.abs(got - want).max() <= 2e-6
    again = ops.edge_feature_sim(torch.from_numpy(src).to(DEV), torch.from_numpy(dst).to(DEV),
                                 torch.from_numpy(x).to(DEV), row_lo, kind).cpu().numpy()
    assert np.array_equal(got, again)                    # no atomics: bitwise reproducible

    if F == 384:
        U, I = 90, 60
        d = synth.make_interactions(U, I, 700, seed=3)
        tu, ti, tr = d.split(0)
        feat = torch.from_numpy(rng.standard_normal((I, F)).astype(np.float32))
        gc = GraphConstructor(item_features={"text": feat}, edge_weight_type=kind, normalize_adj=False)
        gc.build_graph((tu, ti, tr), U, I)
        gc.creat_feature_weight("text")
        g = gc.graph
        pad = np.zeros((U + I, F), np.float32)
        pad[U:] = feat.numpy()
        want = graph_oracle.edge_feature_weights(g.src, g.dst, pad, kind)
        assert np.abs(g.edata["text_weight"].cpu().numpy() - want).max() <= 2e-6
