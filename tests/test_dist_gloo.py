"""World-size-2 check of the item-sharded evaluation path on CPU (gloo).

The partitioning, the all-gather exchange, the per-slice merge and the metric
all-reduce are host logic; the three compute calls are injected.  Here they are
backed by the oracle (test-only stand-in for the HIP ops), so the test pins the
distributed plumbing: the 2-rank result must equal the unsharded oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
METRICS = ["precision", "recall", "hit_rate", "ndcg", "map", "mrr"]


class OracleCompute:
    """score_topk / topk_merge / rank_metrics with the signatures of recommend_amd.ops."""

    @staticmethod
    def score_topk(U, I, K, mask_ptr=None, mask_items=None, item_id_offset=0):
        from oracle import eval_oracle
        mp_ = None if mask_ptr is None else mask_ptr.numpy()
        mi_ = None if mask_items is None else mask_items.numpy()
        if mi_ is not None and mi_.size == 0:
            mp_ = mi_ = None
        s, i = eval_oracle.score_topk(U.numpy(), I.numpy(), K, mp_, mi_, id_offset=item_id_offset)
        return torch.from_numpy(s), torch.from_numpy(i)

    @staticmethod
    def topk_merge(scores, ids, K):
        U, P, Kin = scores.shape
        s = scores.reshape(U, P * Kin).numpy()
        i = ids.reshape(U, P * Kin).numpy().astype(np.int64)
        key_i = np.where(i < 0, np.iinfo(np.int64).max, i)
        order = np.lexsort((key_i, -s.astype(np.float64), i < 0), axis=1)[:, :K]
        return (torch.from_numpy(np.take_along_axis(s, order, 1)),
                torch.from_numpy(np.take_along_axis(i, order, 1).astype(np.int32)))

    @staticmethod
    def rank_metrics(topk_ids, tgt_ptr, tgt_items, k_list, mean_denominator=None, return_sums=False):
        from oracle import eval_oracle
        n = topk_ids.shape[0]
        d = eval_oracle.metrics_from_topk_sparse(topk_ids.numpy(), tgt_ptr.numpy(), tgt_items.numpy(),
                                                 k_list, n_users_total=1)   # sums
        sums = torch.tensor([[d[k][m] for m in METRICS] for k in k_list], dtype=torch.float64)
        out = (sums / float(mean_denominator or n)).float()
        return (out, sums) if return_sums else out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from recommend_amd.dist import ShardedRankingEvaluator, item_range
        rng = np.random.default_rng(0)
        U, I, D = 150, 333, 64
        ue = (np.round(rng.standard_normal((U, D)) * 2) / 2).astype(np.float32)   # ties on purpose
        ie = (np.round(rng.standard_normal((I, D)) * 2) / 2).astype(np.float32)
        tr = np.unique(np.stack([rng.integers(0, U, 900), rng.integers(0, I, 900)], 1), axis=0)
        va = np.unique(np.stack([rng.integers(0, U, 300), rng.integers(0, I, 300)], 1), axis=0)
        ev = ShardedRankingEvaluator((va[:, 0], va[:, 1]), (tr[:, 0], tr[:, 1]), [5, 10, 20], METRICS,
                                     U, I, compute=OracleCompute)
        assert (ev.lo, ev.hi) == item_range(I, rank, world)
        res = ev.evaluate(torch.from_numpy(ue), torch.from_numpy(ie))
        ids = ev.last_ids.numpy()
        # the user-sharded form: same users per rank, no exchange of lists
        from recommend_amd.dist import UserShardedRankingEvaluator
        ev2 = UserShardedRankingEvaluator((va[:, 0], va[:, 1]), (tr[:, 0], tr[:, 1]), [5, 10, 20], METRICS,
                                          U, I, compute=OracleCompute)
        res2 = ev2.evaluate(torch.from_numpy(ue), torch.from_numpy(ie))
        assert (ev2.u_lo, ev2.u_hi) == (ev.u_lo, ev.u_hi)
        assert np.array_equal(ev2.last_ids.numpy(), ids)
        assert {k: float(v) for k, v in res2.items()} == {k: float(v) for k, v in res.items()}
        q.put((rank, {k: float(v) for k, v in res.items()}, ev.u_lo, ev.u_hi, ids))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_item_sharded_eval_world2_equals_unsharded():
    from oracle import eval_oracle
    from recommend_amd.evaluator import pairs_to_csr
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # unsharded oracle on the same inputs
    rng = np.random.default_rng(0)
    U, I, D = 150, 333, 64
    ue = (np.round(rng.standard_normal((U, D)) * 2) / 2).astype(np.float32)
    ie = (np.round(rng.standard_normal((I, D)) * 2) / 2).astype(np.float32)
    tr = np.unique(np.stack([rng.integers(0, U, 900), rng.integers(0, I, 900)], 1), axis=0)
    va = np.unique(np.stack([rng.integers(0, U, 300), rng.integers(0, I, 300)], 1), axis=0)
    mp_, mi_ = pairs_to_csr(tr[:, 0], tr[:, 1], U)
    _s, want_ids = eval_oracle.score_topk(ue, ie, 20, mp_, mi_)
    tp, tc = pairs_to_csr(va[:, 0], va[:, 1], U)
    want = eval_oracle.metrics_from_topk_sparse(want_ids, tp, tc, [5, 10, 20])
    for rank, res, u_lo, u_hi, ids in results:
        np.testing.assert_array_equal(ids, want_ids[u_lo:u_hi])          # merged ids exact, ties included
        for k in (5, 10, 20):
            for m in METRICS:
                assert abs(res[f"{m}@{k}"] - want[k][m]) <= 1e-6, (rank, m, k)
    assert results[0][1] == results[1][1]                                  # every rank holds the same metrics


# ------------------------------------------------------------------ row-sharded propagation
class OracleRowBlock:
    """matmul(h_full) of one row block in float64 scipy (test-only stand-in for HipRowBlock)."""

    def __init__(self, rowptr, col, val, device, n_cols=None):
        import scipy.sparse as sp
        self.n_cols = n_cols if n_cols is not None else (int(col.max()) + 1 if col.size else 1)
        self.m = sp.csr_matrix((np.asarray(val, np.float64), col, rowptr), shape=(rowptr.shape[0] - 1, self.n_cols))

    def remap_columns(self, col, n_cols):
        import scipy.sparse as sp
        m = self.m.tocsr()
        self.n_cols = int(n_cols)
        self.m = sp.csr_matrix((m.data, np.asarray(col), m.indptr), shape=(m.shape[0], self.n_cols))

    def matmul(self, h_full, h0=None, alpha=0.0, beta=1.0):
        h = h_full.detach().numpy().astype(np.float64)[: self.n_cols]
        out = beta * (self.m @ h)
        if h0 is not None:
            out = out + alpha * h0.detach().numpy().astype(np.float64)
        return torch.from_numpy(out.astype(np.float32))


def _prop_case():
    rng = np.random.default_rng(3)
    N, E, D = 101, 700, 16                     # 101 rows over 2 ranks: blocks of 51 and 50
    src = rng.integers(0, N, E)
    dst = rng.integers(0, N, E)
    src[:5], dst[:5] = N - 1, np.arange(5)     # the last node has edges both ways
    dst[5:10], src[5:10] = N - 1, np.arange(5)
    w = rng.random(E).astype(np.float32)       # NOT symmetric: the backward needs the true transpose
    x = rng.standard_normal((N, D)).astype(np.float32)
    g = rng.standard_normal((N, D)).astype(np.float32)
    return N, src, dst, w, x, g


def _prop_worker(rank, world, port, q, exchange="allgather"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from recommend_amd.dist import RowShardedGraph, node_range
        N, src, dst, w, x, g = _prop_case()
        graph = RowShardedGraph(src, dst, w, N, device="cpu", block_factory=OracleRowBlock, exchange=exchange)
        assert (graph.lo, graph.hi) == node_range(N, rank, world)
        if exchange == "halo":
            # exactly the distinct columns outside the rank's block, and no N-row buffer
            for which, rows, cols in (("fwd", dst, src), ("bwd", src, dst)):
                mine = (rows >= graph.lo) & (rows < graph.hi)
                c = cols[mine]
                want = np.unique(c[(c < graph.lo) | (c >= graph.hi)]).shape[0]
                assert graph._plans[which].n_halo == want, (which, graph._plans[which].n_halo, want)
            assert graph.halo_rows() == (graph._plans["fwd"].n_halo, graph._plans["bwd"].n_halo)
            assert not graph._gather_buf
        xl = torch.from_numpy(x[graph.lo:graph.hi]).requires_grad_(True)
        out = graph.aggrate(xl, 3, 0.2)
        out.backward(torch.from_numpy(g[graph.lo:graph.hi]))
        q.put((rank, graph.lo, graph.hi, out.detach().numpy(), xl.grad.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["allgather", "halo"])
def test_row_sharded_propagation_world2_equals_unsharded(exchange):
    import scipy.sparse as sp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_prop_worker, args=(r, 2, port, q, exchange)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    N, src, dst, w, x, g = _prop_case()
    A = sp.csr_matrix((w.astype(np.float64), (dst, src)), shape=(N, N))   # duplicates add up
    h0 = x.astype(np.float64)
    h = h0
    for _ in range(3):
        h = 0.2 * h0 + 0.8 * (A @ h)
    # gradient of sum(out * g) wrt x through the same recurrence
    gh = g.astype(np.float64)
    gx = np.zeros_like(gh)
    for _ in range(3):
        gx += 0.2 * gh
        gh = 0.8 * (A.T @ gh)
    gx += gh
    assert sorted((lo, hi) for _r, lo, hi, _o, _g in results) == [(0, 51), (51, 101)]
    for _rank, lo, hi, out, grad in results:
        assert np.abs(out - h[lo:hi]).max() <= 1e-5 * np.abs(h).max()
        assert np.abs(grad - gx[lo:hi]).max() <= 1e-5 * np.abs(gx).max()
